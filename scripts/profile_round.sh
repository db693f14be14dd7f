#!/bin/bash
# rocprofv3 evidence for one bench.py command: kernel-trace stats + three separate PMC passes (never combined with a trace domain).
#   scripts/profile_round.sh <tag> <bench.py args...>        e.g.  scripts/profile_round.sh r02 --steps 5 --warmup 2
# Writes gpurun_out/<tag>_{prof,pmc_sq,pmc_fetch,pmc_write}/ and gpurun_out/<tag>_kernel_summary.json (scripts/summarize_profiles.py),
# plus gpurun_out/<tag>_kernel_stats.csv (the --stats table).  Copy what should be judged into profiles/.
set -u
tag=$1; shift
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out
mkdir -p $out
args="--no-cpu-baseline $*"
rocprofv3 --output-format csv --kernel-trace --stats -d $out/${tag}_prof -o run -- python3 bench.py $args > $out/${tag}_prof.log 2>&1 || { echo "kernel-trace run failed"; tail -5 $out/${tag}_prof.log; exit 1; }
rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $out/${tag}_pmc_sq -o run -- python3 bench.py $args > $out/${tag}_pmc_sq.log 2>&1 || { echo "pmc sq run failed"; exit 1; }
rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $out/${tag}_pmc_fetch -o run -- python3 bench.py $args > $out/${tag}_pmc_fetch.log 2>&1 || { echo "pmc fetch run failed"; exit 1; }
rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d $out/${tag}_pmc_write -o run -- python3 bench.py $args > $out/${tag}_pmc_write.log 2>&1 || { echo "pmc write run failed"; exit 1; }
python3 scripts/summarize_profiles.py $tag > $out/${tag}_kernel_summary.json
f=$(find $out/${tag}_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $out/${tag}_kernel_stats.csv
# keep the merged output small: the raw traces are tens of MB
find $out/${tag}_prof $out/${tag}_pmc_sq $out/${tag}_pmc_fetch $out/${tag}_pmc_write \( -name "*kernel_trace.csv" -o -name "*counter_collection.csv" -o -name "*.db" \) -delete 2>/dev/null
echo "summary: $out/${tag}_kernel_summary.json"
