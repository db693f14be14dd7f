#!/bin/bash
# PMC passes of the attention-backward kernel alone (scripts/attn_bwd_only.py): bash scripts/attn_bwd_pmc.sh <lse|recompute> <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; k=$1; tag=$2; out=$R/gpurun_out/abw_pmc_$tag; mkdir -p $out
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $out/p1 -o run -- python3 $R/scripts/attn_bwd_only.py $k 6 > $out/p1.log 2>&1 || { echo p1 failed; tail -5 $out/p1.log; exit 1; }
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace -d $out/p2 -o run -- python3 $R/scripts/attn_bwd_only.py $k 6 > $out/p2.log 2>&1 || { echo p2 failed; tail -5 $out/p2.log; exit 1; }
python3 - <<PY
import csv, glob, collections
for p in ("p1", "p2"):
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob("$out/%s/**/*counter_collection.csv" % p, recursive=True):
        for r in csv.DictReader(open(f)):
            if "attn_bwd" not in r["Kernel_Name"]: continue
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k in sorted(acc): print(p, k, acc[k] / max(n[k], 1), n[k])
PY
find $out -name "*.csv" -size +1M -delete; find $out -name "*.db" -delete
