#!/bin/bash
# every bench line quoted in README / DESIGN, one after the other (one GPU):  scripts/bench_round.sh <tag>  -> gpurun_out/<tag>_bench_*.json
set -u
tag=${1:-r}
cd "$(dirname "$0")/.." || exit 1
o=gpurun_out; mkdir -p $o
python3 bench.py > $o/${tag}_bench_vit_b_16.json 2> $o/${tag}_bench.err &&
python3 bench.py --train --no-cpu-baseline > $o/${tag}_bench_train_vit_b_16.json 2>> $o/${tag}_bench.err &&
python3 bench.py --rank-budget 0.5 --no-cpu-baseline > $o/${tag}_bench_rankvit.json 2>> $o/${tag}_bench.err &&
python3 bench.py --rank-budget 0.5 --train --no-cpu-baseline > $o/${tag}_bench_train_rankvit.json 2>> $o/${tag}_bench.err &&
python3 bench.py --model vit_small --batch 512 --steps 20 --warmup 5 --no-cpu-baseline > $o/${tag}_bench_vit_small.json 2>> $o/${tag}_bench.err &&
python3 bench.py --model vit_small --batch 512 --steps 20 --warmup 5 --train --no-cpu-baseline > $o/${tag}_bench_train_vit_small.json 2>> $o/${tag}_bench.err &&
python3 bench.py --model vit_tiny --batch 32 --steps 200 --warmup 100 --no-cpu-baseline > $o/${tag}_bench_vit_tiny.json 2>> $o/${tag}_bench.err
rc=$?
for f in $o/${tag}_bench_*.json; do python3 -c "
import json,sys
d=json.load(open('$f')); print('$f'.split('/')[-1], d['value'], d['unit'], d['ms_per_step'], 'ms', d['dtype'], d.get('all_rows_mode',{}).get('value'), d['config'].get('gflop_per_image_executed'))"; done
exit $rc
