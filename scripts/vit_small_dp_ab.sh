#!/bin/bash
# In-model A/B of the full-row GEMM variants on vit_small (batch 512): ENVVAR=0 / 1 interleaved, three rounds.   scripts/vit_small_dp_ab.sh [ENVVAR]
v=${1:-PV_FULLROW_DP}
for r in 1 2 3; do
for x in 0 1; do
env $v=$x python bench.py --model vit_small --batch 512 --steps 30 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v=$x', d['value'], d['ms_per_step'])"
done; done
