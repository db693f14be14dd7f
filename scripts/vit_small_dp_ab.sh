for r in 1 2 3; do
for dp in 0 1; do
PV_FULLROW_DP=$dp python bench.py --model vit_small --batch 512 --steps 30 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dp=$dp', d['value'], d['ms_per_step'])"
done; done
