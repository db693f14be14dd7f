#!/bin/bash
# In-model A/B of differently built fp16 libraries: swaps peekvit_amd/libpeekvit_hip_<variant>f16.so into the place of the default library and runs
# bench.py once per variant, interleaved ROUNDS times ("base" = the default build).   scripts/lib_ab.sh ROUNDS variant [variant ...] [-- bench args]
# (GPU box only: the swap happens in the box's scratch copy of the repo)
set -u
cd "$(dirname "$0")/.." || exit 1
rounds=$1; shift
vars=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do vars+=("$1"); shift; done
[ $# -gt 0 ] && shift
args="--steps 10 --warmup 3 --no-cpu-baseline $*"
cp peekvit_amd/libpeekvit_hip_f16.so /tmp/pv_base_f16.so
for r in $(seq 1 "$rounds"); do
  for v in base "${vars[@]}"; do
    if [ "$v" = base ]; then cp /tmp/pv_base_f16.so peekvit_amd/libpeekvit_hip_f16.so; else cp "peekvit_amd/libpeekvit_hip_${v}f16.so" peekvit_amd/libpeekvit_hip_f16.so; fi
    python3 bench.py $args 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['roofline']['frac'])"
  done
done
cp /tmp/pv_base_f16.so peekvit_amd/libpeekvit_hip_f16.so
