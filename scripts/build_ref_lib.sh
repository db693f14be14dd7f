#!/bin/bash
# Build the fp16-operand library from the kernel sources of ANOTHER git revision as peekvit_amd/libpeekvit_hip_<tag>f16.so, for same-box A/B runs
# with scripts/lib_ab.sh (run here: hipcc cross-compiles; the .so travels with the snapshot).   scripts/build_ref_lib.sh <git-rev> <tag>
set -eu
cd "$(dirname "$0")/.."
rev=$1; tag=$2
tmp=$(mktemp -d)
git archive "$rev" peekvit_amd/csrc include | tar -x -C "$tmp"
objs=()
for src in "$tmp"/peekvit_amd/csrc/*.hip; do
  b=$(basename "$src" .hip); extra=""
  [ "$b" = pv_attention ] && extra="-mllvm -amdgpu-mfma-vgpr-form=1"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $extra -DPV_OPERAND_F16 -I"$tmp/include" -c "$src" -o "$tmp/$b.o" &
  objs+=("$tmp/$b.o")
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "peekvit_amd/libpeekvit_hip_${tag}f16.so" "${objs[@]}"
rm -rf "$tmp"
echo "peekvit_amd/libpeekvit_hip_${tag}f16.so"
