#!/bin/bash
# Build the fp16-operand library from the kernel sources of ANOTHER git revision as peekvit_amd/libpeekvit_hip_<tag>f16.so, for same-box A/B runs
# with scripts/lib_ab.sh (run here: hipcc cross-compiles; the .so travels with the snapshot).   scripts/build_ref_lib.sh <git-rev> <tag>
set -eu
cd "$(dirname "$0")/.."
rev=$1; tag=$2
tmp=$(mktemp -d)
git archive "$rev" peekvit_amd/csrc peekvit_amd/_build.py include | tar -x -C "$tmp"
objs=()
for src in "$tmp"/peekvit_amd/csrc/*.hip; do
  b=$(basename "$src" .hip)
  # the flags THAT revision's libraries were built with (round-4 review: a hard-coded list here dropped pv_rowops' -fno-slp-vectorize)
  flags=$(python3 - "$tmp/peekvit_amd/_build.py" "$b.hip" <<'PY'
import importlib.util, sys
spec = importlib.util.spec_from_file_location("_build_at_rev", sys.argv[1])
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
print(" ".join(list(m.FLAGS) + list(getattr(m, "FILE_FLAGS", {}).get(sys.argv[2], []))))
PY
)
  /opt/rocm/bin/hipcc $flags -DPV_OPERAND_F16 -I"$tmp/include" -c "$src" -o "$tmp/$b.o" &
  objs+=("$tmp/$b.o")
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "peekvit_amd/libpeekvit_hip_${tag}f16.so" "${objs[@]}"
rm -rf "$tmp"
echo "peekvit_amd/libpeekvit_hip_${tag}f16.so"
