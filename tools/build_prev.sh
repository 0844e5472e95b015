#!/bin/bash
# Build opa-dpo_amd/lib/libopadpo_hip_<name>.so from the csrc/ of another git revision (default: HEAD) for same-box A/B runs with
# tools/ab_lib.sh (whole step) and tools/ab_gemm.sh (sustained per-shape GEMM rates); OPADPO_LIB_PATH selects the library at load time.
#   tools/build_prev.sh [rev] [name] [extra hipcc flags...]      e.g. tools/build_prev.sh HEAD~1 prev
set -e
REV=${1:-HEAD}; NAME=${2:-prev}; shift 2 2>/dev/null || true
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
mkdir -p $T/a/csrc $T/include      # kernels.h includes ../../include/opadpo_hip.h
for f in $(git -C $R ls-tree --name-only $REV opa-dpo_amd/csrc/); do git -C $R show $REV:$f > $T/a/csrc/$(basename $f); done
git -C $R show $REV:include/opadpo_hip.h > $T/include/opadpo_hip.h
OBJS=""
for s in gemm attention elementwise head_optim decode capi ctx; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result "$@" -c $T/a/csrc/$s.hip -o $T/$s.o &
  OBJS="$OBJS $T/$s.o"
done
wait
for o in $OBJS; do [ -f $o ] || { echo "compile failed: $o"; exit 1; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/opa-dpo_amd/lib/libopadpo_hip_$NAME.so $OBJS
rm -rf $T
echo "built opa-dpo_amd/lib/libopadpo_hip_$NAME.so from $REV"
