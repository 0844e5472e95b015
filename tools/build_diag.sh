#!/bin/bash
# Experiment / diagnostic builds of the library: gemm.hip compiled with extra -D flags, everything else from opa-dpo_amd/build/*.o
#   tools/build_diag.sh name1:"-DFLAG=1 -DOTHER=2" name2:"..."      ->  opa-dpo_amd/lib/libopadpo_hip_<name>.so
#   OPADPO_W4S_DIAG=1:    per-tile cycle accounting of the streaming 256x256 GEMM (K-loop / epilogue / set-up; read with tools/w4s_diag.py)
#   (the K-loop itself is generated text since round 5: schedule experiments go through tools/micro/kloop_bisect_gen.py, not through -D flags)
# then   tools/ab_gemm.sh new name1 name2 ...   /   tools/pmc_diag.sh new name1 ...
R=$(cd "$(dirname "$0")/.." && pwd)
python $R/opa-dpo_amd/build.py > /dev/null || exit 1
for spec in "$@"; do
  n=${spec%%:*}; f=${spec#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result $f -c $R/opa-dpo_amd/csrc/${DIAG_SRC:-gemm}.hip -o /tmp/gemm_$n.o 2>/dev/null || { echo "compile failed: $n"; exit 1; }
    OBJS=$(ls $R/opa-dpo_amd/build/*.o | grep -v ${DIAG_SRC:-gemm}.hip.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/opa-dpo_amd/lib/libopadpo_hip_$n.so /tmp/gemm_$n.o $OBJS && echo "built libopadpo_hip_$n.so ($f)" ) &
done
wait
