#!/bin/bash
# Experiment builds of the generated K-loop (csrc/w4_kloop_gen.py, W4K_EXP variants): tools/build_kloop_exp.sh noadv other ...  ->  opa-dpo_amd/lib/libopadpo_hip_<name>.so
# (timing experiments; the shipped library is always built from the committed w4_kloop.inc)
R=$(cd "$(dirname "$0")/.." && pwd)
python $R/opa-dpo_amd/build.py > /dev/null || exit 1
for n in "$@"; do
  ( W4K_EXP=${n//+/,} W4K_OUT=/tmp/w4_kloop_$n.inc python $R/opa-dpo_amd/csrc/w4_kloop_gen.py > /dev/null || { echo "generator failed: $n"; exit 1; }
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result "-DW4K_INC=\"/tmp/w4_kloop_$n.inc\"" -c $R/opa-dpo_amd/csrc/gemm.hip -o /tmp/gemm_k_$n.o 2>/tmp/gemm_k_$n.log || { echo "compile failed: $n"; tail -5 /tmp/gemm_k_$n.log; exit 1; }
    OBJS=$(ls $R/opa-dpo_amd/build/*.o | grep -v gemm.hip.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/opa-dpo_amd/lib/libopadpo_hip_$n.so /tmp/gemm_k_$n.o $OBJS && echo "built libopadpo_hip_$n.so" ) &
done
wait
