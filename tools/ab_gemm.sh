#!/bin/bash
# same-box, sustained-power comparison of several builds of the library on the training GEMM shapes (GB_ITERS launches per shape):
#   tools/ab_gemm.sh [suffix ...]     suffix "" = the shipped build, "prev" -> lib/libopadpo_hip_prev.so, ...
LIBS=${@:-"new prev"}
for i in 1 2; do
  for L in $LIBS; do
    if [ $L = new ]; then unset OPADPO_LIB_PATH; else export OPADPO_LIB_PATH=$PWD/opa-dpo_amd/lib/libopadpo_hip_$L.so; fi
    GB_ITERS=${GB_ITERS:-300} GB_M=${GB_M:-32362} python tools/gemm_bench.py 2>/dev/null | grep -E "glds.: 31" | python -c "
import sys,ast
r=[ast.literal_eval(l) for l in sys.stdin]
print('%-5s' % '$L', ' '.join('%s %.0f' % (x['name'], x['tflops']) for x in r))"
  done
done
