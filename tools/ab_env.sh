#!/bin/bash
# same-box, sustained A/B of environment-switched kernel variants on the training GEMM shapes, interleaved processes:
#   AB_CONFIGS="OPADPO_W4_NT=0 OPADPO_W4_NT=1" tools/ab_env.sh          (each config: comma-separated VAR=value pairs)
for i in 1 2 3; do
  for C in ${AB_CONFIGS:-"X=0"}; do
    env $(echo $C | tr ',' ' ') GB_ONLY=gemm GB_VARIANTS=10 GB_ITERS=${GB_ITERS:-200} GB_M=${GB_M:-24576} python tools/gemm_bench.py 2>/dev/null | grep -E "'glds': 10" | python -c "
import sys,ast
r=[ast.literal_eval(l) for l in sys.stdin]
print('%-40s' % '$C', ' '.join('%s %.0f' % (x['name'], x['tflops']) for x in r))"
  done
done
