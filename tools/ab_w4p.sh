#!/bin/bash
# same-box, sustained A/B of the persistent tile loop (gemm_nt_w4p_kernel, OPADPO_W4P=1, default) against one tile per workgroup
# (OPADPO_W4P=0) on the training GEMM shapes, interleaved processes, GB_ITERS launches per shape (default dispatch both ways)
for i in 1 2 3; do
  for V in 1 0; do
    OPADPO_W4P=$V GB_ONLY=gemm GB_VARIANTS=10 GB_ITERS=${GB_ITERS:-200} GB_M=${GB_M:-24576} python tools/gemm_bench.py 2>/dev/null | grep -E "'glds': 10" | python -c "
import sys,ast
r=[ast.literal_eval(l) for l in sys.stdin]
print('w4p=$V', ' '.join('%s %.0f' % (x['name'], x['tflops']) for x in r))"
  done
done
