#!/bin/bash
# same-box, sustained A/B of the persistent tile loop (gemm_nt_w4p_kernel, OPADPO_W4P=1, default) against one tile per workgroup
# (OPADPO_W4P=0) on the training GEMM shapes, interleaved processes, GB_ITERS launches per shape (default dispatch both ways)
# W4P_CONFIGS="1:0 0:0 1:1 1:2": list of OPADPO_W4P:OPADPO_W4P_DBG pairs
for i in 1 2 3; do
  for C in ${W4P_CONFIGS:-"1:0 0:0"}; do
    V=${C%%:*}; D=${C##*:}
    OPADPO_W4P=$V OPADPO_W4P_DBG=$D GB_ONLY=gemm GB_VARIANTS=10 GB_ITERS=${GB_ITERS:-200} GB_M=${GB_M:-24576} python tools/gemm_bench.py 2>/dev/null | grep -E "'glds': 10" | python -c "
import sys,ast
r=[ast.literal_eval(l) for l in sys.stdin]
print('w4p=$V dbg=$D', ' '.join('%s %.0f' % (x['name'], x['tflops']) for x in r))"
  done
done
