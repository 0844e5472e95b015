#!/bin/bash
# same-box A/B of the attention forward: 32 rows per wave (OPADPO_ATTN64=0) vs 64 rows per wave (OPADPO_ATTN64=1); dense causal (attn3) + packed bench shape (attn2)
mkdir -p gpurun_out
for i in 1 2 3; do
  for V in 0 1; do
    for MODE in attn3 attn2; do
      OPADPO_ATTN64=$V GB_ITERS=${GB_ITERS:-60} GB_ONLY=$MODE python tools/gemm_bench.py 2>/dev/null | python -c "
import sys,ast
r=[ast.literal_eval(l) for l in sys.stdin if l.startswith('{')]
print('attn64=$V %-6s' % ('$MODE'), ' '.join('%s %.3f ms %.0f TF' % (x['kernel'], x['ms'], x['tflops']) for x in r))"
    done
  done
done
