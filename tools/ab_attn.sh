#!/bin/bash
# same-box A/B of the attention kernels (dense causal attn3 + packed bench shape attn2) between library builds, interleaved rounds:
#   tools/ab_attn.sh [suffix ...]     "new" = the shipped build, "prev" -> lib/libopadpo_hip_prev.so
LIBS=${@:-"new prev"}
for i in 1 2 3; do
  for L in $LIBS; do
    if [ $L = new ]; then unset OPADPO_LIB_PATH; else export OPADPO_LIB_PATH=$PWD/opa-dpo_amd/lib/libopadpo_hip_$L.so; fi
    for MODE in attn3 attn2; do
      GB_ITERS=${GB_ITERS:-60} GB_ONLY=$MODE python tools/gemm_bench.py 2>/dev/null | python -c "
import sys,ast
r=[ast.literal_eval(l) for l in sys.stdin if l.startswith('{')]
print('%-5s %-6s' % ('$L', '$MODE'), ' '.join('%s %.3f ms %.0f TF' % (x['kernel'], x['ms'], x['tflops']) for x in r))"
    done
  done
done
