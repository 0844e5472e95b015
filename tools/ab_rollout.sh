#!/bin/bash
# decode ms/step of the rollout (tools/rollout_bench.py) for several library builds / batch sizes on one box
#   tools/ab_rollout.sh "8 16 32 64" new prev          (OPADPO_DEC64_MIN etc. pass through)
BS=$1; shift
for B in $BS; do
  for L in "$@"; do
    if [ $L = new ]; then unset OPADPO_LIB_PATH; else export OPADPO_LIB_PATH=$PWD/opa-dpo_amd/lib/libopadpo_hip_$L.so; fi
    RB_BATCH=$B python tools/rollout_bench.py 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('B=%-3d %-5s decode %.3f ms/step  prefill %.1f ms  %.0f tok/s' % (r['batch'], '$L', r['decode_ms_per_step'], r['prefill_ms'], r['decode_tokens_per_s']))"
  done
done
