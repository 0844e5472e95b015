#!/bin/bash
# ring decode GEMM (<= 64 tokens) of several library builds, one box:   tools/ab_dec.sh new d1 d2 ...
for L in "$@"; do
  if [ $L = new ]; then unset OPADPO_LIB_PATH; else export OPADPO_LIB_PATH=$PWD/opa-dpo_amd/lib/libopadpo_hip_$L.so; fi
  GB_ONLY=dec GB_MS=${GB_MS:-64} python tools/gemm_bench.py 2>/dev/null | grep -E "'kernel': '(${GB_KERNELS:-ring})'" | python -c "
import sys,ast
r=[eval(l, {'nan': 0.0, 'inf': 0.0}) for l in sys.stdin]
print('%-5s' % '$L', ' '.join('%s/%s %.1fus %dGB/s' % (x['name'], x['kernel'], x['us'], x['GBps']) for x in r))"
done
