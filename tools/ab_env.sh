#!/bin/bash
# tools/ab_env.sh VAR v0 v1 ...: tools/ab_stream.py (isolated GEMM shapes, default dispatch and one tile per workgroup) under VAR=v for each value, whole processes alternated twice;
# AB_BENCH=1 adds the bench step the same way
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
VAR=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    echo "== $VAR=$v (rep $rep)"
    env $VAR=$v AB_VENDOR=0 AB_VARIANTS=10,31 AB_M=${AB_M:-24576,32362} AB_SHAPES=${AB_SHAPES:-qkv,o,gate_up,down,lm_head,dgrad_gu} timeout 400 python tools/ab_stream.py 2>&1 | grep -v amdgpu.ids | cut -c1-230
  done
done
if [ "${AB_BENCH:-0}" = "1" ]; then
for rep in 1 2; do
  for v in "$@"; do
    echo "== bench $VAR=$v (rep $rep)"
    env $VAR=$v timeout 900 python bench.py --steps 6 --warmup 2 --no-extra-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
  done
done
fi
