#!/bin/bash
# K-loop text for the >= 128-K-tile products: OPADPO_W4_DEEP=0 default text, 1 DEEP text on one tile per workgroup, 2 (shipped) DEEP text, streaming where eligible.
# Isolated shapes, then the bench step; whole processes alternated
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for rep in 1 2; do
  for v in 0 1 2; do
    echo "== OPADPO_W4_DEEP=$v (rep $rep)"
    OPADPO_W4_DEEP=$v AB_VENDOR=0 AB_VARIANTS=10,31 AB_M=${AB_M:-24576,32362} AB_SHAPES=down,dgrad_gu,dgrad_qkv timeout 300 python tools/ab_stream.py 2>&1 | grep -v amdgpu.ids | cut -c1-230
  done
done
for rep in 1 2; do
  for v in 0 1 2; do
    echo "== bench OPADPO_W4_DEEP=$v (rep $rep)"
    OPADPO_W4_DEEP=$v timeout 900 python bench.py --steps 6 --warmup 2 --no-extra-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
  done
done
