#!/bin/bash
# tools/ab_kloop_exp.sh name1 name2 ...: the shipped library against experiment builds of the generated K-loop (tools/build_kloop_exp.sh), whole processes alternated twice
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do
  for n in shipped "$@"; do
    P=$R/opa-dpo_amd/lib/libopadpo_hip.so; [ $n != shipped ] && P=$R/opa-dpo_amd/lib/libopadpo_hip_$n.so
    echo "== $n (rep $rep)"
    OPADPO_LIB_PATH=$P AB_VENDOR=0 AB_M=${AB_M:-24576} AB_SHAPES=${AB_SHAPES:-o,down,gate_up,dgrad_gu} timeout 300 python tools/ab_stream.py 2>&1 | grep -v amdgpu.ids | cut -c1-220
  done
done
