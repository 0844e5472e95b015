#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "normed" 2>&1 | tail -3
for rep in 1 2; do
echo "B=8 new FUSE=0"; RB_BATCH=8 OPADPO_DEC_FUSE_NORM=0 timeout 600 python tools/rollout_bench.py 2>&1 | tail -1 | cut -c140-260
echo "B=8 new FUSE=1"; RB_BATCH=8 OPADPO_DEC_FUSE_NORM=1 timeout 600 python tools/rollout_bench.py 2>&1 | tail -1 | cut -c140-260
echo "B=8 prev"; RB_BATCH=8 OPADPO_LIB_PATH=/root/repo/opa-dpo_amd/lib/libopadpo_hip_prev.so timeout 600 python tools/rollout_bench.py 2>&1 | tail -1 | cut -c140-260
echo "B=64 new"; RB_BATCH=64 timeout 600 python tools/rollout_bench.py 2>&1 | tail -1 | cut -c140-260
echo "B=64 new ring"; RB_BATCH=64 OPADPO_DEC64_V=1 timeout 600 python tools/rollout_bench.py 2>&1 | tail -1 | cut -c140-260
echo "B=64 prev"; RB_BATCH=64 OPADPO_LIB_PATH=/root/repo/opa-dpo_amd/lib/libopadpo_hip_prev.so timeout 600 python tools/rollout_bench.py 2>&1 | tail -1 | cut -c140-260
done
