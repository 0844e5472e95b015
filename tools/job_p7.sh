#!/bin/bash
# long parity runs of the round: P7 at 7B with the bf16-emulating oracle passes, P7 at 13B (fp32 oracle only)
cd ${GRAFT_REPO_ROOT:-/root/repo}
OPADPO_P7_EMU=1 timeout 2400 python -m pytest tests/test_fullsize_gpu.py -x -q -k "p7_full_depth_32" -s 2>&1 | tail -15
cp gpurun_out/parity_fulldepth.json gpurun_out/r04_parity_fulldepth.json 2>/dev/null
OPADPO_P7_13B=1 timeout 3000 python -m pytest tests/test_fullsize_gpu.py -x -q -k "p7_13b" -s 2>&1 | tail -15
cp gpurun_out/parity_fulldepth_13b.json gpurun_out/r04_parity_fulldepth_13b.json 2>/dev/null
