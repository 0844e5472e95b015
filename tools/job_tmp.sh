python -m pytest tests/test_ops_gpu.py -x -q -k "persistent or split_k_tail" 2>&1 | tail -4
W4P_CONFIGS="0:0 1:0" GB_ITERS=150 bash tools/ab_w4p.sh 2>&1 | tee gpurun_out/r04_ab_w4p_v2.txt
