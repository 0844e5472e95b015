#!/bin/bash
# run the MFMA-shape power probe with a rocm-smi sampler beside it (clock / power once per second)
R=${GRAFT_REPO_ROOT:-/root/repo}
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)|Average Graphics Package Power|Current Socket" | tr '\n' ' '; echo; sleep 1; done ) > /tmp/smi.log 2>&1 &
SMI=$!
$R/tools/micro/mfma_power.bin 0 4 | tee /tmp/mfma_random.txt
$R/tools/micro/mfma_power.bin 1 3 | tee /tmp/mfma_zero.txt
kill $SMI
echo "--- rocm-smi samples (one per second over the runs above)"
cat /tmp/smi.log | sed 's/  */ /g' | cut -c1-200
