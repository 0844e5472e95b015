#!/bin/bash
# Round-end measurement campaign on the GPU box: full -m gpu suite (timed), driver-style bench line, kernel stats, PMC traffic, attention
# SQ counters, sustained config-2 run, full-depth parity with the bf16-emulating pass, 13B line, rollout kernel stats.
# LIGHT=1 skips the attention counters, the full-depth parity re-run with the emulating pass and the native-unit / SFT benches.
# Everything lands under gpurun_out/camp_<tag>/ ; copy what should be judged into profiles/.
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/camp_$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  ( time python -m pytest tests/ -x -q -m gpu ) > $O/gpu_tests.log 2>&1
  tail -4 $O/gpu_tests.log
fi
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python -c "import json;r=json.load(open('$O/bench_default.json'));print('bench', r['value'], r['ms_per_step'], r['roofline']['frac'])"
bash tools/prof.sh ${TAG}_bench python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-rollout --no-exchange-probe --no-side-legs > $O/prof.log 2>&1
cp $R/gpurun_out/${TAG}_bench_kernel_stats.csv $O/bench_7b_kernel_stats.csv
bash tools/pmc_bench.sh > $O/pmc_bench.log 2>&1; cp $R/gpurun_out/pmc_bench.json $O/pmc_traffic.json
if [ "${LIGHT:-0}" != "1" ]; then PMC_V=1 GB_MODE=attn2 bash tools/pmc_attn32.sh > $O/pmc_attn.log 2>&1; cp $R/gpurun_out/pmc_attn32.json $O/pmc_attn_sq.json; fi
python bench.py --steps 220 --warmup 5 --sustained --no-rollout --no-side-legs --no-cpu-baseline --no-exchange-probe > $O/sustained.json 2> $O/sustained.err
python -c "import json;r=json.load(open('$O/sustained.json'));s=r['sustained'];print('sustained', r['value'], s['pairs_per_s_first_20'], s['pairs_per_s_last_20'], s['sclk_mhz'], s['power_w'])"
if [ "${LIGHT:-0}" != "1" ]; then python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -k p7 > $O/p7.log 2>&1; tail -2 $O/p7.log; cp $R/gpurun_out/parity_fulldepth.json $O/parity_fulldepth.json; fi
python bench.py --model 13b --steps 6 --warmup 2 --no-rollout --no-side-legs --no-exchange-probe > $O/bench_13b.json 2> $O/bench_13b.err
python -c "import json;r=json.load(open('$O/bench_13b.json'));print('13b', r['value'], r['ms_per_step'], r['hbm_peak_allocated_GB'])"
GB_M=24576 python tools/gemm_bench.py > $O/gemm_bench.txt 2>/dev/null      # per shape: this library's kernels and the vendor GEMM (torch.matmul -> hipBLASLt) side by side
if [ "${LIGHT:-0}" != "1" ]; then
python tools/sample_bench.py > $O/sample_bench.txt 2>&1      # native OPA-DPO unit (3 responses + CoPO + AncPO)
python tools/sft_bench.py > $O/sft_bench.txt 2>&1            # OPA LoRA-SFT step
fi
cp $R/gpurun_out/parity_bench_config.json $O/ 2>/dev/null
# rollout (BASELINE configs[4]): per-kernel stats of the decode steps at 8 and 64 sequences per device
for RB in 8 64; do
  RB_BATCH=$RB bash tools/prof.sh ${TAG}_rollout_b$RB python $R/tools/rollout_bench.py > $O/prof_rollout_b$RB.log 2>&1
  cp $R/gpurun_out/${TAG}_rollout_b${RB}_kernel_stats.csv $O/rollout_b${RB}_kernel_stats.csv 2>/dev/null
done
GB_ONLY=attn2 GB_ITERS=60 python tools/gemm_bench.py > $O/attn_bench.txt 2>/dev/null      # attention kernels at the packed bench shape
