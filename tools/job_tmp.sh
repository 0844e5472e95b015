cd ${GRAFT_REPO_ROOT:-/root/repo}
python tools/cpu_baseline_full.py 2>/dev/null | tail -1
bash tools/pmc_bench.sh > gpurun_out/r04_pmc_bench.log 2>&1; tail -5 gpurun_out/r04_pmc_bench.log | cut -c1-300
PMC_V=1 GB_MODE=attn2 bash tools/pmc_attn32.sh > gpurun_out/r04_pmc_attn32.log 2>&1; tail -3 gpurun_out/r04_pmc_attn32.log
bash tools/prof.sh r04b_bench_7b python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-rollout --no-exchange-probe --no-side-legs 2>&1 | tail -22
