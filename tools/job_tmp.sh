cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r04b_bench_default.json 2> gpurun_out/r04b_bench_default.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/r04b_bench_default.json; tail -3 gpurun_out/r04b_bench_default.err
