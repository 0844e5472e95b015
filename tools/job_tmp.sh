AB_CONFIGS="X=0 OPADPO_W4_STAGGER=1:2500 OPADPO_W4_STAGGER=1:6000 OPADPO_W4_STAGGER=2:600 OPADPO_W4_STAGGER=3:640 OPADPO_W4_STAGGER=3:200" GB_ITERS=150 bash tools/ab_env.sh 2>&1 | tee gpurun_out/r04_ab_stagger.txt
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-rollout --no-exchange-probe --no-side-legs"
P='import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.3f pairs/s  %.1f ms  frac %.4f" % (r["value"], r["ms_per_step"], r["roofline"]["frac"]))'
for i in 1 2; do
  echo -n "merged reference (default) "; $B 2>/dev/null | python -c "$P"
  echo -n "--no-merge-ref             "; $B --no-merge-ref 2>/dev/null | python -c "$P"
done 2>&1 | tee gpurun_out/r04_ab_merge_ref.txt
