#!/bin/bash
# same-box A/B of environment-switched variants on the WHOLE optimizer step (bench.py, 10 steps after 3 warm-up, no side legs), interleaved:
#   AB_CONFIGS="X=0 OPADPO_W4_NT=0 OPADPO_EW_NT=1" tools/ab_step_env.sh      (each config: comma-separated VAR=value pairs)
B="python bench.py --steps ${AB_STEPS:-10} --warmup 3 --no-cpu-baseline --no-rollout --no-exchange-probe --no-side-legs ${AB_ARGS:-}"
P='import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.3f pairs/s  %.1f ms  frac %.4f" % (r["value"], r["ms_per_step"], r["roofline"]["frac"]))'
mkdir -p gpurun_out
for i in 1 2; do
  for C in ${AB_CONFIGS:-"X=0"}; do
    printf "%-44s" "$C"; env $(echo $C | tr ',' ' ') $B 2>gpurun_out/ab_env.err | python -c "$P" || tail -3 gpurun_out/ab_env.err
  done
done
