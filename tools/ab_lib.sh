B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-rollout --no-exchange-probe --no-side-legs"
P='import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r["value"], r["ms_per_step"], r["roofline"]["frac"])'
mkdir -p gpurun_out
for i in 1 2; do
  echo -n "new  "; $B 2>gpurun_out/ab_new.err | python -c "$P" || tail -5 gpurun_out/ab_new.err
  echo -n "prev "; OPADPO_LIB_PATH=$PWD/opa-dpo_amd/lib/libopadpo_hip_prev.so $B 2>gpurun_out/ab_prev.err | python -c "$P" || tail -5 gpurun_out/ab_prev.err
done
