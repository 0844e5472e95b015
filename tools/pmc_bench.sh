#!/bin/bash
# HBM traffic of the dominant kernel under the bench workload: two separate --pmc passes (FETCH_SIZE, WRITE_SIZE),
# --kernel-trace only (no sys/hip/hsa trace), one optimizer step of the default 22-pair (packed) micro-batch (PAIRS=...).  Summary -> gpurun_out/pmc_bench.json
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_bench
rm -rf $OUT; mkdir -p $OUT
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -- python $R/bench.py --steps 1 --warmup 0 --pairs ${PAIRS:-22} --accum 1 --no-cpu-baseline --no-rollout --no-exchange-probe --no-side-legs > $OUT/$C.log 2>&1
  echo "pass $C rc=$?"
done
python - <<PY
import csv, glob, json, collections
out = "$OUT"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        name = "gemm_nt_w4" if "gemm_nt_w4" in k else "gemm_nt_p8" if "gemm_nt_p8" in k else ("gemm_nt_pp" if "gemm_nt_pp" in k else ("gemm_nt_x" if "gemm_nt_kernel_x" in k else ("gemm_tn" if "gemm_tn" in k else ("attn" if "attn_" in k else None))))
        if name:
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, d in agg.items():
    f = d.get("FETCH_SIZE", [0]); w = d.get("WRITE_SIZE", [0])
    # rocprofv3 reports KiB; gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide coalesced streams -> x2 (MI355X_MICROARCH.md §HBM)
    res[k] = {"launches": len(f), "fetch_KiB_raw_mean": sum(f) / len(f), "write_KiB_mean": sum(w) / max(1, len(w)),
              "hbm_bytes_per_launch_corrected": (2 * sum(f) / len(f) + sum(w) / max(1, len(w))) * 1024}
nt = [res[k] for k in ("gemm_nt_w4", "gemm_nt_p8", "gemm_nt_pp", "gemm_nt_x") if k in res]
tot_l = sum(r["launches"] for r in nt)
full = {"command": "tools/pmc_bench.sh (rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE, separate passes; python bench.py --steps 1 --warmup 0 --pairs ${PAIRS:-22} --accum 1 --no-rollout --no-exchange-probe --no-side-legs; ragged rows, context path)",
        "correction": "FETCH_SIZE x2 on gfx950 for wide coalesced streaming reads (MI355X_MICROARCH.md HBM section); values are KiB in the raw counters; the memory-side counters include Infinity-Cache hits",
        "kernels": res,
        "gemm_nt_avg_hbm_bytes_per_launch": sum(r["hbm_bytes_per_launch_corrected"] * r["launches"] for r in nt) / max(1, tot_l)}
json.dump(full, open("$R/gpurun_out/pmc_bench.json", "w"), indent=1)
print(json.dumps(full, indent=1))
PY
