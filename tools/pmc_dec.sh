#!/bin/bash
# HBM-side traffic of the decode GEMM launches (gemm_nt_dec64x at 64 tokens): rocprofv3 --pmc FETCH_SIZE (x2 on gfx950 for wide streaming reads, KiB)
# next to the weight bytes of each shape.   tools/pmc_dec.sh  ->  gpurun_out/pmc_dec.txt
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_dec
rm -rf $OUT; mkdir -p $OUT
cd /tmp
GB_ONLY=dec GB_MS=64 timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/p0 -- python $R/tools/gemm_bench.py > $OUT/p0.log 2>&1
echo "rc=$?"
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/p0/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "dec64x" in k and r["Counter_Name"] == "FETCH_SIZE":
            agg[(k[k.index("dec64x"):][:40], r.get("Grid_Size", "?"))].append(float(r["Counter_Value"]))
shapes = {"qkv": 12288 * 4096 * 2, "o": 4096 * 4096 * 2, "gate_up": 22016 * 4096 * 2, "down": 4096 * 11008 * 2, "lm_head": 32000 * 4096 * 2}
print("weight bytes (MB):", {k: round(v / 1e6, 1) for k, v in shapes.items()})
for (k, g), v in sorted(agg.items()):
    print("%-42s grid %-8s launches %4d  FETCH_SIZE x2 = %8.1f MB per launch" % (k, g, len(v), sum(v) / len(v) * 2 * 1024 / 1e6))
PY
