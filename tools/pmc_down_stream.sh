#!/bin/bash
# Round 6, VERDICT r05 Next #4a: why does the STREAMING form of the 256x256 kernel lose 1.5-3 % on the K = 11264 down projection (176 K-tiles per output tile)?
# Same shape, same box: one tile per workgroup (OPADPO_W4S_MAXNT=128, the shipped choice) vs streaming (OPADPO_W4S_MAXNT=256) - sustained TF/s, then the L2
# counters (TCC_HIT / TCC_MISS / TCC_REQ per XCD sums), the fabric bytes (FETCH_SIZE x 2 on gfx950, WRITE_SIZE) and the wave-cycle split, one --pmc pass each.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_down
mkdir -p $OUT
cd $R
for NT in 128 256; do
  echo "== OPADPO_W4S_MAXNT=$NT: sustained rate (300 launches), M = 24576"
  OPADPO_W4S_MAXNT=$NT GB_ONLY=gemm GB_VARIANTS=10 GB_ITERS=300 GB_M=24576 python tools/gemm_bench.py 2>/dev/null | grep -E "'down'|'o'" | cut -c1-200
done
cd /tmp
for NT in 128 256; do
  i=0
  for C in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
    i=$((i+1))
    OPADPO_W4S_MAXNT=$NT GB_SHAPES=down GB_M=24576 GB_ONLY=pmc timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/nt${NT}_pass$i -- python $R/tools/gemm_bench.py > $OUT/nt${NT}_pass$i.log 2>&1
    echo "nt $NT pass $i ($C): rc=$?"
  done
done
python - <<PY
import csv, glob, collections
out = "$OUT"
for nt in (128, 256):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(out + f"/nt{nt}_pass*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")[:70]
            if "gemm_nt" in k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"== OPADPO_W4S_MAXNT={nt}")
    for k, d in agg.items():
        print("  ", k)
        for c, v in sorted(d.items()):
            print(f"      {c:32s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
PY
