#!/bin/bash
# PMC passes over the GEMM micro-benchmark (run on the GPU box). Usage: tools/pmc_gemm.sh <outdir>
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$R/gpurun_out/pmc}
mkdir -p $OUT
cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  GB_ONLY=pmc timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pass$i -- python $R/tools/gemm_bench.py > $OUT/pass$i.log 2>&1
  echo "pass $i ($C): rc=$?"
done
python - <<PY
import csv, glob, collections, os
out = "$OUT"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")[:60]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as o:
    for k, d in agg.items():
        o.write(k + "\n")
        for c, v in sorted(d.items()):
            o.write(f"   {c:36s} n={len(v):4d} mean={sum(v)/len(v):.6g}\n")
print(open(out + "/summary.txt").read()[:6000])
PY
