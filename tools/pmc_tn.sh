#!/bin/bash
# SQ / LDS counters of gemm_tn_w4_kernel on the LoRA wgrad shapes (isolated problems of tools/gemm_bench.py), one --pmc pass per counter set
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_tn
rm -rf $OUT; mkdir -p $OUT
cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  GB_ONLY=gemm GB_VARIANTS=10 GB_ITERS=3 GB_M=24576 timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pass$i -- python $R/tools/gemm_bench.py > $OUT/pass$i.log 2>&1
  echo "pass $i: rc=$?"
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "gemm_tn_w4" in k or "gemm_nt_w4_kernel" in k:
            key = ("gemm_tn_w4" if "gemm_tn" in k else "gemm_nt_w4") + " grid " + r.get("Grid_Size", r.get("Grid_Size_X", ""))
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:30s} n={len(v):3d} mean={sum(v)/len(v):.5g}")
    if "SQ_LDS_BANK_CONFLICT" in d and "SQ_LDS_IDX_ACTIVE" in d:
        print("   -> bank conflicts / LDS active = %.3f" % (sum(d["SQ_LDS_BANK_CONFLICT"]) / max(1.0, sum(d["SQ_LDS_IDX_ACTIVE"]))))
    if "SQ_WAIT_ANY" in d and "SQ_WAVE_CYCLES" in d:
        print("   -> wait / wave cycles = %.3f,  MFMA busy cycles / (4 x wave quad-cycles) = %.3f" % (sum(d["SQ_WAIT_ANY"]) / sum(d["SQ_WAVE_CYCLES"]), sum(d["SQ_VALU_MFMA_BUSY_CYCLES"]) / (4.0 * sum(d["SQ_WAVE_CYCLES"]))))
PY
