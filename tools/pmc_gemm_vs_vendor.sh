#!/bin/bash
# SQ / L2 counters of gemm_nt_w4 next to the vendor kernel torch.matmul dispatches (hipBLASLt), same operands, 4096^3 and 8192^3
# (GB_ONLY=cube).  One --pmc pass per counter group, kernel-trace only.  Summary -> gpurun_out/pmc_gemm_vs_vendor.json
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_gemm_vs_vendor
rm -rf $OUT; mkdir -p $OUT
cd /tmp
GB_ONLY=cube GB_VARIANTS=31,-1 python $R/tools/gemm_bench.py > $OUT/time.log 2>&1
PASSES=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
        "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES"
        "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_INST_CYCLES_SALU"
        "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
        "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum TCC_EA0_RD_UNCACHED_32B_sum"
        "FETCH_SIZE"
        "GRBM_GUI_ACTIVE GRBM_COUNT")
i=0
for C in "${PASSES[@]}"; do
  GB_ONLY=cube GB_VARIANTS=31,-1 timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p$i -- python $R/tools/gemm_bench.py > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"; i=$((i+1))
done
python - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        name = "w4" if "gemm_nt_w4" in k else ("vendor" if k.startswith("Cijk") else None)
        if name:
            # 4096^3 and 8192^3 launches differ in grid size: key on it
            agg[name + "_g" + r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
for k, d in sorted(res.items()):
    wc = d.get("SQ_WAVE_CYCLES", 0) or 1
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:30s} {v:18.0f}  {v / wc:8.3f} of WAVE_CYCLES")
json.dump(res, open("$R/gpurun_out/pmc_gemm_vs_vendor.json", "w"), indent=1)
PY
cat $OUT/time.log
grep -l "rror" $OUT/p*.log | head
