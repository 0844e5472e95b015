#!/bin/bash
# Round 6: the vendor library's kernel is 4-7 % ahead of gemm_nt_w4 on the deep-K products (profiles/r06u_ab_stream.txt).  Counters of both on the same shapes,
# one --pmc pass per group: wave-cycle split, instruction mix, LDS, L2.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_vendor
mkdir -p $OUT
cd /tmp
for SH in ${PV_SHAPES:-down o}; do
  i=0
  for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    PV_SHAPE=$SH timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/${SH}_pass$i -- python $R/tools/pmc_vendor.py > $OUT/${SH}_pass$i.log 2>&1
    echo "$SH pass $i ($C): rc=$?"
  done
done
python - <<PY
import csv, glob, collections
out = "$OUT"
for sh in "${PV_SHAPES:-down o}".split():
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    info = {}
    for f in glob.glob(out + f"/{sh}_pass*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if "gemm_nt" in k or "Cijk" in k or "MT" in k:
                agg[k[:150]][r["Counter_Name"]].append(float(r["Counter_Value"]))
                info[k[:150]] = (r.get("Grid_Size"), r.get("Workgroup_Size"), r.get("LDS_Block_Size"), r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"))
    print(f"== shape {sh}")
    for k, d in agg.items():
        print("  ", k)
        print("      grid, workgroup, LDS, VGPR, AGPR, SGPR:", info[k])
        for c, v in sorted(d.items()):
            print(f"      {c:32s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
PY
