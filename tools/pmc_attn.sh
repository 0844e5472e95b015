#!/bin/bash
# SQ stall breakdown of the attention kernels at the bench shape (packed pairs): one --pmc pass, kernel-trace only.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_attn
rm -rf $OUT; mkdir -p $OUT
cd /tmp
PASSES=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
        "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES")
i=0
for C in "${PASSES[@]}"; do
  GB_ONLY=${GB_MODE:-attn2} GB_S=${GB_S:-22} timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p$i -- python $R/tools/gemm_bench.py > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"; i=$((i+1))
done
python - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        name = next((n for n in ("attn_fwd", "attn_bwd_dkdv", "attn_bwd_dq", "gemm_nt_pp", "gemm_nt_p8", "gemm_nt_w4", "gemm_nt_kernel_x", "Cijk", "gemm_tn") if n in k), None)
        if name:
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
for k, d in res.items():
    wc = d.get("SQ_WAVE_CYCLES", 0) or 1
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {v:16.0f}  {v / wc:8.3f} of WAVE_CYCLES")
json.dump(res, open("$R/gpurun_out/pmc_attn.json", "w"), indent=1)
PY
