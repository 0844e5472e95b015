#!/bin/bash
# Timing + SQ counters of the attention kernels, dense causal shape (GB_ONLY=attn3) and packed bench shape (attn2), for the
# 32-rows-per-wave forward (OPADPO_ATTN32=1, default) and the 16-row forward (=0).  One --pmc pass per counter group, kernel-trace only.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_attn32
rm -rf $OUT; mkdir -p $OUT
cd /tmp
for V in 1 0; do
  for MODE in attn3 attn2; do
    OPADPO_ATTN32=$V GB_ONLY=$MODE python $R/tools/gemm_bench.py > $OUT/time_${MODE}_v$V.log 2>&1
  done
done
PASSES=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
        "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES"
        "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES_EQ_64 SQ_INSTS_SMEM")
for V in ${PMC_V:-1}; do
  i=0
  for C in "${PASSES[@]}"; do
    OPADPO_ATTN32=$V GB_ONLY=${GB_MODE:-attn3} timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/v${V}_p$i -- python $R/tools/gemm_bench.py > $OUT/v${V}_p$i.log 2>&1
    echo "v$V pass $i rc=$?"; i=$((i+1))
  done
done
python - <<PY
import csv, glob, collections, json
res = {}
for V in "${PMC_V:-1}".split():
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/v%s_p*/**/*counter_collection.csv" % V, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            name = next((n for n in ("attn_fwd32", "attn_fwd", "attn_bwd_dkdv", "attn_bwd_dq", "attn_delta") if n in k), None)
            if name:
                agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res["v" + V] = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
    for k, d in res["v" + V].items():
        wc = d.get("SQ_WAVE_CYCLES", 0) or 1
        print("v" + V, k)
        for c, v in sorted(d.items()):
            print(f"   {c:28s} {v:16.0f}  {v / wc:8.3f} of WAVE_CYCLES")
json.dump(res, open("$R/gpurun_out/pmc_attn32.json", "w"), indent=1)
PY
cat $OUT/time_*.log
