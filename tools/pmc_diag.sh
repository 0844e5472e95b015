#!/bin/bash
# SQ cycle counters of gemm_nt_w4 (4096^3 / 8192^3) for several builds of the library (tools/build_diag.sh): clock-independent cost of each
# stall source of the K-loop.    tools/pmc_diag.sh new diag2 diag4 ...     ->  gpurun_out/pmc_diag.txt
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_diag
rm -rf $OUT; mkdir -p $OUT
cd /tmp
C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"
for L in "$@"; do
  if [ $L = new ]; then unset OPADPO_LIB_PATH; V=31,-1; else export OPADPO_LIB_PATH=$R/opa-dpo_amd/lib/libopadpo_hip_$L.so; V=31; fi
  GB_ONLY=cube GB_VARIANTS=$V timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$L -- python $R/tools/gemm_bench.py > $OUT/$L.log 2>&1
  echo "$L rc=$?"
done
python - <<PY
import csv, glob, collections, os
for d in sorted(glob.glob("$OUT/*/")):
    lib = os.path.basename(d.rstrip("/"))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            name = "w4" if "gemm_nt_w4" in k else ("vendor" if k.startswith("Cijk") else None)
            if name: agg[name + "_g" + r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in sorted(agg.items()):
        n = {"65536": 256 * 64 * 4, "262144": 1024 * 128 * 4}.get(k.split("_g")[1], 1)
        print("%-8s %-16s " % (lib, k) + "  ".join("%s %.1f" % (c.replace("SQ_", ""), sum(v) / len(v) / n) for c, v in sorted(cs.items())))
PY
