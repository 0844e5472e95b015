#!/bin/bash
# usage: tools/prof_grid.sh <tag> <command...>   -> rocprofv3 --kernel-trace of the command; per (kernel, grid, workgroup) launch count, median / mean duration
# in gpurun_out/<tag>_by_grid.txt (separates the launches of ONE kernel by problem shape: o vs down on the same decode kernel)
TAG=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=/tmp/profg_$TAG
rm -rf $OUT; mkdir -p $OUT $R/gpurun_out
cd /tmp
timeout 1200 rocprofv3 --kernel-trace --output-format csv -d $OUT -- "$@" > $OUT/run.log 2>&1
echo "rocprofv3 rc=$?"; tail -2 $OUT/run.log | cut -c1-300
T=$(find $OUT -name "*kernel_trace.csv" | head -1)
python - "$T" > $R/gpurun_out/${TAG}_by_grid.txt <<'PY'
import csv, sys, collections, statistics
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    key = (r["Kernel_Name"][:70], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")))
    d[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = sorted(d.items(), key=lambda kv: -sum(kv[1]))
tot = sum(sum(v) for v in d.values())
for k, v in rows[:40]:
    print("%-70s grid %8s x %3s wg %4s  calls %6d  median %9.2f us  mean %9.2f us  %5.2f %%" % (k[0], k[1], k[2], k[3], len(v), statistics.median(v) / 1e3, sum(v) / len(v) / 1e3, 100.0 * sum(v) / tot))
PY
head -30 $R/gpurun_out/${TAG}_by_grid.txt
