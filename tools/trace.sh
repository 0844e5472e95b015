#!/bin/bash
# usage: tools/trace.sh <tag> <command...>  -> rocprofv3 --kernel-trace of the command; per-dispatch rows (name, grid, start, end) of
# the kernels in gpurun_out/<tag>_trace.csv (names cut to 60 chars) for per-shape analysis (tools/trace_shapes.py)
TAG=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=/tmp/trace_$TAG
rm -rf $OUT; mkdir -p $OUT $R/gpurun_out
cd /tmp
timeout 1200 rocprofv3 --kernel-trace --output-format csv -d $OUT -- "$@" > $OUT/run.log 2>&1
echo "rocprofv3 rc=$?"
F=$(find $OUT -name "*kernel_trace.csv" | head -1)
python - "$F" "$R/gpurun_out/${TAG}_trace.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
w = csv.writer(open(sys.argv[2], "w"))
w.writerow(["name", "grid", "wg", "start_ns", "dur_ns"])
t0 = int(rows[0]["Start_Timestamp"]) if rows else 0
for r in rows:
    w.writerow([r["Kernel_Name"][:60], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")),
                int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])])
print(len(rows), "dispatches")
PY
