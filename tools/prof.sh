#!/bin/bash
# usage: tools/prof.sh <tag> <command...>   -> rocprofv3 --kernel-trace --stats of the command (run on the GPU box);
# compact per-kernel summary in gpurun_out/<tag>_kernel_stats.csv (kernel names truncated to 90 chars)
TAG=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=/tmp/prof_$TAG
rm -rf $OUT; mkdir -p $OUT $R/gpurun_out
cd /tmp
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- "$@" > $OUT/run.log 2>&1
echo "rocprofv3 rc=$?"; tail -2 $OUT/run.log | cut -c1-300
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
T=$(find $OUT -name "*kernel_trace.csv" | head -1)
python - "$F" "$R/gpurun_out/${TAG}_kernel_stats.csv" "$T" <<'PY'
import csv, sys, collections, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
# per-kernel median and the total without one-off outliers (a first launch can carry code-object loading: seconds), from the trace itself
dur = collections.defaultdict(list)
if len(sys.argv) > 3 and sys.argv[3]:
    for r in csv.DictReader(open(sys.argv[3])):
        dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
w = csv.writer(open(sys.argv[2], "w"))
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "MedianNs", "TotalNsWithoutOutliers", "AverageNsWithoutOutliers", "Outliers"])
for r in rows:
    d = dur.get(r["Name"], [])
    med = statistics.median(d) if d else 0
    keep = [x for x in d if x <= 20 * med] if d else []
    w.writerow([r["Name"][:90], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], int(med), sum(keep),
                int(sum(keep) / len(keep)) if keep else 0, len(d) - len(keep)])
for r in rows[:24]:
    print(r["Name"][:64].ljust(64), r["Calls"].rjust(7), r["AverageNs"].rjust(12), r["Percentage"].rjust(7))
PY
