#!/bin/bash
# round 6, x: TRUE per-kernel totals of KD windows (rocprofv3 --kernel-trace --stats over tools/kd_window_trace.py: 5 windows incl. the first, cold one)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_x; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/tools/kd_window_trace.py > $O/trace.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kd_window16_kernel_stats.csv; rm -rf $O/prof
python3 - <<'PY'
import csv, os
rows = list(csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r06_x/kd_window16_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms over 5 windows (+ set-up):", tot / 1e6)
for r in rows[:40]:
    print(f"{float(r['Percentage']):6.2f}%  calls {int(r['Calls']):6d}  avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms  {r['Name'][:100]}")
PY
