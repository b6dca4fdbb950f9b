#!/bin/bash
# tools/exp_kd_trace.sh <tag> [KD_WINDOW]: kernel timeline of a KD window (tools/kd_window_trace.py)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1; cd /tmp; export TMPDIR=/tmp
[ -n "$2" ] && export KD_WINDOW=$2
rm -rf $O/trace_$T
rocprofv3 --kernel-trace --output-format csv -d $O/trace_$T -- python3 $R/tools/kd_window_trace.py > $O/${T}_kd_trace.log 2>&1
f=$(find $O/trace_$T -name "*kernel_trace.csv" | head -1)
python3 $R/tools/kd_window_trace.py "$f" > $O/${T}_kd_timeline.txt 2>&1
grep "^window" $O/${T}_kd_trace.log >> $O/${T}_kd_timeline.txt
rm -rf $O/trace_$T
