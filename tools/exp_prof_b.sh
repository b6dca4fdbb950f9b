#!/bin/bash
# kernel-stats profile of the decode step at one batch size (graph replay): tools/exp_prof_b.sh <tag> <B>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1; B=$2
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -- python3 $R/tools/time_decode_step.py $B > $O/${T}_b.txt 2> $O/${T}_b.err
f=$(find $O/prof_$T -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${T}_b${B}_kernel_stats.csv; rm -rf $O/prof_$T
cat $O/${T}_b.txt
