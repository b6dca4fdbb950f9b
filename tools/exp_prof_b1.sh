#!/bin/bash
# kernel-stats profile of the batch-1 decode step (graph replay): tools/exp_prof_b1.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -- python3 $R/tools/time_decode_step.py 1 > $O/${T}_b1.txt 2> $O/${T}_b1.err
f=$(find $O/prof_$T -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${T}_b1_kernel_stats.csv; rm -rf $O/prof_$T
cat $O/${T}_b1.txt
