#!/bin/bash
# default bench line + sequential kernel-stats profile (one batch at a time) of the same state
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-r03_g}
python3 $R/bench.py > $O/${T}_default_line.json 2> $O/${T}_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -- python3 $R/bench.py --steps 2 --warmup 1 --pipelines 1 --no-cpu-baseline --kd-optimizer-steps 0 --no-length-mix --no-extra-legs > $O/${T}_prof_line.json 2> $O/${T}_prof.err
f=$(find $O/prof_$T -name "*kernel_stats.csv" | head -1); cp "$f" $O/${T}_kernel_stats.csv; rm -rf $O/prof_$T
