#!/bin/bash
# sequential kernel-stats profile (one batch at a time): tools/exp_prof.sh <tag> [extra bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -- python3 $R/bench.py --steps 2 --warmup 1 --pipelines 1 --no-cpu-baseline --kd-optimizer-steps 0 --no-length-mix --no-extra-legs --no-eos-leg "$@" > $O/${T}_prof_line.json 2> $O/${T}_prof.err
f=$(find $O/prof_$T -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${T}_kernel_stats.csv; rm -rf $O/prof_$T
grep -v "^    @" $O/${T}_prof.err | tail -5
