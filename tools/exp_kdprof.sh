#!/bin/bash
# KD leg almost alone under the profiler: tools/exp_kdprof.sh <tag> [extra bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1; shift
python3 $R/bench.py --batch 1 --steps 1 --warmup 0 --max-new-tokens 2 --pipelines 1 --no-cpu-baseline --kd-optimizer-steps 3 --no-length-mix --no-extra-legs "$@" > $O/${T}_kd_line.json 2> $O/${T}_kd.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -- python3 $R/bench.py --batch 1 --steps 1 --warmup 0 --max-new-tokens 2 --pipelines 1 --no-cpu-baseline --kd-optimizer-steps 3 --kd-local-accum 0 --no-length-mix --no-extra-legs "$@" > $O/${T}_kdprof_line.json 2> $O/${T}_kdprof.err
f=$(find $O/prof_$T -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${T}_kd_kernel_stats.csv; rm -rf $O/prof_$T
