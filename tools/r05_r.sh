#!/bin/bash
# round 5, run r: three batches in flight against two (default), same box
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_r; mkdir -p $O
ARGS="--no-cpu-baseline --no-extra-legs --no-length-mix --kd-optimizer-steps 0 --no-eos-leg"
for cfg in "2 4" "3 6" "2 4" "3 6"; do
  set -- $cfg
  timeout 600 python bench.py --pipelines $1 --steps $2 $ARGS 2> $O/p$1.err | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('pipelines $1 steps $2', d['value'], d['ms_per_step'])" >> $O/pipes.txt
  grep "peak torch" $O/p$1.err | tail -1 >> $O/pipes.txt
done
cat $O/pipes.txt
