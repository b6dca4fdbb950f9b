#!/bin/bash
# KD leg A/B on one box by environment switch: tools/exp_kd_env_ab.sh <tag> <VAR=value>   (alternating default / switched, two rounds)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1; SW=$2
ARGS="--batch 1 --steps 1 --warmup 0 --max-new-tokens 2 --pipelines 1 --no-cpu-baseline --kd-optimizer-steps 3 --no-length-mix --no-extra-legs"
for i in 1 2; do
  for v in default "$SW"; do
    if [ "$v" = default ]; then e=""; else e="$v"; fi
    env $e python3 $R/bench.py $ARGS 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline())['kd_step']; print('$v', d['samples_per_s'], d['window_ms'], d['per_rank_regime_probe']['window_ms'])" >> $O/${T}.txt
  done
done
