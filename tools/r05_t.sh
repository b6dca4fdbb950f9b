#!/bin/bash
# round 5, run t: longer KD A/B of SL_WGRAD_TR (4 alternations, 5 optimizer steps each)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_t; mkdir -p $O
ARGS="--batch 1 --steps 1 --warmup 0 --max-new-tokens 2 --pipelines 1 --no-cpu-baseline --kd-optimizer-steps 5 --no-length-mix --no-extra-legs"
for i in 1 2 3 4; do
  for v in default SL_WGRAD_TR=1; do
    if [ "$v" = default ]; then e=""; else e="$v"; fi
    env $e python3 bench.py $ARGS 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline())['kd_step']; print('$v', d['samples_per_s'], d['window_ms'], d['per_rank_regime_probe']['window_ms'])" >> $O/kd_ab.txt
  done
done
cat $O/kd_ab.txt
