#!/bin/bash
# KD leg A/B on one box: the in-tree library against tools/ab/libspeechllm_base.so, alternating
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1
ARGS="--batch 1 --steps 1 --warmup 0 --max-new-tokens 2 --pipelines 1 --no-cpu-baseline --kd-optimizer-steps 3 --no-length-mix --no-extra-legs"
for i in 1 2; do
  for v in new base; do
    if [ $v = base ]; then export SL_DEV=1 SL_LIB_PATH=$R/tools/ab/libspeechllm_base.so; else unset SL_LIB_PATH SL_DEV; fi
    python3 $R/bench.py $ARGS 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline())['kd_step']; print('$v', d['samples_per_s'], d['window_ms'], d['per_rank_regime_probe']['window_ms'])" >> $O/${T}.txt
  done
done
