#!/bin/bash
# round 5, GPU call G: KD after the lm_head slice threshold + out_proj weight-gradient split-K; KD parity tests; whole suite again (flake check)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_g; mkdir -p $O
timeout 1200 python -m pytest tests/test_train_models_gpu.py tests/test_train_kernels_gpu.py tests/test_dp_gpu.py -q --tb=short 2>&1 | tail -30 > $O/pytest_train.txt
KD="--batch 1 --steps 1 --warmup 0 --max-new-tokens 2 --pipelines 1 --no-cpu-baseline --kd-optimizer-steps 3 --no-length-mix --no-extra-legs --no-eos-leg"
for v in 1 0 1 0; do
  SL_SPLIT_K=$v timeout 600 python bench.py $KD 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
k=d['kd_step']; print('SL_SPLIT_K=$v', 'samples/s', k['samples_per_s'], 'window_ms', k['window_ms'], 'per-rank window', k['per_rank_regime_probe']['window_ms'])" >> $O/kd_ab.txt
done
timeout 2400 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -80 > $O/pytest_all.txt
tail -6 $O/pytest_train.txt; cat $O/kd_ab.txt; tail -12 $O/pytest_all.txt
