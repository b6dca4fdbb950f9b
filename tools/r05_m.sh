#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_m; mkdir -p $O
timeout 1500 python -m pytest tests/test_train_models_gpu.py tests/test_train_kernels_gpu.py tests/test_kernels_gpu.py -q --tb=short -k "attn or attention or dropout or kd or trainer or regular" 2>&1 | tail -8 > $O/pytest.txt
KD="--batch 1 --steps 1 --warmup 0 --max-new-tokens 2 --pipelines 1 --no-cpu-baseline --kd-optimizer-steps 3 --no-length-mix --no-extra-legs --no-eos-leg"
for v in new base new base; do
  SL_DEV=1 SL_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/libspeechllm_$v.so timeout 600 python bench.py $KD 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
k=d['kd_step']; print('lib=$v', 'samples/s', k['samples_per_s'], 'window_ms', k['window_ms'], 'per-rank window', k['per_rank_regime_probe']['window_ms'], 'frac', k['roofline']['frac'])" >> $O/kd_ab.txt
done
cat $O/pytest.txt $O/kd_ab.txt
