#!/bin/bash
# round 5, GPU call O: split-K admitted for the encoder's weight gradients (256 small tiles under 126 slabs)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_o; mkdir -p $O
timeout 1200 python -m pytest tests/test_train_models_gpu.py tests/test_train_kernels_gpu.py tests/test_dp_gpu.py tests/test_kernels_gpu.py -q --tb=short -k "not attn_decode" 2>&1 | tail -8 > $O/pytest.txt
timeout 900 python tools/gemm_vs_vendor.py --rounds 3 --variants p,sk,vendor 2>&1 | grep -E "8064|shape|634 x" > $O/gemm_sk.txt
KD="--batch 1 --steps 1 --warmup 0 --max-new-tokens 2 --pipelines 1 --no-cpu-baseline --kd-optimizer-steps 3 --no-length-mix --no-extra-legs --no-eos-leg"
for v in 1 0 1 0; do
  SL_SPLIT_K=$v timeout 600 python bench.py $KD 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
k=d['kd_step']; print('SL_SPLIT_K=$v', 'samples/s', k['samples_per_s'], 'window_ms', k['window_ms'], 'per-rank window', k['per_rank_regime_probe']['window_ms'])" >> $O/kd_ab.txt
done
cat $O/pytest.txt | tail -3; cat $O/gemm_sk.txt $O/kd_ab.txt
