#!/bin/bash
# round 5, run u: weight gradients from token-major operands (SL_WGRAD_TR=1) — test, per-shape timing, KD leg A/B
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_u; mkdir -p $O
timeout 600 python -m pytest tests/test_train_kernels_gpu.py -x -q -m gpu -k "wgrad" > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
timeout 300 python tools/time_wgrad_tt.py > $O/time_wgrad_tt.txt 2>&1
cat $O/time_wgrad_tt.txt | grep -v amdgpu.ids
SL_WGRAD_TR=1 timeout 900 python -m pytest tests/test_train_models_gpu.py -x -q -m gpu > $O/pytest_models_tr.txt 2>&1
tail -3 $O/pytest_models_tr.txt
bash tools/exp_kd_env_ab.sh r05_u/kd_ab SL_WGRAD_TR=1
cat $O/kd_ab.txt
