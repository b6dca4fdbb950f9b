#!/bin/bash
# round 6, az: the positional conv's weight gradient (16 groups x 64 output rows per utterance) on the token-major kernel, batch index on blockIdx.z
# (SL_TT_BATCHED; before: 16 launches per window of the register-staged loader at 84 TF/s): training parity suites, KD windows A/B in one process
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_az; mkdir -p $O
timeout 1500 python -m pytest tests/test_train_models_gpu.py tests/test_train_kernels_gpu.py tests/test_dp_gpu.py -q -m gpu -rf 2>&1 | grep -E "passed|failed|error|^FAILED" | tail -8 > $O/pytest_train.txt
cat $O/pytest_train.txt
: > $O/kd_windows.txt
timeout 600 python tools/kd_ab_inproc.py SL_TT_BATCHED=0 5 16 2>&1 | grep "window of" >> $O/kd_windows.txt
timeout 600 python tools/kd_ab_inproc.py SL_TT_BATCHED=0 5 2 2>&1 | grep "window of" >> $O/kd_windows.txt
cat $O/kd_windows.txt
