#!/bin/bash
# round 6, ab: the encoder layers' transposed weights made in one batched launch beside the forward (SL_ENC_WT_AHEAD): training parity suites,
# KD windows ahead / a transpose per product in one process
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_ab; mkdir -p $O
timeout 1500 python -m pytest tests/test_train_models_gpu.py tests/test_train_kernels_gpu.py tests/test_dp_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tail -5 > $O/pytest_train.txt
python tools/kd_ab_inproc.py SL_ENC_WT_AHEAD=0 5 2 2>&1 | grep "window of" > $O/kd_windows.txt
python tools/kd_ab_inproc.py SL_ENC_WT_AHEAD=0 4 16 2>&1 | grep "window of" >> $O/kd_windows.txt
cat $O/pytest_train.txt $O/kd_windows.txt
