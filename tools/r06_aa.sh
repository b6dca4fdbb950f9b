#!/bin/bash
# round 6, aa: conv0 backward with the block's four waves folded through LDS before the atomics (SL_CONV0_FOLD): training parity suites,
# KD windows folded / per-wave flush in one process
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_aa; mkdir -p $O
timeout 1500 python -m pytest tests/test_train_models_gpu.py tests/test_train_kernels_gpu.py tests/test_dp_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tail -5 > $O/pytest_train.txt
python tools/kd_ab_inproc.py SL_CONV0_FOLD=0 5 2 2>&1 | grep "window of" > $O/kd_windows.txt
python tools/kd_ab_inproc.py SL_CONV0_FOLD=0 4 16 2>&1 | grep "window of" >> $O/kd_windows.txt
cat $O/pytest_train.txt $O/kd_windows.txt
