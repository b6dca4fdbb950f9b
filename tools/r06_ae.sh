#!/bin/bash
# round 6, ae: RMSNorm backward in the lean form (operands kept as loaded: 150 registers, 3 waves per SIMD; SL_RMSBWD_LEAN): training parity suites,
# KD windows against the float-copy form (256 registers, 1 wave per SIMD) in one process
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_ae; mkdir -p $O
timeout 1500 python -m pytest tests/test_train_models_gpu.py tests/test_train_kernels_gpu.py tests/test_dp_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 > $O/pytest_train.txt
: > $O/kd_windows.txt
python tools/kd_ab_inproc.py SL_RMSBWD_LEAN=0 5 2 2>&1 | grep "window of" >> $O/kd_windows.txt
python tools/kd_ab_inproc.py SL_RMSBWD_LEAN=0 4 16 2>&1 | grep "window of" >> $O/kd_windows.txt
python tools/kd_ab_inproc.py SL_RMSBWD_LEAN=1 4 16 2>&1 | grep "window of" >> $O/kd_windows.txt
python tools/kd_ab_inproc.py SL_RMSBWD_LEAN=3 4 16 2>&1 | grep "window of" >> $O/kd_windows.txt
cat $O/pytest_train.txt $O/kd_windows.txt
