#!/bin/bash
# round 6, r: split-K partials summed inside the norm backward (deferred_splits) + RMSNorm backward with one row per wave at small row counts:
# training parity suites, KD windows fused / unfused in one process
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_r; mkdir -p $O
timeout 1500 python -m pytest tests/test_train_models_gpu.py tests/test_train_kernels_gpu.py tests/test_dp_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tail -5 > $O/pytest_train.txt
python tools/kd_ab_inproc.py SL_TAPE_FUSE=0 4 2 2>&1 | grep "window of" > $O/kd_windows.txt
python tools/kd_ab_inproc.py SL_TAPE_FUSE=0 3 16 2>&1 | grep "window of" >> $O/kd_windows.txt
KD_WINDOW=2 timeout 600 python tools/prof_kd_ops.py > $O/kd_window2_ops.txt 2>&1
cat $O/pytest_train.txt $O/kd_windows.txt; grep -v "^\[W\|Warning\|_warn" $O/kd_window2_ops.txt | cut -c1-52,150-215 | head -16; tail -2 $O/kd_window2_ops.txt
