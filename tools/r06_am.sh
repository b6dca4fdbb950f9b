#!/bin/bash
# round 6, am: LayerNorm backward with the column reduce inside the kernel (last-arriving block; SL_LN_COLRED_INKERNEL): training parity suites, KD windows
# A/B in one process, kernel timeline of the per-rank window (kernel count)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_am; mkdir -p $O
timeout 1500 python -m pytest tests/test_train_models_gpu.py tests/test_train_kernels_gpu.py tests/test_dp_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 > $O/pytest_train.txt
cat $O/pytest_train.txt
: > $O/kd_windows.txt
timeout 600 python tools/kd_ab_inproc.py SL_LN_COLRED_INKERNEL=0 5 2 2>&1 | grep "window of" >> $O/kd_windows.txt
timeout 600 python tools/kd_ab_inproc.py SL_LN_COLRED_INKERNEL=0 4 16 2>&1 | grep "window of" >> $O/kd_windows.txt
cat $O/kd_windows.txt
bash tools/exp_kd_trace.sh r06_am 2
cd "$GRAFT_REPO_ROOT"; head -8 gpurun_out/r06_am_kd_timeline.txt | cut -c1-160; cp gpurun_out/r06_am_kd_timeline.txt $O/kd_window2_timeline.txt
