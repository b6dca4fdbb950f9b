#!/bin/bash
# round 6, q: one join per encoder layer (two copies of the buffers the parameter-gradient groups read): training parity suites, KD windows, timeline
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_q; mkdir -p $O
timeout 1500 python -m pytest tests/test_train_models_gpu.py tests/test_train_kernels_gpu.py tests/test_dp_gpu.py -x -q -m gpu 2>&1 | tail -4 > $O/pytest_train.txt
python tools/kd_ab_inproc.py SL_TAPE_FUSE=0 3 16 2>&1 | grep "window of" > $O/kd_windows.txt
python tools/kd_ab_inproc.py SL_NO_WGRAD_STREAM=1 3 16 2>&1 | grep "window of" >> $O/kd_windows.txt
bash tools/exp_kd_trace.sh r06_q > /dev/null 2>&1
cat $O/pytest_train.txt $O/kd_windows.txt; head -8 gpurun_out/r06_q_kd_timeline.txt
