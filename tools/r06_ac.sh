#!/bin/bash
# round 6, ac: AdamW of the finished gradient buckets beside the rest of the backward (KDTrainer._early_step, SL_KD_OVERLAP_OPT): training
# parity suites (incl. the 2-process data-parallel equivalence), KD windows overlapped / step behind the backward in one process
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_ac; mkdir -p $O
timeout 1500 python -m pytest tests/test_train_models_gpu.py tests/test_train_kernels_gpu.py tests/test_dp_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 > $O/pytest_train.txt
python tools/kd_ab_inproc.py SL_KD_OVERLAP_OPT=0 6 2 2>&1 | grep "window of" > $O/kd_windows.txt
python tools/kd_ab_inproc.py SL_KD_OVERLAP_OPT=0 4 16 2>&1 | grep "window of" >> $O/kd_windows.txt
cat $O/pytest_train.txt $O/kd_windows.txt
