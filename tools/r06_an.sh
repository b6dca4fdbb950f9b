#!/bin/bash
# round 6, an: 256-tile products of 156 tiles cut into three uneven K runs left to the RMSNorm backward (SL_SPLIT_K256): kernel tests, training parity suites,
# KD windows A/B in one process
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_an; mkdir -p $O
timeout 900 python -m pytest tests/test_train_kernels_gpu.py -x -q -m gpu -k "k_runs or deferred or split" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 > $O/pytest_kruns.txt
cat $O/pytest_kruns.txt
: > $O/kd_windows.txt
timeout 600 python tools/kd_ab_inproc.py SL_SPLIT_K256=0 5 16 2>&1 | grep "window of" >> $O/kd_windows.txt
cat $O/kd_windows.txt
timeout 1500 python -m pytest tests/test_train_models_gpu.py tests/test_dp_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 > $O/pytest_train.txt
cat $O/pytest_train.txt
