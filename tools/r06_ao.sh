#!/bin/bash
# round 6, ao: uneven 256-tile K runs also without deferred_splits (reduce launch: the unfused tape keeps the fused tape's bits): kernel tests + training parity suites
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_ao; mkdir -p $O
timeout 900 python -m pytest tests/test_train_kernels_gpu.py -x -q -m gpu -k "k_runs or deferred or split" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 > $O/pytest_kruns.txt
cat $O/pytest_kruns.txt
timeout 1500 python -m pytest tests/test_train_models_gpu.py tests/test_dp_gpu.py -q -m gpu 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 > $O/pytest_train.txt
cat $O/pytest_train.txt
