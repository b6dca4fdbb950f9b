#!/bin/bash
# round 6, c: the fused KD tapes: parity suites of the training path, then the KD window profiles (2 / 16 samples) after the fusions
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_c; mkdir -p $O
timeout 1500 python -m pytest tests/test_train_models_gpu.py tests/test_train_kernels_gpu.py tests/test_dp_gpu.py -x -q -m gpu 2>&1 | tail -25 > $O/pytest_train.txt
KD_WINDOW=2 timeout 600 python tools/prof_kd_ops.py > $O/kd_window2_ops_after.txt 2>&1
timeout 600 python tools/prof_kd_ops.py > $O/kd_window16_ops_after.txt 2>&1
cat $O/pytest_train.txt
for f in kd_window2_ops_after kd_window16_ops_after; do grep -v "^\[W\|Warning\|_warn" $O/$f.txt | cut -c1-52,150-215 | head -44; tail -3 $O/$f.txt; done
