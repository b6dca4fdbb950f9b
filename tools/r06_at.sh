#!/bin/bash
# round 6, at: attention backward with the dK / dV and dQ blocks in one launch (SL_ATTN_BWD_BOTH) where neither pass fills the chip: kernel + training parity
# suites, KD windows A/B in one process
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_at; mkdir -p $O
timeout 1500 python -m pytest tests/test_train_kernels_gpu.py tests/test_train_models_gpu.py tests/test_dp_gpu.py -q -m gpu -rf 2>&1 | grep -E "passed|failed|error|^FAILED" | tail -8 > $O/pytest_train.txt
cat $O/pytest_train.txt
: > $O/kd_windows.txt
timeout 600 python tools/kd_ab_inproc.py SL_ATTN_BWD_BOTH=0 6 2 2>&1 | grep "window of" >> $O/kd_windows.txt
timeout 600 python tools/kd_ab_inproc.py SL_ATTN_BWD_BOTH=0 4 16 2>&1 | grep "window of" >> $O/kd_windows.txt
cat $O/kd_windows.txt
