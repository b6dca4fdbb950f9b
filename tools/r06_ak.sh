#!/bin/bash
# round 6, ak: ring form of the token-major weight-gradient kernel (gemm_tiled_tt_ring_kernel, SL_TT_RING): bit equality with the two-stage kernel,
# the window's weight-gradient shapes three ways in one process, KD windows A/B in one process
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_ak; mkdir -p $O
timeout 900 python -m pytest tests/test_train_kernels_gpu.py -x -q -m gpu -k "wgrad or split" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 > $O/pytest_wgrad.txt
cat $O/pytest_wgrad.txt
timeout 600 python tools/time_wgrad_ring.py 2>&1 | grep -v amdgpu.ids > $O/wgrad_ring.txt
cat $O/wgrad_ring.txt
: > $O/kd_windows.txt
timeout 600 python tools/kd_ab_inproc.py SL_TT_RING=0 5 2 2>&1 | grep "window of" >> $O/kd_windows.txt
timeout 600 python tools/kd_ab_inproc.py SL_TT_RING=0 4 16 2>&1 | grep "window of" >> $O/kd_windows.txt
cat $O/kd_windows.txt
