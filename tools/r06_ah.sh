#!/bin/bash
# round 6, ah: two-stage 128-tile kernel with the next slab's DMA requests between the MFMAs (SL_GLDS_DMAB): bit equality, KD-window rows vs vendor
# (run as pb = burst, p = interleaved when the switch defaulted to 1; today: p = burst, pd = interleaved), KD windows A/B in one process
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_ah; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "ring or split or stream_k" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 > $O/pytest_gemm.txt
cat $O/pytest_gemm.txt
timeout 900 python tools/gemm_vs_vendor.py --mid --rounds 3 --variants pb,p,sk,vendor 2>&1 | grep -v amdgpu.ids > $O/gemm_vs_vendor_mid.txt
cat $O/gemm_vs_vendor_mid.txt
: > $O/kd_windows.txt
timeout 600 python tools/kd_ab_inproc.py SL_GLDS_DMAB=0 5 2 2>&1 | grep "window of" >> $O/kd_windows.txt
timeout 600 python tools/kd_ab_inproc.py SL_GLDS_DMAB=0 4 16 2>&1 | grep "window of" >> $O/kd_windows.txt
cat $O/kd_windows.txt
