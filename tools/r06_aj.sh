#!/bin/bash
# round 6, aj: ring form with a slab's DMA requests spread over two 16-MFMA phases (one per four MFMAs; SL_GLDS_RING=4) against all eight inside
# k-step 1 (204): bit equality, M <= 2 048 rows vs vendor, per-rank KD window A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_aj; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "ring or split or stream_k" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 > $O/pytest_gemm.txt
cat $O/pytest_gemm.txt
timeout 900 python tools/gemm_vs_vendor.py --small --rounds 3 --variants p2,po,p,sk,vendor 2>&1 | grep -v amdgpu.ids > $O/gemm_vs_vendor_small.txt
cat $O/gemm_vs_vendor_small.txt
: > $O/kd_windows.txt
timeout 600 python tools/kd_ab_inproc.py SL_GLDS_RING=204 6 2 2>&1 | grep "window of" >> $O/kd_windows.txt
cat $O/kd_windows.txt
