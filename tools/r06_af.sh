#!/bin/bash
# round 6, af: ring form of the 128-tile GEMM (gemm128.hip, SL_GLDS_RING): bit equality with the two-stage kernel, the M <= 2 048 rows against the
# vendor library (two-stage / ring 4 / ring 3, plain and with the split-K workspace), KD windows ring vs two-stage in one process
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_af; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 > $O/pytest_gemm.txt
cat $O/pytest_gemm.txt
timeout 900 python tools/gemm_vs_vendor.py --small --rounds 3 --variants p2,pu,p,p3,sk2,sk,vendor 2>&1 | grep -v amdgpu.ids > $O/gemm_vs_vendor_small.txt
cat $O/gemm_vs_vendor_small.txt
: > $O/kd_windows.txt
timeout 600 python tools/kd_ab_inproc.py SL_GLDS_RING=0 6 2 2>&1 | grep "window of" >> $O/kd_windows.txt
timeout 600 python tools/kd_ab_inproc.py SL_GLDS_RING=0 4 16 2>&1 | grep "window of" >> $O/kd_windows.txt
cat $O/kd_windows.txt
