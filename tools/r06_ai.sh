#!/bin/bash
# round 6, ai: per-kernel census of the KD windows (2 and 16 samples) on the round's final GEMM code (ring form on), torch profiler, one window each
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_ai; mkdir -p $O
KD_WINDOW=2 timeout 600 python tools/prof_kd_ops.py 2>&1 | grep -v amdgpu.ids | cut -c1-200 > $O/kd_window2_ops.txt
KD_WINDOW=16 timeout 600 python tools/prof_kd_ops.py 2>&1 | grep -v amdgpu.ids | cut -c1-200 > $O/kd_window16_ops.txt
tail -60 $O/kd_window2_ops.txt | cut -c1-150
