#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_q; mkdir -p $O
timeout 600 python tools/prof_kd_ops.py > $O/kd_window_ops.txt 2>&1
grep -v "^\[W\|Warning\|_warn" $O/kd_window_ops.txt | cut -c1-52,150-215 | head -48
