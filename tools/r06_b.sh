#!/bin/bash
# round 6, b: the GEMM epilogue fusions (ABI 7) against the unfused launches; KD window ops with the round-5 tape (the "before" of the fusions)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_b; mkdir -p $O
timeout 900 python -m pytest tests/test_train_kernels_gpu.py -x -q -m gpu -k "epilogue" 2>&1 | tail -25 > $O/pytest_epilogue.txt
KD_WINDOW=2 timeout 600 python tools/prof_kd_ops.py > $O/kd_window2_ops_before.txt 2>&1
timeout 600 python tools/prof_kd_ops.py > $O/kd_window16_ops_before.txt 2>&1
cat $O/pytest_epilogue.txt
for f in kd_window2_ops_before kd_window16_ops_before; do grep -v "^\[W\|Warning\|_warn" $O/$f.txt | cut -c1-52,150-215 | head -44; tail -3 $O/$f.txt; done
