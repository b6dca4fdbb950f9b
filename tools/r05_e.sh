#!/bin/bash
# round 5, GPU call E: where the per-rank KD window (2 samples) spends its time after the split-K change; full parity suite; default line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_e; mkdir -p $O
KD_WINDOW=2 timeout 600 python tools/prof_kd_ops.py > $O/kd_window2_ops.txt 2>&1
KD_WINDOW=2 timeout 900 python tools/kd_gemm_shapes.py > $O/kd_window2_gemm_shapes.txt 2>&1
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/pytest.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -5 $O/pytest.txt
grep -v "^\[W\|Warning\|_warn" $O/kd_window2_ops.txt | cut -c1-52,150-215 | head -40
head -50 $O/kd_window2_gemm_shapes.txt
