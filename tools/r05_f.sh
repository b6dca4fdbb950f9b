#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_f; mkdir -p $O
timeout 1800 python -m pytest tests/test_fullsize_gpu.py -q --tb=short 2>&1 | tail -120 > $O/pytest_full.txt
tail -60 $O/pytest_full.txt
