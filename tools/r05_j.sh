#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_j; mkdir -p $O
timeout 900 python -m pytest tests/test_models_gpu.py -q --tb=short 2>&1 | tail -15 > $O/pytest_models.txt
cat $O/pytest_models.txt
