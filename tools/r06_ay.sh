#!/bin/bash
# round 6, ay: the encoder stack's backward as ONE C call when no reducer listens for finished buckets (four layers per call before: every call boundary
# joins the parameter-gradient side stream): training parity suites, KD windows one call vs chunks of four in one process
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_ay; mkdir -p $O
timeout 1500 python -m pytest tests/test_train_models_gpu.py tests/test_dp_gpu.py -q -m gpu -rf 2>&1 | grep -E "passed|failed|error|^FAILED" | tail -6 > $O/pytest_train.txt
cat $O/pytest_train.txt
: > $O/kd_windows.txt
timeout 600 python tools/kd_ab_inproc.py PY_STACK_CHUNK=4 5 16 2>&1 | grep "window of" >> $O/kd_windows.txt
timeout 600 python tools/kd_ab_inproc.py PY_STACK_CHUNK=4 5 2 2>&1 | grep "window of" >> $O/kd_windows.txt
cat $O/kd_windows.txt
