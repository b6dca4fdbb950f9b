#!/bin/bash
# round 6, w: the per-element mixer of the dropout mask on 24-bit multiplies (dropmix24): every test that replays a mask on the host, the training parity
# suites, attention kernels with dropout timed, KD windows
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_w; mkdir -p $O
timeout 1800 python -m pytest tests/test_train_models_gpu.py tests/test_train_kernels_gpu.py tests/test_kernels_gpu.py tests/test_dp_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tail -5 > $O/pytest.txt
python tools/time_attn_bwd.py 2>&1 | grep "TF/s" > $O/attn_bwd.txt
python tools/kd_ab_inproc.py SL_TAPE_FUSE=0 3 16 2>&1 | grep "window of" > $O/kd_windows.txt
python tools/kd_ab_inproc.py SL_TAPE_FUSE=0 3 2 2>&1 | grep "window of" >> $O/kd_windows.txt
cat $O/pytest.txt $O/attn_bwd.txt $O/kd_windows.txt
