#!/bin/bash
# round 6, ad: the overlapped optimizer step on a CU-masked stream (SL_KD_OPT_CUS CUs per XCD) against the step behind the backward, in one process
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_ad; mkdir -p $O
timeout 900 python -m pytest tests/test_train_models_gpu.py -x -q -m gpu -k "overlapped" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 > $O/pytest_train.txt
: > $O/kd_windows.txt
for n in 4 8 16; do
  echo "SL_KD_OPT_CUS=$n" >> $O/kd_windows.txt
  SL_KD_OPT_CUS=$n python tools/kd_ab_inproc.py SL_KD_OVERLAP_OPT=1 5 2 2>&1 | grep "window of" >> $O/kd_windows.txt
done
echo "SL_KD_OPT_CUS=8, 16 samples" >> $O/kd_windows.txt
SL_KD_OPT_CUS=8 python tools/kd_ab_inproc.py SL_KD_OVERLAP_OPT=1 3 16 2>&1 | grep "window of" >> $O/kd_windows.txt
cat $O/pytest_train.txt $O/kd_windows.txt
