#!/bin/bash
# round 5, run v: LayerNorm backward without register spills (LEAN form) — kernel timing old / new library, its tests, KD leg A/B
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_v; mkdir -p $O
timeout 900 python -m pytest tests/test_train_kernels_gpu.py tests/test_train_models_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
for v in new base new base; do
  if [ $v = base ]; then export SL_DEV=1 SL_LIB_PATH=$PWD/tools/ab/libspeechllm_base.so; else unset SL_LIB_PATH SL_DEV; fi
  echo "== $v" >> $O/time_lnbwd.txt
  timeout 300 python tools/time_lnbwd.py >> $O/time_lnbwd.txt 2>/dev/null
done
unset SL_LIB_PATH SL_DEV
cat $O/time_lnbwd.txt
bash tools/exp_kd_ab.sh r05_v/kd_ab
cat $O/kd_ab.txt
