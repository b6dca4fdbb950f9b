#!/bin/bash
# round 5, run p: LayerNorm-backward block size (SL_LNBWD_NW) and token-major wgrad K runs (SL_TT_MAX_SPLITS): alone and inside KD windows
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_p; mkdir -p $O
timeout 300 python -m pytest tests/test_train_kernels_gpu.py -x -q -m gpu -k "layernorm_backward" > $O/pytest16.txt 2>&1; tail -1 $O/pytest16.txt
SL_LNBWD_NW=8 timeout 300 python -m pytest tests/test_train_kernels_gpu.py -x -q -m gpu -k "layernorm_backward" > $O/pytest8.txt 2>&1; tail -1 $O/pytest8.txt
SL_LNBWD_NW=4 timeout 300 python -m pytest tests/test_train_kernels_gpu.py -x -q -m gpu -k "layernorm_backward" > $O/pytest4.txt 2>&1; tail -1 $O/pytest4.txt
for nw in 16 8 4; do echo "== SL_LNBWD_NW=$nw" >> $O/alone.txt; SL_LNBWD_NW=$nw timeout 200 python tools/time_lnbwd.py 2>/dev/null | grep "gelu=0" >> $O/alone.txt; done
for sp in 8 4 2; do echo "== SL_TT_MAX_SPLITS=$sp" >> $O/alone.txt; SL_TT_MAX_SPLITS=$sp timeout 200 python tools/time_wgrad_tt.py 2>/dev/null | grep tokens= | cut -c1-140 >> $O/alone.txt; done
cat $O/alone.txt
for sw in SL_LNBWD_NW=8 SL_LNBWD_NW=4 SL_TT_MAX_SPLITS=4 SL_TT_MAX_SPLITS=2; do
  echo "== $sw" >> $O/kd.txt
  timeout 400 python tools/kd_ab_inproc.py $sw 6 >> $O/kd.txt 2>/dev/null
done
cat $O/kd.txt
