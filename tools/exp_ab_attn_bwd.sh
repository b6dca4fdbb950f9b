#!/bin/bash
# A/B of two library builds on the attention backward: tools/ab/libspeechllm_{base,new}.so (SL_DEV=1 lets _lib.py honour SL_LIB_PATH)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_k; mkdir -p $O
timeout 900 python -m pytest tests/test_train_kernels_gpu.py -q --tb=short -k "flash_attention" 2>&1 | tail -6 > $O/pytest_attn_bwd.txt
for i in 1 2; do for v in base new; do echo "lib=$v"; SL_DEV=1 SL_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/libspeechllm_$v.so python tools/time_attn_bwd.py 2>&1 | grep "TF/s"; done; done > $O/attn_bwd_ab.txt
cat $O/pytest_attn_bwd.txt $O/attn_bwd_ab.txt
