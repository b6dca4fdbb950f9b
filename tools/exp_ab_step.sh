# A/B of two library builds on one box: tools/exp_ab_step.sh [B ...]   (tools/ab/libspeechllm_{base,new}.so, SL_DEV=1 lets _lib.py honour SL_LIB_PATH)
for i in 1 2 3; do for v in base new; do echo "lib=$v"; SL_DEV=1 SL_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/libspeechllm_$v.so SHARED_PREFIX=9 python tools/time_decode_step.py "$@" 2>&1 | grep "B="; done; done
