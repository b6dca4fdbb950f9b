#!/bin/bash
# round 6, g: the KD window after the blocking uploads became pinned asynchronous copies (the host profile showed 19 stream-draining torch.tensor(...,
# device) calls per window); fused against unfused tapes again, host profile again
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_g; mkdir -p $O
KD_WINDOW=2 python tools/kd_host_profile.py > $O/kd_host_profile_w2.txt 2>&1
python tools/kd_ab_inproc.py SL_TAPE_FUSE=0 5 2 2>&1 | grep "window of" > $O/kd_fuse_ab.txt
python tools/kd_ab_inproc.py SL_TAPE_FUSE=0 4 16 2>&1 | grep "window of" >> $O/kd_fuse_ab.txt
timeout 1500 python -m pytest tests/test_train_models_gpu.py tests/test_dp_gpu.py -x -q -m gpu 2>&1 | tail -4 > $O/pytest_train.txt
head -12 $O/kd_host_profile_w2.txt | cut -c1-150; cat $O/kd_fuse_ab.txt $O/pytest_train.txt
