#!/bin/bash
# round 6, e: attention backward with two 16-row fragments per wave (A/B against SL_ATTN_BWD_KF=1 in one box); KD windows fused / unfused in one process
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_e; mkdir -p $O
timeout 900 python -m pytest tests/test_train_kernels_gpu.py -q --tb=short -k "flash_attention" 2>&1 | tail -4 > $O/pytest_attn_bwd.txt
for i in 1 2; do for v in 1 2; do echo "SL_ATTN_BWD_KF=$v"; SL_ATTN_BWD_KF=$v python tools/time_attn_bwd.py 2>&1 | grep "TF/s"; done; done > $O/attn_bwd_ab.txt
python tools/kd_ab_inproc.py SL_TAPE_FUSE=0 5 2 2>&1 | grep "window of" > $O/kd_fuse_ab.txt
python tools/kd_ab_inproc.py SL_TAPE_FUSE=0 4 16 2>&1 | grep "window of" >> $O/kd_fuse_ab.txt
python tools/kd_ab_inproc.py SL_ATTN_BWD_KF=1 4 16 2>&1 | grep "window of" >> $O/kd_fuse_ab.txt
cat $O/pytest_attn_bwd.txt $O/attn_bwd_ab.txt $O/kd_fuse_ab.txt
