#!/bin/bash
# Round-4 encoder evidence: the encoder GEMM shapes with their epilogue forms under the round-3 loop and the staggered two-phase loop,
# then a kernel trace of the encoder pass (512 x 10 s).   usage: tools/exp_r04_enc.sh <tag>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-r04_enc}; cd /tmp; export TMPDIR=/tmp
{
  echo "=== tools/time_fold_epilogue.py, staggered two-phase loop (default)"; python3 $R/tools/time_fold_epilogue.py
  echo "=== SL_T256_PHASED=0 (round-3 loop)"; SL_T256_PHASED=0 python3 $R/tools/time_fold_epilogue.py
} 2>&1 | grep -v amdgpu.ids > $O/${T}_gemm_epilogue_forms.txt
rm -rf $O/${T}_trace
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_trace -- python3 $R/tools/prof_encoder.py 512 > $O/${T}_encoder.log 2>&1
f=$(find $O/${T}_trace -name "*kernel_stats.csv" | head -1); cp "$f" $O/${T}_encoder_kernel_stats.csv; rm -rf $O/${T}_trace
