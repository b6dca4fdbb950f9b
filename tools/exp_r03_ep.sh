#!/bin/bash
# Round-3 GEMM epilogue evidence: encoder GEMM shapes with each epilogue form, encoder A/B of the LayerNorm fold and the swapped-operand
# epilogue, and the KD window's GEMM shape census.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
{
  echo "=== tools/time_fold_epilogue.py (default: swapped-operand register epilogue where it applies)"; python3 $R/tools/time_fold_epilogue.py
  echo "=== SL_NO_SWAP_EPILOGUE=1 (LDS-turned rows epilogue, compile-time forms)"; SL_NO_SWAP_EPILOGUE=1 python3 $R/tools/time_fold_epilogue.py
  echo "=== tools/time_gelu_epilogue.py"; python3 $R/tools/time_gelu_epilogue.py
} 2>&1 | grep -v amdgpu.ids > $O/r03_ep_gemm_epilogue_forms.txt
bash $R/tools/exp_fold.sh 2>&1 | grep -v amdgpu.ids > $O/r03_ep_encoder_fold_ab.txt
f=$(find $O/fold_on -name "*kernel_stats.csv" | head -1); cp "$f" $O/r03_ep_encoder_kernel_stats.csv
python3 $R/tools/kd_gemm_shapes.py 2>&1 | grep -v amdgpu.ids > $O/r03_ep_kd_gemm_shapes.txt
