#!/bin/bash
# A/B of the encoder GEMM epilogue forms: LayerNorm fold on / off (SL_NO_LN_FOLD=1) x swapped-operand register epilogue on / off
# (SL_NO_SWAP_EPILOGUE=1), same box, alternating runs; then the per-kernel table of the default form.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for i in 1 2 3; do
  echo "default        $(python3 $R/tools/prof_encoder.py 512)"
  echo "no-fold        $(SL_NO_LN_FOLD=1 python3 $R/tools/prof_encoder.py 512)"
  echo "no-swap        $(SL_NO_SWAP_EPILOGUE=1 python3 $R/tools/prof_encoder.py 512)"
  echo "no-fold no-swap $(SL_NO_LN_FOLD=1 SL_NO_SWAP_EPILOGUE=1 python3 $R/tools/prof_encoder.py 512)"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fold_on -o fold -- python3 $R/tools/prof_encoder.py 512 > $O/fold_on.log 2>&1
