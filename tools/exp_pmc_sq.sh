#!/bin/bash
# SQ counters (matrix-core busy, VALU issue, stalls, LDS conflicts) per kernel: the encoder pass and the decode probes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
C="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT"
rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/sq_enc -- python3 $R/tools/prof_encoder.py 256 > $O/${1:-r04}_sq_enc.log 2>&1
python3 $R/tools/pmc_kernels.py $O/sq_enc $O/${1:-r04}_sq_encoder.json >> $O/${1:-r04}_sq_enc.log 2>&1
rm -rf $O/sq_enc
rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/sq_dec -- python3 $R/tools/probe_decode_kernels.py 1024 > $O/${1:-r04}_sq_dec.log 2>&1
python3 $R/tools/pmc_kernels.py $O/sq_dec $O/${1:-r04}_sq_decode.json >> $O/${1:-r04}_sq_dec.log 2>&1
rm -rf $O/sq_dec
