#!/bin/bash
# round 6, al: evidence for the ring kernels — rocprofv3 kernel timeline of the per-rank KD window (kernel count, span, idle) and one SQ counter pass of
# the same window (gemm_tiled_ring_kernel / gemm_tiled_tt_ring_kernel: matrix-core busy, VALU issue, stalls, LDS bank conflicts); --kernel-trace only beside --pmc
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_al; mkdir -p $O
export KD_WINDOW=2
bash $R/tools/exp_kd_trace.sh r06_al 2
cp $R/gpurun_out/r06_al_kd_timeline.txt $O/kd_window2_timeline.txt
C="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT"
cd /tmp
rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/sq_kd2 -- python3 $R/tools/kd_window_trace.py > $O/sq_kd2.log 2>&1
python3 $R/tools/pmc_kernels.py $O/sq_kd2 $O/r06_sq_kd_window2_ring.json > $O/r06_sq_kd_window2_ring.txt 2>&1
rm -rf $O/sq_kd2
cd $R; head -30 $O/kd_window2_timeline.txt | cut -c1-200; cat $O/r06_sq_kd_window2_ring.txt | cut -c1-200 | head -40
