#!/bin/bash
# round 6, k: counter evidence for what rounds 5 and 6 changed.  One --pmc pass each (SQ: matrix-core busy, VALU issue, stalls,
# LDS bank conflicts), --kernel-trace only beside it:
#   * a 16-sample KD window (attn_bwd_dkdv / dq, gemm_tiled_tt_kernel with the bias rider, splitk_reduce_kernel, the fused 256-tile epilogues)
#   * the encoder pass (256 x 10 s)
#   * the decode probes at 1 024 and 2 048 rows
# and the FETCH_SIZE / WRITE_SIZE passes of the two decode kernels (the bench line's roofline.traffic reads the newest committed file).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_k; mkdir -p $O
C="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT"
rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/sq_kd -- python3 $R/tools/kd_window_trace.py > $O/sq_kd.log 2>&1
python3 $R/tools/pmc_kernels.py $O/sq_kd $O/r06_sq_kd_window.json > $O/r06_sq_kd_window.txt 2>&1
rm -rf $O/sq_kd
rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/sq_enc -- python3 $R/tools/prof_encoder.py 256 > $O/sq_enc.log 2>&1
python3 $R/tools/pmc_kernels.py $O/sq_enc $O/r06_sq_encoder.json > $O/r06_sq_encoder.txt 2>&1
rm -rf $O/sq_enc
for B in 1024 2048; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/sq_dec$B -- python3 $R/tools/probe_decode_kernels.py $B > $O/sq_dec$B.log 2>&1
  python3 $R/tools/pmc_kernels.py $O/sq_dec$B $O/r06_sq_decode_$B.json > $O/r06_sq_decode_$B.txt 2>&1
  rm -rf $O/sq_dec$B
done
rm -f $O/r06_pmc_decode_kernels.json
for B in 512 1024; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${B}_$c -- python3 $R/tools/probe_decode_kernels.py $B > /dev/null 2>&1
  done
  python3 $R/tools/pmc_decode.py $O/pmc_${B}_FETCH_SIZE $O/pmc_${B}_WRITE_SIZE $B $O/r06_pmc_decode_kernels.json > $O/pmc_decode_$B.txt 2>&1
  rm -rf $O/pmc_${B}_FETCH_SIZE $O/pmc_${B}_WRITE_SIZE
done
cd $R; for f in r06_sq_kd_window r06_sq_encoder r06_sq_decode_1024 r06_sq_decode_2048; do echo "== $f"; cat $O/$f.txt | cut -c1-170; done; cat $O/pmc_decode_*.txt | cut -c1-300; tail -3 $O/sq_kd.log
