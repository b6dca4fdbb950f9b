#!/bin/bash
# round 6, l: head_dim-64 attention forward with 128-key staged blocks (default) against 64 (SL_ATTN_FWD_ST=1), alternating; encoder pass both ways;
# the KD window's SQ counters again with the anonymous-namespace kernels named
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06_l; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "attn or attention" 2>&1 | tail -3 > $O/pytest_attn.txt
for i in 1 2; do for v in 1 2; do echo "SL_ATTN_FWD_ST=$v"; SL_ATTN_FWD_ST=$v python tools/bench_attn.py 2>&1 | grep "hubert" | cut -c1-110; SL_ATTN_FWD_ST=$v python tools/prof_encoder.py 256 2>&1 | grep "encode ms"; done; done > $O/attn_fwd_st_ab.txt
cat $O/pytest_attn.txt $O/attn_fwd_st_ab.txt
cd /tmp
C="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT"
rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/sq_kd -- python3 $GRAFT_REPO_ROOT/tools/kd_window_trace.py > $O/sq_kd.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_kernels.py $O/sq_kd $O/r06_sq_kd_window.json > $O/r06_sq_kd_window.txt 2>&1
rm -rf $O/sq_kd
python3 - <<'PY'
import json, os
d = json.load(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r06_l/r06_sq_kd_window.json"))
for k, r in d.items():
    if any(s in k for s in ("attn_bwd", "gemm_tiled_tt", "splitk_reduce", "gemm_tiled256p", "gemm_tiled_glds", "attn_fwd")):
        print(f"{r['share_of_kernel_time']:6.3f} mfma_busy {r.get('mfma_busy', 0):5.3f} valu {r.get('valu_issue_per_wave_cycle', 0):5.3f} stalled {r.get('issue_stalled_per_wave_cycle', 0):5.3f} lds_conflict_cyc {r.get('lds_bank_conflict_cycles', 0):9d} x{r['launches']:5d} {k[:80]}")
PY
