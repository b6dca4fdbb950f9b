#!/bin/bash
# round 5, GPU call B: full parity suite, mid-M tile-choice A/B against the vendor library, PMC passes of the decode kernels the graph launches, kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -60 > $O/r05_b_pytest.txt
timeout 900 python tools/gemm_vs_vendor.py --rounds 3 --variants pad0,p,vendor > $O/r05_b_gemm_vs_vendor.txt 2>&1
bash tools/exp_pmc_decode.sh r05 > $O/r05_b_pmc.log 2>&1
bash tools/exp_prof.sh r05_b > $O/r05_b_prof.log 2>&1
tail -8 $O/r05_b_pytest.txt
tail -24 $O/r05_b_gemm_vs_vendor.txt
cat $O/r05_pmc_decode_kernels.json | head -40
