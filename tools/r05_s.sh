#!/bin/bash
# round 5, run s: in-process KD window A/B of SL_WGRAD_TR (full window and the 2-sample per-rank window)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_s; mkdir -p $O
timeout 600 python tools/kd_ab_inproc.py SL_WGRAD_TR=1 8 > $O/ab_full.txt 2> $O/ab_full.err
timeout 600 python tools/kd_ab_inproc.py SL_WGRAD_TR=1 12 2 > $O/ab_w2.txt 2> $O/ab_w2.err
tail -3 $O/ab_full.err; cat $O/ab_full.txt $O/ab_w2.txt
