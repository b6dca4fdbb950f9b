#!/bin/bash
# round 6, ap: how many tiles the ring form of the 128-tile kernel should take (SL_GLDS_RING_MAX_TILES: above 256 its blocks run in two rounds):
# M <= 2 048 rows + the KD-window rows vs vendor at 256 / 320 / 384 / 512, per-rank KD window A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_ap; mkdir -p $O
timeout 900 python tools/gemm_vs_vendor.py --mid --rounds 3 --variants p,m320,m384,m512,vendor 2>&1 | grep -v amdgpu.ids > $O/gemm_vs_vendor_ring_max_tiles.txt
cat $O/gemm_vs_vendor_ring_max_tiles.txt
: > $O/kd_windows.txt
timeout 600 python tools/kd_ab_inproc.py SL_GLDS_RING_MAX_TILES=384 5 2 2>&1 | grep "window of" >> $O/kd_windows.txt
timeout 600 python tools/kd_ab_inproc.py SL_GLDS_RING_MAX_TILES=384 4 16 2>&1 | grep "window of" >> $O/kd_windows.txt
cat $O/kd_windows.txt
