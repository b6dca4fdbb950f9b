#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_k; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/tools/time_attn_bwd.py > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/attn_bwd_kernel_stats.csv; rm -rf $O/prof
cut -c1-200 $O/attn_bwd_kernel_stats.csv | head -20
