#!/bin/bash
# round 6, aw: rocprofv3 kernel timeline of the per-rank KD window on the final code (kernel count after the one-launch attention backward)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r06_aw
bash tools/exp_kd_trace.sh r06_aw 2
cd "$GRAFT_REPO_ROOT"; cp gpurun_out/r06_aw_kd_timeline.txt gpurun_out/r06_aw/kd_window2_timeline.txt; head -12 gpurun_out/r06_aw_kd_timeline.txt | cut -c1-170; tail -5 gpurun_out/r06_aw_kd_timeline.txt
