#!/bin/bash
# round 6, h: GEMM shapes of the 16-sample KD window (per-shape time and rate), and the bench's KD leg + headline after the asynchronous uploads
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_h; mkdir -p $O
timeout 900 python tools/kd_gemm_shapes.py > $O/kd_window16_gemm_shapes.txt 2>&1
timeout 900 python bench.py --no-cpu-baseline --no-length-mix --no-extra-legs --no-eos-leg --steps 4 > $O/bench.json 2> $O/bench.err
head -42 $O/kd_window16_gemm_shapes.txt | cut -c1-175
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06_h/bench.json').read().splitlines() if l.startswith('{')][-1])
print(d['value'], d['kd_step']['samples_per_s'], d['kd_step']['window_ms'], d['kd_per_rank_regime_probe']['window_ms'], d['graded'])
print(d['stage_ms_one_batch_alone'] if 'stage_ms_one_batch_alone' in d else '')
PY
