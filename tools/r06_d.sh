#!/bin/bash
# round 6, d: bias gradients riding on the token-major weight-gradient product; training parity suites; KD window profiles; default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_d; mkdir -p $O
timeout 1500 python -m pytest tests/test_train_models_gpu.py tests/test_train_kernels_gpu.py tests/test_dp_gpu.py -x -q -m gpu 2>&1 | tail -15 > $O/pytest_train.txt
KD_WINDOW=2 timeout 600 python tools/prof_kd_ops.py > $O/kd_window2_ops.txt 2>&1
timeout 600 python tools/prof_kd_ops.py > $O/kd_window16_ops.txt 2>&1
timeout 900 python bench.py --no-cpu-baseline --no-length-mix --no-extra-legs --no-eos-leg --steps 2 > $O/bench_kd.json 2> $O/bench_kd.err
cat $O/pytest_train.txt
for f in kd_window2_ops kd_window16_ops; do grep -v "^\[W\|Warning\|_warn" $O/$f.txt | cut -c1-52,150-215 | head -30; tail -3 $O/$f.txt; done
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06_d/bench_kd.json').read().splitlines() if l.startswith('{')][-1])
print(d['value'], d['kd_step']['samples_per_s'], d['kd_step']['window_ms'], d['kd_per_rank_regime_probe']['window_ms'], d['graded'])
PY
