#!/bin/bash
# round 6, m: whole GPU parity suite, smoke, then the default bench line (wall-clocked) on the round's code
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_m; mkdir -p $O
T0=$(date +%s)
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $O/pytest_gpu.txt
T1=$(date +%s)
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
python bench.py > $O/bench_default_line.json 2> $O/bench_default.err
T2=$(date +%s)
echo "gpu suite wall seconds: $((T1 - T0)); smoke + default bench wall seconds: $((T2 - T1))" > $O/wall.txt
cat $O/pytest_gpu.txt $O/smoke.txt $O/wall.txt; tail -2 $O/bench_default.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06_m/bench_default_line.json').read().splitlines() if l.startswith('{')][-1])
print(d['value'], d['graded'])
print('kd', d['kd_step']['samples_per_s'], d['kd_step']['window_ms'], 'per-rank window', d['kd_per_rank_regime_probe']['window_ms'], 'b1', d['latency_b1']['decode_tokens_per_s'])
print('eos', d['eos_stop_mix']['compacted']['useful_tokens_per_s'], d['eos_stop_mix']['ids_identical_compacted_vs_uncompacted'], 'whisper', d['whisper_pipeline']['tokens_per_s'], 'mix', d['devclean_length_mix']['tokens_per_s'])
print('roofline', d['roofline']); print('cpu', {k: d['cpu_baseline'][k] for k in ('value','unit','cores','kind')})
PY
