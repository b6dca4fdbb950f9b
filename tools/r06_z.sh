#!/bin/bash
# round 6, closing run: whole GPU parity suite, smoke, the default bench line (wall-clocked), and the sequential kernel-stats profile of the same code
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out; T=${1:-r06_z}
T0=$(date +%s)
timeout 2400 python -m pytest tests -m gpu -q -rf 2>&1 | grep -E "passed|failed|error|^FAILED" | tail -6 > $O/${T}_pytest_gpu.txt
T1=$(date +%s)
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i "smoke" > $O/${T}_smoke.txt
python bench.py > $O/${T}_bench_default_line.json 2> $O/${T}_bench_default.err
T2=$(date +%s)
echo "gpu suite wall seconds: $((T1 - T0)); smoke + default bench wall seconds: $((T2 - T1))" > $O/${T}_wall.txt
bash tools/exp_prof.sh $T > $O/${T}_prof.log 2>&1
cd "$GRAFT_REPO_ROOT"
cat $O/${T}_pytest_gpu.txt $O/${T}_smoke.txt $O/${T}_wall.txt
python - $T <<'PY'
import json, sys, csv
T = sys.argv[1]
d=json.loads([l for l in open(f'gpurun_out/{T}_bench_default_line.json').read().splitlines() if l.startswith('{')][-1])
print(d['value'], d['graded'])
print('kd', d['kd_step']['samples_per_s'], d['kd_step']['window_ms'], 'per-rank window', d['kd_per_rank_regime_probe']['window_ms'], 'b1', d['latency_b1']['decode_tokens_per_s'])
print('eos', d['eos_stop_mix']['compacted']['useful_tokens_per_s'], d['eos_stop_mix']['ids_identical_compacted_vs_uncompacted'], 'whisper', d['whisper_pipeline']['tokens_per_s'], 'mix', d['devclean_length_mix']['tokens_per_s'])
print('roofline', {k: d['roofline'][k] for k in ('frac', 'achieved', 'avg_launch_us', 'traffic', 'traffic_source')})
for r in csv.DictReader(open(f'gpurun_out/{T}_kernel_stats.csv')):
    if 'attn_decode_full_kernel' in r['Name']: print('rocprof', r['Name'][:60], r['Calls'], 'avg ns', r['AverageNs'])
PY
