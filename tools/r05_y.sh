#!/bin/bash
# round 5, closing run: whole parity suite, default bench line (wall-clocked), sequential kernel-stats profile
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -40 > $O/r05_y_pytest.txt
T0=$(date +%s)
python bench.py > $O/r05_y_default_line.json 2> $O/r05_y_default.err
T1=$(date +%s)
echo "default bench wall seconds: $((T1 - T0))" > $O/r05_y_wall.txt
bash tools/exp_prof.sh r05_y > $O/r05_y_prof.log 2>&1
tail -4 $O/r05_y_pytest.txt; cat $O/r05_y_wall.txt
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05_y_default_line.json').read().splitlines() if l.startswith('{')][-1])
print(d['value'], d['graded'], d['kd_step']['samples_per_s'], d['kd_per_rank_regime_probe']['window_ms'], d['eos_stop_mix']['compacted']['useful_tokens_per_s'], d['latency_b1']['decode_tokens_per_s'], d['whisper_pipeline']['tokens_per_s'], d['devclean_length_mix']['tokens_per_s'])
PY
