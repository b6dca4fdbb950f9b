#!/bin/bash
# round 5, final measurements: the default bench line (wall-clocked) and the sequential kernel-stats profile of the same code
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
T0=$(date +%s)
python bench.py > $O/r05_z_default_line.json 2> $O/r05_z_default.err
T1=$(date +%s)
echo "default bench wall seconds: $((T1 - T0))" > $O/r05_z_wall.txt
bash tools/exp_prof.sh r05_z > $O/r05_z_prof.log 2>&1
cat $O/r05_z_wall.txt; tail -3 $O/r05_z_default.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05_z_default_line.json').read().splitlines() if l.startswith('{')][-1])
print(d['value'], d['graded'], d['kd_step']['samples_per_s'], d['kd_per_rank_regime_probe']['window_ms'], d['eos_stop_mix']['compacted']['useful_tokens_per_s'], d['latency_b1']['decode_tokens_per_s'], d['whisper_pipeline']['tokens_per_s'], d['devclean_length_mix']['tokens_per_s'])
PY
