#!/bin/bash
# round 6, a: (1) the compaction parity cases on the pinned family, (2) the default bench line of the round's starting code (+ pin, + id checksum),
# (3) the eos_stop_mix leg with each rung picking its own family (SL_COMPACT_PIN=0) for the throughput price of the pin
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
python -m pytest tests/test_fullsize_gpu.py tests/test_models_gpu.py -x -q -m gpu -k "compact" -s 2>&1 | grep -v "^$" | tail -15 > $O/r06_a_compact_tests.txt
T0=$(date +%s)
python bench.py > $O/r06_a_default_line.json 2> $O/r06_a_default.err
T1=$(date +%s)
echo "default bench wall seconds: $((T1 - T0))" > $O/r06_a_wall.txt
SL_COMPACT_PIN=0 python bench.py --steps 2 --warmup 1 --kd-optimizer-steps 0 --no-cpu-baseline --no-length-mix --no-extra-legs > $O/r06_a_nopin_line.json 2> $O/r06_a_nopin.err
cat $O/r06_a_compact_tests.txt $O/r06_a_wall.txt; tail -3 $O/r06_a_default.err
python - <<'PY'
import json
def last(p): return json.loads([l for l in open(p).read().splitlines() if l.startswith('{')][-1])
d=last('gpurun_out/r06_a_default_line.json')
print(d['value'], d['graded'], d['kd_step']['samples_per_s'], d['kd_per_rank_regime_probe']['window_ms'], d['latency_b1']['decode_tokens_per_s'])
print('pinned  ', {k: d['eos_stop_mix'][k] for k in ('compacted','uncompacted','ids_identical_compacted_vs_uncompacted')})
n=last('gpurun_out/r06_a_nopin_line.json')
print('unpinned', {k: n['eos_stop_mix'][k] for k in ('compacted','uncompacted','ids_identical_compacted_vs_uncompacted')})
PY
