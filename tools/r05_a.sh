#!/bin/bash
# round 5, GPU call A: parity suite, default bench line, the 2 048-row step against the 1 024-row one
mkdir -p gpurun_out/r05_a
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > gpurun_out/r05_a/pytest.txt
echo "pytest rc=$?" >> gpurun_out/r05_a/pytest.txt
timeout 900 python bench.py --steps 4 --warmup 1 > gpurun_out/r05_a/bench_default.json 2> gpurun_out/r05_a/bench_default.err
for cfg in "1024 1" "2048 1" "1536 1"; do
  set -- $cfg
  timeout 600 python bench.py --batch $1 --pipelines $2 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --no-length-mix --kd-optimizer-steps 0 --no-eos-leg \
    > gpurun_out/r05_a/bench_b$1_p$2.json 2> gpurun_out/r05_a/bench_b$1_p$2.err
done
tail -5 gpurun_out/r05_a/pytest.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_a/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['value'], d.get('stage_ms_one_batch_alone'), d.get('graded'))
    except Exception as e:
        print(f,'ERR',e)
PY
