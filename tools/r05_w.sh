#!/bin/bash
# round 5, run w: KD leg before vs after the inference legs on the same box (bench.py --kd-order), plus the self-describing small-batch test
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r05_w
timeout 900 python -m pytest tests/test_dp_gpu.py -x -q -m gpu -k "bench_line" > gpurun_out/r05_w/pytest.txt 2>&1
tail -3 gpurun_out/r05_w/pytest.txt
timeout 600 python bench.py --kd-order first > gpurun_out/r05_w/bench_kd_first.json 2> gpurun_out/r05_w/bench_kd_first.err
timeout 600 python bench.py --kd-order last > gpurun_out/r05_w/bench_kd_last.json 2> gpurun_out/r05_w/bench_kd_last.err
python - <<'PY'
import json
for n in ("first", "last"):
    try:
        r = json.loads(open(f"gpurun_out/r05_w/bench_kd_{n}.json").read().strip().splitlines()[-1])
        k = r["kd_step"]
        print(n, r["value"], k.get("samples_per_s"), k.get("window_ms"), k["roofline"]["frac"], k.get("per_rank_regime_probe", {}).get("window_ms"), k.get("error"), r.get("wall_s"))
    except Exception as e:
        print(n, "failed", e)
PY
