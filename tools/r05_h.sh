#!/bin/bash
# round 5, GPU call H: do blocks of two kernels on two streams share a CU when their resources fit (probe)?  + the finer compaction ladder
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_h; mkdir -p $O
hipcc -O3 --offload-arch=gfx950 tools/probe_coresidency.hip -o $O/probe_coresidency 2> $O/probe_build.txt
{
  echo "## hog 130 KiB LDS (256-tile GEMM block) + small 19.5 KiB (64-key attention block)"; timeout 60 $O/probe_coresidency 133120 19968 512 2048 400 50
  echo "## hog 130 KiB + small 38.7 KiB (128-key attention block: does not fit beside)"; timeout 60 $O/probe_coresidency 133120 39629 512 2048 400 50
  echo "## hog 144 KiB (256 x 128 streaming block) + small 19.5 KiB (does not fit)"; timeout 60 $O/probe_coresidency 147456 19968 512 2048 400 50
  echo "## hog 64 KiB + small 19.5 KiB"; timeout 60 $O/probe_coresidency 65536 19968 512 2048 400 50
  echo "## hog 130 KiB one round (256 blocks) + small 19.5 KiB, long small blocks"; timeout 60 $O/probe_coresidency 133120 19968 256 1024 400 200
} > $O/probe_coresidency.txt 2>&1
timeout 900 python -m pytest tests/test_models_gpu.py -q --tb=short -k "compacted or batched_generate or large_batch_decode" 2>&1 | tail -6 > $O/pytest.txt
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --kd-optimizer-steps 0 --no-length-mix > $O/bench_eos.json 2> $O/bench_eos.err
cat $O/probe_coresidency.txt; cat $O/pytest.txt
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05_h/bench_eos.json').read().splitlines() if l.startswith('{')][-1])
print(json.dumps(d['eos_stop_mix']))
PY
