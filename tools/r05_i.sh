#!/bin/bash
# round 5, GPU call I: 64-key decode attention with register prefetch (twice the bytes in flight per block) under the other batch's GEMMs
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_i; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q --tb=short -k "attn_decode" 2>&1 | tail -5 > $O/pytest_attn.txt
for ks in 128 64 65; do
  echo "SL_ATTN_DECODE_KS=$ks" >> $O/step.txt
  SL_ATTN_DECODE_KS=$ks SHARED_PREFIX=9 timeout 600 python tools/time_decode_step.py 1024 2>&1 | grep "B=" >> $O/step.txt
done
B="--steps 4 --warmup 1 --no-cpu-baseline --no-extra-legs --kd-optimizer-steps 0 --no-length-mix --no-eos-leg"
run() { tag=$1; shift; env "$@" timeout 600 python bench.py $B > $O/bench_$tag.json 2> $O/bench_$tag.err; }
run base A=1
run ks65 SL_ATTN_DECODE_KS=65
run base2 A=1
run ks65b SL_ATTN_DECODE_KS=65
cat $O/pytest_attn.txt $O/step.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_i/bench_*.json')):
    try:
        d=json.loads([l for l in open(f).read().splitlines() if l.startswith('{')][-1])
        print(f, d['value'], d['stage_ms'], d.get('stage_ms_one_batch_alone'))
    except Exception as e:
        print(f,'ERR',e)
PY
