#!/bin/bash
# round 6, o: (1) the KD windows after the loss read-back moved behind the optimizer step's launches; (2) VERDICT r5 item 5 by proxy: the same 1 024 rows
# per GPU as ONE batch alone, as TWO 512-row half-batches in flight (each half's attention beside the other's GEMMs, skewed by the host threads), and the
# default two 1 024-row batches in flight
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_o; mkdir -p $O
python tools/kd_ab_inproc.py SL_TAPE_FUSE=0 3 16 2>&1 | grep "window of" > $O/kd_after_readback.txt
python tools/kd_ab_inproc.py SL_TAPE_FUSE=0 3 2 2>&1 | grep "window of" >> $O/kd_after_readback.txt
X="--no-cpu-baseline --no-length-mix --no-extra-legs --no-eos-leg --kd-optimizer-steps 0 --steps 4 --warmup 1"
for cfg in "1 1024" "2 512" "2 1024"; do set -- $cfg; python bench.py $X --pipelines $1 --batch $2 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('pipelines $1 batch $2: %.0f tok/s  ms_per_step %.1f  stage_ms_one_batch_alone %s' % (d['value'], d['ms_per_step'], d.get('stage_ms_one_batch_alone')))"; done > $O/halves_proxy.txt
cat $O/kd_after_readback.txt $O/halves_proxy.txt
