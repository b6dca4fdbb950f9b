#!/bin/bash
# round 6, t: the N = 2 control flow of bench.py on ONE GPU (SL_BENCH_SHARE_GPU=1 puts both ranks on cuda:0, SL_BENCH_BACKEND=gloo replaces RCCL, which refuses two
# ranks per device): per-rank legs, barrier + max-over-ranks bracket, the KD leg behind the inference legs with a real process group, exit status
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
SL_BENCH_SHARE_GPU=1 SL_BENCH_BACKEND=gloo timeout 1200 python bench.py --gpus 2 --batch 256 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --no-length-mix --no-eos-leg > $O/r06_t_two_ranks_one_gpu.json 2> $O/r06_t_two_ranks_one_gpu.err
echo "exit status $?" | tee $O/r06_t_rc.txt
tail -4 $O/r06_t_two_ranks_one_gpu.err | cut -c1-300
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r06_t_two_ranks_one_gpu.json").read().splitlines() if l.startswith("{")][-1])
print(d["n_gpus"], d["value"], d["scaling"], d.get("collective_backend"), [(p["rank"], p["utterances"], p["tokens"]) for p in d["per_rank"]])
k=d["kd_step"]; print({a: k.get(a) for a in ("samples_per_s", "window_ms", "scaling_mode", "error", "position")}); print(k.get("comm"))
PY
