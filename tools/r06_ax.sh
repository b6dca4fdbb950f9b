#!/bin/bash
# round 6, ax: on the FINAL code — (1) the N = 2 control flow of bench.py on one GPU (both ranks on cuda:0 over gloo: the KD leg's 8-sample per-rank windows now run
# the ring kernels, the one-launch attention backward and the side stream), (2) every M <= 2 048 row against the vendor library in one table
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_ax; mkdir -p $O
SL_BENCH_SHARE_GPU=1 SL_BENCH_BACKEND=gloo timeout 1200 python bench.py --gpus 2 --batch 256 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --no-length-mix --no-eos-leg > $O/two_ranks_one_gpu_line.json 2> $O/two_ranks_one_gpu.err
echo "exit status $?" | tee $O/two_ranks_rc.txt
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r06_ax/two_ranks_one_gpu_line.json").read().splitlines() if l.startswith("{")][-1])
print(d["n_gpus"], d["value"], d["scaling"], d.get("collective_backend"), [(p["rank"], p["utterances"], p["tokens"]) for p in d["per_rank"]])
k=d["kd_step"]; print({a: k.get(a) for a in ("samples_per_s", "window_ms", "scaling_mode", "error", "position")}); print(str(k.get("comm"))[:600])
PY
timeout 900 python tools/gemm_vs_vendor.py --small --rounds 3 --variants p2,p,sk,vendor 2>&1 | grep -v amdgpu.ids > $O/gemm_vs_vendor_small_final.txt
cat $O/gemm_vs_vendor_small_final.txt
