"""KD window under `rocprofv3 --kernel-trace`: run with no argument inside the profiler (5 windows), then
    python tools/kd_window_trace.py <kernel_trace.csv>
prints, for the last window, wall span, time with at least one kernel running, per-queue busy time, idle gaps by size, and the
kernels in front of the largest gaps — is the window bound by kernel time or by the host's launch rate?"""
import csv, importlib, os, sys
if len(sys.argv) > 1:
    rows = list(csv.DictReader(open(sys.argv[1])))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")) for r in rows), key=lambda e: e[0])
    marks = [i for i, e in enumerate(ev) if e[2].startswith("adamw_multi_kernel")]
    assert len(marks) >= 2, "need two optimizer steps in the trace"
    w = ev[marks[-2] + 1: marks[-1] + 1]
    t0, t1 = w[0][0], max(e[1] for e in w)
    busy, cur_s, cur_e, gaps = 0, w[0][0], w[0][1], []
    for s, e, name, q in w[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, prev_name, name))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
        prev_name = name
    busy += cur_e - cur_s
    print(f"window: {len(w)} kernels, span {(t1 - t0) / 1e6:.2f} ms, some kernel running {busy / 1e6:.2f} ms ({busy / (t1 - t0):.1%}), idle {(t1 - t0 - busy) / 1e6:.2f} ms in {len(gaps)} gaps")
    per_q = {}
    for s, e, name, q in w:
        per_q[q] = per_q.get(q, 0) + e - s
    print("kernel time per queue (ms):", {q: round(v / 1e6, 2) for q, v in per_q.items()})
    for lo, hi in ((0, 2e3), (2e3, 5e3), (5e3, 2e4), (2e4, 1e5), (1e5, 1e12)):
        sel = [g for g in gaps if lo <= g[0] < hi]
        print(f"  gaps {lo / 1e3:.0f}-{hi / 1e3:.0f} us: {len(sel):5d}, {sum(g[0] for g in sel) / 1e6:.2f} ms")
    for g in sorted(gaps, reverse=True)[:12]:
        print(f"  gap {g[0] / 1e3:8.1f} us  after {g[1][:60]}  before {g[2][:60]}")
    sys.exit(0)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
P = "llm-speech-summarization_amd."
ri, cfgm, weights, enc_mod, llama_mod, utils, training = [importlib.import_module(P + m) for m in ("random_init", "config", "weights", "audio_encoder", "audio_llama", "utils", "training")]
dev = torch.device("cuda:0")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
harch, larch = weights.KNOWN_HUBERT["facebook/hubert-large-ls960-ft"], weights.KNOWN_LLAMA[utils.LLAMA_ID]
conf = cfgm.load_config(os.path.join(REPO, "config", "llama3_hubert.yaml"))
enc = enc_mod.AudioEncoder(conf, dev, dtype=torch.bfloat16, arch=harch)
enc.load_state_dict(ri.hubert_encoder_state_dict(harch, larch.hidden_size, seed=0)).eval().to(dev)
llm = llama_mod.AudioLlamaForCausalLM(larch, bench.gpu_llama_state_dict(larch, 0, dev), torch_dtype=torch.bfloat16, device=dev, max_ctx=512, max_batch=16)
prefix = ri.synthetic_ids(9, larch.vocab_size, seed=7, bos=128000); suffix = ri.synthetic_ids(6, larch.vocab_size, seed=8, bos=128000)
tr = training.KDTrainer(conf, enc, llm, prefix, suffix, total_optimizer_steps=1000, regularizers=training.TrainRegularizers(seed=1234))
g = torch.Generator().manual_seed(99)
text_ids = torch.randint(1, larch.vocab_size, (40,), generator=g); resp_ids = torch.randint(1, larch.vocab_size, (64,), generator=g)
wave = ri.synthetic_waveform(160000, seed=4321).to(dev)
B = int(os.environ.get("KD_WINDOW", tr.local_accum))
tr.local_accum = B
args = ([wave] * B, [text_ids] * B, [resp_ids] * B)
import time
for i in range(5):
    torch.cuda.synchronize(); t = time.perf_counter()
    tr.micro_batch(*args)
    torch.cuda.synchronize()
    print(f"window {i}: {(time.perf_counter() - t) * 1e3:.1f} ms", flush=True)
