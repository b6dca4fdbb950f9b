"""lm_head + greedy selection of one decode step, both forms, for rocprofv3 (--kernel-trace --stats, --pmc FETCH_SIZE / WRITE_SIZE):
  logits : sl_gemm(out_f32) writes (B, 128256) fp32 logits, sl_greedy_select re-reads them       (round 1-2)
  fused  : sl_gemm_ex(amax_*) leaves per-64-column maxima, sl_greedy_select_partial finishes     (round 3)

    python tools/probe_lm_head.py [B=1024] [fused|logits|both]
"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("llm-speech-summarization_amd.ops")
L = importlib.import_module("llm-speech-summarization_amd._lib")
dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
mode = sys.argv[2] if len(sys.argv) > 2 else "both"
H, V = 3072, 128256
W = (torch.randn(V, H, device=dev) * 0.02).to(torch.bfloat16)
x = torch.randn(B, H, device=dev).to(torch.bfloat16)
st = lambda: dict(unfinished=torch.ones(B, dtype=torch.int32, device=dev), ctx=torch.zeros(B, dtype=torch.int32, device=dev),
                  gen=torch.zeros(B, dtype=torch.int32, device=dev), fin=torch.zeros(B, dtype=torch.int32, device=dev),
                  nxt=torch.zeros(B, dtype=torch.int32, device=dev), out=torch.zeros((B, 64), dtype=torch.int32, device=dev))
res = {}
for m in (("logits", "fused") if mode == "both" else (mode,)):
    s = st()
    logits = torch.empty((B, V), device=dev, dtype=torch.float32) if m == "logits" else None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(12):
        if it == 2:
            e0.record()
        if m == "logits":
            ops.gemm(x, W, out_f32=True, out=logits)
            ops.greedy_select(logits, [], 0, False, s["unfinished"], s["ctx"], s["gen"], s["fin"], s["nxt"], s["out"])
        else:
            val, idx = ops.gemm_top1(x, W)
            ops.greedy_select_partial(val, idx, [], 0, False, s["unfinished"], s["ctx"], s["gen"], s["fin"], s["nxt"], s["out"])
    e1.record(); torch.cuda.synchronize()
    res[m] = (e0.elapsed_time(e1) / 10 * 1e3, s["out"][:, :12].clone())
    print(f"{m}: {res[m][0]:.1f} us per lm_head + select at B={B}")
if len(res) == 2:
    assert torch.equal(res["logits"][1], res["fused"][1]), "the two forms picked different tokens"
    print("tokens identical")
