"""Time the decode attention kernel alone (as the decode graph launches it) at B sequences, context ctx.
    python tools/time_decode_attn.py [B=1024] [ctx=264]"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("llm-speech-summarization_amd.ops")
L = importlib.import_module("llm-speech-summarization_amd._lib")
dev = "cuda:0"
nh, nkv, D, max_ctx = 24, 8, 128, 448
for B, ctx_mid in ([(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else [(1024, 264), (512, 264), (1024, 393), (256, 264), (64, 264), (16, 264)]):
    NKV = 4 if B >= 512 else 16
    kc = [(torch.randn(B, nkv, max_ctx, D, device=dev) * 0.5).to(torch.bfloat16) for _ in range(NKV)]
    vc = [(torch.randn(B, nkv, max_ctx, D, device=dev) * 0.5).to(torch.bfloat16) for _ in range(NKV)]
    ctx = torch.full((B,), ctx_mid, device=dev, dtype=torch.int32)
    q = torch.randn(B, nh * D, device=dev).to(torch.bfloat16)
    ao = torch.empty(B, nh * D, device=dev, dtype=torch.bfloat16)
    aws = torch.empty(int(L.lib().sl_attn_decode_workspace_bytes(B, nh, nkv, max_ctx)), dtype=torch.uint8, device=dev)
    fn = lambda i: ops.attn_decode_split(q, nh * D, kc[i % NKV], vc[i % NKV], ctx, nh, nkv, D, max_ctx, D ** -0.5, out=ao, ws=aws)
    for i in range(8):
        fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    n = 56
    with torch.cuda.graph(g):
        for i in range(n):
            fn(i)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    nbytes = B * nkv * ctx_mid * D * 2 * 2 + 2 * B * nh * D * 2
    print(f"B={B:5d} ctx={ctx_mid}: {us:7.1f} us  {nbytes / us / 1e6:6.2f} TB/s  ({nbytes / us / 1e6 / 8:.3f} of 8 TB/s)", flush=True)
    del kc, vc
