"""Decode-shape GEMM microbench: y[M,N] = x[M,K] W[N,K]^T for the four Llama-3.2-3B projections at batch M,
cycling over several weight buffers so that nothing is served from L2 / MALL (as in the real layer loop)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("llm-speech-summarization_amd.ops")
L = importlib.import_module("llm-speech-summarization_amd._lib")
dev = "cuda:0"
NBUF = 8
shapes = [("qkv", 5120, 3072), ("o", 3072, 3072), ("gateup", 16384, 3072), ("down", 3072, 8192), ("lm_head", 128256, 3072)]
Ms = [int(a) for a in sys.argv[1:]] or [16, 64, 128, 256]
for name, N, K in shapes:
    nb = 2 if name == "lm_head" else NBUF
    Ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16) for _ in range(nb)]
    for M in Ms:
        A = torch.randn(M, K, device=dev).to(torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        for w in Ws:
            ops.gemm(A, w, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 4 * nb
        e0.record()
        for i in range(n):
            ops.gemm(A, Ws[i % nb], out=out)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        gbs = (N * K * 2 + M * K * 2 + M * N * 2) / us / 1e3
        print(f"{name:8s} M={M:4d} N={N:6d} K={K:5d}  {us:8.1f} us  {gbs:7.1f} GB/s  {2.0 * M * N * K / us / 1e6:7.1f} TF", flush=True)
    del Ws
