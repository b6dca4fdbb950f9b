"""FFN1-shaped GEMM with and without the GELU epilogue: python tools/time_gelu_epilogue.py"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("llm-speech-summarization_amd.ops")
L = importlib.import_module("llm-speech-summarization_amd._lib")
dev = "cuda:0"
for M, N, K in ((127744, 4096, 1024), (127744, 3072, 1024), (127744, 1024, 4096), (127744, 1024, 1024)):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    Ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16) for _ in range(3)]
    bias = torch.randn(N, device=dev).to(torch.bfloat16)
    res = torch.randn(M, N, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for name, kw in (("plain", {}), ("bias", dict(bias=bias)), ("bias+gelu", dict(bias=bias, act=L.ACT_GELU)), ("bias+res", dict(bias=bias, residual=res))):
        for i in range(3):
            ops.gemm(A, Ws[i], out=out, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(12):
            ops.gemm(A, Ws[i % 3], out=out, **kw)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 12 * 1e3
        print(f"{M} x {N} x {K} {name:10s} {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s", flush=True)
