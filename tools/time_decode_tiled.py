"""Decode-step GEMM shapes of Llama-3.2-3B at 1024 rows through the row-major tiled kernels (what prefill uses), for comparison with the
packed weight-streaming forms of the decode step (profiles/r03_c_decode_gemm_forms.txt): python tools/time_decode_tiled.py [rows]"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("llm-speech-summarization_amd.ops")
L = importlib.import_module("llm-speech-summarization_amd._lib")
dev = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for name, N, K, act in (("qkv", 5120, 3072, L.ACT_NONE), ("o", 3072, 3072, L.ACT_NONE), ("gateup", 16384, 3072, L.ACT_SILU_MUL), ("down", 3072, 8192, L.ACT_NONE),
                        ("lm_head", 128256, 3072, L.ACT_NONE)):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    Ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16) for _ in range(2)]
    res = torch.randn(M, N, device=dev).to(torch.bfloat16) if name in ("o", "down") else None
    for i in range(3):
        ops.gemm(A, Ws[i % 2], act=act, residual=res)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(20):
        ops.gemm(A, Ws[i % 2], act=act, residual=res)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{name:8s} M={M} N={N} K={K}  {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s", flush=True)
