"""Tiled-GEMM epilogue cost at the encoder shapes: plain vs +bias vs +bias+residual (the out_proj / FFN2 form)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("llm-speech-summarization_amd.ops")
dev = "cuda:0"
shapes = [(255488, 1024, 1024), (255488, 1024, 4096), (63872, 3072, 1024), (17408, 3072, 8192)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]]
for M, N, K in shapes:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev).to(torch.bfloat16)
    R = torch.randn(M, N, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    line = f"M={M:6d} N={N:5d} K={K:5d}"
    for name, kw in (("plain", {}), ("bias", {"bias": bias}), ("bias+res", {"bias": bias, "residual": R}), ("in-place res", {"bias": bias, "residual": out})):
        for _ in range(2):
            ops.gemm(A, W, out=out, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.gemm(A, W, out=out, **kw)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        line += f" | {name} {us:8.1f} us {2.0 * M * N * K / us / 1e6:6.1f} TF"
    ops.gemm(A, W, out=out, bias=bias, residual=R)
    ref = A[:4096].float() @ W.float().T + bias.float() + R[:4096].float()
    line += f" | rel err {float((out[:4096].float() - ref).norm() / ref.norm()):.1e}"
    print(line, flush=True)
