"""Tile-choice sweep for the KD-window GEMM shapes: the 256^2 tile's admission threshold (SL_T256_MIN_TILES) at M = 2-8 k rows.
    python tools/sweep_t256.py            # prints TF/s per shape and threshold"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("llm-speech-summarization_amd.ops")
L = importlib.import_module("llm-speech-summarization_amd._lib")
dev = "cuda:0"
shapes = []
for M in (7984, 3200, 5072, 1872):
    enc = M == 7984
    for N, K in (((3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)) if enc else ((5120, 3072), (3072, 3072), (16384, 3072), (3072, 8192))):
        shapes.append((M, N, K))
shapes += [(3200, 3072, 1024), (3200, 4096, 1024), (3200, 1024, 4096)]
print("shape                      " + "".join(f"{t:>10}" for t in ("512", "200", "100", "48", "hipblaslt")))
for M, N, K in shapes:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    Ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16) for _ in range(4)]
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    row = []
    for thr in ("512", "200", "100", "48", "torch"):
        if thr != "torch":
            os.environ["SL_T256_MIN_TILES"] = thr
            L.lib().sl_tuning_reload()
            fn = lambda i: ops.gemm(A, Ws[i % 4], out=out)
        else:
            fn = lambda i: torch.nn.functional.linear(A, Ws[i % 4], out=None)
        for i in range(4):
            fn(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 24
        e0.record()
        for i in range(n):
            fn(i)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        row.append(2.0 * M * N * K / us / 1e6)
    print(f"{M:5d} x {N:5d} x {K:4d}        " + "".join(f"{v:10.0f}" for v in row), flush=True)
os.environ.pop("SL_T256_MIN_TILES", None)
L.lib().sl_tuning_reload()
