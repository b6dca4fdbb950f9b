"""Tiled-GEMM microbench at the encoder / prefill shapes, glds vs register staging, random bf16 data."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("llm-speech-summarization_amd.ops")
L = importlib.import_module("llm-speech-summarization_amd._lib")
dev = "cuda:0"
shapes = [(63872, 1024, 1024), (63872, 4096, 1024), (17408, 16384, 3072), (17408, 3072, 8192), (7984, 1024, 1024), (7984, 3072, 1024), (7984, 4096, 1024), (7984, 1024, 4096), (15999, 512, 1536), (2176, 5120, 3072),
          (2176, 16384, 3072), (2176, 3072, 8192), (3200, 3072, 3072)]
for M, N, K in shapes:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    res = []
    for flag in ("1", "2", "0"):
        os.environ["SL_DISABLE_GLDS"] = flag
        L.lib().sl_tuning_reload()   # the library reads its tuning switches once; re-read after changing them
        for _ in range(3):
            ops.gemm(A, W, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.gemm(A, W, out=out)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        res.append((us, 2.0 * M * N * K / us / 1e6))
    ref = (A.float() @ W.float().T)
    err = float((out.float() - ref).norm() / ref.norm())
    print(f"M={M:6d} N={N:6d} K={K:5d}  regstage {res[0][0]:8.1f} us {res[0][1]:7.1f} TF | glds {res[1][0]:8.1f} us {res[1][1]:7.1f} TF | glds+asm-lds {res[2][0]:8.1f} us {res[2][1]:7.1f} TF | rel err {err:.2e}")
