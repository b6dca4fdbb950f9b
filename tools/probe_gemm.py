"""One tiled-GEMM shape launched a few times (for rocprofv3 --pmc passes): python tools/probe_gemm.py M N K [reps]."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("llm-speech-summarization_amd.ops")
M, N, K = (int(v) for v in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
dev = "cuda:0"
A = torch.randn(M, K, device=dev).to(torch.bfloat16)
W = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for _ in range(reps):
    ops.gemm(A, W, out=out)
torch.cuda.synchronize()
