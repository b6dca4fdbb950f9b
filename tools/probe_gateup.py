"""Launch the dominant decode kernel (packed gate/up weight-streaming GEMM, M = argv[1] rows, default 128) as decode does,
for rocprofv3 --pmc runs."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("llm-speech-summarization_amd.ops")
L = importlib.import_module("llm-speech-summarization_amd._lib")
dev = "cuda:0"
H, F_, NL = 3072, 8192, 28
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
FUSE = not (len(sys.argv) > 2 and sys.argv[2] == "nofuse")   # decode hands gate/up the row scales (rstd_in); "nofuse" = that arithmetic
ws = [ops.pack_weight((torch.randn(2 * F_, H, device=dev) * 0.02).to(torch.bfloat16)) for _ in range(NL)]
x = torch.randn(B, H, device=dev).to(torch.bfloat16)
out = torch.empty(B, F_, device=dev, dtype=torch.bfloat16)
for rep in range(3):
    for w in ws:
        ops.gemm_decode(x, w, 2 * F_, act=L.ACT_SILU_MUL, fuse_rms=FUSE, eps=1e-5, out=out)
torch.cuda.synchronize()
print("algorithmic bytes per launch", 2 * F_ * H * 2 + B * H * 2 + B * F_ * 2)
