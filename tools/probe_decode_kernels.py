"""Launch the two kernels that carry a decode step, as decode launches them, for rocprofv3 --pmc runs:
gate/up weight-streaming GEMM (packed, RMSNorm scale handed in above 26 rows) and split attention over a KV cache.

    python tools/probe_decode_kernels.py [B=256] [ctx=264]
"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("llm-speech-summarization_amd.ops")
L = importlib.import_module("llm-speech-summarization_amd._lib")
dev = "cuda:0"
H, F_, NL, nh, nkv, D = 3072, 8192, 28, 24, 8, 128
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx_mid = int(sys.argv[2]) if len(sys.argv) > 2 else 264
max_ctx = 448
ws = [] if (len(sys.argv) > 1 and int(sys.argv[1]) > 896) else [ops.pack_weight((torch.randn(2 * F_, H, device=dev) * 0.02).to(torch.bfloat16)) for _ in range(NL)]
x = torch.randn(B, H, device=dev).to(torch.bfloat16)
out = torch.empty(B, F_, device=dev, dtype=torch.bfloat16)
bf = L.dtype_code(torch.bfloat16)
sc_o, sc_d = L.lib().sl_gemm_split_count(B, H, nh * D, bf), L.lib().sl_gemm_split_count(B, H, F_, bf)
chain = sc_o > 1 and sc_d > 1
rstd_pass = (not chain) and B > 384 and sc_o == 1 and sc_d == 1
# the decode graph's rule (runtime.hip llama_layer, bench.py): gate/up on the ROW-MAJOR weights through the 256-tile kernel from ~900 rows
tiled_gu = os.environ.get("SL_DECODE_TILED", "1") != "0" and ((chain and B > 896) or rstd_pass)
rstd = torch.rsqrt(x.float().pow(2).mean(-1) + 1e-5) if (chain or rstd_pass) else None
if tiled_gu:
    ws = [(torch.randn(2 * F_, H, device=dev) * 0.02).to(torch.bfloat16) for _ in range(NL)]
NKV = 4   # distinct caches cycled (B=256: 4 x 2 x 235 MB)
kc = [(torch.randn(B, nkv, max_ctx, D, device=dev) * 0.5).to(torch.bfloat16) for _ in range(NKV)]
vc = [(torch.randn(B, nkv, max_ctx, D, device=dev) * 0.5).to(torch.bfloat16) for _ in range(NKV)]
ctx = torch.full((B,), ctx_mid, device=dev, dtype=torch.int32)
q = torch.randn(B, nh * D, device=dev).to(torch.bfloat16)
ao = torch.empty(B, nh * D, device=dev, dtype=torch.bfloat16)
aws = torch.empty(int(L.lib().sl_attn_decode_workspace_bytes(B, nh, nkv, max_ctx)), dtype=torch.uint8, device=dev)
for rep in range(3):
    for i, w in enumerate(ws):
        if tiled_gu:
            ops.gemm(x, w, act=L.ACT_SILU_MUL, out=out)
        else:
            ops.gemm_decode(x, w, 2 * F_, act=L.ACT_SILU_MUL, fuse_rms=True, eps=1e-5, out=out, rstd_in=rstd)
        ops.attn_decode_split(q, nh * D, kc[i % NKV], vc[i % NKV], ctx, nh, nkv, D, max_ctx, D ** -0.5, out=ao, ws=aws)
torch.cuda.synchronize()
print("gate/up form:", "gemm_tiled256p (row-major)" if tiled_gu else "packed streaming / skinny")
print("gemm algorithmic bytes per launch", 2 * F_ * H * 2 + B * H * 2 + B * F_ * 2)
print("attn algorithmic bytes per launch", B * nkv * ctx_mid * D * 2 * 2 + 2 * B * nh * D * 2)
