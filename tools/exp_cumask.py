"""CU-partitioned concurrency probe (round 3): does decode attention (HBM-bound) on a subset of the CUs overlap
with the decode GEMMs (L2 / MFMA-bound) of another batch on the rest?  Streams with CU masks (hipExtStreamCreateWithCUMask);
kernels launched eagerly, as a 1 024-row decode step could afford (40-190 us kernels).

    python tools/exp_cumask.py [B=1024] [ctx=264]
Prints per-kernel times alone under each mask and the wall time of the two kernel chains run concurrently on complementary masks.
"""
import ctypes as C, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("llm-speech-summarization_amd.ops")
L = importlib.import_module("llm-speech-summarization_amd._lib")
dev = "cuda:0"
torch.cuda.set_device(0)
hip = C.CDLL("libamdhip64.so")
H, F_, NL, nh, nkv, D = 3072, 8192, 8, 24, 8, 128
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx_mid = int(sys.argv[2]) if len(sys.argv) > 2 else 264
max_ctx = 448
N_CU, N_XCD = 256, 8


def mask_words(cus):
    w = [0] * 8
    for c in cus:
        w[c // 32] |= 1 << (c % 32)
    return w


def masked_stream(cus):
    arr = (C.c_uint32 * 8)(*mask_words(cus))
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


def per_xcd(lo, hi, layout):
    """CUs lo..hi-1 of every XCD under the assumed bit layout: 'block' = bit 32 x + c, 'inter' = bit 8 c + x."""
    return [(32 * x + c) if layout == "block" else (8 * c + x) for x in range(N_XCD) for c in range(lo, hi)]


ws = [ops.pack_weight((torch.randn(2 * F_, H, device=dev) * 0.02).to(torch.bfloat16)) for _ in range(NL)]
wo = [ops.pack_weight((torch.randn(H, H, device=dev) * 0.02).to(torch.bfloat16)) for _ in range(NL)]
x = torch.randn(B, H, device=dev).to(torch.bfloat16)
res = torch.randn(B, H, device=dev).to(torch.bfloat16)
out = torch.empty(B, F_, device=dev, dtype=torch.bfloat16)
out_o = torch.empty(B, H, device=dev, dtype=torch.bfloat16)
rstd = torch.rsqrt(x.float().pow(2).mean(-1) + 1e-5)
kc = [(torch.randn(B, nkv, max_ctx, D, device=dev) * 0.5).to(torch.bfloat16) for _ in range(4)]
vc = [(torch.randn(B, nkv, max_ctx, D, device=dev) * 0.5).to(torch.bfloat16) for _ in range(4)]
ctx = torch.full((B,), ctx_mid, device=dev, dtype=torch.int32)
q = torch.randn(B, nh * D, device=dev).to(torch.bfloat16)
ao = torch.empty(B, nh * D, device=dev, dtype=torch.bfloat16)
aws = torch.empty(int(L.lib().sl_attn_decode_workspace_bytes(B, nh, nkv, max_ctx)), dtype=torch.uint8, device=dev)
ws2 = torch.zeros(int(L.lib().sl_gemm_split_workspace_bytes(B, H, H, L.dtype_code(torch.bfloat16))) + 256, dtype=torch.uint8, device=dev)


def gemm(i):
    ops.gemm_decode(x, ws[i % NL], 2 * F_, act=L.ACT_SILU_MUL, fuse_rms=True, eps=1e-5, out=out, rstd_in=rstd)


def attn(i):
    ops.attn_decode_split(q, nh * D, kc[i % 4], vc[i % 4], ctx, nh, nkv, D, max_ctx, D ** -0.5, out=ao, ws=aws)


N = 56


def alone(fn, stream):
    with torch.cuda.stream(stream):
        for i in range(8):
            fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(N):
            fn(i)
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N * 1e3


def together(sa, sg, na=N, ng=N):
    import time
    for st, fn in ((sa, attn), (sg, gemm)):
        with torch.cuda.stream(st):
            for i in range(4):
                fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(max(na, ng)):          # interleaved submission, as two decode loops would
        if i < na:
            with torch.cuda.stream(sa):
                attn(i)
        if i < ng:
            with torch.cuda.stream(sg):
                gemm(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6


full = torch.cuda.Stream()
full2 = torch.cuda.Stream()
ta, tg = alone(attn, full), alone(gemm, full)
print(f"B={B} ctx={ctx_mid}: alone on all CUs: attention {ta:.1f} us, gate/up {tg:.1f} us; serial chain of {N}+{N}: {(ta + tg) * N:.0f} us", flush=True)
print(f"two unmasked streams, both chains concurrently: {together(full, full2):.0f} us", flush=True)
for layout in ("block", "inter"):
    for n_att in (8, 12, 16, 20, 24):
        sa = masked_stream(per_xcd(0, n_att, layout))
        sg = masked_stream(per_xcd(n_att, 32, layout))
        a1, g1 = alone(attn, sa), alone(gemm, sg)
        # balance the chain lengths so both finish together: attention launches per GEMM launch
        wall = together(sa, sg)
        print(f"layout {layout:5s} attention on {n_att:2d} CUs/XCD: {a1:6.1f} us ({ta / a1:.2f} of full rate) | gate/up on {32 - n_att:2d}: {g1:6.1f} us ({tg / g1:.2f}) | "
              f"both chains concurrently {wall:7.0f} us vs serial {(ta + tg) * N:7.0f} us -> x{(ta + tg) * N / wall:.2f}", flush=True)
