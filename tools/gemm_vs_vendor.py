"""Tiled GEMM against the vendor library (torch.nn.functional.linear = hipBLASLt) on the shapes the three legs run:
the KD-window rows of profiles/r03_i_tile_choice_sweep.txt, the four encoder products at 127 744 rows and the prefill
products at 140 288 rows.  Random bf16 operands, all variants interleaved in ONE process (guide §5.4 rule 24), median of the
rounds; every variant's result is checked against the fp32 product first.

    python tools/gemm_vs_vendor.py [--rounds 5] [--quick]

Variants: r3 = round-3 loop (SL_T256_PHASED=0), p = staggered two-phase loop (default), pad0 = p with round 4's padding bound
on the whole-rounds tile choice (SL_T256_BY_ROUNDS_PAD=0), vendor."""
import argparse, importlib, os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("llm-speech-summarization_amd.ops")
L = importlib.import_module("llm-speech-summarization_amd._lib")
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--quick", action="store_true")
ap.add_argument("--variants", default="r3,p,vendor")
ap.add_argument("--mid", action="store_true", help="the KD-window rows (M = 1 872 ... 7 984) and the per-rank rows")
ap.add_argument("--small", action="store_true", help="only the per-rank KD window rows (M <= 1 000) and the 2 048-row products")
args = ap.parse_args()
dev = "cuda:0"

shapes = []
for M in (7984, 3200, 5072, 1872):
    enc = M == 7984
    for N, K in (((3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)) if enc else ((5120, 3072), (3072, 3072), (16384, 3072), (3072, 8192))):
        shapes.append((M, N, K))
shapes += [(3200, 3072, 1024), (3200, 4096, 1024), (3200, 1024, 4096)]
# the per-rank KD window (2 samples: 634 LLM rows, 998 encoder frames) and the decode rows
shapes += [(634, 5120, 3072), (634, 3072, 3072), (634, 16384, 3072), (634, 3072, 8192), (634, 8192, 3072), (634, 3072, 16384),
           (4096, 1024, 8064), (1024, 4096, 8064), (3072, 1024, 8064), (1024, 1024, 8064),      # the encoder's weight gradients of a 16-sample window (token rows padded to whole slab pairs)
           (998, 3072, 1024), (998, 4096, 1024), (998, 1024, 4096), (2048, 5120, 3072), (2048, 3072, 3072), (2048, 16384, 3072), (2048, 3072, 8192)]
big = [(127744, 3072, 1024), (127744, 1024, 1024), (127744, 4096, 1024), (127744, 1024, 4096), (140288, 5120, 3072), (140288, 3072, 8192),
       (70144, 16384, 3072)]
if args.quick:
    shapes = [(7984, 3072, 1024), (5072, 5120, 3072), (3200, 3072, 3072)]
    big = [(127744, 3072, 1024), (127744, 1024, 4096), (140288, 5120, 3072)]
shapes = big + shapes
if args.mid:
    shapes = [sh for sh in shapes if sh[0] <= 7984 and sh[2] != 8064]
if args.small:
    shapes = [sh for sh in shapes if sh[0] <= 2048 and sh[2] != 8064] + [(634, 3072, 5120), (998, 1024, 1024), (998, 1024, 3072), (137, 5120, 3072), (137, 16384, 3072)]
variants = args.variants.split(",")


SK = None


def our_gemm(A, W, out, v):
    """ops.gemm, or (variant sk) the same product with the split-K / stream-K workspace a KD tape passes (sl_gemm_ex_args.sk_ws)"""
    global SK
    if not v.startswith("sk"):
        return ops.gemm(A, W, out=out)
    if SK is None:
        SK = ops.streamk_workspace(A.device)
    return ops.gemm_ex(A, W, M=A.shape[0], N=W.shape[0], K=A.shape[1], lda=A.stride(0), ldw=W.stride(0), out=out, sk_ws=SK)


def set_variant(v):
    if v == "vendor":
        return
    os.environ["SL_GLDS_RING_MAX_TILES"] = v[1:] if v[0] == "m" and v[1:].isdigit() else "0"      # m384: the ring takes up to 384 tiles (two rounds)
    if v[0] == "m" and v[1:].isdigit():
        v = "p"
    # p2 / sk2: the two-stage 128-tile kernel (SL_GLDS_RING=0, round 5's rule), p3 / sk3: three ring stages; p / sk: the default (four)
    os.environ["SL_GLDS_RING"] = "0" if v.endswith("2") else ("3" if v.endswith("3") and v != "r3" else ("104" if v.endswith("u") else ("204" if v.endswith("o") else "4")))      # po / sko: the ring with a slab's eight requests inside one 16-MFMA phase
    os.environ["SL_GLDS_DMAB"] = "1" if v.endswith("d") else "0"      # pd: the two-stage kernel with its DMA requests between the MFMAs (default: a burst)
    if v.startswith("sk") or v in ("p2", "p3", "pu", "pd", "po"):      # pu / sku: the ring without the software-pipelined fragment reads
        v = "p"
    os.environ["SL_T256_PHASED"] = {"r3": "0", "p": "1", "pad0": "1"}[v]
    os.environ["SL_T256_BY_ROUNDS_PAD"] = "0" if v == "pad0" else "1"      # pad0 = round 4's row-padding bound on the whole-rounds choice
    L.lib().sl_tuning_reload()


print(f"{'shape':>26} " + "".join(f"{v:>9}" for v in variants) + "   p/vendor   (TF/s, median of %d rounds)" % args.rounds, flush=True)
worst = 10.0
for M, N, K in shapes:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    Ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16) for _ in range(3)]
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    # correctness of every variant on a row sample (the fp32 product of the whole thing is too big at 140 k rows)
    idx = torch.randint(0, M, (512,), device=dev)
    ref = A[idx].float() @ Ws[0].float().T
    for v in variants:
        if v == "vendor":
            continue
        set_variant(v)
        out.zero_()
        our_gemm(A, Ws[0], out, v)
        err = float((out[idx].float() - ref).norm() / ref.norm())
        assert err < 6e-3, (M, N, K, v, err)
    n = 6 if M * N * K > 3e11 else 16
    times = {v: [] for v in variants}
    for rnd in range(args.rounds + 1):
        for v in variants:
            set_variant(v)
            fn = (lambda i: torch.nn.functional.linear(A, Ws[i % 3])) if v == "vendor" else (lambda i, v=v: our_gemm(A, Ws[i % 3], out, v))
            fn(0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(n):
                fn(i)
            e1.record(); torch.cuda.synchronize()
            if rnd:
                times[v].append(e0.elapsed_time(e1) / n * 1e3)
    tf = {v: 2.0 * M * N * K / statistics.median(times[v]) / 1e6 for v in variants}
    ours = tf["p"] if "p" in tf else tf[variants[0]]
    ratio = ours / tf["vendor"] if "vendor" in tf else float("nan")
    worst = min(worst, ratio)
    print(f"{M:7d} x {N:6d} x {K:5d}  " + "".join(f"{tf[v]:9.0f}" for v in variants) + f"   {ratio:6.3f}", flush=True)
print(f"worst p/vendor ratio: {worst:.3f}")
os.environ.pop("SL_T256_PHASED", None)
os.environ.pop("SL_T256_BY_ROUNDS_PAD", None)
os.environ.pop("SL_GLDS_RING", None)
os.environ.pop("SL_GLDS_DMAB", None)
os.environ.pop("SL_GLDS_RING_MAX_TILES", None)
L.lib().sl_tuning_reload()
