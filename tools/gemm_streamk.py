"""Stream-K form of the 256 x 256 GEMM (sl_gemm_ex_args.sk_ws) on the shapes whose tiles do not fill the chip: KD windows, the per-rank
KD regime (M = 400-640 rows), weight gradients of a few dozen tiles under K = 8 000, decode projections at 1 024 rows.
Every form is checked against the fp32 product; interleaved rounds in one process, median.

    python tools/gemm_streamk.py [--rounds 4]
"""
import argparse, importlib, os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("llm-speech-summarization_amd.ops")
L = importlib.import_module("llm-speech-summarization_amd._lib")
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=4)
args = ap.parse_args()
dev = "cuda:0"
# (M, N, K, form)   form: "" plain bf16, "res" + bf16 residual, "acc" fp32 accumulate (weight gradient)
shapes = [(3200, 3072, 16384, ""), (3200, 3072, 5120, ""), (3200, 3072, 3072, ""), (5072, 5120, 3072, ""), (3200, 8192, 3072, ""), (5072, 3072, 8192, "res"),
          (5072, 16384, 3072, ""), (1872, 5120, 3072, ""), (1872, 3072, 8192, ""), (3200, 1024, 4096, ""), (7984, 1024, 1024, ""), (7984, 3072, 1024, ""),
          (4096, 1024, 8000, "acc"), (1024, 4096, 8000, "acc"), (3072, 1024, 8000, "acc"), (1024, 1024, 8000, "acc"),
          (634, 16384, 3072, ""), (634, 3072, 8192, "res"), (634, 5120, 3072, ""), (400, 3072, 16384, ""), (400, 3072, 3072, ""), (1000, 4096, 1024, ""),
          (1024, 3072, 8192, "res"), (1024, 3072, 3072, "res")]
ws = ops.streamk_workspace(dev)
variants = ["tile", "sk", "sk2", "vendor"]
print(f"{'shape':>30} " + "".join(f"{v:>9}" for v in variants) + "   sk/vendor  (TF/s, median; sk = rule, sk2 = forced)", flush=True)
for M, N, K, form in shapes:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    Ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16) for _ in range(3)]
    res = torch.randn(M, N, device=dev).to(torch.bfloat16) if form == "res" else None
    accb = torch.randn(M, N, device=dev) if form == "acc" else None
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if form == "acc" else torch.bfloat16)

    def run(v, i):
        if v == "vendor":
            y = torch.nn.functional.linear(A, Ws[i % 3])
            if res is not None:
                y += res
            return y
        kw = dict(M=M, N=N, K=K, lda=K, ldw=K, out=out)
        if form == "res":
            kw.update(residual=res, ldr=N)
        if form == "acc":
            out.copy_(accb) if i < 0 else None
            kw.update(residual=out, ldr=N, out_f32=True, residual_f32=True)
        if v != "tile":
            kw.update(sk_ws=ws)
        return ops.gemm_ex(A, Ws[i % 3], **kw)

    def setv(v):
        os.environ["SL_STREAM_K"] = "2" if v == "sk2" else "1"
        L.lib().sl_tuning_reload()

    ref = A.float() @ Ws[0].float().T + (res.float() if res is not None else 0) + (accb if accb is not None else 0)
    bad = []
    for v in ("tile", "sk", "sk2"):
        setv(v)
        for rep in range(3):
            out.zero_()
            run(v, -1) if form == "acc" else None
            if form != "acc":
                run(v, 0)
            else:
                out.copy_(accb); kw = None; run(v, 0)
            err = float((out.float() - ref).norm() / ref.norm())
            if not err < 6e-3:
                bad.append((v, rep, err)); break
    times = {v: [] for v in variants}
    n = 16
    for rnd in range(args.rounds + 1):
        for v in variants:
            setv(v)
            run(v, 0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(n):
                run(v, i)
            e1.record(); torch.cuda.synchronize()
            if rnd:
                times[v].append(e0.elapsed_time(e1) / n * 1e3)
    tf = {v: 2.0 * M * N * K / statistics.median(times[v]) / 1e6 for v in variants}
    print(f"{M:6d} x {N:6d} x {K:6d} {form:>4}  " + "".join(f"{tf[v]:9.0f}" for v in variants) + f"   {max(tf['sk'], 0) / tf['vendor']:6.3f}" + (f"   WRONG {bad}" if bad else ""), flush=True)
os.environ.pop("SL_STREAM_K", None)
L.lib().sl_tuning_reload()
