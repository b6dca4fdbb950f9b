"""Race screen of the staggered two-phase GEMM loop: many launches of several shapes, each compared BITWISE with the round-3 loop's
result (same order of additions), while a second stream keeps the chip unevenly loaded.    python tools/gemm_race_screen.py [reps]"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("llm-speech-summarization_amd.ops")
L = importlib.import_module("llm-speech-summarization_amd._lib")
dev = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
side = torch.cuda.Stream()
noise = torch.randn(3000, 3000, device=dev)
bad = 0
for M, N, K in ((127744, 1024, 1024), (140288, 5120, 3072), (7984, 4096, 1024), (66000, 512, 64), (33000, 768, 192), (5072, 16384, 3072)):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    R = torch.randn(M, N, device=dev).to(torch.bfloat16)
    os.environ["SL_T256_PHASED"] = "0"; L.lib().sl_tuning_reload()
    ref = ops.gemm(A, W, residual=R)
    os.environ["SL_T256_PHASED"] = "1"; L.lib().sl_tuning_reload()
    out = torch.empty_like(ref)
    n_bad = 0
    for i in range(reps):
        if i % 3 == 0:
            with torch.cuda.stream(side):
                (noise @ noise).sum()
        out.fill_(float("nan"))
        ops.gemm(A, W, residual=R, out=out)
        if not torch.equal(out, ref):
            n_bad += 1
    print(f"{M} x {N} x {K}: {reps} launches, {n_bad} differ", flush=True)
    bad += n_bad
os.environ.pop("SL_T256_PHASED"); L.lib().sl_tuning_reload()
print("RACE SCREEN", "CLEAN" if bad == 0 else f"FAILED ({bad})")
