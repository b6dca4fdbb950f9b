"""Knock-out timings of the phased 256 x 256 GEMM (library built with `make DEBUG_GEMM=1`): which of {fragment reads, DMA, MFMAs}
sets the loop's pace.  Results of the knocked-out builds are meaningless; only their time is read.  Interleaved rounds, median.

    python tools/gemm_knockout.py [M N K]...
"""
import importlib, os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("llm-speech-summarization_amd.ops")
L = importlib.import_module("llm-speech-summarization_amd._lib")
dev = "cuda:0"
argv = [int(x) for x in sys.argv[1:]]
shapes = [tuple(argv[i:i + 3]) for i in range(0, len(argv), 3)] or [(127744, 3072, 1024), (140288, 5120, 3072), (127744, 1024, 4096)]
names = {0: "full", 1: "no reads", 2: "no DMA", 3: "no reads, no DMA", 4: "no MFMA", 6: "no DMA, no MFMA"}
for M, N, K in shapes:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    times = {k: [] for k in names}
    for rnd in range(6):
        for k in names:
            os.environ["SL_GEMM_KO"] = str(k)
            L.lib().sl_tuning_reload()
            ops.gemm(A, W, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                ops.gemm(A, W, out=out)
            e1.record(); torch.cuda.synchronize()
            if rnd:
                times[k].append(e0.elapsed_time(e1) / 8 * 1e3)
    os.environ.pop("SL_GEMM_KO")
    L.lib().sl_tuning_reload()
    print(f"--- {M} x {N} x {K}")
    for k in names:
        us = statistics.median(times[k])
        print(f"  {names[k]:>18}: {us:8.1f} us  ({2.0 * M * N * K / us / 1e6:6.0f} TF/s-equivalent)")
