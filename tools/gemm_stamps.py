"""In-kernel cycle stamps of the phased 256 x 256 GEMM (instrumented build, SL_GEMM_STAMP_PTR): where a K slab's four phases spend
their cycles, for the leading (waves 0-3) and the lagging (waves 4-7) half of a block.  Medians over all blocks, tiles 4 and 5.

    python tools/gemm_stamps.py [M N K]...
"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("llm-speech-summarization_amd.ops")
L = importlib.import_module("llm-speech-summarization_amd._lib")
dev = "cuda:0"
argv = [int(x) for x in sys.argv[1:]]
shapes = [tuple(argv[i:i + 3]) for i in range(0, len(argv), 3)] or [(127744, 3072, 1024), (140288, 5120, 3072), (127744, 1024, 4096)]
for M, N, K in shapes:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    nblk = ((M + 255) // 256) * ((N + 255) // 256)
    buf = torch.zeros(nblk * 2 * 32, device=dev, dtype=torch.int32)
    for _ in range(30):                      # clocks settle under load
        ops.gemm(A, W, out=out)
    os.environ["SL_GEMM_STAMP_PTR"] = hex(buf.data_ptr())
    L.lib().sl_tuning_reload()
    for _ in range(3):
        ops.gemm(A, W, out=out)
    torch.cuda.synchronize()
    os.environ.pop("SL_GEMM_STAMP_PTR")
    L.lib().sl_tuning_reload()
    s = buf.view(nblk, 2, 32).to(torch.int64) & 0xffffffff
    d = lambda a, b: ((s[:, :, a] - s[:, :, b]) & 0xffffffff).float()
    med = lambda x: [float(x[:, h].median()) for h in range(2)]
    print(f"--- {M} x {N} x {K}: {nblk} blocks, K slabs {K // 64}; cycles, median over blocks [leading half, lagging half]")
    print("  entry -> loop start      ", med(d(1, 0)))
    print("  whole loop               ", med(d(14, 1)), " per slab", [round(v / (K // 64), 1) for v in med(d(14, 1))], "(32 MFMA x 2 phases x 2 halves = 2048 at full rate)")
    for t in range(2):
        b = 2 + 6 * t
        for ph in range(2):
            o = b + 3 * ph
            print(f"  slab {4 + t} phase {ph}: reads+DMA+waits+barrier {med(d(o + 1, o))}  MFMA issue {med(d(o + 2, o + 1))}  closing barrier {med(d(o + 3, o + 2))}")
