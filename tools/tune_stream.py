"""Streaming decode GEMM (gemm_stream.hip) sweep: the Llama-3.2-3B projections at batch M over the packed weights,
checked against a torch fp32 product and timed over rotating weight buffers.

    python tools/tune_stream.py 128,256,512 "default;1,4;2,4;4,2"      # SL_STREAM_CFG = "splits,nwv[,mt]" per config
"""
import importlib, itertools, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("llm-speech-summarization_amd.ops")
L = importlib.import_module("llm-speech-summarization_amd._lib")
dev = "cuda:0"
NBUF = int(os.environ.get("TUNE_NBUF", "6"))
shapes = [("qkv", 5120, 3072), ("o", 3072, 3072), ("gateup", 16384, 3072), ("down", 3072, 8192), ("lm_head", 128256, 3072)]
Ms = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [128]
cfgs = sys.argv[2].split(";") if len(sys.argv) > 2 else ["default"]
PAD = int(os.environ.get("TUNE_LDA_PAD", "0"))
os.environ["SL_STREAM_MIN_M"] = os.environ.get("SL_STREAM_MIN_M", "16")
L.lib().sl_tuning_reload()   # the library reads its tuning switches once; re-read after changing them
for name, N, K in shapes:
    nb = min(2, NBUF) if name == "lm_head" else NBUF
    Wr = [(torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16) for _ in range(nb)]
    Ws = [ops.pack_weight(w) for w in Wr]
    for M in Ms:
        A = torch.randn(M, K, device=dev).to(torch.bfloat16)
        if PAD:   # padded row stride: spreads the rows of one k-stage over the L2 channels
            big = torch.zeros(M, K + PAD, device=dev, dtype=torch.bfloat16)
            big[:, :K] = A
            A = big[:, :K]
        act = L.ACT_SILU_MUL if name == "gateup" else L.ACT_NONE
        ref = A.float() @ Wr[0].float().T
        if name == "gateup":   # rows interleaved in 16-row gate/up blocks
            r4 = ref.view(M, N // 32, 2, 16)
            ref = torch.nn.functional.silu(r4[:, :, 0]) * r4[:, :, 1]
            ref = ref.reshape(M, N // 2)
        for cfg in [c for c0 in cfgs for c in ((c0, c0 + "+rms") if name in ("gateup", "qkv") else (c0,))]:
            fuse = cfg.endswith("+rms")
            cfg = cfg[:-4] if fuse else cfg
            if cfg == "default":
                os.environ.pop("SL_STREAM_CFG", None)
                L.lib().sl_tuning_reload()   # the library reads its tuning switches once; re-read after changing them
            else:
                os.environ["SL_STREAM_CFG"] = cfg
                L.lib().sl_tuning_reload()   # the library reads its tuning switches once; re-read after changing them
            out = ops.gemm_decode(A, Ws[0], N, act=act, fuse_rms=fuse)
            rr = ref
            if fuse:
                rs = torch.rsqrt(A.float().pow(2).mean(-1, keepdim=True) + 1e-5)
                rr = ref * (rs * rs if name == "gateup" else rs) if name != "gateup" else None
            err = float((out.float() - rr).norm() / rr.norm()) if rr is not None else float("nan")
            for w in Ws:
                ops.gemm_decode(A, w, N, act=act, out=out, fuse_rms=fuse)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 5 * nb
            e0.record()
            for i in range(n):
                ops.gemm_decode(A, Ws[i % nb], N, act=act, out=out, fuse_rms=fuse)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / n * 1e3
            gbs = (N * K * 2 + M * K * 2 + out.numel() * 2) / us / 1e3
            print(f"{name:8s} M={M:4d} cfg={cfg + ('+rms' if fuse else ''):12s} {us:8.1f} us  {gbs:7.1f} GB/s  rel_err {err:.2e}", flush=True)
    del Ws, Wr
