"""Token-major weight-gradient products (gemm_tiled_tt_kernel) three ways, interleaved in one process: two-stage kernel on 512 block slots
(SL_TT_RING=0), ring form on 256 slots (default), ring form admitted but the split rule kept at 512 slots (SL_SPLITK_SLOTS=512: only launches
of <= 256 blocks take the ring).  Checks the three results against each other first.   python tools/time_wgrad_ring.py"""
import importlib, os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("llm-speech-summarization_amd.ops")
L = importlib.import_module("llm-speech-summarization_amd._lib")
dev = "cuda:0"
sk = ops.streamk_workspace(dev)
VARS = {"two-stage/512": {"SL_TT_RING": "0"}, "ring/256": {}, "ring/512": {"SL_SPLITK_SLOTS": "512"}}


def setv(name):
    for k in ("SL_TT_RING", "SL_SPLITK_SLOTS"):
        os.environ.pop(k, None)
    os.environ.update(VARS[name])
    L.lib().sl_tuning_reload()


shapes = [(7984, 1024, 1024), (7984, 3072, 1024), (7984, 4096, 1024), (7984, 1024, 4096), (998, 1024, 1024), (998, 3072, 1024), (998, 4096, 1024), (998, 1024, 4096),
          (634, 3072, 1024)]
for M, Nout, Kin in shapes:
    dY = torch.randn(M, Nout, device=dev).to(torch.bfloat16)
    X = torch.randn(M, Kin, device=dev).to(torch.bfloat16)
    bias = torch.zeros(Nout, device=dev)

    def tt(dW, cs=None):
        ops.gemm_ex(dY, X, M=Nout, N=Kin, K=M, lda=Nout, ldw=Kin, out=dW, ldc=Kin, residual=dW, ldr=Kin, out_f32=True, residual_f32=True,
                    trans_a=True, trans_w=True, dtype=dY.dtype, sk_ws=sk, colsum_out=cs)

    ref = dY.float().T @ X.float()
    outs = {}
    for v in VARS:
        setv(v)
        dW = torch.zeros(Nout, Kin, device=dev); cs = torch.zeros(Nout, device=dev)
        tt(dW, cs)
        err = float((dW - ref).norm() / ref.norm()); errb = float((cs - dY.float().sum(0)).norm() / dY.float().sum(0).norm())
        assert err < 2e-3 and errb < 2e-3, (M, Nout, Kin, v, err, errb)
        outs[v] = dW
    times = {v: [] for v in VARS}
    dW = torch.zeros(Nout, Kin, device=dev)
    for rnd in range(4):
        for v in VARS:
            setv(v)
            tt(dW)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                tt(dW)
            e1.record(); torch.cuda.synchronize()
            if rnd:
                times[v].append(e0.elapsed_time(e1) / 20 * 1e3)
    fl = 2.0 * M * Nout * Kin
    print(f"tokens={M:5d} out={Nout:5d} in={Kin:5d}: " + "  ".join(f"{v} {statistics.median(t):7.1f} us ({fl / statistics.median(t) / 1e6:5.0f} TF/s)" for v, t in times.items()), flush=True)
setv("ring/256")
for k in ("SL_TT_RING", "SL_SPLITK_SLOTS"):
    os.environ.pop(k, None)
L.lib().sl_tuning_reload()
