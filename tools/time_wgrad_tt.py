"""Weight-gradient products of a KD window (tokens x features) two ways: K-contiguous transposed copies + the LDS-DMA tile kernels (default),
and token-major operands read by transposing LDS loads (SL_WGRAD_TR=1, gemm_tiled_tt_kernel): python tools/time_wgrad_tt.py"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("llm-speech-summarization_amd.ops")
L = importlib.import_module("llm-speech-summarization_amd._lib")
dev = "cuda:0"
sk = ops.streamk_workspace(dev)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M, Nout, Kin in ((7984, 1024, 1024), (7984, 3072, 1024), (7984, 4096, 1024), (7984, 1024, 4096), (998, 1024, 1024), (998, 4096, 1024), (3200, 1024, 1024)):
    dY = torch.randn(M, Nout, device=dev).to(torch.bfloat16)
    X = torch.randn(M, Kin, device=dev).to(torch.bfloat16)
    dW = torch.zeros(Nout, Kin, device=dev)
    Mp = (M + 127) // 128 * 128

    def copies():
        yt = ops.transpose_pad(dY, M, Nout, Mp)
        xt = ops.transpose_pad(X, M, Kin, Mp)
        ops.gemm_ex(yt, xt, M=Nout, N=Kin, K=Mp, lda=Mp, ldw=Mp, out=dW, ldc=Kin, residual=dW, ldr=Kin, out_f32=True, residual_f32=True, dtype=dY.dtype, sk_ws=sk)

    def tt():
        ops.gemm_ex(dY, X, M=Nout, N=Kin, K=M, lda=Nout, ldw=Kin, out=dW, ldc=Kin, residual=dW, ldr=Kin, out_f32=True, residual_f32=True,
                    trans_a=True, trans_w=True, dtype=dY.dtype, sk_ws=sk)

    t_c = timed(copies)
    os.environ["SL_WGRAD_TR"] = "1"; L.lib().sl_tuning_reload()
    t_t = timed(tt)
    os.environ["SL_WGRAD_TR"] = "0"; L.lib().sl_tuning_reload()
    t_r = timed(tt)      # the register-staged loader these flags selected before
    del os.environ["SL_WGRAD_TR"]; L.lib().sl_tuning_reload()
    fl = 2.0 * M * Nout * Kin
    print(f"tokens={M:5d} out={Nout:5d} in={Kin:5d}: copies+tiles {t_c:7.1f} us ({fl / t_c / 1e6:6.0f} TF/s)  token-major {t_t:7.1f} us ({fl / t_t / 1e6:6.0f} TF/s)  "
          f"register-staged {t_r:7.1f} us", flush=True)
