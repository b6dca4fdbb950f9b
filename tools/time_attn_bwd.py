"""Attention backward (sl_attn_bwd: delta + dK/dV + dQ kernels) on the shapes of a KD window, timed alone:
    python tools/time_attn_bwd.py            # HuBERT layer (16 x 499 frames, 16 heads, D 64, dropout 0.1), Llama layer (16 x 200, 24 / 8 heads, D 128, causal),
                                             # and the per-rank window's (2 sequences) of both
A/B of two builds: SL_DEV=1 SL_LIB_PATH=tools/ab/libspeechllm_{base,new}.so (tools/exp_ab_attn_bwd.sh)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("llm-speech-summarization_amd.ops")
dev = "cuda:0"
dt = torch.bfloat16
cases = [("hubert 16 x 499, 16 heads, D 64, dropout 0.1", [499] * 16, 16, 16, 64, False, 0.1),
         ("llama  16 x 200, 24 / 8 heads, D 128, causal", [200] * 16, 24, 8, 128, True, 0.0),
         ("hubert  2 x 499", [499] * 2, 16, 16, 64, False, 0.1),
         ("llama   2 x 317 (teacher + student rows)", [317] * 2, 24, 8, 128, True, 0.0)]
for name, lens, nh, nkv, D, causal, pdrop in cases:
    n = sum(lens)
    qkv = (torch.randn(n, (nh + 2 * nkv) * D, device=dev) * 0.5).to(dt)
    lse = torch.empty(n, nh, device=dev, dtype=torch.float32)
    out = ops.attn_packed_qkv(qkv, lens, nh, nkv, D, causal, D ** -0.5, dropout_p=pdrop, dropout_seed=7, lse=lse)
    d_out = (torch.randn(n, nh * D, device=dev) * 0.1).to(dt)
    d_qkv = torch.empty_like(qkv)
    fn = lambda: ops.attn_packed_qkv_bwd(qkv, out, d_out, lse, d_qkv, lens, nh, nkv, D, causal, D ** -0.5, dropout_p=pdrop, dropout_seed=7)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for rnd in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    flops = 7 * 2.0 * sum(l * l for l in lens) * D * nh * (0.5 if causal else 1.0)
    t = sorted(ts)[2]
    print(f"{name:50s} {t:8.1f} us   {flops / t / 1e6:7.1f} TF/s   checksum {float(d_qkv.float().abs().sum()):.6e}", flush=True)
