"""Shape census of one KD window's GEMMs and what each shape costs alone.
    python tools/kd_gemm_shapes.py            # runs itself once with SL_GEMM_LOG=1 (one window), then times every distinct un-grouped shape
Prints: count per window, us per launch (same flags, random operands), TF/s, share of the window's GEMM time, 256-tile / 128-tile counts."""
import collections, importlib, os, re, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

if os.environ.get("KD_SHAPES_CHILD") == "1":
    import torch, bench
    P = "llm-speech-summarization_amd."
    ri, cfgm, weights, enc_mod, llama_mod, utils, training = [importlib.import_module(P + m) for m in ("random_init", "config", "weights", "audio_encoder", "audio_llama", "utils", "training")]
    L = importlib.import_module(P + "_lib")
    dev = torch.device("cuda:0")
    harch, larch = weights.KNOWN_HUBERT["facebook/hubert-large-ls960-ft"], weights.KNOWN_LLAMA[utils.LLAMA_ID]
    conf = cfgm.load_config(os.path.join(REPO, "config", "llama3_hubert.yaml"))
    enc = enc_mod.AudioEncoder(conf, dev, dtype=torch.bfloat16, arch=harch)
    enc.load_state_dict(ri.hubert_encoder_state_dict(harch, larch.hidden_size, seed=0)).eval().to(dev)
    llm = llama_mod.AudioLlamaForCausalLM(larch, bench.gpu_llama_state_dict(larch, 0, dev), torch_dtype=torch.bfloat16, device=dev, max_ctx=512, max_batch=16)
    prefix = ri.synthetic_ids(9, larch.vocab_size, seed=7, bos=128000); suffix = ri.synthetic_ids(6, larch.vocab_size, seed=8, bos=128000)
    tr = training.KDTrainer(conf, enc, llm, prefix, suffix, total_optimizer_steps=1000, regularizers=training.TrainRegularizers(seed=1234))
    g = torch.Generator().manual_seed(99)
    text_ids = torch.randint(1, larch.vocab_size, (40,), generator=g); resp_ids = torch.randint(1, larch.vocab_size, (64,), generator=g)
    wave = ri.synthetic_waveform(160000, seed=4321).to(dev)
    B = int(os.environ.get("KD_WINDOW", tr.local_accum))      # KD_WINDOW=2: the per-rank share of an 8-rank step
    tr.local_accum = B
    args = ([wave] * B, [text_ids] * B, [resp_ids] * B)
    for _ in range(2):
        tr.micro_batch(*args)
    torch.cuda.synchronize()
    os.environ["SL_GEMM_LOG"] = "1"; L.lib().sl_tuning_reload()
    sys.stderr.write("SLWINDOW begin\n"); sys.stderr.flush()
    tr.micro_batch(*args)
    torch.cuda.synchronize()
    sys.stderr.write("SLWINDOW end\n"); sys.stderr.flush()
    sys.exit(0)

out = subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, KD_SHAPES_CHILD="1"), capture_output=True, text=True)
lines, on = [], False
for ln in out.stderr.splitlines():
    if ln.startswith("SLWINDOW"):
        on = "begin" in ln
    elif on and ln.startswith("SLGEMM"):
        lines.append(ln)
if not lines:
    print(out.stderr[-3000:]); sys.exit(1)
census = collections.Counter(lines)
import torch
ops = importlib.import_module("llm-speech-summarization_amd.ops")
L = importlib.import_module("llm-speech-summarization_amd._lib")
dev = "cuda:0"
rows = []
for ln, cnt in census.items():
    f = {k: int(v) for k, v in re.findall(r"(\w+)=(-?\d+)", ln)}
    M, N, K = f["M"], f["N"], f["K"]
    if f["grp"] or f["packed"] or f["dt"] != L.dtype_code(torch.bfloat16) or f["batch"] != 1:
        rows.append((ln, cnt, None)); continue
    ta, tw = f["ta"], f["tw"]
    A = torch.randn((K, M) if ta else (M, K), device=dev).to(torch.bfloat16)
    W = (torch.randn((K, N) if tw else (N, K), device=dev) * K ** -0.5).to(torch.bfloat16)
    n_out = N // 2 if f["act"] == L.ACT_SILU_MUL else N
    out_t = torch.zeros((M, n_out), device=dev, dtype=torch.float32 if f["outf32"] else torch.bfloat16)
    res = None
    if f["res"]:
        res = out_t if f["resf32"] else torch.randn((M, n_out), device=dev).to(torch.bfloat16)
    kw = dict(M=M, N=N, K=K, lda=A.stride(0), ldw=W.stride(0), out=out_t, act=f["act"], out_f32=bool(f["outf32"]), trans_a=bool(ta), trans_w=bool(tw),
              residual=res, ldr=n_out if res is not None else 0, residual_f32=bool(f["resf32"]),
              bias=torch.randn(N, device=dev).to(torch.bfloat16) if f["bias"] else None,
              aux_out=torch.empty((M, n_out), device=dev, dtype=torch.bfloat16) if f["aux"] else None)
    try:
        for _ in range(3):
            ops.gemm_ex(A, W, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.gemm_ex(A, W, **kw)
        e1.record(); torch.cuda.synchronize()
        rows.append((ln, cnt, e0.elapsed_time(e1) / 10 * 1e3))
    except Exception as e:  # noqa: BLE001 - a census tool: report and go on
        rows.append((ln, cnt, None)); print("skip", ln, e)
tot = sum(c * u for _, c, u in rows if u)
print(f"{len(lines)} products in the window, {len(census)} distinct; timed shapes sum to {tot / 1e3:.1f} ms")
for ln, cnt, us in sorted(rows, key=lambda r: -(r[1] * (r[2] or 0))):
    f = {k: int(v) for k, v in re.findall(r"(\w+)=(-?\d+)", ln)}
    t256 = -(-f["M"] // 256) * -(-f["N"] // 256); t128 = -(-f["M"] // 128) * -(-f["N"] // 128)
    tf = f"{2.0 * f['M'] * f['N'] * f['K'] / us / 1e6:7.1f} TF/s {100 * cnt * us / tot:5.1f} %" if us else "   (not timed)"
    print(f"x{cnt:3d} {us or 0:8.1f} us {tf}  t256={t256:4d} t128={t128:4d}  {ln[7:]}")
