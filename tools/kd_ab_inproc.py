"""KD window A/B inside ONE process: the same trainer runs full windows alternately with and without an SL_* switch (re-read through
sl_tuning_reload between windows), so that box-to-box and launch-to-launch clock differences cancel:
    python tools/kd_ab_inproc.py SL_WGRAD_TR=1 [pairs] [samples per window]"""
import importlib, os, statistics, sys, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
mod = bench.mod
var, val = sys.argv[1].split("=")
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
window = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
L, ri, cfgm, weights = mod("_lib"), mod("random_init"), mod("config"), mod("weights")
enc_mod, llama_mod, utils, training = mod("audio_encoder"), mod("audio_llama"), mod("utils"), mod("training")
harch = weights.KNOWN_HUBERT["facebook/hubert-large-ls960-ft"]
larch = weights.KNOWN_LLAMA[utils.LLAMA_ID]
conf = cfgm.load_config(os.path.join(REPO, "config", "llama3_hubert.yaml"))
enc = enc_mod.AudioEncoder(conf, dev, dtype=torch.bfloat16, arch=harch)
enc.load_state_dict(ri.hubert_encoder_state_dict(harch, larch.hidden_size, seed=0)).eval().to(dev)
llm = llama_mod.AudioLlamaForCausalLM(larch, bench.gpu_llama_state_dict(larch, 0, dev), torch_dtype=torch.bfloat16, device=dev, max_ctx=512, max_batch=16)
prefix = ri.synthetic_ids(9, larch.vocab_size, seed=7, bos=larch.bos_token_id or 0)
suffix = ri.synthetic_ids(6, larch.vocab_size, seed=8, bos=larch.bos_token_id or 0)
conf.train["per_rank_accum"] = 0
tr = training.KDTrainer(conf, enc, llm, prefix, suffix, total_optimizer_steps=1000, regularizers=training.TrainRegularizers(seed=1234))
if window:
    tr.local_accum = window
if os.environ.get("KD_STACK_CHUNK"):      # 4 = the call boundaries a data-parallel rank keeps for its reducer (default: one call when nobody listens)
    tr.enc_tape.stack_chunk = int(os.environ["KD_STACK_CHUNK"])
B = tr.local_accum
g = torch.Generator().manual_seed(99)
text_ids = torch.randint(1, larch.vocab_size, (40,), generator=g)
resp_ids = torch.randint(1, larch.vocab_size, (64,), generator=g)
wave = ri.synthetic_waveform(160000, seed=4321).to(dev)
waves, texts, resps = [wave] * B, [text_ids] * B, [resp_ids] * B


def windows(n):
    out = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.micro_batch(waves, texts, resps)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) * 1e3)
    return out


windows(3)
res = {"default": [], sys.argv[1]: []}
for _ in range(pairs):
    for name in res:
        if var == "PY_STACK_CHUNK":      # a host-side setting, not a library switch: layers per sl_encoder_stack_train_bwd call (default: the whole stack when nobody listens)
            tr.enc_tape.stack_chunk = None if name == "default" else int(val)
        elif name == "default":
            os.environ.pop(var, None)
        else:
            os.environ[var] = val
        L.lib().sl_tuning_reload()
        windows(1)                      # one window to settle (workspace reuse, caches)
        res[name] += windows(3)
os.environ.pop(var, None)
L.lib().sl_tuning_reload()
for name, v in res.items():
    print(f"{name:>20s}: window of {B} samples  mean {statistics.mean(v):7.2f} ms  median {statistics.median(v):7.2f} ms  min {min(v):7.2f}  ({len(v)} windows)")
