"""Self-consistency of the full-size models at large batch: B copies of one utterance must give B identical embedding blocks
and B identical id rows (catches 32-bit index overflow in any kernel: activations exceed 2^31 elements at these sizes).

    python tools/check_large_batch.py [B=512] [audio_sec=10]
"""
import importlib, os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
mod = bench.mod
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
sec = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
L, ri, cfgm, weights = mod("_lib"), mod("random_init"), mod("config"), mod("weights")
enc_mod, llama_mod, utils = mod("audio_encoder"), mod("audio_llama"), mod("utils")
harch = weights.KNOWN_HUBERT["facebook/hubert-large-ls960-ft"]
larch = weights.KNOWN_LLAMA[utils.LLAMA_ID]
conf = cfgm.load_config(os.path.join(REPO, "config", "llama3_hubert.yaml"))
enc = enc_mod.AudioEncoder(conf, dev, dtype=torch.bfloat16, arch=harch)
enc.load_state_dict(ri.hubert_encoder_state_dict(harch, larch.hidden_size, seed=0)).eval().to(dev)
n = int(sec * 16000)
P = (harch.num_frames(n) - 8) // 4 + 1
wave = ri.synthetic_waveform(n, seed=1).to(dev)
new = 24
S = 9 + P + 4
llm = llama_mod.AudioLlamaForCausalLM(larch, bench.gpu_llama_state_dict(larch, 0, dev), torch_dtype=torch.bfloat16, device=dev,
                                      max_ctx=((S + new + 63) // 64) * 64, max_batch=B)
emb = llm.model.embed_tokens
pre = emb(ri.synthetic_ids(9, larch.vocab_size, seed=7).to(dev))[0]
suf = emb(ri.synthetic_ids(5, larch.vocab_size, seed=8).to(dev))[0, 1:]
x = torch.empty((B * S, larch.hidden_size), device=dev, dtype=torch.bfloat16)
xv = x.view(B, S, -1)
xv[:, :9] = pre
xv[:, 9 + P:] = suf
enc.encode_packed([wave] * B, out=x, out_row_offsets=[b * S + 9 for b in range(B)])
torch.cuda.synchronize()
ref = xv[0].clone()
bad = [b for b in range(B) if not torch.equal(xv[b], ref)]
print(f"encoder: B={B} x {sec:.0f} s, rows differing from utterance 0: {len(bad)} {bad[:8]}", flush=True)
ids, ncols = llm.generate_packed(x, [S] * B, new, use_eos=False)
bad2 = [b for b in range(B) if not torch.equal(ids[b], ids[0])]
print(f"decode: id rows differing from sequence 0: {len(bad2)} {bad2[:8]}; ids[0][:8] = {ids[0][:8].tolist()}", flush=True)
assert not bad and not bad2
print("OK")
