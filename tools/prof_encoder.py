"""Profile helper: HuBERT-large encoder only, 16 x 10 s, bf16 (run under rocprofv3 --kernel-trace --stats)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = "llm-speech-summarization_amd."
ri, cfgm, weights, enc_mod = [importlib.import_module(P + m) for m in ("random_init", "config", "weights", "audio_encoder")]
dev = "cuda:0"
harch = weights.KNOWN_HUBERT["facebook/hubert-large-ls960-ft"]
conf = cfgm.load_config(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "config", "llama3_hubert.yaml"))
enc = enc_mod.AudioEncoder(conf, dev, dtype=torch.bfloat16, arch=harch)
enc.load_state_dict(ri.hubert_encoder_state_dict(harch, 3072, seed=0)).eval().to(dev)
waves = [ri.synthetic_waveform(160000, seed=i).to(dev) for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 16)]
for _ in range(2):
    enc.encode_packed(waves)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    enc.encode_packed(waves)
e1.record(); torch.cuda.synchronize()
print("encode ms", e0.elapsed_time(e1) / 5, "audio-s/s", len(waves) * 10 / (e0.elapsed_time(e1) / 5e3))
