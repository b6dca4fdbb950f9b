"""Which torch (aten) ops still run inside a KD window, and for how long: one window under torch.profiler.
    python tools/prof_kd_ops.py"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
P = "llm-speech-summarization_amd."
ri, cfgm, weights, enc_mod, llama_mod, utils, training = [importlib.import_module(P + m) for m in ("random_init", "config", "weights", "audio_encoder", "audio_llama", "utils", "training")]
dev = torch.device("cuda:0")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
for _ in range(3):
    tr.micro_batch(*args)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.micro_batch(*args)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=50, max_shapes_column_width=70))
