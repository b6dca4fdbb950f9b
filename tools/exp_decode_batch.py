"""Experiment: decode step time vs batch with the packed streaming kernels and with the row-major tiled kernels.
    python tools/exp_decode_batch.py B[,B...] pack|nopack [new_tokens]"""
import os, sys, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
mod = bench.mod
Bs = [int(b) for b in sys.argv[1].split(",")]
pack = sys.argv[2] == "pack"
new = int(sys.argv[3]) if len(sys.argv) > 3 else 64
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
weights, llama_mod, utils = mod("weights"), mod("audio_llama"), mod("utils")
larch = weights.KNOWN_LLAMA[utils.LLAMA_ID]
S = 137
llm = llama_mod.AudioLlamaForCausalLM(larch, bench.gpu_llama_state_dict(larch, 0, dev), torch_dtype=torch.bfloat16, device=dev,
                                      max_ctx=((S + new + 63) // 64) * 64, max_batch=max(Bs), pack_decode=pack)
for B in Bs:
    x = (torch.randn(B * S, larch.hidden_size, device=dev) * 0.02).to(torch.bfloat16)
    llm._kv = None
    for _ in range(2):
        ids, n = llm.generate_packed(x.clone(), [S] * B, new, use_eos=False)
    pre, dec = llm.last_timings_ms
    print(f"B={B} pack={pack}: prefill {pre:.1f} ms, decode {dec / (new - 1):.3f} ms/step -> {B / (dec / (new - 1)) * 1e3:.0f} tok/s decode-only", flush=True)
