"""Decode-step wall time per batch size through sl_greedy_generate (prefill excluded): tools/time_decode_step.py [B ...]"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
P = "llm-speech-summarization_amd."
weights, llama_mod, utils = [importlib.import_module(P + m) for m in ("weights", "audio_llama", "utils")]
dev = torch.device("cuda:0")
larch = weights.KNOWN_LLAMA[utils.LLAMA_ID]
Bs = [int(a) for a in sys.argv[1:]] or [1024, 512, 64, 16, 1]
S, new = 137, 128
llm = llama_mod.AudioLlamaForCausalLM(larch, bench.gpu_llama_state_dict(larch, 0, dev), torch_dtype=torch.bfloat16, device=dev, max_ctx=((S + new + 63) // 64) * 64, max_batch=max(Bs))
for B in Bs:
    llm._kv = None
    x = (torch.randn(B * S, larch.hidden_size, device=dev) * 0.02).to(torch.bfloat16)
    SP = int(os.environ.get('SHARED_PREFIX', '0'))   # timing only: the rows are random, so the promise does not hold and the ids are meaningless
    llm.generate_packed(x.clone(), [S] * B, new, use_eos=False, shared_prefix=SP)
    ts, tp = [], []
    for _ in range(3):
        llm.generate_packed(x.clone(), [S] * B, new, use_eos=False, shared_prefix=SP)
        ts.append(llm.last_timings_ms[1] / (new - 1)); tp.append(llm.last_timings_ms[0])
    print(f"B={B:5d}: decode step {sorted(ts)[1]:8.4f} ms  ({B / sorted(ts)[1]:9.1f} tok/ms)   prefill {sorted(tp)[1]:8.2f} ms", flush=True)
