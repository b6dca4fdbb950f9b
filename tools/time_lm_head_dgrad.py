"""lm_head data gradient of the KD window (1024 tail rows, V = 128 256, H = 3072): one transposed-operand product vs vocabulary slices.
    python tools/time_lm_head_dgrad.py"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("llm-speech-summarization_amd.ops")
training = importlib.import_module("llm-speech-summarization_amd.training")
dev = "cuda:0"
n, V, H = 1024, 128256, 3072
dY = (torch.randn(n, V, device=dev) * 1e-2).to(torch.bfloat16)
Wm = (torch.randn(V, H, device=dev) * H ** -0.5).to(torch.bfloat16)
class W_: pass
class T_(training.LlamaTape):
    def __init__(self): self.w = W_(); self.w.lm_head = Wm
t = T_()
ref = dY.float() @ Wm.float()
for S in (1, 2, 3, 4, 6, 12):
    T_.LM_HEAD_DGRAD_SPLITS = (S,)
    for _ in range(3): out = t._lm_head_dgrad(dY)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): out = t._lm_head_dgrad(dY)
    e1.record(); torch.cuda.synchronize()
    err = float((out.float() - ref).norm() / ref.norm())
    print(f"splits {S:2d}: {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us   rel err {err:.2e}", flush=True)
