"""Diagnostic: gate/up streaming GEMM at M rows with parts of the main loop knocked out (SL_KO bits: 1 weight reloads,
2 x DMA, 4 barriers, 8 LDS reads, 16 MFMAs); prints wall time and in-kernel cycle / clock stamps."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dbg = torch.zeros(4096 * 8, dtype=torch.int64, device="cuda:0")
os.environ["SL_KO_DBG"] = hex(dbg.data_ptr())
ops = importlib.import_module("llm-speech-summarization_amd.ops")
L = importlib.import_module("llm-speech-summarization_amd._lib")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
H, F_ = 3072, 8192
ws = [ops.pack_weight((torch.randn(2 * F_, H, device="cuda:0") * 0.02).to(torch.bfloat16)) for _ in range(6)]
x = torch.randn(M, H, device="cuda:0").to(torch.bfloat16)
out = torch.empty(M, F_, device="cuda:0", dtype=torch.bfloat16)
for _ in range(200):   # settle the clock
    ops.gemm_decode(x, ws[0], 2 * F_, act=L.ACT_SILU_MUL, out=out)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(60):
    ops.gemm_decode(x, ws[i % 6], 2 * F_, act=L.ACT_SILU_MUL, out=out)
e1.record(); torch.cuda.synchronize()
d = dbg.view(-1, 2)[: 512 * 4].cpu().double()
cyc, rt = d[:, 0], d[:, 1]
print(f"KO={os.environ.get('SL_KO')}  {e0.elapsed_time(e1) / 60 * 1e3:7.1f} us   loop cycles/wave median {cyc.median():9.0f}  per stage {cyc.median() / 48:7.1f}   clock {float((cyc / rt).median()) * 100:6.0f} MHz")
