"""Packed-sequence attention forward (attention.hip) at the encoder / prefill shapes: time, TFLOP/s, and the
maximum deviation from a torch fp32 softmax(QK^T)V on a few sequences.  SL_ATTN_GENERIC=1 selects the dtype-generic kernel."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("llm-speech-summarization_amd.ops")
dev = "cuda:0"
torch.manual_seed(0)
cases = [("hubert 512 x 499, 16 heads, D 64", [499] * 512, 16, 16, 64, False),
         ("hubert ragged", [99 + 37 * (i % 40) for i in range(256)], 16, 16, 64, False),
         ("llama prefill 512 x 136, 24/8 heads, D 128 causal", [136] * 512, 24, 8, 128, True),
         ("llama prefill ragged causal", [20 + 23 * (i % 30) for i in range(256)], 24, 8, 128, True)]
for name, lens, nh, nkv, D, causal in cases:
    ntok = sum(lens)
    qkv = (torch.randn(ntok, (nh + 2 * nkv) * D, device=dev) * 1.0).to(torch.bfloat16)
    out = ops.attn_packed_qkv(qkv, lens, nh, nkv, D, causal, D ** -0.5)
    # check the first, a middle and the last sequence
    err = 0.0
    offs = [0]
    for l in lens:
        offs.append(offs[-1] + l)
    for si in (0, len(lens) // 2, len(lens) - 1):
        x = qkv[offs[si]:offs[si + 1]].float()
        q = x[:, :nh * D].view(-1, nh, D).transpose(0, 1)
        k = x[:, nh * D:(nh + nkv) * D].view(-1, nkv, D).transpose(0, 1).repeat_interleave(nh // nkv, 0)
        v = x[:, (nh + nkv) * D:].view(-1, nkv, D).transpose(0, 1).repeat_interleave(nh // nkv, 0)
        s = q @ k.transpose(1, 2) * D ** -0.5
        if causal:
            s = s.masked_fill(torch.ones(s.shape[-2:], device=dev, dtype=torch.bool).triu(1), float("-inf"))
        ref = (s.softmax(-1) @ v).transpose(0, 1).reshape(-1, nh * D)
        err = max(err, float((out[offs[si]:offs[si + 1]].float() - ref).abs().max()))
    for _ in range(2):
        ops.attn_packed_qkv(qkv, lens, nh, nkv, D, causal, D ** -0.5)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.attn_packed_qkv(qkv, lens, nh, nkv, D, causal, D ** -0.5)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    fl = sum(4.0 * l * l * D * nh * (0.5 if causal else 1.0) for l in lens)
    print(f"{name:52s} {us:9.1f} us  {fl / us / 1e6:7.1f} TF/s  max |err| {err:.2e}", flush=True)
