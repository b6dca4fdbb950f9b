import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import pkg, rel_err
ops = pkg("ops")
DEV = "cuda:0"
def run(dt, seqlens, nh, nkv, D, causal):
    g = torch.Generator().manual_seed(1)
    N = sum(seqlens)
    qkv = torch.randn(N, (nh + 2 * nkv) * D, generator=g) * 0.5
    d_att = torch.randn(N, nh * D, generator=g)
    qd, dd = qkv.to(DEV, dt), d_att.to(DEV, dt)
    lse = torch.zeros(N, nh, device=DEV)
    out = ops.attn_packed_qkv(qd, seqlens, nh, nkv, D, causal, D ** -0.5, lse=lse)
    d_qkv = torch.zeros_like(qd)
    ops.attn_packed_qkv_bwd(qd, out, dd, lse, d_qkv, seqlens, nh, nkv, D, causal, D ** -0.5)
    x = qkv.to(dt).float().clone().requires_grad_(True)
    outs, o = [], 0
    rep = nh // nkv
    lses = []
    for S in seqlens:
        qh = x[o:o + S, :nh * D].view(S, nh, D).transpose(0, 1)
        kh = x[o:o + S, nh * D:(nh + nkv) * D].view(S, nkv, D).transpose(0, 1).repeat_interleave(rep, 0)
        vh = x[o:o + S, (nh + nkv) * D:].view(S, nkv, D).transpose(0, 1).repeat_interleave(rep, 0)
        sc = qh @ kh.transpose(1, 2) * D ** -0.5
        if causal:
            sc = sc.masked_fill(torch.triu(torch.ones(S, S, dtype=torch.bool), 1), float("-inf"))
        lses.append(torch.logsumexp(sc, -1).transpose(0, 1))
        outs.append((torch.softmax(sc, -1) @ vh).transpose(0, 1).reshape(S, nh * D))
        o += S
    ref_out = torch.cat(outs)
    ref_out.backward(d_att.to(dt).float())
    gq, gk, gv = x.grad[:, :nh * D], x.grad[:, nh * D:(nh + nkv) * D], x.grad[:, (nh + nkv) * D:]
    dq, dk, dv = d_qkv[:, :nh * D].float().cpu(), d_qkv[:, nh * D:(nh + nkv) * D].float().cpu(), d_qkv[:, (nh + nkv) * D:].float().cpu()
    print(dt, seqlens, nh, nkv, D, causal, "out", f"{rel_err(out.float().cpu(), ref_out.detach()):.2e}", "lse", f"{float((lse.cpu() - torch.cat(lses)).abs().max()):.2e}",
          "dq", f"{rel_err(dq, gq):.2e}", "dk", f"{rel_err(dk, gk):.2e}", "dv", f"{rel_err(dv, gv):.2e}")
    if rel_err(dv, gv) > 1e-2 and N <= 16:
        print("dv ratio", (dv / gv)[:4, :8]); print("dq", dq[:4, :6], gq[:4, :6])
for dt in (torch.float32, torch.bfloat16):
    run(dt, [16], 1, 1, 64, False)
    run(dt, [16], 1, 1, 128, False)
    run(dt, [64], 1, 1, 64, False)
    run(dt, [37, 5, 130], 4, 2, 128, True)
