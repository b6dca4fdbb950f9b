import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import pkg, rel_err
ops = pkg("ops")
DEV = "cuda:0"
torch.set_printoptions(linewidth=200, precision=4, sci_mode=False)
def run(dt, S, D, mode):
    g = torch.Generator().manual_seed(1)
    nh = nkv = 1
    qkv = torch.randn(S, 3 * D, generator=g) * 0.5
    d_att = torch.randn(S, D, generator=g)
    if mode == "ones_dO":
        d_att = torch.ones(S, D)
    if mode == "same_q":
        qkv[:, :D] = qkv[0:1, :D]
        d_att = torch.ones(S, D)
    if mode == "q_one_dim":   # q nonzero only in dim J, k arbitrary: S[q][k] = q[q][J] k[k][J]
        J = 5
        z = torch.zeros(S, D); z[:, J] = qkv[:, J]; qkv[:, :D] = z
        d_att = torch.ones(S, D)
    if mode == "onehot":
        d_att = torch.zeros(S, D); d_att[torch.arange(S), torch.arange(S)] = 1.0
    if mode == "zero_q":
        qkv[:, :D] = 0
    if mode == "index_dO":     # dO[q][d] = q + 100 d : P uniform (zero q) -> dV[k][d] = mean_q = (S-1)/2 + 100 d
        qkv[:, :D] = 0
        d_att = torch.arange(S)[:, None].float() + 100 * torch.arange(D)[None, :].float()
    qd, dd = qkv.to(DEV, dt), d_att.to(DEV, dt)
    lse = torch.zeros(S, nh, device=DEV)
    out = ops.attn_packed_qkv(qd, [S], nh, nkv, D, False, D ** -0.5, lse=lse)
    d_qkv = torch.zeros_like(qd)
    ops.attn_packed_qkv_bwd(qd, out, dd, lse, d_qkv, [S], nh, nkv, D, False, D ** -0.5)
    x = qkv.clone().requires_grad_(True)
    sc = x[:, :D] @ x[:, D:2 * D].T * D ** -0.5
    P = torch.softmax(sc, -1)
    (P @ x[:, 2 * D:]).backward(d_att)
    dv, gv = d_qkv[:, 2 * D:].float().cpu(), x.grad[:, 2 * D:]
    print(mode, dt, "dv err", f"{rel_err(dv, gv):.3e}")
    Pg, Pr = dv[:, :S].T, P.detach()      # dV^T = P
    print(" got P rows 0..3\n", Pg[:4]); print(" ref P rows 0..3\n", Pr[:4])
    # which reference row does each computed row match best?
    for q_ in range(S):
        best = int(((Pr - Pg[q_][None]) ** 2).sum(1).argmin())
        # and try: computed row q_ = exp(S[a] - lse[b])
        sc_ = sc.detach(); lse_ = torch.logsumexp(sc_, -1)
        cand = [(float(((torch.exp(sc_[a] - lse_[b]) - Pg[q_]) ** 2).sum()), a, b) for a in range(S) for b in range(S)]
        print(q_, "nearest ref row", best, "best (err, S-row, lse-row)", min(cand))
for mode in ("onehot",):
    run(torch.float32, 16, 64, mode)
