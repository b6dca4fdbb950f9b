"""GPU: backward / loss kernels of the KD step through the C ABI against PyTorch autograd (fp32 CPU reference).

Tolerances: fp32 2e-5 relative L2 (3e-4 where fp32 atomics reorder a long reduction), bf16 2e-2.
"""
import os
import pytest
import torch
import torch.nn.functional as F

from conftest import pkg, rel_err

pytestmark = pytest.mark.gpu

L = pkg("_lib")
ops = pkg("ops")
weights = pkg("weights")

DT = [torch.float32, torch.bfloat16]
TOL = {torch.float32: 2e-5, torch.bfloat16: 2e-2}
DEV = "cuda:0"


def rnd(*shape, seed=0, std=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * std


def q(x, dt):
    return x.to(dt).float()


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,N,K", [(137, 512, 256), (40, 96, 192), (300, 1024, 384)])
def test_gemm_dgrad_and_wgrad(dt, M, N, K):
    dY, W, X = rnd(M, N, seed=1), rnd(N, K, seed=2, std=N ** -0.5), rnd(M, K, seed=3)
    dX = ops.dgrad(dY.to(DEV, dt), W.to(DEV, dt))
    assert rel_err(dX.float().cpu(), q(dY, dt) @ q(W, dt)) < TOL[dt]
    dW = torch.full((N, K), 0.5, device=DEV, dtype=torch.float32)
    ops.wgrad_acc(dY.to(DEV, dt), X.to(DEV, dt), dW)
    assert rel_err(dW.cpu(), 0.5 + q(dY, dt).T @ q(X, dt)) < 2e-5  # fp32 accumulate in both modes


@pytest.mark.parametrize("dt", DT)
def test_wgrad_over_overlapping_conv_windows(dt):
    Lin, Cc, C2, k, s = 101, 64, 96, 3, 2
    Lo = (Lin - k) // s + 1
    x, dy = rnd(Lin, Cc, seed=4), rnd(Lo, C2, seed=5)
    xr = q(x, dt).T[None].requires_grad_()
    w = torch.zeros(C2, Cc, k, requires_grad=True)
    F.conv1d(xr, w, stride=s)[0].T.backward(q(dy, dt))
    ref_dw = w.grad.permute(0, 2, 1).reshape(C2, k * Cc)
    dW = torch.zeros(C2, k * Cc, device=DEV)
    ops.wgrad_acc(dy.to(DEV, dt), x.to(DEV, dt), dW, ldx=s * Cc, Kin=k * Cc, M=Lo)
    assert rel_err(dW.cpu(), ref_dw) < 2e-5
    # data gradient: dgrad GEMM -> col2im
    wt = rnd(C2, k * Cc, seed=6, std=0.1)
    wr = q(wt, dt).view(C2, k, Cc).permute(0, 2, 1).contiguous()
    xr2 = q(x, dt).T[None].requires_grad_()
    F.conv1d(xr2, wr, stride=s)[0].T.backward(q(dy, dt))
    dcol = ops.dgrad(dy.to(DEV, dt), wt.to(DEV, dt))
    dx = ops.col2im(dcol, Lin, Cc, k, s)
    assert rel_err(dx.float().cpu(), xr2.grad[0].T) < (TOL[dt] if dt == torch.float32 else 3e-2)


@pytest.mark.parametrize("dt", DT)
def test_gelu_silu_rope_backward(dt):
    u, dy = rnd(50, 256, seed=7), rnd(50, 256, seed=8)
    ur = q(u, dt).requires_grad_()
    F.gelu(ur).backward(q(dy, dt))
    assert rel_err(ops.gelu_bwd(dy.to(DEV, dt), u.to(DEV, dt)).float().cpu(), ur.grad) < TOL[dt]
    # SwiGLU on the interleaved layout
    M, Fd = 33, 128
    g, up, d = rnd(M, Fd, seed=9), rnd(M, Fd, seed=10), rnd(M, Fd, seed=11)
    gu = torch.stack([g.view(M, Fd // 16, 16), up.view(M, Fd // 16, 16)], dim=2).reshape(M, 2 * Fd)
    gr, ur2 = q(g, dt).requires_grad_(), q(up, dt).requires_grad_()
    out_ref = F.silu(gr) * ur2
    out_ref.backward(q(d, dt))
    assert rel_err(ops.silu_mul(gu.to(DEV, dt)).float().cpu(), out_ref.detach()) < TOL[dt]
    dgu = ops.silu_mul_bwd(gu.to(DEV, dt), d.to(DEV, dt)).float().cpu().view(M, Fd // 16, 2, 16)
    assert rel_err(dgu[:, :, 0].reshape(M, Fd), gr.grad) < TOL[dt] and rel_err(dgu[:, :, 1].reshape(M, Fd), ur2.grad) < TOL[dt]
    # RoPE: inverse rotation is the exact backward
    arch = weights.LlamaArch(hidden_size=256, num_attention_heads=2, num_key_value_heads=1, head_dim=128)
    cos, sin = weights.rope_tables(arch, 64)
    x = rnd(9, 4 * 128, seed=12)
    pos = torch.arange(9, dtype=torch.int32) * 5
    xd = x.to(DEV, dt).clone()
    ops.rope_inplace(xd, pos.to(DEV), cos.to(DEV), sin.to(DEV), 4, 3, 128)
    assert torch.equal(xd[:, 384:], x.to(DEV, dt)[:, 384:])  # 4th head untouched
    ops.rope_inplace(xd, pos.to(DEV), cos.to(DEV), sin.to(DEV), 4, 3, 128, inverse=True)
    assert rel_err(xd.float().cpu(), q(x, dt)) < (1e-6 if dt == torch.float32 else 1e-2)


@pytest.mark.parametrize("ws", [False, True])
@pytest.mark.parametrize("M,Nout,Kin,ldy,ldx", [(7984, 1024, 1024, 1024, 1024), (7984, 1024, 1024, 3072, 1024), (1000, 1024, 4096, 1024, 4096),
                                                   (256, 128, 128, 128, 136), (3999, 512, 512, 512, 512), (7984, 3072, 1024, 3072, 1024),
                                                   (130, 128, 256, 128, 256)])
def test_wgrad_token_major_operands(M, Nout, Kin, ldy, ldx, ws):
    """SL_WGRAD_TR=1 (the default): dW += dY^T X straight from the token-major operands (gemm_tiled_tt_kernel: LDS-DMA of [64 tokens][128 columns]
    slabs, fragments by transposing LDS reads, token tail from a zero constant, the reduction optionally cut into runs) against the
    fp32 product of the same bf16 values and against the default path through K-contiguous transposed copies."""
    dt = torch.bfloat16
    dYb = rnd(M, ldy, seed=41).to(DEV, dt)
    Xb = rnd(M, ldx, seed=42).to(DEV, dt)
    dW0 = rnd(Nout, Kin, seed=43).to(DEV)
    ref = dW0.double() + dYb[:, :Nout].double().t() @ Xb[:, :Kin].double()
    sk = ops.streamk_workspace(DEV) if ws else None

    def run():
        dW = dW0.clone()
        ops.gemm_ex(dYb, Xb, M=Nout, N=Kin, K=M, lda=ldy, ldw=ldx, out=dW, ldc=Kin, residual=dW, ldr=Kin, out_f32=True, residual_f32=True,
                    trans_a=True, trans_w=True, dtype=dt, sk_ws=sk)
        return dW

    os.environ["SL_WGRAD_TR"] = "0"
    try:
        L.lib().sl_tuning_reload()
        base = run()                  # the register-staged transposed loader these flags selected before
        os.environ["SL_WGRAD_TR"] = "1"
        L.lib().sl_tuning_reload()
        tt = run()
        tt2 = run()
    finally:
        del os.environ["SL_WGRAD_TR"]
        L.lib().sl_tuning_reload()
    assert torch.equal(tt, tt2)       # fixed summation order: reproducible
    assert rel_err(tt.double().cpu(), ref.cpu()) < 1e-5, rel_err(tt.double().cpu(), ref.cpu())
    assert rel_err(tt.cpu(), base.cpu()) < 1e-5
    if sk is not None:
        assert int(sk[:1024].to(torch.int32).sum().item()) == 0      # the flag area in front of the partial tiles stays zero


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("gelu", [False, True])
@pytest.mark.parametrize("rows,cols", [(131, 512), (70, 1024), (5, 128), (20000, 512), (3000, 2048)])
def test_layernorm_backward(dt, gelu, rows, cols):
    x, g, b, dy = rnd(rows, cols, seed=13), 1 + rnd(cols, seed=14, std=0.1), rnd(cols, seed=15, std=0.1), rnd(rows, cols, seed=16)
    xr, gr, br = q(x, dt).requires_grad_(), q(g, dt).requires_grad_(), q(b, dt).requires_grad_()
    y = F.layer_norm(xr, (cols,), gr, br, 1e-5)
    (F.gelu(y) if gelu else y).backward(q(dy, dt))
    dg = torch.full((cols,), 1.0, device=DEV)
    db = torch.zeros(cols, device=DEV)
    dx = ops.layernorm_bwd(x.to(DEV, dt), g.to(DEV, dt), b.to(DEV, dt), dy.to(DEV, dt), 1e-5, dg, db, gelu=gelu)
    assert rel_err(dx.float().cpu(), xr.grad) < TOL[dt]
    assert rel_err(dg.cpu() - 1.0, gr.grad) < 3e-4 and rel_err(db.cpu(), br.grad) < 3e-4
    # the scratch-free entry point (atomics) accumulates the same parameter gradients; the scratch form is reproducible bit for bit
    dg2, db2 = torch.full((cols,), 1.0, device=DEV), torch.zeros(cols, device=DEV)
    dx2 = torch.empty_like(dx)
    xd, gd, bd, dyd = x.to(DEV, dt), g.to(DEV, dt), b.to(DEV, dt), dy.to(DEV, dt)
    L.check(L.lib().sl_layernorm_bwd(L.ptr(xd), L.ptr(gd), L.ptr(bd), L.ptr(dyd), L.ptr(dx2), L.ptr(dg2), L.ptr(db2),
                                     rows, cols, 1e-5, int(gelu), L.dtype_code(dt), L.stream_ptr()), "sl_layernorm_bwd")
    assert torch.equal(dx2, dx) and rel_err(dg2.cpu() - 1.0, gr.grad) < 3e-4 and rel_err(db2.cpu(), br.grad) < 3e-4
    if cols <= 1024:
        dg3, db3 = torch.full((cols,), 1.0, device=DEV), torch.zeros(cols, device=DEV)
        ops.layernorm_bwd(x.to(DEV, dt), g.to(DEV, dt), b.to(DEV, dt), dy.to(DEV, dt), 1e-5, dg3, db3, gelu=gelu)
        assert torch.equal(dg3, dg) and torch.equal(db3, db)


@pytest.mark.parametrize("dt", DT)
def test_rmsnorm_backward_and_colsum(dt):
    x, w, dy = rnd(37, 768, seed=17), 1 + rnd(768, seed=18, std=0.1), rnd(37, 768, seed=19)
    xr = q(x, dt).requires_grad_()
    (q(w, dt) * (xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-5))).backward(q(dy, dt))
    assert rel_err(ops.rmsnorm_bwd(x.to(DEV, dt), w.to(DEV, dt), dy.to(DEV, dt), 1e-5).float().cpu(), xr.grad) < TOL[dt]
    out = torch.zeros(768, device=DEV)
    ops.colsum_acc(dy.to(DEV, dt), out)
    assert rel_err(out.cpu(), q(dy, dt).sum(0)) < 3e-5


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("causal", [False, True])
def test_softmax_rows_and_backward(dt, causal):
    n_mats, rows, cols, ld = 3, 37, 37, 40
    S = rnd(n_mats, rows, ld, seed=20)
    dP = rnd(n_mats, rows, ld, seed=21)
    Sr = S[:, :, :cols].clone().requires_grad_()
    sc = Sr * 0.3
    if causal:
        sc = sc.masked_fill(torch.arange(cols)[None, :] > torch.arange(rows)[:, None], float("-inf"))
    Pref = F.softmax(sc, dim=-1)
    P = ops.softmax_rows(S.to(DEV), n_mats, rows, cols, ld, 0.3, causal, dt)
    assert rel_err(P[:, :, :cols].float().cpu(), Pref.detach()) < (1e-6 if dt == torch.float32 else 5e-3)
    assert float(P[:, :, cols:].abs().max()) == 0
    # backward at the kernel's own P (rounded to dt)
    Pk = P[:, :, :cols].float().cpu()
    ref = 0.3 * Pk * (dP[:, :, :cols] - (dP[:, :, :cols] * Pk).sum(-1, keepdim=True))
    dS = ops.softmax_bwd(P, dP.to(DEV), cols, 0.3)
    assert rel_err(dS[:, :, :cols].float().cpu(), ref) < (1e-5 if dt == torch.float32 else 1e-2)


@pytest.mark.parametrize("dt", DT)
def test_kd_losses_and_gradients(dt):
    n, V = 7, 5003
    s, t = rnd(n, V, seed=22), rnd(n, V, seed=23)
    labels = torch.randint(0, V, (n,), generator=torch.Generator().manual_seed(24))
    sr = s.clone().requires_grad_()
    ce = F.cross_entropy(sr, labels, reduction="sum") * 0.25
    soft = (-(F.softmax(t, -1) * F.log_softmax(sr, -1)).sum(-1)).sum() * 0.5
    (ce + soft).backward()
    loss = torch.zeros(2, device=DEV)
    ds = torch.empty(n, V, device=DEV, dtype=dt)
    ops.ce_loss(s.to(DEV), labels.to(DEV, torch.int32), 0.25, loss[0:1], ds, accumulate=False, dtype=dt)
    ops.soft_ce_loss(s.to(DEV), t.to(DEV), 0.5, loss[1:2], ds, accumulate=True, dtype=dt)
    assert abs(float(loss[0]) - float(ce)) < 1e-4 * float(ce) and abs(float(loss[1]) - float(soft)) < 1e-4 * float(soft)
    assert rel_err(ds.float().cpu(), sr.grad) < (1e-5 if dt == torch.float32 else 1e-2)
    a, b = rnd(11, 256, seed=25), rnd(11, 256, seed=26)
    ar = q(a, dt).requires_grad_()
    m = F.mse_loss(ar, q(b, dt)) * 2.0
    m.backward()
    lm = torch.zeros(1, device=DEV)
    da = torch.empty(11, 256, device=DEV, dtype=dt)
    ops.mse_loss(a.to(DEV, dt), b.to(DEV, dt), 2.0, lm, da)
    assert abs(float(lm) - float(m)) < 1e-5 * float(m) and rel_err(da.float().cpu(), ar.grad) < TOL[dt]


@pytest.mark.parametrize("dt", DT)
def test_avgpool_backward(dt):
    T, H = 49, 128
    x, dy = rnd(T, H, seed=27), rnd(11, H, seed=28)
    xr = q(x, dt).T[None].requires_grad_()
    F.avg_pool1d(xr, 8, 4)[0].T.backward(q(dy, dt))
    assert rel_err(ops.avgpool_bwd(dy.to(DEV, dt), T, 8, 4).float().cpu(), xr.grad[0].T) < TOL[dt]


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("Cc,n", [(64, 4000), (512, 16000)])
def test_conv0_backward(dt, Cc, n):
    wave = rnd(n, seed=29, std=0.1)
    w, b = rnd(Cc, 1, 10, seed=30, std=0.4).requires_grad_(), rnd(Cc, seed=31, std=0.1).requires_grad_()
    g, be = (1 + rnd(Cc, seed=32, std=0.1)).requires_grad_(), rnd(Cc, seed=33, std=0.1).requires_grad_()
    Lo = (n - 10) // 5 + 1
    dy = rnd(Lo, Cc, seed=34)
    y = F.conv1d(wave[None, None], w, b, stride=5)[0].T
    F.gelu(F.layer_norm(y, (Cc,), g, be, 1e-5)).backward(q(dy, dt))
    dw, db, dg, dbe = [torch.zeros(s, device=DEV) for s in ((Cc, 10), (Cc,), (Cc,), (Cc,))]
    ops.hubert_conv0_bwd(wave.to(DEV), w.detach().reshape(Cc, 10).to(DEV), b.detach().to(DEV), g.detach().to(DEV), be.detach().to(DEV),
                         dy.to(DEV, dt), dw, db, dg, dbe)
    for got, ref in ((dw, w.grad.reshape(Cc, 10)), (db, b.grad), (dg, g.grad), (dbe, be.grad)):
        assert rel_err(got.cpu(), ref) < 5e-4


def _attn_autograd(qkv, d_att, seqlens, nh, nkv, D, causal, scale, dt, keep=None, p_drop=0.0):
    """fp32 softmax attention + autograd on the values the kernels see; keep: (N, nh, 65536) bool dropout mask or None."""
    x = q(qkv, dt).clone().requires_grad_(True)
    outs, o = [], 0
    rep = nh // nkv
    for S in seqlens:
        qh = x[o:o + S, :nh * D].view(S, nh, D).transpose(0, 1)
        kh = x[o:o + S, nh * D:(nh + nkv) * D].view(S, nkv, D).transpose(0, 1).repeat_interleave(rep, 0)
        vh = x[o:o + S, (nh + nkv) * D:].view(S, nkv, D).transpose(0, 1).repeat_interleave(rep, 0)
        sc = qh @ kh.transpose(1, 2) * scale
        if causal:
            sc = sc.masked_fill(torch.triu(torch.ones(S, S, dtype=torch.bool), 1), float("-inf"))
        pr = torch.softmax(sc, -1)
        if keep is not None:
            pr = pr * keep[o:o + S, :, :S].transpose(0, 1).float() / (1.0 - p_drop)
        outs.append((pr @ vh).transpose(0, 1).reshape(S, nh * D))
        o += S
    out = torch.cat(outs)
    out.backward(q(d_att, dt))
    return out.detach(), x.grad


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("seqlens", [[37, 5, 130, 64], [499, 1, 200], [16], [333]])
@pytest.mark.parametrize("nh,nkv,D,causal", [(6, 2, 128, True), (4, 4, 64, False), (4, 4, 64, True), (4, 2, 128, False), (3, 1, 128, True)])
def test_flash_attention_backward_ragged_vs_autograd(dt, seqlens, nh, nkv, D, causal):
    """sl_attn_bwd (probabilities recomputed per tile from the forward's log-sum-exp; dK / dV of a GQA group summed in-kernel)
    on a ragged packed batch against autograd through an fp32 softmax attention."""
    N = sum(seqlens)
    qkv = rnd(N, (nh + 2 * nkv) * D, seed=71, std=0.5)
    d_att = rnd(N, nh * D, seed=72)
    scale = D ** -0.5
    qd, dd = qkv.to(DEV, dt), d_att.to(DEV, dt)
    lse = torch.full((N, nh), float("nan"), device=DEV)
    out = ops.attn_packed_qkv(qd, seqlens, nh, nkv, D, causal, scale, lse=lse)
    d_qkv = torch.full_like(qd, float("nan"))
    ops.attn_packed_qkv_bwd(qd, out, dd, lse, d_qkv, seqlens, nh, nkv, D, causal, scale)
    assert bool(torch.isfinite(d_qkv.float()).all()) and bool(torch.isfinite(lse).all())
    ref_out, ref_grad = _attn_autograd(qkv, d_att, seqlens, nh, nkv, D, causal, scale, dt)
    assert rel_err(out.float().cpu(), ref_out) < TOL[dt]
    # lse against the definition
    x = q(qkv, dt)
    S0 = seqlens[0]
    sc = (x[:S0, :D] @ x[:S0, nh * D:nh * D + D].T) * scale
    if causal:
        sc = sc.masked_fill(torch.triu(torch.ones(S0, S0, dtype=torch.bool), 1), float("-inf"))
    assert float((lse[:S0, 0].cpu() - torch.logsumexp(sc, -1)).abs().max()) < (1e-4 if dt == torch.float32 else 2e-2)
    tol = 2e-5 if dt == torch.float32 else 3e-2
    for lo_, hi_ in ((0, nh * D), (nh * D, (nh + nkv) * D), ((nh + nkv) * D, (nh + 2 * nkv) * D)):     # dQ, dK, dV separately
        assert rel_err(d_qkv[:, lo_:hi_].float().cpu(), ref_grad[:, lo_:hi_]) < tol, (lo_, hi_)
    # bitwise reproducible (no atomics)
    d2 = torch.empty_like(qd)
    ops.attn_packed_qkv_bwd(qd, out, dd, lse, d2, seqlens, nh, nkv, D, causal, scale)
    assert torch.equal(d2, d_qkv)


@pytest.mark.parametrize("dt", DT)
def test_flash_attention_backward_with_probability_dropout(dt):
    """Training-mode attention dropout (hf HubertAttention): forward and backward rebuild the same counter-based mask; against
    autograd with the mask restated on the host."""
    nh, D, p_drop, seed = 4, 64, 0.25, 0x0123_4567_89AB_CDEF
    seqlens = [70, 133]
    N = sum(seqlens)
    qkv = rnd(N, 3 * nh * D, seed=73, std=0.5)
    d_att = rnd(N, nh * D, seed=74)
    qd, dd = qkv.to(DEV, dt), d_att.to(DEV, dt)
    lse = torch.empty((N, nh), device=DEV)
    out = ops.attn_packed_qkv(qd, seqlens, nh, nh, D, False, D ** -0.5, dropout_p=p_drop, dropout_seed=seed, lse=lse)
    d_qkv = torch.empty_like(qd)
    ops.attn_packed_qkv_bwd(qd, out, dd, lse, d_qkv, seqlens, nh, nh, D, False, D ** -0.5, dropout_p=p_drop, dropout_seed=seed)
    keep = ops.dropout_keep_mask((N * nh) << 16, p_drop, seed).view(N, nh, 1 << 16)
    ref_out, ref_grad = _attn_autograd(qkv, d_att, seqlens, nh, nh, D, False, D ** -0.5, dt, keep=keep, p_drop=p_drop)
    assert rel_err(out.float().cpu(), ref_out) < TOL[dt]
    assert rel_err(d_qkv.float().cpu(), ref_grad) < (2e-5 if dt == torch.float32 else 3e-2)


@pytest.mark.parametrize("dt", DT)
def test_dropout_kernel_matches_host_mask(dt):
    """sl_dropout: counter-based mask = its host restatement; scaling, residual form, in-place, seeds beyond 32 bits."""
    n, p, seed = 4096 * 24, 0.1, 0xA5A5_1234_5678_9ABC
    x = torch.randn(n, generator=torch.Generator().manual_seed(3)).to(DEV, dt)
    res = torch.randn(n, generator=torch.Generator().manual_seed(4)).to(DEV, dt)
    keep = ops.dropout_keep_mask(n, p, seed).to(DEV)
    assert abs(float(keep.float().mean()) - 0.9) < 0.01
    y = ops.dropout(x, p, seed)
    scale = 1.0 / (1.0 - float(torch.tensor(p, dtype=torch.float32)))
    ref = torch.where(keep, x.float() * scale, torch.zeros_like(x.float()))
    assert torch.equal(y == 0, ~keep | (x == 0))
    assert rel_err(y.float().cpu(), ref.cpu()) < TOL[dt]
    y2 = ops.dropout(x, p, seed, residual=res)
    assert rel_err(y2.float().cpu(), (res.float() + ref).cpu()) < TOL[dt]
    z = x.clone()
    ops.dropout(z, p, seed, out=z)
    assert torch.equal(z, y)
    assert not torch.equal(ops.dropout(x, p, seed + (1 << 40)) == 0, y == 0)   # the high word of the seed matters


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("rows,cols,ld_out", [(100, 72, 128), (257, 64, 257), (64, 200, 64), (7984, 1024, 8000)])
def test_transpose_pad(dt, rows, cols, ld_out):
    vec = 4 if dt == torch.float32 else 8
    x = torch.randn(rows + 3, cols + vec, generator=torch.Generator().manual_seed(rows)).to(DEV, dt)     # a view with a wider row stride
    y = ops.transpose_pad(x[:, :cols], rows, cols, ld_out)
    assert y.shape == (cols, ld_out)
    assert torch.equal(y[:, :rows], x[:rows, :cols].t())
    assert not bool(y[:, rows:].any())


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("V", [1000, 777])        # 16-byte vector path / scalar path (rows not 16-byte aligned)
def test_kd_window_losses_one_launch_equal_torch(dt, V):
    """sl_kd_logit_losses / sl_kd_mse_rows: every utterance's next-token CE, soft CE and feature MSE terms of an accumulation
    window in single launches, against torch (ref:model/audio_llama.py:72-101, ref:utils.py:167-178, ref:trainer.py:358-370)."""
    ns = [5, 9, 2]
    rows = sum(ns)
    s, tch = rnd(rows, V, seed=81, std=2.0), rnd(rows, V, seed=82, std=2.0)
    gen = torch.Generator().manual_seed(83)
    resp = [torch.randint(0, V, (n,), generator=gen) for n in ns]
    w_ntp, w_ld, w_fd, acc = 0.5, 0.25, 1.0, 16.0
    sr = s.clone().requires_grad_()
    o, ce, soft = 0, [], []
    for n, r in zip(ns, resp):
        ce.append(F.cross_entropy(sr[o:o + n - 1], r[1:]))
        soft.append((-(F.softmax(tch[o:o + n], -1) * F.log_softmax(sr[o:o + n], -1)).sum(-1)).mean())
        o += n
    (sum(w_ntp * c + w_ld * so for c, so in zip(ce, soft)) / acc).backward()
    labels = torch.cat([torch.cat([r[1:], torch.tensor([-1])]) for r in resp]).to(DEV, torch.int32)
    coef = torch.tensor([[1 / (n - 1), w_ntp / acc / (n - 1), 1 / n, w_ld / acc / n] for n in ns for _ in range(n)], device=DEV)
    slot = torch.tensor([u for u, n in enumerate(ns) for _ in range(n)], dtype=torch.int32, device=DEV)
    losses = torch.zeros(len(ns), 3, device=DEV)
    ds = torch.full((rows, V), float("nan"), device=DEV, dtype=dt)
    ops.kd_logit_losses(s.to(DEV), tch.to(DEV), labels, coef, slot, losses, ds, dt)
    for u in range(len(ns)):
        assert abs(float(losses[u, 0]) - float(ce[u])) < 1e-4 * float(ce[u]) and abs(float(losses[u, 1]) - float(soft[u])) < 1e-4 * float(soft[u])
    assert rel_err(ds.float().cpu(), sr.grad) < (1e-5 if dt == torch.float32 else 1e-2)
    # without a teacher: CE only
    losses2 = torch.zeros(len(ns), 3, device=DEV)
    ds2 = torch.empty((rows, V), device=DEV, dtype=dt)
    ops.kd_logit_losses(s.to(DEV), None, labels, coef, slot, losses2, ds2, dt)
    sr2 = s.clone().requires_grad_()
    o = 0
    tot = 0
    for n, r in zip(ns, resp):
        tot = tot + w_ntp / acc * F.cross_entropy(sr2[o:o + n - 1], r[1:])
        o += n
    tot.backward()
    assert rel_err(ds2.float().cpu(), sr2.grad) < (1e-5 if dt == torch.float32 else 1e-2) and float(losses2[:, 1].abs().max()) == 0
    # feature MSE of one tap
    H = 256
    a, b = rnd(rows, H, seed=84), rnd(rows, H, seed=85)
    ar = q(a, dt).requires_grad_()
    o, mses = 0, []
    for n in ns:
        mses.append(F.mse_loss(ar[o:o + n], q(b, dt)[o:o + n]))
        o += n
    (sum(mses) * w_fd / acc).backward()
    mcoef = torch.tensor([[1 / (n * H), 2 * w_fd / acc / (n * H)] for n in ns for _ in range(n)], device=DEV)
    da = torch.empty(rows, H, device=DEV, dtype=dt)
    ops.kd_mse_rows(a.to(DEV, dt), b.to(DEV, dt), mcoef, slot, losses, 2, da)
    for u in range(len(ns)):
        assert abs(float(losses[u, 2]) - float(mses[u])) < 1e-5 * float(mses[u])
    assert rel_err(da.float().cpu(), ar.grad) < TOL[dt]


def test_fused_adamw_equals_torch_optim_adamw():
    """sl_adamw_step against torch.optim.AdamW itself (the reference's optimizer, ref:trainer.py:97-105: default eps 1e-8 and
    weight_decay 1e-2) over three steps on tensors of awkward sizes (tails, unaligned views), with the compute-dtype copy written
    by the same launch: bf16 copy == RNE cast of the updated master, fp32 copy == the master."""
    training = pkg("training")
    sizes = [(512, 1024), (3,), (4097,), (1, 1, 10), (777,), (8192 + 5,)]
    gen = torch.Generator().manual_seed(3)
    ref_p = [torch.nn.Parameter((torch.randn(*s, generator=gen) * 0.05).to(DEV)) for s in sizes]
    base = torch.zeros(20000, device=DEV)
    views = [base[1:1 + 3], base[101:101 + 777]]                 # fp32 tensors that do not start on a 16-byte boundary
    my_p = []
    for i, p in enumerate(ref_p):
        if p.numel() == 3:
            views[0].copy_(p.detach()); my_p.append(torch.nn.Parameter(views[0]))
        elif p.numel() == 777:
            views[1].copy_(p.detach()); my_p.append(torch.nn.Parameter(views[1]))
        else:
            my_p.append(torch.nn.Parameter(p.detach().clone()))
    ref_opt = torch.optim.AdamW(ref_p, lr=5e-5, betas=(0.9, 0.999))
    my_opt = torch.optim.AdamW(my_p, lr=5e-5, betas=(0.9, 0.999))
    sched = torch.optim.lr_scheduler.PolynomialLR(my_opt, total_iters=10, power=1.0)
    ref_sched = torch.optim.lr_scheduler.PolynomialLR(ref_opt, total_iters=10, power=1.0)
    dst_bf16 = torch.zeros(512 * 1024 + 64, device=DEV, dtype=torch.bfloat16)
    dst_f32 = torch.zeros(4097, device=DEV, dtype=torch.float32)
    dst_odd = torch.zeros(8192 + 5 + 3, device=DEV, dtype=torch.bfloat16)
    fused = training.FusedAdamW(my_opt, [(f"p{i}", p) for i, p in enumerate(my_p)],
                                {"p0": (dst_bf16, 64), "p2": (dst_f32, 0), "p5": (dst_odd, 3)})     # p5's copy starts 6 bytes into the buffer
    for step in range(3):
        for rp, mp in zip(ref_p, my_p):
            g = torch.randn(rp.shape, generator=gen).to(DEV) * (10.0 ** (step - 2))
            rp.grad, mp.grad = g.clone(), g.clone()
        ref_opt.step(); ref_sched.step()
        fused.step(); sched.step()
        torch.cuda.synchronize()
        for i, (rp, mp) in enumerate(zip(ref_p, my_p)):
            assert rel_err(mp.detach().cpu(), rp.detach().cpu()) < 1e-6, (step, i)
            for key in ("exp_avg", "exp_avg_sq"):
                assert rel_err(my_opt.state[mp][key].cpu(), ref_opt.state[rp][key].cpu()) < 1e-6, (step, i, key)
            assert float(my_opt.state[mp]["step"]) == float(ref_opt.state[rp]["step"]) == step + 1
        assert torch.equal(dst_bf16[64:], my_p[0].detach().reshape(-1).to(torch.bfloat16))
        assert torch.equal(dst_f32, my_p[2].detach().reshape(-1))
        assert torch.equal(dst_odd[3:], my_p[5].detach().reshape(-1).to(torch.bfloat16))
        assert float(dst_bf16[:64].abs().max()) == 0 and float(dst_odd[:3].abs().max()) == 0
    # the state is torch's own: the reference optimizer loads it
    ref_opt.load_state_dict(my_opt.state_dict())


def test_weight_norm_backward_of_the_positional_conv_vs_autograd():
    """sl_weight_norm_bwd: d g and d v of W = g v / ||v|| (norm over all but the tap axis, hf weight_norm(dim = 2)) from the tape's
    gradient of the folded weight in the kernel layout (H, k, Hg), against autograd through torch's own parametrisation formula;
    HuBERT-large's shape and an odd one; bitwise reproducible."""
    import importlib
    L_ = importlib.import_module("llm-speech-summarization_amd._lib")
    for H, Hg, k in ((1024, 64, 128), (48, 12, 16)):
        gen = torch.Generator().manual_seed(H + k)
        v = torch.randn(H, Hg, k, generator=gen, dtype=torch.float64).requires_grad_()
        g = (1.0 + 0.1 * torch.randn(1, 1, k, generator=gen, dtype=torch.float64)).requires_grad_()
        dW = torch.randn(H, Hg, k, generator=gen, dtype=torch.float64)
        W = g * v / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()
        W.backward(dW)
        dWk = dW.permute(0, 2, 1).contiguous().float().cuda()            # the tape's layout (H, k, Hg)
        vd, gd = v.detach().float().cuda(), g.detach().float().cuda()
        ws = torch.empty(int(L_.lib().sl_weight_norm_bwd_workspace_bytes(k)) // 4, device="cuda")
        outs = []
        for _ in range(2):
            dg, dv = torch.empty_like(gd), torch.empty_like(vd)
            L_.check(L_.lib().sl_weight_norm_bwd(L_.ptr(dWk), L_.ptr(vd), L_.ptr(gd), L_.ptr(dg), L_.ptr(dv), L_.ptr(ws), H, Hg, k, L_.stream_ptr()), "sl_weight_norm_bwd")
            outs.append((dg.cpu(), dv.cpu()))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        assert float((outs[0][0].double() - g.grad).norm() / g.grad.norm()) < 2e-6
        assert float((outs[0][1].double() - v.grad).norm() / v.grad.norm()) < 2e-6


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,N,K,sk", [(3200, 1024, 1024, False), (998, 1024, 4096, True), (400, 1024, 4096, False), (5072, 4096, 1024, False), (300, 192, 256, False)])
def test_gemm_epilogue_dropout_equals_the_unfused_launches(dt, M, N, K, sk):
    """sl_gemm_ex_args.post_op = SL_POST_DROPOUT (ABI 7): C = residual + dropout(act(A W^T + bias)) in the GEMM epilogue is bit for bit
    sl_gemm -> sl_dropout(residual) — the HuBERT layer's `h = h + dropout(sublayer)` (hf:models/hubert/modeling_hubert.py:529-540 under
    ref:trainer.py:258 train()) on the 128-tile, 256-tile and split-K (reduce pass) forms; and with GELU + the saved pre-activation
    (FFN1: mid = dropout(gelu(pre)), pre kept for the backward)."""
    A, W, b, R = (rnd(M, K, seed=1).to(DEV, dt), (rnd(N, K, seed=2) * K ** -0.5).to(DEV, dt), rnd(N, seed=3).to(DEV, dt), rnd(M, N, seed=4).to(DEV, dt))
    p_, seed = 0.1, 0x1234567890ABCDEF
    ws = ops.streamk_workspace(DEV) if sk else None
    ref = ops.dropout(ops.gemm_ex(A, W, M=M, N=N, K=K, lda=K, ldw=K, out=torch.empty((M, N), device=DEV, dtype=dt), bias=b, sk_ws=ws), p_, seed, residual=R)
    out = ops.gemm_ex(A, W, M=M, N=N, K=K, lda=K, ldw=K, out=torch.empty((M, N), device=DEV, dtype=dt), bias=b, residual=R, ldr=N, sk_ws=ws,
                      post_op=L.POST_DROPOUT, drop_p=p_, drop_seed=seed, drop_ld=N)
    assert torch.equal(out, ref)
    keep = ops.dropout_keep_mask(M * N, p_, seed).view(M, N)
    assert abs(float((out - R == 0).float().mean()) - p_) < 0.02 and bool(((out.cpu() - R.cpu() == 0) | keep).all())
    if not sk:
        pre_ref = torch.empty((M, N), device=DEV, dtype=dt)
        mid_ref = ops.gemm_ex(A, W, M=M, N=N, K=K, lda=K, ldw=K, out=torch.empty((M, N), device=DEV, dtype=dt), bias=b, act=L.ACT_GELU, aux_out=pre_ref)
        mid_ref = ops.dropout(mid_ref, p_, seed + 1)
        pre = torch.empty((M, N), device=DEV, dtype=dt)
        mid = ops.gemm_ex(A, W, M=M, N=N, K=K, lda=K, ldw=K, out=torch.empty((M, N), device=DEV, dtype=dt), bias=b, act=L.ACT_GELU, aux_out=pre,
                          post_op=L.POST_DROPOUT, drop_p=p_, drop_seed=seed + 1, drop_ld=N)
        assert torch.equal(mid, mid_ref) and torch.equal(pre, pre_ref)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,N,K,p_", [(3200, 4096, 1024, 0.1), (998, 4096, 1024, 0.0), (400, 1024, 1024, 0.1), (5072, 4096, 1024, 0.1), (200, 192, 256, 0.1)])
def test_gemm_epilogue_gelu_backward_and_bias_gradient_equal_the_unfused_launches(dt, M, N, K, p_):
    """post_op = SL_POST_GELU_BWD + colsum_out: d pre = gelu'(pre) * dropout(dY W) and db += colsum(d pre) behind the data-gradient
    product are sl_gemm -> sl_dropout (in place) -> sl_gelu_bwd -> sl_colsum: the values bit for bit, the column sums to fp32 atomics'
    ordering (hf:models/hubert/modeling_hubert.py:467-474 HubertFeedForward backward under ref:trainer.py:373)."""
    dY, Wt, pre = (rnd(M, K, seed=5).to(DEV, dt), (rnd(N, K, seed=6) * K ** -0.5).to(DEV, dt), rnd(M, N, seed=7).to(DEV, dt))
    seed = 0xFEDCBA0987654321
    d_mid = ops.gemm_ex(dY, Wt, M=M, N=N, K=K, lda=K, ldw=K, out=torch.empty((M, N), device=DEV, dtype=dt))
    if p_ > 0:
        ops.dropout(d_mid, p_, seed, out=d_mid)
    ref = ops.gelu_bwd(d_mid, pre)
    db_ref = ops.colsum_acc(ref, torch.full((N,), 0.5, device=DEV, dtype=torch.float32))
    db = torch.full((N,), 0.5, device=DEV, dtype=torch.float32)
    out = ops.gemm_ex(dY, Wt, M=M, N=N, K=K, lda=K, ldw=K, out=torch.empty((M, N), device=DEV, dtype=dt), post_op=L.POST_GELU_BWD, drop_p=p_,
                      drop_seed=seed, drop_ld=N, post_in=pre, post_ld=N, colsum_out=db)
    assert torch.equal(out, ref)
    assert rel_err(db.cpu(), db_ref.cpu()) < 3e-4
    assert rel_err(db.cpu() - 0.5, out.float().sum(0).cpu()) < (3e-4 if dt == torch.float32 else 2e-3)
    # colsum_out without a post-op: the bias gradient of a plain product's stored values
    db2 = torch.zeros((N,), device=DEV, dtype=torch.float32)
    plain = ops.gemm_ex(dY, Wt, M=M, N=N, K=K, lda=K, ldw=K, out=torch.empty((M, N), device=DEV, dtype=dt), colsum_out=db2)
    assert torch.equal(plain, ops.gemm_ex(dY, Wt, M=M, N=N, K=K, lda=K, ldw=K, out=torch.empty((M, N), device=DEV, dtype=dt)))
    assert rel_err(db2.cpu(), plain.float().sum(0).cpu()) < (3e-4 if dt == torch.float32 else 2e-3)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,F_,K,sk", [(634, 8192, 3072, False), (634, 1024, 8192, True), (3200, 2048, 1024, False), (130, 64, 256, False)])
def test_gemm_epilogue_swiglu_backward_equals_the_unfused_launches(dt, M, F_, K, sk):
    """post_op = SL_POST_SILU_MUL_BWD: d [gate | up] (M, 2 F, interleaved in 16-column blocks) = silu_mul_bwd(gu, dX Wdown^T-stored) written by
    the down projection's data-gradient product itself (hf:models/llama/modeling_llama.py:187-189 LlamaMLP backward, frozen weights:
    ref:trainer.py:63-64) — bit for bit sl_gemm -> sl_silu_mul_bwd, on the tile epilogues and the split-K reduce pass."""
    dX, Wt, gu = (rnd(M, K, seed=8).to(DEV, dt), (rnd(F_, K, seed=9) * K ** -0.5).to(DEV, dt), rnd(M, 2 * F_, seed=10).to(DEV, dt))
    ws = ops.streamk_workspace(DEV) if sk else None
    d_mid = ops.gemm_ex(dX, Wt, M=M, N=F_, K=K, lda=K, ldw=K, out=torch.empty((M, F_), device=DEV, dtype=dt), sk_ws=ws)
    ref = ops.silu_mul_bwd(gu, d_mid)
    out = ops.gemm_ex(dX, Wt, M=M, N=F_, K=K, lda=K, ldw=K, out=torch.full((M, 2 * F_), 7.0, device=DEV, dtype=dt), ldc=2 * F_, sk_ws=ws,
                      post_op=L.POST_SILU_MUL_BWD, post_in=gu, post_ld=2 * F_)
    assert torch.equal(out, ref)


def test_gemm_epilogue_fusions_refuse_the_forms_they_do_not_cover():
    A, W = rnd(128, 64).to(DEV, torch.bfloat16), rnd(128, 64).to(DEV, torch.bfloat16)
    out = torch.empty((128, 128), device=DEV, dtype=torch.bfloat16)
    with pytest.raises(L.SpeechLLMError):      # a mask needs its index stride
        ops.gemm_ex(A, W, M=128, N=128, K=64, lda=64, ldw=64, out=out, post_op=L.POST_DROPOUT, drop_p=0.1, drop_seed=1, drop_ld=0)
    with pytest.raises(L.SpeechLLMError):      # transposed operands keep the plain epilogue
        ops.gemm_ex(A, W, M=128, N=128, K=64, lda=64, ldw=64, out=out, trans_w=True, post_op=L.POST_DROPOUT, drop_p=0.1, drop_seed=1, drop_ld=128)
    with pytest.raises(L.SpeechLLMError):      # GELU backward without the saved pre-activation
        ops.gemm_ex(A, W, M=128, N=128, K=64, lda=64, ldw=64, out=out, post_op=L.POST_GELU_BWD)
    with pytest.raises(L.SpeechLLMError):
        ops.gemm_ex(A, W, M=128, N=128, K=64, lda=64, ldw=64, out=out, post_op=9)


@pytest.mark.parametrize("ws", [False, True])
@pytest.mark.parametrize("M,Nout,Kin,ldy", [(3200, 1024, 1024, 1024), (7984, 3072, 1024, 3072), (998, 1024, 4096, 1024), (400, 4096, 1024, 4096), (3200, 1024, 4096, 3072)])
def test_wgrad_token_major_product_carries_the_bias_gradient(M, Nout, Kin, ldy, ws):
    """colsum_out with both operands transposed (ABI 7): db += colsum(dY) comes out of the weight-gradient product itself (the first column
    tile's left-half waves multiply their dY fragments with a fragment of ones) — against sl_colsum on the same dY, with and without K runs,
    on a column slice of a wider buffer (d_qkv -> bqkv); dW keeps the bits it has without the rider."""
    dY_full = rnd(M, ldy, seed=21).to(DEV, torch.bfloat16)
    dY = dY_full[:, :Nout]
    X = rnd(M, Kin, seed=22).to(DEV, torch.bfloat16)
    sk = ops.streamk_workspace(DEV) if ws else None

    def run(db):
        dW = torch.full((Nout, Kin), 0.25, device=DEV, dtype=torch.float32)
        ops.gemm_ex(dY, X, M=Nout, N=Kin, K=M, lda=ldy, ldw=Kin, out=dW, ldc=Kin, residual=dW, ldr=Kin, out_f32=True, residual_f32=True, trans_a=True, trans_w=True,
                    dtype=torch.bfloat16, sk_ws=sk, colsum_out=db)
        return dW

    db = torch.full((Nout,), 0.5, device=DEV, dtype=torch.float32)
    with_rider, plain = run(db), run(None)
    assert torch.equal(with_rider, plain)
    ref = ops.colsum_acc(dY, torch.full((Nout,), 0.5, device=DEV, dtype=torch.float32))
    assert rel_err(db.cpu(), ref.cpu()) < 3e-4
    assert rel_err(db.cpu() - 0.5, dY.float().sum(0).cpu()) < 1e-3
    # ops.wgrad_acc(db=...) takes this path for such shapes and the sl_colsum launch otherwise (a row count below the kernel's range)
    small = torch.zeros((Nout,), device=DEV, dtype=torch.float32)
    ops.wgrad_acc(dY[:100], X[:100], torch.zeros((Nout, Kin), device=DEV, dtype=torch.float32), db=small)
    assert rel_err(small.cpu(), dY[:100].float().sum(0).cpu()) < 1e-3


@pytest.mark.parametrize("M,Nout,Kin,ldy", [(998, 1024, 1024, 1024), (998, 3072, 1024, 3072), (7984, 1024, 1024, 1024), (7984, 4096, 1024, 4096), (1000, 1024, 512, 3072),
                                            (8001, 1024, 1024, 1024), (640, 2048, 2048, 2048), (130, 1024, 1024, 1024)])
def test_wgrad_token_major_ring_form_gives_the_two_stage_kernels_bits(M, Nout, Kin, ldy):
    """gemm_tiled_tt_ring_kernel (launches of at most one block per CU: four stages of DMA, fragment reads pipelined across the barrier, asm
    MFMAs with the DMA requests between them) against gemm_tiled_tt_kernel on the same K runs (SL_SPLITK_SLOTS pins the split rule to 256 block
    slots for both): dW and the bias-gradient rider bit for bit, with and without the workspace; token tails inside a slab (998, 1 000, 8 001
    rows), a last run shorter than the others (8 001 rows: 32 / 32 / 32 / 30 slabs), 130 rows (three slabs: below the ring's admission of eight slabs per run, both switches
    run the two-stage kernel), and the fp64 product."""
    dY_full = rnd(M, ldy, seed=51).to(DEV, torch.bfloat16)
    dY = dY_full[:, :Nout]
    X = rnd(M, Kin, seed=52).to(DEV, torch.bfloat16)
    sk = ops.streamk_workspace(DEV)
    ref = dY.double().t() @ X.double()

    def run(ws):
        dW = torch.zeros((Nout, Kin), device=DEV, dtype=torch.float32)
        db = torch.zeros((Nout,), device=DEV, dtype=torch.float32)
        ops.gemm_ex(dY, X, M=Nout, N=Kin, K=M, lda=ldy, ldw=Kin, out=dW, ldc=Kin, residual=dW, ldr=Kin, out_f32=True, residual_f32=True, trans_a=True, trans_w=True,
                    dtype=torch.bfloat16, sk_ws=ws, colsum_out=db)
        return dW, db

    res = {}
    os.environ["SL_SPLITK_SLOTS"] = "256"
    try:
        for ring in ("0", "4"):
            os.environ["SL_GLDS_RING"] = ring
            L.lib().sl_tuning_reload()
            res[ring] = [run(sk), run(None), run(sk)]
    finally:
        del os.environ["SL_SPLITK_SLOTS"], os.environ["SL_GLDS_RING"]
        L.lib().sl_tuning_reload()
    for (w0, b0), (w1, b1) in zip(res["0"], res["4"]):
        assert torch.equal(w0, w1)
        assert rel_err(b1.cpu(), b0.cpu()) < 1e-6          # atomics across K runs: the order of the adds is not fixed
    assert torch.equal(res["4"][0][0], res["4"][2][0])
    assert rel_err(res["4"][0][0].double().cpu(), ref.cpu()) < 1e-5
    assert rel_err(res["4"][0][1].cpu(), dY.float().sum(0).cpu()) < 1e-3


@pytest.mark.parametrize("T_,G,Hg,k", [(499, 16, 64, 128), (130, 4, 64, 128), (1000, 2, 128, 64)])
def test_batched_token_major_weight_gradient_of_the_positional_conv(T_, G, Hg, k):
    """The positional conv's weight gradient (hf:models/hubert/modeling_hubert.py:113-135 walked backwards): per group g of Hg channels
    dW[g][n][j Hg + c] += sum_t d_pre[t][g Hg + n] * xg[g][t + j][c] — G products of (Hg x T) . (T x k Hg) whose second operand is a set of OVERLAPPING
    windows (row stride Hg, row length k Hg).  Round 6 runs them as ONE batched launch of the token-major kernel (batch index on blockIdx.z; with
    Hg = 64 the tile's upper 64 rows read a zero constant); SL_TT_BATCHED=0 is the register-staged loader it replaced.  Against each other and
    against the fp64 sum, accumulating on top of a non-zero dW."""
    H = G * Hg
    dt = torch.bfloat16
    d_pre = rnd(T_, H, seed=71).to(DEV, dt)
    xg = rnd(G, T_ + k, Hg, seed=72).to(DEV, dt)
    dW0 = rnd(G, Hg, k * Hg, seed=73).to(DEV)
    ref = dW0.double().clone()
    for g in range(G):
        win = torch.stack([xg[g, j:j + T_].double() for j in range(k)], 1).reshape(T_, k * Hg)      # [t][j Hg + c]
        ref[g] += d_pre[:, g * Hg:(g + 1) * Hg].double().t() @ win

    def run():
        dW = dW0.clone()
        ops.gemm_ex(d_pre, xg, M=Hg, N=k * Hg, K=T_, lda=H, ldw=Hg, out=dW, ldc=k * Hg, residual=dW, ldr=k * Hg, out_f32=True, residual_f32=True,
                    trans_a=True, trans_w=True, batch=G, strideA=Hg, strideW=(T_ + k) * Hg, strideC=Hg * k * Hg, strideR=Hg * k * Hg, dtype=dt)
        return dW

    os.environ["SL_TT_BATCHED"] = "0"
    try:
        L.lib().sl_tuning_reload()
        base = run()
    finally:
        del os.environ["SL_TT_BATCHED"]
        L.lib().sl_tuning_reload()
    tt, tt2 = run(), run()
    assert torch.equal(tt, tt2)
    assert rel_err(tt.double().cpu(), ref.cpu()) < 1e-5 and rel_err(base.double().cpu(), ref.cpu()) < 1e-5
    assert rel_err(tt.cpu(), base.cpu()) < 1e-5


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,N,K", [(998, 1024, 4096), (634, 3072, 16384), (400, 1024, 4096)])
def test_gemm_deferred_split_k_partials_sum_to_the_reduced_product(dt, M, N, K):
    """sl_gemm_ex_args.deferred_splits: for a few-tile product under a long reduction the library leaves the S fp32 partial products in sk_ws
    (byte offset 1 024, S slabs of M x N) and skips its reduce pass; summed in run order and rounded to the storage type they are bit for bit
    the product with the reduce pass (what the norm-backward kernels of the KD tapes do while loading dY).  A product the split rule does not
    take reports 0 and writes C as usual."""
    import ctypes as C
    A, W = rnd(M, K, seed=31).to(DEV, dt), (rnd(N, K, seed=32) * K ** -0.5).to(DEV, dt)
    ws = ops.streamk_workspace(DEV)
    ref = ops.gemm_ex(A, W, M=M, N=N, K=K, lda=K, ldw=K, out=torch.empty((M, N), device=DEV, dtype=dt), sk_ws=ws)
    S = C.c_int32(-1)
    out = torch.full((M, N), 3.0, device=DEV, dtype=dt)
    ops.gemm_ex(A, W, M=M, N=N, K=K, lda=K, ldw=K, out=out, sk_ws=ws, deferred_splits=S)
    assert S.value >= 2, S.value
    parts = ws[1024:1024 + S.value * M * N * 4].view(torch.float32).view(S.value, M, N)
    acc = parts[0].clone()
    for z in range(1, S.value):
        acc += parts[z]
    assert torch.equal(acc.to(dt), ref)
    assert bool((out == 3.0).all())                       # C untouched: the consumer owns the sum
    S2 = C.c_int32(-1)                                    # a short reduction: no K runs, the ordinary product
    small = ops.gemm_ex(A[:, :512].contiguous(), W[:, :512].contiguous(), M=M, N=N, K=512, lda=512, ldw=512, out=torch.empty((M, N), device=DEV, dtype=dt),
                        sk_ws=ws, deferred_splits=S2)
    assert S2.value == 0 and torch.equal(small, ops.gemm_ex(A[:, :512].contiguous(), W[:, :512].contiguous(), M=M, N=N, K=512, lda=512, ldw=512,
                                                             out=torch.empty((M, N), device=DEV, dtype=dt)))


@pytest.mark.parametrize("M,N,K,S_want", [(3200, 3072, 16384, 3), (3200, 3072, 9216, 3), (3100, 3072, 12352, 3), (3200, 3072, 5120, 0), (5072, 3072, 16384, 0)])
def test_gemm_256_tile_k_runs_of_uneven_length_left_to_the_consumer(M, N, K, S_want):
    """gemm.hip splitk256_runs (round 6): a 256-tile product that fills 50-66 % of one round of the chip (3 200 x 3 072: 156 tiles) under a long
    reduction, whose caller sums the K runs itself (deferred_splits: the RMSNorm backward of the LLM tape), runs as S = 3 runs of whole slabs
    of UNEVEN length (K = 16 384: 86 / 86 / 84 slabs; 12 352: 65 / 65 / 63) on the phased 256-tile kernel: the fp32 partial products in
    run order sum to the product (fp64 reference on a row sample; against the unsplit bf16 result to a rounding), C is untouched.  A caller
    without deferred_splits gets the reduce launch: the same bits.  Not taken: runs shorter than 48 slabs (K = 5 120), tile counts that fill the chip
    (5 072 rows: 240 tiles), SL_SPLIT_K256=0."""
    import ctypes as C
    dt = torch.bfloat16
    A, W = rnd(M, K, seed=61).to(DEV, dt), (rnd(N, K, seed=62) * K ** -0.5).to(DEV, dt)
    ws = ops.streamk_workspace(DEV)
    plain = ops.gemm_ex(A, W, M=M, N=N, K=K, lda=K, ldw=K, out=torch.empty((M, N), device=DEV, dtype=dt), sk_ws=ws)      # no deferred_splits: the same runs + the reduce launch
    S = C.c_int32(-1)
    out = torch.full((M, N), 3.0, device=DEV, dtype=dt)
    ops.gemm_ex(A, W, M=M, N=N, K=K, lda=K, ldw=K, out=out, sk_ws=ws, deferred_splits=S)
    if S_want == 0:
        assert S.value == 0 and torch.equal(out, plain)
        return
    assert S.value == S_want, S.value
    parts = ws[1024:1024 + S.value * M * N * 4].view(torch.float32).view(S.value, M, N)
    acc = parts[0].clone()
    for z in range(1, S.value):
        acc += parts[z]
    idx = torch.randint(0, M, (256,), generator=torch.Generator().manual_seed(5)).to(DEV)
    ref = A[idx].double() @ W.double().t()
    assert rel_err(acc[idx].double().cpu(), ref.cpu()) < 2e-6
    assert torch.equal(acc.to(dt), plain) and bool((out == 3.0).all())      # the reduce launch adds the runs in the same order
    nows = ops.gemm_ex(A, W, M=M, N=N, K=K, lda=K, ldw=K, out=torch.empty((M, N), device=DEV, dtype=dt))                  # no workspace: one run per tile
    assert rel_err(plain.float().cpu(), nows.float().cpu()) < 4e-3
    ops.gemm_ex(A, W, M=M, N=N, K=K, lda=K, ldw=K, out=out, sk_ws=ws, deferred_splits=S)          # again on the same workspace: the same bits
    assert torch.equal(parts[0] + parts[1] + parts[2], acc)
    os.environ["SL_SPLIT_K256"] = "0"
    try:
        L.lib().sl_tuning_reload()
        S0 = C.c_int32(-1)
        off = ops.gemm_ex(A, W, M=M, N=N, K=K, lda=K, ldw=K, out=torch.empty((M, N), device=DEV, dtype=dt), sk_ws=ws, deferred_splits=S0)
        assert S0.value == 0 and torch.equal(off, nows)
    finally:
        del os.environ["SL_SPLIT_K256"]
        L.lib().sl_tuning_reload()


@pytest.mark.parametrize("dt", DT)
def test_batched_col2im_and_avgpool_backward_equal_the_per_utterance_launches(dt):
    """sl_col2im_batch / sl_avgpool_bwd_batch (ABI 7): the strided-conv data gradient's fold-back and the AvgPool1d backward for every utterance of a packed
    ragged batch in one launch each, written straight into the packed rows — bit for bit the per-utterance launches (hf:models/hubert/modeling_hubert.py:141
    conv layers, ref:model/audio_encoder.py:37-41 pooling, walked backwards by ref:trainer.py:373)."""
    Cc, k, s = 64, 3, 2
    Lin = [401, 37, 999, 5]
    Lout = [(n - k) // s + 1 for n in Lin]
    so, do = [0], [0]
    for a, b in zip(Lout, Lin):
        so.append(so[-1] + a); do.append(do[-1] + b)
    dcol = rnd(so[-1], k * Cc, seed=41).to(DEV, dt)
    out = torch.full((do[-1], Cc), 9.0, device=DEV, dtype=dt)
    ops.col2im_batch(dcol, out, Lout, Lin, so[:-1], do[:-1], Cc, k, s)
    for u in range(len(Lin)):
        assert torch.equal(out[do[u]:do[u + 1]], ops.col2im(dcol[so[u]:so[u + 1]], Lin[u], Cc, k, s)), u
    H, kernel, stride = 128, 8, 4
    T = [499, 12, 123, 8]
    P = [(t_ - kernel) // stride + 1 for t_ in T]
    po, to = [0], [0]
    for a, b in zip(P, T):
        po.append(po[-1] + a); to.append(to[-1] + b)
    dy = rnd(po[-1], H, seed=42).to(DEV, dt)
    dx = torch.full((to[-1], H), 9.0, device=DEV, dtype=dt)
    ops.avgpool_bwd_batch(dy, dx, P, T, po[:-1], to[:-1], kernel, stride)
    for u in range(len(T)):
        assert torch.equal(dx[to[u]:to[u + 1]], ops.avgpool_bwd(dy[po[u]:po[u + 1]], T[u], kernel, stride)), u
