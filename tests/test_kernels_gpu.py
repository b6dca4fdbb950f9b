"""GPU: each HIP kernel, called through the C ABI, against a plain PyTorch fp32 reference of the same op.

Tolerances (relative L2): fp32 mode 2e-5 (exact-fp32 MFMA, only summation order differs); bf16 mode
1.5e-2 with both sides fed the same bf16-rounded inputs (differences = bf16 rounding of the output and
of P in attention).
"""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import pkg, rel_err

pytestmark = pytest.mark.gpu

L = pkg("_lib")
ops = pkg("ops")
weights = pkg("weights")

DT = [torch.float32, torch.bfloat16]
TOL = {torch.float32: 2e-5, torch.bfloat16: 1.5e-2}


def dev():
    return torch.device("cuda:0")


@pytest.fixture
def tuning(monkeypatch):
    """Set an SL_* tuning switch and re-read the library's table; the table is restored in the finalizer, so a failing assert
    between the two reloads cannot leave a switch latched for the rest of the process."""
    def set_(name, value):
        monkeypatch.setenv(name, value)
        L.lib().sl_tuning_reload()

    yield set_
    monkeypatch.undo()
    L.lib().sl_tuning_reload()


def rnd(*shape, seed=0, std=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * std


def q(x, dt):
    """value seen by the kernel: rounded to the storage dtype, back in fp32 for the reference."""
    return x.to(dt).float()


def test_library_loads_and_reports_gfx950():
    import ctypes
    buf = ctypes.create_string_buffer(64)
    L.check(L.lib().sl_device_arch(buf, 64))
    assert buf.value.decode().startswith("gfx950"), buf.value


@pytest.mark.parametrize("M,N,K", [(634, 3072, 3072), (634, 5120, 3072), (998, 1024, 4096), (998, 3072, 1024), (137, 5120, 3072), (130, 96, 512),
                                   (1, 3072, 1024), (255, 16384, 512), (2048, 2048, 1024)])
def test_gemm_ring_form_of_the_128_tile_kernel_gives_the_two_stage_kernels_bits(M, N, K, tuning):
    """gemm128.hip: products of at most one 128 x 128 tile per CU keep a ring of 3 or 4 K slabs in flight (counted vmcnt) instead of the
    two stages of gemm_tiled_glds_kernel.  Same tile, fragment order and epilogues: every form — plain, bias + GELU + pre-activation copy,
    bias + residual, fp32 accumulation into C, batched K runs (split-K with a workspace, where the ring changes the NUMBER of runs and so
    only the fp32 reference applies) — must give the two-stage kernel's bits, and agree with the fp32 product.  Ragged M / N edges, K from
    8 slabs (the admission bound) to 48; 2 048 x 2 048 has exactly 256 tiles (the last product the ring takes)."""
    dt = torch.bfloat16
    A, W, R, b = rnd(M, K, seed=291), rnd(N, K, seed=292, std=K ** -0.5), rnd(M, N, seed=293), rnd(N, seed=294)
    Ad, Wd, Rd, bd = A.to(dev(), dt), W.to(dev(), dt), R.to(dev(), dt), b.to(dev(), dt)
    acc0 = rnd(M, N, seed=295).to(dev())

    def run():
        kw = dict(M=M, N=N, K=K, lda=K, ldw=K)
        o1 = ops.gemm_ex(Ad, Wd, out=torch.empty((M, N), device=dev(), dtype=dt), **kw)
        aux = torch.empty((M, N), device=dev(), dtype=dt)
        o2 = ops.gemm_ex(Ad, Wd, out=torch.empty((M, N), device=dev(), dtype=dt), bias=bd, act=L.ACT_GELU, aux_out=aux, **kw)
        o3 = ops.gemm_ex(Ad, Wd, out=torch.empty((M, N), device=dev(), dtype=dt), bias=bd, residual=Rd, ldr=N, **kw)
        o4 = acc0.clone()
        ops.gemm_ex(Ad, Wd, out=o4, residual=o4, ldr=N, out_f32=True, residual_f32=True, **kw)
        return o1, o2, aux, o3, o4

    tuning("SL_GLDS_RING", "0")
    tuning("SL_GLDS_DMAB", "0")
    two_stage = run()
    ref = q(A, dt) @ q(W, dt).T
    assert rel_err(two_stage[0].float().cpu(), ref) < TOL[dt]
    tuning("SL_GLDS_DMAB", "1")       # the two-stage kernel with the next slab's DMA requests between its MFMAs (A/B form, off by default): same bits
    for a_, b_ in zip(run(), two_stage):
        assert torch.equal(a_, b_)
    for stages in ("4", "3", "104", "204"):
        tuning("SL_GLDS_RING", stages)
        for rep in range(2):
            ring = run()
            for a_, b_ in zip(ring, two_stage):
                assert torch.equal(a_, b_), (stages, rep)
    if K >= 2048 and N % 4 == 0:      # split-K admission: the ring form cuts into 256 / tiles runs instead of 512 / tiles
        ws = ops.streamk_workspace(dev())
        for stages in ("0", "4"):
            tuning("SL_GLDS_RING", stages)
            o = ops.gemm_ex(Ad, Wd, out=torch.empty((M, N), device=dev(), dtype=dt), bias=bd, residual=Rd, ldr=N, M=M, N=N, K=K, lda=K, ldw=K, sk_ws=ws)
            assert rel_err(o.float().cpu(), ref + q(b, dt) + q(R, dt)) < TOL[dt]


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,N,K", [(499, 1024, 1024), (130, 96, 192), (128, 128, 64), (257, 3072, 1024), (1000, 512, 1536)])
def test_gemm_tiled_bias_gelu_residual(dt, M, N, K):
    A, W, b, R = rnd(M, K, seed=1), rnd(N, K, seed=2, std=K ** -0.5), rnd(N, seed=3), rnd(M, N, seed=4)
    ref = F.gelu(q(A, dt) @ q(W, dt).T + q(b, dt)) + q(R, dt)
    out = ops.gemm(A.to(dev(), dt), W.to(dev(), dt), bias=b.to(dev(), dt), residual=R.to(dev(), dt), act=L.ACT_GELU)
    assert rel_err(out.float().cpu(), ref) < TOL[dt]
    ref2 = q(A, dt) @ q(W, dt).T
    out2 = ops.gemm(A.to(dev(), dt), W.to(dev(), dt), out_f32=True)
    assert out2.dtype == torch.float32
    assert rel_err(out2.cpu(), ref2) < 2e-5


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,N,K", [(40000, 1000, 2048), (16500, 2050, 2112), (131072, 256, 2048), (20000, 1000, 192)])
def test_gemm_tiled_256_tile(dt, M, N, K):
    """>= 512 tiles of 256 x 256: the 8-wave kernel (ragged M / N edges, every epilogue)."""
    A, W, b, R = rnd(M, K, seed=51), rnd(N, K, seed=52, std=K ** -0.5), rnd(N, seed=53), rnd(M, N, seed=54)
    Ad, Wd = A.to(dev(), dt), W.to(dev(), dt)
    ref = F.gelu(q(A, dt) @ q(W, dt).T + q(b, dt)) + q(R, dt)
    out = ops.gemm(Ad, Wd, bias=b.to(dev(), dt), residual=R.to(dev(), dt), act=L.ACT_GELU)
    assert rel_err(out.float().cpu(), ref) < TOL[dt]
    out2 = ops.gemm(Ad, Wd, out_f32=True)
    assert rel_err(out2.cpu(), q(A, dt) @ q(W, dt).T) < 2e-5
    if N % 32 == 0:
        g, u = W[: N // 2], W[N // 2:]
        wgu = weights.interleave_gate_up(g, u).to(dev(), dt)
        ref3 = F.silu(q(A, dt) @ q(g, dt).T) * (q(A, dt) @ q(u, dt).T)
        assert rel_err(ops.gemm(Ad, wgu, act=L.ACT_SILU_MUL).float().cpu(), ref3) < TOL[dt]


@pytest.mark.parametrize("dt", DT)
def test_gemm_tiled_256_tile_grouped_ragged(dt):
    """Ragged batch (per-group row counts and offsets) through the 256-tile kernel: one launch for all groups."""
    Ms, N, K = [30000, 70001, 257, 45000], 512, 2048
    offs = [0]
    for m in Ms:
        offs.append(offs[-1] + m)
    A, W, b = rnd(offs[-1], K, seed=55), rnd(N, K, seed=56, std=K ** -0.5), rnd(N, seed=57)
    Ad, Wd, bd = A.to(dev(), dt), W.to(dev(), dt), b.to(dev(), dt)
    out = torch.zeros(offs[-1], N, device=dev(), dtype=dt)
    grp = torch.tensor([[m, offs[i] * K, offs[i] * N, 0] for i, m in enumerate(Ms)], dtype=torch.int64, device=dev())
    ops.gemm_ex(Ad, Wd, M=max(Ms), N=N, K=K, lda=K, ldw=K, out=out, ldc=N, bias=bd, act=L.ACT_GELU, batch=len(Ms), groups=grp, w_mod=1)
    ref = F.gelu(q(A, dt) @ q(W, dt).T + q(b, dt))
    assert rel_err(out.float().cpu(), ref) < TOL[dt]


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M", [1, 3, 16, 17, 32, 50, 64])
@pytest.mark.parametrize("N,K", [(3072, 3072), (160, 8192), (1000, 256), (48, 40)])
def test_gemm_skinny(dt, M, N, K):
    if dt == torch.float32 and K % 4 or dt == torch.bfloat16 and K % 8:
        pytest.skip("K alignment")
    A, W, R = rnd(M, K, seed=5), rnd(N, K, seed=6, std=K ** -0.5), rnd(M, N, seed=7)
    ref = q(A, dt) @ q(W, dt).T + q(R, dt)
    out = ops.gemm(A.to(dev(), dt), W.to(dev(), dt), residual=R.to(dev(), dt))
    assert rel_err(out.float().cpu(), ref) < TOL[dt]
    out32 = ops.gemm(A.to(dev(), dt), W.to(dev(), dt), out_f32=True)
    assert rel_err(out32.cpu(), q(A, dt) @ q(W, dt).T) < 2e-5


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M", [5, 137, 300])
def test_gemm_silu_mul_interleaved(dt, M):
    H, Fd = 256, 512
    x, g, u = rnd(M, H, seed=8), rnd(Fd, H, seed=9, std=H ** -0.5), rnd(Fd, H, seed=10, std=H ** -0.5)
    wgu = weights.interleave_gate_up(g, u)
    ref = F.silu(q(x, dt) @ q(g, dt).T) * (q(x, dt) @ q(u, dt).T)
    out = ops.gemm(x.to(dev(), dt), wgu.to(dev(), dt), act=L.ACT_SILU_MUL)
    assert out.shape == (M, Fd)
    assert rel_err(out.float().cpu(), ref) < TOL[dt]


@pytest.mark.parametrize("dt", DT)
def test_gemm_implicit_conv_overlapping_rows(dt):
    """Conv1d(C->C2, k=3, s=2) on channel-last rows as a GEMM with lda = s*C < K = k*C."""
    Lin, Cc, C2, k, s = 401, 64, 96, 3, 2
    x, w, b = rnd(Lin, Cc, seed=11), rnd(C2, Cc, k, seed=12, std=(Cc * k) ** -0.5), rnd(C2, seed=13)
    ref = F.conv1d(q(x, dt).T[None], q(w, dt), q(b, dt), stride=s)[0].T
    Lo = (Lin - k) // s + 1
    wt = w.permute(0, 2, 1).reshape(C2, k * Cc).contiguous()
    out = ops.gemm(x.to(dev(), dt), wt.to(dev(), dt), bias=b.to(dev(), dt), M=Lo, K=k * Cc, lda=s * Cc)
    assert rel_err(out.float().cpu(), ref) < TOL[dt]


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("cols", [512, 1024, 3072, 128])
def test_layernorm_and_rmsnorm(dt, cols):
    x, g, b = rnd(77, cols, seed=14), 1 + rnd(cols, seed=15, std=0.1), rnd(cols, seed=16, std=0.1)
    xd, gd, bd = x.to(dev(), dt), g.to(dev(), dt), b.to(dev(), dt)
    ref = F.layer_norm(q(x, dt), (cols,), q(g, dt), q(b, dt), 1e-5)
    assert rel_err(ops.layernorm(xd, gd, bd, 1e-5).float().cpu(), ref) < TOL[dt]
    assert rel_err(ops.layernorm(xd, gd, bd, 1e-5, gelu=True).float().cpu(), F.gelu(ref)) < TOL[dt]
    xf = q(x, dt)
    ref = q(g, dt) * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5))
    assert rel_err(ops.rmsnorm(xd, gd, 1e-5).float().cpu(), ref) < TOL[dt]


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("Cc,n", [(512, 16000), (64, 4000), (512, 407)])
def test_hubert_conv0_fused(dt, Cc, n):
    wave = rnd(n, seed=17, std=0.1)
    w, b, g, be = rnd(Cc, 1, 10, seed=18, std=0.4), rnd(Cc, seed=19, std=0.1), 1 + rnd(Cc, seed=20, std=0.1), rnd(Cc, seed=21, std=0.1)
    y = F.conv1d(wave[None, None], w, b, stride=5)[0].T
    ref = F.gelu(F.layer_norm(y, (Cc,), g, be, 1e-5))
    out = ops.hubert_conv0(wave.to(dev()), w.reshape(Cc, 10).to(dev()), b.to(dev()), g.to(dev()), be.to(dev()), dt)
    assert out.shape == ref.shape
    assert rel_err(out.float().cpu(), ref) < (1e-5 if dt == torch.float32 else 6e-3)


@pytest.mark.parametrize("dt", DT)
def test_posconv_stage_and_grouped_conv(dt):
    import ctypes as C
    T, H, G, k = 131, 256, 4, 16
    Hg = H // G
    x, w, b = rnd(T, H, seed=22), rnd(H, Hg, k, seed=23, std=(Hg * k) ** -0.5), rnd(H, seed=24, std=0.1)
    pos = F.conv1d(q(x, dt).T[None], q(w, dt), q(b, dt), padding=k // 2, groups=G)[0][:, :-1].T
    ref = q(x, dt) + F.gelu(pos)
    xd = x.to(dev(), dt)
    xg = ops.posconv_stage(xd, G, k)
    assert xg.shape == (G, T + k, Hg)
    assert torch.equal(xg[:, k // 2:k // 2 + T].permute(1, 0, 2).reshape(T, H), xd) and float(xg[:, :k // 2].abs().max()) == 0
    wd = w.permute(0, 2, 1).reshape(G, Hg, k * Hg).contiguous().to(dev(), dt)
    bd = b.to(dev(), dt)
    out = torch.empty_like(xd)
    a = L.GemmArgs()
    a.A, a.lda, a.strideA = xg.data_ptr(), Hg, (T + k) * Hg
    a.W, a.ldw, a.strideW = wd.data_ptr(), k * Hg, Hg * k * Hg
    a.C, a.ldc, a.strideC = out.data_ptr(), H, Hg
    a.bias, a.strideBias = bd.data_ptr(), Hg
    a.residual, a.ldr, a.strideR = xd.data_ptr(), H, Hg
    a.M, a.N, a.K, a.batch, a.dtype, a.act = T, Hg, k * Hg, G, L.dtype_code(dt), L.ACT_GELU
    ops.gemm_batched(a)
    assert rel_err(out.float().cpu(), ref) < TOL[dt]


@pytest.mark.parametrize("dt", DT)
def test_avgpool_rows_and_ranges(dt):
    x = rnd(49, 128, seed=25)
    xd = x.to(dev(), dt)
    ref = F.avg_pool1d(q(x, dt).T[None], 8, 4)[0].T
    assert rel_err(ops.avgpool_rows(xd, 8, 4).float().cpu(), ref) < TOL[dt]
    ranges = [(0, 3), (3, 4), (4, 11), (11, 30), (30, 49)]
    ref = torch.stack([q(x, dt)[s:e].mean(0) for s, e in ranges])
    r = torch.tensor(ranges, dtype=torch.int32, device=dev())
    assert rel_err(ops.avgpool_rows(xd, ranges=r).float().cpu(), ref) < TOL[dt]


@pytest.mark.parametrize("dt", DT)
def test_embed_gather(dt):
    table = rnd(1000, 256, seed=26).to(dev(), dt)
    ids = torch.tensor([[0, 999, 5, 5, 123]])
    assert torch.equal(ops.embed_gather(table, ids), table[ids.view(-1).to(dev())])


def ref_attention(qh, kh, vh, causal, scale, dt):
    """qh (nh,Sq,D) kh/vh (nkv,Sk,D) fp32 -> (Sq, nh*D); softmax fp32; P rounded to dt like the kernel."""
    nh, Sq, D = qh.shape
    nkv, Sk, _ = kh.shape
    rep = nh // nkv
    kr, vr = kh.repeat_interleave(rep, 0), vh.repeat_interleave(rep, 0)
    s = qh @ kr.transpose(1, 2) * scale
    if causal:
        i = torch.arange(Sq)[:, None] + (Sk - Sq)
        s = s.masked_fill(torch.arange(Sk)[None, :] > i, float("-inf"))
    p = F.softmax(s, dim=-1)
    return (p @ vr).transpose(0, 1).reshape(Sq, nh * D)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("seqlens", [[499], [64], [1, 63, 65, 200], [130, 17], [2999], [5999, 1499]])   # 60 s / 120 s / 30 s utterances (configs[4])
def test_attention_noncausal_d64_varlen(dt, seqlens):
    nh, D = 4, 64
    ntok = sum(seqlens)
    qkv = rnd(ntok, 3 * nh * D, seed=27)
    out = ops.attn_packed_qkv(qkv.to(dev(), dt), seqlens, nh, nh, D, False, D ** -0.5).float().cpu()
    t0 = 0
    for n in seqlens:
        blk = q(qkv[t0:t0 + n], dt).view(n, 3, nh, D)
        ref = ref_attention(blk[:, 0].transpose(0, 1), blk[:, 1].transpose(0, 1), blk[:, 2].transpose(0, 1), False, D ** -0.5, dt)
        assert rel_err(out[t0:t0 + n], ref) < TOL[dt], n
        t0 += n


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("seqlens", [[137], [5, 64, 129], [300], [1800, 777]])   # 1800 ~ the prompt of a 120 s utterance + text prompt
@pytest.mark.parametrize("nh,nkv", [(6, 2), (2, 2)])
def test_attention_causal_gqa_d128(dt, seqlens, nh, nkv):
    D = 128
    ntok = sum(seqlens)
    qkv = rnd(ntok, (nh + 2 * nkv) * D, seed=28)
    out = ops.attn_packed_qkv(qkv.to(dev(), dt), seqlens, nh, nkv, D, True, D ** -0.5).float().cpu()
    t0 = 0
    for n in seqlens:
        blk = q(qkv[t0:t0 + n], dt)
        qh = blk[:, :nh * D].view(n, nh, D).transpose(0, 1)
        kh = blk[:, nh * D:(nh + nkv) * D].view(n, nkv, D).transpose(0, 1)
        vh = blk[:, (nh + nkv) * D:].view(n, nkv, D).transpose(0, 1)
        assert rel_err(out[t0:t0 + n], ref_attention(qh, kh, vh, True, D ** -0.5, dt)) < TOL[dt], n
        t0 += n


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("D,nh,nkv", [(64, 4, 4), (128, 6, 2)])
@pytest.mark.parametrize("causal", [False, True])
def test_attention_keys_longer_than_queries(dt, D, nh, nkv, causal):
    """Separate q / k / v buffers with more keys than queries per sequence (a prompt continued against cached keys): causal
    visibility is j <= i + (klen - qlen); covers both query-tile widths of the bf16 kernel and ragged tails."""
    qlens, klens = [70, 1, 260, 33], [200, 64, 300, 33]
    gq, gk = torch.Generator().manual_seed(41), torch.Generator().manual_seed(42)
    qt = torch.randn(sum(qlens), nh * D, generator=gq)
    kt = torch.randn(sum(klens), nkv * D, generator=gk)
    vt = torch.randn(sum(klens), nkv * D, generator=gk)
    cu_q = torch.tensor([0] + list(torch.tensor(qlens).cumsum(0)), dtype=torch.int32, device=dev())
    cu_k = torch.tensor([0] + list(torch.tensor(klens).cumsum(0)), dtype=torch.int32, device=dev())
    kl = torch.tensor(klens, dtype=torch.int32, device=dev())
    out = torch.empty(sum(qlens), nh * D, device=dev(), dtype=dt)
    ops.attn_fwd(qt.to(dev(), dt), kt.to(dev(), dt), vt.to(dev(), dt), out, cu_q, cu_k, kl, q_strides=(nh * D, D), k_strides=(nkv * D, D),
                 v_strides=(nkv * D, D), o_strides=(nh * D, D), nseq=len(qlens), max_qlen=max(qlens), n_heads=nh, n_kv_heads=nkv, head_dim=D,
                 causal=causal, scale=D ** -0.5)
    out = out.float().cpu()
    q0 = k0 = 0
    for nq, nk in zip(qlens, klens):
        qh = q(qt[q0:q0 + nq], dt).view(nq, nh, D).transpose(0, 1)
        kh = q(kt[k0:k0 + nk], dt).view(nk, nkv, D).transpose(0, 1)
        vh = q(vt[k0:k0 + nk], dt).view(nk, nkv, D).transpose(0, 1)
        assert rel_err(out[q0:q0 + nq], ref_attention(qh, kh, vh, causal, D ** -0.5, dt)) < TOL[dt], (nq, nk)
        q0 += nq
        k0 += nk


@pytest.mark.parametrize("dt", DT)
def test_attention_probability_dropout_matches_host_mask(dt):
    """Training-mode attention (sl_attn_args.dropout_p): probabilities are dropped after the softmax (undropped normaliser)
    with the counter-based mask at index ((token * heads + head) << 16) | key; both forward kernels against a torch
    restatement that rebuilds the mask on the host."""
    nh, D, p_drop, seed = 4, 64, 0.25, 0x1234_5678_9ABC_DEF1
    seqlens = [70, 133]
    ntok = sum(seqlens)
    qkv = rnd(ntok, 3 * nh * D, seed=31)
    out = ops.attn_packed_qkv(qkv.to(dev(), dt), seqlens, nh, nh, D, False, D ** -0.5, dropout_p=p_drop, dropout_seed=seed).float().cpu()
    keep_all = ops.dropout_keep_mask((ntok * nh) << 16, p_drop, seed).view(ntok, nh, 1 << 16)
    t0 = 0
    for n in seqlens:
        blk = q(qkv[t0:t0 + n], dt).view(n, 3, nh, D)
        qh, kh, vh = blk[:, 0].transpose(0, 1), blk[:, 1].transpose(0, 1), blk[:, 2].transpose(0, 1)
        pr = F.softmax(qh @ kh.transpose(1, 2) * D ** -0.5, dim=-1)                      # (nh, n, n)
        keep = keep_all[t0:t0 + n, :, :n].permute(1, 0, 2)
        pr = torch.where(keep, pr / (1.0 - p_drop), torch.zeros_like(pr))
        ref = (pr @ vh).transpose(0, 1).reshape(n, nh * D)
        assert rel_err(out[t0:t0 + n], ref) < TOL[dt], n
        t0 += n
    plain = ops.attn_packed_qkv(qkv.to(dev(), dt), seqlens, nh, nh, D, False, D ** -0.5).float().cpu()
    assert rel_err(out, plain) > 0.05          # the mask did something


@pytest.mark.parametrize("dt", DT)
def test_rope_kv_append_and_decode_attention(dt):
    arch = weights.LlamaArch(hidden_size=256, num_attention_heads=6, num_key_value_heads=2, head_dim=128,
                             rope_scaling=dict(factor=32.0, low_freq_factor=1.0, high_freq_factor=4.0,
                                               original_max_position_embeddings=8192))
    nh, nkv, D, max_ctx, B = 6, 2, 128, 96, 3
    cos, sin = weights.rope_tables(arch, max_ctx)
    lens = [70, 1, 33]  # tokens already in the cache per sequence (prefill), then one decode token each
    ntok = sum(lens)
    qkv = rnd(ntok, (nh + 2 * nkv) * D, seed=29)
    tok_seq = torch.tensor(sum([[s] * n for s, n in enumerate(lens)], []), dtype=torch.int32)
    tok_pos = torch.tensor(sum([list(range(n)) for n in lens], []), dtype=torch.int32)
    kc = torch.zeros(B, nkv, max_ctx, D, device=dev(), dtype=dt)
    vc = torch.zeros_like(kc)
    qd = qkv.to(dev(), dt)
    ops.rope_kv_append(qd, kc, vc, tok_seq.to(dev()), tok_pos.to(dev()), cos.to(dev()), sin.to(dev()), nh, nkv, D, max_ctx)

    def rope(x, pos):  # x (n, heads, D) fp32
        c = torch.cat([cos[pos], cos[pos]], -1)[:, None]
        s = torch.cat([sin[pos], sin[pos]], -1)[:, None]
        x1, x2 = x[..., :D // 2], x[..., D // 2:]
        return x * c + torch.cat([-x2, x1], -1) * s

    qf = q(qkv, dt)
    q_ref = rope(qf[:, :nh * D].view(ntok, nh, D), tok_pos.long())
    k_ref = rope(qf[:, nh * D:(nh + nkv) * D].view(ntok, nkv, D), tok_pos.long())
    v_ref = qf[:, (nh + nkv) * D:].view(ntok, nkv, D)
    tol = 1e-6 if dt == torch.float32 else 6e-3
    assert rel_err(qd[:, :nh * D].float().cpu().view(ntok, nh, D), q_ref) < tol
    t0 = 0
    for s, n in enumerate(lens):
        assert rel_err(kc[s, :, :n].float().cpu(), k_ref[t0:t0 + n].transpose(0, 1)) < tol
        assert torch.equal(vc[s, :, :n].float().cpu(), v_ref[t0:t0 + n].transpose(0, 1))
        t0 += n
    # decode attention: query = last prefill token of each sequence, attends its whole context
    last = torch.tensor([sum(lens[:i + 1]) - 1 for i in range(B)])
    qlast = qd[last.to(dev())].contiguous()
    ctx = torch.tensor(lens, dtype=torch.int32, device=dev())
    out = ops.attn_decode(qlast, qlast.stride(0), kc, vc, ctx, nh, nkv, D, max_ctx, D ** -0.5).float().cpu()
    for s, n in enumerate(lens):
        qh = qlast[s, :nh * D].float().cpu().view(nh, 1, D)
        ref = ref_attention(qh, kc[s, :, :n].float().cpu(), vc[s, :, :n].float().cpu(), False, D ** -0.5, dt)
        assert rel_err(out[s], ref[0]) < TOL[dt]


def test_greedy_select_semantics():
    B, V, max_new = 4, 5000, 6
    logits = rnd(B, V, seed=30)
    logits[0, 777] = 50.0
    logits[1, 10] = 60.0; logits[1, 4000] = 60.0   # tie -> lowest index
    logits[2, 7] = 70.0                             # eos id
    logits[3, 1234] = 80.0                          # row already finished -> pad
    ld = logits.to(dev())
    unfinished = torch.tensor([1, 1, 1, 0], dtype=torch.int32, device=dev())
    ctx = torch.tensor([5, 6, 7, 8], dtype=torch.int32, device=dev())
    gen = torch.tensor([0, 1, 2, 3], dtype=torch.int32, device=dev())
    fin = torch.zeros(B, dtype=torch.int32, device=dev())
    nxt = torch.zeros(B, dtype=torch.int32, device=dev())
    out = torch.full((B, max_new), -1, dtype=torch.int32, device=dev())
    ops.greedy_select(ld, [7, 9], 99, True, unfinished, ctx, gen, fin, nxt, out)
    assert nxt.tolist() == [777, 10, 7, 99]
    assert unfinished.tolist() == [1, 1, 0, 0]
    assert ctx.tolist() == [6, 7, 8, 9] and gen.tolist() == [1, 2, 3, 4] and fin.tolist() == [0, 0, 3, 0]
    assert out[0, 0] == 777 and out[1, 1] == 10 and out[2, 2] == 7 and out[3, 3] == 99


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,N,K", [(260, 1000, 256), (300, 5000, 512), (1024, 20000, 3072), (513, 4097 * 4, 1024), (70, 130, 64)])
def test_fused_lm_head_top1_equals_logits_then_argmax(dt, M, N, K):
    """lm_head + argmax without the logits round trip: the tiled kernels' fused top-1 leaves one (value, column)
    per row and 64-column group; sl_greedy_select_partial finishes.  Against the SAME kernel's fp32 logits + sl_greedy_select the
    tokens are identical bit for bit (same accumulators), ties and planted duplicates included, on both tile sizes (128^2: the
    small products; 256^2: K >= 1024 with >= 512 tiles), ragged last row tile and column group."""
    A, W = rnd(M, K, seed=91), rnd(N, K, seed=92, std=K ** -0.5)
    W[N - 1] = W[3]                                    # duplicate weight rows: exact ties between columns 3 and N - 1 in every row
    if N > 200:
        W[130] = W[3]
    Ad, Wd = A.to(dev(), dt), W.to(dev(), dt)
    bias = rnd(N, seed=93).to(dev(), dt)
    bias[N - 1] = bias[3]
    if N > 200:
        bias[130] = bias[3]
    for b in (None, bias):
        logits = ops.gemm(Ad, Wd, bias=b, out_f32=True)
        val, idx = ops.gemm_top1(Ad, Wd, bias=b)
        ng = (N + 63) // 64
        assert val.shape == (ng, M)
        # every partial is the group's maximum at its lowest column
        pad = torch.full((M, ng * 64 - N), float("-inf"), device=dev())
        grp = torch.cat([logits, pad], 1).view(M, ng, 64)
        gmax, garg = grp.max(dim=2)
        assert torch.equal(val.T.contiguous(), gmax)
        first = (grp == gmax[:, :, None]).float().argmax(dim=2) + torch.arange(ng, device=dev())[None, :] * 64
        assert torch.equal(idx.T.long().contiguous(), first)
        # ... and the two select passes agree on token and bookkeeping
        state = []
        for fused in (False, True):
            unfinished = torch.ones(M, dtype=torch.int32, device=dev()); unfinished[M // 2] = 0
            ctx = torch.arange(M, dtype=torch.int32, device=dev())
            gen = torch.zeros(M, dtype=torch.int32, device=dev())
            fin = torch.zeros(M, dtype=torch.int32, device=dev())
            nxt = torch.zeros(M, dtype=torch.int32, device=dev())
            out = torch.full((M, 3), -1, dtype=torch.int32, device=dev())
            eos = [int(logits[0].argmax()), 3]
            if fused:
                ops.greedy_select_partial(val, idx, eos, 7, True, unfinished, ctx, gen, fin, nxt, out)
            else:
                ops.greedy_select(logits, eos, 7, True, unfinished, ctx, gen, fin, nxt, out)
            state.append([x.cpu() for x in (unfinished, ctx, gen, fin, nxt, out)])
        for x, y in zip(*state):
            assert torch.equal(x, y)
        ref = logits.argmax(dim=1).cpu()          # torch returns the first maximum too
        ref[M // 2] = 7
        assert torch.equal(state[1][4].long(), ref)
        assert int(state[1][0][0]) == 0            # row 0 emitted an EOS id


# ---- decode-path kernels: packed weights, fused RMSNorm, fused RoPE + KV append, flash-decoding ----------
@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M", [1, 8, 16, 17, 26, 27, 32, 40, 64, 100, 128, 200, 256])   # <= 26 rows: skinny kernel, above: gemm_stream.hip
@pytest.mark.parametrize("N,K", [(3072, 3072), (5120, 1024), (1000, 256), (33000, 512), (3072, 8192)])   # the last: down (M <= 8: four steps in flight)
def test_gemm_packed_matches_rowmajor_reference(dt, M, N, K):
    if K == 8192 and M not in (1, 8, 16, 27, 100):
        pytest.skip("down-shaped product: one row count per kernel structure")
    A, W, R = rnd(M, K, seed=31), rnd(N, K, seed=32, std=K ** -0.5), rnd(M, N, seed=33)
    Wd = W.to(dev(), dt)
    Wp = ops.pack_weight(Wd)
    assert Wp.shape[0] == (N + 15) // 16 * 16
    ref = q(A, dt) @ q(W, dt).T + q(R, dt)
    out = ops.gemm_decode(A.to(dev(), dt), Wp, N, residual=R.to(dev(), dt))
    assert rel_err(out.float().cpu(), ref) < TOL[dt]
    assert rel_err(ops.gemm_decode(A.to(dev(), dt), Wp, N, out_f32=True).cpu(), q(A, dt) @ q(W, dt).T) < 2e-5
    # without the K-split scratch buffer the streaming kernel applies the epilogue itself
    out = ops.gemm_decode(A.to(dev(), dt), Wp, N, residual=R.to(dev(), dt), split_k=False)
    assert rel_err(out.float().cpu(), ref) < TOL[dt]


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M", [1, 16, 33, 64, 130])
def test_gemm_packed_fused_rmsnorm_and_silu(dt, M):
    H, Fd = 512, 1024
    x, gain = rnd(M, H, seed=34), 1 + rnd(H, seed=35, std=0.1)
    g, u = rnd(Fd, H, seed=36, std=H ** -0.5), rnd(Fd, H, seed=37, std=H ** -0.5)
    xf = q(x, dt)
    normed = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5)
    wg, wu = q(g * gain[None], dt), q(u * gain[None], dt)      # gain folded into the weight, then rounded
    ref = F.silu(normed @ wg.T) * (normed @ wu.T)
    wgu = weights.interleave_gate_up(g * gain[None], u * gain[None]).to(dev(), dt)
    out = ops.gemm_decode(x.to(dev(), dt), ops.pack_weight(wgu), 2 * Fd, act=L.ACT_SILU_MUL, fuse_rms=True, eps=1e-5)
    assert out.shape == (M, Fd)
    assert rel_err(out.float().cpu(), ref) < TOL[dt]
    out = ops.gemm_decode(x.to(dev(), dt), ops.pack_weight(wgu), 2 * Fd, act=L.ACT_SILU_MUL, fuse_rms=True, eps=1e-5, split_k=False)
    assert rel_err(out.float().cpu(), ref) < TOL[dt]


@pytest.mark.parametrize("M", [384, 500, 513])
def test_gemm_packed_many_rows_wide_matrix(M):
    """384+ rows (three to five 128-row blocks per n-block, placed on one XCD) against 8192+ weight rows, bf16."""
    dt = torch.bfloat16
    H, Fd = 512, 4608
    x = rnd(M, H, seed=61)
    g, u = rnd(Fd, H, seed=62, std=H ** -0.5), rnd(Fd, H, seed=63, std=H ** -0.5)
    xd = x.to(dev(), dt)
    xf = q(x, dt)
    rstd = torch.rsqrt(xf.pow(2).mean(-1) + 1e-5)
    wgu = weights.interleave_gate_up(g, u).to(dev(), dt)
    ref = F.silu((xf * rstd[:, None]) @ q(g, dt).T) * ((xf * rstd[:, None]) @ q(u, dt).T)
    out = ops.gemm_decode(xd, ops.pack_weight(wgu), 2 * Fd, act=L.ACT_SILU_MUL, fuse_rms=True, eps=1e-5, rstd_in=rstd.to(dev()))
    assert rel_err(out.float().cpu(), ref) < TOL[dt]
    W = rnd(8200, H, seed=64, std=H ** -0.5)
    R = rnd(M, 8200, seed=65)
    out2 = ops.gemm_decode(xd, ops.pack_weight(W.to(dev(), dt)), 8200, residual=R.to(dev(), dt))
    assert rel_err(out2.float().cpu(), xf @ q(W, dt).T + q(R, dt)) < TOL[dt]


@pytest.mark.parametrize("M", [385, 512, 700])
@pytest.mark.parametrize("H", [512, 448])
def test_gemm_packed_wide_block_form(M, H, tuning):
    """The 256 x 128 streaming block (gemm_stream_wide_kernel: x and packed weight fragments through LDS-DMA, 8 compute + 4
    loader waves) on shapes the default rule would leave to the 128 x 128 form: ragged last row block, a fragment count that is
    not a multiple of 8, an odd number of 128-byte K stages (H = 448), SILU pairs with handed-over row scales, bias-free
    residual epilogue, fp32 output.  Same tolerance as the other bf16 GEMM tests; equal to the 128 x 128 form within rounding."""
    dt = torch.bfloat16
    Fd = 4608
    x = rnd(M, H, seed=61)
    g, u = rnd(Fd, H, seed=62, std=H ** -0.5), rnd(Fd, H, seed=63, std=H ** -0.5)
    xd, xf = x.to(dev(), dt), q(x, dt)
    rstd = torch.rsqrt(xf.pow(2).mean(-1) + 1e-5)
    wgu = ops.pack_weight(weights.interleave_gate_up(g, u).to(dev(), dt))
    W = rnd(8200, H, seed=64, std=H ** -0.5)
    Wp = ops.pack_weight(W.to(dev(), dt))
    R = rnd(M, 8200, seed=65)
    outs = {}
    for mode in ("0", "2"):
        tuning("SL_STREAM_WIDE", mode)
        a = ops.gemm_decode(xd, wgu, 2 * Fd, act=L.ACT_SILU_MUL, fuse_rms=True, eps=1e-5, rstd_in=rstd.to(dev()), split_k=False)
        b = ops.gemm_decode(xd, Wp, 8200, residual=R.to(dev(), dt), split_k=False)
        c = ops.gemm_decode(xd, Wp, 8200, out_f32=True, split_k=False)
        outs[mode] = (a.float().cpu(), b.float().cpu(), c.cpu())
    ref_a = F.silu((xf * rstd[:, None]) @ q(g, dt).T) * ((xf * rstd[:, None]) @ q(u, dt).T)
    ref_b = xf @ q(W, dt).T + q(R, dt)
    for mode in ("0", "2"):
        assert rel_err(outs[mode][0], ref_a) < TOL[dt] and rel_err(outs[mode][1], ref_b) < TOL[dt]
        assert rel_err(outs[mode][2], xf @ q(W, dt).T) < 2e-5
    assert rel_err(outs["2"][2], outs["0"][2]) < 1e-6      # same k order per output element in both forms


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M", [40, 128, 512, 600])     # 512 / 600 rows in bf16: the 256 x 128 form with its K split
def test_gemm_packed_rstd_handoff(dt, M):
    """K-split reduce pass emits the RMSNorm scale of the rows it stores; the next fused GEMM takes it (rstd_in)
    instead of recomputing it: same result as the in-kernel statistics."""
    H, Fd, K1 = 512, 1024, 2048
    a, wo, res = rnd(M, K1, seed=40), rnd(H, K1, seed=41, std=K1 ** -0.5), rnd(M, H, seed=42)
    assert L.lib().sl_gemm_split_count(M, H, K1, L.dtype_code(dt)) > 1
    rstd = torch.empty(M, device=dev(), dtype=torch.float32)
    x = ops.gemm_decode(a.to(dev(), dt), ops.pack_weight(wo.to(dev(), dt)), H, residual=res.to(dev(), dt), rstd_out=rstd, eps=1e-5)
    xr = q(a, dt) @ q(wo, dt).T + q(res, dt)
    assert rel_err(x.float().cpu(), xr) < TOL[dt]
    xf = x.float().cpu()
    assert rel_err(rstd.cpu(), torch.rsqrt(xf.pow(2).mean(-1) + 1e-5)) < 1e-5
    # norm_out: the same pass also leaves the RMS-normalised rows (what sl_rmsnorm over its output would write), bit for bit
    gain = (1.0 + rnd(H, seed=45, std=0.1)).to(dev(), dt)
    rstd2, h = torch.empty(M, device=dev(), dtype=torch.float32), torch.empty(M, H, device=dev(), dtype=dt)
    x2 = ops.gemm_decode(a.to(dev(), dt), ops.pack_weight(wo.to(dev(), dt)), H, residual=res.to(dev(), dt), rstd_out=rstd2, eps=1e-5, norm_out=h, norm_gain=gain)
    assert torch.equal(x2, x) and torch.equal(rstd2, rstd)
    assert rel_err(h.float().cpu(), ops.rmsnorm(x, gain, 1e-5).float().cpu()) < (1e-6 if dt == torch.float32 else 2e-3)
    assert rel_err(h.float().cpu(), (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5)) * gain.float().cpu()) < TOL[dt]
    g, u = rnd(Fd, H, seed=43, std=H ** -0.5), rnd(Fd, H, seed=44, std=H ** -0.5)
    wp = ops.pack_weight(weights.interleave_gate_up(g, u).to(dev(), dt))
    y0 = ops.gemm_decode(x, wp, 2 * Fd, act=L.ACT_SILU_MUL, fuse_rms=True, eps=1e-5)
    y1 = ops.gemm_decode(x, wp, 2 * Fd, act=L.ACT_SILU_MUL, fuse_rms=True, eps=1e-5, rstd_in=rstd)
    y2 = ops.gemm_decode(x, wp, 2 * Fd, act=L.ACT_SILU_MUL, fuse_rms=True, eps=1e-5, rstd_in=rstd, split_k=False)
    assert rel_err(y1.float().cpu(), y0.float().cpu()) < (1e-6 if dt == torch.float32 else 4e-3)
    assert rel_err(y2.float().cpu(), y0.float().cpu()) < (1e-6 if dt == torch.float32 else 4e-3)


@pytest.mark.parametrize("M", [385, 512, 700])
def test_gemm_packed_wide_split_fixup_equals_reduce_pass(M, tuning):
    """K split of the 256 x 128 form: the in-kernel fix-up (last-arriving block sums the partial records and applies the
    epilogue; a second counter per row block turns the per-tile sums of squares into rstd_out) must store exactly what the
    separate reduce launch stores — same partial records, same summation order — and leave its counters at zero (second call
    on the same workspace gives the same result).  Plain +residual epilogue with the row statistics, and the RoPE / KV epilogue."""
    dt = torch.bfloat16
    H, K1 = 512, 2048
    a, wo, res = rnd(M, K1, seed=40), rnd(H, K1, seed=41, std=K1 ** -0.5), rnd(M, H, seed=42)
    assert L.lib().sl_gemm_split_count(M, H, K1, L.dtype_code(dt)) > 1
    wp = ops.pack_weight(wo.to(dev(), dt))
    arch = weights.LlamaArch(hidden_size=1536, num_attention_heads=6, num_key_value_heads=2, head_dim=128,
                             rope_scaling=dict(factor=32.0, low_freq_factor=1.0, high_freq_factor=4.0, original_max_position_embeddings=8192))
    nh, nkv, D, Hq, max_ctx = 6, 2, 128, 1536, 64
    cos, sin = [t_.to(dev()) for t_ in weights.rope_tables(arch, max_ctx)]
    xq = rnd(M, Hq, seed=38).to(dev(), dt)
    Wq = ops.pack_weight(rnd((nh + 2 * nkv) * D, Hq, seed=39, std=Hq ** -0.5).to(dev(), dt))
    assert L.lib().sl_gemm_split_count(M, (nh + 2 * nkv) * D, Hq, L.dtype_code(dt)) > 1
    pos = torch.tensor([(7 * i + 3) % max_ctx for i in range(M)], dtype=torch.int32, device=dev())
    seq = torch.arange(M, dtype=torch.int32, device=dev())
    got = {}
    for mode in ("0", "1"):
        tuning("SL_STREAM_FIXUP", mode)
        runs = []
        for _ in range(2):
            rstd = torch.zeros(M, device=dev(), dtype=torch.float32)
            x = ops.gemm_decode(a.to(dev(), dt), wp, H, residual=res.to(dev(), dt), rstd_out=rstd, eps=1e-5)
            kc = torch.zeros(M, nkv, max_ctx, D, device=dev(), dtype=dt); vc = torch.zeros_like(kc)
            qo = ops.gemm_decode(xq, Wq, (nh + 2 * nkv) * D, act=L.ACT_ROPE_KV,
                                 rope=dict(cos=cos, sin=sin, pos=pos, seq=seq, k_cache=kc, v_cache=vc, n_heads=nh, n_kv=nkv, max_ctx=max_ctx))
            runs.append((x.cpu(), rstd.cpu(), qo.cpu(), kc.cpu(), vc.cpu()))
        for u, v in zip(runs[0], runs[1]):
            assert torch.equal(u, v)
        got[mode] = runs[0]
    for i in (0, 2, 3, 4):
        assert torch.equal(got["0"][i], got["1"][i]), i
    assert rel_err(got["1"][1], got["0"][1]) < 1e-6
    xr = q(a, dt) @ q(wo, dt).T + q(res, dt)
    assert rel_err(got["1"][0].float(), xr) < TOL[dt]
    assert rel_err(got["1"][1], torch.rsqrt(got["1"][0].float().pow(2).mean(-1) + 1e-5)) < 1e-5


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M", [1, 5, 16, 40, 130])
def test_gemm_packed_rope_kv_epilogue_equals_separate_kernels(dt, M):
    arch = weights.LlamaArch(hidden_size=256, num_attention_heads=6, num_key_value_heads=2, head_dim=128,
                             rope_scaling=dict(factor=32.0, low_freq_factor=1.0, high_freq_factor=4.0,
                                               original_max_position_embeddings=8192))
    nh, nkv, D, H, max_ctx = 6, 2, 128, 256, 64
    cos, sin = [t_.to(dev()) for t_ in weights.rope_tables(arch, max_ctx)]
    x = rnd(M, H, seed=38).to(dev(), dt)
    W = rnd((nh + 2 * nkv) * D, H, seed=39, std=H ** -0.5).to(dev(), dt)
    pos = torch.tensor([(7 * i + 3) % max_ctx for i in range(M)], dtype=torch.int32, device=dev())
    seq = torch.arange(M, dtype=torch.int32, device=dev())
    # reference path: plain GEMM -> sl_rope_kv_append
    qkv = ops.gemm(x, W, out_f32=False)
    kc1 = torch.zeros(M, nkv, max_ctx, D, device=dev(), dtype=dt); vc1 = torch.zeros_like(kc1)
    ops.rope_kv_append(qkv, kc1, vc1, seq, pos, cos, sin, nh, nkv, D, max_ctx)
    # fused path: rows of q/k heads in rotate_half pair order
    blk = torch.arange(16)
    hp = torch.cat([torch.cat([blk + 16 * j, blk + 64 + 16 * j]) for j in range(4)])
    perm = torch.cat([hp + 128 * h for h in range(nh + nkv)] + [torch.arange((nh + nkv) * 128, (nh + 2 * nkv) * 128)])
    Wp = ops.pack_weight(W[perm.to(dev())].contiguous())
    kc2 = torch.zeros_like(kc1); vc2 = torch.zeros_like(kc1)
    qout = ops.gemm_decode(x, Wp, (nh + 2 * nkv) * D, act=L.ACT_ROPE_KV,
                           rope=dict(cos=cos, sin=sin, pos=pos, seq=seq, k_cache=kc2, v_cache=vc2, n_heads=nh, n_kv=nkv, max_ctx=max_ctx))
    tol = 1e-6 if dt == torch.float32 else 8e-3   # bf16: the fused path rounds once instead of twice
    assert rel_err(qout.float().cpu(), qkv[:, :nh * D].float().cpu()) < tol
    assert rel_err(kc2.float().cpu(), kc1.float().cpu()) < tol
    assert rel_err(vc2.float().cpu(), vc1.float().cpu()) < tol


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("nh,nkv", [(6, 2), (2, 2)])
def test_attn_decode_split_matches_reference(dt, nh, nkv):
    D, max_ctx, B = 128, 448, 4
    lens = [1, 64, 65, 393]
    kc = rnd(B, nkv, max_ctx, D, seed=40).to(dev(), dt)
    vc = rnd(B, nkv, max_ctx, D, seed=41).to(dev(), dt)
    qd = rnd(B, nh * D, seed=42).to(dev(), dt)
    ctx = torch.tensor(lens, dtype=torch.int32, device=dev())
    out = ops.attn_decode_split(qd, qd.stride(0), kc, vc, ctx, nh, nkv, D, max_ctx, D ** -0.5).float().cpu()
    out1 = ops.attn_decode(qd, qd.stride(0), kc, vc, ctx, nh, nkv, D, max_ctx, D ** -0.5).float().cpu()
    for s, n in enumerate(lens):
        ref = ref_attention(qd[s].float().cpu().view(nh, 1, D), kc[s, :, :n].float().cpu(), vc[s, :, :n].float().cpu(), False, D ** -0.5, dt)
        assert rel_err(out[s], ref[0]) < TOL[dt]
        assert rel_err(out1[s], ref[0]) < TOL[dt]


@pytest.mark.parametrize("dt", DT)
def test_attn_decode_split_in_launch_merge_equals_combine_launch(dt, tuning):
    """The last block of a (sequence, kv head) to arrive merges the partial records inside the split launch (default for small batches;
    SL_ATTN_SPLIT_MERGE=0 / 1 forces the combine launch / the merge) — sc1 hand-off, arrival counter left at zero — bit for bit what the
    combine launch writes, call after call on the same workspace."""
    D, max_ctx, B, nh, nkv = 128, 448, 3, 6, 2
    lens = [1, 65, 393]
    kc = rnd(B, nkv, max_ctx, D, seed=40).to(dev(), dt)
    vc = rnd(B, nkv, max_ctx, D, seed=41).to(dev(), dt)
    qd = rnd(B, nh * D, seed=42).to(dev(), dt)
    ctx = torch.tensor(lens, dtype=torch.int32, device=dev())
    if dt == torch.bfloat16:
        tuning("SL_ATTN_FORCE_SPLIT", "1")
    ws = torch.empty(int(L.lib().sl_attn_decode_workspace_bytes(B, nh, nkv, max_ctx)), dtype=torch.uint8, device=dev())
    tuning("SL_ATTN_SPLIT_MERGE", "0")
    ref = ops.attn_decode_split(qd, qd.stride(0), kc, vc, ctx, nh, nkv, D, max_ctx, D ** -0.5, ws=ws).cpu()
    tuning("SL_ATTN_SPLIT_MERGE", "1")
    for _ in range(3):
        out = ops.attn_decode_split(qd, qd.stride(0), kc, vc, ctx, nh, nkv, D, max_ctx, D ** -0.5, ws=ws).cpu()
        assert torch.equal(out, ref)


@pytest.mark.parametrize("nh,nkv", [(24, 8), (6, 2), (2, 2), (4, 1)])
def test_attn_decode_single_pass_64_key_chunks_equal_128_key_chunks(nh, nkv, tuning):
    """SL_ATTN_DECODE_KS=64: the single-pass bf16 decode attention with 64-key chunks (a block of 19 KiB of LDS / ~116 VGPRs that can share
    a CU with a 256-tile GEMM block of another stream) against the default 128-key form and the reference, contexts from 1 key to the
    whole cache, with and without a shared prompt prefix."""
    dt = torch.bfloat16
    D, max_ctx, B = 128, 448, 64
    lens = [1 + (53 * i) % max_ctx for i in range(B)]
    lens[:8] = [1, 63, 64, 65, 127, 128, 129, 448]
    kc = rnd(B, nkv, max_ctx, D, seed=60).to(dev(), dt)
    vc = rnd(B, nkv, max_ctx, D, seed=61).to(dev(), dt)
    qd = rnd(B, nh * D, seed=62).to(dev(), dt)
    ctx = torch.tensor(lens, dtype=torch.int32, device=dev())
    ref128 = ops.attn_decode_split(qd, qd.stride(0), kc, vc, ctx, nh, nkv, D, max_ctx, D ** -0.5).float().cpu()
    outs = {}
    for form in ("64", "65"):                     # 65 = 64-key chunks with the next chunk's rows prefetched into a second register set
        tuning("SL_ATTN_DECODE_KS", form)
        out64 = ops.attn_decode_split(qd, qd.stride(0), kc, vc, ctx, nh, nkv, D, max_ctx, D ** -0.5).float().cpu()
        outs[form] = out64
        assert rel_err(out64, ref128) < 4e-3          # same products, online-softmax rescale points differ
        for s in list(range(8)) + [31, 63]:
            n = lens[s]
            ref = ref_attention(qd[s].float().cpu().view(nh, 1, D), kc[s, :, :n].float().cpu(), vc[s, :, :n].float().cpu(), False, D ** -0.5, dt)
            assert rel_err(out64[s], ref[0]) < TOL[dt], s
    assert torch.equal(outs["64"], outs["65"])     # the prefetch changes when rows are requested, not what is computed


@pytest.mark.parametrize("dt", DT)
def test_attn_decode_long_cache_mid_batch(dt):
    """16 sequences against a 2 048-position cache (the long-form leg of bench.py): the dispatcher leaves the single-pass kernel for the
    split form (B x kv heads < 768, cache >= 1 024 positions); contexts from 1 to 1 900 keys against the reference and the per-sequence kernel."""
    D, max_ctx, B, nh, nkv = 128, 2048, 16, 24, 8
    lens = [1, 63, 64, 65, 500, 1023, 1024, 1025, 1500, 1900, 137, 393, 777, 1234, 1782, 2000]
    kc = rnd(B, nkv, max_ctx, D, seed=46).to(dev(), dt)
    vc = rnd(B, nkv, max_ctx, D, seed=47).to(dev(), dt)
    qd = rnd(B, nh * D, seed=48).to(dev(), dt)
    ctx = torch.tensor(lens, dtype=torch.int32, device=dev())
    out = ops.attn_decode_split(qd, qd.stride(0), kc, vc, ctx, nh, nkv, D, max_ctx, D ** -0.5).float().cpu()
    out1 = ops.attn_decode(qd, qd.stride(0), kc, vc, ctx, nh, nkv, D, max_ctx, D ** -0.5).float().cpu()
    assert rel_err(out, out1) < TOL[dt]
    for s in (0, 3, 7, 9, 14, 15):
        n = lens[s]
        ref = ref_attention(qd[s].float().cpu().view(nh, 1, D), kc[s, :, :n].float().cpu(), vc[s, :, :n].float().cpu(), False, D ** -0.5, dt)
        assert rel_err(out[s], ref[0]) < TOL[dt]


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("nh,nkv", [(6, 2), (2, 2)])
def test_attn_decode_split_large_batch(dt, nh, nkv):
    """Decode batches in the hundreds: split + merge against the per-sequence single-pass kernel and the reference."""
    D, max_ctx, B = 128, 200, 512
    lens = [1 + (37 * i) % max_ctx for i in range(B)]
    lens[:6] = [1, 63, 64, 65, 128, 200]
    kc = rnd(B, nkv, max_ctx, D, seed=43).to(dev(), dt)
    vc = rnd(B, nkv, max_ctx, D, seed=44).to(dev(), dt)
    qd = rnd(B, nh * D, seed=45).to(dev(), dt)
    ctx = torch.tensor(lens, dtype=torch.int32, device=dev())
    out = ops.attn_decode_split(qd, qd.stride(0), kc, vc, ctx, nh, nkv, D, max_ctx, D ** -0.5).float().cpu()
    out1 = ops.attn_decode(qd, qd.stride(0), kc, vc, ctx, nh, nkv, D, max_ctx, D ** -0.5).float().cpu()   # per-sequence single-pass kernel
    assert rel_err(out, out1) < TOL[dt]
    for s in list(range(8)) + [100, 311, 511]:
        n = lens[s]
        ref = ref_attention(qd[s].float().cpu().view(nh, 1, D), kc[s, :, :n].float().cpu(), vc[s, :, :n].float().cpu(), False, D ** -0.5, dt)
        assert rel_err(out[s], ref[0]) < TOL[dt]


@pytest.mark.parametrize("V,temperature,top_k,top_p", [(1000, 0.6, 50, 0.9), (128256, 0.6, 50, 0.9), (1000, 1.0, 0, 0.75), (1000, 1.3, 7, 1.0), (777, 0.8, 1, 0.5)])
def test_sample_select_follows_hf_logits_warpers(V, temperature, top_k, top_p):
    """Sampled selection: the survivor set must be the one HF's own TemperatureLogitsWarper ->
    TopKLogitsWarper -> TopPLogitsWarper leave (the installed transformers' classes are the oracle), the drawn token is the
    inverse-CDF pick over that set at the kernel's uniform (restated on the host), finished rows emit pad, top_k = 1 is greedy."""
    from transformers.generation.logits_process import TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper
    B, seed, n_draws = 6, 0xABCDEF0123, 40
    logits = rnd(B, V, seed=51, std=2.5)
    scores = TemperatureLogitsWarper(temperature)(None, logits.clone())
    if top_k > 0:
        scores = TopKLogitsWarper(top_k=top_k)(None, scores)
    if top_p < 1.0:
        scores = TopPLogitsWarper(top_p=top_p)(None, scores)
    allowed = torch.isfinite(scores)
    probs = torch.softmax(scores.double(), -1)
    dl = logits.to(dev())
    unfinished = torch.tensor([1, 1, 0, 1, 1, 1], dtype=torch.int32, device=dev())
    ctx = torch.full((B,), 10, dtype=torch.int32, device=dev())
    cnt = torch.zeros(B, dtype=torch.int32, device=dev())
    fin = torch.zeros(B, dtype=torch.int32, device=dev())
    nxt = torch.zeros(B, dtype=torch.int32, device=dev())
    out = torch.full((B, n_draws), -1, dtype=torch.int32, device=dev())
    for _ in range(n_draws):
        ops.sample_select(dl, temperature, top_k, top_p, seed, [3], 5, True, unfinished, ctx, cnt, fin, nxt, out)
    out = out.cpu()
    assert int(cnt[0]) == n_draws and int(ctx[0]) == 10 + n_draws
    exact = total = 0
    for b in range(B):
        if b == 2:
            assert bool((out[b] == 5).all())                       # finished row: pad
            continue
        for step in range(n_draws):
            tok = int(out[b, step])
            assert bool(allowed[b, tok]), (b, step, tok)           # never outside HF's survivor set
            if tok == 3:                                           # an EOS draw finishes the row: pad from there on
                assert bool((out[b, step + 1:] == 5).all())
                break
            cdf = torch.cumsum(probs[b], 0)
            ref = int(torch.searchsorted(cdf, torch.tensor(ops.sample_uniform(seed, b, step), dtype=torch.float64), right=True))
            total += 1
            exact += int(ref == tok)
    assert exact >= 0.97 * total, (exact, total)                   # fp32 vs fp64 CDF: a draw that lands on a boundary may move by one survivor
    if top_k == 1:
        assert bool((out[0] == int(logits[0].argmax())).all()) or 3 in out[0].tolist()
    # the survivor sets themselves: every surviving token is drawn at some seed when the set is small
    if int(allowed[0].sum()) <= 8:
        seen = set()
        cnt.zero_(); unfinished.fill_(1)
        out2 = torch.zeros((B, 1), dtype=torch.int32, device=dev())
        for sd in range(400):
            cnt.zero_()
            ops.sample_select(dl, temperature, top_k, top_p, sd, [], 5, False, unfinished, ctx, cnt, fin, nxt, out2)
            seen.add(int(out2[0, 0]))
        assert seen == set(torch.nonzero(allowed[0]).flatten().tolist())


@pytest.mark.parametrize("M", [512, 1024])
def test_gemm_packed_at_the_benchmarked_decode_shapes(M):
    """The five Linears of a Llama-3.2-3B decode step at bench.py's row counts, bf16, each as the decode graph launches it
    (gemm_stream_wide_kernel unsplit / K-split + reduce by the library's own rule): qkv-shaped N = 5 120 with in-kernel RMS
    statistics, o (K = 3 072, residual, row scale out), gate/up N = 16 384 with SiLU-mul and the handed-in scale, down (K = 8 192,
    residual), and an lm_head slice in fp32 — against an fp32 reference, and against the skinny family on 16-row slices."""
    dt = torch.bfloat16
    H, Fd, NQ = 3072, 8192, 5120
    x = rnd(M, H, seed=71)
    xd, xf = x.to(dev(), dt), q(x, dt)
    rstd = torch.rsqrt(xf.pow(2).mean(-1) + 1e-5)
    # qkv-shaped: fused RMSNorm (gain folded into W on the host, as weights.build_decode_weights does)
    wq = rnd(NQ, H, seed=72, std=H ** -0.5)
    wqp = ops.pack_weight(wq.to(dev(), dt))
    out = ops.gemm_decode(xd, wqp, NQ, fuse_rms=True, eps=1e-5)
    ref = (xf * rstd[:, None]) @ q(wq, dt).T
    assert rel_err(out.float().cpu(), ref) < TOL[dt]
    sk = torch.cat([ops.gemm_decode(xd[r:r + 16], wqp, NQ, fuse_rms=True, eps=1e-5) for r in (0, M // 2, M - 16)])
    assert rel_err(torch.cat([out[r:r + 16] for r in (0, M // 2, M - 16)]).float().cpu(), sk.float().cpu()) < 6e-3   # one bf16 rounding of the outputs apart
    # o: K = 3072 -> N = 3072 with residual, emitting the next RMSNorm's row scale
    wo, res = rnd(H, H, seed=73, std=H ** -0.5), rnd(M, H, seed=74)
    r_out = torch.empty(M, device=dev(), dtype=torch.float32)
    o = ops.gemm_decode(xd, ops.pack_weight(wo.to(dev(), dt)), H, residual=res.to(dev(), dt), rstd_out=r_out, eps=1e-5)
    o_ref = xf @ q(wo, dt).T + q(res, dt)
    assert rel_err(o.float().cpu(), o_ref) < TOL[dt]
    if L.lib().sl_gemm_split_count(M, H, H, L.dtype_code(dt)) > 1:          # the scale rides on the reduce pass
        assert rel_err(r_out.cpu(), torch.rsqrt(o.float().cpu().pow(2).mean(-1) + 1e-5)) < 2e-3
    # gate/up: N = 16 384 interleaved pairs, SiLU-mul epilogue, handed-in scale
    g, u = rnd(Fd, H, seed=75, std=H ** -0.5), rnd(Fd, H, seed=76, std=H ** -0.5)
    wgu = ops.pack_weight(weights.interleave_gate_up(g, u).to(dev(), dt))
    a = ops.gemm_decode(xd, wgu, 2 * Fd, act=L.ACT_SILU_MUL, fuse_rms=True, eps=1e-5, rstd_in=rstd.to(dev()))
    a_ref = F.silu((xf * rstd[:, None]) @ q(g, dt).T) * ((xf * rstd[:, None]) @ q(u, dt).T)
    assert rel_err(a.float().cpu(), a_ref) < TOL[dt]
    # down: K = 8192 -> N = 3072 with residual
    y = rnd(M, Fd, seed=77)
    wd = rnd(H, Fd, seed=78, std=Fd ** -0.5)
    d = ops.gemm_decode(y.to(dev(), dt), ops.pack_weight(wd.to(dev(), dt)), H, residual=res.to(dev(), dt))
    assert rel_err(d.float().cpu(), q(y, dt) @ q(wd, dt).T + q(res, dt)) < TOL[dt]
    # lm_head slice: fp32 logits of 20 000 vocabulary rows
    wl = rnd(20000, H, seed=79, std=H ** -0.5)
    lg = ops.gemm_decode(xd, ops.pack_weight(wl.to(dev(), dt)), 20000, fuse_rms=True, eps=1e-5, out_f32=True)
    assert rel_err(lg.cpu(), (xf * rstd[:, None]) @ q(wl, dt).T) < 2e-3


@pytest.mark.parametrize("M,N,K", [(499, 1024, 1024), (4100, 1024, 4096), (20000, 1280, 1280), (65, 64, 64)])
def test_gemm_row_statistics_epilogue_equals_sums_of_the_stored_rows(M, N, K):
    """stats_out (the producer side of the LayerNorm fold): {sum, sum of squares} of every 64-column segment of the rows the GEMM
    stored — of the bf16-rounded values, the ones the next kernel reads — with bias and residual in; finalize then gives the
    {mean, rstd} torch computes from the same stored rows, and sl_layernorm_stats (the first layer's input) agrees with both."""
    dt = torch.bfloat16
    A, W, b, R = rnd(M, K, seed=61), rnd(N, K, seed=62, std=K ** -0.5), rnd(N, seed=63), rnd(M, N, seed=64) + 0.5
    assert L.lib().sl_gemm_ln_fold_ok(M, N, K, L.dtype_code(dt)) == 1
    out = torch.empty((M, N), device=dev(), dtype=dt)
    stats = torch.full((M, N // 64, 2), float("nan"), device=dev(), dtype=torch.float32)
    Ad, Wd, bd, Rd = A.to(dev(), dt), W.to(dev(), dt), b.to(dev(), dt), R.to(dev(), dt)
    ops.gemm_ex(Ad, Wd, M=M, N=N, K=K, lda=K, ldw=K, out=out, bias=bd, residual=Rd, ldr=N, stats_out=stats)
    plain = ops.gemm(Ad, Wd, bias=bd, residual=Rd)
    assert torch.equal(out, plain)                                       # the statistics ride along; the product is untouched
    seg = out.float().view(M, N // 64, 64)
    assert rel_err(stats[..., 0].cpu(), seg.sum(-1).cpu()) < 1e-5
    assert rel_err(stats[..., 1].cpu(), (seg * seg).sum(-1).cpu()) < 1e-5
    eps = 1e-5
    mr = ops.layernorm_stats_finalize(stats, N, eps)
    x = out.double()
    mean, var = x.mean(-1), x.var(-1, unbiased=False)
    want = torch.stack([mean, (var + eps).rsqrt()], dim=-1).float()
    assert rel_err(mr.cpu(), want.cpu()) < 1e-5
    assert rel_err(ops.layernorm_stats(out, eps).cpu(), want.cpu()) < 1e-5


@pytest.mark.parametrize("act", [L.ACT_NONE, L.ACT_GELU])
@pytest.mark.parametrize("M,N,K", [(499, 3072, 1024), (4100, 4096, 1024), (20000, 3840, 1280)])
def test_gemm_layernorm_fold_equals_layernorm_then_linear(act, M, N, K):
    """The consumer side: Linear(LayerNorm(x)) = rstd (x W'^T - mean u) + c with W' = W o gain, u = rowsum(W'), c = W beta + bias
    (weights.build_layernorm_fold) against fp32 LayerNorm -> Linear on the same bf16 inputs, and within the same distance of it as
    the unfolded kernels (LayerNorm rounding its output to bf16, then the plain GEMM).  Rows carry a mean of 3 standard
    deviations so the mean term is not a rounding-level correction, and half of them two outlier channels (80 x, + 40)."""
    dt = torch.bfloat16
    x = rnd(M, K, seed=71) * (1 + rnd(M, 1, seed=72).abs()) + 3.0 * rnd(M, 1, seed=73)
    x[::2, 7] *= 80.0                       # every other row with the outlier channels pretrained residual streams carry:
    x[::2, 300] += 40.0                     # the variance comes from sum(x^2) - mean^2 of the producer's fp32 sums
    x = x.to(dt)
    g, beta = (1 + 0.3 * rnd(K, seed=74)).to(dt), (0.2 * rnd(K, seed=75)).to(dt)
    W, b = rnd(N, K, seed=76, std=K ** -0.5).to(dt), rnd(N, seed=77).to(dt)
    eps = 1e-5
    ref = F.layer_norm(x.float(), (K,), g.float(), beta.float(), eps) @ W.float().T + b.float()
    ref = F.gelu(ref) if act == L.ACT_GELU else ref
    xd, gd, bed, Wd, bd = (t.to(dev()) for t in (x, g, beta, W, b))
    Wf = (Wd.float() * gd.float()[None, :]).to(dt)
    u = Wf.float().sum(1).contiguous()
    c = (Wd.float() @ bed.float() + bd.float()).contiguous()
    mr = ops.layernorm_stats(xd, eps)
    out = torch.empty((M, N), device=dev(), dtype=dt)
    ops.gemm_ex(xd, Wf, M=M, N=N, K=K, lda=K, ldw=K, out=out, act=act, ln_mr=mr, ln_u=u, ln_c=c)
    unfolded = ops.gemm(ops.layernorm(xd, gd, bed, eps), Wd, bias=bd, act=act)
    e_fold, e_plain = rel_err(out.float().cpu(), ref), rel_err(unfolded.float().cpu(), ref)
    assert e_fold < TOL[dt] and e_fold < 1.5 * e_plain + 1e-3, (e_fold, e_plain)


def test_gemm_layernorm_fold_arguments_are_checked():
    """ln_* come together, exclude a separate bias, and need a product the LDS-DMA tiled kernels take (bf16, N % 64 == 0; any row count:
    the fold is a property of the model, not of the call)."""
    dt = torch.bfloat16
    M, N, K = 256, 128, 128
    x, W = rnd(M, K, seed=81).to(dev(), dt), rnd(N, K, seed=82).to(dev(), dt)
    mr, u, c = (torch.zeros(n, device=dev()) for n in (2 * M, N, N))
    out = torch.empty((M, N), device=dev(), dtype=dt)
    for kw in (dict(ln_mr=mr, ln_u=u), dict(ln_mr=mr, ln_u=u, ln_c=c, bias=W[0].contiguous())):
        with pytest.raises(RuntimeError, match="LayerNorm fold needs"):
            ops.gemm_ex(x, W, M=M, N=N, K=K, lda=K, ldw=K, out=out, **kw)
    # short products keep the fold (they stay on the tiled kernels): same bits as the first 32 rows of the full product
    full = ops.gemm_ex(x, W, M=M, N=N, K=K, lda=K, ldw=K, out=torch.empty_like(out), ln_mr=mr, ln_u=u, ln_c=c)
    short = ops.gemm_ex(x, W, M=32, N=N, K=K, lda=K, ldw=K, out=torch.empty((32, N), device=dev(), dtype=dt), ln_mr=mr, ln_u=u, ln_c=c)
    assert torch.equal(short, full[:32])
    with pytest.raises(RuntimeError, match="ln_\\* / stats_out need"):
        ops.gemm_ex(x.float(), W.float(), M=M, N=N, K=K, lda=K, ldw=K, out=out.float(), stats_out=mr)
    assert L.lib().sl_gemm_ln_fold_ok(32, N, K, L.dtype_code(dt)) == 1 and L.lib().sl_gemm_ln_fold_ok(M, 100, K, L.dtype_code(dt)) == 0


@pytest.mark.parametrize("M,N,K", [(40000, 1000, 2048), (33000, 3072, 1024), (16500, 2056, 192)])
def test_gemm_swapped_operand_epilogue_equals_lds_turned_rows_epilogue(M, N, K, tuning):
    """The 256-tile kernel's two bf16 store paths — operands exchanged in the MFMA so a lane owns 16 columns of a row and stores
    them from registers (default where rows are 8-element aligned), and the accumulator tile turned through LDS
    (SL_NO_SWAP_EPILOGUE=1) — on the same launches: bias, bias + GELU, bias + residual, ragged M and N edges, a batch of two.
    Same products in the same order; only the store pattern differs, so the outputs are bit-identical."""
    dt = torch.bfloat16
    A, W, b, R = rnd(M, K, seed=91), rnd(N, K, seed=92, std=K ** -0.5), rnd(N, seed=93), rnd(M, N, seed=94)
    Ad, Wd, bd, Rd = A.to(dev(), dt), W.to(dev(), dt), b.to(dev(), dt), R.to(dev(), dt)

    def run():
        outs = [ops.gemm(Ad, Wd, bias=bd), ops.gemm(Ad, Wd, bias=bd, act=L.ACT_GELU), ops.gemm(Ad, Wd, bias=bd, residual=Rd), ops.gemm(Ad, Wd)]
        o2 = torch.empty((2, M // 2, N), device=dev(), dtype=dt)
        ops.gemm_ex(Ad, Wd, M=M // 2, N=N, K=K, lda=K, ldw=K, out=o2, ldc=N, bias=bd, batch=2, strideA=(M // 2) * K, strideC=(M // 2) * N)
        o3, pre = torch.empty((M, N), device=dev(), dtype=dt), torch.empty((M, N), device=dev(), dtype=dt)
        ops.gemm_ex(Ad, Wd, M=M, N=N, K=K, lda=K, ldw=K, out=o3, bias=bd, act=L.ACT_GELU, aux_out=pre)      # the training forward's FFN1: result + pre-activation
        return outs + [o2, o3, pre]

    swapped = run()
    tuning("SL_NO_SWAP_EPILOGUE", "1")
    turned = run()
    for s_, t_ in zip(swapped, turned):
        assert torch.equal(s_, t_)
    ref = F.gelu(q(A, dt) @ q(W, dt).T + q(b, dt))
    assert rel_err(swapped[1].float().cpu(), ref) < TOL[dt]
    assert torch.equal(swapped[4].view(-1, N)[: 2 * (M // 2)], swapped[0][: 2 * (M // 2)])
    assert torch.equal(swapped[5], swapped[1]) and torch.equal(swapped[6], swapped[0])        # GELU output / pre-activation = the bias-only product


@pytest.mark.parametrize("form", ["streamk", "splitk"])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("M,N,K", [(400, 3072, 16384), (634, 1000, 2048), (1300, 520, 4096), (257, 256, 8192), (3200, 3072, 1024), (634, 3072, 8192),
                                   (634, 3072, 3072), (1024, 1024, 8000), (634, 5120, 3072)])
def test_gemm_stream_k_equals_the_tile_kernel_product(M, N, K, dt, form, tuning):
    """Stream-K form of the 256-tile GEMM (sl_gemm_ex_args.sk_ws): every CU takes an equal run of (tile, K slab) units, tiles cut
    between blocks are summed through the workspace inside the launch.  Forced on (SL_STREAM_K=2) over ragged M / N edges, tiles
    split two to five ways, whole tiles inside a block's run, bf16 (register epilogue and, with an odd N, the LDS-turned one) and
    fp32, plain / + residual / fp32 accumulation; repeated launches on one workspace (flags are cleared by their consumers).
    Against the fp32 product, and against the one-block-per-tile result within a few fp32 ulps of re-association.
    form = splitk (round 5, the default where a workspace is given and the product has few tiles): S equal K runs as one batched launch
    of the ordinary tile kernels + a reduce launch that adds the runs in run order (634 x 3 072 x 8 192: 4 runs; x 3 072: 4; 1 024 x 1 024
    x 8 000: 5; 634 x 5 120 x 3 072: 2) — same checks, and bitwise the same result launch after launch."""
    if dt == torch.float32 and M * N * K > 3e10:
        pytest.skip("fp32 MFMA at this size adds nothing")
    A, W, R = rnd(M, K, seed=191), rnd(N, K, seed=192, std=K ** -0.5), rnd(M, N, seed=193)
    Ad, Wd, Rd = A.to(dev(), dt), W.to(dev(), dt), R.to(dev(), dt)
    ws = ops.streamk_workspace(dev())
    acc0 = rnd(M, N, seed=194).to(dev())

    def run(sk):
        kw = dict(M=M, N=N, K=K, lda=K, ldw=K, sk_ws=ws if sk else None)
        o1 = ops.gemm_ex(Ad, Wd, out=torch.empty((M, N), device=dev(), dtype=dt), **kw)
        o2 = ops.gemm_ex(Ad, Wd, out=torch.empty((M, N), device=dev(), dtype=dt), residual=Rd, ldr=N, **kw)
        o3 = acc0.clone()
        ops.gemm_ex(Ad, Wd, out=o3, residual=o3, ldr=N, out_f32=True, residual_f32=True, **kw)
        return o1, o2, o3

    tile = run(False)
    tuning("SL_STREAM_K", "2" if form == "streamk" else "0")
    tuning("SL_SPLIT_K", "0" if form == "streamk" else "1")
    first = None
    for rep in range(3):
        sk = run(True)
        if first is None:
            first = [t_.clone() for t_ in sk]
        assert all(torch.equal(a_, b_) for a_, b_ in zip(sk, first))          # fixed summation order: reproducible bit for bit
        ref = q(A, dt) @ q(W, dt).T
        assert rel_err(sk[0].float().cpu(), ref) < TOL[dt]
        assert rel_err(sk[1].float().cpu(), ref + q(R, dt)) < TOL[dt]
        assert rel_err(sk[2].cpu(), ref + acc0.cpu()) < TOL[dt]
        for a_, b_ in zip(sk, tile):
            assert rel_err(a_.float().cpu(), b_.float().cpu()) < (5e-6 if dt == torch.float32 else 4e-3)     # re-association over up to 16 384 terms
    assert int(ws[:1024].to(torch.int32).abs().sum()) == 0          # every raised flag was consumed and cleared


def test_gemm_layernorm_fold_is_bit_identical_across_tile_kernels(tuning):
    """Row statistics and the folded consumer from the 256-tile kernel's register epilogue, from its LDS-turned rows epilogue
    and from the 128-tile kernel (what a short batch takes): one summation tree, so the same bits — a batch of utterances and
    the utterances alone must encode identically whichever kernel their row count selects."""
    dt = torch.bfloat16
    M, N, K = 33000, 1024, 1024
    A, W, b, R = rnd(M, K, seed=95), rnd(N, K, seed=96, std=K ** -0.5), rnd(N, seed=97), rnd(M, N, seed=98)
    Ad, Wd, bd, Rd = A.to(dev(), dt), W.to(dev(), dt), b.to(dev(), dt), R.to(dev(), dt)
    u, c = rnd(N, seed=99).to(dev()), rnd(N, seed=100).to(dev())

    def run(rows):
        out, st = torch.empty((rows, N), device=dev(), dtype=dt), torch.empty((rows, N // 64, 2), device=dev())
        ops.gemm_ex(Ad, Wd, M=rows, N=N, K=K, lda=K, ldw=K, out=out, bias=bd, residual=Rd, ldr=N, stats_out=st)
        mr = ops.layernorm_stats_finalize(st, N, 1e-5)
        out2 = torch.empty((rows, N), device=dev(), dtype=dt)
        ops.gemm_ex(out, Wd, M=rows, N=N, K=K, lda=K, ldw=K, out=out2, act=L.ACT_GELU, ln_mr=mr, ln_u=u, ln_c=c)
        return out, st, out2

    big = run(M)                       # 256-tile kernel, swapped operands
    small = run(300)                   # 128-tile kernel, rows epilogue
    tuning("SL_NO_SWAP_EPILOGUE", "1")
    turned = run(M)                    # 256-tile kernel, rows epilogue
    for x, y, z in zip(big, turned, small):
        assert torch.equal(x, y) and torch.equal(x[:300], z)


def test_layernorm_fold_build_and_finalize_with_a_dc_offset():
    """sl_layernorm_fold_build (the weight side of the LayerNorm fold, one HIP launch per Linear instead of torch's product + row
    reduction + vendor GEMV) against its defining formulas; and sl_layernorm_stats_finalize on rows whose mean is 30 x their standard
    deviation (segments merged Chan-style: the row-wide E[x^2] - mean^2 it replaces loses three digits there) against the two-pass
    statistics of the same rows."""
    dt = torch.bfloat16
    N, K = 3072, 1024
    W, g, b, bias = rnd(N, K, seed=301, std=K ** -0.5), 1.0 + rnd(K, seed=302, std=0.1), rnd(K, seed=303, std=0.1), rnd(N, seed=304)
    Wd, gd, bd, biasd = (t.to(dev(), dt) for t in (W, g, b, bias))
    Wf, u, c = torch.empty_like(Wd), torch.empty(N, device=dev()), torch.empty(N, device=dev())
    L.check(L.lib().sl_layernorm_fold_build(L.ptr(Wd), L.ptr(gd), L.ptr(bd), L.ptr(biasd), L.ptr(Wf), L.ptr(u), L.ptr(c), N, K, L.dtype_code(dt), L.stream_ptr()),
            "sl_layernorm_fold_build")
    W0 = Wd.float()
    ref_wf = (W0 * gd.float()[None, :]).to(dt)
    assert torch.equal(Wf, ref_wf)
    assert rel_err(u.cpu(), ref_wf.float().sum(dim=1).cpu()) < 1e-6
    assert rel_err(c.cpu(), (W0 @ bd.float() + biasd.float()).cpu()) < 1e-6
    # finalize under a DC offset: 1024-wide rows, mean 30, std 1, values as a bf16 producer stores them
    rows, cols = 4096, 1024
    x = (30.0 + rnd(rows, cols, seed=305)).to(dt)
    xf = x.float()
    segs = cols // 64
    st = torch.stack([xf.view(rows, segs, 64).sum(-1), (xf * xf).view(rows, segs, 64).sum(-1)], dim=-1).contiguous().to(dev())
    mr = ops.layernorm_stats_finalize(st, cols, 1e-5).cpu()
    mean = xf.double().mean(dim=1)
    rstd = 1.0 / torch.sqrt(xf.double().var(dim=1, unbiased=False) + 1e-5)
    assert float((mr[:, 0].double() - mean).abs().max()) < 1e-4
    assert float(((mr[:, 1].double() - rstd) / rstd).abs().max()) < 2e-3        # E[x^2] - mean^2 over the whole row: ~2e-2 here


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_gemm_phased_loop_is_bit_identical_to_the_one_barrier_per_slab_loop(dt, tuning):
    """The staggered two-phase main loop of the 256-tile GEMM (LDS-DMA in flight across barriers, wave halves one barrier apart)
    adds every output element's products in the same order as the round-3 loop (SL_T256_PHASED=0), so the two must agree BIT FOR BIT
    — which makes this a race screen as well: a fragment read that overtook its DMA, or a slot re-filled under a reader, shows up as a
    differing tile.  Shapes with 1, 2, 3 and many K slabs (prologue / tail paths), ragged M and N edges, every epilogue family; each
    launched repeatedly, back to back with other work so that blocks start under uneven load."""
    shapes = [(40000, 1024, 64), (33000, 768, 128), (20000, 1024, 192), (66000, 512, 1024), (9000, 4096, 1024), (2500, 16384, 3072), (70000, 1000, 2048)]
    if dt == torch.float32:
        shapes = [(40000, 1024, 32), (20000, 768, 96), (33000, 512, 512)]
    for M, N, K in shapes:
        A, W, b, R = rnd(M, K, seed=M % 97), rnd(N, K, seed=N % 89, std=K ** -0.5), rnd(N, seed=5), rnd(M, N, seed=6)
        Ad, Wd, bd, Rd = A.to(dev(), dt), W.to(dev(), dt), b.to(dev(), dt), R.to(dev(), dt)
        noise = torch.randn(2048, 2048, device=dev())

        def run():
            outs = [ops.gemm(Ad, Wd), ops.gemm(Ad, Wd, bias=bd, residual=Rd), ops.gemm(Ad, Wd, bias=bd, act=L.ACT_GELU)]
            if N % 32 == 0:
                outs.append(ops.gemm(Ad, Wd, act=L.ACT_SILU_MUL))
            outs.append(ops.gemm(Ad, Wd, out_f32=True))
            return outs

        tuning("SL_T256_PHASED", "0")
        ref = run()
        tuning("SL_T256_PHASED", "1")
        for rep in range(4):
            (noise @ noise).sum()                                   # other kernels in flight around the launches
            for got, want in zip(run(), ref):
                assert torch.equal(got, want), (M, N, K, rep)
    assert rel_err(ref[0].float().cpu(), q(A, dt) @ q(W, dt).T) < TOL[dt]
