"""GPU: whole-model parity of the HIP path (through the C ABI and the host mirrors of the reference
classes) against the golden fixtures generated from the reference and against the CPU oracle.

fp32 mode: hidden states <= 1e-4 relative L2 vs the reference fixtures, greedy ids IDENTICAL.
bf16 mode: weights rounded to bf16 on both sides; hidden states <= 3e-2 relative L2 (stated tolerance).
"""
import os

import pytest
import torch

from conftest import GOLDEN, golden, pkg, rel_err, t
from oracle import hubert_oracle as ho
from oracle import kd_oracle as ko
from oracle import llama_oracle as lo
from oracle.golden_cfgs import TINY_HUBERT, TINY_LLAMA, TINY_MHA, WIDE_HUBERT, WIDE_LLAMA

pytestmark = pytest.mark.gpu

ri = pkg("random_init")
cfgm = pkg("config")
enc_mod = pkg("audio_encoder")
llama_mod = pkg("audio_llama")
weights = pkg("weights")
L = pkg("_lib")
inf_mod = pkg("inference")
utils = pkg("utils")

DEV = "cuda:0"
F32_TOL, BF16_TOL = 1e-4, 3e-2


def hubert_arch(c):
    return weights.HubertArch(c.conv_dim, c.conv_kernel, c.conv_stride, c.hidden_size, c.num_hidden_layers,
                              c.num_attention_heads, c.intermediate_size, c.num_conv_pos_embeddings,
                              c.num_conv_pos_embedding_groups, c.layer_norm_eps)


def llama_arch(c):
    return weights.LlamaArch(c.hidden_size, c.num_hidden_layers, c.num_attention_heads, c.num_key_value_heads, c.head_dim,
                             c.intermediate_size, c.vocab_size, c.rms_norm_eps, c.rope_theta, c.rope_scaling,
                             c.tie_word_embeddings, tuple(c.eos_token_ids), c.pad_token_id)


def make_encoder(c, llm_dim, seed, dtype, method="pool", **sdkw):
    conf = cfgm.from_dict(dict(model=dict(audio_encoder=dict(base="hubert", type="synthetic", downsample_method=method,
                                                             downsample_factor=4, pooling=dict(kernel_size=8, stride=4)),
                                          llm_embedding_channels=llm_dim, llm_type=utils.LLAMA_ID)))
    enc = enc_mod.AudioEncoder(conf, DEV, dtype=dtype, arch=hubert_arch(c))
    sd = ri.hubert_encoder_state_dict(c, llm_dim, seed=seed, downsample=method, **sdkw)
    enc.load_state_dict(sd)
    return enc.eval().to(DEV), sd


def make_llama(c, seed, dtype, max_ctx=256):
    sd = ri.llama_state_dict(c, seed=seed)
    llm = llama_mod.AudioLlamaForCausalLM(llama_arch(c), dict(sd), torch_dtype=dtype, device=DEV, max_ctx=max_ctx)
    return llm, sd


@pytest.mark.parametrize("n", [16000, 32000])
def test_encoder_tiny_fp32_vs_reference_fixture(n):
    g = golden(f"enc_tiny_pool_{n}")
    enc, _ = make_encoder(TINY_HUBERT, 256, int(g["weight_seed"]), torch.float32)
    wave = ri.synthetic_waveform(n, seed=int(g["wave_seed"]))
    out, P, last_hidden, T = enc.encode_packed([wave], want_last_hidden=True)
    assert T[0] == TINY_HUBERT.num_frames(n) and P[0] == g["audio_embeds"].shape[1]
    assert rel_err(last_hidden.cpu(), t(g["last_hidden_state"])[0]) < F32_TOL
    assert rel_err(out.cpu(), t(g["audio_embeds"])[0]) < F32_TOL
    assert rel_err(enc(wave[None].to(DEV)).cpu(), t(g["audio_embeds"])) < F32_TOL


def test_encoder_weight_norm_legacy_keys_and_nested_checkpoint():
    g = golden("enc_tiny_pool_16000")
    enc, sd = make_encoder(TINY_HUBERT, 256, int(g["weight_seed"]), torch.float32, weight_norm_keys="legacy")
    enc.load_state_dict({"audio_encoder": sd, "epoch": 3})  # trainer-style checkpoint (ref:trainer.py:516-528)
    wave = ri.synthetic_waveform(16000, seed=int(g["wave_seed"]))
    assert rel_err(enc(wave[None].to(DEV)).cpu(), t(g["audio_embeds"])) < F32_TOL


def test_encoder_stack_ctcpool_and_batch():
    g = golden("enc_tiny_stack_16000")
    enc, _ = make_encoder(TINY_HUBERT, 256, 12, torch.float32, method="stack")
    wave = ri.synthetic_waveform(16000, seed=int(g["wave_seed"]))
    assert rel_err(enc(wave[None].to(DEV)).cpu(), t(g["audio_embeds"])) < F32_TOL
    # T % 4 == 0: documented divergence from the reference's empty output -> T/4 rows
    g0 = golden("enc_tiny_stack_16720")
    wave = ri.synthetic_waveform(16720, seed=int(g0["wave_seed"]))
    assert enc(wave[None].to(DEV)).shape[1] == int(g0["T"]) // 4
    g = golden("enc_tiny_ctcpool_16000")
    enc, _ = make_encoder(TINY_HUBERT, 256, 13, torch.float32, method="ctc_pool")
    wave = ri.synthetic_waveform(16000, seed=int(g["wave_seed"]))
    out = enc(wave[None].to(DEV), [[tuple(r) for r in g["ranges"].tolist()]])
    assert rel_err(out.cpu(), t(g["audio_embeds"])) < F32_TOL
    g = golden("enc_tiny_pool_batch2")
    enc, _ = make_encoder(TINY_HUBERT, 256, 11, torch.float32)
    wave = torch.stack([ri.synthetic_waveform(24000, seed=int(s)) for s in g["wave_seeds"]])
    assert rel_err(enc(wave.to(DEV)).cpu(), t(g["audio_embeds"])) < F32_TOL


def test_encoder_ragged_batch_equals_single_utterances():
    enc, sd = make_encoder(TINY_HUBERT, 256, 11, torch.float32)
    waves = [ri.synthetic_waveform(n, seed=n) for n in (9000, 16000, 12345)]
    out, P, _, _ = enc.encode_packed(waves)
    r0 = 0
    for w, p in zip(waves, P):
        ref = ho.audio_encoder_forward(sd, TINY_HUBERT, w[None])[0]
        assert ref.shape[0] == p and rel_err(out[r0:r0 + p].cpu(), ref) < F32_TOL
        r0 += p


def test_encoder_short_utterance_alone_equals_inside_a_batch_bf16_fold():
    """the LayerNorm fold is decided per MODEL, not per call — a 1 s utterance (49 frames: fewer rows than the skinny-GEMM
    threshold) encoded alone must come out bit-identical to the same utterance inside a batch (bf16, folded layers)."""
    enc, _ = make_encoder(TINY_HUBERT, 256, 13, torch.bfloat16)
    waves = [ri.synthetic_waveform(n, seed=n) for n in (16000, 40000, 9000)]
    out, P, _, _ = enc.encode_packed(waves)
    r0 = 0
    for w, p in zip(waves, P):
        alone, Pa, _, _ = enc.encode_packed([w])
        assert Pa[0] == p and torch.equal(alone, out[r0:r0 + p]), w.numel()
        r0 += p


@pytest.mark.parametrize("dtype,tol", [(torch.float32, F32_TOL), (torch.bfloat16, BF16_TOL)])
def test_encoder_wide_hubert_large_width(dtype, tol):
    g = golden("enc_wide_pool_32000")
    enc, sd = make_encoder(WIDE_HUBERT, 3072, int(g["weight_seed"]), dtype)
    wave = ri.synthetic_waveform(32000, seed=int(g["wave_seed"]))
    out, P, last_hidden, T = enc.encode_packed([wave], want_last_hidden=True)
    if dtype == torch.float32:
        ref_h, ref_o = t(g["last_hidden_state"])[0], t(g["audio_embeds"])[0]
    else:  # same bf16-rounded weights on the oracle side (conv0 stays fp32 in the kernel)
        keep32 = "conv_layers.0."
        sdq = {k: (v if keep32 in k else v.to(torch.bfloat16).float()) for k, v in sd.items()}
        taps = {}
        ref_o = ho.audio_encoder_forward(sdq, WIDE_HUBERT, wave[None], taps=taps)[0]
        ref_h = taps["last_hidden_state"][0]
    assert rel_err(last_hidden.float().cpu(), ref_h) < tol
    assert rel_err(out.float().cpu(), ref_o) < tol


def test_encoder_layernorm_fold_against_the_unfolded_layers_and_the_oracle(monkeypatch):
    """bf16 inference folds each layer's two LayerNorms into the neighbouring GEMMs (sl_hubert_fold).  Both forms against the
    fp32 oracle on the same bf16-rounded weights: the folded one within the bf16 bound and no further from the oracle than the
    unfolded one by more than a rounding's worth; the two differ bit-wise (the fold is what ran); and after the weights move
    (refresh: a KD optimizer step) the folded copies follow."""
    enc, sd = make_encoder(WIDE_HUBERT, 3072, 31, torch.bfloat16)
    assert enc.weights.struct.fold
    waves = [ri.synthetic_waveform(n, seed=n) for n in (32000, 48000, 25000)]
    keep32 = "conv_layers.0."
    sdq = {k: (v if keep32 in k else v.to(torch.bfloat16).float()) for k, v in sd.items()}
    ref = torch.cat([ho.audio_encoder_forward(sdq, WIDE_HUBERT, w[None])[0] for w in waves])
    folded = enc.encode_packed(waves)[0].float().cpu()
    monkeypatch.setenv("SL_NO_LN_FOLD", "1")
    L.lib().sl_tuning_reload()
    try:
        plain = enc.encode_packed(waves)[0].float().cpu()
    finally:
        monkeypatch.undo()
        L.lib().sl_tuning_reload()
    e_f, e_p = rel_err(folded, ref), rel_err(plain, ref)
    assert e_f < BF16_TOL and e_f < 1.5 * e_p + 2e-3, (e_f, e_p)
    assert not torch.equal(folded, plain)
    gen = torch.Generator().manual_seed(5)
    sd2 = {k: (v * (1 + 0.5 * torch.randn(v.shape, generator=gen)) if k.endswith("layer_norm.weight") and ".layers." in k else v)
           for k, v in sd.items()}
    enc.weights.refresh(sd2)
    sdq2 = {k: (v if keep32 in k else v.to(torch.bfloat16).float()) for k, v in sd2.items()}
    ref2 = torch.cat([ho.audio_encoder_forward(sdq2, WIDE_HUBERT, w[None])[0] for w in waves])
    assert rel_err(enc.encode_packed(waves)[0].float().cpu(), ref2) < BF16_TOL
    assert rel_err(ref2, ref) > 3 * BF16_TOL      # the gains mattered: stale folded copies would have failed the line above


@pytest.mark.parametrize("name,cfg", [("tiny_gqa", TINY_LLAMA), ("tiny_mha", TINY_MHA)])
def test_llama_tiny_fp32_forward_and_generate(name, cfg):
    g = golden(f"llama_{name}")
    llm, _ = make_llama(cfg, int(g["weight_seed"]), torch.float32)
    gen = torch.Generator().manual_seed(int(g["embeds_seed"]))
    x = (torch.randn(1, int(g["S"]), cfg.hidden_size, generator=gen) * 0.05).to(DEV)
    out = llm(inputs_embeds=x, output_hidden_states=True)
    assert rel_err(out.logits.cpu(), t(g["logits"])) < F32_TOL
    assert rel_err(torch.stack(out.hidden_states).cpu(), t(g["hidden_states"])) < F32_TOL
    # response-only next-token loss of list-labels (ref:model/audio_llama.py:72-101) against the reference's own value
    labels = [t(g["labels"]).to(DEV)]
    loss = llm(inputs_embeds=x, labels=labels, attention_mask=torch.ones(1, int(g["S"]), dtype=torch.long, device=DEV)).loss
    assert abs(float(loss) - float(g["loss"])) < 1e-4 * max(1.0, abs(float(g["loss"])))
    # batch mean over two samples with different label lengths equals the oracle's restatement of the same rule
    x2 = torch.cat([x, x.flip(1)], 0)
    lab2 = [labels[0], labels[0][:4]]
    out2 = llm(inputs_embeds=x2, labels=lab2)
    ref2 = lo.response_only_loss(out2.logits.cpu(), [l.cpu() for l in lab2])
    assert abs(float(out2.loss) - float(ref2)) < 1e-4 * max(1.0, abs(float(ref2)))
    llm.generation_config.eos_token_id = None
    assert torch.equal(llm.generate(inputs_embeds=x, max_new_tokens=32).cpu(), t(g["ids_noeos"]))
    llm.generation_config.eos_token_id = list(cfg.eos_token_ids)
    assert torch.equal(llm.generate(inputs_embeds=x, max_new_tokens=32).cpu(), t(g["ids_eos"]))


@pytest.mark.parametrize("name,cfg", [("tiny_gqa", TINY_LLAMA), ("tiny_mha", TINY_MHA)])
def test_llama_left_padded_batch_forward(name, cfg):
    g = golden(f"llama_{name}_padbatch")
    llm, _ = make_llama(cfg, int(g["weight_seed"]), torch.float32)
    out = llm(inputs_embeds=t(g["x"]).to(DEV), attention_mask=t(g["mask"]).to(DEV), output_hidden_states=True)
    m = t(g["mask"]).bool()
    assert rel_err(out.logits.cpu()[m], t(g["logits"])[m]) < F32_TOL
    assert rel_err(out.hidden_states[-1].cpu()[m], t(g["last_hidden"])[m]) < F32_TOL


def test_llama_batched_generate_equals_single():
    cfg = TINY_LLAMA
    llm, sd = make_llama(cfg, 31, torch.float32)
    gen = torch.Generator().manual_seed(5)
    prompts = [torch.randn(n, cfg.hidden_size, generator=gen) * 0.05 for n in (9, 21, 14)]
    llm.generation_config.eos_token_id = list(cfg.eos_token_ids)
    ids = llm.generate(inputs_embeds=[p.to(DEV) for p in prompts], max_new_tokens=24).cpu()
    refs = [lo.greedy_generate(sd, cfg, p[None], 24, use_eos=True)[0] for p in prompts]
    width = max(r.shape[0] for r in refs)
    assert ids.shape[1] == width
    for b, r in enumerate(refs):
        assert torch.equal(ids[b, :r.shape[0]], r)
        assert bool((ids[b, r.shape[0]:] == cfg.pad_token_id).all())


def _decode_step_logits(llm, prompts, next_ids, shared_prefix=0):
    """prefill + ONE decode step through the C ABI (sl_llama_prefill, sl_llama_decode_step): fp32 logits of the new position."""
    import ctypes as C
    L = pkg("_lib")
    w, lib, B = llm._dev(), L.lib(), len(prompts)
    x = torch.cat([p.to(DEV, llm.dtype) for p in prompts]).contiguous()
    cu = [0]
    for p in prompts:
        cu.append(cu[-1] + p.shape[0])
    kv = llm._kv_cache(B, shared_prefix)
    ws = llm._workspace(lib.sl_generate_workspace_bytes(C.byref(w.struct), x.shape[0], B, 1))
    logits = torch.empty((B, llm.arch.vocab_size), device=DEV, dtype=torch.float32)
    ctx = torch.empty(B, device=DEV, dtype=torch.int32)
    L.check(lib.sl_llama_prefill(C.byref(w.struct), C.byref(kv), x.data_ptr(), (C.c_int32 * (B + 1))(*cu), B, logits.data_ptr(), ctx.data_ptr(),
                                 None, ws.data_ptr(), ws.numel(), L.stream_ptr()), "sl_llama_prefill")
    nid = torch.tensor(next_ids, dtype=torch.int32, device=DEV)
    L.check(lib.sl_llama_decode_step(C.byref(w.struct), C.byref(kv), nid.data_ptr(), ctx.data_ptr(), B, logits.data_ptr(), ws.data_ptr(),
                                     ws.numel(), L.stream_ptr()), "sl_llama_decode_step")
    return logits.cpu()


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, BF16_TOL)])
def test_llama_decode_step_large_batch_matches_small_batch(dtype, tol):
    """One decode step at 520 sequences (streaming GEMMs in 128-row blocks, tiled lm_head, single-pass attention in bf16) against
    the same sequences five at a time (skinny GEMMs, split attention): the two kernel families must agree."""
    cfg = TINY_LLAMA
    llm, _ = make_llama(cfg, 33, dtype)
    gen = torch.Generator().manual_seed(8)
    base = [torch.randn(n, cfg.hidden_size, generator=gen) * 0.05 for n in (9, 150, 14, 5, 77)]
    nxt = [11, 222, 3, 444, 55]
    small = _decode_step_logits(llm, base, nxt)
    B = 520
    big = _decode_step_logits(llm, [base[b % 5] for b in range(B)], [nxt[b % 5] for b in range(B)])
    for b in range(B):
        assert rel_err(big[b], small[b % 5]) < tol, b


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_llama_shared_prompt_prefix_is_read_from_slot_zero_with_unchanged_bits(dtype):
    """sl_kv_cache.shared_prefix (one prompt template in front of every utterance, ref:inference.py:95-113): an unshared prefill
    leaves bit-identical K / V rows at the prefix positions of every slot; with the promise given, prefill computes the prefix rows
    once (compacted batch, K / V copied to every slot, the prefix as one more attention sequence) and the single-pass decode
    attention reads them from slot 0 — ids, the decode step's logits and the whole cache after 24 steps are bit for bit those of
    shared_prefix = 0."""
    cfg = TINY_LLAMA
    llm, _ = make_llama(cfg, 35, dtype)
    gen = torch.Generator().manual_seed(12)
    P, B = 11, 40
    pre = torch.randn(P, cfg.hidden_size, generator=gen) * 0.05
    tails = [torch.randn(n, cfg.hidden_size, generator=gen) * 0.05 for n in (9, 140, 14, 5, 77, 30, 21, 1)]
    prompts = [torch.cat([pre, tails[b % len(tails)] * (1.0 + 0.01 * (b // len(tails)))]) for b in range(B)]
    lens = [int(p.shape[0]) for p in prompts]
    x = torch.cat(prompts).to(DEV, dtype)
    ids0, n0 = llm.generate_packed(x.clone(), lens, 24, use_eos=False)
    k0, v0 = llm._kv[0].clone(), llm._kv[1].clone()
    for c in (k0, v0):                                   # (layers, slots, n_kv, max_ctx, D)
        assert torch.equal(c[:, :B, :, :P], c[:, :1, :, :P].expand(-1, B, -1, -1, -1))
    llm._kv[0].zero_(); llm._kv[1].zero_()
    ids1, n1 = llm.generate_packed(x.clone(), lens, 24, use_eos=False, shared_prefix=P)
    assert n0 == n1 and torch.equal(ids0, ids1)
    assert torch.equal(llm._kv[0], k0) and torch.equal(llm._kv[1], v0)
    nxt = [(7 * b + 3) % cfg.vocab_size for b in range(B)]
    assert torch.equal(_decode_step_logits(llm, prompts, nxt, shared_prefix=P), _decode_step_logits(llm, prompts, nxt))
    with pytest.raises(pkg("_lib").SpeechLLMError):
        llm.generate_packed(x.clone(), lens, 4, use_eos=False, shared_prefix=min(lens) + 1)
    # a prompt that is nothing but the prefix: prefill falls back to the per-sequence pass, decode still reads slot 0
    prompts2 = prompts[:17] + [pre.clone()]
    lens2 = [int(p.shape[0]) for p in prompts2]
    x2 = torch.cat(prompts2).to(DEV, dtype)
    a, na = llm.generate_packed(x2.clone(), lens2, 8, use_eos=False)
    b, nb = llm.generate_packed(x2.clone(), lens2, 8, use_eos=False, shared_prefix=P)
    assert na == nb and torch.equal(a, b)


@pytest.mark.parametrize("B", [40, 72, 130, 260, 1100, 2048])      # 2048 = SL_MAX_DECODE_BATCH
def test_llama_large_batch_decode_equals_single(B):
    """Batches above 26 rows decode through gemm_stream.hip (loader wave, K-split + reduce, RMSNorm scales handed down the
    chain); every sequence must still produce exactly the ids it produces alone (fp32: bit-exact vs the oracle)."""
    cfg = TINY_LLAMA
    llm, sd = make_llama(cfg, 31, torch.float32)
    gen = torch.Generator().manual_seed(6)
    base = [torch.randn(n, cfg.hidden_size, generator=gen) * 0.05 for n in (9, 21, 14, 5, 30)]
    llm.generation_config.eos_token_id = list(cfg.eos_token_ids)
    refs = [lo.greedy_generate(sd, cfg, p[None], 20, use_eos=True)[0] for p in base]
    ids = llm.generate(inputs_embeds=[base[b % 5].to(DEV) for b in range(B)], max_new_tokens=20).cpu()
    assert ids.shape[0] == B and ids.shape[1] == max(r.shape[0] for r in refs)
    for b in range(B):
        r = refs[b % 5]
        assert torch.equal(ids[b, :r.shape[0]], r), b
        assert bool((ids[b, r.shape[0]:] == cfg.pad_token_id).all())
    if B == 2048:       # one more sequence than a generate call takes: refused with the instruction to split (the inference surface chunks)
        with pytest.raises(pkg("_lib").SpeechLLMError):
            llm.generate(inputs_embeds=[base[b % 5].to(DEV) for b in range(B + 1)], max_new_tokens=4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_llama_wide_llama32_width(dtype):
    g = golden("llama_wide")
    cfg = WIDE_LLAMA
    llm, sd = make_llama(cfg, int(g["weight_seed"]), dtype)
    gen = torch.Generator().manual_seed(int(g["embeds_seed"]))
    x = (torch.randn(1, int(g["S"]), cfg.hidden_size, generator=gen) * 0.02)
    out = llm(inputs_embeds=x.to(DEV), output_hidden_states=True)
    hs = torch.stack(out.hidden_states).float().cpu()
    if dtype == torch.float32:
        assert rel_err(hs, t(g["hidden_states"])) < F32_TOL
        assert rel_err(out.logits[:, -1].cpu(), t(g["last_logits"])) < F32_TOL
        llm.generation_config.eos_token_id = None
        ids = llm.generate(inputs_embeds=x.to(DEV), max_new_tokens=12).cpu()
        assert float(t(g["margins"]).min()) > 1e-4      # fixture steps are margin-qualified
        assert torch.equal(ids, t(g["ids_noeos"]))      # bit-exact greedy ids at fp32
    else:
        sdq = {k: v.to(torch.bfloat16).float() for k, v in sd.items()}
        ref = lo.llama_forward(sdq, cfg, x.to(torch.bfloat16).float(), output_hidden_states=True)
        assert rel_err(hs, torch.stack(ref["hidden_states"])) < BF16_TOL
        assert rel_err(out.logits[:, -1].cpu(), ref["logits"][:, -1]) < BF16_TOL


class StubTokenizer:
    def __init__(self, table):
        self.table = table

    def __call__(self, text, return_tensors="pt"):
        from types import SimpleNamespace
        return SimpleNamespace(input_ids=self.table[text].clone())

    def batch_decode(self, ids, **kw):
        return [" ".join(str(int(i)) for i in row) for row in ids]


def test_generate_audio_response_pipeline_fp32_ids_match_reference():
    g = golden("pipeline_tiny")
    cfg = TINY_LLAMA
    enc, _ = make_encoder(TINY_HUBERT, cfg.hidden_size, int(g["enc_seed"]), torch.float32)
    llm, _ = make_llama(cfg, int(g["llm_seed"]), torch.float32)
    tok = StubTokenizer({utils.LLAMA_PROMPT_PREFIX: t(g["prefix_ids"]), utils.LLAMA_PROMPT_SUFFIX: t(g["suffix_ids"]),
                         "EXTRA": t(g["text_prompt_ids"])})
    conf = enc.config
    inf = inf_mod.LLMSpeechTextInference(conf, None, DEV, tokenizer=tok, llm=llm, audio_encoder=enc, dtype=torch.float32)
    wave = ri.synthetic_waveform(int(g["n_samples"]), seed=int(g["wave_seed"])).numpy()
    text = inf.generate_audio_response(wave, max_new_tokens=40)
    assert torch.equal(inf.last_generate_ids.cpu(), t(g["ids_audio"]))
    assert text == " ".join(str(int(i)) for i in g["ids_audio"][0])
    inf.generate_audio_response(wave, additional_text_prompt="EXTRA", max_new_tokens=40)
    assert torch.equal(inf.last_generate_ids.cpu(), t(g["ids_text_audio"]))
    # batched extension: the reference utterance with and without the text prompt plus two other lengths, one ragged batch
    w2 = ri.synthetic_waveform(21000, seed=77).numpy()
    w3 = ri.synthetic_waveform(9000, seed=78).numpy()
    singles = []
    for a_, t_ in ((w2, ""), (w3, "EXTRA")):
        inf.generate_audio_response(a_, additional_text_prompt=t_, max_new_tokens=40)
        singles.append(inf.last_generate_ids.cpu()[0])
    texts = inf.generate_audio_responses([wave, wave, w2, w3], ["", "EXTRA", "", "EXTRA"], max_new_tokens=40)
    ids = inf.last_generate_ids.cpu()
    assert len(texts) == 4
    for row, ref in zip(ids, [t(g["ids_audio"])[0], t(g["ids_text_audio"])[0], singles[0], singles[1]]):
        assert torch.equal(row[:ref.shape[0]], ref) and bool((row[ref.shape[0]:] == cfg.pad_token_id).all())
    # more utterances than one generate call takes: answered in consecutive chunks, same ids per utterance (limit lowered for the test)
    L = pkg("_lib")
    keep = L.MAX_DECODE_BATCH
    try:
        L.MAX_DECODE_BATCH = 3
        texts7 = inf.generate_audio_responses([wave, wave, w2, w3, w2, wave, w3], ["", "EXTRA", "", "EXTRA", "", "", "EXTRA"], max_new_tokens=40)
    finally:
        L.MAX_DECODE_BATCH = keep
    ids7 = inf.last_generate_ids.cpu()
    assert len(texts7) == 7 and ids7.shape[0] == 7
    for row, ref in zip(ids7, [t(g["ids_audio"])[0], t(g["ids_text_audio"])[0], singles[0], singles[1], singles[0], t(g["ids_audio"])[0], singles[1]]):
        assert torch.equal(row[:ref.shape[0]], ref) and bool((row[ref.shape[0]:] == cfg.pad_token_id).all())
    assert texts7[0].startswith(text) and texts7[5].startswith(text)      # (the stub tokenizer prints a chunk's pad columns too)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, F32_TOL), (torch.bfloat16, BF16_TOL)])
def test_whisper_path_logmel_and_encoder_vs_reference_fixture(dtype, tol):
    from oracle.golden_cfgs import TINY_WHISPER as WC
    g = golden("whisper_tiny")
    conf = cfgm.from_dict(dict(model=dict(audio_encoder=dict(base="whisper", type="synthetic", downsample_method="pool", downsample_factor=4,
                                                             pooling=dict(kernel_size=8, stride=4)),
                                          llm_embedding_channels=256, llm_type=utils.LLAMA_ID)))
    arch = weights.WhisperArch(WC.d_model, WC.encoder_layers, WC.encoder_attention_heads, WC.encoder_ffn_dim, WC.num_mel_bins, WC.max_source_positions)
    enc = enc_mod.AudioEncoder(conf, DEV, dtype=dtype, arch=arch)
    enc.load_state_dict(ri.whisper_encoder_state_dict(WC, 256, seed=int(g["weight_seed"]))).eval().to(DEV)
    waves = [ri.synthetic_waveform(int(n), seed=int(s)).numpy() for n, s in zip(g["n_samples"], g["wave_seeds"])]
    feats = enc.feature_extractor(waves, return_tensors="pt", sampling_rate=16000).input_features
    assert feats.shape == t(g["input_features"]).shape
    assert float((feats.cpu() - t(g["input_features"])).abs().max()) < 2e-4      # fp32 DFT-as-GEMM vs torch.stft, after log10
    out = enc(feats)
    assert rel_err(out.float().cpu(), t(g["audio_embeds"])) < tol
    with pytest.raises(ValueError):
        enc(feats[:, :, :-2])
    # the trainer-side crop (ref:trainer.py:280-291)
    assert utils.compute_num_audio_embeds(20000) <= out.shape[1]


def test_whisper_generate_audio_response_matches_reference_trainer_path():
    """BASELINE configs[3] end to end through LLMSpeechTextInference with `base: whisper`: log-mel -> encoder -> crop to
    compute_num_audio_embeds -> prompt -> greedy decode; ids identical (fp32) to the fixture generated with the reference's
    AudioEncoder + HF feature extractor + AudioLlamaForCausalLM in the trainer's order of operations."""
    from oracle.golden_cfgs import TINY_WHISPER as WC
    g = golden("whisper_pipeline_tiny")
    cfg = TINY_LLAMA
    conf = cfgm.from_dict(dict(audio=dict(sampling_rate=16000),
                               model=dict(audio_encoder=dict(base="whisper", type="synthetic", downsample_method="pool", downsample_factor=4,
                                                             pooling=dict(kernel_size=8, stride=4)),
                                          llm_embedding_channels=cfg.hidden_size, llm_type=utils.LLAMA_ID)))
    arch = weights.WhisperArch(WC.d_model, WC.encoder_layers, WC.encoder_attention_heads, WC.encoder_ffn_dim, WC.num_mel_bins, WC.max_source_positions)
    enc = enc_mod.AudioEncoder(conf, DEV, dtype=torch.float32, arch=arch)
    enc.load_state_dict(ri.whisper_encoder_state_dict(WC, cfg.hidden_size, seed=int(g["enc_seed"]))).eval().to(DEV)
    llm, _ = make_llama(cfg, int(g["llm_seed"]), torch.float32)
    tok = StubTokenizer({utils.LLAMA_PROMPT_PREFIX: t(g["prefix_ids"]), utils.LLAMA_PROMPT_SUFFIX: t(g["suffix_ids"]), "EXTRA": t(g["text_prompt_ids"])})
    inf = inf_mod.LLMSpeechTextInference(conf, None, DEV, tokenizer=tok, llm=llm, audio_encoder=enc, dtype=torch.float32)
    for i, (n, seed) in enumerate(zip(g["n_samples"], g["wave_seeds"])):
        wave = ri.synthetic_waveform(int(n), seed=int(seed)).numpy()
        assert inf._whisper_audio_embeds(wave).shape[1] == int(g[f"num_audio_embeds_{i}"]) == utils.compute_num_audio_embeds(int(n))
        inf.generate_audio_response(wave, max_new_tokens=32)
        assert torch.equal(inf.last_generate_ids.cpu(), t(g[f"ids_audio_{i}"])), i
        inf.generate_audio_response(wave, additional_text_prompt="EXTRA", max_new_tokens=32)
        assert torch.equal(inf.last_generate_ids.cpu(), t(g[f"ids_text_audio_{i}"])), i
    # the batched surface with `base: whisper` (VERDICT r4 missing #3): every window through ONE log-mel + encoder pass, each utterance's rows
    # cropped to compute_num_audio_embeds, one ragged prefill + decode — per utterance the reference fixture's ids
    waves = [ri.synthetic_waveform(int(n), seed=int(seed)).numpy() for n, seed in zip(g["n_samples"], g["wave_seeds"])]
    k = len(waves)
    inf.generate_audio_responses(waves + waves, [""] * k + ["EXTRA"] * k, max_new_tokens=32)
    ids = inf.last_generate_ids.cpu()
    assert ids.shape[0] == 2 * k
    for i in range(k):
        for row, ref in ((ids[i], t(g[f"ids_audio_{i}"])[0]), (ids[k + i], t(g[f"ids_text_audio_{i}"])[0])):
            assert torch.equal(row[:ref.shape[0]], ref) and bool((row[ref.shape[0]:] == cfg.pad_token_id).all()), i


def _write_hf_llama_dir(path, cfg, sd, bos, eos, pad=None):
    """config.json + model.safetensors in the HF layout AudioLlamaForCausalLM.from_pretrained reads (ref:inference.py:47-52)."""
    import json
    from safetensors.torch import save_file
    hf = dict(architectures=["LlamaForCausalLM"], model_type="llama", hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
              num_attention_heads=cfg.num_attention_heads, num_key_value_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim,
              intermediate_size=cfg.intermediate_size, vocab_size=cfg.vocab_size, rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta,
              rope_scaling=(dict(cfg.rope_scaling, rope_type="llama3") if cfg.rope_scaling else None), tie_word_embeddings=cfg.tie_word_embeddings,
              bos_token_id=bos, eos_token_id=eos, pad_token_id=pad, torch_dtype="float32")
    with open(path / "config.json", "w") as f:
        json.dump(hf, f)
    keys = sorted(sd)
    half = len(keys) // 2                      # two shards, like a hub checkpoint
    save_file({k: sd[k].contiguous() for k in keys[:half]}, str(path / "model-00001-of-00002.safetensors"))
    save_file({k: sd[k].contiguous() for k in keys[half:]}, str(path / "model-00002-of-00002.safetensors"))
    # what Llama-3.2-3B-Instruct ships on the hub: sampling parameters AND do_sample = true
    with open(path / "generation_config.json", "w") as f:
        json.dump(dict(bos_token_id=bos, do_sample=True, eos_token_id=eos, temperature=0.6, top_p=0.9), f)


@pytest.mark.parametrize("family", ["llama3", "minichat"])
def test_reference_constructor_path_real_tokenizer_safetensors_and_local_hubert(tmp_path, family):
    """SURVEY §8 f1 + f2: `LLMSpeechTextInference(config, checkpoint_path, device)` with NOTHING injected — AutoTokenizer from a
    local directory (a byte-level BPE shaped like Llama-3's / a SentencePiece model shaped like MiniChat's, tests/golden/tokenizers),
    AudioLlamaForCausalLM.from_pretrained over sharded safetensors, the HuBERT architecture from a local config.json, the flat
    encoder checkpoint under the released spelling (weight_g / weight_v) via torch.load.  generate_audio_response (with and
    without additional_text_prompt) and generate_text_response must produce the oracle's greedy ids for the ids this tokenizer
    yields, and the text it decodes from them (skip_special_tokens)."""
    import json
    import shutil
    rec = json.load(open(os.path.join(GOLDEN, "tokenizers", "tokenizer_ids.json")))[family]
    name = "Llama-3.2-3B-Instruct" if family == "llama3" else "MiniChat-2-3B"
    llm_dir = tmp_path / name
    shutil.copytree(os.path.join(GOLDEN, "tokenizers", name), llm_dir)
    base = TINY_LLAMA if family == "llama3" else TINY_MHA
    import dataclasses
    cfg = dataclasses.replace(base, vocab_size=512, eos_token_ids=(rec["eos_token_id"],), pad_token_id=None)
    sd = ri.llama_state_dict(cfg, seed=77)
    _write_hf_llama_dir(llm_dir, cfg, sd, rec["bos_token_id"], rec["eos_token_id"])
    hub_dir = tmp_path / "hubert-tiny"
    hub_dir.mkdir()
    hc = TINY_HUBERT
    json.dump(dict(model_type="hubert", conv_dim=list(hc.conv_dim), conv_kernel=list(hc.conv_kernel), conv_stride=list(hc.conv_stride),
                   hidden_size=hc.hidden_size, num_hidden_layers=hc.num_hidden_layers, num_attention_heads=hc.num_attention_heads,
                   intermediate_size=hc.intermediate_size, num_conv_pos_embeddings=hc.num_conv_pos_embeddings,
                   num_conv_pos_embedding_groups=hc.num_conv_pos_embedding_groups, layer_norm_eps=hc.layer_norm_eps, feat_extract_norm="layer",
                   do_stable_layer_norm=True), open(hub_dir / "config.json", "w"))
    enc_sd = ri.hubert_encoder_state_dict(hc, cfg.hidden_size, seed=78, weight_norm_keys="legacy")
    ckpt = tmp_path / "audio_encoder.pt"
    torch.save(enc_sd, ckpt)                                          # flat state-dict, what ref:inference.py:24-26 loads
    conf = cfgm.from_dict(dict(audio=dict(sampling_rate=16000),
                               model=dict(audio_encoder=dict(base="hubert", type=str(hub_dir), downsample_method="pool", downsample_factor=4,
                                                             pooling=dict(kernel_size=8, stride=4)),
                                          llm_embedding_channels=cfg.hidden_size, llm_type=str(llm_dir))))
    inf = inf_mod.LLMSpeechTextInference(conf, str(ckpt), torch.device(DEV), dtype=torch.float32)
    tok = inf.llm_tokenizer
    assert tok.pad_token == tok.eos_token and tok.padding_side == "left"
    # the hub's generation_config.json is read: its sampling parameters are kept, its do_sample flag is recorded but NOT acted on
    # (BASELINE's north_star is greedy decode; generate(do_sample=True) opts in) — the greedy ids below prove it
    gcfg = inf.llm.generation_config
    assert gcfg.hub_do_sample is True and gcfg.do_sample is False and abs(gcfg.temperature - 0.6) < 1e-9 and abs(gcfg.top_p - 0.9) < 1e-9 and gcfg.top_k == 50
    # the four template strings tokenise to the recorded ids (f1), through the object the inference class built itself
    for key in ("prefix", "suffix", "text_prompt", "additional_text_prompt"):
        assert tok(rec["strings"][key], return_tensors="pt").input_ids[0].tolist() == rec["ids"][key], key
    if family == "llama3":
        assert len(rec["ids"]["prefix"]) == 9 and len(rec["ids"]["suffix"]) == 6      # BOS included: 9 + P + 5 prompt rows
    prefix_ids, suffix_ids = t(rec["ids"]["prefix"])[None], t(rec["ids"]["suffix"])[None]
    extra_ids = t(rec["ids"]["additional_text_prompt"])[None]
    wave = ri.synthetic_waveform(24000, seed=5)
    ref_audio = ho.audio_encoder_forward(enc_sd, hc, wave[None])
    for extra_text, extra in (("", None), (rec["strings"]["additional_text_prompt"], extra_ids)):
        text = inf.generate_audio_response(wave.numpy(), additional_text_prompt=extra_text, max_new_tokens=24)
        ref = ko.generate_audio_response_ids(sd, cfg, ref_audio, prefix_ids, suffix_ids, extra, max_new_tokens=24)
        assert torch.equal(inf.last_generate_ids.cpu(), ref)
        assert text == tok.batch_decode(ref, skip_special_tokens=True, clean_up_tokenization_spaces=True)[0]
    # generate_text_response: f"{prefix} {input_text}{suffix} " (ref:inference.py:78, spaces kept)
    text = inf.generate_text_response("hello world", max_new_tokens=16)
    prompt_ids = t(rec["ids"]["text_prompt"])[None]
    ref = lo.greedy_generate(sd, cfg, sd["model.embed_tokens.weight"][prompt_ids], 16, use_eos=True)
    assert torch.equal(inf.last_generate_ids.cpu(), ref)
    assert text == tok.batch_decode(ref, skip_special_tokens=True, clean_up_tokenization_spaces=True)[0]
    # f2: the safetensors loader on its own — logits of a forward pass equal the oracle on the same state dict
    llm2 = llama_mod.AudioLlamaForCausalLM.from_pretrained(str(llm_dir), use_cache=True, torch_dtype=torch.float32).eval().to(DEV)
    x = torch.randn(1, 11, cfg.hidden_size, generator=torch.Generator().manual_seed(3)) * 0.05
    assert rel_err(llm2(inputs_embeds=x.to(DEV)).logits.cpu(), lo.llama_forward(sd, cfg, x)["logits"]) < F32_TOL


def test_utils_soft_cross_entropy_drop_in():
    """ref:utils.py:167-178 through the package's `utils` mirror (and the root `utils` import path)."""
    import utils as root_utils
    gen = torch.Generator().manual_seed(9)
    s, tch = torch.randn(1, 7, 1000, generator=gen) * 3, torch.randn(1, 7, 1000, generator=gen) * 3
    ref = ko.soft_cross_entropy(s, tch)
    out = root_utils.soft_cross_entropy(s.to(DEV), tch.to(DEV))
    assert out.dim() == 0 and abs(float(out) - float(ref)) < 1e-5 * abs(float(ref))
    per = utils.soft_cross_entropy(s.to(DEV), tch.to(DEV), reduction="none")
    assert per.shape == (1, 7) and abs(float(per.mean()) - float(ref)) < 1e-5 * abs(float(ref))


@pytest.mark.parametrize("B,check_every", [(300, 4), (37, 2)])
def test_llama_compacted_batch_mixed_stop_lengths_equals_single(B, check_every):
    """sl_generate with per-sequence token budgets and EOS (answers of different lengths) and compact = 1: as rows finish the batch steps
    down the ladder of row counts (300 -> 256 -> 192 -> ... ; live rows above the rung move into finished rows' places together with their
    K / V slots, output ids and counters) and the decode step runs on the live rows only.  Per sequence the ids are EXACTLY those of the
    uncompacted batch and of the sequence alone (fp32: every kernel family is bit-exact against the oracle), pads after the stop,
    hf:generation/utils.py:2928-2942.  Sampling: the draw of a sequence follows its index in the call, not the row it sits in."""
    cfg = TINY_LLAMA
    llm, sd = make_llama(cfg, 31, torch.float32)
    gen = torch.Generator().manual_seed(6)
    base = [torch.randn(n, cfg.hidden_size, generator=gen) * 0.05 for n in (9, 21, 14, 5, 30)]
    llm.generation_config.eos_token_id = list(cfg.eos_token_ids)
    new = 48
    refs = [lo.greedy_generate(sd, cfg, p[None], new, use_eos=True)[0] for p in base]
    limits = [3 + (11 * b) % 44 for b in range(B)]
    limits[0], limits[1] = 1, new                    # a budget spent by the prefill's own token; a budget of the whole max_new_tokens
    prompts = [base[b % 5] for b in range(B)]
    lens = [int(p.shape[0]) for p in prompts]
    x = torch.cat(prompts).to(DEV, torch.float32)
    ids_c, n_c = llm.generate_packed(x.clone(), lens, new, use_eos=True, row_limits=limits, compact=True, check_every=check_every)
    st = dict(llm.last_generate_stats)
    ids_u, n_u = llm.generate_packed(x.clone(), lens, new, use_eos=True, row_limits=limits, compact=False, check_every=check_every)
    su = dict(llm.last_generate_stats)
    assert n_c == n_u and torch.equal(ids_c[:, :n_c], ids_u[:, :n_u])
    stops = []
    for b in range(B):
        r = refs[b % 5]
        stop = min(int(r.shape[0]), limits[b])          # EOS (the oracle's row ends with it) or the budget, whichever comes first
        stops.append(stop)
        assert torch.equal(ids_c[b, :stop].long(), r[:stop]), b
        assert bool((ids_c[b, stop:n_c] == cfg.pad_token_id).all()), b
    assert n_c == max(stops)
    assert st["compactions"] >= 3 and st["final_rows"] < B and st["row_steps"] < su["row_steps"] and su["compactions"] == 0
    assert su["row_steps"] == B * su["decode_launches"]
    # sampled: the compacted run draws what the uncompacted one draws
    smp = dict(temperature=0.9, top_k=40, top_p=0.95, seed=123)
    a, na = llm.generate_packed(x.clone(), lens, new, use_eos=True, row_limits=limits, compact=True, check_every=check_every, sample=smp)
    assert llm.last_generate_stats["compactions"] >= 3
    b_, nb = llm.generate_packed(x.clone(), lens, new, use_eos=True, row_limits=limits, compact=False, check_every=check_every, sample=smp)
    assert na == nb and torch.equal(a[:, :na], b_[:, :nb])
    assert not torch.equal(a[:, :na], ids_c[:, :na])
    # budgets without any EOS id: rows stop at their budget only; a budget outside [1, max_new_tokens] is refused
    llm.generation_config.eos_token_id = None
    ids_n, n_n = llm.generate_packed(x.clone(), lens, new, use_eos=False, row_limits=limits, check_every=check_every)
    assert n_n == max(limits)
    noeos = [lo.greedy_generate(sd, cfg, p[None], new, use_eos=False)[0] for p in base]
    for b in range(B):
        assert torch.equal(ids_n[b, :limits[b]].long(), noeos[b % 5][:limits[b]]), b
        assert bool((ids_n[b, limits[b]:n_n] == (cfg.pad_token_id if cfg.pad_token_id is not None else 0)).all()), b
    with pytest.raises(pkg("_lib").SpeechLLMError):
        llm.generate_packed(x.clone(), lens, new, row_limits=[new + 1] * B)


def test_llama_compaction_keeps_the_shared_prompt_prefix_promise():
    """Compaction moves K / V slots; the single-pass decode attention reads the shared prompt-prefix positions from slot 0.  Every slot holds
    bit-identical prefix rows (prefill's promise), so whichever sequence ends up in slot 0 the ids must equal those of shared_prefix = 0 and of
    the uncompacted batch — fp32, 40 sequences with a common 11-row prefix, budgets that empty slot 0 early."""
    cfg = TINY_LLAMA
    llm, _ = make_llama(cfg, 35, torch.float32)
    gen = torch.Generator().manual_seed(12)
    P, B, new = 11, 40, 40
    pre = torch.randn(P, cfg.hidden_size, generator=gen) * 0.05
    tails = [torch.randn(n, cfg.hidden_size, generator=gen) * 0.05 for n in (9, 40, 14, 5, 27, 30, 21, 1)]
    prompts = [torch.cat([pre, tails[b % len(tails)] * (1.0 + 0.01 * (b // len(tails)))]) for b in range(B)]
    lens = [int(p.shape[0]) for p in prompts]
    x = torch.cat(prompts).to(DEV, torch.float32)
    limits = [2 + (7 * b) % 37 for b in range(B)]
    limits[0] = 2                                     # slot 0's own sequence leaves first: a mover takes the slot
    llm.generation_config.eos_token_id = None
    ref, n_ref = llm.generate_packed(x.clone(), lens, new, use_eos=False, row_limits=limits, compact=False, shared_prefix=0)
    for sp in (0, P):
        ids, n = llm.generate_packed(x.clone(), lens, new, use_eos=False, row_limits=limits, compact=True, check_every=2, shared_prefix=sp)
        assert llm.last_generate_stats["compactions"] >= 3
        assert n == n_ref and torch.equal(ids[:, :n], ref[:, :n_ref]), sp


def test_sampled_generation_is_reproducible_and_greedy_stays_default():
    """generate(do_sample=True, ...) runs sl_sample_generate (one captured decode graph with the sampling kernels); same seed
    -> same ids, another seed -> other ids, top_k=1 == greedy; without do_sample the reference fixture's greedy ids come out."""
    g = golden("llama_tiny_gqa")
    cfg = TINY_LLAMA
    llm, _ = make_llama(cfg, int(g["weight_seed"]), torch.float32)
    gen = torch.Generator().manual_seed(int(g["embeds_seed"]))
    x = (torch.randn(1, int(g["S"]), cfg.hidden_size, generator=gen) * 0.05).to(DEV)
    llm.generation_config.eos_token_id = None
    greedy = llm.generate(inputs_embeds=x, max_new_tokens=32).cpu()
    assert torch.equal(greedy, t(g["ids_noeos"]))
    a = llm.generate(inputs_embeds=x, max_new_tokens=32, do_sample=True, temperature=0.9, top_k=50, top_p=0.95, seed=11).cpu()
    b = llm.generate(inputs_embeds=x, max_new_tokens=32, do_sample=True, temperature=0.9, top_k=50, top_p=0.95, seed=11).cpu()
    c = llm.generate(inputs_embeds=x, max_new_tokens=32, do_sample=True, temperature=0.9, top_k=50, top_p=0.95, seed=12).cpu()
    assert torch.equal(a, b) and not torch.equal(a, c) and not torch.equal(a, greedy)
    k1 = llm.generate(inputs_embeds=x, max_new_tokens=32, do_sample=True, temperature=0.7, top_k=1, top_p=1.0, seed=5).cpu()
    assert torch.equal(k1, greedy)
    assert torch.equal(llm.generate(inputs_embeds=x, max_new_tokens=32).cpu(), greedy)      # the cached greedy graph is still the greedy one
