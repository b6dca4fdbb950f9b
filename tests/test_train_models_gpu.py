"""GPU: the KD step (ref:trainer.py:270-374) on the HIP path against the reference-generated fixture
`pipeline_tiny.npz` (losses, gradient norms, d ntp / d audio_embeds) and against autograd through the CPU oracle."""
import pytest
import torch

from conftest import golden, pkg, rel_err, t
from oracle import hubert_oracle as ho
from oracle import kd_oracle as ko
from oracle import llama_oracle as lo
from oracle.golden_cfgs import TINY_HUBERT, TINY_LLAMA
from test_models_gpu import DEV, make_encoder, make_llama

pytestmark = pytest.mark.gpu

ri = pkg("random_init")
cfgm = pkg("config")
training = pkg("training")
ops = pkg("ops")
utils = pkg("utils")


def kd_config(ntp=0.5, ld=0.5, fd=1.0, use_ld=True, use_fd=True, taps=(0, 1, 3), accum=16):
    return cfgm.from_dict(dict(train=dict(optimizer=dict(lr=5e-5, beta1=0.9, beta2=0.999), grad_accum_interval=accum, use_ld_loss=use_ld,
                                          use_fd_loss=use_fd, ntp_loss_weight=ntp, ld_loss_weight=ld, fd_loss_weight=fd,
                                          fd_loss_connector_layers=list(taps))))


def build(g, dtype, **kw):
    enc, enc_sd = make_encoder(TINY_HUBERT, TINY_LLAMA.hidden_size, int(g["enc_seed"]), dtype)
    llm, llm_sd = make_llama(TINY_LLAMA, int(g["llm_seed"]), dtype)
    tr = training.KDTrainer(kd_config(**kw), enc, llm, t(g["prefix_ids"]), t(g["suffix_ids"]))
    wave = ri.synthetic_waveform(int(g["n_samples"]), seed=int(g["wave_seed"]))
    return tr, enc_sd, llm_sd, wave


def test_llm_side_ntp_gradient_wrt_audio_embeddings_fp32():
    g = golden("pipeline_tiny")
    tr, _, _, wave = build(g, torch.float32, ntp=1.0, use_ld=False, use_fd=False, accum=1)
    tr.optimizer_step = lambda: None   # inspect the raw gradient, do not update
    losses = tr.micro_step(wave, t(g["text_ids"]), t(g["response_ids"]))
    assert abs(losses["ntp_loss"] - float(g["ntp"])) < 1e-4
    assert rel_err(tr.last_d_audio.cpu(), t(g["d_ntp_d_audio_embeds"])[0]) < 1e-4


def test_kd_micro_step_losses_and_encoder_gradients_fp32_vs_reference_fixture():
    g = golden("pipeline_tiny")
    tr, enc_sd, _, wave = build(g, torch.float32, taps=tuple(g["connector_layers"].tolist()))
    losses = tr.micro_step(wave, t(g["text_ids"]), t(g["response_ids"]))
    for k, ref in (("ntp_loss", "ntp"), ("ld_loss", "ld"), ("fd_loss", "fd"), ("total", "total")):
        assert abs(losses[k] - float(g[ref])) < 1e-4 * max(1.0, abs(float(g[ref]))), k
    grads = training.kernel_grads_to_state_dict(tr.enc, tr.grads, tr.master)
    n = lambda k: float(grads[k].norm())
    assert abs(n("embed_projection.weight") - float(g["grad_norm_embed_projection_weight"])) < 2e-3 * float(g["grad_norm_embed_projection_weight"])
    assert abs(n("encoder.encoder.layers.0.attention.q_proj.weight") - float(g["grad_norm_layer0_q"])) < 2e-3 * float(g["grad_norm_layer0_q"])
    assert abs(n("encoder.feature_extractor.conv_layers.0.conv.weight") - float(g["grad_norm_conv0"])) < 2e-3 * float(g["grad_norm_conv0"])
    total = torch.stack([grads[k].norm() for k in tr.trainable]).norm()
    assert abs(float(total) - float(g["grad_norm_total"])) < 2e-3 * float(g["grad_norm_total"])
    assert len(tr.trainable) == int(g["n_params_with_grad"])


def test_every_encoder_parameter_gradient_matches_oracle_autograd_fp32():
    g = golden("pipeline_tiny")
    tr, enc_sd, llm_sd, wave = build(g, torch.float32, taps=(0, 1, 3))
    tr.micro_step(wave, t(g["text_ids"]), t(g["response_ids"]))
    grads = training.kernel_grads_to_state_dict(tr.enc, tr.grads, tr.master)
    # CPU oracle with autograd over the same state dict
    sd = {k: v.clone().requires_grad_(k != "encoder.masked_spec_embed") for k, v in enc_sd.items()}
    audio = ho.audio_encoder_forward(sd, TINY_HUBERT, wave[None])
    losses = ko.kd_losses(llm_sd, TINY_LLAMA, audio, t(g["text_ids"]), t(g["response_ids"]), t(g["prefix_ids"]), t(g["suffix_ids"]),
                          connector_layers=(0, 1, 3))
    (losses["total"] / 16).backward()
    total = torch.stack([sd[k].grad.norm() for k in tr.trainable]).norm()
    worst = 0.0
    for k in tr.trainable:
        ref = sd[k].grad
        err = float((grads[k].cpu().reshape(ref.shape).double() - ref.double()).norm())
        # k_proj.bias has an exactly-zero true gradient (softmax is shift invariant): judge it on the absolute scale
        bound = 2e-3 * float(ref.norm()) + 1e-6 * float(total)
        worst = max(worst, err / (float(ref.norm()) + 1e-6 * float(total)))
        assert err < bound, (k, err, float(ref.norm()))
    print("worst relative gradient error", worst)


def test_training_mode_regularizers_match_oracle_with_replayed_masks_fp32():
    """Dropout (feature projection / hidden / activation), LayerDrop and SpecAugment of the reference's train() mode
    (ref:trainer.py:258): the oracle replays the kernel path's masks (counter-based hash, host restatement) and its
    SpecAugment rows; losses and every parameter gradient — masked_spec_embed included — must agree."""
    import numpy as np
    g = golden("pipeline_tiny")
    reg = training.TrainRegularizers(feat_proj_dropout=0.1, hidden_dropout=0.1, activation_dropout=0.1, attention_dropout=0.1, layerdrop=0.34, apply_spec_augment=True,
                                     mask_time_prob=0.3, mask_time_length=3, mask_time_min_masks=2, seed=77)
    enc, enc_sd = make_encoder(TINY_HUBERT, TINY_LLAMA.hidden_size, int(g["enc_seed"]), torch.float32)
    enc_sd = dict(enc_sd)
    llm, llm_sd = make_llama(TINY_LLAMA, int(g["llm_seed"]), torch.float32)
    tr = training.KDTrainer(kd_config(taps=(0, 1, 3)), enc, llm, t(g["prefix_ids"]), t(g["suffix_ids"]), regularizers=reg)
    assert "encoder.masked_spec_embed" in tr.trainable
    wave = ri.synthetic_waveform(int(g["n_samples"]), seed=int(g["wave_seed"]))
    T, H = TINY_HUBERT.num_frames(wave.numel()), TINY_HUBERT.hidden_size
    np.random.seed(11)
    state = np.random.get_state()
    spec = training.compute_mask_indices((1, T), reg.mask_time_prob, reg.mask_time_length, reg.mask_time_min_masks)
    np.random.set_state(state)                         # the tape draws the same spans from the global RNG
    losses = tr.micro_step(wave, t(g["text_ids"]), t(g["response_ids"]))
    grads = training.kernel_grads_to_state_dict(tr.enc, tr.grads, tr.master)
    base = (reg.seed * 1000003 + 0) & 0xFFFFFFFFFFFFFFFF
    probs = {"fp": reg.feat_proj_dropout, "pos": reg.hidden_dropout, "attn_out": reg.hidden_dropout, "ffn_out": reg.hidden_dropout,
             "act": reg.activation_dropout}

    def drop(site, layer, v):
        keep = ops.dropout_keep_mask(v.numel(), probs[site], training._site_seed(base, site, layer)).view(v.shape)
        return torch.where(keep, v / (1.0 - float(np.float32(probs[site]))), torch.zeros_like(v))

    nh = TINY_HUBERT.num_attention_heads

    def attn_drop(layer, probs):     # (1, nh, T, T): mask index ((token * nh + head) << 16) | key — one utterance, so token = query row
        n = (T * nh) << 16
        keep = ops.dropout_keep_mask(n, reg.attention_dropout, training._site_seed(base, "attn_prob", layer)).view(T, nh, 1 << 16)[:, :, :T]
        keep = keep.permute(1, 0, 2)[None]
        return torch.where(keep, probs / (1.0 - float(np.float32(reg.attention_dropout))), torch.zeros_like(probs))

    skip = {li for li in range(TINY_HUBERT.num_hidden_layers)
            if (training._site_seed(base, "layerdrop", li) >> 11) * (1.0 / 9007199254740992.0) < reg.layerdrop}
    assert 0 < len(skip) < TINY_HUBERT.num_hidden_layers, "pick a seed that drops some but not all layers of the tiny model"
    assert spec.any()
    sd = {k: v.clone().requires_grad_(True) for k, v in enc_sd.items()}
    audio = ho.audio_encoder_forward(sd, TINY_HUBERT, wave[None], train=dict(drop=drop, skip=skip, spec_mask=torch.from_numpy(spec), attn_drop=attn_drop))
    ref = ko.kd_losses(llm_sd, TINY_LLAMA, audio, t(g["text_ids"]), t(g["response_ids"]), t(g["prefix_ids"]), t(g["suffix_ids"]), connector_layers=(0, 1, 3))
    for k, r in (("ntp_loss", "ntp"), ("ld_loss", "ld"), ("fd_loss", "fd"), ("total", "total")):
        assert abs(losses[k] - float(ref[r])) < 1e-4 * max(1.0, abs(float(ref[r]))), k
    (ref["total"] / 16).backward()
    total = torch.stack([sd[k].grad.norm() for k in tr.trainable if sd[k].grad is not None]).norm()
    for k in tr.trainable:
        rg = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])
        err = float((grads[k].cpu().reshape(rg.shape).double() - rg.double()).norm())
        assert err < 2e-3 * float(rg.norm()) + 1e-6 * float(total), (k, err, float(rg.norm()))
    skipped = next(iter(skip))
    assert float(grads[f"encoder.encoder.layers.{skipped}.attention.out_proj.weight"].abs().max()) == 0.0
    assert float(grads["encoder.masked_spec_embed"].norm()) > 0.0


def test_whisper_kd_step_losses_and_every_gradient_match_oracle_autograd_fp32():
    """KD micro-steps through the Whisper encoder tape (log-mel on the GPU, conv1 / conv2 implicit GEMMs, shared layer stack,
    crop to compute_num_audio_embeds) against autograd through the CPU oracle — two utterances of different lengths packed
    in one window, so the crop and the per-utterance loss scaling are exercised."""
    from oracle import whisper_oracle as wo
    from oracle.golden_cfgs import TINY_WHISPER as WC
    weights = pkg("weights")
    enc_mod = pkg("audio_encoder")
    g = golden("pipeline_tiny")
    conf = cfgm.from_dict(dict(model=dict(audio_encoder=dict(base="whisper", type="synthetic", downsample_method="pool", downsample_factor=4,
                                                             pooling=dict(kernel_size=8, stride=4)), llm_embedding_channels=TINY_LLAMA.hidden_size)))
    arch = weights.WhisperArch(WC.d_model, WC.encoder_layers, WC.encoder_attention_heads, WC.encoder_ffn_dim, WC.num_mel_bins, WC.max_source_positions)
    enc = enc_mod.AudioEncoder(conf, DEV, dtype=torch.float32, arch=arch)
    enc_sd = ri.whisper_encoder_state_dict(WC, TINY_LLAMA.hidden_size, seed=3)
    enc.load_state_dict(enc_sd).eval().to(DEV)
    llm, llm_sd = make_llama(TINY_LLAMA, int(g["llm_seed"]), torch.float32)
    tr = training.KDTrainer(kd_config(taps=(0, 1, 3)), enc, llm, t(g["prefix_ids"]), t(g["suffix_ids"]))
    assert "encoder.embed_positions.weight" not in tr.trainable and "encoder.conv1.weight" in tr.trainable
    waves = [ri.synthetic_waveform(24000, seed=5), ri.synthetic_waveform(17000, seed=6)]        # 1.5 s and ~1.06 s inside the 2 s window
    text_ids, resp_ids = [t(g["text_ids"]), t(g["text_ids"])[:7]], [t(g["response_ids"]), t(g["response_ids"])[:9]]
    losses = tr.micro_batch(waves, text_ids, resp_ids)
    grads = tr.to_state_dict(tr.enc, tr.grads, tr.master)
    sd = {k: v.clone().requires_grad_(k != "encoder.embed_positions.weight") for k, v in enc_sd.items()}
    total_loss = 0.0
    for u, wave in enumerate(waves):
        feats = wo.log_mel(WC, wave)[None]
        n_emb = utils.compute_num_audio_embeds(wave.numel())
        audio = wo.audio_encoder_forward(sd, WC, feats)[:, :n_emb]
        ref = ko.kd_losses(llm_sd, TINY_LLAMA, audio, text_ids[u], resp_ids[u], t(g["prefix_ids"]), t(g["suffix_ids"]), connector_layers=(0, 1, 3))
        for k, r in (("ntp_loss", "ntp"), ("ld_loss", "ld"), ("fd_loss", "fd"), ("total", "total")):
            assert abs(losses[u][k] - float(ref[r])) < 2e-4 * max(1.0, abs(float(ref[r]))), (u, k, losses[u][k], float(ref[r]))
        total_loss = total_loss + ref["total"] / 16
    total_loss.backward()
    total = torch.stack([sd[k].grad.norm() for k in tr.trainable]).norm()
    for k in tr.trainable:
        rg = sd[k].grad
        err = float((grads[k].cpu().reshape(rg.shape).double() - rg.double()).norm())
        assert err < 3e-3 * float(rg.norm()) + 1e-6 * float(total), (k, err, float(rg.norm()))
    tr.optimizer_step()                                   # AdamW over the Whisper parameter set, kernel weights refreshed
    assert not torch.equal(tr.master["encoder.conv1.weight"].cpu(), enc_sd["encoder.conv1.weight"])
    assert torch.equal(tr.master["encoder.embed_positions.weight"].cpu(), enc_sd["encoder.embed_positions.weight"])


def test_optimizer_step_updates_master_and_kernel_weights():
    g = golden("pipeline_tiny")
    tr, enc_sd, _, wave = build(g, torch.float32, accum=2)
    before = {k: v.clone() for k, v in tr.master.items()}
    out0 = tr.enc(wave[None].to(DEV)).clone()
    tr.micro_step(wave, t(g["text_ids"]), t(g["response_ids"]))
    assert all(torch.equal(before[k], tr.master[k]) for k in before)           # accumulating, no update yet
    tr.micro_step(wave, t(g["text_ids"]), t(g["response_ids"]))
    changed = [k for k in tr.trainable if not torch.equal(before[k], tr.master[k])]
    assert len(changed) == len(tr.trainable)
    assert torch.equal(before["encoder.masked_spec_embed"], tr.master["encoder.masked_spec_embed"])
    assert not torch.equal(out0, tr.enc(wave[None].to(DEV)))                    # kernel weights were refreshed
    assert all(float(v.abs().max()) == 0 for v in tr.grads.values())            # accumulators cleared
    # AdamW first step moves every weight by ~lr: |delta| <= lr * (1 + wd*|w|) + eps
    k = "embed_projection.weight"
    assert float((before[k] - tr.master[k]).abs().max()) <= 5e-5 * (1 + 0.01 * float(before[k].abs().max())) + 1e-7


@pytest.mark.parametrize("base_kind,dtype", [("hubert", torch.float32), ("hubert", torch.bfloat16), ("whisper", torch.bfloat16)])
def test_fused_optimizer_step_equals_torch_adamw_plus_full_refresh(base_kind, dtype):
    """KDTrainer.optimizer_step with the one-launch AdamW + in-pass weight refresh (default) against the general path
    (torch.optim.AdamW's foreach step + re-deriving every device tensor from the masters): same masters, same optimizer state,
    and device weights BIT-identical to a full refresh — over three optimizer steps, HuBERT and Whisper parameter sets."""
    g = golden("pipeline_tiny")

    def make(fused):
        if base_kind == "hubert":
            tr, _, _, wave = build(g, dtype, accum=1)
            waves = [wave]
        else:
            from oracle.golden_cfgs import TINY_WHISPER as WC
            weights, enc_mod = pkg("weights"), pkg("audio_encoder")
            conf = cfgm.from_dict(dict(model=dict(audio_encoder=dict(base="whisper", type="synthetic", downsample_method="pool", downsample_factor=4,
                                                                     pooling=dict(kernel_size=8, stride=4)), llm_embedding_channels=TINY_LLAMA.hidden_size)))
            arch = weights.WhisperArch(WC.d_model, WC.encoder_layers, WC.encoder_attention_heads, WC.encoder_ffn_dim, WC.num_mel_bins, WC.max_source_positions)
            enc = enc_mod.AudioEncoder(conf, DEV, dtype=dtype, arch=arch)
            enc.load_state_dict(ri.whisper_encoder_state_dict(WC, TINY_LLAMA.hidden_size, seed=3)).eval().to(DEV)
            llm, _ = make_llama(TINY_LLAMA, int(g["llm_seed"]), dtype)
            tr = training.KDTrainer(kd_config(taps=(0, 1, 3), accum=1), enc, llm, t(g["prefix_ids"]), t(g["suffix_ids"]))
            waves = [ri.synthetic_waveform(24000, seed=5)]
        tr.use_fused_adamw = fused
        return tr, waves

    (a, waves), (b, _) = make(True), make(False)
    for step in range(3):
        for tr in (a, b):
            tr.micro_batch(waves, [t(g["text_ids"])], [t(g["response_ids"])])
        assert a._fused is not None and b._fused is None and a.optimizer_steps == b.optimizer_steps == step + 1
        for k in a.param_names:
            if k.endswith("k_proj.bias"):
                # exactly-zero true gradient (softmax shift invariance): AdamW normalises the summation noise to +-lr per step,
                # and the noise (fp32 atomics) is not reproducible run to run — bounded drift instead of agreement
                assert float((a.master[k] - b.master[k]).abs().max()) <= 2 * (step + 1) * 5e-5 * 1.001, (step, k)
                continue
            assert rel_err(a.master[k].cpu(), b.master[k].cpu()) < (2e-6 if dtype == torch.float32 else 1e-4), (step, k)
        # device weights after the fused step == what a full refresh derives from the same masters, bit for bit
        snap = [w.clone() for w in a.enc.weights._keep]
        a.enc.weights.refresh(a.master)
        for i, (x, y) in enumerate(zip(snap, a.enc.weights._keep)):
            assert torch.equal(x, y), (step, i, tuple(x.shape))
    sa, sb = a.optimizer_state_dict(), b.optimizer_state_dict()
    assert sa["param_groups"][0]["params"] == sb["param_groups"][0]["params"] and set(sa["state"]) == set(sb["state"])
    b.load_optimizer_state_dict(sa)                      # the state the fused step keeps is torch's own layout


def test_merged_teacher_student_pass_equals_two_passes_fp32():
    """The window's teacher (text) and student (audio) sequences through ONE ragged LLM forward, backward over the student rows
    only (default) == the reference's two calls (ref:trainer.py:299-323): same losses, same encoder gradients."""
    g = golden("pipeline_tiny")
    outs = []
    for merged in (True, False):
        tr, _, _, wave = build(g, torch.float32, taps=(0, 1, 3), accum=4)
        tr.merge_teacher_pass = merged
        tr.optimizer_step = lambda: None
        waves = [wave, wave[:20000], wave[3000:30000]]
        texts = [t(g["text_ids"]), t(g["text_ids"])[:5], t(g["text_ids"])[2:]]
        resps = [t(g["response_ids"]), t(g["response_ids"])[:4], t(g["response_ids"])[1:]]
        losses = tr.micro_batch(waves, texts, resps)
        grads = {k: v.clone() for k, v in training.kernel_grads_to_state_dict(tr.enc, tr.grads, tr.master).items()}
        outs.append((losses, grads, tr.trainable))
    (la, ga, names), (lb, gb, _) = outs
    for u in range(3):
        for k in la[u]:
            assert abs(la[u][k] - lb[u][k]) < 1e-5 * max(1.0, abs(lb[u][k])), (u, k)
    total = float(torch.stack([gb[k].norm() for k in names]).norm())
    for k in names:
        assert float((ga[k] - gb[k]).norm()) < 2e-5 * float(gb[k].norm()) + 1e-6 * total, k


def test_kd_micro_step_bf16_runs_and_tracks_fp32():
    g = golden("pipeline_tiny")
    tr, _, _, wave = build(g, torch.bfloat16)
    losses = tr.micro_step(wave, t(g["text_ids"]), t(g["response_ids"]))
    for k, ref in (("ntp_loss", "ntp"), ("ld_loss", "ld")):
        assert abs(losses[k] - float(g[ref])) < 3e-2 * abs(float(g[ref])), (k, losses[k], float(g[ref]))
    grads = training.kernel_grads_to_state_dict(tr.enc, tr.grads, tr.master)
    gn = float(grads["embed_projection.weight"].norm())
    assert abs(gn - float(g["grad_norm_embed_projection_weight"])) < 0.15 * float(g["grad_norm_embed_projection_weight"])


def test_micro_batch_equals_sequential_micro_steps_fp32():
    """A packed ragged batch of micro-steps accumulates the same gradients as the reference's sequential loop."""
    g = golden("pipeline_tiny")
    gen = torch.Generator().manual_seed(77)
    waves = [ri.synthetic_waveform(n, seed=n) for n in (32000, 20000, 26000)]
    texts = [torch.randint(1, TINY_LLAMA.vocab_size, (k,), generator=gen) for k in (11, 5, 8)]
    resps = [torch.randint(1, TINY_LLAMA.vocab_size, (k,), generator=gen) for k in (8, 6, 9)]
    tr1, _, _, _ = build(g, torch.float32, accum=4)
    seq_losses = [tr1.micro_step(w, t_, r) for w, t_, r in zip(waves, texts, resps)]
    tr2, _, _, _ = build(g, torch.float32, accum=4)
    bat_losses = tr2.micro_batch(waves, texts, resps)
    for a, b in zip(seq_losses, bat_losses):
        for k in a:
            assert abs(a[k] - b[k]) < 1e-4 * max(1.0, abs(a[k])), (k, a[k], b[k])
    tot = torch.stack([v.norm() for v in tr1.grads.values()]).norm()
    for k in tr1.grads:
        err = float((tr1.grads[k].double() - tr2.grads[k].double()).norm())
        assert err < 1e-4 * float(tr1.grads[k].norm()) + 1e-6 * float(tot), (k, err)


def test_trainer_loop_validate_checkpoint_roundtrip(tmp_path):
    """ref:trainer.py surface on a synthetic dataset with the reference's row schema: two optimizer steps, validation
    (perplexities + generation), checkpoint in the reference's format, resume."""
    from types import SimpleNamespace
    from test_models_gpu import StubTokenizer
    trainer_mod = pkg("trainer")
    g = golden("pipeline_tiny")
    gen = torch.Generator().manual_seed(5)
    V = TINY_LLAMA.vocab_size

    def row(n, nt, nr):
        return {"audio": {"array": ri.synthetic_waveform(n, seed=n)}, "text": f"utt{n}",
                "text_input_ids": torch.cat([torch.zeros(1, dtype=torch.long), torch.randint(1, V, (nt,), generator=gen)]),
                "response_input_ids": torch.cat([torch.zeros(1, dtype=torch.long), torch.randint(1, V, (nr,), generator=gen)])[None],
                "pool_ranges_4": []}

    train_ds = [row(16000 + 1000 * i, 5 + i % 3, 6 + i % 4) for i in range(8)]
    val_ds = [row(20000, 6, 7), row(24000, 4, 5)]
    conf = cfgm.from_dict(dict(seed_everything=1234, audio=dict(sampling_rate=16000),
                               model=dict(audio_encoder=dict(base="hubert", type="synthetic", downsample_method="pool", downsample_factor=4,
                                                             pooling=dict(kernel_size=8, stride=4)),
                                          llm_embedding_channels=TINY_LLAMA.hidden_size, llm_type=utils.LLAMA_ID),
                               train=dict(optimizer=dict(lr=5e-5, beta1=0.9, beta2=0.999), batch_size=1, grad_accum_interval=4, epochs=1,
                                          use_ld_loss=True, use_fd_loss=True, ntp_loss_weight=0.5, ld_loss_weight=0.5, fd_loss_weight=1.0,
                                          fd_loss_connector_layers=[0, 1, 3]),
                               log=dict(checkpoint_dir=str(tmp_path / "ckpt"), log_dir=str(tmp_path / "logs"), log_interval=4,
                                        validation_interval=1000, num_generate_samples=1)))
    enc, _ = make_encoder(TINY_HUBERT, TINY_LLAMA.hidden_size, 51, torch.float32)
    llm, _ = make_llama(TINY_LLAMA, 52, torch.float32)
    tok = StubTokenizer({utils.LLAMA_PROMPT_PREFIX: t(g["prefix_ids"]), utils.LLAMA_PROMPT_SUFFIX: t(g["suffix_ids"])})
    args = SimpleNamespace(run_name="t", checkpoint_path=None, gpu_idx=0)
    tr = trainer_mod.Trainer(args, conf, DEV, tokenizer=tok, llm=llm, audio_encoder=enc, train_dataset=train_ds, val_dataset=val_ds,
                             dtype=torch.float32)
    w0 = tr.kd.master["embed_projection.weight"].clone()
    tr.train()
    assert tr.step == 8 and not torch.equal(w0, tr.kd.master["embed_projection.weight"])
    ck_path = tmp_path / "ckpt" / "t" / "epoch_0_step_8.pt"
    ck = torch.load(ck_path, map_location="cpu", weights_only=False)
    assert set(ck) >= {"audio_encoder", "optimizer", "lr_scheduler", "epoch", "step"} and ck["step"] == 8
    # the optimizer state is written in the reference's layout (ref:trainer.py:98-105): group 0 = every encoder parameter in
    # audio_encoder.parameters() order, group 1 = the frozen LLM's parameters (no state)
    groups = ck["optimizer"]["param_groups"]
    n_enc = len(ck["audio_encoder"])
    assert len(groups) == 2 and groups[0]["params"] == list(range(n_enc))
    assert groups[1]["params"] == list(range(n_enc, n_enc + 2 + 9 * TINY_LLAMA.num_hidden_layers))
    names = pkg("weights").reference_param_order(ck["audio_encoder"].keys())
    i_proj = names.index("embed_projection.weight")
    assert ck["optimizer"]["state"][i_proj]["exp_avg"].shape == ck["audio_encoder"]["embed_projection.weight"].shape
    assert all(i < n_enc for i in ck["optimizer"]["state"])
    assert names.index("encoder.masked_spec_embed") not in ck["optimizer"]["state"] or tr.kd.reg is not None
    assert torch.equal(ck["audio_encoder"]["embed_projection.weight"], tr.kd.master["embed_projection.weight"].cpu())
    lines = [l for l in open(tmp_path / "logs" / "t" / "metrics.jsonl")]
    assert any("validation/audio_perplexity" in l for l in lines) and any("train/ntp_loss" in l for l in lines)
    # the flat encoder state-dict inside the checkpoint is what ref:inference.py:24-26 loads
    enc2, _ = make_encoder(TINY_HUBERT, TINY_LLAMA.hidden_size, 1, torch.float32)
    enc2.load_state_dict(ck["audio_encoder"])
    wave = ri.synthetic_waveform(16000, seed=3)[None].to(DEV)
    assert rel_err(enc2(wave).cpu(), tr.audio_encoder(wave).cpu()) < 1e-6   # weight-norm folded on the CPU here, on the GPU there
    # resume
    args2 = SimpleNamespace(run_name="t2", checkpoint_path=str(ck_path), gpu_idx=0)
    enc3, _ = make_encoder(TINY_HUBERT, TINY_LLAMA.hidden_size, 51, torch.float32)
    tr2 = trainer_mod.Trainer(args2, conf, DEV, tokenizer=tok, llm=llm, audio_encoder=enc3, train_dataset=train_ds, val_dataset=val_ds,
                              dtype=torch.float32)
    assert tr2.step == 8 and tr2.start_epoch == 0 and torch.equal(tr2.kd.master["embed_projection.weight"], tr.kd.master["embed_projection.weight"])
    assert tr2.kd.micro_batches == tr.kd.micro_batches
    st, st2 = tr.optimizer.state[tr.kd._param["embed_projection.weight"]], tr2.optimizer.state[tr2.kd._param["embed_projection.weight"]]
    assert torch.equal(st["exp_avg_sq"], st2["exp_avg_sq"]) and float(st["step"]) == float(st2["step"]) == 2
    # a checkpoint under the OTHER weight-norm spelling (torch < 2.1 wrote weight_g / weight_v: the released checkpoint) resumes too
    enc4, _ = make_encoder(TINY_HUBERT, TINY_LLAMA.hidden_size, 51, torch.float32, weight_norm_keys="legacy")
    tr3 = trainer_mod.Trainer(SimpleNamespace(run_name="t3", checkpoint_path=str(ck_path), gpu_idx=0), conf, DEV, tokenizer=tok, llm=llm,
                              audio_encoder=enc4, train_dataset=train_ds, val_dataset=val_ds, dtype=torch.float32)
    g_key = "encoder.encoder.pos_conv_embed.conv.weight_g"
    assert g_key in tr3.kd.master and torch.equal(tr3.kd.master[g_key].cpu(), ck["audio_encoder"]["encoder.encoder.pos_conv_embed.conv.parametrizations.weight.original0"])
    assert rel_err(tr3.audio_encoder(wave).cpu(), tr.audio_encoder(wave).cpu()) < 1e-6


def test_trainer_loop_whisper_base(tmp_path):
    """The same Trainer surface with `base: whisper` (ref:config/llama3_whisper.yaml): collate through the GPU log-mel,
    KD steps through the Whisper tape, validation on the uncropped window as the reference does, checkpoint."""
    from types import SimpleNamespace
    from oracle.golden_cfgs import TINY_WHISPER as WC
    from test_models_gpu import StubTokenizer
    trainer_mod, weights, enc_mod = pkg("trainer"), pkg("weights"), pkg("audio_encoder")
    g = golden("pipeline_tiny")
    gen = torch.Generator().manual_seed(6)
    V = TINY_LLAMA.vocab_size

    def row(n, nt, nr):
        return {"audio": {"array": ri.synthetic_waveform(n, seed=n)}, "text": f"utt{n}",
                "text_input_ids": torch.cat([torch.zeros(1, dtype=torch.long), torch.randint(1, V, (nt,), generator=gen)]),
                "response_input_ids": torch.cat([torch.zeros(1, dtype=torch.long), torch.randint(1, V, (nr,), generator=gen)])[None],
                "pool_ranges_4": []}

    train_ds = [row(14000 + 2000 * i, 5 + i % 3, 6 + i % 4) for i in range(4)]
    val_ds = [row(20000, 6, 7)]
    conf = cfgm.from_dict(dict(seed_everything=1234, audio=dict(sampling_rate=16000),
                               model=dict(audio_encoder=dict(base="whisper", type="synthetic", downsample_method="pool", downsample_factor=4,
                                                             pooling=dict(kernel_size=8, stride=4)),
                                          llm_embedding_channels=TINY_LLAMA.hidden_size, llm_type=utils.LLAMA_ID),
                               train=dict(optimizer=dict(lr=5e-5, beta1=0.9, beta2=0.999), batch_size=1, grad_accum_interval=2, epochs=1,
                                          use_ld_loss=True, use_fd_loss=True, ntp_loss_weight=0.5, ld_loss_weight=0.5, fd_loss_weight=1.0,
                                          fd_loss_connector_layers=[0, 1, 3]),
                               log=dict(checkpoint_dir=str(tmp_path / "ckpt"), log_dir=str(tmp_path / "logs"), log_interval=2,
                                        validation_interval=1000, num_generate_samples=1)))
    arch = weights.WhisperArch(WC.d_model, WC.encoder_layers, WC.encoder_attention_heads, WC.encoder_ffn_dim, WC.num_mel_bins, WC.max_source_positions)
    enc = enc_mod.AudioEncoder(conf, DEV, dtype=torch.float32, arch=arch)
    enc.load_state_dict(ri.whisper_encoder_state_dict(WC, TINY_LLAMA.hidden_size, seed=9)).eval().to(DEV)
    llm, _ = make_llama(TINY_LLAMA, 52, torch.float32)
    tok = StubTokenizer({utils.LLAMA_PROMPT_PREFIX: t(g["prefix_ids"]), utils.LLAMA_PROMPT_SUFFIX: t(g["suffix_ids"])})
    tr = trainer_mod.Trainer(SimpleNamespace(run_name="w", checkpoint_path=None, gpu_idx=0), conf, DEV, tokenizer=tok, llm=llm, audio_encoder=enc,
                             train_dataset=train_ds, val_dataset=val_ds, dtype=torch.float32)
    w0 = tr.kd.master["encoder.conv2.weight"].clone()
    tr.train()
    assert tr.step == 4 and not torch.equal(w0, tr.kd.master["encoder.conv2.weight"])
    ck = torch.load(tmp_path / "ckpt" / "w" / "epoch_0_step_4.pt", map_location="cpu", weights_only=False)
    assert torch.equal(ck["audio_encoder"]["encoder.conv2.weight"], tr.kd.master["encoder.conv2.weight"].cpu())
    assert any("validation/audio_perplexity" in l for l in open(tmp_path / "logs" / "w" / "metrics.jsonl"))


def test_validate_perplexities_match_reference_fixture(tmp_path):
    """Trainer.validate (ref:trainer.py:400-514) on a 5-sample validation set — audio-prompt and text-prompt
    perplexities exp(mean(nll)) equal the values the reference's own classes produced (tests/golden/validation_tiny.npz, fp32),
    and the per-sample losses equal the reference's."""
    from types import SimpleNamespace
    from test_models_gpu import StubTokenizer
    trainer_mod = pkg("trainer")
    g, v = golden("pipeline_tiny"), golden("validation_tiny")
    gen = torch.Generator().manual_seed(int(v["gen_seed"]))
    V = TINY_LLAMA.vocab_size
    val_ds = []
    for i, n in enumerate(v["n_samples"]):
        text_ids = torch.randint(1, V, (6 - i % 2,), generator=gen)
        resp_ids = torch.randint(1, V, (7 + i % 3,), generator=gen)
        bos = torch.zeros(1, dtype=torch.long)
        val_ds.append({"audio": {"array": ri.synthetic_waveform(int(n), seed=int(n))}, "text": f"utt{n}", "text_input_ids": torch.cat([bos, text_ids]),
                       "response_input_ids": torch.cat([bos, resp_ids])[None], "pool_ranges_4": []})
    conf = cfgm.from_dict(dict(seed_everything=1234, audio=dict(sampling_rate=16000),
                               model=dict(audio_encoder=dict(base="hubert", type="synthetic", downsample_method="pool", downsample_factor=4,
                                                             pooling=dict(kernel_size=8, stride=4)),
                                          llm_embedding_channels=TINY_LLAMA.hidden_size, llm_type=utils.LLAMA_ID),
                               train=dict(optimizer=dict(lr=5e-5, beta1=0.9, beta2=0.999), batch_size=1, grad_accum_interval=4, epochs=1,
                                          use_ld_loss=True, use_fd_loss=True, ntp_loss_weight=0.5, ld_loss_weight=0.5, fd_loss_weight=1.0,
                                          fd_loss_connector_layers=[0, 1, 3]),
                               log=dict(checkpoint_dir=str(tmp_path / "ckpt"), log_dir=str(tmp_path / "logs"), log_interval=4,
                                        validation_interval=1000, num_generate_samples=2)))
    enc, _ = make_encoder(TINY_HUBERT, TINY_LLAMA.hidden_size, int(v["enc_seed"]), torch.float32)
    llm, _ = make_llama(TINY_LLAMA, int(v["llm_seed"]), torch.float32)
    tok = StubTokenizer({utils.LLAMA_PROMPT_PREFIX: t(g["prefix_ids"]), utils.LLAMA_PROMPT_SUFFIX: t(g["suffix_ids"])})
    tr = trainer_mod.Trainer(SimpleNamespace(run_name="v", checkpoint_path=None, gpu_idx=0), conf, DEV, tokenizer=tok, llm=llm, audio_encoder=enc,
                             train_dataset=val_ds, val_dataset=val_ds, dtype=torch.float32)
    out = tr.validate(0)
    assert abs(out["validation/audio_perplexity"] - float(v["audio_perplexity"])) < 1e-4 * float(v["audio_perplexity"])
    assert abs(out["validation/text_perplexity"] - float(v["text_perplexity"])) < 1e-4 * float(v["text_perplexity"])
    emb = llm.model.embed_tokens
    for i in range(len(val_ds)):
        audio_embeds, text_ids, resp, _ = tr._val_sample(i)
        pre, suf = emb(tr.prefix_ids.to(DEV)), emb(tr.suffix_ids.to(DEV))[:, 1:]
        a_seq = torch.cat([pre, audio_embeds, suf, emb(resp[None])[:, 1:]], dim=1)
        assert abs(float(llm(inputs_embeds=a_seq, labels=[resp]).loss) - float(v["audio_nll"][i])) < 1e-4 * float(v["audio_nll"][i]), i
    line = [l for l in open(tmp_path / "logs" / "v" / "metrics.jsonl")][-1]
    assert "validation/audio_perplexity" in line and "audio_response" in line


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_kd_window_at_benchmark_width_vs_oracle_autograd(dtype):
    """BASELINE configs[2] at the width bench.py's KD leg runs (ref:trainer.py:270-384): HuBERT-large width x 2 layers + Llama-3.2-3B
    width x 2 layers with the full 128 256-way vocabulary, ONE accumulation window of 16 utterances (5-10 s, ragged) as a packed
    micro-batch — the 128^2 / 256^2 GEMMs at M ~ 3 000 rows, the transposed-operand gradient products, the C++ tape, the
    128 256-way loss kernel.  Losses and EVERY encoder parameter gradient against autograd through the CPU oracle, one utterance
    at a time as the reference loops.  fp32: losses <= 1e-5 relative, gradients <= 2e-3 of each parameter's norm (measured ~1e-5);
    bf16 (same bf16-rounded weights on both sides): losses <= 2e-2, gradients <= 0.12 per parameter and <= 5e-2 over all."""
    import os
    from oracle.golden_cfgs import WIDE_HUBERT, WIDE_LLAMA
    HC, LC = WIDE_HUBERT, WIDE_LLAMA
    enc, enc_sd = make_encoder(HC, LC.hidden_size, 91, dtype)
    llm, llm_sd = make_llama(LC, 92, dtype, max_ctx=512)
    prefix, suffix = ri.synthetic_ids(9, LC.vocab_size, seed=7, bos=128000), ri.synthetic_ids(6, LC.vocab_size, seed=8, bos=128000)
    taps = (0, 1, 2)
    tr = training.KDTrainer(kd_config(taps=taps, accum=16), enc, llm, prefix, suffix)
    tr.optimizer_step = lambda: None
    gen = torch.Generator().manual_seed(314)
    B = 16
    waves = [ri.synthetic_waveform(80000 + 5000 * u, seed=700 + u) for u in range(B)]        # 5.0 ... 9.7 s
    texts = [torch.randint(1, LC.vocab_size, (30 + u % 11,), generator=gen) for u in range(B)]
    resps = [torch.randint(1, LC.vocab_size, (48 + (5 * u) % 17,), generator=gen) for u in range(B)]
    losses = tr.micro_batch([w.to(DEV) for w in waves], texts, resps)
    grads = training.kernel_grads_to_state_dict(tr.enc, tr.grads, tr.master)
    torch.cuda.synchronize()
    # oracle: the reference's loop, batch size 1, loss / grad_accum_interval, gradients accumulate
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    rnd_ = (lambda v: v) if dtype == torch.float32 else (lambda v: v.to(torch.bfloat16).float())
    sd = {k: (v.clone() if "conv_layers.0." in k else rnd_(v.clone())).requires_grad_(k != "encoder.masked_spec_embed") for k, v in enc_sd.items()}
    lsd = {k: rnd_(v) for k, v in llm_sd.items()}
    ref_losses = []
    for u in range(B):
        audio = ho.audio_encoder_forward(sd, HC, waves[u][None])
        r = ko.kd_losses(lsd, LC, audio, texts[u], resps[u], prefix, suffix, connector_layers=taps, tail_logits_only=True)
        (r["total"] / 16).backward()
        ref_losses.append({k: float(v) for k, v in r.items()})
    l_tol = 1e-5 if dtype == torch.float32 else 2e-2
    for u in range(B):
        for k, rk in (("ntp_loss", "ntp"), ("ld_loss", "ld"), ("fd_loss", "fd"), ("total", "total")):
            assert abs(losses[u][k] - ref_losses[u][rk]) <= l_tol * max(1.0, abs(ref_losses[u][rk])), (u, k, losses[u][k], ref_losses[u][rk])
    total = float(torch.stack([sd[k].grad.norm() for k in tr.trainable]).norm())
    per_tol = 2e-3 if dtype == torch.float32 else 0.12
    worst, sq = (None, 0.0), 0.0
    for k in tr.trainable:
        ref = sd[k].grad
        err = float((grads[k].cpu().reshape(ref.shape).double() - ref.double()).norm())
        sq += err * err
        scale = float(ref.norm()) + 1e-5 * total      # k_proj.bias: exactly-zero true gradient (softmax shift invariance)
        if err / scale > worst[1]:
            worst = (k, err / scale)
        assert err < per_tol * scale, (k, err, float(ref.norm()))
    overall = sq ** 0.5 / total
    print(f"{dtype}: worst per-parameter gradient error {worst[1]:.2e} ({worst[0]}), all parameters together {overall:.2e}")
    assert overall < (1e-4 if dtype == torch.float32 else 5e-2)


def test_parameter_gradient_side_stream_equals_single_stream(monkeypatch):
    """The encoder backward forks its dW / db products onto a second stream (train_tape.hip SideStream; windows of >= 2048 frames).
    Same kernels, same accumulation order — so every gradient must have the same bits as with SL_NO_WGRAD_STREAM=1, on repeated
    runs (a missing join would show as a run-to-run difference: the products read buffers the main chain recycles)."""
    from oracle.golden_cfgs import WIDE_HUBERT, WIDE_LLAMA
    L_ = pkg("_lib")
    HC, LC = WIDE_HUBERT, WIDE_LLAMA
    enc, _ = make_encoder(HC, LC.hidden_size, 93, torch.bfloat16)
    llm, _ = make_llama(LC, 94, torch.bfloat16, max_ctx=512)
    prefix, suffix = ri.synthetic_ids(9, LC.vocab_size, seed=7, bos=128000), ri.synthetic_ids(6, LC.vocab_size, seed=8, bos=128000)
    tr = training.KDTrainer(kd_config(taps=(0, 1, 2), accum=16), enc, llm, prefix, suffix)
    tr.optimizer_step = lambda: None
    gen = torch.Generator().manual_seed(271)
    B = 16
    waves = [ri.synthetic_waveform(90000 + 4000 * u, seed=800 + u).to(DEV) for u in range(B)]
    texts = [torch.randint(1, LC.vocab_size, (30 + u % 7,), generator=gen) for u in range(B)]
    resps = [torch.randint(1, LC.vocab_size, (40 + (3 * u) % 13,), generator=gen) for u in range(B)]

    def window():
        tr.enc_tape.arena.zero_()
        tr.micro = 0                      # the optimizer step that would close the window (and reset the counter) is stubbed out
        tr.micro_batch(waves, texts, resps)
        torch.cuda.synchronize()
        return tr.enc_tape.arena.flat.clone()

    forked = [window() for _ in range(3)]
    monkeypatch.setenv("SL_NO_WGRAD_STREAM", "1")
    L_.lib().sl_tuning_reload()
    try:
        serial = [window() for _ in range(2)]
    finally:
        monkeypatch.undo()
        L_.lib().sl_tuning_reload()
    assert float(serial[0].abs().max()) > 0
    # bias / conv0 gradients are column sums with atomics: their last bits move from run to run on ONE stream too; every dW is
    # a tiled product accumulated in a fixed order — those must not move at all
    fixed = 0
    for name, (o, n, _shape) in tr.enc_tape.arena.span.items():
        a, b = serial[0][o:o + n], serial[1][o:o + n]
        if torch.equal(a, b):
            fixed += 1
            for f in forked:
                assert torch.equal(f[o:o + n], a), name
        else:
            assert not any(name.endswith(w) for w in (".wqkv", ".wo", ".w1", ".w2")), name
            for f in forked:
                assert float((f[o:o + n] - a).abs().max()) <= 1e-5 * float(a.abs().max()) + 1e-9, name
    assert fixed >= 4 * HC.num_hidden_layers


def test_optimizer_step_overlapped_with_the_backward_equals_the_step_behind_it():
    """KDTrainer steps the parameters of every gradient bucket that is already final (arena prefix: projector, layer N-1, ...) on its own
    stream while the backward still runs (_early_step, opt-in); the rest steps at the end.  AdamW is element-wise, so after ONE window the
    masters, moments and step counts of the layers' weights equal those of `overlap_optimizer=False` bit for bit (bias / conv gradients are
    float-atomic sums whose last bits move from run to run — after a second window everything has seen them, so that one is compared
    to a tolerance)."""
    from oracle.golden_cfgs import WIDE_HUBERT, WIDE_LLAMA
    HC, LC = WIDE_HUBERT, WIDE_LLAMA
    llm, _ = make_llama(LC, 94, torch.bfloat16, max_ctx=512)
    prefix, suffix = ri.synthetic_ids(9, LC.vocab_size, seed=7, bos=128000), ri.synthetic_ids(6, LC.vocab_size, seed=8, bos=128000)
    gen = torch.Generator().manual_seed(55)
    B = 4
    waves = [ri.synthetic_waveform(70000 + 6000 * u, seed=910 + u).to(DEV) for u in range(B)]
    texts = [torch.randint(1, LC.vocab_size, (30 + u,), generator=gen) for u in range(B)]
    resps = [torch.randint(1, LC.vocab_size, (40 + 2 * u,), generator=gen) for u in range(B)]

    def run(overlap):
        enc, _ = make_encoder(HC, LC.hidden_size, 93, torch.bfloat16)
        tr = training.KDTrainer(kd_config(taps=(0, 1, 2), accum=B), enc, llm, prefix, suffix, overlap_optimizer=overlap)
        tr.early_min_bytes = 0                      # every on_bucket call may step (the default waits for 64 MB of final gradients)
        tr.micro_batch(waves, texts, resps)
        torch.cuda.synchronize()
        state = {k: {n: v.clone() for n, v in tr.optimizer.state[tr._param[k]].items()} for k in tr.trainable}
        masters = {k: v.clone() for k, v in tr.master.items()}
        tr.micro_batch(waves, texts, resps)            # a second window: cached record tables, the reset of the early bookkeeping
        torch.cuda.synchronize()
        out = tr.enc(waves[0][None]).clone()
        return tr, masters, state, out, {k: v.clone() for k, v in tr.master.items()}

    tr_o, m_o, s_o, out_o, m2_o = run(True)
    tr_p, m_p, s_p, out_p, m2_p = run(False)
    assert tr_o.early_launches >= 4 and tr_p.early_launches == 0 and tr_o.optimizer_steps == tr_p.optimizer_steps == 2
    for k in m2_o:
        assert float((m2_o[k] - m2_p[k]).abs().max()) <= 2.5e-4, k
    plan = tr_o._early_plan()
    assert sum(n for _, _, n in plan) > 0.9 * sum(tr_o._param[k].numel() for k in tr_o.trainable if ".layers." in k)      # the layers step early
    exact = lambda k: ".layers." in k and k.endswith(".weight")      # (bias / conv gradients are float-atomic sums: their last bits move from run to run)
    n_exact = 0
    for k in m_o:
        if exact(k):
            assert torch.equal(m_o[k], m_p[k]), k
            n_exact += 1
        else:
            assert float((m_o[k] - m_p[k]).abs().max()) <= 2.5e-4, k      # (AdamW's g / (|g| + eps): an element whose gradient is ~0 may move by up to lr per step either way)
    assert n_exact >= 8 * HC.num_hidden_layers
    for k in s_o:
        for n in s_o[k]:
            a, b = s_o[k][n].float().cpu(), s_p[k][n].float().cpu()
            if exact(k):
                assert torch.equal(a, b), (k, n)
            else:
                assert float((a - b).abs().max()) <= 1e-3 * float(b.abs().max()) + 1e-12, (k, n)
    assert float((out_o.float() - out_p.float()).abs().max()) <= 2e-2 * float(out_p.float().abs().max())


@pytest.mark.parametrize("B", [16, 2])      # a 16-sample window (>= 2 048 frames: parameter gradients on the side stream) and the per-rank share of an 8-rank step
def test_weights_transposed_beside_the_forward_give_the_bits_of_a_transpose_per_product(monkeypatch, B):
    """The encoder tape's forward call turns every layer's four weight matrices for the data-gradient products in ONE batched launch on a side
    stream (train_tape.hip WtCache, sl_transpose_pad_batch_impl) and the backward calls on the same workspace read those copies; with
    SL_ENC_WT_AHEAD=0 each product makes its own transpose first, as before.  The copies are exact, so every gradient keeps its bits."""
    _window_bits_equal_under_switch(monkeypatch, B, "SL_ENC_WT_AHEAD")


@pytest.mark.parametrize("B", [16, 2])      # a 16-sample window (>= 2 048 frames: parameter gradients on the side stream) and the per-rank share of an 8-rank step
def test_fused_tape_epilogues_give_the_bits_of_the_unfused_launch_sequence(monkeypatch, B):
    """Round 6: the KD tapes run dropout (forward and backward), GELU', SwiGLU', the b1 bias gradient and the feature-distillation adds inside
    the GEMM epilogues / norm-backward kernels (sl_gemm_ex_args.post_op, train_tape.hip) instead of as launches of their own.  Every fused
    form reproduces the roundings of the sequence it replaces, so with the training-mode regularisers ON (dropouts, LayerDrop,
    SpecAugment — ref:trainer.py:258) the losses and every weight gradient equal those of SL_TAPE_FUSE=0 bit for bit; bias gradients
    (float atomics either way) to their usual last-bit movement."""
    _window_bits_equal_under_switch(monkeypatch, B, "SL_TAPE_FUSE")


@pytest.mark.parametrize("B", [16, 2])
def test_layernorm_backward_in_kernel_column_reduce_gives_the_bits_of_the_reduce_launch(monkeypatch, B):
    """Round 6: in the encoder tape the LayerNorm backward's last-arriving block sums the per-block dgamma / dbeta records itself (arrival counter
    behind the records, agent-scope stores / loads) instead of a norm_colreduce_kernel launch per LayerNorm — same summation order, so every
    LayerNorm gain / bias gradient (and everything else) equals SL_LN_COLRED_INKERNEL=0 bit for bit, window after window on one counter."""
    monkeypatch.setenv("SL_LN_COLRED_INKERNEL", "1")      # (off by default: 52 launches fewer per window but +1.1 ms on the per-rank window, profiles/r06_am_kd_windows.txt)
    pkg("_lib").lib().sl_tuning_reload()
    _window_bits_equal_under_switch(monkeypatch, B, "SL_LN_COLRED_INKERNEL")


def _window_bits_equal_under_switch(monkeypatch, B, switch):
    """One KD window with the regularisers on, default build against `switch`=0 (re-read through sl_tuning_reload)."""
    from oracle.golden_cfgs import WIDE_HUBERT, WIDE_LLAMA
    L_ = pkg("_lib")
    HC, LC = WIDE_HUBERT, WIDE_LLAMA
    enc, _ = make_encoder(HC, LC.hidden_size, 93, torch.bfloat16)
    llm, _ = make_llama(LC, 94, torch.bfloat16, max_ctx=512)
    prefix, suffix = ri.synthetic_ids(9, LC.vocab_size, seed=7, bos=128000), ri.synthetic_ids(6, LC.vocab_size, seed=8, bos=128000)
    gen = torch.Generator().manual_seed(272)
    waves = [ri.synthetic_waveform(90000 + 4000 * u, seed=810 + u).to(DEV) for u in range(B)]
    texts = [torch.randint(1, LC.vocab_size, (30 + u % 7,), generator=gen) for u in range(B)]
    resps = [torch.randint(1, LC.vocab_size, (40 + (3 * u) % 13,), generator=gen) for u in range(B)]

    def window():
        import numpy as np
        np.random.seed(13)                 # SpecAugment spans come from numpy's global generator (hf's _compute_mask_indices)
        torch.manual_seed(13)
        reg = training.TrainRegularizers(feat_proj_dropout=0.1, hidden_dropout=0.1, activation_dropout=0.1, attention_dropout=0.1, layerdrop=0.25,
                                         apply_spec_augment=True, seed=77)        # a fresh counter state: the same masks in every run
        tr = training.KDTrainer(kd_config(taps=(0, 1, 2), accum=B), enc, llm, prefix, suffix, regularizers=reg)
        tr.optimizer_step = lambda: None
        tr.enc_tape.arena.zero_()
        losses = tr.micro_batch(waves, texts, resps)
        torch.cuda.synchronize()
        return tr, tr.enc_tape.arena.flat.clone(), losses

    tr, fused, l_f = window()
    monkeypatch.setenv(switch, "0")
    L_.lib().sl_tuning_reload()
    try:
        _, plain, l_p = window()
    finally:
        monkeypatch.undo()
        L_.lib().sl_tuning_reload()
    assert float(plain.abs().max()) > 0 and len(l_f) == len(l_p)
    for a_, b_ in zip(l_f, l_p):           # (the loss sums are float atomics: their last bits move from run to run)
        for k in a_:
            assert abs(a_[k] - b_[k]) <= 2e-5 * abs(b_[k]) + 1e-7, (k, a_[k], b_[k])
    same = 0
    for name, (o, n, _shape) in tr.enc_tape.arena.span.items():
        a, b = fused[o:o + n], plain[o:o + n]
        if any(name.endswith(w) for w in (".wqkv", ".wo", ".w1", ".w2", ".ln1_g", ".ln1_b", ".ln2_g", ".ln2_b")):
            assert torch.equal(a, b), name
            same += 1
        else:
            assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-9, name
    assert same >= 6 * HC.num_hidden_layers * 0.5      # LayerDrop leaves some layers out
