"""Test helper (not a test): writes HF-layout checkpoint directories from seeded state dicts, the way the hub ships them.

  * HuBERT as HubertForCTC files: `hubert.`-prefixed encoder tensors + an `lm_head` (facebook/hubert-large-ls960-ft's layout),
    safetensors or the older pytorch_model.bin, weight-norm keys in either spelling;
  * Whisper as WhisperForConditionalGeneration files: `model.encoder.*` + a stand-in decoder tensor + `proj_out`;
  * a synthetic KD dataset in the reference's row schema saved with `datasets` (ref:trainer.py:134-199).
"""
import json
import os

import torch


def hubert_config_json(hc, **extra):
    d = dict(model_type="hubert", architectures=["HubertForCTC"], conv_dim=list(hc.conv_dim), conv_kernel=list(hc.conv_kernel),
             conv_stride=list(hc.conv_stride), hidden_size=hc.hidden_size, num_hidden_layers=hc.num_hidden_layers,
             num_attention_heads=hc.num_attention_heads, intermediate_size=hc.intermediate_size,
             num_conv_pos_embeddings=hc.num_conv_pos_embeddings, num_conv_pos_embedding_groups=hc.num_conv_pos_embedding_groups,
             layer_norm_eps=hc.layer_norm_eps, feat_extract_norm="layer", do_stable_layer_norm=True, vocab_size=32)
    d.update(extra)
    return d


def write_hubert_ctc_dir(path, hc, enc_sd, fmt="safetensors", **cfg_extra):
    """enc_sd: AudioEncoder-style state dict (`encoder.*` [+ embed_projection.*, ignored])."""
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(hubert_config_json(hc, **cfg_extra), f)
    files = {"hubert." + k[len("encoder."):]: v.contiguous() for k, v in enc_sd.items() if k.startswith("encoder.")}
    g = torch.Generator().manual_seed(1)
    files["lm_head.weight"] = torch.randn(32, hc.hidden_size, generator=g)
    files["lm_head.bias"] = torch.randn(32, generator=g)
    if fmt == "safetensors":
        from safetensors.torch import save_file
        save_file(files, os.path.join(path, "model.safetensors"))
    else:
        torch.save(files, os.path.join(path, "pytorch_model.bin"))


def whisper_config_json(wc, **extra):
    d = dict(model_type="whisper", architectures=["WhisperForConditionalGeneration"], d_model=wc.d_model, encoder_layers=wc.encoder_layers,
             encoder_attention_heads=wc.encoder_attention_heads, encoder_ffn_dim=wc.encoder_ffn_dim, num_mel_bins=wc.num_mel_bins,
             max_source_positions=wc.max_source_positions, dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, encoder_layerdrop=0.0)
    d.update(extra)
    return d


def write_whisper_gen_dir(path, wc, enc_sd):
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(whisper_config_json(wc), f)
    files = {"model." + k: v.contiguous() for k, v in enc_sd.items() if k.startswith("encoder.")}
    g = torch.Generator().manual_seed(2)
    files["model.decoder.embed_tokens.weight"] = torch.randn(64, wc.d_model, generator=g)
    files["proj_out.weight"] = torch.randn(64, wc.d_model, generator=g)
    from safetensors.torch import save_file
    save_file(files, os.path.join(path, "model.safetensors"))


def write_kd_dataset(path, n_rows, vocab, first_len=16000, step=700, seed=5):
    """`datasets` directory in the reference's preprocessed-row schema: audio{array,sampling_rate}, text, text_input_ids (BOS first),
    response_input_ids (nested [0], BOS first), pool_ranges_4."""
    import importlib
    from datasets import Dataset
    ri = importlib.import_module("llm-speech-summarization_amd.random_init")
    gen = torch.Generator().manual_seed(seed)
    rows = []
    for i in range(n_rows):
        n = first_len + step * i
        rows.append({"audio": {"array": ri.synthetic_waveform(n, seed=n).tolist(), "sampling_rate": 16000}, "text": f"utt{n}",
                     "text_input_ids": [0] + torch.randint(1, vocab, (5 + i % 3,), generator=gen).tolist(),
                     "response_input_ids": [[0] + torch.randint(1, vocab, (6 + i % 4,), generator=gen).tolist()],
                     "pool_ranges_4": [[0, 4], [4, 9]]})
    Dataset.from_list(rows).save_to_disk(path)
