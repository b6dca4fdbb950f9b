"""State-dict -> device weight tables for libspeechllm.

Accepts the checkpoints the reference consumes:
  * AudioEncoder: flat state-dict (ref:inference.py:24-26) or `{"audio_encoder": ...}` (ref:trainer.py:516-528),
    weight-norm keys in either spelling (`weight_g/weight_v` | `parametrizations.weight.original0/1`);
  * LLM: HF LlamaForCausalLM state-dict (ref:inference.py:47-52).
and lays the tensors out for the HIP kernels (layout notes in include/speechllm.h):
  conv i>=1  (C_out, C_in, k) -> (C_out, k*C_in) tap-major      (implicit GEMM over channel-last rows)
  pos-conv   weight-norm folded, (H, H/g, k) -> (g, H/g, k*H/g)
  q/k/v      fused into one (3H | (nh+2nkv)D, H) matrix
  gate/up    fused in 16-row interleaved blocks so silu(gate)*up is a GEMM epilogue
  RoPE       cos/sin tables computed on the host exactly as hf:models/llama/modeling_llama.py:110-127
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib as L


# ------------------------------------------------------------------------------------------------
# architecture descriptions (what HF's config.json carries for the two model families)
# ------------------------------------------------------------------------------------------------
@dataclass
class HubertArch:
    conv_dim: Tuple[int, ...] = (512,) * 7
    conv_kernel: Tuple[int, ...] = (10, 3, 3, 3, 3, 2, 2)
    conv_stride: Tuple[int, ...] = (5, 2, 2, 2, 2, 2, 2)
    hidden_size: int = 1024
    num_hidden_layers: int = 24
    num_attention_heads: int = 16
    intermediate_size: int = 4096
    num_conv_pos_embeddings: int = 128
    num_conv_pos_embedding_groups: int = 16
    layer_norm_eps: float = 1e-5

    def num_frames(self, n: int) -> int:
        for k, s in zip(self.conv_kernel, self.conv_stride):
            n = (n - k) // s + 1
        return n

    @staticmethod
    def from_hf_config(d: dict) -> "HubertArch":
        if d.get("feat_extract_norm", "layer") != "layer" or not d.get("do_stable_layer_norm", True):
            raise L.SpeechLLMError("only the layer-norm / stable-layer-norm HuBERT variant (hubert-large) is built")
        return HubertArch(tuple(d["conv_dim"]), tuple(d["conv_kernel"]), tuple(d["conv_stride"]), d["hidden_size"],
                          d["num_hidden_layers"], d["num_attention_heads"], d["intermediate_size"],
                          d["num_conv_pos_embeddings"], d["num_conv_pos_embedding_groups"], d.get("layer_norm_eps", 1e-5))


@dataclass
class LlamaArch:
    hidden_size: int = 3072
    num_hidden_layers: int = 28
    num_attention_heads: int = 24
    num_key_value_heads: int = 8
    head_dim: int = 128
    intermediate_size: int = 8192
    vocab_size: int = 128256
    rms_norm_eps: float = 1e-5
    rope_theta: float = 500000.0
    rope_scaling: Optional[dict] = None
    tie_word_embeddings: bool = True
    eos_token_ids: Tuple[int, ...] = (128001, 128008, 128009)
    pad_token_id: Optional[int] = None
    bos_token_id: Optional[int] = 128000

    @staticmethod
    def from_hf_config(d: dict) -> "LlamaArch":
        rs = d.get("rope_scaling") or d.get("rope_parameters")
        theta = d.get("rope_theta", (rs or {}).get("rope_theta", 10000.0))
        scaling = None
        if rs and rs.get("rope_type", rs.get("type", "default")) == "llama3":
            scaling = {k: rs[k] for k in ("factor", "low_freq_factor", "high_freq_factor", "original_max_position_embeddings")}
        elif rs and rs.get("rope_type", rs.get("type", "default")) not in ("default", None):
            raise L.SpeechLLMError(f"rope type {rs.get('rope_type')} not built (default, llama3)")
        eos = d.get("eos_token_id", 2)
        eos = tuple(eos) if isinstance(eos, (list, tuple)) else (eos,)
        nh = d["num_attention_heads"]
        return LlamaArch(d["hidden_size"], d["num_hidden_layers"], nh, d.get("num_key_value_heads", nh),
                         d.get("head_dim") or d["hidden_size"] // nh, d["intermediate_size"], d["vocab_size"],
                         d.get("rms_norm_eps", 1e-6), theta, scaling, d.get("tie_word_embeddings", False), eos,
                         d.get("pad_token_id"), d.get("bos_token_id"))


KNOWN_HUBERT = {"facebook/hubert-large-ls960-ft": HubertArch(), "facebook/hubert-large-ll60k": HubertArch()}
KNOWN_LLAMA = {
    "meta-llama/Llama-3.2-3B-Instruct": LlamaArch(
        rope_scaling=dict(factor=32.0, low_freq_factor=1.0, high_freq_factor=4.0, original_max_position_embeddings=8192)),
    "GeneZC/MiniChat-2-3B": LlamaArch(hidden_size=3072, num_hidden_layers=24, num_attention_heads=24, num_key_value_heads=24,
                                      head_dim=128, intermediate_size=8192, vocab_size=49216, rms_norm_eps=1e-5,
                                      rope_theta=10000.0, rope_scaling=None, tie_word_embeddings=False, eos_token_ids=(2,),
                                      pad_token_id=None, bos_token_id=1),
}


# ------------------------------------------------------------------------------------------------
# RoPE tables (host, fp32) — hf:modeling_rope_utils.py:636-660, hf:models/llama/modeling_llama.py:110-127
# ------------------------------------------------------------------------------------------------
def rope_inv_freq(a: LlamaArch) -> torch.Tensor:
    dim = a.head_dim
    inv_freq = 1.0 / (a.rope_theta ** (torch.arange(0, dim, 2, dtype=torch.int64).to(torch.float) / dim))
    if a.rope_scaling is None:
        return inv_freq
    factor, low, high = a.rope_scaling["factor"], a.rope_scaling["low_freq_factor"], a.rope_scaling["high_freq_factor"]
    old_len = a.rope_scaling["original_max_position_embeddings"]
    low_wl, high_wl = old_len / low, old_len / high
    wavelen = 2 * math.pi / inv_freq
    inv_l = torch.where(wavelen > low_wl, inv_freq / factor, inv_freq)
    smooth = (old_len / wavelen - low) / (high - low)
    smoothed = (1 - smooth) * inv_l / factor + smooth * inv_l
    medium = ~(wavelen < high_wl) * ~(wavelen > low_wl)
    return torch.where(medium, smoothed, inv_l)


def rope_tables(a: LlamaArch, length: int) -> Tuple[torch.Tensor, torch.Tensor]:
    freqs = torch.arange(length, dtype=torch.float32)[:, None] * rope_inv_freq(a)[None, :]
    return freqs.cos().contiguous(), freqs.sin().contiguous()


# ------------------------------------------------------------------------------------------------
# HuBERT / AudioEncoder
# ------------------------------------------------------------------------------------------------
def normalize_encoder_state_dict(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    if "audio_encoder" in sd and isinstance(sd["audio_encoder"], dict):
        sd = sd["audio_encoder"]
    return sd


def resolve_pretrained_dir(type_str: str) -> Optional[str]:
    """Where `AutoModel.from_pretrained(type_str)` (ref:model/audio_encoder.py:6-13) would read from without a network: the
    path itself when it is a directory, else the newest snapshot of that hub id in the local HF cache (HF_HUB_CACHE /
    HF_HOME / ~/.cache/huggingface/hub, `models--org--name/snapshots/<rev>/`).  None when neither exists."""
    import os
    if os.path.isdir(type_str):
        return type_str
    roots = [os.environ.get("HF_HUB_CACHE"), os.environ.get("HUGGINGFACE_HUB_CACHE"),
             os.path.join(os.environ["HF_HOME"], "hub") if os.environ.get("HF_HOME") else None,
             os.path.join(os.path.expanduser("~"), ".cache", "huggingface", "hub")]
    for root in roots:
        if not root:
            continue
        snaps = os.path.join(root, "models--" + type_str.replace("/", "--"), "snapshots")
        if os.path.isdir(snaps):
            revs = [os.path.join(snaps, r) for r in os.listdir(snaps)]
            revs = [r for r in revs if os.path.exists(os.path.join(r, "config.json"))]
            if revs:
                return max(revs, key=os.path.getmtime)
    return None


def read_hf_weight_files(path: str) -> Optional[Dict[str, torch.Tensor]]:
    """Every tensor of an HF checkpoint directory: `model.safetensors`, sharded `model-0000x-of-0000y.safetensors`, or the
    older `pytorch_model.bin` / `pytorch_model-*.bin` pickles (tensors only).  None when the directory holds no weights."""
    import os
    names = sorted(os.listdir(path))
    st = [n for n in names if n.endswith(".safetensors")]
    sd: Dict[str, torch.Tensor] = {}
    if st:
        from safetensors.torch import load_file
        for n in st:
            sd.update(load_file(os.path.join(path, n)))
        return sd
    bins = [n for n in names if n.startswith("pytorch_model") and n.endswith(".bin")]
    for n in bins:
        sd.update(torch.load(os.path.join(path, n), map_location="cpu", weights_only=True))
    return sd or None


def pretrained_encoder_state_dict(path: str, base: str) -> Optional[Dict[str, torch.Tensor]]:
    """The `encoder.*` half of the reference AudioEncoder's state-dict, read from an HF checkpoint directory the way
    ref:model/audio_encoder.py:6-13 gets it from `AutoModel.from_pretrained`:
      hubert  -> HubertModel: keys of a bare HubertModel as they are, keys of a head model (HubertForCTC, e.g.
                 hubert-large-ls960-ft) with their `hubert.` base-model prefix stripped and the head (`lm_head.*`) dropped;
      whisper -> WhisperModel(...).encoder: `encoder.*` of a bare WhisperModel or `model.encoder.*` of
                 WhisperForConditionalGeneration; the decoder and `proj_out` are dropped.
    Tensors come back fp32 under `encoder.<HF name>` (the attribute the reference stores the module under).  `embed_projection`
    is not part of any pretrained checkpoint: `init_embed_projection` draws it."""
    raw = read_hf_weight_files(path)
    if raw is None:
        return None
    out: Dict[str, torch.Tensor] = {}
    if base == "hubert":
        prefixed = any(k.startswith("hubert.") for k in raw)
        for k, v in raw.items():
            if prefixed:
                if not k.startswith("hubert."):
                    continue                                   # lm_head.* of the CTC head
                k = k[len("hubert."):]
            elif k.startswith(("lm_head.", "classifier.", "projector.")):
                continue
            out["encoder." + k] = v.float()
        need = "encoder.feature_projection.projection.weight"
    else:
        for k, v in raw.items():
            if k.startswith("model.encoder."):
                out[k[len("model."):]] = v.float()
            elif k.startswith("encoder."):
                out[k] = v.float()
        need = "encoder.conv1.weight"
    if need not in out:
        raise L.SpeechLLMError(f"{path}: no {base} encoder weights among {len(raw)} tensors (first keys {sorted(raw)[:3]})")
    return out


def init_embed_projection(in_features: int, out_features: int, seed: int) -> Dict[str, torch.Tensor]:
    """`nn.Linear(in, out)`'s default initialisation (torch/nn/modules/linear.py reset_parameters: kaiming_uniform_(a=sqrt 5)
    = U(-1/sqrt(in), 1/sqrt(in)) for the weight, the same bound for the bias) — what ref:model/audio_encoder.py:39-52 gets for
    `embed_projection`.  The reference draws it from whatever state the CPU RNG is in (it seeds only the CUDA RNG,
    ref:trainer.py:33); here the draw comes from a private generator seeded by `seed` alone, so every data-parallel rank starts
    from the same projection whatever its own RNG streams are seeded with."""
    g = torch.Generator(device="cpu").manual_seed(int(seed))
    bound = 1.0 / math.sqrt(in_features)
    w = (torch.rand(out_features, in_features, generator=g, dtype=torch.float32) * 2 - 1) * bound
    b = (torch.rand(out_features, generator=g, dtype=torch.float32) * 2 - 1) * bound
    return {"embed_projection.weight": w, "embed_projection.bias": b}


_WN_PAIRS = (("parametrizations.weight.original0", "weight_g"), ("parametrizations.weight.original1", "weight_v"))


def rename_weight_norm_keys(sd: Dict[str, torch.Tensor], like: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Checkpoint keys -> the weight-norm spelling `like` uses (`...conv.weight_g/weight_v` written by torch < 2.1, what the
    released checkpoint carries, vs `...conv.parametrizations.weight.original0/1`; SURVEY.md §5 checkpoint row).  The
    tensors are the same under both spellings (g: (1,1,k), v: (H, H/G, k))."""
    out = {}
    for k, v in sd.items():
        k2 = k
        for new, old in _WN_PAIRS:
            if k.endswith(old) and k not in like and k[:-len(old)] + new in like:
                k2 = k[:-len(old)] + new
            elif k.endswith(new) and k not in like and k[:-len(new)] + old in like:
                k2 = k[:-len(new)] + old
        out[k2] = v
    return out


def reference_param_order(keys) -> List[str]:
    """Encoder state-dict keys in the order `AudioEncoder.parameters()` yields them in the reference (module registration
    order of ref:model/audio_encoder.py:17-54 over HF's HubertModel / WhisperEncoder): this is the index space of the
    reference's optimizer state (ref:trainer.py:98-105, first param group), so checkpoints written here load there and back.
    HubertModel's own parameter (masked_spec_embed) comes before its children's; HF attention registers k, v, q, out."""
    keys = list(keys)

    def rank(k: str):
        if k.startswith("embed_projection."):
            return (9, 0, 0, k.endswith("bias"))
        if k == "encoder.masked_spec_embed":
            return (0, 0, 0, 0)
        body = k[len("encoder."):] if k.startswith("encoder.") else k
        wb = 1 if body.endswith("bias") else 0
        if body.startswith("feature_extractor.conv_layers."):
            i = int(body.split(".")[2])
            return (1, i, 0 if ".conv." in body else 1, wb)
        if body.startswith("feature_projection."):
            return (2, 0, 0 if "layer_norm" in body else 1, wb)
        if body.startswith("encoder.pos_conv_embed."):
            sub = 0 if body.endswith("conv.bias") else (1 if body.endswith(("original0", "weight_g")) else 2)
            return (3, 0, sub, 0)
        if body.startswith("encoder.layer_norm."):
            return (4, 0, 0, wb)
        if body.startswith("encoder.layers."):            # HuBERT
            li = int(body.split(".")[2])
            tail = body.split(".", 3)[3]
            order = ["attention.k_proj", "attention.v_proj", "attention.q_proj", "attention.out_proj", "layer_norm",
                     "feed_forward.intermediate_dense", "feed_forward.output_dense", "final_layer_norm"]
            return (5, li, next(j for j, o in enumerate(order) if tail.startswith(o + ".")), wb)
        # Whisper encoder (hf:models/whisper/modeling_whisper.py): conv1, conv2, embed_positions, layers, layer_norm
        if body.startswith(("conv1.", "conv2.")):
            return (1, int(body[4]), 0, wb)
        if body.startswith("embed_positions."):
            return (2, 0, 0, 0)
        if body.startswith("layers."):
            li = int(body.split(".")[1])
            tail = body.split(".", 2)[2]
            order = ["self_attn.k_proj", "self_attn.v_proj", "self_attn.q_proj", "self_attn.out_proj", "self_attn_layer_norm", "fc1", "fc2",
                     "final_layer_norm"]
            return (5, li, next(j for j, o in enumerate(order) if tail.startswith(o + ".")), wb)
        if body.startswith("layer_norm."):
            return (6, 0, 0, wb)
        raise L.SpeechLLMError(f"reference_param_order: unexpected encoder parameter '{k}'")

    return sorted(keys, key=rank)


def fold_pos_conv_weight(sd: Dict[str, torch.Tensor], prefix: str) -> torch.Tensor:
    if prefix + "parametrizations.weight.original0" in sd:
        g, v = sd[prefix + "parametrizations.weight.original0"], sd[prefix + "parametrizations.weight.original1"]
    elif prefix + "weight_g" in sd:
        g, v = sd[prefix + "weight_g"], sd[prefix + "weight_v"]
    else:
        return sd[prefix + "weight"].float()
    g, v = g.float(), v.float()
    return v * (g / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt())


def build_layernorm_fold(w) -> None:
    """LayerNorm-folded copies of every layer's q|k|v and FFN1 weights for the inference path (sl_hubert_fold, bf16 only): the
    gain goes into the weight's columns (rounded to bf16 once), u = the row sums of that rounded matrix, c = W . beta + bias in
    fp32 — Linear(LayerNorm(x)) = rstd (x W'^T - mean u) + c, so sl_hubert_forward needs no LayerNorm pass inside the layers
    (hf:models/hubert/modeling_hubert.py:515-517,612).  Built from the DEVICE tensors (the values the unfused kernels would
    read); the first call allocates, later calls (weights moved: KD optimizer steps) update in place."""
    w.fold_stale = False
    if w.dtype != torch.bfloat16 or torch.device(w.device).type != "cuda":
        return
    first = not hasattr(w, "_fold_t")
    if first:
        w._fold_t = []
        w._fold = (L.HubertFold * len(w.layer_t))()
    for li, lt in enumerate(w.layer_t):
        if first:
            outs = []
            for wk in ("wqkv", "w1"):
                N_ = lt[wk].shape[0]
                outs += [torch.empty_like(lt[wk]), torch.empty(N_, device=lt[wk].device, dtype=torch.float32), torch.empty(N_, device=lt[wk].device, dtype=torch.float32)]
            w._fold_t.append(outs)
            f = w._fold[li]
            f.wqkv_f, f.uqkv, f.cqkv, f.w1_f, f.u1, f.c1 = (t.data_ptr() for t in outs)
        outs = w._fold_t[li]
        # one HIP launch per Linear (sl_layernorm_fold_build), in place into the tensors the model struct already points at
        for j, (wk, bk, gk, bek) in enumerate((("wqkv", "bqkv", "ln1_g", "ln1_b"), ("w1", "b1", "ln2_g", "ln2_b"))):
            Wd = lt[wk]
            L.check(L.lib().sl_layernorm_fold_build(L.ptr(Wd), L.ptr(lt[gk]), L.ptr(lt[bek]), L.ptr(lt[bk]), L.ptr(outs[3 * j]), L.ptr(outs[3 * j + 1]),
                                                    L.ptr(outs[3 * j + 2]), Wd.shape[0], Wd.shape[1], L.dtype_code(Wd.dtype), L.stream_ptr()),
                    "sl_layernorm_fold_build")
    if first:
        w.struct.fold = C.cast(w._fold, C.POINTER(L.HubertFold))


class HubertDeviceWeights:
    """Owns the device tensors and the sl_hubert_model struct that points at them."""

    def __init__(self, arch: HubertArch, sd: Dict[str, torch.Tensor], llm_dim: int, device, dtype: torch.dtype,
                 pool_kernel: int = 8, pool_stride: int = 4, downsample: str = "pool"):
        sd = normalize_encoder_state_dict(sd)
        self.arch, self.dtype, self.device, self.llm_dim = arch, dtype, device, llm_dim
        self._keep: List[torch.Tensor] = []
        self._pool, self._downsample = (pool_kernel, pool_stride), downsample
        if arch.hidden_size // arch.num_attention_heads != 64:
            raise L.SpeechLLMError("HuBERT head_dim must be 64 for the built attention kernel")
        self.t: Dict[str, torch.Tensor] = {}   # role -> device tensor (the training graph composes ops from these)
        by_ptr: Dict[int, torch.Tensor] = {}

        def dev(t: torch.Tensor, dt=None) -> torch.Tensor:
            t = t.detach().to(device=device, dtype=dt or dtype).contiguous()
            self._keep.append(t)
            by_ptr[t.data_ptr()] = t
            return t

        self._populate(sd, dev, by_ptr)
        build_layernorm_fold(self)

    def refresh(self, sd: Dict[str, torch.Tensor]) -> None:
        """Re-derive every device tensor from `sd` IN PLACE (the optimizer step of the KD trainer: the fp32 master weights
        moved, the compute-dtype kernel copies follow).  Same expressions, same order as construction; no device tensor, struct
        or pointer changes, so captured launches / cached descriptors stay valid and nothing is re-allocated."""
        sd = normalize_encoder_state_dict(sd)
        it = iter(self._keep)

        def dev(t: torch.Tensor, dt=None) -> torch.Tensor:
            dst = next(it)
            dst.copy_(t.detach().reshape(dst.shape))
            return dst

        self._populate(sd, dev, None)
        self.fold_stale = True          # the LayerNorm-folded copies follow lazily, at the next inference-mode encode

    def _populate(self, sd, dev, by_ptr) -> None:
        arch, dtype, llm_dim = self.arch, self.dtype, self.llm_dim
        (pool_kernel, pool_stride), downsample = self._pool, self._downsample
        H, G, kpos = arch.hidden_size, arch.num_conv_pos_embedding_groups, arch.num_conv_pos_embeddings
        m = L.HubertModel()
        m.dtype = L.dtype_code(dtype)
        m.n_conv, m.hidden, m.n_layers = len(arch.conv_dim), H, arch.num_hidden_layers
        m.n_heads, m.ffn, m.pos_k, m.pos_groups = arch.num_attention_heads, arch.intermediate_size, kpos, G
        for i in range(m.n_conv):
            m.conv_dim[i], m.conv_kernel[i], m.conv_stride[i] = arch.conv_dim[i], arch.conv_kernel[i], arch.conv_stride[i]
        m.ln_eps, m.pool_kernel, m.pool_stride, m.llm_dim = arch.layer_norm_eps, pool_kernel, pool_stride, llm_dim
        p = "encoder.feature_extractor.conv_layers."
        m.conv0_w = dev(sd[p + "0.conv.weight"].reshape(arch.conv_dim[0], -1), torch.float32).data_ptr()
        m.conv0_b = dev(sd[p + "0.conv.bias"], torch.float32).data_ptr()
        m.conv0_g = dev(sd[p + "0.layer_norm.weight"], torch.float32).data_ptr()
        m.conv0_beta = dev(sd[p + "0.layer_norm.bias"], torch.float32).data_ptr()
        for i in range(1, m.n_conv):
            w = sd[p + f"{i}.conv.weight"]  # (Cout, Cin, k) -> (Cout, k, Cin) -> (Cout, k*Cin)
            m.conv_w[i] = dev(w.permute(0, 2, 1).reshape(w.shape[0], -1)).data_ptr()
            m.conv_b[i] = dev(sd[p + f"{i}.conv.bias"]).data_ptr()
            m.conv_g[i] = dev(sd[p + f"{i}.layer_norm.weight"]).data_ptr()
            m.conv_beta[i] = dev(sd[p + f"{i}.layer_norm.bias"]).data_ptr()
        p = "encoder.feature_projection."
        m.fp_ln_g, m.fp_ln_b = dev(sd[p + "layer_norm.weight"]).data_ptr(), dev(sd[p + "layer_norm.bias"]).data_ptr()
        m.fp_w, m.fp_b = dev(sd[p + "projection.weight"]).data_ptr(), dev(sd[p + "projection.bias"]).data_ptr()
        p = "encoder.encoder.pos_conv_embed.conv."
        wpos = fold_pos_conv_weight(sd, p)  # (H, H/G, k): out, cin, tap
        Hg = H // G
        m.pos_w = dev(wpos.permute(0, 2, 1).reshape(G, Hg, kpos * Hg)).data_ptr()
        m.pos_b = dev(sd[p + "bias"]).data_ptr()
        layers = (L.HubertLayer * arch.num_hidden_layers)()
        if by_ptr is not None:
            self._layers = layers
        for li in range(arch.num_hidden_layers):
            p = f"encoder.encoder.layers.{li}."
            a = p + "attention."
            lay = layers[li]
            lay.ln1_g, lay.ln1_b = dev(sd[p + "layer_norm.weight"]).data_ptr(), dev(sd[p + "layer_norm.bias"]).data_ptr()
            lay.wqkv = dev(torch.cat([sd[a + "q_proj.weight"], sd[a + "k_proj.weight"], sd[a + "v_proj.weight"]], 0)).data_ptr()
            lay.bqkv = dev(torch.cat([sd[a + "q_proj.bias"], sd[a + "k_proj.bias"], sd[a + "v_proj.bias"]], 0)).data_ptr()
            lay.wo, lay.bo = dev(sd[a + "out_proj.weight"]).data_ptr(), dev(sd[a + "out_proj.bias"]).data_ptr()
            lay.ln2_g = dev(sd[p + "final_layer_norm.weight"]).data_ptr()
            lay.ln2_b = dev(sd[p + "final_layer_norm.bias"]).data_ptr()
            f = p + "feed_forward."
            lay.w1, lay.b1 = dev(sd[f + "intermediate_dense.weight"]).data_ptr(), dev(sd[f + "intermediate_dense.bias"]).data_ptr()
            lay.w2, lay.b2 = dev(sd[f + "output_dense.weight"]).data_ptr(), dev(sd[f + "output_dense.bias"]).data_ptr()
        m.layers = C.cast(layers, C.POINTER(L.HubertLayer))
        m.final_ln_g = dev(sd["encoder.encoder.layer_norm.weight"]).data_ptr()
        m.final_ln_b = dev(sd["encoder.encoder.layer_norm.bias"]).data_ptr()
        proj_w = dev(sd["embed_projection.weight"])
        proj_b = dev(sd["embed_projection.bias"])
        if by_ptr is not None:
            self.proj_w, self.proj_b = proj_w, proj_b
        if downsample == "pool":
            m.proj_w, m.proj_b = proj_w.data_ptr(), proj_b.data_ptr()
        else:  # stack / ctc_pool: the host composes the downsample from ops on last_hidden
            m.proj_w, m.proj_b = None, None
        if by_ptr is None:      # refresh(): tensors, struct and role tables stay as built
            return
        self.struct = m
        g = lambda p_: by_ptr[p_]
        self.t.update(conv0_w=g(m.conv0_w), conv0_b=g(m.conv0_b), conv0_g=g(m.conv0_g), conv0_beta=g(m.conv0_beta),
                      fp_ln_g=g(m.fp_ln_g), fp_ln_b=g(m.fp_ln_b), fp_w=g(m.fp_w), fp_b=g(m.fp_b), pos_w=g(m.pos_w), pos_b=g(m.pos_b),
                      final_ln_g=g(m.final_ln_g), final_ln_b=g(m.final_ln_b), proj_w=self.proj_w, proj_b=self.proj_b)
        for i in range(1, m.n_conv):
            self.t.update({f"conv{i}_w": g(m.conv_w[i]), f"conv{i}_b": g(m.conv_b[i]), f"conv{i}_g": g(m.conv_g[i]),
                           f"conv{i}_beta": g(m.conv_beta[i])})
        self.layer_t = [{n: g(getattr(self._layers[li], n)) for n, _ in L.HubertLayer._fields_} for li in range(arch.num_hidden_layers)]

    def n_params(self) -> int:
        return sum(t.numel() for t in self._keep)

    # -- what the fused optimizer step (training.FusedAdamW) needs: the device tensor of a kernel-layout role, and the roles
    #    whose device tensor is NOT a plain cast of state-dict parameters (re-derived by refresh_indirect) --------------------
    def role_tensor(self, role: str) -> Optional[torch.Tensor]:
        if role in self.t:
            return self.t[role]
        if role.startswith("l") and "." in role and role[1:role.index(".")].isdigit():
            return self.layer_t[int(role[1:role.index(".")])].get(role[role.index(".") + 1:])
        return None

    def indirect_roles(self) -> List[str]:
        return [f"conv{i}_w" for i in range(1, len(self.arch.conv_dim))] + ["pos_w"]

    def constant_roles(self) -> List[str]:
        return []

    def refresh_indirect(self, sd: Dict[str, torch.Tensor]) -> None:
        """In-place refresh of the tensors `indirect_roles` names, same expressions as _populate."""
        arch = self.arch
        H, G, kpos = arch.hidden_size, arch.num_conv_pos_embedding_groups, arch.num_conv_pos_embeddings
        p = "encoder.feature_extractor.conv_layers."
        for i in range(1, len(arch.conv_dim)):
            w = sd[p + f"{i}.conv.weight"]
            dst = self.t[f"conv{i}_w"]
            dst.copy_(w.permute(0, 2, 1).reshape(dst.shape))
        wpos = fold_pos_conv_weight(sd, "encoder.encoder.pos_conv_embed.conv.")
        self.t["pos_w"].copy_(wpos.permute(0, 2, 1).reshape(self.t["pos_w"].shape))
        self.fold_stale = True      # the fused optimizer step rewrote the layer weights: the LayerNorm-folded copies follow lazily


# ------------------------------------------------------------------------------------------------
# Llama
# ------------------------------------------------------------------------------------------------
def interleave_gate_up(gate: torch.Tensor, up: torch.Tensor) -> torch.Tensor:
    """(F,H),(F,H) -> (2F,H) in blocks of 16 gate rows then 16 up rows (SL_ACT_SILU_MUL layout)."""
    F_, H = gate.shape
    if F_ % 16:
        raise L.SpeechLLMError("intermediate_size must be a multiple of 16")
    return torch.stack([gate.reshape(F_ // 16, 16, H), up.reshape(F_ // 16, 16, H)], dim=1).reshape(2 * F_, H)


class LlamaDeviceWeights:
    def __init__(self, arch: LlamaArch, sd: Dict[str, torch.Tensor], device, dtype: torch.dtype, rope_len: int = 8192):
        self.arch, self.dtype, self.device = arch, dtype, device
        self._keep: List[torch.Tensor] = []
        if arch.head_dim != 128:
            raise L.SpeechLLMError("Llama head_dim must be 128 for the built attention kernels")

        def dev(t: torch.Tensor, dt=None) -> torch.Tensor:
            t = t.detach().to(device=device, dtype=dt or dtype).contiguous()
            self._keep.append(t)
            return t

        m = L.LlamaModel()
        m.dtype = L.dtype_code(dtype)
        m.hidden, m.n_layers, m.n_heads, m.n_kv_heads = arch.hidden_size, arch.num_hidden_layers, arch.num_attention_heads, arch.num_key_value_heads
        m.head_dim, m.ffn, m.vocab, m.rms_eps, m.rope_len = arch.head_dim, arch.intermediate_size, arch.vocab_size, arch.rms_norm_eps, rope_len
        self.embed = dev(sd["model.embed_tokens.weight"])
        m.embed = self.embed.data_ptr()
        self.lm_head = dev(sd["lm_head.weight"]) if "lm_head.weight" in sd and not arch.tie_word_embeddings else self.embed
        m.lm_head = self.lm_head.data_ptr()
        m.final_norm = dev(sd["model.norm.weight"]).data_ptr()
        cos, sin = rope_tables(arch, rope_len)
        m.rope_cos, m.rope_sin = dev(cos, torch.float32).data_ptr(), dev(sin, torch.float32).data_ptr()
        self._layers = (L.LlamaLayer * arch.num_hidden_layers)()
        for li in range(arch.num_hidden_layers):
            p = f"model.layers.{li}."
            lay = self._layers[li]
            lay.norm1 = dev(sd[p + "input_layernorm.weight"]).data_ptr()
            lay.wqkv = dev(torch.cat([sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.k_proj.weight"],
                                      sd[p + "self_attn.v_proj.weight"]], 0)).data_ptr()
            lay.wo = dev(sd[p + "self_attn.o_proj.weight"]).data_ptr()
            lay.norm2 = dev(sd[p + "post_attention_layernorm.weight"]).data_ptr()
            lay.wgu = dev(interleave_gate_up(sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"])).data_ptr()
            lay.wdown = dev(sd[p + "mlp.down_proj.weight"]).data_ptr()
        m.layers = C.cast(self._layers, C.POINTER(L.LlamaLayer))
        self.struct = m
        self.decode_packed = False
        by_ptr = {t.data_ptr(): t for t in self._keep}
        self.layer_t = [{n: by_ptr[getattr(self._layers[li], n)] for n in ("norm1", "wqkv", "wo", "norm2", "wgu", "wdown")}
                        for li in range(arch.num_hidden_layers)]
        self.final_norm = by_ptr[m.final_norm]
        self.rope_cos, self.rope_sin = by_ptr[m.rope_cos], by_ptr[m.rope_sin]

    def build_decode_weights(self, fuse_norm: Optional[bool] = None) -> None:
        """Second, decode-only copy of every matrix in the fragment-packed layout the weight-streaming kernel
        reads at full HBM rate (sl_pack_weight), q/k rows in rotate_half pair order for the fused RoPE epilogue
        and — in bf16 mode — the RMSNorm gains folded in (sl_gemm_fused).  Costs one extra copy of the
        weights in HBM (6.4 GB bf16 for Llama-3.2-3B of 288 GB); the prefill path keeps the row-major set."""
        if self.decode_packed:
            return
        a, dt, dev_ = self.arch, self.dtype, self.device
        if fuse_norm is None:
            fuse_norm = dt == torch.bfloat16  # fp32 = parity mode: keep the reference's op order exactly
        lib, code = L.lib(), L.dtype_code(dt)
        kstep = 32 if dt == torch.bfloat16 else 16
        H, D, nh, nkv, F_ = a.hidden_size, a.head_dim, a.num_attention_heads, a.num_key_value_heads, a.intermediate_size
        if H % kstep or (nh * D) % kstep or F_ % kstep:
            return  # shapes the packed kernel does not take: decode stays on the row-major path

        def pack(t: torch.Tensor) -> torch.Tensor:
            t = t.contiguous()
            n, k = t.shape
            out = torch.empty(((n + 15) // 16 * 16, k), device=dev_, dtype=dt)
            L.check(lib.sl_pack_weight(t.data_ptr(), t.stride(0), out.data_ptr(), n, k, code, L.stream_ptr()), "sl_pack_weight")
            self._keep.append(out)
            return out

        def fold(w: torch.Tensor, gain: torch.Tensor) -> torch.Tensor:
            return (w.float() * gain.float()[None, :]).to(dt) if fuse_norm else w

        blk = torch.arange(16, device=dev_)
        head_perm = torch.cat([torch.cat([blk + 16 * j, blk + 64 + 16 * j]) for j in range(4)])  # [0:16],[64:80],[16:32],...
        perm = torch.cat([head_perm + 128 * h for h in range(nh + nkv)] + [torch.arange((nh + nkv) * 128, (nh + 2 * nkv) * 128, device=dev_)])
        by_ptr = {t.data_ptr(): t for t in self._keep}
        self.dec_wgu: List[torch.Tensor] = []
        for li in range(a.num_hidden_layers):
            lay = self._layers[li]
            wqkv, wo, wgu, wdown = by_ptr[lay.wqkv], by_ptr[lay.wo], by_ptr[lay.wgu], by_ptr[lay.wdown]
            n1, n2 = by_ptr[lay.norm1], by_ptr[lay.norm2]
            lay.wqkv_dec = pack(fold(wqkv, n1)[perm]).data_ptr()
            lay.wo_dec = pack(wo).data_ptr()
            wgu_p = pack(fold(wgu, n2))
            self.dec_wgu.append(wgu_p)
            lay.wgu_dec = wgu_p.data_ptr()
            lay.wdown_dec = pack(wdown).data_ptr()
        self.struct.lm_head_dec = pack(fold(self.lm_head, by_ptr[self.struct.final_norm])).data_ptr()
        self.struct.dec_fused_norm = int(bool(fuse_norm))
        torch.cuda.synchronize(dev_)
        self.decode_packed = True

    def n_params(self) -> int:
        return sum(t.numel() for t in self._keep if t.dtype == self.dtype)

    def weight_bytes_per_token(self) -> int:
        """Bytes of weights one decode step streams: every layer matrix + final norm + lm_head."""
        a = self.arch
        per_layer = (a.num_attention_heads + 2 * a.num_key_value_heads) * a.head_dim * a.hidden_size \
            + a.num_attention_heads * a.head_dim * a.hidden_size + 3 * a.intermediate_size * a.hidden_size + 2 * a.hidden_size
        total = per_layer * a.num_hidden_layers + a.hidden_size + a.vocab_size * a.hidden_size
        return total * (2 if self.dtype == torch.bfloat16 else 4)


# ------------------------------------------------------------------------------------------------
# Whisper (alternate encoder, ref:model/audio_encoder.py:10-13; BASELINE configs[3])
# ------------------------------------------------------------------------------------------------
@dataclass
class WhisperArch:
    d_model: int = 1024
    encoder_layers: int = 24
    encoder_attention_heads: int = 16
    encoder_ffn_dim: int = 4096
    num_mel_bins: int = 80
    max_source_positions: int = 1500
    n_fft: int = 400
    hop_length: int = 160
    sampling_rate: int = 16000

    @property
    def hidden_size(self) -> int:   # common name used by the projector code
        return self.d_model

    @property
    def n_frames(self) -> int:
        return 2 * self.max_source_positions

    @property
    def n_samples(self) -> int:
        return self.n_frames * self.hop_length

    @staticmethod
    def from_hf_config(d: dict) -> "WhisperArch":
        return WhisperArch(d["d_model"], d["encoder_layers"], d["encoder_attention_heads"], d["encoder_ffn_dim"], d.get("num_mel_bins", 80),
                           d.get("max_source_positions", 1500))


KNOWN_WHISPER = {"openai/whisper-medium": WhisperArch(), "openai/whisper-medium.en": WhisperArch()}


def slaney_mel_filters(n_freqs: int, n_mels: int, sr: int = 16000, fmin: float = 0.0, fmax: float = 8000.0) -> torch.Tensor:
    """Slaney-scale, area-normalised triangular mel bank (n_freqs, n_mels), float64 -> float32: the published Auditory
    Toolbox / librosa definition that hf:audio_utils.py:638-729 implements for norm="slaney", mel_scale="slaney"."""
    f_sp, min_log_hz = 200.0 / 3.0, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, math.log(6.4) / 27.0

    def hz_to_mel(f: torch.Tensor) -> torch.Tensor:
        return torch.where(f >= min_log_hz, min_log_mel + torch.log(f.clamp(min=1e-10) / min_log_hz) / logstep, f / f_sp)

    def mel_to_hz(m: torch.Tensor) -> torch.Tensor:
        return torch.where(m >= min_log_mel, min_log_hz * torch.exp(logstep * (m - min_log_mel)), f_sp * m)

    d = torch.float64
    mel_pts = torch.linspace(float(hz_to_mel(torch.tensor(fmin, dtype=d))), float(hz_to_mel(torch.tensor(fmax, dtype=d))), n_mels + 2, dtype=d)
    hz_pts = mel_to_hz(mel_pts)
    fft_freqs = torch.linspace(0, sr // 2, n_freqs, dtype=d)
    diff = hz_pts[1:] - hz_pts[:-1]
    slopes = hz_pts[None, :] - fft_freqs[:, None]
    down, up = -slopes[:, :-2] / diff[:-1], slopes[:, 2:] / diff[1:]
    fb = torch.clamp(torch.minimum(down, up), min=0.0)
    fb = fb * (2.0 / (hz_pts[2:n_mels + 2] - hz_pts[:n_mels]))[None, :]
    return fb.to(torch.float32)


def windowed_dft_basis(n_fft: int) -> torch.Tensor:
    """(2*(n_fft/2+1), n_fft) fp32: rows [hann*cos(2 pi k n / N)] then [hann*sin(...)] — torch.stft(window=hann_window(n_fft))
    as a matrix (the sign of the imaginary part is irrelevant for the power spectrum)."""
    d = torch.float64
    n = torch.arange(n_fft, dtype=d)
    k = torch.arange(n_fft // 2 + 1, dtype=d)
    win = 0.5 - 0.5 * torch.cos(2 * math.pi * n / n_fft)          # periodic Hann, torch.hann_window default
    ang = 2 * math.pi * k[:, None] * n[None, :] / n_fft
    return torch.cat([torch.cos(ang) * win, torch.sin(ang) * win], 0).to(torch.float32).contiguous()


class WhisperDeviceWeights:
    """Device tensors + the sl_hubert_model struct in its Whisper layout (include/speechllm.h, sl_whisper_forward)."""

    def __init__(self, arch: WhisperArch, sd: Dict[str, torch.Tensor], llm_dim: int, device, dtype: torch.dtype, pool_kernel: int = 8,
                 pool_stride: int = 4, downsample: str = "pool"):
        sd = normalize_encoder_state_dict(sd)
        self.arch, self.dtype, self.device, self.llm_dim = arch, dtype, device, llm_dim
        self._keep: List[torch.Tensor] = []
        self._pool, self._downsample = (pool_kernel, pool_stride), downsample
        if arch.d_model // arch.encoder_attention_heads != 64:
            raise L.SpeechLLMError("Whisper head_dim must be 64 for the built attention kernel")

        by_ptr: Dict[int, torch.Tensor] = {}

        def dev(t: torch.Tensor, dt=None) -> torch.Tensor:
            t = t.detach().to(device=device, dtype=dt or dtype).contiguous()
            self._keep.append(t)
            by_ptr[t.data_ptr()] = t
            return t

        self._populate(sd, dev, by_ptr)
        build_layernorm_fold(self)

    def refresh(self, sd: Dict[str, torch.Tensor]) -> None:
        """In-place update of every device tensor from `sd` (see HubertDeviceWeights.refresh)."""
        sd = normalize_encoder_state_dict(sd)
        it = iter(self._keep)

        def dev(t: torch.Tensor, dt=None) -> torch.Tensor:
            dst = next(it)
            dst.copy_(t.detach().reshape(dst.shape))
            return dst

        self._populate(sd, dev, None)
        self.fold_stale = True          # the LayerNorm-folded copies follow lazily, at the next inference-mode encode

    def _populate(self, sd, dev, by_ptr) -> None:
        arch, dtype, llm_dim = self.arch, self.dtype, self.llm_dim
        (pool_kernel, pool_stride), downsample = self._pool, self._downsample
        H = arch.d_model
        m = L.HubertModel()
        m.dtype, m.reserved = L.dtype_code(dtype), 1
        m.n_conv, m.hidden, m.n_layers, m.n_heads, m.ffn = 3, H, arch.encoder_layers, arch.encoder_attention_heads, arch.encoder_ffn_dim
        m.pos_k, m.pos_groups = arch.max_source_positions, 1
        m.conv_dim[0], m.conv_dim[1], m.conv_dim[2] = arch.num_mel_bins, H, H
        m.conv_kernel[0] = m.conv_kernel[1] = m.conv_kernel[2] = 3
        m.conv_stride[0], m.conv_stride[1], m.conv_stride[2] = 1, 1, 2
        m.ln_eps, m.pool_kernel, m.pool_stride, m.llm_dim = 1e-5, pool_kernel, pool_stride, llm_dim
        for i, name in ((1, "conv1"), (2, "conv2")):
            w = sd[f"encoder.{name}.weight"]                     # (Cout, Cin, 3) -> (Cout, 3*Cin) tap-major
            m.conv_w[i] = dev(w.permute(0, 2, 1).reshape(w.shape[0], -1)).data_ptr()
            m.conv_b[i] = dev(sd[f"encoder.{name}.bias"]).data_ptr()
        m.pos_w = dev(sd["encoder.embed_positions.weight"]).data_ptr()
        layers = (L.HubertLayer * arch.encoder_layers)()
        if by_ptr is not None:
            self._layers = layers
        for li in range(arch.encoder_layers):
            p = f"encoder.layers.{li}."
            a = p + "self_attn."
            lay = layers[li]
            lay.ln1_g, lay.ln1_b = dev(sd[p + "self_attn_layer_norm.weight"]).data_ptr(), dev(sd[p + "self_attn_layer_norm.bias"]).data_ptr()
            lay.wqkv = dev(torch.cat([sd[a + "q_proj.weight"], sd[a + "k_proj.weight"], sd[a + "v_proj.weight"]], 0)).data_ptr()
            kb = sd.get(a + "k_proj.bias", torch.zeros_like(sd[a + "q_proj.bias"]))   # Whisper's k_proj has no bias
            lay.bqkv = dev(torch.cat([sd[a + "q_proj.bias"], kb, sd[a + "v_proj.bias"]], 0)).data_ptr()
            lay.wo, lay.bo = dev(sd[a + "out_proj.weight"]).data_ptr(), dev(sd[a + "out_proj.bias"]).data_ptr()
            lay.ln2_g, lay.ln2_b = dev(sd[p + "final_layer_norm.weight"]).data_ptr(), dev(sd[p + "final_layer_norm.bias"]).data_ptr()
            lay.w1, lay.b1 = dev(sd[p + "fc1.weight"]).data_ptr(), dev(sd[p + "fc1.bias"]).data_ptr()
            lay.w2, lay.b2 = dev(sd[p + "fc2.weight"]).data_ptr(), dev(sd[p + "fc2.bias"]).data_ptr()
        m.layers = C.cast(layers, C.POINTER(L.HubertLayer))
        m.final_ln_g, m.final_ln_b = dev(sd["encoder.layer_norm.weight"]).data_ptr(), dev(sd["encoder.layer_norm.bias"]).data_ptr()
        proj_w, proj_b = dev(sd["embed_projection.weight"]), dev(sd["embed_projection.bias"])
        if downsample == "pool":
            m.proj_w, m.proj_b = proj_w.data_ptr(), proj_b.data_ptr()
        if by_ptr is None:      # refresh(): the log-mel constants below never change
            return
        self.proj_w, self.proj_b = proj_w, proj_b
        self.struct = m
        # role -> device tensor, as HubertDeviceWeights does (the training tape composes ops from these)
        g = lambda ptr: by_ptr[ptr]
        self.t = dict(conv1_w=g(m.conv_w[1]), conv1_b=g(m.conv_b[1]), conv2_w=g(m.conv_w[2]), conv2_b=g(m.conv_b[2]), pos=g(m.pos_w),
                      final_ln_g=g(m.final_ln_g), final_ln_b=g(m.final_ln_b), proj_w=self.proj_w, proj_b=self.proj_b)
        self.layer_t = [{n: g(getattr(self._layers[li], n)) for n, _ in L.HubertLayer._fields_} for li in range(arch.encoder_layers)]
        # log-mel constants (fp32)
        nb = arch.n_fft // 2 + 1
        ld_pw = (nb + 3) // 4 * 4
        mel = torch.zeros((arch.num_mel_bins, ld_pw), dtype=torch.float32)
        mel[:, :nb] = slaney_mel_filters(nb, arch.num_mel_bins, arch.sampling_rate).T
        self.mel_w = dev(mel, torch.float32)
        self.dft_basis = dev(windowed_dft_basis(arch.n_fft), torch.float32)

    role_tensor = HubertDeviceWeights.role_tensor

    def indirect_roles(self) -> List[str]:
        return ["conv1_w", "conv2_w"]

    def constant_roles(self) -> List[str]:
        return ["pos"] + [f"l{li}.bqkv" for li in range(self.arch.encoder_layers)]      # bqkv: q / v slots are direct, the k slot stays zero

    def refresh_indirect(self, sd: Dict[str, torch.Tensor]) -> None:
        for name in ("conv1", "conv2"):
            w = sd[f"encoder.{name}.weight"]
            dst = self.t[f"{name}_w"]
            dst.copy_(w.permute(0, 2, 1).reshape(dst.shape))
        self.fold_stale = True
