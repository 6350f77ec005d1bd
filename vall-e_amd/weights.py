"""State-dict layout of the reference model + the seeded synthetic-weight generator.

The key names and shapes are those listed in SURVEY.md §8 row a22 (obtained by
instantiating the reference: vall_e/models/base.py:390-494 for the embeddings and
heads, vall_e/models/arch/llama.py:159-162, :486-488, :508-509, :582 for the
stack).  A weights file written by the reference (``fp32.sft``) therefore loads
into this implementation by name.

``synth_state_dict`` is the generator both sides of every parity test use: the
golden-vector script loads its output into the reference model, the GPU tests
load the same tensors into the HIP path, so the GPU box can regenerate identical
weights from this repository alone (no checkpoint travels).
"""
from __future__ import annotations

from collections import OrderedDict

import torch

from .config import Model


def classifier_layout(cfg: Model):
    """(names, n_tokens) per head -- vall_e/models/base.py:390-411."""
    n_resp_tokens = cfg.audio_tokens + (1 if cfg.experimental.causal_size > 0 else 0)
    L = cfg.resp_levels
    emb_tokens = [n_resp_tokens] + [n_resp_tokens - 1] * (L - 1) + [n_resp_tokens]
    cls_tokens = [n_resp_tokens] + [n_resp_tokens - 1] * (L - 1) + [n_resp_tokens - 1]
    names = ["AR:0:0"] + [f"NAR:{i}:{i+1}" for i in range(L - 1)] + ["NAR:0:0"]
    cls_names = list(names) + ["stt"]
    cls_tokens = cls_tokens + [cfg.phoneme_tokens]
    if "len" in cfg.capabilities:
        cls_tokens += [11]
        cls_names += ["len"]
    if cfg.version >= 6:
        cls_tokens += [cfg.text_tokens]
        # base.py:411 rebuilds the name list from the embedding names (which the
        # in-place ``+=`` above has already extended) and appends "text".
        cls_names = cls_names + ["text"]
    return names, emb_tokens, cls_names, cls_tokens


def state_dict_spec_v2(cfg: Model) -> "OrderedDict[str, tuple]":
    """The version >= 7 model (vall_e/models/base_v2.py:412-469, :499; names listed by instantiating the reference):
    plain tables, two AudioEncoders (per-level tables of token_dim = d_model, level position embedding, level weights,
    LayerNorm, a residual FFN), the AudioDecoder (per level: LayerNorm, FFN, head), three auxiliary heads, the llama stack
    WITHOUT its final norm (per_level_normalization moves it into the decoder, base_v2.py:495)."""
    d, ff, L = cfg.dim, cfg.dim * cfg.ffn, cfg.resp_levels
    e = cfg.experimental
    spec = OrderedDict()
    spec["sep"] = (d,)
    spec["len_emb"] = (d,)
    spec["phn_emb.weight"] = (cfg.phoneme_tokens, d)
    spec["text_emb.weight"] = (cfg.text_tokens, d)
    if cfg.langs > 0:
        spec["langs_emb.weight"] = (cfg.langs, d)
    if cfg.tasks > 0:
        spec["tasks_emb.weight"] = (cfg.tasks, d)
    if cfg.tones > 0:
        spec["tones_emb.weight"] = (cfg.tones, d)
    for name, n in (("proms_emb", cfg.audio_tokens), ("resps_emb", cfg.audio_tokens + 2)):      # (+ stop and mask token)
        spec[f"{name}.pos_embedding"] = (1, L, d)
        spec[f"{name}.level_weights"] = (L,)
        for l in range(L):
            spec[f"{name}.embs.{l}.weight"] = (n, d)
        spec[f"{name}.norm.weight"] = (d,)
        spec[f"{name}.norm.bias"] = (d,)
        h = d * e.audio_encoder_ffn_expansion_size
        spec[f"{name}.proj.0.weight"] = (h, d)
        spec[f"{name}.proj.0.bias"] = (h,)
        spec[f"{name}.proj.2.weight"] = (d, h)
        spec[f"{name}.proj.2.bias"] = (d,)
    h = d * e.audio_decoder_ffn_expansion_size
    for l in range(L):
        p = f"audio_decoder.head.{l}."
        spec[p + "0.weight"] = (d,)
        spec[p + "0.bias"] = (d,)
        spec[p + "1.weight"] = (h, d)
        spec[p + "1.bias"] = (h,)
        spec[p + "3.weight"] = (d, h)
        spec[p + "3.bias"] = (d,)
        spec[p + "4.weight"] = (cfg.audio_tokens + 1, d)
        spec[p + "4.bias"] = (cfg.audio_tokens + 1,)
    n_len = 50 if e.len_use_logits else 1
    for name, n in (("len_decoder", n_len), ("phn_decoder", cfg.phoneme_tokens), ("text_decoder", cfg.text_tokens)):
        spec[f"{name}.head.weight"] = (n, d)
        spec[f"{name}.head.bias"] = (n,)
    for l in range(cfg.layers):
        p = f"model.layers.{l}."
        spec[p + "self_attn.q_proj.weight"] = (d, d)
        spec[p + "self_attn.k_proj.weight"] = (d, d)
        spec[p + "self_attn.v_proj.weight"] = (d, d)
        spec[p + "self_attn.o_proj.weight"] = (d, d)
        spec[p + "mlp.gate_proj.weight"] = (ff, d)
        spec[p + "mlp.up_proj.weight"] = (ff, d)
        spec[p + "mlp.down_proj.weight"] = (d, ff)
        spec[p + "input_layernorm.weight"] = (d,)
        spec[p + "post_attention_layernorm.weight"] = (d,)
    if not e.per_level_normalization:
        spec["model.norm.weight"] = (d,)
    return spec


def state_dict_spec(cfg: Model) -> "OrderedDict[str, tuple]":
    """Ordered {name: shape} exactly as ``reference_model.state_dict()`` yields it."""
    if cfg.version >= 7:
        return state_dict_spec_v2(cfg)
    d = cfg.dim
    ff = cfg.dim * cfg.ffn
    L = cfg.resp_levels
    _, emb_tokens, _, cls_tokens = classifier_layout(cfg)
    spec = OrderedDict()
    spec["sep"] = (d,)
    spec["dropout_token"] = (d,)
    spec["text_emb.weight"] = (cfg.phoneme_tokens, d)
    for i in range(L):
        spec[f"proms_emb.embeddings.{i}.weight"] = (cfg.audio_tokens, d)
    for i, n in enumerate(emb_tokens):
        spec[f"resps_emb.embeddings.{i}.weight"] = (n, d)
    if cfg.langs > 0:
        spec["langs_emb.weight"] = (cfg.langs, d)
    if cfg.tasks > 0:
        spec["tasks_emb.weight"] = (cfg.tasks, d)
    if cfg.tones > 0:
        spec["tones_emb.weight"] = (cfg.tones, d)
    spec["rvq_l_emb.weight"] = (L, d)
    spec["len_emb.weight"] = (11, d)
    spec["raw_text_emb.weight"] = (cfg.text_tokens, d)
    for l in range(cfg.layers):
        p = f"model.layers.{l}."
        spec[p + "self_attn.q_proj.weight"] = (d, d)
        spec[p + "self_attn.k_proj.weight"] = (d, d)
        spec[p + "self_attn.v_proj.weight"] = (d, d)
        spec[p + "self_attn.o_proj.weight"] = (d, d)
        spec[p + "mlp.gate_proj.weight"] = (ff, d)
        spec[p + "mlp.up_proj.weight"] = (ff, d)
        spec[p + "mlp.down_proj.weight"] = (d, ff)
        spec[p + "input_layernorm.weight"] = (d,)
        spec[p + "post_attention_layernorm.weight"] = (d,)
    spec["model.norm.weight"] = (d,)
    if cfg.experimental.split_classifiers:
        for k, n in enumerate(cls_tokens):
            spec[f"classifiers.proj.{k}.weight"] = (n, d)
            if cfg.experimental.classifiers_bias:
                spec[f"classifiers.proj.{k}.bias"] = (n,)
    return spec


def _kind(name: str) -> str:
    if name.endswith("layernorm.weight") or name == "model.norm.weight":
        return "norm"
    # version >= 7 names (base_v2.py): encoder / decoder LayerNorms, FFNs, heads, level tables
    if name.endswith("_emb.norm.weight") or (name.startswith("audio_decoder.head.") and name.endswith(".0.weight")):
        return "norm"
    if name.endswith("_emb.norm.bias") or (name.startswith("audio_decoder.head.") and name.endswith(".0.bias")):
        return "head_bias"
    if name.endswith(".pos_embedding"):
        return "pos"
    if name.endswith(".level_weights"):
        return "level_weights"
    if "_emb.proj." in name or name.startswith("audio_decoder.head.") or "_decoder.head." in name:
        if name.endswith(".bias"):
            return "head_bias"
        return "head" if (name.endswith(".4.weight") or "_decoder.head." in name) else "ffn"
    if name.startswith("classifiers.") and name.endswith(".bias"):
        return "head_bias"
    if name.startswith("classifiers."):
        return "head"
    if ".self_attn." in name or ".mlp." in name:
        return "linear"
    return "embedding"


def synth_state_dict(cfg: Model, seed: int = 1234, stop_bias: float | None = None,
                     linear_std: float = 0.03, head_std: float = 0.05) -> "OrderedDict[str, torch.Tensor]":
    """Deterministic fp32 CPU weights: tensor ``idx`` (spec order) is drawn from
    ``torch.Generator().manual_seed(seed + idx)``.

    Scales keep activations O(1) through the stack (so kernel errors are visible
    in logits) and give greedy decisions a healthy top-2 margin.  ``stop_bias``
    overrides the AR head's bias for the stop token (bench: a large negative value
    forces fixed-length generation identically on every implementation).
    """
    out = OrderedDict()
    for idx, (name, shape) in enumerate(state_dict_spec(cfg).items()):
        g = torch.Generator(device="cpu").manual_seed(seed + idx)
        x = torch.randn(shape, generator=g, dtype=torch.float32)
        kind = _kind(name)
        if kind == "norm":
            x = 1.0 + 0.1 * x
        elif kind == "linear":
            x = linear_std * x
        elif kind == "head":
            x = head_std * x
        elif kind == "head_bias":
            x = 0.1 * x
        elif kind == "ffn":
            x = 2.0 * linear_std * x
        elif kind == "pos":
            x = 0.3 * x
        elif kind == "level_weights":
            x = 1.0 / shape[0] ** 0.5 + 0.5 * x
        out[name] = x
    if stop_bias is not None and "classifiers.proj.0.bias" in out:
        out["classifiers.proj.0.bias"][cfg.audio_tokens] = stop_bias
    if stop_bias is not None and cfg.version >= 7:
        for l in range(cfg.resp_levels):
            out[f"audio_decoder.head.{l}.4.bias"][cfg.audio_tokens] = stop_bias
    return out


LORA_TARGETS = ("self_attn.q_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj")


def lora_state_dict_spec(cfg: Model, rank: int) -> "OrderedDict[str, tuple]":
    """{name: shape} of the LoRA tensors the reference's ``apply_lora`` adds under the llama policy
    (vall_e/config.py:519-530: self_attn and mlp linears, k_proj excluded; vall_e/models/lora.py:43-44:
    lora_B (out, rank) then lora_A (rank, in))."""
    base = state_dict_spec(cfg)
    spec = OrderedDict()
    for l in range(cfg.layers):
        for t in LORA_TARGETS:
            out_f, in_f = base[f"model.layers.{l}.{t}.weight"]
            spec[f"model.layers.{l}.{t}.lora_B"] = (out_f, rank)
            spec[f"model.layers.{l}.{t}.lora_A"] = (rank, in_f)
    return spec


def synth_lora_state_dict(cfg: Model, rank: int, seed: int = 4321, std: float = 0.05) -> "OrderedDict[str, torch.Tensor]":
    """Deterministic LoRA tensors for parity tests (a freshly initialised lora_B is zero, which would test nothing)."""
    out = OrderedDict()
    for idx, (name, shape) in enumerate(lora_state_dict_spec(cfg, rank).items()):
        g = torch.Generator(device="cpu").manual_seed(seed + idx)
        out[name] = std * torch.randn(shape, generator=g, dtype=torch.float32)
    return out
