"""The reference's HF / gguf weight layout: ONE embedding table and ONE ``lm_head`` over a contiguous token space
(``convert_to_hf_llama``, vall_e/export.py:15-232; the same ranges are hard-coded in vall_e.cpp/vall_e.cpp:14-41 and
vall_e/models/base.py:64-90).  Every input table of the model occupies a range of ``model.embed_tokens.weight``, in the
order of ``HF_MAPPING``; the classifier of a range, where it has one, sits on the same rows of ``lm_head.weight``.

``io_ranges`` gives the ranges for a model config, ``from_hf_llama`` cuts an HF-layout state dict back into the
reference's own parameter names (what ``AR_NAR.load_state_dict`` takes), ``to_hf_llama`` is the forward direction
(export.py's, plus the head biases, which export.py leaves at ``nn.Linear``'s random init)."""
from __future__ import annotations

from collections import OrderedDict

import torch

from ..config import Model
from ..weights import classifier_layout, state_dict_spec

# (range name as vall_e.cpp spells it, embedding key, classifier name or None) -- export.py:81-108
HF_MAPPING = (
    [("text", "text_emb.weight", "stt"), ("rvq_l", "rvq_l_emb.weight", None), ("lang", "langs_emb.weight", None),
     ("task", "tasks_emb.weight", None), ("len", "len_emb.weight", "len"), ("tone", "tones_emb.weight", None), ("sep", "sep", None)]
    + [(f"prom|{i}", f"proms_emb.embeddings.{i}.weight", None) for i in range(8)]
    + [("resps|AR:0:0", "resps_emb.embeddings.0.weight", "AR:0:0")]
    + [(f"resps|NAR:{i}:{i + 1}", f"resps_emb.embeddings.{i + 1}.weight", f"NAR:{i}:{i + 1}") for i in range(7)]
    + [("resps|NAR:0:0", "resps_emb.embeddings.8.weight", "NAR:0:0")]
)


def io_ranges(cfg: Model):
    """-> OrderedDict name -> (start, end, embedding key, classifier index or -1), e.g. for the published
    ar+nar-len-llama-8 shape: text (0, 256, head 9), ..., resps|NAR:0:0 (16677, 17702, head 8) -- vall_e.cpp:14-41."""
    spec = state_dict_spec(cfg)
    _, _, cls_names, _ = classifier_layout(cfg)
    out, start = OrderedDict(), 0
    for name, key, head in HF_MAPPING:
        if key not in spec:
            raise KeyError(f"{key}: the HF layout needs this table (model config lacks it)")
        n = 1 if len(spec[key]) == 1 else spec[key][0]
        out[name] = (start, start + n, key, cls_names.index(head) if head in cls_names else -1)
        start += n
    return out


def to_hf_llama(sd: dict, cfg: Model) -> dict:
    rng = io_ranges(cfg)
    n_tokens = max(e for _, e, _, _ in rng.values())
    d = cfg.dim
    emb = torch.zeros((n_tokens, d), dtype=sd["text_emb.weight"].dtype)
    head = torch.zeros((n_tokens, d), dtype=emb.dtype)
    bias = torch.zeros(n_tokens, dtype=emb.dtype)
    has_bias = "classifiers.proj.0.bias" in sd
    for s, e, key, k in rng.values():
        emb[s:e] = sd[key].reshape(e - s, d)
        if k >= 0:
            w = sd[f"classifiers.proj.{k}.weight"]
            head[s:s + w.shape[0]] = w                       # export.py:145-146 (a head may be one row shorter than its table)
            if has_bias:
                bias[s:s + w.shape[0]] = sd[f"classifiers.proj.{k}.bias"]
    out = {k: v.clone() for k, v in sd.items() if k.startswith("model.")}
    out["model.embed_tokens.weight"], out["lm_head.weight"] = emb, head
    if has_bias:
        out["lm_head.bias"] = bias
    return out


def from_hf_llama(hf: dict, cfg: Model) -> dict:
    """HF-layout tensors -> the reference's parameter names.  Tables the layout does not carry (``raw_text_emb``,
    ``dropout_token``: export.py's mapping omits them) come back as zeros: raw-text input needs the native weights file."""
    rng = io_ranges(cfg)
    n_tokens = max(e for _, e, _, _ in rng.values())
    emb, head = hf["model.embed_tokens.weight"], hf["lm_head.weight"]
    if emb.shape[0] != n_tokens or head.shape[0] != n_tokens:
        raise ValueError(f"HF layout with {emb.shape[0]} tokens, the model config gives {n_tokens} "
                         f"(ranges: {[(n, s, e) for n, (s, e, _, _) in rng.items()][:4]} ...)")
    spec = state_dict_spec(cfg)
    _, _, _, cls_tokens = classifier_layout(cfg)
    sd = {k: v for k, v in hf.items() if k.startswith("model.") and k != "model.embed_tokens.weight"}
    for s, e, key, k in rng.values():
        sd[key] = emb[s:e].reshape(spec[key]).clone()
        if k >= 0 and cfg.experimental.split_classifiers:
            n = cls_tokens[k]
            sd[f"classifiers.proj.{k}.weight"] = head[s:s + n].clone()
            if cfg.experimental.classifiers_bias:
                if "lm_head.bias" not in hf:
                    raise KeyError("lm_head.bias: the model config has classifier biases")
                sd[f"classifiers.proj.{k}.bias"] = hf["lm_head.bias"][s:s + n].clone()
    for key, shape in spec.items():
        if key not in sd:
            if key in ("raw_text_emb.weight", "dropout_token"):
                sd[key] = torch.zeros(shape, dtype=emb.dtype)
            else:
                raise KeyError(f"{key}: not recoverable from the HF layout")
    return sd
