"""Weight files in the reference's format (vall_e/utils/io.py:92-127, written by Engines.export,
vall_e/engines/base.py:356-401): a safetensors file whose tensors are the model's state dict and whose
string metadata holds JSON for ``config`` (the model's config dict incl. ``experimental``), ``stats`` and
``userdata``.  Only loaders that execute nothing from the file are used (safetensors, JSON).
"""
from __future__ import annotations

import json
from pathlib import Path

import torch

from ..config import Model

SFT_SUFFIXES = (".safetensor", ".safetensors", ".sft")


def _jsonable(v):
    return json.dumps(v) if isinstance(v, (dict, list)) else v


def torch_save(data: dict, path, module_key: str | None = None):
    """``data`` = a state dict, or ``{"module": state_dict, "config": {...}, "stats": {...}, ...}``
    (io.py:61-104): the dict of tensors becomes the payload, everything else string metadata."""
    from safetensors.torch import save_file
    path = Path(path)
    if path.suffix not in SFT_SUFFIXES:
        raise NotImplementedError(f"{path.suffix}: only safetensors files are written (pickled .pth files are not)")
    if all(isinstance(v, torch.Tensor) for v in data.values()):
        tensors, metadata = data, {}
    else:
        target = module_key or next((k for k, v in data.items()
                                     if isinstance(v, dict) and v and all(isinstance(t, torch.Tensor) for t in v.values())), None)
        if target is None:
            raise ValueError("no dict of tensors to save")
        tensors = data[target]
        metadata = {k: _jsonable(v) for k, v in data.items() if k != target and v is not None}
    save_file({k: v.contiguous() for k, v in tensors.items()}, str(path),
              metadata={k: str(v) for k, v in metadata.items()})


def torch_load(path, device="cpu", load_metadata=True, module_key="module") -> dict:
    """-> ``{module_key: state_dict, "config": ..., "stats": ..., "userdata": ...}`` (io.py:106-125)."""
    from safetensors import safe_open
    path = Path(path)
    if path.suffix not in SFT_SUFFIXES:
        raise NotImplementedError(f"{path.suffix}: only safetensors files are read; a pickled .pth would execute code")
    state = {}
    with safe_open(str(path), framework="pt", device=str(device)) as f:
        for k in f.keys():
            state[k] = f.get_tensor(k)
        meta = dict(f.metadata() or {}) if load_metadata else {}
    for k, v in list(meta.items()):
        try:
            meta[k] = json.loads(v)
        except (TypeError, ValueError):
            pass
    return {module_key: state} | meta if load_metadata else state


def load_model_file(path):
    """``cfg.load_model`` + ``load_engines`` for one weights file (vall_e/config.py:120-149,
    vall_e/engines/__init__.py:212-248): -> (Model config from the file's ``config`` metadata, state dict)."""
    blob = torch_load(path)
    if "config" not in blob or not isinstance(blob["config"], dict):
        raise ValueError(f"{path}: no model config in the file's metadata (the reference stores it under 'config')")
    cfg = Model.from_dict(blob["config"])
    sd = blob["module"]
    # older exports name the per-level tables `.embs.` (vall_e/config.py:141-145 probes those keys)
    sd = {k.replace("_emb.embs.", "_emb.embeddings."): v for k, v in sd.items()}
    return cfg, sd
