"""Import harness for the reference implementation (TEST INFRASTRUCTURE ONLY).

This module exists only in the build container, where ``/root/reference`` is
mounted read-only.  It makes ``vall_e.models.get_model`` importable so that
``oracle/gen_golden.py`` can run the *real* reference on seeded inputs and dump
small golden vectors into ``tests/golden/``.  Nothing here travels to the GPU
box as a dependency: tests only read the committed ``.npz`` fixtures.

What it does:
  * inserts no-op stand-ins for *non-arithmetic* third-party modules the
    reference imports at module scope but never calls on the hot path
    (torchmetrics, h5py, coloredlogs, humanize, torchaudio, soundfile,
    phonemizer) -- none of them compute anything that reaches a golden value;
  * restores the ``ROPE_INIT_FUNCTIONS["default"]`` entry that transformers 5.x
    dropped (the reference targets 4.4x): the standard
    ``inv_freq = 1 / theta ** (arange(0, Dh, 2) / Dh)``, scaling 1.0
    (reference use: vall_e/models/arch/llama.py:95-97);
  * loads ``data/config.yaml`` the way ``vall_e.config`` expects.
"""
from __future__ import annotations

import os
import sys
import types
from pathlib import Path

REFERENCE_ROOT = Path(os.environ.get("VALLE_REFERENCE_ROOT", "/root/reference"))


def reference_available() -> bool:
    return (REFERENCE_ROOT / "vall_e" / "models" / "ar_nar.py").exists()


def _stub(name: str, **attrs):
    mod = types.ModuleType(name)
    mod.__dict__.update(attrs)
    mod.__spec__ = None
    sys.modules[name] = mod
    return mod


def _install_stubs():
    import torch
    import transformers  # noqa: F401  (must be imported before the torchaudio stand-in)
    import transformers.models.llama.modeling_llama  # noqa: F401

    class _Metric(torch.nn.Module):
        def __init__(self, num_classes=0, *a, **k):
            super().__init__()
            self.num_classes = num_classes

        def forward(self, *a, **k):  # never reached in inference
            return torch.tensor(0.0)

    if "torchmetrics" not in sys.modules:
        tm = _stub("torchmetrics")
        tm.classification = _stub(
            "torchmetrics.classification",
            BinaryAccuracy=_Metric, MulticlassAccuracy=_Metric, MulticlassPrecision=_Metric,
        )
    for name in ("h5py", "humanize", "soundfile"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                _stub(name)
    if "coloredlogs" not in sys.modules:
        import logging
        _stub("coloredlogs", ColoredFormatter=logging.Formatter)
    if "torchaudio" not in sys.modules:
        try:
            import torchaudio  # noqa: F401
        except Exception:
            ta = _stub("torchaudio")
            ta.functional = _stub("torchaudio.functional")
            ta.transforms = _stub("torchaudio.transforms")
    if "phonemizer" not in sys.modules:
        try:
            import phonemizer  # noqa: F401
        except Exception:
            ph = _stub("phonemizer", phonemize=None)
            ph.backend = _stub("phonemizer.backend", BACKENDS={})


def _install_rope_shim():
    import torch
    from transformers.modeling_rope_utils import ROPE_INIT_FUNCTIONS

    def _default_rope(config, device=None, seq_len=None, **kw):
        base = getattr(config, "rope_theta", None)
        if base is None:
            rp = getattr(config, "rope_parameters", None) or {}
            base = rp.get("rope_theta", 10000.0)
        head_dim = getattr(config, "head_dim", None) or config.hidden_size // config.num_attention_heads
        inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2, dtype=torch.int64).to(device=device, dtype=torch.float) / head_dim))
        return inv_freq, 1.0

    if "default" not in ROPE_INIT_FUNCTIONS:
        ROPE_INIT_FUNCTIONS["default"] = _default_rope
    return _default_rope


_loaded = None


def load_reference():
    """Returns a namespace with the reference modules needed by the generator."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not reference_available():
        raise RuntimeError(f"reference not mounted at {REFERENCE_ROOT}")
    os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
    sys.dont_write_bytecode = True
    _install_stubs()
    rope_fn = _install_rope_shim()
    if str(REFERENCE_ROOT) not in sys.path:
        sys.path.insert(0, str(REFERENCE_ROOT))
    argv = sys.argv
    sys.argv = [argv[0]]  # vall_e.config parses argv at import time
    try:
        from vall_e.config import cfg
        cfg.load_yaml(REFERENCE_ROOT / "data" / "config.yaml")
        cfg.format(training=False)
        cfg.device = "cpu"
        import vall_e.models as ref_models
        from vall_e.models.arch import llama as ref_llama
        import vall_e.samplers as ref_samplers
        # transformers 5.x post_init() expects this on *RotaryEmbedding modules
        if not hasattr(ref_llama.RotaryEmbedding, "compute_default_rope_parameters"):
            ref_llama.RotaryEmbedding.compute_default_rope_parameters = staticmethod(rope_fn)
    finally:
        sys.argv = argv
    _loaded = types.SimpleNamespace(cfg=cfg, models=ref_models, llama=ref_llama, samplers=ref_samplers)
    return _loaded


def build_reference_model(dtype=None, **overrides):
    """Instantiates the reference AR_NAR from data/config.yaml (fp32, eval)."""
    import torch
    ref = load_reference()
    for k, v in overrides.items():
        if hasattr(ref.cfg.model, k):
            setattr(ref.cfg.model, k, v)
        else:
            setattr(ref.cfg.model.experimental, k, v)
    model = ref.models.get_model(ref.cfg.model, training=False)
    model = model.float().eval()
    if dtype is not None and dtype != torch.float32:
        model = model.to(dtype)
    return model
