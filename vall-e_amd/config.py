"""Model configuration: the same field names the reference reads.

Mirrors the hot-path-relevant part of the reference's config tree
(vall_e/config.py:267-354 ``ModelExperimentalSettings``, :357-535 ``Model``) so
that a ``data/config.yaml`` written for the reference configures this
implementation unchanged.  Only fields the AR+NAR forward reads are kept.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field, fields
from pathlib import Path


@dataclass
class ModelExperimentalSettings:
    # vall_e/config.py:271-298 (same names, same defaults)
    split_classifiers: bool = False
    audio_embedding_sums: bool | str = False      # True, False, "prom" or "resp" (base.py:434, :438)
    unified_position_ids: bool = True
    causal_size: int = 1
    masking_ratio: float | str = 0.8
    noncausal_masks: bool = False
    classifiers_bias: bool = True
    max_position_embeddings: int = 70 * 65 * 5


@dataclass
class Model:
    # vall_e/config.py:358-379
    name: str = "ar+nar"
    version: int = 5
    size: str | dict | float = "full"
    tasks: int = 8
    langs: int = 1
    tones: int = 1
    arch_type: str = "llama"
    attention: str = "auto"
    dropout: float = 0.1
    capabilities: list = field(default_factory=lambda: ["ar", "nar"])
    experimental: ModelExperimentalSettings = field(default_factory=ModelExperimentalSettings)

    # --- derived sizes: vall_e/config.py:425-508 --------------------------------
    def _size(self, key, default):
        if isinstance(self.size, dict) and key in self.size:
            return self.size[key]
        return default

    @property
    def resp_levels(self) -> int:
        return self._size("resp_levels", 8)

    @property
    def max_levels(self) -> int:
        return self.resp_levels

    @property
    def audio_tokens(self) -> int:
        return self._size("audio_tokens", 1024)

    @property
    def text_tokens(self) -> int:
        return self._size("text_tokens", 8575)

    @property
    def phoneme_tokens(self) -> int:
        return self._size("phoneme_tokens", 256)

    @property
    def dim(self) -> int:
        if isinstance(self.size, dict) and "dim" in self.size:
            return self.size["dim"]
        if isinstance(self.size, float):
            return math.floor(1024 * self.size)
        return {"quarter": 256, "half": 512}.get(self.size, 1024)

    @property
    def heads(self) -> int:
        if isinstance(self.size, dict) and "heads" in self.size:
            return self.size["heads"]
        if isinstance(self.size, float):
            return math.floor(16 * self.size)
        return {"quarter": 4, "half": 8}.get(self.size, 16)

    @property
    def layers(self) -> int:
        if isinstance(self.size, dict) and "layers" in self.size:
            return self.size["layers"]
        return {"double": 24, "extended": 16}.get(self.size, 12)

    @property
    def ffn(self) -> int:
        return self._size("ffn", 4)

    # --- constructors ---------------------------------------------------------
    @classmethod
    def from_dict(cls, d: dict) -> "Model":
        d = dict(d)
        exp = d.pop("experimental", None) or {}
        if isinstance(exp, ModelExperimentalSettings):
            exp_obj = exp
        else:
            known = {f.name for f in fields(ModelExperimentalSettings)}
            # switches of the reference (config.py:268-354) that change what the forward computes and are not built here:
            # a checkpoint that sets one must not load and silently run with other semantics.  Everything else that is
            # dropped below only steers training (loss factors, dropout odds, level sampling).
            probe = cls(size=d.get("size", "full"))
            dim, heads = probe.dim, probe.heads
            unbuilt = {"hf": False, "interleave": False, "audio_embedding_mode": None, "tie_classifier_to_embedding": False,
                       "monolithic_audio_encoder": False, "logit_normalization": 0, "use_segmented_attention_mask": False,
                       "use_sliding_attention_mask": False, "predict_causally": False}
            for k, off in unbuilt.items():
                if k in exp and exp[k] != off and exp[k]:
                    raise NotImplementedError(f"experimental.{k}={exp[k]!r} is not built on the HIP path")
            if exp.get("kv_heads", 0) not in (0, heads, None):
                raise NotImplementedError(f"experimental.kv_heads={exp['kv_heads']}: grouped-query attention is not built (MHA only)")
            if isinstance(d.get("size"), dict) and d["size"].get("embed_dim", dim) != dim:
                raise NotImplementedError(f"size.embed_dim={d['size']['embed_dim']} != dim: separate embedding width is not built")
            exp_obj = ModelExperimentalSettings(**{k: v for k, v in exp.items() if k in known})
        known = {f.name for f in fields(cls)} - {"experimental"}
        return cls(experimental=exp_obj, **{k: v for k, v in d.items() if k in known})

    @classmethod
    def from_yaml(cls, path: str | Path, index: int = 0) -> "Model":
        """Reads ``models[index]`` of a reference-style YAML (vall_e/config.py:992-998)."""
        import yaml
        with open(path, "r", encoding="utf-8") as f:
            tree = yaml.safe_load(f)
        return cls.from_dict(tree["models"][index])

    def to_dict(self) -> dict:
        d = {f.name: getattr(self, f.name) for f in fields(self) if f.name != "experimental"}
        d["experimental"] = {f.name: getattr(self.experimental, f.name) for f in fields(self.experimental)}
        return d


def repo_default_model() -> Model:
    """``data/config.yaml:5-31`` of the reference (the BASELINE.json configs[0] model)."""
    return Model(
        name="ar+nar", size="full", tasks=9, langs=4, tones=1, arch_type="llama", version=5,
        attention="auto", dropout=0.1, capabilities=["ar", "nar"],
        experimental=ModelExperimentalSettings(
            audio_embedding_sums=True, unified_position_ids=False, split_classifiers=True),
    )


@dataclass
class Inference:
    # vall_e/config.py:849-862
    backend: str = "local"
    weight_dtype: str = "float32"
    amp: bool = False
    normalize: bool = False
    batch_size: int = 16
    audio_backends: dict = field(default_factory=dict)
