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
    # read by the version >= 7 model only (vall_e/config.py:302-332, models/base_v2.py:293-316)
    monolithic_audio_encoder: bool = False
    per_level_normalization: bool = True
    len_use_logits: bool = False
    use_segmented_attention_mask: bool = False
    use_sliding_attention_mask: bool = False
    use_audio_encoder_level_weights: bool = True
    use_audio_encoder_ffn: bool = True
    use_audio_encoder_norm: bool = True
    audio_decoder_ffn_expansion_size: int = 2
    audio_encoder_ffn_expansion_size: int = 2


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
            # Switches of the reference's dataclass (config.py:268-354) that this build does not model are ACCEPTED AND IGNORED,
            # because the reference's version-5 model ignores them too -- a checkpoint that sets one runs with the same
            # semantics there as here:
            #   hf, interleave, logit_normalization, use_segmented_attention_mask, use_sliding_attention_mask: never read by
            #     models/base.py / models/ar_nar.py / models/arch/llama.py (the attention-mask switches belong to base_v2.py:
            #     kept as fields for the version >= 7 model, which refuses them);
            #   tie_classifier_to_embedding, audio_embedding_mode, monolithic_audio_encoder: read into locals of
            #     Base.__init__ (models/base.py:368-369, :381) and not used again (version >= 7 reads the last one);
            #   predict_causally: training only (models/base.py:1033 calc_loss, models/ar_nar.py:182 forward_train);
            #   kv_heads: models/base.py:478 hard-wires num_key_value_heads = n_heads; size.embed_dim: a property of the
            #     dataclass only (config.py:472-476).
            # The switches that DO change the version-5 forward are fields of ModelExperimentalSettings here and are either
            # built (audio_embedding_sums, unified_position_ids, noncausal_masks, classifiers_bias, max_position_embeddings,
            # masking_ratio) or refused by Base.__init__ (split_classifiers=False, causal_size != 1).
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
