"""``vall_e.models`` surface: ``get_model(config, training, **kw)`` -> AR_NAR, or AR_NAR_V2 for version >= 7
(reference models/__init__.py:53-94)."""
from .ar_nar import AR_NAR
from .ar_nar_v2 import AR_NAR_V2
from .base import Base, Logits, Sampled
from .base_v2 import Base_V2


def get_model(config, training=True, **model_kwargs):
    cls = AR_NAR_V2 if getattr(config, "version", 5) >= 7 else AR_NAR           # models/__init__.py:53-60
    kw = dict(
        n_phn_tokens=config.phoneme_tokens, n_audio_tokens=config.audio_tokens, n_text_tokens=config.text_tokens,
        d_model=config.dim, n_heads=config.heads, n_layers=config.layers, n_experts=getattr(config, "experts", 1),
        p_dropout=config.dropout, training=training, config=config,
    )
    kw.update(model_kwargs)
    return cls(**kw)


__all__ = ["AR_NAR", "AR_NAR_V2", "Base", "Base_V2", "Logits", "Sampled", "get_model"]
