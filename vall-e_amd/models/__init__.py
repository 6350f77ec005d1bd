"""``vall_e.models`` surface: ``get_model(config, training, **kw)`` -> AR_NAR (reference models/__init__.py:63-94)."""
from .ar_nar import AR_NAR
from .base import Base, Logits, Sampled


def get_model(config, training=True, **model_kwargs):
    if getattr(config, "version", 5) >= 7:
        raise NotImplementedError("AR_NAR_V2 (version >= 7) is a different model family; not built")
    kw = dict(
        n_phn_tokens=config.phoneme_tokens, n_audio_tokens=config.audio_tokens, n_text_tokens=config.text_tokens,
        d_model=config.dim, n_heads=config.heads, n_layers=config.layers, n_experts=getattr(config, "experts", 1),
        p_dropout=config.dropout, training=training, config=config,
    )
    kw.update(model_kwargs)
    return AR_NAR(**kw)


__all__ = ["AR_NAR", "Base", "Logits", "Sampled", "get_model"]
