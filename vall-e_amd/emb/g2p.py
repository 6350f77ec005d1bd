"""Grapheme-to-phoneme hook (reference ``vall_e/emb/g2p.py``).

The reference phonemizes with espeak-ng through the third-party ``phonemizer`` package (neither is on this
machine).  What is its own -- the language coercion (``g2p.py:47-66``) and the pinyin tone remap (``:118-128``)
-- is here; the backend is a callable the caller registers (``set_backend``) or, when ``phonemizer`` can be
imported, the same espeak backend the reference builds (``g2p.py:68-82``).  ``TTS.encode_text`` calls ``encode``
exactly where ``inference.py:129`` does once a backend exists.
"""
from __future__ import annotations

_backend = None
_cached = {}

PINYIN_TONES = {"1": "ˇ", "2": "ˉ", "3": "ˊ", "4": "ˋ", "5": "_"}      # g2p.py:119-125


def set_backend(fn):
    """fn(text, language, punctuation=True, stress=True, strip=True) -> IPA string (``None`` removes it)."""
    global _backend
    _backend = fn


def has_backend():
    if _backend is not None:
        return True
    try:
        import phonemizer  # noqa: F401
        return True
    except Exception:
        return False


def coerce_language(lang):
    """g2p.py:47-66: en -> en-us, fr -> fr-fr, every zh* dialect -> cmn-latn-pinyin."""
    if lang == "en":
        lang = "en-us"
    if lang == "fr":
        return "fr-fr"
    if lang[:2] == "zh":
        return "cmn-latn-pinyin"
    return lang


def _espeak(text, language, punctuation=True, stress=True, strip=True):
    from phonemizer.backend import BACKENDS                                  # g2p.py:68-82
    key = f"{language}_espeak"
    if key not in _cached:
        _cached[key] = BACKENDS["espeak"](language, preserve_punctuation=punctuation, with_stress=stress)
    out = _cached[key].phonemize([text], strip=strip)
    return out[0] if len(out) else ""


def encode(text, language="auto", backend="auto", punctuation=True, stress=True, strip=True):
    """g2p.py:84-130 -> IPA string.  ``language="auto"`` needs ``langdetect`` in the reference; here the caller
    names the language."""
    if language == "auto":
        raise ValueError("language='auto' needs a language detector (langdetect in the reference): name the language")
    language = coerce_language(language)
    fn = _backend
    if fn is None:
        if not has_backend():
            raise RuntimeError("no phonemizer backend: register one with vall_e_amd.emb.g2p.set_backend(fn) "
                               "(the reference uses espeak-ng through the `phonemizer` package, absent here)")
        fn = _espeak
    phonemes = fn(text, language, punctuation=punctuation, stress=stress, strip=strip)
    if not len(phonemes):
        raise Exception(f"Failed to phonemize, received empty string: {text}")          # g2p.py:112-113
    if language == "cmn-latn-pinyin":
        for k, v in PINYIN_TONES.items():
            phonemes = phonemes.replace(k, v)
    return phonemes
