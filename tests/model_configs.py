"""Reduced-size model configurations of the parity tests (test infrastructure: the product package ships only the
reference's own configurations, vall_e_amd.config.repo_default_model).  Used by tests/, oracle/gen_golden.py and
__graft_entry__.smoke()."""
from vall_e_amd.config import Model, ModelExperimentalSettings, repo_default_model  # noqa: F401


def tiny_model(dim=256, heads=4, layers=2) -> Model:
    """Reduced-size sibling of the repo default (same switches) for fast parity cases."""
    m = repo_default_model()
    m.size = {"dim": dim, "heads": heads, "layers": layers}
    return m


def tiny_len_model(**kw) -> Model:
    """``tiny_model`` with the ``len`` capability: adds the 11-token duration head (base.py:404-406)."""
    m = tiny_model(**kw)
    m.capabilities = ["ar", "nar", "len"]
    return m


def tiny_tone_model(**kw) -> Model:
    """``tiny_len_model`` with the ``tone`` capability: a tone token between prompt and response (base.py:623-624)."""
    m = tiny_len_model(**kw)
    m.capabilities = ["ar", "nar", "len", "tone"]
    return m


def tiny_upi_model(**kw) -> Model:
    """``tiny_model`` with ``unified_position_ids`` (the reference's dataclass default, config.py:292): one arange over
    the whole sequence instead of per-segment position ids."""
    m = tiny_model(**kw)
    m.experimental.unified_position_ids = True
    return m


def tiny_nobias_model(**kw) -> Model:
    """``tiny_model`` with ``classifiers_bias=False``: heads without a bias vector (base.py:179-199)."""
    m = tiny_model(**kw)
    m.experimental.classifiers_bias = False
    return m


def tiny_nosum_model(**kw) -> Model:
    """``tiny_model`` with the reference's dataclass default ``audio_embedding_sums=False`` (config.py:291; data/config.yaml
    switches it on): a code frame is embedded by its last known level alone (base.py:169-173)."""
    m = tiny_model(**kw)
    m.experimental.audio_embedding_sums = False
    return m


def tiny_sumprom_model(**kw) -> Model:
    """``audio_embedding_sums="prom"``: prompt frames summed over their levels, response frames not (base.py:434, :438)."""
    m = tiny_model(**kw)
    m.experimental.audio_embedding_sums = "prom"
    return m


def tiny_nc_model(**kw) -> Model:
    """Tiny sibling of the published ``ar+nar-len-llama-8`` switches: ``len`` head and ``noncausal_masks``
    (NAR levels attend in both directions, base.py:1230-1233)."""
    m = tiny_len_model(**kw)
    m.experimental.noncausal_masks = True
    return m


def tiny_nc_upi_model(**kw) -> Model:
    """``tiny_nc_model`` with the reference's dataclass default ``unified_position_ids=True``: every switch of a
    checkpoint whose config names only ``noncausal_masks`` and the ``len`` capability."""
    m = tiny_nc_model(**kw)
    m.experimental.unified_position_ids = True
    return m


def tiny_v2_model(dim=256, heads=4, layers=2) -> Model:
    """Tiny sibling of a version-7 model (reference models/base_v2.py, models/ar_nar_v2.py: all eight codebooks at once)
    under the switches of data/config.yaml (per-segment position ids, causal masks) and the dataclass defaults of the
    version >= 7 experimental fields."""
    m = tiny_model(dim, heads, layers)
    m.version = 7
    return m


def tiny_v2_nc_model(**kw) -> Model:
    """``tiny_v2_model`` with ``noncausal_masks``: the demasking passes attend in both directions (base_v2.py:1182-1185)."""
    m = tiny_v2_model(**kw)
    m.experimental.noncausal_masks = True
    return m


def tiny_v2_len_model(**kw) -> Model:
    """``tiny_v2_model`` with ``len_use_logits``: the duration head predicts 5 digits (base_v2.py:467, :1302-1306).  With the
    dataclass default (a single float) the reference's own pass fails reshaping the head's (T, 1) output to (-1, 5, 10)
    (base_v2.py:1283) unless T is a multiple of 50."""
    m = tiny_v2_model(**kw)
    m.experimental.len_use_logits = True
    return m
