"""CPU-only checks of the host side: sequence planning vs the oracle, C-ABI symbols, error paths."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden, golden_inputs, model_cfg, synth_weights
import valle_oracle as vo


def host_model(kind="tiny"):
    from vall_e_amd.models import get_model
    m = get_model(model_cfg(kind), training=False)
    m.load_state_dict(synth_weights(kind))
    return m


def table(m):
    sd = m._sd
    return torch.cat([sd[k].reshape(-1, m.d_model) for _, k in m._table_order])


@pytest.mark.parametrize("q,timestep", [(0, False), (1, False), (3, False), (7, False), (0, True)])
def test_plan_matches_oracle_embeddings_and_positions(q, timestep):
    z = load_golden("tiny_fwd")
    phns, proms, langs = golden_inputs(z)
    m = host_model()
    o = vo.Oracle(model_cfg("tiny"), synth_weights("tiny"))
    g = torch.Generator().manual_seed(q)
    resps = [torch.randint(0, 1024, (5 + 3 * i, max(q, 1)), generator=g) for i in range(2)]
    if q == 0:
        resps = [r[:, 0] for r in resps]
    plan = m.plan(phns, langs, proms, resps, q, timestep=timestep)
    E = table(m)
    off = 0
    for i in range(2):
        p = vo.plan_sequence(phns[i], langs[i], proms[i], resps[i], q, timestep=timestep)
        want = o.embed_sequence(p)
        T = want.shape[0]
        idx = torch.from_numpy(plan.idx[off:off + T]).long()
        got = torch.zeros(T, m.d_model)
        started = torch.zeros(T, dtype=torch.bool)
        for j in range(8):
            sel = idx[:, j] >= 0
            got[sel] = torch.where(started[sel, None], got[sel] + E[idx[sel, j]], E[idx[sel, j]])
            started |= sel
        np.testing.assert_array_equal(got.numpy(), want.numpy())
        np.testing.assert_array_equal(plan.pos[off:off + T], p.pos.numpy())
        assert plan.lens[i] == T and m.cls_names[plan.head] == p.head
        off += T
    assert plan.M == off and plan.cu[-1] == off


def _gathered(m, plan, off, T):
    E = table(m)
    idx = torch.from_numpy(plan.idx[off:off + T]).long()
    got = torch.zeros(T, m.d_model)
    started = torch.zeros(T, dtype=torch.bool)
    for j in range(8):
        sel = idx[:, j] >= 0
        got[sel] = torch.where(started[sel, None], got[sel] + E[idx[sel, j]], E[idx[sel, j]])
        started |= sel
    return got


@pytest.mark.parametrize("case", ["task_prompt", "raw_text", "tone", "empty_prompt"])
def test_plan_variants_match_oracle(case):
    """Host planning of the wider input forms == the oracle's (which the reference goldens pin): [task, codes] prompts
    with their position-id quirk, raw text ids, the tone token, the empty prompt of input_prompt_prefix."""
    from vall_e_amd.models.base import RawText
    z = load_golden("tiny_fwd")
    phns, proms, langs = golden_inputs(z)
    kind = "tiny_tone" if case == "tone" else "tiny"
    m = host_model(kind)
    o = vo.Oracle(model_cfg(kind), synth_weights(kind))
    resps = [torch.arange(4 + i) for i in range(2)]
    tones = None
    if case == "task_prompt":
        proms = [["ns", proms[0]], ["sr", proms[1]]]
    elif case == "raw_text":
        phns = [torch.tensor([1, 700, 33, 2]), torch.tensor([1, 4000, 2])]
    elif case == "empty_prompt":
        proms = [torch.zeros(0, dtype=torch.int16) for _ in range(2)]
    elif case == "tone":
        tones = [torch.tensor([0]), torch.tensor([0])]
    plan = m.plan([RawText(p) for p in phns] if case == "raw_text" else phns, langs, proms, resps, 0, tone_list=tones)
    off = 0
    for i in range(2):
        p = vo.plan_sequence(vo.RawText(phns[i]) if case == "raw_text" else phns[i], langs[i], proms[i], resps[i], 0,
                             tone=None if tones is None else tones[i])
        want = o.embed_sequence(p)
        T = want.shape[0]
        np.testing.assert_array_equal(_gathered(m, plan, off, T).numpy(), want.numpy())
        np.testing.assert_array_equal(plan.pos[off:off + T], p.pos.numpy())
        off += T
    assert plan.M == off
    if case == "task_prompt":
        assert plan.pos[plan.lens[0] - 1] == 1                  # the tail the reference fills with 1 (base.py:988-990)


def test_null_cfg_plan_has_no_prompt():
    m = host_model()
    plan = m.plan([torch.tensor([1, 2])], [torch.tensor([0])], None, [torch.arange(9)], 0)
    assert plan.lens == [2 + 1 + 1 + 9 + 3]      # phn, lang, quant_level, resp + 3 seps (SURVEY: R + 7)
    assert plan.pos.tolist() == [0, 1, 2, 0, 1, 0, 1] + list(range(9))


def test_header_symbols_are_exported_and_typed():
    from vall_e_amd import _lib as L
    lib = L.load()
    header = (ROOT / "include" / "valle_hip.h").read_text()
    declared = set(re.findall(r"\b(valle_[a-z0-9_]+)\s*\(", header))
    declared -= {"valle_model"}
    assert declared, "no declarations found"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/valle_hip.h but not exported"
        assert name in L.SIGNATURES, f"{name} has no ctypes signature"
    assert lib.valle_version() == 1


def test_argument_errors_are_reported_without_a_gpu():
    from vall_e_amd import _lib as L
    lib = L.load()
    rc = lib.valle_embed_sum(None, None, None, 4, 1024, L.F32, None)
    assert rc == -1 and b"null pointer" in lib.valle_last_error()
    rc = lib.valle_gemm(C.c_void_p(16), C.c_void_p(16), None, None, C.c_void_p(16), 4, 8, 100, 8, 0, L.F32, None)
    assert rc == -1 and b"multiple of 32" in lib.valle_last_error()
    with pytest.raises(L.ValleHipError):
        L.check(rc)


def test_product_path_refuses_cpu():
    from vall_e_amd import _lib as L
    m = host_model()
    with pytest.raises(L.ValleHipError, match="no CPU fallback"):
        m(task_list=["tts"], phns_list=[torch.tensor([1, 5, 2])], proms_list=[torch.zeros(4, 8, dtype=torch.int16)],
          lang_list=[torch.tensor([0])], max_duration=2, temperature=0.0)


def test_unsupported_options_fail_loudly():
    m = host_model()
    with pytest.raises(NotImplementedError):
        m.sample_params(beam_width=4)
    m.sample_params(mirostat_tau=3.0, mirostat_eta=0.1)                  # accepted and without effect, as in the reference
    with pytest.raises(NotImplementedError):                            # (ar_nar.py:702-704 vs :804-808); an explicit state list is not
        m.sample_params(mirostat=[{"n": 1024, "tau": 3.0, "eta": 0.1}])
    m.sample_params(prefix_silence=0.5)                                  # a forward_ar option (built); the sampler ignores it
    sp = m.sample_params(temperature=1.0, min_temperature=0.5, dry_multiplier=0.8, dry_allowed_length=1)
    assert (sp.temperature64, sp.min_temperature64, sp.dry_multiplier, sp.dry_allowed_length) == (1.0, 0.5, 0.8, 1)
    sp = m.sample_params(temperature=0.7, min_temperature=0.7004)       # closer than 0.001: plain temperature (base.py:1389)
    assert sp.temperature64 == sp.min_temperature64
    with pytest.raises(NotImplementedError):
        m(task_list=["stt"], phns_list=None, resps_list=[torch.zeros(3, 8)])


def test_state_dict_spec_matches_reference_names():
    spec = host_model("full").state_dict_spec()
    assert sum(int(np.prod(s)) for s in spec.values()) == 237_968_641      # parameters as instantiated
    assert spec["resps_emb.embeddings.0.weight"] == (1025, 1024) and spec["classifiers.proj.9.weight"] == (256, 1024)


def test_weights_file_roundtrip_in_reference_format(tmp_path):
    """A file written the way Engines.export does (state dict + JSON `config`/`stats` metadata,
    vall_e/engines/base.py:356-401, utils/io.py:92-104) loads back into the same config and tensors."""
    import torch
    from model_configs import tiny_model
    from vall_e_amd.models import get_model
    from vall_e_amd.utils.io import load_model_file, torch_load, torch_save
    from vall_e_amd.weights import synth_state_dict
    cfg = tiny_model()
    sd = synth_state_dict(cfg, 5)
    path = tmp_path / "fp32.sft"
    torch_save({"module": sd, "config": cfg.to_dict(), "stats": {"global_step": 3}, "userdata": {}}, path)
    blob = torch_load(path)
    assert blob["stats"] == {"global_step": 3} and blob["config"]["experimental"]["split_classifiers"] is True
    cfg2, sd2 = load_model_file(path)
    assert cfg2.to_dict() == cfg.to_dict()
    assert list(sd2) == sorted(sd) or set(sd2) == set(sd)
    for k in sd:
        assert torch.equal(sd[k], sd2[k])
    get_model(cfg2).load_state_dict(sd2)          # shapes and names accepted
    import pytest
    with pytest.raises(NotImplementedError):
        torch_load(tmp_path / "model.pth")


def test_inputs_structure_matches_reference_layout():
    """Base.inputs builds the reference's (name, value) tuple lists (models/base.py:597-649) and
    plan_from_inputs turns them into the same plan as the keyword form."""
    import numpy as np
    import torch
    from model_configs import tiny_model
    from vall_e_amd.models import get_model
    from vall_e_amd.synth import synth_batch
    m = get_model(tiny_model())
    phns, proms, langs = synth_batch(3, 2, n_phn=6, n_prom=5)
    resps = [torch.randint(0, 1024, (7, 2)), torch.randint(0, 1024, (4, 2))]
    inp = m.inputs(phns_list=phns, proms_list=proms, resps_list=resps, lang_list=langs, task_list=["tts"] * 2, quant_levels=[2, 2])
    assert [n for n, _ in inp[0]] == ["task", "phn", "lang", "quant_level", "prom", "resp", "classifier_level"]
    assert inp[0][0] == ("task", "tts") and inp[0][-1] == ("classifier_level", "NAR:1:2")
    a, b = m.plan_from_inputs(inp), m.plan(phns, langs, proms, resps, 2)
    assert np.array_equal(a.idx, b.idx) and np.array_equal(a.pos, b.pos) and a.head == b.head and a.lens == b.lens
    nl = m.inputs(phns_list=phns, proms_list=proms, resps_list=[r[:, 0] for r in resps], lang_list=langs, time_list=[0.5, 0.5], quant_levels=[0, 0])
    assert nl[1][-1] == ("classifier_level", "NAR:0:0")
    assert m.plan_from_inputs(nl).head == m.cls_names.index("NAR:0:0")


def test_len_task_plan_and_inputs():
    """Task ``len`` (base.py:652-683): layout phn|lang|quant_level|prom|len, head ``len``, the len segment is the
    output one (no trailing +1 in its position ids); embeddings and positions equal the oracle's."""
    from vall_e_amd.models import get_model
    z = load_golden("tiny_len")
    phns, proms, langs = golden_inputs(z)
    ws = int(z["weight_seed"])
    m = get_model(model_cfg("tiny_len"), training=False)
    m.load_state_dict(synth_weights("tiny_len", weight_seed=ws))
    o = vo.Oracle(model_cfg("tiny_len"), synth_weights("tiny_len", weight_seed=ws))
    seqs = [torch.tensor([0, 3]), torch.tensor([0]), torch.tensor([0, 1, 10, 4])]
    plan = m.plan(phns, langs, proms, [None] * 3, 0, task="len", len_list=seqs)
    assert m.cls_names[plan.head] == "len" and m.cls_tokens[plan.head] == 11 and plan.out_len == [2, 1, 4]
    E = table(m)
    off = 0
    for i in range(3):
        p = vo.plan_sequence(phns[i], langs[i], proms[i], None, 0, task="len", len_ids=seqs[i])
        want = o.embed_sequence(p)
        T = want.shape[0]
        idx = plan.idx[off:off + T]
        assert (idx[:, 1:] == -1).all()
        np.testing.assert_array_equal(E[torch.from_numpy(idx[:, 0]).long()].numpy(), want.numpy())
        np.testing.assert_array_equal(plan.pos[off:off + T], p.pos.numpy())
        off += T
    inp = m.inputs(task_list=["len"] * 3, phns_list=phns, proms_list=proms, lang_list=langs, len_list=seqs)
    assert [n for n, _ in inp[0]] == ["task", "phn", "lang", "quant_level", "prom", "len", "classifier_level"]
    assert inp[0][-1] == ("classifier_level", "len")
    again = m.plan_from_inputs(inp)
    assert np.array_equal(again.idx, plan.idx) and np.array_equal(again.pos, plan.pos) and again.head == plan.head
    # base.py:679-681: with resps instead of len tokens the digits of the frame count are encoded, 0-led and 10-ended
    enc = m.inputs(task_list=["len"], phns_list=phns[:1], proms_list=proms[:1], lang_list=langs[:1], resps_list=[torch.zeros(137, 8)])
    assert dict((n, v) for n, v in enc[0] if n != "task")["len"].tolist() == [0, 1, 3, 7, 10]
    # a model without the capability has no len head and refuses the task (base.py:654-655)
    plain = host_model()
    with pytest.raises(Exception, match="len"):
        plain.plan(phns, langs, proms, [None] * 3, 0, task="len", len_list=seqs)


def tokenizer_spec(kat):
    """The tokenizer description (the layout of a Hugging Face tokenizer.json) from the fixture's tables."""
    return {"model": {"type": "BPE", "vocab": kat["vocab"], "merges": kat["merges"], "unk_token": kat["unk_token"]},
            "added_tokens": kat["added_tokens"], "normalizer": None, "pre_tokenizer": None,
            "post_processor": {"type": "TemplateProcessing", "single": kat["template_single"],
                               "special_tokens": kat["template_special_tokens"]}}


def test_phoneme_tokenizer_known_answers():
    """utils/tokenizer.py (BPE restatement) against the ids of the library the reference tokenizes with, on the
    reference's own tokenizer.json; includes the phoneme string bundled with data/qnt.enc (BASELINE configs[0])."""
    import json
    from vall_e_amd.utils.tokenizer import PhonemeTokenizer
    from vall_e_amd.synth import QNT_ENC_PHONEME_IDS
    kat = json.loads((ROOT / "tests" / "golden" / "tokenizer_kat.json").read_text(encoding="utf-8"))
    tok = PhonemeTokenizer(tokenizer_spec(kat))
    assert len(kat["cases"]) == 200
    for text, ids in kat["cases"]:
        assert tok.encode(text) == ids, text
    assert tok.encode("wˈʌns ɐɡˈɛn, ˈɛksələnt wˈɜːk.") == QNT_ENC_PHONEME_IDS
    assert tok.encode(list("ðə kˈæt")) == tok.encode("ðə kˈæt")                  # data.py:640-642 joins phoneme lists
    assert tok.encode("ab", add_special_tokens=False) == tok.encode("ab")[1:-1]
    assert tok.decode(tok.encode("ðə kˈæt"), skip_special_tokens=True) == "ðə kˈæt"
    assert tok.get_vocab()["<bos>"] == 1


def test_tts_text_front_end_ids(tmp_path):
    """TTS.encode_text / encode_lang (inference.py:111-137) on phoneme strings; tensors pass through."""
    from vall_e_amd.inference import TTS
    from vall_e_amd.synth import QNT_ENC_PHONEME_IDS
    tts = TTS.__new__(TTS)
    tts.tokenizer = None
    with pytest.raises(RuntimeError, match="tokenizer"):
        tts.encode_text("ðə")
    import json
    spec = tokenizer_spec(json.loads((ROOT / "tests" / "golden" / "tokenizer_kat.json").read_text(encoding="utf-8")))
    (tmp_path / "tokenizer.json").write_text(json.dumps(spec, ensure_ascii=False), encoding="utf-8")
    tts.load_tokenizer(tmp_path / "tokenizer.json")                                # the file layout the reference ships
    ids = tts.encode_text("wˈʌns ɐɡˈɛn, ˈɛksələnt wˈɜːk.")
    assert ids.dtype == torch.int16 and ids.tolist() == QNT_ENC_PHONEME_IDS
    t = torch.tensor([1, 5, 2])
    assert tts.encode_text(t) is t
    assert tts.encode_lang("de").tolist() == [2] and tts.encode_lang("xx").tolist() == [0]


def test_sequences_beyond_the_rope_table_are_refused():
    """Positions index a RoPE table of max_position_embeddings rows: longer sequences raise instead of reading past it."""
    import numpy as np
    from vall_e_amd.models import get_model
    import model_configs as vcfg
    cfg = vcfg.tiny_model()
    cfg.experimental.max_position_embeddings = 64
    m = get_model(cfg, training=False)
    phn = torch.zeros(70, dtype=torch.uint8)
    with pytest.raises(ValueError, match="max_position_embeddings"):
        m.plan([phn], [torch.tensor([0], dtype=torch.uint8)], [None], [torch.zeros(0, dtype=torch.int64)], 0)


def test_derive_seed_separates_chunks_ranks_and_phases():
    from vall_e_amd.inference import derive_seed
    seen = {derive_seed(5, c, r, ph) for c in range(8) for r in range(8) for ph in range(2)}
    assert len(seen) == 128 and all(0 <= s < 2 ** 63 for s in seen)
    assert derive_seed(5, 1, 0, 0) == derive_seed(5, 1, 0, 0) != derive_seed(6, 1, 0, 0)
    assert derive_seed(None) != derive_seed(None)


def test_from_dict_accepts_the_switches_the_reference_ignores():
    """Experimental switches that the reference's version-5 model never reads (see Model.from_dict for the file:line of each)
    load and change nothing, as there; the two that do change its forward and are not built are refused by the model."""
    from vall_e_amd.config import Model
    from vall_e_amd.models import get_model
    ok = Model.from_dict({"size": "full", "experimental": {"audio_embedding_sums": True, "rvq_levels_p": "equal", "layerskip_r": 2,
                                                           "kv_heads": 0, "tie_classifier_to_embedding": False, "split_classifiers": True}})
    assert ok.experimental.audio_embedding_sums is True
    for ignored in ({"tie_classifier_to_embedding": True}, {"interleave": True}, {"hf": True}, {"audio_embedding_mode": "inclusive"},
                    {"logit_normalization": 1.0}, {"use_sliding_attention_mask": True}, {"kv_heads": 4}, {"predict_causally": True}):
        cfg = Model.from_dict({"size": "full", "experimental": dict(ignored, split_classifiers=True, audio_embedding_sums=True)})
        v5_fields = ("audio_embedding_sums", "unified_position_ids", "causal_size", "noncausal_masks", "classifiers_bias", "split_classifiers")
        assert all(getattr(cfg.experimental, k) == getattr(ok.experimental, k) for k in v5_fields)   # nothing of it reaches the version-5 model
        get_model(cfg)
    Model.from_dict({"size": {"dim": 1024, "heads": 16, "layers": 12, "embed_dim": 512}})
    for refused in ({"split_classifiers": False}, {"split_classifiers": True, "causal_size": 2}):
        with pytest.raises(NotImplementedError):
            get_model(Model.from_dict({"size": "full", "experimental": refused}))


def test_hf_token_ranges_and_round_trip():
    """The HF / gguf layout (export.py:81-108): the ranges for the published ar+nar-len-llama-8 shape are the ones
    vall_e.cpp:14-41 and base.py:64-90 hard-code, and to_hf_llama -> from_hf_llama returns every tensor the layout carries."""
    from vall_e_amd.config import Model, ModelExperimentalSettings
    from vall_e_amd.utils.hf_layout import from_hf_llama, io_ranges, to_hf_llama
    from vall_e_amd.weights import synth_state_dict
    pub = Model(name="ar+nar-len", size="full", tasks=9, langs=6, tones=1, capabilities=["ar", "nar", "len"],
                experimental=ModelExperimentalSettings(audio_embedding_sums=True, split_classifiers=True, noncausal_masks=True))
    r = io_ranges(pub)
    want = {"text": (0, 256, 9), "rvq_l": (256, 264, -1), "lang": (264, 270, -1), "task": (270, 279, -1), "len": (279, 290, 10),
            "tone": (290, 291, -1), "sep": (291, 292, -1), "prom|0": (292, 1316, -1), "prom|7": (7460, 8484, -1),
            "resps|AR:0:0": (8484, 9509, 0), "resps|NAR:0:1": (9509, 10533, 1), "resps|NAR:6:7": (15653, 16677, 7),
            "resps|NAR:0:0": (16677, 17702, 8)}
    for k, (s, e, h) in want.items():
        assert (r[k][0], r[k][1], r[k][3]) == (s, e, h), k
    import model_configs as mc
    cfg = mc.tiny_nc_model()
    cfg.langs = 6
    sd = synth_state_dict(cfg, 5)
    hf = to_hf_llama(sd, cfg)
    assert hf["model.embed_tokens.weight"].shape == hf["lm_head.weight"].shape == (17702, cfg.dim)
    back = from_hf_llama(hf, cfg)
    for k, v in sd.items():
        if k in ("raw_text_emb.weight", "dropout_token"):
            assert not back[k].any()
        else:
            assert torch.equal(back[k], v), k
    from vall_e_amd.models import get_model
    get_model(cfg).load_state_dict(back, strict=True)


def test_qnt_code_helpers_known_answers():
    """vall_e_amd.emb.qnt against outputs of the reference's vall_e/emb/qnt.py (golden qnt_kat, oracle/gen_golden.py)."""
    from conftest import load_golden
    from vall_e_amd.emb import qnt
    z = load_golden("qnt_kat")
    clip = torch.from_numpy(z["clip"])
    for t in z["trim_targets"].tolist():
        np.testing.assert_array_equal(qnt.trim(clip, t).numpy(), z[f"trim_{t}"], err_msg=f"trim {t}")
        np.testing.assert_array_equal(qnt.trim(clip.t().contiguous(), t).numpy(), z[f"trimT_{t}"], err_msg=f"trim^T {t}")
    for t in z["extend_targets"].tolist():
        np.testing.assert_array_equal(qnt.repeat_extend_audio(clip, t).numpy(), z[f"extend_{t}"], err_msg=f"extend {t}")
    for i, s in enumerate(z["silence_seconds"].tolist()):
        got = qnt.get_silence(s)
        assert got.dtype == torch.int16
        np.testing.assert_array_equal(got.numpy().reshape(z[f"silence_{i}"].shape), z[f"silence_{i}"])
    for size in (1, 2):
        np.testing.assert_array_equal(qnt.pad_codes_with_silence(clip, size).numpy(), z[f"pad_{size}"])
    np.testing.assert_array_equal(qnt.post_process(clip.t()[None].contiguous()).numpy(), z["post3"])
    np.testing.assert_array_equal(qnt.post_process(clip).numpy(), z["post2"])
    np.testing.assert_array_equal(qnt.concat_audio(clip[:3], None, clip[10:12]).numpy(), z["concat"])
    with pytest.raises(NotImplementedError):
        qnt.decode(clip)


def test_g2p_hook_language_coercion_and_pinyin_tones():
    """emb/g2p.py:47-66, :118-128 restated; the espeak backend itself is a third-party package (absent): a registered
    callable stands in and TTS.encode_text routes plain text through it (inference.py:129)."""
    from vall_e_amd.emb import g2p
    assert [g2p.coerce_language(l) for l in ("en", "fr", "zh", "zh-cn", "ja", "de", "en-gb")] == \
        ["en-us", "fr-fr", "cmn-latn-pinyin", "cmn-latn-pinyin", "ja", "de", "en-gb"]
    calls = []
    def fake(text, language, punctuation=True, stress=True, strip=True):
        calls.append((text, language))
        return "ni3 hao3" if language == "cmn-latn-pinyin" else "həlˈoʊ"
    g2p.set_backend(fake)
    try:
        assert g2p.encode("你好", language="zh") == "niˊ haoˊ"
        assert g2p.encode("hello", language="en") == "həlˈoʊ" and calls[-1] == ("hello", "en-us")
        with pytest.raises(ValueError):
            g2p.encode("hello")
    finally:
        g2p.set_backend(None)
    if not g2p.has_backend():
        with pytest.raises(RuntimeError, match="no phonemizer backend"):
            g2p.encode("hello", language="en")
