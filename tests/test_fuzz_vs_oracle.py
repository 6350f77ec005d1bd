"""Randomised differential test: the HIP path (fp32, tiny model) against the oracle over option combinations that no
single golden covers together -- batch shape, prompt / text lengths, guidance, prefixes, penalties, filters, early exit,
chunking, LoRA -- greedy, so that results are comparable.  A difference is only accepted when the oracle's own top-2 margin
at the first differing decision is below the fp32 logit error (a near tie); anything else fails."""
import numpy as np
import pytest
import torch

from conftest import model_cfg, synth_weights

pytestmark = pytest.mark.gpu

TIE = 2e-4


def _models(kind, lora):
    import valle_oracle as vo
    from vall_e_amd.models import get_model
    from vall_e_amd.weights import synth_lora_state_dict
    cfg = model_cfg(kind)
    sd = synth_weights(kind)
    m = get_model(cfg, training=False)
    m.load_state_dict(sd)
    o = vo.Oracle(cfg, sd)
    if lora is not None:
        rank, alpha, levels = lora
        t = synth_lora_state_dict(cfg, rank)
        m.load_lora(t, rank=rank, alpha=alpha, rvq_levels=levels)
        o.load_lora(t, rank, alpha, levels)
    return m.to("cuda:0", torch.float32), o


def _first_diff(a, b):
    n = min(len(a), len(b))
    for i in range(n):
        if int(a[i]) != int(b[i]):
            return i
    return n if len(a) != len(b) else -1


@pytest.mark.parametrize("seed", list(range(int(__import__("os").environ.get("VALLE_FUZZ_SEEDS", "48")))))
def test_random_option_mix_matches_oracle(gpu_device, seed):
    from vall_e_amd.synth import synth_batch
    rng = np.random.default_rng(1000 + seed)
    kind = ["tiny", "tiny", "tiny_nc", "tiny_upi", "tiny_nosum"][int(rng.integers(5))]
    lora = (4, 8, [(), (0,), (0, 2), (1, 3)][int(rng.integers(4))]) if kind == "tiny" and rng.random() < 0.4 else None
    m, o = _models(kind, lora)
    B = int(rng.integers(1, 6))
    phns, proms, langs = synth_batch(int(rng.integers(1 << 20)), B, n_phn=int(rng.integers(4, 24)),
                                     n_prom=int(rng.integers(3, 40)), ragged=bool(rng.random() < 0.7))
    steps = int(rng.integers(1, 14))
    kw = {}
    if rng.random() < 0.35:
        kw["cfg_strength"] = float(rng.choice([1.5, 3.0]))
    if rng.random() < 0.3:
        kw["repetition_penalty"] = float(rng.choice([1.1, 1.3])); kw["repetition_penalty_decay"] = float(rng.choice([0.0, 0.25]))
    if rng.random() < 0.25:
        kw["length_penalty"] = float(rng.choice([-0.5, 0.4]))
    if rng.random() < 0.25:
        kw["top_k"] = int(rng.choice([3, 20])); kw["top_no"] = 0.0
    if rng.random() < 0.2:
        kw["top_no"] = 0.0
    mode = rng.random()
    if mode < 0.15:
        kw["input_prompt_prefix"] = True
    elif mode < 0.3:
        kw["prefix_silence"] = float(rng.choice([0.03, 0.1]))
    hip_only = dict(step_chunk=int(rng.choice([1, 3, 32])), early_exit=bool(rng.random() < 0.5))
    desc = f"seed {seed}: {kind} lora={lora} B={B} steps={steps} {kw} {hip_only}"
    got = m(task_list=["tts"] * B, phns_list=phns, proms_list=proms, lang_list=langs, max_duration=steps, temperature=0.0,
            **kw, **hip_only)
    want, logits = o.ar(phns, proms, langs, max_duration=steps, temperature=0.0, return_logits=True, **kw)
    usable = []
    for b in range(B):
        g, w = got[b].cpu().numpy(), want[b].numpy()
        d = _first_diff(g, w)
        if d < 0:
            usable.append(b)
            continue
        # the decision at response index d: the oracle's margin there (guided / filtered logits are not kept: be strict then)
        margin = None
        if not kw and d < len(logits):
            row = logits[d][b]
            v = torch.topk(row.float(), 2).values
            margin = float(v[0] - v[1])
        assert margin is not None and margin < TIE, f"{desc}\nutterance {b} differs at {d}: hip {g[:d+2]} oracle {w[:d+2]} margin {margin}"
    # NAR levels on the utterances whose AR codes agree (same inputs on both sides), non-empty responses only
    idx = [b for b in usable if len(want[b]) > 0]
    if not idx or kw.get("input_prompt_prefix"):
        return
    if next(iter({len(want[b]) for b in idx})) == 8:   # reference quirk (ar_nar.py:927-931): when the element the reference
        return                                         # takes from the SET of last dimensions is n_resp_levels -- a level-0
                                                       # response of exactly 8 frames -- the call reads as training; both sides refuse
    nar_kw = {"cfg_strength": kw["cfg_strength"]} if "cfg_strength" in kw and rng.random() < 0.5 else {}
    sub = lambda l: [l[b] for b in idx]
    full = m(task_list=["tts"] * len(idx), phns_list=sub(phns), proms_list=sub(proms), lang_list=sub(langs),
             resps_list=[got[b].clone() for b in idx], temperature=0.0, **nar_kw)
    ref = o.nar(sub(phns), sub(proms), sub(langs), [want[b] for b in idx], temperature=0.0, **nar_kw)
    bad = 0
    for i in range(len(idx)):
        g, w = full[i].cpu().numpy(), ref[i].numpy()
        assert g.shape == w.shape, desc
        # a flipped near tie at one level changes that row's later levels: count rows, not cells
        bad += int((g != w).any(axis=1).sum())
    total = sum(len(want[b]) for b in idx)
    assert bad <= max(1, total // 50), f"{desc}\n{bad} of {total} NAR rows differ"


@pytest.mark.parametrize("seed", list(range(int(__import__("os").environ.get("VALLE_FUZZ_SEEDS", "48")) // 3)))
def test_random_narlen_mix_matches_oracle(gpu_device, seed):
    """NAR-len demasking (level 0 by iterative demasking, then the NAR levels) under random lengths, step counts, guidance
    floors and score options, causal and non-causal models.  Demasking decisions are lowest-k cuts over scores: a single
    near tie moves a position to another step, so agreement is asked of at least 95 % of the frames."""
    from vall_e_amd.synth import synth_batch
    rng = np.random.default_rng(5000 + seed)
    kind = ["tiny", "tiny_nc", "tiny_nc_upi"][int(rng.integers(3))]
    m, o = _models(kind, None)
    B = int(rng.integers(1, 5))
    phns, proms, langs = synth_batch(int(rng.integers(1 << 20)), B, n_phn=int(rng.integers(4, 20)),
                                     n_prom=int(rng.integers(3, 30)), ragged=bool(rng.random() < 0.7))
    lens = [int(rng.integers(3, 30)) for _ in range(B)]
    lens = [l + 1 if l == 8 else l for l in lens]                       # (8 frames would read as training, ar_nar.py:927-931)
    kw = dict(max_steps=int(rng.integers(2, 9)))
    if rng.random() < 0.4:
        kw["cfg_strength"] = float(rng.choice([1.0, 3.5]))
    if rng.random() < 0.3:
        kw["minimum_cfg_strength"] = float(rng.choice([0.0, 1.5]))
    if rng.random() < 0.3:
        kw["sampling_scores_masked_only"] = True
    if rng.random() < 0.3:
        kw["sampling_scores_remask"] = True
    if rng.random() < 0.3:
        kw["annealed_sampling"] = False
    desc = f"seed {seed}: {kind} B={B} lens={lens} {kw}"
    got = m(task_list=["tts"] * B, phns_list=phns, proms_list=proms, lang_list=langs, len_list=lens, temperature=0.0, **kw)
    lvl0 = o.nar_masked(phns, proms, langs, lens, temperature=0.0, **kw)
    want = o.nar(phns, proms, langs, lvl0, temperature=0.0, **({"cfg_strength": kw["cfg_strength"]} if "cfg_strength" in kw else {}))
    bad = total = 0
    for b in range(B):
        g, w = got[b].cpu().numpy(), want[b].numpy()
        assert g.shape == w.shape == (lens[b], 8), desc
        bad += int((g != w).any(axis=1).sum())
        total += lens[b]
    assert bad <= max(1, total // 20), f"{desc}\n{bad} of {total} frames differ"


# ------------------------------------------------------------------------------ version >= 7 family
def _models_v2(kind, upi):
    import model_configs as vcfg
    import valle_oracle_v2 as vo2
    from vall_e_amd.models import get_model
    from vall_e_amd.weights import synth_state_dict
    cfg = {"tiny_v2": vcfg.tiny_v2_model, "tiny_v2_nc": vcfg.tiny_v2_nc_model}[kind]()
    cfg.experimental.unified_position_ids = upi
    sd = synth_state_dict(cfg, 1234)
    m = get_model(cfg, training=False)
    m.load_state_dict(sd)
    return m.to("cuda:0", torch.float32), vo2.OracleV2(cfg, sd)


@pytest.mark.parametrize("seed", list(range(int(__import__("os").environ.get("VALLE_FUZZ_SEEDS", "48")) // 3)))
def test_random_v2_mix_matches_oracle(gpu_device, seed):
    """The version >= 7 loops under random batch shapes and options against the CPU restatement: greedy AR of 8 codes per step
    (KV cache or whole-sequence schedule, any chunking) -- a difference is accepted only at a near tie of the restatement's own
    logits --, and NAR-len of all levels (guidance floor, masked-only / remask scores, annealing, top-n-sigma on or off; the
    restatement under the lower-index tie rule) on >= 95 % of the frames."""
    from vall_e_amd.synth import synth_batch
    rng = np.random.default_rng(9000 + seed)
    kind = ["tiny_v2", "tiny_v2_nc"][int(rng.integers(2))]
    upi = bool(rng.random() < 0.35)
    m, o = _models_v2(kind, upi)
    B = int(rng.integers(1, 5))
    phns, proms, langs = synth_batch(int(rng.integers(1 << 20)), B, n_phn=int(rng.integers(4, 20)),
                                     n_prom=int(rng.integers(3, 30)), ragged=bool(rng.random() < 0.7))
    kwin = dict(task_list=["tts"] * B, phns_list=phns, proms_list=proms, lang_list=langs)
    # ---- AR
    steps = int(rng.integers(1, 9))
    hip_only = dict(step_chunk=int(rng.choice([1, 3, 16])), kv_cache=bool(rng.random() < 0.7))
    top_no = float(rng.choice([0.0, 1.0]))
    desc = f"seed {seed}: {kind} upi={upi} B={B} steps={steps} top_no={top_no} {hip_only}"
    got = m(**kwin, max_duration=steps, temperature=0.0, top_no=top_no, **hip_only)
    want, logits = o.ar_v2(phns, proms, langs, max_duration=steps, temperature=0.0, top_no=top_no, return_logits=True)
    for b in range(B):
        g, w = got[b].cpu().numpy(), want[b].numpy()
        if g.shape == w.shape and (g == w).all():
            continue
        d = next((i for i in range(min(len(g), len(w))) if (g[i] != w[i]).any()), min(len(g), len(w)))
        v = torch.topk(logits[d][b].float(), 2, dim=-1).values                    # (L, 2) at the first differing frame
        margin = float((v[:, 0] - v[:, 1]).min())
        assert margin < TIE, f"{desc}\nutterance {b} differs at frame {d}, margin {margin}"
    # ---- NAR-len
    lens = [int(rng.integers(3, 24)) for _ in range(B)]
    kw = dict(max_steps=int(rng.integers(2, 8)), top_no=float(rng.choice([0.0, 1.0])))
    if rng.random() < 0.4:
        kw["cfg_strength"] = float(rng.choice([1.0, 3.5]))
    if rng.random() < 0.3:
        kw["minimum_cfg_strength"] = float(rng.choice([0.0, 1.5]))
    if rng.random() < 0.3:
        kw["sampling_scores_masked_only"] = True
    if rng.random() < 0.3:
        kw["sampling_scores_remask"] = True
    if rng.random() < 0.3:
        kw["annealed_sampling"] = False
    if max(kw.get("cfg_strength", 2.5), kw.get("minimum_cfg_strength", 2.5)) > 0:
        # the guidance rescales by std(pos) / std(guided) over the ROWS of a column (samplers.py:187-188): over 3-5 nearly equal
        # values that quotient amplifies the fp32 error of the logits (2e-6) to 1e-3 in the scores -- seen at 3 rows -- and one
        # row gives nan on both sides; the comparison is run where the reference's own arithmetic is well conditioned
        lens = [max(l, 6) for l in lens]
    desc = f"seed {seed}: {kind} upi={upi} B={B} lens={lens} {kw}"
    th, to = [], []
    got = m(**kwin, len_list=lens, temperature=0.0, trace=th, **kw)
    want = o.nar_masked_v2(phns, proms, langs, lens, temperature=0.0, tie_rule="lowest_index", trace=to, **kw)
    bad = total = 0
    for b in range(B):
        g, w = got[b].cpu().numpy(), want[b].numpy()
        assert g.shape == w.shape == (lens[b], 8), desc
        bad += int((g != w).sum())
        total += lens[b] * 8
    if bad <= max(1, total // 20):
        return
    # More than 5 % of the codes differ: legitimate only as the cascade of a NEAR TIE -- a logit within the fp32 error of the
    # guided logits (~1e-5) of the top-n-sigma threshold (one more or one fewer survivor changes the row's score, hence the
    # next step's masks), or two candidates within it of each other.  Found at the first step whose scores differ.
    off = np.concatenate(([0], np.cumsum(lens)))
    for s_ in range(len(th)):
        hid, hsc = th[s_][0].cpu().view(8, -1), th[s_][1].cpu().view(8, -1)
        nears, conds = [], []
        for b in range(B):
            oi, osc, olg = to[s_][0][b], to[s_][1][b], to[s_][2][b]
            a, c = hid[:, off[b]:off[b + 1]], hsc[:, off[b]:off[b + 1]]
            if int((a != oi).sum()) == 0 and float((c - osc).abs().max()) <= 1e-4:
                continue
            w_ = int((c - osc).abs().argmax())
            row = olg[w_ // lens[b], w_ % lens[b]].float()
            v = torch.topk(row, 2).values
            near = float(v[0] - v[1])
            if kw["top_no"] > 0:
                near = min(near, float((row - (row.max() - row.std())).abs().min()))
            nears.append(near)
            # conditioning of the guidance's rescale in the columns that carry the row's probability: |x| / std over the rows
            cols = torch.topk(row, 4).indices
            X = olg[w_ // lens[b]][:, cols].float()
            conds.append(float((X.abs().max(dim=0).values / X.std(dim=0)).max()))
        if nears:                                                       # the step at which the two sides part
            # ... or, under guidance, of an ILL-CONDITIONED rescale: the factor std(pos) / std(guided) is taken over the rows of a
            # column (samplers.py:187-188), and where those rows are nearly equal -- the first step of a noncausal model, every
            # response frame the mask token -- the quotient amplifies the 2e-6 of the fp32 logits by |x| / std (seen: 600 x)
            guided = max(kw.get("cfg_strength", 2.5), kw.get("minimum_cfg_strength", 2.5)) > 0
            assert min(nears) < TIE or (guided and max(conds) > 50), \
                f"{desc}\n{bad} of {total} codes differ; step {s_} departs with no near tie ({min(nears):.2e}) and a well-conditioned rescale ({max(conds):.1f})"
            return
    raise AssertionError(f"{desc}\n{bad} of {total} codes differ although every step's ids and scores agree")
