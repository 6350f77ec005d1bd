"""CPU restatement of the reference's AR+NAR forward path  (TEST INFRASTRUCTURE ONLY).

This file is the parity oracle.  It is written from scratch in plain PyTorch (CPU) and
restates, function by function, what the reference computes on this path; it is pinned by
the golden vectors in tests/golden/ that oracle/gen_golden.py produced by running the real
reference in the build container (tests/test_oracle_vs_golden.py).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it; the product
(vall_e_amd) never does and fails loudly if its HIP library is missing.

Layout differences from the reference are deliberate (this is a restatement, not a copy):
utterances are processed as a token-packed matrix (sum(T_i), d) -- linear layers run on the
packed matrix, attention per utterance -- instead of a right-padded (B, Tmax, d) tensor with
an additive (B,1,T,T) mask.  Under the reference's all-causal masks (base.py:1233) a valid
row never sees a padded key, so both give the same values.

Reference map (all paths relative to /root/reference/vall_e):
  segment order / classifier level     models/base.py:560-735   -> ``plan_sequence``
  embedding sums, sep join             models/base.py:101-110, :147-175, :774-922 -> ``embed_sequence``
  per-segment position ids             models/base.py:949-996   -> ``plan_sequence`` (pos)
  RMSNorm / RoPE / attention / GEGLU   models/arch/llama.py:48-132, :224-477, :479-566 -> ``Oracle.stack``
  classifier heads                     models/base.py:179-231   -> ``Oracle.head``
  sampler chain                        models/base.py:1280-1430, samplers.py -> ``sample`` and helpers
  AR loop / NAR levels / NAR-len       models/ar_nar.py:588-890, :466-586, :233-464
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F

TASK_SYMMAP = {"tts": 0, "tts-c": 1, "ns": 2, "sr": 3, "tse": 4, "soe": 5, "mask": 6, "eoe": 7, "stt": 8}  # data.py:670-688


# ------------------------------------------------------------------ sequence planning
@dataclass
class Segment:
    kind: str            # phn | lang | quant_level | prom | resp | len | task
    ids: torch.Tensor    # (n,) or (n, levels) integer codes


@dataclass
class Plan:
    segments: list
    head: str            # classifier level name
    quant_level: int
    pos: torch.Tensor    # (T,) int64 position ids
    out_len: int         # length of the output segment


# data.py:670-688 (get_task_symmap): rows of tasks_emb
TASK_SYMMAP = {"tts": 0, "tts-c": 1, "ns": 2, "sr": 3, "tse": 4, "soe": 5, "mask": 6, "eoe": 7, "stt": 8,
               "len": 0, "nse": 6, "cse": 6, "phn": 0, "un-phn": 0}


def _seg_rows(ids):
    """Embedding rows of a segment; a prompt may be a list of code tensors and task-token strings (base.py:850-856)."""
    if isinstance(ids, list):
        return sum(1 if isinstance(p, str) else int(p.shape[0]) for p in ids)
    return int(ids.shape[0])


def _seg_tokens(ids):
    """What the reference's position ids count (base.py:964-974): in a list prompt only the tensors, so every
    task token leaves the position list one short and the tail is filled with 1 (base.py:988-990)."""
    if isinstance(ids, list):
        return sum(int(p.shape[0]) for p in ids if not isinstance(p, str))
    return int(ids.shape[0])


class RawText:
    """Marks an id tensor as raw text (the reference's ``text_list``: ids of the text tokenizer embedded by
    ``raw_text_emb`` in place of the phonemes, base.py:609-610, :841-842; inference.py:518-519 with phonemize=False)."""

    def __init__(self, ids):
        self.ids = torch.as_tensor(ids).long()


def plan_sequence(phn, lang, prom, resp, quant_level, *, timestep=False, task="tts", unified_pos=False, len_ids=None, tone=None):
    """Ordered segments of one utterance + per-segment position ids.

    TTS order: phn, lang, quant_level, prom, resp (base.py:605-649); every segment but the
    last is followed by one ``sep`` row (base.py:101-110), which the position ids of the
    non-output segments count as one extra position (base.py:984).  The ``len`` task
    (base.py:652-683) has the same order with the length digits in place of resp and its own head."""
    if task not in ("tts", "len"):
        raise NotImplementedError("oracle covers the tts layout (AR, NAR, NAR-len) and the len task")
    segs = []
    if isinstance(phn, RawText):
        segs.append(Segment("text", phn.ids))
    elif phn is not None:
        segs.append(Segment("phn", phn.long()))
    if lang is not None:
        segs.append(Segment("lang", lang.long()))
    segs.append(Segment("quant_level", torch.tensor([quant_level])))
    head = "AR:0:0" if quant_level == 0 else f"NAR:{quant_level-1}:{quant_level}"
    if prom is not None:
        if isinstance(prom, (list, tuple)):                               # [task token, codes] (inference.py:489-493)
            segs.append(Segment("prom", [p if isinstance(p, str) else p.long() for p in prom if p is not None]))
        else:
            segs.append(Segment("prom", prom.long()))
    if tone is not None:                                                  # base.py:623-624, :672-673 (capability "tone")
        segs.append(Segment("tone", tone.long()))
    if timestep:
        head = "NAR:0:0"
    if task == "len":
        head = "len"
        if len_ids is not None:
            segs.append(Segment("len", len_ids.long()))
    elif resp is not None:
        segs.append(Segment("resp", resp.long()))
    out_kind = "len" if task == "len" else "resp"                      # task_outputs, base.py:67-73
    pos = []
    for s in segs:
        pos.append(torch.arange(_seg_tokens(s.ids) + (0 if s.kind == out_kind else 1)))
    pos = torch.cat(pos)
    T = sum(_seg_rows(s.ids) for s in segs) + len(segs) - 1
    if segs[-1].kind != out_kind:   # no output segment: the trailing +1 has no sep row behind it
        pos = pos[:T]
    if pos.shape[0] < T:            # task tokens are rows the position list did not count: tail filled with 1
        pos = torch.cat([pos, torch.ones(T - pos.shape[0], dtype=pos.dtype)])
    if unified_pos:
        pos = torch.arange(T)
    assert pos.shape[0] == T, (pos.shape, T)
    return Plan(segs, head, quant_level, pos, _seg_rows(segs[-1].ids) if segs[-1].kind == out_kind else 0)


# ------------------------------------------------------------------------- samplers
def repetition_penalize(logits, previous, factor=1.0, decay=0.0):
    """samplers.py:16-50 (limit/one_time unused on this path).  In place, like the reference."""
    if factor == 1.0 or previous is None:
        return logits
    n = previous.shape[0]
    is_nar = n == logits.shape[0]
    for i in range(n):
        tok = int(previous[i])
        distance = n - i
        start = i + 1 if (is_nar and i < logits.shape[0]) else None
        logits[start:, tok] /= factor * (distance ** decay)
    return logits


def length_penalize(logits, length, factor=0.0, token=-1):
    """samplers.py:56-61."""
    if factor == 0.0:
        return logits
    logits[:, token] /= (length ** factor)
    return logits


def top_k_top_p_filter(logits, top_k=0, top_p=1.0):
    """samplers.py:93-125 (min_tokens=1).  In place."""
    if top_k > 0:
        k = min(max(top_k, 1), logits.size(-1))
        kth = torch.topk(logits, k)[0][..., -1, None]
        logits[logits < kth] = -float("inf")
    if top_p < 1.0:
        s_logits, s_idx = torch.sort(logits, descending=True)
        cum = torch.cumsum(F.softmax(s_logits, dim=-1), dim=-1)
        remove = cum > top_p
        remove[..., 1:] = remove[..., :-1].clone()
        remove[..., 0] = False
        remove = remove.scatter(1, s_idx, remove)
        logits[remove] = -float("inf")
    return logits


def min_p_filter(logits, min_p=0.0, keep=32):
    """samplers.py:74-90."""
    if min_p <= 0.0:
        return logits
    probs = torch.softmax(logits, dim=-1)
    thr = min_p * probs.max(dim=-1, keepdim=True).values
    order = torch.argsort(logits, descending=True, dim=-1)
    rem = torch.gather(probs < thr, -1, order)
    rem[..., :keep] = False
    rem = rem.scatter(1, order, rem)
    return logits.masked_fill(rem, -float("inf"))


def top_n_sigma(logits, n=1.0):
    """samplers.py:164-172 (unbiased std)."""
    m = logits.max(dim=-1, keepdim=True).values
    sd = torch.std(logits, dim=-1, keepdim=True)
    return torch.where(logits >= m - n * sd, logits, torch.full_like(logits, -float("inf")))


def dynamic_temperature(logits, temperature=1.0, min_temperature=0.0, k=10, center=0.5):
    """samplers.py:128-141.  Per row: p_max = 1 / sum(exp(logit - max)) (the differences are fp32, the
    exponentials and their sum Python doubles), temperature slides from ``temperature`` (flat row) to
    ``min_temperature`` (peaked row) on a sigmoid of p_max; the row is divided by it in place."""
    for i in range(logits.shape[0]):
        diffs = (logits[i] - torch.max(logits[i])).tolist()
        sum_exp = 0.0
        for dlt in diffs:
            sum_exp += math.exp(dlt)
        p_max = 1.0 / sum_exp
        t = temperature - (temperature - min_temperature) / (1 + math.exp(-k * (p_max - center)))
        logits[i] /= t
    return logits


def dry_sampling(logits, previous=None, factor=0.0, base=1.75, allowed_length=2):
    """samplers.py:248-275 AS IT RUNS on tensor tokens: the ``lengths`` dict is keyed by 0-dim tensors, which
    hash by identity, so every POSITION of ``previous`` keeps its own entry (no per-token max), in position
    order; the penalty loop ``break``s at the first entry shorter than ``allowed_length`` -- position 0 always
    has length 1, so with the default allowed_length=2 the processor changes nothing."""
    if factor == 0.0 or previous is None:
        return logits
    prev = [int(t) for t in previous]
    n = len(prev)
    entries = []
    for i in range(n):
        length = 1
        while length < max(allowed_length, 50):
            j = i - length
            if j < 0:
                break
            if prev[j] != prev[n - length - 1]:
                break
            length += 1
        entries.append((prev[i], length))
    for tok, length in entries:
        if length < allowed_length:
            break
        logits[:, tok] -= factor * base ** (length - allowed_length)
    return logits


def cfg_combine(pos_logits, null_logits, strength, lens, rescale=0.0):
    """samplers.py:177-191: in place on the last ``len`` rows of each positive logit tensor.
    For 2-D logits ``dims=()`` makes std a GLOBAL unbiased std over rows x vocab."""
    for i, n in enumerate(lens):
        pos = pos_logits[i][-n:, :]
        neg = null_logits[i][-n:, :]
        mix = neg + (pos - neg) * strength
        if rescale > 0:
            mix = mix * (rescale * (pos.std() / mix.std()) + (1 - rescale))
        pos_logits[i][-n:, :] = mix
    return pos_logits


def mirostat_sample(logits, state, generator=None, forced_index=None):
    """samplers.py:197-238 on one utterance's processed logits (rows, V): the LAST row is sorted, k comes from the
    Zipf-exponent estimate over the 100 leading probability ratios (Python float arithmetic on the fp32 softmax), the
    token is drawn from the softmax of the k + 1 largest logits, the state dict is updated in place and returned.
    forced_index (tests): the position in the sorted order to take instead of drawing."""
    def compute_k(prob, n, tau):
        num = den = 0.0
        for i in range(100):
            b = prob[i] / prob[i + 1]
            t = (i + 2) / (i + 1)
            num += math.log(b) * math.log(t)
            den += math.log(t) ** 2
        s_hat = num / den
        eps = s_hat - 1
        return round(((eps * (2 ** tau)) / (1 - n ** (-eps))) ** (1 / s_hat))
    state.setdefault("max_surprise", state["tau"] * 2)
    state.setdefault("error_surprise", 0)
    state.setdefault("running_total_surprise", 0)
    sorted_logits, sorted_indices = torch.sort(logits[-1, :], descending=True)
    prob_original = torch.softmax(sorted_logits, dim=-1).tolist()
    k = compute_k(prob_original, state["n"], state["max_surprise"]) + 1
    prob_topk = torch.softmax(sorted_logits[0:k], dim=0)
    i = torch.multinomial(prob_topk, num_samples=1, replacement=True, generator=generator) if forced_index is None \
        else torch.tensor([forced_index])
    state["index_surprise"] = math.log2(1 / prob_original[i])
    state["running_total_surprise"] += state["index_surprise"]
    state["error_surprise"] = state["index_surprise"] - state["tau"]
    state["max_surprise"] -= state["eta"] * state["error_surprise"]
    state["token"] = sorted_indices[i]
    state["k"] = k                                                          # (not in the reference's dict: for the tests)
    return state


# ---------------------------------------------------------------------------- entropix (samplers.py:281-566)
LN_2 = 0.69314718056                                                       # samplers.py:278

# samplers.py:364-414 (EntropixSamplerConfig): thresholds, coefficients and clamps of the five branches
ENTROPIX = dict(
    min_p=0.01, low_ent_thresh=0.1, low_vent_thresh=0.1, med_ent_thresh=3.0, high_ent_thresh=5.0, high_vent_thresh=5.0,
    helv_attn_ent_offset=1.3, helv_attn_ent_coef=0.2, lehv_interaction_strength_offset=1.2, lehv_interaction_strength_coef=0.3,
    hehv_attn_ent_coef=0.2, hehv_attn_vent_offset=2.0, hehv_attn_vent_coef=0.5, n_adaptive_samples=5,
    ada_temp_logits=0.3, ada_temp_attn=0.2, ada_temp_agree=0.2, ada_top_p=0.1, ada_top_k_int=0.3, ada_top_k_agree=0.2,
    ada_min_p=0.5, temperature_max=1.25, temperature_min=0.5, top_k_min=1, top_k_max=1024, top_p_min=0.1, top_p_max=1.0,
    min_p_min=0.01, min_p_max=0.5)


def entropix_metrics(last_logits, last_scores):
    """samplers.py:281-321 as sample_entropix calls it (:482): last_logits (V,) = the newest row's logits, last_scores
    (H, T) = the LAST layer's raw attention scores q.k / sqrt(Dh) of that row (what Attention.forward returns under
    output_attentions: the scaled products BEFORE mask and softmax, arch/llama.py:429, :469-477).  The scores are
    soft-maxed here (:295); entropy over keys per head, its unbiased variance over heads, the mean absolute deviation
    of the heads from their mean, the mean |score|."""
    log_probs = F.log_softmax(last_logits[None, :], dim=-1)
    probs = torch.exp(log_probs)
    entropy = -torch.sum(probs * log_probs, dim=-1) / LN_2
    varentropy = torch.sum(probs * (log_probs / LN_2 + entropy.unsqueeze(-1)) ** 2, dim=-1)
    sc = last_scores[None, :, None, :]                                       # (1, H, 1, T)
    ap = F.softmax(sc, dim=-1)
    attn_entropy = -torch.sum(ap * torch.log2(torch.clamp(ap, 1e-10, 1.0)), dim=-1)     # (1, H, 1)
    attn_varentropy = torch.var(attn_entropy, dim=1)
    attn_varentropy = torch.where(torch.isnan(attn_varentropy), torch.zeros_like(attn_varentropy), attn_varentropy)
    mean_attention = torch.mean(ap, dim=1)
    agreement = torch.mean(torch.abs(ap - mean_attention.unsqueeze(1)), dim=(1, 2))
    interaction = torch.mean(torch.abs(sc), dim=(1, 2, 3))
    return {"logits_entropy": torch.mean(entropy).item(), "logits_varentropy": torch.mean(varentropy).item(),
            "attn_entropy": torch.mean(attn_entropy).item(), "attn_varentropy": torch.mean(attn_varentropy).item(),
            "agreement": torch.mean(agreement).item(), "interaction_strength": interaction.item(), "action": -1}


def entropix_pick(logit, temperature, top_k, top_p, min_p, noise=None):
    """_sample_entropix (samplers.py:419-462) on the newest row: probabilities at the clamped temperature, the top_k
    largest in ASCENDING order, those whose smaller neighbours already hold more than top_p dropped (:452), then
    argmax(p / q), q ~ Exponential(1).  The min_p block (:441-444) edits ``logit`` after ``probs`` was taken from it and
    so changes nothing.  noise: q for the ascending positions (tests); None draws from torch's global stream exactly as the
    reference does."""
    C = ENTROPIX
    if top_k == 0:
        top_k = logit.shape[-1]
    temperature = clamp(float(temperature), C["temperature_min"], C["temperature_max"])
    top_p = clamp(float(top_p), C["top_p_min"], C["top_p_max"])
    top_k = clamp(int(top_k), C["top_k_min"], C["top_k_max"])
    probs = F.softmax(logit / temperature, dim=-1)
    top_probs, top_idx = torch.topk(probs, k=min(top_k, probs.shape[-1]))
    ps, pi = torch.flip(top_probs, dims=[-1]), torch.flip(top_idx, dims=[-1])
    csum = torch.cumsum(ps, dim=-1)
    ps = ps * (1 - torch.where(csum - ps > top_p, torch.tensor(1.0), torch.tensor(0.0)))
    ps = ps / torch.sum(ps, dim=-1, keepdim=True)
    q = torch.distributions.exponential.Exponential(1.0).sample(ps.shape) if noise is None else noise[: ps.shape[-1]]
    nxt = torch.argmax(ps / q, dim=-1, keepdim=True)
    return torch.take_along_dim(pi, nxt, dim=-1)


def sample_entropix(logits, last_scores, temperature=1.0, top_k=27, top_p=1.0, min_p=0.0, noise=None):
    """sample_entropix (samplers.py:464-566) for one utterance: logits (T, V) of every row, last_scores (H, T) as in
    entropix_metrics.  Returns (token tensor (1,), metrics dict with the branch taken in "action").  Branch 0 of the
    reference calls ``logits[-1, :].argmax(dim=1)`` on a 1-D row and dies of an IndexError (:492): reproduced.  The
    adaptive branch scores its five draws by sum over ALL rows of log_softmax[row, token] plus a constant (:527-545).
    noise: list of per-draw q vectors (tests)."""
    C = ENTROPIX
    m = entropix_metrics(logits[-1], last_scores)
    ent, vent = m["logits_entropy"], m["logits_varentropy"]
    a_ent, a_vent, agree, inter = m["attn_entropy"], m["attn_varentropy"], m["agreement"], m["interaction_strength"]
    nz = (lambda i: None) if noise is None else (lambda i: noise[i])
    last = logits[-1, :]
    if ent < C["low_ent_thresh"] and vent < C["low_vent_thresh"]:
        m["action"] = 0
        raise IndexError("Dimension out of range (expected to be in range of [-1, 0], but got 1)")
    elif ent > C["high_ent_thresh"] and vent < C["low_vent_thresh"]:
        m["action"] = 1
        temperature *= C["helv_attn_ent_offset"] + C["helv_attn_ent_coef"] * a_ent
        res = entropix_pick(last, temperature, top_k, top_p, min_p, nz(0))
    elif ent < C["high_ent_thresh"] and vent > C["high_vent_thresh"]:
        m["action"] = 2
        temperature *= C["lehv_interaction_strength_offset"] + C["lehv_interaction_strength_coef"] * inter
        top_k = max(5, int(top_k * (1 + 0.5 * (1 - agree))))
        res = entropix_pick(last, temperature, top_k, top_p, min_p, nz(0))
    elif ent > C["med_ent_thresh"] and vent > C["high_vent_thresh"]:
        m["action"] = 3
        temperature *= C["hehv_attn_vent_offset"] + C["hehv_attn_vent_coef"] * a_vent
        top_p = max(0.5, top_p - C["hehv_attn_ent_coef"] * a_ent)
        res = entropix_pick(last, temperature, top_k, top_p, min_p, nz(0))
    else:
        m["action"] = 4
        log_softmax = F.log_softmax(logits, dim=-1)
        lu, au = ent + vent, a_ent + a_vent
        temperature *= 1 + C["ada_temp_logits"] * lu + C["ada_temp_attn"] * au - C["ada_temp_agree"] * agree
        top_p = top_p * (1 + C["ada_top_p"] * a_vent)
        top_k = round(float(top_k * (1 + C["ada_top_k_int"] * inter - C["ada_top_k_agree"] * agree)))
        min_p = C["min_p"] * (1 - C["ada_min_p"] * lu)
        samples = [entropix_pick(last, temperature, top_k, top_p, min_p, nz(i)) for i in range(C["n_adaptive_samples"])]
        scores = [torch.sum(log_softmax[:, int(smp[0])]) for smp in samples]     # (+ one constant for all five)
        res = samples[int(torch.argmax(torch.stack(scores)))]
    return res, m


def sample(logits, prev_list=None, quant_levels=None, *, causal_size=1, stop_token=1024, generator=None,
           temperature=1.0, min_temperature=-1.0, top_k=-100, top_p=1.0, min_p=0.0,
           repetition_penalty=1.0, repetition_penalty_decay=0.0, length_penalty=0.0, top_no=1.0,
           dry_multiplier=0.0, dry_base=1.75, dry_allowed_length=2, beam_width=0, mirostat=None, **_unused):
    """Base.sample (base.py:1280-1430), basic branch and beam branch (no mirostat / entropix).
    Returns (ids, processed_logits, scores)."""
    if min_temperature < 0:
        min_temperature = temperature
    if prev_list is not None:
        prev_list = [p if p.dim() == 1 else p[:, -1] for p in prev_list]
    if quant_levels is not None:
        logits = [l[-len(p):] for l, p in zip(logits, prev_list)]          # :1354-1356
    else:
        logits = [l[-causal_size:] for l in logits]                          # :1358-1360
    # ban_tokens(stop) for NAR is a no-op: ``logits.shape[-1] >= token`` always holds (samplers.py:67)
    if prev_list is not None and repetition_penalty != 1.0:
        logits = [repetition_penalize(l, p, repetition_penalty, repetition_penalty_decay) for l, p in zip(logits, prev_list)]
    if quant_levels is None and prev_list is not None and length_penalty != 0.0:
        logits = [length_penalize(l, len(p) + 1, length_penalty, stop_token) for l, p in zip(logits, prev_list)]
    if min_p > 0.0:
        logits = [min_p_filter(l, min_p) for l in logits]
    if top_k > 0 or top_p < 1.0:
        logits = [top_k_top_p_filter(l, top_k, top_p) for l in logits]
    if abs(temperature - min_temperature) >= 0.001:                         # :1389-1390
        logits = [dynamic_temperature(l, temperature, min_temperature) for l in logits]
    elif temperature > 0.0:
        logits = [l / temperature for l in logits]
    if top_no > 0.0:
        logits = [top_n_sigma(l) for l in logits]
    if dry_multiplier > 0.0 and prev_list is not None:                       # :1399-1400
        logits = [dry_sampling(l, p, dry_multiplier, dry_base, dry_allowed_length) for l, p in zip(logits, prev_list)]
    if mirostat is not None:                                                 # base.py:1404-1407
        states = [mirostat_sample(l, st, generator) for l, st in zip(logits, mirostat)]
        return [st["token"] for st in states], logits, states
    if beam_width > 1:                                                       # base.py:1412-1415 with samplers.py:148-160: the
        flat = torch.cat(logits)                                             # beam_width best (row, token) pairs of ALL rows,
        V = flat.shape[-1]                                                   # best first; candidate i goes to sequence i
        idx = torch.topk(flat.flatten(), beam_width).indices.tolist()
        ids = [torch.tensor([i % V]) for i in idx]
        return ids, logits, [flat[i // V, i % V] for i in idx]
    if temperature <= 0.0:
        ids = [l.argmax(dim=-1) for l in logits]
    else:
        ids = []
        for l in logits:
            p = torch.softmax(l.float(), dim=-1)
            ids.append(torch.multinomial(p, 1, generator=generator).squeeze(-1))
    scores = [F.softmax(l, dim=-1).gather(1, t.unsqueeze(-1)).squeeze(-1) for l, t in zip(logits, ids)]
    return ids, logits, scores


def convert_kwargs(kwargs, prefix):
    """utils/utils.py:46-56."""
    out = dict(kwargs)
    for k in list(out):
        if k.startswith(prefix):
            out[k[len(prefix):]] = out.pop(k)
    return out


def clamp(n, lo, hi):
    return max(lo, min(n, hi))


# ---------------------------------------------------------------------------- model
class Oracle:
    def __init__(self, cfg, state_dict, dtype=torch.float32, threads=None):
        from vall_e_amd.weights import classifier_layout
        self.cfg = cfg
        self.dtype = dtype
        self.w = {k: v.to(dtype) for k, v in state_dict.items()}
        self.d = cfg.dim
        self.H = cfg.heads
        self.Dh = cfg.dim // cfg.heads
        self.L = cfg.layers
        self.n_levels = cfg.resp_levels
        self.stop_token = cfg.audio_tokens
        self.mask_token = cfg.audio_tokens
        self.causal_size = cfg.experimental.causal_size
        self.unified_pos = cfg.experimental.unified_position_ids
        self.emb_names, _, self.cls_names, self.cls_tokens = classifier_layout(cfg)
        self.eps = 1e-6
        v = cfg.experimental.audio_embedding_sums                  # True / False / "prom" / "resp" (base.py:434, :438)
        self.sums = {"proms_emb": v is True or v == "prom", "resps_emb": v is True or v == "resp"}
        self.theta = 10000.0
        self.inv_freq = 1.0 / (self.theta ** (torch.arange(0, self.Dh, 2, dtype=torch.int64).float() / self.Dh))
        if threads:
            torch.set_num_threads(threads)

    # -- LoRA (models/lora.py:19-75; switched per RVQ level, lora.py:183-191 via ar_nar.py:268-269, :539-540, :619-620) --
    lora = None          # (tensors in the reference's names, rank, alpha, rvq_levels)
    lora_on = False
    use_lora = None      # the call's override

    def load_lora(self, tensors, rank, alpha, rvq_levels=()):
        self.lora = ({k: v.to(self.dtype) for k, v in tensors.items()}, rank, alpha, tuple(rvq_levels))

    def set_lora_level(self, level):
        if self.lora is None:
            self.lora_on = False
        elif self.use_lora is not None:
            self.lora_on = bool(self.use_lora)
        else:
            self.lora_on = (not self.lora[3]) or (level in self.lora[3])        # config.py:556-559

    def _lin(self, x, name):
        """LoRALinear.forward (lora.py:67-75): the base product plus (x A^T B^T) * alpha / rank when switched on."""
        r = F.linear(x, self.w[name + ".weight"])
        if self.lora_on and (name + ".lora_A") in self.lora[0]:
            t, rank, alpha, _ = self.lora
            r = r + (x @ t[name + ".lora_A"].transpose(0, 1) @ t[name + ".lora_B"].transpose(0, 1)) * (alpha / rank)
        return r

    # -- embeddings (base.py:774-922) ---------------------------------------------
    tones = None     # tone_list of the current call (every pass of a call carries the same list)

    def _audio_sum(self, table_prefix, first, codes, n):
        acc = None
        for l in (range(n) if self.sums[table_prefix] else (n - 1,)):                  # sums=False: the last level alone (base.py:169-173)
            e = self.w[f"{table_prefix}.embeddings.{first + l}.weight"][codes[:, l]]
            acc = e if acc is None else acc + e
        return acc

    def embed_segment(self, seg, plan):
        q = plan.quant_level
        ids = seg.ids
        if seg.kind == "phn":
            return self.w["text_emb.weight"][ids]
        if seg.kind == "text":
            return self.w["raw_text_emb.weight"][ids]
        if seg.kind == "lang":
            return self.w["langs_emb.weight"][ids]
        if seg.kind == "quant_level":
            return self.w["rvq_l_emb.weight"][ids]
        if seg.kind == "len":
            return self.w["len_emb.weight"][ids]
        if seg.kind == "tone":
            return self.w["tones_emb.weight"][ids]
        if seg.kind == "prom" and isinstance(ids, list):                             # base.py:780-794, :850-856
            parts = []
            for p in ids:
                if isinstance(p, str):
                    parts.append(self.w["tasks_emb.weight"][torch.tensor([TASK_SYMMAP[p]])])
                else:
                    parts.append(self.embed_segment(Segment("prom", p), plan).reshape(-1, self.d))
            return torch.cat(parts, dim=0)
        if seg.kind == "prom":
            if ids.dim() == 1:
                return self.w["proms_emb.embeddings.0.weight"][ids]
            return self._audio_sum("proms_emb", 0, ids, 1 if q == 0 else q)          # base.py:790-794
        if seg.kind == "resp":
            if plan.head == "NAR:0:0":                                               # base.py:869-874
                col = ids if ids.dim() == 1 else ids[:, q]
                return self.w[f"resps_emb.embeddings.{self.emb_names.index('NAR:0:0')}.weight"][col]
            if q == 0:                                                               # base.py:893-899
                col = ids if ids.dim() == 1 else ids[:, 0]
                return self.w["resps_emb.embeddings.0.weight"][col]
            if ids.dim() == 1:
                ids = ids[:, None]
            return self._audio_sum("resps_emb", 1, ids, q)
        raise ValueError(seg.kind)

    def embed_sequence(self, plan):
        parts = []
        sep = self.w["sep"][None]
        for i, seg in enumerate(plan.segments):
            if i:
                parts.append(sep)
            parts.append(self.embed_segment(seg, plan).reshape(-1, self.d))
        return torch.cat(parts, dim=0)

    # -- transformer stack (arch/llama.py) ---------------------------------------------
    def rms_norm(self, x, w):
        xf = x.float()
        xf = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + self.eps)
        return w * xf.to(x.dtype)

    def rope(self, pos):
        freqs = pos.float()[:, None] * self.inv_freq[None, :]
        emb = torch.cat((freqs, freqs), dim=-1)
        return emb.cos().to(self.dtype), emb.sin().to(self.dtype)

    @staticmethod
    def _rot(x, cos, sin):
        h = x.shape[-1] // 2
        rot = torch.cat((-x[..., h:], x[..., :h]), dim=-1)
        return x * cos + rot * sin

    def stack(self, x, pos, lens, kv=None, layer_probe=None, causal=None, score_probe=None, final_norm=True, kv_lens=None):
        """x: packed (sum(lens), d); pos: packed position ids; lens: rows per utterance.
        kv: optional list (per layer) of per-utterance [K, V] caches that are appended to
        (KV-cache mode); attention then runs each new row against cache + new rows.
        causal: per-utterance flags (default all True); False = every row sees every row of its own
        utterance (``noncausal_masks``, arch/llama.py:594-614, :837-841)."""
        cos, sin = self.rope(pos)
        cos, sin = cos[:, None, :], sin[:, None, :]
        offs = [0]
        for n in lens:
            offs.append(offs[-1] + n)
        scale = 1.0 / math.sqrt(self.Dh)
        for l in range(self.L):
            p = f"model.layers.{l}."
            h = self.rms_norm(x, self.w[p + "input_layernorm.weight"])
            q = self._lin(h, p + "self_attn.q_proj").view(-1, self.H, self.Dh)
            k = self._lin(h, p + "self_attn.k_proj").view(-1, self.H, self.Dh)
            v = self._lin(h, p + "self_attn.v_proj").view(-1, self.H, self.Dh)
            q, k = self._rot(q, cos, sin), self._rot(k, cos, sin)
            outs = []
            for b, n in enumerate(lens):
                qb = q[offs[b]:offs[b + 1]].transpose(0, 1)
                kb = k[offs[b]:offs[b + 1]].transpose(0, 1)
                vb = v[offs[b]:offs[b + 1]].transpose(0, 1)
                past = 0
                if kv is not None:
                    if kv[l][b] is not None:
                        past = kv[l][b][0].shape[1]
                        kb = torch.cat([kv[l][b][0], kb], dim=1)
                        vb = torch.cat([kv[l][b][1], vb], dim=1)
                    kv[l][b] = [kb, vb]
                Tk = kb.shape[1]
                mask = torch.ones(n, Tk, dtype=torch.bool)
                if causal is None or causal[b]:
                    mask = mask.tril(diagonal=past)
                if kv_lens is not None:                                      # key-padding mask of a right-padded batch: the rows
                    mask[:, kv_lens[b]:] = False                             # behind kv_lens[b] are queries, never keys
                att = (qb @ kb.transpose(1, 2)) * scale
                if score_probe is not None and l == self.L - 1:              # output_attentions: the raw scaled products of the
                    score_probe.append(att[:, -1, :].float().clone())        # newest row, (H, T) (arch/llama.py:429, :469-477)
                att = att.masked_fill(~mask, float("-inf"))
                att = torch.softmax(att.float(), dim=-1).to(qb.dtype)
                outs.append((att @ vb).transpose(0, 1).reshape(n, self.d))
            a = torch.cat(outs, dim=0)
            x = x + self._lin(a, p + "self_attn.o_proj")
            h = self.rms_norm(x, self.w[p + "post_attention_layernorm.weight"])
            g = F.gelu(self._lin(h, p + "mlp.gate_proj"))
            u = self._lin(h, p + "mlp.up_proj")
            x = x + self._lin(g * u, p + "mlp.down_proj")
            if layer_probe is not None and l == 0:
                layer_probe["layer0"] = x
        if final_norm:                                               # (the version >= 7 model moves it into its decoder)
            x = self.rms_norm(x, self.w["model.norm.weight"])
        if layer_probe is not None:
            layer_probe["final"] = x
        return x

    def head(self, x, name):
        k = self.cls_names.index(name)
        return F.linear(x, self.w[f"classifiers.proj.{k}.weight"], self.w.get(f"classifiers.proj.{k}.bias"))

    def is_causal(self, head):
        """base.py:1230-1233: with ``noncausal_masks`` only the AR / len / phn levels keep the causal mask."""
        if not self.cfg.experimental.noncausal_masks:
            return True
        return head in ("phn", "len") or head.startswith("AR:")

    # -- one forward over a batch of plans: list of (T_i, V) logits (Base.forward) --------------
    def forward(self, plans, rows=None, probe=None, score_probe=None):
        xs = [self.embed_sequence(p) for p in plans]
        lens = [x.shape[0] for x in xs]
        hid = self.stack(torch.cat(xs), torch.cat([p.pos for p in plans]), lens, layer_probe=probe,
                         causal=[self.is_causal(p.head) for p in plans], score_probe=score_probe)
        out, o = [], 0
        for i, (p, n) in enumerate(zip(plans, lens)):
            hi = hid[o:o + n]
            o += n
            if rows is not None:
                hi = hi[-rows[i]:] if rows[i] > 0 else hi[:0]
            out.append(self.head(hi, p.head))
        return out

    def _plans(self, phns, langs, proms, resps, q, timestep=False):
        B = len(resps)
        tones = self.tones if "tone" in self.cfg.capabilities else None  # set by the caller for the duration of a call
        return [plan_sequence(None if phns is None else phns[i], None if langs is None else langs[i],
                              None if proms is None else proms[i], resps[i], q, timestep=timestep,
                              unified_pos=self.unified_pos, tone=None if tones is None else tones[i]) for i in range(B)]

    # -- duration prediction (ar_nar.py:640-692, len task) ------------------------------------
    def predict_len(self, phns, proms, langs, *, return_logits=False, use_lora=None):
        """<= 10 greedy steps over the 11-token ``len`` head, sequences seeded with digit 0.  Like the
        reference it keeps appending to utterances that already stopped until ALL have, and the final
        integer drops every stop token (10) but keeps digits that came after one."""
        B = len(phns)
        stop = 10
        self.use_lora = use_lora
        self.set_lora_level(0)                                           # ar_nar.py:619-620 (the len task runs behind it)
        seqs = [torch.tensor([0]) for _ in range(B)]
        stopped = [False] * B
        kept = []
        for _ in range(10):
            tones = self.tones if "tone" in self.cfg.capabilities else None
            plans = [plan_sequence(phns[i], None if langs is None else langs[i], None if proms is None else proms[i],
                                   None, 0, task="len", unified_pos=self.unified_pos, len_ids=seqs[i],
                                   tone=None if tones is None else tones[i]) for i in range(B)]
            logits = self.forward(plans, rows=[1] * B)
            if return_logits:
                kept.append(torch.stack([l[-1].float() for l in logits]))
            for i in range(B):
                tok = int(logits[i][-1:].argmax(dim=1)[0])
                tok = min(tok, stop)
                stopped[i] = stopped[i] or tok == stop
                seqs[i] = torch.cat([seqs[i], torch.tensor([tok])])
            if all(stopped):
                break
        out = [int("".join(str(int(t)) for t in s if int(t) != stop)) for s in seqs]
        return (out, kept, seqs) if return_logits else out

    # -- AR (ar_nar.py:588-890, tts task) ---------------------------------------------
    def ar(self, phns, proms, langs, *, max_duration=500, use_cache=False, return_logits=False, generator=None,
           prefix_context=None, **kw):
        """prefix_context = (texts, resps, lens) of earlier lines (ar_nar.py:736-742): their phonemes are spliced in
        front of this line's (dropping the inner <eos><bos>), their level-0 codes start the sequence, and the first
        lens[i] tokens are cut off the result (:886-888)."""
        kw = convert_kwargs(dict(kw), "ar_")
        temperature = kw.get("temperature", 1.0)
        cfg_strength = kw.get("cfg_strength", 0.0)
        cfg_rescale = kw.pop("cfg_rescale", 0.7)
        input_prompt_prefix = kw.pop("input_prompt_prefix", False)
        refine_on_stop = kw.pop("refine_on_stop", False)
        prefix_silence = kw.pop("prefix_silence", 0.0)
        beam_width = kw.get("beam_width", 0)
        entropix = kw.pop("entropix_sampling", False)                        # ar_nar.py:782, :807; base.py:1325-1342
        entropix_noise = kw.pop("entropix_noise", None)                      # tests: [step][utterance] -> list of q vectors
        self.entropix_log = []                                               # the metrics dicts, per step and utterance
        B = len(phns)
        self.use_lora = kw.pop("use_lora", None)
        self.set_lora_level(0)                                           # ar_nar.py:619-620
        seqs = [torch.zeros(0, dtype=torch.int64) for _ in range(B)]
        start = [0] * B
        if input_prompt_prefix:                                              # ar_nar.py:727-729: the prompt's level-0 codes
            start = [int(p.shape[0]) for p in proms]                         # open the response, the prompt segment is empty
            seqs = [p[:, 0].long() for p in proms]
            proms = [torch.zeros(0, dtype=torch.int64) for _ in proms]
        elif prefix_silence > 0:                                             # ar_nar.py:730-732 with emb/qnt.py:586-593: floor(s * 75)
            n_sil = math.floor(prefix_silence * 75)                          # frames of EnCodec silence, level 0 = code 62, kept
            seqs = [torch.full((n_sil,), 62, dtype=torch.int64) for _ in range(B)]   # in the result (start_slice stays 0)
        if prefix_context is not None:
            prefix_text, prefix_resps, prefix_lens = prefix_context
            phns = [torch.cat([p[:-1].long(), t[1:].long()]) for p, t in zip(prefix_text, phns)]
            seqs = [(r if r.dim() == 1 else r[:, 0]).long() for r in prefix_resps]
        stopped = [False] * B
        null_phn = [RawText([1, 2]) if isinstance(phns[i], RawText) else torch.tensor([1, 2]) for i in range(B)]   # ar_nar.py:787-788
        last_logits = []
        kv = [[None] * B for _ in range(self.L)] if use_cache else None
        for n in range(max_duration // max(1, self.causal_size)):
            if use_cache and cfg_strength > 0:
                raise NotImplementedError
            if use_cache:
                logits = self._forward_cached(phns, langs, proms, seqs, kv, first=(n == 0))
            else:
                logits = self.forward(self._plans(phns, langs, proms, seqs, 0), rows=None if refine_on_stop else [len(s) + 1 for s in seqs])
            if cfg_strength > 0:
                null = self.forward(self._plans(null_phn, langs, None, seqs, 0), rows=[len(s) + 1 for s in seqs])
                logits = cfg_combine(logits, null, cfg_strength, [len(s) + 1 for s in seqs], cfg_rescale)
            if return_logits:
                last_logits.append(torch.stack([l[-1].float() for l in logits]))
            if entropix:
                # the whole sequence's logits and the last layer's raw scores go to sample_entropix, nothing else of the
                # sampler chain runs (base.py:1327-1342)
                probe = []
                full = self.forward(self._plans(phns, langs, proms, seqs, 0), score_probe=probe)
                got = [sample_entropix(full[i].float(), probe[i], kw.get("temperature", 1.0), kw.get("top_k", -100),
                                       kw.get("top_p", 1.0), kw.get("min_p", 0.0),
                                       None if entropix_noise is None else entropix_noise[n][i]) for i in range(B)]
                ids = [g[0] for g in got]
                self.entropix_log.append([g[1] for g in got])
            else:
                ids, _, _ = sample(logits, prev_list=seqs, causal_size=self.causal_size, stop_token=self.stop_token,
                                   generator=generator, **kw)
            if beam_width > 0 and B == 1:                                    # ar_nar.py:821-832: first step, the batch becomes the
                B = beam_width                                               # beam (the language list is NOT expanded there: with
                if langs is not None and B > 1:                              # one the reference dies of an IndexError next step)
                    raise IndexError("list index out of range")
                phns, proms, seqs, start = phns * B, proms * B, seqs * B, start * B
                null_phn, stopped = null_phn * B, [False] * B
            for i in range(len(ids)):
                if self.stop_token in ids[i]:
                    stopped[i] = True
                seqs[i] = torch.cat([seqs[i], ids[i].long()])
            if all(stopped):
                break
        if beam_width:                                                       # ar_nar.py:865-867: "always candidate 0"
            seqs = seqs[:1]
        out = []
        for i, s in enumerate(seqs):
            hit = (s == self.stop_token).nonzero()
            s = s if len(hit) == 0 else s[: int(hit.min())]
            s = s[start[i]:]                                                  # ar_nar.py:872
            if refine_on_stop:                                               # ar_nar.py:874-883: greedy re-read of the LAST
                n = int(s.shape[0])                                          # iteration's logits, rows counted from its end
                s = logits[i][-n - 1:-1].argmax(dim=-1)
            out.append(s[int(prefix_context[2][i]):] if prefix_context is not None else s)
        return (out, last_logits) if return_logits else out

    def _forward_cached(self, phns, langs, proms, seqs, kv, first):
        """KV-cache step: prefill everything on the first call, afterwards only the newest row."""
        plans = self._plans(phns, langs, proms, seqs, 0)
        if first:
            xs = [self.embed_sequence(p) for p in plans]
            pos = [p.pos for p in plans]
        else:
            xs = [self.embed_sequence(p)[-1:] for p in plans]
            pos = [p.pos[-1:] for p in plans]
        lens = [x.shape[0] for x in xs]
        hid = self.stack(torch.cat(xs), torch.cat(pos), lens, kv=kv)
        out, o = [], 0
        for p, n in zip(plans, lens):
            out.append(self.head(hid[o + n - 1:o + n], p.head))
            o += n
        return out

    # -- NAR levels (ar_nar.py:466-586) ---------------------------------------------
    def nar(self, phns, proms, langs, resps, *, generator=None, **kw):
        kw = convert_kwargs(dict(kw), "nar_")
        max_levels = kw.get("max_levels", 0) or (self.cfg.max_levels - 1)
        cfg_strength = kw.get("cfg_strength", 0.0)
        cfg_rescale = kw.pop("cfg_rescale", 0.7)
        self.use_lora = kw.pop("use_lora", None)
        B = len(resps)
        prev = [r[:, None] if r.dim() == 1 else r for r in resps]
        prev = [p.long() for p in prev]
        null_phn = [RawText([1, 2]) if isinstance(phns[i], RawText) else torch.tensor([1, 2]) for i in range(B)]
        for _ in range(max_levels):
            level = prev[0].shape[-1]
            if level >= max_levels + 1:
                break
            R = [p.shape[0] for p in prev]
            # quirk: the level loop hands only phns_list to inputs() (ar_nar.py:544-551), so raw text (text_list) is
            # absent from the conditional pass of every NAR level; the null pass of the guidance does get it (:560-562)
            cond = [None if isinstance(p, RawText) else p for p in phns]
            self.set_lora_level(level)                                   # ar_nar.py:539-540
            logits = self.forward(self._plans(cond, langs, proms, prev, level), rows=R)
            if cfg_strength > 0:
                null = self.forward(self._plans(null_phn, langs, None, prev, level), rows=R)
                logits = cfg_combine(logits, null, cfg_strength, R, cfg_rescale)
            ids, _, _ = sample(logits, prev_list=prev, quant_levels=[level] * B, stop_token=self.stop_token,
                               generator=generator, **kw)
            prev = [torch.cat([p, t[:, None].long()], dim=-1) for p, t in zip(prev, ids)]
        return prev

    # -- NAR-len level 0 demasking (ar_nar.py:233-464) --------------------------------
    def nar_masked(self, phns, proms, langs, lens, *, generator=None, **kw):
        kw = convert_kwargs(dict(kw), "ar_")
        min_len = kw.pop("min_duration", 1)
        max_len = kw.pop("max_duration", 500)
        max_steps = kw.get("max_steps", 25)
        annealed = kw.get("annealed_sampling", True)
        temperature = kw.pop("temperature", 0.0)
        min_cfg = kw.get("minimum_cfg_strength", 2.5)
        cfg_strength = max(kw.get("cfg_strength", min_cfg), min_cfg)
        cfg_rescale = kw.pop("cfg_rescale", 0.75)
        start, end = kw.get("denoise_start", 0.0), kw.get("denoise_end", 1.0)
        max_steps = math.floor(max_steps * (end - start))
        remask = kw.get("sampling_scores_remask", False)
        masked_only = kw.pop("sampling_scores_masked_only", False)
        lens = [clamp(int(l), min_len, max_len) for l in lens]
        B = len(lens)
        prefix = [torch.zeros(0, dtype=torch.int64) for _ in range(B)]
        prefix_context = kw.pop("prefix_context", None)
        if prefix_context is not None:                                   # ar_nar.py:321-325, :391-392
            phns = [torch.cat([p[:-1].long(), t[1:].long()]) for p, t in zip(prefix_context[0], phns)]
            prefix = [(r if r.dim() == 1 else r[:, 0]).long() for r in prefix_context[1]]
        self.use_lora = kw.pop("use_lora", None)
        self.set_lora_level(0)                                           # ar_nar.py:262-269
        resps = [torch.full((n,), self.mask_token, dtype=torch.int64) for n in lens]
        scores = [torch.ones(n, dtype=torch.float32) for n in lens]
        given = kw.pop("resps_list", None)
        stable_ties = start > 0.0 and given is not None
        if stable_ties:                                                  # ar_nar.py:328-334: denoise an existing sequence; the
            import random                                                # 0 / 1 scores come from Python's global ``random``
            resps = [(r if r.dim() == 1 else r[:, 0]).long()[:n] for r, n in zip(given, lens)]
            noise_p0 = math.cos(start * math.pi * 0.5)
            scores = [torch.tensor([0.0 if random.random() < noise_p0 else 1.0 for _ in range(n)], dtype=torch.float32) for n in lens]
        null_phn = [RawText([1, 2]) if isinstance(phns[i], RawText) else torch.tensor([1, 2]) for i in range(B)]
        for step, t in enumerate(torch.linspace(start, end, max_steps)):
            noise_p = math.cos(t * math.pi * 0.5)
            mask_p = noise_p + (1.0 / (max_steps * 2) if remask else 0)
            prev = resps
            if stable_ties:
                # (the reference's topk leaves the order among tied scores to the torch build; the 0 / 1 start scores of this
                # mode are all ties, so this restatement -- like the HIP kernel -- breaks them by the lower index: NOT pinned)
                idx = [torch.sort(s, stable=True).indices[: clamp(int(mask_p * n), 1, n - step)] for s, n in zip(scores, lens)]
            else:
                idx = [s.topk(clamp(int(mask_p * n), 1, n - step), dim=-1, largest=False).indices for s, n in zip(scores, lens)]
            resps = [r.scatter(0, i, self.mask_token) for r, i in zip(resps, idx)]
            is_masked = [r == self.mask_token for r in resps]
            temp = temperature * (1.0 - float(t)) if annealed else temperature
            inp = [torch.cat([p, r]) for p, r in zip(prefix, resps)]
            logits = self.forward(self._plans(phns, langs, proms, inp, 0, timestep=True), rows=lens)
            if cfg_strength > 0:
                null = self.forward(self._plans(null_phn, langs, None, inp, 0, timestep=True), rows=lens)
                logits = cfg_combine(logits, null, cfg_strength, lens, cfg_rescale)
            # the reference samples twice on the SAME tensors; in-place processors (rep-pen,
            # top-k/top-p) therefore act twice before the "unfiltered" score pass
            ids, _, _ = sample(logits, prev_list=prev, quant_levels=[0] * B, stop_token=self.stop_token,
                               generator=generator, temperature=temp, **kw)
            _, _, sc = sample(logits, prev_list=prev, quant_levels=[0] * B, stop_token=self.stop_token,
                              temperature=0.0, **kw)
            resps = [torch.where(m, i, r) for m, i, r in zip(is_masked, ids, resps)]
            scores = [s.float() for s in sc]
            if masked_only:                                              # ar_nar.py:459-460: kept tokens score 1.0
                scores = [torch.where(m, s, torch.ones_like(s)) for m, s in zip(is_masked, scores)]
        return resps

    def nar_len(self, phns, proms, langs, lens, **kw):
        """AR_NAR.forward with len_list (ar_nar.py:485-496 then the level loop)."""
        level0 = self.nar_masked(phns, proms, langs, lens, **dict(kw))
        kw.pop("prefix_context", None)                                   # the level loop runs on this line alone
        return self.nar(phns, proms, langs, level0, **kw)
