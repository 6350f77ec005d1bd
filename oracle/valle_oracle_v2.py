"""CPU restatement of the reference's version >= 7 model  (TEST INFRASTRUCTURE ONLY, like valle_oracle.py).

The "v2" family (reference vall_e/models/base_v2.py, vall_e/models/ar_nar_v2.py) predicts all RVQ levels of a frame at
once: code frames (T, 8) enter through an AudioEncoder, the same llama stack runs without its final norm, and an
AudioDecoder turns every hidden row into 8 x 1025 logits.  Pinned by goldens made by the real reference
(oracle/gen_golden.py:scen_tiny_v2 -> tests/golden/tiny_v2_*.npz, tests/test_oracle_vs_golden.py).

Reference map (paths relative to /root/reference/vall_e):
  sequence layout phn | lang | prom | [tone] | resp, sep between segments     models/base_v2.py:580-723, :61-65
  AudioEncoder (tables + level position, LayerNorm, residual FFN, weighted sum) models/base_v2.py:86-159
  per-segment position ids                                                     models/base_v2.py:826-873
  causal / non-causal per classifier level                                     models/base_v2.py:1181-1185
  AudioDecoder (per level: LayerNorm, FFN, head)                               models/base_v2.py:164-225
  sampler of this family                                                       models/base_v2.py:1335-1411
  guidance on (levels, rows, vocab) logits                                     samplers.py:177-191
  NAR-len demasking of all levels / AR of 8 codes per step                     models/ar_nar_v2.py:207-372, :433-589
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from valle_oracle import Oracle, clamp, convert_kwargs, top_n_sigma


class OracleV2(Oracle):
    def __init__(self, cfg, state_dict, dtype=torch.float32, threads=None):
        self.cfg, self.dtype = cfg, dtype
        self.w = {k: v.to(dtype) for k, v in state_dict.items()}
        self.d, self.H, self.Dh, self.L = cfg.dim, cfg.heads, cfg.dim // cfg.heads, cfg.layers
        self.n_levels = cfg.resp_levels
        self.stop_token = cfg.audio_tokens
        self.mask_token = cfg.audio_tokens + 1                         # base_v2.py:378-379
        self.causal_size = cfg.experimental.causal_size
        self.unified_pos = cfg.experimental.unified_position_ids
        self.eps = 1e-6
        self.inv_freq = 1.0 / (10000.0 ** (torch.arange(0, self.Dh, 2, dtype=torch.int64).float() / self.Dh))
        if threads:
            torch.set_num_threads(threads)

    # -- AudioEncoder.forward (base_v2.py:126-159) ------------------------------------------------
    def audio_encode(self, name, codes):
        codes = codes.long()
        if codes.shape[0] == 0:
            return torch.zeros((0, self.d), dtype=self.dtype)
        w = self.w
        x = torch.stack([w[f"{name}.embs.{l}.weight"][codes[:, l]] for l in range(self.n_levels)], dim=1)   # (T, L, d)
        x = x + w[f"{name}.pos_embedding"]                               # not monolithic: one (1, L, d) parameter
        x = F.layer_norm(x, (self.d,), w[f"{name}.norm.weight"], w[f"{name}.norm.bias"], 1e-5)
        h = F.linear(F.gelu(F.linear(x, w[f"{name}.proj.0.weight"], w[f"{name}.proj.0.bias"])),
                     w[f"{name}.proj.2.weight"], w[f"{name}.proj.2.bias"])
        x = x + h
        lw = F.softmax(w[f"{name}.level_weights"], dim=0).view(1, -1, 1)
        return (x * lw).sum(dim=1)

    # -- one sequence: embeddings, position ids (base_v2.py:725-799, :826-873) --------------------
    def embed(self, phn, lang, prom, resp, tone=None):
        w = self.w
        segs = [("phn", w["phn_emb.weight"][phn.long()])]
        if lang is not None:
            segs.append(("lang", w["langs_emb.weight"][lang.long()]))
        if prom is not None:
            segs.append(("prom", self.audio_encode("proms_emb", prom)))
        if tone is not None:
            segs.append(("tone", w["tones_emb.weight"][tone.long()]))
        if resp is not None:
            segs.append(("resp", self.audio_encode("resps_emb", resp)))
        x = segs[0][1]
        for _, e in segs[1:]:
            x = torch.cat((x, w["sep"][None], e), dim=0)
        pos = torch.cat([torch.arange(e.shape[0] + (0 if k == "resp" else 1), dtype=torch.int32) for k, e in segs])[: x.shape[0]]
        if self.unified_pos:
            pos = torch.arange(x.shape[0], dtype=torch.int32)
        return x, pos

    # -- AudioDecoder.forward (base_v2.py:215-225): (T, d) -> (L, T, V) ---------------------------
    def audio_decode(self, h):
        w, out = self.w, []
        for l in range(self.n_levels):
            p = f"audio_decoder.head.{l}."
            x = F.layer_norm(h, (self.d,), w[p + "0.weight"], w[p + "0.bias"], 1e-5)
            x = F.linear(F.gelu(F.linear(x, w[p + "1.weight"], w[p + "1.bias"])), w[p + "3.weight"], w[p + "3.bias"])
            out.append(F.linear(x, w[p + "4.weight"], w[p + "4.bias"]))
        return torch.stack(out, dim=0)

    # -- Base_V2.forward for the tts layout: list of (L, T_i, V) logits ----------------------------
    def forward_v2(self, phns, langs, proms, resps, timestep=False, return_embeddings=False):
        B = len(phns)
        xs, ps = zip(*[self.embed(phns[i], None if langs is None else langs[i], None if proms is None else proms[i], resps[i])
                       for i in range(B)])
        lens = [x.shape[0] for x in xs]
        causal = not (self.cfg.experimental.noncausal_masks and timestep)     # NAR:0:0 is not a causal level (base_v2.py:1182-1185)
        hid = self.stack(torch.cat(xs), torch.cat(ps), lens, causal=[causal] * B, final_norm=not self.cfg.experimental.per_level_normalization)
        out, o = [], 0
        for n in lens:
            out.append(self.audio_decode(hid[o:o + n]))
            o += n
        return (out, list(xs), list(ps)) if return_embeddings else out

    # -- Base_V2.sample (base_v2.py:1335-1411), default filters: top-n-sigma then softmax ------------
    @staticmethod
    def sample_v2(logits, n_rows, temperature=0.0, top_no=1.0, generator=None, **_unused):
        logits = [l[..., -n:, :] for l, n in zip(logits, n_rows)]
        if top_no > 0.0:
            logits = [top_n_sigma(l) for l in logits]
        probs = [F.softmax(l, dim=-1) for l in logits]
        if temperature <= 0.0:
            ids = [p.argmax(dim=-1) for p in probs]
        else:
            ids = [torch.distributions.Categorical(logits=l / temperature).sample() for l in logits]
        scores = [torch.gather(p, -1, t.unsqueeze(-1)).squeeze(-1) for p, t in zip(probs, ids)]
        return ids, scores

    @staticmethod
    def cfg_v2(logits, null, strength, lens, rescale):
        """samplers.py:177-191 on (levels, rows, vocab) logits: the rescale statistics run over dim 1 -- the rows -- per
        level and vocabulary column (``dims = tuple(range(1, ndim - 1))``)."""
        for i, n in enumerate(lens):
            pos, neg = logits[i][..., -n:, :], null[i][..., -n:, :]
            s = neg + (pos - neg) * strength
            if rescale > 0:
                s = s * (rescale * (pos.std((1,), keepdim=True) / s.std((1,), keepdim=True)) + (1 - rescale))
            logits[i][..., -n:, :] = s
        return logits

    # -- AR_NAR_V2.forward_nar_masked (ar_nar_v2.py:207-372) ---------------------------------------
    def nar_masked_v2(self, phns, proms, langs, lens, tie_rule="topk", **kw):
        """tie_rule: which positions are masked when the k-th and (k+1)-th lowest scores of a level are equal -- "topk":
        whatever torch.topk returns (the reference's behaviour on this torch build: an internal of its partial sort);
        "lowest_index": the lower index first (a stable sort), the rule valle_mask_lowest implements."""
        kw = convert_kwargs(dict(kw), "ar_")
        lo, hi = kw.pop("min_duration", 1), kw.pop("max_duration", 500)
        lens = [clamp(int(l), lo, hi) for l in lens]
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
        flatten = kw.pop("sampling_scores_flatten", False)
        trace = kw.pop("trace", None)                                    # list: receives (ids, scores, logits) of every step
        B, Lv = len(lens), self.n_levels
        resps = [torch.full((n, Lv), self.mask_token, dtype=torch.int64) for n in lens]
        scores = [torch.ones((n, Lv), dtype=torch.float32) for n in lens]
        null_phn = [torch.tensor([1, 2]) for _ in range(B)]
        for step, t in enumerate(torch.linspace(start, end, max_steps)):
            noise_p = math.cos(t * math.pi * 0.5)
            mask_p = noise_p + (1.0 / (max_steps * 2) if remask else 0)
            ks = [clamp(int(mask_p * n), 1, n - step) for n in lens]
            if tie_rule == "topk":
                idx = [s.topk(k, dim=0, largest=False).indices for s, k in zip(scores, ks)]
            else:
                idx = [s.sort(dim=0, stable=True).indices[:k] for s, k in zip(scores, ks)]
            resps = [torch.stack([r[:, l].scatter(0, i.t()[l], self.mask_token) for l in range(Lv)], dim=-1) for r, i in zip(resps, idx)]
            is_masked = [r == self.mask_token for r in resps]
            temp = temperature * (1.0 - float(t)) if annealed else temperature
            logits = self.forward_v2(phns, langs, proms, resps, timestep=True)
            if cfg_strength > 0:
                null = self.forward_v2(null_phn, langs, None, resps, timestep=True)
                logits = self.cfg_v2(logits, null, cfg_strength, lens, cfg_rescale)
            ids, sc = self.sample_v2(logits, lens, temperature=temp, **{k: v for k, v in kw.items() if k in ("top_no",)})
            if trace is not None:
                trace.append(([i.clone() for i in ids], [x.clone() for x in sc], [l[..., -n:, :].clone() for l, n in zip(logits, lens)]))
            resps = [torch.where(m, i.t(), r) for m, i, r in zip(is_masked, ids, resps)]
            scores = [torch.where(m, s.t(), torch.ones(())) if masked_only else s.t() for m, s in zip(is_masked, sc)]
            if flatten:
                scores = [s.mean(dim=0).repeat(s.shape[0], 1) for s in scores]
        return resps

    # -- AR_NAR_V2.forward_len (ar_nar_v2.py:374-431; base_v2.py:652-678, :1296-1308) with len_use_logits ------------
    def len_v2(self, phns, proms, langs, frames_per_second=75):
        """The head reads ``logit[..., -1, :]`` of the right-PADDED batch (base_v2.py:1298-1300): for an utterance shorter than
        the longest that is a padding row -- zero embedding, position id 1 (base_v2.py:865-867; T_max - 1 under unified ids),
        attending to the utterance's rows, never a key itself."""
        w, xs, ps = self.w, [], []
        for i in range(len(phns)):
            segs = [w["phn_emb.weight"][phns[i].long()]]
            if langs is not None:
                segs.append(w["langs_emb.weight"][langs[i].long()])
            if proms is not None:
                segs.append(self.audio_encode("proms_emb", proms[i]))
            segs.append(w["len_emb"][None])
            x = segs[0]
            for e in segs[1:]:
                x = torch.cat((x, w["sep"][None], e), dim=0)
            pos = torch.cat([torch.arange(e.shape[0] + (0 if j == len(segs) - 1 else 1), dtype=torch.int32) for j, e in enumerate(segs)])[: x.shape[0]]
            if self.unified_pos:
                pos = torch.arange(x.shape[0], dtype=torch.int32)
            xs.append(x), ps.append(pos)
        valid = [x.shape[0] for x in xs]
        T = max(valid)
        for i, n in enumerate(valid):
            if n < T:
                xs[i] = torch.cat((xs[i], torch.zeros((1, self.d), dtype=self.dtype)))
                ps[i] = torch.cat((ps[i], torch.tensor([T - 1 if self.unified_pos else 1], dtype=torch.int32)))
        lens = [x.shape[0] for x in xs]
        causal = not self.cfg.experimental.noncausal_masks                   # "len" is not among the causal levels (base_v2.py:1182)
        hid = self.stack(torch.cat(xs), torch.cat(ps), lens, causal=[causal] * len(lens), kv_lens=valid,
                         final_norm=not self.cfg.experimental.per_level_normalization)
        out, o = [], 0
        for n in lens:
            o += n
            lg = F.linear(hid[o - 1], w["len_decoder.head.weight"], w["len_decoder.head.bias"])
            digits = lg.view(5, 10).argmax(dim=-1)
            out.append(int(int("".join(str(int(t)) for t in digits)) / frames_per_second * frames_per_second))
        return out

    # -- AR_NAR_V2.forward_ar (ar_nar_v2.py:433-589): 8 codes per step from the newest row ---------
    def ar_v2(self, phns, proms, langs, *, max_duration=500, return_logits=False, **kw):
        kw = convert_kwargs(dict(kw), "ar_")
        B, Lv = len(phns), self.n_levels
        seqs = [torch.zeros((0, Lv), dtype=torch.int64) for _ in range(B)]
        stopped = [False] * B
        kept = []
        for _ in range(max_duration // max(1, self.causal_size)):
            logits = self.forward_v2(phns, langs, proms, seqs)
            if return_logits:
                kept.append(torch.stack([l[:, -1, :].float() for l in logits]))          # (B, L, V)
            new = [[] for _ in range(B)]
            for l in range(Lv):
                ids, _ = self.sample_v2([lg[l] for lg in logits], [self.causal_size] * B, temperature=kw.get("temperature", 1.0),
                                        top_no=kw.get("top_no", 1.0))
                for i, tok in enumerate(ids):
                    if self.stop_token in tok:
                        stopped[i] = True
                    new[i].append(tok)
            seqs = [torch.cat([s, torch.stack(n, dim=-1)]) for s, n in zip(seqs, new)]
            if all(stopped):
                break
        out = []
        for s in seqs:
            hit = (s == self.stop_token).nonzero()
            out.append(s[: int(hit[:, 0].min())] if len(hit) else s)
        return (out, kept) if return_logits else out
