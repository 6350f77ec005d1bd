"""AR_NAR_V2: the version >= 7 model's inference entry points on the HIP path.

Same call surface as ``vall_e.models.ar_nar_v2.AR_NAR_V2`` (reference models/ar_nar_v2.py:34-742): ``forward`` dispatches
to ``forward_nar_masked`` (all RVQ levels demasked together), ``forward_len`` and ``forward_ar`` (eight codes per step)
as ar_nar_v2.py:591-742 does.  The demasking state (codes, scores, masks of every level) stays on the device for the
whole loop; the sequence layout is planned and uploaded once per call.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from .. import _lib as L
from .ar_nar import _host_codes, clamp, convert_kwargs
from .base import RawText, _np
from .base_v2 import Base_V2


class AR_NAR_V2(Base_V2):
    # ------------------------------------------------------------------------------ dispatch
    def forward(self, task_list=None, phns_list=None, proms_list=None, resps_list=None, lang_list=None, tone_list=None,
                len_list=None, text_list=None, training=None, disable_tqdm=False, use_lora=None, **sampling_kwargs):
        if use_lora:
            raise NotImplementedError("LoRA on the version >= 7 model")
        if text_list is not None and phns_list is None:
            phns_list = [None if t is None else RawText(t) for t in text_list]
        if training is None and phns_list is not None and resps_list is not None:      # ar_nar_v2.py:625-629
            training = next(iter({int(r.shape[-1]) for r in resps_list})) == self.n_resp_levels
        if training:
            raise NotImplementedError("training is out of scope (resps with all levels mean training in the reference)")
        if (len_list is not None or resps_list is not None) and phns_list is not None:  # ar_nar_v2.py:650-672
            return self.forward_nar_masked(task_list=task_list, phns_list=phns_list, proms_list=proms_list, resps_list=resps_list,
                                           lang_list=lang_list, tone_list=tone_list, len_list=len_list, **sampling_kwargs)
        if task_list is not None and task_list[0] == "len":                            # ar_nar_v2.py:708-724
            return self.forward_len(task_list=task_list, phns_list=phns_list, proms_list=proms_list, lang_list=lang_list,
                                    tone_list=tone_list, **sampling_kwargs)
        return self.forward_ar(task_list=task_list, phns_list=phns_list, proms_list=proms_list, resps_list=resps_list,
                               lang_list=lang_list, tone_list=tone_list, **sampling_kwargs)

    __call__ = forward

    @staticmethod
    def _null_text(phns_list):
        return [RawText(torch.tensor([1, 2])) if isinstance(p, RawText) else torch.tensor([1, 2]) for p in phns_list]

    # ----------------------------------------------------------------------------------- len
    def forward_len(self, task_list=None, phns_list=None, proms_list=None, lang_list=None, tone_list=None, text_list=None, **_unused):
        """Duration prediction (ar_nar_v2.py:374-431): one pass over phn | lang | prom | tone | len, the ``len_decoder`` on the
        last row; seconds -> frames."""
        if text_list is not None and phns_list is None:
            phns_list = [None if t is None else RawText(t) for t in text_list]
        B = len(phns_list if phns_list is not None else proms_list)
        inputs = self.inputs(task_list=["len"] * B, phns_list=[p for p in phns_list] if phns_list is not None else None,
                             proms_list=proms_list, lang_list=lang_list, tone_list=tone_list, quant_levels=[0] * B)
        # RawText entries travel as text
        inputs = [[("text", v.ids) if n == "phn" and isinstance(v, RawText) else (n, v) for n, v in it] for it in inputs]
        out = self.forward_super(inputs=inputs, quant_levels=[0] * B)
        return [int(sec * self.audio_frames_per_second) for sec in out.logits]

    # ------------------------------------------------------------------------------- NAR-len
    def forward_nar_masked(self, task_list=None, phns_list=None, proms_list=None, resps_list=None, lang_list=None, tone_list=None,
                           len_list=None, text_list=None, disable_tqdm=False, use_lora=None, **sampling_kwargs):
        """Iterative demasking of all levels (ar_nar_v2.py:207-372): every step masks, per level, the lowest-scoring
        positions, runs the conditional and the null pass, applies the guidance over (levels, rows, vocab), samples every
        position of every level and keeps the new codes where the input was masked."""
        self._prepare()
        lib = L.load()
        if text_list is not None and phns_list is None:
            phns_list = [None if t is None else RawText(t) for t in text_list]
        kw = convert_kwargs(dict(sampling_kwargs), "ar_")                      # (sic) ar_nar_v2.py:241
        min_length = kw.pop("min_duration", 1)
        max_length = kw.pop("max_duration", 500)
        max_steps = kw.get("max_steps", 25)
        annealed = kw.get("annealed_sampling", True)
        temperature = kw.pop("temperature", 0.0)
        min_cfg = kw.get("minimum_cfg_strength", 2.5)
        cfg_strength = max(kw.get("cfg_strength", min_cfg), min_cfg)           # ar_nar_v2.py:252-254, :276
        cfg_rescale = kw.pop("cfg_rescale", 0.75)
        start_noise, end_noise = kw.get("denoise_start", 0.0), kw.get("denoise_end", 1.0)
        max_steps = math.floor(max_steps * (end_noise - start_noise))
        remasking = kw.get("sampling_scores_remask", False)
        masked_only = kw.pop("sampling_scores_masked_only", False)
        if kw.pop("sampling_scores_flatten", False):
            raise NotImplementedError("sampling_scores_flatten")
        if kw.get("vc_list") is not None:
            kw.pop("vc_list")                                                  # popped and never read again (ar_nar_v2.py:268-270)
        trace = kw.pop("trace", None)                                          # tests: receives (ids, scores) of every step
        seed = kw.pop("seed", 0)
        lens = [clamp(int(l), min_length, max_length) for l in len_list]
        B, Lv, dev = len(lens), self.n_resp_levels, self.device
        off = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
        total = int(off[-1])
        stream = L.stream_ptr(dev)
        # device state, level-major: codes / scores / masks [L][total]; one (level, utterance) segment per masking unit
        codes = self._filled((Lv, total), torch.int32, self.mask_token)         # ar_nar_v2.py:280-282
        scores = self._filled((Lv, total), torch.float32, 1.0)
        is_masked = torch.empty(Lv * total, dtype=torch.uint8, device=dev)
        seg_off = np.concatenate([l * total + off[:-1] for l in range(Lv)] + [[Lv * total]]).astype(np.int32)
        seg_off_d, off_d = self._upload_i32(seg_off, off)
        # the layout does not change between steps: plan, table rows and prompt embeddings are built once per pass kind
        plan = self.plan(phns_list, lang_list, proms_list, lens, tone_list=tone_list)
        x, pdev = self.build_x(plan)
        rows = pdev[3]                                                          # the response rows are the output rows
        if cfg_strength > 0:
            nplan = self.plan(self._null_text(phns_list), lang_list, None, lens, tone_list=tone_list)
            nx, ndev = self.build_x(nplan)
        causal = self.is_causal(True)
        for step, t in enumerate(torch.linspace(start_noise, end_noise, max_steps)):
            noise_p = math.cos(t * math.pi * 0.5)
            mask_p = noise_p + (1.0 / (max_steps * 2) if remasking else 0)
            ks = np.array([clamp(int(mask_p * n), 1, n - step) for n in lens] * Lv, dtype=np.int32)   # ar_nar_v2.py:300
            L.check(lib.valle_mask_lowest(L.ptr(scores), L.ptr(codes), L.ptr(is_masked), L.ptr(seg_off_d), L.ptr(self._i32(ks)),
                                          B * Lv, max(lens), self.mask_token, stream))
            temp = temperature * (1.0 - float(t)) if annealed else temperature
            y = self.encode_frames("resps_emb", codes, total, total)
            self.place_frames("resps_emb", y, total, pdev[3], x)
            logits = self.decode(self.forward_hidden(plan, x, pdev, rows, causal))
            if cfg_strength > 0:
                self.place_frames("resps_emb", y, total, ndev[3], nx)
                null = self.decode(self.forward_hidden(nplan, nx, ndev, ndev[3], causal))
                self.cfg_combine_cols(logits, null, off_d, B, cfg_strength, cfg_rescale)
            sp = self.sample_params_v2(temperature=temp, seed=seed, offset=3000 + step,
                                       **{k: v for k, v in kw.items() if k in ("top_k", "top_p", "min_p", "top_no")})
            ids, sc = self.sample_flat(logits.view(Lv * total, -1), Lv * total, sp)
            if trace is not None:
                trace.append((ids.clone(), sc.clone()))
            L.check(lib.valle_narlen_commit(L.ptr(codes), L.ptr(scores), L.ptr(is_masked), L.ptr(ids), L.ptr(sc), Lv * total,
                                            1 if masked_only else 0, stream))   # ar_nar_v2.py:361-366
        host = codes.cpu().numpy()                                              # (L, total) -> per utterance (frames, L) int16
        return [torch.from_numpy(np.ascontiguousarray(host[:, off[i]:off[i + 1]].T).astype(np.int16)).to(dev) for i in range(B)]

    # ------------------------------------------------------------------------------------ AR
    def forward_ar(self, task_list=None, phns_list=None, proms_list=None, resps_list=None, lang_list=None, tone_list=None,
                   len_list=None, text_list=None, disable_tqdm=False, use_lora=None, **sampling_kwargs):
        """Autoregression over frames, all levels of a frame per step (ar_nar_v2.py:433-589).  The reference runs the whole
        sequence through the stack every step (its state is never used: ``use_cache=False``, base_v2.py:539); here the prompt
        is prefilled once into a KV cache and every step runs the newest frame's encoder row through one decode step of the
        stack (``valle_ar_decode_layers``), the AudioDecoder and the sampler -- the same function of the same inputs, as the
        version-5 loop does it.  ``kv_cache=False`` keeps the reference's schedule (whole sequence per step, the encoder
        output of a frame computed once).  Steps are enqueued in chunks without a host round trip; the stop test reads the
        codes of a chunk back in one copy."""
        self._prepare()
        lib = L.load()
        if text_list is not None and phns_list is None:
            phns_list = [None if t is None else RawText(t) for t in text_list]
        kw = convert_kwargs(dict(sampling_kwargs), "ar_")
        temperature = kw.get("temperature", 1.0)
        if kw.get("cfg_strength", 0.0) > 0:
            # cfg_logits(..., lens=[frames + 1]) runs its rescale statistics over the ROWS of a (levels, rows, vocab) slice
            # (samplers.py:187-188): at the first step that is one row, whose unbiased std is nan -- every logit of the
            # reference's first step is nan under the default rescale of 0.7 (ar_nar_v2.py:472, :551)
            raise NotImplementedError("classifier-free guidance in the version >= 7 AR loop")
        for k in ("beam_width", "entropix_sampling", "layer_skip", "refine_on_stop", "input_prompt_prefix", "prefix_silence"):
            if kw.get(k):
                raise NotImplementedError(f"{k} on the version >= 7 AR loop")       # read into locals, never used (:475-482)
        max_duration = kw.get("max_duration", 500)
        step_chunk = int(kw.get("step_chunk", 16))
        want_logits = kw.get("return_logits")                                      # tests: list receiving (B, L, V) per step
        seed = kw.get("seed", 0)
        use_cache = bool(kw.get("kv_cache", True))
        B, Lv, d, dev = len(phns_list), self.n_resp_levels, self.d_model, self.device
        n_steps = max_duration // max(1, self.causal_size)
        if n_steps <= 0:
            return [torch.zeros((0, Lv), dtype=torch.int64, device=dev) for _ in range(B)]
        # codes of step t: [t][level][utterance] (the sampler's row order), summed encoder rows of step t: [t][utterance][d]
        step_codes = torch.empty((n_steps, Lv, B), dtype=torch.int32, device=dev)
        enc = torch.empty((n_steps, B, d), dtype=self.dtype, device=dev)
        host_codes = np.zeros((0, Lv, B), dtype=np.int64)
        n, done = 0, False
        spk = {k: v for k, v in kw.items() if k in ("top_k", "top_p", "min_p", "top_no")}

        def sample_and_encode(t, logits):                                          # logits (L, B, ld): row l * B + b
            if want_logits is not None:
                want_logits.append(logits[:, :, :self.audio_vocab].transpose(0, 1).clone())
            sp = self.sample_params_v2(temperature=temperature, seed=seed, offset=4000 + t, **spk)
            self.sample_flat(logits.view(Lv * B, -1), Lv * B, sp, ids_out=step_codes[t].view(-1))
            y = self.encode_frames("resps_emb", step_codes[t], B, B)
            self.place_frames("resps_emb", y, B, None, enc[t])

        if use_cache:
            H, Dh, Lyr = self.n_heads, d // self.n_heads, self.n_layers
            plan = self.plan(phns_list, lang_list, proms_list, [0] * B, tone_list=tone_list)
            if plan.max_T + n_steps + 1 > self.max_position_embeddings:
                raise ValueError(f"sequence of {plan.max_T} + {n_steps} steps exceeds max_position_embeddings ({self.max_position_embeddings})")
            cache_T = -(-(plan.max_T + n_steps + 1) // 8) * 8
            stride = B * H * cache_T * Dh
            kc = torch.empty((Lyr, B, H, cache_T, Dh), dtype=self.dtype, device=dev)
            vc = torch.empty_like(kc)
            slot, pos0 = self._i32(np.arange(B)), self._filled(B, torch.int32)
            x, pdev = self.build_x(plan)
            last = self._i32(plan.cu[1:] - 1)
            hidden = self.forward_hidden(plan, x, pdev, last, 1, cache=(kc, vc, cache_T, stride, slot, pos0))
            ctx_len = self._i32(plan.lens)
            # position id of response frame t: t under per-segment ids, rows in front of the response + t under unified ids
            base_pos = np.array(plan.lens if self.unified_position_ids else [0] * B, dtype=np.int32)
            st = L.ARState(B, n_steps, None, None, ctx_len.data_ptr(), None, slot.data_ptr(), None, None, kc.data_ptr(), vc.data_ptr(),
                           cache_T, stride, 0, 0, 0, 0, int(plan.max_T))
            ws_bytes = lib.valle_decode_workspace_bytes(self._handle, B)
            ws = self._workspace(max(ws_bytes, 1))
            fin = 0 if self.per_level_normalization else 1
        while n < n_steps and not done:
            chunk = min(step_chunk, n_steps - n)
            if use_cache:
                pos_chunk = self._i32(np.stack([base_pos + t - 1 for t in range(max(n, 1), n + chunk)]) if n + chunk > max(n, 1) else np.zeros((0, B), np.int32))
            for t in range(n, n + chunk):
                if use_cache:
                    if t > 0:                                                      # frame t - 1 enters the stack, its hidden row predicts frame t
                        st.max_ctx = int(plan.max_T) + t - 1
                        hidden = torch.empty((B, d), dtype=self.dtype, device=dev)
                        L.check(lib.valle_ar_decode_layers(self._handle, C.byref(st), L.ptr(enc[t - 1]), L.ptr(pos_chunk[t - max(n, 1)]), L.ptr(hidden),
                                                           fin, L.ptr(ws), ws.numel(), L.stream_ptr(dev)))
                    sample_and_encode(t, self.decode(hidden))
                    continue
                plan_t = self.plan(phns_list, lang_list, proms_list, [t] * B, tone_list=tone_list)
                x, pdev = self.build_x(plan_t)                                     # resp_dst: utterance-major (b, frame)
                if t > 0:
                    # frame-major kept rows [t'][b] -> the packed rows of (b, t')
                    dst = self._i32(plan_t.resp_dst.reshape(B, t).T.reshape(-1))
                    self.copy_rows(enc, t * B, dst, x)
                last = self._i32(plan_t.cu[1:] - 1)
                sample_and_encode(t, self.decode(self.forward_hidden(plan_t, x, pdev, last, 1)))   # (L, B, ld): row l * B + b
            got = step_codes[n:n + chunk].cpu().numpy().astype(np.int64)
            host_codes = np.concatenate([host_codes, got])
            n += chunk
            stopped = (host_codes == self.stop_token).any(axis=(0, 1))              # ar_nar_v2.py:566-568, :576
            done = bool(stopped.all())
        hit = (host_codes == self.stop_token).any(axis=1)                           # (steps, B): a stop token on any level
        if done:                                                                    # the reference leaves the loop at the first
            all_by = np.maximum.accumulate(hit, axis=0).all(axis=1)                 # step after which every utterance has stopped
            hit, host_codes = hit[: int(np.argmax(all_by)) + 1], host_codes[: int(np.argmax(all_by)) + 1]
        out = []
        for b in range(B):
            seq = host_codes[:, :, b]                                               # (steps, L)
            rows = np.nonzero(hit[:, b])[0]                                         # ar_nar_v2.py:580-587: cut at the first stop
            if len(rows):
                seq = seq[: int(rows[0])]
            out.append(torch.from_numpy(np.ascontiguousarray(seq)).to(dev))
        return out
