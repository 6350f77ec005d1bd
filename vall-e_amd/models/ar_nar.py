"""AR_NAR: the reference's inference entry points on the HIP path.

Same call surface as ``vall_e.models.ar_nar.AR_NAR`` (reference models/ar_nar.py:34-983):
``model(task_list=, phns_list=, proms_list=, resps_list=, lang_list=, len_list=, **sampling_kwargs)``
dispatches to ``forward_ar`` / ``forward_nar`` / ``forward_nar_masked`` exactly as
ar_nar.py:892-983 does.  The AR loop keeps a KV cache on the device (the reference recomputes
the whole sequence every step, base.py:521) and runs steps in chunks with no host sync inside.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from .. import _lib as L
from .base import Base, Logits, RawText, _np, _round_up


def convert_kwargs(kwargs, prefix):
    """utils/utils.py:46-56."""
    for key in list(kwargs):
        if key.startswith(prefix):
            kwargs[key[len(prefix):]] = kwargs.pop(key)
    return kwargs


def clamp(n, lo, hi):
    return max(lo, min(n, hi))


def _host_codes(tensors):
    """list of (R_i, L) integer tensors -> list of int64 numpy arrays, with ONE device->host copy when
    they live on a GPU (a .cpu() per utterance costs ~40 us of synchronisation each)."""
    if all(isinstance(t, torch.Tensor) and t.is_cuda for t in tensors) and len({t.shape[1:] for t in tensors}) == 1:
        t0 = tensors[0]
        chained = len({t.dtype for t in tensors}) == 1 and all(
            t.is_contiguous() and t.untyped_storage().data_ptr() == t0.untyped_storage().data_ptr() for t in tensors) and all(
            b.storage_offset() == a.storage_offset() + a.numel() for a, b in zip(tensors, tensors[1:]))
        if chained:
            # consecutive views of one buffer (what forward_ar / forward_nar return): one copy of that range, no cat kernel
            rows = sum(t.shape[0] for t in tensors)
            whole = t0.new_empty(0).set_(t0.untyped_storage(), t0.storage_offset(), (rows,) + tuple(t0.shape[1:]))
            flat = whole.cpu().numpy().astype(np.int64)
        else:
            flat = torch.cat(list(tensors)).cpu().numpy().astype(np.int64) if len({t.dtype for t in tensors}) == 1 else \
                torch.cat([t.long() for t in tensors]).cpu().numpy()
        cuts = np.cumsum([0] + [t.shape[0] for t in tensors])
        return [flat[cuts[i]:cuts[i + 1]] for i in range(len(tensors))]
    return [_np(t) for t in tensors]


class AR_NAR(Base):
    # ---------------------------------------------------------------- lower-level surface
    def forward_super(self, inputs, quant_levels=None, state=None, output_attentions=False, output_hidden_states=False):
        """Base.forward on an explicit batch description -- all rows' logits.  ``inputs`` is either what
        ``self.inputs(...)`` returns (the reference's list of (name, value) tuples per utterance, ar_nar.py:36-37)
        or a dict with the keyword lists of ``plan`` (phns_list, lang_list, proms_list, resps_list)."""
        if output_hidden_states or state is not None:
            raise NotImplementedError("hidden states / external state are not exposed")
        if isinstance(inputs, (list, tuple)):
            plan = self.plan_from_inputs(inputs)
        else:
            q = 0 if quant_levels is None else int(quant_levels[0])
            if quant_levels is not None and any(int(x) != q for x in quant_levels):
                raise NotImplementedError("one quant level per batch")
            plan = self.plan(inputs.get("phns_list"), inputs.get("lang_list"), inputs.get("proms_list"),
                             inputs["resps_list"], q, timestep=inputs.get("timestep", False))
        rows, off = self.tail_rows(plan, plan.lens)
        V = self.cls_tokens[plan.head]
        attentions = None
        if output_attentions:
            # the reference returns every layer's (B, H, T, T) raw scaled products (arch/llama.py:429, :469-477); here the LAST
            # layer's only -- the one its consumer reads (samplers.py:294) -- as a one-element list, zero-padded like the
            # reference's right-padded batch
            lg, sc = self.forward_dense(plan, rows, score_rows=rows)
            att = torch.zeros((plan.B, self.n_heads, plan.max_T, plan.max_T), dtype=torch.float32, device=self.device)
            for i in range(plan.B):
                n = int(plan.lens[i])
                att[i, :, :n, :n] = sc[off[i]:off[i + 1], :, :n].transpose(0, 1)
            attentions = [att]
        else:
            lg = self.forward_dense(plan, rows)
        return Logits([lg[off[i]:off[i + 1], :V] for i in range(plan.B)], None, inputs, None, attentions, None)

    # ------------------------------------------------------------------------------ dispatch
    def forward(self, task_list=None, phns_list=None, proms_list=None, resps_list=None, lang_list=None, tone_list=None,
                len_list=None, text_list=None, training=None, disable_tqdm=False, use_lora=None, **sampling_kwargs):
        # tone tokens enter a sequence only with the "tone" capability (base.py:623-624); every pass the reference plans
        # for this call -- conditional, null, len -- carries the same list, so it is kept for the duration of the call
        self._call_state.tones = tone_list if "tone" in self.capabilities else None
        self._call_state.use_lora = use_lora
        self._call_state.lora_on = False
        try:
            return self._forward(task_list, phns_list, proms_list, resps_list, lang_list, len_list, text_list, training,
                                 **sampling_kwargs)
        finally:
            self._call_state.tones = None
            self._call_state.lora_on = False

    def _set_lora_level(self, level):
        """enable_lora(self, cfg.lora.active_level(level) if use_lora is None else use_lora) of the reference's three
        generation loops: picks the weight set of the passes that follow."""
        self._call_state.lora_on = self.lora_active_level(level, getattr(self._call_state, "use_lora", None))

    def _forward(self, task_list, phns_list, proms_list, resps_list, lang_list, len_list, text_list, training, **sampling_kwargs):
        if text_list is not None and phns_list is None:                       # raw text in place of phonemes (base.py:609-610):
            phns_list = [None if t is None else RawText(t) for t in text_list]   # one list travels through every path
        if task_list is not None and any(t not in ("tts", "len") for t in task_list):
            raise NotImplementedError(f"tasks {sorted(set(task_list))}: only 'tts' (AR, NAR, NAR-len) and 'len' are built")
        if training:
            raise NotImplementedError("training is out of scope")
        if training is None and phns_list is not None and resps_list is not None:
            levels = {int(r.shape[-1]) for r in resps_list}                # ar_nar.py:927-931: shape[-1] also of a 1-D level-0
            if next(iter(levels)) == self.n_resp_levels:                    # sequence -- exactly 8 frames count as "all levels"
                raise NotImplementedError("resps whose last dimension equals the level count mean training in the reference "
                                          "(also a 1-D sequence of exactly that many frames); out of scope")
        if (len_list is not None or resps_list is not None) and phns_list is not None:
            return self.forward_nar(task_list=task_list, phns_list=phns_list, proms_list=proms_list, resps_list=resps_list,
                                    lang_list=lang_list, len_list=len_list, **sampling_kwargs)
        return self.forward_ar(task_list=task_list, phns_list=phns_list, proms_list=proms_list, resps_list=resps_list,
                               lang_list=lang_list, **sampling_kwargs)

    __call__ = forward

    @staticmethod
    def _null_text(phns_list):
        """The null condition's text <bos><eos>, as phonemes or as raw text like the input (ar_nar.py:787-788)."""
        return [RawText(torch.tensor([1, 2])) if isinstance(p, RawText) else torch.tensor([1, 2]) for p in phns_list]

    # ----------------------------------------------------------------------------------- len
    def forward_len(self, phns_list, proms_list=None, lang_list=None, return_logits=False):
        """Duration prediction, ar_nar.py:640-692: at most 10 greedy steps over the 11-token ``len`` head,
        every utterance seeded with digit 0.  Utterances that already produced the stop token (10) keep
        being extended until all have, and the integer drops stop tokens only -- both as in the reference.
        Each step is one dense pass over phn|lang|quant_level|prom|len (a few hundred rows per utterance)."""
        self._prepare()
        B = len(phns_list)
        stop = 10
        V = self.cls_tokens[self.cls_names.index("len")] if "len" in self.cls_names else 0
        seqs = [np.zeros(1, dtype=np.int64) for _ in range(B)]
        stopped = np.zeros(B, dtype=bool)
        greedy = self.sample_params(temperature=0.0, top_no=0.0)
        row_off = np.arange(B + 1, dtype=np.int32)
        kept = []
        for _ in range(10):
            plan = self.plan(phns_list, lang_list, proms_list, [None] * B, 0, task="len", len_list=seqs)
            rows, _ = self.tail_rows(plan, [1] * B)
            logits = self.forward_dense(plan, rows)
            if return_logits:
                kept.append(logits[:, :V].clone())
            ids, _ = self.sample_rows(logits, V, row_off, None, nar_mode=0, params=greedy)
            toks = np.minimum(ids.cpu().numpy().astype(np.int64), stop)         # ar_nar.py:675-677
            stopped |= toks == stop
            seqs = [np.concatenate((s, toks[i:i + 1])) for i, s in enumerate(seqs)]
            if stopped.all():
                break
        out = [int("".join(str(int(t)) for t in s if t != stop)) for s in seqs]
        return (out, kept) if return_logits else out

    # ------------------------------------------------------------------------------------ AR
    def forward_ar(self, task_list=None, phns_list=None, proms_list=None, resps_list=None, lang_list=None,
                   disable_tqdm=False, use_lora=None, **sampling_kwargs):
        """Level-0 autoregressive loop, ar_nar.py:588-890 (tts task); duration prediction for the len task."""
        if use_lora is not None:                                                # a direct call (forward() keeps it in the call state)
            self._call_state.use_lora = use_lora
        if task_list is not None and task_list[0] == "len":                     # ar_nar.py:640 (behind the LoRA switch of :619-620)
            self._prepare()
            self._set_lora_level(0)
            return self.forward_len(phns_list=phns_list, proms_list=proms_list, lang_list=lang_list,
                                    return_logits=sampling_kwargs.get("return_logits", False))
        self._prepare()
        self._set_lora_level(0)                                               # ar_nar.py:619-620
        lib = L.load()
        kw = convert_kwargs(dict(sampling_kwargs), "ar_")
        cfg_strength = kw.get("cfg_strength", 0.0)
        cfg_rescale = kw.pop("cfg_rescale", 0.7)
        max_duration = kw.get("max_duration", 500)
        step_chunk = int(kw.pop("step_chunk", 32))
        return_logits = kw.pop("return_logits", False)
        # finished utterances leave the decode step (valle_ar_state.early_exit) unless the caller wants what the reference
        # computes behind a stop token: refine_on_stop re-reads those rows, return_logits returns them
        early_exit = bool(kw.pop("early_exit", True))
        noalloc = kw.pop("noalloc", False)  # decode chunks through valle_ar_decode_noalloc (the library's static scratch array)
        prof = kw.pop("profile", None)      # list: receives (start_event, end_event, n_steps, ctx_len_host) per decode chunk
        prefix_context = kw.get("prefix_context", None)
        input_prompt_prefix = kw.pop("input_prompt_prefix", False)
        refine_on_stop = kw.pop("refine_on_stop", False)
        prefix_silence = kw.pop("prefix_silence", 0.0)
        entropix = bool(kw.pop("entropix_sampling", False))                   # ar_nar.py:782, :807
        entropix_noise = kw.pop("entropix_noise", None)                       # tests: [step] -> (B, 5, n) Exponential(1) variates
        trace = kw.pop("trace", None)                                         # list: receives every step's metrics dicts
        beam_width = int(kw.pop("beam_width", 0) or 0)
        if entropix and (beam_width > 1 or kw.get("cfg_strength", 0.0) > 0 or refine_on_stop or return_logits):
            raise NotImplementedError("entropix sampling together with beam search / guidance / refine_on_stop / return_logits")
        if beam_width == 1:
            beam_width = 0          # base.py:1412 needs > 1; a beam of one is plain sampling (the scores it keeps are unused)
        if beam_width > 1:
            # ar_nar.py:821-832: the first step turns the one utterance into beam_width sequences -- text, prompt, sequence,
            # task, but not the language list, which therefore cannot be given (IndexError at the second step in the reference)
            if len(phns_list) != 1:
                raise NotImplementedError("beam search starts from one utterance (ar_nar.py:825)")
            if lang_list is not None:
                raise IndexError("list index out of range (beam search does not expand lang_list, ar_nar.py:825-832)")
            if cfg_strength > 0 or prefix_context is not None or input_prompt_prefix or refine_on_stop or return_logits \
                    or any(isinstance(p, RawText) for p in phns_list):
                raise NotImplementedError("beam search together with guidance / prefix_context / input_prompt_prefix / "
                                          "refine_on_stop / return_logits / raw text")
            phns_list, proms_list = list(phns_list) * beam_width, list(proms_list) * beam_width
        sp = self.sample_params(**kw)
        B = len(phns_list)
        dev = self.device
        n_steps = max_duration // max(1, self.causal_size)
        if refine_on_stop and cfg_strength > 0:
            raise NotImplementedError("refine_on_stop together with AR classifier-free guidance")

        prefix = [np.zeros(0, np.int64) for _ in range(B)]
        prefix_lens = [0] * B
        start_slice = [0] * B
        if input_prompt_prefix:                                               # ar_nar.py:727-729: the prompt's level-0 codes
            start_slice = [int(_np(p).shape[0]) for p in proms_list]          # open the response; the prompt segment is empty
            prefix = [_np(p)[:, 0] for p in proms_list]
            proms_list = [torch.zeros(0, dtype=torch.int16) for _ in range(B)]
        elif prefix_silence > 0:                                              # ar_nar.py:730-732: quantised silence opens the
            from ..emb.qnt import get_silence                                 # response and stays in the result (no start_slice)
            prefix = [_np(get_silence(prefix_silence))[:, 0].astype(np.int64) for _ in range(B)]
        if prefix_context is not None:                                        # ar_nar.py:736-742
            if any(isinstance(p, RawText) for p in phns_list):
                raise TypeError("prefix_context splices phoneme lists; the reference fails the same way with text_list")
            prefix_text, prefix_resps, prefix_lens = prefix_context
            phns_list = [torch.cat([torch.as_tensor(p)[:-1], torch.as_tensor(t)[1:]]) for p, t in zip(prefix_text, phns_list)]
            prefix = [(_np(r) if _np(r).ndim == 1 else _np(r)[:, 0]) for r in prefix_resps]

        passes = [(phns_list, proms_list)]
        if cfg_strength > 0:                                                  # null text / no prompt (ar_nar.py:744-745)
            passes.append((self._null_text(phns_list), None))

        V = self.cls_tokens[0]
        ld = _round_up(V, 8)
        H, Dh, Lyr = self.n_heads, self.d_model // self.n_heads, self.n_layers
        cap = max(len(p) for p in prefix) + n_steps + 1
        seq = self._filled((B, cap), torch.int32)
        for i, p in enumerate(prefix):
            if len(p):
                seq[i, : len(p)] = torch.from_numpy(p.astype(np.int32)).to(dev)
        seq_len = self._i32([len(p) for p in prefix])
        stopped = self._filled(B, torch.int32)
        scores = self._filled(B, torch.float32)
        slot = self._i32(np.arange(B))

        # A [task, codes] prompt makes the reference's position ids one short per task token (base.py:964-990): every row
        # behind the prompt reads its neighbour's id and the LAST row -- the newest token -- always gets 1, so a token's
        # position changes from one step to the next.  A KV cache holds keys rotated once; these calls therefore recompute
        # the sequence every step, as the reference does (measured: cache-path logits off by 1e-2 on the tiny model).
        if entropix:
            return self._ar_entropix(phns_list, proms_list, lang_list, prefix, prefix_lens, prefix_context, start_slice, n_steps,
                                     seq, seq_len, stopped, scores, cap, kw, entropix_noise, trace)
        if not self.unified_position_ids and proms_list is not None and any(
                isinstance(p, (list, tuple)) and any(isinstance(x, str) for x in p) for p in proms_list):
            all_logits = self._ar_recompute(passes, lang_list, prefix, n_steps, seq, seq_len, stopped, scores, sp, cap,
                                            cfg_strength, cfg_rescale, return_logits)
            return self._ar_finish(seq, seq_len, prefix, prefix_lens, prefix_context, start_slice, refine_on_stop, n_steps,
                                   phns_list, lang_list, proms_list, return_logits, all_logits)

        states, first_logits = [], []
        for phns, proms in passes:
            plan = self.plan(phns, lang_list, proms, [torch.from_numpy(p) for p in prefix], 0)
            cache_T = _round_up(plan.max_T + n_steps + 1, 8)
            if plan.max_T + n_steps + 1 > self.max_position_embeddings:       # the RoPE table has that many rows
                raise ValueError(f"sequence of {plan.max_T} + {n_steps} steps exceeds max_position_embeddings "
                                 f"({self.max_position_embeddings})")
            stride = B * H * cache_T * Dh
            kc = torch.empty((Lyr, B, H, cache_T, Dh), dtype=self.dtype, device=dev)
            vc = torch.empty_like(kc)
            # guidance needs the logits of every response row so far (samplers.py:178-189): with a prefix those are
            # the prefill's last len(prefix) + 1 rows
            rows, off = self.tail_rows(plan, [len(p) + 1 for p in prefix] if cfg_strength > 0 else [1] * B)
            pos0 = self._filled(B, torch.int32)
            lg = self.forward_dense(plan, rows, cache=(kc, vc, cache_T, stride, slot, pos0))
            ctx_len = self._i32(plan.lens)
            # RoPE position of response token 0: per-segment position ids restart at 0 (base.py:984); with
            # unified_position_ids (HF default arange over the whole sequence, arch/llama.py:825-831) it is the number of
            # rows in front of the response segment -- which differs between the conditional and the null pass
            resp_pos0 = self._i32([plan.lens[i] - len(prefix[i]) if self.unified_position_ids else 0 for i in range(B)])
            st = L.ARState(B, cap, seq.data_ptr(), seq_len.data_ptr(), ctx_len.data_ptr(), resp_pos0.data_ptr(),
                           slot.data_ptr(), stopped.data_ptr(), scores.data_ptr(), kc.data_ptr(), vc.data_ptr(),
                           cache_T, stride, self._off["resp0"], 0, V,
                           1 if early_exit and not refine_on_stop and not return_logits else 0, int(plan.max_T))
            states.append((st, (kc, vc, ctx_len, resp_pos0), int(plan.max_T)))
            first_logits.append(lg)
        ws_bytes = lib.valle_decode_workspace_bytes(self._handle, B)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        stream = L.stream_ptr(dev)
        all_logits = []

        if beam_width > 1:
            # host-driven steps: the stack runs the newest token of every sequence, the device picks the beam_width best
            # (row, token) pairs of all rows and appends candidate i to sequence i (valle_beam_pick + a forced greedy append)
            proc = torch.empty((B, ld), dtype=torch.float32, device=dev)
            forced = torch.empty((B, ld), dtype=torch.float32, device=dev)
            last = torch.empty((B, ld), dtype=torch.float32, device=dev)
            ids = torch.empty(B, dtype=torch.int32, device=dev)
            greedy = self.sample_params(temperature=0.0, top_no=0.0)
            st = states[0][0]
            for step in range(n_steps):
                if step > 0:
                    st.max_ctx = states[0][2] + step - 1
                    L.check(lib.valle_ar_decode(self._handle, C.byref(st), None, 1, L.ptr(last), ld, L.ptr(ws), ws.numel(), stream))
                src, rows = (first_logits[0], 1) if step == 0 else (last, B)       # step 0: the batch is still the one utterance
                L.check(lib.valle_beam_pick(L.ptr(src), src.stride(0), V, rows, B, C.byref(sp), L.ptr(seq), cap, L.ptr(seq_len),
                                            L.ptr(proc), L.ptr(forced), L.ptr(ids), None, stream))
                self._ar_sample_append(forced, V, seq, seq_len, [states[0][1][2]], stopped, scores, greedy, cap, step=step,
                                       bump_ctx=step > 0)
                if bool(stopped.cpu().numpy().all()):
                    break
            out = self._ar_finish(seq, seq_len, prefix, prefix_lens, prefix_context, start_slice, refine_on_stop, n_steps,
                                  phns_list, lang_list, proms_list, return_logits, all_logits)
            return out[:1]                                                        # ar_nar.py:865-867
        if cfg_strength <= 0:
            # step 0: sample from the prefill's last row, then chunks of cached steps
            st = states[0][0]
            if return_logits:
                all_logits.append(first_logits[0][:, :V].clone())
            self._ar_sample_append(first_logits[0], V, seq, seq_len, [states[0][1][2]], stopped, scores, sp, cap, step=0)
            done = 1
            while done < n_steps:
                if bool(stopped.cpu().numpy().all()):                                # ar_nar.py:846 (checked once per chunk)
                    break
                n = min(step_chunk, n_steps - done)
                lg_out = torch.empty((n, B, ld), dtype=torch.float32, device=dev) if return_logits else None
                sp.offset = done
                st.max_ctx = states[0][2] + done - 1          # every utterance has cached its prefill + (done - 1) tokens
                if prof is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                if noalloc:
                    L.check(lib.valle_ar_decode_noalloc(self._handle, C.byref(st), C.byref(sp), n, L.ptr(lg_out), ld, stream))
                else:
                    L.check(lib.valle_ar_decode(self._handle, C.byref(st), C.byref(sp), n, L.ptr(lg_out), ld,
                                                L.ptr(ws), ws.numel(), stream))
                if prof is not None:
                    e1.record()
                    prof.append((e0, e1, n, [int(t) + done for t in plan.lens]))
                if return_logits:
                    all_logits.extend(lg_out[j, :, :V] for j in range(n))
                done += n
        else:
            # classifier-free guidance: positive and null caches advance together; the combine
            # needs every response row's logits (std over rows x vocab, samplers.py:187-188), which
            # under causality are the last-row logits of the earlier steps -> keep a history
            # utterance b's rows sit right-aligned at hist[k][b, maxp - len(prefix[b]) : maxp + step + 1]
            n_prefix = np.array([len(p) for p in prefix], dtype=np.int64)
            maxp = int(n_prefix.max())
            hcap = maxp + n_steps + 1
            hist = [self._filled((B, hcap, ld), torch.float32) for _ in range(2)]
            foff = np.concatenate(([0], np.cumsum(n_prefix + 1)))
            for k in range(2):                                                 # the prefill's rows open the histories
                for b in range(B):
                    hist[k][b, maxp - n_prefix[b]: maxp + 1] = first_logits[k][foff[b]:foff[b + 1]]
            row_start = self._i32([b * hcap + maxp - int(n_prefix[b]) for b in range(B)])
            row_count0 = self._i32([int(n) + 1 for n in n_prefix])             # rows at step 0; step n reads n more
            last = torch.empty((B, ld), dtype=torch.float32, device=dev)
            # step 0: guidance over the prefill's response rows, last row out, sample + append (nothing to bump yet)
            L.check(lib.valle_cfg_combine_last(L.ptr(hist[0]), L.ptr(hist[1]), ld, V, L.ptr(row_start), L.ptr(row_count0), B,
                                               float(cfg_strength), float(cfg_rescale), L.ptr(last), ld, stream))
            if return_logits:
                all_logits.append(last[:, :V].clone())
            self._ar_sample_append(last, V, seq, seq_len, [], stopped, scores, sp, cap, step=0)
            # the other steps run on the device in chunks (valle_ar_decode_cfg: both caches, the combine over the histories,
            # sampler + append), `stopped` is read once per chunk -- the reference syncs every step (ar_nar.py:846)
            done = 1
            while done < n_steps:
                if bool(stopped.cpu().numpy().all()):
                    break
                n = min(step_chunk, n_steps - done)
                lasts = torch.empty((n, B, ld), dtype=torch.float32, device=dev) if return_logits else last
                sp.offset = done
                for k in range(2):
                    states[k][0].max_ctx = states[k][2] + done - 1
                L.check(lib.valle_ar_decode_cfg(self._handle, C.byref(states[0][0]), C.byref(states[1][0]), C.byref(sp), n, done,
                                                L.ptr(hist[0]), L.ptr(hist[1]), ld, hcap, maxp, L.ptr(row_start), L.ptr(row_count0),
                                                float(cfg_strength), float(cfg_rescale), L.ptr(lasts),
                                                B * ld if return_logits else 0, L.ptr(ws), ws.numel(), stream))
                if return_logits:
                    all_logits.extend(lasts[j, :, :V] for j in range(n))
                done += n
        return self._ar_finish(seq, seq_len, prefix, prefix_lens, prefix_context, start_slice, refine_on_stop, n_steps,
                               phns_list, lang_list, proms_list, return_logits, all_logits)

    def _ar_entropix(self, phns_list, proms_list, lang_list, prefix, prefix_lens, prefix_context, start_slice, n_steps,
                     seq, seq_len, stopped, scores, cap, kw, noise, trace):
        """The AR loop under entropix_sampling (ar_nar.py:778-783, :804-808 -> base.py:1325-1342 -> samplers.py:464-566): every
        step hands the whole sequence's logits and the last layer's raw attention products of the newest row to
        sample_entropix, and nothing else of the sampler chain runs.  Host-driven, one small device->host copy per step
        (the reference does this part on the CPU altogether).  Kept on the device: the KV cache, the running column sums of
        log_softmax over all rows so far (the adaptive branch's ranking, samplers.py:527-545), the sequences."""
        lib = L.load()
        B, dev = len(phns_list), self.device
        if proms_list is not None and not self.unified_position_ids and any(
                isinstance(p, (list, tuple)) and any(isinstance(x, str) for x in p) for p in proms_list):
            raise NotImplementedError("entropix sampling with task tokens in the prompt (their position ids move per step)")
        V = self.cls_tokens[0]
        ld = _round_up(V, 8)
        H, Dh, Lyr = self.n_heads, self.d_model // self.n_heads, self.n_layers
        slot = self._i32(np.arange(B))
        plan = self.plan(phns_list, lang_list, proms_list, [torch.from_numpy(p) for p in prefix], 0)
        if plan.max_T + n_steps + 1 > self.max_position_embeddings:
            raise ValueError(f"sequence of {plan.max_T} + {n_steps} steps exceeds max_position_embeddings ({self.max_position_embeddings})")
        cache_T = _round_up(plan.max_T + n_steps + 1, 8)
        stride = B * H * cache_T * Dh
        kc = torch.empty((Lyr, B, H, cache_T, Dh), dtype=self.dtype, device=dev)
        vc = torch.empty_like(kc)
        rows, off = self.tail_rows(plan, plan.lens)                          # every row: the ranking sums over all of them
        last_rows = (plan.cu[1:] - 1).astype(np.int32)
        pos0 = self._filled(B, torch.int32)
        lg, sc = self.forward_dense(plan, rows, cache=(kc, vc, cache_T, stride, slot, pos0), score_rows=last_rows)
        ctx_len = self._i32(plan.lens)
        resp_pos0 = self._i32([plan.lens[i] - len(prefix[i]) if self.unified_position_ids else 0 for i in range(B)])
        st = L.ARState(B, cap, seq.data_ptr(), seq_len.data_ptr(), ctx_len.data_ptr(), resp_pos0.data_ptr(), slot.data_ptr(),
                       stopped.data_ptr(), scores.data_ptr(), kc.data_ptr(), vc.data_ptr(), cache_T, stride, self._off["resp0"],
                       0, V, 0, int(plan.max_T))
        ws = torch.empty(lib.valle_decode_workspace_bytes(self._handle, B), dtype=torch.uint8, device=dev)
        stream = L.stream_ptr(dev)
        colsum = torch.empty((B, ld), dtype=torch.float32, device=dev)
        off_d = self._i32(off)
        L.check(lib.valle_logsoftmax_colsum(L.ptr(lg), ld, V, L.ptr(off_d), B, L.ptr(colsum), ld, 0, stream))
        step_off = self._i32(np.arange(B + 1))
        greedy = self.sample_params(temperature=0.0, top_no=0.0)
        kws = dict(temperature=kw.get("temperature", 1.0), top_k=kw.get("top_k", -100), top_p=kw.get("top_p", 1.0), seed=kw.get("seed", 0))
        step_lg = torch.empty((B, ld), dtype=torch.float32, device=dev)
        step_sc = self._filled((B, H, cache_T), torch.float32)
        for step in range(n_steps):
            if step == 0:
                cur, cur_rows, cur_sc = lg, last_rows, sc
            else:
                st.max_ctx = int(plan.max_T) + step - 1
                L.check(lib.valle_ar_decode_scores(self._handle, C.byref(st), L.ptr(step_lg), ld, L.ptr(step_sc), cache_T,
                                                   L.ptr(ws), ws.numel(), stream))
                L.check(lib.valle_logsoftmax_colsum(L.ptr(step_lg), ld, V, L.ptr(step_off), B, L.ptr(colsum), ld, 1, stream))
                cur, cur_rows, cur_sc = step_lg, None, step_sc
            ids, metrics, forced = self.entropix_sample(cur, cur_rows, V, cur_sc, [int(n) + step for n in plan.lens], colsum,
                                                        offset=step, noise=None if noise is None else noise[step],
                                                        want_forced=True, **kws)
            if trace is not None:
                trace.append(metrics)
            self._ar_sample_append(forced, V, seq, seq_len, [ctx_len], stopped, scores, greedy, cap, step=step, bump_ctx=step > 0)
            if bool(stopped.cpu().numpy().all()):
                break
        return self._ar_finish(seq, seq_len, prefix, prefix_lens, prefix_context, start_slice, False, n_steps,
                               phns_list, lang_list, proms_list, False, [])

    def _ar_finish(self, seq, seq_len, prefix, prefix_lens, prefix_context, start_slice, refine_on_stop, n_steps,
                   phns_list, lang_list, proms_list, return_logits, all_logits):
        """Host side of the loop's end (ar_nar.py:863-890): prune at the stop token, cut prefixes, optional refine."""
        dev, B = self.device, seq.shape[0]
        torch.cuda.current_stream(dev).synchronize()      # this call's stream only: another stream may be busy with other work
        seq_h, len_h = seq.cpu().numpy(), seq_len.cpu().numpy()
        out = []
        for i in range(B):
            r = seq_h[i, : len_h[i]].astype(np.int64)
            hit = np.nonzero(r == self.stop_token)[0]                         # _prune, base.py:290-299
            if len(hit):
                r = r[: hit.min()]
            out.append(r[start_slice[i]:])                                    # ar_nar.py:872
        if refine_on_stop:
            out = self._refine_on_stop(out, seq_h, len_h, [len(p) for p in prefix], n_steps, phns_list, lang_list, proms_list)
        if prefix_context is not None:                                        # ar_nar.py:886-888
            out = [r[prefix_lens[i]:] for i, r in enumerate(out)]
        # one host->device copy for the whole batch (a copy per utterance costs ~50 us of idle GPU each)
        cuts = np.cumsum([0] + [len(r) for r in out])
        flat = torch.from_numpy(np.ascontiguousarray(np.concatenate(out)) if cuts[-1] else np.zeros(0, np.int64)).to(dev)
        out = [flat[cuts[i]:cuts[i + 1]] for i in range(B)]
        if return_logits:
            return out, all_logits
        return out

    def _ar_recompute(self, passes, lang_list, prefix, n_steps, seq, seq_len, stopped, scores, sp, cap, cfg_strength,
                      cfg_rescale, return_logits):
        """The AR loop without a KV cache: one dense pass over the whole sequence per step (and per guidance pass), the
        reference's own schedule (base.py:521).  Only for inputs whose position ids move between steps."""
        B = seq.shape[0]
        V = self.cls_tokens[0]
        all_logits = []
        for step in range(n_steps):
            len_h = seq_len.cpu().numpy()
            seq_h = seq.cpu().numpy()
            resp_in = [seq_h[i, : len_h[i]].astype(np.int64) for i in range(B)]
            counts = [int(n) + 1 for n in len_h] if cfg_strength > 0 else [1] * B
            outs = []
            for phns, proms in passes:
                plan = self.plan(phns, lang_list, proms, resp_in, 0)
                rows, off = self.tail_rows(plan, counts)
                outs.append(self.forward_dense(plan, rows))
            logits = outs[0]
            if cfg_strength > 0:
                self.cfg_combine(outs[0], outs[1], V, off, cfg_strength, cfg_rescale)
                logits = outs[0].index_select(0, torch.from_numpy(off[1:].astype(np.int64) - 1).to(self.device)).contiguous()
            if return_logits:
                all_logits.append(logits[:, :V].clone())
            self._ar_sample_append(logits, V, seq, seq_len, [], stopped, scores, sp, cap, step=step)
            if bool(stopped.cpu().numpy().all()):
                break
        return all_logits

    def _refine_on_stop(self, out, seq_h, len_h, n_prefix, n_steps, phns_list, lang_list, proms_list):
        """ar_nar.py:874-883: the reference re-reads the logits of its LAST iteration greedily -- for every utterance
        the ``len(result)`` rows that end one before the last row, counted from the end of that iteration's sequence
        (for an utterance that stopped earlier than the others these are not its own first rows; kept).  The last
        iteration is the one in which the last utterance stopped (the decode loop here checks ``stopped`` once per
        chunk and may have run further); its input is recomputed in one dense pass."""
        B = len(out)
        gen = [seq_h[i, n_prefix[i]: len_h[i]] for i in range(B)]
        first_stop = [np.nonzero(g == self.stop_token)[0] for g in gen]
        n_iter = max(int(h[0]) + 1 for h in first_stop) if all(len(h) for h in first_stop) else n_steps
        resp_in = [seq_h[i, : n_prefix[i] + n_iter - 1].astype(np.int64) for i in range(B)]
        plan = self.plan(phns_list, lang_list, proms_list, resp_in, 0)
        rows, off = [], [0]
        for i in range(B):
            T, n = int(plan.lens[i]), len(out[i])
            lo, hi = max(T - n - 1, 0), T - 1                                 # python slice [-n-1:-1] of T rows
            rows.append(int(plan.cu[i]) + np.arange(lo, hi, dtype=np.int32))
            off.append(off[-1] + hi - lo)
        rows = np.concatenate(rows) if off[-1] else np.zeros(0, np.int32)
        if off[-1] == 0:
            return [np.zeros(0, np.int64) for _ in range(B)]
        logits = self.forward_dense(plan, rows)
        off = np.array(off, dtype=np.int32)
        ids, _ = self.sample_rows(logits, self.cls_tokens[0], off, None, nar_mode=1,
                                  params=self.sample_params(temperature=0.0, top_no=0.0))
        ids = ids.cpu().numpy().astype(np.int64)
        return [ids[off[i]:off[i + 1]] for i in range(B)]

    def _ar_sample_append(self, logits, V, seq, seq_len, ctx_lens, stopped, scores, sp, cap, step, bump_ctx=False):
        """Sampling of one AR row per utterance from explicit logits + token append (host-driven
        steps: the prefill row and the CFG path), one launch (valle_sample_append).  prev = the sequence so far
        (ar_nar.py:804-808); cached positions grow when the token was run through the stack (bump_ctx)."""
        lib = L.load()
        B = seq.shape[0]
        ids = torch.empty(B, dtype=torch.int32, device=self.device)
        sp.offset = step
        ctx = [L.ptr(c) for c in ctx_lens] if bump_ctx else []
        ctx += [None] * (2 - len(ctx))
        L.check(lib.valle_sample_append(L.ptr(logits), logits.stride(0), V, B, C.byref(sp), L.ptr(seq), cap, L.ptr(seq_len),
                                        ctx[0], ctx[1], L.ptr(stopped), L.ptr(ids), L.ptr(scores), L.stream_ptr(self.device)))

    # ----------------------------------------------------------------------------------- NAR
    def forward_nar(self, task_list=None, phns_list=None, proms_list=None, resps_list=None, lang_list=None,
                    len_list=None, disable_tqdm=False, use_lora=None, **sampling_kwargs):
        """Levels 1..max_levels, one dense pass per level (ar_nar.py:466-586)."""
        if use_lora is not None:
            self._call_state.use_lora = use_lora
        if len_list is not None:                                              # ar_nar.py:485-496
            resps_list = self.forward_nar_masked(task_list=task_list, phns_list=phns_list, proms_list=proms_list,
                                                 resps_list=resps_list, lang_list=lang_list, len_list=len_list,
                                                 **sampling_kwargs)
        kw = convert_kwargs(dict(sampling_kwargs), "nar_")
        max_levels = kw.get("max_levels", 0)
        cfg_strength = kw.get("cfg_strength", 0.0)
        cfg_rescale = kw.pop("cfg_rescale", 0.7)
        if max_levels == 0:
            max_levels = self.n_max_levels - 1
        B = len(resps_list)
        # the reference mutates the caller's list in place (ar_nar.py:523-525); keep that
        for i, r in enumerate(resps_list):
            if r.dim() == 1:
                resps_list[i] = r.unsqueeze(-1)
        # host mirror of the codes: sequence planning and the sampler's `prev` tables are integer
        # bookkeeping done on the host; one device->host copy here and one per level (the new column)
        prev = _host_codes(resps_list)
        sp = self.sample_params(**kw)
        null_text = self._null_text(phns_list)
        # quirk: the reference's level loop hands only phns_list to inputs() (ar_nar.py:544-551), so raw text
        # (text_list) is absent from the conditional pass of every NAR level; the null pass does get it (:560-562)
        cond_text = [None if isinstance(p, RawText) else p for p in phns_list]
        level0 = prev[0].shape[-1]
        R = [p.shape[0] for p in prev]
        total = int(sum(R))
        levels = [q for q in range(level0, max_levels + 1)]
        if not levels or total == 0:
            return [torch.from_numpy(p).to(self.device) for p in prev]
        # Every level's integer tables are planned on the host up front (the codes a level does not know yet are
        # placeholders) and uploaded; the device then patches the response rows with the codes as they are
        # sampled, so no level waits for a device->host copy and the GPU never idles between levels.
        n_lv = self.n_resp_levels
        off = np.concatenate(([0], np.cumsum(R))).astype(np.int32)
        known = np.concatenate(prev).astype(np.int64)                                          # (total, level0)
        ph = [np.concatenate([p, np.zeros((p.shape[0], n_lv - p.shape[1]), np.int64)], axis=1) for p in prev]
        plans, nplans = {}, {}

        def prepare(n):            # host planning of level n runs while the device is busy with level n - 1
            if n >= len(levels):
                return
            q = levels[n]
            resp_q = [x[:, :q] for x in ph]
            plan = self.plan(cond_text, lang_list, proms_list, resp_q, q)
            rows, _ = self.tail_rows(plan, R)
            plans[n] = (plan, rows, self.upload_plan(plan, rows))
            if cfg_strength > 0:
                nplan = self.plan(null_text, lang_list, None, resp_q, q)
                nrows, _ = self.tail_rows(nplan, R)
                nplans[n] = (nplan, nrows, self.upload_plan(nplan, nrows))

        prepare(0)
        counts = np.diff(off)
        row_item = np.repeat(np.arange(B, dtype=np.int32), counts)
        row_in = np.concatenate([np.arange(c, dtype=np.int32) for c in counts])
        ri, rn, poff, plen, known_d = self._upload_i32(row_item, row_in, off[:-1], counts, known)
        lib, stream = L.load(), L.stream_ptr(self.device)
        bases = [self._off[f"resp{l + 1}"] for l in range(n_lv - 1)]
        # integer tables patched by valle_scatter_cols: resp_idx = embedding rows of the codes so far, flat = the codes
        # themselves (int64, what the call returns), prev_col = the newest level (the sampler's `prev`)
        resp_idx = self._filled((total, 8), torch.int32, -1)
        flat = torch.empty((total, levels[-1] + 1), dtype=torch.int64, device=self.device)
        prev_col = torch.empty(total, dtype=torch.int32, device=self.device)
        for l in range(level0):
            L.check(lib.valle_scatter_cols(L.ptr(resp_idx), 0, 8, None, L.ptr(known_d), level0, total, l, l, 1, bases[l] if l < n_lv - 1 else 0, stream))
        L.check(lib.valle_scatter_cols(L.ptr(flat), 1, flat.shape[1], None, L.ptr(known_d), level0, total, 0, 0, level0, 0, stream))
        L.check(lib.valle_scatter_cols(L.ptr(prev_col), 0, 1, None, L.ptr(known_d), level0, total, 0, level0 - 1, 1, 0, stream))
        for n, q in enumerate(levels):
            self._set_lora_level(q)                                                            # ar_nar.py:539-540
            plan, rows, dev = plans.pop(n)
            def patch(d):                                                                      # the codes decoded since planning
                if self.sum_resp_levels:                                                       # sum of levels < q
                    L.check(lib.valle_scatter_cols(L.ptr(d[0]), 0, 8, L.ptr(d[3]), L.ptr(resp_idx), 8, total, 0, 0, q, 0, stream))
                else:                                                                          # level q - 1 alone (base.py:169-173)
                    L.check(lib.valle_scatter_cols(L.ptr(d[0]), 0, 8, L.ptr(d[3]), L.ptr(resp_idx), 8, total, 0, q - 1, 1, 0, stream))
            patch(dev)
            logits = self.forward_dense(plan, rows, dev=dev)
            prepare(n + 1)
            V = self.cls_tokens[plan.head]
            if cfg_strength > 0:
                nplan, nrows, ndev = nplans.pop(n)
                patch(ndev)
                null = self.forward_dense(nplan, nrows, dev=ndev)
                self.cfg_combine(logits, null, V, off, cfg_strength, cfg_rescale)
            sp.offset = 1000 + n
            ids, _ = self.sample_rows(logits, V, off, None, nar_mode=1, params=sp, dev=(ri, rn, prev_col, poff, plen))
            prev_col = ids
            L.check(lib.valle_scatter_cols(L.ptr(flat), 1, flat.shape[1], None, L.ptr(ids), 1, total, q, 0, 1, 0, stream))
            if q < n_lv - 1:
                L.check(lib.valle_scatter_cols(L.ptr(resp_idx), 0, 8, None, L.ptr(ids), 1, total, q, 0, 1, bases[q], stream))
        return [flat[off[i]:off[i + 1]] for i in range(B)]

    # ------------------------------------------------------------------------------- NAR-len
    def forward_nar_masked(self, task_list=None, phns_list=None, proms_list=None, resps_list=None, lang_list=None,
                           len_list=None, quant_levels=None, disable_tqdm=False, use_lora=None, **sampling_kwargs):
        """Level-0 iterative demasking (ar_nar.py:233-464)."""
        self._prepare()
        if use_lora is not None:
            self._call_state.use_lora = use_lora
        self._set_lora_level(0 if quant_levels is None else quant_levels[0])  # ar_nar.py:262-269
        lib = L.load()
        kw = convert_kwargs(dict(sampling_kwargs), "ar_")                     # (sic) ar_nar.py:282
        min_length = kw.pop("min_duration", 1)
        max_length = kw.pop("max_duration", 500)
        max_steps = kw.get("max_steps", 25)
        annealed = kw.get("annealed_sampling", True)
        temperature = kw.pop("temperature", 0.0)
        min_cfg = kw.get("minimum_cfg_strength", 2.5)
        cfg_strength = max(kw.get("cfg_strength", min_cfg), min_cfg)           # ar_nar.py:293-295, :319
        cfg_rescale = kw.pop("cfg_rescale", 0.75)
        start_noise, end_noise = kw.get("denoise_start", 0.0), kw.get("denoise_end", 1.0)
        max_steps = math.floor(max_steps * (end_noise - start_noise))
        remasking = kw.get("sampling_scores_remask", False)
        prefix_context = kw.pop("prefix_context", None)
        masked_only = kw.pop("sampling_scores_masked_only", False)
        trace = kw.pop("trace", None)          # list: receives (ids, scores) of every demasking step, packed rows (tests)
        if kw.get("vc_list") is not None:
            # ar_nar.py:366-377 writes ``resp.scatter(0, indices, half_mask)``: source element i lands at the i-th index of
            # topk's output, and at the first step every score is 1.0, so WHICH source code lands where is decided by the
            # order torch.topk returns tied entries in -- an internal of the torch build and device, not of the algorithm
            raise NotImplementedError("vc_list is not built on the HIP path (its result depends on torch.topk's order among tied scores)")
        if quant_levels is not None and any(q != 0 for q in quant_levels):
            raise NotImplementedError("demasking levels other than 0")
        lens = [clamp(int(l), min_length, max_length) for l in len_list]
        B, dev = len(lens), self.device
        off = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
        off_d = self._i32(off)
        total = int(off[-1])
        if start_noise > 0.0 and resps_list is not None:
            # denoising from an existing sequence (ar_nar.py:328-334): the given level-0 codes stay, and every position draws a
            # score of 0 with probability cos(start * pi / 2), else 1 -- from Python's global ``random``, as the reference does
            # (seed it the same way and the scores are the same).  The first step then masks the k lowest of these tied scores:
            # among ties valle_mask_lowest takes the lower index, torch.topk whatever its build does -- the masked SET may
            # differ from the reference's there, its size and the "lowest scores first" rule do not.
            import random
            given = _host_codes([r if r.dim() == 1 else r[:, :1] for r in resps_list])
            given = [g.reshape(-1)[:n] for g, n in zip(given, lens)]
            if any(len(g) != n for g, n in zip(given, lens)):
                raise ValueError("denoise_start: resps_list must hold len_list[i] frames per utterance")
            noise_p0 = math.cos(start_noise * math.pi * 0.5)
            sc_h = [[0.0 if random.random() < noise_p0 else 1.0 for _ in range(n)] for n in lens]
            codes = self._i32(np.concatenate(given))
            scores = self._h2d(np.concatenate([np.asarray(x, dtype=np.float32) for x in sc_h]))
        else:
            codes = self._filled((total,), torch.int32, self.mask_token)
            scores = self._filled(total, torch.float32, 1.0)
        is_masked = torch.empty(total, dtype=torch.uint8, device=dev)
        prev_codes = torch.empty(total, dtype=torch.int32, device=dev)
        null_text = self._null_text(phns_list)
        V = self.cls_tokens[self.cls_names.index("NAR:0:0")]
        stream = L.stream_ptr(dev)
        # the sequence layout does not change between demasking steps: both plans are built and uploaded once, and
        # every step patches the response rows' embedding ids with the current codes on the device
        zeros = [np.zeros(n, np.int64) for n in lens]
        if prefix_context is not None:                                        # ar_nar.py:321-325, :391-392: earlier lines'
            prefix_text, prefix_resps, _ = prefix_context                     # phonemes and level-0 codes in front, both passes
            phns_list = [torch.cat([torch.as_tensor(p)[:-1].cpu(), torch.as_tensor(t)[1:].cpu()]) for p, t in zip(prefix_text, phns_list)]
            pre = _host_codes(list(prefix_resps))
            zeros = [np.concatenate([(p if p.ndim == 1 else p[:, 0]).astype(np.int64), z]) for p, z in zip(pre, zeros)]
        plan = self.plan(phns_list, lang_list, proms_list, zeros, 0, timestep=True)
        rows, roff = self.tail_rows(plan, lens)
        pdev = self.upload_plan(plan, rows)
        resp_base = self._off[f"resp{self.emb_names.index('NAR:0:0')}"]
        if cfg_strength > 0:
            nplan = self.plan(null_text, lang_list, None, zeros, 0, timestep=True)
            nrows, _ = self.tail_rows(nplan, lens)
            ndev = self.upload_plan(nplan, nrows)
        counts = np.diff(roff)
        ri, rn, poff, plen = self._upload_i32(np.repeat(np.arange(B, dtype=np.int32), counts),
                                              np.concatenate([np.arange(c, dtype=np.int32) for c in counts]), roff[:-1], counts)
        for step, t in enumerate(torch.linspace(start_noise, end_noise, max_steps)):
            noise_p = math.cos(t * math.pi * 0.5)                               # float32 timestep, as the reference
            mask_p = noise_p + (1.0 / (max_steps * 2) if remasking else 0)
            ks = self._i32([clamp(int(mask_p * n), 1, n - step) for n in lens])  # ar_nar.py:358
            L.check(lib.valle_scatter_cols(L.ptr(prev_codes), 0, 1, None, L.ptr(codes), 1, total, 0, 0, 1, 0, stream))   # the step's `prev`
            L.check(lib.valle_mask_lowest(L.ptr(scores), L.ptr(codes), L.ptr(is_masked), L.ptr(off_d), L.ptr(ks), B,
                                          max(lens), self.mask_token, stream))
            temp = temperature * (1.0 - float(t)) if annealed else temperature
            L.check(lib.valle_scatter_cols(L.ptr(pdev[0]), 0, 8, L.ptr(pdev[3]), L.ptr(codes), 1, total, 0, 0, 1, resp_base, stream))
            logits = self.forward_dense(plan, rows, dev=pdev)
            if cfg_strength > 0:
                L.check(lib.valle_scatter_cols(L.ptr(ndev[0]), 0, 8, L.ptr(ndev[3]), L.ptr(codes), 1, total, 0, 0, 1, resp_base, stream))
                null = self.forward_dense(nplan, nrows, dev=ndev)
                self.cfg_combine(logits, null, V, roff, cfg_strength, cfg_rescale)
            sdev = (ri, rn, prev_codes, poff, plen)
            sp1 = self.sample_params(temperature=temp, **kw)
            sp1.offset = 2000 + step
            ids, _ = self.sample_rows(logits, V, roff, None, nar_mode=1, params=sp1, dev=sdev)
            # the reference's second ("unfiltered") sample call runs on the tensors the first one processed in place: the
            # repetition penalty and -- unless min-p, which filters out of place, sits in between -- top-k / top-p act twice
            refilter = (kw.get("top_k", -100) > 0 or kw.get("top_p", 1.0) < 1.0) and kw.get("min_p", 0.0) <= 0.0
            sp2 = self.sample_params(temperature=0.0, rep_pen_passes=2 if kw.get("repetition_penalty", 1.0) != 1.0 else 1,
                                     filter_passes=2 if refilter else 1, **kw)
            _, sc = self.sample_rows(logits, V, roff, None, nar_mode=1, params=sp2, dev=sdev)
            if trace is not None:
                trace.append((ids.clone(), sc.clone()))
            L.check(lib.valle_narlen_commit(L.ptr(codes), L.ptr(scores), L.ptr(is_masked), L.ptr(ids), L.ptr(sc), total,
                                            1 if masked_only else 0, stream))                  # ar_nar.py:457-462
        codes = codes.to(torch.int16)              # the reference's dtype for this result (ar_nar.py:457)
        return [codes[off[i]:off[i + 1]] for i in range(B)]
