"""Host side of the version >= 7 model family on the HIP path.

Mirrors ``vall_e.models.base_v2.Base_V2`` (reference models/base_v2.py:248-1411) for inference: same constructor
arguments and state-dict names, ``inputs`` / ``forward_super`` / ``sample`` with the reference's argument meaning.  A
frame of L codes enters through the AudioEncoder (per-level table + level position, LayerNorm, residual FFN, softmax-
weighted level sum: base_v2.py:86-159), the llama stack runs without its final norm, the AudioDecoder turns a hidden row
into L x (n_audio_tokens + 1) logits (per level LayerNorm, FFN, head: base_v2.py:164-225).  All arithmetic is in
libvalle_hip.so (csrc/v2.hip, valle_gemm, valle_forward_layers); torch allocates and carries streams.
"""
from __future__ import annotations

import ctypes as C
import threading

import numpy as np
import torch

from .. import _lib as L
from ..config import Model as ModelConfig
from ..weights import state_dict_spec
from .base import TASK_SYMMAP, Base, Logits, RawText, Sampled, _np, _round_up

LN_EPS = 1e-5                      # nn.LayerNorm's default


class PlanV2:
    """Packed batch of version >= 7 sequences: table rows (idx, -1 where a code frame is encoded), position ids, and the
    packed rows that the prompt / response frames land in."""

    __slots__ = ("idx", "pos", "cu", "lens", "B", "M", "max_T", "prom_dst", "prom_codes", "resp_dst", "resp_lens", "out_len", "task")

    def __init__(self, idx, pos, lens, prom_dst, prom_codes, resp_dst, resp_lens, out_len, task):
        self.idx, self.pos, self.lens = idx, pos, lens
        self.B = len(lens)
        self.cu = np.zeros(self.B + 1, dtype=np.int32)
        np.cumsum(lens, out=self.cu[1:])
        self.M = int(self.cu[-1])
        self.max_T = int(max(lens))
        self.prom_dst, self.prom_codes, self.resp_dst, self.resp_lens = prom_dst, prom_codes, resp_dst, resp_lens
        self.out_len, self.task = out_len, task


class Base_V2(Base):
    def __init__(self, n_phn_tokens=256, n_audio_tokens=1024, n_text_tokens=8575, d_model=512, d_ffn=4, n_heads=8,
                 n_layers=12, p_dropout=0.1, n_experts=1, l_padding=0, training=True, attention=None, config=None):
        if config is None:
            raise ValueError("config is required")
        if not isinstance(config, ModelConfig):
            config = ModelConfig.from_dict({k: getattr(config, k) for k in
                                            ("name", "version", "size", "tasks", "langs", "tones", "arch_type", "attention",
                                             "dropout", "capabilities") if hasattr(config, k)} |
                                           {"experimental": getattr(config, "experimental", None)})
        exp = config.experimental
        if config.arch_type != "llama":
            raise NotImplementedError(f"arch_type {config.arch_type!r}: only the llama stack is built")
        if config.version < 7:
            raise ValueError("Base_V2 is the version >= 7 model (models/__init__.py:53-60 of the reference)")
        if exp.causal_size != 1:
            raise NotImplementedError("causal_size != 1")
        if n_experts != 1:
            raise NotImplementedError("MoE")
        if d_model // n_heads != 64:
            raise NotImplementedError("head_dim must be 64")
        # the switches of the version >= 7 forward that are not built (base_v2.py:426-439, :1193-1228)
        if exp.monolithic_audio_encoder:
            raise NotImplementedError("monolithic_audio_encoder")
        if exp.use_segmented_attention_mask or exp.use_sliding_attention_mask:
            raise NotImplementedError("segmented / sliding attention masks (base_v2.py:1193-1228)")
        if not (exp.use_audio_encoder_ffn and exp.use_audio_encoder_norm and exp.use_audio_encoder_level_weights):
            raise NotImplementedError("AudioEncoder without its norm, FFN or level weights")
        self.config = config
        self.training = False
        self.n_phn_tokens, self.n_audio_tokens, self.n_text_tokens = n_phn_tokens, n_audio_tokens, n_text_tokens
        self.d_model, self.n_heads, self.n_layers, self.d_ffn = d_model, n_heads, n_layers, d_ffn
        self.n_resp_levels = config.resp_levels
        self.n_max_levels = config.max_levels
        self.capabilities = list(config.capabilities)
        self.stop_token = n_audio_tokens                                   # base_v2.py:378-379
        self.mask_token = self.stop_token + 1
        self.causal = True
        self.version = config.version
        self.causal_size = exp.causal_size
        self.unified_position_ids = exp.unified_position_ids
        self.noncausal_masks = exp.noncausal_masks
        self.masking_ratio = exp.masking_ratio
        self.max_position_embeddings = exp.max_position_embeddings
        self.per_level_normalization = exp.per_level_normalization
        self.len_use_logits = exp.len_use_logits
        self.audio_frames_per_second = 75                                   # cfg.dataset.frames_per_second (ar_nar_v2.py:632)
        self.device = torch.device("cpu")
        self.dtype = torch.float32
        self._off, self._table_order = self._table_offsets()
        self._sd = None
        self._handles = {}
        self._lora = None
        self._bufs = {}
        self._ws = None
        self._call_state = threading.local()
        self._keep = []
        self._enc = {}

    # ------------------------------------------------------------------ weight preparation
    def _table_offsets(self):
        spec = self.state_dict_spec()
        order = [("phn", "phn_emb.weight"), ("text", "text_emb.weight"), ("lang", "langs_emb.weight"), ("task", "tasks_emb.weight"),
                 ("tone", "tones_emb.weight"), ("sep", "sep"), ("len", "len_emb")]
        order += [(f"prom{l}", f"proms_emb.embs.{l}.weight") for l in range(self.n_resp_levels)]
        order += [(f"resp{l}", f"resps_emb.embs.{l}.weight") for l in range(self.n_resp_levels)]
        order = [(n, k) for n, k in order if k in spec]
        off, row = {}, 0
        for name, key in order:
            off[name] = row
            row += 1 if len(spec[key]) == 1 else spec[key][0]
        return off, order

    def load_lora(self, *a, **k):
        raise NotImplementedError("LoRA on the version >= 7 model")

    def _prepare(self):
        if self._handles:
            return
        if self._sd is None:
            raise RuntimeError("load_state_dict first")
        if self.device.type != "cuda":
            raise L.ValleHipError("the HIP path needs a GPU device (model.to('cuda')); there is no CPU fallback")
        lib = L.load()
        sd, dev, dt = self._sd, self.device, self.dtype
        d, H, ff, Lv = self.d_model, self.n_heads, self.d_model * self.d_ffn, self.n_resp_levels
        b = self._bufs
        b["emb"] = torch.cat([sd[key].reshape(-1, d) for _, key in self._table_order]).to(dt).contiguous().to(dev)
        Dh = d // H
        inv_freq = 1.0 / (10000.0 ** (torch.arange(0, Dh, 2, dtype=torch.int64).float() / Dh))
        pos = torch.arange(self.max_position_embeddings, dtype=torch.float32)
        freqs = (inv_freq[None, :, None] @ pos[None, None, :]).transpose(1, 2)[0]
        b["rope"] = torch.stack((freqs.cos(), freqs.sin()), dim=-1).to(dev).contiguous()
        layers = self._build_layers(sd)
        # per_level_normalization moves the stack's final norm into the decoder (base_v2.py:495): the descriptor then holds a
        # vector of ones that no call applies (valle_forward_layers(final_norm=0))
        norm = sd["model.norm.weight"] if "model.norm.weight" in sd else torch.ones(d)
        b["norm"] = norm.to(dt).contiguous().to(dev)
        cfg = L.Config(d, H, self.n_layers, ff, L.dtype_code(dt), 1e-6, self.max_position_embeddings)
        handle = C.c_void_p()
        L.check(lib.valle_model_create(C.byref(cfg), L.ptr(b["emb"]), L.ptr(b["rope"]), layers, L.ptr(b["norm"]), 0,
                                       None, None, None, None, C.byref(handle)))
        self._handles[False] = handle
        self._keep = [layers]
        up = lambda t: t.to(dt).contiguous().to(dev)
        for name in ("proms_emb", "resps_emb"):
            kind = "prom" if name == "proms_emb" else "resp"
            n_rows = [int(sd[f"{name}.embs.{l}.weight"].shape[0]) for l in range(Lv)]
            lw = torch.softmax(sd[f"{name}.level_weights"].to(dt), dim=0)                 # base_v2.py:156, in the storage dtype
            self._enc[name] = dict(
                row0=(C.c_int32 * Lv)(*[self._off[f"{kind}{l}"] for l in range(Lv)]), n_rows=(C.c_int32 * Lv)(*n_rows),
                pos=up(sd[f"{name}.pos_embedding"].reshape(Lv, d)), nw=up(sd[f"{name}.norm.weight"]), nb=up(sd[f"{name}.norm.bias"]),
                w0=up(sd[f"{name}.proj.0.weight"]), b0=up(sd[f"{name}.proj.0.bias"]),
                w2=up(sd[f"{name}.proj.2.weight"]), b2=up(sd[f"{name}.proj.2.bias"]),
                lw=lw.float().contiguous().to(dev), hdim=int(sd[f"{name}.proj.0.weight"].shape[0]))
        b["one"] = torch.ones(1, dtype=torch.float32).to(dev)
        self._dec = []
        for l in range(Lv):
            p = f"audio_decoder.head.{l}."
            self._dec.append(dict(
                nw=up(sd[p + "0.weight"]) if self.per_level_normalization else None,
                nb=up(sd[p + "0.bias"]) if self.per_level_normalization else None,
                w1=up(sd[p + "1.weight"]), b1=up(sd[p + "1.bias"]), w3=up(sd[p + "3.weight"]), b3=up(sd[p + "3.bias"]),
                w4=up(sd[p + "4.weight"]), b4=sd[p + "4.bias"].to(dt).float().contiguous().to(dev),
                hdim=int(sd[p + "1.weight"].shape[0])))
        # levels of one shape (the reference's AudioDecoder builds them so, base_v2.py:199-206) are also kept as ONE tensor per
        # operand, level-major, so that a few-rows call runs a linear of all levels as one grouped launch (decode below);
        # the per-level entries become views of it
        self._decg = None
        if len({(w["hdim"], tuple(w["w4"].shape)) for w in self._dec}) == 1:
            g = {k: torch.stack([w[k] for w in self._dec]).contiguous() for k in ("w1", "b1", "w3", "b3", "w4", "b4")}
            if self.per_level_normalization:
                g["nw"], g["nb"] = (torch.stack([w[k] for w in self._dec]).contiguous() for k in ("nw", "nb"))
            for l, w in enumerate(self._dec):
                for k in g:
                    w[k] = g[k][l]
            self._decg = g
        self._aux = {n: (up(sd[f"{n}.head.weight"]), sd[f"{n}.head.bias"].to(dt).float().contiguous().to(dev))
                     for n in ("len_decoder", "phn_decoder", "text_decoder")}

    # -------------------------------------------------------------------- sequence planning
    def inputs(self, phns_list=None, text_list=None, proms_list=None, resps_list=None, lang_list=None, tone_list=None,
               len_list=None, task_list=None, time_list=None, quant_levels=None):
        """The reference's per-utterance (name, value) tuples (base_v2.py:580-723) for the ``tts`` and ``len`` layouts."""
        first = next(l for l in (phns_list, text_list, proms_list, resps_list) if l)
        B = len(first)
        out = [[] for _ in range(B)]
        for i in range(B):
            q = int(quant_levels[i]) if quant_levels is not None else 0
            task = task_list[i] if task_list is not None else "tts"
            timestep = time_list[i] if time_list is not None else None
            if task in ("stt", "phn", "text"):
                raise NotImplementedError(f"task {task!r}: only the 'tts' family and 'len' are built")
            if task != "len" and task not in TASK_SYMMAP:
                raise Exception(f"Unrecognized task: {task}")
            out[i].append(("task", task))
            if phns_list is not None and phns_list[i] is not None:
                out[i].append(("phn", phns_list[i]))
            elif text_list is not None and text_list[i] is not None:
                out[i].append(("text", text_list[i]))
            if lang_list is not None and lang_list[i] is not None:
                out[i].append(("lang", lang_list[i]))
            if proms_list is not None and proms_list[i] is not None:
                out[i].append(("prom", proms_list[i]))
            if task == "len":
                if "tone" in self.capabilities and tone_list is not None and tone_list[i] is not None:
                    out[i].append(("tone", tone_list[i]))
                n = resps_list[i].shape[0] if resps_list is not None else 0
                out[i].append(("len", torch.tensor([n])))
                out[i].append(("classifier_level", "len"))
                continue
            if tone_list is not None and tone_list[i] is not None:
                out[i].append(("tone", tone_list[i]))
            if resps_list is not None and resps_list[i] is not None:
                out[i].append(("resp", resps_list[i]))
            out[i].append(("classifier_level", f"{'N' if timestep is not None else ''}AR:{q}:{q}"))
        return out

    def plan(self, phns_list, lang_list, proms_list, resp_lens, tone_list=None, task="tts", pad_pos=None) -> PlanV2:
        """Layout phn | lang | prom | tone | resp (``len``: ... | len) with a ``sep`` row between segments
        (base_v2.py:622-678, :61-65, :797); position ids restart per segment, segments other than the task's output count
        their trailing sep (base_v2.py:860-867).  resp_lens[i] = frames of utterance i's response (None: no resp segment).
        pad_pos[i] (optional; None = no row): position id of ONE zero-embedding padding row appended to utterance i -- the last
        row of the reference's right-padded batch (forward_super, ``len`` task)."""
        off = self._off
        out_kind = "len" if task == "len" else "resp"
        B = len(resp_lens)
        Lv = self.n_resp_levels
        idx_all, pos_all, lens, out_len = [], [], [], []
        prom_dst, prom_codes, resp_dst = [], [], []
        row_base = 0
        sep_row = np.full((1, 8), -1, dtype=np.int32)
        sep_row[0, 0] = off["sep"]

        def rows1(base, ids):
            r = np.full((len(ids), 8), -1, dtype=np.int32)
            r[:, 0] = base + ids
            return r

        for i in range(B):
            segs = []                                  # (kind, rows, n position tokens)
            p = phns_list[i] if phns_list is not None else None
            if isinstance(p, RawText):
                segs.append(("text", rows1(off["text"], _np(p.ids))))
            elif p is not None:
                segs.append(("phn", rows1(off["phn"], _np(p))))
            if lang_list is not None and lang_list[i] is not None and "lang" in off:
                segs.append(("lang", rows1(off["lang"], _np(lang_list[i]).reshape(-1))))
            if proms_list is not None and proms_list[i] is not None:
                plist = proms_list[i] if isinstance(proms_list[i], (list, tuple)) else [proms_list[i]]
                parts, n_tok = [], 0
                for prom in plist:
                    if prom is None:
                        continue
                    if isinstance(prom, str):           # a task token inside the prompt (base_v2.py:731-735)
                        parts.append((False, rows1(off["task"], np.array([TASK_SYMMAP[prom]]))))
                        continue
                    c = _np(prom)
                    if c.ndim != 2 or c.shape[1] != Lv:
                        raise ValueError(f"prompt codes must be (frames, {Lv})")
                    n_tok += c.shape[0]
                    parts.append((True, c))
                segs.append(("prom", parts, n_tok))
            if tone_list is not None and tone_list[i] is not None and "tone" in off and (task != "len" or "tone" in self.capabilities):
                segs.append(("tone", rows1(off["tone"], _np(tone_list[i]).reshape(-1))))
            if task == "len":
                segs.append(("len", rows1(off["len"], np.array([0]))))
            elif resp_lens[i] is not None:
                segs.append(("resp", int(resp_lens[i])))
            parts, pos, row = [], [], 0
            for j, seg in enumerate(segs):
                kind = seg[0]
                if j:
                    parts.append(sep_row)
                    row += 1
                if kind == "prom":
                    for is_codes, c in seg[1]:
                        if not is_codes:                # a task-token table row
                            parts.append(c)
                            row += 1
                        else:
                            n = c.shape[0]
                            parts.append(np.full((n, 8), -1, dtype=np.int32))
                            prom_dst.append(np.arange(row_base + row, row_base + row + n, dtype=np.int32))
                            prom_codes.append(c)
                            row += n
                    n_tok = seg[2]
                elif kind == "resp":
                    n = seg[1]
                    parts.append(np.full((n, 8), -1, dtype=np.int32))
                    resp_dst.append(np.arange(row_base + row, row_base + row + n, dtype=np.int32))
                    row += n
                    n_tok = n
                else:
                    parts.append(seg[1])
                    row += seg[1].shape[0]
                    n_tok = seg[1].shape[0]
                pos.append(np.arange(n_tok + (0 if kind == out_kind else 1), dtype=np.int32))
            idx = np.concatenate(parts)
            pos = np.concatenate(pos)[: idx.shape[0]]
            if pos.shape[0] < idx.shape[0]:                                           # base_v2.py:865-867: tail filled with 1
                pos = np.concatenate([pos, np.ones(idx.shape[0] - pos.shape[0], np.int32)])
            if self.unified_position_ids:
                pos = np.arange(idx.shape[0], dtype=np.int32)
            if pad_pos is not None and pad_pos[i] is not None:
                idx = np.concatenate([idx, np.full((1, 8), -1, dtype=np.int32)])
                pos = np.concatenate([pos, np.array([pad_pos[i]], dtype=np.int32)])
            if idx.shape[0] > self.max_position_embeddings:
                raise ValueError(f"sequence of {idx.shape[0]} rows exceeds max_position_embeddings ({self.max_position_embeddings})")
            idx_all.append(idx), pos_all.append(pos), lens.append(idx.shape[0])
            last = segs[-1]
            out_len.append((last[1] if last[0] == "resp" else last[1].shape[0]) if last[0] == out_kind else 0)
            if task != "len" and resp_lens[i] is None:
                resp_dst.append(np.zeros(0, np.int32))
            row_base += idx.shape[0]
        cat = lambda xs, dt_: np.concatenate(xs) if xs else np.zeros(0, dt_)
        return PlanV2(np.concatenate(idx_all), np.concatenate(pos_all), lens, cat(prom_dst, np.int32),
                      np.concatenate(prom_codes) if prom_codes else np.zeros((0, Lv), np.int64), cat(resp_dst, np.int32),
                      [0 if n is None else int(n) for n in resp_lens], out_len, task)

    # ------------------------------------------------------------------------ encoder / stack / decoder
    def encode_frames(self, name, codes_lm, ld, n):
        """AudioEncoder.forward up to (not including) the level sum for n frames (base_v2.py:138-149): codes_lm = device
        int32, level-major [L][ld].  -> y (n * L, d): row i * L + l = frame i, level l."""
        lib, e, dt = L.load(), self._enc[name], L.dtype_code(self.dtype)
        d, Lv, stream = self.d_model, self.n_resp_levels, L.stream_ptr(self.device)
        rows = n * Lv
        y0 = torch.empty((rows, d), dtype=self.dtype, device=self.device)
        if n == 0:
            return y0
        src = torch.empty(rows, dtype=torch.int32, device=self.device)
        L.check(lib.valle_v2_encode_rows(L.ptr(self._bufs["emb"]), e["row0"], e["n_rows"], Lv, L.ptr_any(codes_lm), int(ld), int(n),
                                         L.ptr(e["pos"]), L.ptr(e["nw"]), L.ptr(e["nb"]), L.ptr(y0), L.ptr(src), d, LN_EPS, dt, stream))
        h = torch.empty((rows, e["hdim"]), dtype=self.dtype, device=self.device)
        gemm = self._gemm(rows)
        L.check(gemm(L.ptr(y0), L.ptr(e["w0"]), L.ptr(e["b0"]), None, L.ptr(h), rows, e["hdim"], d, e["hdim"], L.EPI_GELU, dt, stream))
        y = torch.empty((rows, d), dtype=self.dtype, device=self.device)
        L.check(gemm(L.ptr(h), L.ptr(e["w2"]), L.ptr(e["b2"]), L.ptr(y0), L.ptr(y), rows, d, e["hdim"], d, L.EPI_RESID_BIAS, dt, stream))
        return y

    SKINNY_ROWS = 256                 # up to this many rows the few-rows GEMM is used in bf16 (a decode step's linears)

    def _gemm(self, rows):
        lib = L.load()
        return lib.valle_gemm_skinny if (self.dtype == torch.bfloat16 and rows <= self.SKINNY_ROWS and self.d_model % 128 == 0) else lib.valle_gemm

    def place_frames(self, name, y, n, dst, x):
        """The encoder's weighted level sum (base_v2.py:153-157) of n frames into rows dst (device int32) of x."""
        if n == 0:
            return
        L.check(L.load().valle_v2_level_sum(L.ptr(y), L.ptr(self._enc[name]["lw"]), self.n_resp_levels, int(n), L.ptr_any(dst) if dst is not None else None,
                                            L.ptr_any(x), self.d_model, L.dtype_code(self.dtype), L.stream_ptr(self.device)))

    def copy_rows(self, src, n, dst, x):
        """x[dst[i]] = src[i] (already summed encoder rows kept from an earlier step)."""
        if n == 0:
            return
        L.check(L.load().valle_v2_level_sum(L.ptr_any(src), L.ptr(self._bufs["one"]), 1, int(n), L.ptr_any(dst), L.ptr_any(x),
                                            self.d_model, L.dtype_code(self.dtype), L.stream_ptr(self.device)))

    def _codes_lm(self, codes):
        """host (n, L) codes -> device int32 level-major [L][n]."""
        return self._i32(np.ascontiguousarray(np.asarray(codes, dtype=np.int64).T))

    def build_x(self, plan: PlanV2, resp_codes=None):
        """Embeddings of a packed plan: table rows, encoded prompt frames, and (resp_codes = host (sum resp_lens, L) codes)
        encoded response frames.  -> (x (M, d), dev) with dev = device copies of (pos, cu, prom_dst, resp_dst)."""
        self._prepare()
        lib, dt = L.load(), L.dtype_code(self.dtype)
        idx, pos, cu, pdst, rdst = self._upload_i32(plan.idx, plan.pos, plan.cu, plan.prom_dst, plan.resp_dst)
        x = torch.empty((plan.M, self.d_model), dtype=self.dtype, device=self.device)
        L.check(lib.valle_embed_sum(L.ptr(self._bufs["emb"]), L.ptr(idx), L.ptr(x), plan.M, self.d_model, dt, L.stream_ptr(self.device)))
        n_prom = int(plan.prom_codes.shape[0])
        if n_prom:
            y = self.encode_frames("proms_emb", self._codes_lm(plan.prom_codes), n_prom, n_prom)
            self.place_frames("proms_emb", y, n_prom, pdst, x)
        if resp_codes is not None and len(resp_codes):
            n = int(len(resp_codes))
            y = self.encode_frames("resps_emb", self._codes_lm(resp_codes), n, n)
            self.place_frames("resps_emb", y, n, rdst, x)
        return x, (pos, cu, pdst, rdst)

    def is_causal(self, timestep, task="tts"):
        """base_v2.py:1182-1185: with noncausal_masks only phn / text / AR:l:l levels stay causal."""
        if not self.noncausal_masks:
            return 1
        return 0 if (timestep or task == "len") else 1

    def forward_hidden(self, plan: PlanV2, x, dev, out_rows, causal=1, kv_lens=None, cache=None):
        """The llama stack on embeddings x -> the out_rows (device int32) of the last hidden state, (n_out, d).
        kv_lens (device int32, per utterance): rows behind kv_lens[b] are padding -- queries, never keys.
        cache = (k, v, cache_T, layer_stride, slot, pos0): every row's rotated key and its value are also written there."""
        lib = L.load()
        n_out = int(out_rows.numel())
        hidden = torch.empty((n_out, self.d_model), dtype=self.dtype, device=self.device)
        if n_out == 0:
            return hidden
        ws = self._workspace(lib.valle_dense_workspace_bytes(self._handle, plan.M, n_out))
        kc, vc, cache_T, stride, slot, pos0 = cache if cache is not None else (None, None, 0, 0, None, None)
        L.check(lib.valle_forward_layers(self._handle, L.ptr(x), L.ptr(dev[0]), L.ptr(dev[1]), plan.B, plan.M, plan.max_T, int(causal),
                                         L.ptr(kc), L.ptr(vc), cache_T, stride, L.ptr(slot), L.ptr(pos0), L.ptr_any(out_rows), n_out,
                                         0 if self.per_level_normalization else 1, L.ptr(kv_lens), L.ptr(hidden), L.ptr(ws), ws.numel(),
                                         L.stream_ptr(self.device)))
        return hidden

    @property
    def audio_vocab(self):
        return self.n_audio_tokens + 1

    def decode(self, hidden):
        """AudioDecoder.forward (base_v2.py:215-225): (n, d) -> fp32 logits (L, n, ld), the first n_audio_tokens + 1 columns valid."""
        lib, dt, stream = L.load(), L.dtype_code(self.dtype), L.stream_ptr(self.device)
        n, d, V = int(hidden.shape[0]), self.d_model, self.audio_vocab
        ld = _round_up(V, 8)
        logits = torch.empty((self.n_resp_levels, n, ld), dtype=torch.float32, device=self.device)
        if n == 0:
            return logits
        gemm = self._gemm(n)
        g, Lv = self._decg, self.n_resp_levels
        hd = self._dec[0]["hdim"]
        if g is not None and gemm is lib.valle_gemm_skinny and hd % 128 == 0:
            # a decode step's rows: each linear of all levels in one launch (4 launches instead of 4 per level; every level's
            # arithmetic is the per-level call's)
            src, a_gs = hidden, 0
            if "nw" in g:
                src, a_gs = torch.empty((Lv, n, d), dtype=self.dtype, device=self.device), n * d
                L.check(lib.valle_layernorm_grouped(L.ptr(hidden), L.ptr(g["nw"]), L.ptr(g["nb"]), L.ptr(src), n, d, Lv, LN_EPS, dt, stream))
            t1 = torch.empty((Lv, n, hd), dtype=self.dtype, device=self.device)
            t2 = torch.empty((Lv, n, d), dtype=self.dtype, device=self.device)
            L.check(lib.valle_gemm_skinny_grouped(L.ptr(src), a_gs, L.ptr(g["w1"]), hd * d, L.ptr(g["b1"]), hd, None, 0, L.ptr(t1), n * hd,
                                                  Lv, n, hd, d, hd, L.EPI_GELU, dt, stream))
            L.check(lib.valle_gemm_skinny_grouped(L.ptr(t1), n * hd, L.ptr(g["w3"]), d * hd, L.ptr(g["b3"]), d, None, 0, L.ptr(t2), n * d,
                                                  Lv, n, d, hd, d, L.EPI_STORE, dt, stream))
            L.check(lib.valle_gemm_skinny_grouped(L.ptr(t2), n * d, L.ptr(g["w4"]), V * d, L.ptr(g["b4"]), V, None, 0, L.ptr(logits), n * ld,
                                                  Lv, n, V, d, ld, L.EPI_F32OUT, dt, stream))
            return logits
        t0 = torch.empty((n, d), dtype=self.dtype, device=self.device)
        t2 = torch.empty((n, d), dtype=self.dtype, device=self.device)
        for l, w in enumerate(self._dec):
            t1 = torch.empty((n, w["hdim"]), dtype=self.dtype, device=self.device) if l == 0 or t1.shape[1] != w["hdim"] else t1
            src = hidden
            if w["nw"] is not None:
                L.check(lib.valle_layernorm(L.ptr(hidden), L.ptr(w["nw"]), L.ptr(w["nb"]), L.ptr(t0), n, d, LN_EPS, dt, stream))
                src = t0
            L.check(gemm(L.ptr(src), L.ptr(w["w1"]), L.ptr(w["b1"]), None, L.ptr(t1), n, w["hdim"], d, w["hdim"], L.EPI_GELU, dt, stream))
            L.check(gemm(L.ptr(t1), L.ptr(w["w3"]), L.ptr(w["b3"]), None, L.ptr(t2), n, d, w["hdim"], d, L.EPI_STORE, dt, stream))
            L.check(gemm(L.ptr(t2), L.ptr(w["w4"]), L.ptr(w["b4"]), None, L.ptr_any(logits[l]), n, V, d, ld, L.EPI_F32OUT, dt, stream))
        return logits

    def aux_decode(self, name, hidden):
        """AuxDecoder.forward (base_v2.py:233-246): one nn.Linear -> fp32 logits."""
        lib, dt = L.load(), L.dtype_code(self.dtype)
        w, bias = self._aux[name]
        n, V = int(hidden.shape[0]), int(w.shape[0])
        ld = _round_up(V, 8)
        out = torch.empty((n, ld), dtype=torch.float32, device=self.device)
        L.check(lib.valle_gemm(L.ptr(hidden), L.ptr(w), L.ptr(bias), None, L.ptr(out), n, V, self.d_model, ld, L.EPI_F32OUT, dt,
                               L.stream_ptr(self.device)))
        return out[:, :V]

    def cfg_combine_cols(self, logits, null, row_off_dev, B, strength, rescale):
        """cfg_logits on (levels, rows, vocab) logits (samplers.py:177-191), in place on ``logits``."""
        Lv, rows, ld = logits.shape
        L.check(L.load().valle_cfg_combine_cols(L.ptr(logits), L.ptr(null), ld, self.audio_vocab, Lv, rows, L.ptr(row_off_dev), int(B),
                                                float(strength), float(rescale), L.stream_ptr(self.device)))
        return logits

    # ----------------------------------------------------------------------------- sampling
    def sample_params_v2(self, *, temperature=1.0, top_k=-100, top_p=1.0, min_p=0.0, top_no=1.0, seed=0, offset=0, **unused):
        """Base_V2.sample reads temperature, top_k, top_p, min_p and top_no only (base_v2.py:1386-1404); the other kwargs of
        its signature (penalties, beam, mirostat, DRY, min_temperature) are read into locals and never used."""
        return self.sample_params(temperature=temperature, min_temperature=temperature, top_k=top_k, top_p=top_p, min_p=min_p,
                                  top_no=top_no, seed=seed, offset=offset, score_unscaled=1)

    def sample_flat(self, logits2d, rows, params, ids_out=None):
        """valle_sample over fp32 rows (rows, ld) -> (ids int32, scores fp32)."""
        ids = ids_out if ids_out is not None else torch.empty(max(rows, 1), dtype=torch.int32, device=self.device)
        scores = torch.empty(max(rows, 1), dtype=torch.float32, device=self.device)
        L.check(L.load().valle_sample(L.ptr_any(logits2d), int(logits2d.shape[-1]), self.audio_vocab, int(rows), None, None, None, None, None,
                                      1, C.byref(params), L.ptr_any(ids), L.ptr(scores), L.stream_ptr(self.device)))
        return ids[:rows], scores[:rows]

    def sample(self, logits, prev_list=None, len_list=None, **sampling_kwargs):
        """``Base_V2.sample`` (base_v2.py:1335-1411): logits = per-utterance (rows, vocab) or (levels, rows, vocab) tensors;
        the last len(prev) (or len_list[i], or causal_size) rows of each are sampled.  Scores are returned for the
        non-causal form only, as in the reference."""
        B = len(logits)
        causal = False
        if prev_list is not None:
            counts = [int(p.shape[0]) for p in prev_list]
        elif len_list is not None:
            counts = [int(n) for n in len_list]
        else:
            counts, causal = [self.causal_size] * B, True
        cut = [l[..., -n:, :] if n > 0 else l[..., :0, :] for l, n in zip(logits, counts)]
        V = int(cut[0].shape[-1])
        ld = _round_up(V, 8)
        flat_rows = [c.reshape(-1, V) for c in cut]
        n_rows = [int(f.shape[0]) for f in flat_rows]
        flat = torch.zeros((sum(n_rows), ld), dtype=torch.float32, device=self.device)
        flat[:, :V] = torch.cat([f.to(self.device, torch.float32) for f in flat_rows])
        keys = ("temperature", "top_k", "top_p", "min_p", "top_no", "seed", "offset")
        sp = self.sample_params_v2(**{k: v for k, v in sampling_kwargs.items() if k in keys})
        if V != self.audio_vocab:
            raise NotImplementedError("Base_V2.sample on logits of another head than the audio decoder's")
        ids, sc = self.sample_flat(flat, sum(n_rows), sp)
        offs = np.concatenate(([0], np.cumsum(n_rows)))
        res = [ids[offs[i]:offs[i + 1]].long().reshape(cut[i].shape[:-1]) for i in range(B)]
        scores = [] if causal else [sc[offs[i]:offs[i + 1]].reshape(cut[i].shape[:-1]) for i in range(B)]
        return Sampled(res, cut, scores, None)

    # ---------------------------------------------------------------------- Base_V2.forward
    def forward_super(self, inputs, quant_levels=None, state=None, output_attentions=False, output_hidden_states=False):
        """Base_V2.forward (base_v2.py:1138-1333) on ``inputs()`` tuples: every row's logits, (levels, rows, vocab) per
        utterance; for the ``len`` task the predicted seconds per utterance (base_v2.py:1296-1308)."""
        if output_hidden_states or output_attentions or state is not None:
            raise NotImplementedError("attentions / hidden states / external state are not exposed on the version >= 7 path")
        self._prepare()
        get = lambda item, k: next((v for n, v in item if n == k), None)
        tasks = {get(it, "task") for it in inputs}
        levels = [get(it, "classifier_level") for it in inputs]
        if len(tasks) != 1:
            raise NotImplementedError("one task per batch")
        task = "len" if tasks.pop() == "len" else "tts"
        timestep = all(l.startswith("NAR:") for l in levels)
        if not timestep and any(l.startswith("NAR:") for l in levels):
            raise NotImplementedError("one classifier level kind per batch")
        phn = [RawText(get(it, "text")) if get(it, "phn") is None and get(it, "text") is not None else get(it, "phn") for it in inputs]
        col = lambda k: [get(it, k) for it in inputs] if any(get(it, k) is not None for it in inputs) else None
        resps = [get(it, "resp") for it in inputs]
        resp_lens = [None if r is None else int(r.shape[0]) for r in resps]
        args = (phn if any(p is not None for p in phn) else None, col("lang"), col("prom"), resp_lens)
        plan = self.plan(*args, tone_list=col("tone"), task=task)
        kv_lens, valid_lens = None, list(plan.lens)
        if task == "len" and len(set(plan.lens)) > 1:
            # base_v2.py:1298-1300 reads ``logit[..., -1, :]`` of the right-PADDED batch: for an utterance shorter than the
            # longest one that is a padding row -- a zero embedding at position 1 (the tail of the per-segment ids,
            # base_v2.py:865-867; under unified ids HF's arange gives it T_max - 1) that attends to the utterance's rows and
            # is no key itself.  One such row is appended and read, with the valid rows as the only keys.
            T = plan.max_T
            kv_lens = self._i32(np.array(plan.lens, dtype=np.int32))
            plan = self.plan(*args, tone_list=col("tone"), task=task,
                             pad_pos=[None if n == T else (T - 1 if self.unified_position_ids else 1) for n in plan.lens])
        codes = [_np(r) for r in resps if r is not None and r.shape[0] > 0]
        x, dev = self.build_x(plan, np.concatenate(codes) if codes else None)
        causal = self.is_causal(timestep, task)
        if task == "len":
            if not self.len_use_logits:
                # the reference reshapes every ``len`` utterance's head output to (-1, 5, 10) first (base_v2.py:1279-1283): with
                # the single-float head that is a (T, 1) tensor, and the call dies unless T is a multiple of 50
                for n in valid_lens:
                    if n % 50:
                        raise RuntimeError(f"shape '[-1, 5, 10]' is invalid for input of size {n}")
            rows = self._i32(plan.cu[1:] - 1)
            hidden = self.forward_hidden(plan, x, dev, rows, causal, kv_lens=kv_lens)
            lg = self.aux_decode("len_decoder", hidden)                                # last row of every utterance (:1298-1300)
            if not self.len_use_logits:
                return Logits([float(v) for v in lg[:, 0].cpu()], None, inputs, None, None, None)
            digits = lg.reshape(plan.B, 5, 10).argmax(dim=-1).cpu().numpy()             # :1302-1306
            secs = [int("".join(str(int(t)) for t in row)) / self.audio_frames_per_second for row in digits]
            return Logits(secs, None, inputs, None, None, None)
        rows = self._i32(np.arange(plan.M, dtype=np.int32))
        hidden = self.forward_hidden(plan, x, dev, rows, causal)
        lg = self.decode(hidden)
        V = self.audio_vocab
        out = [lg[:, int(plan.cu[i]):int(plan.cu[i + 1]), :V] for i in range(plan.B)]
        return Logits(out, None, inputs, None, None, None)
