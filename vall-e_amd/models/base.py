"""Host side of the HIP path: weight preparation, sequence planning, kernel-call wrappers.

Mirrors ``vall_e.models.base.Base`` (reference models/base.py:284-1430) for the inference
path: same constructor arguments, same state-dict names, same ``sample`` kwargs.  All
arithmetic happens in libvalle_hip.so; torch is used to allocate device buffers and to move
integer bookkeeping (ids, offsets) to the GPU.
"""
from __future__ import annotations

import ctypes as C
from collections import namedtuple

import threading

import numpy as np
import torch

from .. import _lib as L
from ..config import Model as ModelConfig, repo_default_model
from ..weights import classifier_layout, state_dict_spec

Logits = namedtuple("Logits", ["logits", "state", "inputs", "loss", "attentions", "hidden_states"])  # base.py:46
Sampled = namedtuple("Sampled", ["ids", "logits", "scores", "entropy"])                             # base.py:47

# mirostat_tau / mirostat_eta are accepted and change nothing, as in the reference: forward_ar builds the mirostat
# state list (ar_nar.py:702-704) but its Base.sample call (:804-808) never passes ``mirostat=``, so basic sampling
# runs (golden tiny_mirostat_tau).  Only a caller-made ``mirostat=[state, ...]`` kwarg reaches samplers.py:197-238.
UNSUPPORTED_SAMPLERS = ("beam_width", "layer_skip")


# data.py:670-688 (get_task_symmap): rows of tasks_emb
TASK_SYMMAP = {"tts": 0, "tts-c": 1, "ns": 2, "sr": 3, "tse": 4, "soe": 5, "mask": 6, "eoe": 7, "stt": 8,
               "len": 0, "nse": 6, "cse": 6, "phn": 0, "un-phn": 0}


class RawText:
    """Marks an id tensor as raw text -- the reference's ``text_list`` (ids of the text tokenizer, embedded by
    ``raw_text_emb`` in place of the phonemes: base.py:609-610, :841-842; inference.py:518-519 with phonemize=False).
    ``forward(text_list=...)`` wraps its entries so that one list travels through every call path."""

    def __init__(self, ids):
        self.ids = ids


def _round_up(a, b):
    return (a + b - 1) // b * b


class Plan:
    """Packed description of a batch of sequences (what Base.inputs + inputs_to_embeddings +
    inputs_to_position_ids produce, base.py:560-735, :774-922, :949-996), as integer tables."""

    __slots__ = ("idx", "pos", "cu", "lens", "out_len", "head", "B", "M", "max_T")

    def __init__(self, idx, pos, lens, out_len, head):
        self.idx, self.pos, self.lens, self.out_len, self.head = idx, pos, lens, out_len, head
        self.B = len(lens)
        self.cu = np.zeros(self.B + 1, dtype=np.int32)
        np.cumsum(lens, out=self.cu[1:])
        self.M = int(self.cu[-1])
        self.max_T = int(max(lens))


class Base:
    def __init__(self, n_phn_tokens=256, n_audio_tokens=1024, n_text_tokens=8575, d_model=512, d_ffn=4, n_heads=8,
                 n_layers=12, p_dropout=0.1, n_experts=1, l_padding=0, training=True, attention=None, config=None):
        if config is None:
            raise ValueError("config is required (the reference dereferences it too, models/base.py:340)")
        if not isinstance(config, ModelConfig):
            config = ModelConfig.from_dict({k: getattr(config, k) for k in
                                            ("name", "version", "size", "tasks", "langs", "tones", "arch_type", "attention",
                                             "dropout", "capabilities") if hasattr(config, k)} |
                                           {"experimental": getattr(config, "experimental", None)})
        exp = config.experimental
        if config.arch_type != "llama":
            raise NotImplementedError(f"arch_type {config.arch_type!r}: only the llama stack is built")
        if config.version != 5 or not exp.split_classifiers:
            raise NotImplementedError("built for the repo-default model family: version 5, split classifiers")
        if exp.causal_size != 1:
            raise NotImplementedError("causal_size != 1")
        if n_experts != 1:
            raise NotImplementedError("MoE")
        if d_model // n_heads != 64:
            raise NotImplementedError("head_dim must be 64")
        self.config = config
        self.training = False
        self.n_phn_tokens, self.n_audio_tokens, self.n_text_tokens = n_phn_tokens, n_audio_tokens, n_text_tokens
        self.d_model, self.n_heads, self.n_layers, self.d_ffn = d_model, n_heads, n_layers, d_ffn
        self.n_resp_levels = config.resp_levels
        self.n_max_levels = config.max_levels
        self.capabilities = list(config.capabilities) + ["lang"]          # base.py:444
        self.stop_token = n_audio_tokens
        self.mask_token = self.stop_token
        self.causal = True
        self.version = config.version
        self.causal_size = exp.causal_size
        self.unified_position_ids = exp.unified_position_ids
        # True / "prom" / "resp" (base.py:434, :438); without sums a frame is embedded by its last known level alone (:169-173)
        self.sum_prom_levels = exp.audio_embedding_sums is True or exp.audio_embedding_sums == "prom"
        self.sum_resp_levels = exp.audio_embedding_sums is True or exp.audio_embedding_sums == "resp"
        self.noncausal_masks = exp.noncausal_masks
        self.masking_ratio = exp.masking_ratio
        self.max_position_embeddings = exp.max_position_embeddings
        self.emb_names, self.emb_tokens, self.cls_names, self.cls_tokens = classifier_layout(config)
        self.device = torch.device("cpu")
        self.dtype = torch.float32
        self._off, self._table_order = self._table_offsets()
        self._sd = None
        self._handles = {}                         # False: base weights, True: LoRA merged (when one is loaded)
        self._lora = None                          # (tensors, rank, alpha, rvq_levels)
        self._bufs = {}
        self._ws = None
        self._call_state = threading.local()       # per-thread state of the call in flight (tone list, LoRA on/off)
        self._keep = []

    # ------------------------------------------------------------- nn.Module-ish surface
    def state_dict_spec(self):
        return state_dict_spec(self.config)

    def load_state_dict(self, sd, strict=True):
        spec = self.state_dict_spec()
        missing = [k for k in spec if k not in sd]
        if strict and missing:
            raise KeyError(f"missing keys: {missing[:5]}...")
        for k, shape in spec.items():
            if k in sd and tuple(sd[k].shape) != tuple(shape):
                raise ValueError(f"{k}: shape {tuple(sd[k].shape)} != {shape}")
        self._sd = {k: sd[k].detach().to("cpu", torch.float32) for k in spec if k in sd}
        self._release()
        return self

    def eval(self):
        return self

    def float(self):
        return self.to(dtype=torch.float32)

    def bfloat16(self):
        return self.to(dtype=torch.bfloat16)

    def to(self, device=None, dtype=None):
        if isinstance(device, torch.dtype):
            device, dtype = None, device
        if device is not None:
            self.device = torch.device(device)
        if dtype is not None:
            L.dtype_code(dtype)
            self.dtype = dtype
        self._release()
        return self

    def _release(self):
        for h in self._handles.values():
            L.load().valle_model_destroy(h)
        self._handles, self._bufs, self._ws, self._keep = {}, {}, None, []

    @property
    def _handle(self):
        """The model descriptor of the call in flight: LoRA-merged weights when a LoRA is loaded and switched on for this
        pass (AR_NAR sets it per RVQ level, reference models/lora.py:183-191 through ar_nar.py:268-269, :539-540, :619-620)."""
        if not self._handles:
            return None
        on = bool(getattr(self._call_state, "lora_on", False)) and True in self._handles
        return self._handles[on]

    # ------------------------------------------------------------------------------ LoRA
    def load_lora(self, lora_state_dict, rank, alpha, rvq_levels=()):
        """LoRA weights in the reference's names (``model.layers.N.<linear>.lora_A`` (rank, in) / ``.lora_B`` (out, rank),
        models/lora.py:19-75; targets per config.py:519-530: self_attn and mlp linears except k_proj).  The adapted
        linears get a second set of device weights W + (alpha / rank) B A; ``rvq_levels`` (config.py:547, :556-559)
        lists the RVQ levels whose passes use it (empty: all)."""
        t = {k: v.detach().to("cpu", torch.float32) for k, v in lora_state_dict.items() if "lora_" in k}
        if not t:
            raise ValueError("no lora_A / lora_B tensors in the state dict")
        for k in t:
            if not (k.startswith("model.layers.") and ("self_attn." in k or "mlp." in k)):
                raise NotImplementedError(f"{k}: LoRA on modules other than the stack's linears is not built")
        self._lora = (t, int(rank), float(alpha), tuple(int(x) for x in rvq_levels))
        self._release()
        return self

    def lora_active_level(self, level, use_lora=None):
        """config.py:556-559 ``LoRA.active_level`` with the call's ``use_lora`` override (ar_nar.py:268-269)."""
        if self._lora is None:
            return False
        if use_lora is not None:
            return bool(use_lora)
        levels = self._lora[3]
        return True if not levels else int(level) in levels

    def _merged_lora_sd(self):
        t, rank, alpha, _ = self._lora
        sd = dict(self._sd)
        for ka in [k for k in t if k.endswith(".lora_A")]:
            base = ka[: -len(".lora_A")]
            A, Bm = t[ka], t[base + ".lora_B"]
            W = sd[base + ".weight"]
            if A.shape != (rank, W.shape[1]) or Bm.shape != (W.shape[0], rank):
                raise ValueError(f"{base}: lora_A {tuple(A.shape)} / lora_B {tuple(Bm.shape)} do not fit weight {tuple(W.shape)} at rank {rank}")
            sd[base + ".weight"] = W + (Bm @ A) * (alpha / rank)         # models/lora.py:61-65, merged form of :69-72
        return sd

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    # ------------------------------------------------------------------ weight preparation
    def _table_offsets(self):
        """Row offsets of every embedding table inside the one concatenated table."""
        spec = self.state_dict_spec()
        order = [("text", "text_emb.weight"), ("raw_text", "raw_text_emb.weight"), ("rvq", "rvq_l_emb.weight"), ("lang", "langs_emb.weight"),
                 ("task", "tasks_emb.weight"), ("len", "len_emb.weight"), ("tone", "tones_emb.weight"), ("sep", "sep")]
        order += [(f"prom{l}", f"proms_emb.embeddings.{l}.weight") for l in range(self.n_resp_levels)]
        order += [(f"resp{l}", f"resps_emb.embeddings.{l}.weight") for l in range(len(self.emb_tokens))]
        order = [(n, k) for n, k in order if k in spec]
        off, row = {}, 0
        for name, key in order:
            off[name] = row
            row += 1 if len(spec[key]) == 1 else spec[key][0]
        return off, order

    def _build_layers(self, sd, tag=""):
        """Device copies of the layer weights of state dict ``sd`` in the kernels' layouts (include/valle_hip.h) ->
        valle_layer_weights array; buffers are kept in self._bufs under ``tag``-prefixed names."""
        dev, dt = self.device, self.dtype
        d, H, ff = self.d_model, self.n_heads, self.d_model * self.d_ffn
        Dh = d // H
        store = self._bufs
        class _B(dict):                      # writes go to the shared buffer table under the tag
            def __setitem__(s_, k, v): store[tag + k] = v
            def __getitem__(s_, k): return store[tag + k]
        b = _B()
        # per-head row permutation that makes RoPE pairs adjacent: (0,32,1,33,...)
        perm_head = torch.stack((torch.arange(Dh // 2), torch.arange(Dh // 2) + Dh // 2), dim=1).reshape(-1)
        perm = (torch.arange(H)[:, None] * Dh + perm_head[None, :]).reshape(-1)
        layers = (L.LayerWeights * self.n_layers)()
        for l in range(self.n_layers):
            p = f"model.layers.{l}."
            wq, wk, wv = (sd[p + f"self_attn.{n}_proj.weight"] for n in "qkv")
            host = {"qkv": torch.cat((wq[perm], wk[perm], wv)).to(dt)}          # host copies in the storage dtype
            b[f"qkv{l}"] = host["qkv"].to(dev).contiguous()
            b[f"o{l}"] = sd[p + "self_attn.o_proj.weight"].to(dt).contiguous().to(dev)
            gu = torch.stack((sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"]), dim=1).reshape(2 * ff, d)
            host["gu"] = gu.to(dt)
            b[f"gu{l}"] = host["gu"].to(dev).contiguous()
            b[f"down{l}"] = sd[p + "mlp.down_proj.weight"].to(dt).contiguous().to(dev)
            b[f"ln1_{l}"] = sd[p + "input_layernorm.weight"].to(dt).contiguous().to(dev)
            b[f"ln2_{l}"] = sd[p + "post_attention_layernorm.weight"].to(dt).contiguous().to(dev)
            packed = [None] * 6
            if dt == torch.bfloat16 and d % 64 == 0:
                for j, n in enumerate(("o", "down")):
                    b[f"p_{n}{l}"] = pack_stream_order(narrow_tile_rows(b[f"{n}{l}"], 4))
                    packed[j] = b[f"p_{n}{l}"].data_ptr()
                # norm weight folded in (product in fp32 of the bf16-rounded factors, rounded to bf16 once)
                for j, (n, ln) in enumerate((("qkv", "ln1_"), ("gu", "ln2_"))):
                    # (on the host: weight preparation runs once and leaves no torch arithmetic kernel on the device)
                    ln_host = sd[p + ("input_layernorm.weight" if ln == "ln1_" else "post_attention_layernorm.weight")].to(dt)
                    folded = (host[n].float() * ln_host.float()[None, :]).to(dt)
                    b[f"pn_{n}{l}"] = pack_stream_order(folded).to(dev)
                    packed[2 + j] = b[f"pn_{n}{l}"].data_ptr()
                    b[f"f_{n}{l}"] = folded.to(dev).contiguous()              # row-major, for the dense passes (valle_gemm_normed)
                    packed[4 + j] = b[f"f_{n}{l}"].data_ptr()
            layers[l] = L.LayerWeights(b[f"ln1_{l}"].data_ptr(), b[f"qkv{l}"].data_ptr(), b[f"o{l}"].data_ptr(),
                                       b[f"ln2_{l}"].data_ptr(), b[f"gu{l}"].data_ptr(), b[f"down{l}"].data_ptr(), *packed)
        return layers

    def _prepare(self):
        if self._handles:
            return
        if self._sd is None:
            raise RuntimeError("load_state_dict first")
        if self.device.type != "cuda":
            raise L.ValleHipError("the HIP path needs a GPU device (model.to('cuda')); there is no CPU fallback")
        lib = L.load()
        sd, dev, dt = self._sd, self.device, self.dtype
        d, H, ff = self.d_model, self.n_heads, self.d_model * self.d_ffn
        off = self._off
        # concatenated embedding table (row offsets: _table_offsets)
        tabs = [sd[key].reshape(-1, d) for _, key in self._table_order]
        b = self._bufs
        b["emb"] = torch.cat(tabs).to(dt).contiguous().to(dev)
        # RoPE table, computed exactly as the reference does (arch/llama.py:118-127): fp32
        # outer product of positions and inv_freq, then cos / sin
        Dh = d // H
        inv_freq = 1.0 / (10000.0 ** (torch.arange(0, Dh, 2, dtype=torch.int64).float() / Dh))
        pos = torch.arange(self.max_position_embeddings, dtype=torch.float32)
        freqs = (inv_freq[None, :, None] @ pos[None, None, :]).transpose(1, 2)[0]      # (P, Dh/2)
        b["rope"] = torch.stack((freqs.cos(), freqs.sin()), dim=-1).to(dev).contiguous()  # (P, Dh/2, 2) fp32
        layers = self._build_layers(sd)
        b["norm"] = sd["model.norm.weight"].to(dt).contiguous().to(dev)
        n_cls = len(self.cls_tokens)
        hw, hb, hn, hpn = (C.c_void_p * n_cls)(), (C.c_void_p * n_cls)(), (C.c_int32 * n_cls)(), (C.c_void_p * n_cls)()
        for k in range(n_cls):
            b[f"hw{k}"] = sd[f"classifiers.proj.{k}.weight"].to(dt).contiguous().to(dev)
            hw[k] = b[f"hw{k}"].data_ptr()
            bias = sd.get(f"classifiers.proj.{k}.bias")
            if bias is not None:
                b[f"hb{k}"] = bias.to(dt).to(torch.float32).contiguous().to(dev)     # rounded to the storage dtype, held in fp32 (on the host: no device kernel)
                hb[k] = b[f"hb{k}"].data_ptr()
            else:
                hb[k] = None
            hn[k] = self.cls_tokens[k]
            hpn[k] = None
            if dt == torch.bfloat16 and k == 0:      # the AR head is the one the decode loop streams
                hw_host = sd[f"classifiers.proj.{k}.weight"].to(dt)
                b[f"hpn{k}"] = pack_stream_order((hw_host.float() * sd["model.norm.weight"].to(dt).float()[None, :]).to(dt)).to(dev)
                hpn[k] = b[f"hpn{k}"].data_ptr()
        cfg = L.Config(d, H, self.n_layers, ff, L.dtype_code(dt), 1e-6, self.max_position_embeddings)
        handle = C.c_void_p()
        L.check(lib.valle_model_create(C.byref(cfg), L.ptr(b["emb"]), L.ptr(b["rope"]), layers, L.ptr(b["norm"]), n_cls,
                                       hw, hb, hn, hpn, C.byref(handle)))
        self._handles[False] = handle
        self._keep = [layers, hw, hb, hn, hpn]
        if self._lora is not None:
            # second descriptor: same tables, heads and norms, the adapted linears merged (models/lora.py:61-65)
            layers2 = self._build_layers(self._merged_lora_sd(), tag="lora:")
            handle2 = C.c_void_p()
            L.check(lib.valle_model_create(C.byref(cfg), L.ptr(b["emb"]), L.ptr(b["rope"]), layers2, L.ptr(b["norm"]), n_cls,
                                           hw, hb, hn, hpn, C.byref(handle2)))
            self._handles[True] = handle2
            self._keep.append(layers2)

    # -------------------------------------------------------------------- sequence planning
    def inputs(self, phns_list=None, text_list=None, proms_list=None, resps_list=None, lang_list=None, tone_list=None,
               len_list=None, task_list=None, time_list=None, quant_levels=None):
        """The reference's per-utterance list of (name, value) tuples for the TTS layout (base.py:560-649):
        task, phn, lang, quant_level, prom, resp, classifier_level -- and for the ``len`` task (base.py:652-683):
        task, phn, lang, quant_level, prom, len, classifier_level -- accepted by ``forward_super``."""
        if phns_list is None and text_list is not None:
            phns_list = [None if t is None else RawText(t) for t in text_list]
        first = next(l for l in (phns_list, proms_list, resps_list) if l)
        B = len(first)
        out = [[] for _ in range(B)]
        for i in range(B):
            q = int(quant_levels[i]) if quant_levels is not None else 0
            task = task_list[i] if task_list is not None else "tts"
            if task not in ("tts", "len"):
                raise NotImplementedError(f"task {task!r}: only 'tts' and 'len' are built")
            if task == "tts" and len_list is not None:
                raise NotImplementedError("len inputs belong to the 'len' task")
            timestep = time_list[i] if time_list is not None else None
            out[i].append(("task", task))
            if phns_list is not None and phns_list[i] is not None:
                out[i].append(("text", phns_list[i].ids) if isinstance(phns_list[i], RawText) else ("phn", phns_list[i]))
            if "lang" in self.capabilities and lang_list is not None and lang_list[i] is not None:
                out[i].append(("lang", lang_list[i]))
            out[i].append(("quant_level", torch.tensor([q], dtype=torch.int16)))
            level = "AR:0:0" if q == 0 else f"NAR:{q-1}:{q}"
            if proms_list is not None and proms_list[i] is not None:
                out[i].append(("prom", proms_list[i]))
            if "tone" in self.capabilities and tone_list is not None and tone_list[i] is not None:     # base.py:623-624, :672-673
                out[i].append(("tone", tone_list[i]))
            if task == "len":
                if "len" not in self.cls_names:                              # base.py:654-655
                    raise Exception("Requesting task `len` but corresponding embedding is not defined.")
                if len_list is not None and len_list[i] is not None:
                    out[i].append(("len", len_list[i]))
                elif resps_list is not None and resps_list[i] is not None:   # base.py:679-681: digits of the frame count
                    digits = [0] + [int(c) for c in str(int(resps_list[i].shape[0]))] + [10]
                    out[i].append(("len", torch.tensor(digits, dtype=torch.int16)))
                out[i].append(("classifier_level", "len"))
                continue
            if timestep is not None:
                level = "NAR:0:0"                                           # base.py:628
            if resps_list is not None and resps_list[i] is not None:
                out[i].append(("resp", resps_list[i]))
            out[i].append(("classifier_level", level))
        return out

    def plan_from_inputs(self, inputs):
        """``inputs()`` tuples -> Plan.  One classifier level per batch (one head per dense pass)."""
        col = {k: [dict((n, v) for n, v in item if n != "task").get(k) for item in inputs] for k in ("phn", "text", "lang", "prom", "tone", "resp", "len")}
        tones = col["tone"] if any(t is not None for t in col["tone"]) else None
        col["phn"] = [p if p is not None else (None if t is None else RawText(t)) for p, t in zip(col["phn"], col["text"])]
        levels = {dict(item)["classifier_level"] for item in inputs}
        qs = {int(dict(item)["quant_level"][0]) for item in inputs}
        if len(levels) != 1 or len(qs) != 1:
            raise NotImplementedError("one classifier level per batch")
        level, q = levels.pop(), qs.pop()
        if level == "len":
            return self.plan(col["phn"] if any(p is not None for p in col["phn"]) else None,
                             col["lang"] if any(p is not None for p in col["lang"]) else None,
                             col["prom"] if any(p is not None for p in col["prom"]) else None,
                             [None] * len(inputs), q, task="len", len_list=col["len"], tone_list=tones)
        resps = col["resp"] if any(r is not None for r in col["resp"]) else [None] * len(inputs)
        return self.plan(col["phn"] if any(p is not None for p in col["phn"]) else None,
                         col["lang"] if any(p is not None for p in col["lang"]) else None,
                         col["prom"] if any(p is not None for p in col["prom"]) else None,
                         resps, q, timestep=(level == "NAR:0:0"), tone_list=tones)

    def plan(self, phns_list, lang_list, proms_list, resps_list, quant_level, *, timestep=False, task="tts",
             len_list=None, tone_list=None) -> Plan:
        """TTS layout phn, lang, quant_level, prom, resp with a ``sep`` row between segments
        (base.py:603-649, :101-110); classifier level per base.py:618/:628; position ids restart per
        segment, non-output segments count their trailing sep (base.py:984).  task "len" (base.py:652-683):
        the length digits (``len_list``) stand where resp does and the head is ``len``."""
        off = self._off
        out_kind = "len" if task == "len" else "resp"                                # task_outputs, base.py:53-61
        if task == "len" and "len" not in self.cls_names:
            raise Exception("Requesting task `len` but corresponding embedding is not defined.")
        B = len(resps_list)
        q = quant_level
        if tone_list is None:                         # the call's tone list (AR_NAR.forward keeps it for every pass it plans)
            tone_list = getattr(self._call_state, "tones", None)
        idx_all, pos_all, lens, out_len = [], [], [], []
        sep_row = np.full((1, 8), -1, dtype=np.int32)
        sep_row[0, 0] = off["sep"]

        def rows1(base, ids):
            r = np.full((len(ids), 8), -1, dtype=np.int32)
            r[:, 0] = base + ids
            return r

        for i in range(B):
            segs = []
            if phns_list is not None and isinstance(phns_list[i], RawText):             # base.py:609-610, :841-842
                if "raw_text" not in off:
                    raise Exception("text_list given but the model has no raw_text_emb")
                segs.append(("text", rows1(off["raw_text"], _np(phns_list[i].ids))))
            elif phns_list is not None and phns_list[i] is not None:
                segs.append(("phn", rows1(off["text"], _np(phns_list[i]))))
            if "lang" in self.capabilities and lang_list is not None and lang_list[i] is not None:
                segs.append(("lang", rows1(off["lang"], _np(lang_list[i]))))
            segs.append(("quant_level", rows1(off["rvq"], np.array([q]))))
            if proms_list is not None and proms_list[i] is not None:
                # a prompt is a code tensor or a list of code tensors and task-token strings ([task, codes] for
                # ns / sr, inference.py:489-493; base.py:780-794, :850-856).  The reference's position ids count only
                # the tensors of a list (base.py:964-974), so n_tok < rows when task tokens are present.
                plist = proms_list[i] if isinstance(proms_list[i], (list, tuple)) else [proms_list[i]]
                parts_p, n_tok = [], 0
                for prom in plist:
                    if prom is None:
                        continue
                    if isinstance(prom, str):
                        parts_p.append(rows1(off["task"], np.array([TASK_SYMMAP[prom]])))
                        continue
                    prom = _np(prom)
                    n_tok += prom.shape[0]
                    if prom.ndim == 1:
                        parts_p.append(rows1(off["prom0"], prom))
                    else:
                        n = 1 if q == 0 else q                                        # base.py:790-794
                        r = np.full((prom.shape[0], 8), -1, dtype=np.int32)
                        for j, l in enumerate(range(n) if self.sum_prom_levels else (n - 1,)):   # rows fill from column 0, -1 ends them
                            r[:, j] = off[f"prom{l}"] + prom[:, l]
                        parts_p.append(r)
                segs.append(("prom", np.concatenate(parts_p) if parts_p else np.zeros((0, 8), np.int32), n_tok))
            if "tone" in self.capabilities and tone_list is not None and tone_list[i] is not None:   # base.py:623-624, :672-673
                segs.append(("tone", rows1(off["tone"], _np(tone_list[i]))))
            if task == "len" and len_list is not None and len_list[i] is not None:
                segs.append(("len", rows1(off["len"], _np(len_list[i]))))
            resp = resps_list[i] if task != "len" else None
            if resp is not None:
                resp = _np(resp)
                if timestep:                                                          # NAR-len: base.py:869-874
                    col = resp if resp.ndim == 1 else resp[:, q]
                    segs.append(("resp", rows1(off[f"resp{self.emb_names.index('NAR:0:0')}"], col)))
                elif q == 0:                                                          # AR: base.py:893-899
                    col = resp if resp.ndim == 1 else resp[:, 0]
                    segs.append(("resp", rows1(off["resp0"], col)))
                else:                                                                 # NAR q: sum of levels < q, tables 1..q
                    if resp.ndim == 1:
                        resp = resp[:, None]
                    r = np.full((resp.shape[0], 8), -1, dtype=np.int32)
                    for j, l in enumerate(range(q) if self.sum_resp_levels else (q - 1,)):
                        r[:, j] = off[f"resp{l + 1}"] + resp[:, l]
                    segs.append(("resp", r))
            parts, pos = [], []
            for j, seg in enumerate(segs):
                kind, r = seg[0], seg[1]
                n_tok = seg[2] if len(seg) > 2 else r.shape[0]
                if j:
                    parts.append(sep_row)
                parts.append(r)
                pos.append(np.arange(n_tok + (0 if kind == out_kind else 1), dtype=np.int32))
            idx = np.concatenate(parts)
            pos = np.concatenate(pos)[: idx.shape[0]]
            if pos.shape[0] < idx.shape[0]:                                           # base.py:988-990: tail filled with 1
                pos = np.concatenate([pos, np.ones(idx.shape[0] - pos.shape[0], np.int32)])
            if self.unified_position_ids:
                pos = np.arange(idx.shape[0], dtype=np.int32)
            if idx.shape[0] > self.max_position_embeddings:                           # the RoPE table has that many rows
                raise ValueError(f"sequence of {idx.shape[0]} rows exceeds max_position_embeddings ({self.max_position_embeddings})")
            idx_all.append(idx), pos_all.append(pos), lens.append(idx.shape[0])
            out_len.append(segs[-1][1].shape[0] if segs[-1][0] == out_kind else 0)
        head = "NAR:0:0" if timestep else ("AR:0:0" if q == 0 else f"NAR:{q-1}:{q}")
        if task == "len":
            head = "len"
        return Plan(np.concatenate(idx_all), np.concatenate(pos_all), lens, out_len, self.cls_names.index(head))

    # ------------------------------------------------------------------------ dense forward
    def _workspace(self, nbytes):
        # one workspace per stream: work enqueued on two streams must not share scratch memory
        if self._ws is None:
            self._ws = {}
        key = torch.cuda.current_stream(self.device).cuda_stream
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            self._ws.pop(key, None)
            ws = self._ws[key] = torch.empty(_round_up(nbytes, 1 << 20), dtype=torch.uint8, device=self.device)
        return ws

    def _filled(self, shape, dtype, value=0):
        """A device tensor of a 4-byte dtype with every element = value, initialised by valle_fill32 on the current stream
        (torch.zeros / full / ones would launch a fill kernel of torch's own)."""
        assert dtype in (torch.int32, torch.float32)
        t = torch.empty(shape, dtype=dtype, device=self.device)
        if t.numel():
            word = int(np.array(value, dtype=np.float32 if dtype == torch.float32 else np.int32).view(np.uint32))
            L.check(L.load().valle_fill32(L.ptr(t), word, t.numel(), L.stream_ptr(self.device)))
        return t

    def _i32(self, arr):
        return self._h2d(np.ascontiguousarray(arr, dtype=np.int32))

    def _h2d(self, arr):
        """Asynchronous host->device copy of a numpy array through PINNED staging memory.  An asynchronous copy straight
        from a pageable numpy buffer may still be reading it after this returns, while the buffer is already freed and
        reused by the host (seen as wrong codes of one utterance once a second host thread was allocating concurrently);
        torch's pinned allocator keeps a staging block alive until the copy that uses it has completed."""
        t = torch.from_numpy(arr)
        if self.device.type != "cuda":
            return t.to(self.device)
        return t.pin_memory().to(self.device, non_blocking=True)

    def _upload_i32(self, *arrays):
        """Several integer host arrays -> device int32 views of ONE uploaded buffer (one copy, one sync)."""
        flat = [np.ascontiguousarray(a, dtype=np.int32).reshape(-1) for a in arrays]
        pad = [(-len(f)) % 4 for f in flat]                                   # keep every view 16-byte aligned
        buf = np.concatenate([np.concatenate((f, np.zeros(p, np.int32))) for f, p in zip(flat, pad)]) if flat else np.zeros(0, np.int32)
        dev = self._h2d(buf)
        out, o = [], 0
        for a, f, p in zip(arrays, flat, pad):
            out.append(dev[o:o + len(f)].view(np.asarray(a).shape))
            o += len(f) + p
        return out

    def upload_plan(self, plan: Plan, out_rows):
        """Device copies of a plan's integer tables: (idx [M][8], pos [M], cu [B+1], rows [n_out])."""
        return tuple(self._upload_i32(plan.idx, plan.pos, plan.cu, out_rows))

    def forward_dense(self, plan: Plan, out_rows, *, cache=None, want_hidden=False, dev=None, score_rows=None):
        """Runs the stack on a packed plan; returns fp32 logits (n_out, ld) for the packed row ids
        in ``out_rows`` (head = plan.head).  cache = (k, v, cache_T, layer_stride, slot, pos0);
        dev = upload_plan(...) made earlier (possibly patched on the device since)."""
        self._prepare()
        lib = L.load()
        n_out = int(len(out_rows))
        vocab = self.cls_tokens[plan.head]
        ld = _round_up(vocab, 8)
        idx, pos, cu, rows = dev if dev is not None else self.upload_plan(plan, out_rows)
        logits = torch.empty((n_out, ld), dtype=torch.float32, device=self.device)
        hidden = torch.empty((n_out, self.d_model), dtype=self.dtype, device=self.device) if want_hidden else None
        need = lib.valle_dense_workspace_bytes(self._handle, plan.M, n_out)
        ws = self._workspace(need)
        # base.py:1230-1233: with noncausal_masks the NAR levels (NAR:q-1:q, NAR:0:0) attend in both directions
        level = self.cls_names[plan.head]
        causal = 1 if (not self.noncausal_masks or level in ("phn", "len") or level.startswith("AR:")) else 0
        if not causal and cache is not None:
            raise ValueError("a KV cache only makes sense for a causal pass")
        if cache is None:
            kc = vc = slot = pos0 = None
            cache_T, stride = 0, 0
        else:
            kc, vc, cache_T, stride, slot, pos0 = cache
        if score_rows is not None:
            # output_attentions: also the last layer's raw attention products of the packed rows in score_rows, over every
            # key of their utterance -> (n_score, H, max_T) fp32 (valle_forward_dense_scores)
            srows = self._i32(score_rows)
            scores = self._filled((len(score_rows), self.n_heads, plan.max_T), torch.float32)
            L.check(lib.valle_forward_dense_scores(self._handle, L.ptr(idx), L.ptr(pos), L.ptr(cu), plan.B, plan.M, plan.max_T, causal,
                                                   L.ptr(kc), L.ptr(vc), cache_T, stride, L.ptr(slot), L.ptr(pos0),
                                                   L.ptr(rows), n_out, plan.head, L.ptr(logits), ld, L.ptr(hidden),
                                                   L.ptr(ws), ws.numel(), L.stream_ptr(self.device),
                                                   L.ptr(srows), len(score_rows), L.ptr(scores), plan.max_T))
            return (logits, hidden, scores) if want_hidden else (logits, scores)
        L.check(lib.valle_forward_dense(self._handle, L.ptr(idx), L.ptr(pos), L.ptr(cu), plan.B, plan.M, plan.max_T, causal,
                                        L.ptr(kc), L.ptr(vc), cache_T, stride, L.ptr(slot), L.ptr(pos0),
                                        L.ptr(rows), n_out, plan.head, L.ptr(logits), ld, L.ptr(hidden),
                                        L.ptr(ws), ws.numel(), L.stream_ptr(self.device)))
        return (logits, hidden) if want_hidden else logits

    @staticmethod
    def tail_rows(plan: Plan, counts):
        """Packed row ids of the last counts[i] rows of every sequence + per-utterance offsets."""
        rows, off = [], [0]
        for i, n in enumerate(counts):
            end = int(plan.cu[i + 1])
            rows.append(np.arange(end - n, end, dtype=np.int32))
            off.append(off[-1] + n)
        return np.concatenate(rows) if rows else np.zeros(0, np.int32), np.array(off, dtype=np.int32)

    # ----------------------------------------------------------------------------- sampling
    def sample_params(self, *, temperature=1.0, min_temperature=-1.0, top_k=-100, top_p=1.0, min_p=0.0,
                      repetition_penalty=1.0, repetition_penalty_decay=0.0, length_penalty=0.0, top_no=1.0,
                      rep_pen_passes=1, seed=0, offset=0, dry_multiplier=0.0, dry_base=1.75, dry_allowed_length=2,
                      filter_passes=1, score_unscaled=0, **other) -> L.SampleParams:
        """Base.sample's kwargs (base.py:1288-1309) -> the C struct.  Samplers outside the built
        chain raise instead of being silently ignored."""
        for k in UNSUPPORTED_SAMPLERS:
            if other.get(k) and not (k == "beam_width" and other[k] == 1):
                raise NotImplementedError(f"sampling option {k!r} is not built on the HIP path"
                                          + (" outside forward_ar" if k == "beam_width" else ""))
        if other.get("mirostat") is not None:
            raise NotImplementedError("an explicit mirostat state list goes through Base.sample (not the generation loops, "
                                      "which never pass one in the reference either: ar_nar.py:804-808)")
        if other.get("attentions") is not None or other.get("entropix_sampling"):
            raise NotImplementedError("entropix sampling runs through forward_ar(entropix_sampling=True) / Base.sample(attentions=...), "
                                      "not through the processor chain (the reference returns before it, base.py:1327-1342)")
        if min_temperature < 0:
            min_temperature = temperature
        if abs(float(temperature) - float(min_temperature)) < 0.001:          # base.py:1389: dynamic temperature off
            min_temperature = temperature
        return L.SampleParams(float(temperature), int(top_k), float(top_p), float(min_p), float(top_no),
                              float(repetition_penalty), float(repetition_penalty_decay), float(length_penalty),
                              int(self.stop_token), int(rep_pen_passes), int(seed), int(offset),
                              float(temperature), float(min_temperature), float(dry_multiplier), float(dry_base),
                              int(dry_allowed_length), int(filter_passes), int(score_unscaled))

    # ------------------------------------------------------------------------------ entropix
    # samplers.py:364-414 (EntropixSamplerConfig)
    ENTROPIX = dict(
        min_p=0.01, low_ent_thresh=0.1, low_vent_thresh=0.1, med_ent_thresh=3.0, high_ent_thresh=5.0, high_vent_thresh=5.0,
        helv_attn_ent_offset=1.3, helv_attn_ent_coef=0.2, lehv_interaction_strength_offset=1.2,
        lehv_interaction_strength_coef=0.3, hehv_attn_ent_coef=0.2, hehv_attn_vent_offset=2.0, hehv_attn_vent_coef=0.5,
        n_adaptive_samples=5, ada_temp_logits=0.3, ada_temp_attn=0.2, ada_temp_agree=0.2, ada_top_p=0.1, ada_top_k_int=0.3,
        ada_top_k_agree=0.2, ada_min_p=0.5, temperature_max=1.25, temperature_min=0.5, top_k_min=1, top_k_max=1024,
        top_p_min=0.1, top_p_max=1.0)
    ENTROPIX_KEYS = ("logits_entropy", "logits_varentropy", "attn_entropy", "attn_varentropy", "agreement", "interaction_strength")

    @classmethod
    def entropix_branch(cls, m, temperature, top_k, top_p, vocab):
        """The branch arithmetic of sample_entropix (samplers.py:489-523) and the clamps of _sample_entropix (:428-436) on one
        utterance's metrics (Python floats, as in the reference) -> (action, temperature, top_k, top_p, draws).  Branch 0
        raises the IndexError the reference dies of there (``logits[-1, :].argmax(dim=1)`` on a 1-D row, :492)."""
        C_ = cls.ENTROPIX
        ent, vent = m["logits_entropy"], m["logits_varentropy"]
        a_ent, a_vent, agree, inter = m["attn_entropy"], m["attn_varentropy"], m["agreement"], m["interaction_strength"]
        draws = 1
        if ent < C_["low_ent_thresh"] and vent < C_["low_vent_thresh"]:
            raise IndexError("Dimension out of range (expected to be in range of [-1, 0], but got 1)")
        elif ent > C_["high_ent_thresh"] and vent < C_["low_vent_thresh"]:
            action = 1
            temperature *= C_["helv_attn_ent_offset"] + C_["helv_attn_ent_coef"] * a_ent
        elif ent < C_["high_ent_thresh"] and vent > C_["high_vent_thresh"]:
            action = 2
            temperature *= C_["lehv_interaction_strength_offset"] + C_["lehv_interaction_strength_coef"] * inter
            top_k = max(5, int(top_k * (1 + 0.5 * (1 - agree))))
        elif ent > C_["med_ent_thresh"] and vent > C_["high_vent_thresh"]:
            action = 3
            temperature *= C_["hehv_attn_vent_offset"] + C_["hehv_attn_vent_coef"] * a_vent
            top_p = max(0.5, top_p - C_["hehv_attn_ent_coef"] * a_ent)
        else:
            action = 4
            lu, au = ent + vent, a_ent + a_vent
            temperature *= 1 + C_["ada_temp_logits"] * lu + C_["ada_temp_attn"] * au - C_["ada_temp_agree"] * agree
            top_p = top_p * (1 + C_["ada_top_p"] * a_vent)
            top_k = round(float(top_k * (1 + C_["ada_top_k_int"] * inter - C_["ada_top_k_agree"] * agree)))
            draws = C_["n_adaptive_samples"]
        if top_k == 0:
            top_k = vocab
        clamp = lambda n, lo, hi: max(lo, min(n, hi))
        return (action, clamp(float(temperature), C_["temperature_min"], C_["temperature_max"]),
                clamp(int(top_k), C_["top_k_min"], C_["top_k_max"]), clamp(float(top_p), C_["top_p_min"], C_["top_p_max"]), draws)

    def entropix_sample(self, logits, logit_row, vocab, scores, n_keys, colsum, *, temperature=1.0, top_k=-100, top_p=1.0,
                        seed=0, offset=0, noise=None, want_forced=False):
        """sample_entropix (samplers.py:464-566) for B utterances on the device: metrics of logits row logit_row[b] and
        scores[b] (H, ld) over n_keys[b] keys (valle_entropix_metrics), the branch arithmetic on the host (one small copy --
        the reference does all of it on the CPU), then the draws (valle_entropix_pick).  colsum (B, ld) = sum over every
        row of the sequence of log_softmax (what the adaptive branch ranks its draws by).  noise (B, 5, n): the
        Exponential(1) variates per draw and ascending position (tests); None: the Philox stream of (seed, offset).
        -> (ids int32 (B,), metrics dicts, forced logits or None)."""
        lib = L.load()
        B = len(n_keys)
        stream = L.stream_ptr(self.device)
        met = torch.empty((B, 8), dtype=torch.float32, device=self.device)
        nk = self._i32(n_keys)
        rows = None if logit_row is None else self._i32(logit_row)
        L.check(lib.valle_entropix_metrics(L.ptr(logits), logits.shape[1], vocab, L.ptr(rows), L.ptr(scores), scores.shape[2],
                                           scores.shape[1], L.ptr(nk), B, L.ptr(met), stream))
        host = met.cpu().numpy()
        metrics, par = [], []
        for b in range(B):
            m = {k: float(host[b, i]) for i, k in enumerate(self.ENTROPIX_KEYS)}
            action, t, k, p, draws = self.entropix_branch(m, temperature, top_k, top_p, vocab)
            m["action"] = action
            metrics.append(m)
            par.append((t, k, p, draws))
        f32 = lambda v: self._h2d(np.ascontiguousarray(v, dtype=np.float32))
        t_d, p_d = f32([x[0] for x in par]), f32([x[2] for x in par])
        k_d, n_d = self._i32([x[1] for x in par]), self._i32([x[3] for x in par])
        ids = torch.empty(B, dtype=torch.int32, device=self.device)
        draws = torch.empty((B, 5), dtype=torch.int32, device=self.device)
        forced = torch.empty((B, logits.shape[1]), dtype=torch.float32, device=self.device) if want_forced else None
        if noise is not None:
            noise = noise.to(self.device, torch.float32).contiguous()
        L.check(lib.valle_entropix_pick(L.ptr(logits), logits.shape[1], vocab, L.ptr(rows), B, L.ptr(t_d), L.ptr(k_d), L.ptr(p_d),
                                        L.ptr(n_d), 5, L.ptr(colsum), 0 if colsum is None else colsum.shape[1],
                                        L.ptr(noise), 0 if noise is None else noise.shape[2], int(seed), int(offset),
                                        L.ptr(ids), L.ptr(draws), L.ptr(forced), 0 if forced is None else forced.shape[1], stream))
        return ids, metrics, forced

    def _sample_entropix_call(self, logits, attentions, temperature, top_k, top_p, seed=0, noise=None):
        """Base.sample(attentions=...) (base.py:1325-1342): ``attentions`` as the reference's forward returns them -- a
        sequence over layers of (B, H, T, T) raw scaled products, of which the LAST layer's newest row per utterance is read
        (samplers.py:294, :482) -- or just [last layer], which is what forward_super(output_attentions=True) gives here."""
        lib = L.load()
        B = len(logits)
        V = int(logits[0].shape[-1])
        lens = [int(l.shape[0]) for l in logits]
        last = attentions[-1]
        ld = _round_up(V, 8)
        flat = torch.zeros((sum(lens), ld), dtype=torch.float32, device=self.device)
        flat[:, :V] = torch.cat([l.to(self.device, torch.float32) for l in logits])
        off = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
        scores = torch.zeros((B, last.shape[1], max(lens)), dtype=torch.float32, device=self.device)
        for b in range(B):
            scores[b, :, :lens[b]] = last[b, :, lens[b] - 1, :lens[b]].to(self.device, torch.float32)
        colsum = torch.empty((B, ld), dtype=torch.float32, device=self.device)
        off_d = self._i32(off)
        L.check(lib.valle_logsoftmax_colsum(L.ptr(flat), ld, V, L.ptr(off_d), B, L.ptr(colsum), ld, 0, L.stream_ptr(self.device)))
        ids, metrics, _ = self.entropix_sample(flat, off[1:] - 1, V, scores, lens, colsum, temperature=temperature, top_k=top_k,
                                               top_p=top_p, seed=seed, noise=noise)
        return Sampled([ids[b:b + 1].long() for b in range(B)], logits, None, metrics)

    def sample(self, logits, prev_list=None, **sampling_kwargs):
        """``Base.sample`` (base.py:1280-1430): logits = list of per-utterance (rows, vocab) tensors; with
        ``prev_list`` of equal row counts the call is a NAR one (all rows sampled), otherwise only the last
        ``causal_size`` row of each utterance is (AR).  -> Sampled(ids, logits, scores, entropy=None)."""
        B = len(logits)
        V = int(logits[0].shape[-1])
        if sampling_kwargs.get("attentions") is not None and sampling_kwargs.get("quant_levels") is None:   # base.py:1327
            kw = sampling_kwargs
            return self._sample_entropix_call(logits, kw["attentions"], kw.get("temperature", 1.0), kw.get("top_k", -100),
                                              kw.get("top_p", 1.0), seed=kw.get("seed", 0), noise=kw.get("entropix_noise"))
        if prev_list is not None and all(len(p) == l.shape[0] for p, l in zip(prev_list, logits)):
            nar, rows = 1, logits
        else:
            nar, rows = 0, [l[-self.causal_size:] for l in logits]              # base.py:1358-1360
        counts = [r.shape[0] for r in rows]
        off = np.concatenate(([0], np.cumsum(counts))).astype(np.int32)
        ld = _round_up(V, 8)
        flat = torch.zeros((int(off[-1]), ld), dtype=torch.float32, device=self.device)
        flat[:, :V] = torch.cat([r.to(self.device, torch.float32) for r in rows])
        mirostat = sampling_kwargs.pop("mirostat", None)
        if mirostat is not None:
            return self._sample_mirostat(flat, V, logits, rows, prev_list, mirostat, self.sample_params(**sampling_kwargs))
        ids, scores = self.sample_rows(flat, V, off, prev_list, nar_mode=nar, params=self.sample_params(**sampling_kwargs))
        cut = lambda t: [t[off[i]:off[i + 1]] for i in range(B)]
        return Sampled(cut(ids.long()), logits, cut(scores), None)

    def _sample_mirostat(self, flat, V, logits, rows, prev_list, mirostat, params):
        """Base.sample(mirostat=[state, ...]) (base.py:1404-1407, samplers.py:197-238): one state dict per utterance, the
        utterance's last row is drawn from, the dicts are updated in place and returned as ``scores`` -- as the reference
        does.  The draw uses the Philox stream of ``seed`` (torch.multinomial's stream cannot be reproduced)."""
        lib = L.load()
        B = len(rows)
        if any(r.shape[0] != 1 for r in rows):                                   # mirostat_sample reads logits[-1, :]
            keep = np.cumsum([r.shape[0] for r in rows]) - 1
            flat = flat[torch.from_numpy(keep).to(self.device)].contiguous()
        host = (L.MirostatState * B)()
        for i, st in enumerate(mirostat):
            st.setdefault("max_surprise", st["tau"] * 2)                          # samplers.py:213-220
            st.setdefault("error_surprise", 0)
            st.setdefault("running_total_surprise", 0)
            host[i] = L.MirostatState(float(st["n"]), float(st["tau"]), float(st["eta"]), float(st["max_surprise"]),
                                      float(st["error_surprise"]), float(st["running_total_surprise"]), 0.0, 0, 0)
        nbytes = C.sizeof(host)
        dev_states = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(self.device)
        prev = prev_off = prev_len = None
        if prev_list is not None:
            pl = [(_np(p) if _np(p).ndim == 1 else _np(p)[:, -1]).astype(np.int32) for p in prev_list]
            lens = np.array([len(p) for p in pl], dtype=np.int32)
            prev_off = self._i32(np.concatenate(([0], np.cumsum(lens)[:-1])))
            prev_len = self._i32(lens)
            prev = self._i32(np.concatenate(pl) if lens.sum() else np.zeros(1, np.int32))
        ids = torch.empty(B, dtype=torch.int32, device=self.device)
        scores = torch.empty(B, dtype=torch.float32, device=self.device)
        L.check(lib.valle_sample_mirostat(L.ptr(flat), flat.shape[1], V, B, L.ptr(prev), L.ptr(prev_off), L.ptr(prev_len),
                                          C.byref(params), L.ptr(dev_states), L.ptr(ids), L.ptr(scores), L.stream_ptr(self.device)))
        back = (L.MirostatState * B).from_buffer_copy(dev_states.cpu().numpy().tobytes()[:nbytes])
        for st, b in zip(mirostat, back):
            if b.k < 0:                                                       # compute_k's failures, as Python meets them
                raise {-1: ZeroDivisionError("float division by zero"),         # samplers.py:202 / :209 (e.g. top_no left -inf behind)
                       -2: TypeError("type complex doesn't define __round__ method"),   # negative base ** fraction, :209-210
                       -3: OverflowError("cannot convert float infinity to integer"),   # :210
                       }.get(b.k, RuntimeError("mirostat: empty candidate set (sorted_logits[0:k] with k <= -vocab)"))
            st.update(index_surprise=b.index_surprise, running_total_surprise=b.running_total_surprise,
                      error_surprise=b.error_surprise, max_surprise=b.max_surprise,
                      token=torch.tensor([b.token], device=self.device), k=b.k)
        return Sampled([st["token"] for st in mirostat], logits, list(mirostat), None)

    def sample_rows(self, logits, vocab, row_off, prev_list, nar_mode, params, dev=None):
        """valle_sample over packed logits rows; row_off = per-utterance row offsets (B+1).
        dev = (row_item, row_in_item, prev, prev_off, prev_len) already on the device."""
        lib = L.load()
        B = len(row_off) - 1
        rows = int(row_off[-1])
        if dev is not None:
            ri, rn, prev, prev_off, prev_len = dev
            ids = torch.empty(max(rows, 1), dtype=torch.int32, device=self.device)
            scores = torch.empty(max(rows, 1), dtype=torch.float32, device=self.device)
            L.check(lib.valle_sample(L.ptr(logits), logits.shape[1], vocab, rows, L.ptr(ri), L.ptr(rn), L.ptr(prev),
                                     L.ptr(prev_off), L.ptr(prev_len), int(nar_mode), C.byref(params), L.ptr(ids), L.ptr(scores),
                                     L.stream_ptr(self.device)))
            return ids[:rows], scores[:rows]
        counts = np.diff(row_off)
        row_item = np.repeat(np.arange(B, dtype=np.int32), counts)
        row_in = np.concatenate([np.arange(c, dtype=np.int32) for c in counts]) if rows else np.zeros(0, np.int32)
        prev = prev_off = prev_len = None
        if prev_list is not None:
            pl = [(_np(p) if _np(p).ndim == 1 else _np(p)[:, -1]).astype(np.int32) for p in prev_list]
            lens = np.array([len(p) for p in pl], dtype=np.int32)
            prev_off = self._i32(np.concatenate(([0], np.cumsum(lens)[:-1])))
            prev_len = self._i32(lens)
            flat = np.concatenate(pl) if lens.sum() else np.zeros(1, np.int32)
            prev = self._i32(flat)
        ids = torch.empty(max(rows, 1), dtype=torch.int32, device=self.device)
        scores = torch.empty(max(rows, 1), dtype=torch.float32, device=self.device)
        ri, rn = self._i32(row_item), self._i32(row_in)
        L.check(lib.valle_sample(L.ptr(logits), logits.shape[1], vocab, rows, L.ptr(ri), L.ptr(rn), L.ptr(prev),
                                 L.ptr(prev_off), L.ptr(prev_len), int(nar_mode), C.byref(params), L.ptr(ids), L.ptr(scores),
                                 L.stream_ptr(self.device)))
        return ids[:rows], scores[:rows]

    def cfg_combine(self, pos_logits, neg_logits, vocab, row_off, strength, rescale):
        lib = L.load()
        off = self._i32(row_off)
        B = len(row_off) - 1
        if B >= 8:      # many utterances (NAR-len batches): an utterance's rows over 8 workgroups, partial sums through a workspace
            nbytes = int(lib.valle_cfg_combine_workspace_bytes(B))
            ws = torch.empty(nbytes // 8, dtype=torch.float64, device=self.device)   # (stream-ordered: the allocator keeps it until the launches ran)
            L.check(lib.valle_cfg_combine_ws(L.ptr(pos_logits), L.ptr(neg_logits), pos_logits.shape[1], vocab, L.ptr(off), B,
                                             float(strength), float(rescale), L.ptr(ws), nbytes, L.stream_ptr(self.device)))
            return pos_logits
        L.check(lib.valle_cfg_combine(L.ptr(pos_logits), L.ptr(neg_logits), pos_logits.shape[1], vocab, L.ptr(off),
                                      B, float(strength), float(rescale), L.stream_ptr(self.device)))
        return pos_logits


def pack_stream_order(w: torch.Tensor) -> torch.Tensor:
    """[N][K] -> [ceil(N/16)][K/64][2][64 lanes][8] (valle_hip.h, valle_layer_weights.p_*)."""
    N, K = w.shape
    Np = _round_up(N, 16)
    if Np != N:
        w = torch.cat([w, w.new_zeros(Np - N, K)])
    t = w.reshape(Np // 16, 16, K // 64, 2, 4, 8)        # nt, fr, ks, half, fq, j
    return t.permute(0, 2, 3, 4, 1, 5).contiguous().reshape(-1)


def narrow_tile_rows(w: torch.Tensor, S: int) -> torch.Tensor:
    """[N][K] -> [N*S][K/S]: the rows of a 16-row tile are (K part, output column) pairs of 16/S output columns
    (narrow tiles of o_proj / down_proj, csrc/dec_linear.hip; S == DEC_S_O == DEC_S_DOWN of csrc/common.h)."""
    N, K = w.shape
    C = 16 // S
    assert N % C == 0 and K % (64 * S) == 0, (N, K, S)
    return w.reshape(N // C, C, S, K // S).permute(0, 2, 1, 3).reshape(N * S, K // S).contiguous()


def _np(t):
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().numpy().astype(np.int64)
    return np.asarray(t, dtype=np.int64)
