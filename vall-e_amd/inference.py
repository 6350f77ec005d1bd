"""``TTS``: the reference's API façade for the batched path, on pre-tokenised inputs.

Mirrors ``vall_e.inference.TTS`` (reference inference.py:37-597) for what sits on the hot path:
``batched_inference`` sorts utterances by ``len(text) + 1 + len(prompt) + 1`` (inference.py:288-293),
cuts chunks of ``batch_size`` (:296-311) and runs AR then NAR, or (NAR-len) duration prediction
(task ``len``, unless ``len_list`` is given), demasking then NAR, per chunk (:326-346).  The reference's text front end (phonemizer) and codec (EnCodec/Vocos) are
out of scope: texts are phoneme-id tensors, references are ``(P, 8)`` code tensors, and the
result is the list of ``(R, 8)`` code tensors the reference would hand to ``qnt.decode``
(inference.py:350-355), in the caller's order.

Multi-GPU: utterances are independent, so ``shard_utterances`` deals the length-sorted list
round-robin to ranks and ``gather_codes`` returns every rank's codes with one all-gather
(RCCL on GPUs, gloo in the CPU tests) -- the only collective of the path.
"""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch

from .config import Model as ModelConfig, repo_default_model
from .models import get_model


def sort_key(text, prom):
    """inference.py:288."""
    return int(text.shape[0]) + 1 + (int(prom.shape[0]) if prom is not None else 0) + 1


def shard_utterances(lengths, world_size):
    """Length-sorted utterance indices dealt round-robin: rank r gets sorted[r::world_size], so every
    rank sees the same mix of short and long utterances."""
    order = sorted(range(len(lengths)), key=lambda i: (lengths[i], i))
    return [order[r::world_size] for r in range(world_size)]


def derive_seed(seed, *parts):
    """One Philox seed per model call from (user seed, chunk or line index, rank, phase): the reference reseeds torch per
    batch (``set_seed(seed)``, inference.py:315, :403) and picks a random seed when none is given; a counter-based generator
    needs distinct seeds instead, or every chunk, rank and phase would draw the same uniforms.  splitmix64 over the parts."""
    import os
    x = int(seed) & 0xFFFFFFFFFFFFFFFF if seed is not None else int.from_bytes(os.urandom(8), "little")
    for p in parts:
        x = (x + 0x9E3779B97F4A7C15 + (int(p) & 0xFFFFFFFFFFFFFFFF)) & 0xFFFFFFFFFFFFFFFF
        x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        x ^= x >> 31
    return x & 0x7FFFFFFFFFFFFFFF


def chunk(indices, batch_size):
    """inference.py:296-311."""
    batch_size = max(1, int(batch_size))
    return [indices[i:i + batch_size] for i in range(0, len(indices), batch_size)]


def pack_codes(codes, r_max, levels=8):
    """list of (R_i, levels) integer tensors -> (int16 [n, r_max, levels], int32 [n] lengths)."""
    block = torch.zeros((len(codes), r_max, levels), dtype=torch.int16)
    lens = torch.zeros(len(codes), dtype=torch.int32)
    for i, c in enumerate(codes):
        c = c.detach().to("cpu", torch.int16)
        if c.shape[0] > r_max:
            raise ValueError(f"utterance {i}: {c.shape[0]} frames exceed the gather block ({r_max})")
        block[i, : c.shape[0]] = c
        lens[i] = c.shape[0]
    return block, lens


def gather_codes(local_codes, local_indices, n_total, r_max, group=None, device=None):
    """All ranks' codes in the caller's utterance order.  One all-gather of a fixed-size padded block
    per rank (codes int16 [n_loc_max][r_max][8], lengths, utterance ids); every rank returns the list."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        out = [None] * n_total
        for i, c in zip(local_indices, local_codes):
            out[i] = c
        return out
    n_loc_max = (n_total + world - 1) // world
    block, lens = pack_codes(local_codes, r_max)
    pad = n_loc_max - block.shape[0]
    ids = torch.full((n_loc_max,), -1, dtype=torch.int32)
    ids[: len(local_indices)] = torch.tensor(local_indices, dtype=torch.int32)
    if pad:
        block = torch.cat([block, block.new_zeros((pad,) + tuple(block.shape[1:]))])
        lens = torch.cat([lens, lens.new_zeros(pad)])
    dev = device if device is not None else "cpu"
    block, lens, ids = block.to(dev), lens.to(dev), ids.to(dev)
    block = block.view(torch.uint8)          # neither gloo nor RCCL carries int16: ship the bytes
    blocks = [torch.empty_like(block) for _ in range(world)]
    all_lens = [torch.empty_like(lens) for _ in range(world)]
    all_ids = [torch.empty_like(ids) for _ in range(world)]
    dist.all_gather(blocks, block, group=group)
    dist.all_gather(all_lens, lens, group=group)
    dist.all_gather(all_ids, ids, group=group)
    out = [None] * n_total
    for b, l, u in zip(blocks, all_lens, all_ids):
        b = b.view(torch.int16)
        for j in range(n_loc_max):
            i = int(u[j])
            if i >= 0:
                out[i] = b[j, : int(l[j])].clone()
    return out


class TTS:
    """Constructor keeps the reference's keyword names (inference.py:38); ``config`` is a
    ``vall_e_amd.config.Model`` (or None for the repo default, data/config.yaml) and ``state_dict`` the
    reference's parameter names -> tensors (random-init synthetic weights if omitted)."""

    def __init__(self, config=None, lora=None, device=None, amp=None, dtype=None, attention=None, state_dict=None, seed=1234):
        if attention not in (None, "auto", "valle_hip"):
            raise NotImplementedError(f"attention backend {attention!r}: the HIP path has one attention implementation")
        if isinstance(config, (str, Path)):
            # inference.py:47-89: a weights file carries its own model config in its metadata; a YAML names the model
            path = Path(config)
            if path.suffix in (".yaml", ".yml"):
                config = ModelConfig.from_yaml(path)
            else:
                from .utils.io import load_model_file
                config, state_dict = load_model_file(path)
        self.config = config if isinstance(config, ModelConfig) else repo_default_model()
        self.device = torch.device(device if device is not None else "cuda")
        self.dtype = dtype if dtype is not None else torch.float32        # data/config.yaml:118 weight_dtype float32
        self.amp = bool(amp)
        if self.amp and self.dtype != torch.float32:
            # inference.py:96-98, :516: with amp the reference keeps fp32 weights and runs under torch.autocast(dtype): the
            # linears compute in the low dtype on per-call casts, while embeddings, residual stream and norms stay fp32.
            # That is other arithmetic than model.to(dtype) (the one low-precision form built here), so it is refused
            # rather than silently replaced.  amp with dtype float32 is plain fp32 in the reference as well (autocast to
            # float32 is a no-op), and is accepted.
            raise NotImplementedError("amp=True with a low-precision dtype (fp32 master weights under autocast) is not built; "
                                      "use amp=False with dtype=torch.bfloat16 (model.to(bfloat16) semantics)")
        self.batch_size = 16                                               # config.py:856
        self.tokenizer = None
        if state_dict is None:
            from .weights import synth_state_dict
            state_dict = synth_state_dict(self.config, seed)
        self.model = get_model(self.config).load_state_dict(state_dict).to(self.device, self.dtype)
        if lora is not None:                                               # inference.py:38-45, engines/__init__.py:250-262
            self.load_lora(lora)

    def load_lora(self, lora):
        """A LoRA weights file as Engines.export / extract_lora write it (export.py:190-222: tensors under ``lora`` or
        ``module``, ``config`` = the LoRA dataclass: rank, alpha, rvq_levels), or a dict
        ``{"lora": tensors, "config": {...}}``."""
        if isinstance(lora, (str, Path)):
            from .utils.io import torch_load
            lora = torch_load(lora)
        tensors = lora.get("lora") or lora.get("module")
        conf = lora.get("config") or {}
        self.model.load_lora(tensors, rank=conf.get("rank", 128), alpha=conf.get("alpha", 128), rvq_levels=conf.get("rvq_levels", []))

    def enable_lora(self, enabled=True):
        """inference.py:104-109.  As in the reference, every generation call switches the LoRA again per RVQ level
        (``use_lora=`` overrides), so this only sets the state between calls."""
        self.model._call_state.lora_on = bool(enabled) and self.model._lora is not None

    def disable_lora(self):
        return self.enable_lora(False)

    # ---------------------------------------------------------------- text front end (ids only)
    LANG_SYMMAP = {"en": 0, "ja": 1, "de": 2, "fr": 3, "zh": 4, "ko": 5}     # data.py:650-658

    def load_tokenizer(self, path):
        """The reference's phoneme tokenizer file (data/tokenizer.json, config.py:1170-1184)."""
        from .utils.tokenizer import PhonemeTokenizer
        self.tokenizer = PhonemeTokenizer.from_file(path)
        return self.tokenizer

    def encode_text(self, text, language="auto", precheck=True, phonemize=True):
        """inference.py:111-129 for input that is already phonemes: a tensor passes through, an IPA string is
        tokenised (``<bos> ... <eos>``).  With a grapheme-to-phoneme backend registered (``vall_e_amd.emb.g2p.set_backend``;
        the reference's is espeak through ``phonemizer``, emb/g2p.py) plain text goes through it first, as :129 does;
        without one -- or with ``language="ipa"`` -- the string is taken as phonemes."""
        if isinstance(text, torch.Tensor):
            return text
        if getattr(self, "tokenizer", None) is None:
            raise RuntimeError("no tokenizer loaded: call load_tokenizer(path to the reference's data/tokenizer.json)")
        from .emb import g2p
        if phonemize and language != "ipa" and g2p._backend is not None:        # inference.py:129: tokenize(g2p.encode(text, language))
            text = g2p.encode(text, language=language)
        ids = self.tokenizer.encode(text)
        return torch.tensor(ids, dtype=torch.uint8 if len(self.tokenizer.get_vocab()) < 256 else torch.int16)   # inference.py:284

    def encode_audio(self, paths, trim_length=0.0):
        """inference.py:139-169: a code tensor is returned as it is (``trim_length`` applies only to clips the
        reference encodes from files, through ``repeat_extend_audio`` -- vall_e_amd.emb.qnt has it for callers that
        hold codes); audio files need the codec networks, which are not built."""
        if isinstance(paths, torch.Tensor):
            return paths
        raise NotImplementedError("encode_audio(path): waveform -> codes needs the codec (EnCodec / Vocos), not built; "
                                  "pass the prompt as a (frames, 8) code tensor")

    def encode_lang(self, language):
        """inference.py:131-137."""
        return torch.tensor([self.LANG_SYMMAP.get(language, 0)], dtype=torch.uint8)

    def modality(self, modality):
        """inference.py:210-214: 'auto' resolves to the model's name when that names a modality."""
        if modality in (None, "auto"):
            name = getattr(self.config, "name", "ar+nar")
            return name if name in ("ar+nar", "nar-len") else "ar+nar"
        return modality

    @torch.inference_mode()
    def inference(self, text, references, language="auto", text_language=None, task="tts", out_path=None, play=False,
                  **sampling_kwargs):
        """inference.py:361-597, the serial line-by-line path, for tasks ``tts``, ``ns`` and ``sr``: every line (a list of phoneme strings
        / id tensors, or one string split at newlines -- sentence splitting and language detection belong to the text
        front end) runs AR then NAR, or len -> NAR-len; with ``context_history`` = n > 0 the AR call of a line gets the
        previous n lines' phonemes and codes as ``prefix_context`` (:575-584).  Returns the list of per-line (R, 8)
        code tensors the reference hands to ``qnt.decode_to_file`` one after the other (:587).  ``max_duration`` may be
        a list (one value per line)."""
        if task not in ("tts", "ns", "sr"):
            raise NotImplementedError(f"task {task!r}: TTS.inference is built for 'tts', 'ns' and 'sr'")
        seed = sampling_kwargs.pop("seed", None)              # None: a fresh one per call, as the reference does
        if seed is None:
            seed = derive_seed(None)
        input_prompt_length = sampling_kwargs.pop("input_prompt_length", 0)
        for k in ("tqdm", "dtype", "amp", "voice_convert", "split_text_by"):
            sampling_kwargs.pop(k, None)
        phonemize = sampling_kwargs.pop("phonemize", True)     # False: lines are raw-text ids, sent as text_list (inference.py:518-519)
        modality = self.modality(sampling_kwargs.pop("modality", "auto"))
        duration_padding = sampling_kwargs.pop("duration_padding", 1.05)
        context_history = int(sampling_kwargs.pop("context_history", 0))
        max_duration = sampling_kwargs.pop("max_duration", None)
        lines = text.split("\n") if isinstance(text, str) else list(text)
        lines = [l for l in lines if not isinstance(l, str) or l.strip()] or [""]
        prom = self.encode_audio(references, trim_length=input_prompt_length) if references is not None else None   # inference.py:486
        if task in ("ns", "sr"):                                                  # inference.py:489-493: the task token rides
            prom = [task, prom]                                                   # in front of the prompt; the model task stays tts
        lang = self.encode_lang(language) if isinstance(language, str) else language
        prefix_context, history, out = None, [], []
        for n, line in enumerate(lines):
            if phonemize:
                phns = self.encode_text(line)
                kw = dict(phns_list=[phns], proms_list=[prom], lang_list=[lang])
            else:
                if isinstance(line, str):
                    raise NotImplementedError("phonemize=False takes ids of the reference's text tokenizer, not strings")
                phns = torch.as_tensor(line)
                kw = dict(text_list=[phns], proms_list=[prom], lang_list=[lang])
            per_line = dict(sampling_kwargs)
            if max_duration is not None:
                per_line["max_duration"] = max_duration[n] if isinstance(max_duration, (list, tuple)) else max_duration
            if modality == "nar-len":
                if task in ("ns", "sr"):
                    lens = [int(prom[1].shape[0])]                                # inference.py:527-528: as long as the input
                else:
                    lens = self.model(**kw, task_list=["len"], max_duration=5)
                    lens = [int(max(l, 75) * duration_padding) for l in lens]     # inference.py:535-538 (75 frames per second)
                if prefix_context is not None:
                    per_line["prefix_context"] = prefix_context
                resps = self.model(**kw, task_list=["tts"], len_list=lens, seed=derive_seed(seed, n, 0, 1), **per_line)
            else:
                extra = {"prefix_context": prefix_context} if prefix_context is not None else {}
                resps = self.model(**kw, task_list=["tts"], seed=derive_seed(seed, n, 0, 0), **per_line, **extra)
                resps = self.model(**kw, task_list=["tts"], resps_list=resps, seed=derive_seed(seed, n, 0, 1), **per_line)
            r = resps[0]
            out.append(r)
            if context_history > 0:                                               # inference.py:575-584
                history.append((phns, r, int(r.shape[0])))
                last = history[-context_history:]
                prefix_context = ([torch.cat([torch.as_tensor(x[0]) for x in last])], [torch.cat([x[1] for x in last])],
                                  [sum(x[2] for x in last)])
        return out

    @torch.inference_mode()
    def batched_inference(self, texts, references=None, languages=None, text_languages=None, out_paths=None,
                          len_list=None, rank=0, world_size=1, **sampling_kwargs):
        """-> list of (R_i, 8) code tensors for THIS rank's utterances plus their indices when
        world_size > 1 (use ``gather_codes`` to assemble), else the full list in input order."""
        batch_size = sampling_kwargs.pop("batch_size", self.batch_size)
        modality = self.modality(sampling_kwargs.pop("modality", "auto"))
        seed = sampling_kwargs.pop("seed", None)              # inference.py:315: set_seed per chunk; None = a fresh seed
        if seed is None:
            seed = derive_seed(None)
        input_prompt_length = sampling_kwargs.pop("input_prompt_length", 0)
        for k in ("tqdm", "dtype", "amp"):
            sampling_kwargs.pop(k, None)
        duration_padding = sampling_kwargs.pop("duration_padding", 1.05)
        n = len(texts)
        texts = [self.encode_text(t) for t in texts]                          # IPA strings -> ids (inference.py:284)
        if languages and isinstance(languages[0], str):
            languages = [self.encode_lang(l) for l in languages]              # inference.py:286
        references = list(references) if references else [None] * n
        references = [self.encode_audio(r, trim_length=input_prompt_length) if r is not None else None for r in references]   # inference.py:281
        languages = list(languages) if languages else [torch.tensor([0], dtype=torch.uint8)] * n
        lengths = [sort_key(t, r) for t, r in zip(texts, references)]
        mine = shard_utterances(lengths, world_size)[rank]
        results = {}
        for ci, idx in enumerate(chunk(mine, batch_size)):
            kw = dict(task_list=["tts"] * len(idx), phns_list=[texts[i] for i in idx],
                      proms_list=[references[i] for i in idx], lang_list=[languages[i] for i in idx])
            if modality == "nar-len":
                if len_list is not None:
                    lens = [len_list[i] for i in idx]
                else:                                                            # inference.py:328-332
                    lens = self.model(**{**kw, "task_list": ["len"] * len(idx)}, max_duration=5)
                    lens = [int(l * duration_padding) for l in lens]
                resps = self.model(**kw, len_list=lens, seed=derive_seed(seed, ci, rank, 1), **sampling_kwargs)
            else:
                resps = self.model(**kw, seed=derive_seed(seed, ci, rank, 0), **sampling_kwargs)
                resps = self.model(**kw, resps_list=resps, seed=derive_seed(seed, ci, rank, 1), **sampling_kwargs)
            for i, r in zip(idx, resps):
                results[i] = r
        if world_size == 1:
            return [results[i] for i in range(n)]
        return [results[i] for i in mine], mine
