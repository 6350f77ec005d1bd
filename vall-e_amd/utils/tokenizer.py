"""Phoneme tokenizer: a restatement of what the reference gets from ``PreTrainedTokenizerFast(tokenizer_file=
data/tokenizer.json)`` (reference config.py:1170-1184; call sites data.py:640-643 ``tokenize`` and
inference.py:129 ``encode_text``) for the one tokenizer shape that file has: a BPE model with no normalizer and
no pre-tokenizer (the whole phoneme string is one word), special tokens matched verbatim first, unknown
characters mapped to ``<unk>`` one by one, and a TemplateProcessing post-processor that wraps the ids in
``<bos> ... <eos>``.  Third-party algorithm (Hugging Face ``tokenizers`` BPE): lowest-rank adjacent pair merges
first, leftmost occurrence first; pinned against that library's output in tests/golden/tokenizer_kat.json.

Host-side only (it produces the uint8/int16 id tensors the model consumes); the phonemizer in front of it
(espeak through ``phonemizer``, reference emb/g2p.py) is not part of this package: ``encode`` takes IPA strings.
"""
from __future__ import annotations

import heapq
import json
from pathlib import Path


class PhonemeTokenizer:
    def __init__(self, spec: dict):
        model = spec["model"]
        if model.get("type") != "BPE" or spec.get("normalizer") or spec.get("pre_tokenizer"):
            raise NotImplementedError("only the reference's tokenizer shape is restated: BPE, no normalizer, no pre-tokenizer")
        if model.get("byte_fallback") or model.get("continuing_subword_prefix") or model.get("end_of_word_suffix") or model.get("dropout"):
            raise NotImplementedError("BPE options outside the reference's tokenizer.json")
        self.vocab = dict(model["vocab"])
        self.unk = self.vocab[model["unk_token"]] if model.get("unk_token") else None
        self.fuse_unk = bool(model.get("fuse_unk"))
        merges = [tuple(m.split(" ")) if isinstance(m, str) else tuple(m) for m in model["merges"]]
        self.merge_ids = {}                                                # (id_a, id_b) -> (rank, merged id)
        for r, (x, y) in enumerate(merges):
            self.merge_ids.setdefault((self.vocab[x], self.vocab[y]), (r, self.vocab[x + y]))
        self.special = sorted((t["content"] for t in spec.get("added_tokens", [])), key=len, reverse=True)
        self.special_ids = {t["content"]: t["id"] for t in spec.get("added_tokens", [])}
        self.prefix, self.suffix = [], []
        post = spec.get("post_processor")
        if post:
            if post.get("type") != "TemplateProcessing":
                raise NotImplementedError(f"post processor {post.get('type')}")
            ids = {k: v["ids"] for k, v in post["special_tokens"].items()}
            seen_seq = False
            for piece in post["single"]:
                if "Sequence" in piece:
                    seen_seq = True
                else:
                    (self.suffix if seen_seq else self.prefix).extend(ids[piece["SpecialToken"]["id"]])
        self.id_to_token = {}
        for tok, i in self.vocab.items():                                  # several strings share an id; the library's pick
            self.id_to_token.setdefault(i, tok)                            # follows hash order and changes between
                                                                           # processes -- here: the first in file order

    @classmethod
    def from_file(cls, path):
        return cls(json.loads(Path(path).read_text(encoding="utf-8")))

    def get_vocab(self):
        """data.py:637 ``get_phone_symmap``."""
        return dict(self.vocab) | dict(self.special_ids)

    # ------------------------------------------------------------------------------- encode
    def _split_special(self, text):
        """[(piece, is_special)]: added tokens are cut out of the raw string before the model sees it."""
        out, i, start = [], 0, 0
        while i < len(text):
            hit = next((s for s in self.special if text.startswith(s, i)), None)
            if hit is None:
                i += 1
                continue
            if i > start:
                out.append((text[start:i], False))
            out.append((hit, True))
            i += len(hit)
            start = i
        if start < len(text):
            out.append((text[start:], False))
        return out

    def _bpe(self, word):
        """Merges run on token IDS, as in the library: tokenizer.json maps several strings to one id (e.g. the
        look-alike letters of other scripts), and a merge rule written for one of them applies to all of them."""
        syms = []
        for ch in word:
            if ch in self.vocab:
                syms.append(self.vocab[ch])
            elif self.unk is not None and not (self.fuse_unk and syms and syms[-1] == self.unk):
                syms.append(self.unk)
        n = len(syms)
        if n == 0:
            return []
        nxt = list(range(1, n)) + [-1]
        prv = [-1] + list(range(n - 1))
        alive = [True] * n
        heap = []
        for i in range(n - 1):
            m = self.merge_ids.get((syms[i], syms[i + 1]))
            if m is not None:
                heap.append((m[0], i, syms[i], syms[i + 1]))
        heapq.heapify(heap)
        while heap:
            r, i, a, b = heapq.heappop(heap)
            j = nxt[i] if alive[i] else -1
            if j < 0 or not alive[j] or syms[i] != a or syms[j] != b:
                continue                                                   # stale entry: a neighbour was merged away
            syms[i] = self.merge_ids[(a, b)][1]
            alive[j] = False
            nxt[i] = nxt[j]
            if nxt[j] >= 0:
                prv[nxt[j]] = i
            p, q = prv[i], nxt[i]
            if p >= 0:
                m = self.merge_ids.get((syms[p], syms[i]))
                if m is not None:
                    heapq.heappush(heap, (m[0], p, syms[p], syms[i]))
            if q >= 0:
                m = self.merge_ids.get((syms[i], syms[q]))
                if m is not None:
                    heapq.heappush(heap, (m[0], i, syms[i], syms[q]))
        ids, i = [], 0
        while i >= 0:
            ids.append(syms[i])
            i = nxt[i]
        return ids

    def encode(self, text, add_special_tokens=True):
        if isinstance(text, (list, tuple)):
            text = "".join(text)                                           # data.py:640-642
        ids = []
        for piece, special in self._split_special(text):
            ids.extend([self.special_ids[piece]] if special else self._bpe(piece))
        return self.prefix + ids + self.suffix if add_special_tokens else ids

    def decode(self, ids, skip_special_tokens=False):
        """Concatenated token strings (the ByteLevel decoder of the file maps these code points to themselves)."""
        special = set(self.special_ids.values())
        return "".join(self.id_to_token.get(int(i), "") for i in ids if not (skip_special_tokens and int(i) in special))
