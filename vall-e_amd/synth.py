"""Seeded synthetic inputs (SURVEY.md §8d "Synthetic inputs").

The same generator feeds the reference (when goldens are made), the oracle, the
HIP path and ``bench.py``; dtypes are the ones ``TTS.batched_inference`` sends to
the model (vall_e/inference.py:284-286): phonemes uint8, prompt int16, lang uint8.
"""
from __future__ import annotations

import torch


def synth_utterance(seed: int, n_phn: int = 80, n_prom: int = 225, n_levels: int = 8,
                    n_audio_tokens: int = 1024, lang: int = 0):
    """One utterance: ``[1] + randint(3,256,(n_phn-2)) + [2]`` phonemes, random prompt codes."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    body = torch.randint(3, 256, (max(n_phn - 2, 0),), generator=g, dtype=torch.int64)
    phn = torch.cat([torch.tensor([1]), body, torch.tensor([2])]).to(torch.uint8)
    prom = torch.randint(0, n_audio_tokens, (n_prom, n_levels), generator=g, dtype=torch.int64).to(torch.int16)
    return phn, prom, torch.tensor([lang], dtype=torch.uint8)


def synth_batch(seed: int, batch: int, n_phn=80, n_prom=225, ragged: bool = False, **kw):
    """``batch`` utterances; ``ragged`` draws Lp in [n_phn/2, 3n_phn/2], P in [2P/3, 4P/3]."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    phns, proms, langs = [], [], []
    for i in range(batch):
        lp, p = n_phn, n_prom
        if ragged:
            lp = int(torch.randint(max(n_phn // 2, 3), n_phn * 3 // 2 + 1, (1,), generator=g))
            p = int(torch.randint(max(n_prom * 2 // 3, 1), n_prom * 4 // 3 + 1, (1,), generator=g))
        a, b, c = synth_utterance(seed * 1000 + i + 1, lp, p, **kw)
        phns.append(a), proms.append(b), langs.append(c)
    return phns, proms, langs


# The phoneme string bundled with the reference's data/qnt.enc fixture
# ('wˈʌns ɐɡˈɛn, ˈɛksələnt wˈɜːk.') tokenised with data/tokenizer.json;
# oracle/gen_golden.py re-derives and checks these ids when the reference is mounted.
QNT_ENC_PHONEME_IDS = [1, 211, 33, 4, 48, 57, 123, 9, 4, 83, 171, 111, 139, 4, 37, 115, 25, 11, 2]
