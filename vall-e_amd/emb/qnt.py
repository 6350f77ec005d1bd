"""Integer helpers over RVQ code tensors, mirroring ``vall_e/emb/qnt.py`` (reference lines cited per function).

Only the functions that touch codes -- prompt preparation (``inference.py:139-169``) and the AR loop's
``prefix_silence`` (``models/ar_nar.py:730-732``) -- are here; they run on the host on a few KB of int16, as in the
reference.  Encoding / decoding waveforms needs the codec networks (EnCodec / Vocos, third-party, weights not
available offline): those entry points raise.  Known-answer vectors: ``tests/golden/qnt_kat.npz``.
"""
from __future__ import annotations

import math
import random

import torch

FRAMES_PER_SECOND = 75                      # 24 kHz EnCodec / Vocos (emb/qnt.py:570-583)
ENCODEC_SILENCE = (62, 424, 786, 673, 622, 986, 570, 948)   # one frame of quantised silence (emb/qnt.py:591)


def _time_axis_is_rows(qnt):
    return qnt.shape[0] > qnt.shape[1]


def _cut(qnt, start, end):
    return qnt[start:end] if _time_axis_is_rows(qnt) else qnt[:, start:end]


def post_process(codes):
    """emb/qnt.py:447-454: drop a leading batch axis, return (frame, codebook)."""
    if codes.dim() == 3:
        codes = codes[0]
    return codes.t() if codes.shape[0] < codes.shape[1] else codes


def trim(qnt, target, reencode=False, device=None):
    """emb/qnt.py:459-475.  target > 0: the first ``target`` frames -- or, when the clip is not longer than that, the
    slice [length - target, length) with Python's meaning of a negative start (the tail of a too-short clip);
    target <= 0: the last ``-target`` frames (0: nothing)."""
    if reencode:
        raise NotImplementedError("trim(reencode=True) decodes and re-encodes a waveform: needs the codec")
    length = max(qnt.shape[0], qnt.shape[1])
    if target > 0:
        start, end = (0, target) if target < length else (length - target, length)
    else:
        start, end = max(length + target, 0), length
    return _cut(qnt, start, end)


def trim_random(qnt, target):
    """emb/qnt.py:488-496 (Python's ``random``: the window is not reproducible across implementations)."""
    length = max(qnt.shape[0], qnt.shape[1])
    start = int(length * random.random())
    end = start + target
    if end >= length:
        start, end = length - target, length
    return _cut(qnt, start, end)


def repeat_extend_audio(qnt, target):
    """emb/qnt.py:499-506: tile the clip along dim 0 until it has at least ``target`` rows, then ``trim``."""
    n = 0
    rows = 0
    while rows < target:
        n += 1
        rows += qnt.shape[0]
    return trim(torch.cat([qnt] * n), target)      # target <= 0: torch.cat of nothing raises, as in the reference


def concat_audio(*args, reencode=False, device=None):
    """emb/qnt.py:528-533 (codes joined as they are)."""
    if reencode:
        raise NotImplementedError("concat_audio(reencode=True) needs the codec")
    return torch.concat([q for q in args if q is not None])


def get_framerate(backend=None, sample_rate=None):
    """emb/qnt.py:570-583 for the built backends."""
    if backend == "dac" and sample_rate in (44_100, 16_000):
        return 87 if sample_rate == 44_100 else 50
    return FRAMES_PER_SECOND


def get_silence(length, device=None, codes=None):
    """emb/qnt.py:586-593: floor(seconds * 75) frames of the EnCodec silence codes, int16 (frame, 8)."""
    n = math.floor(length * get_framerate())
    return torch.tensor([list(ENCODEC_SILENCE) for _ in range(n)], device=device, dtype=torch.int16).reshape(n, len(ENCODEC_SILENCE))


def pad_codes_with_silence(codes, size=1):
    """emb/qnt.py:596-604, including its unit slip: the 'duration' is frames * 75, so the padding is ``size`` seconds
    split as floor(size / 2 * 75) frames behind the clip and the rest in front."""
    duration = codes.shape[0] * get_framerate()
    difference = math.ceil(duration + size) - duration
    silence = get_silence(difference, device=codes.device)[:, : codes.shape[-1]]
    half = math.floor(difference / 2 * get_framerate())
    return torch.concat([silence[half:, :], codes, silence[:half, :]], dim=0)


def encode(*a, **k):
    raise NotImplementedError("waveform -> codes needs the codec networks (EnCodec / Vocos): not built (SURVEY.md §8 f2)")


decode = decode_to_wave = decode_to_file = encode_from_file = encode_from_files = merge_audio = encode
