"""Timing of the BASELINE.json configurations that are parity cases, not the bench line (bench.py is configs[1]):
configs[2] NAR-len demasking at batch 64 and configs[4] long-form sampling at batch 32.  Synthetic inputs and
weights like bench.py; prints one JSON line per configuration."""
import json, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from vall_e_amd import config as vcfg
from vall_e_amd.models import get_model
from vall_e_amd.synth import synth_batch
from vall_e_amd.weights import synth_state_dict

dev = torch.device("cuda", 0)
cfg = vcfg.repo_default_model()
model = get_model(cfg).load_state_dict(synth_state_dict(cfg, 1234, stop_bias=-30000.0)).to(dev, torch.bfloat16)


def timed(fn, reps):
    fn(); torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / reps, out


which = sys.argv[1:] or ["narlen", "longform"]
if "narlen" in which:
    B, R = 64, 450
    phns, proms, langs = synth_batch(7, B, n_phn=80, n_prom=225)
    kw = dict(task_list=["tts"] * B, phns_list=phns, proms_list=proms, lang_list=langs)
    dt, out = timed(lambda: model(**kw, len_list=[R] * B, max_steps=25, temperature=0.0), 2)
    assert out[0].shape == (R, 8)
    print(json.dumps({"config": "configs[2] NAR-len: batch 64, 450 frames, 25 demasking steps + 7 NAR levels, greedy", "ms": round(dt * 1e3, 1),
                      "frames_per_s": round(B * R / dt, 1)}), flush=True)
if "longform" in which:
    B, R = 32, 2250
    phns, proms, langs = synth_batch(9, B, n_phn=400, n_prom=225)
    kw = dict(task_list=["tts"] * B, phns_list=phns, proms_list=proms, lang_list=langs)
    samp = dict(temperature=0.95, top_k=50, top_p=0.9, top_no=0.0, repetition_penalty=1.2, repetition_penalty_decay=0.5, length_penalty=0.3)
    dt, codes = timed(lambda: model(**kw, max_duration=R, step_chunk=64, seed=3, **samp), 1)
    assert codes[0].shape[0] == R
    print(json.dumps({"config": "configs[4] long-form AR: batch 32, 2250 frames (context to ~2.9k), top-k 50 / top-p 0.9 / repetition + length penalties",
                      "ms": round(dt * 1e3, 1), "level0_frames_per_s": round(B * R / dt, 1), "us_per_step": round(dt * 1e6 / R, 1)}), flush=True)
