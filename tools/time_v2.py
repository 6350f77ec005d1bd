"""Timing of the version >= 7 model family at the bench's sizes (d_model 1,024, 12 layers, bf16, batch 16, 80 phonemes, 225 prompt
frames): NAR-len demasking of all 8 levels at 450 frames x 25 steps, and the AR loop of 8 codes per step.  One JSON line each."""
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
cfg.version = 7
model = get_model(cfg).load_state_dict(synth_state_dict(cfg, 1234, stop_bias=-30000.0)).to(dev, torch.bfloat16)
B = 16
phns, proms, langs = synth_batch(7, B, n_phn=80, n_prom=225)
kw = dict(task_list=["tts"] * B, phns_list=phns, proms_list=proms, lang_list=langs)


def timed(fn, reps):
    fn(); torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / reps, out


which = sys.argv[1:] or ["narlen", "ar"]
if "narlen" in which:
    R = 450
    dt, out = timed(lambda: model(**kw, len_list=[R] * B, max_steps=25, temperature=0.0), 2)
    assert out[0].shape == (R, 8)
    print(json.dumps({"config": "v2 NAR-len: batch 16, 450 frames x 8 levels, 25 demasking steps with guidance, greedy", "ms": round(dt * 1e3, 1),
                      "rvq_frames_per_s": round(B * R / dt, 1)}), flush=True)
if "ar" in which:
    R = int(sys.argv[sys.argv.index("ar") + 1]) if "ar" in sys.argv and len(sys.argv) > sys.argv.index("ar") + 1 else 120
    dt, out = timed(lambda: model(**kw, max_duration=R, temperature=0.0, step_chunk=R), 1)
    assert out[0].shape == (R, 8), out[0].shape
    print(json.dumps({"config": f"v2 AR: batch 16, {R} frames x 8 levels, greedy", "ms": round(dt * 1e3, 1),
                      "rvq_frames_per_s": round(B * R / dt, 1), "ms_per_step": round(dt * 1e3 / R, 3)}), flush=True)
