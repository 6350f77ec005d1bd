"""Dev timing of valle_sample option by option (32 rows, vocab 1025, 2,000 previous tokens per row)."""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from vall_e_amd import config as vcfg
from vall_e_amd.models import get_model
from vall_e_amd.weights import synth_state_dict
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import model_configs as _mc
cfg = _mc.tiny_model()
m = get_model(cfg).load_state_dict(synth_state_dict(cfg, 1)).to("cuda:0", torch.float32)
B, V, NP = 32, 1025, 2000
g = torch.Generator().manual_seed(0)
lg = torch.zeros(B, 1032, device="cuda:0"); lg[:, :V] = torch.randn(B, V, generator=g).cuda() * 3
prev = [torch.randint(0, 1024, (NP,), generator=g) for _ in range(B)]
off = np.arange(B + 1, dtype=np.int32)
ri = torch.arange(B, dtype=torch.int32, device="cuda:0"); rn = torch.zeros(B, dtype=torch.int32, device="cuda:0")
pv = torch.cat(prev).int().cuda(); po = (torch.arange(B, dtype=torch.int32) * NP).cuda(); pl = torch.full((B,), NP, dtype=torch.int32).cuda()
dev = (ri, rn, pv, po, pl)
cases = {"greedy fast path": dict(temperature=0.0), "rep-pen": dict(temperature=0.0, repetition_penalty=1.2),
         "rep-pen+decay": dict(temperature=0.0, repetition_penalty=1.2, repetition_penalty_decay=0.5),
         "length-pen": dict(temperature=0.0, length_penalty=0.3), "top-k": dict(temperature=0.0, top_k=50, top_no=0.0),
         "top-p": dict(temperature=0.0, top_p=0.9, top_no=0.0), "categorical": dict(temperature=0.95, top_no=0.0),
         "dry": dict(temperature=0.0, dry_multiplier=0.8, dry_allowed_length=1),
         "all (long-form)": dict(temperature=0.95, top_k=50, top_p=0.9, top_no=0.0, repetition_penalty=1.2, repetition_penalty_decay=0.5, length_penalty=0.3)}
for name, kw in cases.items():
    sp = m.sample_params(**kw)
    for _ in range(3):
        m.sample_rows(lg, V, off, None, 0, sp, dev=dev)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        m.sample_rows(lg, V, off, None, 0, sp, dev=dev)
    e1.record(); torch.cuda.synchronize()
    print(f"{name:20s} {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us", flush=True)
