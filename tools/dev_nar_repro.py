"""dev: is a NAR run / a dense pass reproducible call after call in one process?  Three utterances of the bench shape with 10
response frames (the rank-0 shard of tests/test_bench_ranks.py); every result is compared bit for bit with the first one.
usage: python tools/dev_nar_repro.py N"""
import os
import sys

import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from vall_e_amd import config as vcfg                    # noqa: E402
from vall_e_amd.models import get_model                  # noqa: E402
from vall_e_amd.synth import synth_batch                 # noqa: E402
from vall_e_amd.weights import synth_state_dict          # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
cfg = vcfg.repo_default_model()
model = get_model(cfg).load_state_dict(synth_state_dict(cfg, 1234, stop_bias=-30000.0)).to(dev, torch.bfloat16)
gp, gr, gl = synth_batch(1234, 6, n_phn=80, n_prom=225)
sel = [0, 2, 4]
phns, proms, langs = [gp[i] for i in sel], [gr[i] for i in sel], [gl[i] for i in sel]
kw = dict(task_list=["tts"] * 3, phns_list=phns, proms_list=proms, lang_list=langs)
codes = model(**kw, max_duration=10, temperature=0.0)
full0 = [f.cpu() for f in model(**kw, resps_list=[c.clone() for c in codes], temperature=0.0)]
bad_nar = 0
for it in range(N):
    full = [f.cpu() for f in model(**kw, resps_list=[c.clone() for c in codes], temperature=0.0)]
    for b in range(3):
        if not torch.equal(full[b], full0[b]):
            d = (full[b] != full0[b]).nonzero()
            print("NAR iteration", it, "utterance", b, "differs at", d[:6].tolist(), flush=True)
            bad_nar += 1
# a second stream of this process keeps the GPU busy meanwhile (HOG=1): other kernels share the CUs and the memory system,
# but nothing is time-sliced between processes
hog = os.environ.get("HOG")
if hog:
    side = torch.cuda.Stream(device=dev)
    a = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
    b = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
    big = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    def poke(n=int(hog)):
        with torch.cuda.stream(side):
            for _ in range(n):
                torch.mm(a, b)
                big.add_(1)
else:
    def poke():
        pass
# one dense pass (level 3) on the final codes, the same plan every time
resp = [f[:, :3] for f in full0]
plan = model.plan(phns, langs, proms, resp, 3)
rows, off = model.tail_rows(plan, [10] * 3)
ref = model.forward_dense(plan, rows).clone()
bad = 0
for it in range(N * 4):
    poke()
    lg = model.forward_dense(plan, rows)
    if not torch.equal(lg, ref):
        d = (lg != ref).any(dim=1).nonzero().flatten().tolist()
        print("dense pass", it, "rows differ:", d, "max abs", float((lg - ref).abs().max()), flush=True)
        bad += 1
print("done: NAR mismatches", bad_nar, "of", N, "; dense mismatches", bad, "of", N * 4, flush=True)
