"""dev: where a fuzz case of the version >= 7 NAR-len loop leaves the restatement (first differing step, what differs)."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "oracle"), str(ROOT / "tests")]
from test_fuzz_vs_oracle import _models_v2
from vall_e_amd.synth import synth_batch
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 72
rng = np.random.default_rng(9000 + seed)
kind = ["tiny_v2", "tiny_v2_nc"][int(rng.integers(2))]
upi = bool(rng.random() < 0.35)
m, o = _models_v2(kind, upi)
B = int(rng.integers(1, 5))
phns, proms, langs = synth_batch(int(rng.integers(1 << 20)), B, n_phn=int(rng.integers(4, 20)), n_prom=int(rng.integers(3, 30)), ragged=bool(rng.random() < 0.7))
steps = int(rng.integers(1, 9)); rng.choice([1, 3, 16]); rng.random(); rng.choice([0.0, 1.0])
lens = [int(rng.integers(3, 24)) for _ in range(B)]
kw = dict(max_steps=int(rng.integers(2, 8)), top_no=float(rng.choice([0.0, 1.0])))
if rng.random() < 0.4: kw["cfg_strength"] = float(rng.choice([1.0, 3.5]))
if rng.random() < 0.3: kw["minimum_cfg_strength"] = float(rng.choice([0.0, 1.5]))
if rng.random() < 0.3: kw["sampling_scores_masked_only"] = True
if rng.random() < 0.3: kw["sampling_scores_remask"] = True
if rng.random() < 0.3: kw["annealed_sampling"] = False
if max(kw.get("cfg_strength", 2.5), kw.get("minimum_cfg_strength", 2.5)) > 0:
    lens = [max(l, 6) for l in lens]
print(kind, upi, B, lens, kw)
th, to = [], []
got = m(task_list=["tts"] * B, phns_list=phns, proms_list=proms, lang_list=langs, len_list=lens, temperature=0.0, trace=th, **kw)
want = o.nar_masked_v2(phns, proms, langs, lens, temperature=0.0, tie_rule="lowest_index", trace=to, **kw)
off = np.concatenate(([0], np.cumsum(lens))); total = off[-1]
for s in range(len(th)):
    hid, hsc = th[s][0].cpu().view(8, total), th[s][1].cpu().view(8, total)
    for b in range(B):
        oi, osc, olg = to[s][0][b], to[s][1][b], to[s][2][b]
        a, c = hid[:, off[b]:off[b + 1]], hsc[:, off[b]:off[b + 1]]
        nd = int((a != oi).sum()); ds = float((c - osc).abs().max())
        if nd or ds > 1e-4:
            w = (c - osc).abs().argmax(); l, r = int(w // lens[b]), int(w % lens[b])
            row = olg[l, r]
            thr = row.max() - row.std()
            near = (row - thr).abs().min().item()
            print(f"step {s} utt {b}: {nd} ids differ, max score diff {ds:.3e} at level {l} row {r}: hip {c[l, r]:.6f} oracle {osc[l, r]:.6f}; "
                  f"nearest logit to the top-n-sigma threshold {near:.2e}; survivors {(row >= thr).sum().item()}")
    if any(int((hid[:, off[b]:off[b + 1]] != to[s][0][b]).sum()) for b in range(B)) or any(float((hsc[:, off[b]:off[b + 1]] - to[s][1][b]).abs().max()) > 1e-4 for b in range(B)):
        break
