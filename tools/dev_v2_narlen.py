"""dev: first demasking step of the version >= 7 NAR-len loop, HIP against the CPU restatement, stage by stage."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "oracle"), str(ROOT / "tests")]
from conftest import load_golden, golden_inputs, model_cfg, synth_weights
import valle_oracle_v2 as vo2
from vall_e_amd.models import get_model

z = load_golden("tiny_v2_narlen")
phns, proms, langs = golden_inputs(z)
lens = [int(x) for x in z["lens"]]
m = get_model(model_cfg("tiny_v2"), training=False); m.load_state_dict(synth_weights("tiny_v2")); m.to("cuda:0", torch.float32)
o = vo2.OracleV2(model_cfg("tiny_v2"), synth_weights("tiny_v2"))
resps = [torch.full((n, 8), o.mask_token, dtype=torch.int64) for n in lens]
t = torch.tensor(0.0)
inp = m.inputs(task_list=["tts"] * 2, phns_list=phns, proms_list=proms, resps_list=resps, lang_list=langs, time_list=[t, t], quant_levels=[0, 0])
got = m.forward_super(inputs=inp).logits
want = o.forward_v2(phns, langs, proms, resps, timestep=True)
for i in range(2):
    print("cond", i, (got[i].cpu() - want[i]).abs().max().item())
null_phn = [torch.tensor([1, 2]) for _ in range(2)]
inp = m.inputs(task_list=["tts"] * 2, phns_list=null_phn, proms_list=None, resps_list=resps, lang_list=langs, time_list=[t, t], quant_levels=[0, 0])
gotn = m.forward_super(inputs=inp).logits
wantn = o.forward_v2(null_phn, langs, None, resps, timestep=True)
for i in range(2):
    print("null", i, (gotn[i].cpu() - wantn[i]).abs().max().item())
trace = []
out = m(task_list=["tts"] * 2, phns_list=phns, proms_list=proms, lang_list=langs, len_list=lens, max_steps=1, temperature=0.0, trace=trace)
ref = o.nar_masked_v2(phns, proms, langs, lens, max_steps=1, temperature=0.0)
for b in range(2):
    print("1 step", b, (out[b].cpu().numpy() != ref[b].numpy()).sum(), "of", ref[b].numel())
for steps in (2, 3, 4):
    out = m(task_list=["tts"] * 2, phns_list=phns, proms_list=proms, lang_list=langs, len_list=lens, max_steps=steps, temperature=0.0)
    ref = o.nar_masked_v2(phns, proms, langs, lens, max_steps=steps, temperature=0.0)
    for b in range(2):
        print(steps, "steps", b, (out[b].cpu().numpy() != ref[b].numpy()).sum(), "of", ref[b].numel())
# step-0 scores
lg = o.forward_v2(phns, langs, proms, resps, timestep=True)
nl = o.forward_v2(null_phn, langs, None, resps, timestep=True)
lg = o.cfg_v2(lg, nl, 2.5, lens, 0.75)
ids, sc = o.sample_v2(lg, lens, temperature=0.0, top_no=1.0)
hid, hsc = trace[0]
off = np.concatenate(([0], np.cumsum(lens)))
total = off[-1]
hsc = hsc.cpu().view(8, total); hid = hid.cpu().view(8, total)
for b in range(2):
    a = hsc[:, off[b]:off[b + 1]]
    print("scores", b, (a - sc[b]).abs().max().item(), "ids eq", bool((hid[:, off[b]:off[b+1]] == ids[b]).all()))
    print(" argmin per level hip", a.argmin(dim=1).tolist(), "oracle", sc[b].argmin(dim=1).tolist())
    print(" hip", a[0].tolist()); print(" ora", sc[b][0].tolist())
# step 1 by hand
r1 = [torch.stack([ids[b][l] for l in range(8)], dim=-1) for b in range(2)]            # (n, 8) codes after step 0
scores = [sc[b].t() for b in range(2)]
idx = [s.topk(1, dim=0, largest=False).indices for s in scores]
r1m = [torch.stack([r[:, l].scatter(0, i.t()[l], o.mask_token) for l in range(8)], dim=-1) for r, i in zip(r1, idx)]
t1 = torch.tensor(1.0)
inp = m.inputs(task_list=["tts"] * 2, phns_list=phns, proms_list=proms, resps_list=r1m, lang_list=langs, time_list=[t1, t1], quant_levels=[0, 0])
g1 = m.forward_super(inputs=inp).logits
w1 = o.forward_v2(phns, langs, proms, r1m, timestep=True)
for i in range(2):
    print("step1 cond", i, (g1[i].cpu() - w1[i]).abs().max().item())
out2 = m(task_list=["tts"] * 2, phns_list=phns, proms_list=proms, lang_list=langs, len_list=lens, max_steps=2, temperature=0.0, trace=(tr2 := []))
ref2 = o.nar_masked_v2(phns, proms, langs, lens, max_steps=2, temperature=0.0)
print("diff positions utt0 (frame, level):", np.argwhere(out2[0].cpu().numpy() != ref2[0].numpy()).tolist())
print("masked by oracle (level -> frame):", [int(idx[0][0, l]) for l in range(8)])
print("hip out", out2[0].cpu().numpy()[:, 0].tolist()); print("ref out", ref2[0].numpy()[:, 0].tolist()); print("after step 0", r1[0][:, 0].tolist())
