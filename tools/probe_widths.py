"""Dev probe: one short bf16 AR run at widths 512 and 768 (kernel trace shows which decode kernels ran)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vall_e_amd import config as vcfg
from vall_e_amd.models import get_model
from vall_e_amd.weights import synth_state_dict
from vall_e_amd.synth import synth_batch
for dim, heads in ((512, 8), (768, 12)):
    cfg = vcfg.tiny_model(dim=dim, heads=heads)
    m = get_model(cfg, training=False)
    m.load_state_dict(synth_state_dict(cfg, 1234))
    m = m.to("cuda:0", torch.bfloat16)
    phns, proms, langs = synth_batch(5, 4, n_phn=10, n_prom=12, ragged=True)
    out = m(task_list=["tts"] * 4, phns_list=phns, proms_list=proms, lang_list=langs, max_duration=6, temperature=0.0)
    print(dim, [tuple(o.shape) for o in out])
