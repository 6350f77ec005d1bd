import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "oracle"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: full-size CPU cases (tens of seconds)")
    torch.set_num_threads(min(8, os.cpu_count() or 1))


def load_golden(name):
    return dict(np.load(GOLDEN / f"{name}.npz"))


def golden_inputs(z):
    """(phns, proms, langs) lists with the dtypes TTS sends (uint8 / int16 / uint8)."""
    B = int(z["batch"]) if "batch" in z else 1
    phns = [torch.from_numpy(z[f"phn_{i}"]) for i in range(B)]
    proms = [torch.from_numpy(z[f"prom_{i}"]) for i in range(B)]
    langs = [torch.from_numpy(z[f"lang_{i}"]) for i in range(B)]
    return phns, proms, langs


def qnt_inputs():
    z = load_golden("qnt_enc_prompt")
    prom = torch.from_numpy(z["codes"][0].T.astype(np.int16)).contiguous()
    phn = torch.tensor(z["phoneme_ids"], dtype=torch.uint8)
    return [phn], [prom], [torch.tensor([0], dtype=torch.uint8)]


_cache = {}


def model_cfg(kind):
    import model_configs as vcfg
    return {"full": vcfg.repo_default_model, "tiny": vcfg.tiny_model, "tiny_len": vcfg.tiny_len_model,
            "tiny_nc": vcfg.tiny_nc_model, "tiny_tone": vcfg.tiny_tone_model, "tiny_upi": vcfg.tiny_upi_model, "tiny_nobias": vcfg.tiny_nobias_model, "tiny_nc_upi": vcfg.tiny_nc_upi_model, "tiny_nosum": vcfg.tiny_nosum_model, "tiny_sumprom": vcfg.tiny_sumprom_model,
            "half2": lambda: vcfg.tiny_model(dim=512, heads=8), "s075_2": lambda: vcfg.tiny_model(dim=768, heads=12),
            "tiny_v2": vcfg.tiny_v2_model, "tiny_v2_nc": vcfg.tiny_v2_nc_model, "tiny_v2_len": vcfg.tiny_v2_len_model}[kind]()


def synth_weights(kind, **kw):
    from vall_e_amd.weights import synth_state_dict
    key = (kind, tuple(sorted(kw.items())))
    if key not in _cache:
        if len(_cache) > 2:
            _cache.clear()
        kw = dict(kw)
        _cache[key] = synth_state_dict(model_cfg(kind), kw.pop("weight_seed", 1234), **kw)
    return _cache[key]


@pytest.fixture(scope="session")
def gpu_device():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
