"""valle_ar_decode_noalloc (include/valle_hip.h): the decode loop on the library's static scratch array, for
callers that cannot allocate.  Bar: bit-identical codes and logits to valle_ar_decode (same kernels, only the
workspace differs), on two calls in a row -- the second with fewer utterances, so stale bytes of the first
call sit in the shared array -- in bf16 (fused decode kernels) and fp32 (generic kernels)."""
import ctypes as C

import pytest
import torch

from test_hip_parity import hip_model
from vall_e_amd import _lib as L
from vall_e_amd.synth import synth_batch

pytestmark = pytest.mark.gpu


def _run(m, B, seed, noalloc):
    phns, proms, langs = synth_batch(seed, B, n_phn=10, n_prom=12, ragged=True)
    codes, logits = m(task_list=["tts"] * B, phns_list=phns, proms_list=proms, lang_list=langs, max_duration=9,
                      temperature=0.0, return_logits=True, step_chunk=4, noalloc=noalloc)
    return codes, torch.stack([l.float() for l in logits])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_noalloc_decode_equals_workspace_decode(gpu_device, dtype):
    m = hip_model("tiny", dtype)
    for B, seed in ((5, 301), (2, 302)):        # second call: smaller batch over the first call's leftovers
        want_c, want_l = _run(m, B, seed, noalloc=False)
        got_c, got_l = _run(m, B, seed, noalloc=True)
        for a, b in zip(got_c, want_c):
            assert torch.equal(a, b)
        assert torch.equal(got_l, want_l)


def test_noalloc_refuses_oversized_batches(gpu_device):
    m = hip_model("tiny", torch.float32)
    m._prepare()
    lib = L.load()
    assert lib.valle_noalloc_scratch_bytes() == 2 << 20
    B = 400
    assert lib.valle_decode_workspace_bytes(m._handle, B) > lib.valle_noalloc_scratch_bytes()
    st = L.ARState()
    st.B = B
    sp = L.SampleParams()
    rc = lib.valle_ar_decode_noalloc(m._handle, C.byref(st), C.byref(sp), 1, None, 0, None)
    assert rc == -4, rc                                  # VALLE_E_WORKSPACE, before anything is enqueued
    assert b"scratch" in lib.valle_last_error()
