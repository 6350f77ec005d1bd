"""GPU parity of the version >= 7 model family (reference models/base_v2.py, models/ar_nar_v2.py) on the HIP path:
operators against torch, passes and generation loops against the reference's goldens and the CPU restatement."""
import numpy as np
import pytest
import torch

from conftest import load_golden, golden_inputs, model_cfg, synth_weights

pytestmark = pytest.mark.gpu


def hip_model(kind, dtype=torch.float32, **kw):
    from vall_e_amd.models import get_model
    m = get_model(model_cfg(kind), training=False)
    m.load_state_dict(synth_weights(kind, **kw))
    return m.to("cuda:0", dtype)


def oracle_v2(kind, dtype=torch.float32, **kw):
    import valle_oracle_v2 as vo2
    return vo2.OracleV2(model_cfg(kind), synth_weights(kind, **kw), dtype=dtype)


def full_v2_model():
    import model_configs as vcfg
    m = vcfg.repo_default_model()
    m.version = 7
    return m


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm()).item()


# ------------------------------------------------------------------------------ operators
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,d", [(37, 1024), (5, 256), (130, 768)])
def test_layernorm(gpu_device, dtype, rows, d):
    from vall_e_amd import _lib as L
    g = torch.Generator().manual_seed(rows + d)
    x = (torch.randn(rows, d, generator=g) * 2 + 0.3).to(dtype)
    w = (1 + 0.1 * torch.randn(d, generator=g)).to(dtype)
    b = (0.1 * torch.randn(d, generator=g)).to(dtype)
    want = torch.nn.functional.layer_norm(x, (d,), w, b, 1e-5)
    xd, wd, bd = x.to(gpu_device), w.to(gpu_device), b.to(gpu_device)
    out = torch.empty_like(xd)
    L.check(L.load().valle_layernorm(L.ptr(xd), L.ptr(wd), L.ptr(bd), L.ptr(out), rows, d, 1e-5, L.dtype_code(dtype), L.stream_ptr()))
    tol = 3e-6 if dtype == torch.float32 else 1.6e-2
    torch.testing.assert_close(out.cpu().float(), want.float(), rtol=tol, atol=tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(200, 512, 256), (70, 256, 512), (16, 2048, 1024), (1300, 1024, 2048), (2300, 2048, 1024)])
def test_gemm_gelu_and_resid_bias(gpu_device, dtype, shape):
    """The two epilogues of the encoder / decoder FFNs: gelu(A W^T + b) and resid + (A W^T + b)."""
    from vall_e_amd import _lib as L
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(M, K, generator=g).to(dtype)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dtype)
    bias = (0.3 * torch.randn(N, generator=g)).to(dtype)
    resid = torch.randn(M, N, generator=g).to(dtype)
    acc = A.double() @ W.double().T + bias.double()
    Ad, Wd, bd, Rd = A.to(gpu_device), W.to(gpu_device), bias.to(gpu_device), resid.to(gpu_device)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    for epi, want in ((L.EPI_GELU, torch.nn.functional.gelu(acc)), (L.EPI_RESID_BIAS, resid.double() + acc), (L.EPI_STORE, acc)):
        Cd = torch.zeros((M, N), dtype=dtype, device=gpu_device)
        L.check(L.load().valle_gemm(L.ptr(Ad), L.ptr(Wd), L.ptr(bd), L.ptr(Rd) if epi == L.EPI_RESID_BIAS else None, L.ptr(Cd),
                                    M, N, K, N, epi, L.dtype_code(dtype), L.stream_ptr()))
        assert rel_l2(Cd, want) < tol, epi


def test_cfg_combine_cols(gpu_device):
    from vall_e_amd import _lib as L
    g = torch.Generator().manual_seed(5)
    Lv, V, ld, lens = 8, 1025, 1032, [9, 1, 12, 40]
    total = sum(lens)
    pos = torch.randn(Lv, total, ld, generator=g)
    neg = torch.randn(Lv, total, ld, generator=g)
    off = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
    for strength, rescale in ((2.5, 0.75), (3.0, 0.0)):
        want = pos.clone()
        for i in range(len(lens)):
            p, q = pos[:, off[i]:off[i + 1], :V], neg[:, off[i]:off[i + 1], :V]
            s = q + (p - q) * strength
            if rescale > 0:
                s = s * (rescale * (p.std((1,), keepdim=True) / s.std((1,), keepdim=True)) + (1 - rescale))
            want[:, off[i]:off[i + 1], :V] = s
        pd, nd = pos.to(gpu_device), neg.to(gpu_device)
        od = torch.from_numpy(off).to(gpu_device)
        L.check(L.load().valle_cfg_combine_cols(L.ptr(pd), L.ptr(nd), ld, V, Lv, total, L.ptr(od), len(lens), strength, rescale, L.stream_ptr()))
        got = pd.cpu()
        keep = [i for i in range(len(lens)) if not (rescale > 0 and lens[i] == 1)]
        for i in keep:
            torch.testing.assert_close(got[:, off[i]:off[i + 1], :V], want[:, off[i]:off[i + 1], :V], rtol=2e-5, atol=2e-5)
        if rescale > 0:                                     # one row: the unbiased std is nan, in torch too
            assert torch.isnan(got[:, off[1]:off[2], :V]).all() and torch.isnan(want[:, off[1]:off[2], :V]).all()


# ------------------------------------------------------------------------------ passes
def test_tiny_v2_forward_vs_reference(gpu_device):
    """AudioEncoder embeddings, position ids and the (8, T, 1025) logits of an AR-style and a demasking-style pass against the
    reference's own outputs (golden tiny_v2_fwd)."""
    z = load_golden("tiny_v2_fwd")
    phns, proms, langs = golden_inputs(z)
    m = hip_model("tiny_v2")
    resps = [torch.from_numpy(z[f"ar_resp_{i}"]) for i in range(2)]
    plan = m.plan(phns, langs, proms, [r.shape[0] for r in resps])
    x, _ = m.build_x(plan, np.concatenate([r.numpy() for r in resps]))
    x = x.cpu()
    for i in range(2):
        a, b = int(plan.cu[i]), int(plan.cu[i + 1])
        torch.testing.assert_close(x[a:b], torch.from_numpy(z[f"ar_emb_{i}"]), rtol=1e-5, atol=1e-5)
        np.testing.assert_array_equal(plan.pos[a:b], z["ar_pos"][i][: b - a])
    inputs = m.inputs(task_list=["tts"] * 2, phns_list=phns, proms_list=proms, resps_list=resps, lang_list=langs, quant_levels=[0, 0])
    out = m.forward_super(inputs=inputs, quant_levels=[0, 0])
    for i in range(2):
        torch.testing.assert_close(out.logits[i][:, -3:].cpu(), torch.from_numpy(z[f"ar_logits_{i}"]), rtol=1e-4, atol=1e-4)
    masked = [torch.from_numpy(z[f"nar_resp_{i}"]) for i in range(2)]
    t = torch.tensor(0.3)
    inputs = m.inputs(task_list=["tts"] * 2, phns_list=phns, proms_list=proms, resps_list=masked, lang_list=langs, time_list=[t, t], quant_levels=[0, 0])
    out = m.forward_super(inputs=inputs, quant_levels=[0, 0])
    for i in range(2):
        n = masked[i].shape[0]
        torch.testing.assert_close(out.logits[i][:, -n:][:, ::3].cpu(), torch.from_numpy(z[f"nar_logits_{i}"]), rtol=1e-4, atol=1e-4)


def test_tiny_v2_ar_vs_reference(gpu_device):
    """Greedy AR of 8 codes per step == the reference's codes (golden tiny_v2_ar), on the KV cache and on the reference's own
    schedule (the whole sequence per step), whatever the step chunking; the logits of every step agree between the two
    schedules far inside the golden's smallest top-2 margin."""
    z = load_golden("tiny_v2_ar")
    phns, proms, langs = golden_inputs(z)
    m = hip_model("tiny_v2")
    kept = {}
    for cache in (True, False):
        for chunk in (16, 1, 4):
            lg = []
            out = m(task_list=["tts"] * 2, phns_list=phns, proms_list=proms, lang_list=langs, max_duration=int(z["steps"]), temperature=0.0,
                    step_chunk=chunk, kv_cache=cache, return_logits=lg)
            for b in range(2):
                np.testing.assert_array_equal(out[b].cpu().numpy(), z[f"ar_codes_{b}"])
            kept[cache] = torch.stack(lg)
    err = (kept[True] - kept[False]).abs().max().item()
    assert err * 10 < float(z["min_top2_gap"]), (err, float(z["min_top2_gap"]))


def test_v2_ar_unified_positions_cache_equals_recompute(gpu_device):
    """unified_position_ids (the dataclass default): the cached step's position is rows-in-front + frame; 10 greedy steps on a
    ragged batch of 3 give the same codes on the cache as on the whole-sequence schedule."""
    import model_configs as vcfg
    from vall_e_amd.models import get_model
    from vall_e_amd.weights import synth_state_dict
    from vall_e_amd.synth import synth_batch
    cfg = vcfg.tiny_v2_model()
    cfg.experimental.unified_position_ids = True
    m = get_model(cfg, training=False)
    m.load_state_dict(synth_state_dict(cfg, 1234))
    m.to("cuda:0", torch.float32)
    phns, proms, langs = synth_batch(5, 3, n_phn=12, n_prom=20, ragged=True)
    a = m(task_list=["tts"] * 3, phns_list=phns, proms_list=proms, lang_list=langs, max_duration=10, temperature=0.0, kv_cache=True)
    b = m(task_list=["tts"] * 3, phns_list=phns, proms_list=proms, lang_list=langs, max_duration=10, temperature=0.0, kv_cache=False, step_chunk=3)
    for x, y in zip(a, b):
        assert x.shape == (10, 8) and torch.equal(x, y)


@pytest.mark.parametrize("name,kind", [("tiny_v2_narlen_noties", "tiny_v2"), ("tiny_v2_nc_narlen_noties", "tiny_v2_nc"),
                                       ("tiny_v2_narlen_masked_only", "tiny_v2")])
def test_tiny_v2_narlen_vs_reference(gpu_device, name, kind):
    """NAR-len demasking of all levels with the guidance over (levels, rows, vocab), causal and with noncausal_masks, and with
    sampling_scores_masked_only == the reference's codes.  These goldens have no masking cut on tied scores (``tied_cuts`` = 0,
    counted while the reference ran), so the result does not hang on torch.topk's order among ties."""
    z = load_golden(name)
    assert int(z["tied_cuts"]) == 0
    phns, proms, langs = golden_inputs(z)
    kw = {k[3:]: (bool(z[k]) if "scores" in k else float(z[k])) for k in z if k.startswith("kw_")}
    m = hip_model(kind)
    out = m(task_list=["tts"] * 2, phns_list=phns, proms_list=proms, lang_list=langs, len_list=[int(x) for x in z["lens"]],
            max_steps=int(z["max_steps"]), temperature=0.0, **kw)
    for b in range(2):
        assert out[b].dtype == torch.int16
        np.testing.assert_array_equal(out[b].cpu().numpy(), z[f"codes_{b}"])


@pytest.mark.parametrize("name,kind", [("tiny_v2_narlen", "tiny_v2"), ("tiny_v2_nc_narlen", "tiny_v2_nc")])
def test_tiny_v2_narlen_default_filter_vs_oracle(gpu_device, name, kind):
    """Under the default top-n-sigma filter the tiny model's rows keep one candidate in places: probability exactly 1.0, tied
    scores, and 7-10 masking cuts of these goldens fall on ties -- where the reference masks whichever position torch.topk's
    partial sort returns first (parity with the reference UNPINNED there).  valle_mask_lowest takes the lower index; against
    the CPU restatement under that same rule the codes are identical (the restatement under torch.topk's order equals the
    reference's goldens, tests/test_oracle_vs_golden.py)."""
    z = load_golden(name)
    assert int(z["tied_cuts"]) > 0
    phns, proms, langs = golden_inputs(z)
    lens = [int(x) for x in z["lens"]]
    m = hip_model(kind)
    out = m(task_list=["tts"] * 2, phns_list=phns, proms_list=proms, lang_list=langs, len_list=lens, max_steps=int(z["max_steps"]), temperature=0.0)
    ref = oracle_v2(kind).nar_masked_v2(phns, proms, langs, lens, max_steps=int(z["max_steps"]), temperature=0.0, tie_rule="lowest_index")
    for b in range(2):
        np.testing.assert_array_equal(out[b].cpu().numpy(), ref[b].numpy())


def test_v2_stop_token_ends_ar(gpu_device):
    """A stop bias on every level's head: generation ends at the first step, the result is empty (cut at the first stop row)."""
    z = load_golden("tiny_v2_ar")
    phns, proms, langs = golden_inputs(z)
    m = hip_model("tiny_v2", stop_bias=50.0)
    out = m(task_list=["tts"] * 2, phns_list=phns, proms_list=proms, lang_list=langs, max_duration=40, temperature=0.0, step_chunk=4)
    assert all(o.shape == (0, 8) for o in out)


def test_v2_sampling_is_seeded_and_scores_are_unscaled(gpu_device):
    """temperature > 0: the draw is reproducible under a seed, differs across seeds, and the score is the softmax of the
    processed logits BEFORE the temperature at the drawn token (base_v2.py:1398-1410)."""
    m = hip_model("tiny_v2")
    m._prepare()
    g = torch.Generator().manual_seed(11)
    lg = [torch.randn(8, 6, 1025, generator=g) * 3 for _ in range(2)]
    prev = [torch.zeros(6, 8, dtype=torch.long)] * 2
    a = m.sample(lg, prev_list=prev, temperature=0.8, top_no=0.0, seed=3)
    b = m.sample(lg, prev_list=prev, temperature=0.8, top_no=0.0, seed=3)
    c = m.sample(lg, prev_list=prev, temperature=0.8, top_no=0.0, seed=4)
    for i in range(2):
        assert torch.equal(a.ids[i], b.ids[i]) and a.ids[i].shape == (8, 6)
        want = torch.softmax(lg[i], dim=-1).gather(-1, a.ids[i].cpu().unsqueeze(-1)).squeeze(-1)
        torch.testing.assert_close(a.scores[i].cpu(), want, rtol=1e-4, atol=1e-6)
    assert any(not torch.equal(a.ids[i], c.ids[i]) for i in range(2))
    # the causal form (AR): the last row only, no scores
    d = m.sample([l[0] for l in lg], temperature=0.0)
    assert d.scores == [] and all(t.shape == (1,) for t in d.ids)
    assert int(d.ids[0][0]) == int(torch.where(lg[0][0, -1] >= lg[0][0, -1].max() - lg[0][0, -1].std(), lg[0][0, -1], -torch.inf).argmax())


# ------------------------------------------------------------------------------ at the reference's full width, bf16
def test_v2_full_bf16_vs_oracle(gpu_device):
    """d_model 1,024 / 12 layers in bf16: the logits of one demasking-style pass of 2 ragged utterances against the bf16 CPU
    restatement (rel-L2 <= 4e-2, as for the version-5 passes), and 3 greedy AR steps of 8 codes in fp32 == the restatement."""
    import model_configs as vcfg
    import valle_oracle_v2 as vo2
    from vall_e_amd.models import get_model
    from vall_e_amd.weights import synth_state_dict
    from vall_e_amd.synth import synth_batch
    cfg = full_v2_model()
    sd = synth_state_dict(cfg, 1234)
    phns, proms, langs = synth_batch(77, 2, n_phn=24, n_prom=70, ragged=True)
    g = torch.Generator().manual_seed(1)
    resps = [torch.randint(0, 1026, (n, 8), generator=g) for n in (90, 61)]
    m = get_model(cfg, training=False)
    m.load_state_dict(sd)
    m.to("cuda:0", torch.bfloat16)
    t = torch.tensor(0.5)
    inputs = m.inputs(task_list=["tts"] * 2, phns_list=phns, proms_list=proms, resps_list=resps, lang_list=langs, time_list=[t, t], quant_levels=[0, 0])
    got = m.forward_super(inputs=inputs, quant_levels=[0, 0]).logits
    want = vo2.OracleV2(cfg, sd, dtype=torch.bfloat16).forward_v2(phns, langs, proms, resps, timestep=True)
    for i in range(2):
        n = resps[i].shape[0]
        assert rel_l2(got[i][:, -n:], want[i][:, -n:].float()) < 4e-2
    # bf16 AR on the fused decode kernels (KV cache) against the whole-sequence schedule: the logits of 4 steps within the bf16 bar
    la, lb = [], []
    m(task_list=["tts"] * 2, phns_list=phns, proms_list=proms, lang_list=langs, max_duration=4, temperature=0.0, kv_cache=True, return_logits=la)
    m(task_list=["tts"] * 2, phns_list=phns, proms_list=proms, lang_list=langs, max_duration=4, temperature=0.0, kv_cache=False, return_logits=lb)
    assert rel_l2(la[0], lb[0]) < 4e-2
    m.to("cuda:0", torch.float32)
    ref = vo2.OracleV2(cfg, sd).ar_v2(phns, proms, langs, max_duration=3, temperature=0.0)
    for cache in (True, False):
        out = m(task_list=["tts"] * 2, phns_list=phns, proms_list=proms, lang_list=langs, max_duration=3, temperature=0.0, kv_cache=cache)
        for b in range(2):
            np.testing.assert_array_equal(out[b].cpu().numpy(), ref[b].numpy())


def test_tiny_v2_len_vs_reference(gpu_device):
    """Duration prediction with the 5-digit head (ar_nar_v2.py:374-431) == the reference's frames on a ragged batch -- where the
    reference reads the shorter utterance's LAST PADDING ROW (base_v2.py:1298-1300 on the right-padded batch): reproduced by
    one zero row that is a query and never a key (valle_forward_layers(kv_lens=)); alone, each utterance gives what the
    restatement gives alone.  With the dataclass default head (one float) the reference's pass dies on a reshape: same here."""
    z = load_golden("tiny_v2_len")
    phns, proms, langs = golden_inputs(z)
    m = hip_model("tiny_v2_len")
    got = m(task_list=["len"] * 2, phns_list=phns, proms_list=proms, lang_list=langs)
    assert got == [int(x) for x in z["frames"]]
    o = oracle_v2("tiny_v2_len")
    for i in range(2):
        solo = m(task_list=["len"], phns_list=phns[i:i + 1], proms_list=proms[i:i + 1], lang_list=langs[i:i + 1])
        assert solo == o.len_v2(phns[i:i + 1], proms[i:i + 1], langs[i:i + 1])
    with pytest.raises(RuntimeError, match="is invalid for input of size 27"):
        hip_model("tiny_v2")(task_list=["len"] * 2, phns_list=phns, proms_list=proms, lang_list=langs)


def test_v2_tts_batched_inference_narlen(gpu_device):
    """TTS.batched_inference(modality="nar-len") on a version-7 model (inference.py:326-336: the duration head, then the
    demasking of all levels) == the restatement fed the same calls; the "ar+nar" modality's second call hands (frames, 8)
    codes back to the model, which the reference takes for a training batch (ar_nar_v2.py:625-629): refused by name."""
    from vall_e_amd.inference import TTS
    z = load_golden("tiny_v2_len")
    phns, proms, langs = golden_inputs(z)
    tts = TTS(config=model_cfg("tiny_v2_len"), device="cuda:0", dtype=torch.float32, state_dict=synth_weights("tiny_v2_len"))
    kw = dict(max_steps=3, top_no=0.0, max_duration=14, temperature=0.0)
    out = tts.batched_inference(phns, references=proms, languages=langs, modality="nar-len", seed=5, **kw)
    o = oracle_v2("tiny_v2_len")
    lens = [int(l * 1.05) for l in o.len_v2(phns, proms, langs)]
    ref = o.nar_masked_v2(phns, proms, langs, lens, tie_rule="lowest_index", **kw)
    for b in range(2):
        assert out[b].shape == (14, 8)
        np.testing.assert_array_equal(out[b].cpu().numpy(), ref[b].numpy())
    with pytest.raises(NotImplementedError, match="training"):
        tts.batched_inference(phns, references=proms, languages=langs, modality="ar+nar", max_duration=3, temperature=0.0)


@pytest.mark.parametrize("shape", [(16, 2048, 1024), (16, 1025, 1024), (128, 1024, 2048), (5, 48, 128), (200, 512, 256), (33, 1000, 640)])
def test_gemm_skinny(gpu_device, shape):
    """The few-rows GEMM (valle_gemm_skinny) against fp64, every epilogue it builds, ragged M and N."""
    from vall_e_amd import _lib as L
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(M, K, generator=g).bfloat16()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16()
    bias = (0.3 * torch.randn(N, generator=g))
    resid = torch.randn(M, N, generator=g).bfloat16()
    acc = A.double() @ W.double().T
    Ad, Wd, Rd = A.to(gpu_device), W.to(gpu_device), resid.to(gpu_device)
    b16, b32 = bias.bfloat16().to(gpu_device), bias.to(gpu_device)
    accb = acc + bias.bfloat16().double()
    cases = ((L.EPI_STORE, accb, b16, torch.bfloat16), (L.EPI_GELU, torch.nn.functional.gelu(accb), b16, torch.bfloat16),
             (L.EPI_RESID_BIAS, resid.double() + accb, b16, torch.bfloat16), (L.EPI_RESID, resid.double() + acc, None, torch.bfloat16),
             (L.EPI_F32OUT, acc + bias.double(), b32, torch.float32))
    ldc = (N + 7) // 8 * 8
    for epi, want, bd, odt in cases:
        Cd = torch.zeros((M, ldc), dtype=odt, device=gpu_device)
        Rp = None
        if epi in (L.EPI_RESID, L.EPI_RESID_BIAS):
            Rp = torch.zeros((M, ldc), dtype=torch.bfloat16, device=gpu_device)
            Rp[:, :N] = Rd
        L.check(L.load().valle_gemm_skinny(L.ptr(Ad), L.ptr(Wd), L.ptr(bd), L.ptr(Rp), L.ptr(Cd), M, N, K, ldc, epi, L.BF16, L.stream_ptr()))
        assert rel_l2(Cd[:, :N], want) < (2e-2 if odt == torch.bfloat16 else 2e-5), (epi, shape)
        assert float(Cd[:, N:].abs().max() if ldc > N else 0) == 0.0          # nothing written past N


@pytest.mark.parametrize("shared_a", [False, True])
def test_gemm_skinny_grouped_equals_per_group_calls(gpu_device, shared_a):
    """valle_gemm_skinny_grouped (G products of one shape in one launch: the AudioDecoder's per-level linears) gives, group
    by group, the bits of valle_gemm_skinny; every epilogue, a shared (stride 0) or per-group A, ragged N."""
    from vall_e_amd import _lib as L
    lib = L.load()
    G, M, N, K = 8, 16, 1025, 1024
    ldc = (N + 7) // 8 * 8
    g = torch.Generator().manual_seed(5)
    A = torch.randn(1 if shared_a else G, M, K, generator=g).bfloat16().to(gpu_device)
    W = (torch.randn(G, N, K, generator=g) / K ** 0.5).bfloat16().to(gpu_device)
    bias = 0.3 * torch.randn(G, N, generator=g)
    R = torch.randn(G, M, ldc, generator=g).bfloat16().to(gpu_device)
    a_gs = 0 if shared_a else M * K
    for epi, odt in ((L.EPI_STORE, torch.bfloat16), (L.EPI_GELU, torch.bfloat16), (L.EPI_RESID_BIAS, torch.bfloat16),
                     (L.EPI_RESID, torch.bfloat16), (L.EPI_F32OUT, torch.float32)):
        bd = (bias if odt == torch.float32 else bias.bfloat16()).contiguous().to(gpu_device)
        use_r = epi in (L.EPI_RESID, L.EPI_RESID_BIAS)
        bp = None if epi == L.EPI_RESID else bd
        Cg = torch.zeros((G, M, ldc), dtype=odt, device=gpu_device)
        L.check(lib.valle_gemm_skinny_grouped(L.ptr(A), a_gs, L.ptr(W), N * K, L.ptr(bp), N, L.ptr(R) if use_r else None, M * ldc,
                                              L.ptr(Cg), M * ldc, G, M, N, K, ldc, epi, L.BF16, L.stream_ptr()))
        for i in range(G):
            C1 = torch.zeros((M, ldc), dtype=odt, device=gpu_device)
            L.check(lib.valle_gemm_skinny(L.ptr(A[0 if shared_a else i]), L.ptr(W[i]), L.ptr(None if bp is None else bp[i]),
                                          L.ptr(R[i]) if use_r else None, L.ptr(C1), M, N, K, ldc, epi, L.BF16, L.stream_ptr()))
            assert torch.equal(Cg[i], C1), (epi, i)
    # group strides that would break the 16-byte fragment loads are refused
    with pytest.raises(L.ValleHipError):
        L.check(lib.valle_gemm_skinny_grouped(L.ptr(A), 4, L.ptr(W), N * K, None, 0, None, 0, L.ptr(Cg), M * ldc, G, M, N, K, ldc,
                                              L.EPI_STORE, L.BF16, L.stream_ptr()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_layernorm_grouped_equals_per_group_calls(gpu_device, dtype):
    from vall_e_amd import _lib as L
    lib = L.load()
    G, rows, d = 8, 19, 512
    g = torch.Generator().manual_seed(6)
    x = torch.randn(rows, d, generator=g).to(dtype).to(gpu_device)
    w = (1 + 0.2 * torch.randn(G, d, generator=g)).to(dtype).to(gpu_device)
    b = (0.2 * torch.randn(G, d, generator=g)).to(dtype).to(gpu_device)
    out = torch.zeros((G, rows, d), dtype=dtype, device=gpu_device)
    L.check(lib.valle_layernorm_grouped(L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(out), rows, d, G, 1e-5, L.dtype_code(dtype), L.stream_ptr()))
    for i in range(G):
        o1 = torch.zeros((rows, d), dtype=dtype, device=gpu_device)
        L.check(lib.valle_layernorm(L.ptr(x), L.ptr(w[i]), L.ptr(b[i]), L.ptr(o1), rows, d, 1e-5, L.dtype_code(dtype), L.stream_ptr()))
        assert torch.equal(out[i], o1), i
