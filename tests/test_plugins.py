"""Operator-level drop-ins on the reference's own plug-in signatures (vall_e_amd/plugins.py):
the attention backend callable ``fn(q, k, v, causal=, softmax_scale=, dropout_p=)`` on (B, H, T, Dh) tensors
(vall_e/models/arch/llama.py:396-403) and the ``replace_linear`` module class (vall_e/utils/utils.py:364-462)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref_attn(q, k, v, causal, scale, seqlens=None):
    """fp64 softmax(q k^T scale + mask) v on (B, H, T, Dh)."""
    q, k, v = q.double(), k.double(), v.double()
    B, H, T, _ = q.shape
    s = q @ k.transpose(-1, -2) * scale
    mask = torch.zeros(B, 1, T, T, dtype=torch.bool, device=q.device)
    if causal:
        mask |= torch.ones(T, T, dtype=torch.bool, device=q.device).triu(1)
    if seqlens is not None:
        mask |= (torch.arange(T, device=q.device)[None, :] >= seqlens[:, None].to(q.device))[:, None, None, :]
    p = torch.softmax(s.masked_fill(mask, float("-inf")), dim=-1)
    return p @ v


@pytest.mark.parametrize("causal", [True, False])
@pytest.mark.parametrize("T", [1, 61, 200, 333])
def test_attention_backend_signature_hnd(gpu_device, causal, T):
    from vall_e_amd.plugins import valle_hip_attn_func
    g = torch.Generator().manual_seed(T)
    B, H, Dh = 3, 4, 64
    q, k, v = (torch.randn(B, H, T, Dh, generator=g).to(gpu_device, torch.bfloat16) for _ in range(3))
    scale = 1.0 / math.sqrt(Dh)
    out = valle_hip_attn_func(q, k, v, causal=causal, softmax_scale=scale, dropout_p=0.0)      # the reference's call, llama.py:396-403
    assert out.shape == q.shape and out.dtype == torch.bfloat16
    want = _ref_attn(q, k, v, causal, scale)
    err = (out.double() - want).abs().max().item()
    assert err < 2e-2, err                                    # bf16 output of O(1) values: half an ulp is 4e-3; P is rounded to bf16 too


def test_attention_strided_views_seqlens_and_scale(gpu_device):
    """q, k, v as views into one fused (B, T, 3, H, Dh) projection (what `view(...).transpose(1, 2)` of the reference's
    projections gives, llama.py:311-316), right-padded utterances masked through seqlens, a non-default scale, and the
    (B, T, H, Dh) layout of the flash_attn branch."""
    from vall_e_amd.plugins import valle_hip_attn_func
    g = torch.Generator().manual_seed(9)
    B, H, T, Dh = 4, 4, 150, 64
    qkv = torch.randn(B, T, 3, H, Dh, generator=g).to(gpu_device, torch.bfloat16)
    q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))            # (B, H, T, Dh), strides (3TH*Dh, Dh, 3H*Dh, 1)
    assert not q.is_contiguous()
    seqlens = torch.tensor([150, 17, 64, 129])
    out = valle_hip_attn_func(q, k, v, causal=True, softmax_scale=0.2, seqlens=seqlens)
    want = _ref_attn(q, k, v, True, 0.2, seqlens)
    for b in range(B):
        n = int(seqlens[b])
        assert (out[b, :, :n].double() - want[b, :, :n]).abs().max().item() < 2e-2
        assert (out[b, :, n:] == 0).all()
    nhd = valle_hip_attn_func(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], causal=True, softmax_scale=0.2, seqlens=seqlens, layout="NHD")
    assert nhd.shape == (B, T, H, Dh) and torch.equal(nhd.transpose(1, 2), out)
    with pytest.raises(NotImplementedError):
        valle_hip_attn_func(q, k, v, dropout_p=0.1)


def test_attention_backend_equals_the_models_own_kernel(gpu_device):
    """Same kernel, two address forms: packed qkv through valle_attn_prefill == strided tensors through valle_attn_prefill_hnd."""
    from vall_e_amd import _lib as L
    from vall_e_amd.plugins import valle_hip_attn_func
    g = torch.Generator().manual_seed(2)
    H, Dh, lens = 4, 64, [70, 131]
    M, d = sum(lens), H * Dh
    qkv = torch.randn(M, 3 * d, generator=g).to(gpu_device, torch.bfloat16)
    cu = torch.tensor([0, 70, 201], dtype=torch.int32, device=gpu_device)
    packed = torch.empty(M, d, dtype=torch.bfloat16, device=gpu_device)
    L.check(L.load().valle_attn_prefill(L.ptr(qkv), L.ptr(packed), L.ptr(cu), 2, M, max(lens), H, Dh, 1, L.BF16, L.stream_ptr(gpu_device)))
    for b, (lo, n) in enumerate(((0, 70), (70, 131))):
        x = qkv[lo:lo + n].view(1, n, 3, H, Dh)
        out = valle_hip_attn_func(x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2), causal=True)
        assert torch.equal(out[0].transpose(0, 1).reshape(n, d), packed[lo:lo + n])


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 2e-2)])
def test_hip_linear_replaces_nn_linear(gpu_device, dtype, tol):
    from vall_e_amd.plugins import HipLinear
    torch.manual_seed(0)
    lin = torch.nn.Linear(256, 384, bias=True).to(gpu_device, dtype)
    hip = HipLinear.from_linear(lin)
    x = torch.randn(2, 37, 256, device=gpu_device).to(dtype)
    want = (x.double() @ lin.weight.double().T + lin.bias.double())
    got = hip(x)
    assert got.shape == (2, 37, 384) and got.dtype == dtype
    assert ((got.double() - want).abs().max() / want.abs().max()).item() < tol
    nb = HipLinear(256, 64, bias=False, device=gpu_device, dtype=dtype)
    torch.nn.init.normal_(nb.weight)
    assert ((nb(x).double() - x.double() @ nb.weight.double().T).abs().max().item() / 16) < tol


def test_hip_embedding_replaces_nn_embedding(gpu_device):
    """``replace_embedding``'s klass (vall_e/utils/utils.py:400-432): same rows as nn.Embedding, bit for bit, for the index
    dtypes the reference sends (uint8 / int16 / int64)."""
    from vall_e_amd.plugins import HipEmbedding
    torch.manual_seed(1)
    for dtype in (torch.float32, torch.bfloat16):
        emb = torch.nn.Embedding(300, 256).to(gpu_device, dtype)
        hip = HipEmbedding(num_embeddings=300, embedding_dim=256, padding_idx=None, max_norm=None, norm_type=2.0,
                           scale_grad_by_freq=False, sparse=False).to(device=gpu_device, dtype=dtype)   # as replace_embedding builds it
        hip.load_state_dict(emb.state_dict())
        for idt in (torch.uint8, torch.int16, torch.int64):
            idx = torch.randint(0, 255 if idt == torch.uint8 else 300, (3, 17)).to(idt)
            got = hip(idx.to(gpu_device))
            assert got.shape == (3, 17, 256) and torch.equal(got, emb(idx.to(gpu_device).long()))
    with pytest.raises(IndexError):
        hip(torch.tensor([300], device=gpu_device))


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 3e-2)])
def test_hip_attention_replaces_the_attention_module(gpu_device, dtype, tol):
    """``replace_attention``'s klass (vall_e/utils/utils.py:435-462; module contract vall_e/models/arch/llama.py:134-477): built
    as klass(config=, layer_idx=, mode=), weights through the state dict, forward(hidden_states, attention_mask, is_causal,
    position_ids) -> (attn_output, None, past_key_value) against an fp64 restatement of the module: q/k/v projections, HF
    half-split RoPE at per-segment position ids, softmax(q k^T / 8 + mask) v per utterance (one causal, one not, right
    padding through the additive mask), o_proj."""
    from types import SimpleNamespace
    from vall_e_amd.plugins import HipAttention
    g = torch.Generator().manual_seed(4)
    B, T, H, Dh = 2, 45, 4, 64
    d = H * Dh
    cfg = SimpleNamespace(hidden_size=d, num_attention_heads=H, num_key_value_heads=H, head_dim=Dh, attention_bias=False,
                          attention_dropout=0.0, rope_theta=10000.0, attn_mode="valle_hip")
    att = HipAttention(config=cfg, layer_idx=0, mode="valle_hip").to(device=gpu_device, dtype=dtype)
    W = {n: (torch.randn(d, d, generator=g) / d ** 0.5) for n in "qkvo"}
    att.load_state_dict({f"{n}_proj.weight": W[n].to(dtype) for n in "qkvo"})
    x = torch.randn(B, T, d, generator=g).to(dtype)
    lens, causal = [45, 30], [True, False]
    pos = torch.stack([torch.cat([torch.arange(20), torch.arange(T - 20)]) for _ in range(B)])     # ids restart per segment
    mask = torch.full((B, 1, T, T), torch.finfo(dtype).min, dtype=dtype)
    for b in range(B):
        m = torch.zeros(T, T, dtype=dtype)
        if causal[b]:
            m = m.masked_fill(torch.ones(T, T, dtype=torch.bool).triu(1), torch.finfo(dtype).min)
        m[:, lens[b]:] = torch.finfo(dtype).min
        mask[b, 0] = m
    out, sc, kv = att(x.to(gpu_device), attention_mask=mask.to(gpu_device), is_causal=causal, position_ids=pos.to(gpu_device))
    assert sc is None and kv is None and out.shape == (B, T, d) and out.dtype == dtype
    # fp64 restatement on the same (rounded) operands
    Wd = {n: W[n].to(dtype).double() for n in "qkvo"}
    xd = x.double()
    inv = 1.0 / (10000.0 ** (torch.arange(0, Dh, 2).double() / Dh))
    fr = pos.double()[..., None] * inv
    cos, sin = torch.cat((fr, fr), -1).cos()[:, None], torch.cat((fr, fr), -1).sin()[:, None]
    rot = lambda t: torch.cat((-t[..., Dh // 2:], t[..., :Dh // 2]), dim=-1)
    q, k, v = ((xd @ Wd[n].T).view(B, T, H, Dh).transpose(1, 2) for n in "qkv")
    q, k = q * cos + rot(q) * sin, k * cos + rot(k) * sin
    want = torch.zeros(B, T, d, dtype=torch.float64)
    for b in range(B):
        n = lens[b]
        s = q[b, :, :n] @ k[b, :, :n].transpose(-1, -2) / 8.0
        if causal[b]:
            s = s.masked_fill(torch.ones(n, n, dtype=torch.bool).triu(1), float("-inf"))
        want[b, :n] = ((torch.softmax(s, -1) @ v[b, :, :n]).transpose(0, 1).reshape(n, d)) @ Wd["o"].T
    got = out.double().cpu()
    for b in range(B):
        n = lens[b]
        err = ((got[b, :n] - want[b, :n]).norm() / want[b, :n].norm()).item()
        assert err < tol, (b, err)
    with pytest.raises(NotImplementedError):
        att(x.to(gpu_device), output_attentions=True)
