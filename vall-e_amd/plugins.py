"""Operator-level drop-ins for the reference's own plug-in points, next to the whole-model drop-in.

* ``valle_hip_attn_func`` has the signature the reference calls its fused attention backend with
  (``fused_attn_func(q, k, v, causal=, softmax_scale=, dropout_p=)`` on ``(B, H, T, Dh)`` tensors,
  vall_e/models/arch/llama.py:396-403; registered through ``AVAILABLE_ATTENTIONS``,
  vall_e/models/arch/attention/__init__.py:11-198).
* ``HipLinear`` is a module for ``ml.replace_linear(model, klass)`` (vall_e/utils/utils.py:364-462,
  applied in ``load_engines`` under ``cfg.optimizations.replace``, vall_e/engines/__init__.py:95-100).

Both call libvalle_hip.so through ctypes; tensors stay torch tensors (torch is the allocator and stream carrier).
Inference only: no autograd."""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib as L


def _strides3(t):
    return (C.c_int64 * 3)(t.stride(0), t.stride(1), t.stride(2))


def valle_hip_attn_func(q, k, v, causal=False, softmax_scale=None, dropout_p=0.0, seqlens=None, layout="HND"):
    """softmax(q k^T * softmax_scale [+ causal mask]) v on ``(B, H, T, Dh)`` tensors (``layout="NHD"``: ``(B, T, H, Dh)``,
    the flash_attn branch of llama.py:328-339) -> same shape and layout, bf16, Dh == 64.  Any strides with a contiguous
    last dimension (views of a fused qkv projection work as they are).  ``seqlens`` (int tensor [B], optional) masks right
    padding per utterance; output rows beyond it are zero."""
    if dropout_p:
        raise NotImplementedError("attention dropout is a training feature; the HIP path is inference only")
    if q.dtype != torch.bfloat16 or q.shape[-1] != 64:
        raise L.ValleHipError(f"valle_hip_attn_func is built for bf16 and head_dim 64 (got {q.dtype}, head_dim {q.shape[-1]})")
    if not (q.is_cuda and k.is_cuda and v.is_cuda):
        raise L.ValleHipError("valle_hip_attn_func needs GPU tensors; there is no CPU fallback")
    if layout == "NHD":
        q, k, v = q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)
    B, H, T, Dh = q.shape
    if k.shape != q.shape or v.shape != q.shape:
        raise ValueError(f"q, k, v must share one shape (self-attention without a cache): {tuple(q.shape)}, {tuple(k.shape)}, {tuple(v.shape)}")
    fix = lambda t: t if t.stride(-1) == 1 and all(s % 8 == 0 for s in t.stride()[:3]) and t.data_ptr() % 16 == 0 else t.contiguous()
    q, k, v = fix(q), fix(k), fix(v)
    out = (torch.zeros if seqlens is not None else torch.empty)((B, H, T, Dh), dtype=q.dtype, device=q.device)
    if seqlens is not None:
        seqlens = seqlens.to(device=q.device, dtype=torch.int32).contiguous()
    scale = float(softmax_scale) if softmax_scale is not None else 1.0 / math.sqrt(Dh)
    L.check(L.load().valle_attn_prefill_hnd(L.ptr_any(q), L.ptr_any(k), L.ptr_any(v), L.ptr_any(out), B, H, T, Dh,
                                            _strides3(q), _strides3(k), _strides3(v), _strides3(out), 1 if causal else 0,
                                            scale, L.ptr(seqlens), L.BF16, L.stream_ptr(q.device)))
    return out.transpose(1, 2) if layout == "NHD" else out


class HipLinear(torch.nn.Module):
    """``nn.Linear`` on ``valle_gemm`` (bf16 or fp32 weights, fp32 accumulate): ``klass`` for the reference's
    ``replace_linear`` -- constructed as ``klass(in_features, out_features, bias=...)`` and loaded through the state dict
    like the module it replaces (vall_e/utils/utils.py:400-428)."""

    def __init__(self, in_features, out_features, bias=True, device=None, dtype=None):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = torch.nn.Parameter(torch.empty((out_features, in_features), device=device, dtype=dtype), requires_grad=False)
        self.bias = torch.nn.Parameter(torch.zeros(out_features, device=device, dtype=dtype), requires_grad=False) if bias else None

    @classmethod
    def from_linear(cls, lin: torch.nn.Linear):
        m = cls(lin.in_features, lin.out_features, bias=lin.bias is not None, device=lin.weight.device, dtype=lin.weight.dtype)
        m.load_state_dict(lin.state_dict())
        return m

    def forward(self, x):
        w = self.weight
        if not x.is_cuda:
            raise L.ValleHipError("HipLinear needs GPU tensors; there is no CPU fallback")
        if x.dtype != w.dtype:
            raise TypeError(f"HipLinear: input {x.dtype} vs weight {w.dtype}")
        if self.in_features % 32:
            raise L.ValleHipError("HipLinear: in_features must be a multiple of 32 (valle_gemm)")
        x2 = x.reshape(-1, self.in_features).contiguous()
        M, N = x2.shape[0], self.out_features
        out = torch.empty((M, N), dtype=x.dtype, device=x.device)
        if M:
            L.check(L.load().valle_gemm(L.ptr(x2), L.ptr(w.contiguous()), L.ptr(self.bias), None, L.ptr(out), M, N, self.in_features, N,
                                        L.EPI_STORE, L.dtype_code(x.dtype), L.stream_ptr(x.device)))
        return out.reshape(*x.shape[:-1], N)


class HipEmbedding(torch.nn.Module):
    """``nn.Embedding`` on ``valle_embed_sum`` (a one-column gather): ``klass`` for the reference's ``replace_embedding``,
    which constructs it as ``klass(num_embeddings=, embedding_dim=, padding_idx=, max_norm=, norm_type=, scale_grad_by_freq=,
    sparse=)`` (vall_e/utils/utils.py:400-432).  Indices may be any integer dtype (the reference patches ``Embedding.forward``
    to upcast int16 / uint8 indices, utils/utils.py:86-104; ml.py:203-208).  Inference only; ``max_norm`` renormalisation is a
    training-time side effect and is refused."""

    def __init__(self, num_embeddings, embedding_dim, padding_idx=None, max_norm=None, norm_type=2.0, scale_grad_by_freq=False,
                 sparse=False, device=None, dtype=None):
        super().__init__()
        if max_norm is not None:
            raise NotImplementedError("HipEmbedding: max_norm renormalises the table in place (training); not built")
        self.num_embeddings, self.embedding_dim, self.padding_idx = num_embeddings, embedding_dim, padding_idx
        self.weight = torch.nn.Parameter(torch.empty((num_embeddings, embedding_dim), device=device, dtype=dtype), requires_grad=False)

    @classmethod
    def from_embedding(cls, emb: torch.nn.Embedding):
        m = cls(emb.num_embeddings, emb.embedding_dim, padding_idx=emb.padding_idx, device=emb.weight.device, dtype=emb.weight.dtype)
        m.load_state_dict(emb.state_dict())
        return m

    def forward(self, x):
        w = self.weight
        if not w.is_cuda:
            raise L.ValleHipError("HipEmbedding needs GPU tensors; there is no CPU fallback")
        n = x.numel()
        out = torch.empty((n, self.embedding_dim), dtype=w.dtype, device=w.device)
        if n:
            import numpy as np
            ids = x.detach().reshape(-1).cpu().numpy().astype(np.int64)        # integer bookkeeping on the host, as the model's plans
            if ids.min() < 0 or ids.max() >= self.num_embeddings:
                raise IndexError("index out of range in self")
            tab = np.full((n, 8), -1, dtype=np.int32)
            tab[:, 0] = ids
            idx = torch.from_numpy(tab).to(w.device)
            L.check(L.load().valle_embed_sum(L.ptr(w.contiguous()), L.ptr(idx), L.ptr(out), n, self.embedding_dim, L.dtype_code(w.dtype),
                                             L.stream_ptr(w.device)))
        return out.reshape(*x.shape, self.embedding_dim)


class HipAttention(torch.nn.Module):
    """The reference's ``Attention`` module (vall_e/models/arch/llama.py:134-477) on the HIP kernels: ``klass`` for
    ``replace_attention(model, klass, target, mode)``, which constructs it as ``klass(config=, layer_idx=, mode=)``
    (vall_e/utils/utils.py:435-462) and then loads ``q_proj / k_proj / v_proj / o_proj`` through the state dict.
    forward takes the reference's arguments and returns ``(attn_output, None, past_key_value)``:
    one fused q/k/v GEMM (rows of q and k pair-interleaved per head so a RoPE pair is adjacent) -> ``valle_rope_kv`` at
    ``position_ids`` -> ``valle_attn_prefill`` per utterance (causal or full, right padding cut off at the length the
    additive / boolean mask encodes) -> o_proj.  head_dim 64, no KV cache (the reference never passes one, base.py:521),
    no attention dropout, no attention maps (``output_attentions`` is what AR_NAR.forward_super offers)."""

    def __init__(self, config, layer_idx=0, mode="valle_hip", device=None, dtype=None):
        super().__init__()
        self.config, self.layer_idx, self.attn_mode = config, layer_idx, mode
        self.hidden_size, self.num_heads = config.hidden_size, config.num_attention_heads
        self.head_dim = getattr(config, "head_dim", None) or config.hidden_size // config.num_attention_heads
        if self.head_dim != 64 or getattr(config, "num_key_value_heads", self.num_heads) != self.num_heads:
            raise NotImplementedError("HipAttention: head_dim 64, as many key/value heads as query heads")
        if getattr(config, "attention_bias", False):
            raise NotImplementedError("HipAttention: attention_bias")
        d = self.num_heads * self.head_dim
        mk = lambda i, o: HipLinear(i, o, bias=False, device=device, dtype=dtype)
        self.q_proj, self.k_proj, self.v_proj, self.o_proj = mk(self.hidden_size, d), mk(self.hidden_size, d), mk(self.hidden_size, d), mk(d, self.hidden_size)
        self._fused = None          # (key, fused weight): q and k rows permuted (0, 32, 1, 33, ...) per head
        self._rope = None           # (device, rows, table [rows][32][2] fp32)

    def _fused_weight(self):
        ws = (self.q_proj.weight, self.k_proj.weight, self.v_proj.weight)
        key = tuple((w.data_ptr(), w._version, w.dtype, w.device) for w in ws)
        if self._fused is None or self._fused[0] != key:
            Dh, H = self.head_dim, self.num_heads
            perm_head = torch.stack((torch.arange(Dh // 2), torch.arange(Dh // 2) + Dh // 2), dim=1).reshape(-1)
            perm = (torch.arange(H)[:, None] * Dh + perm_head[None, :]).reshape(-1).to(ws[0].device)
            self._fused = (key, torch.cat((ws[0].detach()[perm], ws[1].detach()[perm], ws[2].detach())).contiguous())   # (one-off weight preparation)
        return self._fused[1]

    def _rope_table(self, device, rows):
        if self._rope is None or self._rope[0] != device or self._rope[1] < rows:
            theta = getattr(self.config, "rope_theta", None) or (getattr(self.config, "rope_parameters", None) or {}).get("rope_theta", 10000.0)
            rows = max(rows, 1024)
            inv = 1.0 / (theta ** (torch.arange(0, self.head_dim, 2, dtype=torch.int64).float() / self.head_dim))
            fr = torch.arange(rows, dtype=torch.float32)[:, None] * inv[None, :]
            self._rope = (device, rows, torch.stack((fr.cos(), fr.sin()), dim=-1).contiguous().to(device))
        return self._rope[2]

    def forward(self, hidden_states, attention_mask=None, is_causal=True, position_ids=None, past_key_value=None,
                output_attentions=False, use_cache=False, cache_position=None, position_embeddings=None, **kwargs):
        import numpy as np
        if output_attentions:
            raise NotImplementedError("HipAttention: attention maps (AR_NAR.forward_super(output_attentions=True) returns the last layer's)")
        if past_key_value is not None or use_cache:
            raise NotImplementedError("HipAttention: KV cache (the reference's forward never passes one, models/base.py:521)")
        x = hidden_states
        if not x.is_cuda:
            raise L.ValleHipError("HipAttention needs GPU tensors; there is no CPU fallback")
        lib, B, T, d = L.load(), x.shape[0], x.shape[1], self.num_heads * self.head_dim
        dt, stream = L.dtype_code(x.dtype), L.stream_ptr(x.device)
        causal = [bool(c) for c in is_causal] if isinstance(is_causal, (list, tuple)) else [bool(is_causal)] * B
        lens = [T] * B
        if attention_mask is not None:                       # right padding: keys some row of the utterance may see
            m = attention_mask
            seen = (m if m.dtype == torch.bool else m > torch.finfo(m.dtype).min / 2)
            seen = seen.reshape(B, -1, seen.shape[-1]).any(dim=1) if seen.dim() > 2 else seen
            lens = [int(v) for v in seen[:, :T].sum(dim=-1).cpu()]
        if position_ids is None:
            position_ids = torch.arange(T, device=x.device)[None, :].expand(B, T)
        pos = position_ids.to(device=x.device, dtype=torch.int32).contiguous().reshape(-1)
        rope = self._rope_table(x.device, int(pos.max().item()) + 1 if pos.numel() else 1)
        M = B * T
        x2 = x.reshape(M, self.hidden_size).contiguous()
        qkv = torch.empty((M, 3 * d), dtype=x.dtype, device=x.device)
        L.check(lib.valle_gemm(L.ptr(x2), L.ptr(self._fused_weight()), None, None, L.ptr(qkv), M, 3 * d, self.hidden_size, 3 * d,
                               L.EPI_STORE, dt, stream))
        cu_all = torch.from_numpy(np.arange(B + 1, dtype=np.int32) * T).to(x.device)
        L.check(lib.valle_rope_kv(L.ptr(qkv), L.ptr(pos), L.ptr(rope), L.ptr(cu_all), B, M, self.num_heads, self.head_dim, None, None, 0,
                                  None, None, dt, stream))
        att = torch.zeros((M, d), dtype=x.dtype, device=x.device)                  # padded rows stay zero
        cu1 = {n: torch.tensor([0, n], dtype=torch.int32, device=x.device) for n in set(lens)}
        es = x.element_size()
        for b in range(B):                                    # per utterance: its own length and causality
            if lens[b] == 0:
                continue
            L.check(lib.valle_attn_prefill(C.c_void_p(qkv.data_ptr() + b * T * 3 * d * es), C.c_void_p(att.data_ptr() + b * T * d * es),
                                           L.ptr(cu1[lens[b]]), 1, lens[b], lens[b], self.num_heads, self.head_dim, 1 if causal[b] else 0,
                                           dt, stream))
        out = self.o_proj(att.reshape(B, T, d))
        return out, None, past_key_value
