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
