"""Dev: times valle_gemm on given shapes (bf16, random operands).  usage: time_gemm.py M,N,K,epi[,ldc] ...  (epi by name:
store resid geglu f32out gelu resid_bias)"""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from vall_e_amd import _lib as L

lib = L.load()
dev = torch.device("cuda", 0)
EPI = dict(store=L.EPI_STORE, resid=L.EPI_RESID, geglu=L.EPI_GEGLU, f32out=L.EPI_F32OUT, gelu=L.EPI_GELU, resid_bias=L.EPI_RESID_BIAS)
for spec in sys.argv[1:]:
    f = spec.split(",")
    M, N, K, epi = int(f[0]), int(f[1]), int(f[2]), f[3]
    Nout = N // 2 if epi == "geglu" else N
    ldc = int(f[4]) if len(f) > 4 else (Nout + 7) // 8 * 8
    A = torch.randn(M, K, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    bias = torch.randn(N, device=dev)
    bias = bias if epi == "f32out" else bias.bfloat16()
    R = torch.randn(M, ldc, device=dev).bfloat16()
    Cc = torch.empty((M, ldc), dtype=torch.float32 if epi == "f32out" else torch.bfloat16, device=dev)
    bp = None if epi in ("resid", "geglu") else bias
    rp = R if epi in ("resid", "resid_bias") else None
    run = lambda: L.check(lib.valle_gemm(L.ptr(A), L.ptr(W), L.ptr(bp), L.ptr(rp), L.ptr(Cc), M, N, K, ldc, EPI[epi], L.BF16, L.stream_ptr(dev)))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"{spec:34s} {us:8.1f} us  {2.0 * M * N * K / us * 1e-6:7.1f} TFLOP/s", flush=True)
