"""How fast does the vendor library (hipBLASLt through torch.matmul) run the dense-pass GEMM shapes?  A yardstick
for csrc/gemm.hip, not part of the product path."""
import torch, time
dev = "cuda:0"
shapes = [(12176, 3072, 1024, "qkv"), (12176, 1024, 1024, "o"), (12176, 8192, 1024, "gate/up"), (12176, 1024, 4096, "down"),
          (12176, 4096, 4096, "square-ish")]
for M, N, K, name in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        c = a @ w.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        c = a @ w.t()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{name:12s} M={M} N={N} K={K}: {us:8.1f} us  {2*M*N*K/us/1e6:7.1f} TFLOP/s", flush=True)
