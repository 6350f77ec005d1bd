"""Dev: the dense-pass attention alone (valle_attn_prefill, bf16, 16 x 761 tokens, 16 heads, causal), timed with
events; run under rocprofv3 --pmc for counters.  usage: python tools/time_attn.py [T] [B] [reps]"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from vall_e_amd import _lib as L

T = int(sys.argv[1]) if len(sys.argv) > 1 else 761
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
H, Dh = 16, 64
dev = torch.device("cuda", 0)
lib = L.load()
g = torch.Generator().manual_seed(1)
qkv = torch.randn(B * T, 3 * H * Dh, generator=g).to(dev, torch.bfloat16)
out = torch.empty(B * T, H * Dh, dtype=torch.bfloat16, device=dev)
cu = torch.arange(B + 1, dtype=torch.int32, device=dev) * T
s = L.stream_ptr(dev)
call = lambda: L.check(lib.valle_attn_prefill(L.ptr(qkv), L.ptr(out), L.ptr(cu), B, B * T, T, H, Dh, 1, L.dtype_code(torch.bfloat16), s))
for _ in range(3):
    call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    call()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / reps
fl = B * H * 4 * T * T * Dh / 2
print(f"causal attention B={B} T={T}: {us:.1f} us, {fl / us * 1e-6:.1f} TFLOP/s (causal half counted)")
