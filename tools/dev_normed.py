# dev: per-row ratio of valle_gemm_normed to the fp64 RMSNorm -> Linear
import sys, torch
sys.path.insert(0, ".")
from vall_e_amd import _lib as L
lib = L.load(); dev = "cuda:0"
M, N, K = 333, 768, 256
g = torch.Generator().manual_seed(N + K)
x = (torch.randn(M, K, generator=g) * (0.5 + 3 * torch.rand(M, 1, generator=g))).bfloat16()
w = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16()
ln = (1 + 0.1 * torch.randn(K, generator=g)).bfloat16()
wf = (w.float() * ln.float()[None, :]).bfloat16()
xd, wfd = x.to(dev), wf.to(dev)
got = torch.zeros((M, N), dtype=torch.bfloat16, device=dev)
L.check(lib.valle_gemm_normed(L.ptr(xd), L.ptr(wfd), L.ptr(got), M, N, K, N, 0, 1e-6, None, None, L.stream_ptr()))
torch.cuda.synchronize()
xn = x.double() * torch.rsqrt(x.double().pow(2).mean(-1, keepdim=True) + 1e-6) * ln.double()
want = xn @ w.double().T
got = got.double().cpu()
r = (got * want).sum(-1) / (want * want).sum(-1)            # least-squares scale per row
print("row scale got/want, rows 0..15:", [round(float(v), 4) for v in r[:16]])
print("rows 64..79:", [round(float(v), 4) for v in r[64:80]])
print("rows 128..143:", [round(float(v), 4) for v in r[128:144]])
rc = (got[:, :128] * want[:, :128]).sum(-1) / (want[:, :128] ** 2).sum(-1)
rc2 = (got[:, 640:] * want[:, 640:]).sum(-1) / (want[:, 640:] ** 2).sum(-1)
print("cols 0..127 vs 640..767, rows 0..7:", [round(float(v), 4) for v in rc[:8]], [round(float(v), 4) for v in rc2[:8]])
ss = x.float().pow(2).sum(-1)
print("true ssq rows 0..7", [round(float(v), 1) for v in ss[:8]])
print("implied ssq rows 0..7", [round(float(v / (s * s)), 1) for v, s in zip(ss[:8], r[:8])])
