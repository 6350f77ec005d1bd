"""Dev: the erf GELU whose result is rounded to bf16 (gelu_erf_bf16out in csrc/common.h: Abramowitz & Stegun 7.1.26 erfc form)
against the correctly rounded function and against the fp32-erf form it replaces in the bf16 epilogues, over every finite
bf16 input (host arithmetic in float32; the device's v_rcp_f32 / v_exp_f32 differ in the last bits, the class is the same)."""
import numpy as np, torch
from scipy.special import erf

u = np.arange(65536, dtype=np.uint32)
x = torch.from_numpy((u << 16).astype(np.uint32).view(np.float32).copy())
x = x[torch.isfinite(x)]
xd = x.double().numpy()
exact = torch.from_numpy(0.5 * xd * (1 + erf(xd / np.sqrt(2)))).bfloat16()
cur = (0.5 * x * (1 + torch.erf(x * 0.70710678118654752440))).bfloat16()


def fast(x):
    z = x.abs() * np.float32(0.70710678118654752440)
    t = 1.0 / (1.0 + np.float32(0.3275911) * z)
    a1, a2, a3, a4, a5 = [np.float32(v) for v in (0.254829592, -0.284496736, 1.421413741, -1.453152027, 1.061405429)]
    erfc = ((((a5 * t + a4) * t + a3) * t + a2) * t + a1) * t * torch.exp(-z * z)
    return torch.where(x >= 0, 0.5 * x * (2.0 - erfc), 0.5 * x * erfc)


def differ(a, b, m=None):
    d = a.float() != b.float()
    if m is not None:
        d = d & m
    return int(d.sum())


f = fast(x).bfloat16()
m = (x.abs() < 8) & (x.abs() > 2 ** -10)
print("finite bf16 inputs:", len(x), " of them with 2^-10 < |x| < 8:", int(m.sum()))
print("fp32-erf form != correctly rounded:", differ(cur, exact), " in that range:", differ(cur, exact, m))
print("erfc form     != correctly rounded:", differ(f, exact), " in that range:", differ(f, exact, m))
print("erfc form     != fp32-erf form    :", differ(f, cur))
