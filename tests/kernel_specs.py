"""Closed-form statements (plain torch, any device/dtype) of what each second-order HIP
kernel must compute.  tests/test_kernel_specs.py checks these formulas against torch
autograd on the CPU; the GPU tests then check the kernels against these formulas."""
import math

import torch


def ln_stats(x, eps):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    rstd = 1.0 / torch.sqrt(var + eps)
    return (x - mu) * rstd, rstd


def proj(z, xh):
    return z - z.mean(-1, keepdim=True) - xh * (z * xh).mean(-1, keepdim=True)


def ln_fwd(x, g, b, eps):
    xh, _ = ln_stats(x, eps)
    return xh * g + b


def ln_bwd(x, dy, g, eps):
    xh, rstd = ln_stats(x, eps)
    dx = rstd * proj(dy * g, xh)
    return dx, (dy * xh).reshape(-1, x.shape[-1]).sum(0), dy.reshape(-1, x.shape[-1]).sum(0)


def ln_jvp(x, xdot, g, eps):
    xh, rstd = ln_stats(x, eps)
    return g * rstd * proj(xdot, xh)


def ln_bwd2(x, xdot, gy, ybar, g, eps):
    """xbar, dgamma, dbeta of  phi = <ybar, LN(x)> + <gy, LN_jvp(x; xdot)>."""
    xh, rstd = ln_stats(x, eps)
    H = x.shape[-1]
    w = gy * g
    pd = proj(xdot, xh)
    pw = proj(w, xh)
    a = (w * xh).mean(-1, keepdim=True)
    b = (xdot * xh).mean(-1, keepdim=True)
    a0 = (w * pd).mean(-1, keepdim=True)
    xbar = -(rstd ** 2) * (xh * a0 + b * pw + a * pd)
    dg = (gy * rstd * pd).reshape(-1, H).sum(0)
    db = torch.zeros(H, dtype=x.dtype, device=x.device)
    if ybar is not None:
        xbar = xbar + rstd * proj(ybar * g, xh)
        dg = dg + (ybar * xh).reshape(-1, H).sum(0)
        db = ybar.reshape(-1, H).sum(0)
    return xbar, dg, db


def gelu(x):
    return 0.5 * x * (1 + torch.erf(x / math.sqrt(2)))


def gelu_d1(x):
    return 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)


def gelu_d2(x):
    return torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi) * (2 - x * x)


def gelu_bwd2(u, udot, ggu, gubar):
    out = gelu_d1(u) * gubar
    if udot is not None:
        out = out + gelu_d2(u) * udot * ggu
    return out


def softmax_bwd(dP, P):
    return P * (dP - (dP * P).sum(-1, keepdim=True))


def softmax_jvp(Sdot, P):
    return P * (Sdot - (P * Sdot).sum(-1, keepdim=True))


def softmax_bwd2(P, Sdot, gP, X):
    """Sbar, Pdot given  X = dphi/dP through the linear uses of P and Pdot's adjoint gP."""
    r = (P * Sdot).sum(-1, keepdim=True)
    c = (gP * P).sum(-1, keepdim=True)
    Pdot = P * (Sdot - r)
    Pbar = X + gP * (Sdot - r) - c * Sdot
    Sbar = P * (Pbar - (Pbar * P).sum(-1, keepdim=True))
    return Sbar, Pdot
