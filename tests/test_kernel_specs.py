"""CPU checks of the closed-form second-order formulas (tests/kernel_specs.py) against
torch autograd in fp64, plus the identity the gradient-penalty backward is built on."""
import torch

from tests import kernel_specs as K


def dir_term(fn, primals, tangents, cot):
    """<cot, J(primals) tangents> as a differentiable scalar, built the way the reference
    builds its penalty (loss.py:74-81): reverse-mode gradient with create_graph, then a dot
    product with the fixed direction."""
    s = (cot * fn(*primals)).sum()
    gs = torch.autograd.grad(s, primals, create_graph=True)
    return sum((g * t).sum() for g, t in zip(gs, tangents))


DT = torch.float64
torch.manual_seed(0)


def test_ln_formulas():
    R, H, eps = 5, 64, 1e-12
    x = torch.randn(R, H, dtype=DT, requires_grad=True)
    g = (1 + 0.1 * torch.randn(H, dtype=DT)).requires_grad_(True)
    b = (0.1 * torch.randn(H, dtype=DT)).requires_grad_(True)
    xdot, gy, ybar = (torch.randn(R, H, dtype=DT) for _ in range(3))
    f = lambda t: torch.nn.functional.layer_norm(t, (H,), g, b, eps)  # noqa: E731
    y = f(x)
    assert torch.allclose(K.ln_fwd(x, g, b, eps), y)
    dx, dg, db = torch.autograd.grad(y, (x, g, b), gy, retain_graph=True)
    sdx, sdg, sdb = K.ln_bwd(x.detach(), gy, g.detach(), eps)
    assert torch.allclose(sdx, dx) and torch.allclose(sdg, dg) and torch.allclose(sdb, db)
    _, ydot = torch.func.jvp(f, (x.detach(),), (xdot,))
    assert torch.allclose(K.ln_jvp(x.detach(), xdot, g.detach(), eps), ydot.detach())
    phi = (ybar * y).sum() + dir_term(f, (x,), (xdot,), gy)
    rx, rg, rb = torch.autograd.grad(phi, (x, g, b))
    sx, sg, sb = K.ln_bwd2(x.detach(), xdot, gy, ybar, g.detach(), eps)
    assert torch.allclose(sx, rx, atol=1e-10) and torch.allclose(sg, rg, atol=1e-10) and torch.allclose(sb, rb)


def test_gelu_formulas():
    u = torch.randn(7, 33, dtype=DT, requires_grad=True)
    udot, ggu, gubar = (torch.randn(7, 33, dtype=DT) for _ in range(3))
    y = torch.nn.functional.gelu(u)
    assert torch.allclose(K.gelu(u), y)
    phi = (gubar * y).sum() + dir_term(torch.nn.functional.gelu, (u,), (udot,), ggu)
    ref, = torch.autograd.grad(phi, u)
    assert torch.allclose(K.gelu_bwd2(u.detach(), udot, ggu, gubar), ref, atol=1e-12)


def test_attention_second_order():
    """Full attention block: phi = <ctxbar, ctx> + <gctx, ctxdot>; check qbar,kbar,vbar."""
    Sq, d = 9, 8
    q, k, v = (torch.randn(Sq, d, dtype=DT, requires_grad=True) for _ in range(3))
    qd, kd, vd, gctx, ctxbar = (torch.randn(Sq, d, dtype=DT) for _ in range(5))
    sc = d ** -0.5

    def attn(q, k, v):
        return torch.softmax(q @ k.T * sc, -1) @ v

    phi = (ctxbar * attn(q, k, v)).sum() + dir_term(attn, (q, k, v), (qd, kd, vd), gctx)
    rq, rk, rv = torch.autograd.grad(phi, (q, k, v))
    with torch.no_grad():
        P = torch.softmax(q @ k.T * sc, -1)
        Sdot = (qd @ k.T + q @ kd.T) * sc
        gP = gctx @ v.T
        gS = K.softmax_bwd(gP, P)
        X = ctxbar @ v.T + gctx @ vd.T
        Sbar, Pdot = K.softmax_bwd2(P, Sdot, gP, X)
        assert torch.allclose(Pdot, K.softmax_jvp(Sdot, P))
        qbar = (Sbar @ k + gS @ kd) * sc
        kbar = (Sbar.T @ q + gS.T @ qd) * sc
        vbar = P.T @ ctxbar + Pdot.T @ gctx
    assert torch.allclose(qbar, rq, atol=1e-10)
    assert torch.allclose(kbar, rk, atol=1e-10)
    assert torch.allclose(vbar, rv, atol=1e-10)


def test_gp_identity():
    """d gp/d theta == d/d theta <v, dS/dx> with v = (2/B)(n-1)/n * g * m held constant."""
    B, N = 3, 10
    W1 = torch.randn(N, N, dtype=DT, requires_grad=True)
    w2 = torch.randn(N, dtype=DT, requires_grad=True)
    x = torch.randn(B, N, dtype=DT, requires_grad=True)
    m = (torch.rand(B, N) > 0.2).to(DT)

    def score(x):
        return (torch.tanh(x @ W1) * torch.nn.functional.gelu(x @ W1.T)) @ w2

    g, = torch.autograd.grad(score(x).sum(), x, create_graph=True)
    n = (g * m).flatten(1).norm(dim=1)
    gp = ((n - 1) ** 2).mean()
    ref = torch.autograd.grad(gp, (W1, w2), retain_graph=True)
    v = ((2.0 / B) * (n - 1) / n)[:, None].detach() * (g * m).detach()
    sur = (v * g).sum()
    got = torch.autograd.grad(sur, (W1, w2))
    for a, b in zip(ref, got):
        assert torch.allclose(a, b, atol=1e-12)
