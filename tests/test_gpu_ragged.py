"""Ragged critic passes on a real MI355X: every row of a padded batch runs on its own leading columns (up to the
last one its attention mask shows), rows laid back to back.  The reference pads to 512 and masks
(dataset.py:19-42, loss.py:41-66, models.py:87-93); the skipped columns are never read by a visible row and
get zero gradient, so results must equal the rectangular pass.

  * kernels: the four ragged attention kernels and the ragged key-side products against the rectangular
    kernels run on one sequence at a time (same arithmetic per row: identical bits), through the C ABI;
  * model: train.critic_update with ragged rows against the CPU oracle on the padded inputs (fp32: loss 2e-4,
    gradients 2e-3 of the largest, `grad is None` sets equal, untouched position rows exactly zero), and
    against the rectangular HIP update (fp32 to 1e-5, bf16 to rounding: 2e-2 relative L2 per tensor)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import protbert_cpu as O  # noqa: E402
from tests.test_gpu_model import CONFIGS, build, make_cfg, rel_close, rel_l2  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16


@pytest.fixture(scope="module")
def pg():
    import pgload
    return pgload.load()


def _ragged(pg, lens, S):
    from protgen_gan_amd.ragged import Ragged
    return Ragged(lens, S, DEV)


def test_plan_lengths_and_extents(pg):
    from protgen_gan_amd.ragged import extents, plan_lengths
    m = torch.zeros(4, 32, device=DEV)
    m[0, :20] = 1
    m[1, :32] = 1
    m[2, 5] = 1       # a hole before and after: the extent is the last visible column + 1
    m[2, 17] = 1
    ext = extents(m)  # row 3 shows nothing: counts as full
    assert ext == [20, 32, 18, 32]
    lens = plan_lengths(ext, 32)
    assert all(a >= b and a % 8 == 0 and a <= 32 for a, b in zip(lens, ext))
    assert sum(lens) % 64 == 0 or all(v == 32 for v in lens)
    assert plan_lengths([3, 9], 64, floor=[16, 8]) [0] >= 16


@pytest.mark.parametrize("resident,S,lens", [(1, 200, [72, 200, 8, 136]), (0, 200, [72, 200, 8, 136]), (1, 1024, [600, 1024, 72, 520])])
def test_ragged_attention_kernels_match_per_sequence(pg, resident, S, lens):
    """bf16 fused kernels: forward (P written), first-order backward, tangent and second-order passes, and the
    key-side products (one or two reduction segments) over a ragged batch against one rectangular launch per
    sequence.  Also: score-shaped outputs carry zeros from a sequence's end to the next multiple of 64.
    `resident`: the forward / first-order backward with K and V resident in LDS, or the ring kernels (pg_tune 25)."""
    lib, ops = pg._lib.lib, pg.ops
    lib.pg_tune(25, resident)
    try:
        _ragged_attention_case(pg, lib, ops, S, lens)
    finally:
        lib.pg_tune(25, 1)


def _ragged_attention_case(pg, lib, ops, S, lens):
    heads = 2
    H = heads * 64
    rg = _ragged(pg, lens, S)
    T, B, ld = rg.T, len(lens), ops.score_ld(S)
    g = torch.Generator().manual_seed(5)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(BF).to(DEV)  # noqa: E731
    qkv, tqkv = rnd(T, 3 * H), rnd(T, 3 * H, sc=0.5)
    dctx, gctx, ctxbar = rnd(T, H), rnd(T, H), rnd(T, H)
    mask = torch.ones(T, device=DEV)
    mask[rg.off[1] + lens[1] - 50:rg.off[1] + lens[1] - 40] = 0   # holes inside a sequence
    mask[rg.off[3] + 3] = 0
    st = ops.stream()
    dt = 1
    so = rg.seq_off.data_ptr()

    def slots():
        return torch.full((B * heads, ld, ld), float("nan"), dtype=BF, device=DEV)

    P, ctx = slots(), torch.empty(T, H, dtype=BF, device=DEV)
    assert lib.pg_attention_fwd_p_ragged(dt, qkv.data_ptr(), mask.data_ptr(), P.data_ptr(), ctx.data_ptr(), so, B, S, 0, heads,
                                         3 * H, H, st) == 0
    dS, dq = slots(), torch.zeros(T, 3 * H, dtype=BF, device=DEV)
    assert lib.pg_attention_bwd_dq_ragged(dt, qkv.data_ptr(), P.data_ptr(), ctx.data_ptr(), dctx.data_ptr(), dS.data_ptr(),
                                          dq.data_ptr(), so, B, S, 0, heads, 3 * H, H, 3 * H, st) == 0
    ctxdot = torch.empty(T, H, dtype=BF, device=DEV)
    rho = torch.zeros(B * heads, S, device=DEV)
    assert lib.pg_attention_jvp_ragged(dt, qkv.data_ptr(), tqkv.data_ptr(), P.data_ptr(), ctx.data_ptr(), ctxdot.data_ptr(),
                                       rho.data_ptr(), so, B, S, 0, heads, 3 * H, H, H, st) == 0
    Sbar, Pdot = slots(), slots()
    assert lib.pg_attention_bwd2_ragged(dt, qkv.data_ptr(), tqkv.data_ptr(), P.data_ptr(), ctx.data_ptr(), ctxdot.data_ptr(),
                                        gctx.data_ptr(), ctxbar.data_ptr(), rho.data_ptr(), Sbar.data_ptr(), Pdot.data_ptr(),
                                        so, B, S, 0, heads, 3 * H, H, st) == 0

    # key-side products over the ragged slots: dK = dS^T Q / 8 (one segment), kbar = (Sbar^T Q + dS^T Qdot) / 8 (two)
    def kside(A, Bm, A2=None, B2=None, a_kmajor=False, col=0):
        out = torch.zeros(T, 3 * H, dtype=BF, device=DEV)
        d = pg._lib.GemmDesc()
        d.A, d.B, d.C = A.data_ptr(), Bm.data_ptr() + 2 * col, out.data_ptr() + 2 * H
        d.dtype = d.c_dtype = 1
        d.a_kmajor, d.b_kmajor = int(a_kmajor), 0
        d.M, d.N, d.K = S, 64, S
        d.lda, d.ldb, d.ldc = ld, 3 * H, 3 * H
        d.batch, d.batch_inner = B * heads, heads
        d.sAo, d.sAi = heads * ld * ld, ld * ld
        d.sBi = d.sCi = 64
        d.alpha, d.beta = 0.125, 0.0
        if A2 is not None:
            d.A2, d.B2, d.K2 = A2.data_ptr(), B2.data_ptr() + 2 * col, S
        d.seq_off, d.seq_tok0, d.seq_ops = so, 0, 2 | 4
        assert lib.pg_gemm(C.byref(d), st) == 0
        return out[:, H:H + heads * 64]
    dK = kside(dS, qkv)
    # the whole first-order backward in two launches (dq, then the key-stationary dk / dv kernel), with and without dS kept
    dqkv_a, dS_a = ops.attention_bwd(qkv, P, ctx, dctx, heads, seq_off=rg.seq_off, max_len=S, keep_dS=True)
    dqkv_b, _ = ops.attention_bwd(qkv, P, ctx, dctx, heads, seq_off=rg.seq_off, max_len=S, keep_dS=False)
    kbar = kside(Sbar, qkv, dS, tqkv)
    qbar = kside(Sbar, qkv, dS, tqkv, a_kmajor=True, col=H)  # (Sbar K + dS Kdot) / 8
    torch.cuda.synchronize()

    for b, L in enumerate(lens):
        r0 = rg.off[b]
        sl = slice(r0, r0 + L)
        q1, t1 = qkv[sl].contiguous(), tqkv[sl].contiguous()
        m1 = mask[sl].contiguous()
        c1, P1 = ops.attention_fwd_p(q1, m1, 1, L, heads)
        ld1 = ops.score_ld(L)
        assert torch.equal(ctx[sl], c1), b
        Pb = P[b * heads:(b + 1) * heads]
        assert torch.equal(Pb[:, :L, :ld1], P1), b
        pad = (L + 63) // 64 * 64
        assert float(Pb[:, L:pad, :pad].float().abs().max() if pad > L else 0.0) == 0.0
        dS1, dq1 = ops.attention_bwd_dq(q1, P1, c1, dctx[sl].contiguous(), 1, L, heads)
        assert torch.equal(dS[b * heads:(b + 1) * heads][:, :L, :ld1], dS1), b
        assert torch.equal(dq[sl, :H], dq1), b
        dqkv1, _ = ops.attention_bwd(q1, P1, c1, dctx[sl].contiguous(), heads, B=1, S=L)
        assert torch.equal(dS_a[b * heads:(b + 1) * heads][:, :L, :ld1], dS1), b
        assert torch.equal(dqkv_a[sl], dqkv1) and torch.equal(dqkv_b[sl], dqkv1), b
        for h in range(heads):  # dk = dS^T Q / 8 and dv = P^T dO against fp32 matmuls over the very bf16 operands
            Q1, dO1 = q1[:, h * 64:(h + 1) * 64].float(), dctx[sl][:, h * 64:(h + 1) * 64].float()
            for col, ref in ((H, dS1[h, :, :L].float().T @ Q1 / 8), (2 * H, P1[h, :, :L].float().T @ dO1)):
                gh = dqkv1[:, col + h * 64:col + (h + 1) * 64].float()
                assert float((gh - ref).abs().max()) <= 1e-2 * float(ref.abs().max()) + 1e-6, (b, h, col)
        cd1, rho1 = ops.attention_jvp(q1, t1, P1, c1, 1, L, heads)
        assert torch.equal(ctxdot[sl], cd1), b
        assert torch.equal(rho[b * heads:(b + 1) * heads, :L], rho1), b
        Sb1, Pd1 = ops.attention_bwd2(q1, t1, P1, c1, cd1, gctx[sl].contiguous(), ctxbar[sl].contiguous(), rho1, 1, L, heads)
        assert torch.equal(Sbar[b * heads:(b + 1) * heads][:, :L, :ld1], Sb1), b
        assert torch.equal(Pdot[b * heads:(b + 1) * heads][:, :L, :ld1], Pd1), b
        for Z in (dS, Sbar, Pdot):
            Zb = Z[b * heads:(b + 1) * heads]
            if pad > L:
                assert float(Zb[:, L:pad, :pad].float().abs().max()) == 0.0, b
        # key-side products against fp32 matmuls over the very bf16 operands
        for h in range(heads):
            dSh = dS1[h, :, :L].float()
            Sbh = Sb1[h, :, :L].float()
            Q, Qd = q1[:, h * 64:(h + 1) * 64].float(), t1[:, h * 64:(h + 1) * 64].float()
            Kk, Kd = q1[:, H + h * 64:H + (h + 1) * 64].float(), t1[:, H + h * 64:H + (h + 1) * 64].float()
            for got, ref in ((dK, dSh.T @ Q / 8), (kbar, (Sbh.T @ Q + dSh.T @ Qd) / 8), (qbar, (Sbh @ Kk + dSh @ Kd) / 8)):
                gh = got[sl, h * 64:(h + 1) * 64].float()
                assert float((gh - ref).abs().max()) <= 1e-2 * float(ref.abs().max()) + 1e-6, (b, h)


class _NoStep:
    def __init__(self, critic):
        self.critic = critic

    def zero_grad(self):
        self.critic.zero_grad(set_to_none=True)

    def step(self):
        pass


def _padded_batches(B, S, ext_r, ext_f, seed, holes=True):
    g = torch.Generator().manual_seed(seed)
    real = torch.randint(5, 25, (B, S), generator=g)
    fake = torch.randint(5, 25, (B, S), generator=g)
    real[:, 0] = fake[:, 0] = O.CLS
    rm = torch.zeros(B, S)
    for b in range(B):
        rm[b, :ext_r[b]] = 1
        real[b, ext_r[b]:] = O.PAD
        fake[b, ext_f[b]:] = O.PAD
    if holes:  # a pad token the generator sampled inside a fake row, and a hidden column inside a real row
        fake[0, 5] = O.PAD
        rm[B - 1, 3] = 0
    fm = (fake != O.PAD).long()
    return real, fake, rm, fm


@pytest.mark.parametrize("host_lengths", [False, True])
def test_ragged_critic_update_fp32_vs_oracle(pg, host_lengths):
    """One critic update in fp32 with ragged rows against the oracle on the padded inputs."""
    from protgen_gan_amd import train as T
    c, P, PC, Ch, _, critic = build(pg, "tiny", torch.float32)
    B, S = 4, 64
    ext_r, ext_f = [64, 30, 17, 41], [50, 64, 9, 40]
    real, fake, rm, fm = _padded_batches(B, S, ext_r, ext_f, 81)
    torch.manual_seed(99)
    alpha = torch.rand(B, 1, 1)
    loss, gp, rs, fs, grads = O.critic_step_grads(PC, Ch, c, real, fake, rm, fm, alpha, 5.0)
    d = lambda t: t.to(DEV)  # noqa: E731
    kw = dict(real_lengths=ext_r, fake_lengths=ext_f) if host_lengths else {}
    calls = []
    orig = T._critic_update_ragged
    T._critic_update_ragged = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        l_h = T.critic_update(critic, _NoStep(critic), d(real), d(rm), d(fake), T.GanHyper(lambda_gp=5.0), alpha=alpha,
                              ragged=True, **kw)
    finally:
        T._critic_update_ragged = orig
    assert calls, "the ragged path did not run"
    rel_close(l_h.item(), loss.numpy(), 2e-4, "loss")
    named = dict(critic.named_parameters())
    assert {n for n, p in named.items() if p.grad is not None} == set(grads.keys())
    gmax = max(float(v.abs().max()) for v in grads.values())
    for n, gref in grads.items():
        rel_close(named[n].grad.cpu().numpy(), gref.numpy(), 2e-3, f"grad {n}", atol=1e-6 * gmax)
    # positions no row reaches: exactly zero, as in the reference
    gpos = named["protbert.bert.embeddings.position_embeddings.weight"].grad
    assert float(gpos[S:].abs().max()) == 0.0


def test_ragged_matches_rectangular_fp32(pg):
    from protgen_gan_amd import train as T
    c, P, PC, Ch, _, critic = build(pg, "mid", torch.float32)
    B, S = 3, 136
    real, fake, rm, fm = _padded_batches(B, S, [136, 70, 33], [90, 136, 64], 82)
    alpha = torch.rand(B, 1, 1, generator=torch.Generator().manual_seed(3))
    d = lambda t: t.to(DEV)  # noqa: E731
    out = {}
    for ragged in (False, True):
        loss = T.critic_update(critic, _NoStep(critic), d(real), d(rm), d(fake), T.GanHyper(lambda_gp=5.0), alpha=alpha,
                               ragged=ragged)
        out[ragged] = (float(loss), {n: p.grad.double().cpu() for n, p in critic.named_parameters() if p.grad is not None})
    assert abs(out[True][0] - out[False][0]) <= 1e-5 * abs(out[False][0]) + 1e-7
    assert out[True][1].keys() == out[False][1].keys()
    gmax = max(float(v.abs().max()) for v in out[False][1].values())
    for n, ref in out[False][1].items():
        err = float((out[True][1][n] - ref).abs().max())
        assert err <= 2e-5 * float(ref.abs().max()) + 1e-7 * gmax, (n, err)


def test_ragged_matches_rectangular_bf16(pg):
    """ProtBERT's layer width, 4 layers, bf16 (the fused ragged kernels and the ragged key-side products): loss
    and every gradient tensor of a ragged update against the rectangular one."""
    from protgen_gan_amd import train as T
    cfg = pg.ProtBertConfig(num_hidden_layers=4, max_position_embeddings=256)
    critic = pg.Critic(pg.ProtBertForMaskedLM(cfg, device=DEV, compute_dtype=BF).init_random(seed=4))
    B, S = 4, 256
    real, fake, rm, fm = _padded_batches(B, S, [256, 130, 77, 201], [180, 256, 100, 199], 83)
    alpha = torch.rand(B, 1, 1, generator=torch.Generator().manual_seed(4))
    d = lambda t: t.to(DEV)  # noqa: E731
    out = {}
    for ragged in (False, True):
        loss = T.critic_update(critic, _NoStep(critic), d(real), d(rm), d(fake), T.GanHyper(lambda_gp=5.0), alpha=alpha,
                               ragged=ragged)
        torch.cuda.synchronize()
        out[ragged] = (float(loss), {n: p.grad.double().cpu() for n, p in critic.named_parameters() if p.grad is not None})
    assert abs(out[True][0] - out[False][0]) <= 2e-3 * abs(out[False][0]) + 1e-4
    assert out[True][1].keys() == out[False][1].keys()
    worst = max((float((out[True][1][n] - ref).norm() / (ref.norm() + 1e-12)), n) for n, ref in out[False][1].items()
                if "key.bias" not in n)
    assert worst[0] < 2e-2, worst


@pytest.mark.parametrize("dtype,ltol,gtol", [(torch.float32, 2e-4, 2e-3), (BF, 2e-2, 6e-2)])
def test_ragged_frozen_encoder_update_vs_oracle(pg, dtype, ltol, gtol):
    """The trainers' first epoch (freeze_protbert: encoder frozen, the head learns, fully_masked_train.py:119-136) on
    ragged rows: loss and the head's gradients against the oracle with train_encoder=False; the encoder's parameters
    keep `grad is None`."""
    from protgen_gan_amd import train as T
    c, P, PC, Ch, _, critic = build(pg, "tiny", dtype)
    T.freeze_protbert(critic, True)
    B, S = 4, 64
    ext_r, ext_f = [64, 30, 17, 41], [50, 64, 9, 40]
    real, fake, rm, fm = _padded_batches(B, S, ext_r, ext_f, 84)
    alpha = torch.rand(B, 1, 1, generator=torch.Generator().manual_seed(5))
    loss, gp, rs, fs, grads = O.critic_step_grads(PC, Ch, c, real, fake, rm, fm, alpha, 5.0, train_encoder=False)
    d = lambda t: t.to(DEV)  # noqa: E731
    calls = []
    orig = T._critic_update_ragged
    T._critic_update_ragged = lambda *a, **k: (calls.append(a[-1] if not k else k.get("train_encoder", a[-1])), orig(*a, **k))[1]
    try:
        l_h = T.critic_update(critic, _NoStep(critic), d(real), d(rm), d(fake), T.GanHyper(lambda_gp=5.0), alpha=alpha)
    finally:
        T._critic_update_ragged = orig
    torch.cuda.synchronize()
    assert calls == [False], "the frozen ragged path did not run"
    rel_close(l_h.item(), loss.numpy(), ltol, "loss")
    named = dict(critic.named_parameters())
    assert {n for n, p in named.items() if p.grad is not None} == set(grads.keys())
    assert all(n.startswith("classifier.") for n in grads)
    gmax = max(float(v.abs().max()) for v in grads.values())
    if dtype == torch.float32:  # (bf16: the real and fake terms nearly cancel and a hidden state rounded to bf16 can flip
        for n, gref in grads.items():  # a gate of the head, so a per-tensor bound against fp32 says nothing; see below)
            rel_close(named[n].grad.cpu().numpy(), gref.numpy(), gtol, f"grad {n}", atol=1e-6 * gmax)
    # and equal to the call-by-call form through autograd (rectangular passes, the same precision)
    ref = {n: p.grad.clone() for n, p in named.items() if p.grad is not None}
    l2 = T.critic_update(critic, _NoStep(critic), d(real), d(rm), d(fake), T.GanHyper(lambda_gp=5.0), alpha=alpha, ragged=False)
    torch.cuda.synchronize()
    assert abs(float(l2) - float(l_h)) <= ltol * abs(float(l_h)) + 1e-6
    for n, gref in ref.items():
        assert rel_l2(named[n].grad.cpu().numpy(), gref.cpu().numpy()) < (1e-4 if dtype == torch.float32 else 5e-2), n


@pytest.mark.parametrize("dtype,gtol", [(torch.float32, 1e-4), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("host_lengths", [False, True])
def test_ragged_generator_update_matches_autograd_form(pg, dtype, gtol, host_lengths):
    """generator_update (g_loss = -mean critic(fake), backward into the critic) with the fake rows on their own
    lengths against the call through autograd on the padded batch: same loss, same gradients, same `grad is None`
    set, and gradients ADD to what the critic's buffers held (the reference does not clear them before this step)."""
    from protgen_gan_amd import train as T

    class GenOpt:
        def zero_grad(self):
            pass

        def step(self):
            pass

    B, S = 4, 64
    ext_f = [50, 64, 9, 40]
    _, fake, _, _ = _padded_batches(B, S, [64] * B, ext_f, 83)
    res = []
    for ragged in (True, False):
        c, P, PC, Ch, _, critic = build(pg, "tiny", dtype)
        critic.zero_grad(set_to_none=True)
        calls = []
        orig = T._generator_scores_ragged
        T._generator_scores_ragged = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        try:
            kw = dict(fake_lengths=ext_f) if host_lengths else {}
            g1 = T.generator_update(critic, GenOpt(), fake.to(DEV), ragged=ragged, **kw)
            g2 = T.generator_update(critic, GenOpt(), fake.to(DEV), ragged=ragged, **kw)  # accumulates
        finally:
            T._generator_scores_ragged = orig
        assert bool(calls) == ragged
        res.append((float(g1), float(g2), {n: (None if p.grad is None else p.grad.float().cpu().clone())
                                            for n, p in critic.named_parameters()}))
    (a1, a2, ga), (b1, b2, gb) = res
    rel_close(a1, b1, 2e-4 if dtype == torch.float32 else 2e-2, "g_loss")
    assert a1 == a2 and b1 == b2
    assert {n for n, g in ga.items() if g is not None} == {n for n, g in gb.items() if g is not None}
    for n, g in gb.items():
        if g is None:
            continue
        if dtype == torch.float32:
            rel_close(ga[n].numpy(), g.numpy(), gtol, f"grad {n}", atol=1e-7)
        else:
            assert rel_l2(ga[n], g) <= gtol or float(g.abs().max()) < 1e-6, (n, rel_l2(ga[n], g))
