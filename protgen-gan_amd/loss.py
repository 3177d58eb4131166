"""WGAN-GP losses with the reference's surface (loss.py:3-96).

`compute_gradient_penalty` returns a 0-dim tensor that plugs into
`critic_loss(...).backward()` exactly like the reference's, but the second-order
differentiation is not left to autograd.  With S = sum_b s_b the critic score,
g = dS/dx and n_b = ||g_b * m||:

    gp = mean_b (n_b - 1)^2 ,     d gp / d theta = d/d theta <v, dS/dx>   with v held fixed,
    v_b = (2/B) (n_b - 1)/n_b * g_b * m     (so <v, dS/dx> is the directional derivative of S along v).

Forward = encoder forward (primal stash) + first-order reverse to the input (adjoint
stash).  Backward = one tangent-forward pass along v (tangent stash) + one reverse pass
over the (primal, tangent) pair, which adds the parameter gradients straight into the
flat gradient buffer.  Every pass is one C call into libprotgen_hip.so."""
from __future__ import annotations

import torch

from . import _lib, ops
from ._lib import check, lib


def critic_loss(real_scores, fake_scores, gradient_penalty, lambda_gp=10):
    return -torch.mean(real_scores) + torch.mean(fake_scores) + lambda_gp * gradient_penalty


def generator_loss(fake_scores):
    return -torch.mean(fake_scores)


class _GradientPenaltyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, critic, real, fake, joint, alpha, *params):
        m = critic.protbert
        head = critic.classifier
        B, S = real.shape
        H = m.config.hidden_size
        dt = ops.dt_code(m.compute_dtype)
        st = ops.stream()
        enc_train = m.encoder_trainable()
        head_train = head[0].weight.requires_grad
        dev = real.device

        # embeddings of both batches, interpolation (loss.py:22-29)
        e_r = torch.empty(B, S, H, dtype=m.compute_dtype, device=dev)
        e_f = torch.empty_like(e_r)
        sum_r = torch.empty_like(e_r) if enc_train else None
        sum_f = torch.empty_like(e_r) if enc_train else None
        m._embed(real, e_r, sum_r)
        m._embed(fake, e_f, sum_f)
        x = torch.empty_like(e_r)
        check(lib.pg_interpolate(dt, e_r.data_ptr(), e_f.data_ptr(), alpha.data_ptr(), x.data_ptr(), B, S * H, st),
              "pg_interpolate")
        del e_r, e_f

        # pass A: primal forward with stash; critic head
        hidden, stash = m._encode(None, x, joint, want_stash=True)
        scores = torch.empty(B, 1, dtype=torch.float32, device=dev)
        hstash = torch.empty(B, lib.pg_head_stash_floats(H), dtype=torch.float32, device=dev)
        check(lib.pg_head_forward(dt, hidden.data_ptr(), S * H, head.store.master.data_ptr(), B, H, scores.data_ptr(),
                                  hstash.data_ptr(), st), "pg_head_forward")
        # pass B: g = d sum(scores) / dx, keeping every layer's adjoints
        dh = torch.zeros_like(hidden)
        check(lib.pg_head_backward(dt, head.store.master.data_ptr(), hstash.data_ptr(), None, B, H, None, dh.data_ptr(),
                                   S * H, st), "pg_head_backward")
        need_second = enc_train
        adj = m._new_stash(B, S) if need_second else None
        g = m._encode_backward(stash, dh, B, S, param_grads=False, adj_stash=adj)
        norms = torch.empty(B, dtype=torch.float32, device=dev)
        gp = torch.empty((), dtype=torch.float32, device=dev)
        check(lib.pg_gp_norms(dt, g.data_ptr(), joint.data_ptr(), B, S, H, norms.data_ptr(), gp.data_ptr(), st),
              "pg_gp_norms")

        ctx.critic, ctx.real, ctx.fake, ctx.joint, ctx.alpha = critic, real, fake, joint, alpha
        ctx.stash, ctx.adj, ctx.hstash, ctx.g, ctx.norms = stash, adj, hstash, g, norms
        ctx.sums = (sum_r, sum_f)
        ctx.enc_train, ctx.head_train = enc_train, head_train
        return gp

    @staticmethod
    def backward(ctx, grad_gp):
        critic = ctx.critic
        m, head = critic.protbert, critic.classifier
        B, S = ctx.real.shape
        H = m.config.hidden_size
        dt = ops.dt_code(m.compute_dtype)
        st = ops.stream()
        dev = ctx.real.device
        scale = grad_gp.to(torch.float32).contiguous()

        # GP direction v (the incoming loss weight is read on the device)
        v = torch.empty_like(ctx.g)
        check(lib.pg_gp_direction(dt, ctx.g.data_ptr(), ctx.joint.data_ptr(), ctx.norms.data_ptr(), B, S, H,
                                  scale.data_ptr(), v.data_ptr(), None, st), "pg_gp_direction")
        # pass C: tangent forward along v
        tan = m._new_stash(B, S)
        hdot = m._encode_jvp(ctx.stash, v, B, S, tan)
        ghead = None
        if ctx.head_train:
            head.store.begin_accumulate()
            ghead = head.store.grad
        tstash = torch.empty_like(ctx.hstash)
        check(lib.pg_head_jvp_grad(dt, head.store.master.data_ptr(), ctx.hstash.data_ptr(), hdot.data_ptr(), S * H, B, H,
                                   1.0, ops.ptr(ghead), tstash.data_ptr(), None, st), "pg_head_jvp_grad")
        # pass D: second-order reverse -> encoder parameter gradients and d/dx
        if ctx.enc_train:
            m.store.begin_accumulate()
            xbar = m._encode_backward2(ctx.stash, ctx.adj, tan, B, S, want_xbar=True)
            d_r = torch.empty_like(xbar)
            d_f = torch.empty_like(xbar)
            check(lib.pg_interpolate_bwd(dt, xbar.data_ptr(), ctx.alpha.data_ptr(), d_r.data_ptr(), d_f.data_ptr(), B,
                                         S * H, st), "pg_interpolate_bwd")
            m._embed_backward(ctx.real, ctx.sums[0], d_r)
            m._embed_backward(ctx.fake, ctx.sums[1], d_f)
        ctx.stash = ctx.adj = ctx.g = None
        return (None,) * len(ctx.needs_input_grad)


def compute_gradient_penalty(critic, real_data, fake_data, real_mask, fake_mask, device, alpha=None):
    """loss.py:10-96.  alpha ~ U[0,1) of shape [B,1,1] is drawn from the CPU generator as
    the reference does (loss.py:15) unless given.  Masks may be bool, integer or float
    0/1 tensors (the reference's `&` rejects float masks; casting to bool first is the
    documented superset)."""
    m = critic.protbert
    real = m._check_ids(real_data)
    fake = m._check_ids(fake_data)
    if real.shape != fake.shape:
        raise ValueError(f"real {tuple(real.shape)} and fake {tuple(fake.shape)} batches differ in shape")
    B, S = real.shape
    if alpha is None:
        alpha = torch.rand(B, 1, 1)
    alpha = alpha.to(device=real.device, dtype=torch.float32).contiguous().view(B)
    joint = (real_mask.to(real.device).bool() & fake_mask.to(real.device).bool()).to(torch.float32).contiguous()
    params = [p for p in critic.parameters() if p.requires_grad]
    return _GradientPenaltyFn.apply(critic, real, fake, joint, alpha, *params)
