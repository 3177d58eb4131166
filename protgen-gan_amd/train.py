"""The trainers' inner loops (fully_masked_train.py:195-352, 10p_train.py:132-184,296-370) as
library functions over this package's Generator / Critic: fake-batch generation in blind
(fully masked) and seeded (90 %-masked) mode, one critic update, one generator update, one
GAN step (n_critic critic updates + one generator update), and the freeze schedule.

Order of random draws follows the reference: Python `random` for sequence lengths
(val_metrics.py:48-57), one CPU `torch.rand(1)` per fill iteration for the temperature
(fully_masked_train.py:223), a device `torch.rand(ids.shape)` for the 90 % mask
(10p_train.py:142), the device generator for the sampling noise, one CPU
`torch.rand(B,1,1)` for the gradient-penalty interpolation (loss.py:15)."""
from __future__ import annotations

import os
import random
import threading
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence

import torch
import torch.nn.functional as F

from . import ops
from ._lib import check, lib
from .loss import compute_gradient_penalty, critic_loss, generator_loss
from .models import Critic, Generator, RowPacking
from .ragged import Ragged, extents, plan_lengths


# A/B switch (PG_ONE_FORWARD=0 restores the call-by-call critic update everywhere)
ONE_FORWARD_DEFAULT = os.environ.get("PG_ONE_FORWARD", "1") != "0"
# with one forward: also one first-order backward over the 3B sequences (PG_ONE_BACKWARD=0: two, B and 2B)
ONE_BACKWARD_DEFAULT = os.environ.get("PG_ONE_BACKWARD", "1") != "0"
# with one forward: every row runs on its own leading columns up to the last one its mask shows (PG_RAGGED=0: all S)
RAGGED_DEFAULT = os.environ.get("PG_RAGGED", "1") != "0"
# ragged update: the real and fake rows' backward on a second stream beside the interpolates' three passes
# (PG_TWO_STREAM_BACKWARD=0: one first-order backward over all 3B sequences, then the other two passes)
TWO_STREAM_BACKWARD_DEFAULT = os.environ.get("PG_TWO_STREAM_BACKWARD", "1") != "0"
_SIDE_STREAMS: dict = {}


def _side_stream(device) -> "torch.cuda.Stream":
    """One extra stream per (device, stream it serves): the second lane of a critic update."""
    key = (torch.device(device), torch.cuda.current_stream(device).cuda_stream)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device)
    return _SIDE_STREAMS[key]


# Throughput bookkeeping: the sequence lengths every update actually ran, appended as
# (kind, lengths of the gradient-penalty interpolates, lengths of the scored rows).  A benchmark clears the list,
# runs, and prices the executed work from it; nothing in the training path reads it.
EXEC_LOG: list = []


# Diagnostics only: called with the critic head's stash ([3B][pg_head_stash_floats] fp32: input copy, the three
# post-ReLU activations, backward vectors) right after the head's forward of a one-forward critic update, before
# any backward reads it.  A precision experiment records one run's ReLU gates here and imposes them on another's.
HEAD_STASH_HOOK = None


def _note_rows(kind: str, gp_lens, score_lens):
    if len(EXEC_LOG) < 1 << 20:
        EXEC_LOG.append((kind, [int(v) for v in gp_lens], [int(v) for v in score_lens]))


@dataclass
class GanHyper:
    n_critic: int = 8                   # fully_masked_train.py:21
    lambda_gp: float = 10.0             # loss.py:3 (the SLURM grids pass 5)
    initial_masking_rate: float = 0.9   # fully_masked_train.py:83
    iteration_fill_rate: float = 0.1    # :84
    min_temp: float = 0.8               # :85
    max_temp: float = 1.2               # :86
    max_len: int = 512                  # :60, :309
    gen_batch_blind: int = 4            # :339
    warmup_head_lr: float = 1e-5        # :124
    micro_batch: Optional[int] = None   # critic updates made this many rows at a time (critic_update; no counterpart)


def sample_sequence_length(length_pool: Sequence[int], variation: float = 0.1, rng=random) -> int:
    """val_metrics.py:48-57 with the file's line lengths passed in as `length_pool`."""
    base = rng.choice(length_pool)
    var = int(base * variation)
    return min(500, max(1, base + rng.randint(-var, var)))


def sample_fake_lengths(length_pool: Sequence[int], batch_size: int, rng=random, rank: int = 0, world: int = 1) -> List[int]:
    """The template lengths (CLS and SEP included) of one blind-mode fake batch (fully_masked_train.py:203-207).
    world > 1: `rng` must be seeded alike on every rank; the lengths of the world * batch_size rows of the global
    fake batch are drawn from it and dealt by length (data.balanced_deal), so every rank fills and scores about the
    same number of tokens and no rank is the one the others wait for at each gradient reduction."""
    if world <= 1:
        return [sample_sequence_length(length_pool, rng=rng) + 2 for _ in range(batch_size)]
    from .data import balanced_deal
    glob = [sample_sequence_length(length_pool, rng=rng) + 2 for _ in range(batch_size * world)]
    return [glob[i] for i in balanced_deal(glob, world)[rank]]


def freeze_protbert(model, freeze: bool = True):
    """fully_masked_train.py:119-121."""
    for p in model.protbert.parameters():
        p.requires_grad = not freeze


def generate_fake_batch(generator: Generator, batch_size_local: int, length_pool: Sequence[int], hp: GanHyper,
                        device, rng=random, mask_id=4, cls_id=2, sep_id=3, pad_id=0,
                        lengths: Optional[Sequence[int]] = None) -> torch.Tensor:
    """Blind mode (fully_masked_train.py:195-258): fully masked templates of sampled lengths,
    attention mask `ids != MASK` recomputed every iteration, per-iteration temperature, then
    pad / truncate to max_len.  The template is assembled on the host and copied once."""
    if lengths is None:
        lengths = [sample_sequence_length(length_pool, rng=rng) + 2 for _ in range(batch_size_local)]
    S = max(lengths)
    ids = torch.full((batch_size_local, S), pad_id, dtype=torch.long)
    for i, L in enumerate(lengths):
        ids[i, :L] = mask_id
        ids[i, 0] = cls_id
        ids[i, L - 1] = sep_id
    ids = ids.to(device)
    rate = hp.initial_masking_rate
    while rate > 0:
        attn = (ids != mask_id).long()
        temperature = hp.min_temp + torch.rand(1).item() * (hp.max_temp - hp.min_temp)
        ids = generator.generate(ids, attn, temperature=temperature, keep_percent=hp.iteration_fill_rate,
                                 current_rate=rate)
        rate = max(0, rate - hp.iteration_fill_rate)
        if int(generator.last_counts[:, 1].sum().item()) == 0:
            break
    if ids.size(1) < hp.max_len:
        ids = F.pad(ids, (0, hp.max_len - ids.size(1)), value=pad_id)
    elif ids.size(1) > hp.max_len:
        ids = ids[:, :hp.max_len].contiguous()
    return ids


def generate_fakes_for_batch(generator: Generator, input_ids: torch.Tensor, hp: GanHyper, mask_id=4, cls_id=2,
                             sep_id=3, pad_id=0, unk_id=1, random_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Seeded mode (10p_train.py:132-184): mask 90 % of the meaningful positions of real
    sequences, attention mask `ids != PAD`, fill until no mask is left.  random_mask (tests): the boolean draw of
    10p_train.py:142 instead of the device generator's."""
    meaningful = (input_ids != pad_id) & (input_ids != unk_id) & (input_ids != cls_id) & (input_ids != sep_id)
    if random_mask is None:
        random_mask = torch.rand(input_ids.shape, device=input_ids.device) < hp.initial_masking_rate
    ids = input_ids.clone()
    ids[meaningful & random_mask] = mask_id
    rate = hp.initial_masking_rate
    while rate > 0:
        attn = (ids != pad_id).long()
        temperature = hp.min_temp + torch.rand(1).item() * (hp.max_temp - hp.min_temp)
        ids = generator.generate(ids, attn, temperature=temperature, keep_percent=hp.iteration_fill_rate,
                                 current_rate=rate)
        rate = max(0, rate - hp.iteration_fill_rate)
        if int(generator.last_counts[:, 1].sum().item()) == 0:
            break
    return ids


def generate_fakes_for_batches(generator: Generator, inputs: Sequence[torch.Tensor], hp: GanHyper, mask_id=4, cls_id=2,
                               sep_id=3, pad_id=0, unk_id=1, packed: bool = True,
                               random_masks: Optional[Sequence[torch.Tensor]] = None,
                               temps_per_batch: Optional[Sequence[Sequence[float]]] = None) -> List[torch.Tensor]:
    """Seeded-mode fakes for several real batches at once (the generator does not change between the
    critic updates of one iteration): every batch is masked as generate_fakes_for_batch masks it (its own
    device draw for the 90 % mask, its own CPU draw of a temperature per fill iteration, in batch order),
    then all rows are filled as one stacked batch.  The attention mask is `ids != PAD`, so nothing beyond a
    row's last non-pad position is ever attended or committed: in bf16 the encoder pass runs on each row's
    own length only (rows laid back to back), about 30 % fewer token rows at the reference's lengths."""
    if not inputs:
        return []
    S = inputs[0].size(1)
    if any(x.size(1) != S for x in inputs):
        raise ValueError("seeded batches must share one padded width")
    given_temps = temps_per_batch  # (drawn by the caller: a fill on a second host thread must leave the CPU generator alone)
    parts, temps_per_batch, spans, r = [], [], [], 0
    for bi, x in enumerate(inputs):
        meaningful = (x != pad_id) & (x != unk_id) & (x != cls_id) & (x != sep_id)
        random_mask = (random_masks[bi] if random_masks is not None
                       else torch.rand(x.shape, device=x.device) < hp.initial_masking_rate)
        ids = x.clone()
        ids[meaningful & random_mask] = mask_id
        parts.append(ids)
        temps_per_batch.append(fill_temperatures(hp) if given_temps is None else list(given_temps[bi]))
        spans.append((r, r + x.size(0)))
        r += x.size(0)
    ids = torch.cat(parts, dim=0).contiguous()
    packing = None
    if packed and generator.protbert.compute_dtype == torch.bfloat16:
        cols = torch.arange(1, S + 1, device=ids.device)
        widths = ((ids != pad_id) * cols).amax(dim=1).clamp_(min=1).tolist()  # last non-pad position + 1
        if sum(widths) < ids.numel():
            packing = RowPacking(widths, S, ids.device)
    rate, it = hp.initial_masking_rate, 0
    while rate > 0:
        attn = (ids != pad_id).long()
        groups = [(a, b, temps[it]) for (a, b), temps in zip(spans, temps_per_batch)]
        ids = generator.generate_groups(ids, attn, groups, keep_percent=hp.iteration_fill_rate, current_rate=rate,
                                        packing=packing)
        if generator.fill_shapes and generator.fill_shapes[-1] == (ids.size(0), S):
            # throughput bookkeeping: what the pass had to cover (each row's own length when packed)
            generator.fill_shapes[-1:] = ([(1, w) for w in packing.widths] if packing is not None
                                          else [(b - a, S) for a, b in spans])
        rate = max(0, rate - hp.iteration_fill_rate)
        it += 1
        if int(generator.last_counts[:, 1].sum().item()) == 0:
            break
    return [ids[a:b].contiguous() for a, b in spans]


def fill_temperatures(hp: GanHyper) -> List[float]:
    """The per-iteration temperatures one blind-mode fake batch draws (fully_masked_train.py:223):
    one CPU `torch.rand(1)` per pass of the `while rate > 0` loop, all ten of the rate schedule.

    The reference's loop stops drawing for a batch once that batch has no mask left
    (fully_masked_train.py:244-248, 10p_train.py:178-182).  That happens before the schedule ends when every
    row of the batch is short: a blind row with fewer than ten open positions, or a seeded row whose mask count
    is at most nine times its per-pass quota, is complete after nine passes or fewer.  The stacked fill still
    takes the full schedule's draws for such a batch, so from that point on the CPU generator is one or more
    draws ahead of the one-after-the-other order (later temperatures and gradient-penalty alphas differ); the
    tokens of the batch itself are not affected, it had nothing left to fill."""
    temps, rate = [], hp.initial_masking_rate
    while rate > 0:
        temps.append(hp.min_temp + torch.rand(1).item() * (hp.max_temp - hp.min_temp))
        rate = max(0, rate - hp.iteration_fill_rate)
    return temps


def generate_fake_batches(generator: Generator, lengths_per_batch: Sequence[Sequence[int]],
                          temps_per_batch: Sequence[Sequence[float]], hp: GanHyper, device, mask_id=4, cls_id=2,
                          sep_id=3, pad_id=0, packed: bool = True) -> List[torch.Tensor]:
    """Several blind-mode fake batches filled together: the generator's weights do not change
    between them, so their fill passes run as one stacked batch (one encoder pass per fill
    iteration for all of them) instead of one batch after the other.  Each batch keeps what a
    separate generate_fake_batch call would give it: its own template width (positions beyond
    it are excluded from attention, positions inside it follow `ids != MASK`), its own
    temperature per iteration, and padding to max_len at the end."""
    widths = [max(ls) for ls in lengths_per_batch]
    S = max(widths)
    rows = sum(len(ls) for ls in lengths_per_batch)
    ids = torch.full((rows, S), pad_id, dtype=torch.long)
    inside = torch.zeros(rows, S, dtype=torch.bool)
    spans, r = [], 0
    for ls, w in zip(lengths_per_batch, widths):
        spans.append((r, r + len(ls)))
        for L in ls:
            ids[r, :L] = mask_id
            ids[r, 0] = cls_id
            ids[r, L - 1] = sep_id
            inside[r, :w] = True
            r += 1
    ids, inside = ids.to(device), inside.to(device)
    # bf16: the encoder passes skip the columns beyond each batch's own width (they are outside its attention
    # and its output); fp32 keeps the rectangular pass
    packing = None
    if packed and generator.protbert.compute_dtype == torch.bfloat16 and len(set(widths)) > 1:
        packing = RowPacking([w for ls, w in zip(lengths_per_batch, widths) for _ in ls], S, device)
    rate, it = hp.initial_masking_rate, 0
    while rate > 0:
        attn = ((ids != mask_id) & inside).long()
        groups = [(a, b, temps[it]) for (a, b), temps in zip(spans, temps_per_batch)]
        ids = generator.generate_groups(ids, attn, groups, keep_percent=hp.iteration_fill_rate, current_rate=rate,
                                        packing=packing)
        if generator.fill_shapes and generator.fill_shapes[-1] == (rows, S):
            # throughput bookkeeping counts each batch at its own width, not the stacked one
            generator.fill_shapes[-1:] = [(b - a, w) for (a, b), w in zip(spans, widths)]
        rate = max(0, rate - hp.iteration_fill_rate)
        it += 1
        if int(generator.last_counts[:, 1].sum().item()) == 0:
            break
    out = []
    for (a, b), w in zip(spans, widths):
        x = ids[a:b, :min(w, hp.max_len)]
        out.append(F.pad(x, (0, hp.max_len - x.size(1)), value=pad_id).contiguous())
    return out


def _critic_update_one_forward(critic: Critic, critic_optimizer, real, attn_mask_real, fake, fake_mask, hp: GanHyper,
                               grad_sync, alpha, ragged=False, real_lengths=None, fake_lengths=None,
                               train_encoder=True, part=None) -> Optional[torch.Tensor]:
    """The critic update with ONE encoder forward over 3B sequences: the gradient-penalty interpolates
    (loss.py:22-66) and the real and fake rows (fully_masked_train.py:321-323) use the same weights, and at
    B = 8 a forward over 4096 or 8192 token rows runs at little more than half the rate of one over 12 288.
    Everything after the forward is what `compute_gradient_penalty` + `critic(...)` + `c_loss.backward()` do,
    issued by hand on views of the shared stash: input-gradient pass and penalty on the first B sequences,
    score backward on the other 2B, tangent and second-order passes on the first B.

    `train_encoder=False` (the trainers' first epoch: encoder frozen, the head learns) exists in the ragged form
    only; None is returned when that form does not apply and nothing has been done.

    `part = (rows of the whole batch, is first, is last)`: these rows are one slice of a larger batch whose update
    is made slice by slice (critic_update's micro_batch).  The losses are means over the whole batch, so a slice
    enters with the whole batch's 1 / B; gradients add up across the slices; the first slice clears them, the last
    one reduces them across ranks and steps the optimizer.  The return value is this slice's share of c_loss."""
    m, head = critic.protbert, critic.classifier
    B, S = real.shape
    Btot, first, last = part if part is not None else (B, True, True)
    H = m.config.hidden_size
    cd = m.compute_dtype
    dt = ops.dt_code(cd)
    st = ops.stream()
    dev = real.device
    lam = float(hp.lambda_gp)
    rmask = attn_mask_real.to(dev)
    if not train_encoder and not (ragged and ONE_BACKWARD_DEFAULT and S % 8 == 0):
        return None
    if first:
        critic_optimizer.zero_grad()
    if alpha is None:
        alpha = torch.rand(B, 1, 1)
    alpha = alpha.to(device=dev, dtype=torch.float32).contiguous().view(B)
    joint = (rmask.bool() & fake_mask.bool()).to(torch.float32)
    if ragged and ONE_BACKWARD_DEFAULT and S % 8 == 0:
        # columns each row needs: up to the last one its mask shows (bounds given by the caller, or read back)
        er = [min(S, int(v)) for v in real_lengths] if real_lengths is not None else extents(rmask)
        ef = [min(S, int(v)) for v in fake_lengths] if fake_lengths is not None else extents(fake_mask)
        lens_j = plan_lengths([min(a, b) for a, b in zip(er, ef)], S)
        lens_rf = plan_lengths(er + ef, S, floor=lens_j + lens_j)
        if sum(lens_j) + sum(lens_rf) < 3 * B * S or not train_encoder:
            return _critic_update_ragged(critic, critic_optimizer, real, rmask.to(torch.float32), fake,
                                         fake_mask.to(torch.float32), joint, lens_j, lens_rf[:B], lens_rf[B:], hp, grad_sync,
                                         alpha, train_encoder, part=(Btot, first, last))
    mask3 = torch.cat([joint, rmask.to(torch.float32), fake_mask.to(torch.float32)], dim=0).contiguous()
    _note_rows("critic", [S] * B, [S] * (2 * B))

    # embeddings of both batches straight into their rows of the 3B input, interpolates into the first B
    X = torch.empty(3 * B, S, H, dtype=cd, device=dev)
    sum_r = torch.empty(B, S, H, dtype=cd, device=dev)
    sum_f = torch.empty_like(sum_r)
    m._embed(real, X[B:2 * B], sum_r)
    m._embed(fake, X[2 * B:], sum_f)
    check(lib.pg_interpolate(dt, X[B:2 * B].data_ptr(), X[2 * B:].data_ptr(), alpha.data_ptr(), X[:B].data_ptr(), B, S * H, st),
          "pg_interpolate")
    hidden, stash = m._encode(None, X, mask3, want_stash=True, stash_role="primal")
    del X
    scores = torch.empty(3 * B, 1, dtype=torch.float32, device=dev)
    hstash = torch.empty(3 * B, lib.pg_head_stash_floats(H), dtype=torch.float32, device=dev)
    check(lib.pg_head_forward(dt, hidden.data_ptr(), S * H, head.store.master.data_ptr(), 3 * B, H, scores.data_ptr(),
                              hstash.data_ptr(), st), "pg_head_forward")
    if HEAD_STASH_HOOK is not None:
        HEAD_STASH_HOOK(hstash)
    m.store.begin_accumulate()
    head.store.begin_accumulate()
    ghead = head.store.grad
    w = torch.cat([torch.full((B,), -1.0 / Btot), torch.full((B,), 1.0 / Btot)]).to(dev)
    norms = torch.empty(B, dtype=torch.float32, device=dev)
    gp = torch.empty((), dtype=torch.float32, device=dev)
    joint = joint.contiguous()
    if ONE_BACKWARD_DEFAULT:
        # ONE first-order backward over the 3B sequences: the interpolates' rows start from the head's unit
        # backward (g = d sum(scores)/dx, loss.py:68-75) and only leave their input gradient and adjoints, the
        # real and fake rows start from d c_loss / d scores = -1/B, +1/B and are the only ones whose parameter
        # gradients are taken (grad_row0 = B)
        dh3 = torch.zeros(3 * B, S, H, dtype=cd, device=dev)
        check(lib.pg_head_backward(dt, head.store.master.data_ptr(), hstash.data_ptr(), None, B, H, None, dh3.data_ptr(), S * H,
                                   st), "pg_head_backward")
        check(lib.pg_head_backward(dt, head.store.master.data_ptr(), hstash[B:].data_ptr(), w.data_ptr(), 2 * B, H,
                                   ghead.data_ptr(), dh3[B:].data_ptr(), S * H, st), "pg_head_backward")
        adj = m._new_stash(3 * B, S, role="adjoint")
        adj_rows = 3 * B
        dx3 = m._encode_backward(stash, dh3, 3 * B, S, param_grads=True, adj_stash=adj, grad_row0=B)
        g = dx3[:B]
        dx2 = dx3[B:]
    else:
        # penalty: g = d sum(scores of the interpolates) / dx, keeping every layer's adjoints
        dh = torch.zeros(B, S, H, dtype=cd, device=dev)
        check(lib.pg_head_backward(dt, head.store.master.data_ptr(), hstash.data_ptr(), None, B, H, None, dh.data_ptr(), S * H,
                                   st), "pg_head_backward")
        adj = m._new_stash(B, S, role="adjoint")
        adj_rows = 0
        g = m._encode_backward(stash, dh, B, S, param_grads=False, adj_stash=adj, stash_rows=3 * B, row0=0)
        dh2 = torch.zeros(2 * B, S, H, dtype=cd, device=dev)
        check(lib.pg_head_backward(dt, head.store.master.data_ptr(), hstash[B:].data_ptr(), w.data_ptr(), 2 * B, H,
                                   ghead.data_ptr(), dh2.data_ptr(), S * H, st), "pg_head_backward")
        dx2 = m._encode_backward(stash, dh2, 2 * B, S, param_grads=True, stash_rows=3 * B, row0=B)
    check(lib.pg_gp_norms(dt, g.data_ptr(), joint.data_ptr(), B, S, H, norms.data_ptr(), gp.data_ptr(), st), "pg_gp_norms")
    c_loss = critic_loss(scores[B:2 * B], scores[2 * B:], gp, lam) * (B / Btot)
    m._embed_backward(real, sum_r, dx2[:B])
    m._embed_backward(fake, sum_f, dx2[B:])
    # penalty direction, tangent pass, second-order pass (loss.py's create_graph backward)
    # (the direction kernel divides by the rows it is given: a slice of a larger batch is rescaled to the batch's 2 / B)
    scale = torch.full((), lam * B / Btot, dtype=torch.float32, device=dev)
    v = torch.empty_like(g)
    check(lib.pg_gp_direction(dt, g.data_ptr(), joint.data_ptr(), norms.data_ptr(), B, S, H, scale.data_ptr(), v.data_ptr(),
                              None, st), "pg_gp_direction")
    tan = m._new_stash(B, S, role="tangent")
    hdot = m._encode_jvp(stash, v, B, S, tan, stash_rows=3 * B, row0=0)
    tstash = torch.empty(B, hstash.shape[1], dtype=torch.float32, device=dev)
    check(lib.pg_head_jvp_grad(dt, head.store.master.data_ptr(), hstash.data_ptr(), hdot.data_ptr(), S * H, B, H, 1.0,
                               ghead.data_ptr(), tstash.data_ptr(), None, st), "pg_head_jvp_grad")
    xbar = m._encode_backward2(stash, adj, tan, B, S, want_xbar=True, stash_rows=3 * B, row0=0, adj_rows=adj_rows)
    d_r = torch.empty_like(xbar)
    d_f = torch.empty_like(xbar)
    check(lib.pg_interpolate_bwd(dt, xbar.data_ptr(), alpha.data_ptr(), d_r.data_ptr(), d_f.data_ptr(), B, S * H, st),
          "pg_interpolate_bwd")
    m._embed_backward(real, sum_r, d_r)
    m._embed_backward(fake, sum_f, d_f)
    if last:
        if grad_sync is not None:
            grad_sync()
        critic_optimizer.step()
    return c_loss.detach()


def _critic_update_ragged(critic: Critic, critic_optimizer, real, rmask, fake, fake_mask, joint, lens_j, lens_r, lens_f,
                          hp: GanHyper, grad_sync, alpha, train_encoder=True, part=None) -> torch.Tensor:
    """_critic_update_one_forward over ragged rows: sequence b of the interpolates, of the real and of the fake
    batch runs on its lens_j[b] / lens_r[b] / lens_f[b] leading columns (lens_j[b] <= both others), all 3B
    sequences laid back to back.  Columns left out are hidden by the masks in every pass of the reference
    (loss.py:41-66, models.py:87-93) and get zero gradient there; here they are not computed.

    `train_encoder=False`: the encoder is frozen and only the head takes gradients (fully_masked_train.py:119-136).
    The penalty then needs the input gradient of the interpolates and the tangent pass through the encoder, but
    no parameter gradient of it: one first-order backward over the first B sequences without weight gradients,
    no second-order pass, no embedding backward."""
    m, head = critic.protbert, critic.classifier
    B, S = real.shape
    Btot, _, last = part if part is not None else (B, True, True)
    H = m.config.hidden_size
    cd = m.compute_dtype
    dt = ops.dt_code(cd)
    st = ops.stream()
    dev = real.device
    lam = float(hp.lambda_gp)
    _note_rows("critic", lens_j, list(lens_r) + list(lens_f))
    rg3 = Ragged(list(lens_j) + list(lens_r) + list(lens_f), S, dev)
    rgJ = Ragged(lens_j, S, dev)
    rgRF = Ragged(list(lens_r) + list(lens_f), S, dev)
    Tj, Trf, T3 = rgJ.T, rgRF.T, rg3.T
    Tr = rgRF.off[B]
    off_r = rgRF.seq_off[:B]
    off_f = rgRF.seq_off[B:2 * B]  # (rows of the fake batch inside the real+fake buffers: offsets from their start)
    mask3 = torch.cat([rgJ.pack(joint), rgRF.pack(torch.cat([rmask, fake_mask], dim=0))]).contiguous()
    ids_rf = rgRF.pack(torch.cat([real, fake], dim=0)).contiguous()

    # embeddings of the real and fake rows straight into their rows of the input, interpolates in front
    X = torch.empty(T3, H, dtype=cd, device=dev)
    sums = torch.empty(Trf, H, dtype=cd, device=dev) if train_encoder else None
    m._embed_packed(ids_rf, rgRF.pos, X[Tj:], sums)
    check(lib.pg_interpolate_ragged(dt, X[Tj:].data_ptr(), X[Tj:].data_ptr(), alpha.data_ptr(), X.data_ptr(),
                                    rgJ.seq_id.data_ptr(), rgJ.seq_off.data_ptr(), off_r.data_ptr(), off_f.data_ptr(), Tj, H, st),
          "pg_interpolate_ragged")
    hidden, stash = m._encode_ragged(rg3, x=X, mask_p=mask3, want_stash=True, stash_role="primal")
    del X
    cls = hidden.index_select(0, rg3.cls_rows)  # [3B, H]: the rows the critic reads (models.py:104)
    scores = torch.empty(3 * B, 1, dtype=torch.float32, device=dev)
    hstash = torch.empty(3 * B, lib.pg_head_stash_floats(H), dtype=torch.float32, device=dev)
    check(lib.pg_head_forward(dt, cls.data_ptr(), H, head.store.master.data_ptr(), 3 * B, H, scores.data_ptr(),
                              hstash.data_ptr(), st), "pg_head_forward")
    if HEAD_STASH_HOOK is not None:
        HEAD_STASH_HOOK(hstash)
    if train_encoder:
        m.store.begin_accumulate()
    head.store.begin_accumulate()
    ghead = head.store.grad
    w = torch.cat([torch.full((B,), -1.0 / Btot), torch.full((B,), 1.0 / Btot)]).to(dev)
    norms = torch.empty(B, dtype=torch.float32, device=dev)
    gp = torch.empty((), dtype=torch.float32, device=dev)
    dcls = torch.empty(3 * B, H, dtype=cd, device=dev)
    check(lib.pg_head_backward(dt, head.store.master.data_ptr(), hstash.data_ptr(), None, B, H, None, dcls.data_ptr(), H, st),
          "pg_head_backward")
    two_streams = train_encoder and TWO_STREAM_BACKWARD_DEFAULT
    rgd = rg3.desc(0, B, stash_T=T3, adj_T=Tj if two_streams else T3)
    side = None
    if two_streams:
        # Two lanes.  The real and fake rows need one first-order backward with weight gradients (2B sequences); the
        # interpolates need a first-order backward that keeps adjoints, the tangent pass and the second-order pass, one
        # after the other (B sequences: launches that cover a sixth to a half of the chip).  The lanes share nothing
        # but the primal stash (read) and the gradient buffers: the second-order pass, the only other writer, makes
        # every layer wait for the event the real / fake backward records once that layer's gradients are issued.
        check(lib.pg_head_backward(dt, head.store.master.data_ptr(), hstash[B:].data_ptr(), w.data_ptr(), 2 * B, H,
                                   ghead.data_ptr(), dcls[B:].data_ptr(), H, st), "pg_head_backward")
        dh_rf = torch.zeros(Trf, H, dtype=cd, device=dev).index_copy_(0, rgRF.cls_rows, dcls[B:])
        dhj = torch.zeros(Tj, H, dtype=cd, device=dev).index_copy_(0, rgJ.cls_rows, dcls[:B])
        m.enable_layer_events()
        main = torch.cuda.current_stream(dev)
        side = _side_stream(dev)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            dx2 = m._encode_backward(stash, dh_rf, 2 * B, S, param_grads=True, stash_rows=3 * B, row0=B,
                                     rgd=rg3.desc(B, 2 * B, stash_T=T3), T=Trf)
        dh_rf.record_stream(side)
        dx2.record_stream(main)
        adj = m._new_stash(B, S, Tj, role="adjoint")
        g = m._encode_backward(stash, dhj, B, S, param_grads=False, adj_stash=adj, stash_rows=3 * B, row0=0, rgd=rgd, T=Tj)
    elif train_encoder:
        # one first-order backward over the 3B sequences (see _critic_update_one_forward)
        check(lib.pg_head_backward(dt, head.store.master.data_ptr(), hstash[B:].data_ptr(), w.data_ptr(), 2 * B, H,
                                   ghead.data_ptr(), dcls[B:].data_ptr(), H, st), "pg_head_backward")
        dh3 = torch.zeros(T3, H, dtype=cd, device=dev).index_copy_(0, rg3.cls_rows, dcls)
        adj = m._new_stash(3 * B, S, T3, role="adjoint")
        dx3 = m._encode_backward(stash, dh3, 3 * B, S, param_grads=True, adj_stash=adj, grad_row0=B, rgd=rg3.desc(), T=T3)
        g, dx2 = dx3[:Tj], dx3[Tj:]
    else:
        # the head's gradient from the real and fake scores; the encoder only hands the interpolates' input gradient back
        check(lib.pg_head_backward(dt, head.store.master.data_ptr(), hstash[B:].data_ptr(), w.data_ptr(), 2 * B, H,
                                   ghead.data_ptr(), None, H, st), "pg_head_backward")
        dhj = torch.zeros(Tj, H, dtype=cd, device=dev).index_copy_(0, rgJ.cls_rows, dcls[:B])
        g = m._encode_backward(stash, dhj, B, S, param_grads=False, stash_rows=3 * B, row0=0, rgd=rgd, T=Tj)
    check(lib.pg_gp_norms_ragged(dt, g.data_ptr(), mask3.data_ptr(), rgJ.seq_off.data_ptr(), B, S, H, norms.data_ptr(),
                                 gp.data_ptr(), st), "pg_gp_norms_ragged")
    c_loss = critic_loss(scores[B:2 * B], scores[2 * B:], gp, lam) * (B / Btot)
    # penalty direction, tangent pass, second-order pass (loss.py's create_graph backward) on the first B sequences
    scale = torch.full((), lam * B / Btot, dtype=torch.float32, device=dev)
    v = torch.empty_like(g)
    check(lib.pg_gp_direction_ragged(dt, g.data_ptr(), mask3.data_ptr(), norms.data_ptr(), rgJ.seq_off.data_ptr(), B, S, H,
                                     scale.data_ptr(), v.data_ptr(), None, st), "pg_gp_direction_ragged")
    tan = m._new_stash(B, S, Tj, role="tangent")
    hdot = m._encode_jvp(stash, v, B, S, tan, stash_rows=3 * B, row0=0, rgd=rgd, T=Tj)
    hdot_cls = hdot.index_select(0, rgJ.cls_rows)
    tstash = torch.empty(B, hstash.shape[1], dtype=torch.float32, device=dev)
    check(lib.pg_head_jvp_grad(dt, head.store.master.data_ptr(), hstash.data_ptr(), hdot_cls.data_ptr(), H, B, H, 1.0,
                               ghead.data_ptr(), tstash.data_ptr(), None, st), "pg_head_jvp_grad")
    if not train_encoder:
        if last:
            if grad_sync is not None:
                grad_sync()
            critic_optimizer.step()
        return c_loss.detach()
    xbar = m._encode_backward2(stash, adj, tan, B, S, want_xbar=True, stash_rows=3 * B, row0=0,
                               adj_rows=0 if two_streams else 3 * B, rgd=rgd, T=Tj, wait_layer_events=two_streams)
    if side is not None:
        torch.cuda.current_stream(dev).wait_stream(side)
    # the interpolates' input gradient joins the real and fake rows' own (loss.py:29), then one embedding backward
    check(lib.pg_interpolate_bwd_ragged(dt, xbar.data_ptr(), alpha.data_ptr(), dx2.data_ptr(), dx2.data_ptr(),
                                        rgJ.seq_id.data_ptr(), rgJ.seq_off.data_ptr(), off_r.data_ptr(), off_f.data_ptr(), Tj, H,
                                        1, st), "pg_interpolate_bwd_ragged")
    m._embed_backward_packed(ids_rf, rgRF.pos, sums, dx2)
    if last:
        if grad_sync is not None:
            grad_sync()
        critic_optimizer.step()
    return c_loss.detach()


def critic_update(critic: Critic, critic_optimizer, real_data, attn_mask_real, fake_data, hp: GanHyper,
                  pad_id=0, grad_sync: Optional[Callable[[], None]] = None, alpha=None,
                  one_forward: Optional[bool] = None, real_lengths=None, fake_lengths=None,
                  ragged: Optional[bool] = None, micro_batch: Optional[int] = None) -> torch.Tensor:
    """fully_masked_train.py:297-326 / 10p_train.py:327-338.  `grad_sync` (data-parallel
    gradient averaging) runs between backward and the optimizer step.  Returns c_loss
    (device scalar; the caller decides when to synchronise for logging).

    `one_forward` (default: whenever every critic parameter trains): interpolates, real and fake rows
    share one encoder forward, see _critic_update_one_forward; False = the reference's sequence of calls
    through autograd.

    `ragged` (default on, with one_forward): every row runs on its leading columns up to the last one its mask
    shows, see _critic_update_ragged.  `real_lengths` / `fake_lengths`: per row, an upper bound of that column
    count known on the host (the tokenizer's lengths, the fake templates' lengths); without them the masks are
    read back once (one device-to-host copy per update).

    `micro_batch` (with one_forward): the update is made `micro_batch` rows at a time, gradients summed, one
    optimizer step at the end: the same update (losses are batch means, the penalty is per sample) with the
    activation memory of the smaller batch.  Three stashes of a 30-layer pass over 3 x 32 sequences of 1024 tokens do
    not fit one GPU's 288 GB; at 8 rows they take about 60 GB."""
    fake = fake_data.detach()
    fake_mask = (fake != pad_id).long()
    head_trains = critic.classifier[0].weight.requires_grad
    trainable = critic.protbert.encoder_trainable() and head_trains
    # (the trainers' first epoch: every encoder parameter frozen, the head learns)
    head_only = head_trains and not any(p.requires_grad for p in critic.protbert.parameters())
    if one_forward is None and head_only and ONE_FORWARD_DEFAULT and (ragged if ragged is not None else RAGGED_DEFAULT):
        real = critic.protbert._check_ids(real_data, values=False)
        fk = critic.protbert._check_ids(fake, values=False)
        if real.shape == fk.shape:
            done = _critic_update_one_forward(critic, critic_optimizer, real, attn_mask_real, fk, fake_mask, hp, grad_sync, alpha,
                                              True, real_lengths, fake_lengths, train_encoder=False)
            if done is not None:
                return done
    if one_forward is None:
        one_forward = trainable and ONE_FORWARD_DEFAULT
    if one_forward:
        if not trainable:
            raise ValueError("one_forward needs the encoder and the head trainable")
        # (ids here come from the tokenizer and from the fill kernels: no read-back of their range per update)
        real = critic.protbert._check_ids(real_data, values=False)
        fk = critic.protbert._check_ids(fake, values=False)
        if real.shape != fk.shape:
            raise ValueError(f"real {tuple(real.shape)} and fake {tuple(fk.shape)} batches differ in shape")
        if ragged is None:
            ragged = RAGGED_DEFAULT
        B = real.size(0)
        if micro_batch and 0 < micro_batch < B:
            if alpha is None:
                alpha = torch.rand(B, 1, 1)  # (one draw for the whole batch, as loss.py:15 makes it)
            total = None
            for lo in range(0, B, micro_batch):
                hi = min(B, lo + micro_batch)
                c = _critic_update_one_forward(
                    critic, critic_optimizer, real[lo:hi], attn_mask_real[lo:hi], fk[lo:hi], fake_mask[lo:hi], hp, grad_sync,
                    alpha[lo:hi], ragged, real_lengths[lo:hi] if real_lengths is not None else None,
                    fake_lengths[lo:hi] if fake_lengths is not None else None, part=(B, lo == 0, hi == B))
                total = c if total is None else total + c
            return total
        return _critic_update_one_forward(critic, critic_optimizer, real, attn_mask_real, fk, fake_mask, hp, grad_sync, alpha,
                                          ragged, real_lengths, fake_lengths)
    critic_optimizer.zero_grad()
    _note_rows("critic", [real_data.size(1)] * real_data.size(0), [real_data.size(1)] * (2 * real_data.size(0)))
    gp = compute_gradient_penalty(critic, real_data, fake, attn_mask_real, fake_mask, real_data.device, alpha=alpha)
    # real and fake rows go through the encoder as one 2B-row batch (rows are independent, so the
    # scores equal those of two separate calls); the larger GEMMs fill the chip better
    B = real_data.size(0)
    scores = critic(torch.cat([real_data, fake], dim=0),
                    attention_mask=torch.cat([attn_mask_real.to(fake_mask.dtype), fake_mask], dim=0))
    real_scores, fake_scores = scores[:B], scores[B:]
    c_loss = critic_loss(real_scores, fake_scores, gp, hp.lambda_gp)
    c_loss.backward()
    if grad_sync is not None:
        grad_sync()
    critic_optimizer.step()
    return c_loss.detach()


def _generator_scores_ragged(critic: Critic, fake, fake_mask, lens) -> torch.Tensor:
    """`g_loss = -mean(critic(fake))` and its backward into the critic (fully_masked_train.py:349-351) with every
    fake row on its lens[b] leading columns: the columns behind are pads under attention mask 0 (models.py:87-93),
    which no kept row attends and whose own rows reach no score.  Same kernels as the critic update's real and fake
    rows; gradients accumulate on top of what the critic's buffers hold, as autograd's would."""
    m, head = critic.protbert, critic.classifier
    B, S = fake.shape
    H = m.config.hidden_size
    cd = m.compute_dtype
    dt = ops.dt_code(cd)
    st = ops.stream()
    dev = fake.device
    train_encoder = m.encoder_trainable()
    _note_rows("generator", [], lens)
    rg = Ragged(list(lens), S, dev)
    T = rg.T
    ids_p = rg.pack(fake).contiguous()
    mask_p = rg.pack(fake_mask.to(torch.float32)).contiguous()
    X = torch.empty(T, H, dtype=cd, device=dev)
    sums = torch.empty(T, H, dtype=cd, device=dev) if train_encoder else None
    m._embed_packed(ids_p, rg.pos, X, sums)
    hidden, stash = m._encode_ragged(rg, x=X, mask_p=mask_p, want_stash=train_encoder, stash_role="primal")
    del X
    cls = hidden.index_select(0, rg.cls_rows)
    scores = torch.empty(B, 1, dtype=torch.float32, device=dev)
    hstash = torch.empty(B, lib.pg_head_stash_floats(H), dtype=torch.float32, device=dev)
    check(lib.pg_head_forward(dt, cls.data_ptr(), H, head.store.master.data_ptr(), B, H, scores.data_ptr(), hstash.data_ptr(), st),
          "pg_head_forward")
    g_loss = generator_loss(scores)
    head.store.begin_accumulate()
    w = torch.full((B,), -1.0 / B, dtype=torch.float32, device=dev)  # d g_loss / d score
    dcls = torch.empty(B, H, dtype=cd, device=dev) if train_encoder else None
    check(lib.pg_head_backward(dt, head.store.master.data_ptr(), hstash.data_ptr(), w.data_ptr(), B, H,
                               head.store.grad.data_ptr(), ops.ptr(dcls), H, st), "pg_head_backward")
    if train_encoder:
        m.store.begin_accumulate()
        dh = torch.zeros(T, H, dtype=cd, device=dev).index_copy_(0, rg.cls_rows, dcls)
        dx = m._encode_backward(stash, dh, B, S, param_grads=True, rgd=rg.desc(), T=T)
        m._embed_backward_packed(ids_p, rg.pos, sums, dx)
    return g_loss


def generator_update(critic: Critic, gen_optimizer, fake_data, pad_id=0, fake_lengths=None,
                     ragged: Optional[bool] = None) -> torch.Tensor:
    """fully_masked_train.py:335-352.  Token ids carry no gradient, so this back-propagates
    into the critic only and `gen_optimizer.step()` changes nothing — kept because the
    reference does the same work.

    `ragged` (default on, when the critic's encoder and head both train): the fake rows run on their
    leading columns up to the last one the mask `fake != PAD` shows (`fake_lengths`: host-side upper bounds of
    that count, else the mask is read back once), as in critic_update; False = the call through autograd."""
    gen_optimizer.zero_grad()
    m = critic.protbert
    fake = fake_data.detach()
    if ragged is None:
        ragged = RAGGED_DEFAULT
    S = fake.shape[1]
    if (ragged and ONE_BACKWARD_DEFAULT and S % 8 == 0 and m.encoder_trainable()
            and critic.classifier[0].weight.requires_grad):
        fk = m._check_ids(fake, values=False)
        fake_mask = (fk != pad_id).long()
        ext = [min(S, int(v)) for v in fake_lengths] if fake_lengths is not None else extents(fake_mask)
        lens = plan_lengths(ext, S)
        if sum(lens) < fk.numel():
            g_loss = _generator_scores_ragged(critic, fk, fake_mask, lens)
            gen_optimizer.step()
            return g_loss.detach()
    _note_rows("generator", [], [S] * fake.shape[0])
    fake_scores = critic(fake_data, attention_mask=(fake_data != pad_id).long())
    g_loss = generator_loss(fake_scores)
    g_loss.backward()
    gen_optimizer.step()
    return g_loss.detach()


class _frozen_parameters:
    """Parameters of `module` take no gradient inside the block (their requires_grad flags come back after)."""

    def __init__(self, module):
        self.params = [p for p in module.parameters() if p.requires_grad]

    def __enter__(self):
        for p in self.params:
            p.requires_grad_(False)

    def __exit__(self, *exc):
        for p in self.params:
            p.requires_grad_(True)


def generator_update_soft(generator: Generator, critic: Critic, gen_optimizer, masked_input_ids, attention_mask,
                          temperature: float = 1.0, critic_attention_mask=None, grad_sync=None,
                          critic_param_grads: bool = False) -> torch.Tensor:
    """The generator update the reference's history prescribes in place of the one above
    (docs/HISTORY.md:169-179): the generator's logits over the masked template become probabilities, the
    probabilities average the critic's word-embedding rows, and the critic scores those embeddings through its
    3-D branch (models.py:94-102), so the loss reaches the generator's weights: MLM head, encoder, embeddings.

    The critic's own parameters would collect a gradient on the way (the snippet leaves them trainable) that its
    next update throws away; `critic_param_grads=False` skips that work, the generator's gradients are the same.
    `grad_sync`: called between backward and the optimizer step (data-parallel average of the generator's
    gradients)."""
    gen_optimizer.zero_grad()
    logits = generator.protbert(input_ids=masked_input_ids, attention_mask=attention_mask).logits
    soft_probs = F.softmax(logits / temperature, dim=-1)
    table = critic.protbert.bert.embeddings.word_embeddings.weight
    cmask = attention_mask if critic_attention_mask is None else critic_attention_mask
    if critic_param_grads:
        fake_scores = critic(soft_probs @ table, attention_mask=cmask)
    else:
        with _frozen_parameters(critic):
            fake_scores = critic(soft_probs @ table.detach(), attention_mask=cmask)
    g_loss = generator_loss(fake_scores)
    if g_loss.requires_grad:  # (not in the frozen first epoch: freeze_protbert leaves the generator nothing to train)
        g_loss.backward()
        if grad_sync is not None:
            grad_sync()
    gen_optimizer.step()
    return g_loss.detach()


# FillAhead(low_priority=True): the fill's stream is created at the lowest priority the device offers, so that its
# workgroups take the CUs the updates leave idle instead of competing with them launch by launch (one GPU, same box:
# 704.3 -> 691.0 ms per GAN step).  Only without a gradient reduction beside it: with the layer-wise all-reduce on its
# own stream the low-priority fill was starved until the updates had finished (947 against 721 ms per step in a one-rank
# RCCL rehearsal), so the data-parallel callers leave it off.  The updates stay on the caller's stream either way: giving
# THEM a high-priority stream instead hung a blocking host-to-device copy of the main thread about once in four runs of
# the trainer tests; this form ran those tests 20 times over.  PG_FILL_LOW_PRIORITY=0 turns it off everywhere.
FILL_STREAM_LOW_PRIORITY = os.environ.get("PG_FILL_LOW_PRIORITY", "1") == "1"


def _low_priority_stream(device) -> "torch.cuda.Stream":
    """A stream of the device's least priority (torch.cuda.Stream offers normal and high only): created through the HIP
    runtime torch itself uses, wrapped as an external stream, never destroyed (it lives as long as the process)."""
    import ctypes as C
    torch.cuda.init()
    path = None
    with open("/proc/self/maps") as maps:  # the copy of the HIP runtime torch loaded (torch ships its own), not another one
        for line in maps:
            if "libamdhip64" in line:
                path = line.split()[-1]
                break
    if path is None:
        raise RuntimeError("no HIP runtime is loaded in this process")
    hip = C.CDLL(path)
    least, greatest = C.c_int(0), C.c_int(0)
    if hip.hipDeviceGetStreamPriorityRange(C.byref(least), C.byref(greatest)) != 0:
        raise RuntimeError("hipDeviceGetStreamPriorityRange failed")
    handle = C.c_void_p()
    with torch.cuda.device(device):
        if hip.hipStreamCreateWithPriority(C.byref(handle), C.c_uint(1), least) != 0:  # 1 = hipStreamNonBlocking
            raise RuntimeError("hipStreamCreateWithPriority failed")
    return torch.cuda.ExternalStream(handle.value, device=device)


class FillAhead:
    """The NEXT iteration's fake batches, filled while THIS iteration's updates run.

    Nothing changes the generator inside the reference's loop (its optimizer step is a no-op, see generator_update),
    so the fakes of iteration s + 1 may be filled before iteration s has finished.  The critic's updates leave CUs
    idle (their N = 1024 products cover about half the chip, the passes over the 8 interpolates a sixth), the fill's
    launches cover it several times over: issued on a second stream by a second host thread, the fill takes those CUs.
    Measured on configs[1]: fill 304 ms + updates 477 ms one after the other 775 ms, at once 716 ms.

    The host random draws of an iteration (lengths, temperatures, gradient-penalty alphas) are made in the main thread
    in the order the stacked form makes them, one whole iteration after the other; the sampling noise comes from the
    device generator in fill order.  So the fake batches, the alphas and with them every loss are the ones the
    one-iteration-at-a-time schedule produces.  Not for the soft generator update (it changes the generator)."""

    def __init__(self, device, low_priority: bool = False):
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.stream = None
        if low_priority and FILL_STREAM_LOW_PRIORITY:
            try:
                self.stream = _low_priority_stream(self.device)
            except (RuntimeError, OSError, AttributeError):  # (a scheduling hint only: any stream serves)
                self.stream = None
        if self.stream is None:
            self.stream = torch.cuda.Stream(self.device)
        self._thread: Optional[threading.Thread] = None
        self._out = None
        self._err: Optional[BaseException] = None

    def pending(self) -> bool:
        return self._thread is not None

    def start(self, job: Callable[[], object]):
        """Run job() in a second thread with the side stream current; its result waits for take()."""
        if self._thread is not None:
            raise RuntimeError("a fill is already in flight")
        self._out = self._err = None
        self.stream.wait_stream(torch.cuda.current_stream(self.device))

        def run():
            try:
                torch.cuda.set_device(self.device)
                with torch.cuda.stream(self.stream):
                    self._out = job()
            except BaseException as e:  # noqa: BLE001  (re-raised in the main thread by take / wait_host)
                self._err = e

        # (not a daemon: an interpreter that exits on an error elsewhere waits for the fill instead of tearing the device
        # runtime down under it)
        self._thread = threading.Thread(target=run, name="protgen-fill-ahead")
        self._thread.start()

    def wait_host(self):
        """Until the fill thread has issued everything (its last read-back also means the device is done with it)."""
        if self._thread is not None and self._thread.is_alive():
            self._thread.join()
        if self._err is not None:
            err, self._err, self._thread = self._err, None, None
            raise err

    def take(self):
        """The finished job's result, ordered into the current stream; None when nothing was started."""
        if self._thread is None:
            return None
        self.wait_host()
        self._thread = None
        cur = torch.cuda.current_stream(self.device)
        cur.wait_stream(self.stream)
        out, self._out = self._out, None

        def keep(o):  # the tensors were allocated on the side stream and are about to be used on this one
            if isinstance(o, torch.Tensor):
                if o.is_cuda:
                    o.record_stream(cur)
            elif isinstance(o, (list, tuple)):
                for v in o:
                    keep(v)
        keep(out)
        return out

    def close(self):
        """Drop a fill nobody will use (end of training)."""
        try:
            self.take()
        finally:
            self._thread = self._out = None


def blind_template(lengths: Sequence[int], device, mask_id=4, cls_id=2, sep_id=3, pad_id=0) -> torch.Tensor:
    """`[CLS] MASK.. [SEP] PAD..` rows of the given lengths (CLS and SEP included), as wide as the longest
    (fully_masked_train.py:203-215)."""
    ids = torch.full((len(lengths), max(lengths)), pad_id, dtype=torch.long)
    for i, L in enumerate(lengths):
        ids[i, :L] = mask_id
        ids[i, 0] = cls_id
        ids[i, L - 1] = sep_id
    return ids.to(device)


def gan_step_blind(generator: Generator, critic: Critic, gen_optimizer, critic_optimizer,
                   real_batches: List[dict], length_pool: Sequence[int], hp: GanHyper, batch_size: int, device,
                   rng=random, grad_sync=None, stacked_generation: bool = True, soft_generator_update: bool = False,
                   gen_grad_sync=None, rank: int = 0, world: int = 1, fill_ahead: Optional[FillAhead] = None,
                   more_follows: bool = True):
    """One iteration of the blind-mode training loop: n_critic critic updates (each with a
    fresh fake batch) followed by one generator update on `gen_batch_blind` fakes.

    Nothing updates the generator inside the iteration (its optimizer step is a no-op, see
    generator_update), so with `stacked_generation` all n_critic + 1 fake batches are filled
    up front as one stacked batch.  The host random draws are made in the order the
    one-after-the-other loop makes them (lengths from `rng`; temperatures and the
    gradient-penalty alpha from the CPU torch generator).

    `rank`, `world` (data-parallel, stacked form): `rng` is then the generator all ranks seed alike, and the fake
    lengths are this rank's balanced share of the global draw (sample_fake_lengths).

    `fill_ahead` (stacked form, token-id generator update): the iteration takes the fakes a previous call left in
    flight there (or fills its own when there are none) and, when `more_follows`, starts the fill of the next
    iteration's fakes on the side stream before its own updates; see FillAhead.  Real batches must then have
    `batch_size` rows.

    `soft_generator_update`: the generator update is generator_update_soft on the fully masked template of the
    generator batch (keys visible as in the fill: `ids != MASK`; the critic sees `ids != PAD`), at the first
    temperature drawn for that batch; the batch is not filled.  The generator then changes every iteration, which
    the stacked fill allows: all fills of an iteration happen before its generator update."""
    real_batches = real_batches[:hp.n_critic]
    if soft_generator_update and not stacked_generation:
        raise ValueError("the soft generator update is wired into the stacked form only")
    if not stacked_generation:
        c_sum = None
        for batch in real_batches:
            fake = generate_fake_batch(generator, batch_size, length_pool, hp, device, rng=rng,
                                       lengths=sample_fake_lengths(length_pool, batch_size, rng, rank, world))
            c = critic_update(critic, critic_optimizer, batch["input_ids"], batch["attention_mask"], fake, hp,
                              grad_sync=grad_sync)
            c_sum = c if c_sum is None else c_sum + c
        fake = generate_fake_batch(generator, hp.gen_batch_blind, length_pool, hp, device, rng=rng,
                                   lengths=sample_fake_lengths(length_pool, hp.gen_batch_blind, rng, rank, world))
        g = generator_update(critic, gen_optimizer, fake)
        return c_sum / hp.n_critic, g

    def draw_plan(n_updates, rows):
        """The host draws of one iteration, in the order the one-after-the-other loop makes them."""
        lengths, temps, alphas = [], [], []
        for _ in range(n_updates):
            lengths.append(sample_fake_lengths(length_pool, batch_size, rng, rank, world))
            temps.append(fill_temperatures(hp))
            alphas.append(torch.rand(rows, 1, 1))
        lengths.append(sample_fake_lengths(length_pool, hp.gen_batch_blind, rng, rank, world))
        temps.append(fill_temperatures(hp))
        return lengths, temps, alphas

    n_fill_skip = 1 if soft_generator_update else 0
    ahead = fill_ahead if (fill_ahead is not None and not soft_generator_update) else None
    taken = ahead.take() if ahead is not None else None
    if taken is not None:
        lengths, temps, alphas, fakes = taken
        if len(alphas) != len(real_batches) or any(a.size(0) != b["input_ids"].size(0) for a, b in zip(alphas, real_batches)):
            raise ValueError("the fakes filled ahead were planned for another number or size of real batches")
    else:
        lengths, temps, alphas = draw_plan(len(real_batches), real_batches[0]["input_ids"].size(0) if real_batches else batch_size)
        n_fill = len(lengths) - n_fill_skip
        fakes = generate_fake_batches(generator, lengths[:n_fill], temps[:n_fill], hp, device)
    if ahead is not None and more_follows:
        nl, nt, na = draw_plan(hp.n_critic, batch_size)
        ahead.start(lambda: (nl, nt, na, generate_fake_batches(generator, nl, nt, hp, device)))
    c, g = _blind_updates(critic, generator, gen_optimizer, critic_optimizer, real_batches, fakes, alphas, lengths,
                          temps, hp, device, grad_sync, soft_generator_update, gen_grad_sync)
    if ahead is not None:
        ahead.wait_host()  # (the generator is free again when this call returns; the result stays for the next call)
    return c, g


def _blind_updates(critic, generator, gen_optimizer, critic_optimizer, real_batches, fakes, alphas, lengths, temps, hp, device,
                   grad_sync, soft_generator_update, gen_grad_sync):
    """The n_critic critic updates and the generator update of one blind-mode iteration, on the current stream."""
    c_sum = None
    for batch, fake, alpha, fl in zip(real_batches, fakes, alphas, lengths):
        # (a fake row's non-pad columns end inside its template: positions past it are never masks, so never filled)
        c = critic_update(critic, critic_optimizer, batch["input_ids"], batch["attention_mask"], fake, hp,
                          grad_sync=grad_sync, alpha=alpha, real_lengths=batch.get("lengths"), fake_lengths=fl,
                          micro_batch=hp.micro_batch)
        c_sum = c if c_sum is None else c_sum + c
    if soft_generator_update:
        tpl = blind_template([min(L, hp.max_len) for L in lengths[-1]], device)
        width = tpl.size(1)
        if width % 8:  # the backward passes want a multiple of 8 columns: pad columns no key mask lets in
            tpl = F.pad(tpl, (0, 8 - width % 8), value=0)
        visible = (tpl != 4)
        visible[:, width:] = False
        g = generator_update_soft(generator, critic, gen_optimizer, tpl, visible.long(), temperature=temps[-1][0],
                                  critic_attention_mask=(tpl != 0).long(), grad_sync=gen_grad_sync)
    else:
        g = generator_update(critic, gen_optimizer, fakes[-1], fake_lengths=lengths[-1])
    return c_sum / hp.n_critic, g


def gan_step_seeded(generator: Generator, critic: Critic, gen_optimizer, critic_optimizer,
                    real_batches: List[dict], gen_batches: List[dict], hp: GanHyper, grad_sync=None,
                    stacked_generation: bool = True, soft_generator_update: bool = False, gen_grad_sync=None,
                    fill_ahead: Optional[FillAhead] = None, next_gen_batches: Optional[List[dict]] = None):
    """One iteration of the seeded-mode loop (10p_train.py:296-370): needs n_critic critic
    batches and n_critic + 1 generator batches.  With `stacked_generation` the n_critic + 1 fake
    batches are filled up front as one stacked batch (generate_fakes_for_batches).

    `fill_ahead` + `next_gen_batches` (stacked form, token-id generator update): the fakes grown from the NEXT
    iteration's generator batches are filled on the side stream beside this iteration's updates, and this iteration
    takes the ones a previous call left there (FillAhead).  The next iteration's temperatures are then drawn from
    the CPU generator before this iteration's gradient-penalty alphas instead of after them."""
    c_sum = None
    if soft_generator_update and not stacked_generation:
        raise ValueError("the soft generator update is wired into the stacked form only")
    if stacked_generation and fill_ahead is not None and not soft_generator_update:
        n_fill = hp.n_critic + 1
        fakes = fill_ahead.take()
        if fakes is None:
            fakes = generate_fakes_for_batches(generator, [gb["input_ids"] for gb in gen_batches[:n_fill]], hp)
        if next_gen_batches is not None:
            ins = [gb["input_ids"] for gb in next_gen_batches[:n_fill]]
            temps = [fill_temperatures(hp) for _ in ins]
            fill_ahead.start(lambda: generate_fakes_for_batches(generator, ins, hp, temps_per_batch=temps))

        def updates():
            total = None
            for batch, fake, gb in zip(real_batches[:hp.n_critic], fakes, gen_batches):
                c = critic_update(critic, critic_optimizer, batch["input_ids"], batch["attention_mask"], fake, hp,
                                  grad_sync=grad_sync, real_lengths=batch.get("lengths"), fake_lengths=gb.get("lengths"),
                                  micro_batch=hp.micro_batch)
                total = c if total is None else total + c
            return total / hp.n_critic, generator_update(critic, gen_optimizer, fakes[hp.n_critic],
                                                         fake_lengths=gen_batches[hp.n_critic].get("lengths"))

        c, g = updates()
        fill_ahead.wait_host()
        return c, g
    if stacked_generation:
        n_fill = hp.n_critic + (0 if soft_generator_update else 1)
        fakes = generate_fakes_for_batches(generator, [gb["input_ids"] for gb in gen_batches[:n_fill]], hp)
        for batch, fake, gb in zip(real_batches[:hp.n_critic], fakes, gen_batches):
            # (a seeded fake keeps the pad columns of the sequence it started from)
            c = critic_update(critic, critic_optimizer, batch["input_ids"], batch["attention_mask"], fake, hp,
                              grad_sync=grad_sync, real_lengths=batch.get("lengths"), fake_lengths=gb.get("lengths"),
                              micro_batch=hp.micro_batch)
            c_sum = c if c_sum is None else c_sum + c
        if soft_generator_update:
            # the generator batch masked as a fill would mask it (10p_train.py:137-145), one pass, keys `ids != PAD`
            x = gen_batches[hp.n_critic]["input_ids"]
            meaningful = (x != 0) & (x != 1) & (x != 2) & (x != 3)
            tpl = x.clone()
            tpl[meaningful & (torch.rand(x.shape, device=x.device) < hp.initial_masking_rate)] = 4
            temperature = hp.min_temp + torch.rand(1).item() * (hp.max_temp - hp.min_temp)
            g = generator_update_soft(generator, critic, gen_optimizer, tpl, (tpl != 0).long(), temperature=temperature,
                                      grad_sync=gen_grad_sync)
        else:
            g = generator_update(critic, gen_optimizer, fakes[hp.n_critic], fake_lengths=gen_batches[hp.n_critic].get("lengths"))
        return c_sum / hp.n_critic, g
    for batch, gb in zip(real_batches[:hp.n_critic], gen_batches[:hp.n_critic]):
        fake = generate_fakes_for_batch(generator, gb["input_ids"], hp)
        c = critic_update(critic, critic_optimizer, batch["input_ids"], batch["attention_mask"], fake, hp,
                          grad_sync=grad_sync)
        c_sum = c if c_sum is None else c_sum + c
    fake = generate_fakes_for_batch(generator, gen_batches[hp.n_critic]["input_ids"], hp)
    g = generator_update(critic, gen_optimizer, fake)
    return c_sum / hp.n_critic, g


@torch.no_grad()
def generate_eval_sequences(generator: Generator, tokenizer, num_sequences: int, length_pool: Sequence[int], device,
                            rng=random, noise_fn=None) -> List[str]:
    """The fill schedule of the in-training evaluation (val_metrics.py:96-140): one sequence at a time, a sampled
    length, `[CLS] MASK.. [SEP]` under an all-ones attention mask, rates 1.0, 0.9, ... while above zero (eleven
    iterations: the last one, at 1.4e-16, commits every mask left), temperature 1.0.  The reference also draws one
    CPU uniform(0.8, 1.2) per iteration and never uses it; the draw is made here too, it moves the CPU generator.
    noise_fn(n_rows, vocab) (tests): the Exp(1) draw of each iteration instead of the device generator's."""
    mask_id, cls_id, sep_id = tokenizer.mask_token_id, tokenizer.cls_token_id, tokenizer.sep_token_id
    out = []
    for _ in range(num_sequences):
        L = sample_sequence_length(length_pool, rng=rng)
        ids = torch.full((1, L + 2), mask_id, dtype=torch.long)
        ids[0, 0], ids[0, -1] = cls_id, sep_id
        ids = ids.to(device)
        attn = torch.ones_like(ids)
        rate = 1.0
        while rate > 0:
            torch.empty(1).uniform_(0.8, 1.2)
            noise = noise_fn(ids.numel(), generator.protbert.config.vocab_size) if noise_fn is not None else None
            ids = generator.generate(ids, attn, keep_percent=0.1, current_rate=rate, temperature=1.0, noise=noise)
            rate = max(0, rate - 0.1)
        out.append(tokenizer.decode(ids[0].cpu(), skip_special_tokens=True).replace(" ", "")[:600])
    return out


def synthetic_real_batch(batch_size: int, length_pool: Sequence[int], max_len: int, device, rng: random.Random,
                         gen: torch.Generator, rank: int = 0, world: int = 1):
    """A dataset.py-shaped batch (ids [B,max_len] int64, attention_mask [B,max_len] float) of
    i.i.d. residues (ids 5..24) with lengths from the pool: the stand-in for the DNMT set,
    which is not distributed with the reference.  world > 1: `rng` is seeded alike on every rank, draws the
    lengths of the global batch, and this rank keeps its balanced share (data.balanced_deal), as a BatchLoader
    does with real data; the residues come from `gen`, which each rank seeds for itself."""
    ids = torch.zeros(batch_size, max_len, dtype=torch.long)
    mask = torch.zeros(batch_size, max_len, dtype=torch.float32)
    lengths = []
    drawn = [min(rng.choice(length_pool), max_len - 2) for _ in range(batch_size * max(1, world))]
    if world > 1:
        from .data import balanced_deal
        drawn = [drawn[i] for i in balanced_deal(drawn, world)[rank]]
    for b in range(batch_size):
        L = drawn[b]
        ids[b, 0] = 2
        ids[b, 1:L + 1] = torch.randint(5, 25, (L,), generator=gen)
        ids[b, L + 1] = 3
        mask[b, :L + 2] = 1.0
        lengths.append(L + 2)
    # "lengths": what a tokenizer knows on the host (tokens per row before padding, dataset.py:19-25)
    return {"input_ids": ids.to(device), "attention_mask": mask.to(device), "lengths": lengths}
