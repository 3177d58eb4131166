"""Parity at BASELINE.json's own sizes, full 30-layer ProtBERT:

* configs[0] (generate.py full mode, batch 4, L 510 = S 512, fp32): two teacher-forced fill iterations of the fp32
  engine against the CPU oracle on the same Exp(1) draw — logits within 1e-4, identical sampled tokens at every
  open position, identical committed positions and tokens; the smallest margins of the decisions are printed so a
  near-tie would be visible (models.py:18-64);
* configs[1]'s critic update at batch 2 x S 512 in bf16 against the fp32 oracle: losses and penalty within 2e-2, the
  CLS hidden state within 2e-2, and the scores within what that hidden-state distance moves the head by
  (first-order bound through the head's own Jacobian);
* the bf16 / fp32 gradient distance with the head as initialised, and again with the fp32 run's ReLU gates imposed
  on the bf16 run: the distance that the free gates produce goes away (models.py:76-84 is piecewise linear)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import protbert_cpu as O  # noqa: E402
from tests.test_gpu_model import make_cfg, rel_close, rel_l2  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16


@pytest.fixture(scope="module")
def pg():
    import pgload
    return pgload.load()


@pytest.fixture(scope="module")
def full():
    c = O.BertCfg(max_position_embeddings=512)
    return c, O.init_params(c, seed=0, bias_std=0.02)


def _margins(probs, noise, pred, conf, ids, k_of_row):
    """Smallest relative gap between the winning and the second p/q among open positions, and between the last
    committed and the first uncommitted confidence of a row."""
    B, S, V = probs.shape
    ratio = (probs.view(-1, V) / noise).view(B, S, V)
    top2 = ratio.topk(2, dim=-1).values
    gap = ((top2[..., 0] - top2[..., 1]) / top2[..., 0])[ids == O.MASK]
    kth = []
    for b in range(B):
        open_ = ids[b] == O.MASK
        c = torch.sort(conf[b][open_], descending=True).values
        k = k_of_row[b]
        if 0 < k < c.numel():
            kth.append(float((c[k - 1] - c[k]) / c[k - 1]))
    return float(gap.min()), (min(kth) if kth else float("nan"))


def test_fill_iterations_full_size_fp32_b4_s512_vs_oracle(pg, full):
    c, P = full
    gen = pg.Generator(pg.ProtBertForMaskedLM.from_params(make_cfg(pg, c), P, device=DEV, compute_dtype=torch.float32))
    B, L = 4, 510
    S = L + 2
    ids0, attn = O.build_fully_masked_batch([L] * B)
    g = torch.Generator().manual_seed(5)
    # iteration 1: the template at rate 1.0; iteration 2: the oracle's result filled on to about one half, rate 0.5
    state = ids0.clone()
    for it, rate in enumerate((1.0, 0.5)):
        if it == 1:
            open_pos = (state == O.MASK).nonzero()
            take = open_pos[torch.randperm(open_pos.size(0), generator=g)[: open_pos.size(0) * 4 // 9]]
            state[take[:, 0], take[:, 1]] = torch.randint(5, 25, (take.size(0),), generator=g)
        noise = O.draw_noise(B * S, c.vocab_size, g)
        with torch.no_grad():
            logits = O.generator_forward(P, c, state, attn)
            probs, pred, conf = O.sample_from_logits(logits, 1.0, noise)
            want = state.clone()
            committed = O.commit_step(want, pred, conf, 0.1, rate)
            got_logits = gen(state.to(DEV), attn.to(DEV)).cpu()
        rel_close(got_logits.numpy(), logits.numpy(), 1e-4, f"logits, iteration {it + 1}")
        got = gen.generate(state.clone().to(DEV), attention_mask=attn.to(DEV), temperature=1.0, keep_percent=0.1,
                           current_rate=rate, noise=noise).cpu()
        gap, kth = _margins(probs, noise, pred, conf, state, committed)
        print(f"iteration {it + 1}: committed {committed}, logits max err "
              f"{float((got_logits - logits).abs().max()):.2e} on scale {float(logits.abs().max()):.2f}; smallest "
              f"relative margin of the sampled token's p/q {gap:.2e}, of the k-th confidence {kth:.2e}")
        changed = got != state
        assert torch.equal(changed, want != state), f"iteration {it + 1}: committed positions differ"
        assert torch.equal(got, want), f"iteration {it + 1}: committed tokens differ"
        # (a sampled [MASK] leaves its position open: committed counts bound the changed positions from above)
        assert int(changed.sum()) <= sum(committed)
        state = want


def _critic_pair(pg, c, P, C, dtype):
    critic = pg.Critic(pg.ProtBertForMaskedLM.from_params(make_cfg(pg, c), P, device=DEV, compute_dtype=dtype))
    critic.classifier.load_state_dict(C)
    return critic


def test_critic_update_full_size_bf16_b2_s512_vs_oracle(pg, full):
    """The quantity bench.py's parity probe prints, asserted.  The head maps the CLS row h to a score through three
    ReLU layers; a bf16 encoder moves h by about one per cent of its norm, and the score moves by J dh (J = the
    head's Jacobian at h) — that, not a fraction of the score's own size, is the bound that holds: a randomly
    initialised head puts out scores of the size of its last bias (0.07 here) whatever the size of h."""
    c, P = full
    C = O.init_classifier(c.hidden_size, seed=2)
    critic = _critic_pair(pg, c, P, C, BF)
    g = torch.Generator().manual_seed(0)
    B, S = 2, 512
    real = torch.randint(5, 25, (B, S), generator=g)
    fake = torch.randint(5, 25, (B, S), generator=g)
    ones = torch.ones(B, S, dtype=torch.long)
    alpha = torch.rand(B, 1, 1, generator=g)
    d = lambda t: t.to(DEV)  # noqa: E731
    loss, gp, rs, fs, _ = O.critic_step_grads(P, C, c, real, fake, ones, ones, alpha, 5.0)
    gp_h = pg.compute_gradient_penalty(critic, d(real), d(fake), d(ones), d(ones), DEV, alpha=alpha)
    with torch.no_grad():
        rs_h, fs_h = critic(d(real), attention_mask=d(ones)), critic(d(fake), attention_mask=d(ones))
    l_h = pg.critic_loss(rs_h, fs_h, gp_h, 5.0)
    rel_close(float(gp_h.detach()), float(gp.detach()), 2e-2, "gp")
    rel_close(float(l_h.detach()), float(loss.detach()), 2e-2, "c_loss")
    # the CLS hidden state of both batches, engine against oracle, and the head's Jacobian there
    x = torch.cat([real, fake])
    with torch.no_grad():
        h_ref = O.encoder(P, c, O.embeddings(P, c, x), torch.cat([ones, ones]))[:, 0, :]
        out = critic.protbert(input_ids=d(x), attention_mask=d(torch.cat([ones, ones])), output_hidden_states=True)
        h_eng = out.hidden_states[-1][:, 0, :].float().cpu()
    hh = h_ref.clone().requires_grad_(True)
    s_ref = O.classifier(C, hh)
    J = torch.autograd.grad(s_ref.sum(), hh)[0]
    s_eng = torch.cat([rs_h, fs_h]).float().cpu().view(-1)
    dh = h_eng - h_ref
    rel_h = dh.norm(dim=1) / h_ref.norm(dim=1)
    first_order = (J * dh).sum(dim=1)
    err = s_eng - s_ref.detach().view(-1)
    print("bf16 B2xS512, 30 layers: |dh|/|h| per row", [f"{float(v):.3e}" for v in rel_h], "score error", err.tolist(),
          "J.dh", first_order.tolist(), "scores", s_ref.detach().view(-1).tolist())
    assert float(rel_h.max()) < 2e-2
    # the score error is the hidden-state error carried through the head (gates that flip add second-order terms)
    bound = J.norm(dim=1) * dh.norm(dim=1)
    assert bool((err.abs() <= 1.5 * bound + 1e-5).all()), (err.tolist(), bound.tolist())
    assert float((err - first_order).abs().max()) <= 0.5 * float(bound.max()) + 1e-5


def test_bf16_gradients_with_the_fp32_runs_gate_pattern(pg):
    """One critic update, 4 layers of ProtBERT's width, batch 8 x S 512, head as initialised: bf16 gradients against
    fp32 gradients per tensor, free (measured: tens of per cent at the median, whole tensors moved by a head unit whose
    ReLU gate flips) and with the fp32 run's gates imposed on the bf16 run's head stash (the rounding level)."""
    from protgen_gan_amd import train as T
    from tests.test_gpu_bench_shapes import _NoStep, _bench_batches, _grads
    cfg = pg.ProtBertConfig(num_hidden_layers=4)
    real, fake, alpha = _bench_batches(pg, 8, 512, 5)
    H = cfg.hidden_size
    lo, hi = H, H + H // 2 + H // 4 + H // 16  # the three post-ReLU activation slots of a stash row
    gates = {}

    def record(hstash):
        gates["fp32"] = (hstash[:, lo:hi] > 0).clone()

    def impose(hstash):
        a = hstash[:, lo:hi]
        gates["flipped"] = int(((a > 0) != gates["fp32"]).sum())
        a.copy_(torch.where(gates["fp32"], a.clamp_min(1e-30), torch.zeros_like(a)))

    def run(dtype, hook):
        torch.manual_seed(3)
        critic = pg.Critic(pg.ProtBertForMaskedLM(cfg, device=DEV, compute_dtype=dtype).init_random(seed=2))
        T.HEAD_STASH_HOOK = hook
        try:
            loss = T.critic_update(critic, _NoStep(critic), real["input_ids"], real["attention_mask"], fake,
                                   T.GanHyper(lambda_gp=5.0), alpha=alpha, real_lengths=real["lengths"])
            torch.cuda.synchronize()
        finally:
            T.HEAD_STASH_HOOK = None
        return float(loss), _grads(critic)

    l0, g0 = run(torch.float32, record)
    l1, g1 = run(BF, None)
    l2, g2 = run(BF, impose)
    typical = sorted(float(v.norm()) for v in g0.values())[len(g0) // 2]
    names = [n for n in g0 if float(g0[n].norm()) > 0.05 * typical and "key.bias" not in n]
    free = {n: rel_l2(g1[n].numpy(), g0[n].numpy()) for n in names}
    held = {n: rel_l2(g2[n].numpy(), g0[n].numpy()) for n in names}
    med = lambda dct: float(np.median(list(dct.values())))  # noqa: E731
    print(f"gates free: median {med(free):.3f} worst {max(free.values()):.3f}; fp32 gates imposed ({gates['flipped']} of "
          f"{gates['fp32'].numel()} units differed): median {med(held):.3f} worst {max(held.values()):.3f}; "
          f"losses {l0:.5f} {l1:.5f} {l2:.5f}")
    assert med(held) < 3e-2 and max(held.values()) < 1e-1, (med(held), max(held.items(), key=lambda kv: kv[1]))
    if gates["flipped"]:
        assert med(held) < med(free)
