"""The differentiable generator update (docs/HISTORY.md:169-179: softmax of the generator's logits, averaged critic
word embeddings, the critic's 3-D branch of models.py:94-102) on the HIP engine against torch autograd over the CPU
oracle.  The reference's trainers do not contain this update yet (the history file prescribes it), so no run of the
reference pins the numbers: the oracle restates the snippet."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import protbert_cpu as O  # noqa: E402
from tests.test_gpu_model import CONFIGS, DEV, build, rel_l2  # noqa: E402


@pytest.fixture(scope="module")
def pg():
    import pgload
    return pgload.load()


def _template(B, S, seed):
    g = torch.Generator().manual_seed(seed)
    ids = torch.full((B, S), 4, dtype=torch.long)
    ids[:, 0] = 2
    for b in range(B):
        L = int(torch.randint(S // 2, S + 1, (1,), generator=g))
        ids[b, L - 1] = 3
        ids[b, L:] = 0
        keep = torch.rand(L, generator=g) < 0.3  # some positions already filled
        fill = torch.randint(5, 25, (L,), generator=g)
        for j in range(1, L - 1):
            if keep[j]:
                ids[b, j] = fill[j]
    return ids


@pytest.mark.parametrize("name,dtype,bound", [("tiny", torch.float32, 2e-4), ("mid", torch.float32, 2e-4),
                                              ("mid", torch.bfloat16, 6e-2)])
def test_mlm_head_backward_vs_autograd(pg, name, dtype, bound):
    """logits.backward(dlogits) through the engine's MLM head: gradient of the hidden state and of every head
    parameter (dense, LayerNorm, decoder bias, tied word table) against autograd on the oracle's head."""
    c, P, PC, C, gen, critic = build(pg, name, dtype)
    m = gen.protbert
    g = torch.Generator().manual_seed(3)
    B, S = 3, 40
    hidden = torch.randn(B, S, c.hidden_size, generator=g)
    dlogits = torch.randn(B, S, c.vocab_size, generator=g)
    Pl = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    h_ref = hidden.to(dtype).float().detach().requires_grad_(True)
    O.mlm_head(Pl, c, h_ref).backward(dlogits)

    h = hidden.to(DEV).to(dtype).detach().requires_grad_(True)
    for p in m.parameters():
        p.grad = None
    logits = m.lm_logits(h)
    assert logits.requires_grad
    logits.backward(dlogits.to(DEV))
    torch.cuda.synchronize()
    assert rel_l2(h.grad.float().cpu().numpy(), h_ref.grad.numpy()) < bound
    got = {n: p.grad for n, p in m.named_parameters() if p.grad is not None}
    want = {k: v.grad for k, v in Pl.items() if v.grad is not None}
    # (a backward attaches the whole flat gradient buffer: parameters this loss does not reach read zero)
    assert set(want) <= set(got), set(want) - set(got)
    for k in want:
        assert rel_l2(got[k].cpu().numpy(), want[k].numpy()) < bound, k
    for k in set(got) - set(want):
        assert not got[k].any(), k


@pytest.mark.parametrize("name,dtype,bound", [("tiny", torch.float32, 2e-3), ("mid", torch.float32, 2e-3),
                                              ("mid", torch.bfloat16, 1.5e-1)])
def test_generator_soft_update_gradients_vs_oracle(pg, name, dtype, bound):
    """One soft generator update with an optimizer that does not step: loss and every generator gradient against the
    oracle; the set of parameters with a gradient is the same; the critic collects none."""
    from protgen_gan_amd import train as T
    c, P, PC, C, gen, critic = build(pg, name, dtype)
    ids = _template(4, 48, 5)
    gmask = (ids != 4).long()   # blind-mode visibility
    cmask = (ids != 0).long()
    temperature = 0.9

    class NoStep:
        def zero_grad(self, set_to_none=True):
            for p in gen.parameters():
                p.grad = None

        def step(self):
            pass

    for p in critic.parameters():
        p.grad = None
    loss = T.generator_update_soft(gen, critic, NoStep(), ids.to(DEV), gmask.to(DEV), temperature=temperature,
                                   critic_attention_mask=cmask.to(DEV))
    torch.cuda.synchronize()
    l_ref, g_ref = O.generator_soft_step_grads(P, PC, C, c, ids, gmask, temperature, critic_mask=cmask)
    assert abs(float(loss) - float(l_ref)) <= (1e-4 if dtype == torch.float32 else 2e-2) * abs(float(l_ref)) + 1e-5
    got = {n[len("protbert."):]: p.grad for n, p in gen.named_parameters() if p.grad is not None}
    assert set(got) == set(g_ref), set(got) ^ set(g_ref)
    typical = float(np.median([float(v.norm()) for v in g_ref.values()]))
    worst = ("", 0.0)
    for k, v in g_ref.items():
        if float(v.norm()) < 1e-3 * typical:
            continue
        d = rel_l2(got[k].cpu().numpy(), v.numpy())
        if d > worst[1]:
            worst = (k, d)
    print("soft update", name, dtype, "loss", float(loss), float(l_ref), "worst", worst)
    assert worst[1] < bound, worst
    assert all(p.grad is None for p in critic.parameters())
    assert all(p.requires_grad for p in critic.parameters())


def test_generator_soft_update_moves_the_generator_only(pg):
    """With AdamW: generator weights change (encoder, embeddings and MLM head), the critic's do not; with
    critic_param_grads the critic collects gradients as the snippet's autograd would, and the generator's
    gradients are the same either way."""
    from protgen_gan_amd import train as T
    c, P, PC, C, gen, critic = build(pg, "tiny", torch.float32)
    ids = _template(2, 32, 9).to(DEV)
    mask = (ids != 0).long()
    before_g = gen.protbert.store.master.clone()
    before_c = critic.protbert.store.master.clone()
    opt = pg.AdamW(gen.parameters(), lr=1e-3)
    T.generator_update_soft(gen, critic, opt, ids, mask)
    torch.cuda.synchronize()
    assert torch.equal(critic.protbert.store.master, before_c)
    for n in ("bert.encoder.layer.0.attention.self.query.weight", "bert.embeddings.word_embeddings.weight",
              "cls.predictions.transform.dense.weight", "cls.predictions.bias"):
        p = gen.protbert.get_parameter(n)
        off = gen.protbert._offsets[n]
        assert not torch.equal(p.detach().flatten(), before_g[off:off + p.numel()]), n
    g0 = {n: p.grad.clone() for n, p in gen.named_parameters() if p.grad is not None}

    class NoStep:
        def zero_grad(self, set_to_none=True):
            for p in gen.parameters():
                p.grad = None

        def step(self):
            pass

    gen2 = pg.Generator(pg.ProtBertForMaskedLM.from_params(gen.protbert.config, P, device=DEV, compute_dtype=torch.float32))
    T.generator_update_soft(gen2, critic, NoStep(), ids, mask, critic_param_grads=True)
    torch.cuda.synchronize()
    assert any(p.grad is not None for p in critic.parameters())
    for n, p in gen2.named_parameters():
        if n in g0:
            assert rel_l2(p.grad.cpu().numpy(), g0[n].cpu().numpy()) < 1e-5, n


@pytest.mark.parametrize("mode", ["blind", "seeded"])
def test_trainer_with_soft_generator_update(pg, tmp_path, mode):
    """train_cli --soft_generator_update, two epochs on a small random model: the frozen first epoch leaves the
    generator bit-unchanged, the second one moves its encoder and its MLM head; the generator's optimizer state is
    saved next to the critic's and a resumed run loads it."""
    import os
    import random
    from protgen_gan_amd import train_cli as TC
    rng = random.Random(5)
    data = tmp_path / "d.txt"
    data.write_text("\n".join("".join(rng.choice("ACDEFGHIKLMNPQRSTVWY") for _ in range(rng.randint(20, 44))) for _ in range(36)) + "\n")
    lens = tmp_path / "l.txt"
    lens.write_text("\n".join("A" * rng.randint(20, 44) for _ in range(30)) + "\n")
    ck = tmp_path / "ck"
    argv = ["--mode", mode, "--init_random", "--layers", "2", "--data_file", str(data), "--lengths_file", str(lens),
            "--checkpoint_dir", str(ck), "--run_name", "s", "--batch_size", "2", "--n_critic", "2", "--n_epochs", "2",
            "--num_eval_sequences", "0", "--max_len", "64", "--dtype", "bf16", "--lr_gen", "1e-3", "--soft_generator_update"]
    hist = TC.main(argv)
    assert len(hist) == 2
    e1, e2 = ck / "s" / "epoch_1", ck / "s" / "epoch_2"
    assert os.path.exists(e2 / "generator_optimizer.pt")
    init = pg.ProtBertForMaskedLM(pg.ProtBertConfig(num_hidden_layers=2), device=DEV, compute_dtype=torch.bfloat16).init_random(seed=0)
    g1 = pg.ProtBertForMaskedLM.from_pretrained(str(e1 / "generator_bert"), device=DEV)
    g2 = pg.ProtBertForMaskedLM.from_pretrained(str(e2 / "generator_bert"), device=DEV)
    for (n, a), (_, b) in zip(init.named_parameters(), g1.named_parameters()):
        assert torch.equal(a, b), n
    for n in ("bert.encoder.layer.1.output.dense.weight", "bert.embeddings.word_embeddings.weight",
              "cls.predictions.transform.dense.weight"):
        assert not torch.equal(g2.get_parameter(n), init.get_parameter(n)), n
    hist = TC.main(argv + ["--resume", str(e2), "--run_name", "s2", "--n_epochs", "3"])
    assert len(hist) == 1 and os.path.exists(ck / "s2" / "epoch_3" / "generator_optimizer.pt")
