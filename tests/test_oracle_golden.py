"""The CPU oracle must reproduce what the reference itself produced (tests/golden/ref_*.npz,
written by oracle/make_golden.py from /root/reference/{models,loss,generate}.py over HF BERT)."""
import os

import numpy as np
import pytest
import torch

from oracle import protbert_cpu as O

CONFIGS = {
    "tiny": O.BertCfg(hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                      intermediate_size=512, max_position_embeddings=640),
    "mid": O.BertCfg(hidden_size=256, num_hidden_layers=4, num_attention_heads=4,
                     intermediate_size=1024, max_position_embeddings=640),
}


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, f"ref_{name}.npz"), allow_pickle=False)


@pytest.fixture(scope="module", params=["tiny", "mid"])
def setup(request, golden_dir):
    name = request.param
    cfg = CONFIGS[name]
    G = load(golden_dir, name)
    P = O.init_params(cfg, seed=100, bias_std=0.02)
    C = O.init_classifier(cfg.hidden_size, seed=101)
    PC = O.init_params(cfg, seed=102, bias_std=0.02)
    chk = [float(sum(v.double().sum() for v in d.values())) for d in (P, PC, C)]
    np.testing.assert_allclose(chk, G["param_checksum"], rtol=1e-12)
    return name, cfg, G, P, PC, C


def test_logits(setup):
    _, cfg, G, P, _, _ = setup
    ids = torch.from_numpy(G["fwd_ids"])
    mask = torch.from_numpy(G["fwd_mask"])
    with torch.no_grad():
        a = O.generator_forward(P, cfg, ids, mask).numpy()
        b = O.generator_forward(P, cfg, ids, None).numpy()
    np.testing.assert_allclose(a, G["fwd_logits"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(b, G["fwd_logits_nomask"], rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("kind", ["gen", "blind"])
def test_generate_teacher_forced(setup, kind):
    """Every iteration's decisions, fed the reference's own input ids of that iteration."""
    _, cfg, G, P, _, _ = setup
    traj = G[f"{kind}_traj"]
    noise = G[f"{kind}_noise"]
    rates = G[f"{kind}_rates"]
    for it in range(len(rates)):
        ids = torch.from_numpy(traj[it]).clone()
        if kind == "gen":
            mask = torch.from_numpy(G["gen_attn"])
            temp = float(G["gen_temps"][it])
        else:
            mask = (ids != O.MASK).long()
            temp = 1.1
        with torch.no_grad():
            out = O.generate_step(P, cfg, ids, mask, temp, 0.1, float(rates[it]),
                                  torch.from_numpy(noise[it]))
        assert np.array_equal(out.numpy(), traj[it + 1]), f"{kind} iteration {it}"


def test_generate_free_running(setup):
    _, cfg, G, P, _, _ = setup
    traj = G["gen_traj"]
    ids = torch.from_numpy(traj[0]).clone()
    mask = torch.from_numpy(G["gen_attn"])
    for it in range(len(G["gen_rates"])):
        with torch.no_grad():
            ids = O.generate_step(P, cfg, ids, mask, float(G["gen_temps"][it]), 0.1,
                                  float(G["gen_rates"][it]), torch.from_numpy(G["gen_noise"][it]))
    assert np.array_equal(ids.numpy(), traj[-1])


def test_critic_scores(setup):
    _, cfg, G, _, PC, C = setup
    real = torch.from_numpy(G["real"])
    rm = torch.from_numpy(G["real_mask"])
    with torch.no_grad():
        s = O.critic_forward(PC, C, cfg, real, rm).numpy()
        e = O.embeddings(PC, cfg, real)
        np.testing.assert_allclose(e.numpy(), G["critic_emb_in"], rtol=1e-5, atol=1e-6)
        s3 = O.critic_forward(PC, C, cfg, torch.from_numpy(G["critic_emb_in"]), rm).numpy()
    np.testing.assert_allclose(s, G["critic_real_scores"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(s3, G["critic_emb_scores"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("regime", ["unfrozen", "frozen"])
def test_critic_step(setup, regime):
    _, cfg, G, _, PC, C = setup
    real, fake = torch.from_numpy(G["real"]), torch.from_numpy(G["fake"])
    rm, fm = torch.from_numpy(G["real_mask"]), torch.from_numpy(G["fake_mask"])
    alpha = torch.from_numpy(G[f"{regime}_alpha"])
    loss, gp, rs, fs, grads = O.critic_step_grads(PC, C, cfg, real, fake, rm, fm, alpha, 5.0,
                                                  train_encoder=(regime == "unfrozen"))
    np.testing.assert_allclose(gp.numpy(), G[f"{regime}_gp"], rtol=2e-4)
    np.testing.assert_allclose(loss.numpy(), G[f"{regime}_c_loss"], rtol=2e-4)
    np.testing.assert_allclose(rs.numpy(), G[f"{regime}_real_scores"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(fs.numpy(), G[f"{regime}_fake_scores"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(O.generator_loss(fs).numpy(), G[f"{regime}_g_loss"], rtol=1e-4)
    # same set of parameters without gradient as the reference
    ref_names = set(G[f"{regime}_grad_names"].tolist())
    assert set(grads.keys()) == ref_names
    for key in G.files:
        if key.startswith(f"{regime}_grad::"):
            n = key.split("::", 1)[1]
            ref = G[key]
            got = grads[n].numpy()
            scale = np.abs(ref).max() + 1e-12
            assert np.abs(got - ref).max() <= 2e-3 * scale + 1e-7, n
    gsq = sum(float((g.double() ** 2).sum()) for g in grads.values())
    np.testing.assert_allclose(gsq, float(G[f"{regime}_grad_sqnorm"]), rtol=2e-3)


def test_generate_py_loops(golden_dir):
    G = np.load(os.path.join(golden_dir, "ref_generate_py.npz"), allow_pickle=False)
    cfg = CONFIGS["tiny"]
    P = O.init_params(cfg, seed=100, bias_std=0.02)
    tok = O.ProtTokenizer()
    assert tok.encode(" ".join(list("ACDXUBZOMK"))) == G["enc_probe"].tolist()
    assert tok.decode(torch.tensor([2, 5, 4, 25, 0, 3, 1, 9])) == str(G["dec_probe"])
    pool = G["pool"].tolist()
    seeds = G["seeds"].tolist()
    with torch.no_grad():
        torch.manual_seed(555)
        full = O.generate_full_mode_sequences(P, cfg, tok, 5, pool, batch_size=3, rng_seed=42)
        torch.manual_seed(556)
        seeded = O.generate_seeded_mode_sequences(P, cfg, tok, 5, seeds, batch_size=3, rng_seed=123)
    assert full == G["full"].tolist()
    assert seeded == G["seeded"].tolist()


def test_adamw_matches_torch():
    torch.manual_seed(0)
    p0 = torch.randn(257)
    g = [torch.randn(257) for _ in range(3)]
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=3e-3, betas=(0.9, 0.999), weight_decay=0.01)
    p, m, v = p0.clone(), torch.zeros(257), torch.zeros(257)
    for t, gi in enumerate(g, 1):
        ref.grad = gi.clone()
        opt.step()
        O.adamw_step(p, gi, m, v, t, 3e-3)
    np.testing.assert_allclose(p.numpy(), ref.detach().numpy(), rtol=1e-6, atol=1e-7)
