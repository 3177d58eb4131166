#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (dev container only).

Imports /root/reference/{models,loss,generate}.py unmodified and drives them over
`transformers.BertForMaskedLM` (eager attention, eval mode) with seeded synthetic
weights, then records inputs and outputs as plain arrays.  Nothing from the reference
is copied into the repository: the .npz files hold data only.  The GPU box never runs
this script (it has neither /root/reference nor needs transformers).

    SOURCE_DIR=/tmp python oracle/make_golden.py
"""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SOURCE_DIR", "/tmp/protgen_golden_src")
sys.path.insert(0, "/root/reference")

from transformers import AutoTokenizer, BertConfig, BertForMaskedLM  # noqa: E402

import generate as ref_generate  # noqa: E402  (reference)
import loss as ref_loss  # noqa: E402  (reference)
import models as ref_models  # noqa: E402  (reference)
from oracle import protbert_cpu as O  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

CONFIGS = {
    "tiny": O.BertCfg(hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                      intermediate_size=512, max_position_embeddings=640),
    "mid": O.BertCfg(hidden_size=256, num_hidden_layers=4, num_attention_heads=4,
                     intermediate_size=1024, max_position_embeddings=640),
}


def hf_model(cfg: O.BertCfg, params, dtype):
    hc = BertConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size,
                    num_hidden_layers=cfg.num_hidden_layers,
                    num_attention_heads=cfg.num_attention_heads,
                    intermediate_size=cfg.intermediate_size,
                    max_position_embeddings=cfg.max_position_embeddings,
                    type_vocab_size=cfg.type_vocab_size, hidden_dropout_prob=0.0,
                    attention_probs_dropout_prob=0.0, layer_norm_eps=cfg.layer_norm_eps,
                    attn_implementation="eager")
    m = BertForMaskedLM(hc)
    sd = {k: v.clone() for k, v in params.items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(("decoder" in k) or ("position_ids" in k) for k in missing), missing
    m.tie_weights()
    return m.to(dtype).eval()


def rand_protein_batch(g, B, S, min_len):
    ids = torch.full((B, S), O.PAD, dtype=torch.long)
    mask = torch.zeros(B, S, dtype=torch.long)
    for b in range(B):
        L = int(torch.randint(min_len, S - 1, (1,), generator=g))
        ids[b, 0] = O.CLS
        ids[b, 1:L + 1] = torch.randint(5, 25, (L,), generator=g)
        ids[b, L + 1] = O.SEP
        mask[b, :L + 2] = 1
    return ids, mask


def run_config(name, cfg, B, S):
    out = {}
    P = O.init_params(cfg, seed=100, bias_std=0.02)
    C = O.init_classifier(cfg.hidden_size, seed=101)
    gen_model = hf_model(cfg, P, torch.float32)
    generator = ref_models.Generator(gen_model)
    PC = O.init_params(cfg, seed=102, bias_std=0.02)
    critic = ref_models.Critic(hf_model(cfg, PC, torch.float32))
    critic.classifier.load_state_dict(C)
    critic.eval()

    g = torch.Generator().manual_seed(7)
    ids, mask = rand_protein_batch(g, B, S, S // 2)
    out["fwd_ids"], out["fwd_mask"] = ids.numpy(), mask.numpy()
    with torch.no_grad():
        out["fwd_logits"] = generator(ids, mask).numpy()
        out["fwd_logits_nomask"] = generator(ids, None).numpy()

    # --- Generator.generate trajectories (teacher-forcing data: every iteration's input and output)
    lengths = [S - 2, S - 7, S // 2, S // 3][:B]
    t_ids, t_attn = ref_generate.build_fully_masked_batch(lengths, O.CLS, O.SEP, O.MASK, O.PAD, "cpu")
    out["gen_lengths"] = np.array(lengths)
    traj, noises, rates, temps = [t_ids.clone().numpy()], [], [], []
    rate = 1.0
    cur = t_ids.clone()
    for it in range(11):
        if int((cur == O.MASK).sum()) == 0:
            break
        temp = 1.0 if it % 2 == 0 else 0.9 + 0.05 * it
        torch.manual_seed(1000 + it)
        noise = torch.empty(cur.numel(), cfg.vocab_size).exponential_(1)
        torch.manual_seed(1000 + it)
        with torch.no_grad():
            cur = generator.generate(cur, attention_mask=t_attn, temperature=temp,
                                     keep_percent=0.1, current_rate=rate)
        traj.append(cur.clone().numpy())
        noises.append(noise.numpy())
        rates.append(rate)
        temps.append(temp)
        rate = max(0.0, rate - 0.1)
    out["gen_attn"] = t_attn.numpy()
    out["gen_traj"] = np.stack(traj)
    out["gen_noise"] = np.stack(noises)
    out["gen_rates"] = np.array(rates, dtype=np.float64)
    out["gen_temps"] = np.array(temps, dtype=np.float64)

    # blind-trainer style: attention mask = ids != MASK recomputed per iteration, rate from 0.9
    cur = t_ids.clone()
    traj2, noises2, rates2 = [cur.clone().numpy()], [], []
    rate = 0.9
    it = 0
    while rate > 0:
        am = (cur != O.MASK).long()
        torch.manual_seed(2000 + it)
        noise = torch.empty(cur.numel(), cfg.vocab_size).exponential_(1)
        torch.manual_seed(2000 + it)
        with torch.no_grad():
            cur = generator.generate(cur, am, temperature=1.1, keep_percent=0.1, current_rate=rate)
        traj2.append(cur.clone().numpy())
        noises2.append(noise.numpy())
        rates2.append(rate)
        rate = max(0, rate - 0.1)
        it += 1
        if int((cur == O.MASK).sum()) == 0:
            break
    out["blind_traj"] = np.stack(traj2)
    out["blind_noise"] = np.stack(noises2)
    out["blind_rates"] = np.array(rates2, dtype=np.float64)

    # --- critic scores, GP, loss, gradients
    real, real_mask = rand_protein_batch(g, B, S, S // 2)
    fake = torch.from_numpy(traj[-1]).clone()
    fake_mask = (fake != O.PAD).long()
    out["real"], out["real_mask"], out["fake"], out["fake_mask"] = (
        real.numpy(), real_mask.numpy(), fake.numpy(), fake_mask.numpy())
    with torch.no_grad():
        out["critic_real_scores"] = critic(real, real_mask).numpy()
        emb = critic.protbert.bert.embeddings(input_ids=real)
        out["critic_emb_in"] = emb.numpy()
        out["critic_emb_scores"] = critic(emb, real_mask).numpy()

    for regime in ("unfrozen", "frozen"):
        for p in critic.protbert.parameters():
            p.requires_grad = regime == "unfrozen"
        critic.zero_grad(set_to_none=True)
        torch.manual_seed(4242)
        alpha = torch.rand(B, 1, 1)
        torch.manual_seed(4242)
        gp = ref_loss.compute_gradient_penalty(critic, real, fake, real_mask, fake_mask, "cpu")
        rs = critic(real, attention_mask=real_mask)
        fs = critic(fake, attention_mask=fake_mask)
        c_loss = ref_loss.critic_loss(rs, fs, gp, 5.0)
        c_loss.backward()
        out[f"{regime}_alpha"] = alpha.numpy()
        out[f"{regime}_gp"] = gp.detach().numpy()
        out[f"{regime}_c_loss"] = c_loss.detach().numpy()
        out[f"{regime}_real_scores"] = rs.detach().numpy()
        out[f"{regime}_fake_scores"] = fs.detach().numpy()
        out[f"{regime}_g_loss"] = ref_loss.generator_loss(fs).detach().numpy()
        names, none_names = [], []
        for n, p in critic.named_parameters():
            if p.grad is None:
                none_names.append(n)
            else:
                names.append(n)
        out[f"{regime}_grad_none"] = np.array(none_names)
        keep = [n for n in names if n.startswith("classifier")]
        L = cfg.num_hidden_layers
        if name == "tiny":
            sel = ("embeddings.", "layer.0.", f"layer.{L - 1}.")
        else:
            sel = ("embeddings.word_embeddings", "embeddings.LayerNorm", "layer.0.attention.self.query",
                   "layer.0.intermediate.dense.bias", f"layer.{L - 1}.attention.self.value.bias",
                   f"layer.{L - 1}.output.LayerNorm", "layer.1.attention.output.dense.bias",
                   "layer.2.attention.self.key.weight")
        for n in names:
            if any(s in n for s in sel):
                keep.append(n)
        for n in dict.fromkeys(keep):
            out[f"{regime}_grad::{n}"] = dict(critic.named_parameters())[n].grad.numpy().copy()
        out[f"{regime}_grad_names"] = np.array(names)
        gsq = sum(float((p.grad.double() ** 2).sum()) for p in critic.parameters() if p.grad is not None)
        out[f"{regime}_grad_sqnorm"] = np.array(gsq)
    out["param_checksum"] = np.array([float(sum(v.double().sum() for v in P.values())),
                                      float(sum(v.double().sum() for v in PC.values())),
                                      float(sum(v.double().sum() for v in C.values()))])
    np.savez_compressed(os.path.join(GOLD, f"ref_{name}.npz"), **out)
    print(name, "saved", len(out), "arrays")


def run_generate_py():
    """generate.py's two loops end to end with the tiny model and a local vocab.txt."""
    cfg = CONFIGS["tiny"]
    P = O.init_params(cfg, seed=100, bias_std=0.02)
    generator = ref_models.Generator(hf_model(cfg, P, torch.float32))
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "vocab.txt"), "w") as f:
            f.write("\n".join(O.VOCAB) + "\n")
        with open(os.path.join(d, "tokenizer_config.json"), "w") as f:
            f.write('{"do_lower_case": false, "tokenizer_class": "BertTokenizer"}')
        tok = AutoTokenizer.from_pretrained(d, do_lower_case=False, local_files_only=True)
    assert (tok.pad_token_id, tok.cls_token_id, tok.sep_token_id, tok.mask_token_id) == (0, 2, 3, 4)
    pool = [23, 30, 37, 41, 52]
    seeds = ["MKVLAAGIVGLLLAQPSRAEDTPKGGSW", "ACDEFGHIKLMNPQRSTVWYACDEFGHIKLMNPQRSTVWY",
             "GGSGGSLVPRGSHMASMTGGQQMGRDLYDDDDK"]
    torch.manual_seed(555)
    full = ref_generate.generate_full_mode_sequences(generator, tok, 5, pool, keep_percent=0.1,
                                                     temperature=1.0, batch_size=3, device="cpu",
                                                     rng_seed=42)
    torch.manual_seed(556)
    seeded = ref_generate.generate_seeded_mode_sequences(generator, tok, 5, seeds, keep_init=0.1,
                                                         keep_percent=0.1, temperature=1.0,
                                                         batch_size=3, device="cpu", rng_seed=123)
    enc = np.array(tok.encode(" ".join(list("ACDXUBZOMK")), add_special_tokens=False))
    np.savez_compressed(os.path.join(GOLD, "ref_generate_py.npz"),
                        pool=np.array(pool), seeds=np.array(seeds), full=np.array(full),
                        seeded=np.array(seeded), enc_probe=enc,
                        dec_probe=np.array(tok.decode(torch.tensor([2, 5, 4, 25, 0, 3, 1, 9]),
                                                      skip_special_tokens=True)))
    print("generate.py golden:", full[:2], seeded[:2])


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    run_config("tiny", CONFIGS["tiny"], B=3, S=40)
    run_config("mid", CONFIGS["mid"], B=2, S=64)
    run_generate_py()
