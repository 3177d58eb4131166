"""Diagnostic: per-term (real score / fake score / gp) gradient error, fp32 and bf16."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgload
pg = pgload.load()
from oracle import protbert_cpu as O
from tests.test_gpu_model import build, CONFIGS
DEV = "cuda"
name = sys.argv[1] if len(sys.argv) > 1 else "tiny"
cfg = CONFIGS[name]
G = np.load(os.path.join(ROOT, "tests/golden", f"ref_{name}.npz"))
PC = O.init_params(cfg, seed=102, bias_std=0.02); C = O.init_classifier(cfg.hidden_size, seed=101)
real, fake = torch.from_numpy(G["real"]), torch.from_numpy(G["fake"])
rm, fm = torch.from_numpy(G["real_mask"]), torch.from_numpy(G["fake_mask"])
alpha = torch.from_numpy(G["unfrozen_alpha"])
def oracle_term(which):
    Pl = {k: v.clone().requires_grad_(True) for k, v in PC.items()}
    Cl = {k: v.clone().requires_grad_(True) for k, v in C.items()}
    if which == "real": t = -O.critic_forward(Pl, Cl, cfg, real, rm).mean()
    elif which == "fake": t = O.critic_forward(Pl, Cl, cfg, fake, fm).mean()
    else: t = 5 * O.compute_gradient_penalty(Pl, Cl, cfg, real, fake, rm, fm, alpha)
    t.backward()
    out = {"protbert." + k: v.grad for k, v in Pl.items() if v.grad is not None}
    out.update({"classifier." + k: v.grad for k, v in Cl.items() if v.grad is not None})
    return out
sel = ["classifier.0.weight", "classifier.4.weight", "protbert.bert.embeddings.word_embeddings.weight",
       "protbert.bert.embeddings.position_embeddings.weight", "protbert.bert.embeddings.LayerNorm.weight",
       "protbert.bert.encoder.layer.0.attention.self.query.weight", "protbert.bert.encoder.layer.0.attention.self.value.weight",
       "protbert.bert.encoder.layer.0.intermediate.dense.weight", "protbert.bert.encoder.layer.1.output.dense.weight",
       "protbert.bert.encoder.layer.1.output.LayerNorm.weight", "protbert.bert.encoder.layer.1.attention.output.dense.bias"]
for which in ("real", "fake", "gp"):
    ref = oracle_term(which)
    for dtype in (torch.float32, torch.bfloat16):
        _, _, _, _, _, critic = build(pg, name, dtype)
        d = lambda t: t.to(DEV)
        if which == "real": t = -critic(d(real), attention_mask=d(rm)).mean()
        elif which == "fake": t = critic(d(fake), attention_mask=d(fm)).mean()
        else: t = 5 * pg.compute_gradient_penalty(critic, d(real), d(fake), d(rm), d(fm), DEV, alpha=alpha)
        t.backward()
        named = dict(critic.named_parameters())
        print(f"== term {which} {dtype}")
        for n in sel:
            a = named[n].grad.double().cpu(); b = ref[n].double()
            print(f"   {n:66s} relL2 {float((a-b).norm()/b.norm()):.3e}  |ref| {float(b.norm()):.3e}")
