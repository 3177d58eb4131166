"""Diagnostic: per-tensor gradient error of the bf16 critic step vs the reference fixtures."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgload
pg = pgload.load()
from oracle import protbert_cpu as O
from tests.test_gpu_model import build, CONFIGS
DEV = "cuda"
for name in ("tiny", "mid"):
    G = np.load(os.path.join(ROOT, "tests/golden", f"ref_{name}.npz"))
    for dtype in (torch.float32, torch.bfloat16):
        for regime in ("unfrozen", "frozen"):
            _, _, _, _, _, critic = build(pg, name, dtype)
            for p in critic.protbert.parameters():
                p.requires_grad = regime == "unfrozen"
            real, fake = torch.from_numpy(G["real"]).to(DEV), torch.from_numpy(G["fake"]).to(DEV)
            rm, fm = torch.from_numpy(G["real_mask"]).to(DEV), torch.from_numpy(G["fake_mask"]).to(DEV)
            alpha = torch.from_numpy(G[f"{regime}_alpha"])
            gp = pg.compute_gradient_penalty(critic, real, fake, rm, fm, DEV, alpha=alpha)
            rs = critic(real, attention_mask=rm); fs = critic(fake, attention_mask=fm)
            loss = pg.critic_loss(rs, fs, gp, 5.0); loss.backward()
            print(f"== {name} {dtype} {regime}: gp {gp.item():.6f} ref {float(G[regime+'_gp']):.6f} loss {loss.item():.6f} ref {float(G[regime+'_c_loss']):.6f}")
            named = dict(critic.named_parameters())
            num = den = dot = n1 = n2 = 0.0
            for key in G.files:
                if key.startswith(f"{regime}_grad::"):
                    n = key.split("::", 1)[1]
                    a = named[n].grad.double().cpu().numpy(); b = G[key].astype(np.float64)
                    l2 = np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)
                    mx = np.abs(a - b).max() / (np.abs(b).max() + 1e-30)
                    dot += (a * b).sum(); n1 += (a * a).sum(); n2 += (b * b).sum()
                    if dtype == torch.bfloat16:
                        print(f"   {n:70s} relL2 {l2:.3e} relmax {mx:.3e} |ref| {np.abs(b).max():.3e}")
            print(f"   cosine over listed tensors: {dot / np.sqrt(n1 * n2):.6f}")
