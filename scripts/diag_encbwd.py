import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgload
pg = pgload.load()
from oracle import protbert_cpu as O
from tests.test_gpu_model import build, CONFIGS
DEV = "cuda"
name = "tiny"; cfg = CONFIGS[name]
G = np.load(os.path.join(ROOT, "tests/golden", f"ref_{name}.npz"))
PC = O.init_params(cfg, seed=102, bias_std=0.02)
real = torch.from_numpy(G["real"]); rm = torch.from_numpy(G["real_mask"])
g = torch.Generator().manual_seed(5)
R = torch.randn(real.shape[0], real.shape[1], cfg.hidden_size, generator=g)
Pl = {k: v.clone().requires_grad_(True) for k, v in PC.items()}
h = O.encoder(Pl, cfg, O.embeddings(Pl, cfg, real), rm)
(h * R).sum().backward()
for dtype in (torch.float32, torch.bfloat16):
    _, _, _, _, _, critic = build(pg, name, dtype)
    m = critic.protbert
    out = m(input_ids=real.to(DEV), attention_mask=rm.to(DEV), output_hidden_states=True).hidden_states[-1]
    print(dtype, "hidden relL2", float((out.float().cpu() - h.detach()).norm() / h.detach().norm()))
    (out.float() * R.to(DEV)).sum().backward()
    named = dict(m.named_parameters())
    for n in ["bert.embeddings.word_embeddings.weight", "bert.embeddings.LayerNorm.weight", "bert.encoder.layer.0.attention.self.query.weight",
              "bert.encoder.layer.0.attention.self.key.weight", "bert.encoder.layer.0.attention.self.value.weight",
              "bert.encoder.layer.0.attention.output.dense.weight", "bert.encoder.layer.0.intermediate.dense.weight",
              "bert.encoder.layer.0.output.dense.weight", "bert.encoder.layer.1.output.dense.weight", "bert.encoder.layer.1.output.LayerNorm.weight",
              "bert.encoder.layer.1.output.dense.bias", "bert.encoder.layer.1.attention.self.query.bias"]:
        a = named[n].grad.double().cpu(); b = Pl[n].grad.double()
        print(f"   {n:60s} relL2 {float((a-b).norm()/b.norm()):.3e} |ref| {float(b.norm()):.3e}")
