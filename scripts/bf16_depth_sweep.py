"""Relative L2 difference between the bf16 and fp32 engines on the gradient of a penalty-dominated critic
loss (lambda_gp = 500), against encoder depth."""
import sys, torch
sys.path.insert(0, ".")
import pgload
pg = pgload.load()
DEV = "cuda"
g = torch.Generator().manual_seed(11)
B, S = 2, 24
real = torch.randint(5, 25, (B, S), generator=g).to(DEV)
fake = torch.randint(5, 25, (B, S), generator=g).to(DEV)
real[:, 0] = fake[:, 0] = 2
rm = torch.ones(B, S, device=DEV)
alpha = torch.rand(B, 1, 1, generator=g)
OPEN = len(sys.argv) > 1 and sys.argv[1] == "open"  # biases that keep every ReLU gate of the head open
for L in (1, 2, 4, 8, 16, 30):
    out = {}
    for dt in (torch.float32, torch.bfloat16):
        cfg = pg.ProtBertConfig(num_hidden_layers=L, max_position_embeddings=64)
        critic = pg.Critic(pg.ProtBertForMaskedLM(cfg, device=DEV, compute_dtype=dt).init_random(seed=1))
        torch.manual_seed(2)
        for p in critic.classifier.parameters():
            torch.nn.init.normal_(p, std=0.05)
        if OPEN:
            with torch.no_grad():
                for k in (0, 2, 4):
                    critic.classifier[k].bias.fill_(10.0)
        critic.zero_grad(set_to_none=True)
        gp = pg.compute_gradient_penalty(critic, real, fake, rm, rm, DEV, alpha=alpha)
        (500.0 * gp).backward()
        out[dt] = (float(gp), {n: p.grad.double().cpu() for n, p in critic.named_parameters() if p.grad is not None})
    (g32, a), (g16, b) = out[torch.float32], out[torch.bfloat16]
    names = ["protbert.bert.encoder.layer.0.attention.self.query.weight", "protbert.bert.encoder.layer.0.intermediate.dense.weight",
             f"protbert.bert.encoder.layer.{L-1}.output.dense.weight", "classifier.0.weight"]
    errs = [float((a[n] - b[n]).norm() / a[n].norm()) for n in names]
    print(f"L={L:2d} gp32={g32:.5f} gp16={g16:.5f} " + " ".join(f"{e:.4f}" for e in errs), flush=True)
