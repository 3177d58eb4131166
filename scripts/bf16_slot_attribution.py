"""Which stored quantity does a bf16 run owe its distance from fp32 to?  The gradient of a penalty-dominated critic
loss (lambda_gp = 500, every ReLU gate of the head held open) from the fp32 engine, from the bf16 engine, and from
the fp32 engine with ONE class of stored values rounded to bf16 at the end of every layer (pg_tune key 18) or with
the weights rounded to bf16 beforehand.  Prints the relative L2 distance from the fp32 gradient, median and worst
over the encoder's weight matrices."""
import sys, torch
sys.path.insert(0, ".")
import pgload
pg = pgload.load()
from protgen_gan_amd._lib import lib
DEV = "cuda"
L = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B, S = 2, 24
g = torch.Generator().manual_seed(11)
real = torch.randint(5, 25, (B, S), generator=g).to(DEV)
fake = torch.randint(5, 25, (B, S), generator=g).to(DEV)
real[:, 0] = fake[:, 0] = 2
rm = torch.ones(B, S, device=DEV)
alpha = torch.rand(B, 1, 1, generator=g)


def run(dt, mask=0, round_weights=False):
    cfg = pg.ProtBertConfig(num_hidden_layers=L, max_position_embeddings=64)
    critic = pg.Critic(pg.ProtBertForMaskedLM(cfg, device=DEV, compute_dtype=dt).init_random(seed=1))
    torch.manual_seed(2)
    for p in critic.classifier.parameters():
        torch.nn.init.normal_(p, std=0.05)
    with torch.no_grad():
        for k in (0, 2, 4):
            critic.classifier[k].bias.fill_(10.0)
        if round_weights:
            m = critic.protbert.store.master
            m.copy_(m.bfloat16().float())
    critic.zero_grad(set_to_none=True)
    lib.pg_tune(18, mask)
    gp = pg.compute_gradient_penalty(critic, real, fake, rm, rm, DEV, alpha=alpha)
    (500.0 * gp).backward()
    torch.cuda.synchronize()
    lib.pg_tune(18, 0)
    return float(gp), {n: p.grad.double().cpu() for n, p in critic.named_parameters()
                       if p.grad is not None and n.endswith("weight") and "encoder.layer" in n and "LayerNorm" not in n}


def dist(a, b):
    v = sorted(float((a[n] - b[n]).norm() / b[n].norm()) for n in b)
    return v[len(v) // 2], v[-1]


gp0, g0 = run(torch.float32)
rows = [("bf16 engine", run(torch.bfloat16)),
        ("fp32, weights rounded to bf16", run(torch.float32, 0, True)),
        ("fp32, forward token-row slots rounded", run(torch.float32, 1)),
        ("fp32, attention probabilities rounded", run(torch.float32, 2)),
        ("fp32, adjoint token-row slots rounded", run(torch.float32, 4)),
        ("fp32, adjoint scores rounded", run(torch.float32, 8)),
        ("fp32, tangent token-row slots rounded", run(torch.float32, 16)),
        ("fp32, tangent scores rounded", run(torch.float32, 32)),
        ("fp32, every slot and the weights rounded", run(torch.float32, 63, True))]
print(f"{L} layers, B {B}, S {S}: gp fp32 {gp0:.6f}")
for name, (gp, gr) in rows:
    med, worst = dist(gr, g0)
    print(f"  {name:44s} gp {gp:.6f}  median {med:.4f}  worst {worst:.4f}", flush=True)
