"""bf16 engine against the fp32 engine on one critic update at bench-like shapes: relative L2 distance of the
gradient tensors (median / worst), with the head's ReLU gates as initialised and with every gate open, and per loss
term (lambda_gp = 0 leaves D(fake) - D(real))."""
import random, sys
import numpy as np, torch
sys.path.insert(0, ".")
import pgload
pg = pgload.load()
from protgen_gan_amd import train as T
DEV, BF = "cuda", torch.bfloat16


class NoStep:
    def __init__(self, m):
        self.param_groups = [{"params": list(m.parameters())}]
    def zero_grad(self, set_to_none=True):
        for p in self.param_groups[0]["params"]:
            p.grad = None
    def step(self):
        pass


def batches(B, S, seed):
    rng, tg = random.Random(seed), torch.Generator().manual_seed(seed)
    pool = list(range(S // 3, S - 2))
    real = T.synthetic_real_batch(B, pool, S, DEV, rng, tg)
    fake = torch.randint(5, 25, (B, S), generator=tg).to(DEV)
    fake[:, 0] = 2
    for i in range(B):
        L = rng.choice(pool) + 2
        fake[i, L - 1] = 3
        fake[i, L:] = 0
    return real, fake, torch.rand(B, 1, 1, generator=tg)


for S, L, B in ((512, 4, 4), (1024, 4, 4), (512, 12, 4)):
    for gates_open in (False, True):
        for lam in (5.0, 0.0):
            real, fake, alpha = batches(B, S, 7)
            out = {}
            for dt in (torch.float32, BF):
                cfg = pg.ProtBertConfig(num_hidden_layers=L, max_position_embeddings=S)
                torch.manual_seed(3)
                critic = pg.Critic(pg.ProtBertForMaskedLM(cfg, device=DEV, compute_dtype=dt).init_random(seed=2))
                if gates_open:
                    with torch.no_grad():
                        for k in (0, 2, 4):
                            critic.classifier[k].bias.fill_(10.0)
                loss = T.critic_update(critic, NoStep(critic), real["input_ids"], real["attention_mask"], fake,
                                       T.GanHyper(lambda_gp=lam, max_len=S), alpha=alpha, real_lengths=real["lengths"])
                torch.cuda.synchronize()
                out[dt] = (float(loss), {n: p.grad.double().cpu() for n, p in critic.named_parameters() if p.grad is not None})
            (l0, g0), (l1, g1) = out[torch.float32], out[BF]
            norms = sorted(float(v.norm()) for v in g0.values())
            typical = norms[len(norms) // 2]
            dist = {n: float((g1[n] - g0[n]).norm() / g0[n].norm()) for n in g0
                    if float(g0[n].norm()) > 0.05 * typical and "key.bias" not in n}
            vals = sorted(dist.values())
            worst = max(dist.items(), key=lambda kv: kv[1])
            print(f"S={S} L={L} open={gates_open} lambda={lam}: loss {l0:.5f} {l1:.5f}  median {vals[len(vals)//2]:.4f} "
                  f"p90 {vals[int(0.9*len(vals))]:.4f} worst {worst[1]:.4f} {worst[0]}", flush=True)
