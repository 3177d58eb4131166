"""One critic update at the benchmark's shape (30 layers, B 8, S 512, bf16) with ragged rows and with the
rectangular passes, from the same weights and inputs: loss and per-tensor gradient distance."""
import random
import sys

import torch

sys.path.insert(0, ".")
import pgload

pg = pgload.load()
from protgen_gan_amd import train as T

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = "cuda"
cfg = pg.ProtBertConfig(num_hidden_layers=layers)
critic = pg.Critic(pg.ProtBertForMaskedLM(cfg, device=dev, compute_dtype=torch.bfloat16).init_random(seed=1))
rng = random.Random(5)
tg = torch.Generator().manual_seed(5)
pool = list(range(200, 501))
B, S = 8, 512
real = T.synthetic_real_batch(B, pool, S, dev, rng, tg)
fk = T.synthetic_real_batch(B, pool, S, dev, rng, tg)
fake = fk["input_ids"]
alpha = torch.rand(B, 1, 1, generator=tg)


class NoStep:
    def zero_grad(self):
        critic.zero_grad(set_to_none=True)

    def step(self):
        pass


out = {}
for ragged in (False, True):
    loss = T.critic_update(critic, NoStep(), real["input_ids"], real["attention_mask"], fake, T.GanHyper(lambda_gp=5.0),
                           alpha=alpha, ragged=ragged)
    torch.cuda.synchronize()
    out[ragged] = (float(loss), {n: p.grad.double().cpu() for n, p in critic.named_parameters() if p.grad is not None})
print("loss rect", out[False][0], "ragged", out[True][0])
worst = []
for n, ref in out[False][1].items():
    got = out[True][1][n]
    worst.append((float((got - ref).norm() / (ref.norm() + 1e-30)), float(ref.norm()), n))
worst.sort(reverse=True)
for w in worst[:12]:
    print("%.3e  |ref| %.3e  %s" % w)
print("median rel l2", sorted(w[0] for w in worst)[len(worst) // 2])
