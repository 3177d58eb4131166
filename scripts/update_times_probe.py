import random, sys, time, torch
sys.path.insert(0, ".")
import pgload
pg = pgload.load()
from protgen_gan_amd import train as T
dev = torch.device("cuda", 0)
cfg = pg.ProtBertConfig()
torch.manual_seed(2)
critic = pg.Critic(pg.ProtBertForMaskedLM(cfg, device=dev, compute_dtype=torch.bfloat16).init_random(seed=1))
gen = pg.Generator(pg.ProtBertForMaskedLM(cfg, device=dev, compute_dtype=torch.bfloat16).init_random(seed=0))
gopt = pg.AdamW(gen.parameters(), lr=5e-6); copt = pg.AdamW(critic.parameters(), lr=5e-5)
hp = T.GanHyper(n_critic=8, lambda_gp=5.0)
rng, tg = random.Random(1), torch.Generator().manual_seed(1)
pool = list(range(200, 501))
real = T.synthetic_real_batch(8, pool, 512, dev, rng, tg)
fake8 = T.synthetic_real_batch(8, pool, 512, dev, rng, tg)
fake4 = T.synthetic_real_batch(4, pool, 512, dev, rng, tg)
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("critic update", t(lambda: T.critic_update(critic, copt, real["input_ids"], real["attention_mask"], fake8["input_ids"], hp, alpha=torch.rand(8,1,1), real_lengths=real["lengths"], fake_lengths=fake8["lengths"])))
print("generator update", t(lambda: T.generator_update(critic, gopt, fake4["input_ids"], fake_lengths=fake4["lengths"])))
