"""Does the next iteration's stacked fill (the generator's ten encoder passes over 68 sequences: full-chip
launches) hide in the CUs the critic updates leave idle (their N = 1024 products cover half the chip)?  Times, at
configs[1]'s sizes: the fill alone, the eight critic updates + generator update alone, one after the other, and
both at once (fill on a side stream driven by a second host thread).  The generator does not change inside the
reference's loop, so the next iteration's fakes may be filled while this iteration's updates run."""
import random
import sys
import threading
import time

import torch
sys.path.insert(0, ".")
import pgload
pg = pgload.load()
from protgen_gan_amd import train as T

dev = torch.device("cuda", 0)
bf = torch.bfloat16
cfg = pg.ProtBertConfig()
generator = pg.Generator(pg.ProtBertForMaskedLM(cfg, device=dev, compute_dtype=bf).init_random(seed=0))
torch.manual_seed(2)
critic = pg.Critic(pg.ProtBertForMaskedLM(cfg, device=dev, compute_dtype=bf).init_random(seed=1))
hp = T.GanHyper(n_critic=8, lambda_gp=5.0, max_len=512, gen_batch_blind=4)
gen_opt = pg.AdamW(generator.parameters(), lr=5e-6)
critic_opt = pg.AdamW(critic.parameters(), lr=5e-5)
rng, tgen = random.Random(1234), torch.Generator().manual_seed(1234)
pool = list(range(200, 501))
side = torch.cuda.Stream()


def plan():
    lengths = [T.sample_fake_lengths(pool, 8, rng) for _ in range(8)] + [T.sample_fake_lengths(pool, 4, rng)]
    temps = [T.fill_temperatures(hp) for _ in range(9)]
    return lengths, temps


def fill(p):
    return T.generate_fake_batches(generator, p[0], p[1], hp, dev)


def updates(p, fakes, real):
    for batch, fake, fl in zip(real, fakes, p[0]):
        T.critic_update(critic, critic_opt, batch["input_ids"], batch["attention_mask"], fake, hp,
                        alpha=torch.rand(8, 1, 1), real_lengths=batch["lengths"], fake_lengths=fl)
    T.generator_update(critic, gen_opt, fakes[-1], fake_lengths=p[0][-1])


def timed(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


real = [T.synthetic_real_batch(8, pool, 512, dev, rng, tgen) for _ in range(8)]
p0 = plan()
f0 = fill(p0)
torch.cuda.synchronize()
t_fill = timed(lambda: fill(p0))
t_upd = timed(lambda: updates(p0, f0, real))
t_seq = timed(lambda: (fill(p0), updates(p0, f0, real)))


def both():
    def job():
        with torch.cuda.stream(side):
            fill(p0)
    side.wait_stream(torch.cuda.current_stream())
    th = threading.Thread(target=job)
    th.start()
    updates(p0, f0, real)
    th.join()
    torch.cuda.current_stream().wait_stream(side)


t_both = timed(both)
hi = torch.cuda.Stream(priority=-1)


def both_prio(fill_high):
    def job():
        with torch.cuda.stream(hi if fill_high else side):
            fill(p0)
    cur = torch.cuda.current_stream()
    (hi if fill_high else side).wait_stream(cur)
    th = threading.Thread(target=job)
    th.start()
    if fill_high:
        updates(p0, f0, real)
    else:
        hi.wait_stream(cur)
        with torch.cuda.stream(hi):
            updates(p0, f0, real)
        cur.wait_stream(hi)
    th.join()
    cur.wait_stream(hi if fill_high else side)


t_upd_hi = timed(lambda: both_prio(False))
t_fill_hi = timed(lambda: both_prio(True))
print(f"at once, updates on a high-priority stream {t_upd_hi:.1f} ms; fill on a high-priority stream {t_fill_hi:.1f} ms")
print(f"fill alone {t_fill:.1f} ms, updates alone {t_upd:.1f} ms, one after the other {t_seq:.1f} ms, at once {t_both:.1f} ms",
      flush=True)
