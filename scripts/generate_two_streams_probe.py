"""Mass generation (generate.py full mode, batch 64, lengths 200..500) on one stream against two host threads with a
stream each, every thread filling its own batches.  Batches are independent; a packed batch's N = 1024 products cover
1.25 - 1.5 rounds of the chip, so a second stream has CUs to take."""
import sys
import threading
import time

import torch
sys.path.insert(0, ".")
import pgload
pg = pgload.load()
from protgen_gan_amd import generate as gen_py

dev = torch.device("cuda", 0)
generator = pg.Generator(pg.ProtBertForMaskedLM(pg.ProtBertConfig(), device=dev, compute_dtype=torch.bfloat16).init_random(seed=0))
tok = pg.ProtTokenizer()
pool = list(range(200, 501))


def run(n, seed):
    return gen_py.generate_full_mode_sequences(generator, tok, n, pool, batch_size=64, device=str(dev), rng_seed=seed)


run(64, 1)
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    run(64 * 8, 42)
    torch.cuda.synchronize()
    t1 = time.perf_counter() - t0
    streams = [torch.cuda.Stream() for _ in range(2)]

    def job(i):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(streams[i]):
            run(64 * 4, 42 + i)
            streams[i].synchronize()

    t0 = time.perf_counter()
    th = [threading.Thread(target=job, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    t2 = time.perf_counter() - t0
    print(f"512 sequences: one stream {512 / t1:.1f} sequences/s, two streams {512 / t2:.1f} sequences/s", flush=True)
