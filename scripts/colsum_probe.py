"""Column-sum launch time against the number of row chunks (pg_tune key 12)."""
import sys, torch
sys.path.insert(0, ".")
import pgload
pg = pgload.load()
from protgen_gan_amd import ops
from protgen_gan_amd._lib import lib
for rows, N in ((8192, 4096), (4096, 4096), (8192, 3072), (4096, 3072)):
    x = torch.randn(rows, N, device="cuda").bfloat16()
    db = torch.zeros(N, device="cuda")
    for chunks in (16, 32, 64, 128, 256):
        lib.pg_tune(12, chunks)
        for _ in range(5):
            ops.colsum_acc(x, db)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            ops.colsum_acc(x, db)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        print(f"rows {rows} N {N} chunks {chunks:4d}: {us:6.1f} us  {rows * N * 2 / us / 1e6:5.2f} TB/s", flush=True)
