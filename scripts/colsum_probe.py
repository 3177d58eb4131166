"""Column-sum launch time against the number of row chunks (pg_tune key 12)."""
import sys, torch
sys.path.insert(0, ".")
import pgload
pg = pgload.load()
from protgen_gan_amd import ops
from protgen_gan_amd._lib import lib
big = torch.empty(300 << 20, dtype=torch.uint8, device="cuda")
for rows, N in ((5376, 4096), (5376, 3072), (2688, 4096), (8192, 4096)):
  x = torch.randn(rows, N, device="cuda").bfloat16()
  db = torch.zeros(N, device="cuda")
  for waves in (4, 16):
    lib.pg_tune(17, waves)
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
        # cold: another 300 MB written between the calls
        cold = 0.0
        for _ in range(5):
            big.fill_(1)
            e0.record()
            ops.colsum_acc(x, db)
            e1.record()
            torch.cuda.synchronize()
            cold += e0.elapsed_time(e1) / 5 * 1e3
        print(f"rows {rows} N {N} waves {waves:2d} chunks {chunks:4d}: {us:6.1f} us  {rows * N * 2 / us / 1e6:5.2f} TB/s   cold {cold:6.1f} us", flush=True)
