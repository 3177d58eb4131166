"""One full-size encoder forward (inference and stash variants) for kernel-trace inspection."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgload
pg = pgload.load()
for kv in sys.argv[2:]:
    pg._lib.lib.pg_tune(*map(int, kv.split("=")))
dev = "cuda"
cfg = pg.ProtBertConfig()
m = pg.ProtBertForMaskedLM(cfg, device=dev, compute_dtype=torch.bfloat16).init_random(0)
B, S = 8, int(sys.argv[1]) if len(sys.argv) > 1 else 512
ids = torch.randint(5, 25, (B, S), device=dev)
mask = torch.ones(B, S, device=dev)
for it in range(3):
    h, _ = m._encode(ids, None, mask, want_stash=False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for it in range(5):
    h, _ = m._encode(ids, None, mask, want_stash=False)
e1.record(); torch.cuda.synchronize()
print("inference forward ms:", e0.elapsed_time(e1) / 5)
e0.record()
for it in range(3):
    h, st = m._encode(ids, None, mask, want_stash=True)
e1.record(); torch.cuda.synchronize()
print("stash forward ms:", e0.elapsed_time(e1) / 3)
