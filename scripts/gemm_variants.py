"""Which tile kernel is fastest for the critic's GEMM shapes?  Every product is timed over a ring of 16
different weight matrices and activation buffers (as in a pass over the layers: nothing of one launch's
operands is left in the caches for the next), for each kernel the dispatcher can be forced to:
128x128 (two-buffer / four-buffer ring), 128x64, 256x256."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgload

pg = pgload.load()
from protgen_gan_amd._lib import lib

dev = "cuda"
Ms = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2432, 5696, 8128, 12288]
NK = [("qkv NT", 3072, 1024, True), ("wo  NT", 1024, 1024, True), ("w1  NT", 4096, 1024, True), ("w2  NT", 1024, 4096, True),
      ("dgu NN", 4096, 1024, False), ("dh  NN", 1024, 4096, False), ("dx  NN", 1024, 3072, False), ("dc  NN", 1024, 1024, False)]
VARIANTS = [("128/2", {0: 2, 1: 1, 2: 0}), ("128/4", {0: 4, 1: 1, 2: 0}), ("128x64", {0: 1, 1: 2, 2: 0, "force_narrow": 1}),
            ("256", {0: 1, 1: 1, 2: 2})]
RING = int(os.environ.get("PG_VAR_RING", "16"))
EPI = os.environ.get("PG_VAR_EPI") == "1"  # the forward epilogues: bias everywhere, GELU on w1, residual on wo / w2
g = torch.Generator(device=dev).manual_seed(0)


def set_variant(v):
    lib.pg_tune(0, v[0])
    lib.pg_tune(1, v[1])
    lib.pg_tune(2, v[2])


for M in Ms:
    for name, N, K, bk in NK:
        As = [torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16) for _ in range(RING)]
        Bs = [(torch.randn((N, K) if bk else (K, N), device=dev, generator=g) * 0.05).to(torch.bfloat16) for _ in range(RING)]
        Cs = [torch.zeros(M, N, dtype=torch.bfloat16, device=dev) for _ in range(RING)]
        kw = {}
        if EPI and bk:
            kw["bias"] = torch.randn(N, device=dev)
            if name.startswith("w1"):
                kw["act"] = 1
            if name.startswith(("wo", "w2")):
                kw["residual"] = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16)
        out = []
        for vname, v in VARIANTS:
            if vname == "128x64" and (M + 127) // 128 * ((N + 127) // 128) > 256:
                out.append((vname, float("nan")))
                continue
            set_variant(v)
            best = 1e9
            for rep in range(3):
                for i in range(RING):
                    pg.ops.gemm(As[i], Bs[i], Cs[i], a_kmajor=True, b_kmajor=bk, **kw)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(2):
                    for i in range(RING):
                        pg.ops.gemm(As[i], Bs[i], Cs[i], a_kmajor=True, b_kmajor=bk, **kw)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / (2 * RING) * 1e3)
            out.append((vname, best))
        lib.pg_tune(0, 1); lib.pg_tune(1, 1); lib.pg_tune(2, 1)
        fl = 2.0 * M * N * K
        cells = " | ".join(f"{vn} {us:6.1f}us {fl / us / 1e6:5.0f}TF" for vn, us in out)
        print(f"M{M:6d} {name} N{N} K{K}: {cells}", flush=True)
