"""A/B of the two 256x256-tile kernels for k-major x k-major products (pg_tune key 21: 0 = two 64 KiB stages,
1 = five-slot ring of 32-k half-stages) on the fill pass's and the critic's forward shapes, each timed over 16
rotating operand sets with the forward epilogues (bias; GELU on W1; residual on Wo / W2)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgload

pg = pgload.load()
from protgen_gan_amd._lib import lib

dev = "cuda"
Ms = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [32768, 12288, 8192]
NK = [("qkv", 3072, 1024), ("wo", 1024, 1024), ("w1", 4096, 1024), ("w2", 1024, 4096)]
RING = 16
g = torch.Generator(device=dev).manual_seed(0)
for M in Ms:
    for name, N, K in NK:
        n = RING if M * max(N, K) < 2 ** 27 else 4
        As = [torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16) for _ in range(n)]
        Bs = [(torch.randn(N, K, device=dev, generator=g) * 0.05).to(torch.bfloat16) for _ in range(n)]
        Cs = [torch.zeros(M, N, dtype=torch.bfloat16, device=dev) for _ in range(n)]
        kw = {"bias": torch.randn(N, device=dev)}
        if name == "w1":
            kw["act"] = 1
        if name in ("wo", "w2"):
            kw["residual"] = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16)
        res = []
        outs = []
        runs = [("two-stage", {21: 0}), ("ring", {21: 1}), ("two-stage", {21: 0}), ("ring", {21: 1})]
        if os.environ.get("PG_AB_DBG"):  # loop time without the in-loop loads (results are garbage)
            runs = [("two-stage", {21: 0}), ("ring", {21: 1}), ("two-stage", {21: 0, 3: 1}), ("ring", {21: 2})]
        for _, tune in runs:
            for k, v in tune.items():
                lib.pg_tune(k, v)
            lib.pg_tune(2, 2)
            best = 1e9
            for rep in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for i in range(n):
                    pg.ops.gemm(As[i], Bs[i], Cs[i], **kw)
                e0.record()
                for _ in range(2):
                    for i in range(n):
                        pg.ops.gemm(As[i], Bs[i], Cs[i], **kw)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / (2 * n) * 1e3)
            res.append(best)
            outs.append(Cs[0].float().clone())
            lib.pg_tune(3, 0)
        lib.pg_tune(2, 1)
        lib.pg_tune(21, 0)
        fl = 2.0 * M * N * K
        same = torch.equal(outs[0], outs[1])
        print(f"M{M:6d} {name:4s} N{N} K{K}: two-stage {res[0]:6.1f} / {res[2]:6.1f} us ({fl / res[2] / 1e6:5.0f} TF)   "
              f"ring {res[1]:6.1f} / {res[3]:6.1f} us ({fl / res[3] / 1e6:5.0f} TF)   same bits: {same}", flush=True)
