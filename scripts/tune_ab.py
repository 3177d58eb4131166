"""A/B of one pg_tune key on plain k-major x k-major bf16 products (bias + residual epilogue), 16 rotating operand
sets.  usage: python scripts/tune_ab.py KEY VALUE_A VALUE_B M1,M2,... [N:K,N:K,...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgload

pg = pgload.load()
from protgen_gan_amd._lib import lib

dev = "cuda"
key, va, vb = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
Ms = [int(v) for v in sys.argv[4].split(",")]
NK = [tuple(int(x) for x in p.split(":")) for p in (sys.argv[5] if len(sys.argv) > 5 else "1024:1024,1024:4096").split(",")]
g = torch.Generator(device=dev).manual_seed(0)
for M in Ms:
    for N, K in NK:
        n = 16
        As = [torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16) for _ in range(n)]
        Bs = [(torch.randn(N, K, device=dev, generator=g) * 0.05).to(torch.bfloat16) for _ in range(n)]
        Cs = [torch.zeros(M, N, dtype=torch.bfloat16, device=dev) for _ in range(n)]
        Rs = [torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16) for _ in range(n)]
        bias = torch.randn(N, device=dev)
        res, outs = [], []
        for v in (va, vb, va, vb):
            lib.pg_tune(key, v)
            best = 1e9
            for rep in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for i in range(n):
                    pg.ops.gemm(As[i], Bs[i], Cs[i], bias=bias, residual=Rs[i])
                e0.record()
                for _ in range(2):
                    for i in range(n):
                        pg.ops.gemm(As[i], Bs[i], Cs[i], bias=bias, residual=Rs[i])
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / (2 * n) * 1e3)
            res.append(best)
            outs.append(Cs[0].float().clone())
        lib.pg_tune(key, va)
        print(f"M{M:6d} N{N} K{K}: key {key}={va}: {res[0]:6.1f} / {res[2]:6.1f} us   ={vb}: {res[1]:6.1f} / {res[3]:6.1f} us   "
              f"same bits: {torch.equal(outs[0], outs[1])}", flush=True)
