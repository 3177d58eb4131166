"""A/B of the persistent 256x256-tile GEMM (pg_tune key 24: 0 = one workgroup per tile, 1 = one workgroup per CU
walking its tiles) on the forward and backward shapes of the encoder, 16 rotating operand sets."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgload

pg = pgload.load()
from protgen_gan_amd._lib import lib

dev = "cuda"
Ms = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [32768, 8192, 2432]
SH = [("wo  +res", 1024, 1024, True, "res"), ("w2  +res", 1024, 4096, True, "res"), ("w1  gelu", 4096, 1024, True, "gelu"),
      ("dgu xdgelu", 4096, 1024, False, "dgelu"), ("dx  +res", 1024, 3072, False, "res"), ("qkv bias", 3072, 1024, True, "bias")]
g = torch.Generator(device=dev).manual_seed(0)
for M in Ms:
    for name, N, K, bk, kind in SH:
        n = 16 if M * max(N, K) < 2 ** 27 else 4
        As = [torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16) for _ in range(n)]
        Bs = [(torch.randn((N, K) if bk else (K, N), device=dev, generator=g) * 0.05).to(torch.bfloat16) for _ in range(n)]
        Cs = [torch.zeros(M, N, dtype=torch.bfloat16, device=dev) for _ in range(n)]
        Rs = [torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16) for _ in range(n)]
        bias = torch.randn(N, device=dev)
        res, outs = [], []
        for v in (0, 1, 0, 1):
            lib.pg_tune(24, v)
            best = 1e9
            for rep in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

                def run(i):
                    kw = dict(a_kmajor=True, b_kmajor=bk)
                    if kind == "res":
                        kw["residual"] = Rs[i]
                    if kind == "gelu":
                        kw["act"] = 1
                    if kind == "dgelu":
                        kw["act"], kw["aux"] = pg._lib.PG_ACT_MUL_DGELU, Rs[i]
                    if bk:
                        kw["bias"] = bias
                    pg.ops.gemm(As[i], Bs[i], Cs[i], **kw)
                for i in range(n):
                    run(i)
                e0.record()
                for _ in range(2):
                    for i in range(n):
                        run(i)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / (2 * n) * 1e3)
            res.append(best)
            outs.append(Cs[0].float().clone())
        lib.pg_tune(24, 0)
        print(f"M{M:6d} {name:10s} N{N} K{K}: per tile {res[0]:6.1f} / {res[2]:6.1f} us   persistent {res[1]:6.1f} / {res[3]:6.1f} us   "
              f"same bits: {torch.equal(outs[0], outs[1])}", flush=True)
