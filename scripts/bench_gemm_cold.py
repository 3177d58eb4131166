"""GEMM timing with operands rotating through > 256 MB (Infinity-Cache-cold), as inside a 30-layer pass."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgload
pg = pgload.load()
from protgen_gan_amd._lib import lib
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
T = 4096
shapes = [("qkv", T, 3072, 1024), ("wo", T, 1024, 1024), ("w1", T, 4096, 1024), ("w2", T, 1024, 4096)]
for name, M, N, K in shapes:
    nA = max(2, int(400e6 // (M * K * 2)))
    As = [torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16) for _ in range(min(nA, 24))]
    Bs = [(torch.randn(N, K, device=dev, generator=g) * 0.05).to(torch.bfloat16) for _ in range(30)]
    C = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    bias = torch.randn(N, device=dev)
    for path in (2, 4):
        lib.pg_tune(0, path)
        for hot in (True, False):
            for _ in range(5):
                pg.ops.gemm(As[0], Bs[0], C, bias=bias)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            n = 120
            for i in range(n):
                pg.ops.gemm(As[0] if hot else As[i % len(As)], Bs[0] if hot else Bs[i % 30], C, bias=bias)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / n * 1e3
            print(f"{name:4s} glds{path} {'hot ' if hot else 'cold'}: {us:6.1f} us {2.0*M*N*K/us/1e6:6.0f} TF", flush=True)
    lib.pg_tune(0, 1)
