"""Short program for rocprofv3 --pmc passes: a few launches of the dominant GEMM shapes of the GAN step
(fill-pass W1 with bias+GELU on the 256x256 kernel, fill-pass W2 with bias+residual, critic W2 at T=8192)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgload
pg = pgload.load()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
bf = torch.bfloat16
for name, M, N, K, kw in [("w1_fill", 34176, 4096, 1024, "gelu"), ("w2_fill", 34176, 1024, 4096, "res"), ("w2_critic", 8192, 1024, 4096, "res")]:
    A = torch.randn(M, K, device=dev, generator=g).to(bf)
    B = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(bf)
    C = torch.empty(M, N, dtype=bf, device=dev)
    bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev, generator=g).to(bf)
    for _ in range(4):
        if kw == "gelu":
            pg.ops.gemm(A, B, C, bias=bias, act=1)
        else:
            pg.ops.gemm(A, B, C, bias=bias, residual=res)
    torch.cuda.synchronize()
    print(name, M, N, K, "algorithmic operand+output bytes", (M * K + N * K + M * N * (2 if kw == "res" else 1)) * 2)
