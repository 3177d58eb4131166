"""Sustained GEMM run: does the rate drop once the chip has been busy for a while?"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgload
pg = pgload.load()
from protgen_gan_amd._lib import lib
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
M, N, K = 4096, 4096, 1024
A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
Ws = [(torch.randn(N, K, device=dev, generator=g) * 0.05).to(torch.bfloat16) for _ in range(30)]
bias = torch.randn(N, device=dev)
C = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
aux = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
wide = int(sys.argv[1]) if len(sys.argv) > 1 else 1
lib.pg_tune(2, wide)
for label, kw in (("plain", {}), ("bias", {"bias": bias}), ("bias+gelu", {"bias": bias, "act": 1}), ("bias+gelu+aux", {"bias": bias, "act": 1, "aux": aux})):
    for chunk in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 300
        for i in range(n):
            pg.ops.gemm(A, Ws[i % 30], C, **kw)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        print(f"{label:14s} chunk {chunk}: {us:6.1f} us  {2.0*M*N*K/us/1e6:6.0f} TF", flush=True)
