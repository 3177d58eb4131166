"""GEMM micro-benchmark on the ProtBERT shapes (random bf16 data), both kernel paths interleaved."""
import os, sys, ctypes
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgload
pg = pgload.load()
from protgen_gan_amd._lib import lib
dev = "cuda"
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
shapes = [  # name, M, N, K, ak, bk, out
    ("qkv  NT", T, 3072, 1024, True, True, torch.bfloat16),
    ("wo   NT", T, 1024, 1024, True, True, torch.bfloat16),
    ("w1   NT", T, 4096, 1024, True, True, torch.bfloat16),
    ("w2   NT", T, 1024, 4096, True, True, torch.bfloat16),
    ("dgu  NN", T, 4096, 1024, True, False, torch.bfloat16),
    ("dh   NN", T, 1024, 4096, True, False, torch.bfloat16),
    ("dx   NN", T, 1024, 3072, True, False, torch.bfloat16),
    ("pv   NN", 512, 64, 512, True, False, torch.bfloat16),
    ("dk   TN", 512, 64, 512, False, False, torch.bfloat16),
    ("dW1  TN", 4096, 1024, T, False, False, torch.float32),
    ("dW2  TN", 1024, 4096, T, False, False, torch.float32),
    ("dWo  TN", 1024, 1024, T, False, False, torch.float32),
    ("dWqk TN", 3072, 1024, T, False, False, torch.float32),
]
g = torch.Generator(device=dev).manual_seed(0)
for name, M, N, K, ak, bk, od in shapes:
    A = torch.randn((M, K) if ak else (K, M), device=dev, generator=g).to(torch.bfloat16)
    B = (torch.randn((N, K) if bk else (K, N), device=dev, generator=g) * 0.05).to(torch.bfloat16)
    C = torch.zeros(M, N, dtype=od, device=dev)
    res = {}
    for rnd in range(3):
        for path in (0, 2, 4):
            lib.pg_tune(0, path if path else 1)
            lib.pg_tune(1, 2 if path == 0 else 1)
            for _ in range(3):
                pg.ops.gemm(A, B, C, a_kmajor=ak, b_kmajor=bk)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            n = 20
            for _ in range(n):
                pg.ops.gemm(A, B, C, a_kmajor=ak, b_kmajor=bk)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / n * 1e3
            res.setdefault(path, []).append(us)
    lib.pg_tune(0, 1); lib.pg_tune(1, 1)
    fl = 2.0 * M * N * K
    r0, r1, r2 = min(res[0]), min(res[2]), min(res[4])
    print(f"{name} M{M} N{N} K{K}: n64 {r0:7.1f} us {fl/r0/1e6:6.0f} TF | glds2 {r1:7.1f} us {fl/r1/1e6:6.0f} TF | glds4 {r2:7.1f} us {fl/r2/1e6:6.0f} TF")
