"""Target finder: this library's GEMM next to torch.matmul (hipBLASLt/rocBLAS) on the ProtBERT shapes,
hot (same operands) and cold (operands rotating through > 256 MB).  Measurement aid only."""
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
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
shapes = [("qkv", T, 3072, 1024), ("wo", T, 1024, 1024), ("w1", T, 4096, 1024), ("w2", T, 1024, 4096)]


def timed(fn, n=120):
    for i in range(5):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, M, N, K in shapes:
    nA = max(2, min(24, int(600e6 // (M * K * 2))))
    As = [torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16) for _ in range(nA)]
    Bs = [(torch.randn(N, K, device=dev, generator=g) * 0.05).to(torch.bfloat16) for _ in range(30)]
    C = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    fl = 2.0 * M * N * K
    for hot in (True, False):
        a = (lambda i: As[0]) if hot else (lambda i: As[i % nA])
        b = (lambda i: Bs[0]) if hot else (lambda i: Bs[i % 30])
        row = []
        for path in (2, 4):
            lib.pg_tune(0, path)
            lib.pg_tune(2, 0)
            us = timed(lambda i: pg.ops.gemm(a(i), b(i), C))
            row.append(f"glds{path} {us:6.1f} us {fl/us/1e6:5.0f} TF")
        lib.pg_tune(0, 1)
        lib.pg_tune(2, 2)
        us = timed(lambda i: pg.ops.gemm(a(i), b(i), C))
        row.append(f"wide {us:6.1f} us {fl/us/1e6:5.0f} TF")
        lib.pg_tune(2, 1)
        us = timed(lambda i: torch.matmul(a(i), b(i).t(), out=C))
        row.append(f"lib {us:6.1f} us {fl/us/1e6:5.0f} TF")
        print(f"{name:5s} {'hot ' if hot else 'cold'}: " + " | ".join(row), flush=True)
