"""Which side bounds the 256x256 GEMM: loads only / compute only / both (debug switches via pg_tune key 3)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgload
pg = pgload.load()
from protgen_gan_amd._lib import lib
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
def timed(fn, n=100):
    for i in range(5): fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for name, M, N, K in [("w1", T, 4096, 1024), ("w1k4", T, 4096, 4096), ("qkv", T, 3072, 1024)]:
    nA = max(2, min(8, int(600e6 // (M * K * 2))))
    As = [torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16) for _ in range(nA)]
    Bs = [(torch.randn(N, K, device=dev, generator=g) * 0.05).to(torch.bfloat16) for _ in range(30)]
    C = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    fl = 2.0 * M * N * K
    for dbg, label in ((0, "full"), (1, "no loads"), (2, "no mfma"), (3, "neither")):
        lib.pg_tune(3, dbg)
        hot = timed(lambda i: pg.ops.gemm(As[0], Bs[0], C))
        cold = timed(lambda i: pg.ops.gemm(As[i % nA], Bs[i % 30], C))
        print(f"{name:5s} {label:9s}: hot {hot:6.1f} us ({fl/hot/1e6:5.0f} TF)  cold {cold:6.1f} us ({fl/cold/1e6:5.0f} TF)", flush=True)
    lib.pg_tune(3, 0)
