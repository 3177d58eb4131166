"""Short program for rocprofv3 --pmc passes: four launches each of the dominant GEMM shapes of the GAN step (the
fill pass's QKV, W1 with bias+GELU, W2 and Wo with bias+residual at its packed token count; the critic's W1 and W2
at the ragged token count of one update).  Prints each shape's operand and output bytes."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgload
pg = pgload.load()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
bf = torch.bfloat16
SHAPES = [("fill_qkv", 33024, 3072, 1024, "bias"), ("fill_w1", 33024, 4096, 1024, "gelu"), ("fill_w2", 33024, 1024, 4096, "res"),
          ("fill_wo", 33024, 1024, 1024, "res"), ("critic_w1", 8064, 4096, 1024, "gelu"), ("critic_w2", 8064, 1024, 4096, "res")]
out = []
for name, M, N, K, kw in SHAPES:
    A = torch.randn(M, K, device=dev, generator=g).to(bf)
    B = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(bf)
    C = torch.empty(M, N, dtype=bf, device=dev)
    bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev, generator=g).to(bf)
    for _ in range(4):
        if kw == "gelu":
            pg.ops.gemm(A, B, C, bias=bias, act=1)
        elif kw == "res":
            pg.ops.gemm(A, B, C, bias=bias, residual=res)
        else:
            pg.ops.gemm(A, B, C, bias=bias)
    torch.cuda.synchronize()
    out.append({"shape": name, "M": M, "N": N, "K": K, "read_bytes": (M * K + N * K + (M * N if kw == "res" else 0)) * 2,
                "write_bytes": M * N * 2})
print(json.dumps(out))
