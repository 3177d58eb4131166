"""Times the four fused attention kernels of a critic update (forward with P written, first-order backward,
tangent pass, second-order pass) and the key-side products at the benchmark's shape: 24 ragged sequences of
200..500 tokens (3B of one update; the tangent and second-order passes run on the first 8), 16 heads.
Each kernel runs over NSETS rotating score-shaped buffer sets so that P / dS come from HBM as in a pass over
30 layers.  PG_ATTN_ONCE=1: one launch of each (for rocprofv3 --pmc passes)."""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgload

pg = pgload.load()
from protgen_gan_amd._lib import lib
from protgen_gan_amd.ragged import Ragged

for kv in os.environ.get("PG_TUNE", "").split(","):  # e.g. PG_TUNE=23=0
    if kv:
        lib.pg_tune(*[int(x) for x in kv.split("=")])
DEV, BF = "cuda", torch.bfloat16
heads, S = 16, 512
H = heads * 64
rng = random.Random(7)
lens = [(rng.randint(200, 500) + 2 + 7) // 8 * 8 for _ in range(24)]
once = os.environ.get("PG_ATTN_ONCE") == "1"
NSETS = 1 if once else 6
ops = pg.ops
st = ops.stream()
g = torch.Generator(device=DEV).manual_seed(1)


def setup(lens):
    rg = Ragged(lens, S, DEV)
    T, B, ld = rg.T, len(lens), ops.score_ld(S)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, device=DEV, generator=g) * sc).to(BF)  # noqa: E731
    d = dict(rg=rg, T=T, B=B, ld=ld, qkv=rnd(T, 3 * H), tqkv=rnd(T, 3 * H, sc=0.5), dctx=rnd(T, H), gctx=rnd(T, H),
             ctxbar=rnd(T, H), mask=torch.ones(T, device=DEV), ctx=torch.empty(T, H, dtype=BF, device=DEV),
             ctxdot=torch.empty(T, H, dtype=BF, device=DEV), dq=torch.zeros(T, 3 * H, dtype=BF, device=DEV),
             rho=torch.zeros(B * heads, S, device=DEV))
    for name in ("P", "dS", "Sbar", "Pdot"):
        d[name] = [torch.zeros(B * heads, ld, ld, dtype=BF, device=DEV) for _ in range(NSETS)]
    return d


def timeit(fn, n_sets):
    reps = 1 if once else 3
    best = 1e9
    for _ in range(reps):
        for i in range(n_sets):
            fn(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n_sets):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n_sets * 1e3)
    return best


full, part = setup(lens), setup(lens[:8])
for name, d in (("3B = 24 sequences", full), ("B = 8 sequences", part)):
    so, B, T = d["rg"].seq_off.data_ptr(), d["B"], d["T"]
    p = lambda t: t.data_ptr()  # noqa: E731
    score_bytes = sum(((L + 63) // 64 * 64) ** 2 for L in (lens if d is full else lens[:8])) * heads * 2
    fl_unit = sum(L * L for L in (lens if d is full else lens[:8])) * heads * 64 * 2  # one S x S x 64 product
    kern = {
        "fwd_p": (lambda i: lib.pg_attention_fwd_p_ragged(1, p(d["qkv"]), p(d["mask"]), p(d["P"][i]), p(d["ctx"]), so, B, S, 0,
                                                         heads, 3 * H, H, st), 1, 2),
        "bwd_dq": (lambda i: lib.pg_attention_bwd_dq_ragged(1, p(d["qkv"]), p(d["P"][i]), p(d["ctx"]), p(d["dctx"]), p(d["dS"][i]),
                                                            p(d["dq"]), so, B, S, 0, heads, 3 * H, H, 3 * H, st), 2, 2),
        "jvp": (lambda i: lib.pg_attention_jvp_ragged(1, p(d["qkv"]), p(d["tqkv"]), p(d["P"][i]), p(d["ctx"]), p(d["ctxdot"]),
                                                      p(d["rho"]), so, B, S, 0, heads, 3 * H, H, H, st), 1, 4),
        "bwd2": (lambda i: lib.pg_attention_bwd2_ragged(1, p(d["qkv"]), p(d["tqkv"]), p(d["P"][i]), p(d["ctx"]), p(d["ctxdot"]),
                                                        p(d["gctx"]), p(d["ctxbar"]), p(d["rho"]), p(d["Sbar"][i]), p(d["Pdot"][i]),
                                                        so, B, S, 0, heads, 3 * H, H, st), 3, 5),
    }
    delta = torch.empty(B * heads, S, device=DEV)
    kern["bwd, dS kept"] = (lambda i: lib.pg_attention_bwd(1, p(d["qkv"]), p(d["P"][i]), p(d["ctx"]), p(d["dctx"]), p(d["dS"][i]),
                                                           p(d["dq"]), p(delta), so, B, S, 0, heads, 3 * H, H, 3 * H, st), 3, 5)
    kern["bwd, no dS"] = (lambda i: lib.pg_attention_bwd(1, p(d["qkv"]), p(d["P"][i]), p(d["ctx"]), p(d["dctx"]), None,
                                                         p(d["dq"]), p(delta), so, B, S, 0, heads, 3 * H, H, 3 * H, st), 2, 5)
    print(f"{name}: {T} token rows, score-shaped buffer {score_bytes / 1e6:.0f} MB per slot")
    for k, (fn, nbuf, nprod) in kern.items():
        assert fn(0) == 0
        us = timeit(fn, NSETS)
        print(f"  {k:12s} {us:7.1f} us   score-shaped traffic {nbuf * score_bytes / us / 1e6:5.2f} TB/s   "
              f"{nprod * fl_unit / us / 1e6:5.0f} TFLOP/s ({nprod} products)", flush=True)

# the inference forward of the fill loop: 68 sequences, all keys visible / half of them (walk over the visible-key list)
lens68 = [(rng.randint(200, 500) + 2 + 7) // 8 * 8 for _ in range(68)]
rg68 = Ragged(lens68, S, DEV)
T68 = rg68.T
qkv68 = (torch.randn(T68, 3 * H, device=DEV, generator=g)).to(BF)
ctx68 = torch.empty(T68, H, dtype=BF, device=DEV)
fl68 = sum(L * L for L in lens68) * heads * 64 * 2 * 2
for name, frac in (("all keys", 1.0), ("half the keys", 0.5)):
    m = (torch.rand(T68, device=DEV, generator=g) < frac).float() if frac < 1 else torch.ones(T68, device=DEV)
    idx = torch.empty(T68, dtype=torch.int32, device=DEV)
    cnt = torch.empty(68, dtype=torch.int32, device=DEV)
    assert lib.pg_attention_key_lists(m.data_ptr(), rg68.seq_off.data_ptr(), 68, S, idx.data_ptr(), cnt.data_ptr(), st) == 0
    fn = lambda i: lib.pg_attention_fwd_keys(1, qkv68.data_ptr(), m.data_ptr(), ctx68.data_ptr(), rg68.seq_off.data_ptr(), 68, S,  # noqa: E731
                                             heads, 3 * H, H, idx.data_ptr(), cnt.data_ptr(), st)
    assert fn(0) == 0
    us = timeit(fn, 4)
    print(f"inference forward, 68 sequences, {name}: {us:7.1f} us   {fl68 * frac / us / 1e6:5.0f} TFLOP/s", flush=True)
