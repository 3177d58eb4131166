"""Kernel-level parity on a real MI355X: every HIP kernel, called through the C ABI
(ctypes), against closed-form CPU statements (tests/kernel_specs.py, themselves checked
against autograd on the CPU) or the oracle's functions.  Tolerances: fp32 kernels 2e-5
relative to the tensor scale (exact fp32 FMA chains, different summation order than the
CPU); bf16 kernels 2e-2 (inputs rounded to bf16 first, so only accumulation order and the
output rounding differ)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import protbert_cpu as O  # noqa: E402
from tests import kernel_specs as K  # noqa: E402

K_gelu = K.gelu


@pytest.fixture(scope="module")
def pg():
    import pgload
    return pgload.load()


DEV = "cuda"
DTYPES = [torch.float32, torch.bfloat16]


def tol(dtype):
    return 2e-5 if dtype == torch.float32 else 2e-2


def close(got, ref, dtype, scale=None, mult=1.0):
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    s = float(ref.abs().max()) if scale is None else scale
    err = float((got - ref).abs().max())
    assert err <= mult * tol(dtype) * max(s, 1e-6), f"max err {err:.3e} vs scale {s:.3e}"


def rnd(*shape, dtype=torch.float32, scale=1.0, gen=None):
    """Random data already representable in `dtype` (so CPU fp64 references see the same inputs)."""
    t = torch.randn(*shape, generator=gen) * scale
    return t.to(dtype).to(torch.float32)


# ----------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("ak,bk", [(True, True), (True, False), (False, True), (False, False)])
@pytest.mark.parametrize("kfull", [False, True])
@pytest.mark.parametrize("narrow", [False, True])
def test_gemm_exact_layouts(pg, dtype, ak, bk, kfull, narrow):
    """Small-integer operands: every product and partial sum is exact in bf16/fp32, so any
    fragment-layout or transposed-read mistake shows as a wrong integer.  A and B are
    asymmetric; sizes leave ragged M/N/K tails."""
    g = torch.Generator().manual_seed(1)
    epc = 4 if dtype == torch.float32 else 8
    M, N, K = 136 + epc, 200, 72 + epc
    if kfull:  # whole k-tiles: the bf16 direct-to-LDS kernel (ragged K takes the register-staged one)
        K = 192
    if not ak:
        M = 144
    if not bk:
        N = 208
    if narrow:  # N <= 64: the 128x64-tile kernel (bf16, whole k-tiles)
        N = 56
    A = torch.randint(-3, 4, (M, K), generator=g).float()
    B = torch.randint(-3, 4, (N, K), generator=g).float()
    A[:, 0] += torch.arange(M) % 5
    B[:, 1] -= torch.arange(N) % 7
    A[:, K - 1] += torch.arange(M) % 3
    B[:, K // 2] += torch.arange(N) % 4
    ref = A.double() @ B.double().T
    Ad = (A if ak else A.T.contiguous()).to(dtype).to(DEV)
    Bd = (B if bk else B.T.contiguous()).to(dtype).to(DEV)
    out = pg.ops.gemm(Ad, Bd, a_kmajor=ak, b_kmajor=bk, out_dtype=torch.float32)
    assert torch.equal(out.cpu().double(), ref)


@pytest.fixture(params=[(0, 0)], ids=["two_stage"])
def ring(request, pg):
    """The 256x256-tile kernel the library ships: two 64 KiB stages, one workgroup per tile.  (A five-slot ring and a
    persistent form were measured against it and lost; their sources are kept under scripts/probes/.)"""
    yield request.param


@pytest.mark.parametrize("ak,bk", [(True, True), (True, False), (False, True), (False, False)])
@pytest.mark.parametrize("shape", [(520, 776, 192), (256, 256, 64), (264, 520, 320), (304, 264, 1088), (256, 512, 128)])
def test_gemm_wide_tile_exact(pg, ring, ak, bk, shape):
    """The 256x256-tile bf16 kernel (forced; normally only large-N products take it): exact
    integers over several tiles with ragged M and N edges, all four operand layouts, and both
    output types."""
    g = torch.Generator().manual_seed(11)
    M, N, K = shape
    A = torch.randint(-3, 4, (M, K), generator=g).float()
    B = torch.randint(-3, 4, (N, K), generator=g).float()
    A[:, 0] += torch.arange(M) % 5
    B[:, 1] -= torch.arange(N) % 7
    A[:, K - 1] += torch.arange(M) % 3
    B[:, K // 2] += torch.arange(N) % 4
    ref = A.double() @ B.double().T
    Ad = (A if ak else A.T.contiguous()).to(torch.bfloat16).to(DEV)
    Bd = (B if bk else B.T.contiguous()).to(torch.bfloat16).to(DEV)
    pg._lib.lib.pg_tune(2, 2)
    try:
        out = pg.ops.gemm(Ad, Bd, a_kmajor=ak, b_kmajor=bk, out_dtype=torch.float32)
        out16 = pg.ops.gemm(Ad, Bd, a_kmajor=ak, b_kmajor=bk, alpha=1.0 / 64)
    finally:
        pg._lib.lib.pg_tune(2, 1)
    assert torch.equal(out.cpu().double(), ref)
    close(out16, ref / 64, torch.bfloat16)


def test_gemm_wide_tile_epilogues(pg, ring):
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(12)
    M, N, Kd = 520, 512, 256
    A, B = rnd(M, Kd, dtype=dtype, gen=g), rnd(N, Kd, dtype=dtype, scale=0.05, gen=g)
    bias = torch.randn(N, generator=g)
    res = rnd(M, N, dtype=dtype, gen=g)
    U = rnd(M, N, dtype=dtype, gen=g)
    Ad, Bd = A.to(dtype).to(DEV), B.to(dtype).to(DEV)
    base = A.double() @ B.double().T
    pg._lib.lib.pg_tune(2, 2)
    try:
        out = pg.ops.gemm(Ad, Bd, bias=bias.to(DEV), residual=res.to(dtype).to(DEV))
        close(out, base + bias.double() + res.double(), dtype)
        aux = torch.empty(M, N, dtype=dtype, device=DEV)
        out = pg.ops.gemm(Ad, Bd, bias=bias.to(DEV), act=pg._lib.PG_ACT_GELU, aux=aux)
        close(aux, base + bias.double(), dtype)
        close(out, K.gelu(base + bias.double()), dtype)
        aux2 = torch.empty(M, N, dtype=dtype, device=DEV)
        out = pg.ops.gemm(Ad, Bd, act=pg._lib.PG_ACT_MUL_DGELU, aux=U.to(dtype).to(DEV), aux2=aux2)
        close(aux2, base, dtype)
        close(out, base * K.gelu_d1(U.double()), dtype)
        C0 = torch.randn(M, N, generator=g)
        Cd = C0.clone().to(DEV)
        pg.ops.gemm(Ad, Bd, Cd, alpha=0.5, beta=1.0)
        close(Cd, C0.double() + 0.5 * base, dtype)
    finally:
        pg._lib.lib.pg_tune(2, 1)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_epilogues(pg, dtype):
    g = torch.Generator().manual_seed(2)
    M, N, Kd = 200, 264, 320
    A, B = rnd(M, Kd, dtype=dtype, gen=g), rnd(N, Kd, dtype=dtype, scale=0.05, gen=g)
    bias = torch.randn(N, generator=g)
    res = rnd(M, N, dtype=dtype, gen=g)
    Ad, Bd = A.to(dtype).to(DEV), B.to(dtype).to(DEV)
    base = A.double() @ B.double().T
    # bias + residual
    out = pg.ops.gemm(Ad, Bd, bias=bias.to(DEV), residual=res.to(dtype).to(DEV))
    close(out, base + bias.double() + res.double(), dtype)
    # bias + GELU with the pre-activation stored
    aux = torch.empty(M, N, dtype=dtype, device=DEV)
    out = pg.ops.gemm(Ad, Bd, bias=bias.to(DEV), act=pg._lib.PG_ACT_GELU, aux=aux)
    pre = base + bias.double()
    close(aux, pre, dtype)
    close(out, K.gelu(pre), dtype)
    # multiply by gelu'(U), storing the unmultiplied value
    U = rnd(M, N, dtype=dtype, gen=g)
    aux2 = torch.empty(M, N, dtype=dtype, device=DEV)
    out = pg.ops.gemm(Ad, Bd, act=pg._lib.PG_ACT_MUL_DGELU, aux=U.to(dtype).to(DEV), aux2=aux2)
    close(aux2, base, dtype)
    close(out, base * K.gelu_d1(U.double()), dtype)
    # alpha / beta accumulation into an fp32 C (the weight-gradient form)
    C0 = torch.randn(M, N, generator=g)
    Cd = C0.clone().to(DEV)
    pg.ops.gemm(Ad, Bd, Cd, alpha=0.5, beta=1.0)
    close(Cd, C0.double() + 0.5 * base, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_weight_gradient_shape(pg, dtype):
    """dW = dY^T X with a token count that is not a multiple of the k tile."""
    g = torch.Generator().manual_seed(3)
    T, No, Ni = 328, 192, 128
    dY, X = rnd(T, No, dtype=dtype, gen=g), rnd(T, Ni, dtype=dtype, gen=g)
    out = pg.ops.gemm(dY.to(dtype).to(DEV), X.to(dtype).to(DEV), a_kmajor=False, b_kmajor=False, out_dtype=torch.float32)
    close(out, dY.double().T @ X.double(), dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("T", [256, 328])
def test_gemm_pair_weight_gradients(pg, dtype, T):
    """pg_gemm_pair: two accumulating weight-gradient products (different output shapes) in one
    launch equal two pg_gemm calls.  T = 256 takes the single-launch bf16 kernel, T = 328 (not a
    whole number of k-tiles) and fp32 the two-launch fallback."""
    g = torch.Generator().manual_seed(31)
    dY0, X0 = rnd(T, 136, dtype=dtype, gen=g), rnd(T, 128, dtype=dtype, gen=g)
    dY1, X1 = rnd(T, 392, dtype=dtype, gen=g), rnd(T, 200, dtype=dtype, gen=g)
    C0, C1 = torch.randn(136, 128, generator=g), torch.randn(392, 200, generator=g)
    d = lambda t: t.to(dtype).to(DEV)  # noqa: E731
    c0, c1 = C0.clone().to(DEV), C1.clone().to(DEV)
    pg.ops.gemm_pair(d(dY0), d(X0), c0, d(dY1), d(X1), c1, a_kmajor=False, b_kmajor=False, alpha=1.0, beta=1.0)
    close(c0, C0.double() + dY0.double().T @ X0.double(), dtype)
    close(c1, C1.double() + dY1.double().T @ X1.double(), dtype)
    e0, e1 = C0.clone().to(DEV), C1.clone().to(DEV)
    pg.ops.gemm(d(dY0), d(X0), e0, a_kmajor=False, b_kmajor=False, beta=1.0)
    pg.ops.gemm(d(dY1), d(X1), e1, a_kmajor=False, b_kmajor=False, beta=1.0)
    assert torch.equal(c0, e0) and torch.equal(c1, e1)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("T,T2", [(256, 192), (328, 64)])
def test_gemm_two_reduction_segments(pg, dtype, T, T2):
    """dW += dY^T X + dYb^T Xb as one descriptor (the second-order pass's weight gradients):
    whole-k-tile bf16 walks both segments in one launch, everything else falls back to two."""
    g = torch.Generator().manual_seed(32)
    dY, X = rnd(T, 136, dtype=dtype, gen=g), rnd(T, 200, dtype=dtype, gen=g)
    dYb, Xb = rnd(T2, 136, dtype=dtype, gen=g), rnd(T2, 200, dtype=dtype, gen=g)
    C0 = torch.randn(136, 200, generator=g)
    d = lambda t: t.to(dtype).to(DEV)  # noqa: E731
    c = C0.clone().to(DEV)
    pg.ops.gemm(d(dY), d(X), c, a_kmajor=False, b_kmajor=False, beta=1.0, A2=d(dYb), B2=d(Xb))
    close(c, C0.double() + dY.double().T @ X.double() + dYb.double().T @ Xb.double(), dtype)
    args = dict(a_kmajor=False, b_kmajor=False, bias=torch.ones(56, device=DEV), A2=d(dYb)[:, :56].contiguous(),
                B2=d(Xb)[:, :56].contiguous())
    if dtype == torch.float32 or T % 64 or T2 % 64:
        with pytest.raises(pg._lib.PgError):  # two launches: an epilogue cannot be split over them
            pg.ops.gemm(d(dY)[:, :56].contiguous(), d(X)[:, :56].contiguous(), None, **args)
    else:  # one launch walks both segments, so a bias is fine
        out = pg.ops.gemm(d(dY)[:, :56].contiguous(), d(X)[:, :56].contiguous(), None, out_dtype=torch.float32, **args)
        close(out, dY[:, :56].double().T @ X[:, :56].double() + dYb[:, :56].double().T @ Xb[:, :56].double() + 1.0, dtype)


def test_gemm_wide_tile_split_last_round(pg, ring):
    """A large-N product whose 256x256 tiles fill the CUs once plus a few: the rows past the last
    full round of tiles go to a second launch.  Exact integers plus every row-offset epilogue
    operand (residual, stored pre-activation) across the seam."""
    g = torch.Generator().manual_seed(41)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    N, K = 2048, 128
    M = (cus // 8 + 1) * 256 + 40  # one full round of tiles, then one and a bit tile rows
    A = torch.randint(-3, 4, (M, K), generator=g).float()
    B = torch.randint(-3, 4, (N, K), generator=g).float()
    A[:, 0] += torch.arange(M) % 5
    B[:, 1] -= torch.arange(N) % 7
    ref = A.double() @ B.double().T
    bf = torch.bfloat16
    Ad, Bd = A.to(bf).to(DEV), B.to(bf).to(DEV)
    out = pg.ops.gemm(Ad, Bd, out_dtype=torch.float32)
    assert torch.equal(out.cpu().double(), ref)
    res = torch.randint(-8, 9, (M, N), generator=g).float()
    bias = torch.randint(-4, 5, (N,), generator=g).float()
    aux = torch.empty(M, N, dtype=bf, device=DEV)
    out = pg.ops.gemm(Ad, Bd, alpha=1.0 / 64, bias=bias.to(DEV), residual=res.to(bf).to(DEV))
    close(out, ref / 64 + bias.double() + res.double(), bf)
    out = pg.ops.gemm(Ad, Bd, alpha=1.0 / 64, bias=bias.to(DEV), act=pg._lib.PG_ACT_GELU, aux=aux)
    close(aux, ref / 64 + bias.double(), bf)
    close(out, K_gelu(ref / 64 + bias.double()), bf)


@pytest.mark.parametrize("kernel", ["128", "256", "n64"])
@pytest.mark.parametrize("M", [8, 72, 136, 250])
def test_gemm_lower_edge_rows(pg, kernel, M):
    """A tile that hangs over the lower edge of C leaves through the vector epilogue with the rows past M unstored:
    for every epilogue option, rows [0, M) carry the same bits as the same rows of a product whose M fills whole
    tiles, and the rows of the output buffers behind M keep what they held (outputs are views of taller buffers)."""
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(900 + M)
    Mfull, N, Kd = 512, (64 if kernel == "n64" else 512), 192
    A = rnd(Mfull, Kd, dtype=bf, gen=g).to(bf).to(DEV)
    B = rnd(N, Kd, dtype=bf, scale=0.05, gen=g).to(bf).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    res = rnd(Mfull, N, dtype=bf, gen=g).to(bf).to(DEV)
    U = rnd(Mfull, N, dtype=bf, gen=g).to(bf).to(DEV)
    C0 = torch.randn(Mfull, N, generator=g)
    SENT = 12345.0

    def run(m):
        outs = {}
        tall = lambda dt=bf: torch.full((Mfull, N), SENT, dtype=dt, device=DEV)  # noqa: E731
        c = tall(); pg.ops.gemm(A[:m], B, c[:m], bias=bias, residual=res[:m]); outs["bias_res"] = c
        c, a = tall(), tall(); pg.ops.gemm(A[:m], B, c[:m], bias=bias, act=pg._lib.PG_ACT_GELU, aux=a[:m])
        outs["gelu"], outs["gelu_aux"] = c, a
        c, a = tall(), tall(); pg.ops.gemm(A[:m], B, c[:m], act=pg._lib.PG_ACT_MUL_DGELU, aux=U[:m], aux2=a[:m])
        outs["dgelu"], outs["dgelu_aux2"] = c, a
        c = tall(torch.float32); c[:m] = C0[:m].to(DEV); pg.ops.gemm(A[:m], B, c[:m], alpha=0.5, beta=1.0); outs["acc_f32"] = c
        c = tall(); c[:m] = C0[:m].to(bf).to(DEV); pg.ops.gemm(A[:m], B, c[:m], beta=1.0); outs["acc_bf16"] = c
        return outs

    pg._lib.lib.pg_tune(2, 2 if kernel == "256" else (0 if kernel == "128" else 1))
    try:
        edge, full = run(M), run(Mfull)
    finally:
        pg._lib.lib.pg_tune(2, 1)
    for name in edge:
        assert torch.equal(edge[name][:M], full[name][:M]), name
        assert bool((edge[name][M:] == SENT).all()), name


def test_gemm_split_last_round_two_segments(pg, ring):
    """The same split with a second reduction segment and a k-major A: the rows that go to the second
    launch must take the same rows of A2.  Exact integers against two plain products."""
    g = torch.Generator().manual_seed(43)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    N, K, K2 = 1024, 1024, 128
    M = (cus // 4) * 256 + 296  # one full round of 256x256 tiles (N = 4 tiles wide), then a sparse one
    A = torch.randint(-2, 3, (M, K), generator=g).float()
    B = torch.randint(-2, 3, (N, K), generator=g).float()
    A2 = torch.randint(-2, 3, (M, K2), generator=g).float()
    B2 = torch.randint(-2, 3, (N, K2), generator=g).float()
    A2[:, 0] += torch.arange(M) % 5  # rows of the second segment are told apart
    bf = torch.bfloat16
    # (A2, B2) share the leading dimensions of (A, B): views into wider buffers
    A2w = torch.zeros(M, K, dtype=bf)
    A2w[:, :K2] = A2.to(bf)
    B2w = torch.zeros(N, K, dtype=bf)
    B2w[:, :K2] = B2.to(bf)
    out = pg.ops.gemm(A.to(bf).to(DEV), B.to(bf).to(DEV), out_dtype=torch.float32, A2=A2w.to(DEV)[:, :K2],
                      B2=B2w.to(DEV)[:, :K2])
    ref = A.double() @ B.double().T + A2.double() @ B2.double().T
    assert torch.equal(out.cpu().double(), ref)


@pytest.mark.parametrize("ak,bk", [(True, True), (True, False), (False, True), (False, False)])
@pytest.mark.parametrize("shape", [(256, 1024, 4096), (200, 392, 1024), (128, 128, 2048), (640, 1024, 1536)])
def test_gemm_small_grid_exact(pg, ak, bk, shape):
    """With pg_tune key 1 = 2, products of at most 256 tiles of 128x128 take the 128x64-tile kernel: small-integer
    operands, exact against fp64, ragged M and N edges, every layout; equal to the 128x128 kernel's result."""
    from protgen_gan_amd._lib import lib
    g = torch.Generator().manual_seed(77)
    M, N, K = shape
    if not ak:
        M = (M + 7) // 8 * 8
    A = torch.randint(-2, 3, (M, K), generator=g).float()
    B = torch.randint(-2, 3, (N, K), generator=g).float()
    A[:, 0] += torch.arange(M) % 5
    B[:, K - 1] -= torch.arange(N) % 3
    A[:, K // 2 + 1] += torch.arange(M) % 4
    ref = A.double() @ B.double().T
    bf = torch.bfloat16
    Ad = (A if ak else A.T.contiguous()).to(bf).to(DEV)
    Bd = (B if bk else B.T.contiguous()).to(bf).to(DEV)
    lib.pg_tune(1, 2)
    try:
        out = pg.ops.gemm(Ad, Bd, a_kmajor=ak, b_kmajor=bk, out_dtype=torch.float32)
    finally:
        lib.pg_tune(1, 1)
    assert torch.equal(out.cpu().double(), ref)
    plain = pg.ops.gemm(Ad, Bd, a_kmajor=ak, b_kmajor=bk, out_dtype=torch.float32)
    assert torch.equal(plain, out)


def test_gemm_small_grid_epilogues(pg):
    """bias + GELU (with the pre-activation stored), bias + residual and the GELU-derivative epilogue on the
    128x64-tile kernel for a 256-row product, bf16 outputs, against the 128x128 kernel (same arithmetic)."""
    from protgen_gan_amd._lib import lib
    g = torch.Generator().manual_seed(78)
    bf = torch.bfloat16
    M, N, K = 256, 1024, 1024
    A = (torch.randn(M, K, generator=g) * 0.5).to(bf).to(DEV)
    B = (torch.randn(N, K, generator=g) * 0.05).to(bf).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    res = torch.randn(M, N, generator=g).to(bf).to(DEV)
    u = torch.randn(M, N, generator=g).to(bf).to(DEV)

    def run():
        o1, aux = torch.empty(M, N, dtype=bf, device=DEV), torch.empty(M, N, dtype=bf, device=DEV)
        pg.ops.gemm(A, B, o1, bias=bias, act=1, aux=aux)
        o2 = pg.ops.gemm(A, B, bias=bias, residual=res)
        o3, gu = torch.empty(M, N, dtype=bf, device=DEV), torch.empty(M, N, dtype=bf, device=DEV)
        pg.ops.gemm(A, B, o3, act=2, aux=u, aux2=gu)
        torch.cuda.synchronize()
        return [o1, aux, o2, o3, gu]

    lib.pg_tune(1, 2)
    try:
        small = run()
    finally:
        lib.pg_tune(1, 1)
    plain = run()
    for a, c in zip(small, plain):
        assert torch.equal(a, c)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("T", [128, 200])
def test_gemm_group_weight_gradients(pg, dtype, T):
    """pg_gemm_group: four accumulating weight-gradient products of different shapes (enough
    256x256 tiles between them to take the grouped kernel in bf16 when T is whole k-tiles), two of
    them with a second reduction segment, against one pg_gemm per product."""
    g = torch.Generator().manual_seed(33)
    shapes = [(2056, 2296), (2296, 2056), (1536, 520), (520, 520)]  # 81 + 81 + 18 + 9 tiles of 256x256
    d = lambda t: t.to(dtype).to(DEV)  # noqa: E731
    probs, refs, singles = [], [], []
    for i, (No, Ni) in enumerate(shapes):
        dY, X = rnd(T, No, dtype=dtype, gen=g), rnd(T, Ni, dtype=dtype, gen=g)
        C0 = torch.randn(No, Ni, generator=g)
        ref = C0.double() + dY.double().T @ X.double()
        pr = [d(dY), d(X), C0.clone().to(DEV)]
        if i % 2 == 0:
            dYb, Xb = rnd(64, No, dtype=dtype, gen=g), rnd(64, Ni, dtype=dtype, gen=g)
            ref = ref + dYb.double().T @ Xb.double()
            pr += [d(dYb), d(Xb)]
        probs.append(tuple(pr))
        refs.append(ref)
        e = C0.clone().to(DEV)
        pg.ops.gemm(pr[0], pr[1], e, a_kmajor=False, b_kmajor=False, beta=1.0, **({"A2": pr[3], "B2": pr[4]} if len(pr) == 5 else {}))
        singles.append(e)
    outs = pg.ops.gemm_group(probs, a_kmajor=False, b_kmajor=False, alpha=1.0, beta=1.0)
    for o, r, e in zip(outs, refs, singles):
        close(o, r, dtype)
        assert torch.equal(o, e)


def test_gemm_rejects_bad_arguments(pg):
    with pytest.raises(pg._lib.PgError):  # fp32 leading dimension not a multiple of 4 elements
        pg.ops.gemm(torch.zeros(64, 34, device=DEV), torch.zeros(64, 34, device=DEV))
    flat = torch.zeros(64 * 36 + 1, device=DEV)
    with pytest.raises(pg._lib.PgError):  # operand not 16-byte aligned
        pg.ops.gemm(flat[1:].view(64, 36), torch.zeros(64, 36, device=DEV))
    with pytest.raises(pg._lib.PgError):  # m-major operand whose free extent is not a multiple of the chunk
        pg.ops.gemm(torch.zeros(36, 66, device=DEV)[:, :62], torch.zeros(64, 36, device=DEV), a_kmajor=False)


# ----------------------------------------------------------------------------- LayerNorm family
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("H", [64, 128, 256, 1024])
def test_layernorm_family(pg, dtype, H):
    g = torch.Generator().manual_seed(4)
    R, eps = 37, 1e-12
    x, xdot, gy, ybar = (rnd(R, H, dtype=dtype, gen=g) for _ in range(4))
    gam = 1 + 0.1 * torch.randn(H, generator=g)
    bet = 0.1 * torch.randn(H, generator=g)
    d = lambda t: t.to(dtype).to(DEV)  # noqa: E731
    gd, bd = gam.to(DEV), bet.to(DEV)
    close(pg.ops.layernorm(d(x), gd, bd, eps), K.ln_fwd(x.double(), gam.double(), bet.double(), eps), dtype)
    dg = torch.full((H,), 0.5, device=DEV)
    db = torch.full((H,), -0.25, device=DEV)
    dxs = torch.full((H,), 2.0, device=DEV)
    dx = pg.ops.layernorm_bwd(d(x), d(gy), gd, eps, dg, db, dxs)
    rdx, rdg, rdb = K.ln_bwd(x.double(), gy.double(), gam.double(), eps)
    close(dx, rdx, dtype)
    close(dg, rdg + 0.5, torch.float32, mult=10)
    close(db, rdb - 0.25, torch.float32, mult=10)
    # column sums of the output (the bias gradient of the Linear in front): sums of the fp32 values
    close(dxs, rdx.sum(0) + 2.0, torch.float32, scale=float(rdx.abs().sum(0).max()), mult=10)
    only = torch.zeros(H, device=DEV)
    pg.ops.layernorm_bwd(d(x), d(gy), gd, eps, None, None, only)
    close(only, rdx.sum(0), torch.float32, scale=float(rdx.abs().sum(0).max()), mult=10)
    close(pg.ops.layernorm_jvp(d(x), d(xdot), gd, eps), K.ln_jvp(x.double(), xdot.double(), gam.double(), eps), dtype)
    for yb in (ybar, None):
        dg.zero_()
        db.zero_()
        dxs.zero_()
        xb = pg.ops.layernorm_bwd2(d(x), d(xdot), d(gy), None if yb is None else d(yb), gd, eps, dg, db, dxs)
        rx, rg, rb = K.ln_bwd2(x.double(), xdot.double(), gy.double(), None if yb is None else yb.double(), gam.double(), eps)
        close(xb, rx, dtype, mult=2)
        close(dxs, rx.sum(0), torch.float32, scale=float(rx.abs().sum(0).max()), mult=20)
        close(dg, rg, torch.float32, mult=20)
        close(db, rb, torch.float32, scale=max(float(rb.abs().max()), 1.0), mult=20)


def test_layernorm_many_rows(pg):
    """More rows than partial-sum slots: the grid-stride path of the parameter gradients."""
    g = torch.Generator().manual_seed(5)
    R, H, eps = 2500, 128, 1e-12
    x, gy = torch.randn(R, H, generator=g), torch.randn(R, H, generator=g)
    gam = 1 + 0.1 * torch.randn(H, generator=g)
    dg, db = torch.zeros(H, device=DEV), torch.zeros(H, device=DEV)
    dx = pg.ops.layernorm_bwd(x.to(DEV), gy.to(DEV), gam.to(DEV), eps, dg, db)
    rdx, rdg, rdb = K.ln_bwd(x.double(), gy.double(), gam.double(), eps)
    close(dx, rdx, torch.float32)
    close(dg, rdg, torch.float32, mult=10)
    close(db, rdb, torch.float32, mult=10)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gelu_bwd2_and_colsum(pg, dtype):
    g = torch.Generator().manual_seed(6)
    u, udot, ggu, gub = (rnd(50, 96, dtype=dtype, gen=g) for _ in range(4))
    d = lambda t: t.to(dtype).to(DEV)  # noqa: E731
    close(pg.ops.gelu_bwd2(d(u), d(udot), d(ggu), d(gub)), K.gelu_bwd2(u.double(), udot.double(), ggu.double(), gub.double()), dtype)
    close(pg.ops.gelu_bwd2(d(u), None, None, d(gub)), K.gelu_bwd2(u.double(), None, None, gub.double()), dtype)
    dY = rnd(777, 200, dtype=dtype, gen=g)
    db = torch.ones(200, device=DEV)
    pg.ops.colsum_acc(d(dY), db)
    close(db, dY.double().sum(0) + 1, torch.float32, mult=20)
    # a column count that is not a multiple of the 16-byte vector takes the scalar kernel
    db = torch.ones(197, device=DEV)
    pg.ops.colsum_acc(d(dY)[:, :197], db)
    close(db, dY[:, :197].double().sum(0) + 1, torch.float32, mult=20)
    # wide and tall: several column blocks and row chunks of the vector kernel
    dY = rnd(4100, 1032, dtype=dtype, gen=g)
    db = torch.zeros(1032, device=DEV)
    pg.ops.colsum_acc(d(dY), db)
    close(db, dY.double().sum(0), torch.float32, mult=200)


# ----------------------------------------------------------------------------- softmax family
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("S", [40, 64, 200])
def test_softmax_family(pg, dtype, S):
    g = torch.Generator().manual_seed(7)
    B, nh = 2, 2
    Z, ld = B * nh, pg.ops.score_ld(S)
    sc = torch.zeros(Z, S, ld)
    sc[:, :, :S] = torch.randn(Z, S, S, generator=g) * 2
    sc[:, :, S:] = float("nan")  # padding columns must never be read
    mask = torch.ones(B, S)
    mask[0, S - 7:] = 0
    mask[1, 3] = 0
    ext = O.extended_mask(mask, torch.float64)  # [B,1,1,S]
    ref = torch.softmax(sc[:, :, :S].double().view(B, nh, S, S) + ext, -1).view(Z, S, S)
    P = pg.ops.softmax_fwd(sc.to(DEV), mask.to(DEV), B, nh, S, dtype)
    close(P[:, :, :S], ref, dtype)
    assert float(P[:, :, S:].float().abs().max() if ld > S else 0.0) == 0.0
    assert float(P.float()[0, :, S - 7:S].abs().max()) == 0.0  # masked keys get exactly zero
    # backward / tangent / second order use the device's own P so only the new op is measured
    Pd = P.float().cpu().double()[:, :, :S]
    dP, Sdot, gP, X = (torch.randn(Z, S, S, generator=g) for _ in range(4))

    def pad(t):
        o = torch.full((Z, S, ld), float("nan"))
        o[:, :, :S] = t
        return o.to(DEV)

    dS = pg.ops.softmax_bwd(pad(dP), P, S)
    close(dS[:, :, :S], K.softmax_bwd(dP.double(), Pd), dtype)
    Sd_out, Pdot = pg.ops.softmax_jvp(pad(Sdot), P, S)
    close(Pdot[:, :, :S], K.softmax_jvp(Sdot.double(), Pd), dtype)
    close(Sd_out[:, :, :S], Sdot.double(), dtype)
    Sd_in = Sd_out.float().cpu().double()[:, :, :S]
    Sbar, Pdot2 = pg.ops.softmax_bwd2(P, Sd_out, pad(gP), pad(X), S)
    rS, rP = K.softmax_bwd2(Pd, Sd_in, gP.double(), X.double())
    close(Sbar[:, :, :S], rS, dtype, mult=2)
    close(Pdot2[:, :, :S], rP, dtype)
    if ld > S:
        assert float(Sbar[:, :, S:].float().abs().max()) == 0.0


def test_fused_attention_forward_packed_rows(pg):
    """Sequences of different lengths laid back to back (1, a few, one query tile, several tiles, not
    multiples of anything) against eager attention per sequence in fp64; the grid is sized for the longest,
    the query tiles beyond a shorter sequence's end leave without touching memory."""
    g = torch.Generator().manual_seed(77)
    nh = 3
    H = nh * 64
    lens = [1, 7, 128, 130, 333, 64, 497]
    off = [0]
    for n in lens:
        off.append(off[-1] + n)
    T = off[-1]
    qkv = rnd(T, 3 * H, dtype=torch.bfloat16, gen=g)
    qkv[:, :H] *= 2.0
    mask = (torch.rand(T, generator=g) > 0.2).float()
    mask[off[3]:off[4]] = 1.0  # one sequence fully visible
    canary = 12288.0  # representable in bf16
    for m in (mask, None):
        ref = torch.empty(T, H, dtype=torch.float64)
        for a, b in zip(off, off[1:]):
            q, k, v = (qkv[a:b, i * H:(i + 1) * H].double().view(b - a, nh, 64).transpose(0, 1) for i in range(3))
            sc = q @ k.transpose(1, 2) * 0.125
            if m is not None:
                sc = sc + O.extended_mask(m[a:b].view(1, -1), torch.float64).view(1, 1, -1)
            ref[a:b] = (torch.softmax(sc, -1) @ v).transpose(0, 1).reshape(b - a, H)
        out = pg.ops.attention_fwd_packed(qkv.to(torch.bfloat16).to(DEV), None if m is None else m.to(DEV),
                                          torch.tensor(off, dtype=torch.int32, device=DEV), max(lens), nh)
        close(out, ref, torch.bfloat16)
    # rows after the last sequence are not written
    pad = torch.full((T + 64, 3 * H), 0.5, dtype=torch.bfloat16)
    pad[:T] = qkv
    big = pad.to(DEV)
    ctx = torch.full((T + 64, H), canary, dtype=torch.bfloat16, device=DEV)
    from protgen_gan_amd._lib import check, lib
    so = torch.tensor(off, dtype=torch.int32, device=DEV)
    check(lib.pg_attention_fwd_packed(1, big.data_ptr(), None, ctx.data_ptr(), so.data_ptr(), len(lens), max(lens), nh,
                                      big.stride(0), ctx.stride(0), pg.ops.stream()), "pg_attention_fwd_packed")
    assert bool((ctx[T:].float() == canary).all()) and bool((ctx[:T].float() != canary).all())


def test_fused_attention_forward_visible_keys(pg):
    """The forward over the visible keys only (key lists built on the device) against eager attention in fp64 with the
    additive mask: sparse masks (2 visible keys of 500: the first fill iteration), dense ones, a fully visible
    sequence, a sequence with NO visible key (walked whole: uniform weights, as the reference's softmax gives) and a
    mask with a value that is neither 0 nor 1 (walked whole under the additive mask); packed rows and a rectangle."""
    g = torch.Generator().manual_seed(91)
    nh = 2
    H = nh * 64
    lens = [500, 333, 64, 130, 7, 200, 97]
    off = [0]
    for n in lens:
        off.append(off[-1] + n)
    T = off[-1]
    qkv = rnd(T, 3 * H, dtype=torch.bfloat16, gen=g)
    qkv[:, :H] *= 2.0
    mask = (torch.rand(T, generator=g) > 0.5).float()
    mask[off[0]:off[1]] = 0.0
    mask[off[0]] = 1.0
    mask[off[0] + 417] = 1.0            # two visible keys of 500
    mask[off[2]:off[3]] = 1.0           # fully visible
    mask[off[3]:off[4]] = 0.0           # nothing visible
    mask[off[5] + 3] = 0.5              # not a 0/1 mask
    ref = torch.empty(T, H, dtype=torch.float64)
    for a, b in zip(off, off[1:]):
        q, k, v = (qkv[a:b, i * H:(i + 1) * H].double().view(b - a, nh, 64).transpose(0, 1) for i in range(3))
        sc = q @ k.transpose(1, 2) * 0.125 + O.extended_mask(mask[a:b].view(1, -1), torch.float64).view(1, 1, -1)
        ref[a:b] = (torch.softmax(sc, -1) @ v).transpose(0, 1).reshape(b - a, H)
    so = torch.tensor(off, dtype=torch.int32, device=DEV)
    out, cnt = pg.ops.attention_fwd_visible(qkv.to(torch.bfloat16).to(DEV), mask.to(DEV), nh, seq_off=so, max_len=max(lens))
    want = [int(mask[a:b].sum()) for a, b in zip(off, off[1:])]
    want[3] = -1
    want[5] = -1
    assert cnt.tolist() == want
    close(out, ref, torch.bfloat16)
    # rectangle: 3 x 200
    B, S = 3, 200
    qkv2 = rnd(B * S, 3 * H, dtype=torch.bfloat16, gen=g)
    m2 = (torch.rand(B, S, generator=g) > 0.7).float()
    m2[:, 0] = 1.0
    q, k, v = (qkv2[:, i * H:(i + 1) * H].double().view(B, S, nh, 64).transpose(1, 2) for i in range(3))
    ref2 = (torch.softmax(q @ k.transpose(2, 3) * 0.125 + O.extended_mask(m2, torch.float64), -1) @ v).transpose(1, 2).reshape(B * S, H)
    out2, cnt2 = pg.ops.attention_fwd_visible(qkv2.to(torch.bfloat16).to(DEV), m2.to(DEV).contiguous(), nh, B=B, S=S)
    assert cnt2.tolist() == [int(x) for x in m2.sum(1)]
    close(out2, ref2, torch.bfloat16)


def test_embed_ln_packed_positions(pg):
    """Embedding + LayerNorm with explicit positions equals the rectangular kernel on the same (id, position) pairs."""
    from protgen_gan_amd._lib import check, lib
    g = torch.Generator().manual_seed(78)
    B, S, Hd, V, P = 3, 20, 128, 30, 64
    word, pos, typ = rnd(V, Hd, gen=g).to(DEV), rnd(P, Hd, gen=g).to(DEV), rnd(2, Hd, gen=g).to(DEV)
    gam, bet = (1 + 0.1 * rnd(Hd, gen=g)).to(DEV), rnd(Hd, gen=g).to(DEV)
    ids = torch.randint(0, V, (B, S), generator=g).to(DEV)
    dense = torch.empty(B, S, Hd, dtype=torch.bfloat16, device=DEV)
    check(lib.pg_embed_ln2(1, ids.data_ptr(), word.data_ptr(), pos.data_ptr(), typ.data_ptr(), gam.data_ptr(), bet.data_ptr(),
                           1e-12, dense.data_ptr(), None, B, S, Hd, pg.ops.stream()), "pg_embed_ln2")
    widths = [20, 11, 3]
    idx = torch.cat([torch.arange(w) + b * S for b, w in enumerate(widths)]).to(DEV)
    pid = torch.cat([torch.arange(w) for w in widths]).to(torch.int32).to(DEV)
    packed = torch.empty(idx.numel(), Hd, dtype=torch.bfloat16, device=DEV)
    ids_p = ids.view(-1)[idx].contiguous()
    check(lib.pg_embed_ln_packed(1, ids_p.data_ptr(), pid.data_ptr(), word.data_ptr(), pos.data_ptr(), typ.data_ptr(),
                                 gam.data_ptr(), bet.data_ptr(), 1e-12, packed.data_ptr(), idx.numel(), Hd, pg.ops.stream()),
          "pg_embed_ln_packed")
    assert torch.equal(packed, dense.view(B * S, Hd)[idx])


@pytest.mark.parametrize("S", [40, 64, 200, 512, 497])
def test_fused_attention_forward(pg, S):
    """Flash-style forward vs eager attention (HF eager_attention_forward) in fp64 on the same
    bf16-rounded q, k, v; ragged S, masked keys, a fully visible row."""
    g = torch.Generator().manual_seed(70 + S)
    B, nh = 2, 3
    H = nh * 64
    qkv = rnd(B * S, 3 * H, dtype=torch.bfloat16, gen=g)
    qkv[:, :H] *= 2.0  # sharper softmax
    mask = torch.ones(B, S)
    mask[0, S - 9:] = 0
    mask[1, 5] = 0
    q, k, v = (qkv[:, i * H:(i + 1) * H].double().view(B, S, nh, 64).transpose(1, 2) for i in range(3))
    sc = q @ k.transpose(2, 3) * 0.125 + O.extended_mask(mask, torch.float64)
    ref = (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(B * S, H)
    for m in (mask, None):
        if m is None:
            ref = (torch.softmax(q @ k.transpose(2, 3) * 0.125, -1) @ v).transpose(1, 2).reshape(B * S, H)
        out = pg.ops.attention_fwd(qkv.to(torch.bfloat16).to(DEV), None if m is None else m.to(DEV), B, S, nh)
        close(out, ref, torch.bfloat16)
        out2, P = pg.ops.attention_fwd_p(qkv.to(torch.bfloat16).to(DEV), None if m is None else m.to(DEV), B, S, nh)
        close(out2, ref, torch.bfloat16)
        pref = torch.softmax(q @ k.transpose(2, 3) * 0.125 + (O.extended_mask(m, torch.float64) if m is not None else 0), -1)
        close(P[:, :, :S], pref.reshape(B * nh, S, S), torch.bfloat16)
        if P.shape[2] > S:
            assert float(P[:, :, S:].float().abs().max()) == 0.0


@pytest.fixture
def attention_kernels(pg, request):
    """pg_tune key 25: 1 = K / V resident in LDS up to 512 keys (what runs by default), 0 = the ring kernels at every length."""
    pg._lib.lib.pg_tune(25, request.param)
    yield request.param
    pg._lib.lib.pg_tune(25, 1)


@pytest.mark.parametrize("S,attention_kernels", [(40, 1), (200, 1), (512, 1), (40, 0), (200, 0), (512, 0), (576, 1)],
                         indirect=["attention_kernels"])
def test_fused_attention_training_passes(pg, S, attention_kernels):
    """The three kernels that work on the stored probabilities P, against fp64 on the same
    bf16-rounded operands (P as the forward kernel stored it), with the forward and the first-order backward
    in both their forms (K and V resident in LDS / streamed through the two-buffer ring; 576 keys: ring only):
      first-order backward   dS = P (dP - delta), dq = dS K / 8
      tangent pass           dctx = Pdot V + P dV, rho = rowsum(P dS)
      second-order scores    Sbar, Pdot of the softmax double backward (tests/kernel_specs.py)."""
    g = torch.Generator().manual_seed(90 + S)
    B, nh = 2, 2
    H, Z = nh * 64, B * nh
    bf = torch.bfloat16
    qkv, tqkv = rnd(B * S, 3 * H, dtype=bf, gen=g), rnd(B * S, 3 * H, dtype=bf, gen=g)
    gctx, ctxbar = rnd(B * S, H, dtype=bf, gen=g), rnd(B * S, H, dtype=bf, gen=g)
    mask = torch.ones(B, S)
    mask[0, S - 7:] = 0
    mask[1, 3] = 0
    d = lambda t: t.to(bf).to(DEV)  # noqa: E731
    ctx_d, P_d = pg.ops.attention_fwd_p(d(qkv), mask.to(DEV), B, S, nh)
    heads = lambda t, i=0: t[:, i * H:(i + 1) * H].double().view(B, S, nh, 64).transpose(1, 2)  # noqa: E731
    q, k, v = (heads(qkv, i) for i in range(3))
    qd, kd, vd = (heads(tqkv, i) for i in range(3))
    P = P_d[:, :, :S].double().cpu().view(B, nh, S, S)
    ctx = ctx_d.double().cpu()
    flat = lambda t: t.transpose(1, 2).reshape(B * S, H)  # noqa: E731
    # first-order backward
    dP = heads(gctx) @ v.transpose(2, 3)
    delta = (gctx.double() * ctx).view(B, S, nh, 64).sum(-1).transpose(1, 2).unsqueeze(-1)
    dS_ref = P * (dP - delta)
    dS, dq = pg.ops.attention_bwd_dq(d(qkv), P_d, ctx_d, d(gctx), B, S, nh)
    close(dS[:, :, :S], dS_ref.reshape(Z, S, S), bf)
    close(dq, flat(dS_ref @ k) * 0.125, bf)
    # the whole backward in two launches: dq as above, dk / dv by the key-stationary kernel (dS formed again from P)
    for keep in (True, False):
        dqkv, dS2 = pg.ops.attention_bwd(d(qkv), P_d, ctx_d, d(gctx), nh, B=B, S=S, keep_dS=keep)
        assert (dS2 is None) == (not keep)
        if keep:
            assert torch.equal(dS2, dS)
        assert torch.equal(dqkv[:, :H], dq)
        close(dqkv[:, H:2 * H], flat(dS_ref.transpose(2, 3) @ q) * 0.125, bf)
        close(dqkv[:, 2 * H:], flat(P.transpose(2, 3) @ heads(gctx)), bf)
    # tangent pass
    Sd = (qd @ k.transpose(2, 3) + q @ kd.transpose(2, 3)) * 0.125
    rho_ref = (P * Sd).sum(-1, keepdim=True)
    Pd_ref = P * (Sd - rho_ref)
    ctxdot, rho = pg.ops.attention_jvp(d(qkv), d(tqkv), P_d, ctx_d, B, S, nh)
    close(rho, rho_ref.reshape(Z, S), bf)
    close(ctxdot, flat(Pd_ref @ v + P @ vd), bf)
    # second-order scores (with the device's own dctx and rho, as the engine chains them)
    cd = ctxdot.double().cpu()
    gam = heads(gctx) @ v.transpose(2, 3)
    X = heads(ctxbar) @ v.transpose(2, 3) + heads(gctx) @ vd.transpose(2, 3)
    Sbar_ref, Pd2_ref = K.softmax_bwd2(P, Sd, gam, X)
    Sbar, Pd = pg.ops.attention_bwd2(d(qkv), d(tqkv), P_d, ctx_d, ctxdot, d(gctx), d(ctxbar), rho, B, S, nh)
    close(Pd[:, :, :S], Pd2_ref.reshape(Z, S, S), bf)
    close(Sbar[:, :, :S], Sbar_ref.reshape(Z, S, S), bf, mult=2)
    del cd


# ----------------------------------------------------------------------------- sampler
def test_mlm_decode_sample_commit(pg):
    g = torch.Generator().manual_seed(8)
    B, S, H, V = 3, 50, 128, 30
    t = torch.randn(B * S, H, generator=g)
    word = torch.randn(V, H, generator=g) * 0.3
    bias = torch.randn(V, generator=g) * 0.1
    for dtype in DTYPES:
        td = t.to(dtype)
        logits = pg.ops.mlm_decode(td.to(DEV), word.to(DEV), bias.to(DEV))
        close(logits, td.double() @ word.double().T + bias.double(), torch.float32, mult=4)
    logits = (t @ word.T + bias).view(B, S, V)
    noise = O.draw_noise(B * S, V, g)
    for temp in (1.0, 0.83):
        _, pred_ref, conf_ref = O.sample_from_logits(logits, temp, noise)
        pred, conf = pg.ops.sample_tokens(logits.view(-1, V).to(DEV), noise.to(DEV), temp)
        assert torch.equal(pred.cpu().view(B, S), pred_ref)
        np.testing.assert_allclose(conf.cpu().view(B, S).numpy(), conf_ref.numpy(), rtol=2e-5)
        # commit: every branch of models.py:35-63 (row without masks, fill-all, keep_percent, k clamps)
        ids = torch.full((B, S), O.PAD, dtype=torch.long)
        ids[:, 0] = O.CLS
        ids[0, 1:41] = O.MASK
        ids[0, 41] = O.SEP
        ids[1, 1:31] = torch.randint(5, 25, (30,), generator=g)
        ids[1, 31] = O.SEP  # no mask at all
        ids[2, 1:48] = torch.randint(5, 25, (47,), generator=g)
        ids[2, [5, 9, 33]] = O.MASK  # fewer masks than keep_percent * meaningful
        ids[2, 48] = O.SEP
        for rate in (1.0, 0.5, 0.10000000000000014, 0.1, 1.3877787807814457e-16, None):
            ref_ids = ids.clone()
            committed = O.commit_step(ref_ids, pred_ref, conf_ref, 0.1, rate)
            got = ids.clone().to(DEV)
            counts = torch.zeros(B, 2, dtype=torch.int32, device=DEV)
            fill_all = rate is not None and rate <= 0.1
            pg.ops.commit_topk(got, pred, conf_ref.reshape(-1).to(DEV), 0.1, fill_all, counts=counts)
            assert torch.equal(got.cpu(), ref_ids), f"rate {rate}"
            assert counts[:, 0].cpu().tolist() == committed
            assert counts[:, 1].cpu().tolist() == (ref_ids == O.MASK).sum(1).tolist()


def test_commit_ties_lowest_index(pg):
    ids = torch.tensor([[O.CLS] + [O.MASK] * 20 + [O.SEP]], device=DEV)
    pred = torch.full((1, 22), 7, dtype=torch.long, device=DEV)
    conf = torch.full((1, 22), 0.25, device=DEV)
    pg.ops.commit_topk(ids, pred.view(-1), conf.view(-1), 0.1, False)
    assert ids[0].tolist() == [O.CLS, 7, 7] + [O.MASK] * 18 + [O.SEP]  # int(0.1*20) = 2 lowest positions


# ----------------------------------------------------------------------------- critic head
@pytest.mark.parametrize("dtype", DTYPES)
def test_head_forward_backward_tangent(pg, dtype):
    from protgen_gan_amd._lib import lib
    g = torch.Generator().manual_seed(9)
    B, S, H = 3, 5, 128
    Cw = O.init_classifier(H, seed=11)
    hid = rnd(B, S, H, dtype=dtype, gen=g)
    flat = torch.cat([Cw[f"{k}.{n}"].reshape(-1) for k in (0, 2, 4, 6) for n in ("weight", "bias")]).to(DEV)
    assert flat.numel() == lib.pg_head_total(H)
    hd = hid.to(dtype).to(DEV)
    scores = torch.empty(B, 1, device=DEV)
    stash = torch.empty(B, lib.pg_head_stash_floats(H), device=DEV)
    st = pg.ops.stream()
    dt = pg.ops.dt_code(dtype)
    assert lib.pg_head_forward(dt, hd.data_ptr(), S * H, flat.data_ptr(), B, H, scores.data_ptr(), stash.data_ptr(), st) == 0
    # CPU reference with autograd (fp64)
    Cd = {k: v.double().requires_grad_(True) for k, v in Cw.items()}
    x = hid[:, 0, :].double().requires_grad_(True)
    s_ref = O.classifier(Cd, x)
    close(scores, s_ref, torch.float32, mult=4)
    w = torch.tensor([0.7, -1.3, 0.4])
    (s_ref.squeeze(1) * w.double()).sum().backward()
    ghead = torch.zeros_like(flat)
    dh = torch.zeros(B, S, H, dtype=dtype, device=DEV)
    assert lib.pg_head_backward(dt, flat.data_ptr(), stash.data_ptr(), w.to(DEV).data_ptr(), B, H, ghead.data_ptr(),
                                dh.data_ptr(), S * H, st) == 0
    close(dh[:, 0, :], x.grad, dtype)
    assert float(dh[:, 1:, :].float().abs().max()) == 0.0
    gref = torch.cat([Cd[f"{k}.{n}"].grad.reshape(-1) for k in (0, 2, 4, 6) for n in ("weight", "bias")])
    close(ghead, gref, torch.float32, mult=8)
    # tangent + gradient of sum(sdot) w.r.t. the head parameters
    xdot = rnd(B, S, H, dtype=dtype, gen=g)
    Cd2 = {k: v.double().requires_grad_(True) for k, v in Cw.items()}
    x2 = hid[:, 0, :].double().requires_grad_(True)
    s2 = O.classifier(Cd2, x2).sum()
    gx, = torch.autograd.grad(s2, x2, create_graph=True)
    sur = (gx * xdot[:, 0, :].double()).sum()
    sur.backward()
    ghead.zero_()
    tst = torch.empty_like(stash)
    sdot = torch.empty(B, device=DEV)
    assert lib.pg_head_jvp_grad(dt, flat.data_ptr(), stash.data_ptr(), xdot.to(dtype).to(DEV).data_ptr(), S * H, B, H, 2.0,
                                ghead.data_ptr(), tst.data_ptr(), sdot.data_ptr(), st) == 0
    gref2 = torch.cat([(Cd2[f"{k}.{n}"].grad if Cd2[f"{k}.{n}"].grad is not None else torch.zeros_like(Cd2[f"{k}.{n}"])).reshape(-1)
                       for k in (0, 2, 4, 6) for n in ("weight", "bias")])
    close(ghead, 2.0 * gref2, torch.float32, mult=8)
    close(sdot.sum(), sur.detach(), torch.float32, mult=8)


# ----------------------------------------------------------------------------- GP reductions, interpolation, scatter, AdamW
@pytest.mark.parametrize("dtype", DTYPES)
def test_gp_reductions_and_interpolation(pg, dtype):
    from protgen_gan_amd._lib import lib
    g = torch.Generator().manual_seed(10)
    B, S, H = 3, 24, 64
    gr = rnd(B, S, H, dtype=dtype, scale=0.05, gen=g)
    mask = (torch.rand(B, S, generator=g) > 0.3).float()
    d = lambda t: t.to(dtype).to(DEV)  # noqa: E731
    st, dt = pg.ops.stream(), pg.ops.dt_code(dtype)
    norms = torch.empty(B, device=DEV)
    gp = torch.empty((), device=DEV)
    assert lib.pg_gp_norms(dt, d(gr).data_ptr(), mask.to(DEV).data_ptr(), B, S, H, norms.data_ptr(), gp.data_ptr(), st) == 0
    n_ref = (gr.double() * mask[..., None].double()).flatten(1).norm(dim=1)
    close(norms, n_ref, torch.float32, mult=4)
    close(gp, ((n_ref - 1) ** 2).mean(), torch.float32, mult=4)
    v = torch.empty(B, S, H, dtype=dtype, device=DEV)
    scale = torch.tensor(5.0, device=DEV)
    assert lib.pg_gp_direction(dt, d(gr).data_ptr(), mask.to(DEV).data_ptr(), norms.data_ptr(), B, S, H, scale.data_ptr(),
                               v.data_ptr(), None, st) == 0
    v_ref = 5.0 * (2.0 / B) * ((n_ref - 1) / n_ref)[:, None, None] * gr.double() * mask[..., None].double()
    close(v, v_ref, dtype)
    er, ef = rnd(B, S, H, dtype=dtype, gen=g), rnd(B, S, H, dtype=dtype, gen=g)
    alpha = torch.rand(B, generator=g)
    x = torch.empty(B, S, H, dtype=dtype, device=DEV)
    assert lib.pg_interpolate(dt, d(er).data_ptr(), d(ef).data_ptr(), alpha.to(DEV).data_ptr(), x.data_ptr(), B, S * H, st) == 0
    a = alpha.double()[:, None, None]
    close(x, a * er.double() + (1 - a) * ef.double(), dtype)
    o1, o2 = torch.empty_like(x), torch.empty_like(x)
    assert lib.pg_interpolate_bwd(dt, d(er).data_ptr(), alpha.to(DEV).data_ptr(), o1.data_ptr(), o2.data_ptr(), B, S * H, st) == 0
    close(o1, a * er.double(), dtype)
    close(o2, (1 - a) * er.double(), dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_embedding_forward_and_scatter(pg, dtype):
    from protgen_gan_amd._lib import lib
    g = torch.Generator().manual_seed(11)
    B, S, H, V = 2, 19, 128, 30
    word, pos, typ = torch.randn(V, H, generator=g), torch.randn(40, H, generator=g), torch.randn(2, H, generator=g)
    gam, bet = 1 + 0.1 * torch.randn(H, generator=g), 0.1 * torch.randn(H, generator=g)
    ids = torch.randint(0, V, (B, S), generator=g)
    st, dt = pg.ops.stream(), pg.ops.dt_code(dtype)
    out = torch.empty(B, S, H, dtype=dtype, device=DEV)
    esum = torch.empty_like(out)
    dw, dp, dty, dg, db, di = (t.to(DEV) for t in (word, pos, typ, gam, bet, ids))
    assert lib.pg_embed_ln2(dt, di.data_ptr(), dw.data_ptr(), dp.data_ptr(), dty.data_ptr(), dg.data_ptr(), db.data_ptr(),
                            1e-12, out.data_ptr(), esum.data_ptr(), B, S, H, st) == 0
    sref = (word[ids] + typ[0]) + pos[:S][None]
    close(esum, sref, dtype)
    close(out, K.ln_fwd(esum.float().cpu().double(), gam.double(), bet.double(), 1e-12), dtype)
    dsum = rnd(B, S, H, dtype=dtype, gen=g)
    gw, gpz, gt = torch.zeros(V, H, device=DEV), torch.zeros(40, H, device=DEV), torch.zeros(H, device=DEV)
    assert lib.pg_embed_scatter(dt, dsum.to(dtype).to(DEV).data_ptr(), di.data_ptr(), gw.data_ptr(), gpz.data_ptr(),
                                gt.data_ptr(), B, S, H, st) == 0
    rw = torch.zeros(V, H, dtype=torch.float64).index_add_(0, ids.reshape(-1), dsum.double().reshape(-1, H))
    close(gw, rw, torch.float32, mult=8)
    rp = torch.zeros(40, H, dtype=torch.float64)
    rp[:S] = dsum.double().sum(0)
    close(gpz, rp, torch.float32, mult=8)
    close(gt, dsum.double().sum((0, 1)), torch.float32, mult=8)


def test_adamw_and_casts(pg):
    g = torch.Generator().manual_seed(12)
    n = 100_003
    p0 = torch.randn(n, generator=g)
    p, m, v = p0.clone(), torch.zeros(n), torch.zeros(n)
    pd, md, vd = p0.clone().to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    sh = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    for step in (1, 2, 3):
        gr = torch.randn(n, generator=g)
        O.adamw_step(p, gr, m, v, step, 3e-3, wd=0.01)
        pg.ops.adamw(pd, gr.to(DEV), md, vd, sh, 3e-3, 0.9, 0.999, 1e-8, 0.01, step)
    np.testing.assert_allclose(pd.cpu().numpy(), p.numpy(), rtol=2e-6, atol=2e-7)
    assert torch.equal(sh.cpu(), pd.cpu().to(torch.bfloat16))
