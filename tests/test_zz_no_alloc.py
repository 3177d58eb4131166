"""pg_layernorm_bwd_static (the entry point that allocates nothing: its partial sums live in a device
array inside the library) against pg_layernorm_bwd, through the C ABI, two calls in a row with the second
on a smaller input.  dx is the same kernel on the same rows: identical bits.  The parameter gradients end
in float atomics over row chunks, whose order is not fixed: 1e-5 of the tensor scale in fp32, as stated."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.fixture(scope="module")
def pg():
    import pgload
    return pgload.load()


def _both(pg, dtype, rows, H, seed):
    lib, ops = pg._lib.lib, pg.ops
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(rows, H, generator=g).to(dtype).to(DEV)
    dy = torch.randn(rows, H, generator=g).to(dtype).to(DEV)
    gamma = (1.0 + 0.1 * torch.randn(H, generator=g)).to(DEV)
    out = []
    for static in (False, True):
        dx = torch.empty_like(x)
        acc = [torch.zeros(H, dtype=torch.float32, device=DEV) for _ in range(3)]
        if static:
            rc = lib.pg_layernorm_bwd_static(ops.dt_code(dtype), x.data_ptr(), dy.data_ptr(), gamma.data_ptr(), 1e-12,
                                             dx.data_ptr(), acc[0].data_ptr(), acc[1].data_ptr(), acc[2].data_ptr(), rows, H,
                                             ops.stream())
        else:
            part = torch.empty(3 * H * lib.pg_ln_partial_rows(rows), dtype=torch.float32, device=DEV)
            rc = lib.pg_layernorm_bwd(ops.dt_code(dtype), x.data_ptr(), dy.data_ptr(), gamma.data_ptr(), 1e-12, dx.data_ptr(),
                                      acc[0].data_ptr(), acc[1].data_ptr(), acc[2].data_ptr(), part.data_ptr(), rows, H,
                                      ops.stream())
        assert rc == 0, lib.pg_strerror(rc)
        torch.cuda.synchronize()
        out.append((dx.cpu(), [a.cpu() for a in acc]))
    return out


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_static_scratch_matches_workspace_form(pg, dtype):
    # first call: more rows than workgroups (every partial row of the scratch is used); second: fewer rows,
    # so stale partial rows of the first call sit behind the ones this call writes
    for rows, H, seed in ((5000, 1024, 1), (300, 1024, 2), (77, 256, 3)):
        (dx0, acc0), (dx1, acc1) = _both(pg, dtype, rows, H, seed)
        assert torch.equal(dx0, dx1)
        for a, b in zip(acc0, acc1):
            scale = float(a.abs().max())
            assert float((a - b).abs().max()) <= 1e-5 * max(scale, 1e-6)


def test_static_scratch_rejects_oversize(pg):
    lib = pg._lib.lib
    x = torch.zeros(4, 2048, device=DEV)
    g = torch.ones(2048, device=DEV)
    rc = lib.pg_layernorm_bwd_static(0, x.data_ptr(), x.data_ptr(), g.data_ptr(), 1e-12, x.data_ptr(), g.data_ptr(), None,
                                     None, 4, 2048, pg.ops.stream())
    assert rc == -6 and b"static" in lib.pg_strerror(rc)
