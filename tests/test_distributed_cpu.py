"""The N > 1 paths on CPU with the gloo backend, world_size 2: gradient averaging over the flat
buffers reproduces the global-batch gradient (the oracle provides the per-shard gradients),
and sharded generation covers every sequence index exactly once."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import pgload
from oracle import protbert_cpu as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeStore:
    """Flat store stand-in on CPU tensors (the real FlatStore lives on the GPU)."""

    def __init__(self, params):
        self.views, off = [], 0
        total = sum(p.numel() for p in params) + 8 * len(params)
        self.grad = torch.zeros(total)
        for p in params:
            n = p.numel()
            p.grad = self.grad[off:off + n].view(p.shape)
            self.views.append((p, off, n))
            off += (n + 7) // 8 * 8


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    pgload.load()
    from protgen_gan_amd import distributed as D
    r, _, w = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    cfg = O.BertCfg(hidden_size=64, num_hidden_layers=1, num_attention_heads=1, intermediate_size=128,
                    max_position_embeddings=32)
    P = O.init_params(cfg, seed=5, bias_std=0.02)
    C = O.init_classifier(64, seed=6)
    g = torch.Generator().manual_seed(9)
    B, S = 4, 16
    real = torch.randint(5, 25, (B, S), generator=g)
    fake = torch.randint(5, 25, (B, S), generator=g)
    ones = torch.ones(B, S, dtype=torch.long)
    alpha = torch.rand(B, 1, 1, generator=g)
    lo, hi = D.shard_range(B, rank, world)
    _, _, _, _, grads = O.critic_step_grads(P, C, cfg, real[lo:hi], fake[lo:hi], ones[lo:hi], ones[lo:hi], alpha[lo:hi], 5.0)
    names = sorted(grads)
    params = [torch.nn.Parameter(torch.zeros_like(grads[n])) for n in names]
    store = _FakeStore(params)
    for p, n in zip(params, names):
        p.grad.copy_(grads[n])
    D.GradSync([store], world, bucket_bytes=4096)()
    if rank == 0:
        _, _, _, _, full = O.critic_step_grads(P, C, cfg, real, fake, ones, ones, alpha, 5.0)
        err = max(float((p.grad - full[n]).abs().max() / (full[n].abs().max() + 1e-12)) for p, n in zip(params, names))
        q.put(("grad_err", err))
    # sharded generation bookkeeping: every index exactly once, per-rank seeds differ
    got = D.generate_sharded(lambda num_sequences, rng_seed: [f"{rng_seed}:{i}" for i in range(num_sequences)], 7, rank,
                             world, rng_seed=100)
    allseq = D.gather_strings(got, rank, world)
    if rank == 0:
        q.put(("gathered", allseq))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_sync_and_sharding_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert out["grad_err"] < 1e-5  # mean of per-shard gradients == global-batch gradient
    assert out["gathered"] == ["100:0", "100:1", "100:2", "100:3", "101:0", "101:1", "101:2"]


def test_shard_and_bucket_helpers():
    pgload.load()
    from protgen_gan_amd import distributed as D
    for n, w in ((120000, 8), (7, 3), (2, 4)):
        spans = [D.shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
    assert D.bucket_ranges([(0, 10), (20, 45)], 10) == [(0, 10), (20, 30), (30, 40), (40, 45)]


def test_layerwise_plan_order_and_coverage():
    """GradSync's issue order with an encoder to overlap with: the buckets of the last layer first,
    layer 0 last among the layers, then what lies outside the layers (embedding tables, head); every
    live element exactly once; the position table cut at the promised width."""
    pgload.load()
    from protgen_gan_amd import distributed as D
    assert D.clip_ranges([(0, 10), (20, 45)], 5, 30) == [(5, 10), (20, 30)]
    assert D.cut_ranges([(0, 100)], [(10, 20), (40, 50)]) == [(0, 10), (20, 40), (50, 100)]
    assert D.cut_ranges([(0, 10), (30, 60)], [(5, 35)]) == [(0, 5), (35, 60)]

    L, per_layer, pos_rows, H = 3, 96, 16, 8
    pos = torch.nn.Parameter(torch.zeros(pos_rows, H))
    head = torch.nn.Parameter(torch.zeros(24))
    layers = [torch.nn.Parameter(torch.zeros(per_layer)) for _ in range(L)]
    store = _FakeStore([pos] + layers + [head])
    store.live_elems = {}

    class Model:
        config = type("Cfg", (), {"num_hidden_layers": L})()

        def __init__(self):
            self.store = store

        def enable_layer_events(self):
            raise AssertionError("events are for more than one rank")

        def layer_param_range(self, l):
            off = store.views[1 + l][1]
            return off, off + per_layer

        def limit_position_rows(self, rows):
            store.live_elems[store.views[0][1]] = rows * H

    sync = D.GradSync([store], world=1, bucket_bytes=4 * 40, overlap=Model(), position_rows=4)
    plan = sync._plan()
    order = [l for l, *_ in plan]
    assert order == [2, 2, 2, 1, 1, 1, 0, 0, 0, None, None]
    assert all(b - a <= 40 for _, _, a, b in plan)
    seen = torch.zeros(store.grad.numel(), dtype=torch.int32)
    for _, _, a, b in plan:
        seen[a:b] += 1
    want = torch.zeros_like(seen)
    want[:4 * H] = 1  # the position rows inside the bound
    for p, off, n in store.views[1:]:
        want[off:off + n] = 1
    assert torch.equal(seen, want)
