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
    # the same identity when the rows of the global batch are dealt by length instead of cut in order: lengths 16, 8,
    # 12, 4 -> rank 0 holds rows {0, 3}, rank 1 rows {1, 2}; each row keeps its own alpha
    from protgen_gan_amd import data as DD
    lens = [16, 8, 12, 4]
    rm = torch.zeros(B, S, dtype=torch.long)
    real2, fake2 = real.clone(), fake.clone()
    for b, n in enumerate(lens):
        rm[b, :n] = 1
        real2[b, n:] = O.PAD
        fake2[b, n:] = O.PAD
    mine = DD.balanced_deal(lens, world)[rank]
    tok = sum(lens[i] for i in mine)
    idx = torch.tensor(mine)
    _, _, _, _, grads = O.critic_step_grads(P, C, cfg, real2[idx], fake2[idx], rm[idx], rm[idx], alpha[idx], 5.0)
    for p, n in zip(params, names):
        p.grad.copy_(grads[n])
    D.GradSync([store], world, bucket_bytes=4096)()
    toks = [None] * world
    dist.all_gather_object(toks, (sorted(mine), tok))
    if rank == 0:
        _, _, _, _, full = O.critic_step_grads(P, C, cfg, real2, fake2, rm, rm, alpha, 5.0)
        err = max(float((p.grad - full[n]).abs().max() / (full[n].abs().max() + 1e-12)) for p, n in zip(params, names))
        q.put(("dealt_grad_err", err))
        q.put(("dealt", toks))
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
    out = dict(q.get(timeout=240) for _ in range(4))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert out["grad_err"] < 1e-5  # mean of per-shard gradients == global-batch gradient
    assert out["dealt_grad_err"] < 1e-5  # ... whichever rank holds which row
    assert out["dealt"] == [([0, 3], 20), ([1, 2], 20)]
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


def test_balanced_deal_token_counts():
    """data.balanced_deal on global batches of 8 rows per rank with the benchmark's lengths (uniform 200..500 + CLS/SEP):
    every rank the same number of rows, every row exactly once, per-rank token counts within 2 % of each other
    (dealing rows r, r + world, ... leaves them up to 50 % apart), the same deal on every rank."""
    import random
    pgload.load()
    from protgen_gan_amd import data as DD
    from protgen_gan_amd import train as T
    for world in (2, 4, 8):
        worst = naive = 0.0
        for seed in range(200):
            rng = random.Random(seed)
            L = [rng.randint(200, 500) + 2 for _ in range(8 * world)]
            deal = DD.balanced_deal(L, world)
            assert deal == DD.balanced_deal(list(L), world)
            assert sorted(i for d in deal for i in d) == list(range(8 * world)) and all(len(d) == 8 for d in deal)
            sums = [sum(L[i] for i in d) for d in deal]
            worst = max(worst, (max(sums) - min(sums)) / (sum(sums) / world))
            strided = [sum(L[r::world]) for r in range(world)]
            naive = max(naive, (max(strided) - min(strided)) / (sum(strided) / world))
        assert worst < 0.02 < naive, (world, worst, naive)
    with pytest.raises(ValueError):
        DD.balanced_deal([5, 6, 7], 2)
    # the trainers' draws: every rank seeds the length stream alike and keeps its share of the global draw
    pool = list(range(200, 501))
    shares = [T.sample_fake_lengths(pool, 8, random.Random(3), r, 4) for r in range(4)]
    glob = [T.sample_sequence_length(pool, rng=rng) + 2 for rng in [random.Random(3)] for _ in range(32)]
    assert sorted(v for s in shares for v in s) == sorted(glob)
    sums = [sum(s) for s in shares]
    assert (max(sums) - min(sums)) / (sum(sums) / 4) < 0.02
    assert T.sample_fake_lengths(pool, 8, random.Random(3)) == glob[:8]  # one rank: the plain draw
    g = torch.Generator().manual_seed(0)
    parts = [T.synthetic_real_batch(8, pool, 512, "cpu", random.Random(5), g, r, 2) for r in range(2)]
    tot = [sum(p["lengths"]) for p in parts]
    assert abs(tot[0] - tot[1]) / (sum(tot) / 2) < 0.02
    for p in parts:
        assert p["attention_mask"].sum(dim=1).tolist() == [float(v) for v in p["lengths"]]


def test_batch_loader_deals_global_batches():
    """BatchLoader at world 2: both ranks walk the same permutation, never share a row, cover every global batch, and
    (balance on) hold token counts within 2 % of each other per batch."""
    pgload.load()
    from protgen_gan_amd import data as DD
    import random
    rng = random.Random(0)
    N, S = 64, 512
    lengths = [rng.randint(200, 500) + 2 for _ in range(N)]
    ids = torch.zeros(N, S, dtype=torch.long)
    ids[:, 0] = torch.arange(N)  # row tag
    mask = torch.zeros(N, S)
    for i, n in enumerate(lengths):
        mask[i, :n] = 1
    data = DD.TokenizedSet(ids, mask, lengths)
    for balance in (True, False):
        loaders = [DD.BatchLoader(data, 8, "cpu", generator=torch.Generator().manual_seed(4), rank=r, world=2, balance=balance)
                   for r in range(2)]
        assert len(loaders[0]) == 4
        seen = []
        for b0, b1 in zip(*loaders):
            r0, r1 = b0["input_ids"][:, 0].tolist(), b1["input_ids"][:, 0].tolist()
            assert len(r0) == len(r1) == 8 and not set(r0) & set(r1)
            seen += r0 + r1
            if balance:
                t0, t1 = sum(b0["lengths"]), sum(b1["lengths"])
                assert abs(t0 - t1) / ((t0 + t1) / 2) < 0.02
        assert sorted(seen) == list(range(N))


def test_layerwise_plan_on_the_full_protbert_layout():
    """GradSync's plan on ProtBERT's real parameter layout (30 layers x 1024, offsets from the library), positions
    bounded at 512: one 50.4 MB bucket per layer, layers 29 -> 0, then the embedding tables and the head; the bytes on
    the wire are the critic's reachable parameters minus the position rows no batch touches."""
    import ctypes as C
    pg = pgload.load()
    from protgen_gan_amd import _lib, bert
    from protgen_gan_amd import distributed as D
    cfg = pg.ProtBertConfig()
    ccfg = cfg.to_c()
    lib = _lib.lib
    H, I, L = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers

    class Store:
        def __init__(self, total):
            self.grad = torch.empty(total, device="meta")
            self.views, self.live_elems = [], {}

        def add(self, off, shape):
            p = torch.nn.Parameter(torch.empty(shape, device="meta"))
            p.grad = torch.empty(shape, device="meta")
            self.views.append((p, off, p.numel()))

    store = Store(lib.pg_param_total(C.byref(ccfg)))
    for name, which in bert._GLOBAL_NAMES:
        if not name.startswith("cls."):  # the MLM head is not on the critic's path: no gradient, not reduced
            store.add(lib.pg_param_offset_global(C.byref(ccfg), which), bert._shape_of(cfg, name))
    spans = []
    for l in range(L):
        offs = []
        for suffix, which in bert._LAYER_NAMES:
            off = lib.pg_param_offset_layer(C.byref(ccfg), l, which)
            shape = bert._shape_of(cfg, f"bert.encoder.layer.{l}.{suffix}")
            store.add(off, shape)
            offs.append((off, off + int(torch.Size(shape).numel())))
        spans.append((min(a for a, _ in offs), max(b for _, b in offs)))
    head = Store(lib.pg_head_total(H))
    head.add(0, (lib.pg_head_total(H),))

    class Model:
        config = cfg

        def __init__(self):
            self.store = store

        def enable_layer_events(self):
            return []

        def layer_param_range(self, l):
            return spans[l]

        def limit_position_rows(self, rows):
            store.live_elems[lib.pg_param_offset_global(C.byref(ccfg), _lib.P_POS_EMB)] = rows * H

    sync = D.GradSync([store, head], world=8, overlap=Model(), position_rows=512)
    plan = sync._plan()
    per_layer = 4 * H * H + 2 * H * I + 9 * H + I
    assert per_layer == 12596224
    layer_part = [(l, b - a) for l, st, a, b in plan if l is not None]
    assert layer_part == [(l, per_layer) for l in reversed(range(L))]  # one bucket per layer (64 MiB holds 50.4 MB)
    rest = [(st, a, b) for l, st, a, b in plan if l is None]
    assert plan[:L] == [(l, st, a, b) for l, st, a, b in plan if l is not None]  # the layers come first
    tables = sum(b - a for st, a, b in rest if st is store)
    assert tables == cfg.vocab_size * H + 512 * H + cfg.type_vocab_size * H + 2 * H
    assert sum(b - a for st, a, b in rest if st is head) == 672641
    wire = 4 * (L * per_layer + tables + 672641)
    reachable = 4 * (L * per_layer + cfg.vocab_size * H + cfg.max_position_embeddings * H + cfg.type_vocab_size * H + 2 * H
                     + 672641)
    assert reachable - wire == 4 * (cfg.max_position_embeddings - 512) * H  # 161.7 MB of zeros stay home
    assert abs(wire / 1e9 - 1.5168) < 1e-3, wire
