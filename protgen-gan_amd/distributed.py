"""Multi-GPU: one process per GPU, `torch.distributed` over RCCL/xGMI (backend "nccl"), or
gloo on CPU tensors for tests.

* Training is data-parallel (absent from the reference, which is single-GPU): every rank
  holds full replicas, draws its own real batch and generates its own fakes; `critic_loss`
  and the gradient penalty are batch means (loss.py:4,95), so averaging the per-rank
  gradients gives the gradient of the global-batch loss.  `GradSync` all-reduces the flat
  gradient buffers in contiguous buckets (the buffers are already flat, so no packing).
* Mass generation shards sequence indices over ranks with no collective at all
  (the reference's own pattern: independent SLURM array tasks, slurms/mass_eval.sh:34-37)."""
from __future__ import annotations

import os
from typing import List, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None) -> Tuple[int, int, int]:
    """(rank, local_rank, world) from torchrun's environment; initialises the process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # PG_REHEARSE_ONE_GPU=1: every rank on device 0 over gloo, to exercise the multi-rank code path
        # on a one-GPU box (RCCL refuses two ranks on one device); never a measurement.
        rehearse = os.environ.get("PG_REHEARSE_ONE_GPU") == "1"
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() and not rehearse else "gloo"
        if rehearse:
            local = 0
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif os.environ.get("PG_REHEARSE_ONE_GPU") == "1":
        local = 0
    return rank, local, world


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of `n_items` for `rank`; sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def bucket_ranges(ranges: List[Tuple[int, int]], bucket_elems: int) -> List[Tuple[int, int]]:
    """Split contiguous element ranges into pieces of at most `bucket_elems`."""
    out = []
    for a, b in ranges:
        while a < b:
            e = min(b, a + bucket_elems)
            out.append((a, e))
            a = e
    return out


def live_ranges(store) -> List[Tuple[int, int]]:
    """Contiguous ranges of a flat store covered by parameters that currently hold a gradient."""
    bound = getattr(store, "live_elems", {})
    rs = sorted((off, off + min(n, bound.get(off, n))) for p, off, n in store.views
                if p.requires_grad and p.grad is not None)
    merged: List[List[int]] = []
    for a, b in rs:
        if merged and a <= merged[-1][1] + 8:
            merged[-1][1] = max(merged[-1][1], b)
        else:
            merged.append([a, b])
    return [(a, b) for a, b in merged]


def clip_ranges(ranges: List[Tuple[int, int]], lo: int, hi: int) -> List[Tuple[int, int]]:
    """The parts of `ranges` inside [lo, hi)."""
    return [(max(a, lo), min(b, hi)) for a, b in ranges if max(a, lo) < min(b, hi)]


def cut_ranges(ranges: List[Tuple[int, int]], holes: List[Tuple[int, int]]) -> List[Tuple[int, int]]:
    """`ranges` minus the (disjoint) `holes`."""
    out = list(ranges)
    for lo, hi in holes:
        nxt = []
        for a, b in out:
            if a < min(b, lo):
                nxt.append((a, min(b, lo)))
            if max(a, hi) < b:
                nxt.append((max(a, hi), b))
        out = nxt
    return sorted(out)


class GradSync:
    """Averages the critic's (or any model's) flat gradient buffers across ranks.

    Bucket size: xGMI is point-to-point (7 links per GPU), so a ring all-reduce is bound by one
    link; 64 MiB fp32 buckets keep each collective bandwidth-bound (not latency-bound) while
    letting RCCL pipeline consecutive buckets.

    `overlap` (an encoder model whose store is among `stores`; RCCL only): the backward passes finish
    the layers from the last to the first and record one event per layer; each layer's bucket is
    queued behind its event on a side stream, so it travels while the rest of the pass still computes,
    and only the embedding tables and the head wait for the end of the backward.  The host has
    issued the whole backward when this object is called; the device is still early in it.

    `position_rows`: bound on the positions any rank ever feeds (the padded width); the rest of the
    position table holds zero gradients everywhere and is left out of the reduction."""

    def __init__(self, stores, world: int, bucket_bytes: int = 64 << 20, overlap=None, position_rows=None):
        self.stores = list(stores)
        self.world = world
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.overlap = overlap
        self._side = None
        self._tail = None
        self._plan_cache = None
        self._layer_spans = None
        if overlap is not None:
            if not any(st is overlap.store for st in self.stores):
                raise ValueError("overlap model's store is not among the stores to reduce")
            if world > 1:
                overlap.enable_layer_events()
        if position_rows is not None:
            for m in ([overlap] if overlap is not None else []):
                m.limit_position_rows(position_rows)
            if overlap is None:
                raise ValueError("position_rows needs the model (overlap=...) whose position table it bounds")

    def _plan(self):
        """[(layer or None, store, lo, hi)] in issue order: layers last to first, then everything else."""
        lives = [live_ranges(store) if store.grad is not None else None for store in self.stores]
        key = tuple(tuple(x) if x is not None else None for x in lives)
        if self._plan_cache is not None and self._plan_cache[0] == key:
            return self._plan_cache[1]
        plan = []
        rest = []
        for store, live in zip(self.stores, lives):
            if live is None:
                continue
            if self.overlap is not None and store is self.overlap.store:
                if self._layer_spans is None:
                    L = self.overlap.config.num_hidden_layers
                    self._layer_spans = [self.overlap.layer_param_range(l) for l in range(L)]
                spans = []
                for l in reversed(range(len(self._layer_spans))):
                    lo, hi = self._layer_spans[l]
                    spans.append((lo, hi))
                    for a, b in bucket_ranges(clip_ranges(live, lo, hi), self.bucket_elems):
                        plan.append((l, store, a, b))
                live = cut_ranges(live, sorted(spans))
            for a, b in bucket_ranges(live, self.bucket_elems):
                rest.append((None, store, a, b))
        self._plan_cache = (key, plan + rest)
        return plan + rest

    def __call__(self):
        if self.world <= 1:
            return
        plan = self._plan()
        rccl = dist.get_backend() == "nccl"
        if rccl and self.overlap is not None and any(l is not None for l, *_ in plan):
            return self._overlapped(plan)
        works = []
        for _, store, a, b in plan:
            chunk = store.grad[a:b]
            if chunk.is_cuda and rccl:
                works.append(dist.all_reduce(chunk, op=dist.ReduceOp.AVG, async_op=True))
            elif chunk.is_cuda:  # gloo (rehearsal only): stage through the host, it has no AVG either
                host = chunk.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM)
                chunk.copy_(host.div_(self.world))
            else:
                dist.all_reduce(chunk, op=dist.ReduceOp.SUM)
                chunk.div_(self.world)
        for w in works:
            w.wait()

    def _overlapped(self, plan):
        dev = self.overlap.device
        main = torch.cuda.current_stream(dev)
        if self._side is None:
            self._side = torch.cuda.Stream(dev)
            self._tail = torch.cuda.Event()
        events = self.overlap.enable_layer_events()
        self._tail.record(main)  # everything the backward has issued
        works = []
        with torch.cuda.stream(self._side):
            waited = None
            for l, store, a, b in plan:
                if l != waited:
                    self._side.wait_event(events[l] if l is not None else self._tail)
                    waited = l
                # the collective is ordered behind the side stream, i.e. behind that layer's event only
                works.append(dist.all_reduce(store.grad[a:b], op=dist.ReduceOp.AVG, async_op=True))
        for w in works:
            w.wait()  # the optimizer step on the main stream comes after every bucket


def generate_sharded(generate_fn, num_sequences: int, rank: int, world: int, rng_seed: int, **kw) -> List[str]:
    """Run `generate_fn(num_sequences=<this rank's share>, rng_seed=rng_seed + rank, **kw)`."""
    lo, hi = shard_range(num_sequences, rank, world)
    if hi <= lo:
        return []
    return generate_fn(num_sequences=hi - lo, rng_seed=rng_seed + rank, **kw)


def gather_strings(local: List[str], rank: int, world: int) -> List[str]:
    """Rank-ordered concatenation on rank 0 (host side, after generation has finished)."""
    if world <= 1:
        return local
    out = [None] * world
    dist.all_gather_object(out, local)
    return [s for part in out for s in part] if rank == 0 else []
