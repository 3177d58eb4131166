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
    rs = sorted((off, off + n) for p, off, n in store.views if p.requires_grad and p.grad is not None)
    merged: List[List[int]] = []
    for a, b in rs:
        if merged and a <= merged[-1][1] + 8:
            merged[-1][1] = max(merged[-1][1], b)
        else:
            merged.append([a, b])
    return [(a, b) for a, b in merged]


class GradSync:
    """Averages the critic's (or any model's) flat gradient buffers across ranks.

    Bucket size: xGMI is point-to-point (7 links per GPU), so a ring all-reduce is bound by one
    link; 64 MiB fp32 buckets keep each collective bandwidth-bound (not latency-bound) while
    letting RCCL pipeline consecutive buckets."""

    def __init__(self, stores, world: int, bucket_bytes: int = 64 << 20):
        self.stores = list(stores)
        self.world = world
        self.bucket_elems = max(1, bucket_bytes // 4)

    def __call__(self):
        if self.world <= 1:
            return
        works = []
        for store in self.stores:
            if store.grad is None:
                continue
            for a, b in bucket_ranges(live_ranges(store), self.bucket_elems):
                chunk = store.grad[a:b]
                if chunk.is_cuda and dist.get_backend() == "nccl":
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
