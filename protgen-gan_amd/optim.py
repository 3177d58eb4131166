"""Fused multi-tensor AdamW over the flat parameter stores (torch.optim.AdamW semantics as
used at fully_masked_train.py:74-75,326): decoupled weight decay, bias correction with a step
count per parameter (a parameter that starts receiving gradients later starts at step 1, as in
torch), parameters whose gradient is None are skipped entirely (no decay), `param_groups[i]["lr"]`
may be edited between steps (the trainers' warm-up does).  One kernel launch per contiguous run of
parameters that share a group and a step count updates fp32 master weights, both moments and the
bf16 shadow.

The moments and step counts live on the flat stores (FlatStore.adam_m / adam_v / adam_steps), so one
optimizer per store: two AdamW instances over the same parameters would share them.  state_dict() /
load_state_dict() carry them (the reference saves no optimizer state, CLAUDE.md:339-352; resuming a run
needs it)."""
from __future__ import annotations

import torch

from . import ops


def _merge_ranges(ranges):
    ranges = sorted(ranges)
    out = []
    for a, b in ranges:
        if out and a <= out[-1][1] + 8:  # tensors are padded to 8-element boundaries
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_scale=1.0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.grad_scale = grad_scale
        self._stores = None

    def _index(self):
        """Map every parameter to (store, offset, numel) by locating its storage in a flat store."""
        from .bert import FlatStore  # noqa: F401
        stores = {}
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                st = getattr(p, "_pg_store", None)
                if st is None:
                    raise TypeError("protgen AdamW only drives parameters of ProtBertForMaskedLM / CriticHead")
                store, off, n = st
                stores.setdefault(id(store), (store, []))[1].append((gi, p, off, n))
        self._stores = list(stores.values())

    @torch.no_grad()
    def step(self, closure=None):
        if self._stores is None:
            self._index()
        for store, entries in self._stores:
            if store.grad is None:
                continue
            if store.adam_m is None:
                store.adam_m = torch.zeros_like(store.master)
                store.adam_v = torch.zeros_like(store.master)
            per_group = {}
            for gi, p, off, n in entries:
                if p.grad is None or not p.requires_grad:
                    continue
                step = store.adam_steps.get(off, 0) + 1
                store.adam_steps[off] = step
                per_group.setdefault((gi, step), []).append((off, off + n))
            if not per_group:
                continue
            store.adam_step = max(store.adam_steps.values())
            keep_shadow = store.shadow is not None and store._shadow_version == store.master._version
            for (gi, step), ranges in per_group.items():
                g = self.param_groups[gi]
                b1, b2 = g["betas"]
                for a, b in _merge_ranges(ranges):
                    ops.adamw(store.master[a:b], store.grad[a:b], store.adam_m[a:b], store.adam_v[a:b],
                              store.shadow[a:b] if store.shadow is not None else None, float(g["lr"]), b1, b2,
                              g["eps"], g["weight_decay"], step, self.grad_scale)
            # kernel writes do not bump the autograd version counter, so a shadow that was
            # current before the step (and was rewritten by the kernel) is still current
            if keep_shadow:
                store.mark_shadow_fresh()
            elif store.shadow is not None:
                store._shadow_version = -1
        return None

    # -- checkpointing: param_groups as torch keeps them, plus the flat moments and step counts of every store
    def state_dict(self):
        if self._stores is None:
            self._index()
        sd = super().state_dict()
        sd["flat_stores"] = [{"numel": store.master.numel(),
                              "adam_m": None if store.adam_m is None else store.adam_m.detach().clone(),
                              "adam_v": None if store.adam_v is None else store.adam_v.detach().clone(),
                              "adam_steps": dict(store.adam_steps)} for store, _ in self._stores]
        return sd

    def load_state_dict(self, state_dict):
        if self._stores is None:
            self._index()
        flat = state_dict.get("flat_stores")
        if flat is None or len(flat) != len(self._stores):
            raise ValueError("optimizer state does not belong to these parameter stores")
        for (store, _), rec in zip(self._stores, flat):
            if rec["numel"] != store.master.numel():
                raise ValueError("optimizer state was saved for a model of another size")
        super().load_state_dict({k: v for k, v in state_dict.items() if k != "flat_stores"})
        for (store, _), rec in zip(self._stores, flat):
            for name in ("adam_m", "adam_v"):
                t = rec[name]
                setattr(store, name, None if t is None else t.to(device=store.master.device, dtype=torch.float32).clone())
            store.adam_steps = {int(k): int(v) for k, v in rec["adam_steps"].items()}
            store.adam_step = max(store.adam_steps.values(), default=0)
