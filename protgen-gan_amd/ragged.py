"""Row lengths of a padded [B, S] batch, as the index tensors and C descriptors the ragged encoder passes take.

The reference pads every row to 512 (dataset.py:19-42, fully_masked_train.py:252-256) and hides the padding
with the attention mask.  Columns after a row's last visible one are never read by a visible row and receive
exactly zero gradient, so the critic's passes run on each row's leading columns only, rows laid back to back."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import torch

from . import _lib


def extents(mask: torch.Tensor) -> List[int]:
    """Last visible column + 1 of every row of a 0/1 mask [B, S] (a row without a visible column counts as
    full: its softmax is uniform over all S keys).  One device-to-host copy."""
    S = mask.shape[1]
    cols = torch.arange(1, S + 1, device=mask.device)
    ext = ((mask != 0) * cols).amax(dim=1)
    ext = torch.where(ext == 0, torch.full_like(ext, S), ext)
    return [int(v) for v in ext.tolist()]


def plan_lengths(ext: Sequence[int], S: int, floor: Optional[Sequence[int]] = None, multiple: int = 64) -> List[int]:
    """Lengths the kernels accept for rows whose visible columns end at `ext`: multiples of 8 in 8..S, at least
    `floor[b]`, with extra chunks of 8 handed out so that the total is a multiple of `multiple` whenever some row
    has room (the weight-gradient products reduce over the token rows in whole 64-row tiles).  Rows gained this
    way are padding columns of the source batch: dead rows, as in the rectangular pass."""
    if S % 8:
        raise ValueError("the padded width must be a multiple of 8")
    lens = []
    for b, e in enumerate(ext):
        if e < 0 or e > S:
            raise ValueError(f"row extent {e} outside 0..{S}")
        n = max(8, (int(e) + 7) // 8 * 8)
        if floor is not None:
            n = max(n, int(floor[b]))
        lens.append(min(S, n))
    need = (-sum(lens)) % multiple
    b = 0
    while need > 0 and b < len(lens):
        room = S - lens[b]
        take = min(room, need)
        lens[b] += take
        need -= take
        b += 1
    return lens


class Ragged:
    """`lengths[b]` leading columns of row b of a [B, S] batch, rows laid back to back (T token rows)."""

    def __init__(self, lengths: Sequence[int], S: int, device):
        lengths = [int(v) for v in lengths]
        if not lengths or min(lengths) < 8 or max(lengths) > S or any(v % 8 for v in lengths):
            raise ValueError("ragged lengths must be multiples of 8 in 8..S")
        off = [0]
        for v in lengths:
            off.append(off[-1] + v)
        self.lengths, self.S, self.B, self.T, self.off = lengths, int(S), len(lengths), off[-1], off
        self.device = torch.device(device)
        self.off_host = (C.c_int32 * (self.B + 1))(*off)  # the host copy the C side reads; lives as long as this object
        rows = torch.repeat_interleave(torch.arange(self.B), torch.tensor(lengths))
        pos = torch.arange(self.T) - torch.tensor(off[:-1])[rows]
        packed = torch.stack([rows * S + pos, pos, rows]).to(self.device)  # one host-to-device copy
        self.idx = packed[0]                                 # flat index into [B*S]
        self.pos = packed[1].to(torch.int32)                 # position of each token row in its sequence
        self.seq_id = packed[2].to(torch.int32)              # sequence of each token row
        self.seq_off = torch.tensor(off, dtype=torch.int32).to(self.device)
        self.cls_rows = self.seq_off[:-1].to(torch.int64)    # token row of each sequence's first position

    def saves(self) -> bool:
        return self.T < self.B * self.S

    def pack(self, x: torch.Tensor) -> torch.Tensor:
        """[B, S, ...] -> [T, ...]"""
        return x.reshape(self.B * self.S, *x.shape[2:]).index_select(0, self.idx)

    def unpack(self, xp: torch.Tensor) -> torch.Tensor:
        """[T, ...] -> [B, S, ...] with zeros in the columns left out."""
        out = torch.zeros(self.B * self.S, *xp.shape[1:], dtype=xp.dtype, device=xp.device)
        out.index_copy_(0, self.idx, xp)
        return out.view(self.B, self.S, *xp.shape[1:])

    def desc(self, first: int = 0, count: Optional[int] = None, stash_T: int = 0, adj_T: int = 0) -> "_lib.RaggedDesc":
        """C descriptor of the sequences [first, first + count) (a pass over part of what a forward covered)."""
        count = self.B - first if count is None else count
        if first < 0 or count < 1 or first + count > self.B:
            raise ValueError("sequence range outside the batch")
        d = _lib.RaggedDesc()
        d.max_len = self.S
        d.seq_off = self.seq_off.data_ptr() + 4 * first
        d.seq_off_host = C.addressof(self.off_host) + 4 * first
        d.stash_T, d.adj_T = int(stash_T), int(adj_T)
        d._keep = self  # the descriptor points into this object's buffers
        return d

    def tokens(self, first: int = 0, count: Optional[int] = None) -> int:
        count = self.B - first if count is None else count
        return self.off[first + count] - self.off[first]
