"""The data side of the trainers, natively: one-sequence-per-line text files -> fixed-width id / mask tensors,
per-epoch shuffled batches, the per-epoch generator / critic split of the seeded trainer, and the length pool of
the blind trainer.  Stands in for the reference's HuggingFace `datasets` + `DataLoader` pipeline
(dataset.py:5-94) without either dependency.

Contract kept from dataset.py: every row is `[CLS] residues [SEP]` truncated and padded with [PAD] to
`max_length` (dataset.py:19-25), ids int64, attention mask float32 with 1 on the tokens (dataset.py:39-42);
batches are drawn without replacement from a fresh permutation each epoch and an incomplete last batch is
dropped (dataset.py:59-64).  Added: every batch also carries its rows' token counts as a host list
(`lengths`), which the critic's ragged passes use instead of reading the masks back from the device, and a
batch iterator can hand a rank its share of every global batch (data-parallel training)."""
from __future__ import annotations

from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import torch

from .tokenizer import ProtTokenizer


def read_sequences(path: str) -> List[str]:
    """Non-empty lines; residues may be separated by blanks ("M K T") or not ("MKT")."""
    out = []
    with open(path, "r", encoding="utf-8") as fh:
        for raw in fh:
            s = "".join(raw.split())
            if s:
                out.append(s)
    return out


class TokenizedSet:
    """ids [N, max_length] int64, mask [N, max_length] float32 (host tensors), lengths [N] (tokens incl. CLS/SEP)."""

    def __init__(self, ids: torch.Tensor, mask: torch.Tensor, lengths: List[int]):
        self.ids, self.mask, self.lengths = ids, mask, lengths

    def __len__(self):
        return self.ids.shape[0]

    @classmethod
    def from_sequences(cls, seqs: Sequence[str], tokenizer: ProtTokenizer, max_length: int = 512) -> "TokenizedSet":
        N = len(seqs)
        ids = torch.full((N, max_length), tokenizer.pad_token_id, dtype=torch.long)
        mask = torch.zeros(N, max_length, dtype=torch.float32)
        lengths = []
        for i, s in enumerate(seqs):
            toks = tokenizer.encode(" ".join(s[:max_length - 2]), add_special_tokens=True)
            ids[i, :len(toks)] = torch.as_tensor(toks)
            mask[i, :len(toks)] = 1.0
            lengths.append(len(toks))
        return cls(ids, mask, lengths)

    @classmethod
    def from_file(cls, path: str, tokenizer: ProtTokenizer, max_length: int = 512) -> "TokenizedSet":
        return cls.from_sequences(read_sequences(path), tokenizer, max_length)

    def batch(self, index: Sequence[int], device) -> Dict[str, object]:
        idx = torch.as_tensor(list(index), dtype=torch.long)
        return {"input_ids": self.ids[idx].to(device), "attention_mask": self.mask[idx].to(device),
                "lengths": [self.lengths[i] for i in idx.tolist()]}


def balanced_deal(lengths: Sequence[int], world: int) -> List[List[int]]:
    """Deals the rows of one global batch to `world` ranks so that every rank gets the same number of rows and
    nearly the same number of tokens: rows sorted by length, longest first, handed out in rounds of `world` whose
    direction alternates (0..world-1, then world-1..0, ...), then single rows traded between the fullest and the
    emptiest rank while that helps.  Deterministic: every rank computes the same deal from the same lengths.  Returns, per rank, the indices into `lengths` it
    receives.  Which rank holds which row does not change the averaged gradient (the losses are batch means over
    rows of equal weight, loss.py:4,95); it changes how long the slowest rank of a step takes, since every rank
    waits for that one at each gradient reduction.  len(lengths) must be a multiple of world."""
    n = len(lengths)
    if world <= 0 or n % world:
        raise ValueError(f"{n} rows cannot be dealt evenly to {world} ranks")
    order = sorted(range(n), key=lambda i: (-int(lengths[i]), i))
    out: List[List[int]] = [[] for _ in range(world)]
    for rnd in range(n // world):
        chunk = order[rnd * world:(rnd + 1) * world]
        if rnd % 2:
            chunk = chunk[::-1]
        for r, i in enumerate(chunk):
            out[r].append(i)
    # then trade single rows between the fullest and the emptiest rank while that narrows the gap between them
    sums = [sum(int(lengths[i]) for i in rows) for rows in out]
    for _ in range(4 * n):
        hi = max(range(world), key=lambda r: (sums[r], r))
        lo = min(range(world), key=lambda r: (sums[r], r))
        gap = sums[hi] - sums[lo]
        best = None
        for a, i in enumerate(out[hi]):
            for b, j in enumerate(out[lo]):
                d = int(lengths[i]) - int(lengths[j])
                if 0 < d < gap and (best is None or abs(gap - 2 * d) < best[0]):
                    best = (abs(gap - 2 * d), a, b, d)
        if best is None or best[0] >= gap:
            break
        _, a, b, d = best
        out[hi][a], out[lo][b] = out[lo][b], out[hi][a]
        sums[hi] -= d
        sums[lo] += d
    return out


class BatchLoader:
    """Batches of `batch_size` rows of `data` over the row indices `subset` (all rows by default), reshuffled on every
    `iter()` from `generator` (a CPU torch.Generator; the global one if None), incomplete last batch dropped.  With
    world > 1 a global batch is world * batch_size rows of the permutation every rank builds alike (same generator
    seed), and no two ranks see the same row: with `balance` (default) the global batch is dealt by length
    (balanced_deal), so the ranks' token counts stay within a few per cent of each other; without it this rank gets
    rows rank, rank + world, ... of the global batch."""

    def __init__(self, data: TokenizedSet, batch_size: int, device, subset: Optional[Sequence[int]] = None,
                 generator: Optional[torch.Generator] = None, rank: int = 0, world: int = 1, balance: bool = True):
        self.data, self.batch_size, self.device = data, batch_size, device
        self.subset = list(range(len(data))) if subset is None else list(subset)
        self.generator, self.rank, self.world, self.balance = generator, rank, world, balance

    def __len__(self):
        return len(self.subset) // (self.batch_size * self.world)

    def __iter__(self) -> Iterator[Dict[str, object]]:
        perm = torch.randperm(len(self.subset), generator=self.generator).tolist()
        g = self.batch_size * self.world
        for b in range(len(self)):
            if self.world > 1 and self.balance:
                glob = [self.subset[r] for r in perm[b * g:(b + 1) * g]]
                mine = balanced_deal([self.data.lengths[i] for i in glob], self.world)[self.rank]
                yield self.data.batch([glob[i] for i in mine], self.device)
                continue
            rows = perm[b * g + self.rank:(b + 1) * g:self.world]
            yield self.data.batch([self.subset[r] for r in rows], self.device)


def dynamic_split(n_rows: int, n_critic: int, generator: Optional[torch.Generator] = None) -> Tuple[List[int], List[int]]:
    """dataset.py:72-94: a fresh permutation every epoch; the first n_rows // (n_critic + 1) rows feed the generator,
    the rest the critic."""
    perm = torch.randperm(n_rows, generator=generator).tolist()
    k = n_rows // (n_critic + 1)
    return perm[:k], perm[k:]


def length_pool(path: str) -> List[int]:
    """Line lengths of the file the blind trainer samples fake lengths from (val_metrics.py:48-53)."""
    return [len(s) for s in read_sequences(path)]
