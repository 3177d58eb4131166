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


class BatchLoader:
    """Batches of `batch_size` rows of `data` over the row indices `subset` (all rows by default), reshuffled on every
    `iter()` from `generator` (a CPU torch.Generator; the global one if None), incomplete last batch dropped.  With
    world > 1 a global batch is world * batch_size rows and this rank gets rows rank, rank + world, ... of it, so
    every rank builds the same permutation (same generator seed) and no two ranks see the same row."""

    def __init__(self, data: TokenizedSet, batch_size: int, device, subset: Optional[Sequence[int]] = None,
                 generator: Optional[torch.Generator] = None, rank: int = 0, world: int = 1):
        self.data, self.batch_size, self.device = data, batch_size, device
        self.subset = list(range(len(data))) if subset is None else list(subset)
        self.generator, self.rank, self.world = generator, rank, world

    def __len__(self):
        return len(self.subset) // (self.batch_size * self.world)

    def __iter__(self) -> Iterator[Dict[str, object]]:
        perm = torch.randperm(len(self.subset), generator=self.generator).tolist()
        g = self.batch_size * self.world
        for b in range(len(self)):
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
