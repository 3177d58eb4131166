#!/usr/bin/env python3
"""Mass generation behind the reference's entry points and flags (generate.py:19-275), on the HIP engine.

What is kept from the reference is its surface: function names and signatures, CLI flags and their
defaults, the output file names, which exception each failure raises, and the order in which the Python
`random.Random(rng_seed)` stream is consumed (one `choice` per length, one `randrange` per seed, one
`sample` per seeded row, rows in batch order), because generated sets are reproducible by seed.  The rest
is this package's own:

  * batch templates are built as whole tensors on the host and copied to the device once;
  * the fill loops never wait for the device just to learn whether masks are left.  The host keeps a lower
    bound of every row's open masks (what the schedule leaves if no sampled token re-opens a position or
    lowers the row's commit quota; either can only leave MORE masks open), and only when that bound reaches
    zero for every row is the device count read, which is where the reference's `break` can fire at all;
  * sharded mass generation (one process per GPU, no collective) lives in distributed.py."""
from __future__ import annotations

import argparse
import math
import os
import random
from pathlib import Path
from typing import Iterator, List, Sequence, Tuple

import torch

from .bert import ProtBertForMaskedLM
from .models import Generator, RowPacking
from .tokenizer import ProtTokenizer

_FINETUNED_IDS = ("finetuned_protbert", "finetuned-protbert")
_BASE_IDS = ("protbert_base", "protbert-base")


def _source_dirs() -> Tuple[str, str, str]:
    """(checkpoint root, fine-tuned ProtBERT dir, base ProtBERT dir) under $SOURCE_DIR, the layout of the
    reference's config.py:24-28.  Looked up on use, so importing this module needs no environment."""
    root = os.environ.get("SOURCE_DIR")
    if not root:
        raise EnvironmentError("SOURCE_DIR is not set: export SOURCE_DIR=<dir holding checkpoints/ and models/>")
    models = os.path.join(root, "models")
    return os.path.join(root, "checkpoints"), os.path.join(models, "dnmt_protbert"), os.path.join(models, "prot_bert")


# ----------------------------------------------------------------------------- inputs
def _sequences_in_range(path: str, min_len: int, max_len: int) -> Iterator[str]:
    """Non-empty lines of a one-sequence-per-line file, spaces inside a line removed, length within bounds."""
    with open(path, "r", encoding="utf-8") as fh:
        for raw in fh:
            seq = raw.strip().replace(" ", "")
            if seq and min_len <= len(seq) <= max_len:
                yield seq


def load_length_distribution(dataset_path: str, max_len: int, min_len: int = 30) -> List[int]:
    pool = [len(s) for s in _sequences_in_range(dataset_path, min_len, max_len)]
    if not pool:
        raise ValueError(f"No sequence lengths found in range [{min_len}, {max_len}] from {dataset_path}.")
    return pool


def load_seed_sequences(seed_file: str, max_len: int, min_len: int = 30) -> List[str]:
    pool = list(_sequences_in_range(seed_file, min_len, max_len))
    if not pool:
        raise ValueError(f"No usable seed sequences in range [{min_len}, {max_len}] from {seed_file}.")
    return pool


def sample_sequence_length(lengths: List[int], rng: random.Random) -> int:
    return rng.choice(lengths)


def to_token_ids(seq: str, tokenizer) -> List[int]:
    return tokenizer.encode(" ".join(seq), add_special_tokens=False)


def safe_local_from_pretrained(path: str, device: str, compute_dtype=torch.bfloat16):
    for missing, what in ((not os.path.isdir(path), f"Model dir not found: {path}"),
                          (not os.path.exists(os.path.join(path, "config.json")),
                           f"Missing config.json in {path} (did save_pretrained run?)")):
        if missing:
            raise FileNotFoundError(what)
    return ProtBertForMaskedLM.from_pretrained(path, device=device, compute_dtype=compute_dtype)


def resolve_model_path(ckpt_id: str) -> Tuple[str, str]:
    key = ckpt_id.strip().strip("/")
    ckpt_root, finetuned, base = _source_dirs()
    if key in _FINETUNED_IDS:
        return finetuned, "finetuned"
    if key in _BASE_IDS:
        return base, "base"
    return os.path.join(ckpt_root, key, "generator_bert"), "trained"


def derive_out_prefix(ckpt_id: str) -> str:
    key = ckpt_id.strip().strip("/")
    return key if key in _FINETUNED_IDS + _BASE_IDS + ("base_protbert",) else "_".join(key.split("/"))


# ----------------------------------------------------------------------------- batch templates
def _template(lengths: Sequence[int], cls_id: int, sep_id: int, mask_id: int, pad_id: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Host tensors: ids [B, max L + 2] = CLS, L masks, SEP, pads; attention mask = 1 on the first L + 2 columns."""
    L = torch.as_tensor(list(lengths), dtype=torch.long).unsqueeze(1)
    col = torch.arange(int(L.max()) + 2).unsqueeze(0)
    ids = torch.full((L.shape[0], col.shape[1]), pad_id, dtype=torch.long)
    ids = torch.where(col <= L, torch.full_like(ids, mask_id), ids)
    ids = torch.where(col == L + 1, torch.full_like(ids, sep_id), ids)
    ids[:, 0] = cls_id
    return ids, (col <= L + 1).to(torch.long)


def build_fully_masked_batch(batch_sizes: List[int], cls_id: int, sep_id: int, mask_id: int, pad_id: int,
                             device) -> Tuple[torch.Tensor, torch.Tensor]:
    ids, attn = _template(batch_sizes, cls_id, sep_id, mask_id, pad_id)
    return ids.to(device), attn.to(device)


def build_seeded_batch(seqs: List[str], tokenizer, keep_init: float, cls_id: int, sep_id: int, mask_id: int,
                       pad_id: int, device, rng: random.Random) -> Tuple[torch.Tensor, torch.Tensor]:
    rows = [to_token_ids(s, tokenizer) for s in seqs]
    ids, attn = _template([len(r) for r in rows], cls_id, sep_id, mask_id, pad_id)
    for b, toks in enumerate(rows):  # (one `sample` per row, in row order: the seed's draw order)
        if toks:
            kept = torch.as_tensor(rng.sample(range(1, len(toks) + 1), max(1, int(round(keep_init * len(toks))))))
            ids[b, kept] = torch.as_tensor(toks)[kept - 1]
    return ids.to(device), attn.to(device)


# ----------------------------------------------------------------------------- fill loops
def _special_ids(tokenizer) -> Tuple[int, int, int, int]:
    if tokenizer.pad_token_id is None:
        raise ValueError("Tokenizer has no pad_token_id; please set one.")
    return tokenizer.mask_token_id, tokenizer.cls_token_id, tokenizer.sep_token_id, tokenizer.pad_token_id


def _row_packing(generator: Generator, ids: torch.Tensor, widths: List[int]):
    """Each row of a template holds its sequence in the leading `widths[b]` columns; the columns behind hold pad ids
    under attention mask 0 (generate.py:92-127): never a key, never a mask to fill.  The bf16 encoder pass therefore
    runs on the leading columns only, rows laid back to back — with lengths of 200..500 in a batch of 64 that is
    about 30 % fewer token rows — and yields the tokens of the full-width pass."""
    if generator.protbert.compute_dtype != torch.bfloat16 or sum(widths) >= ids.numel():
        return None
    return RowPacking(widths, ids.shape[1], ids.device)


def _fill(generator: Generator, ids: torch.Tensor, attn: torch.Tensor, open_masks: List[int], meaningful: List[int],
          first_rate: float, steps: int, keep_percent: float, temperature: float, mask_id: int,
          widths: List[int]) -> torch.Tensor:
    """Up to `steps` fill iterations at rates first_rate, first_rate - keep_percent, ...; stops before an iteration
    that would find no mask anywhere (the reference's per-iteration check, generate.py:147-149,185-187).
    open_masks / meaningful: per row, the masks in the template and its tokens outside {PAD, CLS, SEP};
    widths: per row, the columns its sequence occupies (the ones of its attention mask)."""
    low = list(open_masks)  # lower bound of each row's open masks (module docstring)
    rate = first_rate
    packing = _row_packing(generator, ids, widths)
    for _ in range(steps):
        if not any(low) and int((ids == mask_id).sum().item()) == 0:
            break
        kw = {} if packing is None else {"packing": packing}
        ids = generator.generate(ids, attention_mask=attn, keep_percent=keep_percent, current_rate=rate,
                                 temperature=temperature, **kw)
        fill_all = rate <= 0.1
        low = [0 if fill_all else m - min(max(1, int(keep_percent * n)), m) for m, n in zip(low, meaningful)]
        rate = max(0.0, rate - keep_percent)
    return ids


def _decode_rows(tokenizer, ids: torch.Tensor) -> List[str]:
    return [tokenizer.decode(row, skip_special_tokens=True).replace(" ", "") for row in ids.cpu().tolist()]


@torch.no_grad()
def generate_full_mode_sequences(generator: Generator, tokenizer, num_sequences: int, length_pool: List[int],
                                 keep_percent: float = 0.10, temperature: float = 1.0, batch_size: int = 64,
                                 device: str = "cuda", rng_seed: int = 42) -> List[str]:
    assert 0.0 < keep_percent <= 1.0
    dev = torch.device(device)
    generator.eval()
    rng = random.Random(rng_seed)
    mask_id, cls_id, sep_id, pad_id = _special_ids(tokenizer)
    steps = math.ceil(1.0 / keep_percent)
    sequences: List[str] = []
    for start in range(0, num_sequences, batch_size):
        lengths = [sample_sequence_length(length_pool, rng) for _ in range(min(batch_size, num_sequences - start))]
        ids, attn = build_fully_masked_batch(lengths, cls_id, sep_id, mask_id, pad_id, dev)
        ids = _fill(generator, ids, attn, lengths, lengths, 1.0, steps, keep_percent, temperature, mask_id,
                    [L + 2 for L in lengths])
        sequences += _decode_rows(tokenizer, ids)
    return sequences


@torch.no_grad()
def generate_seeded_mode_sequences(generator: Generator, tokenizer, num_sequences: int, seed_pool: List[str],
                                   keep_init: float = 0.10, keep_percent: float = 0.10, temperature: float = 1.0,
                                   batch_size: int = 64, device: str = "cuda", rng_seed: int = 123) -> List[str]:
    assert 0.0 < keep_init < 1.0 and 0.0 < keep_percent <= 1.0
    dev = torch.device(device)
    generator.eval()
    rng = random.Random(rng_seed)
    mask_id, cls_id, sep_id, pad_id = _special_ids(tokenizer)
    steps = max(1, math.ceil((1.0 - keep_init) / keep_percent)) + 1
    specials = (pad_id, cls_id, sep_id)
    sequences: List[str] = []
    for start in range(0, num_sequences, batch_size):
        n = min(batch_size, num_sequences - start)
        seeds = [seed_pool[rng.randrange(len(seed_pool))] for _ in range(n)]
        ids, attn = build_seeded_batch(seeds, tokenizer, keep_init, cls_id, sep_id, mask_id, pad_id, dev, rng)
        host = ids.cpu()
        open_masks = (host == mask_id).sum(dim=1).tolist()
        inner = host[:, 1:]  # (the row's own CLS / SEP / pads do not count; a residue that encodes to one of them does not either)
        meaningful = [int(((row != specials[0]) & (row != specials[1]) & (row != specials[2])).sum()) for row in inner]
        ids = _fill(generator, ids, attn, open_masks, meaningful, 1.0 - keep_init, steps, keep_percent, temperature, mask_id,
                    attn.sum(dim=1).tolist())
        sequences += _decode_rows(tokenizer, ids)
    return sequences


# ----------------------------------------------------------------------------- command line
def _parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(description="Generate sequences in full & seeded modes.")
    ap.add_argument("--ckpt_id", required=True)
    for flag, kind, default in (("--dataset_path", str, "./data/dnmt_unformatted.txt"), ("--seed_file", str, None),
                                ("--num_full", int, 5000), ("--num_seeded", int, 5000), ("--keep_init", float, 0.10),
                                ("--keep_percent", float, 0.10), ("--temperature", float, 1.0), ("--max_len", int, 600),
                                ("--min_len", int, 30), ("--batch_size", int, 64), ("--device", str, "cuda"),
                                ("--rng_seed", int, None), ("--out_dir", str, "./results/")):
        ap.add_argument(flag, type=kind, default=default)
    ap.add_argument("--tokenizer_dir", type=str, default=None, help="Tokenizer path (defaults to finetuned).")
    ap.add_argument("--dtype", choices=["bf16", "fp32"], default="fp32",
                    help="compute dtype (the reference generates in fp32; bf16 is the fast path)")
    return ap


def _write_lines(path: Path, lines: Sequence[str]) -> None:
    with open(path, "w", encoding="utf-8") as fh:
        fh.writelines(s + "\n" for s in lines)


def main(argv=None):
    args = _parser().parse_args(argv)
    if args.rng_seed is None:  # (the reference's default: a fresh seed per run)
        args.rng_seed = random.randint(1, 10000)
    out_dir = Path(args.out_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    prefix = derive_out_prefix(args.ckpt_id)
    model_dir, tag = resolve_model_path(args.ckpt_id)

    tokenizer = ProtTokenizer.from_pretrained(args.tokenizer_dir or _source_dirs()[1], do_lower_case=False)
    if tokenizer.pad_token is None:
        tokenizer.add_special_tokens({"pad_token": "[PAD]"})
    protbert = safe_local_from_pretrained(model_dir, args.device, torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    vocab = protbert.config.vocab_size
    too_big = {name: i for name, i in (("pad", tokenizer.pad_token_id), ("cls", tokenizer.cls_token_id),
                                       ("sep", tokenizer.sep_token_id), ("mask", tokenizer.mask_token_id))
               if i is None or not 0 <= i < vocab}
    if too_big:  # (the embedding table is not resized here: ids past it would index other tensors)
        raise ValueError(f"special token ids {too_big} do not fit the checkpoint's vocabulary of {vocab}")
    generator = Generator(protbert_model=protbert).eval()

    lengths = load_length_distribution(args.dataset_path, max_len=args.max_len, min_len=args.min_len)
    seeds = load_seed_sequences(args.seed_file or args.dataset_path, max_len=args.max_len, min_len=args.min_len)
    common = dict(keep_percent=args.keep_percent, temperature=args.temperature, batch_size=args.batch_size, device=args.device)
    counts = {}
    if args.num_full > 0:
        seqs = generate_full_mode_sequences(generator, tokenizer, args.num_full, lengths, rng_seed=args.rng_seed, **common)
        _write_lines(out_dir / f"{prefix}_full.txt", seqs)
        counts["full"] = len(seqs)
    if args.num_seeded > 0:
        seqs = generate_seeded_mode_sequences(generator, tokenizer, args.num_seeded, seeds, keep_init=args.keep_init,
                                              rng_seed=args.rng_seed + 1337, **common)
        _write_lines(out_dir / f"{prefix}_seeded.txt", seqs)
        counts["seeded"] = len(seqs)
    print(f"{tag} checkpoint {prefix}: " + ", ".join(f"{n} {mode}-mode sequences" for mode, n in counts.items()) +
          f" written under {out_dir}")


if __name__ == "__main__":
    main()
