#!/usr/bin/env python3
"""Mass generation with the reference's entry points and flags (generate.py:19-275), on the
HIP engine.  Helper names, argument meaning, RNG use (Python `random.Random(rng_seed)` for
lengths / seeds / kept positions, the device generator for sampling noise) and error
behaviour follow the reference; the model and tokenizer are this package's own.

Sharded mass generation (one process per GPU, no collective) lives in distributed.py."""
from __future__ import annotations

import argparse
import math
import os
import random
from pathlib import Path
from typing import List, Tuple

import torch

from .bert import ProtBertForMaskedLM
from .models import Generator
from .tokenizer import ProtTokenizer


def _paths():
    """config.py:12-28 — resolved lazily so importing this module never raises."""
    src = os.environ.get("SOURCE_DIR")
    if not src:
        raise EnvironmentError("SOURCE_DIR is not set: export SOURCE_DIR=<dir holding checkpoints/ and models/>")
    return (os.path.join(src, "checkpoints"), os.path.join(src, "models", "dnmt_protbert"),
            os.path.join(src, "models", "prot_bert"))


# ---------------- utils ----------------
def load_length_distribution(dataset_path: str, max_len: int, min_len: int = 30) -> List[int]:
    lengths = []
    with open(dataset_path, "r", encoding="utf-8") as f:
        for line in f:
            seq = line.strip().replace(" ", "")
            if not seq:
                continue
            if min_len <= len(seq) <= max_len:
                lengths.append(len(seq))
    if not lengths:
        raise ValueError(f"No sequence lengths found in range [{min_len}, {max_len}] from {dataset_path}.")
    return lengths


def load_seed_sequences(seed_file: str, max_len: int, min_len: int = 30) -> List[str]:
    seqs = []
    with open(seed_file, "r", encoding="utf-8") as f:
        for line in f:
            s = line.strip().replace(" ", "")
            if not s:
                continue
            if min_len <= len(s) <= max_len:
                seqs.append(s[:max_len])
    if not seqs:
        raise ValueError(f"No usable seed sequences in range [{min_len}, {max_len}] from {seed_file}.")
    return seqs


def sample_sequence_length(lengths: List[int], rng: random.Random) -> int:
    return rng.choice(lengths)


def to_token_ids(seq: str, tokenizer) -> List[int]:
    return tokenizer.encode(" ".join(list(seq)), add_special_tokens=False)


def safe_local_from_pretrained(path: str, device: str, compute_dtype=torch.bfloat16):
    if not os.path.isdir(path):
        raise FileNotFoundError(f"Model dir not found: {path}")
    if not os.path.exists(os.path.join(path, "config.json")):
        raise FileNotFoundError(f"Missing config.json in {path} (did save_pretrained run?)")
    return ProtBertForMaskedLM.from_pretrained(path, device=device, compute_dtype=compute_dtype)


def resolve_model_path(ckpt_id: str) -> Tuple[str, str]:
    ckpt_root, finetuned, base = _paths()
    ckpt_id = ckpt_id.strip().strip("/")
    if ckpt_id in {"finetuned_protbert", "finetuned-protbert"}:
        return finetuned, "finetuned"
    if ckpt_id in {"protbert_base", "protbert-base"}:
        return base, "base"
    return os.path.join(ckpt_root, ckpt_id, "generator_bert"), "trained"


def derive_out_prefix(ckpt_id: str) -> str:
    ckpt_id = ckpt_id.strip().strip("/")
    if ckpt_id in {"finetuned-protbert", "finetuned_protbert", "protbert_base", "protbert-base", "base_protbert"}:
        return ckpt_id
    parts = ckpt_id.split("/")
    if len(parts) == 2:
        return f"{parts[0]}_{parts[1]}"
    return ckpt_id.replace("/", "_")


# -------------- builders (templates are assembled on the host, one copy to the device) --------------
def build_fully_masked_batch(batch_sizes: List[int], cls_id: int, sep_id: int, mask_id: int, pad_id: int,
                             device) -> Tuple[torch.Tensor, torch.Tensor]:
    max_L = max(batch_sizes)
    B = len(batch_sizes)
    ids = torch.full((B, max_L + 2), pad_id, dtype=torch.long)
    attn = torch.zeros_like(ids)
    for i, L in enumerate(batch_sizes):
        attn[i, :L + 2] = 1
        ids[i, 0] = cls_id
        ids[i, L + 1] = sep_id
        if L > 0:
            ids[i, 1:L + 1] = mask_id
    return ids.to(device), attn.to(device)


def build_seeded_batch(seqs: List[str], tokenizer, keep_init: float, cls_id: int, sep_id: int, mask_id: int,
                       pad_id: int, device, rng: random.Random) -> Tuple[torch.Tensor, torch.Tensor]:
    tok_lists = [to_token_ids(s, tokenizer) for s in seqs]
    max_L = max(len(t) for t in tok_lists)
    B = len(seqs)
    ids = torch.full((B, max_L + 2), pad_id, dtype=torch.long)
    attn = torch.zeros_like(ids)
    for i, toks in enumerate(tok_lists):
        L = len(toks)
        attn[i, :L + 2] = 1
        ids[i, 0] = cls_id
        ids[i, L + 1] = sep_id
        if L > 0:
            ids[i, 1:L + 1] = mask_id
            k = max(1, int(round(keep_init * L)))
            for pos in rng.sample(range(1, L + 1), k):
                ids[i, pos] = toks[pos - 1]
    return ids.to(device), attn.to(device)


def _decode_batch(tokenizer, ids: torch.Tensor) -> List[str]:
    rows = ids.cpu()
    return [tokenizer.decode(rows[i], skip_special_tokens=True).replace(" ", "") for i in range(rows.shape[0])]


def _masks_left(ids, mask_id) -> int:
    return int((ids == mask_id).sum().item())


# -------------- generators --------------
@torch.no_grad()
def generate_full_mode_sequences(generator: Generator, tokenizer, num_sequences: int, length_pool: List[int],
                                 keep_percent: float = 0.10, temperature: float = 1.0, batch_size: int = 64,
                                 device: str = "cuda", rng_seed: int = 42) -> List[str]:
    assert 0.0 < keep_percent <= 1.0
    device = torch.device(device)
    generator.eval()
    rng = random.Random(rng_seed)
    mask_id, cls_id, sep_id, pad_id = (tokenizer.mask_token_id, tokenizer.cls_token_id, tokenizer.sep_token_id,
                                       tokenizer.pad_token_id)
    if pad_id is None:
        raise ValueError("Tokenizer has no pad_token_id; please set one.")
    out, n_done = [], 0
    max_steps = math.ceil(1.0 / keep_percent)
    while n_done < num_sequences:
        this_batch = min(batch_size, num_sequences - n_done)
        batch_lengths = [sample_sequence_length(length_pool, rng) for _ in range(this_batch)]
        input_ids, attn = build_fully_masked_batch(batch_lengths, cls_id, sep_id, mask_id, pad_id, device)
        current_rate = 1.0
        for _ in range(max_steps):
            if _masks_left(input_ids, mask_id) == 0:
                break
            input_ids = generator.generate(input_ids, attention_mask=attn, keep_percent=keep_percent,
                                           current_rate=current_rate, temperature=temperature)
            current_rate = max(0.0, current_rate - keep_percent)
        out.extend(_decode_batch(tokenizer, input_ids))
        n_done += this_batch
    return out


@torch.no_grad()
def generate_seeded_mode_sequences(generator: Generator, tokenizer, num_sequences: int, seed_pool: List[str],
                                   keep_init: float = 0.10, keep_percent: float = 0.10, temperature: float = 1.0,
                                   batch_size: int = 64, device: str = "cuda", rng_seed: int = 123) -> List[str]:
    assert 0.0 < keep_init < 1.0 and 0.0 < keep_percent <= 1.0
    device = torch.device(device)
    generator.eval()
    rng = random.Random(rng_seed)
    mask_id, cls_id, sep_id, pad_id = (tokenizer.mask_token_id, tokenizer.cls_token_id, tokenizer.sep_token_id,
                                       tokenizer.pad_token_id)
    if pad_id is None:
        raise ValueError("Tokenizer has no pad_token_id; please set one.")
    out, n_done = [], 0
    max_steps = max(1, math.ceil((1.0 - keep_init) / keep_percent)) + 1
    N = len(seed_pool)
    while n_done < num_sequences:
        this_batch = min(batch_size, num_sequences - n_done)
        batch_seqs = [seed_pool[rng.randrange(N)] for _ in range(this_batch)]
        input_ids, attn = build_seeded_batch(batch_seqs, tokenizer, keep_init, cls_id, sep_id, mask_id, pad_id, device, rng)
        current_rate = 1.0 - keep_init
        for _ in range(max_steps):
            if _masks_left(input_ids, mask_id) == 0:
                break
            input_ids = generator.generate(input_ids, attention_mask=attn, keep_percent=keep_percent,
                                           current_rate=current_rate, temperature=temperature)
            current_rate = max(0.0, current_rate - keep_percent)
        out.extend(_decode_batch(tokenizer, input_ids))
        n_done += this_batch
    return out


# -------------- main --------------
def main(argv=None):
    random_seed = random.randint(1, 10000)
    ap = argparse.ArgumentParser(description="Generate sequences in full & seeded modes.")
    ap.add_argument("--ckpt_id", required=True)
    ap.add_argument("--dataset_path", type=str, default="./data/dnmt_unformatted.txt")
    ap.add_argument("--seed_file", type=str, default=None)
    ap.add_argument("--num_full", type=int, default=5000)
    ap.add_argument("--num_seeded", type=int, default=5000)
    ap.add_argument("--keep_init", type=float, default=0.10)
    ap.add_argument("--keep_percent", type=float, default=0.10)
    ap.add_argument("--temperature", type=float, default=1.0)
    ap.add_argument("--max_len", type=int, default=600)
    ap.add_argument("--min_len", type=int, default=30)
    ap.add_argument("--batch_size", type=int, default=64)
    ap.add_argument("--device", type=str, default="cuda")
    ap.add_argument("--rng_seed", type=int, default=random_seed)
    ap.add_argument("--tokenizer_dir", type=str, default=None, help="Tokenizer path (defaults to finetuned).")
    ap.add_argument("--out_dir", type=str, default="./results/")
    ap.add_argument("--dtype", choices=["bf16", "fp32"], default="fp32",
                    help="compute dtype (the reference generates in fp32; bf16 is the fast path)")
    args = ap.parse_args(argv)

    os.makedirs(args.out_dir, exist_ok=True)
    out_prefix = derive_out_prefix(args.ckpt_id)
    full_out = str(Path(args.out_dir) / f"{out_prefix}_full.txt")
    seeded_out = str(Path(args.out_dir) / f"{out_prefix}_seeded.txt")

    model_path, tag = resolve_model_path(args.ckpt_id)
    tokenizer = ProtTokenizer.from_pretrained(args.tokenizer_dir or _paths()[1], do_lower_case=False)
    if tokenizer.pad_token is None:
        tokenizer.add_special_tokens({"pad_token": "[PAD]"})
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    protbert = safe_local_from_pretrained(model_path, args.device, dtype)
    generator = Generator(protbert_model=protbert)
    generator.eval()

    lengths = load_length_distribution(args.dataset_path, max_len=args.max_len, min_len=args.min_len)
    seed_source = args.seed_file if args.seed_file is not None else args.dataset_path
    seeds = load_seed_sequences(seed_source, max_len=args.max_len, min_len=args.min_len)

    seqs_full = []
    if args.num_full > 0:
        seqs_full = generate_full_mode_sequences(generator=generator, tokenizer=tokenizer, num_sequences=args.num_full,
                                                 length_pool=lengths, keep_percent=args.keep_percent,
                                                 temperature=args.temperature, batch_size=args.batch_size,
                                                 device=args.device, rng_seed=args.rng_seed)
        with open(full_out, "w", encoding="utf-8") as f:
            for s in seqs_full:
                f.write(s + "\n")
        print(f"[{tag}] Wrote {len(seqs_full)} full-mode sequences → {full_out}")
    seqs_seeded = []
    if args.num_seeded > 0:
        seqs_seeded = generate_seeded_mode_sequences(generator=generator, tokenizer=tokenizer,
                                                     num_sequences=args.num_seeded, seed_pool=seeds,
                                                     keep_init=args.keep_init, keep_percent=args.keep_percent,
                                                     temperature=args.temperature, batch_size=args.batch_size,
                                                     device=args.device, rng_seed=args.rng_seed + 1337)
        with open(seeded_out, "w", encoding="utf-8") as f:
            for s in seqs_seeded:
                f.write(s + "\n")
        print(f"[{tag}] Wrote {len(seqs_seeded)} seeded-10% sequences → {seeded_out}")
    print(f"[DONE] {out_prefix} | full={len(seqs_full)} | seeded={len(seqs_seeded)}")


if __name__ == "__main__":
    main()
