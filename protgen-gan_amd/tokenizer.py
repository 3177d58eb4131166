"""The slice of BertTokenizer the path uses (generate.py:50-52,139,160,231): per-residue
encode, special-token ids, decode(skip_special_tokens=True).  Reads `vocab.txt` from a
checkpoint/tokenizer directory when present, else the public ProtBERT vocabulary."""
from __future__ import annotations

import os
from typing import List, Optional

PROTBERT_VOCAB = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]",
                  "L", "A", "G", "V", "E", "S", "I", "K", "R", "D", "T", "P", "N", "Q", "F", "Y",
                  "M", "H", "C", "W", "X", "U", "B", "Z", "O"]


class ProtTokenizer:
    def __init__(self, vocab: Optional[List[str]] = None):
        self.vocab = list(vocab or PROTBERT_VOCAB)
        self.tok2id = {t: i for i, t in enumerate(self.vocab)}
        for name in ("pad", "unk", "cls", "sep", "mask"):
            tok = f"[{name.upper()}]"
            setattr(self, f"{name}_token", tok if tok in self.tok2id else None)
            setattr(self, f"{name}_token_id", self.tok2id.get(tok))
        self._special = {i for i in (self.pad_token_id, self.unk_token_id, self.cls_token_id, self.sep_token_id,
                                     self.mask_token_id) if i is not None}

    @classmethod
    def from_pretrained(cls, path: str, do_lower_case: bool = False, **_):
        vf = os.path.join(path, "vocab.txt") if os.path.isdir(path) else path
        if os.path.exists(vf):
            with open(vf, encoding="utf-8") as f:
                vocab = [line.rstrip("\n") for line in f if line.rstrip("\n") != ""]
            return cls(vocab)
        return cls()

    def __len__(self):
        return len(self.vocab)

    def add_special_tokens(self, mapping):
        for key, tok in mapping.items():
            if tok not in self.tok2id:
                self.tok2id[tok] = len(self.vocab)
                self.vocab.append(tok)
            name = key.replace("_token", "")
            setattr(self, f"{name}_token", tok)
            setattr(self, f"{name}_token_id", self.tok2id[tok])
            self._special.add(self.tok2id[tok])

    def convert_ids_to_tokens(self, ids):
        return [self.vocab[i] for i in ids]

    def encode(self, text: str, add_special_tokens: bool = True) -> List[int]:
        unk = self.unk_token_id
        ids = [self.tok2id.get(t, unk) for t in text.split()]
        if add_special_tokens:
            ids = [self.cls_token_id] + ids + [self.sep_token_id]
        return ids

    def decode(self, ids, skip_special_tokens: bool = False) -> str:
        if hasattr(ids, "tolist"):
            ids = ids.tolist()
        toks = [self.vocab[i] for i in ids if not (skip_special_tokens and i in self._special)]
        return " ".join(toks)

    def save_pretrained(self, path: str):
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "vocab.txt"), "w", encoding="utf-8") as f:
            f.write("\n".join(self.vocab) + "\n")
