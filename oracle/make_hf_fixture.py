#!/usr/bin/env python3
"""Write a small HuggingFace-format checkpoint with `transformers` itself (dev container only): the on-disk
layout the reference's scripts produce and read (`save_pretrained` at fully_masked_train.py:378-389,
`from_pretrained` at generate.py:54-60), so the loader in protgen-gan_amd/bert.py is tested against files
it did not write.

  tests/golden/hf_tiny/       config.json + model.safetensors   (transformers' own save_pretrained)
  tests/golden/hf_tiny_bin/   config.json + pytorch_model.bin   (a torch.save of state_dict(): older layout,
                                                                 includes the tied cls.predictions.decoder.* keys)
  tests/golden/hf_tiny_io.npz input ids / mask and the logits transformers computes for them (fp32, eager)

    python oracle/make_hf_fixture.py
"""
import os
import shutil
import sys

import numpy as np
import torch
from transformers import BertConfig, BertForMaskedLM

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def main():
    torch.manual_seed(2024)
    cfg = BertConfig(vocab_size=30, hidden_size=64, num_hidden_layers=2, num_attention_heads=1, intermediate_size=128,
                     max_position_embeddings=48, type_vocab_size=2, hidden_dropout_prob=0.0,
                     attention_probs_dropout_prob=0.0, layer_norm_eps=1e-12, attn_implementation="eager")
    m = BertForMaskedLM(cfg).eval()
    with torch.no_grad():  # biases and LayerNorm parameters away from their 0 / 1 initial values
        for n, p in m.named_parameters():
            if n.endswith("bias"):
                p.normal_(0.0, 0.05)
            elif "LayerNorm.weight" in n:
                p.add_(0.1 * torch.randn_like(p))
    d1, d2 = os.path.join(GOLD, "hf_tiny"), os.path.join(GOLD, "hf_tiny_bin")
    for d in (d1, d2):
        shutil.rmtree(d, ignore_errors=True)
    m.save_pretrained(d1)
    os.makedirs(d2)
    shutil.copy(os.path.join(d1, "config.json"), os.path.join(d2, "config.json"))
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    assert "cls.predictions.decoder.weight" in sd
    torch.save(sd, os.path.join(d2, "pytorch_model.bin"))
    g = torch.Generator().manual_seed(7)
    ids = torch.randint(5, 25, (3, 24), generator=g)
    ids[:, 0] = 2
    mask = torch.ones(3, 24, dtype=torch.long)
    mask[1, 17:] = 0
    ids[1, 17:] = 0
    ids[2, 5:9] = 4
    with torch.no_grad():
        logits = m(input_ids=ids, attention_mask=mask).logits
    np.savez(os.path.join(GOLD, "hf_tiny_io.npz"), ids=ids.numpy(), mask=mask.numpy(), logits=logits.numpy())
    print(sorted(os.listdir(d1)), sorted(os.listdir(d2)), float(logits.abs().max()))


if __name__ == "__main__":
    sys.exit(main())
