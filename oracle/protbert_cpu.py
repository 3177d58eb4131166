"""CPU oracle for the ProtBERT GAN hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain PyTorch CPU ops (fp32 or fp64, no HuggingFace, no HIP),
the arithmetic the reference executes on its hot path.  It exists so the HIP engine
can be checked on a machine that has neither the reference nor `transformers`.
Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may
import it; the product package must never route through it.

Pinning: the reference ships no tests or golden vectors for this path, so this
restatement is pinned against outputs of the reference itself
(`/root/reference/models.py`, `/root/reference/loss.py`, `/root/reference/generate.py`
driven over `transformers.BertForMaskedLM`), captured by `oracle/make_golden.py` into
`tests/golden/*.npz` and checked by `tests/test_oracle_golden.py`.

What each function follows (reference file:line; HF = transformers/models/bert/modeling_bert.py):
  embeddings               HF:53-108      (word + token_type[0] + position -> LayerNorm eps 1e-12)
  extended_mask            transformers/modeling_utils.py:1012-1024  ((1-m) * finfo.min, key side)
  layer                    HF:111-136,164-203,282-293,325-351, activations.py:83 (erf GELU)
  mlm_head                 HF:466-496     (dense -> GELU -> LayerNorm -> tied decoder + bias)
  generator_forward        models.py:14-16
  generate_step            models.py:18-64
  critic_forward           models.py:86-106
  compute_gradient_penalty loss.py:10-96
  critic_loss/generator_loss  loss.py:3-7
  adamw_step               torch.optim.AdamW as used at fully_masked_train.py:74-75,326
  build_fully_masked_batch / build_seeded_batch / *_mode_sequences   generate.py:92-201
"""
from __future__ import annotations

import math
import random
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

PAD, UNK, CLS, SEP, MASK = 0, 1, 2, 3, 4

# ProtBERT vocabulary order (ids 0..29); only the five special ids are pinned by the
# reference (models.py:6), the residue order is the public Rostlab/prot_bert vocab.txt.
VOCAB = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]",
         "L", "A", "G", "V", "E", "S", "I", "K", "R", "D", "T", "P", "N", "Q", "F", "Y",
         "M", "H", "C", "W", "X", "U", "B", "Z", "O"]


@dataclass
class BertCfg:
    vocab_size: int = 30
    hidden_size: int = 1024
    num_hidden_layers: int = 30
    num_attention_heads: int = 16
    intermediate_size: int = 4096
    max_position_embeddings: int = 40000
    type_vocab_size: int = 2
    layer_norm_eps: float = 1e-12
    initializer_range: float = 0.02

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads


def param_names(cfg: BertCfg) -> List[str]:
    """HF state_dict key order of BertForMaskedLM (tied decoder weight omitted)."""
    n = ["bert.embeddings.word_embeddings.weight",
         "bert.embeddings.position_embeddings.weight",
         "bert.embeddings.token_type_embeddings.weight",
         "bert.embeddings.LayerNorm.weight",
         "bert.embeddings.LayerNorm.bias"]
    for i in range(cfg.num_hidden_layers):
        p = f"bert.encoder.layer.{i}."
        for s in ("attention.self.query", "attention.self.key", "attention.self.value",
                  "attention.output.dense"):
            n += [p + s + ".weight", p + s + ".bias"]
        n += [p + "attention.output.LayerNorm.weight", p + "attention.output.LayerNorm.bias"]
        n += [p + "intermediate.dense.weight", p + "intermediate.dense.bias",
              p + "output.dense.weight", p + "output.dense.bias",
              p + "output.LayerNorm.weight", p + "output.LayerNorm.bias"]
    n += ["cls.predictions.bias",
          "cls.predictions.transform.dense.weight", "cls.predictions.transform.dense.bias",
          "cls.predictions.transform.LayerNorm.weight", "cls.predictions.transform.LayerNorm.bias"]
    return n


def param_shape(cfg: BertCfg, name: str) -> Tuple[int, ...]:
    H, I, V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
    if name.endswith("word_embeddings.weight"):
        return (V, H)
    if name.endswith("position_embeddings.weight"):
        return (cfg.max_position_embeddings, H)
    if name.endswith("token_type_embeddings.weight"):
        return (cfg.type_vocab_size, H)
    if name == "cls.predictions.bias":
        return (V,)
    if "intermediate.dense" in name:
        return (I, H) if name.endswith("weight") else (I,)
    if ".output.dense" in name and "attention" not in name:
        return (H, I) if name.endswith("weight") else (H,)
    if "LayerNorm" in name:
        return (H,)
    return (H, H) if name.endswith("weight") else (H,)


def init_params(cfg: BertCfg, seed: int, dtype=torch.float32, bias_std: float = 0.0) -> Dict[str, torch.Tensor]:
    """Deterministic synthetic weights: N(0, initializer_range) matrices, LayerNorm 1/0,
    biases 0 (HF `_init_weights`), drawn name by name from one CPU generator.  `bias_std`
    > 0 gives biases and LayerNorm parameters a random perturbation so parity tests
    exercise them."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for n in param_names(cfg):
        shp = param_shape(cfg, n)
        if "LayerNorm.weight" in n:
            t = torch.ones(shp, dtype=torch.float32)
            if bias_std:
                t = t + bias_std * torch.randn(shp, generator=g)
        elif n.endswith("bias"):
            t = torch.zeros(shp, dtype=torch.float32)
            if bias_std:
                t = bias_std * torch.randn(shp, generator=g)
        else:
            t = torch.randn(shp, generator=g) * cfg.initializer_range
        out[n] = t.to(dtype)
    return out


def init_classifier(hidden: int, seed: int, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """nn.Sequential(Linear,ReLU,Linear,ReLU,Linear,ReLU,Linear) keys 0,2,4,6 (models.py:76-84)
    with nn.Linear's default init (uniform +-1/sqrt(fan_in)) from one CPU generator."""
    g = torch.Generator().manual_seed(seed)
    dims = [hidden, hidden // 2, hidden // 4, hidden // 16, 1]
    out = {}
    for li, k in enumerate((0, 2, 4, 6)):
        fi, fo = dims[li], dims[li + 1]
        bound = 1.0 / math.sqrt(fi)
        out[f"{k}.weight"] = ((torch.rand(fo, fi, generator=g) * 2 - 1) * bound).to(dtype)
        out[f"{k}.bias"] = ((torch.rand(fo, generator=g) * 2 - 1) * bound).to(dtype)
    return out


# ----------------------------------------------------------------------------- encoder
def gelu_erf(x):
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def layer_norm(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def embeddings(P, cfg: BertCfg, ids):
    S = ids.shape[1]
    e = P["bert.embeddings.word_embeddings.weight"][ids]
    e = e + P["bert.embeddings.token_type_embeddings.weight"][0]
    e = e + P["bert.embeddings.position_embeddings.weight"][:S].unsqueeze(0)
    return layer_norm(e, P["bert.embeddings.LayerNorm.weight"], P["bert.embeddings.LayerNorm.bias"],
                      cfg.layer_norm_eps)


def extended_mask(mask, dtype):
    """[B,S] 0/1 (any dtype) -> additive [B,1,1,S]."""
    m = mask.to(dtype)
    return (1.0 - m)[:, None, None, :] * torch.finfo(dtype).min


def layer(P, cfg: BertCfg, i: int, x, ext):
    p = f"bert.encoder.layer.{i}."
    B, S, H = x.shape
    nh, d = cfg.num_attention_heads, cfg.head_dim

    def lin(t, name):
        return F.linear(t, P[p + name + ".weight"], P[p + name + ".bias"])

    q = lin(x, "attention.self.query").view(B, S, nh, d).transpose(1, 2)
    k = lin(x, "attention.self.key").view(B, S, nh, d).transpose(1, 2)
    v = lin(x, "attention.self.value").view(B, S, nh, d).transpose(1, 2)
    sc = torch.matmul(q, k.transpose(2, 3)) * (d ** -0.5)
    if ext is not None:
        sc = sc + ext
    pr = torch.softmax(sc, dim=-1)
    ctx = torch.matmul(pr, v).transpose(1, 2).reshape(B, S, H)
    a = layer_norm(lin(ctx, "attention.output.dense") + x,
                   P[p + "attention.output.LayerNorm.weight"], P[p + "attention.output.LayerNorm.bias"],
                   cfg.layer_norm_eps)
    u = gelu_erf(lin(a, "intermediate.dense"))
    return layer_norm(lin(u, "output.dense") + a,
                      P[p + "output.LayerNorm.weight"], P[p + "output.LayerNorm.bias"], cfg.layer_norm_eps)


def encoder(P, cfg: BertCfg, x, mask):
    ext = None if mask is None else extended_mask(mask, x.dtype)
    for i in range(cfg.num_hidden_layers):
        x = layer(P, cfg, i, x, ext)
    return x


def mlm_head(P, cfg: BertCfg, h):
    t = F.linear(h, P["cls.predictions.transform.dense.weight"], P["cls.predictions.transform.dense.bias"])
    t = gelu_erf(t)
    t = layer_norm(t, P["cls.predictions.transform.LayerNorm.weight"],
                   P["cls.predictions.transform.LayerNorm.bias"], cfg.layer_norm_eps)
    return F.linear(t, P["bert.embeddings.word_embeddings.weight"], P["cls.predictions.bias"])


def generator_forward(P, cfg: BertCfg, ids, mask=None):
    """models.py:14-16 -> logits [B,S,V]."""
    return mlm_head(P, cfg, encoder(P, cfg, embeddings(P, cfg, ids), mask))


# ----------------------------------------------------------------------------- fill step
def fill_count(keep_percent: float, meaningful: int, remaining: int, fill_all: bool) -> int:
    """models.py:42-54 (Python double multiply, truncation)."""
    if fill_all:
        return remaining
    return min(max(1, int(keep_percent * meaningful)), remaining)


def select_topk_masked(conf_row: torch.Tensor, is_mask: torch.Tensor, k: int) -> torch.Tensor:
    """Positions the reference commits (models.py:57-63): top-k of confidence with non-mask
    positions zeroed, ties broken towards the lowest index (torch.topk leaves tie order
    implementation-defined; this oracle and the HIP kernel both define lowest-index-wins).
    Only masked positions are ever committed."""
    c = torch.where(is_mask, conf_row, torch.zeros_like(conf_row))
    idx = torch.arange(c.numel())
    order = sorted(idx[is_mask].tolist(), key=lambda j: (-float(c[j]), j))
    return torch.tensor(order[:k], dtype=torch.long)


def sample_from_logits(logits, temperature: float, noise):
    """models.py:21-28 with multinomial(p,1) restated as argmax(p / q), q ~ Exp(1) (`noise`,
    shape [B*S,V]); identical to torch.multinomial under the same generator state."""
    B, S, V = logits.shape
    probs = F.softmax(logits / temperature, dim=-1)
    pred = torch.argmax(probs.view(-1, V) / noise, dim=-1).view(B, S)
    conf = probs.gather(-1, pred.unsqueeze(-1)).squeeze(-1)
    return probs, pred, conf


def commit_step(ids, pred, conf, keep_percent, current_rate):
    """models.py:35-63, in place on `ids`; returns per-row committed counts."""
    B = ids.shape[0]
    fill_all = current_rate is not None and current_rate <= 0.1
    committed = []
    for i in range(B):
        is_mask = ids[i] == MASK
        rem = int(is_mask.sum())
        if rem == 0:
            committed.append(0)
            continue
        meaningful = int(((ids[i] != PAD) & (ids[i] != CLS) & (ids[i] != SEP)).sum())
        k = fill_count(keep_percent, meaningful, rem, fill_all)
        pos = select_topk_masked(conf[i], is_mask, k)
        ids[i, pos] = pred[i, pos]
        committed.append(int(pos.numel()))
    return committed


def generate_step(P, cfg, ids, mask, temperature, keep_percent, current_rate, noise):
    """One `Generator.generate` call (models.py:18-64), mutating and returning `ids`."""
    logits = generator_forward(P, cfg, ids, mask)
    _, pred, conf = sample_from_logits(logits, temperature, noise)
    commit_step(ids, pred, conf, keep_percent, current_rate)
    return ids


def draw_noise(n_rows: int, V: int, generator=None, dtype=torch.float32):
    """The Exp(1) draw torch.multinomial makes for an [n_rows,V] probability matrix."""
    return torch.empty(n_rows, V, dtype=dtype).exponential_(1, generator=generator)


# ----------------------------------------------------------------------------- critic
def classifier(C, x):
    x = F.relu(F.linear(x, C["0.weight"], C["0.bias"]))
    x = F.relu(F.linear(x, C["2.weight"], C["2.bias"]))
    x = F.relu(F.linear(x, C["4.weight"], C["4.bias"]))
    return F.linear(x, C["6.weight"], C["6.bias"])


def critic_forward(P, C, cfg: BertCfg, data, mask=None):
    """models.py:86-106: 2-D ids or 3-D embeddings -> [B,1]."""
    if data.dim() == 2:
        h = encoder(P, cfg, embeddings(P, cfg, data), mask)
    else:
        h = encoder(P, cfg, data, mask)
    return classifier(C, h[:, 0, :])


def critic_loss(real_scores, fake_scores, gp, lambda_gp=10):
    return -torch.mean(real_scores) + torch.mean(fake_scores) + lambda_gp * gp


def generator_loss(fake_scores):
    return -torch.mean(fake_scores)


def compute_gradient_penalty(P, C, cfg: BertCfg, real, fake, real_mask, fake_mask, alpha):
    """loss.py:10-96 with the CPU draw `alpha` ([B,1,1]) passed in.  Masks may be
    bool/long/float (the reference's `&` only accepts integer/bool; this accepts all by
    casting to bool first).  Differentiable to second order through torch autograd."""
    e_r = embeddings(P, cfg, real)
    e_f = embeddings(P, cfg, fake)
    x = alpha * e_r + (1 - alpha) * e_f
    if not x.requires_grad:
        x.requires_grad_(True)
    joint = (real_mask.bool() & fake_mask.bool()).to(x.dtype)
    h = encoder(P, cfg, x, joint)
    s = classifier(C, h[:, 0, :])
    g = torch.autograd.grad(s, x, torch.ones_like(s), create_graph=True, retain_graph=True)[0]
    g = g * joint.unsqueeze(-1)
    n = g.reshape(g.shape[0], -1).norm(2, dim=1)
    return ((n - 1) ** 2).mean()


def critic_step_grads(P, C, cfg, real, fake, real_mask, fake_mask, alpha, lambda_gp,
                      train_encoder=True):
    """One critic update's loss and gradients (fully_masked_train.py:315-325).  Returns
    (c_loss, gp, real_scores, fake_scores, grads dict over 'protbert.'+name / 'classifier.'+name;
    parameters the loss does not reach are absent, as `grad is None` in the reference)."""
    Pl = {k: v.detach().clone().requires_grad_(train_encoder) for k, v in P.items()}
    Cl = {k: v.detach().clone().requires_grad_(True) for k, v in C.items()}
    gp = compute_gradient_penalty(Pl, Cl, cfg, real, fake, real_mask, fake_mask, alpha)
    rs = critic_forward(Pl, Cl, cfg, real, real_mask)
    fs = critic_forward(Pl, Cl, cfg, fake, fake_mask)
    loss = critic_loss(rs, fs, gp, lambda_gp)
    loss.backward()
    grads = {}
    for k, v in Pl.items():
        if v.grad is not None:
            grads["protbert." + k] = v.grad
    for k, v in Cl.items():
        if v.grad is not None:
            grads["classifier." + k] = v.grad
    return loss.detach(), gp.detach(), rs.detach(), fs.detach(), grads


def generator_soft_step_grads(PG, PC, C, cfg, masked_ids, mask, temperature=1.0, critic_mask=None):
    """The differentiable generator update docs/HISTORY.md:169-179 prescribes (restated from its snippet; the
    reference's trainers do not contain it yet, so no run of the reference pins it): softmax(logits / T) of the
    generator over the masked template, averaged word embeddings of the critic, the critic's 3-D branch
    (models.py:94-102), generator_loss.  Returns (g_loss, grads over the generator's parameter names; parameters
    the loss does not reach are absent)."""
    Pl = {k: v.detach().clone().requires_grad_(True) for k, v in PG.items()}
    logits = generator_forward(Pl, cfg, masked_ids, mask)
    soft = F.softmax(logits / temperature, dim=-1) @ PC["bert.embeddings.word_embeddings.weight"]
    scores = critic_forward(PC, C, cfg, soft, mask if critic_mask is None else critic_mask)
    loss = generator_loss(scores)
    loss.backward()
    return loss.detach(), {k: v.grad for k, v in Pl.items() if v.grad is not None}



def adamw_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, wd=0.01):
    """torch.optim.AdamW single-tensor update (decoupled weight decay), in place; `step`
    is the 1-based step count after increment."""
    p.mul_(1 - lr * wd)
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)
    return p


# ----------------------------------------------------------------------------- generate.py
class ProtTokenizer:
    """The slice of BertTokenizer behaviour the path uses: per-residue encode, special ids,
    decode(skip_special_tokens=True) (generate.py:50-52,139,160)."""
    pad_token_id, unk_token_id, cls_token_id, sep_token_id, mask_token_id = PAD, UNK, CLS, SEP, MASK

    def __init__(self, vocab: Optional[List[str]] = None):
        self.vocab = list(vocab or VOCAB)
        self.tok2id = {t: i for i, t in enumerate(self.vocab)}

    def __len__(self):
        return len(self.vocab)

    def encode(self, text: str, add_special_tokens: bool = False) -> List[int]:
        ids = [self.tok2id.get(t, UNK) for t in text.split()]
        return [CLS] + ids + [SEP] if add_special_tokens else ids

    def decode(self, ids, skip_special_tokens: bool = True) -> str:
        toks = []
        for i in (ids.tolist() if hasattr(ids, "tolist") else ids):
            if skip_special_tokens and i in (PAD, UNK, CLS, SEP, MASK):
                continue
            toks.append(self.vocab[i])
        return " ".join(toks)


def build_fully_masked_batch(lengths: List[int]):
    """generate.py:92-103."""
    B, mx = len(lengths), max(lengths)
    ids = torch.full((B, mx + 2), PAD, dtype=torch.long)
    attn = torch.zeros_like(ids)
    for i, L in enumerate(lengths):
        attn[i, :L + 2] = 1
        ids[i, 0] = CLS
        ids[i, L + 1] = SEP
        if L > 0:
            ids[i, 1:L + 1] = MASK
    return ids, attn


def build_seeded_batch(seqs: List[str], tok: ProtTokenizer, keep_init: float, rng: random.Random):
    """generate.py:105-127."""
    toks = [tok.encode(" ".join(list(s))) for s in seqs]
    mx = max(len(t) for t in toks)
    B = len(seqs)
    ids = torch.full((B, mx + 2), PAD, dtype=torch.long)
    attn = torch.zeros_like(ids)
    for i, t in enumerate(toks):
        L = len(t)
        attn[i, :L + 2] = 1
        ids[i, 0] = CLS
        ids[i, L + 1] = SEP
        if L > 0:
            ids[i, 1:L + 1] = MASK
            k = max(1, int(round(keep_init * L)))
            for pos in rng.sample(range(1, L + 1), k):
                ids[i, pos] = t[pos - 1]
    return ids, attn


def generate_full_mode_sequences(P, cfg, tok, num_sequences, length_pool, keep_percent=0.10,
                                 temperature=1.0, batch_size=64, rng_seed=42, torch_gen=None,
                                 trace: Optional[list] = None):
    """generate.py:130-162.  `torch_gen` supplies the multinomial noise stream."""
    rng = random.Random(rng_seed)
    out, n_done = [], 0
    max_steps = math.ceil(1.0 / keep_percent)
    while n_done < num_sequences:
        nb = min(batch_size, num_sequences - n_done)
        lengths = [rng.choice(length_pool) for _ in range(nb)]
        ids, attn = build_fully_masked_batch(lengths)
        rate = 1.0
        for _ in range(max_steps):
            if int((ids == MASK).sum()) == 0:
                break
            noise = draw_noise(ids.numel(), cfg.vocab_size, torch_gen)
            ids = generate_step(P, cfg, ids, attn, temperature, keep_percent, rate, noise)
            if trace is not None:
                trace.append(ids.clone())
            rate = max(0.0, rate - keep_percent)
        for i in range(nb):
            out.append(tok.decode(ids[i], skip_special_tokens=True).replace(" ", ""))
        n_done += nb
    return out


def generate_seeded_mode_sequences(P, cfg, tok, num_sequences, seed_pool, keep_init=0.10,
                                   keep_percent=0.10, temperature=1.0, batch_size=64, rng_seed=123,
                                   torch_gen=None, trace: Optional[list] = None):
    """generate.py:164-201."""
    rng = random.Random(rng_seed)
    out, n_done = [], 0
    max_steps = max(1, math.ceil((1.0 - keep_init) / keep_percent)) + 1
    N = len(seed_pool)
    while n_done < num_sequences:
        nb = min(batch_size, num_sequences - n_done)
        seqs = [seed_pool[rng.randrange(N)] for _ in range(nb)]
        ids, attn = build_seeded_batch(seqs, tok, keep_init, rng)
        rate = 1.0 - keep_init
        for _ in range(max_steps):
            if int((ids == MASK).sum()) == 0:
                break
            noise = draw_noise(ids.numel(), cfg.vocab_size, torch_gen)
            ids = generate_step(P, cfg, ids, attn, temperature, keep_percent, rate, noise)
            if trace is not None:
                trace.append(ids.clone())
            rate = max(0.0, rate - keep_percent)
        for i in range(nb):
            out.append(tok.decode(ids[i], skip_special_tokens=True).replace(" ", ""))
        n_done += nb
    return out


# ----------------------------------------------------------------------------- the trainers' fake-batch loops
# (10p_train.py and fully_masked_train.py cannot be imported: they parse arguments, load models and data at
# module level.  The two loops below restate their fake-batch functions by reading; every arithmetic piece
# they call, generate_step above, is pinned by the reference's own outputs.  The compositions are not.)
def trainer_fakes_seeded(P, cfg, input_ids, random_mask, temperatures, noises, initial_mask_rate=0.9, iteration_rate=0.1,
                         unk_id=1):
    """10p_train.py:132-184.  random_mask: the boolean draw `torch.rand(shape) < rate` (:142); temperatures: one per
    fill iteration (:158); noises: one Exp(1) draw [B*S, V] per iteration.  Returns (ids, iterations run)."""
    meaningful = (input_ids != PAD) & (input_ids != unk_id) & (input_ids != CLS) & (input_ids != SEP)
    ids = input_ids.clone()
    ids[meaningful & random_mask] = MASK
    rate, it = initial_mask_rate, 0
    while rate > 0:
        attn = (ids != PAD).long()                       # :153-156
        with torch.no_grad():
            ids = generate_step(P, cfg, ids, attn, temperatures[it], iteration_rate, rate, noises[it])
        it += 1
        rate = max(0, rate - iteration_rate)
        if int((ids == MASK).sum()) == 0:                # :179-182
            break
    return ids, it


def trainer_fakes_blind(P, cfg, lengths, temperatures, noises, max_len, initial_mask_rate=0.9, iteration_rate=0.1):
    """fully_masked_train.py:195-258.  lengths: total lengths (sampled length + 2, :205); attention mask
    `ids != MASK` recomputed every iteration (:222); the result is padded or cut to max_len (:252-256)."""
    S = max(lengths)
    ids = torch.full((len(lengths), S), PAD, dtype=torch.long)
    for i, L in enumerate(lengths):
        ids[i, :L] = MASK
        ids[i, 0] = CLS
        ids[i, L - 1] = SEP
    rate, it = initial_mask_rate, 0
    while rate > 0:
        attn = (ids != MASK).long()
        with torch.no_grad():
            ids = generate_step(P, cfg, ids, attn, temperatures[it], iteration_rate, rate, noises[it])
        it += 1
        rate = max(0, rate - iteration_rate)
        if int((ids == MASK).sum()) == 0:
            break
    if S < max_len:
        ids = torch.nn.functional.pad(ids, (0, max_len - S), value=PAD)
    elif S > max_len:
        ids = ids[:, :max_len]
    return ids, it
