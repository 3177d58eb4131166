"""ProtBERT (`BertForMaskedLM`) on the MI355X engine, with the attribute surface the
reference reaches into (models.py:15-16,88-102; loss.py:22-23,55-66; generate.py:54-60;
fully_masked_train.py:378-389):

    model(input_ids=, attention_mask=, output_hidden_states=) -> .logits / .hidden_states[-1]
    model.config.hidden_size, model.parameters(), model.save_pretrained(dir)
    model.bert.embeddings(input_ids=), model.bert.get_extended_attention_mask(mask, shape),
    model.bert.encoder(x, attention_mask=ext).last_hidden_state

Parameters are `nn.Parameter` views into one flat fp32 buffer laid out by the C library
(pg_param_offset_*), named exactly like HuggingFace's state_dict so checkpoints are
interchangeable.  Gradients are accumulated by the kernels straight into a second flat
buffer whose views become the parameters' `.grad`.  All arithmetic runs in
libprotgen_hip.so; there is no PyTorch fallback."""
from __future__ import annotations

import ctypes as C
import json
import math
import weakref
import os
from dataclasses import asdict, dataclass
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import PG_BWD_PARAM_GRADS, PG_BWD_SAVE_ADJOINTS, PG_BWD_WAIT_LAYER_DONE, check, lib


# ----------------------------------------------------------------------------- config
@dataclass
class ProtBertConfig:
    vocab_size: int = 30
    hidden_size: int = 1024
    num_hidden_layers: int = 30
    num_attention_heads: int = 16
    intermediate_size: int = 4096
    max_position_embeddings: int = 40000
    type_vocab_size: int = 2
    layer_norm_eps: float = 1e-12
    initializer_range: float = 0.02
    hidden_act: str = "gelu"
    hidden_dropout_prob: float = 0.0
    attention_probs_dropout_prob: float = 0.0
    pad_token_id: int = 0
    tie_word_embeddings: bool = True
    model_type: str = "bert"

    def to_c(self) -> _lib.ModelCfg:
        return _lib.ModelCfg(self.vocab_size, self.hidden_size, self.num_hidden_layers, self.num_attention_heads,
                             self.intermediate_size, self.max_position_embeddings, self.type_vocab_size,
                             self.layer_norm_eps)

    def to_dict(self):
        d = asdict(self)
        d["architectures"] = ["BertForMaskedLM"]
        return d

    @classmethod
    def from_dict(cls, d):
        known = {f for f in cls.__dataclass_fields__}
        cfg = cls(**{k: v for k, v in d.items() if k in known})
        if d.get("hidden_act", "gelu") != "gelu":
            raise ValueError("only the exact (erf) GELU of ProtBERT is implemented")
        if cfg.hidden_size // cfg.num_attention_heads != 64:
            raise ValueError("the attention kernels are written for head_dim 64 (ProtBERT: 1024/16)")
        return cfg


# ----------------------------------------------------------------------------- shared workspace
_WORKSPACES: Dict[tuple, torch.Tensor] = {}


def workspace(device, nbytes: int) -> torch.Tensor:
    """One grow-only scratch buffer per (device, stream), shared by every model: the passes issued on one stream
    are ordered, so they can reuse it; a caller that runs passes on a second stream beside them gets a second one."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    w = _WORKSPACES.get(key)
    if w is None or w.numel() < nbytes:
        if w is not None:
            torch.cuda.synchronize(device)
        w = torch.empty(int(nbytes * 1.05) + 4096, dtype=torch.uint8, device=device)
        _WORKSPACES[key] = w
    return w


_STASH_POOLS: Dict[tuple, torch.Tensor] = {}


def pooled_stash(device, role: str, nbytes: int, reserve: int = 0) -> torch.Tensor:
    """`nbytes` of a grow-only buffer kept per (device, role), for the stashes of the hand-scheduled update passes
    (roles: the primal stash of a forward, the adjoint stash of a first-order backward, the tangent stash).  Their
    sizes follow the packed token count, which differs from one update to the next; taken from the caching allocator
    they leave a trail of multi-gigabyte blocks none of which fits the next request, until the allocator gives
    everything back to the driver in the middle of a step (seconds).  `reserve`: the size to allocate when the buffer
    has to grow (the largest the caller will ever ask for).  All users are ordered on the current stream; a buffer
    handed out under a role is dead by the time the role is asked for again."""
    key = (device, role)
    w = _STASH_POOLS.get(key)
    if w is None or w.numel() < nbytes:
        if w is not None:
            torch.cuda.synchronize(device)
            del _STASH_POOLS[key], w
        w = torch.empty(max(int(nbytes), int(reserve)) + 4096, dtype=torch.uint8, device=device)
        _STASH_POOLS[key] = w
    return w[:nbytes]


# ----------------------------------------------------------------------------- flat parameter store
class FlatStore:
    """fp32 master parameters in one buffer (+ bf16 shadow, + flat gradient buffer)."""

    def __init__(self, total: int, device):
        self.master = torch.zeros(total, dtype=torch.float32, device=device)
        self.grad: Optional[torch.Tensor] = None
        self.shadow: Optional[torch.Tensor] = None
        self._shadow_version = -1
        self.views = []  # (param, offset, numel) for every parameter a backward can reach
        self.adam_m = None
        self.adam_v = None
        self.adam_step = 0                   # largest step count of any parameter (reporting only)
        self.adam_steps: Dict[int, int] = {}  # parameter offset -> optimizer steps it has taken
        # offset -> leading elements of that view a backward may write (set by a data-parallel caller that
        # promises a bound, e.g. the rows of the position table within the padded width)
        self.live_elems: Dict[int, int] = {}

    def shadow_ptr(self):
        v = self.master._version
        if self.shadow is None:
            self.shadow = torch.empty(self.master.numel(), dtype=torch.bfloat16, device=self.master.device)
            self._shadow_version = -1
        if self._shadow_version != v:
            ops.cast_bf16(self.master, self.shadow)
            self._shadow_version = v
        return self.shadow.data_ptr()

    def mark_shadow_fresh(self):
        self._shadow_version = self.master._version

    def begin_accumulate(self):
        """Called at the start of every backward that adds into this store: (re)attach views of
        the flat gradient buffer as the parameters' .grad.  When the optimizer dropped every
        gradient (zero_grad(set_to_none=True)) the buffer is cleared with one memset."""
        if self.grad is None:
            self.grad = torch.zeros_like(self.master)
        live = [v for v in self.views if v[0].requires_grad]
        missing = [v for v in live if v[0].grad is None]
        if not missing:
            return
        if len(missing) == len(live):
            self.grad.zero_()
        for p, off, n in missing:
            g = self.grad[off:off + n].view(p.shape)
            if len(missing) != len(live):
                g.zero_()
            p.grad = g


def _params_struct(store: FlatStore, dtype: torch.dtype, with_grad: bool) -> _lib.Params:
    return _lib.Params(store.master.data_ptr(), store.grad.data_ptr() if (with_grad and store.grad is not None) else None,
                       store.shadow_ptr() if dtype == torch.bfloat16 else None)


# ----------------------------------------------------------------------------- outputs
class _HiddenStates:
    """`hidden_states` of a forward that keeps only what the path consumes: index -1 (the
    final LayerNorm output) and 0 (the embedding output)."""

    def __init__(self, first, last, n):
        self._first, self._last, self._n = first, last, n

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if i < 0:
            i += self._n
        if i == self._n - 1:
            return self._last
        if i == 0 and self._first is not None:
            return self._first
        raise IndexError("only hidden_states[0] and hidden_states[-1] are materialised by the fused encoder")


class MaskedLMOutput:
    def __init__(self, model, hidden, hidden_states):
        self._model, self._hidden = model, hidden
        self.hidden_states = hidden_states
        self._logits = None

    @property
    def logits(self):
        if self._logits is None:
            self._logits = self._model.lm_logits(self._hidden)
        return self._logits


class EncoderOutput:
    def __init__(self, last_hidden_state):
        self.last_hidden_state = last_hidden_state

    def __getitem__(self, i):
        return (self.last_hidden_state,)[i]


# ----------------------------------------------------------------------------- autograd glue
class _EmbedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, model, _sentinel):
        B, S = ids.shape
        dtype = model.compute_dtype
        out = torch.empty(B, S, model.config.hidden_size, dtype=dtype, device=ids.device)
        esum = torch.empty_like(out) if ctx.needs_input_grad[2] else None
        model._embed(ids, out, esum)
        ctx.model, ctx.ids, ctx.esum = model, ids, esum
        return out

    @staticmethod
    def backward(ctx, dout):
        model = ctx.model
        model.store.begin_accumulate()
        model._embed_backward(ctx.ids, ctx.esum, dout.contiguous())
        return None, None, None


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask, model, _sentinel):
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[3]
        hidden, stash = model._encode(None, x, mask, want_stash=need)
        ctx.model, ctx.stash, ctx.mask = model, stash, mask
        ctx.shape = x.shape
        return hidden

    @staticmethod
    def backward(ctx, dh):
        model = ctx.model
        grads = ctx.needs_input_grad[3]
        if grads:
            model.store.begin_accumulate()
        B, S, _ = ctx.shape
        dx = model._encode_backward(ctx.stash, dh.contiguous(), B, S, param_grads=grads)
        ctx.stash = None
        return (dx if ctx.needs_input_grad[0] else None), None, None, None


class _MlmHeadFn(torch.autograd.Function):
    """The MLM head with a backward (dense -> GELU -> LayerNorm -> tied decoder + bias): the head is evaluated again
    in the backward, so only `hidden` is kept."""

    @staticmethod
    def forward(ctx, hidden, model, _sentinel):
        ctx.model, ctx.hidden = model, hidden
        ctx.grads = ctx.needs_input_grad[2]
        return model._lm_logits_raw(hidden)

    @staticmethod
    def backward(ctx, dlogits):
        m, hidden = ctx.model, ctx.hidden
        B, S, H = hidden.shape
        T = B * S
        if ctx.grads:
            m._reach_head()
            m.store.begin_accumulate()
        dt = ops.dt_code(m.compute_dtype)
        n = lib.pg_mlm_backward_workspace_bytes(C.byref(m._ccfg), dt, T)
        ws = workspace(m.device, n)
        dhidden = torch.empty_like(hidden)
        pr = _params_struct(m.store, m.compute_dtype, ctx.grads)
        check(lib.pg_mlm_backward(C.byref(m._ccfg), C.byref(pr), dt, hidden.data_ptr(), T,
                                  dlogits.to(torch.float32).contiguous().data_ptr(), 1 if ctx.grads else 0,
                                  dhidden.data_ptr(), ws.data_ptr(), n, ops.stream()), "pg_mlm_backward")
        ctx.hidden = None
        return (dhidden if ctx.needs_input_grad[0] else None), None, None


# ----------------------------------------------------------------------------- module tree
class _Node(nn.Module):
    """Name-only container so parameters carry HuggingFace's dotted names."""

    def __getitem__(self, i):
        return getattr(self, str(i))

    def __len__(self):
        return len(self._modules)


class _Embeddings(_Node):
    def __init__(self, owner):
        super().__init__()
        object.__setattr__(self, "_owner", owner)

    def forward(self, input_ids=None, **_):
        m = self._owner
        return _EmbedFn.apply(m._check_ids(input_ids), m, self.word_embeddings.weight)


class _Encoder(_Node):
    def __init__(self, owner):
        super().__init__()
        object.__setattr__(self, "_owner", owner)

    def forward(self, hidden_states, attention_mask=None, **_):
        m = self._owner
        mask = m._mask01(attention_mask, hidden_states.shape[0], hidden_states.shape[1])
        x = hidden_states.to(m.compute_dtype).contiguous()
        sentinel = self.layer[0].attention.self.query.weight
        return EncoderOutput(_EncoderFn.apply(x, mask, m, sentinel))


class _BertModel(_Node):
    def __init__(self, owner):
        super().__init__()
        object.__setattr__(self, "_owner", owner)

    def get_extended_attention_mask(self, attention_mask, input_shape=None, dtype=None):
        """[B,S] 0/1 -> additive [B,1,1,S] with finfo.min at masked keys (the HF contract);
        the encoder recognises key j as visible iff the entry is 0."""
        if attention_mask is None:
            return None
        dtype = dtype or torch.float32
        m = attention_mask.to(dtype)
        return (1.0 - m)[:, None, None, :] * torch.finfo(dtype).min


_GLOBAL_NAMES = [
    ("bert.embeddings.word_embeddings.weight", _lib.P_WORD_EMB), ("bert.embeddings.position_embeddings.weight", _lib.P_POS_EMB),
    ("bert.embeddings.token_type_embeddings.weight", _lib.P_TYPE_EMB), ("bert.embeddings.LayerNorm.weight", _lib.P_EMB_LN_G),
    ("bert.embeddings.LayerNorm.bias", _lib.P_EMB_LN_B), ("cls.predictions.bias", _lib.P_MLM_BIAS),
    ("cls.predictions.transform.dense.weight", _lib.P_MLM_W), ("cls.predictions.transform.dense.bias", _lib.P_MLM_B),
    ("cls.predictions.transform.LayerNorm.weight", _lib.P_MLM_LN_G), ("cls.predictions.transform.LayerNorm.bias", _lib.P_MLM_LN_B),
]
_LAYER_NAMES = [
    ("attention.self.query.weight", _lib.L_WQ), ("attention.self.key.weight", _lib.L_WK),
    ("attention.self.value.weight", _lib.L_WV), ("attention.self.query.bias", _lib.L_BQ),
    ("attention.self.key.bias", _lib.L_BK), ("attention.self.value.bias", _lib.L_BV),
    ("attention.output.dense.weight", _lib.L_WO), ("attention.output.dense.bias", _lib.L_BO),
    ("attention.output.LayerNorm.weight", _lib.L_LN1_G), ("attention.output.LayerNorm.bias", _lib.L_LN1_B),
    ("intermediate.dense.weight", _lib.L_W1), ("intermediate.dense.bias", _lib.L_B1),
    ("output.dense.weight", _lib.L_W2), ("output.dense.bias", _lib.L_B2),
    ("output.LayerNorm.weight", _lib.L_LN2_G), ("output.LayerNorm.bias", _lib.L_LN2_B),
]


def _shape_of(cfg: ProtBertConfig, name: str):
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


class ProtBertForMaskedLM(nn.Module):
    def __init__(self, config: ProtBertConfig, device="cuda", compute_dtype=torch.bfloat16):
        super().__init__()
        if config.hidden_size // config.num_attention_heads != 64:
            raise ValueError("head_dim must be 64")
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("protgen-gan_amd runs on an MI355X (HIP) device only; there is no CPU path")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.config = config
        self.compute_dtype = compute_dtype
        self._ccfg = config.to_c()
        total = lib.pg_param_total(C.byref(self._ccfg))
        self.store = FlatStore(total, device)
        self.bert = _BertModel(self)
        self.bert.embeddings = _Embeddings(self)
        self.bert.encoder = _Encoder(self)
        self.cls = _Node()
        self._offsets: Dict[str, int] = {}
        for name, which in _GLOBAL_NAMES:
            self._add_param(name, lib.pg_param_offset_global(C.byref(self._ccfg), which),
                            reachable=not name.startswith("cls."))
        for l in range(config.num_hidden_layers):
            for suffix, which in _LAYER_NAMES:
                self._add_param(f"bert.encoder.layer.{l}.{suffix}",
                                lib.pg_param_offset_layer(C.byref(self._ccfg), l, which), reachable=True)
        self._layer_events = None
        self._layer_event_arr = None
        self.reset_parameters(seed=None)

    # -- per-layer "gradients issued" events for a data-parallel caller
    def enable_layer_events(self):
        """One event per encoder layer; the backward passes record event l on the current stream once
        layer l's parameter-gradient launches are issued (pg_encoder_backward / pg_encoder_backward2)."""
        if self._layer_events is None:
            L = self.config.num_hidden_layers
            with torch.cuda.device(self.device):
                evs = [torch.cuda.Event() for _ in range(L)]
                for e in evs:
                    e.record()  # the handle exists only after the first record
            self._layer_events = evs
            self._layer_event_arr = (C.c_void_p * L)(*[e.cuda_event for e in evs])
        return self._layer_events

    def layer_param_range(self, l: int):
        """[lo, hi) of layer l's parameters in the flat store."""
        pre = f"bert.encoder.layer.{l}."
        spans = [(off, off + p.numel()) for (name, off), p in
                 ((kv, self.get_parameter(kv[0])) for kv in self._offsets.items() if kv[0].startswith(pre))]
        return min(a for a, _ in spans), max(b for _, b in spans)

    def limit_position_rows(self, rows: int):
        """Promise that no backward sees more than `rows` positions: gradient reductions then skip the
        rest of the position table (rows nobody writes are zero on every rank)."""
        off = lib.pg_param_offset_global(C.byref(self._ccfg), _lib.P_POS_EMB)
        self.store.live_elems[off] = min(rows, self.config.max_position_embeddings) * self.config.hidden_size

    # -- construction helpers
    def _add_param(self, name, offset, reachable):
        shape = _shape_of(self.config, name)
        n = math.prod(shape)
        p = nn.Parameter(self.store.master[offset:offset + n].view(shape))
        node = self
        parts = name.split(".")
        for part in parts[:-1]:
            if not hasattr(node, part):
                node.add_module(part, _Node())
            node = getattr(node, part)
        node.register_parameter(parts[-1], p)
        p._pg_store = (self.store, offset, n)
        self._offsets[name] = offset
        if reachable:
            self.store.views.append((p, offset, n))

    def _apply(self, fn, *a, **k):
        probe = self.store.master[:1]
        moved = fn(probe)
        if moved.device != probe.device or moved.dtype != probe.dtype:
            raise RuntimeError("ProtBertForMaskedLM keeps its parameters in one flat HIP buffer; construct it on the "
                               "target device (device=...) and choose the compute dtype with compute_dtype=...")
        return self

    @property
    def device(self):
        return self.store.master.device

    def get_input_embeddings(self):
        return self.bert.embeddings.word_embeddings

    def reset_parameters(self, seed: Optional[int] = 0, bias_std: float = 0.0):
        """HF `_init_weights`: N(0, initializer_range) matrices, LayerNorm 1/0, biases 0, drawn
        name by name from one CPU generator (same stream as the CPU oracle's init_params)."""
        g = torch.Generator().manual_seed(seed) if seed is not None else None
        with torch.no_grad():
            for name, p in self.named_parameters():
                if "LayerNorm.weight" in name:
                    t = torch.ones(p.shape)
                    if bias_std and g is not None:
                        t = t + bias_std * torch.randn(p.shape, generator=g)
                elif name.endswith("bias"):
                    t = torch.zeros(p.shape)
                    if bias_std and g is not None:
                        t = bias_std * torch.randn(p.shape, generator=g)
                elif g is None:
                    continue
                else:
                    t = torch.randn(p.shape, generator=g) * self.config.initializer_range
                p.copy_(t)

    def init_random(self, seed: int = 0):
        """Synthetic ProtBERT-shaped weights drawn on the device: N(0, initializer_range)
        everywhere, then LayerNorm weights 1 and every bias 0 (HF `_init_weights`)."""
        g = torch.Generator(device=self.device).manual_seed(seed)
        with torch.no_grad():
            self.store.master.normal_(0.0, self.config.initializer_range, generator=g)
            for name, p in self.named_parameters():
                if "LayerNorm.weight" in name:
                    p.fill_(1.0)
                elif name.endswith("bias"):
                    p.zero_()
        return self

    @classmethod
    def from_params(cls, config, params: Dict[str, torch.Tensor], device="cuda", compute_dtype=torch.bfloat16):
        m = cls(config, device=device, compute_dtype=compute_dtype)
        m.load_state_dict(params, strict=False)
        return m

    def load_state_dict(self, state_dict, strict=True, assign=False):
        sd = {k: v for k, v in state_dict.items()
              if not k.startswith("cls.predictions.decoder.") and not k.endswith("position_ids")}
        return super().load_state_dict(sd, strict=strict)

    # -- HF directory format (generate.py:54-60, fully_masked_train.py:378-389)
    def save_pretrained(self, path: str):
        from safetensors.torch import save_file
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(self.config.to_dict(), f, indent=2)
        sd = {k: v.detach().cpu().contiguous().clone() for k, v in self.state_dict().items()}
        save_file(sd, os.path.join(path, "model.safetensors"), metadata={"format": "pt"})

    @classmethod
    def from_pretrained(cls, path: str, device="cuda", compute_dtype=torch.bfloat16, local_files_only=True):
        if not os.path.isdir(path):
            raise FileNotFoundError(f"Model dir not found: {path}")
        cpath = os.path.join(path, "config.json")
        if not os.path.exists(cpath):
            raise FileNotFoundError(f"Missing config.json in {path} (did save_pretrained run?)")
        with open(cpath) as f:
            config = ProtBertConfig.from_dict(json.load(f))
        st = os.path.join(path, "model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu", weights_only=True)
        m = cls(config, device=device, compute_dtype=compute_dtype)
        missing, unexpected = m.load_state_dict(sd, strict=False)
        if missing:
            raise KeyError(f"checkpoint {path} lacks tensors: {missing[:5]}...")
        return m.eval()

    # -- input normalisation
    def _guard_device(self):
        """Kernels are launched on the current device's current stream: the model's device has to be that device."""
        if self.device.type == "cuda" and torch.cuda.current_device() != self.device.index:
            raise RuntimeError(f"model on {self.device} but the current device is cuda:{torch.cuda.current_device()}: "
                               f"call torch.cuda.set_device({self.device.index}) (or use `with torch.cuda.device(...)`)")

    def _check_ids(self, ids, values=True):
        """Shape, type and device of an id matrix; with `values`, also that every id is a row of the embedding table
        (torch's embedding raises on such an id; the kernels here would read or add into a neighbouring table).
        The value check reads two numbers back from the device; the tensor OBJECT that passed is remembered (weak
        reference + version counter), so a fill loop that hands the same tensor back every iteration pays once, while
        a new batch that the allocator happens to place at the old address is checked again.  The fill kernels write
        into such a tensor without moving its version counter; what they write are sampled vocabulary indices."""
        if ids.dim() != 2 or ids.dtype != torch.int64:
            raise TypeError("input_ids must be a [B,S] int64 tensor")
        if ids.device != self.device:
            raise ValueError(f"input_ids on {ids.device}, model on {self.device}")
        self._guard_device()
        if ids.shape[1] > self.config.max_position_embeddings:
            raise ValueError("sequence longer than max_position_embeddings")
        given = ids
        ids = ids.contiguous()
        if values and ids.numel():
            seen = getattr(self, "_ids_checked", None)
            if seen is None or seen[0]() is not given or seen[1] != given._version:
                lo, hi = (int(v) for v in torch.aminmax(ids))
                if lo < 0 or hi >= self.config.vocab_size:
                    raise IndexError(f"token id out of range: ids span [{lo}, {hi}], vocabulary has {self.config.vocab_size}")
                self._ids_checked = (weakref.ref(given), given._version)
        return ids

    def _mask01(self, mask, B, S):
        if mask is None:
            return None
        if mask.dim() == 4:  # extended additive mask [B,1,1,S]
            mask = (mask[:, 0, 0, :] == 0)
        if mask.shape != (B, S):
            raise ValueError(f"attention_mask shape {tuple(mask.shape)} != {(B, S)}")
        return mask.to(device=self.device, dtype=torch.float32).contiguous()

    def _gp(self, which):
        off = lib.pg_param_offset_global(C.byref(self._ccfg), which)
        return self.store.master.data_ptr() + 4 * off

    def _gg(self, which):
        off = lib.pg_param_offset_global(C.byref(self._ccfg), which)
        return self.store.grad.data_ptr() + 4 * off

    def encoder_trainable(self) -> bool:
        return self.bert.encoder.layer[0].attention.self.query.weight.requires_grad

    # -- engine passes
    def _embed(self, ids, out, esum):
        B, S = ids.shape
        check(lib.pg_embed_ln2(ops.dt_code(self.compute_dtype), ids.data_ptr(), self._gp(_lib.P_WORD_EMB),
                               self._gp(_lib.P_POS_EMB), self._gp(_lib.P_TYPE_EMB), self._gp(_lib.P_EMB_LN_G),
                               self._gp(_lib.P_EMB_LN_B), self.config.layer_norm_eps, out.data_ptr(), ops.ptr(esum),
                               B, S, self.config.hidden_size, ops.stream()), "pg_embed_ln2")

    def _embed_backward(self, ids, esum, dout):
        """LayerNorm backward of BertEmbeddings, then scatter into the three tables."""
        B, S = ids.shape
        H = self.config.hidden_size
        bound = self.store.live_elems.get(lib.pg_param_offset_global(C.byref(self._ccfg), _lib.P_POS_EMB))
        if bound is not None and S * H > bound:
            raise ValueError(f"{S} positions exceed the bound of {bound // H} given to limit_position_rows")
        dt = ops.dt_code(self.compute_dtype)
        dsum = torch.empty_like(esum)
        part = torch.empty(3 * H * lib.pg_ln_partial_rows(B * S), dtype=torch.float32, device=ids.device)
        check(lib.pg_layernorm_bwd(dt, esum.data_ptr(), dout.data_ptr(), self._gp(_lib.P_EMB_LN_G),
                                   self.config.layer_norm_eps, dsum.data_ptr(), self._gg(_lib.P_EMB_LN_G),
                                   self._gg(_lib.P_EMB_LN_B), None, part.data_ptr(), B * S, H, ops.stream()), "pg_layernorm_bwd")
        check(lib.pg_embed_scatter(dt, dsum.data_ptr(), ids.data_ptr(), self._gg(_lib.P_WORD_EMB), self._gg(_lib.P_POS_EMB),
                                   self._gg(_lib.P_TYPE_EMB), B, S, H, ops.stream()), "pg_embed_scatter")

    def _ws(self, B, S, T=None):
        """Workspace for a pass over B x S, or (T given) over T ragged token rows of B sequences of at most S."""
        dt = ops.dt_code(self.compute_dtype)
        n = (lib.pg_workspace_bytes(C.byref(self._ccfg), dt, B, S) if T is None
             else lib.pg_workspace_bytes_ragged(C.byref(self._ccfg), dt, B, S, T))
        if n < 0:
            raise ValueError("unsupported shape")
        return workspace(self.device, n), n

    def _new_stash(self, B, S, T=None, role=None):
        """Stash bytes for a pass over B x S (or T ragged token rows).  `role`: take them from the per-role pool
        (pooled_stash) instead of the allocator; only for callers that are done with the stash before they ask for
        the role again."""
        dt = ops.dt_code(self.compute_dtype)
        n = (lib.pg_stash_bytes(C.byref(self._ccfg), dt, B, S) if T is None
             else lib.pg_stash_bytes_ragged(C.byref(self._ccfg), dt, B, S, T))
        if role is None:
            return torch.empty(n, dtype=torch.uint8, device=self.device)
        # (packed token counts wander by a few per cent from update to update: a fifth of headroom, at most the
        # rectangle's size, lets the pool settle after the first few updates)
        most = n if T is None else min(lib.pg_stash_bytes_ragged(C.byref(self._ccfg), dt, B, S, B * S), n + n // 5)
        return pooled_stash(self.device, role, n, reserve=most)

    # -- ragged passes (rows of different lengths laid back to back; see ragged.py)
    def _embed_packed(self, ids_p, pos_p, out, esum):
        """Embeddings + LayerNorm of T packed tokens (int64 ids, int32 positions) into out [T,H] (and the
        pre-LayerNorm sum into esum, when given)."""
        T = ids_p.numel()
        check(lib.pg_embed_ln2_packed(ops.dt_code(self.compute_dtype), ids_p.data_ptr(), pos_p.data_ptr(),
                                      self._gp(_lib.P_WORD_EMB), self._gp(_lib.P_POS_EMB), self._gp(_lib.P_TYPE_EMB),
                                      self._gp(_lib.P_EMB_LN_G), self._gp(_lib.P_EMB_LN_B), self.config.layer_norm_eps,
                                      out.data_ptr(), ops.ptr(esum), T, self.config.hidden_size, ops.stream()),
              "pg_embed_ln2_packed")

    def _embed_backward_packed(self, ids_p, pos_p, esum, dout):
        T = ids_p.numel()
        H = self.config.hidden_size
        dt = ops.dt_code(self.compute_dtype)
        dsum = torch.empty_like(esum)
        part = torch.empty(3 * H * lib.pg_ln_partial_rows(T), dtype=torch.float32, device=ids_p.device)
        check(lib.pg_layernorm_bwd(dt, esum.data_ptr(), dout.data_ptr(), self._gp(_lib.P_EMB_LN_G),
                                   self.config.layer_norm_eps, dsum.data_ptr(), self._gg(_lib.P_EMB_LN_G),
                                   self._gg(_lib.P_EMB_LN_B), None, part.data_ptr(), T, H, ops.stream()), "pg_layernorm_bwd")
        check(lib.pg_embed_scatter_packed(dt, dsum.data_ptr(), ids_p.data_ptr(), pos_p.data_ptr(), self._gg(_lib.P_WORD_EMB),
                                          self._gg(_lib.P_POS_EMB), self._gg(_lib.P_TYPE_EMB), T, H, ops.stream()),
              "pg_embed_scatter_packed")

    def _encode_ragged(self, rg, ids_p=None, pos_p=None, x=None, mask_p=None, want_stash=False, esum=None, stash_role=None):
        """Forward over the ragged batch `rg` (a ragged.Ragged): packed ids + positions, or x [T,H]; mask_p fp32 [T]
        or None.  Returns (hidden [T,H], stash or None).  `stash_role`: see _new_stash."""
        dt = ops.dt_code(self.compute_dtype)
        ws, n = self._ws(rg.B, rg.S, rg.T)
        stash = self._new_stash(rg.B, rg.S, rg.T, role=stash_role) if want_stash else None
        hidden = torch.empty(rg.T, self.config.hidden_size, dtype=self.compute_dtype, device=self.device)
        pr = _params_struct(self.store, self.compute_dtype, False)
        d = rg.desc()
        check(lib.pg_encoder_forward_ragged(C.byref(self._ccfg), C.byref(pr), dt, rg.B, C.byref(d), ops.ptr(ids_p), ops.ptr(pos_p),
                                            ops.ptr(x), ops.ptr(mask_p), ws.data_ptr(), n, ops.ptr(stash), hidden.data_ptr(),
                                            ops.ptr(esum), ops.stream()), "pg_encoder_forward_ragged")
        return hidden, stash

    def _encode(self, ids, x, mask, want_stash=False, esum=None, stash_role=None):
        """Embeddings (if ids) + all encoder layers.  Returns (hidden [B,S,H], stash or None)."""
        B, S = (ids.shape if ids is not None else x.shape[:2])
        dt = ops.dt_code(self.compute_dtype)
        ws, n = self._ws(B, S)
        stash = self._new_stash(B, S, role=stash_role) if want_stash else None
        hidden = torch.empty(B, S, self.config.hidden_size, dtype=self.compute_dtype, device=self.device)
        pr = _params_struct(self.store, self.compute_dtype, False)
        check(lib.pg_encoder_forward(C.byref(self._ccfg), C.byref(pr), dt, B, S, ops.ptr(ids), ops.ptr(x), ops.ptr(mask),
                                     ws.data_ptr(), n, ops.ptr(stash), hidden.data_ptr(), ops.ptr(esum), ops.stream()),
              "pg_encoder_forward")
        return hidden, stash

    def _encode_packed(self, ids_p, pos_ids, seq_off, mask_p, nseq, max_len):
        """Inference forward over `nseq` sequences of different lengths laid back to back in the T rows of
        `ids_p` (int64), `pos_ids` (int32), `mask_p` (fp32 0/1); `seq_off` int32 [nseq+1].  -> hidden [T,H]."""
        if self.compute_dtype != torch.bfloat16:
            raise ValueError("the packed forward exists for bf16 inference only")
        T = ids_p.numel()
        dt = ops.dt_code(self.compute_dtype)
        ws, n = self._ws(nseq, max_len)
        hidden = torch.empty(T, self.config.hidden_size, dtype=self.compute_dtype, device=self.device)
        pr = _params_struct(self.store, self.compute_dtype, False)
        check(lib.pg_encoder_forward_packed(C.byref(self._ccfg), C.byref(pr), dt, nseq, max_len, seq_off.data_ptr(), T,
                                            ids_p.data_ptr(), pos_ids.data_ptr(), ops.ptr(mask_p), ws.data_ptr(), n,
                                            hidden.data_ptr(), ops.stream()), "pg_encoder_forward_packed")
        return hidden

    # `rgd` below: a _lib.RaggedDesc (ragged.Ragged.desc(...)) when the pass is ragged, with `T` its token rows.
    def _encode_backward(self, stash, dh, B, S, param_grads, adj_stash=None, stash_rows=0, row0=0, grad_row0=0, rgd=None, T=None):
        dt = ops.dt_code(self.compute_dtype)
        ws, n = self._ws(B, S, T if rgd is not None else None)
        flags = (PG_BWD_PARAM_GRADS if param_grads else 0) | (PG_BWD_SAVE_ADJOINTS if adj_stash is not None else 0)
        dx = torch.empty_like(dh)
        pr = _params_struct(self.store, self.compute_dtype, param_grads)
        check(lib.pg_encoder_backward(C.byref(self._ccfg), C.byref(pr), dt, B, S, stash.data_ptr(), dh.data_ptr(), None,
                                      flags, ops.ptr(adj_stash), dx.data_ptr(), ws.data_ptr(), n, ops.stream(),
                                      self._layer_event_arr if param_grads else None, stash_rows, row0, grad_row0,
                                      C.byref(rgd) if rgd is not None else None),
              "pg_encoder_backward")
        return dx

    def _encode_jvp(self, stash, xdot, B, S, tan_stash, stash_rows=0, row0=0, rgd=None, T=None):
        dt = ops.dt_code(self.compute_dtype)
        ws, n = self._ws(B, S, T if rgd is not None else None)
        hdot = torch.empty_like(xdot)
        pr = _params_struct(self.store, self.compute_dtype, False)
        check(lib.pg_encoder_jvp(C.byref(self._ccfg), C.byref(pr), dt, B, S, stash.data_ptr(), xdot.data_ptr(),
                                 tan_stash.data_ptr(), hdot.data_ptr(), ws.data_ptr(), n, ops.stream(), stash_rows, row0,
                                 C.byref(rgd) if rgd is not None else None),
              "pg_encoder_jvp")
        return hdot

    def _encode_backward2(self, stash, adj, tan, B, S, want_xbar=True, stash_rows=0, row0=0, adj_rows=0, rgd=None, T=None,
                          wait_layer_events=False):
        """`wait_layer_events`: every layer waits for its "gradients issued" event as a first-order backward on another
        stream recorded it, before it adds into the same gradient buffers."""
        dt = ops.dt_code(self.compute_dtype)
        ws, n = self._ws(B, S, T if rgd is not None else None)
        H = self.config.hidden_size
        shape = (T, H) if rgd is not None else (B, S, H)
        xbar = torch.empty(*shape, dtype=self.compute_dtype, device=self.device) if want_xbar else None
        pr = _params_struct(self.store, self.compute_dtype, True)
        flags = PG_BWD_PARAM_GRADS
        if wait_layer_events:
            self.enable_layer_events()
            flags |= PG_BWD_WAIT_LAYER_DONE
        check(lib.pg_encoder_backward2(C.byref(self._ccfg), C.byref(pr), dt, B, S, stash.data_ptr(), adj.data_ptr(),
                                       tan.data_ptr(), flags, ops.ptr(xbar), ws.data_ptr(), n, ops.stream(),
                                       self._layer_event_arr, stash_rows, row0, adj_rows,
                                       C.byref(rgd) if rgd is not None else None),
              "pg_encoder_backward2")
        return xbar

    def lm_logits(self, hidden, ws_shape=None):
        """MLM head (dense -> GELU -> LayerNorm -> tied decoder + bias) -> fp32 [B,S,V].
        `ws_shape`: the (B, S) whose workspace is large enough (packed rows: the unpacked shape).
        Under autograd (a hidden state that carries a graph, or head parameters that take gradients) the
        logits carry the head's backward."""
        if torch.is_grad_enabled() and (hidden.requires_grad or self._head_sentinel().requires_grad):
            return _MlmHeadFn.apply(hidden.contiguous(), self, self._head_sentinel())
        return self._lm_logits_raw(hidden, ws_shape)

    def _head_sentinel(self):
        return self.cls.predictions.transform.dense.weight

    def _reach_head(self):
        """The MLM head's parameters join the ones a backward can reach the first time a head backward runs.
        (Until then they have no .grad, like the head of a model whose logits never feed a loss: the optimizer
        leaves such parameters alone, weight decay included.)"""
        if not getattr(self, "_head_reached", False):
            for name, _ in _GLOBAL_NAMES:
                if name.startswith("cls."):
                    p = self.get_parameter(name)
                    _, off, n = p._pg_store
                    self.store.views.append((p, off, n))
            self._head_reached = True

    def _lm_logits_raw(self, hidden, ws_shape=None):
        B, S, H = hidden.shape
        T = B * S
        ws, n = self._ws(*(ws_shape or (B, S)))
        logits = torch.empty(B, S, self.config.vocab_size, dtype=torch.float32, device=self.device)
        pr = _params_struct(self.store, self.compute_dtype, False)
        check(lib.pg_mlm_logits(C.byref(self._ccfg), C.byref(pr), ops.dt_code(self.compute_dtype), hidden.data_ptr(), T,
                                ws.data_ptr(), n, logits.data_ptr(), ops.stream()), "pg_mlm_logits")
        return logits

    # -- the BertForMaskedLM call (models.py:15,20,88-92)
    def forward(self, input_ids=None, attention_mask=None, output_hidden_states=False, inputs_embeds=None, **_):
        if input_ids is not None:
            ids = self._check_ids(input_ids)
            B, S = ids.shape
            mask = self._mask01(attention_mask, B, S)
            if torch.is_grad_enabled() and self.encoder_trainable():
                emb = self.bert.embeddings(input_ids=ids)
                hidden = _EncoderFn.apply(emb, mask, self, self.bert.encoder.layer[0].attention.self.query.weight)
                first = emb
            else:
                hidden, _ = self._encode(ids, None, mask, want_stash=False)
                first = None
        else:
            x = inputs_embeds.to(self.compute_dtype).contiguous()
            mask = self._mask01(attention_mask, x.shape[0], x.shape[1])
            hidden = _EncoderFn.apply(x, mask, self, self.bert.encoder.layer[0].attention.self.query.weight)
            first = x
        hs = _HiddenStates(first, hidden, self.config.num_hidden_layers + 1) if output_hidden_states else None
        return MaskedLMOutput(self, hidden, hs)
