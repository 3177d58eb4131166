"""Generator and Critic with the reference's surface (models.py:5-106), on the HIP engine.

`Generator.generate` is one fill iteration (models.py:18-64) executed as four kernel
launches after the encoder forward (MLM head, sampler, per-row top-k commit) with no
host synchronisation; `Critic` is ProtBERT + the 1024-512-256-64-1 ReLU head
(models.py:76-84) evaluated by a fused head kernel."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from ._lib import check, lib
from .bert import FlatStore, ProtBertForMaskedLM


class RowPacking:
    """Which columns of a [B,S] id matrix belong to each row's own batch (`widths[b]` leading columns),
    as index tensors for the packed encoder pass: built once, reused by every fill iteration."""

    def __init__(self, widths, S: int, device):
        widths = [int(w) for w in widths]
        if not widths or min(widths) < 1 or max(widths) > S:
            raise ValueError("row widths must lie in 1..S")
        off = [0]
        for w in widths:
            off.append(off[-1] + w)
        self.widths, self.S, self.T = widths, S, off[-1]
        self.offsets = off
        rows = torch.repeat_interleave(torch.arange(len(widths)), torch.tensor(widths))
        pos = torch.arange(self.T) - torch.tensor(off[:-1])[rows]
        self.idx = (rows * S + pos).to(device)                       # flat index into [B*S]
        self.pos = pos.to(torch.int32).to(device)
        self.seq_off = torch.tensor(off, dtype=torch.int32).to(device)


class Generator(nn.Module):
    def __init__(self, protbert_model, cls_token_id=2, sep_token_id=3, mask_token_id=4, pad_token_id=0):
        super().__init__()
        self.protbert = protbert_model
        self.mask_token_id = mask_token_id
        self.pad_token_id = pad_token_id
        self.cls_token_id = cls_token_id
        self.sep_token_id = sep_token_id
        self.last_counts = None  # int32 [B,2] on device: (committed, masks left) of the last generate()
        self.fill_calls = 0      # bookkeeping for throughput reports: calls and their (B, S)
        self.fill_shapes = []
        self.fill_widths = []    # per call: None (every row at full width) or the rows' own widths of a packed pass

    def forward(self, input_ids, attention_mask=None):
        outputs = self.protbert(input_ids=input_ids, attention_mask=attention_mask)
        return outputs.logits

    @torch.no_grad()
    def generate(self, input_ids, attention_mask=None, temperature=1.0, keep_percent=0.1, current_rate=None,
                 noise=None, packing: "RowPacking | None" = None):
        """One iterative-fill step; mutates and returns `input_ids` (models.py:18-64).

        Every position of every row is sampled (pad and already-filled ones too), so the
        Exp(1) noise has shape [B*S, V] exactly like the draw `torch.multinomial` makes; it
        is taken from the device's global generator unless `noise` is given (tests inject
        the reference's draw).  Ties in the top-k are broken towards the lowest index.

        `packing` (not in the reference's signature; bf16): see generate_groups."""
        return self._fill_step(input_ids, attention_mask, [(0, input_ids.shape[0], float(temperature))], keep_percent,
                               current_rate, noise, packing)

    @torch.no_grad()
    def generate_groups(self, input_ids, attention_mask, groups, keep_percent=0.1, current_rate=None, noise=None,
                        packing: "RowPacking | None" = None):
        """The same fill step for several independent batches stacked along dim 0, one encoder
        pass for all of them.  `groups` = [(row_begin, row_end, temperature), ...] covers the
        rows; each group is sampled at its own temperature, exactly as separate generate()
        calls on the row slices would (rows never interact: attention is per sequence and the
        top-k commit is per row).

        `packing` (bf16 only): the batches have different padded widths and the columns beyond a
        row's own batch hold nothing (pad ids, never attended).  The encoder pass then runs on
        the kept columns only, laid back to back; tokens, confidences and commits are those of
        the full-width pass."""
        return self._fill_step(input_ids, attention_mask, [(int(a), int(b), float(t)) for a, b, t in groups],
                               keep_percent, current_rate, noise, packing)

    def _fill_step(self, input_ids, attention_mask, groups, keep_percent, current_rate, noise, packing=None):
        m: ProtBertForMaskedLM = self.protbert
        ids = m._check_ids(input_ids)
        inplace = ids.data_ptr() == input_ids.data_ptr()
        B, S = ids.shape
        V = m.config.vocab_size
        covered = sorted((a, b) for a, b, _ in groups)
        if covered[0][0] != 0 or covered[-1][1] != B or any(x[1] != y[0] for x, y in zip(covered, covered[1:])):
            raise ValueError("groups must partition the rows of input_ids")
        mask = m._mask01(attention_mask, B, S)
        if packing is not None and packing.T < B * S:
            return self._fill_step_packed(input_ids, ids, mask, groups, keep_percent, current_rate, noise, packing)
        hidden, _ = m._encode(ids, None, mask, want_stash=False)
        logits = m.lm_logits(hidden).view(B * S, V)
        if noise is None:
            noise = torch.empty(B * S, V, dtype=torch.float32, device=ids.device).exponential_(1.0)
        else:
            noise = noise.to(device=ids.device, dtype=torch.float32).contiguous().view(B * S, V)
        if len(groups) == 1:
            pred, conf = ops.sample_tokens(logits, noise, groups[0][2])
        else:
            pred = torch.empty(B * S, dtype=torch.long, device=ids.device)
            conf = torch.empty(B * S, dtype=torch.float32, device=ids.device)
            for a, b, t in groups:
                ops.sample_tokens(logits[a * S:b * S], noise[a * S:b * S], t, pred=pred[a * S:b * S], conf=conf[a * S:b * S])
        fill_all = current_rate is not None and current_rate <= 0.1
        counts = torch.empty(B, 2, dtype=torch.int32, device=ids.device)
        ops.commit_topk(ids, pred, conf, float(keep_percent), fill_all, self.mask_token_id, self.pad_token_id,
                        self.cls_token_id, self.sep_token_id, counts)
        self.last_counts = counts
        self.fill_calls += 1
        if len(self.fill_shapes) < 100000:
            self.fill_shapes.append((B, S))
            self.fill_widths.append(None)
        if not inplace:
            input_ids.copy_(ids)
        return input_ids


    def _fill_step_packed(self, input_ids, ids, mask, groups, keep_percent, current_rate, noise, pk):
        m: ProtBertForMaskedLM = self.protbert
        B, S = ids.shape
        V = m.config.vocab_size
        if pk.S != S or len(pk.widths) != B:
            raise ValueError("packing was built for another shape")
        if mask is None:
            mask = torch.ones(B, S, dtype=torch.float32, device=ids.device)
        T = pk.T
        hidden = m._encode_packed(ids.view(-1)[pk.idx], pk.pos, pk.seq_off, mask.view(-1)[pk.idx], B, S)
        logits = m.lm_logits(hidden.view(1, T, -1), ws_shape=(B, S)).view(T, V)
        if noise is None:
            # the full-width pass's draw (every position of every row, models.py:24-27), of which the kept columns'
            # part is used: the device generator ends in the same state with and without packing
            noise = torch.empty(B * S, V, dtype=torch.float32, device=ids.device).exponential_(1.0)[pk.idx]
        else:
            noise = noise.to(device=ids.device, dtype=torch.float32).contiguous().view(B * S, V)[pk.idx]
        pred_p = torch.empty(T, dtype=torch.long, device=ids.device)
        conf_p = torch.empty(T, dtype=torch.float32, device=ids.device)
        for a, b, t in groups:
            lo, hi = pk.offsets[a], pk.offsets[b]
            ops.sample_tokens(logits[lo:hi], noise[lo:hi], t, pred=pred_p[lo:hi], conf=conf_p[lo:hi])
        # columns outside a row's batch hold pad ids: the commit never looks at their prediction
        pred = torch.zeros(B * S, dtype=torch.long, device=ids.device).index_copy_(0, pk.idx, pred_p)
        conf = torch.zeros(B * S, dtype=torch.float32, device=ids.device).index_copy_(0, pk.idx, conf_p)
        fill_all = current_rate is not None and current_rate <= 0.1
        counts = torch.empty(B, 2, dtype=torch.int32, device=ids.device)
        ops.commit_topk(ids, pred, conf, float(keep_percent), fill_all, self.mask_token_id, self.pad_token_id,
                        self.cls_token_id, self.sep_token_id, counts)
        self.last_counts = counts
        self.fill_calls += 1
        if len(self.fill_shapes) < 100000:
            self.fill_shapes.append((B, S))
            self.fill_widths.append(pk.widths)
        if ids.data_ptr() != input_ids.data_ptr():
            input_ids.copy_(ids)
        return input_ids


class _HeadFn(torch.autograd.Function):
    """scores[b] = MLP(cls[b]) (models.py:104-105) and its backward, both in fused kernels."""

    @staticmethod
    def forward(ctx, cls_rows, head, *params):
        B, H = cls_rows.shape
        if cls_rows.stride(1) != 1:
            cls_rows = cls_rows.contiguous()
        dt = ops.dt_code(cls_rows.dtype)
        scores = torch.empty(B, 1, dtype=torch.float32, device=cls_rows.device)
        stash = torch.empty(B, lib.pg_head_stash_floats(H), dtype=torch.float32, device=cls_rows.device)
        check(lib.pg_head_forward(dt, cls_rows.data_ptr(), cls_rows.stride(0), head.store.master.data_ptr(), B, H,
                                  scores.data_ptr(), stash.data_ptr(), ops.stream()), "pg_head_forward")
        ctx.head, ctx.stash, ctx.H, ctx.dtype = head, stash, H, cls_rows.dtype
        ctx.train_head = any(ctx.needs_input_grad[2:])
        return scores

    @staticmethod
    def backward(ctx, gscores):
        head, B, H = ctx.head, ctx.stash.shape[0], ctx.H
        w = gscores.to(torch.float32).contiguous().view(B)
        ghead = None
        if ctx.train_head:
            head.store.begin_accumulate()
            ghead = head.store.grad
        dcls = torch.empty(B, H, dtype=ctx.dtype, device=w.device) if ctx.needs_input_grad[0] else None
        check(lib.pg_head_backward(ops.dt_code(ctx.dtype), head.store.master.data_ptr(), ctx.stash.data_ptr(), w.data_ptr(),
                                   B, H, ops.ptr(ghead), ops.ptr(dcls), H, ops.stream()), "pg_head_backward")
        return (dcls, None) + (None,) * 8


class CriticHead(nn.Sequential):
    """nn.Sequential(Linear, ReLU, Linear, ReLU, Linear, ReLU, Linear) with state-dict keys
    0/2/4/6.{weight,bias} (fully_masked_train.py:385-386); parameters are views into one flat
    fp32 buffer and the forward is a single fused kernel."""

    def __init__(self, hidden_size: int, device):
        dims = [hidden_size, hidden_size // 2, hidden_size // 4, hidden_size // 16, 1]
        layers = []
        for i in range(4):
            layers.append(nn.Linear(dims[i], dims[i + 1]))
            if i < 3:
                layers.append(nn.ReLU())
        super().__init__(*layers)
        self.hidden_size = hidden_size
        store = FlatStore(lib.pg_head_total(hidden_size), device)
        object.__setattr__(self, "store", store)
        for li, k in enumerate((0, 2, 4, 6)):
            lin = self[k]
            for is_bias, pname in ((0, "weight"), (1, "bias")):
                old = getattr(lin, pname)
                off = lib.pg_head_offset(hidden_size, li, is_bias)
                view = store.master[off:off + old.numel()].view(old.shape)
                with torch.no_grad():
                    view.copy_(old.detach())
                p = nn.Parameter(view)
                lin.register_parameter(pname, p)
                p._pg_store = (store, off, old.numel())
                store.views.append((p, off, old.numel()))

    def _apply(self, fn, *a, **k):
        probe = self.store.master[:1]
        moved = fn(probe)
        if moved.device != probe.device or moved.dtype != probe.dtype:
            raise RuntimeError("CriticHead parameters live in one flat fp32 HIP buffer and cannot be moved or cast")
        return self

    def forward(self, cls_output):
        params = [p for k in (0, 2, 4, 6) for p in (self[k].weight, self[k].bias)]
        return _HeadFn.apply(cls_output, self, *params)


class Critic(nn.Module):
    def __init__(self, protbert_model):
        super().__init__()
        self.protbert = protbert_model
        hidden_size = self.protbert.config.hidden_size
        self.classifier = CriticHead(hidden_size, self.protbert.device)

    def forward(self, input_data, attention_mask=None):
        if input_data.dim() == 2:  # token ids
            outputs = self.protbert(input_ids=input_data, attention_mask=attention_mask, output_hidden_states=True)
            last_hidden_state = outputs.hidden_states[-1]
        elif input_data.dim() == 3:  # embeddings straight into the encoder (models.py:94-102)
            extended_mask = self.protbert.bert.get_extended_attention_mask(attention_mask, input_data.shape[:2])
            last_hidden_state = self.protbert.bert.encoder(input_data, attention_mask=extended_mask).last_hidden_state
        else:
            raise ValueError("input_data must be [B,S] token ids or [B,S,H] embeddings")
        cls_output = last_hidden_state[:, 0, :]
        return self.classifier(cls_output)
