"""Thin tensor-level wrappers over the C ABI (one Python function per kernel entry point).

PyTorch is plumbing here: it owns device memory and the current HIP stream; every
arithmetic step is a call into libprotgen_hip.so."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import PG_ACT_GELU, PG_ACT_MUL_DGELU, PG_ACT_NONE, PG_BF16, PG_F32, check, lib

_DT = {torch.float32: PG_F32, torch.bfloat16: PG_BF16}


def dt_code(dtype: torch.dtype) -> int:
    try:
        return _DT[dtype]
    except KeyError:
        raise TypeError(f"unsupported activation dtype {dtype}; use torch.float32 or torch.bfloat16")


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise ValueError("protgen-gan_amd kernels need tensors on a HIP device (there is no CPU path)")


def _plain_desc(A, B, C_out, a_kmajor, b_kmajor, M, N, K, alpha, beta):
    d = _lib.GemmDesc()
    d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), C_out.data_ptr()
    d.dtype, d.c_dtype = dt_code(A.dtype), dt_code(C_out.dtype)
    d.a_kmajor, d.b_kmajor = int(a_kmajor), int(b_kmajor)
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc = A.stride(0), B.stride(0), C_out.stride(0)
    d.batch, d.batch_inner = 1, 1
    d.alpha, d.beta = alpha, beta
    return d


def gemm_pair(A0, B0, C0, A1, B1, C1, *, a_kmajor=True, b_kmajor=True, alpha=1.0, beta=0.0):
    """Two 2-D GEMMs of the same layout in one launch (pg_gemm_pair); outputs must be given."""
    _need_cuda(A0, B0)
    descs = []
    for A, B, Cm in ((A0, B0, C0), (A1, B1, C1)):
        M, K = (A.shape if a_kmajor else (A.shape[1], A.shape[0]))
        N, K2 = (B.shape if b_kmajor else (B.shape[1], B.shape[0]))
        assert K == K2 and A.stride(1) == 1 and B.stride(1) == 1 and Cm.shape == (M, N)
        descs.append(_plain_desc(A, B, Cm, a_kmajor, b_kmajor, M, N, K, alpha, beta))
    check(lib.pg_gemm_pair(C.byref(descs[0]), C.byref(descs[1]), stream()), "pg_gemm_pair")
    return C0, C1


def gemm_group(problems, *, a_kmajor=True, b_kmajor=True, alpha=1.0, beta=0.0):
    """Up to sixteen 2-D GEMMs of one layout in one launch (pg_gemm_group).  `problems` is a list of
    (A, B, C) or (A, B, C, A2, B2) tuples; outputs must be given."""
    assert 1 <= len(problems) <= 16
    arr = (_lib.GemmDesc * len(problems))()
    for i, pr in enumerate(problems):
        A, B, Cm = pr[:3]
        _need_cuda(A, B)
        M, K = (A.shape if a_kmajor else (A.shape[1], A.shape[0]))
        N, K2 = (B.shape if b_kmajor else (B.shape[1], B.shape[0]))
        assert K == K2 and A.stride(1) == 1 and B.stride(1) == 1 and Cm.shape == (M, N)
        d = _plain_desc(A, B, Cm, a_kmajor, b_kmajor, M, N, K, alpha, beta)
        if len(pr) == 5:
            A2, B2 = pr[3], pr[4]
            assert A2.stride(0) == A.stride(0) and B2.stride(0) == B.stride(0)
            d.A2, d.B2 = A2.data_ptr(), B2.data_ptr()
            d.K2 = A2.shape[1] if a_kmajor else A2.shape[0]
        arr[i] = d
    check(lib.pg_gemm_group(arr, len(problems), stream()), "pg_gemm_group")
    return [pr[2] for pr in problems]


def gemm(A, B, C_out=None, *, a_kmajor=True, b_kmajor=True, out_dtype=None, alpha=1.0, beta=0.0, bias=None,
         residual=None, act=PG_ACT_NONE, aux=None, aux2=None, A2=None, B2=None):
    """2-D GEMM.  A is [M,K] (a_kmajor) or [K,M]; B is [N,K] (b_kmajor) or [K,N].  (A2, B2): an
    optional second reduction segment with the row strides of (A, B): C = epilogue(alpha (A.B + A2.B2))."""
    _need_cuda(A, B)
    M, K = (A.shape if a_kmajor else (A.shape[1], A.shape[0]))
    N, K2 = (B.shape if b_kmajor else (B.shape[1], B.shape[0]))
    assert K == K2, (A.shape, B.shape)
    assert A.stride(1) == 1 and B.stride(1) == 1
    out_dtype = out_dtype or A.dtype
    if C_out is None:
        C_out = torch.empty(M, N, dtype=out_dtype, device=A.device)
    d = _plain_desc(A, B, C_out, a_kmajor, b_kmajor, M, N, K, alpha, beta)
    d.bias = ptr(bias)
    if residual is not None:
        d.residual, d.ldr = residual.data_ptr(), residual.stride(0)
    if aux is not None:
        d.aux, d.ldaux = aux.data_ptr(), aux.stride(0)
    if aux2 is not None:
        d.aux2 = aux2.data_ptr()
    d.act = act
    if A2 is not None:
        assert A2.stride(0) == A.stride(0) and B2.stride(0) == B.stride(0) and A2.stride(1) == 1 and B2.stride(1) == 1
        K2 = A2.shape[1] if a_kmajor else A2.shape[0]
        assert K2 == (B2.shape[1] if b_kmajor else B2.shape[0])
        d.A2, d.B2, d.K2 = A2.data_ptr(), B2.data_ptr(), K2
    check(lib.pg_gemm(C.byref(d), stream()), "pg_gemm")
    return C_out


def layernorm(x, gamma, beta, eps):
    y = torch.empty_like(x)
    rows, H = x.numel() // x.shape[-1], x.shape[-1]
    check(lib.pg_layernorm(dt_code(x.dtype), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), eps, y.data_ptr(), rows,
                           H, stream()), "pg_layernorm")
    return y


def _ln_partial(rows, H, device):
    return torch.empty(3 * H * lib.pg_ln_partial_rows(rows), dtype=torch.float32, device=device)


def layernorm_bwd(x, dy, gamma, eps, dgamma=None, dbeta=None, dxsum=None):
    dx = torch.empty_like(x)
    rows, H = x.numel() // x.shape[-1], x.shape[-1]
    part = _ln_partial(rows, H, x.device) if any(t is not None for t in (dgamma, dbeta, dxsum)) else None
    check(lib.pg_layernorm_bwd(dt_code(x.dtype), x.data_ptr(), dy.data_ptr(), gamma.data_ptr(), eps, dx.data_ptr(),
                               ptr(dgamma), ptr(dbeta), ptr(dxsum), ptr(part), rows, H, stream()), "pg_layernorm_bwd")
    return dx


def layernorm_jvp(x, xdot, gamma, eps):
    y = torch.empty_like(x)
    rows, H = x.numel() // x.shape[-1], x.shape[-1]
    check(lib.pg_layernorm_jvp(dt_code(x.dtype), x.data_ptr(), xdot.data_ptr(), gamma.data_ptr(), eps, y.data_ptr(),
                               rows, H, stream()), "pg_layernorm_jvp")
    return y


def layernorm_bwd2(x, xdot, gy, ybar, gamma, eps, dgamma=None, dbeta=None, dxsum=None):
    xbar = torch.empty_like(x)
    rows, H = x.numel() // x.shape[-1], x.shape[-1]
    part = _ln_partial(rows, H, x.device) if any(t is not None for t in (dgamma, dbeta, dxsum)) else None
    check(lib.pg_layernorm_bwd2(dt_code(x.dtype), x.data_ptr(), xdot.data_ptr(), gy.data_ptr(), ptr(ybar),
                                gamma.data_ptr(), eps, xbar.data_ptr(), ptr(dgamma), ptr(dbeta), ptr(dxsum), ptr(part),
                                rows, H, stream()), "pg_layernorm_bwd2")
    return xbar


def gelu_bwd2(u, udot, ggu, gubar):
    out = torch.empty_like(u)
    check(lib.pg_gelu_bwd2(dt_code(u.dtype), u.data_ptr(), ptr(udot), ptr(ggu), gubar.data_ptr(), out.data_ptr(),
                           u.numel(), stream()), "pg_gelu_bwd2")
    return out


def colsum_acc(dY, db):
    rows, N = dY.shape
    check(lib.pg_colsum_acc(dt_code(dY.dtype), dY.data_ptr(), dY.stride(0), rows, N, db.data_ptr(), stream()),
          "pg_colsum_acc")
    return db


def score_ld(S: int) -> int:
    return lib.pg_score_ld(S)


def softmax_fwd(scores, mask, B, heads, S, dtype):
    P = torch.empty(B * heads, S, score_ld(S), dtype=dtype, device=scores.device)
    check(lib.pg_softmax_fwd(dt_code(dtype), scores.data_ptr(), ptr(mask), P.data_ptr(), B, heads, S, stream()),
          "pg_softmax_fwd")
    return P


def softmax_bwd(dP, P, S):
    dS = torch.empty_like(P)
    check(lib.pg_softmax_bwd(dt_code(P.dtype), dP.data_ptr(), P.data_ptr(), dS.data_ptr(), P.shape[0], S, stream()),
          "pg_softmax_bwd")
    return dS


def softmax_jvp(Sdot, P, S):
    Sd, Pd = torch.empty_like(P), torch.empty_like(P)
    check(lib.pg_softmax_jvp(dt_code(P.dtype), Sdot.data_ptr(), P.data_ptr(), Sd.data_ptr(), Pd.data_ptr(), P.shape[0],
                             S, stream()), "pg_softmax_jvp")
    return Sd, Pd


def softmax_bwd2(P, Sdot, gP, X, S):
    Sbar, Pd = torch.empty_like(P), torch.empty_like(P)
    check(lib.pg_softmax_bwd2(dt_code(P.dtype), P.data_ptr(), Sdot.data_ptr(), gP.data_ptr(), X.data_ptr(),
                              Sbar.data_ptr(), Pd.data_ptr(), P.shape[0], S, stream()), "pg_softmax_bwd2")
    return Sbar, Pd


def mlm_decode(t, word, bias):
    T, H = t.shape
    V = word.shape[0]
    logits = torch.empty(T, V, dtype=torch.float32, device=t.device)
    check(lib.pg_mlm_decode(dt_code(t.dtype), t.data_ptr(), word.data_ptr(), bias.data_ptr(), logits.data_ptr(), T, H,
                            V, stream()), "pg_mlm_decode")
    return logits


def sample_tokens(logits, noise, temperature: float, pred=None, conf=None):
    T, V = logits.shape
    assert logits.is_contiguous() and noise.is_contiguous() and noise.shape == logits.shape
    if pred is None:
        pred = torch.empty(T, dtype=torch.int64, device=logits.device)
    if conf is None:
        conf = torch.empty(T, dtype=torch.float32, device=logits.device)
    assert pred.is_contiguous() and conf.is_contiguous() and pred.numel() == T and conf.numel() == T
    assert pred.dtype == torch.int64 and conf.dtype == torch.float32
    check(lib.pg_sample_tokens(logits.data_ptr(), noise.data_ptr(), float(temperature), T, V, pred.data_ptr(),
                               conf.data_ptr(), stream()), "pg_sample_tokens")
    return pred, conf


def commit_topk(ids, pred, conf, keep_percent: float, fill_all: bool, mask_id=4, pad_id=0, cls_id=2, sep_id=3,
                counts=None):
    B, S = ids.shape
    assert ids.is_contiguous() and ids.dtype == torch.int64
    check(lib.pg_commit_topk(ids.data_ptr(), pred.data_ptr(), conf.data_ptr(), B, S, float(keep_percent), int(fill_all),
                             mask_id, pad_id, cls_id, sep_id, ptr(counts), stream()), "pg_commit_topk")
    return ids


def adamw(p, g, m, v, shadow, lr, beta1, beta2, eps, wd, step, grad_scale=1.0):
    check(lib.pg_adamw(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), ptr(shadow), p.numel(), lr, beta1, beta2,
                       eps, wd, step, grad_scale, stream()), "pg_adamw")


def cast_bf16(src, dst):
    check(lib.pg_cast_bf16(src.data_ptr(), dst.data_ptr(), src.numel(), stream()), "pg_cast_bf16")


def attention_fwd(qkv, mask, B, S, heads):
    """qkv: [B*S, 3*heads*64] bf16 packed (q | k | v); returns ctx [B*S, heads*64]."""
    H = heads * 64
    ctx = torch.empty(B * S, H, dtype=qkv.dtype, device=qkv.device)
    check(lib.pg_attention_fwd(dt_code(qkv.dtype), qkv.data_ptr(), ptr(mask), ctx.data_ptr(), B, S, heads, qkv.stride(0),
                               ctx.stride(0), stream()), "pg_attention_fwd")
    return ctx


def attention_fwd_packed(qkv, mask, seq_off, max_len, heads):
    """As attention_fwd over sequences of different lengths laid back to back: qkv [T, 3*heads*64],
    mask [T] fp32 or None, seq_off int32 [nseq+1] on the device."""
    ctx = torch.empty(qkv.shape[0], heads * 64, dtype=qkv.dtype, device=qkv.device)
    check(lib.pg_attention_fwd_packed(dt_code(qkv.dtype), qkv.data_ptr(), ptr(mask), ctx.data_ptr(), seq_off.data_ptr(),
                                      seq_off.numel() - 1, max_len, heads, qkv.stride(0), ctx.stride(0), stream()),
          "pg_attention_fwd_packed")
    return ctx


def attention_fwd_visible(qkv, mask, heads, B=None, S=None, seq_off=None, max_len=None):
    """attention_fwd / attention_fwd_packed walking only the keys whose mask is 1 (pg_attention_key_lists +
    pg_attention_fwd_keys).  Rectangle: pass B, S; packed rows: pass seq_off, max_len.  Returns (ctx, key_cnt)."""
    packed = seq_off is not None
    n = seq_off.numel() - 1 if packed else B
    width = max_len if packed else S
    T = qkv.shape[0]
    idx = torch.empty(T, dtype=torch.int32, device=qkv.device)
    cnt = torch.empty(n, dtype=torch.int32, device=qkv.device)
    so = seq_off.data_ptr() if packed else None
    check(lib.pg_attention_key_lists(mask.data_ptr(), so, n, width, idx.data_ptr(), cnt.data_ptr(), stream()),
          "pg_attention_key_lists")
    ctx = torch.empty(T, heads * 64, dtype=qkv.dtype, device=qkv.device)
    check(lib.pg_attention_fwd_keys(dt_code(qkv.dtype), qkv.data_ptr(), mask.data_ptr(), ctx.data_ptr(), so, n, width, heads,
                                    qkv.stride(0), ctx.stride(0), idx.data_ptr(), cnt.data_ptr(), stream()),
          "pg_attention_fwd_keys")
    return ctx, cnt


def attention_bwd_dq(qkv, P, ctx, dctx, B, S, heads):
    """First half of the attention backward over stored P: returns (dS [B*heads,S,ld], dq [B*S,H])."""
    H = heads * 64
    dS = torch.zeros(B * heads, S, score_ld(S), dtype=qkv.dtype, device=qkv.device)
    dq = torch.empty(B * S, H, dtype=qkv.dtype, device=qkv.device)
    check(lib.pg_attention_bwd_dq(dt_code(qkv.dtype), qkv.data_ptr(), P.data_ptr(), ctx.data_ptr(), dctx.data_ptr(),
                                  dS.data_ptr(), dq.data_ptr(), B, S, heads, qkv.stride(0), ctx.stride(0), dq.stride(0),
                                  stream()), "pg_attention_bwd_dq")
    return dS, dq


def attention_bwd(qkv, P, ctx, dctx, heads, B=None, S=None, seq_off=None, max_len=None, keep_dS=True):
    """The whole attention backward over stored P (pg_attention_bwd: dq with the row constants, then the key-stationary
    dk / dv kernel).  Rectangle: pass B, S; ragged rows: pass seq_off, max_len.  Returns (dqkv [T, 3H], dS or None)."""
    n, width = (B, S) if seq_off is None else (seq_off.numel() - 1, max_len)
    ld = score_ld(width)
    rows = width if seq_off is None else ld
    dS = torch.zeros(n * heads, rows, ld, dtype=qkv.dtype, device=qkv.device) if keep_dS else None
    dqkv = torch.zeros_like(qkv)
    delta = torch.empty(n * heads, width, dtype=torch.float32, device=qkv.device)
    check(lib.pg_attention_bwd(dt_code(qkv.dtype), qkv.data_ptr(), P.data_ptr(), ctx.data_ptr(), dctx.data_ptr(), ptr(dS),
                               dqkv.data_ptr(), delta.data_ptr(), ptr(seq_off), n, width, 0, heads, qkv.stride(0),
                               ctx.stride(0), dqkv.stride(0), stream()), "pg_attention_bwd")
    return dqkv, dS


def attention_jvp(qkv, tqkv, P, ctx, B, S, heads):
    """Tangent pass over stored P: returns (dctx [B*S,H], rho [B*heads,S] fp32)."""
    H = heads * 64
    ctxdot = torch.empty(B * S, H, dtype=qkv.dtype, device=qkv.device)
    rho = torch.empty(B * heads, S, dtype=torch.float32, device=qkv.device)
    check(lib.pg_attention_jvp(dt_code(qkv.dtype), qkv.data_ptr(), tqkv.data_ptr(), P.data_ptr(), ctx.data_ptr(),
                               ctxdot.data_ptr(), rho.data_ptr(), B, S, heads, qkv.stride(0), ctx.stride(0),
                               ctxdot.stride(0), stream()), "pg_attention_jvp")
    return ctxdot, rho


def attention_bwd2(qkv, tqkv, P, ctx, ctxdot, gctx, ctxbar, rho, B, S, heads):
    """Score side of the second-order reverse pass: returns (Sbar, dP), both [B*heads,S,ld]."""
    Sbar = torch.zeros(B * heads, S, score_ld(S), dtype=qkv.dtype, device=qkv.device)
    Pdot = torch.zeros_like(Sbar)
    check(lib.pg_attention_bwd2(dt_code(qkv.dtype), qkv.data_ptr(), tqkv.data_ptr(), P.data_ptr(), ctx.data_ptr(),
                                ctxdot.data_ptr(), gctx.data_ptr(), ctxbar.data_ptr(), rho.data_ptr(), Sbar.data_ptr(),
                                Pdot.data_ptr(), B, S, heads, qkv.stride(0), ctx.stride(0), stream()), "pg_attention_bwd2")
    return Sbar, Pdot


def attention_fwd_p(qkv, mask, B, S, heads):
    """As attention_fwd, also returning P [B*heads, S, score_ld(S)]."""
    H = heads * 64
    ctx = torch.empty(B * S, H, dtype=qkv.dtype, device=qkv.device)
    P = torch.empty(B * heads, S, score_ld(S), dtype=qkv.dtype, device=qkv.device)
    check(lib.pg_attention_fwd_p(dt_code(qkv.dtype), qkv.data_ptr(), ptr(mask), P.data_ptr(), ctx.data_ptr(), B, S, heads,
                                 qkv.stride(0), ctx.stride(0), stream()), "pg_attention_fwd_p")
    return ctx, P
