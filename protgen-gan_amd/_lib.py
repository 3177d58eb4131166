"""ctypes binding of libprotgen_hip.so (the C ABI declared in include/protgen_hip.h).

The library is the product: there is no CPU or PyTorch fallback.  If it is missing or
fails to load, importing this module raises."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# (PG_LIB: another build of the same library, for A/B timing of kernel variants inside one GPU session)
LIB_PATH = os.environ.get("PG_LIB") or os.path.join(HERE, "libprotgen_hip.so")

PG_F32, PG_BF16 = 0, 1
PG_ACT_NONE, PG_ACT_GELU, PG_ACT_MUL_DGELU = 0, 1, 2
PG_BWD_PARAM_GRADS, PG_BWD_SAVE_ADJOINTS, PG_BWD_WAIT_LAYER_DONE = 1, 2, 4

# enum order of include/protgen_hip.h
P_WORD_EMB, P_POS_EMB, P_TYPE_EMB, P_EMB_LN_G, P_EMB_LN_B, P_MLM_BIAS, P_MLM_W, P_MLM_B, P_MLM_LN_G, P_MLM_LN_B = range(10)
(L_WQ, L_WK, L_WV, L_BQ, L_BK, L_BV, L_WO, L_BO, L_LN1_G, L_LN1_B, L_W1, L_B1, L_W2, L_B2, L_LN2_G,
 L_LN2_B) = range(16)


class ModelCfg(C.Structure):
    _fields_ = [("vocab_size", C.c_int), ("hidden", C.c_int), ("layers", C.c_int), ("heads", C.c_int),
                ("inter", C.c_int), ("max_pos", C.c_int), ("type_vocab", C.c_int), ("ln_eps", C.c_float)]


class Params(C.Structure):
    _fields_ = [("master", C.c_void_p), ("grad", C.c_void_p), ("shadow", C.c_void_p)]


class GemmDesc(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
                ("dtype", C.c_int), ("c_dtype", C.c_int), ("a_kmajor", C.c_int), ("b_kmajor", C.c_int),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("lda", C.c_int64), ("ldb", C.c_int64), ("ldc", C.c_int64),
                ("batch", C.c_int), ("batch_inner", C.c_int),
                ("sAo", C.c_int64), ("sAi", C.c_int64), ("sBo", C.c_int64), ("sBi", C.c_int64),
                ("sCo", C.c_int64), ("sCi", C.c_int64),
                ("alpha", C.c_float), ("beta", C.c_float),
                ("bias", C.c_void_p), ("residual", C.c_void_p), ("ldr", C.c_int64),
                ("aux", C.c_void_p), ("ldaux", C.c_int64), ("aux2", C.c_void_p), ("act", C.c_int),
                ("A2", C.c_void_p), ("B2", C.c_void_p), ("K2", C.c_int),
                ("seq_off", C.c_void_p), ("seq_tok0", C.c_int), ("seq_ops", C.c_int)]


class RaggedDesc(C.Structure):
    _fields_ = [("max_len", C.c_int), ("seq_off", C.c_void_p), ("seq_off_host", C.c_void_p), ("stash_T", C.c_int),
                ("adj_T", C.c_int)]


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python protgen-gan_amd/build.py` "
        "(hipcc --offload-arch=gfx950).  There is no fallback path.")
# torch first: its wheel carries a HIP runtime of its own, and the streams and device pointers this library is handed
# come from that copy.  Opened before torch, the library would bind to the system's runtime instead and every launch
# on a torch stream would fail.
import torch  # noqa: E402,F401

lib = C.CDLL(LIB_PATH)

vp, i32, i64, f32, f64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double
cfgp = C.POINTER(ModelCfg)
parp = C.POINTER(Params)
ragp = C.POINTER(RaggedDesc)

SIGNATURES = {
    "pg_strerror": (C.c_char_p, [i32]),
    "pg_abi_version": (i32, []),
    "pg_param_total": (i64, [cfgp]),
    "pg_param_offset_global": (i64, [cfgp, i32]),
    "pg_param_offset_layer": (i64, [cfgp, i32, i32]),
    "pg_param_size_global": (i64, [cfgp, i32]),
    "pg_param_size_layer": (i64, [cfgp, i32]),
    "pg_head_total": (i64, [i32]),
    "pg_head_offset": (i64, [i32, i32, i32]),
    "pg_gemm": (i32, [C.POINTER(GemmDesc), vp]),
    "pg_gemm_pair": (i32, [C.POINTER(GemmDesc), C.POINTER(GemmDesc), vp]),
    "pg_gemm_group": (i32, [C.POINTER(GemmDesc), i32, vp]),
    "pg_tune": (i32, [i32, i32]),
    "pg_profile_begin": (i32, [i32]),
    "pg_profile_end": (i32, [C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "pg_embed_ln": (i32, [i32, vp, vp, vp, vp, vp, vp, f32, vp, i32, i32, i32, vp]),
    "pg_embed_ln2": (i32, [i32, vp, vp, vp, vp, vp, vp, f32, vp, vp, i32, i32, i32, vp]),
    "pg_layernorm": (i32, [i32, vp, vp, vp, f32, vp, i32, i32, vp]),
    "pg_ln_partial_rows": (i32, [i32]),
    "pg_layernorm_bwd": (i32, [i32, vp, vp, vp, f32, vp, vp, vp, vp, vp, i32, i32, vp]),
    "pg_layernorm_bwd_static": (i32, [i32, vp, vp, vp, f32, vp, vp, vp, vp, i32, i32, vp]),
    "pg_layernorm_jvp": (i32, [i32, vp, vp, vp, f32, vp, i32, i32, vp]),
    "pg_layernorm_bwd2": (i32, [i32, vp, vp, vp, vp, vp, f32, vp, vp, vp, vp, vp, i32, i32, vp]),
    "pg_gelu_bwd2": (i32, [i32, vp, vp, vp, vp, vp, i64, vp]),
    "pg_colsum_acc": (i32, [i32, vp, i64, i32, i32, vp, vp]),
    "pg_attention_fwd": (i32, [i32, vp, vp, vp, i32, i32, i32, i64, i64, vp]),
    "pg_attention_fwd_p": (i32, [i32, vp, vp, vp, vp, i32, i32, i32, i64, i64, vp]),
    "pg_attention_bwd_dq": (i32, [i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i64, i64, i64, vp]),
    "pg_attention_jvp": (i32, [i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i64, i64, i64, vp]),
    "pg_attention_bwd2": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i64, i64, vp]),
    "pg_score_ld": (i32, [i32]),
    "pg_softmax_fwd": (i32, [i32, vp, vp, vp, i32, i32, i32, vp]),
    "pg_softmax_bwd": (i32, [i32, vp, vp, vp, i32, i32, vp]),
    "pg_softmax_jvp": (i32, [i32, vp, vp, vp, vp, i32, i32, vp]),
    "pg_softmax_bwd2": (i32, [i32, vp, vp, vp, vp, vp, vp, i32, i32, vp]),
    "pg_mlm_decode": (i32, [i32, vp, vp, vp, vp, i32, i32, i32, vp]),
    "pg_mlm_decode_bwd": (i32, [i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "pg_sample_tokens": (i32, [vp, vp, f32, i32, i32, vp, vp, vp]),
    "pg_commit_topk": (i32, [vp, vp, vp, i32, i32, f64, i32, i32, i32, i32, i32, vp, vp]),
    "pg_head_stash_floats": (i64, [i32]),
    "pg_head_forward": (i32, [i32, vp, i64, vp, i32, i32, vp, vp, vp]),
    "pg_head_backward": (i32, [i32, vp, vp, vp, i32, i32, vp, vp, i64, vp]),
    "pg_head_jvp_grad": (i32, [i32, vp, vp, vp, i64, i32, i32, f32, vp, vp, vp, vp]),
    "pg_gp_norms": (i32, [i32, vp, vp, i32, i32, i32, vp, vp, vp]),
    "pg_gp_direction": (i32, [i32, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp]),
    "pg_interpolate": (i32, [i32, vp, vp, vp, vp, i32, i64, vp]),
    "pg_interpolate_bwd": (i32, [i32, vp, vp, vp, vp, i32, i64, vp]),
    "pg_embed_scatter": (i32, [i32, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "pg_adamw": (i32, [vp, vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, f32, vp]),
    "pg_cast_bf16": (i32, [vp, vp, i64, vp]),
    "pg_cast_f32": (i32, [vp, vp, i64, vp]),
    "pg_scale_f32": (i32, [vp, f32, i64, vp]),
    "pg_workspace_bytes": (i64, [cfgp, i32, i32, i32]),
    "pg_stash_bytes": (i64, [cfgp, i32, i32, i32]),
    "pg_encoder_forward": (i32, [cfgp, parp, i32, i32, i32, vp, vp, vp, vp, i64, vp, vp, vp, vp]),
    "pg_round_bf16_f32": (i32, [vp, i64, vp]),
    "pg_attention_key_lists": (i32, [vp, vp, i32, i32, vp, vp, vp]),
    "pg_attention_fwd_keys": (i32, [i32, vp, vp, vp, vp, i32, i32, i32, i64, i64, vp, vp, vp]),
    "pg_mlm_logits": (i32, [cfgp, parp, i32, vp, i32, vp, i64, vp, vp]),
    "pg_mlm_backward_workspace_bytes": (i64, [cfgp, i32, i32]),
    "pg_mlm_backward": (i32, [cfgp, parp, i32, vp, i32, vp, i32, vp, vp, i64, vp]),
    "pg_encoder_backward": (i32, [cfgp, parp, i32, i32, i32, vp, vp, vp, i32, vp, vp, vp, i64, vp, vp, i32, i32, i32, ragp]),
    "pg_encoder_forward_ragged": (i32, [cfgp, parp, i32, i32, ragp, vp, vp, vp, vp, vp, i64, vp, vp, vp, vp]),
    "pg_workspace_bytes_ragged": (i64, [cfgp, i32, i32, i32, i32]),
    "pg_stash_bytes_ragged": (i64, [cfgp, i32, i32, i32, i32]),
    "pg_attention_fwd_p_ragged": (i32, [i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i64, i64, vp]),
    "pg_attention_bwd_dq_ragged": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i64, i64, i64, vp]),
    "pg_attention_bwd": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i64, i64, i64, vp]),
    "pg_attention_jvp_ragged": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i64, i64, i64, vp]),
    "pg_attention_bwd2_ragged": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i64, i64, vp]),
    "pg_embed_ln2_packed": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, f32, vp, vp, i32, i32, vp]),
    "pg_embed_scatter_packed": (i32, [i32, vp, vp, vp, vp, vp, vp, i32, i32, vp]),
    "pg_gp_norms_ragged": (i32, [i32, vp, vp, vp, i32, i32, i32, vp, vp, vp]),
    "pg_gp_direction_ragged": (i32, [i32, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp]),
    "pg_interpolate_ragged": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]),
    "pg_interpolate_bwd_ragged": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "pg_encoder_forward_packed": (i32, [cfgp, parp, i32, i32, i32, vp, i32, vp, vp, vp, vp, i64, vp, vp]),
    "pg_embed_ln_packed": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, f32, vp, i32, i32, vp]),
    "pg_attention_fwd_packed": (i32, [i32, vp, vp, vp, vp, i32, i32, i32, i64, i64, vp]),
    "pg_encoder_jvp": (i32, [cfgp, parp, i32, i32, i32, vp, vp, vp, vp, vp, i64, vp, i32, i32, ragp]),
    "pg_encoder_backward2": (i32, [cfgp, parp, i32, i32, i32, vp, vp, vp, i32, vp, vp, i64, vp, vp, i32, i32, i32, ragp]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here means the .so is stale: rebuild
    _fn.restype = _res
    _fn.argtypes = _args


class PgError(RuntimeError):
    pass


def check(rc: int, what: str = ""):
    if rc != 0:
        raise PgError(f"{what or 'protgen_hip'}: {lib.pg_strerror(rc).decode()} (code {rc})")
