// Whole-pass orchestration of the ProtBERT encoder on one HIP stream: forward,
// first-order backward, gradient-penalty tangent pass and second-order reverse pass.
// Host code only; every launch goes through the kernels of gemm.hip / rowwise.hip /
// misc.hip.  No allocation, no host synchronisation: the caller owns parameters,
// workspace and stash buffers (sizes from pg_workspace_bytes / pg_stash_bytes).
//
// Stash layout (activation dtype), shared by the primal, adjoint and tangent stashes so
// that the four passes address the same slot names:
//   xs[l]  l = 0..L   input of layer l (xs[L] = output of the last layer)          [T][H]
//   per layer: qkv [T][3H] | ss [Z][S][ldS] | ctx [T][H] | a [T][H] (pre-LN1) | h [T][H]
//              u [T][I] (pre-GELU) | gu [T][I] | f [T][H] (pre-LN2)
#include <string.h>
#include "pg_common.h"
#include "protgen_hip.h"

extern int pg_internal_fused_gp;  // pg_tune key 6 (defined beside pg_tune)

namespace {

struct Dims {
  int B, S, H, I, nh, L, V, ldS;
  int Srows;  // rows of one score slot: S, or (ragged passes) ldS so that whole 64-row tiles stay inside it
  long T, Z;  // token rows (B*S, or the packed count of a ragged pass), score slots
  int es;  // activation element size
};

// Ragged pass: sequence b = token rows [off_h[b] - tok0, off_h[b+1] - tok0) of this pass's buffers.
struct Rag {
  const int32_t* off = nullptr;    // device
  const int32_t* off_h = nullptr;  // host copy
  int tok0 = 0, B = 0;
  bool on() const { return off != nullptr; }
  long row(int b) const { return (long)off_h[b] - tok0; }
  int len(int b) const { return off_h[b + 1] - off_h[b]; }
};

inline long align_up(long v, long a) { return (v + a - 1) / a * a; }

struct StashLayout {
  long xs, layer0, per_layer;
  long o_qkv, o_ss, o_ctx, o_a, o_h, o_u, o_gu, o_f;
  long total_bytes;
};

StashLayout stash_layout(const Dims& d) {
  StashLayout s;
  const long th = align_up(d.T * d.H * d.es, 256);
  const long ti = align_up(d.T * d.I * d.es, 256);
  const long t3 = align_up(d.T * 3 * d.H * d.es, 256);
  const long ss = align_up(d.Z * d.Srows * d.ldS * d.es, 256);
  s.xs = 0;
  s.layer0 = th * (d.L + 1);
  long o = 0;
  s.o_qkv = o; o += t3;
  s.o_ss = o; o += ss;
  s.o_ctx = o; o += th;
  s.o_a = o; o += th;
  s.o_h = o; o += th;
  s.o_u = o; o += ti;
  s.o_gu = o; o += ti;
  s.o_f = o; o += th;
  s.per_layer = o;
  s.total_bytes = s.layer0 + s.per_layer * d.L;
  return s;
}

struct Slots {
  char* x; char* y; char* qkv; char* ss; char* ctx; char* a; char* h; char* u; char* gu; char* f;
};

struct Workspace {
  // fp32 score-sized buffers
  float* sc0; float* sc1;
  // activation-dtype temporaries
  char* ss0; char* ss1;
  char* qkv; char* ctx; char* a; char* h; char* u; char* gu; char* f; char* x0; char* x1;
  float* ln_partial;
  int32_t* key_idx; int32_t* key_cnt;  // visible-key lists of an inference pass (one int per token row, one per sequence)
  float* mlm_tmp;  // unused by the encoder passes
  // the operands a layer's weight gradients read (f, u, qkv, a), four sets: the gradients of up to four
  // layers are launched together, so a set must outlive the three layers that follow (set 0 = f, u, qkv, a)
  char* rf[4]; char* ru[4]; char* rqkv[4]; char* ra[4];
  long total_bytes;
};

Workspace ws_layout(const Dims& d, char* base) {
  Workspace w;
  long o = 0;
  auto take = [&](long bytes) { char* p = base ? base + o : nullptr; o += align_up(bytes, 256); return p; };
  const long scb = d.Z * d.Srows * d.ldS * 4L;
  w.sc0 = (float*)take(scb);
  w.sc1 = (float*)take(scb);
  w.ss0 = take(d.Z * d.Srows * d.ldS * (long)d.es);
  w.ss1 = take(d.Z * d.Srows * d.ldS * (long)d.es);
  w.qkv = take(d.T * 3 * d.H * d.es);
  w.ctx = take(d.T * d.H * d.es);
  w.a = take(d.T * d.H * d.es);
  w.h = take(d.T * d.H * d.es);
  w.u = take(d.T * d.I * d.es);
  w.gu = take(d.T * d.I * d.es);
  w.f = take(d.T * d.H * d.es);
  w.x0 = take(d.T * d.H * d.es);
  w.x1 = take(d.T * d.H * d.es);
  w.ln_partial = (float*)take(3L * d.H * 1024 * 4);
  w.key_idx = (int32_t*)take(d.T * 4);
  w.key_cnt = (int32_t*)take((long)d.B * 4);
  w.mlm_tmp = nullptr;
  w.rf[0] = w.f; w.ru[0] = w.u; w.rqkv[0] = w.qkv; w.ra[0] = w.a;
  for (int i = 1; i < 4; ++i) {
    w.rf[i] = take(d.T * d.H * d.es);
    w.ru[i] = take(d.T * d.I * d.es);
    w.rqkv[i] = take(d.T * 3 * d.H * d.es);
    w.ra[i] = take(d.T * d.H * d.es);
  }
  w.total_bytes = o;
  return w;
}

bool make_dims(const pg_model_cfg* c, int dtype, int B, int S, Dims& d) {
  if (!c || B <= 0 || S <= 0) return false;
  if (c->hidden % c->heads || c->hidden / c->heads != 64) return false;
  if (dtype != PG_F32 && dtype != PG_BF16) return false;
  d.B = B; d.S = S; d.H = c->hidden; d.I = c->inter; d.nh = c->heads; d.L = c->layers; d.V = c->vocab_size;
  d.ldS = pg_score_ld(S);
  d.Srows = S;
  d.T = (long)B * S; d.Z = (long)B * c->heads;
  d.es = dtype == PG_F32 ? 4 : 2;
  return true;
}

// Dims of a ragged pass (rg != NULL) or of the B x S rectangle.  Lengths must be multiples of 8 in 8..max_len.
bool make_dims_rg(const pg_model_cfg* c, int dtype, int B, int S, const pg_ragged* rg, Dims& d, Rag& r) {
  r = Rag();
  if (!rg) return make_dims(c, dtype, B, S, d);
  if (!rg->seq_off || !rg->seq_off_host || rg->max_len != S || (S % 8)) return false;
  if (!make_dims(c, dtype, B, S, d)) return false;
  r.off = rg->seq_off; r.off_h = rg->seq_off_host; r.tok0 = rg->seq_off_host[0]; r.B = B;
  if (r.tok0 < 0) return false;
  for (int b = 0; b < B; ++b) {
    const int len = r.len(b);
    if (len < 8 || len > S || (len % 8)) return false;
  }
  d.T = r.row(B);
  d.Srows = d.ldS;
  return true;
}

// The attention of a pass as independent sub-problems for the kernels that take a B x S rectangle: the whole
// pass (rectangular), or one sequence at a time (ragged; each keeps its scores as [heads][len][ld(len)] at the
// start of its slots).  tok = first token row, slot = first element of its score slots.
struct Sub { Dims d; long tok, slot; };
inline int n_subs(const Dims& d, const Rag& r) { (void)d; return r.on() ? r.B : 1; }
inline Sub sub_at(const Dims& d, const Rag& r, int i) {
  Sub u;
  u.d = d; u.tok = 0; u.slot = 0;
  if (r.on()) {
    const int len = r.len(i);
    u.d.B = 1; u.d.S = len; u.d.ldS = pg_score_ld(len); u.d.Srows = len; u.d.T = len; u.d.Z = d.nh;
    u.tok = r.row(i);
    u.slot = (long)i * d.nh * d.Srows * d.ldS;
  }
  return u;
}

struct ParamView {
  const pg_model_cfg* c; const pg_params* p; int dtype;
  const float* m(long off) const { return p->master + off; }
  float* g(long off) const { return p->grad ? p->grad + off : nullptr; }
  // matrix operand in the compute dtype
  const void* w(long off) const {
    return dtype == PG_F32 ? (const void*)(p->master + off) : (const void*)((const bf16_t*)p->shadow + off);
  }
  long lo(int l, int which) const { return pg_param_offset_layer(c, l, which); }
};

#define CK(expr) do { int _rc = (expr); if (_rc != PG_OK) return _rc; } while (0)

// Plain 2-D GEMM helper.
struct G {
  int dtype; hipStream_t st;
  int run(const void* A, bool ak, long lda, const void* B, bool bk, long ldb, void* C, int cdt, long ldc, int M, int N,
          int K, float alpha, float beta, const float* bias = nullptr, const void* res = nullptr, long ldr = 0,
          int act = PG_ACT_NONE, void* aux = nullptr, long ldaux = 0, void* aux2 = nullptr) const {
    pg_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.A = A; d.B = B; d.C = C; d.dtype = dtype; d.c_dtype = cdt; d.a_kmajor = ak; d.b_kmajor = bk;
    d.M = M; d.N = N; d.K = K; d.lda = lda; d.ldb = ldb; d.ldc = ldc; d.batch = 1; d.batch_inner = 1;
    d.alpha = alpha; d.beta = beta; d.bias = bias; d.residual = res; d.ldr = ldr; d.aux = aux; d.ldaux = ldaux;
    d.aux2 = aux2; d.act = act;
    return pg_gemm(&d, st);
  }
  // dW += dY^T X (+ dYb^T Xb when a second operand pair is given), fp32 accumulate.
  pg_gemm_desc dw_desc(const void* dY, const void* X, float* dW, int No, int Ni, int T, const void* dYb = nullptr,
                       const void* Xb = nullptr) const {
    pg_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.A = dY; d.B = X; d.C = dW; d.dtype = dtype; d.c_dtype = PG_F32; d.a_kmajor = 0; d.b_kmajor = 0;
    d.M = No; d.N = Ni; d.K = T; d.lda = No; d.ldb = Ni; d.ldc = Ni; d.batch = 1; d.batch_inner = 1;
    d.alpha = 1.f; d.beta = 1.f;
    if (dYb) { d.A2 = dYb; d.B2 = Xb; d.K2 = T; }
    return d;
  }
  int dw(const pg_gemm_desc& d) const { return pg_gemm(&d, st); }
  int dw_pair(const pg_gemm_desc& d0, const pg_gemm_desc& d1) const { return pg_gemm_pair(&d0, &d1, st); }
  // Batched over (b, head).  Operands living in a [T][ld] activation buffer are addressed with
  // (outer = S*ld, inner = 64); score-shaped operands with (outer = nh*S*ldS, inner = S*ldS).
  // (A2, B2): optional second reduction segment in the layout of (A, B), N <= 64 products only.
  // rag (fused bf16 passes over ragged sequences, N = 64, M = K = S): the token-row operands follow seq_off.
  int bat(const Dims& dm, const void* A, bool ak, long lda, bool a_score, const void* B, bool bk, long ldb, bool b_score,
          void* C, int cdt, long ldc, bool c_score, int M, int N, int K, float alpha, float beta, const Rag* rag = nullptr,
          const void* A2 = nullptr, const void* B2 = nullptr) const {
    pg_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.A = A; d.B = B; d.C = C; d.dtype = dtype; d.c_dtype = cdt; d.a_kmajor = ak; d.b_kmajor = bk;
    d.M = M; d.N = N; d.K = K; d.lda = lda; d.ldb = ldb; d.ldc = ldc;
    d.batch = (int)dm.Z; d.batch_inner = dm.nh;
    const long so = (long)dm.nh * dm.Srows * dm.ldS, si = (long)dm.Srows * dm.ldS;
    d.sAo = a_score ? so : (long)dm.S * lda; d.sAi = a_score ? si : 64;
    d.sBo = b_score ? so : (long)dm.S * ldb; d.sBi = b_score ? si : 64;
    d.sCo = c_score ? so : (long)dm.S * ldc; d.sCi = c_score ? si : 64;
    d.alpha = alpha; d.beta = beta;
    if (A2) { d.A2 = A2; d.B2 = B2; d.K2 = K; }
    if (rag && rag->on()) {
      d.seq_off = rag->off; d.seq_tok0 = rag->tok0;
      d.seq_ops = (a_score ? 0 : 1) | (b_score ? 0 : 2) | (c_score ? 0 : 4);
    }
    return pg_gemm(&d, st);
  }
};

// Weight gradients of consecutive layers, launched together.  One layer's four products make
// 64 + 64 + 48 + 16 = 192 tiles of 256x256 at ProtBERT's sizes: three quarters of the 256 CUs for the length
// of one tile.  Four layers make 768 = three full rounds.  `depth` layers are collected (their operands stay
// in separate buffer sets meanwhile), then one grouped launch; the layers' "gradients issued" events follow it.
struct DwQueue {
  pg_gemm_desc d[16];
  int layers[4];
  int n = 0, nl = 0, depth = 1;
  hipStream_t st = nullptr;
  void* const* done = nullptr;
  int flush() {
    if (!n) return PG_OK;
    const int rc = pg_gemm_group(d, n, st);
    if (rc != PG_OK) return rc;
    if (done)
      for (int i = 0; i < nl; ++i)
        if (hipEventRecord((hipEvent_t)done[layers[i]], st) != hipSuccess) return PG_ERR_LAUNCH;
    n = nl = 0;
    return PG_OK;
  }
  int push(int layer, const pg_gemm_desc* four) {
    for (int i = 0; i < 4; ++i) d[n++] = four[i];
    layers[nl++] = layer;
    return nl >= depth ? flush() : PG_OK;
  }
};
}  // namespace
int pg_internal_fused_dgelu_save = 1;  // pg_tune key 15: GELU backward in the GEMM epilogue also when adjoints are kept
int pg_internal_dw_depth = 0;  // 0 = choose by tile count; 1..4 = force (pg_tune key 13)
int pg_internal_round_mask = 0;  // pg_tune key 18
int pg_internal_visible_keys = 1;  // pg_tune key 20: inference attention over the visible keys only
int pg_internal_attn_key_stationary = 1;  // pg_tune key 26: 0 = dk, dv of the first-order backward as two GEMMs over dS and P
namespace {
// Layers per grouped launch: the count in 1..4 whose tiles fill whole rounds of the 256 CUs best.
inline int dw_depth(const Dims& d, int dtype) {
  if (pg_internal_dw_depth >= 1 && pg_internal_dw_depth <= 4) return pg_internal_dw_depth;
  if (dtype != PG_BF16 || d.T % 64) return 1;
  auto t = [](long m, long n) { return ((m + 255) / 256) * ((n + 255) / 256); };
  const long per = 2 * t(d.H, d.I) + t(3L * d.H, d.H) + t(d.H, d.H);
  if (per * 2 < 256) return 1;  // such products are not launched as one group of 256x256 tiles anyway
  int best = 1;
  double best_eff = 0.0;
  for (int n = 1; n <= 4; ++n) {
    const long tiles = per * n, rounds = (tiles + 255) / 256;
    const double eff = (double)tiles / (double)(rounds * 256);
    if (eff > best_eff + 1e-9) { best_eff = eff; best = n; }
  }
  return best;
}

// The tangent and second-order passes must agree on this: with the fused kernels the score slot
// of the tangent stash holds rho ([B*heads][S] fp32), otherwise the tangent scores.
inline bool fused_gp_attention(int dtype, int S) { return pg_internal_fused_gp && dtype == PG_BF16 && S <= 2048; }

inline char* off(char* p, long elems, int es) { return p + elems * es; }
inline const char* off(const char* p, long elems, int es) { return p + elems * es; }

Slots layer_slots(char* base, const StashLayout& sl, const Dims& d, int l) {
  Slots s;
  const long th = align_up(d.T * d.H * d.es, 256);
  char* lb = base + sl.layer0 + sl.per_layer * l;
  s.x = base + sl.xs + th * l;
  s.y = base + sl.xs + th * (l + 1);
  s.qkv = lb + sl.o_qkv; s.ss = lb + sl.o_ss; s.ctx = lb + sl.o_ctx; s.a = lb + sl.o_a; s.h = lb + sl.o_h;
  s.u = lb + sl.o_u; s.gu = lb + sl.o_gu; s.f = lb + sl.o_f;
  return s;
}

// The primal stash of a pass may have been written by one forward over more sequences than the pass works on
// (several batches pushed through the encoder together): `rows` sequences laid out, this pass's B sequences
// start at sequence `row0`.  Every slot is sequence-major, so the view is the slot plus an offset.
struct StashView {
  Dims dl;
  StashLayout sl;
  long row0;     // first sequence of the view
  long tok_off;  // its first token row inside the stash
};
// pass: the viewing pass's dims (its T rows must fit).  Ragged: the stash holds view_T token rows (0 = the pass's
// own count) and the view starts at token row tok_abs; rectangular: rows * S token rows, view at row0 * S.
bool make_view(const pg_model_cfg* cfg, int dtype, const Dims& pass, const Rag& r, int stash_rows, int stash_row0, int view_T,
               long tok_abs, StashView& v) {
  const int rows = stash_rows > 0 ? stash_rows : pass.B;
  if (stash_row0 < 0 || stash_row0 + pass.B > rows) return false;
  if (!make_dims(cfg, dtype, rows, pass.S, v.dl)) return false;
  v.row0 = stash_row0;
  if (r.on()) {
    v.dl.T = view_T > 0 ? view_T : pass.T;
    v.dl.Srows = v.dl.ldS;
    v.tok_off = tok_abs;
    if (tok_abs < 0 || tok_abs + pass.T > v.dl.T) return false;
  } else {
    v.tok_off = (long)stash_row0 * pass.S;
  }
  v.sl = stash_layout(v.dl);
  return true;
}
Slots view_slots(char* base, const StashView& v, int l) {
  Slots s = layer_slots(base, v.sl, v.dl, l);
  const Dims& d = v.dl;
  const long t = v.tok_off * d.es;
  s.x += t * d.H; s.y += t * d.H; s.ctx += t * d.H; s.a += t * d.H; s.h += t * d.H; s.f += t * d.H;
  s.qkv += t * 3 * d.H;
  s.u += t * d.I; s.gu += t * d.I;
  s.ss += v.row0 * d.nh * d.Srows * d.ldS * d.es;
  return s;
}

}  // namespace

// ------------------------------------------------------------------ parameter layout
namespace {
long gsize(const pg_model_cfg* c, int which) {
  const long H = c->hidden;
  switch (which) {
    case PG_P_WORD_EMB: return (long)c->vocab_size * H;
    case PG_P_POS_EMB: return (long)c->max_pos * H;
    case PG_P_TYPE_EMB: return (long)c->type_vocab * H;
    case PG_P_EMB_LN_G: case PG_P_EMB_LN_B: return H;
    case PG_P_MLM_BIAS: return c->vocab_size;
    case PG_P_MLM_W: return H * H;
    case PG_P_MLM_B: case PG_P_MLM_LN_G: case PG_P_MLM_LN_B: return H;
  }
  return -1;
}
long lsize(const pg_model_cfg* c, int which) {
  const long H = c->hidden, I = c->inter;
  switch (which) {
    case PG_L_WQ: case PG_L_WK: case PG_L_WV: case PG_L_WO: return H * H;
    case PG_L_BQ: case PG_L_BK: case PG_L_BV: case PG_L_BO: return H;
    case PG_L_LN1_G: case PG_L_LN1_B: case PG_L_LN2_G: case PG_L_LN2_B: return H;
    case PG_L_W1: return I * H;
    case PG_L_B1: return I;
    case PG_L_W2: return H * I;
    case PG_L_B2: return H;
  }
  return -1;
}
// every tensor starts on an 8-element boundary so bf16 operands stay 16-byte aligned
long pad8(long n) { return (n + 7) / 8 * 8; }
long globals_total(const pg_model_cfg* c) {
  long o = 0;
  for (int i = 0; i < PG_P_GLOBAL_COUNT; ++i) o += pad8(gsize(c, i));
  return o;
}
long layer_total(const pg_model_cfg* c) {
  long o = 0;
  for (int i = 0; i < PG_L_COUNT; ++i) o += pad8(lsize(c, i));
  return o;
}
}  // namespace

extern "C" int64_t pg_param_size_global(const pg_model_cfg* c, int which) { return c ? gsize(c, which) : -1; }
extern "C" int64_t pg_param_size_layer(const pg_model_cfg* c, int which) { return c ? lsize(c, which) : -1; }
extern "C" int64_t pg_param_offset_global(const pg_model_cfg* c, int which) {
  if (!c || which < 0 || which >= PG_P_GLOBAL_COUNT) return -1;
  long o = 0;
  for (int i = 0; i < which; ++i) o += pad8(gsize(c, i));
  return o;
}
extern "C" int64_t pg_param_offset_layer(const pg_model_cfg* c, int layer, int which) {
  if (!c || which < 0 || which >= PG_L_COUNT || layer < 0 || layer >= c->layers) return -1;
  long o = globals_total(c) + layer_total(c) * layer;
  for (int i = 0; i < which; ++i) o += pad8(lsize(c, i));
  return o;
}
extern "C" int64_t pg_param_total(const pg_model_cfg* c) { return c ? globals_total(c) + layer_total(c) * c->layers : -1; }

extern "C" int64_t pg_workspace_bytes(const pg_model_cfg* cfg, int dtype, int B, int S) {
  Dims d;
  if (!make_dims(cfg, dtype, B, S, d)) return -1;
  return ws_layout(d, nullptr).total_bytes;
}
extern "C" int64_t pg_stash_bytes(const pg_model_cfg* cfg, int dtype, int B, int S) {
  Dims d;
  if (!make_dims(cfg, dtype, B, S, d)) return -1;
  return stash_layout(d).total_bytes;
}
extern "C" int64_t pg_workspace_bytes_ragged(const pg_model_cfg* cfg, int dtype, int B, int max_len, int T) {
  Dims d;
  if (T <= 0 || !make_dims(cfg, dtype, B, max_len, d)) return -1;
  d.T = T; d.Srows = d.ldS;
  return ws_layout(d, nullptr).total_bytes;
}
extern "C" int64_t pg_stash_bytes_ragged(const pg_model_cfg* cfg, int dtype, int B, int max_len, int T) {
  Dims d;
  if (T <= 0 || !make_dims(cfg, dtype, B, max_len, d)) return -1;
  d.T = T; d.Srows = d.ldS;
  return stash_layout(d).total_bytes;
}

// Precision experiments (pg_tune key 18; fp32 passes): the slots of one layer rounded to bf16 values once the layer
// is done, token-row slots under `tok_bit`, the score slot under `score_bit`.
static int round_layer_slots(const Slots& s, const Dims& d, int dtype, int tok_bit, int score_bit, bool with_y, hipStream_t st) {
  if (dtype != PG_F32 || !(pg_internal_round_mask & (tok_bit | score_bit))) return PG_OK;
  const long T = d.T;
  if (pg_internal_round_mask & tok_bit) {
    struct { char* p; long n; } v[] = {{s.qkv, T * 3 * d.H}, {s.ctx, T * d.H}, {s.a, T * d.H}, {s.h, T * d.H}, {s.u, T * d.I},
                                       {s.gu, T * d.I}, {s.f, T * d.H}, {with_y ? s.y : s.x, T * d.H}};
    for (auto& e : v)
      if (e.p) CK(pg_round_bf16_f32((float*)e.p, e.n, st));
  }
  if ((pg_internal_round_mask & score_bit) && s.ss) CK(pg_round_bf16_f32((float*)s.ss, d.Z * d.Srows * d.ldS, st));
  return PG_OK;
}

// ------------------------------------------------------------------ forward
// Packed inference form (seq_off != NULL; bf16, ids only, nothing stashed): the B sequences have their own lengths
// seq_off[b+1] - seq_off[b] <= S and lie back to back in T_packed token rows; every token-wise kernel
// runs on T_packed rows, attention walks each sequence's own keys.  Buffers are laid out for B x S.
// Ragged form (rg != NULL): the same with every option of the rectangular pass (stash, fp32, x_in); buffers are
// laid out for the packed token count.
static int forward_impl(const pg_model_cfg* cfg, const pg_params* p, int dtype, int B, int S, const int64_t* ids,
                        const void* x_in, const float* mask, void* workspace, int64_t workspace_bytes, void* stash,
                        void* hidden_out, void* emb_sum_out, void* stream, const int32_t* seq_off, const int32_t* pos_ids,
                        int T_packed, const pg_ragged* rg) {
  Dims d;
  Rag rag;
  if (!make_dims_rg(cfg, dtype, B, S, rg, d, rag) || !p || !p->master || (!ids && !x_in) || !workspace) return PG_ERR_ARG;
  if (dtype == PG_BF16 && !p->shadow) return PG_ERR_ARG;
  if (!stash && !hidden_out) return PG_ERR_ARG;
  if (d.ldS > 1024 && !(!stash && dtype == PG_BF16 && S <= 2048)) return PG_ERR_UNSUPPORTED;
  if (seq_off && (rg || !ids || !pos_ids || stash || emb_sum_out || T_packed <= 0 || T_packed > d.T)) return PG_ERR_ARG;
  if (seq_off && !(dtype == PG_BF16 && S <= 2048)) return PG_ERR_UNSUPPORTED;
  if (rag.on() && (rag.tok0 != 0 || (ids && !pos_ids))) return PG_ERR_ARG;  // a forward owns all of its token rows
  Workspace w = ws_layout(d, (char*)workspace);
  if (w.total_bytes > workspace_bytes) return PG_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  ParamView pv{cfg, p, dtype};
  G g{dtype, st};
  const int es = d.es, H = d.H, I = d.I, T = seq_off ? T_packed : (int)d.T;
  StashLayout sl = stash_layout(d);
  const bool fused = dtype == PG_BF16 && S <= 2048;

  char* x = stash ? (char*)stash + sl.xs : w.x0;
  if (seq_off) {
    CK(pg_embed_ln_packed(dtype, ids, pos_ids, pv.m(pg_param_offset_global(cfg, PG_P_WORD_EMB)),
                          pv.m(pg_param_offset_global(cfg, PG_P_POS_EMB)), pv.m(pg_param_offset_global(cfg, PG_P_TYPE_EMB)),
                          pv.m(pg_param_offset_global(cfg, PG_P_EMB_LN_G)), pv.m(pg_param_offset_global(cfg, PG_P_EMB_LN_B)),
                          cfg->ln_eps, x, T, H, st));
  } else if (ids && rag.on()) {
    CK(pg_embed_ln2_packed(dtype, ids, pos_ids, pv.m(pg_param_offset_global(cfg, PG_P_WORD_EMB)),
                           pv.m(pg_param_offset_global(cfg, PG_P_POS_EMB)), pv.m(pg_param_offset_global(cfg, PG_P_TYPE_EMB)),
                           pv.m(pg_param_offset_global(cfg, PG_P_EMB_LN_G)), pv.m(pg_param_offset_global(cfg, PG_P_EMB_LN_B)),
                           cfg->ln_eps, x, emb_sum_out, T, H, st));
  } else if (ids) {
    CK(pg_embed_ln2(dtype, ids, pv.m(pg_param_offset_global(cfg, PG_P_WORD_EMB)), pv.m(pg_param_offset_global(cfg, PG_P_POS_EMB)),
                    pv.m(pg_param_offset_global(cfg, PG_P_TYPE_EMB)), pv.m(pg_param_offset_global(cfg, PG_P_EMB_LN_G)),
                    pv.m(pg_param_offset_global(cfg, PG_P_EMB_LN_B)), cfg->ln_eps, x, emb_sum_out, B, S, H, st));
  } else {
    if (hipMemcpyAsync(x, x_in, (size_t)T * H * es, hipMemcpyDeviceToDevice, st) != hipSuccess) return PG_ERR_LAUNCH;
  }
  // nothing kept for a backward: the fused attention walks each sequence's visible keys only (one list per pass)
  const int32_t* inf_off = seq_off ? seq_off : (rag.on() ? rag.off : nullptr);
  const bool keys = pg_internal_visible_keys && !stash && fused && mask;
  if (keys) CK(pg_attention_key_lists(mask, inf_off, B, S, w.key_idx, w.key_cnt, st));
  for (int l = 0; l < d.L; ++l) {
    Slots s;
    if (stash) {
      s = layer_slots((char*)stash, sl, d, l);
    } else {
      s.x = x; s.y = (x == w.x0) ? w.x1 : w.x0;
      s.qkv = w.qkv; s.ss = w.ss0; s.ctx = w.ctx; s.a = w.a; s.h = w.h; s.u = nullptr; s.gu = w.gu; s.f = w.f;
    }
    if (l == d.L - 1 && !stash) s.y = (char*)hidden_out;
    // QKV projection (BertSelfAttention.query/key/value as one [3H,H] matrix)
    CK(g.run(s.x, true, H, pv.w(pv.lo(l, PG_L_WQ)), true, H, s.qkv, dtype, 3 * H, T, 3 * H, H, 1.f, 0.f,
             pv.m(pv.lo(l, PG_L_BQ))));
    if (seq_off || (!stash && fused)) {
      // nothing is kept for a backward: fused attention, no score matrix in HBM
      CK(pg_attention_fwd_keys(dtype, s.qkv, mask, s.ctx, inf_off, B, S, d.nh, 3 * H, H, keys ? w.key_idx : nullptr,
                               keys ? w.key_cnt : nullptr, st));
    } else if (fused && rag.on()) {
      CK(pg_attention_fwd_p_ragged(dtype, s.qkv, mask, s.ss, s.ctx, rag.off, B, S, 0, d.nh, 3 * H, H, st));
    } else if (fused) {
      // fused attention that also writes P once for the backward passes
      CK(pg_attention_fwd_p(dtype, s.qkv, mask, s.ss, s.ctx, B, S, d.nh, 3 * H, H, st));
    } else {
      for (int i = 0; i < n_subs(d, rag); ++i) {
        const Sub u = sub_at(d, rag, i);
        const char* qkv = s.qkv + u.tok * 3 * H * es;
        char* ss = s.ss + u.slot * es;
        // scores = Q K^T / 8
        CK(g.bat(u.d, qkv, true, 3 * H, false, off(qkv, H, es), true, 3 * H, false, w.sc0, PG_F32, u.d.ldS, true, u.d.S, u.d.S,
                 64, 0.125f, 0.f));
        CK(pg_softmax_fwd(dtype, w.sc0, mask ? mask + u.tok : nullptr, ss, u.d.B, d.nh, u.d.S, st));
        // ctx = P V
        CK(g.bat(u.d, ss, true, u.d.ldS, true, off(qkv, 2 * H, es), false, 3 * H, false, s.ctx + u.tok * H * es, dtype, H, false,
                 u.d.S, 64, u.d.S, 1.f, 0.f));
      }
    }
    // a = ctx Wo^T + bo + x ; h = LN1(a)
    CK(g.run(s.ctx, true, H, pv.w(pv.lo(l, PG_L_WO)), true, H, s.a, dtype, H, T, H, H, 1.f, 0.f, pv.m(pv.lo(l, PG_L_BO)),
             s.x, H));
    CK(pg_layernorm(dtype, s.a, pv.m(pv.lo(l, PG_L_LN1_G)), pv.m(pv.lo(l, PG_L_LN1_B)), cfg->ln_eps, s.h, T, H, st));
    // gu = gelu(h W1^T + b1) ; f = gu W2^T + b2 + h ; y = LN2(f)
    CK(g.run(s.h, true, H, pv.w(pv.lo(l, PG_L_W1)), true, H, s.gu, dtype, I, T, I, H, 1.f, 0.f, pv.m(pv.lo(l, PG_L_B1)),
             nullptr, 0, PG_ACT_GELU, s.u, I));
    CK(g.run(s.gu, true, I, pv.w(pv.lo(l, PG_L_W2)), true, I, s.f, dtype, H, T, H, I, 1.f, 0.f, pv.m(pv.lo(l, PG_L_B2)),
             s.h, H));
    CK(pg_layernorm(dtype, s.f, pv.m(pv.lo(l, PG_L_LN2_G)), pv.m(pv.lo(l, PG_L_LN2_B)), cfg->ln_eps, s.y, T, H, st));
    if (stash) CK(round_layer_slots(s, d, dtype, 1, 2, true, st));
    x = s.y;
  }
  if (stash && hidden_out) {
    if (hipMemcpyAsync(hidden_out, x, (size_t)T * H * es, hipMemcpyDeviceToDevice, st) != hipSuccess) return PG_ERR_LAUNCH;
  }
  return PG_OK;
}

extern "C" int pg_encoder_forward(const pg_model_cfg* cfg, const pg_params* p, int dtype, int B, int S, const int64_t* ids,
                                  const void* x_in, const float* mask, void* workspace, int64_t workspace_bytes,
                                  void* stash, void* hidden_out, void* emb_sum_out, void* stream) {
  return forward_impl(cfg, p, dtype, B, S, ids, x_in, mask, workspace, workspace_bytes, stash, hidden_out, emb_sum_out, stream,
                      nullptr, nullptr, 0, nullptr);
}

extern "C" int pg_encoder_forward_packed(const pg_model_cfg* cfg, const pg_params* p, int dtype, int nseq, int max_len,
                                         const int32_t* seq_off, int T, const int64_t* ids, const int32_t* pos_ids,
                                         const float* mask, void* workspace, int64_t workspace_bytes, void* hidden_out,
                                         void* stream) {
  if (!seq_off || !hidden_out) return PG_ERR_ARG;
  return forward_impl(cfg, p, dtype, nseq, max_len, ids, nullptr, mask, workspace, workspace_bytes, nullptr, hidden_out, nullptr,
                      stream, seq_off, pos_ids, T, nullptr);
}

extern "C" int pg_encoder_forward_ragged(const pg_model_cfg* cfg, const pg_params* p, int dtype, int B, const pg_ragged* rg,
                                         const int64_t* ids, const int32_t* pos_ids, const void* x_in, const float* mask,
                                         void* workspace, int64_t workspace_bytes, void* stash, void* hidden_out,
                                         void* emb_sum_out, void* stream) {
  if (!rg) return PG_ERR_ARG;
  return forward_impl(cfg, p, dtype, B, rg->max_len, ids, x_in, mask, workspace, workspace_bytes, stash, hidden_out, emb_sum_out,
                      stream, nullptr, pos_ids, 0, rg);
}

// MLM head: t = LN(gelu(hidden W^T + b)) ; logits = t word^T + bias
extern "C" int pg_mlm_logits(const pg_model_cfg* cfg, const pg_params* p, int dtype, const void* hidden, int T,
                             void* workspace, int64_t workspace_bytes, float* logits, void* stream) {
  if (!cfg || !p || !hidden || !workspace || !logits || T <= 0) return PG_ERR_ARG;
  const int H = cfg->hidden, es = dtype == PG_F32 ? 4 : 2;
  const long need = 2 * align_up((long)T * H * es, 256);
  if (need > workspace_bytes) return PG_ERR_WORKSPACE;
  char* t0 = (char*)workspace;
  char* t1 = t0 + align_up((long)T * H * es, 256);
  hipStream_t st = (hipStream_t)stream;
  ParamView pv{cfg, p, dtype};
  G g{dtype, st};
  CK(g.run(hidden, true, H, pv.w(pg_param_offset_global(cfg, PG_P_MLM_W)), true, H, t0, dtype, H, T, H, H, 1.f, 0.f,
           pv.m(pg_param_offset_global(cfg, PG_P_MLM_B)), nullptr, 0, PG_ACT_GELU, nullptr, 0));
  CK(pg_layernorm(dtype, t0, pv.m(pg_param_offset_global(cfg, PG_P_MLM_LN_G)), pv.m(pg_param_offset_global(cfg, PG_P_MLM_LN_B)),
                  cfg->ln_eps, t1, T, H, st));
  return pg_mlm_decode(dtype, t1, pv.m(pg_param_offset_global(cfg, PG_P_WORD_EMB)),
                       pv.m(pg_param_offset_global(cfg, PG_P_MLM_BIAS)), logits, T, H, cfg->vocab_size, st);
}

// MLM head, backward.  Nothing of the forward is kept: u = hidden W^T + b, gelu(u) and the LayerNorm output are
// computed again here (one H x H product, against the encoder's 30 layers).
//   dt = dlogits word ; g_word += dlogits^T t ; g_bias += colsum(dlogits)
//   dg = LN'(gelu(u))^T dt (+ g_gamma, g_beta) ; du = dg * gelu'(u) ; g_b += colsum(du) ; g_W += du^T hidden
//   dhidden = du W
extern "C" int64_t pg_mlm_backward_workspace_bytes(const pg_model_cfg* cfg, int dtype, int T) {
  if (!cfg || T <= 0) return -1;
  const int es = dtype == PG_F32 ? 4 : 2;
  return 4 * align_up((long)T * cfg->hidden * es, 256) + align_up(3L * cfg->hidden * pg_ln_partial_rows(T) * 4, 256);
}

extern "C" int pg_mlm_backward(const pg_model_cfg* cfg, const pg_params* p, int dtype, const void* hidden, int T,
                               const float* dlogits, int param_grads, void* dhidden, void* workspace, int64_t workspace_bytes,
                               void* stream) {
  if (!cfg || !p || !p->master || !hidden || !dlogits || !dhidden || !workspace || T <= 0) return PG_ERR_ARG;
  if (dtype == PG_BF16 && !p->shadow) return PG_ERR_ARG;
  if (param_grads && !p->grad) return PG_ERR_ARG;
  if (pg_mlm_backward_workspace_bytes(cfg, dtype, T) > workspace_bytes) return PG_ERR_WORKSPACE;
  const int H = cfg->hidden, V = cfg->vocab_size, es = dtype == PG_F32 ? 4 : 2;
  const long sz = align_up((long)T * H * es, 256);
  char* u = (char*)workspace;       // pre-GELU
  char* gl = u + sz;                // gelu(u), later du
  char* t = gl + sz;                // LayerNorm output, later dt
  char* dg = t + sz;
  float* partial = (float*)(dg + sz);
  hipStream_t st = (hipStream_t)stream;
  ParamView pv{cfg, p, dtype};
  G g{dtype, st};
  auto go = [&](int which) { return pg_param_offset_global(cfg, which); };
  auto gr = [&](int which) -> float* { return param_grads ? pv.g(go(which)) : nullptr; };
  CK(g.run(hidden, true, H, pv.w(go(PG_P_MLM_W)), true, H, gl, dtype, H, T, H, H, 1.f, 0.f, pv.m(go(PG_P_MLM_B)), nullptr, 0,
           PG_ACT_GELU, u, H));
  if (param_grads)
    CK(pg_layernorm(dtype, gl, pv.m(go(PG_P_MLM_LN_G)), pv.m(go(PG_P_MLM_LN_B)), cfg->ln_eps, t, T, H, st));
  // the decoder's weight gradient reads t before dt takes its place
  if (param_grads)
    CK(pg_mlm_decode_bwd(dtype, dlogits, t, nullptr, nullptr, gr(PG_P_WORD_EMB), gr(PG_P_MLM_BIAS), T, H, V, st));
  CK(pg_mlm_decode_bwd(dtype, dlogits, nullptr, pv.m(go(PG_P_WORD_EMB)), t, nullptr, nullptr, T, H, V, st));
  CK(pg_layernorm_bwd(dtype, gl, t, pv.m(go(PG_P_MLM_LN_G)), cfg->ln_eps, dg, gr(PG_P_MLM_LN_G), gr(PG_P_MLM_LN_B), nullptr,
                      partial, T, H, st));
  char* du = gl;
  CK(pg_gelu_bwd2(dtype, u, nullptr, nullptr, dg, du, (int64_t)T * H, st));
  if (param_grads) {
    CK(pg_colsum_acc(dtype, du, H, T, H, gr(PG_P_MLM_B), st));
    CK(g.dw(g.dw_desc(du, hidden, gr(PG_P_MLM_W), H, H, T)));
  }
  return g.run(du, true, H, pv.w(go(PG_P_MLM_W)), false, H, dhidden, dtype, H, T, H, H, 1.f, 0.f);
}

// ------------------------------------------------------------------ first-order backward
extern "C" int pg_encoder_backward(const pg_model_cfg* cfg, const pg_params* p, int dtype, int B, int S, const void* stash,
                                   void* d_hidden, const float* mask, int flags, void* adj_stash, void* dx_out,
                                   void* workspace, int64_t workspace_bytes, void* stream, void* const* layer_done,
                                   int stash_rows, int stash_row0, int grad_row0, const pg_ragged* rg) {
  (void)mask;
  Dims d;
  Rag rag;
  if (!make_dims_rg(cfg, dtype, B, S, rg, d, rag) || !p || !stash || !d_hidden || !workspace) return PG_ERR_ARG;
  StashView sv;
  if (!make_view(cfg, dtype, d, rag, stash_rows, stash_row0, rg ? rg->stash_T : 0, rag.tok0, sv)) return PG_ERR_ARG;
  const bool grads = flags & PG_BWD_PARAM_GRADS, save = flags & PG_BWD_SAVE_ADJOINTS;
  if (grads && !p->grad) return PG_ERR_ARG;
  if (save && !adj_stash) return PG_ERR_ARG;
  if (dtype == PG_BF16 && !p->shadow) return PG_ERR_ARG;
  const int epc = dtype == PG_F32 ? 4 : 8;
  if (S % epc) return PG_ERR_UNSUPPORTED;  // m-major operands of extent S
  Workspace w = ws_layout(d, (char*)workspace);
  if (w.total_bytes > workspace_bytes) return PG_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  ParamView pv{cfg, p, dtype};
  G g{dtype, st};
  const int es = d.es, H = d.H, I = d.I, T = (int)d.T;
  StashLayout sl = stash_layout(d);
  const size_t th_bytes = (size_t)d.T * H * es;
  const bool fused = dtype == PG_BF16 && S <= 2048;
  const Rag* frag = rag.on() ? &rag : nullptr;
  // parameter gradients are taken from the sequences grad_row0.. only (the others ride along for their input
  // gradient and adjoints: the gradient-penalty interpolates in front of the real and fake rows)
  if (grad_row0 < 0 || grad_row0 >= B) return PG_ERR_ARG;
  const long r0 = rag.on() ? rag.row(grad_row0) : (long)grad_row0 * S;
  const int Tg = T - (int)r0;
  auto ln_bwd = [&](const char* x, const char* dyp, const float* gamma, char* out, float* gG, float* gB, float* gX) -> int {
    if (!grads) return pg_layernorm_bwd(dtype, x, dyp, gamma, cfg->ln_eps, out, nullptr, nullptr, nullptr, w.ln_partial, T, H, st);
    if (r0 == 0) return pg_layernorm_bwd(dtype, x, dyp, gamma, cfg->ln_eps, out, gG, gB, gX, w.ln_partial, T, H, st);
    const int rc = pg_layernorm_bwd(dtype, x, dyp, gamma, cfg->ln_eps, out, nullptr, nullptr, nullptr, w.ln_partial, (int)r0, H, st);
    if (rc != PG_OK) return rc;
    const long o = r0 * H * es;
    return pg_layernorm_bwd(dtype, x + o, dyp + o, gamma, cfg->ln_eps, out + o, gG, gB, gX, w.ln_partial, Tg, H, st);
  };

  char* dy = (char*)d_hidden;
  if (save) {
    Slots last = layer_slots((char*)adj_stash, sl, d, d.L - 1);
    if (hipMemcpyAsync(last.y, dy, th_bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return PG_ERR_LAUNCH;
  }
  DwQueue dwq;
  dwq.st = st; dwq.done = layer_done; dwq.depth = grads ? dw_depth(d, dtype) : 1;
  for (int l = d.L - 1; l >= 0; --l) {
    const Slots s = view_slots((char*)const_cast<void*>(stash), sv, l);
    Slots a;  // adjoint buffers of this layer
    if (save) {
      a = layer_slots((char*)adj_stash, sl, d, l);
    } else {
      const int r = (d.L - 1 - l) % dwq.depth;
      a.qkv = w.rqkv[r]; a.ss = w.ss0; a.ctx = w.ctx; a.a = w.ra[r]; a.h = w.h; a.u = w.ru[r]; a.gu = w.gu; a.f = w.rf[r];
      a.x = (dy == w.x0) ? w.x1 : w.x0;
    }
    if (l == 0 && dx_out && !save) a.x = (char*)dx_out;
    // LN2
    // (the column sums of a LayerNorm backward's output are the bias gradient of the Linear before it)
    CK(ln_bwd(s.f, dy, pv.m(pv.lo(l, PG_L_LN2_G)), a.f, pv.g(pv.lo(l, PG_L_LN2_G)), pv.g(pv.lo(l, PG_L_LN2_B)),
              pv.g(pv.lo(l, PG_L_B2))));
    // (the four weight gradients of the layer are launched together once dqkv exists: see below)
    // d gu = df W2 ; du = dgu * gelu'(u)
    if (save && !pg_internal_fused_dgelu_save) {
      CK(g.run(a.f, true, H, pv.w(pv.lo(l, PG_L_W2)), false, I, a.gu, dtype, I, T, I, H, 1.f, 0.f));
      CK(pg_gelu_bwd2(dtype, s.u, nullptr, nullptr, a.gu, a.u, (int64_t)d.T * I, st));
    } else {
      // (the adjoint of the GELU output, which the second-order pass needs, leaves the same epilogue when adjoints are kept)
      CK(g.run(a.f, true, H, pv.w(pv.lo(l, PG_L_W2)), false, I, a.u, dtype, I, T, I, H, 1.f, 0.f, nullptr, nullptr, 0,
               PG_ACT_MUL_DGELU, s.u, I, save ? a.gu : nullptr));
    }
    if (grads) CK(pg_colsum_acc(dtype, a.u + r0 * I * es, I, Tg, I, pv.g(pv.lo(l, PG_L_B1)), st));
    // dh = du W1 + df
    CK(g.run(a.u, true, I, pv.w(pv.lo(l, PG_L_W1)), false, H, a.h, dtype, H, T, H, I, 1.f, 0.f, nullptr, a.f, H));
    // LN1
    CK(ln_bwd(s.a, a.h, pv.m(pv.lo(l, PG_L_LN1_G)), a.a, pv.g(pv.lo(l, PG_L_LN1_G)), pv.g(pv.lo(l, PG_L_LN1_B)),
              pv.g(pv.lo(l, PG_L_BO))));
    // dctx = da Wo   (dWo = da^T ctx waits for dWqkv below: 64 + 192 tiles fill the chip together)
    CK(g.run(a.a, true, H, pv.w(pv.lo(l, PG_L_WO)), false, H, a.ctx, dtype, H, T, H, H, 1.f, 0.f));
    // attention: dP = dctx V^T ; dS = softmax_bwd ; dq = dS K/8 ; dk = dS^T Q/8 ; dv = P^T dctx
    if (fused && pg_internal_attn_key_stationary) {
      // dq, then dk and dv from P alone: dS is written only where the second-order pass will read it (kept adjoints)
      CK(pg_attention_bwd(dtype, s.qkv, s.ss, s.ctx, a.ctx, save ? a.ss : nullptr, a.qkv, w.sc0, rag.on() ? rag.off : nullptr, B,
                          S, rag.on() ? rag.tok0 : 0, d.nh, 3 * H, H, 3 * H, st));
    } else if (fused) {
      if (rag.on())
        CK(pg_attention_bwd_dq_ragged(dtype, s.qkv, s.ss, s.ctx, a.ctx, a.ss, a.qkv, rag.off, B, S, rag.tok0, d.nh, 3 * H, H,
                                      3 * H, st));
      else
        CK(pg_attention_bwd_dq(dtype, s.qkv, s.ss, s.ctx, a.ctx, a.ss, a.qkv, B, S, d.nh, 3 * H, H, 3 * H, st));
      CK(g.bat(d, a.ss, false, d.ldS, true, s.qkv, false, 3 * H, false, off(a.qkv, H, es), dtype, 3 * H, false, S, 64, S,
               0.125f, 0.f, frag));
      CK(g.bat(d, s.ss, false, d.ldS, true, a.ctx, false, H, false, off(a.qkv, 2 * H, es), dtype, 3 * H, false, S, 64, S, 1.f,
               0.f, frag));
    } else {
      for (int i = 0; i < n_subs(d, rag); ++i) {
        const Sub u = sub_at(d, rag, i);
        const char* qkv = s.qkv + u.tok * 3 * H * es;
        const char* ss = s.ss + u.slot * es;
        char* ass = a.ss + u.slot * es;
        char* aqkv = a.qkv + u.tok * 3 * H * es;
        const char* actx = a.ctx + u.tok * H * es;
        const int Su = u.d.S, ld = u.d.ldS;
        CK(g.bat(u.d, actx, true, H, false, off(qkv, 2 * H, es), true, 3 * H, false, w.sc0, PG_F32, ld, true, Su, Su, 64, 1.f, 0.f));
        CK(pg_softmax_bwd(dtype, w.sc0, ss, ass, (int)u.d.Z, Su, st));
        CK(g.bat(u.d, ass, true, ld, true, off(qkv, H, es), false, 3 * H, false, aqkv, dtype, 3 * H, false, Su, 64, Su, 0.125f, 0.f));
        CK(g.bat(u.d, ass, false, ld, true, qkv, false, 3 * H, false, off(aqkv, H, es), dtype, 3 * H, false, Su, 64, Su, 0.125f, 0.f));
        CK(g.bat(u.d, ss, false, ld, true, actx, false, H, false, off(aqkv, 2 * H, es), dtype, 3 * H, false, Su, 64, Su, 1.f, 0.f));
      }
    }
    if (grads) {
      // dW2 = df^T gu, dW1 = du^T h, dWqkv = dqkv^T x, dWo = da^T ctx: 64 + 64 + 48 + 16 tiles of 256x256
      const long oH = r0 * H * es, oI = r0 * I * es, oQ = r0 * 3 * H * es;
      const pg_gemm_desc dws[4] = {g.dw_desc(a.f + oH, s.gu + oI, pv.g(pv.lo(l, PG_L_W2)), H, I, Tg),
                                   g.dw_desc(a.u + oI, s.h + oH, pv.g(pv.lo(l, PG_L_W1)), I, H, Tg),
                                   g.dw_desc(a.qkv + oQ, s.x + oH, pv.g(pv.lo(l, PG_L_WQ)), 3 * H, H, Tg),
                                   g.dw_desc(a.a + oH, s.ctx + oH, pv.g(pv.lo(l, PG_L_WO)), H, H, Tg)};
      CK(pg_colsum_acc(dtype, a.qkv + oQ, 3 * H, Tg, 3 * H, pv.g(pv.lo(l, PG_L_BQ)), st));
      CK(dwq.push(l, dws));
    }
    // dx = dqkv Wqkv + da
    CK(g.run(a.qkv, true, 3 * H, pv.w(pv.lo(l, PG_L_WQ)), false, H, a.x, dtype, H, T, H, 3 * H, 1.f, 0.f, nullptr, a.a, H));
    if (save) CK(round_layer_slots(a, d, dtype, 4, 8, false, st));
    dy = a.x;
  }
  CK(dwq.flush());
  if (dx_out && dy != (char*)dx_out) {
    if (hipMemcpyAsync(dx_out, dy, th_bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return PG_ERR_LAUNCH;
  }
  return PG_OK;
}

// ------------------------------------------------------------------ GP tangent-forward pass
extern "C" int pg_encoder_jvp(const pg_model_cfg* cfg, const pg_params* p, int dtype, int B, int S, const void* stash,
                              const void* xdot_in, void* tan_stash, void* hdot_out, void* workspace, int64_t workspace_bytes,
                              void* stream, int stash_rows, int stash_row0, const pg_ragged* rg) {
  Dims d;
  Rag rag;
  if (!make_dims_rg(cfg, dtype, B, S, rg, d, rag) || !p || !stash || !xdot_in || !tan_stash || !workspace) return PG_ERR_ARG;
  StashView sv;
  if (!make_view(cfg, dtype, d, rag, stash_rows, stash_row0, rg ? rg->stash_T : 0, rag.tok0, sv)) return PG_ERR_ARG;
  if (dtype == PG_BF16 && !p->shadow) return PG_ERR_ARG;
  Workspace w = ws_layout(d, (char*)workspace);
  if (w.total_bytes > workspace_bytes) return PG_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  ParamView pv{cfg, p, dtype};
  G g{dtype, st};
  const int es = d.es, H = d.H, I = d.I, T = (int)d.T;
  StashLayout sl = stash_layout(d);
  const size_t th_bytes = (size_t)d.T * H * es;
  if (hipMemcpyAsync((char*)tan_stash + sl.xs, xdot_in, th_bytes, hipMemcpyDeviceToDevice, st) != hipSuccess)
    return PG_ERR_LAUNCH;
  for (int l = 0; l < d.L; ++l) {
    const Slots s = view_slots((char*)const_cast<void*>(stash), sv, l);
    const Slots t = layer_slots((char*)tan_stash, sl, d, l);
    CK(g.run(t.x, true, H, pv.w(pv.lo(l, PG_L_WQ)), true, H, t.qkv, dtype, 3 * H, T, 3 * H, H, 1.f, 0.f));
    if (fused_gp_attention(dtype, S)) {
      // one sweep; the score slot of the tangent stash keeps only rho = rowsum(P Sdot)
      if (rag.on())
        CK(pg_attention_jvp_ragged(dtype, s.qkv, t.qkv, s.ss, s.ctx, t.ctx, (float*)t.ss, rag.off, B, S, rag.tok0, d.nh, 3 * H,
                                   H, H, st));
      else
        CK(pg_attention_jvp(dtype, s.qkv, t.qkv, s.ss, s.ctx, t.ctx, (float*)t.ss, B, S, d.nh, 3 * H, H, H, st));
    } else {
      for (int i = 0; i < n_subs(d, rag); ++i) {
        const Sub u = sub_at(d, rag, i);
        const char* qkv = s.qkv + u.tok * 3 * H * es;
        const char* tqkv = t.qkv + u.tok * 3 * H * es;
        const char* ss = s.ss + u.slot * es;
        char* tss = t.ss + u.slot * es;
        char* tctx = t.ctx + u.tok * H * es;
        const int Su = u.d.S, ld = u.d.ldS;
        // Sdot = (Qdot K^T + Q Kdot^T)/8
        CK(g.bat(u.d, tqkv, true, 3 * H, false, off(qkv, H, es), true, 3 * H, false, w.sc0, PG_F32, ld, true, Su, Su, 64, 0.125f, 0.f));
        CK(g.bat(u.d, qkv, true, 3 * H, false, off(tqkv, H, es), true, 3 * H, false, w.sc0, PG_F32, ld, true, Su, Su, 64, 0.125f, 1.f));
        CK(pg_softmax_jvp(dtype, w.sc0, ss, tss, w.ss0, (int)u.d.Z, Su, st));
        // ctxdot = Pdot V + P Vdot
        CK(g.bat(u.d, w.ss0, true, ld, true, off(qkv, 2 * H, es), false, 3 * H, false, tctx, dtype, H, false, Su, 64, Su, 1.f, 0.f));
        CK(g.bat(u.d, ss, true, ld, true, off(tqkv, 2 * H, es), false, 3 * H, false, tctx, dtype, H, false, Su, 64, Su, 1.f, 1.f));
      }
    }
    CK(g.run(t.ctx, true, H, pv.w(pv.lo(l, PG_L_WO)), true, H, t.a, dtype, H, T, H, H, 1.f, 0.f, nullptr, t.x, H));
    CK(pg_layernorm_jvp(dtype, s.a, t.a, pv.m(pv.lo(l, PG_L_LN1_G)), cfg->ln_eps, t.h, T, H, st));
    CK(g.run(t.h, true, H, pv.w(pv.lo(l, PG_L_W1)), true, H, t.gu, dtype, I, T, I, H, 1.f, 0.f, nullptr, nullptr, 0,
             PG_ACT_MUL_DGELU, s.u, I, t.u));
    CK(g.run(t.gu, true, I, pv.w(pv.lo(l, PG_L_W2)), true, I, t.f, dtype, H, T, H, I, 1.f, 0.f, nullptr, t.h, H));
    CK(pg_layernorm_jvp(dtype, s.f, t.f, pv.m(pv.lo(l, PG_L_LN2_G)), cfg->ln_eps, t.y, T, H, st));
    CK(round_layer_slots(t, d, dtype, 16, 32, true, st));
  }
  if (hdot_out) {
    const Slots t = layer_slots((char*)tan_stash, sl, d, d.L - 1);
    if (hipMemcpyAsync(hdot_out, t.y, th_bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return PG_ERR_LAUNCH;
  }
  return PG_OK;
}

// ------------------------------------------------------------------ GP second-order reverse pass
extern "C" int pg_encoder_backward2(const pg_model_cfg* cfg, const pg_params* p, int dtype, int B, int S, const void* stash,
                                    const void* adj_stash, const void* tan_stash, int flags, void* xbar_out,
                                    void* workspace, int64_t workspace_bytes, void* stream, void* const* layer_done,
                                    int stash_rows, int stash_row0, int adj_rows, const pg_ragged* rg) {
  Dims d;
  Rag rag;
  if (!make_dims_rg(cfg, dtype, B, S, rg, d, rag) || !p || !stash || !adj_stash || !tan_stash || !workspace) return PG_ERR_ARG;
  StashView sv, sva;
  if (!make_view(cfg, dtype, d, rag, stash_rows, stash_row0, rg ? rg->stash_T : 0, rag.tok0, sv)) return PG_ERR_ARG;
  // adjoints of a pass over adj_rows >= B sequences, of which this pass's are the first
  if (!make_view(cfg, dtype, d, rag, adj_rows, 0, rg ? rg->adj_T : 0, 0, sva)) return PG_ERR_ARG;
  const bool grads = flags & PG_BWD_PARAM_GRADS;
  const bool wait_done = (flags & PG_BWD_WAIT_LAYER_DONE) && grads;
  if (grads && !p->grad) return PG_ERR_ARG;
  if (wait_done && !layer_done) return PG_ERR_ARG;
  if (dtype == PG_BF16 && !p->shadow) return PG_ERR_ARG;
  const int epc = dtype == PG_F32 ? 4 : 8;
  if (S % epc) return PG_ERR_UNSUPPORTED;
  Workspace w = ws_layout(d, (char*)workspace);
  if (w.total_bytes > workspace_bytes) return PG_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  ParamView pv{cfg, p, dtype};
  G g{dtype, st};
  const int es = d.es, H = d.H, I = d.I, T = (int)d.T;
  StashLayout sl = stash_layout(d);
  const size_t th_bytes = (size_t)d.T * H * es;
  const Rag* frag = rag.on() ? &rag : nullptr;

  char* ybar = nullptr;  // second-order adjoint of the layer output (zero above the last layer)
  DwQueue dwq;
  dwq.st = st; dwq.done = layer_done; dwq.depth = grads ? dw_depth(d, dtype) : 1;
  for (int l = d.L - 1; l >= 0; --l) {
    const Slots s = view_slots((char*)const_cast<void*>(stash), sv, l);
    const Slots a = view_slots((char*)const_cast<void*>(adj_stash), sva, l);
    const Slots t = layer_slots((char*)const_cast<void*>(tan_stash), sl, d, l);
    char* xbar = (ybar == w.x0) ? w.x1 : w.x0;
    const int r = (d.L - 1 - l) % dwq.depth;
    char* const wf = w.rf[r]; char* const wu = w.ru[r]; char* const wqkv = w.rqkv[r]; char* const wa = w.ra[r];
    // (the other pass's record of this layer's event is still its latest: this pass records event l only after the
    // weight-gradient launch that contains layer l, which comes later on this stream)
    if (wait_done && hipStreamWaitEvent(st, (hipEvent_t)layer_done[l], 0) != hipSuccess) return PG_ERR_LAUNCH;
    // LN2: fbar
    CK(pg_layernorm_bwd2(dtype, s.f, t.f, a.y, ybar, pv.m(pv.lo(l, PG_L_LN2_G)), cfg->ln_eps, wf,
                         grads ? pv.g(pv.lo(l, PG_L_LN2_G)) : nullptr, grads ? pv.g(pv.lo(l, PG_L_LN2_B)) : nullptr,
                         grads ? pv.g(pv.lo(l, PG_L_B2)) : nullptr, w.ln_partial, T, H, st));
    // (W2bar = fbar^T gu + gf^T gudot and the other three weight gradients: one grouped launch at the end of the layer)
    // gubar = fbar W2 ; ubar = gelu'(u) gubar + gelu''(u) udot ggu
    CK(g.run(wf, true, H, pv.w(pv.lo(l, PG_L_W2)), false, I, w.gu, dtype, I, T, I, H, 1.f, 0.f));
    CK(pg_gelu_bwd2(dtype, s.u, t.u, a.gu, w.gu, wu, (int64_t)d.T * I, st));
    if (grads) CK(pg_colsum_acc(dtype, wu, I, T, I, pv.g(pv.lo(l, PG_L_B1)), st));
    // hbar = ubar W1 + fbar
    CK(g.run(wu, true, I, pv.w(pv.lo(l, PG_L_W1)), false, H, w.h, dtype, H, T, H, I, 1.f, 0.f, nullptr, wf, H));
    // LN1: abar
    CK(pg_layernorm_bwd2(dtype, s.a, t.a, a.h, w.h, pv.m(pv.lo(l, PG_L_LN1_G)), cfg->ln_eps, wa,
                         grads ? pv.g(pv.lo(l, PG_L_LN1_G)) : nullptr, grads ? pv.g(pv.lo(l, PG_L_LN1_B)) : nullptr,
                         grads ? pv.g(pv.lo(l, PG_L_BO)) : nullptr, w.ln_partial, T, H, st));  // (the Wo products pair with the QKV ones below)
    // ctxbar = abar Wo
    CK(g.run(wa, true, H, pv.w(pv.lo(l, PG_L_WO)), false, H, w.ctx, dtype, H, T, H, H, 1.f, 0.f));
    // attention second order
    if (fused_gp_attention(dtype, S)) {
      if (rag.on())
        CK(pg_attention_bwd2_ragged(dtype, s.qkv, t.qkv, s.ss, s.ctx, t.ctx, a.ctx, w.ctx, (const float*)t.ss, w.ss0 /*Sbar*/,
                                    w.ss1 /*Pdot*/, rag.off, B, S, rag.tok0, d.nh, 3 * H, H, st));
      else
        CK(pg_attention_bwd2(dtype, s.qkv, t.qkv, s.ss, s.ctx, t.ctx, a.ctx, w.ctx, (const float*)t.ss, w.ss0 /*Sbar*/,
                             w.ss1 /*Pdot*/, B, S, d.nh, 3 * H, H, st));
      // qbar = (Sbar K + gS Kdot)/8 ; kbar = (Sbar^T Q + gS^T Qdot)/8 ; vbar = P^T ctxbar + Pdot^T gctx:
      // each one launch over two reduction segments
      CK(g.bat(d, w.ss0, true, d.ldS, true, off(s.qkv, H, es), false, 3 * H, false, wqkv, dtype, 3 * H, false, S, 64, S,
               0.125f, 0.f, frag, a.ss, off(t.qkv, H, es)));
      CK(g.bat(d, w.ss0, false, d.ldS, true, s.qkv, false, 3 * H, false, off(wqkv, H, es), dtype, 3 * H, false, S, 64, S,
               0.125f, 0.f, frag, a.ss, t.qkv));
      // (ctxbar and gctx have row stride H: the two segments of vbar must share a leading dimension, they do)
      CK(g.bat(d, s.ss, false, d.ldS, true, w.ctx, false, H, false, off(wqkv, 2 * H, es), dtype, 3 * H, false, S, 64, S, 1.f,
               0.f, frag, w.ss1, a.ctx));
    } else {
      for (int i = 0; i < n_subs(d, rag); ++i) {
        const Sub u = sub_at(d, rag, i);
        const char* qkv = s.qkv + u.tok * 3 * H * es;
        const char* tqkv = t.qkv + u.tok * 3 * H * es;
        const char* ss = s.ss + u.slot * es;
        const char* tss = t.ss + u.slot * es;
        const char* ass = a.ss + u.slot * es;
        const char* actx = a.ctx + u.tok * H * es;
        const char* cbar = w.ctx + u.tok * H * es;
        char* qb = wqkv + u.tok * 3 * H * es;
        const char* v = off(qkv, 2 * H, es);
        const char* vdot = off(tqkv, 2 * H, es);
        const int Su = u.d.S, ld = u.d.ldS;
        CK(g.bat(u.d, actx, true, H, false, v, true, 3 * H, false, w.sc0, PG_F32, ld, true, Su, Su, 64, 1.f, 0.f));      // gP
        CK(g.bat(u.d, cbar, true, H, false, v, true, 3 * H, false, w.sc1, PG_F32, ld, true, Su, Su, 64, 1.f, 0.f));      // X
        CK(g.bat(u.d, actx, true, H, false, vdot, true, 3 * H, false, w.sc1, PG_F32, ld, true, Su, Su, 64, 1.f, 1.f));
        CK(pg_softmax_bwd2(dtype, ss, tss, w.sc0, w.sc1, w.ss0 /*Sbar*/, w.ss1 /*Pdot*/, (int)u.d.Z, Su, st));
        // qbar = (Sbar K + gS Kdot)/8
        CK(g.bat(u.d, w.ss0, true, ld, true, off(qkv, H, es), false, 3 * H, false, qb, dtype, 3 * H, false, Su, 64, Su, 0.125f, 0.f));
        CK(g.bat(u.d, ass, true, ld, true, off(tqkv, H, es), false, 3 * H, false, qb, dtype, 3 * H, false, Su, 64, Su, 0.125f, 1.f));
        // kbar = (Sbar^T Q + gS^T Qdot)/8
        CK(g.bat(u.d, w.ss0, false, ld, true, qkv, false, 3 * H, false, off(qb, H, es), dtype, 3 * H, false, Su, 64, Su, 0.125f, 0.f));
        CK(g.bat(u.d, ass, false, ld, true, tqkv, false, 3 * H, false, off(qb, H, es), dtype, 3 * H, false, Su, 64, Su, 0.125f, 1.f));
        // vbar = P^T ctxbar + Pdot^T gctx
        CK(g.bat(u.d, ss, false, ld, true, cbar, false, H, false, off(qb, 2 * H, es), dtype, 3 * H, false, Su, 64, Su, 1.f, 0.f));
        CK(g.bat(u.d, w.ss1, false, ld, true, actx, false, H, false, off(qb, 2 * H, es), dtype, 3 * H, false, Su, 64, Su, 1.f, 1.f));
      }
    }
    if (grads) {
      const pg_gemm_desc dws[4] = {g.dw_desc(wf, s.gu, pv.g(pv.lo(l, PG_L_W2)), H, I, T, a.f, t.gu),
                                   g.dw_desc(wu, s.h, pv.g(pv.lo(l, PG_L_W1)), I, H, T, a.u, t.h),
                                   g.dw_desc(wqkv, s.x, pv.g(pv.lo(l, PG_L_WQ)), 3 * H, H, T, a.qkv, t.x),
                                   g.dw_desc(wa, s.ctx, pv.g(pv.lo(l, PG_L_WO)), H, H, T, a.a, t.ctx)};
      CK(pg_colsum_acc(dtype, wqkv, 3 * H, T, 3 * H, pv.g(pv.lo(l, PG_L_BQ)), st));
      CK(dwq.push(l, dws));
    }
    // xbar = qkvbar Wqkv + abar
    CK(g.run(wqkv, true, 3 * H, pv.w(pv.lo(l, PG_L_WQ)), false, H, xbar, dtype, H, T, H, 3 * H, 1.f, 0.f, nullptr, wa, H));
    ybar = xbar;
  }
  CK(dwq.flush());
  if (xbar_out) {
    if (hipMemcpyAsync(xbar_out, ybar, th_bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return PG_ERR_LAUNCH;
  }
  return PG_OK;
}
