/* protgen_hip.h — C ABI of the MI355X (gfx950) engine for the ProtBERT GAN hot path.
 *
 * The reference (UfukTanriverdi8/ProtGen-GAN) has no FFI: its hot path is Python
 * (models.py, loss.py, generate.py) over torch/transformers kernels.  Every entry point
 * below therefore names the reference Python call sites whose arithmetic it replaces.
 * All pointers are device pointers owned by the caller unless stated otherwise; nothing is
 * retained after a call returns; `stream` is a hipStream_t passed as void*.  Functions
 * return PG_OK (0) or a negative PG_ERR_* code; pg_strerror() names it.  No entry point
 * allocates or frees device memory.
 *
 * Activation dtype `dtype` is PG_F32 (exact fp32 MFMA path, used for token-identical
 * parity) or PG_BF16 (bf16 storage and MFMA inputs, fp32 accumulation and statistics).
 * Parameters are always fp32 ("master"); bf16 compute additionally reads a bf16 copy of
 * the same flat layout ("shadow").
 */
#ifndef PROTGEN_HIP_H
#define PROTGEN_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PG_OK 0
#define PG_ERR_ARG (-1)
#define PG_ERR_ALIGN (-2)
#define PG_ERR_LAUNCH (-3)
#define PG_ERR_UNSUPPORTED (-4)
#define PG_ERR_WORKSPACE (-5)
#define PG_ERR_SCRATCH (-6) /* beyond the capacity of the library's static device scratch */

#define PG_F32 0
#define PG_BF16 1

#define PG_ACT_NONE 0
#define PG_ACT_GELU 1      /* v = gelu(v); pre-activation optionally stored to aux */
#define PG_ACT_MUL_DGELU 2 /* v *= gelu'(aux[m][n]) */

const char* pg_strerror(int code);
int pg_abi_version(void);

/* ------------------------------------------------------------------ model description */
typedef struct {
  int vocab_size, hidden, layers, heads, inter, max_pos, type_vocab;
  float ln_eps;
} pg_model_cfg;

/* Flat parameter layout (element offsets into the fp32 master / grad / bf16 shadow
 * buffers).  `which` indexes the tensors of HF BertForMaskedLM
 * (transformers/models/bert/modeling_bert.py) in this order: */
enum {
  PG_P_WORD_EMB = 0, PG_P_POS_EMB, PG_P_TYPE_EMB, PG_P_EMB_LN_G, PG_P_EMB_LN_B,
  PG_P_MLM_BIAS, PG_P_MLM_W, PG_P_MLM_B, PG_P_MLM_LN_G, PG_P_MLM_LN_B,
  PG_P_GLOBAL_COUNT
};
enum { /* per encoder layer; Q,K,V weights are adjacent so they form one [3H,H] matrix */
  PG_L_WQ = 0, PG_L_WK, PG_L_WV, PG_L_BQ, PG_L_BK, PG_L_BV, PG_L_WO, PG_L_BO, PG_L_LN1_G, PG_L_LN1_B,
  PG_L_W1, PG_L_B1, PG_L_W2, PG_L_B2, PG_L_LN2_G, PG_L_LN2_B,
  PG_L_COUNT
};
int64_t pg_param_total(const pg_model_cfg* cfg);
int64_t pg_param_offset_global(const pg_model_cfg* cfg, int which);
int64_t pg_param_offset_layer(const pg_model_cfg* cfg, int layer, int which);
int64_t pg_param_size_global(const pg_model_cfg* cfg, int which);
int64_t pg_param_size_layer(const pg_model_cfg* cfg, int which);

/* Critic classification head nn.Sequential(Linear,ReLU,Linear,ReLU,Linear,ReLU,Linear)
 * (models.py:76-84): flat fp32 layout W0,b0,W2,b2,W4,b4,W6,b6 with dims H,H/2,H/4,H/16,1. */
int64_t pg_head_total(int hidden);
int64_t pg_head_offset(int hidden, int linear_index /*0..3*/, int is_bias);

typedef struct {
  const float* master; /* flat fp32 parameters */
  float* grad;         /* flat fp32 gradients (same layout), or NULL */
  const void* shadow;  /* flat bf16 copy of master, required when dtype == PG_BF16 */
} pg_params;

/* ------------------------------------------------------------------ K1/K7: GEMM
 * C = epilogue(alpha * A.B) on the matrix cores; replaces every torch.matmul / F.linear of
 * HF BertSelfAttention / BertSelfOutput / BertIntermediate / BertOutput
 * (modeling_bert.py:111-136,175-177,282-293,325-351) and their autograd backward.
 * a_kmajor: A stored [M][K] (1) or [K][M] (0); b_kmajor: B stored [N][K] (1) or [K][N] (0).
 * Batch index z in [0,batch) splits as (z / batch_inner, z % batch_inner) with strides
 * (s?o, s?i) in elements.  bias is fp32 [N]; residual/aux have the activation dtype.   */
typedef struct {
  const void* A; const void* B; void* C;
  int dtype, c_dtype, a_kmajor, b_kmajor;
  int M, N, K;
  int64_t lda, ldb, ldc;
  int batch, batch_inner;
  int64_t sAo, sAi, sBo, sBi, sCo, sCi;
  float alpha, beta;
  const float* bias;
  const void* residual; int64_t ldr;
  void* aux; int64_t ldaux;
  void* aux2; /* PG_ACT_MUL_DGELU only: optional store of v before the multiply (ld = ldaux) */
  int act;
  /* Optional second reduction segment (unbatched only): C = epilogue(alpha (A.B + A2.B2)), A2 / B2 in
   * the layout and leading dimensions of A / B with reduction extent K2 (0 = none).  The gradient
   * penalty's second-order pass sums two products into each weight gradient. */
  const void* A2; const void* B2; int K2;
  /* Optional ragged batches (bf16, N <= 64, B stored [K][N]; the key-side contractions of attention over
   * sequences of different lengths laid back to back): batch (zo, zi) = (sequence, head); device int32
   * seq_off[batch / batch_inner + 1].  Operands named in seq_ops (1 = A, 2 = B, 4 = C) are token-row
   * buffers: sequence zo starts at row seq_off[zo] - seq_tok0 (the outer batch stride is ignored) and
   * M = K = seq_off[zo+1] - seq_off[zo] (the descriptor's M = K bound them and size the grid).  The other
   * operands are score-shaped slots addressed by the batch strides; whole 64-row tiles of them are read, so
   * the rows / columns between a sequence's extent and the next multiple of 64 must hold zeros. */
  const int32_t* seq_off; int seq_tok0; int seq_ops;
} pg_gemm_desc;
int pg_gemm(const pg_gemm_desc* d, void* stream);
/* Two unbatched products in one launch (results identical to pg_gemm(d0) then pg_gemm(d1)):
 * the autograd backward of BertSelfOutput.dense and of the query/key/value Linears
 * (modeling_bert.py:282-293, 111-136) yields two weight-gradient products at the same point,
 * one of which alone covers a quarter of the chip.  Falls back to two launches when the two
 * descriptors do not share dtype / layout or are not whole-k-tile bf16 products. */
int pg_gemm_pair(const pg_gemm_desc* d0, const pg_gemm_desc* d1, void* stream);
/* n <= 16 unbatched products (d[0..n)) in one launch of 256x256 tiles when they share dtype / layout and
 * together cover at least half the CUs; otherwise pairs (pg_gemm_pair) and a single.  The autograd backward
 * of one BertLayer yields the weight gradients of its four Linears (modeling_bert.py:111-136, 282-293,
 * 325-351): none of them has enough 256x256 tiles to cover the chip alone; the encoder backward passes
 * collect those of up to four layers (192 tiles per layer at ProtBERT's sizes, 768 = three rounds of 256 CUs). */
int pg_gemm_group(const pg_gemm_desc* d, int n, void* stream);
/* Test and measurement hook (no reference counterpart): sets one dispatch threshold or switch of the library.
 * The settings are process-global and unsynchronised: call it from the thread that launches, never beside a launch.
 * A product run never calls it; the defaults are the measured best.  Everything else in this header is free of
 * hidden state: a call reads its arguments, launches on the given stream and keeps no pointer, so calls on different
 * streams / devices from different threads are independent (pg_profile_* excepted, see above).
 * key 0: bf16 GEMM path (0 register-staged, 1 direct-to-LDS auto, 2/4 ring depth); key 1: 128x64 tiles (0 never,
 * 1 N<=64, 2 also grids of at most 256 128x128 tiles); key 2: 256x256 tiles (0 never, 1 large-N products, 2 always);
 * key 4: 16-byte bf16 epilogue accesses (0/1); key 5: pg_gemm_pair in one launch (0/1); key 6: fused attention kernels
 * in the gradient-penalty passes (0/1); key 7: two-segment products in one launch (0/1); key 8: tile rows per group of
 * the grouped tile order (>= 1); key 9: a last round of 256x256 tiles less than this many per cent full sends its rows
 * to a second launch (0 = never, default 50); key 10: smallest K at which a narrow-N product with >= 512 tiles takes
 * the 256x256 kernel; key 11: pg_gemm_group in one launch (0/1); key 12: most row chunks of a pg_colsum_acc launch
 * below 8192 rows (twice that from there on); key 13: layers whose weight gradients share one grouped launch (0 = by
 * tile count, 1..4); key 14: fewest 256x256 tiles at which a product with 512 <= N < 2048 takes the 256x256 kernel
 * (default 129); key 15: GELU backward inside the GEMM epilogue also in passes that keep adjoints (0/1); key 16:
 * largest 128x128 grid (workgroups) that takes the four-buffer ring (default 256); key 17: waves per workgroup of a
 * pg_colsum_acc launch (4 or 16); key 18 (precision experiments, fp32 passes only): bit mask of the stored quantities
 * rounded to bf16 values at the end of every layer: 1 forward token-row slots, 2 attention probabilities, 4 adjoint
 * token-row slots of a backward that keeps adjoints, 8 adjoint scores, 16 tangent token-row slots, 32 tangent scores;
 * key 20: inference passes walk the visible keys only (0/1, default 1); key 23: GEMM epilogues fetch their per-element
 * operand several row blocks at a time (0/1, default 1); key 25: the forward that writes P keeps K / V of sequences of at
 * most 512 keys resident in LDS and makes one pass (0 = two sweeps over the K / V ring at every length; default 1);
 * key 26: dk, dv of the first-order attention backward from the key-stationary kernel of pg_attention_bwd (0 = two
 * GEMMs over dS and P; default 1).  Other keys: PG_ERR_ARG.
 * PG_TUNE="key=value,..." in the environment applies the same settings once, when the library is loaded. */
int pg_tune(int key, int value);
/* Measurement aid (no reference counterpart): time every `sample_every`-th pg_gemm launch
 * with HIP events on its own stream.  pg_profile_end synchronises those events, stops the
 * sampling and fills out[16][3] = {launches sampled, FLOPs sampled, milliseconds} per kernel
 * variant, index 8*(bf16 inputs) + 4*(fp32 output) + 2*(a_kmajor) + b_kmajor.  Host state is
 * process-global and not thread-safe. */
int pg_profile_begin(int sample_every);
int pg_profile_end(double* out, int64_t* total_launches);

/* ------------------------------------------------------------------ row-wise kernels (K3/K4/K9/K10)
 * H must be 64, 128, 256, 512 or 1024 (one wavefront per row).                          */

/* BertEmbeddings (modeling_bert.py:53-108): word[ids]+type[0]+pos[s] -> LayerNorm. ids int64 [B,S]. */
int pg_embed_ln(int dtype, const int64_t* ids, const float* word, const float* pos, const float* type0,
                const float* gamma, const float* beta, float eps, void* out, int B, int S, int H, void* stream);
/* Same, also storing the pre-LayerNorm sum (needed by the embedding backward). */
int pg_embed_ln2(int dtype, const int64_t* ids, const float* word, const float* pos, const float* type0,
                 const float* gamma, const float* beta, float eps, void* out, void* sum_out, int B, int S, int H,
                 void* stream);
/* Same for T tokens with explicit positions (device int32 pos_ids[T]): rows of several sequences back to back. */
int pg_embed_ln_packed(int dtype, const int64_t* ids, const int32_t* pos_ids, const float* word, const float* pos,
                       const float* type0, const float* gamma, const float* beta, float eps, void* out, int T, int H,
                       void* stream);
/* pg_embed_ln_packed that also stores the pre-LayerNorm sum (for the embedding backward). */
int pg_embed_ln2_packed(int dtype, const int64_t* ids, const int32_t* pos_ids, const float* word, const float* pos,
                        const float* type0, const float* gamma, const float* beta, float eps, void* out, void* sum_out,
                        int T, int H, void* stream);
/* y = LayerNorm(x) (modeling_bert.py:289-293,347-351; residual is added by the GEMM epilogue). */
int pg_layernorm(int dtype, const void* x, const float* gamma, const float* beta, float eps, void* y,
                 int rows, int H, void* stream);
/* dx = LN'(x)^T dy ; dgamma += sum dy*xhat ; dbeta += sum dy ; dxsum += column sums of dx, which
 * is the bias gradient of the Linear whose output (plus residual) this LayerNorm normalises
 * (any of the three may be NULL).  `partial` is fp32 scratch of 3*H*pg_ln_partial_rows(rows) floats. */
int pg_ln_partial_rows(int rows);
int pg_layernorm_bwd(int dtype, const void* x, const void* dy, const float* gamma, float eps, void* dx,
                     float* dgamma, float* dbeta, float* dxsum, float* partial, int rows, int H, void* stream);
/* pg_layernorm_bwd without the `partial` argument, for callers that cannot allocate device memory (code
 * under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the
 * reference.  The partial sums live in a 12 MiB device array inside the library (one copy per GPU, held
 * until the process ends); limits: H <= 1024 (PG_ERR_SCRATCH beyond), any row count.  Calls no hipMalloc /
 * hipFree.  The device of `stream` must be current.  All calls on one GPU share the array, so two calls
 * that overlap on one GPU (different streams) race; calls on one stream are ordered and safe. */
int pg_layernorm_bwd_static(int dtype, const void* x, const void* dy, const float* gamma, float eps, void* dx,
                            float* dgamma, float* dbeta, float* dxsum, int rows, int H, void* stream);
/* Tangent (JVP): ydot = gamma * rstd * Proj(xdot). */
int pg_layernorm_jvp(int dtype, const void* x, const void* xdot, const float* gamma, float eps, void* ydot,
                     int rows, int H, void* stream);
/* Second-order reverse of the (primal, tangent) pair used by the gradient penalty:
 *   xbar = LN'(x)^T ybar + d/dx <gy, LN_jvp(x; xdot)>,
 *   dgamma += sum ybar*xhat + sum gy*rstd*Proj(xdot), dbeta += sum ybar, dxsum += column sums of xbar.
 * ybar may be NULL (treated as zero).                                                    */
int pg_layernorm_bwd2(int dtype, const void* x, const void* xdot, const void* gy, const void* ybar,
                      const float* gamma, float eps, void* xbar, float* dgamma, float* dbeta, float* dxsum,
                      float* partial, int rows, int H, void* stream);
/* ubar = gelu'(u)*gubar + gelu''(u)*udot*ggu  (GP second-order reverse of BertIntermediate's GELU). */
int pg_gelu_bwd2(int dtype, const void* u, const void* udot, const void* ggu, const void* gubar, void* ubar,
                 int64_t n, void* stream);
/* db[n] += sum_rows dY[row][n]  (bias gradients). */
int pg_colsum_acc(int dtype, const void* dY, int64_t ld, int rows, int N, float* db, void* stream);

/* ------------------------------------------------------------------ attention over materialised scores (K2/K8/K10)
 * scores/grad buffers are fp32 [Z][S][ldS] with ldS = pg_score_ld(S); mask is float [B][S]
 * of 0/1 (NULL = all ones), Z = B*heads.  Softmax follows eager_attention_forward
 * (modeling_bert.py:125-130) with the additive mask (1-m)*finfo(float32).min
 * (modeling_utils.py:1012-1024).  Outputs of activation dtype have row stride ldS and
 * zeros in columns >= S.                                                                */
/* Fused (flash-style) forward for passes that keep nothing for a backward: ctx = softmax(Q K^T/8
 * + keymask) V straight from the packed [T][3H] QKV buffer, bf16 only, S <= 2048. */
int pg_attention_fwd(int dtype, const void* qkv, const float* mask, void* ctx, int B, int S, int heads,
                     int64_t ld_qkv, int64_t ld_ctx, void* stream);
/* Same over sequences of different lengths laid back to back: sequence b = token rows
 * [seq_off[b], seq_off[b+1]) of qkv / mask / ctx (device int32 seq_off[nseq+1], lengths <= max_len). */
int pg_attention_fwd_packed(int dtype, const void* qkv, const float* mask, void* ctx, const int32_t* seq_off, int nseq,
                            int max_len, int heads, int64_t ld_qkv, int64_t ld_ctx, void* stream);
/* The fused forward over the visible keys only.  pg_attention_key_lists writes, per sequence (B x S rectangle, or
 * seq_off != NULL: packed rows), the positions of the keys whose mask is 1 in order (key_idx, one int per token row)
 * and their number (key_cnt; -1 = walk the sequence whole under the additive mask: a mask value other than 0 / 1,
 * or no visible key, where modeling_bert.py's softmax is uniform over all keys).  pg_attention_fwd_keys is
 * pg_attention_fwd / pg_attention_fwd_packed walking those lists (NULL lists: all keys): a masked key has weight
 * exactly zero in the reference's softmax, so the result is the same sum without its zero terms. */
int pg_attention_key_lists(const float* mask, const int32_t* seq_off, int nseq, int S, int32_t* key_idx, int32_t* key_cnt,
                           void* stream);
int pg_attention_fwd_keys(int dtype, const void* qkv, const float* mask, void* ctx, const int32_t* seq_off, int nseq, int S,
                          int heads, int64_t ld_qkv, int64_t ld_ctx, const int32_t* key_idx, const int32_t* key_cnt,
                          void* stream);
/* Same, additionally writing the probabilities P (activation dtype, [B*heads][S][pg_score_ld(S)],
 * zeros in the padding columns) that the backward and gradient-penalty passes consume. */
int pg_attention_fwd_p(int dtype, const void* qkv, const float* mask, void* P, void* ctx, int B, int S, int heads,
                       int64_t ld_qkv, int64_t ld_ctx, void* stream);
/* First half of the attention backward from the stored P: dS = P*(dO V^T - <dO,O>) (written,
 * same layout as P) and dQ = dS K / 8 (written to dq + head offset, row stride ld_dq).  The
 * second half is two GEMMs: dK = dS^T Q / 8, dV = P^T dO. */
int pg_attention_bwd_dq(int dtype, const void* qkv, const void* P, const void* ctx, const void* dctx, void* dS, void* dq,
                        int B, int S, int heads, int64_t ld_qkv, int64_t ld_ctx, int64_t ld_dq, void* stream);
/* Gradient-penalty passes of one attention block over the stored probabilities P
 * (the double backward torch.autograd.grad(..., create_graph=True) takes through
 * eager_attention_forward, modeling_bert.py:111-136, for loss.py:33-48).  bf16 only, S <= 2048.
 * pg_attention_jvp: tangent pass.  tqkv holds (dQ | dK | dV) in the layout of qkv; writes
 *   dctx = dP V + P dV with dP = P (dS - rho), dS = (dQ K^T + Q dK^T)/8, and rho = rowsum(P dS)
 *   ([B*heads][S], fp32) for the pass below.
 * pg_attention_bwd2: score side of the second-order reverse pass.  gctx is the first-order
 *   adjoint of ctx, ctxbar the second-order one; writes (bf16, [B*heads][S][pg_score_ld(S)])
 *   Sbar = P (X + gamma (dS - rho) - c dS - e) and dP, where gamma = gctx V^T,
 *   X = ctxbar V^T + gctx dV^T, c = <gctx, ctx>, e = <ctxbar, ctx> + <gctx, dctx> - rho c. */
int pg_attention_jvp(int dtype, const void* qkv, const void* tqkv, const void* P, const void* ctx, void* ctxdot, float* rho,
                     int B, int S, int heads, int64_t ld_qkv, int64_t ld_ctx, int64_t ld_out, void* stream);
int pg_attention_bwd2(int dtype, const void* qkv, const void* tqkv, const void* P, const void* ctx, const void* ctxdot,
                      const void* gctx, const void* ctxbar, const float* rho, void* Sbar, void* Pdot, int B, int S, int heads,
                      int64_t ld_qkv, int64_t ld_ctx, void* stream);

/* The four training kernels over sequences of different lengths laid back to back ("ragged"): sequence b is
 * the token rows [seq_off[b] - tok0, seq_off[b+1] - tok0) of every [T][ld] operand (device int32
 * seq_off[nseq+1]; lengths are multiples of 8, at most max_len).  Score-shaped operands (P, dS, Sbar, Pdot)
 * hold one slot of ld x ld elements per (sequence, head), ld = pg_score_ld(max_len), of which the leading
 * len rows and columns are used; outputs carry zeros from there to the next multiple of 64 in both
 * directions.  rho stays [nseq*heads][max_len].  The reference pads every row to 512 (dataset.py:19-42),
 * lengths are 200-500: the rows and keys skipped here are the ones its attention mask hides. */
int pg_attention_fwd_p_ragged(int dtype, const void* qkv, const float* mask, void* P, void* ctx, const int32_t* seq_off,
                              int nseq, int max_len, int tok0, int heads, int64_t ld_qkv, int64_t ld_ctx, void* stream);
int pg_attention_bwd_dq_ragged(int dtype, const void* qkv, const void* P, const void* ctx, const void* dctx, void* dS,
                               void* dq, const int32_t* seq_off, int nseq, int max_len, int tok0, int heads,
                               int64_t ld_qkv, int64_t ld_ctx, int64_t ld_dq, void* stream);
int pg_attention_jvp_ragged(int dtype, const void* qkv, const void* tqkv, const void* P, const void* ctx, void* ctxdot,
                            float* rho, const int32_t* seq_off, int nseq, int max_len, int tok0, int heads, int64_t ld_qkv,
                            int64_t ld_ctx, int64_t ld_out, void* stream);
int pg_attention_bwd2_ragged(int dtype, const void* qkv, const void* tqkv, const void* P, const void* ctx,
                             const void* ctxdot, const void* gctx, const void* ctxbar, const float* rho, void* Sbar,
                             void* Pdot, const int32_t* seq_off, int nseq, int max_len, int tok0, int heads, int64_t ld_qkv,
                             int64_t ld_ctx, void* stream);

/* The whole attention backward from the stored P in two launches.  First dQ as pg_attention_bwd_dq (dS is written
 * only when dS != NULL: the gradient penalty keeps it, nothing here reads it) together with the row constants
 * delta = <dO, O> (fp32 workspace of nseq*heads*max_len); then a key-stationary kernel forms dS again from P, dO,
 * V and delta in registers and accumulates dK = dS^T Q / 8 and dV = P^T dO per 128 keys: P is read twice in all,
 * where the two key-side GEMMs read P and dS once each after dS was written.  dqkv: [T][ld_dq] with dQ | dK | dV at
 * column offsets 0 | H | 2H (H = heads*64).  seq_off == NULL: nseq sequences of max_len rows each (a rectangle).
 * modeling_bert.py:111-136 (eager_attention_forward) is the forward whose backward this is. */
int pg_attention_bwd(int dtype, const void* qkv, const void* P, const void* ctx, const void* dctx, void* dS, void* dqkv,
                     float* delta, const int32_t* seq_off, int nseq, int max_len, int tok0, int heads, int64_t ld_qkv,
                     int64_t ld_ctx, int64_t ld_dq, void* stream);

int pg_score_ld(int S);
int pg_softmax_fwd(int dtype, const float* scores, const float* mask, void* P, int B, int heads, int S, void* stream);
/* dS = P * (dP - rowsum(dP*P)) */
int pg_softmax_bwd(int dtype, const float* dP, const void* P, void* dS, int Z, int S, void* stream);
/* Sdot_out = Sdot ; Pdot = P * (Sdot - rowsum(P*Sdot)) */
int pg_softmax_jvp(int dtype, const float* Sdot, const void* P, void* Sdot_out, void* Pdot, int Z, int S, void* stream);
/* GP second-order reverse: given P, Sdot, gP = gctx.V^T, X = ctxbar.V^T + gctx.Vdot^T:
 *   Pdot = P*(Sdot - r), r = rowsum(P*Sdot), c = rowsum(gP*P)
 *   Pbar = X + gP*(Sdot - r) - c*Sdot ;  Sbar = P*(Pbar - rowsum(Pbar*P))              */
int pg_softmax_bwd2(int dtype, const void* P, const void* Sdot, const float* gP, const float* X, void* Sbar,
                    void* Pdot, int Z, int S, void* stream);

/* ------------------------------------------------------------------ K5/K6: MLM head + sampler + unmask
 * Replaces models.py:21-63 (softmax(logits/T), multinomial over every position,
 * confidence gather, per-row top-k commit).                                              */
/* logits[t][v] = hidden_t . word[v] + bias[v] (tied decoder, modeling_bert.py:484-496), fp32 out. */
int pg_mlm_decode(int dtype, const void* t, const float* word, const float* bias, float* logits, int T, int H, int V,
                  void* stream);
/* Backward of pg_mlm_decode: dt[T][H] = dlogits word (when dt != NULL); gword[V][H] += dlogits^T t and
 * gbias[V] += column sums of dlogits (when gword != NULL; fp32 atomics).  V <= 64. */
int pg_mlm_decode_bwd(int dtype, const float* dlogits, const void* t, const float* word, void* dt, float* gword,
                      float* gbias, int T, int H, int V, void* stream);
/* Per token: p = softmax(logits/temperature); pred = argmax_v p_v / noise_v (== torch.multinomial
 * under the same Exp(1) draw); conf = p[pred].  noise fp32 [T][V].                       */
int pg_sample_tokens(const float* logits, const float* noise, float temperature, int T, int V, int64_t* pred,
                     float* conf, void* stream);
/* Per batch row (models.py:35-63): masks = ids==mask_id; k = all masks if fill_all else
 * min(max(1, int(keep_percent * #{ids not in {pad,cls,sep}})), #masks); commit pred at the
 * k masked positions of highest conf (ties: lowest index).  ids int64 [B][S] in place.
 * counts int32 [B][2] = {committed, masks remaining after the commit} (may be NULL).     */
int pg_commit_topk(int64_t* ids, const int64_t* pred, const float* conf, int B, int S, double keep_percent,
                   int fill_all, int mask_id, int pad_id, int cls_id, int sep_id, int32_t* counts, void* stream);

/* ------------------------------------------------------------------ critic head (models.py:76-84,104-105)
 * All fp32.  x is the CLS row of the last hidden state: row b at hidden + b*row_stride
 * (activation dtype).  `stash` holds the three post-ReLU activations and the input copy:
 * pg_head_stash_floats(H) floats per batch row.                                          */
int64_t pg_head_stash_floats(int hidden);
int pg_head_forward(int dtype, const void* hidden, int64_t row_stride, const float* head, int B, int H, float* scores,
                    float* stash, void* stream);
/* Backward of sum_b w[b]*score_b: accumulates head grads (if ghead != NULL) and writes
 * dx rows (activation dtype) to dhidden + b*row_stride (other rows untouched).          */
int pg_head_backward(int dtype, const float* head, float* stash, const float* w, int B, int H, float* ghead,
                     void* dhidden, int64_t row_stride, void* stream);
/* GP second-order: tangent xdot (row b at hdot + b*row_stride) through the head with the
 * primal's ReLU gates; accumulates ghead += scale * d<1, sdot>/dhead.  tstash is scratch of the
 * same size as stash; sdot_out [B] optional. */
int pg_head_jvp_grad(int dtype, const float* head, float* stash, const void* hdot, int64_t row_stride, int B,
                     int H, float scale, float* ghead, float* tstash, float* sdot_out, void* stream);

/* ------------------------------------------------------------------ K11: gradient-penalty reductions (loss.py:87-95)
 * g [B][S*H] activation dtype, mask float [B][S].  norms[b] = ||g_b * mask||_2 ;
 * gp = mean((norm-1)^2) ; v = scale * (2/B) * (norm-1)/norm * g * mask (the GP direction);
 * scale is read from device memory (*scale_dev, NULL = 1) so the loss weight arriving from
 * autograd needs no host synchronisation.                                                */
int pg_gp_norms(int dtype, const void* g, const float* mask, int B, int S, int H, float* norms, float* gp_out,
                void* stream);
int pg_gp_direction(int dtype, const void* g, const float* mask, const float* norms, int B, int S, int H,
                    const float* scale_dev, void* v, float* gp_out, void* stream);
/* The same two reductions when sample b is the token rows [seq_off[b], seq_off[b+1]) of g / mask / v (every
 * sample packed to its own length: rows past it have mask 0 in loss.py:87-88 and add nothing). */
int pg_gp_norms_ragged(int dtype, const void* g, const float* mask, const int32_t* seq_off, int B, int max_len, int H,
                       float* norms, float* gp_out, void* stream);
int pg_gp_direction_ragged(int dtype, const void* g, const float* mask, const float* norms, const int32_t* seq_off, int B,
                           int max_len, int H, const float* scale_dev, void* v, float* gp_out, void* stream);
/* x = alpha_b * er + (1-alpha_b) * ef (loss.py:29); alpha fp32 [B]. */
int pg_interpolate(int dtype, const void* er, const void* ef, const float* alpha, void* x, int B, int64_t per_sample,
                   void* stream);
/* der = alpha_b * dx ; def = (1-alpha_b) * dx */
int pg_interpolate_bwd(int dtype, const void* dx, const float* alpha, void* der, void* def, int B,
                       int64_t per_sample, void* stream);
/* loss.py:29 over packed rows: token row t of x is row t - off_x[b] of sequence b = seq_id[t]; the real and fake
 * embeddings of that sequence start at rows off_r[b], off_f[b] of er, ef (each batch packed to its own
 * lengths, none shorter than the interpolate's).  All index arrays are device int32.  The backward writes
 * (accumulate = 0) or adds (1) alpha_b dx and (1 - alpha_b) dx into the same rows of der, def and touches no
 * other row. */
int pg_interpolate_ragged(int dtype, const void* er, const void* ef, const float* alpha, void* x, const int32_t* seq_id,
                          const int32_t* off_x, const int32_t* off_r, const int32_t* off_f, int T, int H, void* stream);
int pg_interpolate_bwd_ragged(int dtype, const void* dx, const float* alpha, void* der, void* def, const int32_t* seq_id,
                              const int32_t* off_x, const int32_t* off_r, const int32_t* off_f, int T, int H, int accumulate,
                              void* stream);
/* Embedding backward: LayerNorm backward of BertEmbeddings followed by scatter-add into the
 * word / position / token-type tables (fp32 atomics).                                    */
int pg_embed_scatter(int dtype, const void* dsum, const int64_t* ids, float* gword, float* gpos, float* gtype0, int B,
                     int S, int H, void* stream);

/* The same for T packed token rows with explicit positions (device int32 pos_ids[T]). */
int pg_embed_scatter_packed(int dtype, const void* dsum, const int64_t* ids, const int32_t* pos_ids, float* gword,
                            float* gpos, float* gtype0, int T, int H, void* stream);

/* ------------------------------------------------------------------ K12: optimizer
 * torch.optim.AdamW semantics (fully_masked_train.py:74-75,326): decoupled weight decay,
 * bias correction with the 1-based `step`.  Optionally refreshes the bf16 shadow.        */
int pg_adamw(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr, float beta1,
             float beta2, float eps, float wd, int step, float grad_scale, void* stream);
int pg_cast_bf16(const float* src, void* dst, int64_t n, void* stream);
int pg_cast_f32(const void* src_bf16, float* dst, int64_t n, void* stream);
int pg_scale_f32(float* p, float s, int64_t n, void* stream);
/* p[i] = the nearest bf16 value of p[i], kept as fp32 (precision experiments: which stored quantity's rounding a
 * bf16 run owes its distance from fp32 to; see pg_tune key 18). */
int pg_round_bf16_f32(float* p, int64_t n, void* stream);

/* ------------------------------------------------------------------ engine: whole-pass orchestration
 * One call runs a complete pass over all layers with the kernels above (no host sync).
 * The caller provides the workspace and (for training) the stash buffers; sizes come from
 * pg_workspace_bytes / pg_stash_bytes.  Requirements: head_dim == 64; for passes that
 * stash, S % 64 == 0 is not required but score rows are padded to pg_score_ld(S).       */
int64_t pg_workspace_bytes(const pg_model_cfg* cfg, int dtype, int B, int S);
int64_t pg_stash_bytes(const pg_model_cfg* cfg, int dtype, int B, int S);

/* Ragged form of the training passes: the B sequences of a pass have their own lengths (multiples of 8, from 8
 * to max_len) and lie back to back in T token rows; every token-wise kernel runs on T rows and attention walks
 * each sequence's own keys.  The critic of the reference sees rows padded to 512 (dataset.py:19-42) whose
 * attention mask hides the padding (models.py:87-102, loss.py:41-66): hidden columns are never read by a visible
 * row, the critic reads the first row only (models.py:104), and their gradients are exactly zero, so a pass over
 * each row's leading columns up to its last visible one gives the same scores and gradients.  `mask` (per token,
 * T floats) still applies inside a sequence.
 *   seq_off / seq_off_host: device and host copies of int32 [B+1]; seq_off[b] = token row of sequence b in the
 *     buffers of the forward that wrote the primal stash (so a pass over a sub-range of that forward's sequences
 *     passes a pointer into the same array), seq_off[B] = end of the last one.
 *   stash_T: token rows of that forward (0 = this pass covers all of them).
 *   adj_T (pg_encoder_backward2): token rows of the pass that wrote the adjoint stash (0 = same as this pass).
 * Workspace and stash sizes: the *_ragged functions below (T token rows, score slots for B x max_len). */
typedef struct {
  int max_len;
  const int32_t* seq_off;
  const int32_t* seq_off_host;
  int stash_T;
  int adj_T;
} pg_ragged;
int64_t pg_workspace_bytes_ragged(const pg_model_cfg* cfg, int dtype, int B, int max_len, int T);
int64_t pg_stash_bytes_ragged(const pg_model_cfg* cfg, int dtype, int B, int max_len, int T);

/* Forward through embeddings (if ids != NULL, else x_in [B,S,H] is the encoder input, as in
 * models.py:94-102 / loss.py:62-66) and all encoder layers.  mask: float [B,S] or NULL.
 * hidden_out [B,S,H] activation dtype.  stash may be NULL (inference).  emb_sum_out
 * (nullable, ids path only) receives the pre-LayerNorm embedding sum for the backward.   */
int pg_encoder_forward(const pg_model_cfg* cfg, const pg_params* p, int dtype, int B, int S, const int64_t* ids,
                       const void* x_in, const float* mask, void* workspace, int64_t workspace_bytes, void* stash,
                       void* hidden_out, void* emb_sum_out, void* stream);
/* The same forward for sequences of different lengths laid back to back ("packed"; bf16, inference only):
 * sequence b is token rows [seq_off[b], seq_off[b+1]) of ids / pos_ids / mask / hidden_out (all T rows,
 * T <= nseq*max_len; seq_off and pos_ids are device int32, pos_ids[t] = position of token t in its
 * sequence).  Used by the stacked fill (several generate() batches of different padded widths in one
 * pass, models.py:18-64 per batch): columns beyond a batch's own width cost nothing.  workspace as for
 * pg_workspace_bytes(cfg, dtype, nseq, max_len).                                                      */
int pg_encoder_forward_packed(const pg_model_cfg* cfg, const pg_params* p, int dtype, int nseq, int max_len,
                              const int32_t* seq_off, int T, const int64_t* ids, const int32_t* pos_ids, const float* mask,
                              void* workspace, int64_t workspace_bytes, void* hidden_out, void* stream);
/* pg_encoder_forward over ragged sequences, with every option of the rectangular form: ids + pos_ids (device
 * int64 / int32 [T]) or x_in [T,H]; mask float [T] or NULL; stash for a backward (pg_stash_bytes_ragged);
 * hidden_out [T,H]; emb_sum_out [T,H] (ids path).  fp32 runs the attention of each sequence as its own launches
 * (seq_off_host), bf16 the fused ragged kernels. */
int pg_encoder_forward_ragged(const pg_model_cfg* cfg, const pg_params* p, int dtype, int B, const pg_ragged* rg,
                              const int64_t* ids, const int32_t* pos_ids, const void* x_in, const float* mask,
                              void* workspace, int64_t workspace_bytes, void* stash, void* hidden_out, void* emb_sum_out,
                              void* stream);
/* MLM head up to logits (Generator.forward, models.py:14-16): fp32 logits [T][V]. */
int pg_mlm_logits(const pg_model_cfg* cfg, const pg_params* p, int dtype, const void* hidden, int T, void* workspace,
                  int64_t workspace_bytes, float* logits, void* stream);
/* Backward of pg_mlm_logits (the reference takes it from autograd when the generator's logits feed a loss:
 * docs/HISTORY.md:169-179; head = BertOnlyMLMHead, tied decoder).  dlogits fp32 [T][V]; dhidden [T][H] in the
 * activation type.  With param_grads, the gradients of the head's dense, LayerNorm, decoder bias and of the tied
 * word-embedding table are added into p->grad.  Nothing of the forward is kept: the head is evaluated again. */
int64_t pg_mlm_backward_workspace_bytes(const pg_model_cfg* cfg, int dtype, int T);
int pg_mlm_backward(const pg_model_cfg* cfg, const pg_params* p, int dtype, const void* hidden, int T,
                    const float* dlogits, int param_grads, void* dhidden, void* workspace, int64_t workspace_bytes,
                    void* stream);
/* First-order backward of pg_encoder_forward from d_hidden [B,S,H] (activation dtype).
 * flags: bit0 = accumulate parameter gradients into p->grad; bit1 = write every layer's
 * adjoints into adj_stash (same layout as the forward stash; needed by the GP).
 * dx_out (nullable) receives the gradient w.r.t. the encoder input (the embedding output;
 * its own backward is pg_layernorm_bwd + pg_embed_scatter).
 * layer_done (nullable; also on pg_encoder_backward2): cfg->layers hipEvent_t handles.  With bit0 set,
 * event l is recorded on `stream` once every launch that adds into layer l's parameter gradients
 * has been issued (layers finish from the last to the first), so that a data-parallel caller can
 * start reducing a layer's gradients on another stream while the pass goes on.  The reference
 * trains on one GPU and has no counterpart.
 * stash_rows / stash_row0 (also on pg_encoder_jvp and pg_encoder_backward2; 0, 0 = the stash is this pass's
 * own): the primal stash was written by one pg_encoder_forward over stash_rows >= B sequences (several
 * batches through the encoder together: the gradient-penalty interpolates with the real and fake rows of
 * fully_masked_train.py:315-323) and this pass works on the B sequences from sequence stash_row0 on.  The
 * The tangent stash always holds exactly this pass's B sequences.
 * grad_row0 (pg_encoder_backward): parameter gradients are accumulated from the sequences grad_row0.. only;
 * the sequences before them get their input gradient and adjoints (the interpolates of the gradient penalty
 * in one backward with the real and fake rows).  adj_rows (pg_encoder_backward2; 0 = B): the adjoint stash was
 * written by such a pass over adj_rows sequences, this pass's B sequences are its first.
 * rg (last argument of the three passes; NULL = B x S rectangle): the pass is ragged, S = rg->max_len, and
 * d_hidden / dx_out / xdot_in / hdot_out / xbar_out hold this pass's T = seq_off[B] - seq_off[0] token rows.  */
#define PG_BWD_PARAM_GRADS 1
#define PG_BWD_SAVE_ADJOINTS 2
/* pg_encoder_backward2 only: before layer l writes any parameter gradient, the stream waits for layer_done[l] as an
 * earlier pass recorded it (a first-order backward over other rows of the same update that runs on another stream
 * and adds into the same gradient buffers); the pass then records the events anew as it goes. */
#define PG_BWD_WAIT_LAYER_DONE 4
int pg_encoder_backward(const pg_model_cfg* cfg, const pg_params* p, int dtype, int B, int S, const void* stash,
                        void* d_hidden, const float* mask, int flags, void* adj_stash, void* dx_out,
                        void* workspace, int64_t workspace_bytes, void* stream, void* const* layer_done,
                        int stash_rows, int stash_row0, int grad_row0, const pg_ragged* rg);
/* GP tangent-forward pass: xdot_in [B,S,H] -> hdot_out [B,S,H]; writes the tangent stash. */
int pg_encoder_jvp(const pg_model_cfg* cfg, const pg_params* p, int dtype, int B, int S, const void* stash,
                   const void* xdot_in, void* tan_stash, void* hdot_out, void* workspace, int64_t workspace_bytes,
                   void* stream, int stash_rows, int stash_row0, const pg_ragged* rg);
/* GP second-order reverse over (primal stash, adjoint stash, tangent stash): accumulates
 * d<v, ds/dx>/dtheta into p->grad (flags bit0; any loss weight is folded into v by
 * pg_gp_direction) and writes the gradient w.r.t. the encoder input x to xbar_out (nullable). */
int pg_encoder_backward2(const pg_model_cfg* cfg, const pg_params* p, int dtype, int B, int S, const void* stash,
                         const void* adj_stash, const void* tan_stash, int flags, void* xbar_out,
                         void* workspace, int64_t workspace_bytes, void* stream, void* const* layer_done,
                         int stash_rows, int stash_row0, int adj_rows, const pg_ragged* rg);

#ifdef __cplusplus
}
#endif
#endif
