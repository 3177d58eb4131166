// MLM decoder + sampler + confidence-ranked unmask (the fill step), critic head,
// gradient-penalty reductions, embedding gradient scatter, AdamW and dtype casts.
#include <float.h>
#include <math.h>
#include "pg_common.h"
#include "protgen_hip.h"

namespace {

inline int check_launch() { return hipGetLastError() == hipSuccess ? PG_OK : PG_ERR_LAUNCH; }

// ------------------------------------------------------------------ tied MLM decoder
// logits[t][v] = <x_t, word[v]> + bias[v].  The V x H fp32 table sits in LDS (120 KB for
// ProtBERT); each wave owns one token at a time: lanes hold interleaved 4-element chunks of
// the hidden row (conflict-free ds_read_b128 of the table), V wave reductions per token.
template <typename T, int VW, int NCH>
__global__ __launch_bounds__(1024) void mlm_decode_kernel(const T* __restrict__ x, const float* __restrict__ word,
                                                           const float* __restrict__ bias, float* __restrict__ logits,
                                                           int Tn, int H, int V) {
  extern __shared__ __align__(16) float wlds[];
  constexpr int N = VW * NCH;
  for (int i = threadIdx.x; i < V * H; i += blockDim.x) wlds[i] = word[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int waves_per_block = blockDim.x >> 6;
  for (long t = (long)blockIdx.x * waves_per_block + (threadIdx.x >> 6); t < Tn; t += (long)gridDim.x * waves_per_block) {
    float h[N];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      if constexpr (VW == 4) VecIO<T, 4>::load(x + t * H + (i * 64 + lane) * 4, &h[i * 4]);
      else h[i] = to_f32<T>(x[t * H + i * 64 + lane]);
    }
    float mine = 0.f;
    for (int v = 0; v < V; ++v) {
      const float* w = wlds + v * H;
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        if constexpr (VW == 4) {
          const float4 wv = *reinterpret_cast<const float4*>(w + (i * 64 + lane) * 4);
          s += h[i * 4] * wv.x + h[i * 4 + 1] * wv.y + h[i * 4 + 2] * wv.z + h[i * 4 + 3] * wv.w;
        } else {
          s += h[i] * w[i * 64 + lane];
        }
      }
      s = wave_sum(s);
      if (lane == v) mine = s + bias[v];
    }
    if (lane < V) logits[t * V + lane] = mine;
  }
}

constexpr int MAXV = 64;

// ------------------------------------------------------------------ tied MLM decoder, backward
// dt[t][h] = sum_v dl[t][v] * word[v][h].  A block takes 16 token rows, whose dl rows sit in LDS; a thread owns
// the columns h, h + 256, ... and walks the V rows of the table (coalesced, served by L2 after the first block).
template <typename T>
__global__ __launch_bounds__(256) void mlm_decode_bwd_x_kernel(const float* __restrict__ dl, const float* __restrict__ word,
                                                               T* __restrict__ dt, int Tn, int H, int V) {
  constexpr int ROWS = 16;
  __shared__ float s[ROWS * MAXV];
  const long t0 = (long)blockIdx.x * ROWS;
  const int rows = (int)min((long)ROWS, (long)Tn - t0);
  for (int i = threadIdx.x; i < ROWS * MAXV; i += blockDim.x) {
    const int r = i / MAXV, v = i % MAXV;
    s[i] = (r < rows && v < V) ? dl[(t0 + r) * V + v] : 0.f;
  }
  __syncthreads();
  for (int h = threadIdx.x; h < H; h += blockDim.x) {
    float acc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = 0.f;
    for (int v = 0; v < V; ++v) {
      const float w = word[(long)v * H + h];
#pragma unroll
      for (int r = 0; r < ROWS; ++r) acc[r] += s[r * MAXV + v] * w;
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
      if (r < rows) dt[(t0 + r) * H + h] = from_f32<T>(acc[r]);
  }
}

// gword[v][h] += sum_t dl[t][v] * x[t][h] ; gbias[v] += sum_t dl[t][v].  A block takes a span of token rows and 256
// columns; a thread owns one column and keeps one partial sum per vocabulary entry, added to the fp32 gradients
// with atomics at the end (blocks of different spans meet there).
template <typename T>
__global__ __launch_bounds__(256) void mlm_decode_bwd_w_kernel(const float* __restrict__ dl, const T* __restrict__ x,
                                                               float* __restrict__ gword, float* __restrict__ gbias, int Tn,
                                                               int H, int V, int span) {
  constexpr int CH = 16;
  __shared__ float s[CH * MAXV];
  const int h = blockIdx.x * 256 + threadIdx.x;
  const long t0 = (long)blockIdx.y * span, t1 = min((long)Tn, t0 + span);
  float acc[MAXV];
#pragma unroll
  for (int v = 0; v < MAXV; ++v) acc[v] = 0.f;
  float bsum = 0.f;
  for (long tc = t0; tc < t1; tc += CH) {
    __syncthreads();
    for (int i = threadIdx.x; i < CH * MAXV; i += blockDim.x) {
      const int r = i / MAXV, v = i % MAXV;
      s[i] = (tc + r < t1 && v < V) ? dl[(tc + r) * V + v] : 0.f;
    }
    __syncthreads();
    const int rows = (int)min((long)CH, t1 - tc);
    for (int r = 0; r < rows; ++r) {
      const float xv = h < H ? to_f32<T>(x[(tc + r) * H + h]) : 0.f;
#pragma unroll
      for (int v = 0; v < MAXV; ++v) acc[v] += s[r * MAXV + v] * xv;
      if (blockIdx.x == 0 && threadIdx.x < MAXV) bsum += s[r * MAXV + threadIdx.x];
    }
  }
  if (h < H) {
#pragma unroll
    for (int v = 0; v < MAXV; ++v)
      if (v < V) atomicAdd(gword + (long)v * H + h, acc[v]);
  }
  if (gbias && blockIdx.x == 0 && (int)threadIdx.x < V) atomicAdd(gbias + threadIdx.x, bsum);
}

// ------------------------------------------------------------------ sampler: one thread per token
__global__ __launch_bounds__(256) void sample_kernel(const float* __restrict__ logits, const float* __restrict__ noise,
                                                      float temperature, int Tn, int V, int64_t* __restrict__ pred,
                                                      float* __restrict__ conf) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= Tn) return;
  const float* l = logits + t * V;
  float mx = -FLT_MAX;
  for (int v = 0; v < V; ++v) mx = fmaxf(mx, l[v] / temperature);
  float sum = 0.f;
  for (int v = 0; v < V; ++v) sum += expf(l[v] / temperature - mx);
  float best = -1.f, bp = 0.f;
  int bi = 0;
  for (int v = 0; v < V; ++v) {
    const float p = expf(l[v] / temperature - mx) / sum;
    const float r = p / noise[t * V + v];
    if (r > best) { best = r; bi = v; bp = p; }
  }
  pred[t] = bi;
  conf[t] = bp;
}

// ------------------------------------------------------------------ per-row top-k commit
// One workgroup per batch row.  rank(i) = #{masked j : conf_j > conf_i or (conf_j == conf_i
// and j < i)}; position i is committed iff rank(i) < k.
__global__ __launch_bounds__(256) void commit_kernel(int64_t* __restrict__ ids, const int64_t* __restrict__ pred,
                                                      const float* __restrict__ conf, int S, double keep_percent,
                                                      int fill_all, int mask_id, int pad_id, int cls_id, int sep_id,
                                                      int32_t* __restrict__ counts) {
  extern __shared__ float c[];  // S floats: confidence of masked positions, -1 elsewhere
  __shared__ int s_nmask, s_nmean, s_commit, s_refill;
  const int b = blockIdx.x;
  int64_t* row = ids + (long)b * S;
  if (threadIdx.x == 0) { s_nmask = 0; s_nmean = 0; s_commit = 0; s_refill = 0; }
  __syncthreads();
  int nm = 0, nmean = 0;
  for (int i = threadIdx.x; i < S; i += blockDim.x) {
    const int64_t id = row[i];
    const bool is_mask = id == mask_id;
    nm += is_mask;
    nmean += (id != pad_id && id != cls_id && id != sep_id);
    c[i] = is_mask ? conf[(long)b * S + i] : -1.f;
  }
  atomicAdd(&s_nmask, nm);
  atomicAdd(&s_nmean, nmean);
  __syncthreads();
  const int nmask = s_nmask;
  if (nmask == 0) {
    if (counts && threadIdx.x == 0) { counts[2 * b] = 0; counts[2 * b + 1] = 0; }
    return;
  }
  int k;
  if (fill_all) {
    k = nmask;
  } else {
    k = (int)(keep_percent * (double)s_nmean);  // Python int(float * int): truncation
    if (k < 1) k = 1;
    if (k > nmask) k = nmask;
  }
  int ncommit = 0, nstill = 0;
  for (int i = threadIdx.x; i < S; i += blockDim.x) {
    const float ci = c[i];
    if (ci < 0.f) continue;
    int rank = 0;
    for (int j = 0; j < S; ++j) {
      const float cj = c[j];
      rank += (cj > ci) || (cj == ci && j < i);
    }
    if (rank < k) {
      const int64_t p = pred[(long)b * S + i];
      row[i] = p;
      ++ncommit;
      nstill += (p == mask_id);
    }
  }
  atomicAdd(&s_commit, ncommit);
  atomicAdd(&s_refill, nstill);
  __syncthreads();
  if (counts && threadIdx.x == 0) {
    counts[2 * b] = s_commit;
    counts[2 * b + 1] = nmask - s_commit + s_refill;
  }
}

// ------------------------------------------------------------------ critic head
// dims: d0 = H, d1 = H/2, d2 = H/4, d3 = H/16, out 1.  stash per batch row (floats):
//   [x d0][a1 d1][a2 d2][a3 d3][v1 d1][v2 d2][v3 d3]   (a = post-ReLU, v = backward vectors)
struct HeadDims { int d[5]; long w[4], b[4]; long stash; };
__host__ __device__ inline HeadDims head_dims(int H) {
  HeadDims h;
  h.d[0] = H; h.d[1] = H / 2; h.d[2] = H / 4; h.d[3] = H / 16; h.d[4] = 1;
  long o = 0;
  for (int i = 0; i < 4; ++i) {
    h.w[i] = o; o += (long)h.d[i + 1] * h.d[i];
    h.b[i] = o; o += h.d[i + 1];
  }
  h.stash = (long)h.d[0] + 2L * (h.d[1] + h.d[2] + h.d[3]);
  return h;
}

// y[o] = act(b[o] + sum_i W[o][i] x[i]): a wave takes four output rows at a time with 16-byte loads, so
// sixteen weight loads are in flight per lane instead of a dependent chain per row (n_in % 4 == 0).
__device__ __forceinline__ void block_matvec(const float* __restrict__ W, const float* __restrict__ bias,
                                             const float* xin, float* yout, int n_out, int n_in, bool relu) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int o0 = wave * 4; o0 < n_out; o0 += nw * 4) {
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = lane * 4; i < n_in; i += 256) {
      const float4 xv = *reinterpret_cast<const float4*>(xin + i);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = o0 + r < n_out ? o0 + r : n_out - 1;
        const float4 wv = *reinterpret_cast<const float4*>(W + (long)o * n_in + i);
        s[r] += (wv.x * xv.x + wv.y * xv.y) + (wv.z * xv.z + wv.w * xv.w);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float t = wave_sum(s[r]);
      if (lane == 0 && o0 + r < n_out) {
        const float v = t + (bias ? bias[o0 + r] : 0.f);
        yout[o0 + r] = relu ? fmaxf(v, 0.f) : v;
      }
    }
  }
}
// y[i] = gate[i] * sum_o W[o][i] v[o]   (transposed mat-vec; threads own columns)
__device__ __forceinline__ void block_matvec_t(const float* __restrict__ W, const float* vin, const float* gate_act,
                                               float* yout, int n_out, int n_in) {
  for (int i = threadIdx.x; i < n_in; i += blockDim.x) {
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // eight loads in flight, not one dependent chain
    int o = 0;
    for (; o + 8 <= n_out; o += 8) {
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += W[(long)(o + e) * n_in + i] * vin[o + e];
    }
    for (; o < n_out; ++o) s[0] += W[(long)o * n_in + i] * vin[o];
    const float t = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    yout[i] = (gate_act == nullptr || gate_act[i] > 0.f) ? t : 0.f;
  }
}

template <typename T>
__global__ __launch_bounds__(1024) void head_fwd_kernel(const T* __restrict__ hidden, long row_stride,
                                                        const float* __restrict__ head, int H, float* __restrict__ scores,
                                                        float* __restrict__ stash) {
  const HeadDims hd = head_dims(H);
  const int b = blockIdx.x;
  float* st = stash + (long)b * hd.stash;
  float* x = st;
  float* a1 = x + hd.d[0];
  float* a2 = a1 + hd.d[1];
  float* a3 = a2 + hd.d[2];
  for (int i = threadIdx.x; i < H; i += blockDim.x) x[i] = to_f32<T>(hidden[(long)b * row_stride + i]);
  __syncthreads();
  block_matvec(head + hd.w[0], head + hd.b[0], x, a1, hd.d[1], hd.d[0], true);
  __syncthreads();
  block_matvec(head + hd.w[1], head + hd.b[1], a1, a2, hd.d[2], hd.d[1], true);
  __syncthreads();
  block_matvec(head + hd.w[2], head + hd.b[2], a2, a3, hd.d[3], hd.d[2], true);
  __syncthreads();
  block_matvec(head + hd.w[3], head + hd.b[3], a3, scores + b, 1, hd.d[3], false);
}

// Backward vectors of w[b] * score_b: v3 = w*W6*gate3, v2 = (W4^T v3)*gate2, v1 = (W2^T v2)*gate1,
// dx = W0^T v1.  With TANGENT, additionally pushes xdot forward through the gated head and
// overwrites the activation slots' roles: t1,t2,t3 are written where a1..a3 were read from a
// second stash (tstash), keeping the primal stash intact.
template <typename T>
__global__ __launch_bounds__(1024) void head_bwd_vec_kernel(const float* __restrict__ head, float* __restrict__ stash,
                                                            const float* __restrict__ w, int H, T* __restrict__ dhidden,
                                                            long row_stride) {
  const HeadDims hd = head_dims(H);
  const int b = blockIdx.x;
  float* st = stash + (long)b * hd.stash;
  float* a1 = st + hd.d[0];
  float* a2 = a1 + hd.d[1];
  float* a3 = a2 + hd.d[2];
  float* v1 = a3 + hd.d[3];
  float* v2 = v1 + hd.d[1];
  float* v3 = v2 + hd.d[2];
  const float wb = w ? w[b] : 1.f;
  for (int i = threadIdx.x; i < hd.d[3]; i += blockDim.x) v3[i] = a3[i] > 0.f ? wb * head[hd.w[3] + i] : 0.f;
  __syncthreads();
  block_matvec_t(head + hd.w[2], v3, a2, v2, hd.d[3], hd.d[2]);
  __syncthreads();
  block_matvec_t(head + hd.w[1], v2, a1, v1, hd.d[2], hd.d[1]);
  __syncthreads();
  if (dhidden) {
    // (eight rows of W0 in flight per lane: one dependent chain of d1 loads is a latency walk of 200 us)
    for (int i = threadIdx.x; i < H; i += blockDim.x) {
      float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const float* w0 = head + hd.w[0] + i;
      int o = 0;
      for (; o + 8 <= hd.d[1]; o += 8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += w0[(long)(o + e) * H] * v1[o + e];
      }
      for (; o < hd.d[1]; ++o) s[0] += w0[(long)o * H] * v1[o];
      dhidden[(long)b * row_stride + i] = from_f32<T>(((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7])));
    }
  }
}

// Tangent through the gated head: t1 = gate1*(W0 xdot), t2 = gate2*(W2 t1), t3 = gate3*(W4 t2),
// sdot = W6 t3.  tstash layout: [xdot d0][t1 d1][t2 d2][t3 d3].
template <typename T>
__global__ __launch_bounds__(1024) void head_jvp_kernel(const float* __restrict__ head, const float* __restrict__ stash,
                                                        const T* __restrict__ hdot, long row_stride, int H,
                                                        float* __restrict__ tstash, float* __restrict__ sdot) {
  const HeadDims hd = head_dims(H);
  const int b = blockIdx.x;
  const float* st = stash + (long)b * hd.stash;
  const float* a1 = st + hd.d[0];
  const float* a2 = a1 + hd.d[1];
  const float* a3 = a2 + hd.d[2];
  float* ts = tstash + (long)b * hd.stash;
  float* xd = ts;
  float* t1 = xd + hd.d[0];
  float* t2 = t1 + hd.d[1];
  float* t3 = t2 + hd.d[2];
  for (int i = threadIdx.x; i < H; i += blockDim.x) xd[i] = to_f32<T>(hdot[(long)b * row_stride + i]);
  __syncthreads();
  block_matvec(head + hd.w[0], nullptr, xd, t1, hd.d[1], hd.d[0], false);
  __syncthreads();
  for (int i = threadIdx.x; i < hd.d[1]; i += blockDim.x) if (!(a1[i] > 0.f)) t1[i] = 0.f;
  __syncthreads();
  block_matvec(head + hd.w[1], nullptr, t1, t2, hd.d[2], hd.d[1], false);
  __syncthreads();
  for (int i = threadIdx.x; i < hd.d[2]; i += blockDim.x) if (!(a2[i] > 0.f)) t2[i] = 0.f;
  __syncthreads();
  block_matvec(head + hd.w[2], nullptr, t2, t3, hd.d[3], hd.d[2], false);
  __syncthreads();
  for (int i = threadIdx.x; i < hd.d[3]; i += blockDim.x) if (!(a3[i] > 0.f)) t3[i] = 0.f;
  __syncthreads();
  float s = 0.f;
  for (int i = threadIdx.x; i < hd.d[3]; i += blockDim.x) s += head[hd.w[3] + i] * t3[i];
  __shared__ float red[16];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0 && sdot) {
    float t = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
    sdot[b] = t;
  }
}

// Weight gradients: gW_l[o][i] += scale * sum_b vec_l[b][o] * act_l[b][i]; gb_l[o] += scale*sum_b vec_l[b][o].
// vec comes from `vstash` (slots v1..v3, and the scalar w for the last layer), act from `astash`
// (slots x, a1, a2, a3).  One thread per weight element, deterministic.
__global__ __launch_bounds__(256) void head_wgrad_kernel(const float* __restrict__ vstash, const float* __restrict__ astash,
                                                          const float* __restrict__ w, int B, int H, float scale,
                                                          int with_bias, float* __restrict__ ghead) {
  const HeadDims hd = head_dims(H);
  const long total = hd.b[3] + 1;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  int l = 0;
  while (l < 3 && e >= hd.w[l + 1]) ++l;
  const bool is_bias = e >= hd.b[l];
  if (is_bias && !with_bias) return;
  const long vo[4] = {hd.d[0] + hd.d[1] + hd.d[2] + hd.d[3], hd.d[0] + 2L * hd.d[1] + hd.d[2] + hd.d[3],
                      hd.d[0] + 2L * hd.d[1] + 2L * hd.d[2] + hd.d[3], -1};
  const long ao[4] = {0, hd.d[0], hd.d[0] + hd.d[1], hd.d[0] + hd.d[1] + hd.d[2]};
  int o, i = 0;
  if (is_bias) { o = (int)(e - hd.b[l]); }
  else { o = (int)((e - hd.w[l]) / hd.d[l]); i = (int)((e - hd.w[l]) % hd.d[l]); }
  float s = 0.f;
  for (int b = 0; b < B; ++b) {
    const float v = l == 3 ? (w ? w[b] : 1.f) : vstash[(long)b * hd.stash + vo[l] + o];
    s += is_bias ? v : v * astash[(long)b * hd.stash + ao[l] + i];
  }
  ghead[e] += scale * s;
}

// ------------------------------------------------------------------ gradient-penalty reductions
template <typename T>
__global__ __launch_bounds__(256) void gp_sumsq_kernel(const T* __restrict__ g, const float* __restrict__ mask, int S, int H,
                                                        float* __restrict__ sumsq, const int* __restrict__ seq_off) {
  const int b = blockIdx.y;
  // sample b: S rows from row b*S, or (ragged) the rows [seq_off[b], seq_off[b+1])
  const long first = seq_off ? seq_off[b] : (long)b * S;
  const long per = (seq_off ? (long)(seq_off[b + 1] - seq_off[b]) : (long)S) * H;
  const long n4 = per / 4;
  float s = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float v[4];
    VecIO<T, 4>::load(g + first * H + i * 4, v);
    const float m = mask ? mask[first + (i * 4) / H] : 1.f;
    s += m * m * (v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
  }
  __shared__ float red[4];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(sumsq + b, red[0] + red[1] + red[2] + red[3]);
}
// norms[b] = sqrt(sumsq[b]) in place; gp = mean((norm - 1)^2)  (single workgroup)
__global__ void gp_sqrt_kernel(float* norms, int B, float* gp_out) {
  float s = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const float n = sqrtf(norms[b]);
    norms[b] = n;
    s += (n - 1.0f) * (n - 1.0f);
  }
  s = wave_sum(s);
  if (threadIdx.x == 0 && gp_out) *gp_out = s / (float)B;
}
template <typename T>
__global__ __launch_bounds__(256) void gp_direction_kernel(const T* __restrict__ g, const float* __restrict__ mask,
                                                            const float* __restrict__ norms, int B, int S, int H,
                                                            const float* __restrict__ scale_dev, T* __restrict__ v,
                                                            float* __restrict__ gp_out, const int* __restrict__ seq_off) {
  const int b = blockIdx.y;
  const long first = seq_off ? seq_off[b] : (long)b * S;
  const long per = (seq_off ? (long)(seq_off[b + 1] - seq_off[b]) : (long)S) * H;
  const long n4 = per / 4;
  const float n = norms[b];
  const float scale = scale_dev ? *scale_dev : 1.0f;
  const float coef = n > 0.f ? scale * (2.0f / (float)B) * (n - 1.0f) / n : 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float x[4];
    VecIO<T, 4>::load(g + first * H + i * 4, x);
    const float m = mask ? mask[first + (i * 4) / H] : 1.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] *= coef * m;
    VecIO<T, 4>::store(v + first * H + i * 4, x);
  }
  if (gp_out && b == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
    float s = 0.f;
    for (int j = 0; j < B; ++j) { const float d = norms[j] - 1.0f; s += d * d; }
    *gp_out = s / (float)B;
  }
}

template <typename T, bool BWD>
__global__ __launch_bounds__(256) void interp_kernel(const T* __restrict__ a, const T* __restrict__ bsrc,
                                                      const float* __restrict__ alpha, T* __restrict__ o1,
                                                      T* __restrict__ o2, long per4) {
  const int b = blockIdx.y;
  const float al = alpha[b];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per4; i += (long)gridDim.x * blockDim.x) {
    const long off = ((long)b * per4 + i) * 4;
    float x[4], y[4];
    VecIO<T, 4>::load(a + off, x);
    if (!BWD) {
      VecIO<T, 4>::load(bsrc + off, y);
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = al * x[e] + (1.0f - al) * y[e];
      VecIO<T, 4>::store(o1 + off, x);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) { y[e] = (1.0f - al) * x[e]; x[e] = al * x[e]; }
      VecIO<T, 4>::store(o1 + off, x);
      VecIO<T, 4>::store(o2 + off, y);
    }
  }
}

// Ragged interpolation: token row t of x belongs to sequence b = seq_id[t] and is its row j = t - off_x[b];
// the real and fake embeddings of that sequence start at rows off_r[b] and off_f[b] of er / ef (each batch
// packed to its own lengths, every one at least the interpolates' length).  One wave per row.
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void interp_ragged_kernel(const T* __restrict__ a, const T* __restrict__ bsrc,
                                                             const float* __restrict__ alpha, T* __restrict__ o1,
                                                             T* __restrict__ o2, const int* __restrict__ seq_id,
                                                             const int* __restrict__ off_x, const int* __restrict__ off_r,
                                                             const int* __restrict__ off_f, int Tn, int H, int accumulate) {
  const int lane = threadIdx.x & 63;
  const long t = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= Tn) return;
  const int b = seq_id[t];
  const long j = t - off_x[b];
  const long rr = (long)off_r[b] + j, rf = (long)off_f[b] + j;
  const float al = alpha[b];
  for (int c = lane * 4; c < H; c += 256) {
    float x[4], y[4];
    if (!BWD) {
      VecIO<T, 4>::load(a + rr * H + c, x);
      VecIO<T, 4>::load(bsrc + rf * H + c, y);
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = al * x[e] + (1.0f - al) * y[e];
      VecIO<T, 4>::store(o1 + t * H + c, x);
    } else {
      VecIO<T, 4>::load(a + t * H + c, x);
#pragma unroll
      for (int e = 0; e < 4; ++e) { y[e] = (1.0f - al) * x[e]; x[e] = al * x[e]; }
      if (accumulate) {
        float p[4], q[4];
        VecIO<T, 4>::load(o1 + rr * H + c, p);
        VecIO<T, 4>::load(o2 + rf * H + c, q);
#pragma unroll
        for (int e = 0; e < 4; ++e) { x[e] += p[e]; y[e] += q[e]; }
      }
      VecIO<T, 4>::store(o1 + rr * H + c, x);
      VecIO<T, 4>::store(o2 + rf * H + c, y);
    }
  }
}

// Scatter of the embedding-sum gradient into the word and position tables (fp32 atomics);
// the token-type row receives the column sum through pg_colsum_acc.
template <typename T>
__global__ __launch_bounds__(256) void embed_scatter_kernel(const T* __restrict__ dsum, const int64_t* __restrict__ ids,
                                                             float* __restrict__ gword, float* __restrict__ gpos, int S,
                                                             int H, long n_tokens, const int* __restrict__ pos_ids) {
  const long total = n_tokens * H;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long t = e / H;
    const int c = (int)(e % H);
    const float v = to_f32<T>(dsum[e]);
    if (gword) atomicAdd(gword + ids[t] * H + c, v);
    if (gpos) atomicAdd(gpos + (pos_ids ? (long)pos_ids[t] : t % S) * H + c, v);
  }
}

// ------------------------------------------------------------------ optimizer / casts
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                     float* __restrict__ v, bf16_t* __restrict__ shadow, long n, float lr,
                                                     float b1, float b2, float eps, float wd, float bc1, float sqrt_bc2,
                                                     float gscale) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gi = g[i] * gscale;
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = m[i] * b1 + (1.0f - b1) * gi;
    const float vi = v[i] * b2 + (1.0f - b2) * gi * gi;
    const float denom = sqrtf(vi) / sqrt_bc2 + eps;
    pi = pi - (lr / bc1) * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
    if (shadow) shadow[i] = f32_to_bf16_bits(pi);
  }
}
// Four parameters per lane: 16-byte accesses to p, g, m, v and 8-byte stores of the bf16 copy.
__global__ __launch_bounds__(256) void adamw_vec4_kernel(float4* __restrict__ p, const float4* __restrict__ g,
                                                          float4* __restrict__ m, float4* __restrict__ v,
                                                          uint2* __restrict__ shadow, long n4, float lr, float b1, float b2,
                                                          float eps, float wd, float bc1, float sqrt_bc2, float gscale) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    // read once, written once, 12.6 GB per critic update: nontemporal accesses leave the caches to the passes'
    // operands (834 / 839 against 840 / 859 ms per GAN step)
    typedef float nt4 __attribute__((ext_vector_type(4)));
    const nt4 gq = __builtin_nontemporal_load(reinterpret_cast<const nt4*>(g) + i);
    const nt4 pq = __builtin_nontemporal_load(reinterpret_cast<const nt4*>(p) + i);
    const nt4 mq = __builtin_nontemporal_load(reinterpret_cast<const nt4*>(m) + i);
    const nt4 vq = __builtin_nontemporal_load(reinterpret_cast<const nt4*>(v) + i);
    const float4 g4 = make_float4(gq[0], gq[1], gq[2], gq[3]);
    float4 p4 = make_float4(pq[0], pq[1], pq[2], pq[3]), m4 = make_float4(mq[0], mq[1], mq[2], mq[3]),
           v4 = make_float4(vq[0], vq[1], vq[2], vq[3]);
    const float gs[4] = {g4.x, g4.y, g4.z, g4.w};
    float ps[4] = {p4.x, p4.y, p4.z, p4.w}, ms[4] = {m4.x, m4.y, m4.z, m4.w}, vs[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gi = gs[e] * gscale;
      float pi = ps[e] * (1.0f - lr * wd);
      const float mi = ms[e] * b1 + (1.0f - b1) * gi;
      const float vi = vs[e] * b2 + (1.0f - b2) * gi * gi;
      const float denom = sqrtf(vi) / sqrt_bc2 + eps;
      pi = pi - (lr / bc1) * (mi / denom);
      ps[e] = pi; ms[e] = mi; vs[e] = vi;
    }
    __builtin_nontemporal_store((nt4){ps[0], ps[1], ps[2], ps[3]}, reinterpret_cast<nt4*>(p) + i);
    __builtin_nontemporal_store((nt4){ms[0], ms[1], ms[2], ms[3]}, reinterpret_cast<nt4*>(m) + i);
    __builtin_nontemporal_store((nt4){vs[0], vs[1], vs[2], vs[3]}, reinterpret_cast<nt4*>(v) + i);
    if (shadow) {
      uint2 o;
      o.x = (unsigned)f32_to_bf16_bits(ps[0]) | ((unsigned)f32_to_bf16_bits(ps[1]) << 16);
      o.y = (unsigned)f32_to_bf16_bits(ps[2]) | ((unsigned)f32_to_bf16_bits(ps[3]) << 16);
      shadow[i] = o;
    }
  }
}
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ s, bf16_t* __restrict__ d, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    d[i] = f32_to_bf16_bits(s[i]);
}
__global__ __launch_bounds__(256) void cast_f32_kernel(const bf16_t* __restrict__ s, float* __restrict__ d, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    d[i] = bf16_bits_to_f32(s[i]);
}
__global__ __launch_bounds__(256) void round_bf16_kernel(float* __restrict__ p, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    p[i] = bf16_bits_to_f32(f32_to_bf16_bits(p[i]));
}
__global__ __launch_bounds__(256) void scale_kernel(float* __restrict__ p, float s, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] *= s;
}

inline int ew_blocks(long n) { long b = (n + 255) / 256; return (int)(b < 8192 ? (b > 0 ? b : 1) : 8192); }

}  // namespace

extern "C" int64_t pg_head_total(int hidden) { return head_dims(hidden).b[3] + 1; }
extern "C" int64_t pg_head_offset(int hidden, int li, int is_bias) {
  if (li < 0 || li > 3) return -1;
  const HeadDims h = head_dims(hidden);
  return is_bias ? h.b[li] : h.w[li];
}
extern "C" int64_t pg_head_stash_floats(int hidden) { return head_dims(hidden).stash; }

extern "C" int pg_mlm_decode(int dtype, const void* t, const float* word, const float* bias, float* logits, int T, int H,
                             int V, void* stream) {
  if (!t || !word || !bias || !logits || T <= 0 || V <= 0 || V > 64) return PG_ERR_ARG;
  const size_t lds = (size_t)V * H * sizeof(float);
  if (lds > 160 * 1024 - 512) return PG_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  int blocks = (T + 15) / 16;
  if (blocks > 256) blocks = 256;
#define DECODE_CASE(TT, VWv, NCHv)                                                                                    \
  {                                                                                                                   \
    auto kfn = mlm_decode_kernel<TT, VWv, NCHv>;                                                                      \
    if (lds > 64 * 1024 &&                                                                                            \
        hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)    \
      return PG_ERR_LAUNCH;                                                                                           \
    hipLaunchKernelGGL(kfn, dim3(blocks), dim3(1024), lds, st, (const TT*)t, word, bias, logits, T, H, V);           \
  }
#define DECODE_DISPATCH(TT)                                   \
  switch (H) {                                                \
    case 64: DECODE_CASE(TT, 1, 1) break;                     \
    case 128: DECODE_CASE(TT, 1, 2) break;                    \
    case 256: DECODE_CASE(TT, 4, 1) break;                    \
    case 512: DECODE_CASE(TT, 4, 2) break;                    \
    case 1024: DECODE_CASE(TT, 4, 4) break;                   \
    default: return PG_ERR_UNSUPPORTED;                       \
  }
  if (dtype == PG_F32) { DECODE_DISPATCH(float) } else { DECODE_DISPATCH(bf16_t) }
  return check_launch();
}

extern "C" int pg_mlm_decode_bwd(int dtype, const float* dlogits, const void* t, const float* word, void* dt, float* gword,
                                 float* gbias, int T, int H, int V, void* stream) {
  if (!dlogits || T <= 0 || H <= 0 || V <= 0 || V > MAXV) return PG_ERR_ARG;
  if ((dt && !word) || ((gword || gbias) && !t) || (gbias && !gword)) return PG_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dt) {
    const int blocks = (T + 15) / 16;
    if (dtype == PG_F32)
      hipLaunchKernelGGL(mlm_decode_bwd_x_kernel<float>, dim3(blocks), dim3(256), 0, st, dlogits, word, (float*)dt, T, H, V);
    else
      hipLaunchKernelGGL(mlm_decode_bwd_x_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, dlogits, word, (bf16_t*)dt, T, H, V);
  }
  if (gword) {
    int spans = (T + 63) / 64;
    if (spans > 128) spans = 128;
    const int span = ((T + spans - 1) / spans + 15) / 16 * 16;
    spans = (T + span - 1) / span;
    const dim3 grid((H + 255) / 256, spans);
    if (dtype == PG_F32)
      hipLaunchKernelGGL(mlm_decode_bwd_w_kernel<float>, grid, dim3(256), 0, st, dlogits, (const float*)t, gword, gbias, T, H, V,
                         span);
    else
      hipLaunchKernelGGL(mlm_decode_bwd_w_kernel<bf16_t>, grid, dim3(256), 0, st, dlogits, (const bf16_t*)t, gword, gbias, T, H,
                         V, span);
  }
  return check_launch();
}

extern "C" int pg_sample_tokens(const float* logits, const float* noise, float temperature, int T, int V, int64_t* pred,
                                float* conf, void* stream) {
  if (!logits || !noise || !pred || !conf || T <= 0 || V <= 0 || V > MAXV || !(temperature > 0.f)) return PG_ERR_ARG;
  hipLaunchKernelGGL(sample_kernel, dim3((T + 255) / 256), dim3(256), 0, (hipStream_t)stream, logits, noise, temperature,
                     T, V, pred, conf);
  return check_launch();
}

extern "C" int pg_commit_topk(int64_t* ids, const int64_t* pred, const float* conf, int B, int S, double keep_percent,
                              int fill_all, int mask_id, int pad_id, int cls_id, int sep_id, int32_t* counts,
                              void* stream) {
  if (!ids || !pred || !conf || B <= 0 || S <= 0) return PG_ERR_ARG;
  if ((size_t)S * sizeof(float) > 60 * 1024) return PG_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(commit_kernel, dim3(B), dim3(256), (size_t)S * sizeof(float), (hipStream_t)stream, ids, pred, conf, S,
                     keep_percent, fill_all, mask_id, pad_id, cls_id, sep_id, counts);
  return check_launch();
}

extern "C" int pg_head_forward(int dtype, const void* hidden, int64_t row_stride, const float* head, int B, int H,
                               float* scores, float* stash, void* stream) {
  if (!hidden || !head || !scores || !stash || B <= 0 || H < 64 || (H % 64)) return PG_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == PG_F32)
    hipLaunchKernelGGL(head_fwd_kernel<float>, dim3(B), dim3(1024), 0, st, (const float*)hidden, (long)row_stride, head, H,
                       scores, stash);
  else
    hipLaunchKernelGGL(head_fwd_kernel<bf16_t>, dim3(B), dim3(1024), 0, st, (const bf16_t*)hidden, (long)row_stride, head,
                       H, scores, stash);
  return check_launch();
}

extern "C" int pg_head_backward(int dtype, const float* head, float* stash, const float* w, int B, int H, float* ghead,
                                void* dhidden, int64_t row_stride, void* stream) {
  if (!head || !stash || B <= 0 || H < 64 || (H % 64)) return PG_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == PG_F32)
    hipLaunchKernelGGL(head_bwd_vec_kernel<float>, dim3(B), dim3(1024), 0, st, head, stash, w, H, (float*)dhidden,
                       (long)row_stride);
  else
    hipLaunchKernelGGL(head_bwd_vec_kernel<bf16_t>, dim3(B), dim3(1024), 0, st, head, stash, w, H, (bf16_t*)dhidden,
                       (long)row_stride);
  if (hipGetLastError() != hipSuccess) return PG_ERR_LAUNCH;
  if (ghead) {
    const long total = pg_head_total(H);
    hipLaunchKernelGGL(head_wgrad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, stash, stash, w, B, H,
                       1.0f, 1, ghead);
  }
  return check_launch();
}

extern "C" int pg_head_jvp_grad(int dtype, const float* head, float* stash, const void* hdot, int64_t row_stride, int B,
                                int H, float scale, float* ghead, float* tstash, float* sdot_out, void* stream) {
  if (!head || !stash || !hdot || !tstash || B <= 0 || H < 64 || (H % 64)) return PG_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  // unit-weight backward vectors (the adjoints of the tangents) into the primal stash's v slots
  hipLaunchKernelGGL(head_bwd_vec_kernel<float>, dim3(B), dim3(1024), 0, st, head, stash, (const float*)nullptr, H,
                     (float*)nullptr, 0L);
  if (dtype == PG_F32)
    hipLaunchKernelGGL(head_jvp_kernel<float>, dim3(B), dim3(1024), 0, st, head, stash, (const float*)hdot,
                       (long)row_stride, H, tstash, sdot_out);
  else
    hipLaunchKernelGGL(head_jvp_kernel<bf16_t>, dim3(B), dim3(1024), 0, st, head, stash, (const bf16_t*)hdot,
                       (long)row_stride, H, tstash, sdot_out);
  if (hipGetLastError() != hipSuccess) return PG_ERR_LAUNCH;
  if (ghead) {
    const long total = pg_head_total(H);
    hipLaunchKernelGGL(head_wgrad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, stash, tstash,
                       (const float*)nullptr, B, H, scale, 0, ghead);
  }
  return check_launch();
}

static int gp_norms_launch(int dtype, const void* g, const float* mask, const int32_t* seq_off, int B, int S, int H,
                           float* norms, float* gp_out, void* stream) {
  if (!g || !norms || B <= 0 || S <= 0 || (H % 4)) return PG_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(norms, 0, sizeof(float) * B, st) != hipSuccess) return PG_ERR_LAUNCH;
  const long n4 = (long)S * H / 4;
  int bx = (int)((n4 + 255) / 256);
  if (bx > 256) bx = 256;
  dim3 grid(bx, B);
  if (dtype == PG_F32)
    hipLaunchKernelGGL(gp_sumsq_kernel<float>, grid, dim3(256), 0, st, (const float*)g, mask, S, H, norms, seq_off);
  else
    hipLaunchKernelGGL(gp_sumsq_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)g, mask, S, H, norms, seq_off);
  hipLaunchKernelGGL(gp_sqrt_kernel, dim3(1), dim3(64), 0, st, norms, B, gp_out);
  return check_launch();
}
extern "C" int pg_gp_norms(int dtype, const void* g, const float* mask, int B, int S, int H, float* norms, float* gp_out,
                           void* stream) {
  return gp_norms_launch(dtype, g, mask, nullptr, B, S, H, norms, gp_out, stream);
}
extern "C" int pg_gp_norms_ragged(int dtype, const void* g, const float* mask, const int32_t* seq_off, int B, int max_len,
                                  int H, float* norms, float* gp_out, void* stream) {
  if (!seq_off) return PG_ERR_ARG;
  return gp_norms_launch(dtype, g, mask, seq_off, B, max_len, H, norms, gp_out, stream);
}

static int gp_direction_launch(int dtype, const void* g, const float* mask, const float* norms, const int32_t* seq_off, int B,
                               int S, int H, const float* scale_dev, void* v, float* gp_out, void* stream) {
  if (!g || !norms || !v || B <= 0 || S <= 0 || (H % 4)) return PG_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const long n4 = (long)S * H / 4;
  int bx = (int)((n4 + 255) / 256);
  if (bx > 256) bx = 256;
  dim3 grid(bx, B);
  if (dtype == PG_F32)
    hipLaunchKernelGGL(gp_direction_kernel<float>, grid, dim3(256), 0, st, (const float*)g, mask, norms, B, S, H,
                       scale_dev, (float*)v, gp_out, seq_off);
  else
    hipLaunchKernelGGL(gp_direction_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)g, mask, norms, B, S, H,
                       scale_dev, (bf16_t*)v, gp_out, seq_off);
  return check_launch();
}
extern "C" int pg_gp_direction(int dtype, const void* g, const float* mask, const float* norms, int B, int S, int H,
                               const float* scale_dev, void* v, float* gp_out, void* stream) {
  return gp_direction_launch(dtype, g, mask, norms, nullptr, B, S, H, scale_dev, v, gp_out, stream);
}
extern "C" int pg_gp_direction_ragged(int dtype, const void* g, const float* mask, const float* norms, const int32_t* seq_off,
                                      int B, int max_len, int H, const float* scale_dev, void* v, float* gp_out, void* stream) {
  if (!seq_off) return PG_ERR_ARG;
  return gp_direction_launch(dtype, g, mask, norms, seq_off, B, max_len, H, scale_dev, v, gp_out, stream);
}

extern "C" int pg_interpolate_ragged(int dtype, const void* er, const void* ef, const float* alpha, void* x,
                                     const int32_t* seq_id, const int32_t* off_x, const int32_t* off_r, const int32_t* off_f,
                                     int T, int H, void* stream) {
  if (!er || !ef || !alpha || !x || !seq_id || !off_x || !off_r || !off_f || T <= 0 || H <= 0 || (H & 3)) return PG_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((T + 3) / 4);
  if (dtype == PG_F32)
    hipLaunchKernelGGL((interp_ragged_kernel<float, false>), grid, dim3(256), 0, st, (const float*)er, (const float*)ef, alpha,
                       (float*)x, (float*)nullptr, seq_id, off_x, off_r, off_f, T, H, 0);
  else
    hipLaunchKernelGGL((interp_ragged_kernel<bf16_t, false>), grid, dim3(256), 0, st, (const bf16_t*)er, (const bf16_t*)ef,
                       alpha, (bf16_t*)x, (bf16_t*)nullptr, seq_id, off_x, off_r, off_f, T, H, 0);
  return check_launch();
}
extern "C" int pg_interpolate_bwd_ragged(int dtype, const void* dx, const float* alpha, void* der, void* def,
                                         const int32_t* seq_id, const int32_t* off_x, const int32_t* off_r,
                                         const int32_t* off_f, int T, int H, int accumulate, void* stream) {
  if (!dx || !alpha || !der || !def || !seq_id || !off_x || !off_r || !off_f || T <= 0 || H <= 0 || (H & 3)) return PG_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((T + 3) / 4);
  if (dtype == PG_F32)
    hipLaunchKernelGGL((interp_ragged_kernel<float, true>), grid, dim3(256), 0, st, (const float*)dx, (const float*)nullptr,
                       alpha, (float*)der, (float*)def, seq_id, off_x, off_r, off_f, T, H, accumulate);
  else
    hipLaunchKernelGGL((interp_ragged_kernel<bf16_t, true>), grid, dim3(256), 0, st, (const bf16_t*)dx, (const bf16_t*)nullptr,
                       alpha, (bf16_t*)der, (bf16_t*)def, seq_id, off_x, off_r, off_f, T, H, accumulate);
  return check_launch();
}

extern "C" int pg_interpolate(int dtype, const void* er, const void* ef, const float* alpha, void* x, int B,
                              int64_t per_sample, void* stream) {
  if (!er || !ef || !alpha || !x || B <= 0 || per_sample <= 0 || (per_sample & 3)) return PG_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const long per4 = per_sample / 4;
  int bx = (int)((per4 + 255) / 256);
  if (bx > 512) bx = 512;
  dim3 grid(bx, B);
  if (dtype == PG_F32)
    hipLaunchKernelGGL((interp_kernel<float, false>), grid, dim3(256), 0, st, (const float*)er, (const float*)ef, alpha,
                       (float*)x, (float*)nullptr, per4);
  else
    hipLaunchKernelGGL((interp_kernel<bf16_t, false>), grid, dim3(256), 0, st, (const bf16_t*)er, (const bf16_t*)ef, alpha,
                       (bf16_t*)x, (bf16_t*)nullptr, per4);
  return check_launch();
}
extern "C" int pg_interpolate_bwd(int dtype, const void* dx, const float* alpha, void* der, void* def, int B,
                                  int64_t per_sample, void* stream) {
  if (!dx || !alpha || !der || !def || B <= 0 || per_sample <= 0 || (per_sample & 3)) return PG_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const long per4 = per_sample / 4;
  int bx = (int)((per4 + 255) / 256);
  if (bx > 512) bx = 512;
  dim3 grid(bx, B);
  if (dtype == PG_F32)
    hipLaunchKernelGGL((interp_kernel<float, true>), grid, dim3(256), 0, st, (const float*)dx, (const float*)nullptr, alpha,
                       (float*)der, (float*)def, per4);
  else
    hipLaunchKernelGGL((interp_kernel<bf16_t, true>), grid, dim3(256), 0, st, (const bf16_t*)dx, (const bf16_t*)nullptr,
                       alpha, (bf16_t*)der, (bf16_t*)def, per4);
  return check_launch();
}

static int embed_scatter_launch(int dtype, const void* dsum, const int64_t* ids, const int32_t* pos_ids, float* gword,
                                float* gpos, float* gtype0, int B, int S, int H, void* stream) {
  if (!dsum || !ids || B <= 0 || S <= 0 || H <= 0) return PG_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const long n_tokens = (long)B * S;
  if (gword || gpos) {
    if (dtype == PG_F32)
      hipLaunchKernelGGL(embed_scatter_kernel<float>, dim3(ew_blocks(n_tokens * H)), dim3(256), 0, st, (const float*)dsum,
                         ids, gword, gpos, S, H, n_tokens, pos_ids);
    else
      hipLaunchKernelGGL(embed_scatter_kernel<bf16_t>, dim3(ew_blocks(n_tokens * H)), dim3(256), 0, st, (const bf16_t*)dsum,
                         ids, gword, gpos, S, H, n_tokens, pos_ids);
    if (hipGetLastError() != hipSuccess) return PG_ERR_LAUNCH;
  }
  if (gtype0) return pg_colsum_acc(dtype, dsum, H, (int)n_tokens, H, gtype0, stream);
  return PG_OK;
}
extern "C" int pg_embed_scatter(int dtype, const void* dsum, const int64_t* ids, float* gword, float* gpos, float* gtype0,
                                int B, int S, int H, void* stream) {
  return embed_scatter_launch(dtype, dsum, ids, nullptr, gword, gpos, gtype0, B, S, H, stream);
}
extern "C" int pg_embed_scatter_packed(int dtype, const void* dsum, const int64_t* ids, const int32_t* pos_ids, float* gword,
                                       float* gpos, float* gtype0, int T, int H, void* stream) {
  if (!pos_ids) return PG_ERR_ARG;
  return embed_scatter_launch(dtype, dsum, ids, pos_ids, gword, gpos, gtype0, 1, T, H, stream);
}

extern "C" int pg_adamw(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr, float beta1,
                        float beta2, float eps, float wd, int step, float grad_scale, void* stream) {
  if (!p || !g || !m || !v || n <= 0 || step < 1) return PG_ERR_ARG;
  const float bc1 = 1.0f - (float)pow((double)beta1, step);
  const float sqrt_bc2 = (float)sqrt(1.0 - pow((double)beta2, step));
  hipStream_t st = (hipStream_t)stream;
  const bool aligned = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16 == 0) && ((uintptr_t)shadow_bf16 % 8 == 0);
  const long n4 = aligned ? n / 4 : 0;
  if (n4 > 0)
    hipLaunchKernelGGL(adamw_vec4_kernel, dim3(ew_blocks(n4)), dim3(256), 0, st, (float4*)p, (const float4*)g, (float4*)m,
                       (float4*)v, (uint2*)shadow_bf16, n4, lr, beta1, beta2, eps, wd, bc1, sqrt_bc2, grad_scale);
  const long done = n4 * 4;
  if (done < n)
    hipLaunchKernelGGL(adamw_kernel, dim3(ew_blocks(n - done)), dim3(256), 0, st, p + done, g + done, m + done, v + done,
                       shadow_bf16 ? (bf16_t*)shadow_bf16 + done : nullptr, (long)(n - done), lr, beta1, beta2, eps, wd, bc1,
                       sqrt_bc2, grad_scale);
  return check_launch();
}
extern "C" int pg_cast_bf16(const float* src, void* dst, int64_t n, void* stream) {
  if (!src || !dst || n <= 0) return PG_ERR_ARG;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, (long)n);
  return check_launch();
}
extern "C" int pg_cast_f32(const void* src, float* dst, int64_t n, void* stream) {
  if (!src || !dst || n <= 0) return PG_ERR_ARG;
  hipLaunchKernelGGL(cast_f32_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, dst,
                     (long)n);
  return check_launch();
}
extern "C" int pg_round_bf16_f32(float* p, int64_t n, void* stream) {
  if (!p || n <= 0) return PG_ERR_ARG;
  hipLaunchKernelGGL(round_bf16_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, p, (long)n);
  return check_launch();
}
extern "C" int pg_scale_f32(float* p, float s, int64_t n, void* stream) {
  if (!p || n <= 0) return PG_ERR_ARG;
  hipLaunchKernelGGL(scale_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, p, s, (long)n);
  return check_launch();
}

extern "C" const char* pg_strerror(int code) {
  switch (code) {
    case PG_OK: return "ok";
    case PG_ERR_ARG: return "invalid argument";
    case PG_ERR_ALIGN: return "pointer, leading dimension or stride breaks the 16-byte alignment rule";
    case PG_ERR_LAUNCH: return "kernel launch failed";
    case PG_ERR_UNSUPPORTED: return "shape not supported by this kernel";
    case PG_ERR_WORKSPACE: return "workspace missing or too small";
    case PG_ERR_SCRATCH: return "size exceeds the capacity of the library's static device scratch";
    default: return "unknown error";
  }
}
extern "C" int pg_abi_version(void) { return 1; }
