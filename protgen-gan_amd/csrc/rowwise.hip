// Row-wise kernels of the ProtBERT path: embeddings + LayerNorm, LayerNorm forward /
// backward / tangent / second-order reverse, softmax over materialised attention scores
// (forward, backward, tangent, second-order reverse), GELU second-order reverse, bias
// gradient column sums.  All are HBM-bound: one 64-lane wavefront owns one row, every
// global access is a 4-element vector (8 B bf16 / 16 B fp32) with consecutive lanes on
// consecutive chunks, statistics are fp32 wave reductions (no LDS, no barriers).
#include <float.h>
#include "pg_common.h"
#include "protgen_hip.h"

namespace {

constexpr int ROWS_PER_BLOCK = 4;  // 256 threads

// Row ownership: element index of (chunk i, lane, e) = (i*64 + lane)*VW + e.
template <typename T, int VW, int NCH>
__device__ __forceinline__ void row_load(const T* row, int lane, float (&v)[VW * NCH]) {
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    if constexpr (VW == 4) {
      VecIO<T, 4>::load(row + (i * 64 + lane) * 4, &v[i * 4]);
    } else {
      v[i] = to_f32<T>(row[i * 64 + lane]);
    }
  }
}
template <typename T, int VW, int NCH>
__device__ __forceinline__ void row_store(T* row, int lane, const float (&v)[VW * NCH]) {
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    if constexpr (VW == 4) {
      VecIO<T, 4>::store(row + (i * 64 + lane) * 4, &v[i * 4]);
    } else {
      row[i * 64 + lane] = from_f32<T>(v[i]);
    }
  }
}
template <int N> __device__ __forceinline__ float arr_sum(const float (&v)[N]) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < N; ++i) s += v[i];
  return s;
}

template <int N>
__device__ __forceinline__ void ln_stats(const float (&x)[N], int H, float eps, float& mean, float& rstd) {
  mean = wave_sum(arr_sum(x)) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < N; ++i) { const float d = x[i] - mean; q += d * d; }
  const float var = wave_sum(q) / (float)H;
  rstd = 1.0f / sqrtf(var + eps);
}

// ------------------------------------------------------------------ embeddings + LN
template <typename T, int VW, int NCH>
__global__ __launch_bounds__(256) void embed_ln_kernel(const int64_t* __restrict__ ids, const float* __restrict__ word,
                                                        const float* __restrict__ pos, const float* __restrict__ type0,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float eps, T* __restrict__ out, T* __restrict__ sum_out, int B,
                                                        int S, int H, const int* __restrict__ pos_ids) {
  constexpr int N = VW * NCH;
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= (long)B * S) return;
  const int s = pos_ids ? pos_ids[row] : (int)(row % S);
  const long id = ids[row];
  float w[N], p[N], t[N], g[N], b[N], x[N];
  row_load<float, VW, NCH>(word + id * H, lane, w);
  row_load<float, VW, NCH>(type0, lane, t);
  row_load<float, VW, NCH>(pos + (long)s * H, lane, p);
  row_load<float, VW, NCH>(gamma, lane, g);
  row_load<float, VW, NCH>(beta, lane, b);
#pragma unroll
  for (int i = 0; i < N; ++i) x[i] = (w[i] + t[i]) + p[i];
  if (sum_out) {
    row_store<T, VW, NCH>(sum_out + row * H, lane, x);
    // the backward recomputes statistics from the stored (possibly rounded) sum
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = to_f32<T>(from_f32<T>(x[i]));
  }
  float mean, rstd;
  ln_stats<N>(x, H, eps, mean, rstd);
#pragma unroll
  for (int i = 0; i < N; ++i) x[i] = (x[i] - mean) * rstd * g[i] + b[i];
  row_store<T, VW, NCH>(out + row * H, lane, x);
}

// ------------------------------------------------------------------ LN forward
template <typename T, int VW, int NCH>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ xin, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float eps, T* __restrict__ y,
                                                      int rows, int H) {
  constexpr int N = VW * NCH;
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= rows) return;
  float x[N], g[N], b[N];
  row_load<T, VW, NCH>(xin + row * H, lane, x);
  row_load<float, VW, NCH>(gamma, lane, g);
  row_load<float, VW, NCH>(beta, lane, b);
  float mean, rstd;
  ln_stats<N>(x, H, eps, mean, rstd);
#pragma unroll
  for (int i = 0; i < N; ++i) x[i] = (x[i] - mean) * rstd * g[i] + b[i];
  row_store<T, VW, NCH>(y + row * H, lane, x);
}

// ------------------------------------------------------------------ LN tangent
template <typename T, int VW, int NCH>
__global__ __launch_bounds__(256) void ln_jvp_kernel(const T* __restrict__ xin, const T* __restrict__ xdot,
                                                      const float* __restrict__ gamma, float eps, T* __restrict__ ydot,
                                                      int rows, int H) {
  constexpr int N = VW * NCH;
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= rows) return;
  float x[N], d[N], g[N];
  row_load<T, VW, NCH>(xin + row * H, lane, x);
  row_load<T, VW, NCH>(xdot + row * H, lane, d);
  row_load<float, VW, NCH>(gamma, lane, g);
  float mean, rstd;
  ln_stats<N>(x, H, eps, mean, rstd);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    x[i] = (x[i] - mean) * rstd;
    s1 += d[i];
    s2 += d[i] * x[i];
  }
  const float md = wave_sum(s1) / (float)H, b = wave_sum(s2) / (float)H;
#pragma unroll
  for (int i = 0; i < N; ++i) d[i] = g[i] * rstd * (d[i] - md - x[i] * b);
  row_store<T, VW, NCH>(ydot + row * H, lane, d);
}

// ------------------------------------------------------------------ LN backward (first and second order)
// Each wave walks rows wave_id, wave_id + n_waves, ... and keeps its dgamma/dbeta partial
// sums (and the column sums of its output rows: the bias gradient of the Linear that feeds
// this LayerNorm) in registers; partials go to `partial[wave][3][H]` and are reduced by
// ln_param_reduce_kernel.
template <typename T, int VW, int NCH, bool SECOND>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ xin, const T* __restrict__ xdot,
                                                      const T* __restrict__ gy, const T* __restrict__ ybar,
                                                      const float* __restrict__ gamma, float eps, T* __restrict__ xbar,
                                                      float* __restrict__ partial, int rows, int H, int n_waves) {
  constexpr int N = VW * NCH;
  __shared__ float red[ROWS_PER_BLOCK - 1][64 * N];  // waves 1.. park one partial array at a time
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const int wave = blockIdx.x * ROWS_PER_BLOCK + wib;
  float g[N], dg[N], db[N], dxs[N];
  row_load<float, VW, NCH>(gamma, lane, g);
#pragma unroll
  for (int i = 0; i < N; ++i) { dg[i] = 0.f; db[i] = 0.f; dxs[i] = 0.f; }
  const float invH = 1.0f / (float)H;
  for (long row = wave; row < rows; row += n_waves) {  // (waves past n_waves own no row)
    float x[N], out[N];
    row_load<T, VW, NCH>(xin + row * H, lane, x);
    float mean, rstd;
    ln_stats<N>(x, H, eps, mean, rstd);
#pragma unroll
    for (int i = 0; i < N; ++i) { x[i] = (x[i] - mean) * rstd; out[i] = 0.f; }
    if (!SECOND || ybar) {
      float y[N];
      row_load<T, VW, NCH>((SECOND ? ybar : gy) + row * H, lane, y);
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < N; ++i) {
        dg[i] += y[i] * x[i];
        db[i] += y[i];
        y[i] *= g[i];
        s1 += y[i];
        s2 += y[i] * x[i];
      }
      const float m1 = wave_sum(s1) * invH, m2 = wave_sum(s2) * invH;
#pragma unroll
      for (int i = 0; i < N; ++i) out[i] = rstd * (y[i] - m1 - x[i] * m2);
    }
    if (SECOND) {
      float d[N], w[N], gyv[N];
      row_load<T, VW, NCH>(xdot + row * H, lane, d);
      row_load<T, VW, NCH>(gy + row * H, lane, gyv);
      float sd = 0.f, sdx = 0.f, sw = 0.f, swx = 0.f;
#pragma unroll
      for (int i = 0; i < N; ++i) {
        w[i] = gyv[i] * g[i];
        sd += d[i];
        sdx += d[i] * x[i];
        sw += w[i];
        swx += w[i] * x[i];
      }
      const float md = wave_sum(sd) * invH, b = wave_sum(sdx) * invH;
      const float mw = wave_sum(sw) * invH, a = wave_sum(swx) * invH;
      float s0 = 0.f;
#pragma unroll
      for (int i = 0; i < N; ++i) {
        d[i] = d[i] - md - x[i] * b;   // Proj(xdot)
        s0 += w[i] * d[i];
      }
      const float a0 = wave_sum(s0) * invH;
      const float r2 = rstd * rstd;
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const float pw = w[i] - mw - x[i] * a;  // Proj(w)
        out[i] -= r2 * (x[i] * a0 + b * pw + a * d[i]);
        dg[i] += gyv[i] * rstd * d[i];  // d<gy, ydot>/dgamma

      }
    }
    row_store<T, VW, NCH>(xbar + row * H, lane, out);
#pragma unroll
    for (int i = 0; i < N; ++i) dxs[i] += out[i];
  }
  if (partial) {  // one partial row per workgroup: the four waves combine through LDS first
    float* pg = partial + (long)blockIdx.x * 3 * H;
#pragma unroll
    for (int arr = 0; arr < 3; ++arr) {
      float(&v)[N] = arr == 0 ? dg : (arr == 1 ? db : dxs);
      if (wib > 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) red[wib - 1][i * 64 + lane] = v[i];
      }
      __syncthreads();
      if (wib == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
#pragma unroll
          for (int w = 0; w < ROWS_PER_BLOCK - 1; ++w) v[i] += red[w][i * 64 + lane];
        }
        row_store<float, VW, NCH>(pg + arr * H, lane, v);
      }
      __syncthreads();
    }
  }
}

// grid (ceil(3H/64), row_chunks): thread (c = tid&63, rg = tid>>6) sums partial rows rg, rg+4, ...
// of its chunk for column c; the four row groups combine through LDS; one atomic per column.
__global__ __launch_bounds__(256) void ln_param_reduce_kernel(const float* __restrict__ partial, int n_waves, int H,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              float* __restrict__ dxsum, int rows_per_chunk) {
  __shared__ float red[256];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;  // column in [0, 3H)
  float* dst = c < H ? dgamma : (c < 2 * H ? dbeta : dxsum);
  const int r0 = blockIdx.y * rows_per_chunk;
  const int r1 = min(n_waves, r0 + rows_per_chunk);
  float s = 0.f;
  if (c < 3 * H && dst)
    for (int w = r0 + rg; w < r1; w += 4) s += partial[(long)w * 3 * H + c];
  red[threadIdx.x] = s;
  __syncthreads();
  if (rg == 0 && c < 3 * H && dst) atomicAdd(dst + c % H, red[cl] + red[64 + cl] + red[128 + cl] + red[192 + cl]);
}

// ------------------------------------------------------------------ GELU second-order reverse
template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd2_kernel(const T* __restrict__ u, const T* __restrict__ udot,
                                                         const T* __restrict__ ggu, const T* __restrict__ gubar,
                                                         T* __restrict__ ubar, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float a[4], b[4], o[4];
    VecIO<T, 4>::load(u + i * 4, a);
    VecIO<T, 4>::load(gubar + i * 4, b);
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = gelu_d1(a[e]) * b[e];
    if (udot) {
      float c[4], d[4];
      VecIO<T, 4>::load(udot + i * 4, c);
      VecIO<T, 4>::load(ggu + i * 4, d);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] += gelu_d2(a[e]) * c[e] * d[e];
    }
    VecIO<T, 4>::store(ubar + i * 4, o);
  }
}

// ------------------------------------------------------------------ bias gradients
// grid (ceil(N/64), row_chunks): thread (c = tid&63, rg = tid>>6) sums rows rg, rg+4, ... of
// its chunk for column n, the four row groups combine through LDS, one atomic per column.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ dY, long ld, int rows, int N,
                                                      float* __restrict__ db, int rows_per_chunk) {
  __shared__ float red[256];
  const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + c;
  const int r0 = blockIdx.y * rows_per_chunk;
  const int r1 = min(rows, r0 + rows_per_chunk);
  float s = 0.f;
  if (n < N)
    for (int r = r0 + rg; r < r1; r += 4) s += to_f32<T>(dY[(long)r * ld + n]);
  red[threadIdx.x] = s;
  __syncthreads();
  if (rg == 0 && n < N) atomicAdd(db + n, red[c] + red[64 + c] + red[128 + c] + red[192 + c]);
}

// Vector form (16-byte-aligned rows, N a multiple of the 16-byte element count): a lane owns E
// consecutive columns and reads them as one 16-byte load per row, so a wave covers 1 KiB of a
// row; grid (ceil(N / (64 E)), row_chunks).
template <typename T, int NW>
__global__ __launch_bounds__(64 * NW) void colsum_vec_kernel(const T* __restrict__ dY, long ld, int rows, int N,
                                                             float* __restrict__ db, int rows_per_chunk) {
  constexpr int E = 16 / sizeof(T);
  __shared__ float red[NW - 1][64 * E];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n0 = (blockIdx.x * 64 + lane) * E;
  const int r0 = blockIdx.y * rows_per_chunk;
  const int r1 = min(rows, r0 + rows_per_chunk);
  float s[E];
#pragma unroll
  for (int e = 0; e < E; ++e) s[e] = 0.f;
  if (n0 < N) {
    const T* p = dY + n0;
#pragma unroll 8
    for (int r = r0 + w; r < r1; r += NW) {
      float v[E];
      VecIO<T, E>::load(p + (long)r * ld, v);
#pragma unroll
      for (int e = 0; e < E; ++e) s[e] += v[e];
    }
  }
  if (w > 0) {
#pragma unroll
    for (int e = 0; e < E; ++e) red[w - 1][e * 64 + lane] = s[e];
  }
  __syncthreads();
  if (w == 0 && n0 < N) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      float t = s[e];
#pragma unroll
      for (int i = 0; i < NW - 1; ++i) t += red[i][e * 64 + lane];
      atomicAdd(db + n0 + e, t);
    }
  }
}

// ------------------------------------------------------------------ softmax family
// One wave per score row; the row (<= 64*MAXI columns) lives in registers.
constexpr int MAXI = 16;

template <typename T>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ sc, const float* __restrict__ mask,
                                                           T* __restrict__ P, long n_rows, int heads, int S, int ldS) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const long z = row / S;
  const int b = (int)(z / heads);
  const float* x = sc + row * ldS;
  const int ni = ldS / 64;
  float v[MAXI];
  float mx = -FLT_MAX;
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    if (i < ni) {
      const int j = i * 64 + lane;
      float s = -FLT_MAX;
      if (j < S) {
        s = x[j];
        if (mask) s = s + (1.0f - mask[(long)b * S + j]) * -FLT_MAX;
      }
      v[i] = s;
      mx = fmaxf(mx, s);
    }
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    if (i < ni) {
      const int j = i * 64 + lane;
      const float e = j < S ? expf(v[i] - mx) : 0.f;
      v[i] = e;
      sum += e;
    }
  }
  sum = wave_sum(sum);
  T* o = P + row * ldS;
#pragma unroll
  for (int i = 0; i < MAXI; ++i)
    if (i < ni) o[i * 64 + lane] = from_f32<T>(v[i] / sum);
}

template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ dP, const T* __restrict__ P,
                                                           T* __restrict__ dS, long n_rows, int S, int ldS) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int ni = ldS / 64;
  float p[MAXI], d[MAXI];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    if (i < ni) {
      const int j = i * 64 + lane;
      p[i] = to_f32<T>(P[row * ldS + j]);
      d[i] = j < S ? dP[row * ldS + j] : 0.f;
      s += p[i] * d[i];
    }
  }
  s = wave_sum(s);
#pragma unroll
  for (int i = 0; i < MAXI; ++i)
    if (i < ni) dS[row * ldS + i * 64 + lane] = from_f32<T>(p[i] * (d[i] - s));
}

template <typename T>
__global__ __launch_bounds__(256) void softmax_jvp_kernel(const float* __restrict__ Sd, const T* __restrict__ P,
                                                           T* __restrict__ Sd_out, T* __restrict__ Pd, long n_rows, int S,
                                                           int ldS) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int ni = ldS / 64;
  float p[MAXI], d[MAXI];
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    if (i < ni) {
      const int j = i * 64 + lane;
      p[i] = to_f32<T>(P[row * ldS + j]);
      d[i] = j < S ? Sd[row * ldS + j] : 0.f;
      r += p[i] * d[i];
    }
  }
  r = wave_sum(r);
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    if (i < ni) {
      const long o = row * ldS + i * 64 + lane;
      Sd_out[o] = from_f32<T>(d[i]);
      Pd[o] = from_f32<T>(p[i] * (d[i] - r));
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd2_kernel(const T* __restrict__ P, const T* __restrict__ Sd,
                                                            const float* __restrict__ gP, const float* __restrict__ X,
                                                            T* __restrict__ Sbar, T* __restrict__ Pd, long n_rows, int S,
                                                            int ldS) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int ni = ldS / 64;
  float p[MAXI], d[MAXI], g[MAXI], pb[MAXI];
  float r = 0.f, c = 0.f;
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    if (i < ni) {
      const int j = i * 64 + lane;
      const long o = row * ldS + j;
      p[i] = to_f32<T>(P[o]);
      d[i] = to_f32<T>(Sd[o]);
      g[i] = j < S ? gP[o] : 0.f;
      pb[i] = j < S ? X[o] : 0.f;
      r += p[i] * d[i];
      c += p[i] * g[i];
    }
  }
  r = wave_sum(r);
  c = wave_sum(c);
  float e = 0.f;
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    if (i < ni) {
      pb[i] = pb[i] + g[i] * (d[i] - r) - c * d[i];
      e += pb[i] * p[i];
    }
  }
  e = wave_sum(e);
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    if (i < ni) {
      const long o = row * ldS + i * 64 + lane;
      Sbar[o] = from_f32<T>(p[i] * (pb[i] - e));
      Pd[o] = from_f32<T>(p[i] * (d[i] - r));
    }
  }
}

// ------------------------------------------------------------------ dispatch helpers
#define ROW_DISPATCH(H, ...)                                          \
  switch (H) {                                                        \
    case 64:   { constexpr int VW = 1, NCH = 1; __VA_ARGS__; } break; \
    case 128:  { constexpr int VW = 1, NCH = 2; __VA_ARGS__; } break; \
    case 256:  { constexpr int VW = 4, NCH = 1; __VA_ARGS__; } break; \
    case 512:  { constexpr int VW = 4, NCH = 2; __VA_ARGS__; } break; \
    case 1024: { constexpr int VW = 4, NCH = 4; __VA_ARGS__; } break; \
    default: return PG_ERR_UNSUPPORTED;                               \
  }

inline int check_launch() { return hipGetLastError() == hipSuccess ? PG_OK : PG_ERR_LAUNCH; }
inline int row_blocks(long rows) { return (int)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK); }

}  // namespace

extern "C" int pg_ln_partial_rows(int rows) { return rows < 1024 ? rows : 1024; }
extern "C" int pg_score_ld(int S) { return ((S + 63) / 64) * 64; }

static int embed_ln_launch(int dtype, const int64_t* ids, const int* pos_ids, const float* word, const float* pos,
                           const float* type0, const float* gamma, const float* beta, float eps, void* out, void* sum_out,
                           int B, int S, int H, void* stream) {
  if (!ids || !word || !pos || !type0 || !gamma || !beta || !out || B <= 0 || S <= 0) return PG_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int blocks = row_blocks((long)B * S);
  if (dtype == PG_F32) {
    ROW_DISPATCH(H, hipLaunchKernelGGL((embed_ln_kernel<float, VW, NCH>), dim3(blocks), dim3(256), 0, st, ids, word, pos,
                                        type0, gamma, beta, eps, (float*)out, (float*)sum_out, B, S, H, pos_ids));
  } else {
    ROW_DISPATCH(H, hipLaunchKernelGGL((embed_ln_kernel<bf16_t, VW, NCH>), dim3(blocks), dim3(256), 0, st, ids, word, pos,
                                        type0, gamma, beta, eps, (bf16_t*)out, (bf16_t*)sum_out, B, S, H, pos_ids));
  }
  return check_launch();
}
extern "C" int pg_embed_ln2(int dtype, const int64_t* ids, const float* word, const float* pos, const float* type0,
                            const float* gamma, const float* beta, float eps, void* out, void* sum_out, int B, int S, int H,
                            void* stream) {
  return embed_ln_launch(dtype, ids, nullptr, word, pos, type0, gamma, beta, eps, out, sum_out, B, S, H, stream);
}
extern "C" int pg_embed_ln_packed(int dtype, const int64_t* ids, const int32_t* pos_ids, const float* word, const float* pos,
                                  const float* type0, const float* gamma, const float* beta, float eps, void* out, int T,
                                  int H, void* stream) {
  if (!pos_ids) return PG_ERR_ARG;
  return embed_ln_launch(dtype, ids, pos_ids, word, pos, type0, gamma, beta, eps, out, nullptr, 1, T, H, stream);
}
extern "C" int pg_embed_ln2_packed(int dtype, const int64_t* ids, const int32_t* pos_ids, const float* word, const float* pos,
                                   const float* type0, const float* gamma, const float* beta, float eps, void* out,
                                   void* sum_out, int T, int H, void* stream) {
  if (!pos_ids) return PG_ERR_ARG;
  return embed_ln_launch(dtype, ids, pos_ids, word, pos, type0, gamma, beta, eps, out, sum_out, 1, T, H, stream);
}
extern "C" int pg_embed_ln(int dtype, const int64_t* ids, const float* word, const float* pos, const float* type0,
                           const float* gamma, const float* beta, float eps, void* out, int B, int S, int H, void* stream) {
  return pg_embed_ln2(dtype, ids, word, pos, type0, gamma, beta, eps, out, nullptr, B, S, H, stream);
}

extern "C" int pg_layernorm(int dtype, const void* x, const float* gamma, const float* beta, float eps, void* y, int rows,
                            int H, void* stream) {
  if (!x || !gamma || !beta || !y || rows <= 0) return PG_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int blocks = row_blocks(rows);
  if (dtype == PG_F32) {
    ROW_DISPATCH(H, hipLaunchKernelGGL((ln_fwd_kernel<float, VW, NCH>), dim3(blocks), dim3(256), 0, st, (const float*)x,
                                        gamma, beta, eps, (float*)y, rows, H));
  } else {
    ROW_DISPATCH(H, hipLaunchKernelGGL((ln_fwd_kernel<bf16_t, VW, NCH>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)x,
                                        gamma, beta, eps, (bf16_t*)y, rows, H));
  }
  return check_launch();
}

extern "C" int pg_layernorm_jvp(int dtype, const void* x, const void* xdot, const float* gamma, float eps, void* ydot,
                                int rows, int H, void* stream) {
  if (!x || !xdot || !gamma || !ydot || rows <= 0) return PG_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int blocks = row_blocks(rows);
  if (dtype == PG_F32) {
    ROW_DISPATCH(H, hipLaunchKernelGGL((ln_jvp_kernel<float, VW, NCH>), dim3(blocks), dim3(256), 0, st, (const float*)x,
                                        (const float*)xdot, gamma, eps, (float*)ydot, rows, H));
  } else {
    ROW_DISPATCH(H, hipLaunchKernelGGL((ln_jvp_kernel<bf16_t, VW, NCH>), dim3(blocks), dim3(256), 0, st,
                                        (const bf16_t*)x, (const bf16_t*)xdot, gamma, eps, (bf16_t*)ydot, rows, H));
  }
  return check_launch();
}

namespace {
template <bool SECOND>
int ln_bwd_launch(int dtype, const void* x, const void* xdot, const void* gy, const void* ybar, const float* gamma,
                  float eps, void* xbar, float* dgamma, float* dbeta, float* dxsum, float* partial, int rows, int H,
                  hipStream_t st) {
  const bool want_params = dgamma || dbeta || dxsum;
  if (want_params && !partial) return PG_ERR_WORKSPACE;
  // up to 1024 workgroups of four waves (16 waves per CU keep enough rows in flight); one partial
  // row per workgroup, so the scratch stays at pg_ln_partial_rows(rows) rows
  const int blocks = rows < 4096 ? row_blocks(rows) : 1024;
  const int n_waves = blocks * ROWS_PER_BLOCK;
  float* part = want_params ? partial : nullptr;
  if (dtype == PG_F32) {
    ROW_DISPATCH(H, hipLaunchKernelGGL((ln_bwd_kernel<float, VW, NCH, SECOND>), dim3(blocks), dim3(256), 0, st,
                                        (const float*)x, (const float*)xdot, (const float*)gy, (const float*)ybar, gamma,
                                        eps, (float*)xbar, part, rows, H, n_waves));
  } else {
    ROW_DISPATCH(H, hipLaunchKernelGGL((ln_bwd_kernel<bf16_t, VW, NCH, SECOND>), dim3(blocks), dim3(256), 0, st,
                                        (const bf16_t*)x, (const bf16_t*)xdot, (const bf16_t*)gy, (const bf16_t*)ybar,
                                        gamma, eps, (bf16_t*)xbar, part, rows, H, n_waves));
  }
  if (hipGetLastError() != hipSuccess) return PG_ERR_LAUNCH;
  if (want_params) {
    int chunks = (blocks + 63) / 64;
    if (chunks > 16) chunks = 16;
    const int rpc = (blocks + chunks - 1) / chunks;
    hipLaunchKernelGGL(ln_param_reduce_kernel, dim3((3 * H + 63) / 64, chunks), dim3(256), 0, st, partial, blocks, H,
                       dgamma, dbeta, dxsum, rpc);
  }
  return check_launch();
}
}  // namespace

extern "C" int pg_layernorm_bwd(int dtype, const void* x, const void* dy, const float* gamma, float eps, void* dx,
                                float* dgamma, float* dbeta, float* dxsum, float* partial, int rows, int H, void* stream) {
  if (!x || !dy || !gamma || !dx || rows <= 0) return PG_ERR_ARG;
  return ln_bwd_launch<false>(dtype, x, nullptr, dy, nullptr, gamma, eps, dx, dgamma, dbeta, dxsum, partial, rows, H,
                              (hipStream_t)stream);
}
// ---- allocation-free LayerNorm backward -----------------------------------------------------
// Same kernels as pg_layernorm_bwd, with the partial sums in a file-scope device array instead of
// caller memory.  Every GPU the library runs on holds its own copy of the array (12 MiB) at its
// own address; the address is looked up once per device and only that pointer is ever handed to
// launches.  All calls on one GPU share the array: calls that overlap on one GPU race.
constexpr int LN_STATIC_MAX_H = 1024;
constexpr int LN_STATIC_ROWS = 1024;  // pg_ln_partial_rows never exceeds this
static __device__ __align__(256) float ln_static_scratch[3 * LN_STATIC_MAX_H * LN_STATIC_ROWS];
constexpr int LN_STATIC_MAX_DEVICES = 64;
static float* g_ln_static_ptr[LN_STATIC_MAX_DEVICES];  // zero-initialised: not looked up yet

extern "C" int pg_layernorm_bwd_static(int dtype, const void* x, const void* dy, const float* gamma, float eps, void* dx,
                                       float* dgamma, float* dbeta, float* dxsum, int rows, int H, void* stream) {
  if (!x || !dy || !gamma || !dx || rows <= 0) return PG_ERR_ARG;
  if (H > LN_STATIC_MAX_H || pg_ln_partial_rows(rows) > LN_STATIC_ROWS) return PG_ERR_SCRATCH;
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= LN_STATIC_MAX_DEVICES) return PG_ERR_LAUNCH;
  if (!g_ln_static_ptr[dev]) {
    float* p = nullptr;
    if (hipGetSymbolAddress((void**)&p, HIP_SYMBOL(ln_static_scratch)) != hipSuccess || !p) return PG_ERR_LAUNCH;
    g_ln_static_ptr[dev] = p;
  }
  // (every partial row the reduction reads is written by the launch before it: nothing to clear)
  return ln_bwd_launch<false>(dtype, x, nullptr, dy, nullptr, gamma, eps, dx, dgamma, dbeta, dxsum, g_ln_static_ptr[dev],
                              rows, H, (hipStream_t)stream);
}

extern "C" int pg_layernorm_bwd2(int dtype, const void* x, const void* xdot, const void* gy, const void* ybar,
                                 const float* gamma, float eps, void* xbar, float* dgamma, float* dbeta, float* dxsum,
                                 float* partial, int rows, int H, void* stream) {
  if (!x || !xdot || !gy || !gamma || !xbar || rows <= 0) return PG_ERR_ARG;
  return ln_bwd_launch<true>(dtype, x, xdot, gy, ybar, gamma, eps, xbar, dgamma, dbeta, dxsum, partial, rows, H,
                             (hipStream_t)stream);
}

extern "C" int pg_gelu_bwd2(int dtype, const void* u, const void* udot, const void* ggu, const void* gubar, void* ubar,
                            int64_t n, void* stream) {
  if (!u || !gubar || !ubar || n <= 0 || (n & 3)) return PG_ERR_ARG;
  if ((udot == nullptr) != (ggu == nullptr)) return PG_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const long n4 = n / 4;
  const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  if (dtype == PG_F32)
    hipLaunchKernelGGL(gelu_bwd2_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)u, (const float*)udot,
                       (const float*)ggu, (const float*)gubar, (float*)ubar, n4);
  else
    hipLaunchKernelGGL(gelu_bwd2_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)u, (const bf16_t*)udot,
                       (const bf16_t*)ggu, (const bf16_t*)gubar, (bf16_t*)ubar, n4);
  return check_launch();
}

int pg_internal_colsum_waves = 16;  // waves per workgroup of a column-sum launch, 4 or 16 (pg_tune key 17)
int pg_internal_colsum_chunks = 32;  // most row chunks of one column-sum launch (pg_tune key 12)
extern "C" int pg_colsum_acc(int dtype, const void* dY, int64_t ld, int rows, int N, float* db, void* stream) {
  if (!dY || !db || rows <= 0 || N <= 0) return PG_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int E = dtype == PG_F32 ? 4 : 8;
  if (ld % E == 0 && N % E == 0 && ((uintptr_t)dY) % 16 == 0) {
    // at most 32 row chunks: every chunk ends in one atomic per column, and many workgroups adding
    // into the same few KiB at once serialise
    const int gx = (N + 64 * E - 1) / (64 * E);
    // Sixteen waves per workgroup keep four times the loads in flight per atomic: 12.6-13.3 us against 21.7 us for
    // 5376 x 4096 bf16, and 32 chunks stay best up to 8192 rows (four waves: 64 chunks from 8192 rows on).
    const int nw = pg_internal_colsum_waves == 16 ? 16 : 4;
    int chunks = rows / 64;
    const int cap = (rows >= 8192 && nw == 4) ? 2 * pg_internal_colsum_chunks : pg_internal_colsum_chunks;
    chunks = chunks < 1 ? 1 : (chunks > cap ? cap : chunks);
    const int rpc = ((rows + chunks - 1) / chunks + nw - 1) / nw * nw;
    dim3 grid(gx, (rows + rpc - 1) / rpc);
    if (dtype == PG_F32) {
      if (nw == 16)
        hipLaunchKernelGGL((colsum_vec_kernel<float, 16>), grid, dim3(1024), 0, st, (const float*)dY, (long)ld, rows, N, db, rpc);
      else
        hipLaunchKernelGGL((colsum_vec_kernel<float, 4>), grid, dim3(256), 0, st, (const float*)dY, (long)ld, rows, N, db, rpc);
    } else {
      if (nw == 16)
        hipLaunchKernelGGL((colsum_vec_kernel<bf16_t, 16>), grid, dim3(1024), 0, st, (const bf16_t*)dY, (long)ld, rows, N, db, rpc);
      else
        hipLaunchKernelGGL((colsum_vec_kernel<bf16_t, 4>), grid, dim3(256), 0, st, (const bf16_t*)dY, (long)ld, rows, N, db, rpc);
    }
    return check_launch();
  }
  int chunks = (rows + 255) / 256;
  if (chunks > 64) chunks = 64;
  const int rpc = (rows + chunks - 1) / chunks;
  dim3 grid((N + 63) / 64, chunks);
  if (dtype == PG_F32)
    hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, st, (const float*)dY, (long)ld, rows, N, db, rpc);
  else
    hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)dY, (long)ld, rows, N, db, rpc);
  return check_launch();
}

extern "C" int pg_softmax_fwd(int dtype, const float* scores, const float* mask, void* P, int B, int heads, int S,
                              void* stream) {
  if (!scores || !P || B <= 0 || heads <= 0 || S <= 0) return PG_ERR_ARG;
  const int ldS = pg_score_ld(S);
  if (ldS > 64 * MAXI) return PG_ERR_UNSUPPORTED;
  const long n_rows = (long)B * heads * S;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == PG_F32)
    hipLaunchKernelGGL(softmax_fwd_kernel<float>, dim3(row_blocks(n_rows)), dim3(256), 0, st, scores, mask, (float*)P,
                       n_rows, heads, S, ldS);
  else
    hipLaunchKernelGGL(softmax_fwd_kernel<bf16_t>, dim3(row_blocks(n_rows)), dim3(256), 0, st, scores, mask, (bf16_t*)P,
                       n_rows, heads, S, ldS);
  return check_launch();
}
extern "C" int pg_softmax_bwd(int dtype, const float* dP, const void* P, void* dS, int Z, int S, void* stream) {
  if (!dP || !P || !dS || Z <= 0 || S <= 0) return PG_ERR_ARG;
  const int ldS = pg_score_ld(S);
  if (ldS > 64 * MAXI) return PG_ERR_UNSUPPORTED;
  const long n_rows = (long)Z * S;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == PG_F32)
    hipLaunchKernelGGL(softmax_bwd_kernel<float>, dim3(row_blocks(n_rows)), dim3(256), 0, st, dP, (const float*)P,
                       (float*)dS, n_rows, S, ldS);
  else
    hipLaunchKernelGGL(softmax_bwd_kernel<bf16_t>, dim3(row_blocks(n_rows)), dim3(256), 0, st, dP, (const bf16_t*)P,
                       (bf16_t*)dS, n_rows, S, ldS);
  return check_launch();
}
extern "C" int pg_softmax_jvp(int dtype, const float* Sdot, const void* P, void* Sdot_out, void* Pdot, int Z, int S,
                              void* stream) {
  if (!Sdot || !P || !Sdot_out || !Pdot || Z <= 0 || S <= 0) return PG_ERR_ARG;
  const int ldS = pg_score_ld(S);
  if (ldS > 64 * MAXI) return PG_ERR_UNSUPPORTED;
  const long n_rows = (long)Z * S;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == PG_F32)
    hipLaunchKernelGGL(softmax_jvp_kernel<float>, dim3(row_blocks(n_rows)), dim3(256), 0, st, Sdot, (const float*)P,
                       (float*)Sdot_out, (float*)Pdot, n_rows, S, ldS);
  else
    hipLaunchKernelGGL(softmax_jvp_kernel<bf16_t>, dim3(row_blocks(n_rows)), dim3(256), 0, st, Sdot, (const bf16_t*)P,
                       (bf16_t*)Sdot_out, (bf16_t*)Pdot, n_rows, S, ldS);
  return check_launch();
}
extern "C" int pg_softmax_bwd2(int dtype, const void* P, const void* Sdot, const float* gP, const float* X, void* Sbar,
                               void* Pdot, int Z, int S, void* stream) {
  if (!P || !Sdot || !gP || !X || !Sbar || !Pdot || Z <= 0 || S <= 0) return PG_ERR_ARG;
  const int ldS = pg_score_ld(S);
  if (ldS > 64 * MAXI) return PG_ERR_UNSUPPORTED;
  const long n_rows = (long)Z * S;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == PG_F32)
    hipLaunchKernelGGL(softmax_bwd2_kernel<float>, dim3(row_blocks(n_rows)), dim3(256), 0, st, (const float*)P,
                       (const float*)Sdot, gP, X, (float*)Sbar, (float*)Pdot, n_rows, S, ldS);
  else
    hipLaunchKernelGGL(softmax_bwd2_kernel<bf16_t>, dim3(row_blocks(n_rows)), dim3(256), 0, st, (const bf16_t*)P,
                       (const bf16_t*)Sdot, gP, X, (bf16_t*)Sbar, (bf16_t*)Pdot, n_rows, S, ldS);
  return check_launch();
}
