// Fused attention forward (bf16, head_dim 64) for the inference passes of the fill loop:
//   ctx = softmax(Q K^T / 8 + keymask) V      (modeling_bert.py:111-136, eager_attention_forward)
// without materialising the [S,S] score matrix.  One workgroup = 4 waves = 128 query rows of
// one (batch, head); each wave owns 32 query rows (two 16-row groups) and walks the keys in
// 64-key tiles.  K and V tiles arrive by LDS-DMA (global_load_lds_dwordx4) into a two-buffer
// ring, XOR-swizzled on the source address (chunk c of row r sits at position c ^ (r & 7)).
//
// Both products are issued "transposed" so the query index sits on the MFMA lane:
//   S^T = K Q^T : lane (q = lane & 15, g = lane >> 4) holds keys 16*blk + 4g + reg of query q
//   O^T = V^T P^T: the exponentiated scores of two 16-key blocks are, as they stand in the
//                  accumulator registers, the B operand of a 32-deep MFMA step whose k-slots
//                  are keys {4g + j, 16 + 4g + j}; the A operand (V^T) for exactly those slots
//                  is what two transposing LDS reads (ds_read_b64_tr_b16) deliver.
// So the online-softmax statistics are per lane (plus two cross-lane shuffles), nothing goes
// through LDS between the two products, and each lane ends with 4 consecutive head-dim
// values of one query row: 8-byte output stores.
#include <float.h>
#include <math.h>
#include "pg_common.h"
#include "protgen_hip.h"

int pg_internal_attn_resident = 1;  // 0 = the forward that writes P streams K / V through the ring at every length (pg_tune key 25)

namespace {

typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int QB = 128;      // query rows per workgroup
constexpr int KT = 64;       // keys per tile
constexpr int TILE = KT * 64;  // bf16 elements of one K (or V) tile: 8 KiB
constexpr int MAX_S = 2048;    // additive key mask of one batch row lives in LDS

template <int IMM>
__device__ __forceinline__ u32x4 rd128(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM) : "memory");
  return v;
}
template <int IMM>
__device__ __forceinline__ u32x2 rdtr(unsigned addr) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM) : "memory");
  return v;
}

// Sequences of different lengths laid back to back ("ragged"): sequence b of a launch is the token rows
// [seq_off[b] - tok0, seq_off[b+1] - tok0) of every [T][ld] activation operand; score-shaped operands keep
// one slot of slot_rows x ldS per (sequence, head), of which the leading len rows / columns are used.  A
// kernel that writes score-shaped output also writes zeros into the rows [len, 64-multiple): the key-side
// contractions over these buffers reduce over whole 64-row tiles.  seq_off == NULL: B x S rectangle.
struct RaggedArgs { const int* seq_off; int tok0; int slot_rows; };
struct SeqSpan { long row0; int len, store_rows, slot_rows; };
__device__ __forceinline__ SeqSpan seq_span(const RaggedArgs& rg, int b, int S) {
  SeqSpan sp;
  if (rg.seq_off) {
    const int a = rg.seq_off[b];
    sp.row0 = a - rg.tok0;
    sp.len = rg.seq_off[b + 1] - a;
    const int pad = (sp.len + 63) & ~63;
    sp.store_rows = pad < rg.slot_rows ? pad : rg.slot_rows;
    sp.slot_rows = rg.slot_rows;
  } else {
    sp.row0 = (long)b * S;
    sp.len = S;
    sp.store_rows = S;
    sp.slot_rows = S;
  }
  return sp;
}

// One 64-row x 128-byte tile: 8 wave-instructions of 1 KiB, two per wave.
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ base, long ld, int row0, int rows_max, bf16_t* lds_tile,
                                           int wid, int lane) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int i = wid * 2 + j;
    const int r = 8 * i + (lane >> 3);
    const int c = (lane & 7) ^ (lane >> 3);
    int rg = row0 + r;
    rg = rg < rows_max ? rg : rows_max - 1;
    __builtin_amdgcn_global_load_lds(base + (long)rg * ld + c * 8, lds_tile + i * 512, 16, 0, 0);
  }
}

// The visible keys of every sequence, in order: idx[row0 + j] = position inside the sequence of its j-th key whose
// mask is 1, cnt[b] = their number; cnt[b] = -1 when the sequence is to be walked whole under the additive mask
// (a mask value other than 0 / 1, or no visible key at all: the reference's softmax is then uniform over all keys).
// A masked key has weight exactly 0 (exp(-FLT_MAX - max) = 0), so leaving it out changes no sum; a fill pass
// that starts from a fully masked template attends 2 keys of 500 in its first iteration and half of them midway.
__global__ __launch_bounds__(256) void key_compact_kernel(const float* __restrict__ mask, const int* __restrict__ seq_off, int S,
                                                          int* __restrict__ idx, int* __restrict__ cnt) {
  __shared__ int wsum[4];
  __shared__ int bad;
  const int b = blockIdx.x;
  long row0 = (long)b * S;
  int len = S;
  if (seq_off) {
    row0 = seq_off[b];
    len = seq_off[b + 1] - seq_off[b];
  }
  if (threadIdx.x == 0) bad = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int running = 0;
  for (int c0 = 0; c0 < len; c0 += 256) {
    const int k = c0 + threadIdx.x;
    const float m = k < len ? mask[row0 + k] : 0.f;
    const bool vis = k < len && m == 1.f;
    if (k < len && m != 0.f && m != 1.f) bad = 1;
    const unsigned long long bal = __ballot(vis);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[w] = __popcll(bal);
    __syncthreads();
    int off = running;
    for (int i = 0; i < w; ++i) off += wsum[i];
    if (vis) idx[row0 + off + before] = k;
    running += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
  if (threadIdx.x == 0) cnt[b] = (bad || running == 0) ? -1 : running;
}

// 2^x as the one hardware instruction (v_exp_f32; arguments here are <= 0, results below 2^-126 may flush to zero, which
// is what a probability that small rounds to in bf16 anyway)
__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }
constexpr float kScaleLog2e = 0.125f * 1.44269504088896340736f;

template <int IMM>
__device__ __forceinline__ int rd32(unsigned addr) {
  int v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM) : "memory");
  return v;
}

// stage_tile with the key rows taken from a table in LDS (table_addr: LDS byte address of entry 0; entry k = row of
// the k-th key of the walk, valid for every k below the 64-multiple the walk ends at).
__device__ __forceinline__ void stage_tile_tab(const bf16_t* __restrict__ kbase, const bf16_t* __restrict__ vbase, long ld, int k0,
                                               unsigned table_addr, bf16_t* lds_k, bf16_t* lds_v, int wid, int lane) {
  const unsigned a = table_addr + (unsigned)((k0 + 16 * wid + (lane >> 3)) * 4);
  const int r0 = rd32<0>(a), r1 = rd32<32>(a);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);  // (nothing that uses r0 / r1 may be scheduled above the wait)
  const int c = (lane & 7) ^ (lane >> 3);
  const int i = wid * 2;
  __builtin_amdgcn_global_load_lds(kbase + (long)r0 * ld + c * 8, lds_k + i * 512, 16, 0, 0);
  __builtin_amdgcn_global_load_lds(kbase + (long)r1 * ld + c * 8, lds_k + (i + 1) * 512, 16, 0, 0);
  __builtin_amdgcn_global_load_lds(vbase + (long)r0 * ld + c * 8, lds_v + i * 512, 16, 0, 0);
  __builtin_amdgcn_global_load_lds(vbase + (long)r1 * ld + c * 8, lds_v + (i + 1) * 512, 16, 0, 0);
}

__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ mask,
                                                           bf16_t* __restrict__ ctx, int S, int heads, long ld_qkv, long ld_ctx,
                                                           const int* __restrict__ seq_off, const int* __restrict__ key_idx,
                                                           const int* __restrict__ key_cnt) {
  // [buffer][K|V] tiles, then the additive key mask (fp32) of this batch row, then the rows of the keys the walk
  // visits (int).  One array only: a second __shared__ object would make the compiler drain the LDS-DMA before
  // LDS reads.
  __shared__ __align__(16) bf16_t smem[4 * TILE + 4 * MAX_S];
  const int lid = xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
  const int bh = lid / gridDim.x, qt = lid % gridDim.x;  // the query tiles of one head share an XCD's L2
  const int b = bh / heads, h = bh % heads;
  // packed rows: sequence b occupies token rows [seq_off[b], seq_off[b+1]) and the grid is sized for the longest
  long row0 = (long)b * S;
  if (seq_off) {
    row0 = seq_off[b];
    S = seq_off[b + 1] - seq_off[b];
    if (qt * QB >= S) return;
  }
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int H = heads * 64;
  const bf16_t* Q = qkv + row0 * ld_qkv + h * 64;
  const bf16_t* K = Q + H;
  const bf16_t* V = Q + 2 * H;
  const float* mrow = mask ? mask + row0 : nullptr;
  const int q0 = qt * QB + wid * 32;
  // keys walked: the sequence's visible keys when a list of them came along, else all S under the additive mask
  int Sk = S;
  const int* kidx = nullptr;
  if (key_cnt) {
    const int c = key_cnt[b];
    if (c > 0 && c <= S) {  // (anything else, -1 included: the whole walk under the additive mask)
      Sk = c;
      kidx = key_idx + row0;
    }
  }

  // Q fragments (B operand of S^T = K Q^T): lane holds Q[q][32kk + 8g .. +8]
  bf16x8_t qf[2][2];
#pragma unroll
  for (int qg = 0; qg < 2; ++qg) {
    int q = q0 + qg * 16 + r;
    q = q < S ? q : S - 1;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint4 v = *reinterpret_cast<const uint4*>(Q + (long)q * ld_qkv + kk * 32 + 8 * g);
      qf[qg][kk] = __builtin_bit_cast(bf16x8_t, v);
    }
  }
  {
    float* ml = reinterpret_cast<float*>(smem + 4 * TILE);
    int* il = reinterpret_cast<int*>(smem + 4 * TILE + 2 * MAX_S);
    const int s_pad = ((Sk + KT - 1) / KT) * KT;
    for (int k = threadIdx.x; k < s_pad; k += 256) {
      float a = 0.f;
      if (k >= Sk) a = -INFINITY;                              // beyond the walk: weight exactly 0
      else if (!kidx && mrow) a = (1.0f - mrow[k]) * -FLT_MAX;  // HF additive mask, finfo(float32).min
      ml[k] = a;
      const int kc = k < Sk ? k : Sk - 1;                       // (past the walk: any valid row, its weight is 0)
      il[k] = kidx ? kidx[kc] : kc;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  f32x4 o[2][4];
  float m_run[2], l_run[2];
#pragma unroll
  for (int qg = 0; qg < 2; ++qg) {
    m_run[qg] = -INFINITY;
    l_run[qg] = 0.f;
#pragma unroll
    for (int d = 0; d < 4; ++d) o[qg][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  // per-lane LDS byte offsets inside a tile
  unsigned koff[2];  // K fragment: row 16*blk + r (blk adds 2048), chunk 4kk + g
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) koff[kk] = (unsigned)(r * 128 + (((kk * 4 + g) ^ (r & 7)) << 4));
  unsigned voff[2][4];  // V^T fragment: key row 4g + q' (+16), d-block dblk; k-step s2 adds 4096
  {
    const int qq = r >> 2, pp = r & 3;
#pragma unroll
    for (int hi = 0; hi < 2; ++hi)
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int k = 4 * g + qq + 16 * hi;
        const int c = 2 * d + (pp >> 1);
        voff[hi][d] = (unsigned)(k * 128 + ((c ^ (k & 7)) << 4) + 8 * (pp & 1));
      }
  }
  const unsigned smem_addr = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)smem;
  const unsigned mask_addr = smem_addr + 4 * TILE * 2;
  const unsigned tab_addr = mask_addr + MAX_S * 4;

  const int nt = (Sk + KT - 1) / KT;
  stage_tile_tab(K, V, ld_qkv, 0, tab_addr, smem, smem + TILE, wid, lane);
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    if (t + 1 < nt) {
      bf16_t* nb = smem + ((t + 1) & 1) * 2 * TILE;
      stage_tile_tab(K, V, ld_qkv, (t + 1) * KT, tab_addr, nb, nb + TILE, wid, lane);
    }
    const unsigned ka = smem_addr + (t & 1) * 4 * TILE;
    const unsigned va = ka + 2 * TILE;
    // ---- additive mask of this lane's 16 keys of the tile (4 consecutive keys per block)
    u32x4 mraw[4];
    {
      const unsigned ma = mask_addr + (unsigned)((t * KT + 4 * g) * 4);
      mraw[0] = rd128<0>(ma);
      mraw[1] = rd128<64>(ma);
      mraw[2] = rd128<128>(ma);
      mraw[3] = rd128<192>(ma);
    }
    // ---- K fragments, then S^T = K Q^T
    u32x4 kf[4][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      kf[0][kk] = rd128<0>(ka + koff[kk]);
      kf[1][kk] = rd128<2048>(ka + koff[kk]);
      kf[2][kk] = rd128<4096>(ka + koff[kk]);
      kf[3][kk] = rd128<6144>(ka + koff[kk]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    f32x4 s[2][4];
#pragma unroll
    for (int qg = 0; qg < 2; ++qg)
#pragma unroll
      for (int blk = 0; blk < 4; ++blk) {
        f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf[blk][0]), qf[qg][0], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf[blk][1]), qf[qg][1], a, 0, 0, 0);
        s[qg][blk] = a;
      }
    // ---- V^T fragments in flight under the softmax arithmetic
    u32x2 vlo[2][4], vhi[2][4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      vlo[0][d] = rdtr<0>(va + voff[0][d]);
      vhi[0][d] = rdtr<0>(va + voff[1][d]);
      vlo[1][d] = rdtr<4096>(va + voff[0][d]);
      vhi[1][d] = rdtr<4096>(va + voff[1][d]);
    }
    float madd[4][4];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk)
#pragma unroll
      for (int e = 0; e < 4; ++e) madd[blk][e] = __uint_as_float(mraw[blk][e]);
    bf16x8_t pf[2][2];
#pragma unroll
    for (int qg = 0; qg < 2; ++qg) {
      float mx = -INFINITY;
#pragma unroll
      for (int blk = 0; blk < 4; ++blk)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = s[qg][blk][e] * kScaleLog2e + madd[blk][e];  // (base-2 exponent: exp(x / 8) = 2^(x log2(e) / 8))
          s[qg][blk][e] = v;
          mx = fmaxf(mx, v);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run[qg], mx);
      const float alpha = exp2_fast(m_run[qg] - m_new);
      float sum = 0.f;
#pragma unroll
      for (int blk = 0; blk < 4; ++blk)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float p = exp2_fast(s[qg][blk][e] - m_new);
          s[qg][blk][e] = p;
          sum += p;
        }
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      l_run[qg] = l_run[qg] * alpha + sum;
      m_run[qg] = m_new;
#pragma unroll
      for (int d = 0; d < 4; ++d) o[qg][d] *= alpha;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8_t p8;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          p8[e] = (__bf16)s[qg][2 * s2][e];
          p8[4 + e] = (__bf16)s[qg][2 * s2 + 1][e];
        }
        pf[qg][s2] = p8;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // ---- O^T += V^T P^T
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const u32x4 vv = (u32x4){vlo[s2][d][0], vlo[s2][d][1], vhi[s2][d][0], vhi[s2][d][1]};
        const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, vv);
#pragma unroll
        for (int qg = 0; qg < 2; ++qg) o[qg][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qg][s2], o[qg][d], 0, 0, 0);
      }
    __syncthreads();
  }
  // ---- normalise and store: lane holds ctx[q][h*64 + 16d + 4g .. +4]
#pragma unroll
  for (int qg = 0; qg < 2; ++qg) {
    const int q = q0 + qg * 16 + r;
    if (q >= S) continue;
    const float inv = 1.0f / l_run[qg];
    bf16_t* dst = ctx + (row0 + q) * ld_ctx + h * 64 + 4 * g;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      float v[4] = {o[qg][d][0] * inv, o[qg][d][1] * inv, o[qg][d][2] * inv, o[qg][d][3] * inv};
      VecIO<bf16_t, 4>::store(dst + d * 16, v);
    }
  }
}

// Per-lane LDS offsets of the kernels that read or write score-shaped operands.  Key order inside a 64-key tile: row m of MFMA block blk is key
// 32 (blk >> 1) + 8 (m >> 2) + 4 (blk & 1) + (m & 3), so a lane's accumulators of blocks 2j and 2j + 1 are the 8 consecutive
// keys 32 j + 8 g .. + 7 of its query: score-shaped operands move 16 bytes per lane, 64 contiguous bytes per row and
// instruction (keys 16 blk + 4 g .. + 3, the order of the inference forward, would move 8 and 32, and the issue of those
// pieces is what bounds these kernels).
struct ResOffsets {
  unsigned koff[2][2];  // k-major fragment of block (2j + b1), k-step kk: + 4096 j
  unsigned voff[2][4];  // transposing read of k-slots 8 g + 4 hi + (0..3), head-dim block d: + 4096 s2
};
__device__ __forceinline__ ResOffsets res_offsets(int r, int g) {
  ResOffsets o;
#pragma unroll
  for (int b1 = 0; b1 < 2; ++b1)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int row = 8 * (r >> 2) + 4 * b1 + (r & 3);
      o.koff[b1][kk] = (unsigned)(row * 128 + (((kk * 4 + g) ^ (row & 7)) << 4));
    }
  const int qq = r >> 2, pp = r & 3;
#pragma unroll
  for (int hi = 0; hi < 2; ++hi)
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const int k = 8 * g + qq + 4 * hi;
      const int c = 2 * d + (pp >> 1);
      o.voff[hi][d] = (unsigned)(k * 128 + ((c ^ (k & 7)) << 4) + 8 * (pp & 1));
    }
  return o;
}

// Training-pass variant: the same products, but the normalised probabilities P (bf16,
// [B*heads][S][ldS], zeros in the padding columns) are written once for the backward and the
// gradient-penalty passes.  Two sweeps over the keys: the first computes each query row's
// max and sum (Q K^T only), the second recomputes the scores, writes P and accumulates P V.
// That replaces an fp32 score matrix round trip (GEMM store, softmax load/store, GEMM load).
__global__ __launch_bounds__(256, 2) void attn_fwd_p_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ mask,
                                                             bf16_t* __restrict__ P, bf16_t* __restrict__ ctx, int S, int ldS,
                                                             int heads, long ld_qkv, long ld_ctx, RaggedArgs rg) {
  __shared__ __align__(16) bf16_t smem[4 * TILE + 2 * MAX_S];
  const int lid = xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
  const int bh = lid / gridDim.x, qt = lid % gridDim.x;
  const int b = bh / heads, h = bh % heads;
  const SeqSpan sp = seq_span(rg, b, S);
  if (qt * QB >= sp.store_rows) return;
  S = sp.len;
  const long row0 = sp.row0;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int H = heads * 64;
  const bf16_t* Q = qkv + row0 * ld_qkv + h * 64;
  const bf16_t* K = Q + H;
  const bf16_t* V = Q + 2 * H;
  const float* mrow = mask ? mask + row0 : nullptr;
  const int q0 = qt * QB + wid * 32;

  bf16x8_t qf[2][2];
#pragma unroll
  for (int qg = 0; qg < 2; ++qg) {
    int q = q0 + qg * 16 + r;
    q = q < S ? q : S - 1;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint4 v = *reinterpret_cast<const uint4*>(Q + (long)q * ld_qkv + kk * 32 + 8 * g);
      qf[qg][kk] = __builtin_bit_cast(bf16x8_t, v);
    }
  }
  {
    float* ml = reinterpret_cast<float*>(smem + 4 * TILE);
    const int s_pad = ((S + KT - 1) / KT) * KT;
    for (int k = threadIdx.x; k < s_pad; k += 256) {
      float a = 0.f;
      if (k >= S) a = -INFINITY;
      else if (mrow) a = (1.0f - mrow[k]) * -FLT_MAX;
      ml[k] = a;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  const ResOffsets ro = res_offsets(r, g);
  const unsigned smem_addr = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)smem;
  const unsigned mask_addr = smem_addr + 4 * TILE * 2;
  const int nt = (S + KT - 1) / KT;

  // scores of one key tile for both query groups, scaled and masked
  auto scores = [&](unsigned ka, int t, f32x4 (&s)[2][4]) {
    u32x4 mraw[4];
    const unsigned ma = mask_addr + (unsigned)((t * KT + 8 * g) * 4);
    mraw[0] = rd128<0>(ma);
    mraw[1] = rd128<16>(ma);
    mraw[2] = rd128<128>(ma);
    mraw[3] = rd128<144>(ma);
    u32x4 kf[4][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      kf[0][kk] = rd128<0>(ka + ro.koff[0][kk]);
      kf[1][kk] = rd128<0>(ka + ro.koff[1][kk]);
      kf[2][kk] = rd128<4096>(ka + ro.koff[0][kk]);
      kf[3][kk] = rd128<4096>(ka + ro.koff[1][kk]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int qg = 0; qg < 2; ++qg)
#pragma unroll
      for (int blk = 0; blk < 4; ++blk) {
        f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf[blk][0]), qf[qg][0], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf[blk][1]), qf[qg][1], a, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = a[e] * 0.125f + __uint_as_float(mraw[blk][e]);
        s[qg][blk] = a;
      }
  };

  // ---- sweep 1: row max and sum
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
  stage_tile(K, ld_qkv, 0, S, smem, wid, lane);
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    if (t + 1 < nt) stage_tile(K, ld_qkv, (t + 1) * KT, S, smem + ((t + 1) & 1) * 2 * TILE, wid, lane);
    f32x4 s[2][4];
    scores(smem_addr + (t & 1) * 4 * TILE, t, s);
#pragma unroll
    for (int qg = 0; qg < 2; ++qg) {
      float mx = -INFINITY;
#pragma unroll
      for (int blk = 0; blk < 4; ++blk)
#pragma unroll
        for (int e = 0; e < 4; ++e) mx = fmaxf(mx, s[qg][blk][e]);
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run[qg], mx);
      float sum = 0.f;
#pragma unroll
      for (int blk = 0; blk < 4; ++blk)
#pragma unroll
        for (int e = 0; e < 4; ++e) sum += __expf(s[qg][blk][e] - m_new);
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      l_run[qg] = l_run[qg] * __expf(m_run[qg] - m_new) + sum;
      m_run[qg] = m_new;
    }
    __syncthreads();
  }
  const float inv_l[2] = {1.0f / l_run[0], 1.0f / l_run[1]};

  // ---- sweep 2: P and P V
  f32x4 o[2][4];
#pragma unroll
  for (int qg = 0; qg < 2; ++qg)
#pragma unroll
    for (int d = 0; d < 4; ++d) o[qg][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
  stage_tile(K, ld_qkv, 0, S, smem, wid, lane);
  stage_tile(V, ld_qkv, 0, S, smem + TILE, wid, lane);
  __syncthreads();
  bf16_t* Pz = P + (long)bh * sp.slot_rows * ldS;
  for (int t = 0; t < nt; ++t) {
    if (t + 1 < nt) {
      bf16_t* nb = smem + ((t + 1) & 1) * 2 * TILE;
      stage_tile(K, ld_qkv, (t + 1) * KT, S, nb, wid, lane);
      stage_tile(V, ld_qkv, (t + 1) * KT, S, nb + TILE, wid, lane);
    }
    const unsigned ka = smem_addr + (t & 1) * 4 * TILE;
    const unsigned va = ka + 2 * TILE;
    f32x4 s[2][4];
    scores(ka, t, s);
    u32x2 vlo[2][4], vhi[2][4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      vlo[0][d] = rdtr<0>(va + ro.voff[0][d]);
      vhi[0][d] = rdtr<0>(va + ro.voff[1][d]);
      vlo[1][d] = rdtr<4096>(va + ro.voff[0][d]);
      vhi[1][d] = rdtr<4096>(va + ro.voff[1][d]);
    }
    bf16x8_t pf[2][2];
#pragma unroll
    for (int qg = 0; qg < 2; ++qg) {
      const int q = q0 + qg * 16 + r;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float pv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          pv[e] = q < S ? __expf(s[qg][2 * j + (e >> 2)][e & 3] - m_run[qg]) * inv_l[qg] : 0.f;  // (rows past the sequence: zeros)
          s[qg][2 * j + (e >> 2)][e & 3] = pv[e];
        }
        if (q < sp.store_rows) VecIO<bf16_t, 8>::store(Pz + (long)q * ldS + t * KT + j * 32 + 8 * g, pv);
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8_t p8;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          p8[e] = (__bf16)s[qg][2 * s2][e];
          p8[4 + e] = (__bf16)s[qg][2 * s2 + 1][e];
        }
        pf[qg][s2] = p8;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const u32x4 vv = (u32x4){vlo[s2][d][0], vlo[s2][d][1], vhi[s2][d][0], vhi[s2][d][1]};
        const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, vv);
#pragma unroll
        for (int qg = 0; qg < 2; ++qg) o[qg][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qg][s2], o[qg][d], 0, 0, 0);
      }
    __syncthreads();
  }
#pragma unroll
  for (int qg = 0; qg < 2; ++qg) {
    const int q = q0 + qg * 16 + r;
    if (q >= S) continue;
    bf16_t* dst = ctx + (row0 + q) * ld_ctx + h * 64 + 4 * g;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      float v[4] = {o[qg][d][0], o[qg][d][1], o[qg][d][2], o[qg][d][3]};
      VecIO<bf16_t, 4>::store(dst + d * 16, v);
    }
  }
}

// First half of the attention backward, consuming the stored probabilities P:
//   dP = dO V^T ; delta_q = <dO_q, O_q> ; dS = P * (dP - delta) ; dQ = dS K / 8
// in the same query-on-the-lane formulation as the forward (dP^T = V dO^T has the layout of
// the score tile; dS, rounded to bf16, is the B operand of dQ^T = K^T dS^T as it stands).
// dS is written out ([B*heads][S][ldS], like P) when STORE_DS: the gradient penalty keeps it, and the
// second half as two GEMMs (dK = dS^T Q / 8, dV = P^T dO) reads it; attn_bwd_dkv_kernel below forms
// it again from P and the row constants delta ([B*heads][S of the launch], written when asked for).
template <bool STORE_DS>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ P,
                                                              const bf16_t* __restrict__ ctx, const bf16_t* __restrict__ dctx,
                                                              bf16_t* __restrict__ dS, bf16_t* __restrict__ dq,
                                                              float* __restrict__ delta_out, int S, int ldS, int heads,
                                                              long ld_qkv, long ld_ctx, long ld_dq, RaggedArgs rg) {
  __shared__ __align__(16) bf16_t smem[4 * TILE];
  const int lid = xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
  const int bh = lid / gridDim.x, qt = lid % gridDim.x;
  const int b = bh / heads, h = bh % heads;
  const SeqSpan sp = seq_span(rg, b, S);
  if (qt * QB >= sp.store_rows) return;
  const int delta_stride = S;  // delta stays [B*heads][S of the launch]
  S = sp.len;
  const long row0 = sp.row0;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int H = heads * 64;
  const bf16_t* K = qkv + row0 * ld_qkv + h * 64 + H;
  const bf16_t* V = K + H;
  const int q0 = qt * QB + wid * 32;

  // dO fragments (B operand of dP^T = V dO^T) and delta_q = <dO_q, O_q>
  bf16x8_t df[2][2];
  float delta[2];
  int qrow[2];
#pragma unroll
  for (int qg = 0; qg < 2; ++qg) {
    int q = q0 + qg * 16 + r;
    q = q < S ? q : S - 1;
    qrow[qg] = q;
    float acc = 0.f;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const long o = (row0 + q) * ld_ctx + h * 64 + kk * 32 + 8 * g;
      uint4 dv = *reinterpret_cast<const uint4*>(dctx + o);
      uint4 ov = *reinterpret_cast<const uint4*>(ctx + o);
      df[qg][kk] = __builtin_bit_cast(bf16x8_t, dv);
      const bf16_t* de = reinterpret_cast<const bf16_t*>(&dv);
      const bf16_t* oe = reinterpret_cast<const bf16_t*>(&ov);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc += bf16_bits_to_f32(de[e]) * bf16_bits_to_f32(oe[e]);
    }
    acc += __shfl_xor(acc, 16, 64);
    acc += __shfl_xor(acc, 32, 64);
    delta[qg] = acc;
    if (delta_out && g == 0 && q0 + qg * 16 + r < S) delta_out[(long)bh * delta_stride + q] = acc;
  }

  const ResOffsets ro = res_offsets(r, g);
  const unsigned smem_addr = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)smem;
  const int nt = (S + KT - 1) / KT;
  const bf16_t* Pz = P + (long)bh * sp.slot_rows * ldS;
  bf16_t* dSz = dS + (long)bh * sp.slot_rows * ldS;

  f32x4 o[2][4];
#pragma unroll
  for (int qg = 0; qg < 2; ++qg)
#pragma unroll
    for (int d = 0; d < 4; ++d) o[qg][d] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // P of the current tile is fetched one iteration ahead, before that iteration's LDS-DMA is
  // issued, so its wait never drains the DMA ring.
  uint4 pcur[2][2], pnext[2][2];
#pragma unroll
  for (int qg = 0; qg < 2; ++qg)
#pragma unroll
    for (int j = 0; j < 2; ++j) pcur[qg][j] = *reinterpret_cast<const uint4*>(Pz + (long)qrow[qg] * ldS + j * 32 + 8 * g);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  stage_tile(V, ld_qkv, 0, S, smem, wid, lane);          // slot 0: V (k-major in d for dP^T = V dO^T)
  stage_tile(K, ld_qkv, 0, S, smem + TILE, wid, lane);   // slot 1: K (transposed reads for dQ^T = K^T dS^T)
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    if (t + 1 < nt) {
#pragma unroll
      for (int qg = 0; qg < 2; ++qg)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          pnext[qg][j] = *reinterpret_cast<const uint4*>(Pz + (long)qrow[qg] * ldS + (t + 1) * KT + j * 32 + 8 * g);
      bf16_t* nb = smem + ((t + 1) & 1) * 2 * TILE;
      stage_tile(V, ld_qkv, (t + 1) * KT, S, nb, wid, lane);
      stage_tile(K, ld_qkv, (t + 1) * KT, S, nb + TILE, wid, lane);
    }
    const unsigned va = smem_addr + (t & 1) * 4 * TILE;
    const unsigned ka = va + 2 * TILE;
    u32x4 vf[4][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      vf[0][kk] = rd128<0>(va + ro.koff[0][kk]);
      vf[1][kk] = rd128<0>(va + ro.koff[1][kk]);
      vf[2][kk] = rd128<4096>(va + ro.koff[0][kk]);
      vf[3][kk] = rd128<4096>(va + ro.koff[1][kk]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    f32x4 dp[2][4];
#pragma unroll
    for (int qg = 0; qg < 2; ++qg)
#pragma unroll
      for (int blk = 0; blk < 4; ++blk) {
        f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vf[blk][0]), df[qg][0], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vf[blk][1]), df[qg][1], a, 0, 0, 0);
        dp[qg][blk] = a;
      }
    u32x2 klo[2][4], khi[2][4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      klo[0][d] = rdtr<0>(ka + ro.voff[0][d]);
      khi[0][d] = rdtr<0>(ka + ro.voff[1][d]);
      klo[1][d] = rdtr<4096>(ka + ro.voff[0][d]);
      khi[1][d] = rdtr<4096>(ka + ro.voff[1][d]);
    }
    bf16x8_t sf[2][2];
#pragma unroll
    for (int qg = 0; qg < 2; ++qg) {
      const int q = q0 + qg * 16 + r;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const uint4 pr = pcur[qg][j];
        const unsigned pw[4] = {pr.x, pr.y, pr.z, pr.w};
        float ds[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float pe = __uint_as_float((e & 1) ? (pw[e >> 1] & 0xffff0000u) : (pw[e >> 1] << 16));
          float v = pe * (dp[qg][2 * j + (e >> 2)][e & 3] - delta[qg]);
          if (q >= S) v = 0.f;  // (rows past the sequence: zeros)
          ds[e] = v;
          dp[qg][2 * j + (e >> 2)][e & 3] = v;
        }
        if (STORE_DS && q < sp.store_rows) VecIO<bf16_t, 8>::store(dSz + (long)q * ldS + t * KT + j * 32 + 8 * g, ds);
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8_t p8;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          p8[e] = (__bf16)dp[qg][2 * s2][e];
          p8[4 + e] = (__bf16)dp[qg][2 * s2 + 1][e];
        }
        sf[qg][s2] = p8;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const u32x4 kv = (u32x4){klo[s2][d][0], klo[s2][d][1], khi[s2][d][0], khi[s2][d][1]};
        const bf16x8_t kf = __builtin_bit_cast(bf16x8_t, kv);
#pragma unroll
        for (int qg = 0; qg < 2; ++qg) o[qg][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, sf[qg][s2], o[qg][d], 0, 0, 0);
      }
#pragma unroll
    for (int qg = 0; qg < 2; ++qg)
#pragma unroll
      for (int j = 0; j < 2; ++j) pcur[qg][j] = pnext[qg][j];
    __syncthreads();
  }
#pragma unroll
  for (int qg = 0; qg < 2; ++qg) {
    const int q = q0 + qg * 16 + r;
    if (q >= S) continue;
    bf16_t* dst = dq + (row0 + q) * ld_dq + h * 64 + 4 * g;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      float v[4] = {o[qg][d][0] * 0.125f, o[qg][d][1] * 0.125f, o[qg][d][2] * 0.125f, o[qg][d][3] * 0.125f};
      VecIO<bf16_t, 4>::store(dst + d * 16, v);
    }
  }
}


// Kernels for sequences of at most RES_KEYS keys keep all K and V tiles of the head in LDS: staged once, then no LDS-DMA in
// flight, no barrier and no drained wait while the keys are swept.  A workgroup is 8 waves = 256 query rows, every wave
// on its own after the one barrier behind the staging.
constexpr int RES_NT = 8;              // 64-key tiles that fit: K and V of 512 keys are 128 KiB
constexpr int RES_KEYS = RES_NT * KT;
constexpr int RES_QB = 256;            // query rows per workgroup

// Stages tiles [0, nt) of two [rows][64] operands into tile slots of 2 * TILE elements ([A | B] per tile).
__device__ __forceinline__ void stage_resident(const bf16_t* __restrict__ A, const bf16_t* __restrict__ Bm, long ld, int nt, int rows,
                                               bf16_t* smem, int wid, int nw, int lane) {
  const int rl = lane >> 3;
  const int c = (lane & 7) ^ rl;
  for (int i = wid; i < nt * 16; i += nw) {
    const int tile = i >> 4, op = (i >> 3) & 1, pi = i & 7;
    int rg = tile * KT + 8 * pi + rl;
    rg = rg < rows ? rg : rows - 1;
    const bf16_t* src = (op ? Bm : A) + (long)rg * ld + c * 8;
    __builtin_amdgcn_global_load_lds(src, smem + (tile * 2 + op) * TILE + pi * 512, 16, 0, 0);
  }
}

// attn_fwd_p_kernel for sequences of at most RES_KEYS keys, K and V resident.  A wave takes its two 16-query groups one
// after the other and keeps the whole score row of a group in registers (NT tiles x 16 floats per lane): Q K^T and the
// exponentials are computed once, where the two sweeps of the ring kernel compute both twice; no barrier after the staging.
// NT = the sequence's tile count rounded up to even (the register array wants a compile-time extent): a tile past the
// last one is staged like the others (clamped rows), its keys carry the -inf of the padding, its P is not stored.
template <int NT>
__device__ __forceinline__ void fwd_p_rows(const bf16_t* __restrict__ Q, bf16_t* __restrict__ Pz, bf16_t* __restrict__ ctx_rows,
                                           int S, int nt, int store_rows, int ldS, long ld_qkv, long ld_ctx, int q0, int r, int g,
                                           const ResOffsets& ro, unsigned smem_addr, unsigned mask_addr) {
#pragma unroll 1
  for (int qg = 0; qg < 2; ++qg) {
    const int q = q0 + qg * 16 + r;
    if (q0 + qg * 16 >= store_rows) break;  // (groups past the padded rows of a short sequence; no barrier follows)
    const int qc = q < S ? q : S - 1;
    bf16x8_t qf[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint4 v = *reinterpret_cast<const uint4*>(Q + (long)qc * ld_qkv + kk * 32 + 8 * g);
      qf[kk] = __builtin_bit_cast(bf16x8_t, v);
    }
    // ---- scores of the whole row, scaled and masked; row max
    f32x4 s[NT][4];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const unsigned ka = smem_addr + t * 4 * TILE;
      const unsigned ma = mask_addr + (unsigned)((t * KT + 8 * g) * 4);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        u32x4 mraw[2], kf[2][2];
        if (j == 0) {
          mraw[0] = rd128<0>(ma);
          mraw[1] = rd128<16>(ma);
        } else {
          mraw[0] = rd128<128>(ma);
          mraw[1] = rd128<144>(ma);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          if (j == 0) {
            kf[0][kk] = rd128<0>(ka + ro.koff[0][kk]);
            kf[1][kk] = rd128<0>(ka + ro.koff[1][kk]);
          } else {
            kf[0][kk] = rd128<4096>(ka + ro.koff[0][kk]);
            kf[1][kk] = rd128<4096>(ka + ro.koff[1][kk]);
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b1 = 0; b1 < 2; ++b1) {
          f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf[b1][0]), qf[0], a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf[b1][1]), qf[1], a, 0, 0, 0);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            a[e] = a[e] * 0.125f + __uint_as_float(mraw[b1][e]);
            mx = fmaxf(mx, a[e]);
          }
          s[t][2 * j + b1] = a;
        }
        __builtin_amdgcn_sched_barrier(0);  // (a few fragments at a time: the score row already fills half the registers)
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    // ---- exponentials, once; row sum
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int blk = 0; blk < 4; ++blk)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float pe = __expf(s[t][blk][e] - mx);
          s[t][blk][e] = pe;
          sum += pe;
        }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv_l = q < S ? 1.0f / sum : 0.f;  // (rows past the sequence: zeros)
    // ---- P and P V
    f32x4 o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) o[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const unsigned va = smem_addr + t * 4 * TILE + 2 * TILE;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        u32x2 vlo[4], vhi[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          if (j == 0) {
            vlo[d] = rdtr<0>(va + ro.voff[0][d]);
            vhi[d] = rdtr<0>(va + ro.voff[1][d]);
          } else {
            vlo[d] = rdtr<4096>(va + ro.voff[0][d]);
            vhi[d] = rdtr<4096>(va + ro.voff[1][d]);
          }
        }
        float pv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) pv[e] = s[t][2 * j + (e >> 2)][e & 3] * inv_l;
        if (t < nt && q < store_rows) VecIO<bf16_t, 8>::store(Pz + (long)q * ldS + t * KT + j * 32 + 8 * g, pv);
        bf16x8_t p8;
#pragma unroll
        for (int e = 0; e < 8; ++e) p8[e] = (__bf16)pv[e];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const u32x4 vv = (u32x4){vlo[d][0], vlo[d][1], vhi[d][0], vhi[d][1]};
          o[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vv), p8, o[d], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (q < S) {
      bf16_t* dst = ctx_rows + (long)q * ld_ctx + 4 * g;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        float v[4] = {o[d][0], o[d][1], o[d][2], o[d][3]};
        VecIO<bf16_t, 4>::store(dst + d * 16, v);
      }
    }
  }
}

__global__ __launch_bounds__(512, 1) void attn_fwd_p_res_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ mask,
                                                                 bf16_t* __restrict__ P, bf16_t* __restrict__ ctx, int S, int ldS,
                                                                 int heads, long ld_qkv, long ld_ctx, RaggedArgs rg) {
  __shared__ __align__(16) bf16_t smem[RES_NT * 2 * TILE + 2 * RES_KEYS];  // tile t: [K | V]; then the additive key mask (fp32)
  const int lid = xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
  const int bh = lid / gridDim.x, qt = lid % gridDim.x;
  const int b = bh / heads, h = bh % heads;
  const SeqSpan sp = seq_span(rg, b, S);
  if (qt * RES_QB >= sp.store_rows) return;
  S = sp.len;
  const long row0 = sp.row0;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int H = heads * 64;
  const bf16_t* Q = qkv + row0 * ld_qkv + h * 64;
  const float* mrow = mask ? mask + row0 : nullptr;
  const int q0 = qt * RES_QB + wid * 32;
  const int nt = (S + KT - 1) / KT;
  const int ntv = (nt + 1) & ~1;
  stage_resident(Q + H, Q + 2 * H, ld_qkv, ntv, S, smem, wid, 8, lane);
  {
    float* ml = reinterpret_cast<float*>(smem + RES_NT * 2 * TILE);
    for (int k = threadIdx.x; k < ntv * KT; k += 512) {
      float a = 0.f;
      if (k >= S) a = -INFINITY;
      else if (mrow) a = (1.0f - mrow[k]) * -FLT_MAX;
      ml[k] = a;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const ResOffsets ro = res_offsets(r, g);
  const unsigned smem_addr = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)smem;
  const unsigned mask_addr = smem_addr + RES_NT * 2 * TILE * 2;
  bf16_t* Pz = P + (long)bh * sp.slot_rows * ldS;
  bf16_t* ctx_rows = ctx + row0 * ld_ctx + h * 64;
#define PG_FWD_P_ROWS(N) \
  fwd_p_rows<N>(Q, Pz, ctx_rows, S, nt, sp.store_rows, ldS, ld_qkv, ld_ctx, q0, r, g, ro, smem_addr, mask_addr)
  if (ntv == 8) PG_FWD_P_ROWS(8);
  else if (ntv == 6) PG_FWD_P_ROWS(6);
  else if (ntv == 4) PG_FWD_P_ROWS(4);
  else PG_FWD_P_ROWS(2);
#undef PG_FWD_P_ROWS
}


__device__ __forceinline__ void read_kmajor_frags(unsigned tile_addr, const unsigned (&koff)[2][2], u32x4 (&f)[4][2]) {
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    f[0][kk] = rd128<0>(tile_addr + koff[0][kk]);
    f[1][kk] = rd128<0>(tile_addr + koff[1][kk]);
    f[2][kk] = rd128<4096>(tile_addr + koff[0][kk]);
    f[3][kk] = rd128<4096>(tile_addr + koff[1][kk]);
  }
}

// Second half of the attention backward, key-stationary, from the stored P:
//   dV = P^T dO ;  dK = dS^T Q / 8  with  dS = P * (dO V^T - delta)  formed again in registers
// so P is read once and dS neither read nor (for this half) written.  One workgroup = 4 waves = 128 keys of one
// (sequence, head), a wave owns 32 of them (two 16-key groups) and keeps their dK^T and dV^T in 64 accumulator
// registers while the workgroup walks the queries in 64-row tiles: P [64][128 keys], Q [64][64], dO [64][64] and
// the 64 delta arrive by LDS-DMA into a two-buffer ring.
// Products are issued key-on-the-lane: dP[q][k] = dO_q . V_k with dO rows as the A operand (row m of block qb is
// query 32 (qb >> 1) + 8 (m >> 2) + 4 (qb & 1) + (m & 3), the order of res_offsets) and the wave's V rows, read once,
// as the B operand; a lane (k = lane & 15, g = lane >> 4) then holds, in blocks 2j and 2j + 1, the queries
// 32 j + 8 g .. + 7 of its key, which is the B operand of a 32-deep step of dV^T += dO^T P and dK^T += Q^T dS
// as it stands.  P reaches that layout through transposing LDS reads; -delta is the initial accumulator of dP.
constexpr int DKV_KB = 128;                        // keys per workgroup
constexpr int DKV_STAGE = 2 * TILE + 64 * DKV_KB + 128;  // bf16 elements: dO | Q | P | delta (64 floats)

__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ P,
                                                               const bf16_t* __restrict__ dctx, const float* __restrict__ delta,
                                                               bf16_t* __restrict__ dkv, int S, int ldS, int heads, long ld_qkv,
                                                               long ld_ctx, long ld_dq, RaggedArgs rg) {
  __shared__ __align__(16) bf16_t smem[2 * DKV_STAGE];
  const int lid = xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
  const int bh = lid / gridDim.x, kb = lid % gridDim.x;
  const int b = bh / heads, h = bh % heads;
  const SeqSpan sp = seq_span(rg, b, S);
  if (kb * DKV_KB >= sp.len) return;
  const float* drow = delta + (long)bh * S;  // delta stays [B*heads][S of the launch]
  S = sp.len;
  const long row0 = sp.row0;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int H = heads * 64;
  const bf16_t* Q = qkv + row0 * ld_qkv + h * 64;
  const bf16_t* V = Q + 2 * H;
  const bf16_t* dO = dctx + row0 * ld_ctx + h * 64;
  const bf16_t* Pz = P + (long)bh * sp.slot_rows * ldS + kb * DKV_KB;
  const int nq = (S + 63) / 64;
  const int chunk_max = (ldS - kb * DKV_KB) / 8 - 1;  // last 16-byte chunk of a P row that belongs to it

  // this wave's pieces of a stage: 4 of P (4 rows x 256 B each), 2 of dO and 2 of Q (8 rows x 128 B each); wave 0: delta
  auto stage = [&](int t, bf16_t* buf) {
    const int q0 = t * 64;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = wid * 4 + j;
      int row = q0 + 4 * i + (lane >> 4);
      row = row < sp.store_rows ? row : sp.store_rows - 1;
      const int rl = (4 * i + (lane >> 4));
      int c = (lane & 15) ^ (2 * ((rl & 3) + 4 * ((rl >> 3) & 1)));
      c = c < chunk_max ? c : chunk_max;
      __builtin_amdgcn_global_load_lds(Pz + (long)row * ldS + c * 8, buf + 2 * TILE + i * 512, 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int i = wid * 2 + j;
      int row = q0 + 8 * i + (lane >> 3);
      row = row < S ? row : S - 1;
      const int c = (lane & 7) ^ (lane >> 3);
      __builtin_amdgcn_global_load_lds(dO + (long)row * ld_ctx + c * 8, buf + i * 512, 16, 0, 0);
      __builtin_amdgcn_global_load_lds(Q + (long)row * ld_qkv + c * 8, buf + TILE + i * 512, 16, 0, 0);
    }
    if (wid == 0) {
      int row = q0 + lane;
      row = row < S ? row : S - 1;
      __builtin_amdgcn_global_load_lds(drow + row, buf + 2 * TILE + 64 * DKV_KB, 4, 0, 0);
    }
  };

  // V rows of this wave's keys (B operand of dP): lane holds V[k][32 kk + 8 g .. + 8]
  bf16x8_t vfr[2][2];
  int key[2];
#pragma unroll
  for (int kg = 0; kg < 2; ++kg) {
    key[kg] = kb * DKV_KB + wid * 32 + kg * 16 + r;
    const int kc = key[kg] < S ? key[kg] : S - 1;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint4 v = *reinterpret_cast<const uint4*>(V + (long)kc * ld_qkv + kk * 32 + 8 * g);
      vfr[kg][kk] = __builtin_bit_cast(bf16x8_t, v);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  const ResOffsets ro = res_offsets(r, g);
  unsigned poff[2][2];  // transposing read of P: rows 8 g + 4 hi + (0..3) (+ 32 j), the wave's 16-key group kg
  {
    const int qq = r >> 2, pp = r & 3;
    const int sw = 2 * (qq + 4 * (g & 1));
#pragma unroll
    for (int kg = 0; kg < 2; ++kg)
#pragma unroll
      for (int hi = 0; hi < 2; ++hi) {
        const int row = 8 * g + 4 * hi + qq;
        const int c = 2 * (wid * 2 + kg) + (pp >> 1);
        poff[kg][hi] = (unsigned)(row * 256 + ((c ^ sw) << 4) + 8 * (pp & 1));
      }
  }
  const unsigned smem_addr = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)smem;

  f32x4 oV[2][4], oK[2][4];
#pragma unroll
  for (int kg = 0; kg < 2; ++kg)
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      oV[kg][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
      oK[kg][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

  stage(0, smem);
  __syncthreads();
  for (int t = 0; t < nq; ++t) {
    if (t + 1 < nq) stage(t + 1, smem + ((t + 1) & 1) * DKV_STAGE);
    const unsigned da = smem_addr + (t & 1) * DKV_STAGE * 2;  // dO tile
    const unsigned qa = da + 2 * TILE;                         // Q tile
    const unsigned pa = qa + 2 * TILE;                         // P tile
    const unsigned la = pa + 64 * DKV_KB * 2;                  // delta
    // ---- dP - delta for both key groups
    f32x4 dp[2][4];
    {
      u32x4 af[4][2];
      read_kmajor_frags(da, ro.koff, af);
      u32x4 dl[4];
      dl[0] = rd128<0>(la + 32 * g);
      dl[1] = rd128<16>(la + 32 * g);
      dl[2] = rd128<128>(la + 32 * g);
      dl[3] = rd128<144>(la + 32 * g);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kg = 0; kg < 2; ++kg)
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
          f32x4 a = (f32x4){-__uint_as_float(dl[qb][0]), -__uint_as_float(dl[qb][1]), -__uint_as_float(dl[qb][2]),
                            -__uint_as_float(dl[qb][3])};
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[qb][0]), vfr[kg][0], a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[qb][1]), vfr[kg][1], a, 0, 0, 0);
          dp[kg][qb] = a;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- P (raw bits: B operand of dV^T) and dS = P (dP - delta) (B operand of dK^T)
    bf16x8_t pf[2][2], sf[2][2];
    {
      u32x2 pr[2][2][2];
#pragma unroll
      for (int kg = 0; kg < 2; ++kg)
#pragma unroll
        for (int hi = 0; hi < 2; ++hi) {
          pr[kg][0][hi] = rdtr<0>(pa + poff[kg][hi]);
          pr[kg][1][hi] = rdtr<8192>(pa + poff[kg][hi]);
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      const bool ragged_end = (t == nq - 1) && (S & 63);  // rows past the sequence inside the last tile: P counts as 0
#pragma unroll
      for (int kg = 0; kg < 2; ++kg)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          unsigned pw[4] = {pr[kg][j][0][0], pr[kg][j][0][1], pr[kg][j][1][0], pr[kg][j][1][1]};
          if (ragged_end) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const int q = t * 64 + 32 * j + 8 * g + e;
              if (q >= S) pw[e >> 1] &= (e & 1) ? 0x0000ffffu : 0xffff0000u;
            }
          }
          bf16x8_t s8;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float pe = __uint_as_float((e & 1) ? (pw[e >> 1] & 0xffff0000u) : (pw[e >> 1] << 16));
            s8[e] = (__bf16)(pe * dp[kg][2 * j + (e >> 2)][e & 3]);
          }
          sf[kg][j] = s8;
          pf[kg][j] = __builtin_bit_cast(bf16x8_t, ((u32x4){pw[0], pw[1], pw[2], pw[3]}));
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- dV^T += dO^T P ; dK^T += Q^T dS
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      const unsigned ta = which ? qa : da;
      u32x2 lo[2][4], hi[2][4];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        lo[0][d] = rdtr<0>(ta + ro.voff[0][d]);
        hi[0][d] = rdtr<0>(ta + ro.voff[1][d]);
        lo[1][d] = rdtr<4096>(ta + ro.voff[0][d]);
        hi[1][d] = rdtr<4096>(ta + ro.voff[1][d]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const u32x4 av = (u32x4){lo[j][d][0], lo[j][d][1], hi[j][d][0], hi[j][d][1]};
          const bf16x8_t a8 = __builtin_bit_cast(bf16x8_t, av);
#pragma unroll
          for (int kg = 0; kg < 2; ++kg) {
            if (which) oK[kg][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, sf[kg][j], oK[kg][d], 0, 0, 0);
            else oV[kg][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, pf[kg][j], oV[kg][d], 0, 0, 0);
          }
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }
  // ---- store: lane holds dK[k][h*64 + 16d + 4g .. +4] and the same of dV
#pragma unroll
  for (int kg = 0; kg < 2; ++kg) {
    if (key[kg] >= S) continue;
    bf16_t* dst = dkv + (row0 + key[kg]) * ld_dq + H + h * 64 + 4 * g;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      float kv[4] = {oK[kg][d][0] * 0.125f, oK[kg][d][1] * 0.125f, oK[kg][d][2] * 0.125f, oK[kg][d][3] * 0.125f};
      float vv[4] = {oV[kg][d][0], oV[kg][d][1], oV[kg][d][2], oV[kg][d][3]};
      VecIO<bf16_t, 4>::store(dst + d * 16, kv);
      VecIO<bf16_t, 4>::store(dst + H + d * 16, vv);
    }
  }
}

// ---- gradient-penalty passes over the stored probabilities ---------------------------------
// Both kernels below keep the query-on-the-lane formulation: a score-shaped product X^T = T R^T
// takes its A operand from a 64-key LDS tile T (K, dK, V or dV rows, k-major in the head
// dimension) and its B operand from per-lane row fragments R[q][32kk + 8g .. +8] read once from
// global memory; lane (q = lane & 15, g = lane >> 4) then holds, in blocks 2j and 2j + 1, the 8 consecutive keys
// 32 j + 8 g .. + 7 of query q (the key order of res_offsets).

// Row fragments of one [T][ld] activation (head h): both 16-query groups, both k-steps.
__device__ __forceinline__ void load_row_frags(const bf16_t* __restrict__ base, long ld, const int (&qrow)[2], int g,
                                               bf16x8_t (&f)[2][2]) {
#pragma unroll
  for (int qg = 0; qg < 2; ++qg)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint4 v = *reinterpret_cast<const uint4*>(base + (long)qrow[qg] * ld + kk * 32 + 8 * g);
      f[qg][kk] = __builtin_bit_cast(bf16x8_t, v);
    }
}

// <a_q, b_q> over the 64 head-dim values of each of the lane's two query rows.
__device__ __forceinline__ void row_dots(const bf16x8_t (&a)[2][2], const bf16x8_t (&b)[2][2], float (&out)[2]) {
#pragma unroll
  for (int qg = 0; qg < 2; ++qg) {
    float acc = 0.f;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc += (float)a[qg][kk][e] * (float)b[qg][kk][e];
    acc += __shfl_xor(acc, 16, 64);
    acc += __shfl_xor(acc, 32, 64);
    out[qg] = acc;
  }
}

template <bool ZERO>
__device__ __forceinline__ void score_product(const u32x4 (&tf)[4][2], const bf16x8_t (&rf)[2][2], f32x4 (&acc)[2][4]) {
#pragma unroll
  for (int qg = 0; qg < 2; ++qg)
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
      f32x4 a = ZERO ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[qg][blk];
      a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, tf[blk][0]), rf[qg][0], a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, tf[blk][1]), rf[qg][1], a, 0, 0, 0);
      acc[qg][blk] = a;
    }
}

template <bool ZERO>
__device__ __forceinline__ void score_product1(const u32x4 (&tf)[4][2], const bf16x8_t (&rf)[2], f32x4 (&acc)[4]) {
#pragma unroll
  for (int blk = 0; blk < 4; ++blk) {
    f32x4 a = ZERO ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[blk];
    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, tf[blk][0]), rf[0], a, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, tf[blk][1]), rf[1], a, 0, 0, 0);
    acc[blk] = a;
  }
}

// Tangent (forward-mode) pass of one attention head block, given P:
//   dS = (dQ K^T + Q dK^T) / 8 ; rho_q = sum_k P dS ; dP = P (dS - rho)
//   dctx = dP V + P dV = (P dS) V - rho ctx + P dV
// One sweep over the keys; nothing score-shaped is written.  rho ([B*heads][S], fp32) is kept
// for the second-order reverse pass, which recomputes dS from the same operands.
__global__ __launch_bounds__(256, 2) void attn_jvp_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ tqkv,
                                                           const bf16_t* __restrict__ P, const bf16_t* __restrict__ ctx,
                                                           bf16_t* __restrict__ ctxdot, float* __restrict__ rho, int S, int ldS,
                                                           int heads, long ld_qkv, long ld_ctx, long ld_out, RaggedArgs rg) {
  __shared__ __align__(16) bf16_t smem[8 * TILE];  // [buffer][K | dK | V | dV]
  const int lid = xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
  const int bh = lid / gridDim.x, qt = lid % gridDim.x;
  const int b = bh / heads, h = bh % heads;
  const SeqSpan sp = seq_span(rg, b, S);
  if (qt * QB >= sp.len) return;
  const int rho_stride = S;  // rho stays [B*heads][S of the launch]
  S = sp.len;
  const long row0 = sp.row0;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int H = heads * 64;
  const bf16_t* Q = qkv + row0 * ld_qkv + h * 64;
  const bf16_t* Qd = tqkv + row0 * ld_qkv + h * 64;
  const int q0 = qt * QB + wid * 32;
  int qrow[2];
#pragma unroll
  for (int qg = 0; qg < 2; ++qg) {
    const int q = q0 + qg * 16 + r;
    qrow[qg] = q < S ? q : S - 1;
  }
  bf16x8_t qf[2][2], qdf[2][2];
  load_row_frags(Q, ld_qkv, qrow, g, qf);
  load_row_frags(Qd, ld_qkv, qrow, g, qdf);

  const ResOffsets ro = res_offsets(r, g);
  const unsigned smem_addr = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)smem;
  const int nt = (S + KT - 1) / KT;
  const bf16_t* Pz = P + (long)bh * sp.slot_rows * ldS;

  f32x4 o[2][4];
  float racc[2] = {0.f, 0.f};
#pragma unroll
  for (int qg = 0; qg < 2; ++qg)
#pragma unroll
    for (int d = 0; d < 4; ++d) o[qg][d] = (f32x4){0.f, 0.f, 0.f, 0.f};

  uint4 pcur[2][2], pnext[2][2];
#pragma unroll
  for (int qg = 0; qg < 2; ++qg)
#pragma unroll
    for (int j = 0; j < 2; ++j) pcur[qg][j] = *reinterpret_cast<const uint4*>(Pz + (long)qrow[qg] * ldS + j * 32 + 8 * g);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  stage_tile(Q + H, ld_qkv, 0, S, smem, wid, lane);
  stage_tile(Qd + H, ld_qkv, 0, S, smem + TILE, wid, lane);
  stage_tile(Q + 2 * H, ld_qkv, 0, S, smem + 2 * TILE, wid, lane);
  stage_tile(Qd + 2 * H, ld_qkv, 0, S, smem + 3 * TILE, wid, lane);
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    if (t + 1 < nt) {
#pragma unroll
      for (int qg = 0; qg < 2; ++qg)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          pnext[qg][j] = *reinterpret_cast<const uint4*>(Pz + (long)qrow[qg] * ldS + (t + 1) * KT + j * 32 + 8 * g);
      bf16_t* nb = smem + ((t + 1) & 1) * 4 * TILE;
      stage_tile(Q + H, ld_qkv, (t + 1) * KT, S, nb, wid, lane);
      stage_tile(Qd + H, ld_qkv, (t + 1) * KT, S, nb + TILE, wid, lane);
      stage_tile(Q + 2 * H, ld_qkv, (t + 1) * KT, S, nb + 2 * TILE, wid, lane);
      stage_tile(Qd + 2 * H, ld_qkv, (t + 1) * KT, S, nb + 3 * TILE, wid, lane);
    }
    const unsigned ka = smem_addr + (t & 1) * 8 * TILE;
    const unsigned kda = ka + 2 * TILE, va = ka + 4 * TILE, vda = ka + 6 * TILE;
    f32x4 ds[2][4];
    {
      u32x4 kf[4][2];
      read_kmajor_frags(ka, ro.koff, kf);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      score_product<true>(kf, qdf, ds);
      __builtin_amdgcn_sched_barrier(0);
      read_kmajor_frags(kda, ro.koff, kf);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      score_product<false>(kf, qf, ds);
      __builtin_amdgcn_sched_barrier(0);
    }
    // V^T fragments in flight under the elementwise part
    u32x2 vlo[2][4], vhi[2][4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      vlo[0][d] = rdtr<0>(va + ro.voff[0][d]);
      vhi[0][d] = rdtr<0>(va + ro.voff[1][d]);
      vlo[1][d] = rdtr<4096>(va + ro.voff[0][d]);
      vhi[1][d] = rdtr<4096>(va + ro.voff[1][d]);
    }
    bf16x8_t pdf[2][2], pf[2][2];
#pragma unroll
    for (int qg = 0; qg < 2; ++qg) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const uint4 pr = pcur[qg][j];
        const unsigned pw[4] = {pr.x, pr.y, pr.z, pr.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float pe = __uint_as_float((e & 1) ? (pw[e >> 1] & 0xffff0000u) : (pw[e >> 1] << 16));
          const float pd = pe * ds[qg][2 * j + (e >> 2)][e & 3] * 0.125f;
          racc[qg] += pd;
          ds[qg][2 * j + (e >> 2)][e & 3] = pd;
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8_t p8;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          p8[e] = (__bf16)ds[qg][2 * s2][e];
          p8[4 + e] = (__bf16)ds[qg][2 * s2 + 1][e];
        }
        pdf[qg][s2] = p8;
        const u32x4 raw = (u32x4){pcur[qg][s2].x, pcur[qg][s2].y, pcur[qg][s2].z, pcur[qg][s2].w};
        pf[qg][s2] = __builtin_bit_cast(bf16x8_t, raw);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const u32x4 vv = (u32x4){vlo[s2][d][0], vlo[s2][d][1], vhi[s2][d][0], vhi[s2][d][1]};
#pragma unroll
        for (int qg = 0; qg < 2; ++qg)
          o[qg][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vv), pdf[qg][s2], o[qg][d], 0, 0, 0);
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int d = 0; d < 4; ++d) {  // dV^T into the same registers
      vlo[0][d] = rdtr<0>(vda + ro.voff[0][d]);
      vhi[0][d] = rdtr<0>(vda + ro.voff[1][d]);
      vlo[1][d] = rdtr<4096>(vda + ro.voff[0][d]);
      vhi[1][d] = rdtr<4096>(vda + ro.voff[1][d]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const u32x4 ww = (u32x4){vlo[s2][d][0], vlo[s2][d][1], vhi[s2][d][0], vhi[s2][d][1]};
#pragma unroll
        for (int qg = 0; qg < 2; ++qg)
          o[qg][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ww), pf[qg][s2], o[qg][d], 0, 0, 0);
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int qg = 0; qg < 2; ++qg)
#pragma unroll
      for (int j = 0; j < 2; ++j) pcur[qg][j] = pnext[qg][j];
    __syncthreads();
  }
#pragma unroll
  for (int qg = 0; qg < 2; ++qg) {
    float rq = racc[qg];
    rq += __shfl_xor(rq, 16, 64);
    rq += __shfl_xor(rq, 32, 64);
    const int q = q0 + qg * 16 + r;
    if (q >= S) continue;
    if (g == 0) rho[(long)bh * rho_stride + q] = rq;
    const bf16_t* cs = ctx + (row0 + q) * ld_ctx + h * 64 + 4 * g;
    bf16_t* dst = ctxdot + (row0 + q) * ld_out + h * 64 + 4 * g;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      float c4[4];
      VecIO<bf16_t, 4>::load(cs + d * 16, c4);
      float v[4] = {o[qg][d][0] - rq * c4[0], o[qg][d][1] - rq * c4[1], o[qg][d][2] - rq * c4[2], o[qg][d][3] - rq * c4[3]};
      VecIO<bf16_t, 4>::store(dst + d * 16, v);
    }
  }
}

// Score side of the second-order reverse pass of one attention head block.  With P, the
// tangent dS (recomputed), gamma = gctx V^T, X = ctxbar V^T + gctx dV^T and the row scalars
//   rho = sum P dS (from the tangent pass), c = <gctx, ctx>, e = <ctxbar, ctx> + <gctx, dctx> - rho c
// it writes   Sbar = P (X + gamma (dS - rho) - c dS - e)   and   dP = P (dS - rho)
// (both bf16, [B*heads][S][ldS]); the key-side contractions over them are plain GEMMs.
__global__ __launch_bounds__(256, 2) void attn_bwd2_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ tqkv,
                                                            const bf16_t* __restrict__ P, const bf16_t* __restrict__ ctx,
                                                            const bf16_t* __restrict__ ctxdot, const bf16_t* __restrict__ gctx,
                                                            const bf16_t* __restrict__ ctxbar, const float* __restrict__ rho,
                                                            bf16_t* __restrict__ Sbar, bf16_t* __restrict__ Pdot, int S, int ldS,
                                                            int heads, long ld_qkv, long ld_ctx, RaggedArgs rg) {
  __shared__ __align__(16) bf16_t smem[8 * TILE];  // [buffer][K | dK | V | dV]
  const int lid = xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
  const int bh = lid / gridDim.x, qt = lid % gridDim.x;
  const int b = bh / heads, h = bh % heads;
  const SeqSpan sp = seq_span(rg, b, S);
  if (qt * QB >= sp.store_rows) return;
  const int rho_stride = S;
  S = sp.len;
  const long row0 = sp.row0;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int H = heads * 64;
  const bf16_t* Q = qkv + row0 * ld_qkv + h * 64;
  const bf16_t* Qd = tqkv + row0 * ld_qkv + h * 64;
  const int q0 = qt * QB + wid * 32;
  int qrow[2];
#pragma unroll
  for (int qg = 0; qg < 2; ++qg) {
    const int q = q0 + qg * 16 + r;
    qrow[qg] = q < S ? q : S - 1;
  }
  bf16x8_t qf[2][2], qdf[2][2], gf[2][2], cbf[2][2];
  load_row_frags(Q, ld_qkv, qrow, g, qf);
  load_row_frags(Qd, ld_qkv, qrow, g, qdf);
  load_row_frags(gctx + row0 * ld_ctx + h * 64, ld_ctx, qrow, g, gf);
  load_row_frags(ctxbar + row0 * ld_ctx + h * 64, ld_ctx, qrow, g, cbf);
  float rq[2], cq[2], eq[2];
  {
    bf16x8_t cf[2][2], cdf[2][2];
    load_row_frags(ctx + row0 * ld_ctx + h * 64, ld_ctx, qrow, g, cf);
    load_row_frags(ctxdot + row0 * ld_ctx + h * 64, ld_ctx, qrow, g, cdf);
    float e1[2], e2[2];
    row_dots(gf, cf, cq);
    row_dots(cbf, cf, e1);
    row_dots(gf, cdf, e2);
#pragma unroll
    for (int qg = 0; qg < 2; ++qg) {
      rq[qg] = rho[(long)bh * rho_stride + qrow[qg]];
      eq[qg] = e1[qg] + e2[qg] - rq[qg] * cq[qg];
    }
  }

  const ResOffsets ro = res_offsets(r, g);
  const unsigned smem_addr = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)smem;
  const int nt = (S + KT - 1) / KT;
  const bf16_t* Pz = P + (long)bh * sp.slot_rows * ldS;
  bf16_t* Sbz = Sbar + (long)bh * sp.slot_rows * ldS;
  bf16_t* Pdz = Pdot + (long)bh * sp.slot_rows * ldS;

  uint4 pcur[2][2], pnext[2][2];
#pragma unroll
  for (int qg = 0; qg < 2; ++qg)
#pragma unroll
    for (int j = 0; j < 2; ++j) pcur[qg][j] = *reinterpret_cast<const uint4*>(Pz + (long)qrow[qg] * ldS + j * 32 + 8 * g);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  stage_tile(Q + H, ld_qkv, 0, S, smem, wid, lane);
  stage_tile(Qd + H, ld_qkv, 0, S, smem + TILE, wid, lane);
  stage_tile(Q + 2 * H, ld_qkv, 0, S, smem + 2 * TILE, wid, lane);
  stage_tile(Qd + 2 * H, ld_qkv, 0, S, smem + 3 * TILE, wid, lane);
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    if (t + 1 < nt) {
#pragma unroll
      for (int qg = 0; qg < 2; ++qg)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          pnext[qg][j] = *reinterpret_cast<const uint4*>(Pz + (long)qrow[qg] * ldS + (t + 1) * KT + j * 32 + 8 * g);
      bf16_t* nb = smem + ((t + 1) & 1) * 4 * TILE;
      stage_tile(Q + H, ld_qkv, (t + 1) * KT, S, nb, wid, lane);
      stage_tile(Qd + H, ld_qkv, (t + 1) * KT, S, nb + TILE, wid, lane);
      stage_tile(Q + 2 * H, ld_qkv, (t + 1) * KT, S, nb + 2 * TILE, wid, lane);
      stage_tile(Qd + 2 * H, ld_qkv, (t + 1) * KT, S, nb + 3 * TILE, wid, lane);
    }
    const unsigned ka = smem_addr + (t & 1) * 8 * TILE;
    const unsigned kda = ka + 2 * TILE, va = ka + 4 * TILE, vda = ka + 6 * TILE;
    // one 16-query group at a time: three score-shaped accumulators of one group are live, the
    // tile fragments are re-read for the second group (the kernel is bound by its stores)
#pragma unroll
    for (int qg = 0; qg < 2; ++qg) {
      f32x4 ds[4], gm[4], xx[4];
      u32x4 tf[4][2];
      read_kmajor_frags(ka, ro.koff, tf);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      score_product1<true>(tf, qdf[qg], ds);
      __builtin_amdgcn_sched_barrier(0);
      read_kmajor_frags(kda, ro.koff, tf);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      score_product1<false>(tf, qf[qg], ds);
      __builtin_amdgcn_sched_barrier(0);
      read_kmajor_frags(va, ro.koff, tf);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      score_product1<true>(tf, gf[qg], gm);
      score_product1<true>(tf, cbf[qg], xx);
      __builtin_amdgcn_sched_barrier(0);
      read_kmajor_frags(vda, ro.koff, tf);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      score_product1<false>(tf, gf[qg], xx);
      __builtin_amdgcn_sched_barrier(0);
      const int q = q0 + qg * 16 + r;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const uint4 pr = pcur[qg][j];
        const unsigned pw[4] = {pr.x, pr.y, pr.z, pr.w};
        float sb[8], pd[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int blk = 2 * j + (e >> 2), ee = e & 3;
          const float pv = __uint_as_float((e & 1) ? (pw[e >> 1] & 0xffff0000u) : (pw[e >> 1] << 16));
          const float d = ds[blk][ee] * 0.125f;
          const float pbar = xx[blk][ee] + gm[blk][ee] * (d - rq[qg]) - cq[qg] * d;
          sb[e] = q < S ? pv * (pbar - eq[qg]) : 0.f;  // (rows past the sequence: zeros)
          pd[e] = q < S ? pv * (d - rq[qg]) : 0.f;
        }
        if (q < sp.store_rows) {
          const long o = (long)q * ldS + t * KT + j * 32 + 8 * g;
          VecIO<bf16_t, 8>::store(Sbz + o, sb);
          VecIO<bf16_t, 8>::store(Pdz + o, pd);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int qg = 0; qg < 2; ++qg)
#pragma unroll
      for (int j = 0; j < 2; ++j) pcur[qg][j] = pnext[qg][j];
    __syncthreads();
  }
}

}  // namespace

extern "C" int pg_attention_key_lists(const float* mask, const int32_t* seq_off, int nseq, int S, int32_t* key_idx,
                                      int32_t* key_cnt, void* stream) {
  if (!mask || !key_idx || !key_cnt || nseq <= 0 || S <= 0) return PG_ERR_ARG;
  hipLaunchKernelGGL(key_compact_kernel, dim3(nseq), dim3(256), 0, (hipStream_t)stream, mask, seq_off, S, key_idx, key_cnt);
  return hipGetLastError() == hipSuccess ? PG_OK : PG_ERR_LAUNCH;
}

extern "C" int pg_attention_fwd_keys(int dtype, const void* qkv, const float* mask, void* ctx, const int32_t* seq_off, int nseq,
                                     int S, int heads, int64_t ld_qkv, int64_t ld_ctx, const int32_t* key_idx,
                                     const int32_t* key_cnt, void* stream) {
  if (!qkv || !ctx || nseq <= 0 || S <= 0 || heads <= 0 || ((key_idx == nullptr) != (key_cnt == nullptr))) return PG_ERR_ARG;
  if (dtype != PG_BF16 || S > MAX_S) return PG_ERR_UNSUPPORTED;
  if ((ld_qkv % 8) || (ld_ctx % 4) || (((uintptr_t)qkv) & 15) || (((uintptr_t)ctx) & 7)) return PG_ERR_ALIGN;
  dim3 grid((S + QB - 1) / QB, nseq * heads);
  hipLaunchKernelGGL(attn_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, mask, (bf16_t*)ctx, S, heads,
                     (long)ld_qkv, (long)ld_ctx, seq_off, key_idx, key_cnt);
  return hipGetLastError() == hipSuccess ? PG_OK : PG_ERR_LAUNCH;
}

extern "C" int pg_attention_fwd(int dtype, const void* qkv, const float* mask, void* ctx, int B, int S, int heads,
                                int64_t ld_qkv, int64_t ld_ctx, void* stream) {
  return pg_attention_fwd_keys(dtype, qkv, mask, ctx, nullptr, B, S, heads, ld_qkv, ld_ctx, nullptr, nullptr, stream);
}

extern "C" int pg_attention_fwd_packed(int dtype, const void* qkv, const float* mask, void* ctx, const int32_t* seq_off,
                                       int nseq, int max_len, int heads, int64_t ld_qkv, int64_t ld_ctx, void* stream) {
  if (!seq_off) return PG_ERR_ARG;
  return pg_attention_fwd_keys(dtype, qkv, mask, ctx, seq_off, nseq, max_len, heads, ld_qkv, ld_ctx, nullptr, nullptr, stream);
}

namespace {
struct RaggedHost { RaggedArgs rg; int S, ldS; dim3 grid; };
// Rectangular launch (seq_off == NULL: B sequences of S) or ragged launch (nseq sequences of at most max_len).
inline RaggedHost ragged_host(const int32_t* seq_off, int n, int S, int tok0, int heads) {
  RaggedHost h;
  h.S = S;
  h.ldS = pg_score_ld(S);
  h.rg = RaggedArgs{seq_off, tok0, seq_off ? h.ldS : S};
  h.grid = dim3(((seq_off ? h.ldS : S) + QB - 1) / QB, n * heads);
  return h;
}

int launch_fwd_p(int dtype, const void* qkv, const float* mask, void* P, void* ctx, const int32_t* seq_off, int n, int S,
                 int tok0, int heads, int64_t ld_qkv, int64_t ld_ctx, void* stream) {
  if (!qkv || !ctx || !P || n <= 0 || S <= 0 || heads <= 0 || tok0 < 0) return PG_ERR_ARG;
  if (dtype != PG_BF16 || S > MAX_S) return PG_ERR_UNSUPPORTED;
  if ((ld_qkv % 8) || (ld_ctx % 4) || (((uintptr_t)qkv) & 15) || (((uintptr_t)ctx) & 7) || (((uintptr_t)P) & 15)) return PG_ERR_ALIGN;
  const RaggedHost h = ragged_host(seq_off, n, S, tok0, heads);
  if (pg_internal_attn_resident && S <= RES_KEYS) {
    const dim3 grid(((seq_off ? h.ldS : S) + RES_QB - 1) / RES_QB, n * heads);
    hipLaunchKernelGGL(attn_fwd_p_res_kernel, grid, dim3(512), 0, (hipStream_t)stream, (const bf16_t*)qkv, mask, (bf16_t*)P,
                       (bf16_t*)ctx, S, h.ldS, heads, (long)ld_qkv, (long)ld_ctx, h.rg);
    return hipGetLastError() == hipSuccess ? PG_OK : PG_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(attn_fwd_p_kernel, h.grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, mask, (bf16_t*)P,
                     (bf16_t*)ctx, S, h.ldS, heads, (long)ld_qkv, (long)ld_ctx, h.rg);
  return hipGetLastError() == hipSuccess ? PG_OK : PG_ERR_LAUNCH;
}

// dS == NULL: not written (then delta != NULL: a caller without dS has the key side formed from P and delta).
int launch_bwd_dq(int dtype, const void* qkv, const void* P, const void* ctx, const void* dctx, void* dS, void* dq, float* delta,
                  const int32_t* seq_off, int n, int S, int tok0, int heads, int64_t ld_qkv, int64_t ld_ctx, int64_t ld_dq,
                  void* stream) {
  if (!qkv || !P || !ctx || !dctx || (!dS && !delta) || !dq || n <= 0 || S <= 0 || heads <= 0 || tok0 < 0) return PG_ERR_ARG;
  if (dtype != PG_BF16 || S > MAX_S) return PG_ERR_UNSUPPORTED;
  if ((ld_qkv % 8) || (ld_ctx % 8) || (ld_dq % 4)) return PG_ERR_ALIGN;
  if ((((uintptr_t)qkv) | ((uintptr_t)ctx) | ((uintptr_t)dctx)) & 15) return PG_ERR_ALIGN;
  if (((uintptr_t)dq) & 7) return PG_ERR_ALIGN;
  if ((((uintptr_t)P) | ((uintptr_t)dS)) & 15) return PG_ERR_ALIGN;
  const RaggedHost h = ragged_host(seq_off, n, S, tok0, heads);
  if (dS)
    hipLaunchKernelGGL(attn_bwd_dq_kernel<true>, h.grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, (const bf16_t*)P,
                       (const bf16_t*)ctx, (const bf16_t*)dctx, (bf16_t*)dS, (bf16_t*)dq, delta, S, h.ldS, heads, (long)ld_qkv,
                       (long)ld_ctx, (long)ld_dq, h.rg);
  else
    hipLaunchKernelGGL(attn_bwd_dq_kernel<false>, h.grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, (const bf16_t*)P,
                       (const bf16_t*)ctx, (const bf16_t*)dctx, (bf16_t*)nullptr, (bf16_t*)dq, delta, S, h.ldS, heads,
                       (long)ld_qkv, (long)ld_ctx, (long)ld_dq, h.rg);
  return hipGetLastError() == hipSuccess ? PG_OK : PG_ERR_LAUNCH;
}

int launch_bwd_dkv(int dtype, const void* qkv, const void* P, const void* dctx, const float* delta, void* dkv,
                   const int32_t* seq_off, int n, int S, int tok0, int heads, int64_t ld_qkv, int64_t ld_ctx, int64_t ld_dq,
                   void* stream) {
  if (!qkv || !P || !dctx || !delta || !dkv || n <= 0 || S <= 0 || heads <= 0 || tok0 < 0) return PG_ERR_ARG;
  if (dtype != PG_BF16 || S > MAX_S) return PG_ERR_UNSUPPORTED;
  if ((ld_qkv % 8) || (ld_ctx % 8) || (ld_dq % 4)) return PG_ERR_ALIGN;
  if ((((uintptr_t)qkv) | ((uintptr_t)dctx) | ((uintptr_t)P)) & 15) return PG_ERR_ALIGN;
  if ((((uintptr_t)dkv) & 7) || (((uintptr_t)delta) & 3)) return PG_ERR_ALIGN;
  const RaggedHost h = ragged_host(seq_off, n, S, tok0, heads);
  const dim3 grid(((seq_off ? h.ldS : S) + DKV_KB - 1) / DKV_KB, n * heads);
  hipLaunchKernelGGL(attn_bwd_dkv_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, (const bf16_t*)P,
                     (const bf16_t*)dctx, delta, (bf16_t*)dkv, S, h.ldS, heads, (long)ld_qkv, (long)ld_ctx, (long)ld_dq, h.rg);
  return hipGetLastError() == hipSuccess ? PG_OK : PG_ERR_LAUNCH;
}

int launch_jvp(int dtype, const void* qkv, const void* tqkv, const void* P, const void* ctx, void* ctxdot, float* rho,
               const int32_t* seq_off, int n, int S, int tok0, int heads, int64_t ld_qkv, int64_t ld_ctx, int64_t ld_out,
               void* stream) {
  if (!qkv || !tqkv || !P || !ctx || !ctxdot || !rho || n <= 0 || S <= 0 || heads <= 0 || tok0 < 0) return PG_ERR_ARG;
  if (dtype != PG_BF16 || S > MAX_S) return PG_ERR_UNSUPPORTED;
  if ((ld_qkv % 8) || (ld_ctx % 4) || (ld_out % 4)) return PG_ERR_ALIGN;
  if ((((uintptr_t)qkv) | ((uintptr_t)tqkv)) & 15) return PG_ERR_ALIGN;
  if ((((uintptr_t)ctx) | ((uintptr_t)ctxdot)) & 7) return PG_ERR_ALIGN;
  if (((uintptr_t)P) & 15) return PG_ERR_ALIGN;
  const RaggedHost h = ragged_host(seq_off, n, S, tok0, heads);
  hipLaunchKernelGGL(attn_jvp_kernel, h.grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, (const bf16_t*)tqkv,
                     (const bf16_t*)P, (const bf16_t*)ctx, (bf16_t*)ctxdot, rho, S, h.ldS, heads, (long)ld_qkv, (long)ld_ctx,
                     (long)ld_out, h.rg);
  return hipGetLastError() == hipSuccess ? PG_OK : PG_ERR_LAUNCH;
}

int launch_bwd2(int dtype, const void* qkv, const void* tqkv, const void* P, const void* ctx, const void* ctxdot,
                const void* gctx, const void* ctxbar, const float* rho, void* Sbar, void* Pdot, const int32_t* seq_off, int n,
                int S, int tok0, int heads, int64_t ld_qkv, int64_t ld_ctx, void* stream) {
  if (!qkv || !tqkv || !P || !ctx || !ctxdot || !gctx || !ctxbar || !rho || !Sbar || !Pdot) return PG_ERR_ARG;
  if (n <= 0 || S <= 0 || heads <= 0 || tok0 < 0) return PG_ERR_ARG;
  if (dtype != PG_BF16 || S > MAX_S) return PG_ERR_UNSUPPORTED;
  if ((ld_qkv % 8) || (ld_ctx % 8)) return PG_ERR_ALIGN;
  if ((((uintptr_t)qkv) | ((uintptr_t)tqkv) | ((uintptr_t)ctx) | ((uintptr_t)ctxdot) | ((uintptr_t)gctx) | ((uintptr_t)ctxbar)) & 15)
    return PG_ERR_ALIGN;
  if ((((uintptr_t)P) | ((uintptr_t)Sbar) | ((uintptr_t)Pdot)) & 15) return PG_ERR_ALIGN;
  const RaggedHost h = ragged_host(seq_off, n, S, tok0, heads);
  hipLaunchKernelGGL(attn_bwd2_kernel, h.grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, (const bf16_t*)tqkv,
                     (const bf16_t*)P, (const bf16_t*)ctx, (const bf16_t*)ctxdot, (const bf16_t*)gctx, (const bf16_t*)ctxbar, rho,
                     (bf16_t*)Sbar, (bf16_t*)Pdot, S, h.ldS, heads, (long)ld_qkv, (long)ld_ctx, h.rg);
  return hipGetLastError() == hipSuccess ? PG_OK : PG_ERR_LAUNCH;
}
}  // namespace

extern "C" int pg_attention_fwd_p(int dtype, const void* qkv, const float* mask, void* P, void* ctx, int B, int S, int heads,
                                  int64_t ld_qkv, int64_t ld_ctx, void* stream) {
  return launch_fwd_p(dtype, qkv, mask, P, ctx, nullptr, B, S, 0, heads, ld_qkv, ld_ctx, stream);
}
extern "C" int pg_attention_fwd_p_ragged(int dtype, const void* qkv, const float* mask, void* P, void* ctx,
                                         const int32_t* seq_off, int nseq, int max_len, int tok0, int heads, int64_t ld_qkv,
                                         int64_t ld_ctx, void* stream) {
  if (!seq_off) return PG_ERR_ARG;
  return launch_fwd_p(dtype, qkv, mask, P, ctx, seq_off, nseq, max_len, tok0, heads, ld_qkv, ld_ctx, stream);
}

extern "C" int pg_attention_bwd_dq(int dtype, const void* qkv, const void* P, const void* ctx, const void* dctx, void* dS,
                                   void* dq, int B, int S, int heads, int64_t ld_qkv, int64_t ld_ctx, int64_t ld_dq,
                                   void* stream) {
  if (!dS) return PG_ERR_ARG;
  return launch_bwd_dq(dtype, qkv, P, ctx, dctx, dS, dq, nullptr, nullptr, B, S, 0, heads, ld_qkv, ld_ctx, ld_dq, stream);
}
extern "C" int pg_attention_bwd_dq_ragged(int dtype, const void* qkv, const void* P, const void* ctx, const void* dctx,
                                          void* dS, void* dq, const int32_t* seq_off, int nseq, int max_len, int tok0,
                                          int heads, int64_t ld_qkv, int64_t ld_ctx, int64_t ld_dq, void* stream) {
  if (!seq_off) return PG_ERR_ARG;
  if (!dS) return PG_ERR_ARG;
  return launch_bwd_dq(dtype, qkv, P, ctx, dctx, dS, dq, nullptr, seq_off, nseq, max_len, tok0, heads, ld_qkv, ld_ctx, ld_dq, stream);
}

extern "C" int pg_attention_bwd(int dtype, const void* qkv, const void* P, const void* ctx, const void* dctx, void* dS,
                                void* dqkv, float* delta, const int32_t* seq_off, int nseq, int max_len, int tok0, int heads,
                                int64_t ld_qkv, int64_t ld_ctx, int64_t ld_dq, void* stream) {
  if (!delta) return PG_ERR_ARG;
  const int rc = launch_bwd_dq(dtype, qkv, P, ctx, dctx, dS, dqkv, delta, seq_off, nseq, max_len, tok0, heads, ld_qkv, ld_ctx,
                               ld_dq, stream);
  if (rc != PG_OK) return rc;
  return launch_bwd_dkv(dtype, qkv, P, dctx, delta, dqkv, seq_off, nseq, max_len, tok0, heads, ld_qkv, ld_ctx, ld_dq, stream);
}

extern "C" int pg_attention_jvp(int dtype, const void* qkv, const void* tqkv, const void* P, const void* ctx, void* ctxdot,
                                float* rho, int B, int S, int heads, int64_t ld_qkv, int64_t ld_ctx, int64_t ld_out,
                                void* stream) {
  return launch_jvp(dtype, qkv, tqkv, P, ctx, ctxdot, rho, nullptr, B, S, 0, heads, ld_qkv, ld_ctx, ld_out, stream);
}
extern "C" int pg_attention_jvp_ragged(int dtype, const void* qkv, const void* tqkv, const void* P, const void* ctx,
                                       void* ctxdot, float* rho, const int32_t* seq_off, int nseq, int max_len, int tok0,
                                       int heads, int64_t ld_qkv, int64_t ld_ctx, int64_t ld_out, void* stream) {
  if (!seq_off) return PG_ERR_ARG;
  return launch_jvp(dtype, qkv, tqkv, P, ctx, ctxdot, rho, seq_off, nseq, max_len, tok0, heads, ld_qkv, ld_ctx, ld_out, stream);
}

extern "C" int pg_attention_bwd2(int dtype, const void* qkv, const void* tqkv, const void* P, const void* ctx,
                                 const void* ctxdot, const void* gctx, const void* ctxbar, const float* rho, void* Sbar,
                                 void* Pdot, int B, int S, int heads, int64_t ld_qkv, int64_t ld_ctx, void* stream) {
  return launch_bwd2(dtype, qkv, tqkv, P, ctx, ctxdot, gctx, ctxbar, rho, Sbar, Pdot, nullptr, B, S, 0, heads, ld_qkv, ld_ctx,
                     stream);
}
extern "C" int pg_attention_bwd2_ragged(int dtype, const void* qkv, const void* tqkv, const void* P, const void* ctx,
                                        const void* ctxdot, const void* gctx, const void* ctxbar, const float* rho, void* Sbar,
                                        void* Pdot, const int32_t* seq_off, int nseq, int max_len, int tok0, int heads,
                                        int64_t ld_qkv, int64_t ld_ctx, void* stream) {
  if (!seq_off) return PG_ERR_ARG;
  return launch_bwd2(dtype, qkv, tqkv, P, ctx, ctxdot, gctx, ctxbar, rho, Sbar, Pdot, seq_off, nseq, max_len, tok0, heads,
                     ld_qkv, ld_ctx, stream);
}
