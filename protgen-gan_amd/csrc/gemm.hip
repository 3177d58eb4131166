// Batched strided GEMM on the CDNA4 matrix cores.
//
//   C[m][n] = epilogue( alpha * sum_k A(m,k) * B(k,n) )
//
// One kernel template covers every contraction of the ProtBERT path: the Linear layers
// (X W^T), their input gradients (dY W), their weight gradients (dY^T X), and the
// attention products over materialised score matrices.  Each operand is either
// "k-major" (the reduction index is the contiguous one: A[m][k], B[n][k]) or
// "m-major" (the free index is contiguous: A stored [k][m], B stored [k][n]).
//
// Tiling: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave,
// 4x4 MFMA tiles of 16x16).  bf16 uses v_mfma_f32_16x16x32_bf16, fp32 uses
// v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain).  Tiles are staged through LDS with a
// two-buffer pipeline (global loads for tile t+1 are issued before the MFMAs of tile t
// and written to the other buffer afterwards).  k-major fragments are read with
// ds_read_b128; m-major bf16 fragments with the transposing ds_read_b64_tr_b16.
// The MFMA is issued with the operands swapped (D = Bt * At) so that a lane owns four
// consecutive n of one output row, which makes every epilogue access an 8/16-byte vector.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include <deque>
#include <mutex>
#include "pg_common.h"
#include "protgen_hip.h"

namespace {

// Optional sampled timing of GEMM launches with HIP events on the launch stream (used by the
// benchmark to report achieved FLOP/s per kernel variant without a profiler attached).
struct ProfEntry { hipEvent_t a, b; int variant; double flops; };
// (a deque: an entry's address stays valid while launches of another host thread append theirs)
struct ProfState { int every = 0; long counter = 0; long launches = 0; std::deque<ProfEntry> entries; std::mutex mu; };
ProfState g_prof;

#include "gemm_shared.h"
constexpr int BM = 128, BN = 128, NTHREADS = 256;

template <typename T> struct GemmCfg;
template <> struct GemmCfg<bf16_t> {
  static constexpr int BK = 64, EPC = 8, KSTEP = 32;
  static constexpr int LDK = BK + 8;     // k-major row stride (elements): 144 B
  static constexpr int LDM = 128 + 16;   // m-major row stride (elements): 288 B
};
template <> struct GemmCfg<float> {
  static constexpr int BK = 16, EPC = 4, KSTEP = 4;
  static constexpr int LDK = BK + 4;     // 80 B
  static constexpr int LDM = 128 + 4;    // 528 B
};

template <typename T> constexpr int op_elems() {
  return (BM * GemmCfg<T>::LDK > GemmCfg<T>::BK * GemmCfg<T>::LDM) ? BM * GemmCfg<T>::LDK
                                                                   : GemmCfg<T>::BK * GemmCfg<T>::LDM;
}


// ---- global -> register staging of one operand tile -------------------------------------
// CHECK = false is the hot path: every chunk of the tile lies inside the reduction extent, so
// the loads are unconditional and stay in flight together until the LDS write that consumes
// them.  CHECK = true (only the last, ragged k-tile) zero-fills whatever lies beyond K.
template <typename T, bool KMAJOR, bool CHECK>
__device__ __forceinline__ void stage_load(const T* __restrict__ base, long ld, int row0, int rows_max,
                                           int k0, int K, uint4 (&regs)[GemmCfg<T>::BK * 128 / GemmCfg<T>::EPC / NTHREADS]) {
  using C = GemmCfg<T>;
  constexpr int NCH = C::BK * 128 / C::EPC / NTHREADS;
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = t + i * NTHREADS;
    if (KMAJOR) {
      constexpr int CPR = C::BK / C::EPC;
      const int r = c / CPR, kc = c % CPR;
      const int k = k0 + kc * C::EPC;
      int rg = row0 + r;
      rg = rg < rows_max ? rg : rows_max - 1;
      if (!CHECK) {
        regs[i] = *reinterpret_cast<const uint4*>(base + (long)rg * ld + k);
      } else {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (k < K) {
          v = *reinterpret_cast<const uint4*>(base + (long)rg * ld + k);
          if (k + C::EPC > K) {  // chunk straddles the end of the reduction: zero the tail
            T* e = reinterpret_cast<T*>(&v);
#pragma unroll
            for (int j = 0; j < C::EPC; ++j)
              if (k + j >= K) e[j] = (T)0;
          }
        }
        regs[i] = v;
      }
    } else {
      constexpr int CPR = 128 / C::EPC;
      const int kr = c / CPR, mc = c % CPR;
      const int k = k0 + kr;
      int m = row0 + mc * C::EPC;
      m = m <= rows_max - C::EPC ? m : rows_max - C::EPC;
      if (!CHECK) {
        regs[i] = *reinterpret_cast<const uint4*>(base + (long)k * ld + m);
      } else {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (k < K) v = *reinterpret_cast<const uint4*>(base + (long)k * ld + m);
        regs[i] = v;
      }
    }
  }
}

template <typename T, bool KMAJOR>
__device__ __forceinline__ void stage_store(T* lds, const uint4 (&regs)[GemmCfg<T>::BK * 128 / GemmCfg<T>::EPC / NTHREADS]) {
  using C = GemmCfg<T>;
  constexpr int NCH = C::BK * 128 / C::EPC / NTHREADS;
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = t + i * NTHREADS;
    if (KMAJOR) {
      constexpr int CPR = C::BK / C::EPC;
      const int r = c / CPR, kc = c % CPR;
      *reinterpret_cast<uint4*>(lds + r * C::LDK + kc * C::EPC) = regs[i];
    } else {
      constexpr int CPR = 128 / C::EPC;
      const int kr = c / CPR, mc = c % CPR;
      *reinterpret_cast<uint4*>(lds + kr * C::LDM + mc * C::EPC) = regs[i];
    }
  }
}

// ---- LDS -> MFMA fragments --------------------------------------------------------------
// bf16 fragment of the 16x16x32 MFMA: lane (r = lane&15, g = lane>>4) holds 8 values
// X[row0 + r][kk*32 + 8g + j], j = 0..7.
template <bool KMAJOR>
__device__ __forceinline__ bf16x8_t frag_bf16(const bf16_t* lds, int row0, int kk, int lane) {
  using C = GemmCfg<bf16_t>;
  const int r = lane & 15, g = lane >> 4;
  if (KMAJOR) {
    uint4 v = *reinterpret_cast<const uint4*>(lds + (row0 + r) * C::LDK + kk * 32 + 8 * g);
    return __builtin_bit_cast(bf16x8_t, v);
  } else {
    // Transposing read: within a 16-lane group, lane 4q+p addresses row q, columns 4p..4p+3
    // of a 4x16 block and receives column (lane&15) of its 4 rows.
    const int q = r >> 2, p = r & 3;
    const bf16_t* a0 = lds + (kk * 32 + 8 * g + q) * C::LDM + row0 + 4 * p;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * C::LDM));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
  }
}
// fp32 fragment of the 16x16x4 MFMA: lane holds X[row0 + r][kk*4 + g].
template <bool KMAJOR>
__device__ __forceinline__ float frag_f32(const float* lds, int row0, int kk, int lane) {
  using C = GemmCfg<float>;
  const int r = lane & 15, g = lane >> 4;
  if (KMAJOR) return lds[(row0 + r) * C::LDK + kk * 4 + g];
  return lds[(kk * 4 + g) * C::LDM + row0 + r];
}

template <bool AK, bool BK_>
__device__ __forceinline__ void mma_tile_bf16(const bf16_t* la, const bf16_t* lb, f32x4 (&acc)[4][4], int wr,
                                              int wc, int lane) {
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    bf16x8_t af[4], bfr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = frag_bf16<AK>(la, wr * 64 + i * 16, kk, lane);
#pragma unroll
    for (int j = 0; j < 4; ++j) bfr[j] = frag_bf16<BK_>(lb, wc * 64 + j * 16, kk, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
  }
}
template <bool AK, bool BK_>
__device__ __forceinline__ void mma_tile_f32(const float* la, const float* lb, f32x4 (&acc)[4][4], int wr, int wc,
                                             int lane) {
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    float af[4], bfr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = frag_f32<AK>(la, wr * 64 + i * 16, kk, lane);
#pragma unroll
    for (int j = 0; j < 4; ++j) bfr[j] = frag_f32<BK_>(lb, wc * 64 + j * 16, kk, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bfr[j], af[i], acc[i][j], 0, 0, 0);
  }
}

template <typename T, typename TC, bool AK, bool BK_>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(GemmArgs g) {
  using C = GemmCfg<T>;
  constexpr int OPE = op_elems<T>();
  constexpr int NCH = C::BK * 128 / C::EPC / NTHREADS;
  __shared__ __align__(16) T smem[4 * OPE];

  const int z = blockIdx.z;
  const int zo = z / g.batch_inner, zi = z % g.batch_inner;
  const T* A = reinterpret_cast<const T*>(g.A) + zo * g.sAo + zi * g.sAi;
  const T* B = reinterpret_cast<const T*>(g.B) + zo * g.sBo + zi * g.sBi;
  TC* Cp = reinterpret_cast<TC*>(g.C) + zo * g.sCo + zi * g.sCi;

  const int bm = blockIdx.y * BM, bn = blockIdx.x * BN;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wr = wid >> 1, wc = wid & 1;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  uint4 ra[NCH], rb[NCH];
  const int nt = (g.K + C::BK - 1) / C::BK;
  const bool ragged = (g.K % C::BK) != 0;  // only the last k-tile can reach beyond K
  if (ragged && nt == 1) {
    stage_load<T, AK, true>(A, g.lda, bm, g.M, 0, g.K, ra);
    stage_load<T, BK_, true>(B, g.ldb, bn, g.N, 0, g.K, rb);
  } else {
    stage_load<T, AK, false>(A, g.lda, bm, g.M, 0, g.K, ra);
    stage_load<T, BK_, false>(B, g.ldb, bn, g.N, 0, g.K, rb);
  }
  stage_store<T, AK>(smem, ra);
  stage_store<T, BK_>(smem + OPE, rb);
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    T* la = smem + cur * 2 * OPE;
    T* lb = la + OPE;
    if (t + 1 < nt) {
      if (ragged && t + 2 == nt) {
        stage_load<T, AK, true>(A, g.lda, bm, g.M, (t + 1) * C::BK, g.K, ra);
        stage_load<T, BK_, true>(B, g.ldb, bn, g.N, (t + 1) * C::BK, g.K, rb);
      } else {
        stage_load<T, AK, false>(A, g.lda, bm, g.M, (t + 1) * C::BK, g.K, ra);
        stage_load<T, BK_, false>(B, g.ldb, bn, g.N, (t + 1) * C::BK, g.K, rb);
      }
    }
    if constexpr (sizeof(T) == 2)
      mma_tile_bf16<AK, BK_>(reinterpret_cast<const bf16_t*>(la), reinterpret_cast<const bf16_t*>(lb), acc, wr, wc, lane);
    else
      mma_tile_f32<AK, BK_>(reinterpret_cast<const float*>(la), reinterpret_cast<const float*>(lb), acc, wr, wc, lane);
    if (t + 1 < nt) {
      T* na = smem + (cur ^ 1) * 2 * OPE;
      stage_store<T, AK>(na, ra);
      stage_store<T, BK_>(na + OPE, rb);
    }
    __syncthreads();
  }

  gemm_epilogue<T, TC, 4>(acc, g, Cp, g.vec_ok && bn + BN <= g.N, bm + wr * 64, bn + wc * 64, lane);
}

template <typename T, typename TC>
int launch_layout(const pg_gemm_desc* d, const GemmArgs& a, dim3 grid, hipStream_t st) {
  if (d->a_kmajor && d->b_kmajor) hipLaunchKernelGGL((gemm_kernel<T, TC, true, true>), grid, dim3(NTHREADS), 0, st, a);
  else if (d->a_kmajor && !d->b_kmajor) hipLaunchKernelGGL((gemm_kernel<T, TC, true, false>), grid, dim3(NTHREADS), 0, st, a);
  else if (!d->a_kmajor && d->b_kmajor) hipLaunchKernelGGL((gemm_kernel<T, TC, false, true>), grid, dim3(NTHREADS), 0, st, a);
  else hipLaunchKernelGGL((gemm_kernel<T, TC, false, false>), grid, dim3(NTHREADS), 0, st, a);
  return hipGetLastError() == hipSuccess ? PG_OK : PG_ERR_LAUNCH;
}

// ======================================================================================
// bf16 fast path: operand tiles go global -> LDS directly (global_load_lds_dwordx4, no
// staging registers, no ds_write), two LDS buffers, the loads of k-tile t+1 in flight under
// the MFMAs of tile t.  A wave instruction writes 1 KiB of LDS linearly (lane * 16 B), so
// the bank-conflict swizzle is applied to the per-lane SOURCE address and undone by the
// same XOR on the fragment read:
//   k-major tile [128 rows][64 bf16] (128-B rows):  chunk c of row r sits at position c ^ (r & 7)
//   m-major tile [64 k][128 bf16]   (256-B rows):  chunk c of row k sits at position c ^ swz(k),
//                                                  swz(k) = ((k & 3) << 2) | ((k >> 2) & 3)
// Requires K % 64 == 0 (whole tiles only: nothing can be masked on the way into LDS).
#include "gemm_tile256.h"  // GBK, GT_ELEMS, swizzles, fragment offsets / reads, the 256-tile k-tile body

template <bool KMAJOR>
__device__ __forceinline__ void glds_stage(const bf16_t* __restrict__ base, long ld, int row0, int rows_max, int k0,
                                           bf16_t* lds_tile, int wid, int lane) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = wid * 4 + j;  // wave instruction index 0..15, 1 KiB of the tile each
    const bf16_t* src;
    if (KMAJOR) {
      const int r = 8 * i + (lane >> 3);
      const int c = (lane & 7) ^ (lane >> 3);  // (r & 7) == lane >> 3
      int rg = row0 + r;
      rg = rg < rows_max ? rg : rows_max - 1;
      src = base + (long)rg * ld + k0 + c * 8;
    } else {
      const int k = 4 * i + (lane >> 4);
      const int c = (lane & 15) ^ swz_m(k);
      int m = row0 + c * 8;
      m = m <= rows_max - 8 ? m : rows_max - 8;
      src = base + (long)(k0 + k) * ld + m;
    }
    __builtin_amdgcn_global_load_lds(src, lds_tile + i * 512, 16, 0, 0);
  }
}


template <bool AK, bool BK_>
__device__ __forceinline__ void glds_compute(unsigned a_addr, unsigned b_addr, const FragOff<AK>& fa, const FragOff<BK_>& fb,
                                             f32x4 (&acc)[4][4]) {
  u32x4 a0[4], b0[4], a1[4], b1[4];
  read_frags<AK, 0>(a_addr, fa, a0);
  read_frags<BK_, 0>(b_addr, fb, b0);
  read_frags<AK, 1>(a_addr, fa, a1);
  read_frags<BK_, 1>(b_addr, fb, b1);
  constexpr int second = (AK ? 4 : 8) + (BK_ ? 4 : 8);  // reads issued for k-step 1
  if constexpr (second <= 8) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
  else if constexpr (second <= 12) asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
  else asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, b0[j]), __builtin_bit_cast(bf16x8_t, a0[i]),
                                                          acc[i][j], 0, 0, 0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, b1[j]), __builtin_bit_cast(bf16x8_t, a1[i]),
                                                          acc[i][j], 0, 0, 0);
}


// NST = 2: two LDS buffers, one k-tile in flight under the MFMAs, two workgroups per CU.
// NST = 4: four-buffer ring, up to three k-tiles in flight behind a counted vmcnt and a raw
//          s_barrier (one barrier per k-tile), one workgroup per CU: for grids of about one
//          workgroup per CU, where nothing else hides the HBM/L2 latency.
template <typename TC, bool AK, bool BK_, int NST>
__device__ __forceinline__ void glds_body(bf16_t* smem, const GemmArgs& g, const bf16_t* A, const bf16_t* B, TC* Cp, int bm, int bn) {
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wid >> 1, wc = wid & 1;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // k-tiles [0, nt1) come from (A, B), k-tiles [nt1, nt) from the optional second segment (A2, B2)
  const int nt1 = g.K / GBK, nt = nt1 + g.K2 / GBK;
  const bf16_t* A2 = reinterpret_cast<const bf16_t*>(g.A2);
  const bf16_t* B2 = reinterpret_cast<const bf16_t*>(g.B2);
  auto stage = [&](int t, bf16_t* dst) {
    const bool second = t >= nt1;
    const int k0 = (second ? t - nt1 : t) * GBK;
    glds_stage<AK>(second ? A2 : A, g.lda, bm, g.M, k0, dst, wid, lane);
    glds_stage<BK_>(second ? B2 : B, g.ldb, bn, g.N, k0, dst + GT_ELEMS, wid, lane);
  };
  const FragOff<AK> fa = frag_offsets<AK>(wr, lane);
  const FragOff<BK_> fb = frag_offsets<BK_>(wc, lane);
  const unsigned smem_addr = lds_addr(smem);
  if constexpr (NST == 2) {
    stage(0, smem);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
      const unsigned la = smem_addr + (t & 1) * 4 * GT_ELEMS;
      if (t + 1 < nt) {
        stage(t + 1, smem + ((t + 1) & 1) * 2 * GT_ELEMS);
      }
      glds_compute<AK, BK_>(la, la + 2 * GT_ELEMS, fa, fb, acc);
      __syncthreads();
    }
  } else {
    // every wave issues 8 LDS-DMA loads per k-tile; vmcnt(8 * tiles allowed to stay in flight)
    const int pre = nt < NST - 1 ? nt : NST - 1;
    for (int t = 0; t < pre; ++t) {
      stage(t, smem + t * 2 * GT_ELEMS);
    }
    for (int t = 0; t < nt; ++t) {
      const int ahead = nt - 1 - t;  // tiles issued after tile t that may remain in flight
      if (ahead >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // tile t landed for every wave; buffer (t-1)%NST is free again
      if (t + NST - 1 < nt) {
        stage(t + NST - 1, smem + ((t + NST - 1) % NST) * 2 * GT_ELEMS);
      }
      const unsigned la = smem_addr + (t % NST) * 4 * GT_ELEMS;
      glds_compute<AK, BK_>(la, la + 2 * GT_ELEMS, fa, fb, acc);
    }
  }
  gemm_epilogue<bf16_t, TC, 4>(acc, g, Cp, g.vec_ok && bn + BN <= g.N, bm + wr * 64, bn + wc * 64, lane);
}


template <typename TC, bool AK, bool BK_, int NST>
__global__ __launch_bounds__(NTHREADS, NST == 2 ? 2 : 1) void gemm_glds_kernel(GemmArgs g) {
  __shared__ __align__(16) bf16_t smem[NST * 2 * GT_ELEMS];  // [buffer][A|B][16 KiB]
  const int gxy = gridDim.x * gridDim.y;
  const int lid = xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gxy * gridDim.z);
  const int z = lid / gxy, txy = lid % gxy;
  const int zo = z / g.batch_inner, zi = z % g.batch_inner;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(g.A) + zo * g.sAo + zi * g.sAi;
  const bf16_t* B = reinterpret_cast<const bf16_t*>(g.B) + zo * g.sBo + zi * g.sBi;
  TC* Cp = reinterpret_cast<TC*>(g.C) + zo * g.sCo + zi * g.sCi;
  // wide outputs (more than eight tile columns): groups of eight tile rows walked down the rows
  // first, so that the workgroups an XCD runs together share operand panels in its L2
  const int gx = gridDim.x, gy = gridDim.y;
  const int GM = gx > 8 ? g.group_m : 1;
  const int grp = txy / (GM * gx), first = grp * GM, rem = txy - grp * GM * gx;
  const int gsz = gy - first < GM ? gy - first : GM;
  glds_body<TC, AK, BK_, NST>(smem, g, A, B, Cp, (first + rem % gsz) * BM, (rem / gsz) * BN);
}

// Two independent products of the same layout in one launch (unbatched): workgroups [0, nb0)
// tile the first, the rest the second.  Used to pair a product that has too few tiles to cover
// the chip (the attention-output weight gradient: 64 tiles) with one that is ready at the same
// time (the QKV weight gradient: 192 tiles).
struct GemmArgs2 { GemmArgs p0, p1; int nb0, gx0, gx1; };

template <typename TC, bool AK, bool BK_, int NST>
__global__ __launch_bounds__(NTHREADS, NST == 2 ? 2 : 1) void gemm_glds_pair_kernel(GemmArgs2 a) {
  __shared__ __align__(16) bf16_t smem[NST * 2 * GT_ELEMS];
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const bool second = lid >= a.nb0;
  const GemmArgs& g = second ? a.p1 : a.p0;
  const int txy = second ? lid - a.nb0 : lid, gx = second ? a.gx1 : a.gx0;
  glds_body<TC, AK, BK_, NST>(smem, g, reinterpret_cast<const bf16_t*>(g.A), reinterpret_cast<const bf16_t*>(g.B),
                              reinterpret_cast<TC*>(g.C), (txy / gx) * BM, (txy % gx) * BN);
}

template <typename TC, int NST>
int launch_glds(const pg_gemm_desc* d, const GemmArgs& a, dim3 grid, hipStream_t st) {
  if (d->a_kmajor && d->b_kmajor) hipLaunchKernelGGL((gemm_glds_kernel<TC, true, true, NST>), grid, dim3(NTHREADS), 0, st, a);
  else if (d->a_kmajor && !d->b_kmajor) hipLaunchKernelGGL((gemm_glds_kernel<TC, true, false, NST>), grid, dim3(NTHREADS), 0, st, a);
  else if (!d->a_kmajor && d->b_kmajor) hipLaunchKernelGGL((gemm_glds_kernel<TC, false, true, NST>), grid, dim3(NTHREADS), 0, st, a);
  else hipLaunchKernelGGL((gemm_glds_kernel<TC, false, false, NST>), grid, dim3(NTHREADS), 0, st, a);
  return hipGetLastError() == hipSuccess ? PG_OK : PG_ERR_LAUNCH;
}


template <typename TC, int NST>
int launch_glds_pair(const pg_gemm_desc* d, const GemmArgs2& a, int blocks, hipStream_t st) {
  if (d->a_kmajor && d->b_kmajor) hipLaunchKernelGGL((gemm_glds_pair_kernel<TC, true, true, NST>), dim3(blocks), dim3(NTHREADS), 0, st, a);
  else if (d->a_kmajor && !d->b_kmajor) hipLaunchKernelGGL((gemm_glds_pair_kernel<TC, true, false, NST>), dim3(blocks), dim3(NTHREADS), 0, st, a);
  else if (!d->a_kmajor && d->b_kmajor) hipLaunchKernelGGL((gemm_glds_pair_kernel<TC, false, true, NST>), dim3(blocks), dim3(NTHREADS), 0, st, a);
  else hipLaunchKernelGGL((gemm_glds_pair_kernel<TC, false, false, NST>), dim3(blocks), dim3(NTHREADS), 0, st, a);
  return hipGetLastError() == hipSuccess ? PG_OK : PG_ERR_LAUNCH;
}

// ---- wide variant: 256 x 256 output tile, 512 threads (8 waves as 2 x 4, 128 x 64 per wave) --
// A 128x128 tile moves 256 operand rows through L2 per 128x128 outputs; at the matrix-core rate
// that is more than the L2 -> LDS path delivers, so the large-N products (N >= 2048: the
// feed-forward expansion, the fused QKV projection and the input gradient of the contraction)
// use a tile with twice the reuse.  One workgroup per CU, two waves per SIMD, two 64 KiB LDS
// stages (A and B each as two independent 128-row sub-tiles in the layouts above, so staging and
// fragment addressing are shared with the 128x128 kernel).  A wave keeps 128 accumulator
// registers and walks a k-tile in four chunks (k-step x row half), each chunk's fragment reads
// issued one chunk ahead of the MFMAs that consume them.


__device__ __forceinline__ void mfma_4x4(f32x4 (&acc)[8][4], int i0, const u32x4 (&a)[4], const u32x4 (&b)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      acc[i0 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, b[j]), __builtin_bit_cast(bf16x8_t, a[i]),
                                                               acc[i0 + i][j], 0, 0, 0);
}


template <bool AK, bool BK_, int PH, int DBG>
__device__ __forceinline__ void glds256_mainloop(const GemmArgs& g, const bf16_t* A, const bf16_t* B, const bf16_t* A2,
                                                 const bf16_t* B2, int bm, int bn, bf16_t* smem, int wid, int lane,
                                                 f32x4 (&acc)[8][4]) {
  const int wr = wid >> 2, wc = wid & 3, w4 = wid & 3, half = wid >> 2;
  // k-tiles [0, nt1) come from (A, B), k-tiles [nt1, nt) from the optional second segment (A2, B2)
  const int nt1 = g.K / GBK, nt = nt1 + g.K2 / GBK;
  const FragOff<AK> fa_lo = frag_offsets<AK>(0, lane), fa_hi = frag_offsets<AK>(1, lane);
  const FragOff<BK_> fb = frag_offsets<BK_>(wc & 1, lane);
  const unsigned smem_addr = lds_addr(smem);
  const unsigned a_off = wr * 2 * GT_ELEMS, b_off = (2 + (wc >> 1)) * 2 * GT_ELEMS;  // bytes inside a stage
  unsigned aoff[4], boff[4];
  piece_offsets<AK>(g.lda, bm, half, g.M, w4, lane, aoff);
  piece_offsets<BK_>(g.ldb, bn, half, g.N, w4, lane, boff);
  // uniform source bases of k-tile 0 and their per-tile strides (bytes)
  const char* a_src = reinterpret_cast<const char*>(AK ? A + (long)bm * g.lda : A + bm);
  const char* b_src = reinterpret_cast<const char*>(BK_ ? B + (long)bn * g.ldb : B + bn);
  const long a_step = AK ? GBK * 2 : (long)GBK * g.lda * 2, b_step = BK_ ? GBK * 2 : (long)GBK * g.ldb * 2;
  const int dst0 = half * GT_ELEMS + w4 * 4 * 512;  // this wave's first piece inside a stage (A part; B is 2 sub-tiles on)

  // k-tile 0 (nothing to overlap it with)
#pragma unroll
  for (int j = 0; j < 4; ++j) __builtin_amdgcn_global_load_lds(a_src + aoff[j], smem + dst0 + j * 512, 16, 0, 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) __builtin_amdgcn_global_load_lds(b_src + boff[j], smem + 2 * GT_ELEMS + dst0 + j * 512, 16, 0, 0);
  for (int t = 0; t < nt - 1; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // tile t landed for every wave; the other stage is no longer being read
    if (t + 1 == nt1) {  // the next k-tile opens the second segment
      a_src = reinterpret_cast<const char*>(AK ? A2 + (long)bm * g.lda : A2 + bm);
      b_src = reinterpret_cast<const char*>(BK_ ? B2 + (long)bn * g.ldb : B2 + bn);
    } else {
      a_src += a_step;
      b_src += b_step;
    }
    const unsigned st = smem_addr + (t & 1) * 8 * GT_ELEMS;
    bf16_t* nx = smem + ((t + 1) & 1) * 4 * GT_ELEMS + dst0;
    glds256_tile<AK, BK_, PH, !(DBG & 1), !(DBG & 2)>(st + a_off, st + b_off, fa_lo, fa_hi, fb, acc, a_src, b_src, aoff, boff, nx,
                                                     nx + 2 * GT_ELEMS);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const unsigned st = smem_addr + ((nt - 1) & 1) * 8 * GT_ELEMS;
  glds256_tile<AK, BK_, PH, false, !(DBG & 2)>(st + a_off, st + b_off, fa_lo, fa_hi, fb, acc, a_src, b_src, aoff, boff, smem, smem);
}

// One 256 x 256 output tile: tile index txy inside a gx x gy grid of tiles.  Tile order: groups of
// eight tile rows, walked down the rows first.  The ~32 workgroups an XCD runs at a time then cover
// about 8 x 4 tiles (12 distinct operand panels) instead of 2 x 16 (18): fewer panel fetches from
// beyond that XCD's L2 (fabric reads of the fill-pass W1 product measured 4.6x its operand bytes
// with the row-major order).
template <typename TC, bool AK, bool BK_, int DBG>
__device__ __forceinline__ void glds256_block(const GemmArgs& g, const bf16_t* A, const bf16_t* B, TC* Cp, int txy, int gx, int gy,
                                              bf16_t* smem) {
  const int GM = gx > 4 ? g.group_m : 1;  // (a tile row of four or fewer tiles is already a compact patch)
  const int grp = txy / (GM * gx), first = grp * GM, rem = txy - grp * GM * gx;
  const int gsz = gy - first < GM ? gy - first : GM;
  const int bm = (first + rem % gsz) * 256, bn = (rem / gsz) * 256;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wid >> 2, wc = wid & 3;
  const bf16_t* A2 = reinterpret_cast<const bf16_t*>(g.A2);
  const bf16_t* B2 = reinterpret_cast<const bf16_t*>(g.B2);

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // waves w and w + 4 share a SIMD: opposite piece phases
  if (wr == 0) glds256_mainloop<AK, BK_, 0, DBG>(g, A, B, A2, B2, bm, bn, smem, wid, lane, acc);
  else glds256_mainloop<AK, BK_, 1, DBG>(g, A, B, A2, B2, bm, bn, smem, wid, lane, acc);
  gemm_epilogue<bf16_t, TC, 8>(acc, g, Cp, g.vec_ok && bn + 256 <= g.N, bm + wr * 128, bn + wc * 64, lane);
}

template <typename TC, bool AK, bool BK_, int DBG = 0>
__global__ __launch_bounds__(NT256, 1) void gemm_glds256_kernel(GemmArgs g) {
  __shared__ __align__(16) bf16_t smem[2 * 4 * GT_ELEMS];  // [stage][A0 A1 B0 B1][16 KiB]
  const int gxy = gridDim.x * gridDim.y;
  const int lid = xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gxy * gridDim.z);
  const int z = lid / gxy, txy = lid % gxy;
  const int zo = z / g.batch_inner, zi = z % g.batch_inner;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(g.A) + zo * g.sAo + zi * g.sAi;
  const bf16_t* B = reinterpret_cast<const bf16_t*>(g.B) + zo * g.sBo + zi * g.sBi;
  TC* Cp = reinterpret_cast<TC*>(g.C) + zo * g.sCo + zi * g.sCi;
  glds256_block<TC, AK, BK_, DBG>(g, A, B, Cp, txy, gridDim.x, gridDim.y, smem);
}

// Up to sixteen unbatched products of one layout in one launch of 256x256 tiles: workgroups
// [first[i], first[i+1]) tile product i.  The four weight gradients of a transformer layer have
// 64 + 64 + 48 + 16 such tiles: none covers the chip alone, together they cover three quarters, those of
// four layers three full rounds, at twice the operand reuse of 128x128 tiles.
constexpr int GROUP_MAX = 16;
struct GemmArgsG { GemmArgs p[GROUP_MAX]; int first[GROUP_MAX + 1]; int gx[GROUP_MAX]; int gy[GROUP_MAX]; };
static_assert(sizeof(GemmArgsG) <= 4096, "kernel arguments are limited to 4 KiB");

template <typename TC, bool AK, bool BK_>
__global__ __launch_bounds__(NT256, 1) void gemm_glds256_group_kernel(GemmArgsG a) {
  __shared__ __align__(16) bf16_t smem[2 * 4 * GT_ELEMS];
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  int i = 0;
#pragma unroll
  for (int j = 1; j < GROUP_MAX; ++j)
    if (lid >= a.first[j]) i = j;
  const GemmArgs& g = a.p[i];
  glds256_block<TC, AK, BK_, 0>(g, reinterpret_cast<const bf16_t*>(g.A), reinterpret_cast<const bf16_t*>(g.B),
                                reinterpret_cast<TC*>(g.C), lid - a.first[i], a.gx[i], a.gy[i], smem);
}

template <typename TC>
int launch_glds256_group(const pg_gemm_desc* d, const GemmArgsG& a, int blocks, hipStream_t st) {
  if (d->a_kmajor && d->b_kmajor) hipLaunchKernelGGL((gemm_glds256_group_kernel<TC, true, true>), dim3(blocks), dim3(NT256), 0, st, a);
  else if (d->a_kmajor && !d->b_kmajor) hipLaunchKernelGGL((gemm_glds256_group_kernel<TC, true, false>), dim3(blocks), dim3(NT256), 0, st, a);
  else if (!d->a_kmajor && d->b_kmajor) hipLaunchKernelGGL((gemm_glds256_group_kernel<TC, false, true>), dim3(blocks), dim3(NT256), 0, st, a);
  else hipLaunchKernelGGL((gemm_glds256_group_kernel<TC, false, false>), dim3(blocks), dim3(NT256), 0, st, a);
  return hipGetLastError() == hipSuccess ? PG_OK : PG_ERR_LAUNCH;
}

template <typename TC>
int launch_glds256(const pg_gemm_desc* d, const GemmArgs& a, hipStream_t st) {
  dim3 grid((d->N + 255) / 256, (d->M + 255) / 256, d->batch);
  if (d->a_kmajor && d->b_kmajor) hipLaunchKernelGGL((gemm_glds256_kernel<TC, true, true>), grid, dim3(NT256), 0, st, a);
  else if (d->a_kmajor && !d->b_kmajor) hipLaunchKernelGGL((gemm_glds256_kernel<TC, true, false>), grid, dim3(NT256), 0, st, a);
  else if (!d->a_kmajor && d->b_kmajor) hipLaunchKernelGGL((gemm_glds256_kernel<TC, false, true>), grid, dim3(NT256), 0, st, a);
  else hipLaunchKernelGGL((gemm_glds256_kernel<TC, false, false>), grid, dim3(NT256), 0, st, a);
  return hipGetLastError() == hipSuccess ? PG_OK : PG_ERR_LAUNCH;
}

// ---- narrow variant: 128 x 64 output tile (4 waves x 32 x 64), for N <= 64 attention products
// (P V, dS K, dS^T Q, P^T dO ...) and as an option for grids that would otherwise leave CUs idle.
// B tile: 64 rows of 128 bytes either way (k-major [n][k], or n-major [k][n] with the chunk
// swizzle c ^ (k & 7)); two LDS buffers of 24 KiB.
constexpr int NB_ELEMS = 64 * GBK;  // 4096 bf16 = 8 KiB

template <bool KMAJOR>
__device__ __forceinline__ void glds_stage_b64(const bf16_t* __restrict__ base, long ld, int n0, int n_max, int k0,
                                               bf16_t* lds_tile, int wid, int lane, int k_max) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int i = wid * 2 + j;            // wave instruction 0..7: 8 rows of 128 B
    const int row = 8 * i + (lane >> 3);  // n (k-major) or k (n-major) inside the tile
    const int c = (lane & 7) ^ (lane >> 3);
    const bf16_t* src;
    if (KMAJOR) {
      int ng = n0 + row;
      ng = ng < n_max ? ng : n_max - 1;
      src = base + (long)ng * ld + k0 + c * 8;
    } else {
      int n = n0 + c * 8;
      n = n <= n_max - 8 ? n : n_max - 8;
      int kr = k0 + row;  // (a ragged reduction extent ends inside the tile: rows past it meet zeros of A)
      kr = kr < k_max ? kr : k_max - 1;
      src = base + (long)kr * ld + n;
    }
    __builtin_amdgcn_global_load_lds(src, lds_tile + i * 512, 16, 0, 0);
  }
}

template <typename TC, bool AK, bool BK_>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_glds_n64_kernel(GemmArgs g) {
  __shared__ __align__(16) bf16_t smem[2 * (GT_ELEMS + NB_ELEMS)];
  const int gxy = gridDim.x * gridDim.y;
  const int lid = xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gxy * gridDim.z);
  const int z = lid / gxy, txy = lid % gxy;
  const int zo = z / g.batch_inner, zi = z % g.batch_inner;
  const int bm = (txy / gridDim.x) * BM, bn = (txy % gridDim.x) * 64;
  long oA = zo * g.sAo, oB = zo * g.sBo, oC = zo * g.sCo;
  GemmArgs gl = g;  // (M, K of this batch)
  if (g.seq_off) {
    const int a0 = g.seq_off[zo];
    const long r0 = a0 - g.seq_tok0;
    gl.M = gl.K = g.seq_off[zo + 1] - a0;
    if (bm >= gl.M) return;
    if (g.seq_ops & 1) oA = r0 * g.lda;
    if (g.seq_ops & 2) oB = r0 * g.ldb;
    if (g.seq_ops & 4) oC = r0 * g.ldc;
  }
  const bf16_t* A = reinterpret_cast<const bf16_t*>(g.A) + oA + zi * g.sAi;
  const bf16_t* B = reinterpret_cast<const bf16_t*>(g.B) + oB + zi * g.sBi;
  const bf16_t* A2 = g.K2 > 0 ? reinterpret_cast<const bf16_t*>(g.A2) + oA + zi * g.sAi : nullptr;
  const bf16_t* B2 = g.K2 > 0 ? reinterpret_cast<const bf16_t*>(g.B2) + oB + zi * g.sBi : nullptr;
  TC* Cp = reinterpret_cast<TC*>(g.C) + oC + zi * g.sCi;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, gq = lane >> 4, qq = r >> 2, pp = r & 3;

  // per-lane fragment offsets (bytes inside a tile)
  unsigned offA[AK ? 2 : 4], offB[BK_ ? 2 : 8];
  if constexpr (AK) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) offA[kk] = (unsigned)((wid * 32 + r) * 128 + (((kk * 4 + gq) ^ (r & 7)) << 4));
  } else {
#pragma unroll
    for (int hi = 0; hi < 2; ++hi)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int k = 8 * gq + qq + 4 * hi;
        const int ch = wid * 4 + 2 * i + (pp >> 1);
        offA[hi * 2 + i] = (unsigned)(k * 256 + ((ch ^ swz_m(k)) << 4) + 8 * (pp & 1));
      }
  }
  if constexpr (BK_) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) offB[kk] = (unsigned)(r * 128 + (((kk * 4 + gq) ^ (r & 7)) << 4));
  } else {
#pragma unroll
    for (int hi = 0; hi < 2; ++hi)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = 8 * gq + qq + 4 * hi;
        const int c = 2 * j + (pp >> 1);
        offB[hi * 4 + j] = (unsigned)(k * 128 + ((c ^ (k & 7)) << 4) + 8 * (pp & 1));
      }
  }

  f32x4 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const unsigned smem_addr = lds_addr(smem);
  constexpr int BUF = GT_ELEMS + NB_ELEMS;
  // k-tiles [0, nt1) come from (A, B), k-tiles [nt1, nt) from the optional second segment (A2, B2); a ragged
  // batch has the same extent in both
  const int nt1 = (gl.K + GBK - 1) / GBK;
  const int K2 = g.K2 > 0 ? (g.seq_off ? gl.K : g.K2) : 0;
  const int nt = nt1 + (K2 + GBK - 1) / GBK;
  auto stage = [&](int t, bf16_t* dst) {
    const bool second = t >= nt1;
    const int k0 = (second ? t - nt1 : t) * GBK;
    glds_stage<AK>(second ? A2 : A, g.lda, bm, gl.M, k0, dst, wid, lane);
    glds_stage_b64<BK_>(second ? B2 : B, g.ldb, bn, g.N, k0, dst + GT_ELEMS, wid, lane, second ? K2 : gl.K);
  };
  stage(0, smem);
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    if (t + 1 < nt) stage(t + 1, smem + ((t + 1) & 1) * BUF);
    const unsigned la = smem_addr + (t & 1) * BUF * 2;
    const unsigned lb = la + GT_ELEMS * 2;
    u32x4 af[2][2], bf[2][4];  // [kk][tile]
    if constexpr (AK) {
      af[0][0] = lds_read_b128<0>(la + offA[0]);
      af[0][1] = lds_read_b128<2048>(la + offA[0]);
      af[1][0] = lds_read_b128<0>(la + offA[1]);
      af[1][1] = lds_read_b128<2048>(la + offA[1]);
    } else {
#define PG_A_TR(KK, I)                                                         \
      {                                                                        \
        u32x2 lo = lds_read_tr_b64<KK * 8192>(la + offA[I]);                  \
        u32x2 hi = lds_read_tr_b64<KK * 8192>(la + offA[2 + I]);              \
        af[KK][I] = (u32x4){lo[0], lo[1], hi[0], hi[1]};                      \
      }
      PG_A_TR(0, 0) PG_A_TR(0, 1) PG_A_TR(1, 0) PG_A_TR(1, 1)
#undef PG_A_TR
    }
    if constexpr (BK_) {
#define PG_B_K(KK)                                        \
      bf[KK][0] = lds_read_b128<0>(lb + offB[KK]);        \
      bf[KK][1] = lds_read_b128<2048>(lb + offB[KK]);     \
      bf[KK][2] = lds_read_b128<4096>(lb + offB[KK]);     \
      bf[KK][3] = lds_read_b128<6144>(lb + offB[KK]);
      PG_B_K(0) PG_B_K(1)
#undef PG_B_K
    } else {
#define PG_B_TR(KK, J)                                                         \
      {                                                                        \
        u32x2 lo = lds_read_tr_b64<KK * 4096>(lb + offB[J]);                  \
        u32x2 hi = lds_read_tr_b64<KK * 4096>(lb + offB[4 + J]);              \
        bf[KK][J] = (u32x4){lo[0], lo[1], hi[0], hi[1]};                      \
      }
      PG_B_TR(0, 0) PG_B_TR(0, 1) PG_B_TR(0, 2) PG_B_TR(0, 3) PG_B_TR(1, 0) PG_B_TR(1, 1) PG_B_TR(1, 2) PG_B_TR(1, 3)
#undef PG_B_TR
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bf[kk][j]),
                                                              __builtin_bit_cast(bf16x8_t, af[kk][i]), acc[i][j], 0, 0, 0);
    __syncthreads();
  }
  gemm_epilogue<bf16_t, TC, 2>(acc, gl, Cp, g.vec_ok && bn + 64 <= g.N, bm + wid * 32, bn, lane);
}

template <typename TC>
int launch_glds_n64(const pg_gemm_desc* d, const GemmArgs& a, hipStream_t st) {
  dim3 grid((d->N + 63) / 64, (d->M + BM - 1) / BM, d->batch);
  if (d->a_kmajor && d->b_kmajor) hipLaunchKernelGGL((gemm_glds_n64_kernel<TC, true, true>), grid, dim3(NTHREADS), 0, st, a);
  else if (d->a_kmajor && !d->b_kmajor) hipLaunchKernelGGL((gemm_glds_n64_kernel<TC, true, false>), grid, dim3(NTHREADS), 0, st, a);
  else if (!d->a_kmajor && d->b_kmajor) hipLaunchKernelGGL((gemm_glds_n64_kernel<TC, false, true>), grid, dim3(NTHREADS), 0, st, a);
  else hipLaunchKernelGGL((gemm_glds_n64_kernel<TC, false, false>), grid, dim3(NTHREADS), 0, st, a);
  return hipGetLastError() == hipSuccess ? PG_OK : PG_ERR_LAUNCH;
}

// Dispatch thresholds and switches.  Process-global, set through pg_tune by tests and A/B measurements only; the
// values here are what every product run uses.  Not synchronised: pg_tune must not run beside a launch.
int g_gemm_epi_prefetch = 1;  // bf16 epilogues fetch their per-element operand (pre-activation, residual or C) for the whole wave tile up front (pg_tune key 23)
int g_gemm_narrow = 1;  // 1 = 128x64 tiles when N <= 64, 2 = also whenever the 128x128 grid has <= 256 workgroups, 0 = never

int g_gemm_wide = 1;  // 1 = 256x256 tiles for large-N products with enough tiles to cover the chip, 2 = always, 0 = never

int g_gemm_ep8 = 1;   // 16-byte epilogue accesses for bf16 outputs (0 = 8-byte accesses only)
int g_gemm_pair = 1;  // 0 = pg_gemm_pair launches its two products one after the other
int g_gemm_dualk = 1;  // 0 = a two-segment product runs as two launches
int g_gemm_group_m = 8;  // tile rows per group of the grouped tile order (1 = plain row-major)
int g_gemm_split_tail = 50;  // 256-tile products: the rows of a last round less than this many per cent full go to a second launch (0 = never)
int g_gemm_wide_min_k = 1024;  // smallest K for which a narrow-N product with many tiles takes the 256-tile kernel
int g_gemm_group = 1;  // 0 = pg_gemm_group launches its products in pairs / singly
int g_gemm_wide_min_tiles = 129;  // fewest 256x256 tiles at which a narrow-N (512 <= N < 2048) product takes the 256x256 kernel (pg_tune key 14)
int g_gemm_path = 1;  // 0 = register-staged kernel, 1 = direct-to-LDS (ring depth by grid size), 2/4 = force that depth
// largest 128x128 grid that takes the four-buffer ring (one workgroup per CU): measured on 16 cold operand sets, the
// ring wins at 152 workgroups (N = 1024 products of 2432 rows: 44 vs 65 us at K = 4096) and loses from 360 on
// (5696 rows: 85 vs 72 us), where two workgroups per CU with two buffers each hide the latency better
int g_gemm_deep_max_blocks = 256;

}  // namespace

int pg_internal_fused_gp = 1;  // fused attention kernels in the gradient-penalty passes (engine)

extern int pg_internal_colsum_chunks;
extern int pg_internal_colsum_waves;
extern int pg_internal_dw_depth;
extern int pg_internal_round_mask;
extern int pg_internal_visible_keys;
extern int pg_internal_fused_dgelu_save;
extern int pg_internal_attn_resident;
extern int pg_internal_attn_key_stationary;
extern "C" int pg_tune(int key, int value) {
  if (key == 0) { g_gemm_path = value; return PG_OK; }
  if (key == 1) { g_gemm_narrow = value; return PG_OK; }
  if (key == 2) { g_gemm_wide = value; return PG_OK; }
  if (key == 4) { g_gemm_ep8 = value; return PG_OK; }
  if (key == 5) { g_gemm_pair = value; return PG_OK; }
  if (key == 6) { pg_internal_fused_gp = value; return PG_OK; }
  if (key == 7) { g_gemm_dualk = value; return PG_OK; }
  if (key == 11) { g_gemm_group = value; return PG_OK; }
  if (key == 15) { pg_internal_fused_dgelu_save = value != 0; return PG_OK; }
  if (key == 16) { if (value < 0) return PG_ERR_ARG; g_gemm_deep_max_blocks = value; return PG_OK; }
  if (key == 14) { if (value < 1) return PG_ERR_ARG; g_gemm_wide_min_tiles = value; return PG_OK; }
  if (key == 20) { pg_internal_visible_keys = value ? 1 : 0; return PG_OK; }
  if (key == 18) { if (value < 0 || value > 63) return PG_ERR_ARG; pg_internal_round_mask = value; return PG_OK; }
  if (key == 13) { if (value < 0 || value > 4) return PG_ERR_ARG; pg_internal_dw_depth = value; return PG_OK; }
  if (key == 12) { if (value < 1) return PG_ERR_ARG; pg_internal_colsum_chunks = value; return PG_OK; }
  if (key == 17) { if (value != 4 && value != 16) return PG_ERR_ARG; pg_internal_colsum_waves = value; return PG_OK; }
  if (key == 10) { g_gemm_wide_min_k = value; return PG_OK; }
  if (key == 9) { if (value < 0 || value > 100) return PG_ERR_ARG; g_gemm_split_tail = value == 1 ? 50 : value; return PG_OK; }
  if (key == 8) { if (value < 1) return PG_ERR_ARG; g_gemm_group_m = value; return PG_OK; }
  if (key == 23) { g_gemm_epi_prefetch = value ? 1 : 0; return PG_OK; }
  if (key == 25) { pg_internal_attn_resident = value ? 1 : 0; return PG_OK; }
  if (key == 26) { pg_internal_attn_key_stationary = value ? 1 : 0; return PG_OK; }
  return PG_ERR_ARG;
}

namespace {
// PG_TUNE="key=value,key=value" in the environment is applied once, when the library is loaded (the same keys as pg_tune):
// a deployment that wants a switch off sets it there and never calls pg_tune; an entry that does not parse or that
// pg_tune refuses is reported on stderr and skipped.
struct TuneFromEnv {
  TuneFromEnv() {
    const char* e = getenv("PG_TUNE");
    if (!e) return;
    std::string s(e);
    size_t pos = 0;
    while (pos < s.size()) {
      size_t end = s.find(',', pos);
      if (end == std::string::npos) end = s.size();
      const std::string item = s.substr(pos, end - pos);
      pos = end + 1;
      if (item.empty()) continue;
      int k = 0, v = 0;
      if (sscanf(item.c_str(), "%d=%d", &k, &v) != 2 || pg_tune(k, v) != PG_OK)
        fprintf(stderr, "libprotgen_hip: PG_TUNE entry '%s' ignored\n", item.c_str());
    }
  }
};
TuneFromEnv g_tune_from_env;

int gemm_prepare(const pg_gemm_desc* d, GemmArgs& a);
ProfEntry* prof_begin(int variant, double flops, hipStream_t st);
}  // namespace

static int device_cus() {
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
    if (n_cu <= 0) n_cu = 256;
  }
  return n_cu;
}

static long tiles256(const pg_gemm_desc* d) { return (long)((d->M + 255) / 256) * ((d->N + 255) / 256) * d->batch; }

extern "C" int pg_gemm(const pg_gemm_desc* d, void* stream) {
  GemmArgs a;
  const int prc = gemm_prepare(d, a);
  if (prc != PG_OK) return prc;
  dim3 grid((d->N + BN - 1) / BN, (d->M + BM - 1) / BM, d->batch);
  hipStream_t st = (hipStream_t)stream;
  if (d->seq_off) {  // ragged batches: the 128x64-tile kernel only (checked in gemm_prepare)
    ProfEntry* pe = prof_begin(8 + (d->c_dtype == PG_F32 ? 4 : 0) + (d->a_kmajor ? 2 : 0) + (d->b_kmajor ? 1 : 0),
                               2.0 * d->M * d->N * (d->K2 > 0 ? 2.0 : 1.0) * d->K * d->batch, st);
    const int rc = d->c_dtype == PG_F32 ? launch_glds_n64<float>(d, a, st) : launch_glds_n64<bf16_t>(d, a, st);
    if (pe) hipEventRecord(pe->b, st);
    return rc;
  }
  const bool fast = g_gemm_path != 0 && d->dtype == PG_BF16 && d->K % GBK == 0 && d->K2 % GBK == 0 && (d->a_kmajor || d->M >= 8) &&
                    (d->b_kmajor || d->N >= 8);
  const long blocks128 = (long)grid.x * grid.y * grid.z;
  // (key 19: a product of few 128x128 tiles on 128x64 tiles, twice the workgroups.  Timed alone on cold operands
  // the 256 rows a sparse last round of 256x256 tiles leaves over take 14 us that way against 25-30 us at K = 1024;
  // inside the GAN step the four-buffer 128x128 kernel is the faster one, 846 against 850-854 ms per step: off)
  const bool narrow = fast && ((g_gemm_narrow >= 1 && d->N <= 64) || (d->K2 == 0 && g_gemm_narrow == 2 && blocks128 <= 256) ||
                              (d->K2 == 0 && g_gemm_narrow == 3 && blocks128 <= 128 && d->K >= 1024));
  const bool wide = fast && !narrow &&
                    (g_gemm_wide == 2 || (g_gemm_wide >= 1 && d->N >= 2048 && tiles256(d) >= 160) ||
                     // narrow N but long K: the halved L2 -> LDS traffic outweighs a partial round as soon as the
                     // 128x128 tiles no longer fit the chip at once (192 tiles on 256 CUs beat 768 tiles of 128x128
                     // by 17 %; 130 tiles beat 520 by 5 % at K 1024 and 10 % at K 4096; at 128 tiles = 512 of
                     // 128x128, all resident together, they lose 28 against 37 us)
                     (g_gemm_wide == 1 && d->N >= 512 && d->K >= g_gemm_wide_min_k && tiles256(d) >= g_gemm_wide_min_tiles));
  if (d->K2 > 0 && (!fast || !g_gemm_dualk || (d->batch != 1 && !narrow))) {
    // only the 128x128 and 256x256 direct-to-LDS kernels walk two reduction segments: elsewhere two launches
    if (d->bias || d->residual || d->act != PG_ACT_NONE) return PG_ERR_UNSUPPORTED;  // the epilogue would apply twice
    pg_gemm_desc one = *d;
    one.K2 = 0; one.A2 = nullptr; one.B2 = nullptr;
    const int rc1 = pg_gemm(&one, stream);
    if (rc1 != PG_OK) return rc1;
    one.A = d->A2; one.B = d->B2; one.K = d->K2; one.beta = 1.f;
    return pg_gemm(&one, stream);
  }
  if (wide && g_gemm_wide == 1 && g_gemm_split_tail && d->batch == 1 && d->a_kmajor) {
    // One 256x256 tile per CU and round: when the last round would be less than half full (pg_tune key 9), the
    // rows beyond the last full round go to a second launch (which takes the 128x128 kernel) instead of costing
    // a whole round.  (Measured at thresholds 30 / 40 / 50 / 60 %: mass generation at batch 64, whose N = 1024
    // products have 1.25 - 1.5 rounds of tiles, 240 / 246 / 248 / 249 sequences/s; the GAN step unchanged.)
    const int cus = device_cus();
    const long tiles = tiles256(d), full = tiles / cus;
    const int gxw = (d->N + 255) / 256;
    const long rows1 = full * cus / gxw * 256;  // whole tile rows inside the full rounds
    if (full >= 1 && tiles % cus != 0 && (tiles % cus) * 100 < (long)g_gemm_split_tail * cus && rows1 > 0 && rows1 < d->M) {
      const int esz = 2, csz = d->c_dtype == PG_F32 ? 4 : 2;
      pg_gemm_desc top = *d, rest = *d;
      top.M = (int)rows1;
      rest.M = d->M - (int)rows1;
      rest.A = (const char*)d->A + rows1 * d->lda * esz;
      rest.C = (char*)d->C + rows1 * d->ldc * csz;
      if (d->residual) rest.residual = (const char*)d->residual + rows1 * d->ldr * esz;
      if (d->aux) rest.aux = (char*)d->aux + rows1 * d->ldaux * esz;
      if (d->aux2) rest.aux2 = (char*)d->aux2 + rows1 * d->ldaux * esz;
      if (d->K2 > 0) rest.A2 = (const char*)d->A2 + rows1 * d->lda * esz;  // second segment: same rows of A2
      const int rc1 = pg_gemm(&top, stream);
      return rc1 != PG_OK ? rc1 : pg_gemm(&rest, stream);
    }
  }
  ProfEntry* pe = prof_begin((d->dtype == PG_BF16 ? 8 : 0) + (d->c_dtype == PG_F32 ? 4 : 0) + (d->a_kmajor ? 2 : 0) + (d->b_kmajor ? 1 : 0),
                             2.0 * d->M * d->N * ((double)d->K + d->K2) * d->batch, st);
  int rc = PG_ERR_ARG;
  if (narrow) {
    if (d->c_dtype == PG_F32) rc = launch_glds_n64<float>(d, a, st);
    else rc = launch_glds_n64<bf16_t>(d, a, st);
  } else if (wide) {
    if (d->c_dtype == PG_F32) rc = launch_glds256<float>(d, a, st);
    else rc = launch_glds256<bf16_t>(d, a, st);
  } else if (fast) {
    const long blocks = blocks128;
    const bool deep = g_gemm_path == 4 || (g_gemm_path == 1 && blocks <= g_gemm_deep_max_blocks && d->K >= 4 * GBK);
    if (deep) {
      if (d->c_dtype == PG_F32) rc = launch_glds<float, 4>(d, a, grid, st);
      else rc = launch_glds<bf16_t, 4>(d, a, grid, st);
    } else {
      if (d->c_dtype == PG_F32) rc = launch_glds<float, 2>(d, a, grid, st);
      else rc = launch_glds<bf16_t, 2>(d, a, grid, st);
    }
  } else if (d->dtype == PG_BF16) {
    if (d->c_dtype == PG_F32) rc = launch_layout<bf16_t, float>(d, a, grid, st);
    else rc = launch_layout<bf16_t, bf16_t>(d, a, grid, st);
  } else if (d->dtype == PG_F32) {
    rc = launch_layout<float, float>(d, a, grid, st);
  }
  if (pe) hipEventRecord(pe->b, st);
  return rc;
}
namespace {
ProfEntry* prof_begin(int variant, double flops, hipStream_t st) {
  if (g_prof.every <= 0) return nullptr;
  ProfEntry e;
  {
    std::lock_guard<std::mutex> lock(g_prof.mu);
    ++g_prof.launches;
    if (++g_prof.counter % g_prof.every != 0) return nullptr;
  }
  if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) return nullptr;
  e.variant = variant;
  e.flops = flops;
  ProfEntry* pe;
  {
    std::lock_guard<std::mutex> lock(g_prof.mu);
    g_prof.entries.push_back(e);
    pe = &g_prof.entries.back();
  }
  hipEventRecord(pe->a, st);
  return pe;
}

int gemm_prepare(const pg_gemm_desc* d, GemmArgs& a) {
  if (!d || !d->A || !d->B || !d->C) return PG_ERR_ARG;
  if (d->M <= 0 || d->N <= 0 || d->K <= 0 || d->batch <= 0 || d->batch_inner <= 0) return PG_ERR_ARG;
  if (d->batch % d->batch_inner) return PG_ERR_ARG;
  const int epc = d->dtype == PG_BF16 ? 8 : 4;
  // 16-byte chunk loads: leading dimensions and batch strides must keep chunks aligned.
  if (d->lda % epc || d->ldb % epc || d->sAo % epc || d->sAi % epc || d->sBo % epc || d->sBi % epc) return PG_ERR_ALIGN;
  if (((uintptr_t)d->A | (uintptr_t)d->B) & 15) return PG_ERR_ALIGN;
  if (d->a_kmajor) { if (d->lda < ((d->K + epc - 1) / epc) * epc) return PG_ERR_ARG; }
  else { if (d->M % epc || d->M < epc || d->lda < d->M) return PG_ERR_ARG; }
  if (d->b_kmajor) { if (d->ldb < ((d->K + epc - 1) / epc) * epc) return PG_ERR_ARG; }
  else { if (d->N % epc || d->N < epc || d->ldb < d->N) return PG_ERR_ARG; }
  if ((d->residual || d->aux) && d->batch != 1) return PG_ERR_ARG;
  if (d->act == PG_ACT_MUL_DGELU && !d->aux) return PG_ERR_ARG;
  if (d->c_dtype != PG_F32 && d->c_dtype != d->dtype) return PG_ERR_ARG;

  a.A = d->A; a.B = d->B; a.C = d->C;
  a.M = d->M; a.N = d->N; a.K = d->K;
  a.lda = d->lda; a.ldb = d->ldb; a.ldc = d->ldc;
  a.batch_inner = d->batch_inner;
  a.sAo = d->sAo; a.sAi = d->sAi; a.sBo = d->sBo; a.sBi = d->sBi; a.sCo = d->sCo; a.sCi = d->sCi;
  a.alpha = d->alpha; a.beta = d->beta;
  a.bias = d->bias; a.residual = d->residual; a.ldr = d->ldr; a.aux = d->aux; a.ldaux = d->ldaux;
  a.aux2 = d->act == PG_ACT_MUL_DGELU ? d->aux2 : nullptr;
  a.act = d->act;
  a.A2 = nullptr; a.B2 = nullptr; a.K2 = 0;
  a.group_m = g_gemm_group_m;
  a.seq_off = d->seq_off; a.seq_tok0 = d->seq_tok0; a.seq_ops = d->seq_ops;
  if (d->seq_off) {
    // sequence zo of every (zo, zi) batch has its own M = K <= the descriptor's: whole 64-row tiles of a
    // score-shaped operand are walked (zeros past the extent), token-row operands are clamped
    if (d->dtype != PG_BF16 || d->N > 64 || d->N % 8 || d->M != d->K || d->seq_tok0 < 0 || (d->seq_ops & ~7)) return PG_ERR_UNSUPPORTED;
    if (d->b_kmajor || d->bias || d->residual || d->aux) return PG_ERR_UNSUPPORTED;
    // A must be the score-shaped operand: its reduction rows past a sequence's extent are zeros by the layout's
    // contract, which is what ends the sum; a token-row A would run on into the next sequence's rows
    if (d->seq_ops & 1) return PG_ERR_UNSUPPORTED;
  }
  if (d->K2 > 0) {
    if (!d->A2 || !d->B2 || (d->batch != 1 && d->N > 64)) return PG_ERR_ARG;
    if (((uintptr_t)d->A2 | (uintptr_t)d->B2) & 15) return PG_ERR_ALIGN;
    if (d->a_kmajor && d->lda < ((d->K2 + epc - 1) / epc) * epc) return PG_ERR_ARG;
    if (d->b_kmajor && d->ldb < ((d->K2 + epc - 1) / epc) * epc) return PG_ERR_ARG;
    a.A2 = d->A2; a.B2 = d->B2; a.K2 = d->K2;
  } else if (d->K2 < 0) {
    return PG_ERR_ARG;
  }
  const int csz = d->c_dtype == PG_F32 ? 4 : 2;
  const int tsz = d->dtype == PG_F32 ? 4 : 2;
  bool vec = (d->ldc % 4 == 0) && (d->sCo % 4 == 0) && (d->sCi % 4 == 0) && (((uintptr_t)d->C) % (4 * csz) == 0);
  if (d->residual) vec = vec && (d->ldr % 4 == 0) && (((uintptr_t)d->residual) % (4 * tsz) == 0);
  if (d->aux) vec = vec && (d->ldaux % 4 == 0) && (((uintptr_t)d->aux) % (4 * tsz) == 0);
  if (d->bias) vec = vec && (((uintptr_t)d->bias) % 16 == 0);
  if (a.aux2) vec = vec && (((uintptr_t)a.aux2) % (4 * tsz) == 0);
  a.vec_ok = vec ? 1 : 0;
  if (vec && d->dtype == PG_BF16 && d->c_dtype == PG_BF16) {  // 16-byte epilogue accesses possible?
    bool v8 = (d->ldc % 8 == 0) && (d->sCo % 8 == 0) && (d->sCi % 8 == 0) && (((uintptr_t)d->C) % 16 == 0);
    if (d->residual) v8 = v8 && (d->ldr % 8 == 0) && (((uintptr_t)d->residual) % 16 == 0);
    if (d->aux) v8 = v8 && (d->ldaux % 8 == 0) && (((uintptr_t)d->aux) % 16 == 0);
    if (a.aux2) v8 = v8 && (((uintptr_t)a.aux2) % 16 == 0);
    if (v8 && g_gemm_ep8) {
      a.vec_ok = 2;
    }
  }
  if (a.vec_ok && g_gemm_epi_prefetch) a.vec_ok |= 4;  // (vec_ok: bits 0-1 = access width class, bit 2 = operand prefetch)

  return PG_OK;
}
}  // namespace


extern "C" int pg_gemm_pair(const pg_gemm_desc* d0, const pg_gemm_desc* d1, void* stream) {
  GemmArgs2 a;
  int rc = gemm_prepare(d0, a.p0);
  if (rc != PG_OK) return rc;
  rc = gemm_prepare(d1, a.p1);
  if (rc != PG_OK) return rc;
  auto fast = [](const pg_gemm_desc* d) {
    return d->dtype == PG_BF16 && d->K % GBK == 0 && d->K2 % GBK == 0 && d->batch == 1 && d->N > 64 &&
           (d->a_kmajor || d->M >= 8) && (d->b_kmajor || d->N >= 8);
  };
  const bool seg_ok = (d0->K2 == 0 && d1->K2 == 0) || g_gemm_dualk != 0;
  const bool together = g_gemm_path != 0 && g_gemm_pair != 0 && seg_ok && fast(d0) && fast(d1) && d0->a_kmajor == d1->a_kmajor &&
                        d0->b_kmajor == d1->b_kmajor && d0->c_dtype == d1->c_dtype;
  if (!together) {
    rc = pg_gemm(d0, stream);
    return rc != PG_OK ? rc : pg_gemm(d1, stream);
  }
  a.gx0 = (d0->N + BN - 1) / BN;
  a.gx1 = (d1->N + BN - 1) / BN;
  a.nb0 = a.gx0 * ((d0->M + BM - 1) / BM);
  const int blocks = a.nb0 + a.gx1 * ((d1->M + BM - 1) / BM);
  hipStream_t st = (hipStream_t)stream;
  ProfEntry* pe = prof_begin(8 + (d0->c_dtype == PG_F32 ? 4 : 0) + (d0->a_kmajor ? 2 : 0) + (d0->b_kmajor ? 1 : 0),
                             2.0 * d0->M * d0->N * ((double)d0->K + d0->K2) + 2.0 * d1->M * d1->N * ((double)d1->K + d1->K2), st);
  const int kmin = d0->K < d1->K ? d0->K : d1->K;
  const bool deep = g_gemm_path == 4 || (g_gemm_path == 1 && blocks <= g_gemm_deep_max_blocks && kmin >= 4 * GBK);
  if (deep) rc = d0->c_dtype == PG_F32 ? launch_glds_pair<float, 4>(d0, a, blocks, st) : launch_glds_pair<bf16_t, 4>(d0, a, blocks, st);
  else rc = d0->c_dtype == PG_F32 ? launch_glds_pair<float, 2>(d0, a, blocks, st) : launch_glds_pair<bf16_t, 2>(d0, a, blocks, st);
  if (pe) hipEventRecord(pe->b, st);
  return rc;
}

extern "C" int pg_gemm_group(const pg_gemm_desc* d, int n, void* stream) {
  if (!d || n < 1 || n > GROUP_MAX) return PG_ERR_ARG;
  GemmArgsG a;
  bool together = g_gemm_path != 0 && g_gemm_group != 0 && n >= 2;
  long tiles = 0;
  double flops = 0;
  for (int i = 0; i < n; ++i) {
    const int rc = gemm_prepare(&d[i], a.p[i]);
    if (rc != PG_OK) return rc;
    const pg_gemm_desc* di = &d[i];
    together = together && di->dtype == PG_BF16 && di->K % GBK == 0 && di->K2 % GBK == 0 && di->batch == 1 && di->N > 64 &&
               (di->a_kmajor || di->M >= 8) && (di->b_kmajor || di->N >= 8) && di->a_kmajor == d[0].a_kmajor &&
               di->b_kmajor == d[0].b_kmajor && di->c_dtype == d[0].c_dtype && (di->K2 == 0 || g_gemm_dualk != 0);
    a.gx[i] = (di->N + 255) / 256;
    a.gy[i] = (di->M + 255) / 256;
    a.first[i] = (int)tiles;
    tiles += (long)a.gx[i] * a.gy[i];
    flops += 2.0 * di->M * di->N * ((double)di->K + di->K2);
  }
  for (int i = n; i < GROUP_MAX; ++i) { a.p[i] = a.p[0]; a.gx[i] = a.gy[i] = 1; a.first[i] = 0x7fffffff; }
  a.first[GROUP_MAX] = 0x7fffffff;
  // worth it only if the 256x256 tiles of all products together occupy most of the chip
  const int cus = device_cus();
  together = together && tiles * 2 >= cus && tiles <= 0x7fffffff / 2;
  if (!together) {
    int i = 0;
    for (; i + 1 < n; i += 2) {
      const int rc = pg_gemm_pair(&d[i], &d[i + 1], stream);
      if (rc != PG_OK) return rc;
    }
    return i < n ? pg_gemm(&d[i], stream) : PG_OK;
  }
  hipStream_t st = (hipStream_t)stream;
  ProfEntry* pe = prof_begin(8 + (d[0].c_dtype == PG_F32 ? 4 : 0) + (d[0].a_kmajor ? 2 : 0) + (d[0].b_kmajor ? 1 : 0), flops, st);
  const int rc = d[0].c_dtype == PG_F32 ? launch_glds256_group<float>(&d[0], a, (int)tiles, st)
                                        : launch_glds256_group<bf16_t>(&d[0], a, (int)tiles, st);
  if (pe) hipEventRecord(pe->b, st);
  return rc;
}

// variant index = 8*(bf16) + 4*(fp32 output) + 2*(A k-major) + (B k-major)
extern "C" int pg_profile_begin(int sample_every) {
  std::lock_guard<std::mutex> lock(g_prof.mu);
  for (auto& e : g_prof.entries) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
  g_prof.entries.clear();
  g_prof.every = sample_every > 0 ? sample_every : 0;
  g_prof.counter = 0;
  g_prof.launches = 0;
  return PG_OK;
}
// out[16][3] = {sampled launches, sampled flops, sampled milliseconds} per variant; stops sampling.
extern "C" int pg_profile_end(double* out, int64_t* total_launches) {
  if (!out) return PG_ERR_ARG;
  std::lock_guard<std::mutex> lock(g_prof.mu);
  g_prof.every = 0;
  for (int i = 0; i < 48; ++i) out[i] = 0.0;
  for (auto& e : g_prof.entries) {
    float ms = 0.f;
    if (hipEventSynchronize(e.b) == hipSuccess && hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
      out[e.variant * 3 + 0] += 1.0;
      out[e.variant * 3 + 1] += e.flops;
      out[e.variant * 3 + 2] += ms;
    }
    hipEventDestroy(e.a);
    hipEventDestroy(e.b);
  }
  g_prof.entries.clear();
  if (total_launches) *total_launches = g_prof.launches;
  return PG_OK;
}
