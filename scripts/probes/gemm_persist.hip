// PROBE, not part of libprotgen_hip.so: measured against the shipped 256x256 kernel and not faster (results under
// profiles/r02_d_ring_ab*.txt, profiles/r02_f_persist_ab*.txt).  To rebuild the experiment, add this file to SOURCES in
// protgen-gan_amd/build.py and route pg_gemm's wide branch to its pg_internal_launch_* entry point.
// Persistent form of the 256 x 256-tile direct-to-LDS GEMM (bf16 operands, A k-major, unbatched).
// Off by default (pg_tune key 24): measured slower than one workgroup per tile, see below.
//
// One workgroup per CU walks tiles id, id + gridDim.x, id + 2 gridDim.x, ... of the same XCD-aware
// tile order the one-tile-per-workgroup kernel uses.  The k-tiles of consecutive output tiles form one
// stream through the two LDS stages: the LDS-DMA pieces of the NEXT output tile's first k-tile are
// issued inside the last k-tile of the current one (the slots every other k-tile uses for its
// successor), so a tile after the first pays neither a workgroup launch nor a cold first fetch.
//
// Measured on MI355X against the one-tile-per-workgroup kernel (same bits, 16 rotating operand sets, us):
//   M 32768: N1024 K1024 84 -> 93,  N4096 K1024 gelu 310 -> 338,  N4096 K1024 x gelu' 350 -> 399,  N3072 K1024 207 -> 233
//   M  8192: N4096 K1024 gelu 81 -> 92,  N3072 K1024 68 -> 81;  products of at most one tile per CU are untouched.
// With the wait for the next tile's first k-tile placed after the epilogue instead of before it the losses are
// about half of these (N4096 gelu 310 -> 330, N3072 206 -> 225) and the N = 1024 shapes break even.
// Why: loads (LDS-DMA included) and stores share the vmcnt counter and complete in no fixed order, so the first
// "my pieces have landed" wait after an epilogue also waits for the acknowledgement of that epilogue's 128 - 256 KiB
// of stores, which all CUs issue at about the same time.  A workgroup that ends after its tile never waits for
// them: the next workgroup's launch and first fetch overlap the drain, and that overlap is worth more than the
// launch and the cold fetch cost.
#include <hip/hip_runtime.h>
#include "pg_common.h"
#include "protgen_hip.h"

namespace {
#include "gemm_shared.h"
#include "gemm_tile256.h"

// Tile txy of a gx x gy grid of 256 x 256 tiles -> first row / column: groups of group_m tile rows,
// walked down the rows first (the order of the one-tile-per-workgroup kernel).
__device__ __forceinline__ void tile_origin(const GemmArgs& g, int txy, int gx, int gy, int& bm, int& bn) {
  const int GM = gx > 4 ? g.group_m : 1;
  const int grp = txy / (GM * gx), first = grp * GM, rem = txy - grp * GM * gx;
  const int gsz = gy - first < GM ? gy - first : GM;
  bm = (first + rem % gsz) * 256;
  bn = (rem / gsz) * 256;
}

template <typename TC, bool BK_, int PH>
__device__ __forceinline__ void persist_run(const GemmArgs& g, bf16_t* smem, int gx, int gy, int wid, int lane) {
  const int tiles = gx * gy;
  const int wr = wid >> 2, wc = wid & 3, w4 = wid & 3, half = wid >> 2;
  const int nt1 = g.K / GBK, nt = nt1 + g.K2 / GBK;  // k-tiles [0, nt1) from (A, B), [nt1, nt) from (A2, B2)
  const bf16_t* A = reinterpret_cast<const bf16_t*>(g.A);
  const bf16_t* B = reinterpret_cast<const bf16_t*>(g.B);
  const bf16_t* A2 = reinterpret_cast<const bf16_t*>(g.A2);
  const bf16_t* B2 = reinterpret_cast<const bf16_t*>(g.B2);
  TC* Cp = reinterpret_cast<TC*>(g.C);
  const FragOff<true> fa_lo = frag_offsets<true>(0, lane), fa_hi = frag_offsets<true>(1, lane);
  const FragOff<BK_> fb = frag_offsets<BK_>(wc & 1, lane);
  const unsigned smem_addr = lds_addr(smem);
  const unsigned a_off = wr * 2 * GT_ELEMS, b_off = (2 + (wc >> 1)) * 2 * GT_ELEMS;  // bytes inside a stage
  const long a_step = GBK * 2, b_step = BK_ ? GBK * 2 : (long)GBK * g.ldb * 2;
  const int dst0 = half * GT_ELEMS + w4 * 4 * 512;  // this wave's first piece inside a stage (A part; B is 2 sub-tiles on)

  int id = blockIdx.x;
  int bm, bn;
  tile_origin(g, xcd_remap(id, tiles), gx, gy, bm, bn);
  unsigned aoff[4], boff[4];
  piece_offsets<true>(g.lda, bm, half, g.M, w4, lane, aoff);
  piece_offsets<BK_>(g.ldb, bn, half, g.N, w4, lane, boff);
  const char* a_src = reinterpret_cast<const char*>(A + (long)bm * g.lda);
  const char* b_src = reinterpret_cast<const char*>(BK_ ? B + (long)bn * g.ldb : B + bn);
  // the first k-tile of this workgroup's first output tile (nothing to overlap it with)
#pragma unroll
  for (int j = 0; j < 4; ++j) __builtin_amdgcn_global_load_lds(a_src + aoff[j], smem + dst0 + j * 512, 16, 0, 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) __builtin_amdgcn_global_load_lds(b_src + boff[j], smem + 2 * GT_ELEMS + dst0 + j * 512, 16, 0, 0);

  f32x4 acc[8][4];
  int p = 0;  // k-tiles consumed so far by this workgroup: k-tile p sits in stage p & 1
  // Every k-tile ends with "my pieces of the next one have landed" + a barrier (all waves' pieces have, and the
  // stage just read is free).  After the last k-tile of an output tile that wait comes BEFORE the epilogue's
  // stores: loads and stores share one counter and complete in no fixed order, so a wait placed after them would
  // also wait for the stores' acknowledgements, which a workgroup that ends after its tile never pays.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (;;) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < nt - 1; ++t, ++p) {
      if (t + 1 == nt1) {  // the next k-tile opens the second segment
        a_src = reinterpret_cast<const char*>(A2 + (long)bm * g.lda);
        b_src = reinterpret_cast<const char*>(BK_ ? B2 + (long)bn * g.ldb : B2 + bn);
      } else {
        a_src += a_step;
        b_src += b_step;
      }
      const unsigned st = smem_addr + (p & 1) * 8 * GT_ELEMS;
      bf16_t* nx = smem + ((p + 1) & 1) * 4 * GT_ELEMS + dst0;
      glds256_tile<true, BK_, PH, true, true>(st + a_off, st + b_off, fa_lo, fa_hi, fb, acc, a_src, b_src, aoff, boff, nx, nx + 2 * GT_ELEMS);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    const int cm = bm, cn = bn;  // the tile whose last k-tile this is
    const unsigned st = smem_addr + (p & 1) * 8 * GT_ELEMS;
    const int next = id + (int)gridDim.x;
    const bool more = next < tiles;
    // Its MFMAs carry the first k-tile of the next output tile into the other stage.  (After the last tile the
    // slots refetch this tile's own first k-tile, which nothing reads: one copy of the k-tile body instead of two
    // whose accumulators would have to meet in the same registers.)
    tile_origin(g, xcd_remap(more ? next : id, tiles), gx, gy, bm, bn);
    piece_offsets<true>(g.lda, bm, half, g.M, w4, lane, aoff);
    piece_offsets<BK_>(g.ldb, bn, half, g.N, w4, lane, boff);
    a_src = reinterpret_cast<const char*>(A + (long)bm * g.lda);
    b_src = reinterpret_cast<const char*>(BK_ ? B + (long)bn * g.ldb : B + bn);
    bf16_t* nx = smem + ((p + 1) & 1) * 4 * GT_ELEMS + dst0;
    glds256_tile<true, BK_, PH, true, true>(st + a_off, st + b_off, fa_lo, fa_hi, fb, acc, a_src, b_src, aoff, boff, nx, nx + 2 * GT_ELEMS);
    ++p;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // (the epilogue's per-lane offsets are recomputed per tile: as loop invariants they would stay live through
    // the k-loop, which has no registers to spare)
    int elane = lane;
    asm volatile("" : "+v"(elane));
    gemm_epilogue<bf16_t, TC, 8>(acc, g, Cp, g.vec_ok && cn + 256 <= g.N, cm + wr * 128, cn + wc * 64, elane);
    if (!more) break;  // (uniform: every wave of the workgroup leaves here)
    id = next;
  }
}

template <typename TC, bool BK_>
__global__ __launch_bounds__(NT256, 1) void gemm_persist256_kernel(GemmArgs g, int gx, int gy) {
  __shared__ __align__(16) bf16_t smem[2 * 4 * GT_ELEMS];  // [stage][A0 A1 B0 B1][16 KiB]
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // waves w and w + 4 share a SIMD: opposite piece phases
  if ((wid >> 2) == 0) persist_run<TC, BK_, 0>(g, smem, gx, gy, wid, lane);
  else persist_run<TC, BK_, 1>(g, smem, gx, gy, wid, lane);
}

}  // namespace

// args: the GemmArgs gemm.hip prepared.  workgroups: at most one per CU, a multiple of eight when the
// product has more tiles than that (so that a workgroup's tiles all belong to its own XCD's share).
int pg_internal_launch_persist256(const void* args, int M, int N, int b_kmajor, int c_is_f32, int workgroups, hipStream_t st) {
  const GemmArgs& a = *reinterpret_cast<const GemmArgs*>(args);
  const int gx = (N + 255) / 256, gy = (M + 255) / 256;
  long tiles = (long)gx * gy;
  if (tiles > 0x7fffffff / 2 || workgroups < 1) return PG_ERR_ARG;
  int wg = tiles < workgroups ? (int)tiles : workgroups;
  if (wg < tiles && wg >= 8) wg &= ~7;
  if (b_kmajor) {
    if (c_is_f32) hipLaunchKernelGGL((gemm_persist256_kernel<float, true>), dim3(wg), dim3(NT256), 0, st, a, gx, gy);
    else hipLaunchKernelGGL((gemm_persist256_kernel<bf16_t, true>), dim3(wg), dim3(NT256), 0, st, a, gx, gy);
  } else {
    if (c_is_f32) hipLaunchKernelGGL((gemm_persist256_kernel<float, false>), dim3(wg), dim3(NT256), 0, st, a, gx, gy);
    else hipLaunchKernelGGL((gemm_persist256_kernel<bf16_t, false>), dim3(wg), dim3(NT256), 0, st, a, gx, gy);
  }
  return hipGetLastError() == hipSuccess ? PG_OK : PG_ERR_LAUNCH;
}
