// PROBE, not part of libprotgen_hip.so: measured against the shipped 256x256 kernel and not faster (results under
// profiles/r02_d_ring_ab*.txt, profiles/r02_f_persist_ab*.txt).  To rebuild the experiment, add this file to SOURCES in
// protgen-gan_amd/build.py and route pg_gemm's wide branch to its pg_internal_launch_* entry point.
// 256 x 256-tile bf16 GEMM over a five-slot LDS ring (k-major x k-major operands); a translation unit of
// its own so that it builds in seconds.  pg_gemm (gemm.hip) decides which products take it.
#include "pg_common.h"
#include "protgen_hip.h"

namespace {
#include "gemm_shared.h"

constexpr int GBK = 64;
constexpr int NT256 = 512;

// ---- 256 x 256 tile over a ring of five 32-k half-stages (both operands k-major) -------------
// The two-stage kernel of gemm.hip keeps one 64 KiB k-tile in flight and has to see it land within
// the MFMAs of one k-tile (about 1.2 us); a k-tile takes it 1.5 - 1.65 us.  Here the LDS (all 160 KiB)
// is a ring of five 32 KiB slots, each holding 32 k of the 256 A rows and the 256 B rows.  Slots are
// consumed one by one (32 MFMAs per wave), and the loads of a whole 64-k tile (two slots) are issued
// as soon as the second of its slots is free: three half-stages = 1.5 k-tiles before the first of
// them is read, with up to two k-tiles (128 KiB) in flight.
// Measured: the same speed as the two-stage kernel on every forward shape (a k-tile still takes
// 1.65 us; 1.24 us with the loads compiled out, in both kernels), and 1 % slower inside the GAN step:
// what the loop waits for is the rate at which the L2s deliver 64 KiB per k-tile to 256 CUs, not the
// latency of one tile.  Off by default (pg_tune key 21); kept as the measured alternative.
// The two 64-byte halves of every 128-byte operand line are requested back to back (pieces ordered
// row group first, half second): with the halves half a stage apart the vector L1 had dropped the
// line before its second half was asked for and the kernel ran 10 - 15 % slower.
//   slot layout: A rows then B rows, 64 bytes per row; 16-byte chunk c of row r at position
//   c ^ ((r >> 1) & 3), so that the eight lanes an LDS cycle serves (rows r..r+7, one chunk) fall
//   into eight different 16-byte columns of the 256-byte bank line.
//   DMA piece: 16 rows x 64 B (lane -> row lane >> 2, position lane & 3); a wave copies two A and
//   two B pieces per slot.
constexpr int RING_SLOTS = 5;
constexpr int RING_SLOT_ELEMS = 2 * 256 * 32;  // bf16 elements per slot (32 KiB)

__device__ __forceinline__ void ring_piece_offsets(long ld, int blk0, int extent, int wid, int lane, unsigned (&off)[2]) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int rl = 32 * wid + 16 * j + (lane >> 2);
    const int c = (lane & 3) ^ ((lane >> 3) & 3);  // ((rl >> 1) & 3) == (lane >> 3) & 3
    int rg = blk0 + rl;
    rg = rg < extent ? rg : extent - 1;
    off[j] = (unsigned)(((long)(rg - blk0) * ld + c * 8) * 2);
  }
}

template <int N> __device__ __forceinline__ void vm_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Fragments of one slot a wave reads ahead: its four B column blocks and the first four of its eight A row blocks.
struct RingFrags { u32x4 b[4], alo[4]; };

__device__ __forceinline__ void ring_read_ahead(unsigned a_addr, unsigned b_addr, RingFrags& f) {
  f.b[0] = lds_read_b128<0>(b_addr);
  f.b[1] = lds_read_b128<1024>(b_addr);
  f.b[2] = lds_read_b128<2048>(b_addr);
  f.b[3] = lds_read_b128<3072>(b_addr);
  f.alo[0] = lds_read_b128<0>(a_addr);
  f.alo[1] = lds_read_b128<1024>(a_addr);
  f.alo[2] = lds_read_b128<2048>(a_addr);
  f.alo[3] = lds_read_b128<3072>(a_addr);
}

// One slot of one wave: 16 MFMAs on rows 0..63 of the wave's 128 (fragments `cur`, read during the previous
// slot), then 16 on rows 64..127 (read at the top).  Between the two the wave waits for the NEXT slot to have
// landed (all but its VM youngest pieces), meets the other waves at the barrier and reads that slot's `nxt`
// fragments, so no MFMA ever waits on an LDS read issued just before it.  After that barrier every wave holds
// all fragments of this slot in registers: the slot is free, and the loads of the k-tile that will reuse it go
// out behind the barrier — NP = 1: pieces 0..3 (first slot of that k-tile) inside the second MFMA group;
// NP = 2: pieces 4..7 (second slot) inside the first group of the following slot.  VM < 0: last slot.
// PH as in the two-stage kernel (the two waves of a SIMD issue their pieces in opposite phases).
template <int PH, int NP, int VM>
__device__ __forceinline__ void ring_half(const RingFrags& cur, RingFrags& nxt, unsigned a_addr, unsigned a_next, unsigned b_next,
                                          f32x4 (&acc)[8][4], const char* a_src, const char* b_src, const unsigned (&aoff)[2],
                                          const unsigned (&boff)[2], bf16_t* d0, bf16_t* d1) {
  u32x4 ahi[4];
  ahi[0] = lds_read_b128<4096>(a_addr);
  ahi[1] = lds_read_b128<5120>(a_addr);
  ahi[2] = lds_read_b128<6144>(a_addr);
  ahi[3] = lds_read_b128<7168>(a_addr);
  lgkm_wait<4>();
  __builtin_amdgcn_sched_barrier(0);
  auto piece = [&](int n) {  // n is a compile-time constant after unrolling
    const int h = n & 1, ab = n >> 2, j = (n >> 1) & 1;  // the two 64-byte halves of a line back to back
    const char* src = (ab ? b_src + boff[j] : a_src + aoff[j]) + h * 64;
    bf16_t* dst = (h ? d1 : d0) + ab * (256 * 32) + j * 512;
    __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if constexpr (NP == 2 && PH == 0) piece(4 + g);
    mfma_row(acc[g], cur.alo[g], cur.b);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NP == 2 && PH == 1) piece(4 + g);
  }
  lgkm_wait<0>();
  if constexpr (VM >= 0) {
    vm_wait<VM>();
    __builtin_amdgcn_s_barrier();
    ring_read_ahead(a_next, b_next, nxt);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if constexpr (NP == 1 && PH == 0) piece(g);
    mfma_row(acc[4 + g], ahi[g], cur.b);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NP == 1 && PH == 1) piece(g);
  }
}

template <int PH, int DBG>
__device__ __forceinline__ void ring256_mainloop(const GemmArgs& g, const bf16_t* A, const bf16_t* B, const bf16_t* A2,
                                                 const bf16_t* B2, int bm, int bn, bf16_t* smem, int wid, int lane,
                                                 f32x4 (&acc)[8][4]) {
  const int wr = wid >> 2, wc = wid & 3;
  // k-tiles (64 k = two slots) [0, nt1) come from (A, B), [nt1, nt) from the optional second segment
  const int nt1 = g.K / GBK, nt = nt1 + g.K2 / GBK;
  unsigned aoff[2], boff[2];
  ring_piece_offsets(g.lda, bm, g.M, wid, lane, aoff);
  ring_piece_offsets(g.ldb, bn, g.N, wid, lane, boff);
  const int r = lane & 15, gq = lane >> 4;
  const unsigned foff = (unsigned)(r * 64 + ((gq ^ ((r >> 1) & 3)) << 4));
  const unsigned smem_addr = lds_addr(smem);
  const unsigned a_rd = smem_addr + wr * (128 * 64) + foff, b_rd = smem_addr + 256 * 64 + wc * (64 * 64) + foff;
  const char* a_seg0 = reinterpret_cast<const char*>(A + (long)bm * g.lda);
  const char* b_seg0 = reinterpret_cast<const char*>(B + (long)bn * g.ldb);
  const char* a_seg1 = reinterpret_cast<const char*>(A2 + (long)bm * g.lda);
  const char* b_seg1 = reinterpret_cast<const char*>(B2 + (long)bn * g.ldb);
  bf16_t* wdst = smem + 2 * wid * 512;  // this wave's first A piece inside slot 0

  // k-tiles 0 and 1 fill slots 0..3
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const char* as = p < nt1 ? a_seg0 + p * 128 : a_seg1 + (p - nt1) * 128;
    const char* bs = p < nt1 ? b_seg0 + p * 128 : b_seg1 + (p - nt1) * 128;
    bf16_t* d = wdst + 2 * p * RING_SLOT_ELEMS;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h) __builtin_amdgcn_global_load_lds(as + aoff[j] + h * 64, d + h * RING_SLOT_ELEMS + j * 512, 16, 0, 0);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        __builtin_amdgcn_global_load_lds(bs + boff[j] + h * 64, d + h * RING_SLOT_ELEMS + 256 * 32 + j * 512, 16, 0, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  const unsigned SB = RING_SLOT_ELEMS * 2;  // slot size in bytes
  RingFrags fx, fy;
  vm_wait<9>();  // slot 0 has landed: the last piece of its k-tile and the next k-tile are younger
  __builtin_amdgcn_s_barrier();
  ring_read_ahead(a_rd, b_rd, fx);
  // Slot 2q is at ring position e, 2q + 1 at o, 2q + 2 at e2.  Waits at the barriers inside the slots: in
  // slot 2q for slot 2q + 1 (younger: k-tile q + 1 = 8 pieces), in slot 2q + 1 for slot 2q + 2 (younger: the
  // second slot of k-tile q + 1 and the eight pieces of k-tile q + 2 = 12).  nt >= 2 (checked on the host);
  // the last two k-tiles, which issue no loads and wait for fewer pieces, are peeled so that the loop body
  // has no branches.
  int e = 0;
  for (int q = 0; q < nt - 2; ++q) {
    const int o = e == RING_SLOTS - 1 ? 0 : e + 1;
    const int e2 = o == RING_SLOTS - 1 ? 0 : o + 1;
    const int s0 = e == 0 ? RING_SLOTS - 1 : e - 1;  // ring position of slot 2q + 4
    const int p = q + 2;
    const char* as = p < nt1 ? a_seg0 + p * 128 : a_seg1 + (p - nt1) * 128;
    const char* bs = p < nt1 ? b_seg0 + p * 128 : b_seg1 + (p - nt1) * 128;
    bf16_t* d0 = wdst + s0 * RING_SLOT_ELEMS;
    bf16_t* d1 = wdst + e * RING_SLOT_ELEMS;
    ring_half<PH, DBG ? 0 : 1, 8>(fx, fy, a_rd + e * SB, a_rd + o * SB, b_rd + o * SB, acc, as, bs, aoff, boff, d0, d1);
    __builtin_amdgcn_sched_barrier(0);
    ring_half<PH, DBG ? 0 : 2, 9>(fy, fx, a_rd + o * SB, a_rd + e2 * SB, b_rd + e2 * SB, acc, as, bs, aoff, boff, d0, d1);
    __builtin_amdgcn_sched_barrier(0);
    e = e2;
  }
  {  // k-tile nt - 2
    const int o = e == RING_SLOTS - 1 ? 0 : e + 1;
    const int e2 = o == RING_SLOTS - 1 ? 0 : o + 1;
    ring_half<PH, 0, 8>(fx, fy, a_rd + e * SB, a_rd + o * SB, b_rd + o * SB, acc, a_seg0, b_seg0, aoff, boff, smem, smem);
    __builtin_amdgcn_sched_barrier(0);
    ring_half<PH, 0, 1>(fy, fx, a_rd + o * SB, a_rd + e2 * SB, b_rd + e2 * SB, acc, a_seg0, b_seg0, aoff, boff, smem, smem);
    __builtin_amdgcn_sched_barrier(0);
    e = e2;
  }
  {  // k-tile nt - 1
    const int o = e == RING_SLOTS - 1 ? 0 : e + 1;
    ring_half<PH, 0, 0>(fx, fy, a_rd + e * SB, a_rd + o * SB, b_rd + o * SB, acc, a_seg0, b_seg0, aoff, boff, smem, smem);
    __builtin_amdgcn_sched_barrier(0);
    ring_half<PH, 0, -1>(fy, fx, a_rd + o * SB, a_rd, b_rd, acc, a_seg0, b_seg0, aoff, boff, smem, smem);
  }
}

// DBG = 1 (timing experiments only, wrong results): no loads after the first two k-tiles
template <typename TC, int DBG = 0>
__global__ __launch_bounds__(NT256, 1) void gemm_ring256_kernel(GemmArgs g) {
  __shared__ __align__(16) bf16_t smem[RING_SLOTS * RING_SLOT_ELEMS];  // 160 KiB: the whole LDS of a CU
  const int gx = gridDim.x, gy = gridDim.y;
  const int txy = xcd_remap(blockIdx.x + gx * blockIdx.y, gx * gy);
  const bf16_t* A = reinterpret_cast<const bf16_t*>(g.A);
  const bf16_t* B = reinterpret_cast<const bf16_t*>(g.B);
  TC* Cp = reinterpret_cast<TC*>(g.C);
  // tile order as in the two-stage kernel: groups of group_m tile rows, walked down the rows first
  const int GM = gx > 4 ? g.group_m : 1;
  const int grp = txy / (GM * gx), first = grp * GM, rem = txy - grp * GM * gx;
  const int gsz = gy - first < GM ? gy - first : GM;
  const int bm = (first + rem % gsz) * 256, bn = (rem / gsz) * 256;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wid >> 2, wc = wid & 3;
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bf16_t* A2 = reinterpret_cast<const bf16_t*>(g.A2);
  const bf16_t* B2 = reinterpret_cast<const bf16_t*>(g.B2);
  if (wr == 0) ring256_mainloop<0, DBG>(g, A, B, A2, B2, bm, bn, smem, wid, lane, acc);
  else ring256_mainloop<1, DBG>(g, A, B, A2, B2, bm, bn, smem, wid, lane, acc);
  gemm_epilogue<bf16_t, TC, 8>(acc, g, Cp, g.vec_ok && bn + 256 <= g.N, bm + wr * 128, bn + wc * 64, lane);
}

}  // namespace

int pg_internal_ring_dbg = 0;
int pg_internal_launch_ring256(const void* args, int M, int N, int c_is_f32, hipStream_t st) {
  const GemmArgs& a = *reinterpret_cast<const GemmArgs*>(args);
  dim3 grid((N + 255) / 256, (M + 255) / 256, 1);
  if (pg_internal_ring_dbg == 1 && !c_is_f32) hipLaunchKernelGGL((gemm_ring256_kernel<bf16_t, 1>), grid, dim3(NT256), 0, st, a);
  else if (c_is_f32) hipLaunchKernelGGL((gemm_ring256_kernel<float>), grid, dim3(NT256), 0, st, a);
  else hipLaunchKernelGGL((gemm_ring256_kernel<bf16_t>), grid, dim3(NT256), 0, st, a);
  return hipGetLastError() == hipSuccess ? PG_OK : PG_ERR_LAUNCH;
}
