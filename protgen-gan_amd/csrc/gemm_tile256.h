// Pieces of the 256 x 256-tile direct-to-LDS GEMM shared by its kernels (included inside the
// anonymous namespace of each translation unit, after gemm_shared.h): tile constants, the LDS
// swizzles, per-lane fragment offsets and reads, and one k-tile of one wave.
#pragma once

constexpr int GBK = 64;
constexpr int GT_ELEMS = 128 * GBK;  // 8192 bf16 = 16 KiB per operand tile

__device__ __forceinline__ int swz_m(int k) { return ((k & 3) << 2) | ((k >> 2) & 3); }

// Fragment reads are inline asm: a compiler-visible ds_read of the staging array makes hipcc
// wait vmcnt(0) first (it cannot tell which in-flight LDS-DMA it may alias), which would
// drain the prefetch ring before every k-tile.  The waits are therefore counted by hand:
// reads return in issue order, lgkmcnt(n) = "all but the n youngest have returned".
// Per-lane byte offsets (inside one 16 KiB operand tile) of the fragments a lane reads; they do
// not depend on the k-tile, so they are computed once per kernel.
//   k-major: off[kk]; tile row block i adds i*2048 (immediate)
//   m-major: off[hi*4 + i]; k-step kk adds kk*8192 (immediate)
template <bool KMAJOR> struct FragOff { unsigned off[KMAJOR ? 2 : 8]; };

template <bool KMAJOR>
__device__ __forceinline__ FragOff<KMAJOR> frag_offsets(int wsel, int lane) {
  FragOff<KMAJOR> f;
  const int r = lane & 15, g = lane >> 4;
  if constexpr (KMAJOR) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) f.off[kk] = (unsigned)((wsel * 64 + r) * 128 + (((kk * 4 + g) ^ (r & 7)) << 4));
  } else {
    const int q = r >> 2, pp = r & 3;
#pragma unroll
    for (int hi = 0; hi < 2; ++hi)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = 8 * g + q + 4 * hi;
        const int ch = wsel * 8 + 2 * i + (pp >> 1);
        f.off[hi * 4 + i] = (unsigned)(k * 256 + ((ch ^ swz_m(k)) << 4) + 8 * (pp & 1));
      }
  }
  return f;
}

template <bool KMAJOR, int KK>
__device__ __forceinline__ void read_frags(unsigned tile_addr, const FragOff<KMAJOR>& f, u32x4 (&out)[4]) {
  if constexpr (KMAJOR) {
    const unsigned a = tile_addr + f.off[KK];
    out[0] = lds_read_b128<0>(a);
    out[1] = lds_read_b128<2048>(a);
    out[2] = lds_read_b128<4096>(a);
    out[3] = lds_read_b128<6144>(a);
  } else {
#define PG_TR_PAIR(I)                                                         \
    {                                                                         \
      u32x2 lo = lds_read_tr_b64<KK * 8192>(tile_addr + f.off[I]);           \
      u32x2 hi = lds_read_tr_b64<KK * 8192>(tile_addr + f.off[4 + I]);       \
      out[I] = (u32x4){lo[0], lo[1], hi[0], hi[1]};                          \
    }
    PG_TR_PAIR(0) PG_TR_PAIR(1) PG_TR_PAIR(2) PG_TR_PAIR(3)
#undef PG_TR_PAIR
  }
}

constexpr int NT256 = 512;

// Per-lane byte offsets (from the workgroup's first operand row / column) of the four 1 KiB
// pieces a wave copies of one 128-row sub-tile; the k-tile only moves the uniform base.
template <bool KMAJOR>
__device__ __forceinline__ void piece_offsets(long ld, int blk0, int half, int extent, int w4, int lane, unsigned (&off)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = w4 * 4 + j;
    if constexpr (KMAJOR) {
      const int c = (lane & 7) ^ (lane >> 3);
      int rg = blk0 + half * 128 + 8 * i + (lane >> 3);
      rg = rg < extent ? rg : extent - 1;
      off[j] = (unsigned)(((long)(rg - blk0) * ld + c * 8) * 2);
    } else {
      const int k = 4 * i + (lane >> 4);
      const int c = (lane & 15) ^ swz_m(k);
      int m = blk0 + half * 128 + c * 8;
      m = m <= extent - 8 ? m : extent - 8;
      off[j] = (unsigned)(((long)k * ld + (m - blk0)) * 2);
    }
  }
}


// One k-tile of one wave: 64 MFMAs in four chunks (k-step x row half), each chunk's fragment
// reads issued one chunk ahead.  With LOAD the wave's eight LDS-DMA pieces of the next k-tile
// are issued one per four MFMAs inside the first two chunks: a piece holds the issuing wave for
// tens of cycles, so a burst of eight at the top of the tile would idle the matrix pipe.  The
// two waves that share a SIMD use opposite phases (PH: piece before / after its four MFMAs) so
// one of them is always feeding the pipe.
template <bool AK, bool BK_, int PH, bool LOAD, bool MATH>
__device__ __forceinline__ void glds256_tile(unsigned a_addr, unsigned b_addr, const FragOff<AK>& fa_lo, const FragOff<AK>& fa_hi,
                                             const FragOff<BK_>& fb, f32x4 (&acc)[8][4], const char* a_src, const char* b_src,
                                             const unsigned (&aoff)[4], const unsigned (&boff)[4], bf16_t* a_dst, bf16_t* b_dst) {
  constexpr int NA = AK ? 4 : 8, NB = BK_ ? 4 : 8;  // LDS reads per group of four fragments
  if constexpr (!MATH) {
    if constexpr (LOAD) {
#pragma unroll
      for (int j = 0; j < 4; ++j) __builtin_amdgcn_global_load_lds(a_src + aoff[j], a_dst + j * 512, 16, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) __builtin_amdgcn_global_load_lds(b_src + boff[j], b_dst + j * 512, 16, 0, 0);
    }
    return;
  }
  u32x4 b0[4], b1[4], a00[4], a01[4], a10[4], a11[4];
  read_frags<BK_, 0>(b_addr, fb, b0);
  read_frags<AK, 0>(a_addr, fa_lo, a00);
  read_frags<AK, 0>(a_addr, fa_hi, a01);
  lgkm_wait<NA>();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if constexpr (LOAD && PH == 0) {
      __builtin_amdgcn_global_load_lds(a_src + aoff[g], a_dst + g * 512, 16, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    mfma_row(acc[g], a00[g], b0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (LOAD && PH == 1) {
      __builtin_amdgcn_global_load_lds(a_src + aoff[g], a_dst + g * 512, 16, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  read_frags<BK_, 1>(b_addr, fb, b1);
  read_frags<AK, 1>(a_addr, fa_lo, a10);
  lgkm_wait<NA + NB>();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if constexpr (LOAD && PH == 0) {
      __builtin_amdgcn_global_load_lds(b_src + boff[g], b_dst + g * 512, 16, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    mfma_row(acc[4 + g], a01[g], b0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (LOAD && PH == 1) {
      __builtin_amdgcn_global_load_lds(b_src + boff[g], b_dst + g * 512, 16, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  read_frags<AK, 1>(a_addr, fa_hi, a11);
  lgkm_wait<NA>();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int g = 0; g < 4; ++g) mfma_row(acc[g], a10[g], b1);
  __builtin_amdgcn_sched_barrier(0);
  lgkm_wait<0>();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int g = 0; g < 4; ++g) mfma_row(acc[4 + g], a11[g], b1);
}
