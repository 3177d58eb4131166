// Shared by the GEMM translation units (included inside their anonymous namespaces): the kernel
// argument block, the epilogue every tile shape ends in, and the inline-asm LDS read helpers.
#pragma once

struct GemmArgs {
  const void* A; const void* B; void* C;
  int M, N, K;
  long lda, ldb, ldc;
  int batch_inner;
  long sAo, sAi, sBo, sBi, sCo, sCi;
  float alpha, beta;
  const float* bias;
  const void* residual; long ldr;
  void* aux; long ldaux;
  void* aux2;
  int act;
  int vec_ok;
  const void* A2; const void* B2; int K2;  // optional second reduction segment (same layout and strides)
  int group_m;  // tile rows per group of the grouped tile order (1 = row-major)
  // Ragged batches (128x64-tile kernel only): batch (zo, zi) is sequence zo, head zi.  Its token-row operands
  // (bits of seq_ops: 1 = A, 2 = B, 4 = C) start at row seq_off[zo] - seq_tok0 of their buffers and M = K =
  // seq_off[zo+1] - seq_off[zo]; score-shaped operands keep their batch strides.
  const int* seq_off; int seq_tok0; int seq_ops;
};

typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int IMM>
__device__ __forceinline__ u32x4 lds_read_b128(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM) : "memory");
  return v;
}
template <int IMM>
__device__ __forceinline__ u32x2 lds_read_tr_b64(unsigned addr) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM) : "memory");
  return v;
}

__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}

template <int N> __device__ __forceinline__ void lgkm_wait() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N < 15 ? N : 15) : "memory");
}

__device__ __forceinline__ void mfma_row(f32x4 (&acc)[4], const u32x4& a, const u32x4 (&b)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, b[j]), __builtin_bit_cast(bf16x8_t, a), acc[j], 0, 0, 0);
}

template <typename TC> __device__ __forceinline__ void load4(const TC* p, float* o, bool vec, int nvalid) {
  if (vec) {
    VecIO<TC, 4>::load(p, o);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = j < nvalid ? to_f32<TC>(p[j]) : 0.f;
  }
}
template <typename TC> __device__ __forceinline__ void store4(TC* p, const float* o, bool vec, int nvalid) {
  if (vec) {
    VecIO<TC, 4>::store(p, o);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < nvalid) p[j] = from_f32<TC>(o[j]);
  }
}

// ---- epilogue: lane owns C[m][n..n+3], m = ... + (lane&15), n = ... + 4*(lane>>4)
// Interior tiles with vector-aligned operands take a branch-free path: the bias is fetched
// once per column group as a float4, and the residual / auxiliary / accumulate loads of one
// output row group are all issued before the first is consumed.
template <typename T> __device__ __forceinline__ float gelu_epi(float x);
template <> __device__ __forceinline__ float gelu_epi<float>(float x) { return gelu_f(x); }
template <> __device__ __forceinline__ float gelu_epi<bf16_t>(float x) { return gelu_fast(x); }
template <typename T> __device__ __forceinline__ float gelu_d1_epi(float x);
template <> __device__ __forceinline__ float gelu_d1_epi<float>(float x) { return gelu_d1(x); }
template <> __device__ __forceinline__ float gelu_d1_epi<bf16_t>(float x) { return gelu_d1_fast(x); }

template <typename T, typename TC, int MI>
__device__ __forceinline__ void gemm_epilogue_interior(f32x4 (&acc)[MI][4], const GemmArgs& g, TC* Cp, int m_base, int n_base,
                                                       int lane) {
  const int n0 = n_base + 4 * (lane >> 4);
  const int m0 = m_base + (lane & 15);
  // A tile over the lower edge of C: rows past the last one are computed and never stored; what the epilogue
  // reads for them comes from the last row, so every load stays unconditional.
  const int m_last = g.M - 1;
  float bv[4][4];
  if (g.bias) {
#pragma unroll
    for (int j = 0; j < 4; ++j) VecIO<float, 4>::load(g.bias + n0 + j * 16, bv[j]);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[j][e] = 0.f;
  }
  const T* aux = reinterpret_cast<const T*>(g.aux);
  const T* res = reinterpret_cast<const T*>(g.residual);
  // An accumulating product with no other per-element operand (the weight gradients: C += dY^T X) fetches C two row
  // blocks at a time, eight loads in flight together; inside the loop below each load waits alone in its branch.
  constexpr int PB = MI < 2 ? MI : 2;
  const bool pre_c = (g.vec_ok & 4) && g.beta != 0.f && g.act != PG_ACT_MUL_DGELU && !res;
  float cpre[PB][4][4];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const bool row_ok = m0 + i * 16 <= m_last;
    const long m = row_ok ? m0 + i * 16 : m_last;
    if (i % PB == 0 && pre_c) {
#pragma unroll
      for (int i2 = 0; i2 < PB; ++i2) {
        const int mp = m0 + (i + i2) * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) VecIO<TC, 4>::load(Cp + (long)(mp <= m_last ? mp : m_last) * g.ldc + n0 + j * 16, cpre[i2][j]);
      }
    }
#pragma unroll
    for (int jh = 0; jh < 4; jh += 2) {  // two column groups at a time keeps the live set small
      float v[2][4], u[2][4], r[2][4], c0[2][4];
      if (g.act == PG_ACT_MUL_DGELU) {
#pragma unroll
        for (int j = 0; j < 2; ++j) VecIO<T, 4>::load(aux + m * g.ldaux + n0 + (jh + j) * 16, u[j]);
      }
      if (res) {
#pragma unroll
        for (int j = 0; j < 2; ++j) VecIO<T, 4>::load(res + m * g.ldr + n0 + (jh + j) * 16, r[j]);
      }
      if (g.beta != 0.f) {
        if (pre_c) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) c0[j][e] = cpre[i % PB][jh + j][e];
        } else {
#pragma unroll
          for (int j = 0; j < 2; ++j) VecIO<TC, 4>::load(Cp + m * g.ldc + n0 + (jh + j) * 16, c0[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[j][e] = acc[i][jh + j][e] * g.alpha + bv[jh + j][e];
      if (g.act == PG_ACT_GELU) {
        if (g.aux && row_ok) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
            VecIO<T, 4>::store(reinterpret_cast<T*>(g.aux) + m * g.ldaux + n0 + (jh + j) * 16, v[j]);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[j][e] = gelu_epi<T>(v[j][e]);
      } else if (g.act == PG_ACT_MUL_DGELU) {
        if (g.aux2 && row_ok) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
            VecIO<T, 4>::store(reinterpret_cast<T*>(g.aux2) + m * g.ldaux + n0 + (jh + j) * 16, v[j]);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[j][e] *= gelu_d1_epi<T>(u[j][e]);
      }
      if (res) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[j][e] += r[j][e];
      }
      if (g.beta != 0.f) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[j][e] += g.beta * c0[j][e];
      }
      if (row_ok) {
#pragma unroll
        for (int j = 0; j < 2; ++j) VecIO<TC, 4>::store(Cp + m * g.ldc + n0 + (jh + j) * 16, v[j]);
      }
    }
  }
}

// bf16 output with 16-byte-aligned operands: the two lanes that hold columns 4g..4g+3 of the
// 16x16 tiles j and j+1 (g = lane >> 4) first trade halves across 16-lane rows
// (v_permlane16_swap), after which a lane owns eight consecutive columns of ONE tile: even g
// keeps tile j, odd g tile j+1, columns 8*(g>>1)..+7.  Every access of the epilogue is then a
// 16-byte vector and the number of store instructions, which is what bounds the store tail of a
// workgroup that owns a whole CU, is halved.
template <int MI>
__device__ __forceinline__ void gemm_epilogue_interior_b8(f32x4 (&acc)[MI][4], const GemmArgs& g, bf16_t* Cp, int m_base,
                                                          int n_base, int lane) {
  const int gq = lane >> 4;
  const int m0 = m_base + (lane & 15);
  const int m_last = g.M - 1;  // (rows past it: computed, never stored, their operands read from row m_last)
  const bf16_t* aux = reinterpret_cast<const bf16_t*>(g.aux);
  const bf16_t* res = reinterpret_cast<const bf16_t*>(g.residual);
  // this lane's two column groups (jh = 0, 2) and their biases
  int nn[2];
  float bv[2][8];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    nn[h] = n_base + (2 * h + (gq & 1)) * 16 + 8 * (gq >> 1);
    if (g.bias) {
      VecIO<float, 4>::load(g.bias + nn[h], bv[h]);
      VecIO<float, 4>::load(g.bias + nn[h] + 4, bv[h] + 4);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) bv[h][e] = 0.f;
    }
  }
  // The operand the epilogue reads per output element (the pre-activation of x GELU', else the residual, else C
  // for an accumulate) is fetched four row blocks at a time, eight loads in flight together: read inside the
  // loop below, each load sits in a branch of its own behind a full wait and the latencies follow one another.
  // (All 2 MI loads of a 256x256 tile's wave at once cost that kernel twelve spilled registers.)
  constexpr int PB = MI < 4 ? MI : 4;  // row blocks per batch
  int pre_kind = 0;  // 1 = pre-activation (aux), 2 = residual, 3 = C
  const bf16_t* pp = nullptr;
  long pld = 0;
  if (g.vec_ok & 4) {
    if (g.act == PG_ACT_MUL_DGELU) { pp = aux; pld = g.ldaux; pre_kind = 1; }
    else if (res) { pp = res; pld = g.ldr; pre_kind = 2; }
    else if (g.beta != 0.f) { pp = Cp; pld = g.ldc; pre_kind = 3; }
  }
  uint4 pre[PB][2];
  auto unpack8 = [](const uint4& q, float* o) {
    o[0] = __uint_as_float(q.x << 16); o[1] = __uint_as_float(q.x & 0xffff0000u);
    o[2] = __uint_as_float(q.y << 16); o[3] = __uint_as_float(q.y & 0xffff0000u);
    o[4] = __uint_as_float(q.z << 16); o[5] = __uint_as_float(q.z & 0xffff0000u);
    o[6] = __uint_as_float(q.w << 16); o[7] = __uint_as_float(q.w & 0xffff0000u);
  };
  // Row block outermost: the two 64-byte halves of every 128-byte output line leave in consecutive
  // store instructions.  (Column group outermost put eight stores between them, and fabric-side
  // write bytes of a 4096-wide output measured 1.4x the output size: lines written back half full,
  // then again.)
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const bool row_ok = m0 + i * 16 <= m_last;
    const long m = row_ok ? m0 + i * 16 : m_last;
    if (i % PB == 0 && pre_kind) {
#pragma unroll
      for (int i2 = 0; i2 < PB; ++i2) {
        const int mp = m0 + (i + i2) * 16;
#pragma unroll
        for (int h = 0; h < 2; ++h) pre[i2][h] = *reinterpret_cast<const uint4*>(pp + (long)(mp <= m_last ? mp : m_last) * pld + nn[h]);
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int jh = 2 * h;
      float v[8], u[8], r[8];
      if (g.act == PG_ACT_MUL_DGELU) {
        if (pre_kind == 1) unpack8(pre[i % PB][h], u);
        else VecIO<bf16_t, 8>::load(aux + m * g.ldaux + nn[h], u);
      }
      if (res) {
        if (pre_kind == 2) unpack8(pre[i % PB][h], r);
        else VecIO<bf16_t, 8>::load(res + m * g.ldr + nn[h], r);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[i][jh][e]), __float_as_uint(acc[i][jh + 1][e]), false, false);
        v[e] = __uint_as_float(sw[0]) * g.alpha + bv[h][e];
        v[4 + e] = __uint_as_float(sw[1]) * g.alpha + bv[h][4 + e];
      }
      if (g.act == PG_ACT_GELU) {
        if (g.aux && row_ok) VecIO<bf16_t, 8>::store(reinterpret_cast<bf16_t*>(g.aux) + m * g.ldaux + nn[h], v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_fast(v[e]);
      } else if (g.act == PG_ACT_MUL_DGELU) {
        if (g.aux2 && row_ok) VecIO<bf16_t, 8>::store(reinterpret_cast<bf16_t*>(g.aux2) + m * g.ldaux + nn[h], v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= gelu_d1_fast(u[e]);
      }
      if (res) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += r[e];
      }
      if (g.beta != 0.f) {
        float c0[8];
        if (pre_kind == 3) unpack8(pre[i % PB][h], c0);
        else VecIO<bf16_t, 8>::load(Cp + m * g.ldc + nn[h], c0);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += g.beta * c0[e];
      }
      if (row_ok) VecIO<bf16_t, 8>::store(Cp + m * g.ldc + nn[h], v);
    }
  }
}

// `interior`: the workgroup tile's columns all lie inside C and every operand is vector-aligned (the tile may
// hang over the lower edge: the vector paths skip the rows past M).
// (m_base, n_base): first row / column of this wave's MI x 4 grid of 16x16 tiles.
template <typename T, typename TC, int MI>
__device__ __forceinline__ void gemm_epilogue(f32x4 (&acc)[MI][4], const GemmArgs& g, TC* Cp, bool interior, int m_base,
                                              int n_base, int lane) {
  if (interior) {
    if constexpr (sizeof(T) == 2 && sizeof(TC) == 2) {
      if ((g.vec_ok & 3) >= 2) {
        gemm_epilogue_interior_b8<MI>(acc, g, Cp, m_base, n_base, lane);
        return;
      }
    }
    gemm_epilogue_interior<T, TC, MI>(acc, g, Cp, m_base, n_base, lane);
    return;
  }
  const bool vec = g.vec_ok != 0;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = m_base + i * 16 + (lane & 15);
    if (m >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n_base + j * 16 + 4 * (lane >> 4);
      if (n >= g.N) continue;
      const int nvalid = g.N - n < 4 ? g.N - n : 4;
      const bool v4 = vec && nvalid == 4;
      float v[4] = {acc[i][j][0] * g.alpha, acc[i][j][1] * g.alpha, acc[i][j][2] * g.alpha, acc[i][j][3] * g.alpha};
      if (g.bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (e < nvalid) v[e] += g.bias[n + e];
      }
      if (g.act == PG_ACT_GELU) {
        if (g.aux) store4<T>(reinterpret_cast<T*>(g.aux) + (long)m * g.ldaux + n, v, v4, nvalid);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_epi<T>(v[e]);
      } else if (g.act == PG_ACT_MUL_DGELU) {
        float u[4];
        load4<T>(reinterpret_cast<const T*>(g.aux) + (long)m * g.ldaux + n, u, v4, nvalid);
        if (g.aux2) store4<T>(reinterpret_cast<T*>(g.aux2) + (long)m * g.ldaux + n, v, v4, nvalid);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= gelu_d1_epi<T>(u[e]);
      }
      if (g.residual) {
        float r[4];
        load4<T>(reinterpret_cast<const T*>(g.residual) + (long)m * g.ldr + n, r, v4, nvalid);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += r[e];
      }
      TC* cp = Cp + (long)m * g.ldc + n;
      if (g.beta != 0.f) {
        float c0[4];
        load4<TC>(cp, c0, v4, nvalid);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += g.beta * c0[e];
      }
      store4<TC>(cp, v, v4, nvalid);
    }
  }
}

