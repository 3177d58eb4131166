// Shared device helpers for the ProtBERT GAN engine (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bf16 storage

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define PG_WAVE 64

__device__ __forceinline__ float bf16_bits_to_f32(bf16_t v) {
  return __uint_as_float(((uint32_t)v) << 16);
}
// Round-to-nearest-even through the hardware convert (keeps NaN a NaN).
__device__ __forceinline__ bf16_t f32_to_bf16_bits(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(uint16_t, h);
}

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return bf16_bits_to_f32(v); }

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return f32_to_bf16_bits(v); }

// Exact (erf) GELU and its first two derivatives, as torch.nn.functional.gelu computes it.
__device__ __forceinline__ float gelu_f(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float gelu_d1(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
__device__ __forceinline__ float gelu_d2(float x) {
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return pdf * (2.0f - x * x);
}

// GELU for values that are rounded to bf16 anyway (GEMM epilogues of the bf16 path; the fp32 path
// keeps the library erff):  gelu(x) ~= x * sigmoid(t(x)),  t = x (a + b x^2 + c x^4), a minimax fit
// to the erf form over the real line: |error| <= 2.6e-5 for the value and <= 1.1e-4 for the first
// derivative (a bf16 rounding step is 3.9e-3 relative).  c < 0, so t is evaluated on x clamped to
// [-8, 8], where the sigmoid has long saturated.  Seven 4-cycle VALU operations, one v_exp_f32 and
// one v_rcp_f32 per value, against fifteen plus two for a rational erf.
constexpr float GELU_SIG_A = 1.59501577f, GELU_SIG_B = 7.40112920e-2f, GELU_SIG_C = -7.03033578e-4f;
constexpr float NEG_LOG2E = -1.44269504088896340736f;
__device__ __forceinline__ float gelu_fast(float x) {
  const float xc = __builtin_amdgcn_fmed3f(x, -8.0f, 8.0f);
  const float x2 = xc * xc;
  const float q = fmaf(x2, fmaf(x2, GELU_SIG_C * NEG_LOG2E, GELU_SIG_B * NEG_LOG2E), GELU_SIG_A * NEG_LOG2E);
  const float e = __builtin_amdgcn_exp2f(xc * q);  // exp(-t)
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float gelu_d1_fast(float x) {
  const float xc = __builtin_amdgcn_fmed3f(x, -8.0f, 8.0f);
  const float x2 = xc * xc;
  const float q = fmaf(x2, fmaf(x2, GELU_SIG_C * NEG_LOG2E, GELU_SIG_B * NEG_LOG2E), GELU_SIG_A * NEG_LOG2E);
  const float tp = fmaf(x2, fmaf(x2, 5.0f * GELU_SIG_C, 3.0f * GELU_SIG_B), GELU_SIG_A);  // dt/dx
  const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(xc * q));
  return fmaf(x * tp, s - s * s, s);
}

// Workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so hardware ids b and
// b + 8 share an L2.  This bijection hands every XCD a contiguous range of logical ids, so
// neighbouring tiles (same A rows / same B columns / same attention head) meet in one L2.
// Placement is a speed assumption only; nothing depends on it for correctness.
__device__ __forceinline__ int xcd_remap(int id, int n) {
  const int q = n >> 3, r = n & 7, x = id & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Vector load/store of N consecutive activation elements as float.
template <typename T, int N> struct VecIO;
template <> struct VecIO<float, 4> {
  static __device__ __forceinline__ void load(const float* p, float* o) {
    float4 v = *reinterpret_cast<const float4*>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  }
  static __device__ __forceinline__ void store(float* p, const float* o) {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
  }
};
template <> struct VecIO<bf16_t, 8> {
  static __device__ __forceinline__ void load(const bf16_t* p, float* o) {
    uint4 v = *reinterpret_cast<const uint4*>(p);
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
    o[4] = __uint_as_float(v.z << 16); o[5] = __uint_as_float(v.z & 0xffff0000u);
    o[6] = __uint_as_float(v.w << 16); o[7] = __uint_as_float(v.w & 0xffff0000u);
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float* o) {
    uint4 v;
    v.x = (uint32_t)f32_to_bf16_bits(o[0]) | ((uint32_t)f32_to_bf16_bits(o[1]) << 16);
    v.y = (uint32_t)f32_to_bf16_bits(o[2]) | ((uint32_t)f32_to_bf16_bits(o[3]) << 16);
    v.z = (uint32_t)f32_to_bf16_bits(o[4]) | ((uint32_t)f32_to_bf16_bits(o[5]) << 16);
    v.w = (uint32_t)f32_to_bf16_bits(o[6]) | ((uint32_t)f32_to_bf16_bits(o[7]) << 16);
    *reinterpret_cast<uint4*>(p) = v;
  }
  // streaming store: the line is not kept in the caches on its way out
  static __device__ __forceinline__ void store_nt(bf16_t* p, const float* o) {
    typedef __attribute__((ext_vector_type(4))) unsigned nt_u4;
    nt_u4 v;
    v[0] = (uint32_t)f32_to_bf16_bits(o[0]) | ((uint32_t)f32_to_bf16_bits(o[1]) << 16);
    v[1] = (uint32_t)f32_to_bf16_bits(o[2]) | ((uint32_t)f32_to_bf16_bits(o[3]) << 16);
    v[2] = (uint32_t)f32_to_bf16_bits(o[4]) | ((uint32_t)f32_to_bf16_bits(o[5]) << 16);
    v[3] = (uint32_t)f32_to_bf16_bits(o[6]) | ((uint32_t)f32_to_bf16_bits(o[7]) << 16);
    __builtin_nontemporal_store(v, reinterpret_cast<nt_u4*>(p));
  }
};
template <> struct VecIO<bf16_t, 4> {
  static __device__ __forceinline__ void load(const bf16_t* p, float* o) {
    uint2 v = *reinterpret_cast<const uint2*>(p);
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float* o) {
    uint2 v;
    v.x = (uint32_t)f32_to_bf16_bits(o[0]) | ((uint32_t)f32_to_bf16_bits(o[1]) << 16);
    v.y = (uint32_t)f32_to_bf16_bits(o[2]) | ((uint32_t)f32_to_bf16_bits(o[3]) << 16);
    *reinterpret_cast<uint2*>(p) = v;
  }
};
