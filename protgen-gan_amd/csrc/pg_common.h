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

// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7): used where the result is rounded
// to bf16 anyway; the fp32 path keeps the library erff.
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __frcp_rn(1.0f + 0.3275911f * ax);
  const float p = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float r = 1.0f - p * __expf(-ax * ax);
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_fast(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_d1_fast(float x) {
  const float cdf = 0.5f * (1.0f + erf_fast(x * 0.70710678118654752440f));
  return cdf + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
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
