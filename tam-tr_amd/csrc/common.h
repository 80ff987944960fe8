// common.h - shared device helpers for the TAM-TR gfx950 kernels (wave64, CDNA4 only: no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tamtr_hip.h"

#define WAVE 64

typedef uint16_t bf16_t;  // raw bf16 bits
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// round-to-nearest-even; NaN stays NaN (plain cast path of the guide: v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}

template <typename T>
struct Elt;
template <>
struct Elt<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
  static __device__ __forceinline__ void ld4(const float* p, float (&o)[4]) {
    float4 t = *reinterpret_cast<const float4*>(p);
    o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
  }
  static __device__ __forceinline__ void st4(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
};
template <>
struct Elt<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
  static __device__ __forceinline__ void ld4(const bf16_t* p, float (&o)[4]) {
    uint2 t = *reinterpret_cast<const uint2*>(p);
    o[0] = __uint_as_float(t.x << 16); o[1] = __uint_as_float(t.x & 0xffff0000u);
    o[2] = __uint_as_float(t.y << 16); o[3] = __uint_as_float(t.y & 0xffff0000u);
  }
  static __device__ __forceinline__ void st4(bf16_t* p, const float (&v)[4]) {
    uint2 t;
    t.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
    t.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
    *reinterpret_cast<uint2*>(p) = t;
  }
};

// butterfly reductions inside an aligned group of WIDTH lanes (WIDTH power of two <= 64)
template <int WIDTH>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = WIDTH / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
  return v;
}
template <int WIDTH>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int o = WIDTH / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, WAVE));
  return v;
}

static inline int tamtr_launch_status() { return hipGetLastError() == hipSuccess ? TAMTR_OK : TAMTR_ELAUNCH; }
