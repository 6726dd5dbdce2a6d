// Shared device/host helpers for libcoral_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/coral_amd.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define CA_WAVE 64

void ca_set_error(const char* fmt, ...);

#define CA_CHECK_ARG(cond, ...)     \
  do {                              \
    if (!(cond)) {                  \
      ca_set_error(__VA_ARGS__);    \
      return CA_ERR_ARG;            \
    }                               \
  } while (0)

#define CA_CHECK_LAUNCH(name)                                              \
  do {                                                                     \
    hipError_t e__ = hipGetLastError();                                    \
    if (e__ != hipSuccess) {                                               \
      ca_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return CA_ERR_LAUNCH;                                                \
    }                                                                      \
  } while (0)

// ---- bf16 <-> f32 (round-to-nearest-even via the hardware cast; NaN stays NaN) ----
__device__ __forceinline__ float bf2f(unsigned short u) {
  return __builtin_bit_cast(float, ((unsigned int)u) << 16);
}
__device__ __forceinline__ unsigned short f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}

// exact (erf) GELU and its derivative, fp32
__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float dgelu_erf(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// ---- 64-lane wavefront reductions ----
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

// counter-based hash for fused dropout masks (same bits in forward and backward)
__device__ __forceinline__ uint32_t ca_hash32(uint64_t seed, uint64_t idx) {
  uint64_t z = idx + seed * 0x9E3779B97F4A7C15ull + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (uint32_t)(z >> 32);
}
__device__ __forceinline__ uint64_t ca_hash64(uint64_t seed, uint64_t idx) {
  uint64_t z = idx + seed * 0x9E3779B97F4A7C15ull + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// Keep decision for element `idx` (a flat index): one 64-bit hash serves the 4 elements of an
// aligned group (16 bits each), so the mask depends only on (seed, idx) -- identical in the forward
// GELU epilogue and the backward GELU' epilogue -- at a quarter of the hashing cost.
__device__ __forceinline__ bool ca_dropout_keep(uint64_t seed, uint64_t idx, float p) {
  const uint64_t h = ca_hash64(seed, idx >> 2);
  const unsigned int bits = (unsigned int)(h >> (16 * (idx & 3))) & 0xFFFFu;
  return (float)bits * (1.0f / 65536.0f) >= p;
}
// the same decision for 4 consecutive elements starting at a multiple of 4 (one hash)
__device__ __forceinline__ unsigned int ca_dropout_keep4(uint64_t seed, uint64_t idx4, float p) {
  const uint64_t h = ca_hash64(seed, idx4 >> 2);
  unsigned int m = 0;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    m |= ((float)((unsigned int)(h >> (16 * e)) & 0xFFFFu) * (1.0f / 65536.0f) >= p ? 1u : 0u) << e;
  return m;
}

// out[i] (+)= sum_p partial[p*stride + i]  (defined in norm.hip)
void ca_reduce_partials_launch(const float* partial, int nparts, int64_t stride, int n, float* out,
                               int accumulate, hipStream_t s);
