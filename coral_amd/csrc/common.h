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
// Exact-erf GELU ($TF activations "gelu" = F.gelu(approximate="none")) evaluated without libm's
// branchy erff: erfc(u) = t (a1 + t (a2 + t (a3 + t (a4 + t a5)))) exp(-u^2), t = 1/(1 + p u), u = |x|/sqrt(2)
// (Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7 on erf).  Phi(x) is taken from the complementary
// form on the negative side, so there is no 1 + erf cancellation: against the float64 value this
// rounds to the same bf16 more often than torch's own fp32 GELU does (tools/dev notes in DESIGN.md §4.4).
// Round 5: the FFN epilogues are bound by vector-instruction issue (24 VALU per element for GELU + dropout, two
// waves per SIMD), so the evaluation is written for instruction count:
//   * he = 0.5 erfc(|x| / sqrt 2) with the 0.5 folded into the coefficients (an exact scaling) and the two constant
//     factors in front of exp2 folded into one (w = |x| sqrt(log2(e) / 2), exp(-x^2/2) = exp2(-w^2));
//   * gelu(x) = max(x, 0) - |x| he (one fma; for x >= 0: x - x he = x (1 - he), for x < 0: x he) instead of forming
//     Phi(x) with a compare and a select and multiplying;
//   * |x| is clamped to 14 inside the erfc part (he underflows to 0 beyond), so an infinite x stays infinite.
// 9 VALU slots per element with packed fp32 arithmetic instead of 14; the exp(-x^2/2) is shared with the derivative.
__device__ __forceinline__ void ca_half_erfc(float x, float& he, float& e, float& ax) {
  ax = fminf(fabsf(x), 14.0f);
  const float w = ax * 0.84932180028801904272f;  // sqrt(log2(e) / 2)
  e = __builtin_amdgcn_exp2f(-w * w);            // exp(-x^2/2)
  const float t = __builtin_amdgcn_rcpf(fmaf(ax, 0.23164189f, 1.0f));  // 0.3275911 / sqrt 2
  float poly = fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f);
  poly = fmaf(t, poly, 0.5f * 1.421413741f);
  poly = fmaf(t, poly, 0.5f * -0.284496736f);
  poly = fmaf(t, poly, 0.5f * 0.254829592f);
  he = t * poly * e;
}
__device__ __forceinline__ void ca_gauss_cdf_pdf(float x, float& cdf, float& e) {
  float he, ax;
  ca_half_erfc(x, he, e, ax);
  cdf = x >= 0.f ? 1.0f - he : he;
}
__device__ __forceinline__ float gelu_erf(float x) {
  float he, e, ax;
  ca_half_erfc(x, he, e, ax);
  return fmaf(-ax, he, fmaxf(x, 0.f));
}
__device__ __forceinline__ float dgelu_erf(float x) {
  float he, e, ax;
  ca_half_erfc(x, he, e, ax);
  const float cdf = x >= 0.f ? 1.0f - he : he;
  // x pdf(x) with the clamped |x| (beyond 14 the term is below 1e-40): an infinite x gives 1 or 0, not inf * 0
  return fmaf(__builtin_copysignf(ax, x) * 0.39894228040143267794f, e, cdf);
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
// The same sum without LDS-crossbar traffic: DPP butterflies inside each row of 16 lanes (quad swaps, then row
// rotations by 4 and 8: every lane ends with its row's total), then the four row totals through v_readlane.  A
// __shfl_xor butterfly is six dependent ds_bpermute_b32 per reduction; this form is ~15 dependent VALU / SALU steps.
// Measured (same box, rocprofv3): it pays where a wave has little else in flight - the fused conv0 + LayerNorm + GELU
// kernels, one output frame per wave at a time: forward 881 -> 761 us, backward 478 -> 331 us - and costs 15 % in the
// LayerNorm kernels, whose many resident waves already hide the crossbar latency and are short of VALU issue slots.
// The result is wave-uniform; all 64 lanes must be active.
template <int CTRL>
__device__ __forceinline__ float ca_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float ca_lane(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += ca_dpp<0xB1>(v);   // quad_perm [1,0,3,2]
  v += ca_dpp<0x4E>(v);   // quad_perm [2,3,0,1]
  v += ca_dpp<0x124>(v);  // row_ror:4
  v += ca_dpp<0x128>(v);  // row_ror:8
  return (ca_lane(v, 0) + ca_lane(v, 16)) + (ca_lane(v, 32) + ca_lane(v, 48));
}
// ---- LayerNorm row arithmetic (one expression wherever a normalised value is formed) ----
__device__ __forceinline__ float ln_apply(float v, float mean, float rstd, float gm, float bt) {
  return (v - mean) * rstd * gm + bt;
}

// counter-based hash for fused dropout masks (same bits in forward and backward)
__device__ __forceinline__ uint32_t ca_hash32(uint64_t seed, uint64_t idx) {
  uint64_t z = idx + seed * 0x9E3779B97F4A7C15ull + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (uint32_t)(z >> 32);
}
// Dropout keep decisions.  The mask of element `idx` (a flat index) depends only on (seed, idx), so the
// forward GELU epilogue and the backward GELU' epilogue regenerate the same mask instead of storing
// it.  One 32-bit integer hash (two multiply-xorshift rounds) plus one cheap second word serve the 4
// elements of an aligned group, 16 bits each, compared against an integer threshold.
__device__ __forceinline__ unsigned int ca_mix32(unsigned int x) {
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ unsigned int ca_dropout_threshold(float p) { return (unsigned int)(p * 65536.0f); }
__device__ __forceinline__ void ca_dropout_words(uint64_t seed, uint64_t group, unsigned int& w0, unsigned int& w1) {
  const unsigned int s = (unsigned int)seed * 0x9E3779B9u + (unsigned int)(seed >> 32);
  w0 = ca_mix32((unsigned int)group ^ s ^ ((unsigned int)(group >> 32) * 0x85EBCA6Bu));
  w1 = w0 * 0xC2B2AE35u;
  w1 ^= w1 >> 15;
}
__device__ __forceinline__ bool ca_dropout_keep(uint64_t seed, uint64_t idx, float p) {
  unsigned int w0, w1;
  ca_dropout_words(seed, idx >> 2, w0, w1);
  const unsigned int w = (idx & 2) ? w1 : w0;
  return ((w >> (16 * (idx & 1))) & 0xFFFFu) >= ca_dropout_threshold(p);
}
// the same decision for 4 consecutive elements starting at a multiple of 4 (one hash)
__device__ __forceinline__ unsigned int ca_dropout_keep4(uint64_t seed, uint64_t idx4, float p) {
  unsigned int w0, w1;
  ca_dropout_words(seed, idx4 >> 2, w0, w1);
  const unsigned int thr = ca_dropout_threshold(p);
  return ((w0 & 0xFFFFu) >= thr ? 1u : 0u) | ((w0 >> 16) >= thr ? 2u : 0u) | ((w1 & 0xFFFFu) >= thr ? 4u : 0u) |
         ((w1 >> 16) >= thr ? 8u : 0u);
}
// ca_dropout_keep4 with the hash input split by the caller: s_eff = seed word ^ (high group word * 0x85EBCA6B),
// group_lo = low 32 bits of (flat element index / 4) - for callers that walk a range in which the high word is constant
__device__ __forceinline__ unsigned int ca_dropout_keep4_lo(unsigned int s_eff, unsigned int group_lo, unsigned int thr) {
  const unsigned int w0 = ca_mix32(group_lo ^ s_eff);
  unsigned int w1 = w0 * 0xC2B2AE35u;
  w1 ^= w1 >> 15;
  return ((w0 & 0xFFFFu) >= thr ? 1u : 0u) | ((w0 >> 16) >= thr ? 2u : 0u) | ((w1 & 0xFFFFu) >= thr ? 4u : 0u) |
         ((w1 >> 16) >= thr ? 8u : 0u);
}

// out[i] (+)= sum_p partial[p*stride + i]  (defined in norm.hip)
void ca_reduce_partials_launch(const float* partial, int nparts, int64_t stride, int n, float* out,
                               int accumulate, hipStream_t s);

// Key split of the single-query attention forms (rule in attention.hip; shared with decode.hip): the keys of (clip, head)
// item i go to `ns` workgroups for i < tail_start and to `ns_tail` for the others.  Parts of an item are consecutive
// sub-items; sub-items are numbered items-first.
struct CaKeySplit {
  int ns, tail_start, ns_tail;
};
CaKeySplit ca_attn_key_split(int bh, int ncu, int cap);
__host__ __device__ __forceinline__ int ca_key_split_parts(const CaKeySplit& s, int bh) {
  return s.tail_start * s.ns + (bh - s.tail_start) * s.ns_tail;
}
// sub-item `it` -> its item, its place among the item's n parts, the item's first sub-item
__device__ __forceinline__ void ca_key_split_item(const CaKeySplit& s, int it, int& bh, int& sp, int& n, int& part0) {
  const int head = s.tail_start * s.ns;
  if (it < head) {
    n = s.ns;
    bh = it / n;
    part0 = bh * n;
  } else {
    n = s.ns_tail;
    const int t = (it - head) / n;
    bh = s.tail_start + t;
    part0 = head + t * n;
  }
  sp = it - part0;
}

// host stub of the AdamW kernel the trainer runs in the background (defined in misc.hip; for hipFuncGetAttributes)
const void* ca_adamw_background_kernel();
