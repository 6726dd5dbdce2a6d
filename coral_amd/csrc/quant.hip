// Per-tensor fp8 (OCP e4m3) quantisation of bf16 tensors for ca_gemm_fp8 (BASELINE.json configs[4]: fp8 weights /
// CDNA4 fp8 MFMA).  Two HBM-bound passes: amax (an order-independent max: deterministic with atomics), then
// q = e4m3(x * 448 / amax) with v_cvt_pk_fp8_f32 (round to nearest even; the clamp makes the saturation explicit).
#include "common.h"

#define FP8_MAX 448.0f

__global__ __launch_bounds__(256) void fp8_amax_kernel(const unsigned short* __restrict__ x, int64_t n8,
                                                       unsigned int* __restrict__ amax_bits) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    const u16x8_t u = *(const u16x8_t*)(x + i * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(bf2f(u[e])));
  }
  m = wave_max(m);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0)  // one atomic per workgroup; non-negative floats order like their bits
    atomicMax(amax_bits, __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
}

__global__ __launch_bounds__(256) void fp8_cast_kernel(const unsigned short* __restrict__ x, int64_t n8,
                                                       const float* __restrict__ amax, unsigned int* __restrict__ q,
                                                       float* __restrict__ inv_scale) {
  const float am = amax[0];
  const float scale = am > 0.f ? FP8_MAX / am : 1.f;
  if (blockIdx.x == 0 && threadIdx.x == 0) inv_scale[0] = am > 0.f ? am / FP8_MAX : 1.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    const u16x8_t u = *(const u16x8_t*)(x + i * 8);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fminf(fmaxf(bf2f(u[e]) * scale, -FP8_MAX), FP8_MAX);
    unsigned int w0 = 0, w1 = 0;
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], w0, false);
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w0, true);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], w1, false);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], w1, true);
    q[i * 2] = w0;
    q[i * 2 + 1] = w1;
  }
}

extern "C" int ca_quantize_fp8(const void* x_bf16, int64_t n, void* q_fp8, float* inv_scale, float* amax_ws,
                               void* stream) {
  CA_CHECK_ARG(x_bf16 && q_fp8 && inv_scale && amax_ws && n > 0 && (n % 8) == 0,
               "ca_quantize_fp8: null pointer or n not a multiple of 8");
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(amax_ws, 0, sizeof(float), s) != hipSuccess) {
    ca_set_error("ca_quantize_fp8: memset failed");
    return CA_ERR_LAUNCH;
  }
  const int64_t n8 = n / 8;
  int64_t g = (n8 + 255) / 256;
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(fp8_amax_kernel, dim3((unsigned)g), dim3(256), 0, s, (const unsigned short*)x_bf16, n8,
                     (unsigned int*)amax_ws);
  hipLaunchKernelGGL(fp8_cast_kernel, dim3((unsigned)g), dim3(256), 0, s, (const unsigned short*)x_bf16, n8, amax_ws,
                     (unsigned int*)q_fp8, inv_scale);
  CA_CHECK_LAUNCH("ca_quantize_fp8");
  return CA_OK;
}

// ---- delayed scaling: one pass per tensor and step -----------------------------------------------------------------
// The two-pass form above reads a weight matrix twice per optimiser step (amax, then the cast) behind a memset: 1.65 ms
// per whisper-large-turbo step for 64 matrices, about what their fp8 GEMMs save.  Weights move by ~1e-4 of their range
// per step, so the scale of step t may come from the amax of step t - 1: ONE pass that quantises with the given scale
// and leaves this step's amax for the next one (order-independent atomic max: the same bits on every run).  Values
// beyond the old amax saturate (clamp to +-448).  ca_fp8_amax_rotate turns all accumulated amax words into the next
// step's scales in one tiny launch.
__global__ __launch_bounds__(256) void fp8_cast_delayed_kernel(const unsigned short* __restrict__ x, int64_t n8,
                                                               const float* __restrict__ scale_p,
                                                               unsigned int* __restrict__ q,
                                                               unsigned int* __restrict__ amax_next) {
  const float scale = scale_p[0];
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    const u16x8_t u = *(const u16x8_t*)(x + i * 8);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float f = bf2f(u[e]);
      m = fmaxf(m, fabsf(f));
      v[e] = fminf(fmaxf(f * scale, -FP8_MAX), FP8_MAX);
    }
    unsigned int w0 = 0, w1 = 0;
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], w0, false);
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w0, true);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], w1, false);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], w1, true);
    q[i * 2] = w0;
    q[i * 2 + 1] = w1;
  }
  m = wave_max(m);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  // (one atomic per workgroup, spread over the tensor's CA_FP8_AMAX_SLOTS words: atomics on ONE address serialise at
  // ~12 ns each - 1024 of them were 13 of this kernel's 16 us)
  if (threadIdx.x == 0)
    atomicMax(amax_next + (blockIdx.x & (CA_FP8_AMAX_SLOTS - 1)),
              __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
}

extern "C" int ca_quantize_fp8_delayed(const void* x_bf16, int64_t n, void* q_fp8, const float* scale,
                                       uint32_t* amax_next, void* stream) {
  CA_CHECK_ARG(x_bf16 && q_fp8 && scale && amax_next && n > 0 && (n % 8) == 0,
               "ca_quantize_fp8_delayed: null pointer or n not a multiple of 8");
  const int64_t n8 = n / 8;
  int64_t g = (n8 + 255) / 256;
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(fp8_cast_delayed_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)x_bf16, n8, scale, (unsigned int*)q_fp8, (unsigned int*)amax_next);
  CA_CHECK_LAUNCH("ca_quantize_fp8_delayed");
  return CA_OK;
}

// ---- one launch per layer: straight + transposed e4m3 copies of up to eight weight matrices -------------------------
// The per-step refresh of a whisper-large-turbo encoder layer was seven launches (four ca_quantize_fp8_delayed, three
// ca_quantize_fp8_transposed: 224 launches and 1.6 ms per step, each matrix read twice).  ca_fp8_refresh_group walks the
// 64 x 64 tiles of all of a layer's matrices in ONE launch: a tile is read once, quantised with the matrix's delayed
// scale, stored straight (16 bytes per thread) and - where asked - transposed through LDS; the matrix's amax
// accumulates as in fp8_cast_delayed_kernel.  Same arithmetic per element as the two kernels above: the copies are
// bit-identical, the amax words hold the same maximum.
struct Fp8GroupArgs {
  CaFp8RefreshTask t[CA_FP8_GROUP_MAX];
  int first[CA_FP8_GROUP_MAX + 1];  // first tile of task i; first[count] = the grid
  int count;
};
__global__ __launch_bounds__(256) void fp8_refresh_group_kernel(const Fp8GroupArgs a) {
  __shared__ unsigned char tile[64][64 + 8];  // tile[c][r]
  __shared__ float red[4];
  int ti = 0;
#pragma unroll
  for (int i = 1; i < CA_FP8_GROUP_MAX; ++i)
    if (i < a.count && (int)blockIdx.x >= a.first[i]) ti = i;
  const CaFp8RefreshTask t = a.t[ti];
  const int rows = t.rows, cols = t.cols;
  const int tcols = (cols + 63) / 64;
  const int lt = (int)blockIdx.x - a.first[ti];
  const int r0 = (lt / tcols) * 64, c0 = (lt % tcols) * 64;
  const unsigned short* x = (const unsigned short*)t.x_bf16;
  unsigned char* q = (unsigned char*)t.q_fp8;
  unsigned char* qt = (unsigned char*)t.q_fp8_t;
  const float scale = t.scale[0];
  float m = 0.f;
  {  // thread -> (row of the tile, 16 consecutive columns)
    const int r = threadIdx.x >> 2, cq = (threadIdx.x & 3) * 16;
    if (r0 + r < rows) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c = cq + 8 * h;
        if (c0 + c < cols) {  // cols % 8 == 0: a chunk is all-or-nothing
          const u16x8_t u = *(const u16x8_t*)(x + (int64_t)(r0 + r) * cols + c0 + c);
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float f = bf2f(u[e]);
            m = fmaxf(m, fabsf(f));
            v[e] = fminf(fmaxf(f * scale, -FP8_MAX), FP8_MAX);
          }
          unsigned int w0 = 0, w1 = 0;
          w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], w0, false);
          w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w0, true);
          w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], w1, false);
          w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], w1, true);
          *(uint2*)(q + (int64_t)(r0 + r) * cols + c0 + c) = make_uint2(w0, w1);
          if (qt) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              tile[c + e][r] = (unsigned char)((w0 >> (8 * e)) & 0xffu);
              tile[c + 4 + e][r] = (unsigned char)((w1 >> (8 * e)) & 0xffu);
            }
          }
        }
      }
    }
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0 && t.amax_next)
    atomicMax(t.amax_next + (lt & (CA_FP8_AMAX_SLOTS - 1)),
              __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
  if (qt) {  // thread -> (column of x = row of q_t, 16 consecutive rows of x)
    const int c = threadIdx.x >> 2, rq = (threadIdx.x & 3) * 16;
    if (c0 + c < cols) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r = rq + 8 * h;
        if (r0 + r + 8 <= rows) {
          *(uint2*)(qt + (int64_t)(c0 + c) * rows + r0 + r) = *(const uint2*)&tile[c][r];
        } else {
          for (int e = 0; e < 8; ++e)
            if (r0 + r + e < rows) qt[(int64_t)(c0 + c) * rows + r0 + r + e] = tile[c][r + e];
        }
      }
    }
  }
}
extern "C" int ca_fp8_refresh_group(const CaFp8RefreshTask* tasks, int32_t count, void* stream) {
  CA_CHECK_ARG(tasks && count > 0 && count <= CA_FP8_GROUP_MAX, "ca_fp8_refresh_group: 1 .. CA_FP8_GROUP_MAX tasks");
  Fp8GroupArgs a;
  int64_t total = 0;
  for (int i = 0; i < count; ++i) {
    const CaFp8RefreshTask& t = tasks[i];
    CA_CHECK_ARG(t.x_bf16 && t.q_fp8 && t.scale && t.rows > 0 && t.cols > 0 && (t.cols % 8) == 0 &&
                     ((uintptr_t)t.x_bf16 % 16) == 0 && ((uintptr_t)t.q_fp8 % 8) == 0,
                 "ca_fp8_refresh_group: cols must be a multiple of 8, x 16-byte and q 8-byte aligned");
    CA_CHECK_ARG(!t.q_fp8_t || ((t.rows % 8) == 0 && ((uintptr_t)t.q_fp8_t % 8) == 0),
                 "ca_fp8_refresh_group: a transposed copy needs rows % 8 == 0 and an 8-byte aligned destination");
    a.t[i] = t;
    a.first[i] = (int)total;
    total += (int64_t)((t.rows + 63) / 64) * ((t.cols + 63) / 64);
    CA_CHECK_ARG(total < (1ll << 31), "ca_fp8_refresh_group: too many tiles");
  }
  for (int i = count; i <= CA_FP8_GROUP_MAX; ++i) a.first[i] = (int)total;
  a.count = count;
  hipLaunchKernelGGL(fp8_refresh_group_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, a);
  CA_CHECK_LAUNCH("ca_fp8_refresh_group");
  return CA_OK;
}

// amax of tensor i = max over its CA_FP8_AMAX_SLOTS words; scale[i] = 448 / (margin * amax), inv_scale[i] = its reciprocal (the dequantisation factor ca_gemm_fp8 takes),
// amax_next[i] = 0 - for every tensor whose amax was measured since the last rotation (a zero word keeps the old scale).
__global__ void fp8_rotate_kernel(unsigned int* __restrict__ amax_next, float* __restrict__ scale,
                                  float* __restrict__ inv_scale, int count, float margin) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  unsigned int* w = amax_next + (int64_t)i * CA_FP8_AMAX_SLOTS;
  float am = 0.f;
  for (int k = 0; k < CA_FP8_AMAX_SLOTS; ++k) am = fmaxf(am, __uint_as_float(w[k]));
  if (am > 0.f) {
    scale[i] = FP8_MAX / (am * margin);
    inv_scale[i] = am * margin / FP8_MAX;
    for (int k = 0; k < CA_FP8_AMAX_SLOTS; ++k) w[k] = 0u;
  }
}
extern "C" int ca_fp8_amax_rotate(uint32_t* amax_next, float* scale, float* inv_scale, int32_t count, float margin,
                                  void* stream) {
  CA_CHECK_ARG(amax_next && scale && inv_scale && count > 0 && margin >= 1.f, "ca_fp8_amax_rotate: bad argument");
  hipLaunchKernelGGL(fp8_rotate_kernel, dim3((count + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     (unsigned int*)amax_next, scale, inv_scale, count, margin);
  CA_CHECK_LAUNCH("ca_fp8_amax_rotate");
  return CA_OK;
}

// ---- data gradients on the fp8 path ---------------------------------------------------------------------------------
// dX = dY W needs (a) dY as e4m3 and (b) W with the contraction index contiguous, i.e. a TRANSPOSED e4m3 copy.
// (a) The gradient entering a sub-layer already goes through one elementwise pass when the sub-layer's output was dropped
// (hidden-state dropout: dY = dropout(dH) with the forward's mask, ca_dropout_bf16); this is that pass as a row-per-wave
// kernel that also leaves the row as e4m3 with its own scale (the wave holds the whole row, as in ca_layernorm_fwd_fp8).
// p = 0: a plain per-row quantiser (y may be NULL).
template <int NCH>
__global__ __launch_bounds__(256) void dropout_rows_fp8_kernel(const unsigned short* __restrict__ x,
                                                               unsigned short* __restrict__ y, unsigned int* __restrict__ q,
                                                               float* __restrict__ row_scale, int64_t rows, int C, float p,
                                                               uint64_t seed) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int nchunk = C >> 3;
  const float ks = p > 0.f ? 1.f / (1.f - p) : 1.f;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    float v[NCH][8];
    float am = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + c * 64;
      if (ch < nchunk) {
        const int64_t i = row * C + ch * 8;
        const u16x8_t u = *(const u16x8_t*)(x + i);
        unsigned int keep = 0xFFu;
        if (p > 0.f) keep = ca_dropout_keep4(seed, (uint64_t)i, p) | (ca_dropout_keep4(seed, (uint64_t)i + 4, p) << 4);
        u16x8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          // (the bf16 value ca_dropout_bf16 would have written: the weight gradient reads that one)
          o[e] = ((keep >> e) & 1u) ? (p > 0.f ? f2bf(bf2f(u[e]) * ks) : u[e]) : (unsigned short)0;
          v[c][e] = bf2f(o[e]);
          am = fmaxf(am, fabsf(v[c][e]));
        }
        if (y) *(u16x8_t*)(y + i) = o;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[c][e] = 0.f;
      }
    }
    am = wave_max(am);
    const float scale = am > 0.f ? FP8_MAX / am : 1.f;
    if (lane == 0) row_scale[row] = am > 0.f ? am / FP8_MAX : 1.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + c * 64;
      if (ch < nchunk) {
        float t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = fminf(fmaxf(v[c][e] * scale, -FP8_MAX), FP8_MAX);
        unsigned int w0 = 0, w1 = 0;
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(t[0], t[1], w0, false);
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(t[2], t[3], w0, true);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(t[4], t[5], w1, false);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(t[6], t[7], w1, true);
        q[(row * C + ch * 8) / 4] = w0;
        q[(row * C + ch * 8) / 4 + 1] = w1;
      }
    }
  }
}

extern "C" int ca_dropout_rows_fp8(const void* x, void* y, void* q_fp8, float* row_scale, int64_t rows, int32_t C, float p,
                                   uint64_t seed, void* stream) {
  CA_CHECK_ARG(x && q_fp8 && row_scale && rows > 0 && C > 0 && (C % 16) == 0 && C <= 4096,
               "ca_dropout_rows_fp8: C=%d must be a multiple of 16 and <= 4096", C);
  CA_CHECK_ARG(p >= 0.f && p < 1.f && (p == 0.f || y != nullptr), "ca_dropout_rows_fp8: bad p (p > 0 needs y)");
  const int nch = (C / 8 + 63) / 64;
  int64_t g = (rows + 3) / 4;
  if (g > 4096) g = 4096;
  hipStream_t s = (hipStream_t)stream;
#define DRF(N)                                                                                                   \
  hipLaunchKernelGGL((dropout_rows_fp8_kernel<N>), dim3((unsigned)g), dim3(256), 0, s, (const unsigned short*)x, \
                     (unsigned short*)y, (unsigned int*)q_fp8, row_scale, rows, C, p, seed)
  switch (nch) {
    case 1: DRF(1); break;
    case 2: DRF(2); break;
    case 3: DRF(3); break;
    case 4: DRF(4); break;
    default: DRF(8); break;
  }
#undef DRF
  CA_CHECK_LAUNCH("ca_dropout_rows_fp8");
  return CA_OK;
}

// (b) q_t[c][r] = e4m3(clamp(x[r][c] * scale[0])): the transposed e4m3 copy of a [rows, cols] bf16 matrix, 64 x 64 tiles
// through LDS (16-byte reads along the rows of x, 8-byte stores along the rows of q_t).  Uses the scale of the matrix's
// own (untransposed) copy: no amax of its own.
__global__ __launch_bounds__(256) void fp8_cast_transposed_kernel(const unsigned short* __restrict__ x, int rows, int cols,
                                                                  const float* __restrict__ scale_p,
                                                                  unsigned char* __restrict__ qt) {
  __shared__ unsigned char tile[64][64 + 8];  // tile[c][r]
  const float scale = scale_p[0];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  {  // thread -> (row of the tile, 16 consecutive columns)
    const int r = threadIdx.x >> 2, cq = (threadIdx.x & 3) * 16;
    if (r0 + r < rows) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c = cq + 8 * h;
        if (c0 + c < cols) {  // cols % 8 == 0: a chunk is all-or-nothing
          const u16x8_t u = *(const u16x8_t*)(x + (int64_t)(r0 + r) * cols + c0 + c);
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            const float a = fminf(fmaxf(bf2f(u[e]) * scale, -FP8_MAX), FP8_MAX);
            const float b = fminf(fmaxf(bf2f(u[e + 1]) * scale, -FP8_MAX), FP8_MAX);
            const unsigned int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0u, false);
            tile[c + e][r] = (unsigned char)(w & 0xffu);
            tile[c + e + 1][r] = (unsigned char)((w >> 8) & 0xffu);
          }
        }
      }
    }
  }
  __syncthreads();
  {  // thread -> (column of x = row of q_t, 16 consecutive rows of x)
    const int c = threadIdx.x >> 2, rq = (threadIdx.x & 3) * 16;
    if (c0 + c < cols) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r = rq + 8 * h;
        if (r0 + r + 8 <= rows) {
          *(uint2*)(qt + (int64_t)(c0 + c) * rows + r0 + r) = *(const uint2*)&tile[c][r];
        } else {
          for (int e = 0; e < 8; ++e)
            if (r0 + r + e < rows) qt[(int64_t)(c0 + c) * rows + r0 + r + e] = tile[c][r + e];
        }
      }
    }
  }
}
extern "C" int ca_quantize_fp8_transposed(const void* x_bf16, int32_t rows, int32_t cols, void* q_fp8_t, const float* scale,
                                          void* stream) {
  CA_CHECK_ARG(x_bf16 && q_fp8_t && scale && rows > 0 && cols > 0 && (rows % 8) == 0 && (cols % 8) == 0 &&
                   ((uintptr_t)q_fp8_t % 8) == 0,
               "ca_quantize_fp8_transposed: rows and cols must be multiples of 8");
  hipLaunchKernelGGL(fp8_cast_transposed_kernel, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)x_bf16, rows, cols, scale, (unsigned char*)q_fp8_t);
  CA_CHECK_LAUNCH("ca_quantize_fp8_transposed");
  return CA_OK;
}
