// Small HBM-bound pieces: SpecAugment/padding masks, grouped-conv regrouping, weight-norm for
// the positional conv, dtype casts / weight reorders, gradient-norm + fused AdamW, embeddings.
#include "common.h"
#include <cstdlib>

// ---- SpecAugment + padding --------------------------------------------------------------------
// $TF/models/wav2vec2/modeling_wav2vec2.py:1272-1316 (masked_spec_embed on time spans, zeros on
// feature spans) and :752-755 (padded frames -> 0).
__global__ __launch_bounds__(256) void mask_frames_kernel(unsigned short* __restrict__ h,
                                                          const uint8_t* __restrict__ tmask,
                                                          const uint8_t* __restrict__ fmask,
                                                          const unsigned short* __restrict__ embed,
                                                          const int32_t* __restrict__ flen, int B,
                                                          int T, int C) {
  const int cch = C >> 3;
  const int64_t total = (int64_t)B * T * cch;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % cch);
    const int64_t bt = i / cch;
    const int t = (int)(bt % T), b = (int)(bt / T);
    const bool tm = tmask && tmask[bt];
    const bool pad = flen && t >= flen[b];
    bool anyf = false;
    if (fmask) {
#pragma unroll
      for (int e = 0; e < 8; ++e) anyf |= fmask[(int64_t)b * C + c8 * 8 + e] != 0;
    }
    if (!tm && !pad && !anyf) continue;
    unsigned short* p = h + bt * C + c8 * 8;
    u16x8_t v = *(u16x8_t*)p;
    if (tm) v = *(const u16x8_t*)(embed + c8 * 8);
    if (fmask) {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (fmask[(int64_t)b * C + c8 * 8 + e]) v[e] = 0;
    }
    if (pad) v = (u16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
    *(u16x8_t*)p = v;
  }
}

extern "C" int ca_mask_frames(void* h, const uint8_t* tmask, const uint8_t* fmask,
                              const void* embed, const int32_t* flen, int32_t B, int32_t T,
                              int32_t C, void* stream) {
  CA_CHECK_ARG(h && B > 0 && T > 0 && C > 0 && (C % 8) == 0, "ca_mask_frames: bad argument");
  CA_CHECK_ARG(!tmask || embed, "ca_mask_frames: tmask needs embed");
  int64_t g = ((int64_t)B * T * (C / 8) + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(mask_frames_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream,
                     (unsigned short*)h, tmask, fmask, (const unsigned short*)embed, flen, B, T, C);
  CA_CHECK_LAUNCH("ca_mask_frames");
  return CA_OK;
}

// ---- x [B,T,G*Cg] -> xg [B,G,T+2*pad,Cg], zero time padding -----------------------------------
__global__ __launch_bounds__(256) void regroup_pad_kernel(const unsigned short* __restrict__ x,
                                                          unsigned short* __restrict__ xg, int B,
                                                          int T, int G, int Cg, int pad) {
  const int cch = Cg >> 3;
  const int Tp = T + 2 * pad;
  const int64_t total = (int64_t)B * G * Tp * cch;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % cch);
    int64_t r = i / cch;
    const int tp = (int)(r % Tp);
    r /= Tp;
    const int g = (int)(r % G), b = (int)(r / G);
    const int t = tp - pad;
    u16x8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (t >= 0 && t < T) v = *(const u16x8_t*)(x + ((int64_t)b * T + t) * (G * Cg) + g * Cg + c8 * 8);
    *(u16x8_t*)(xg + i * 8) = v;
  }
}

extern "C" int ca_regroup_pad(const void* x, void* xg, int32_t B, int32_t T, int32_t G,
                              int32_t Cg, int32_t pad, void* stream) {
  CA_CHECK_ARG(x && xg && B > 0 && T > 0 && G > 0 && Cg > 0 && (Cg % 8) == 0 && pad >= 0,
               "ca_regroup_pad: bad argument");
  int64_t g = ((int64_t)B * G * (T + 2 * pad) * (Cg / 8) + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(regroup_pad_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)x, (unsigned short*)xg, B, T, G, Cg, pad);
  CA_CHECK_LAUNCH("ca_regroup_pad");
  return CA_OK;
}

// ---- positional-conv weight norm (dim=2): w = g * v / ||v||_(0,1) -----------------------------
// $TF/models/wav2vec2/modeling_wav2vec2.py:326-358 (nn.utils.parametrizations.weight_norm).
// v fp32 [d][Cg][K]; per-tap norm over all (co, ci).
#define PCW_SLABS 1024
// partial[slab][K]: sum over the slab's rows of a[row][j]*b[row][j]
__global__ __launch_bounds__(256) void tap_dot_kernel(const float* __restrict__ a,
                                                      const float* __restrict__ b,
                                                      float* __restrict__ partial, int64_t nrows,
                                                      int K, int bmode, int Cg, int G) {
  // bmode 0: b has the same [row][K] layout as a.
  // bmode 1: a = v [d][Cg][K], b = dwf [G][Cg_out][K][Cg_in] (gradient in GEMM layout).
  const int j = threadIdx.x % K;  // K = 128 -> two row lanes per 256-thread block
  const int rl = threadIdx.x / K;
  const int nrl = 256 / K;
  const int64_t per = (nrows + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * per;
  int64_t r1 = r0 + per;
  if (r1 > nrows) r1 = nrows;
  // four independent rows in flight per thread (one dependent 4-byte load per iteration left the kernel at 0.5 TB/s:
  // 222 us for the 118 MB of XLS-R-2B's positional-conv weight, twice per step)
  auto term = [&](int64_t r) -> float {
    const float av = a[r * K + j];
    float bv;
    if (bmode == 0) {
      bv = b[r * K + j];
    } else {
      const int ci = (int)(r % Cg);
      const int64_t co_g = r / Cg;  // global output channel = g*Cg + co
      bv = b[(co_g * K + j) * Cg + ci];
    }
    return av * bv;
  };
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
  int64_t r = r0 + rl;
  for (; r + 3 * nrl < r1; r += 4 * nrl) {
    acc0 += term(r);
    acc1 += term(r + nrl);
    acc2 += term(r + 2 * nrl);
    acc3 += term(r + 3 * nrl);
  }
  for (; r < r1; r += nrl) acc0 += term(r);
  const float acc = (acc0 + acc1) + (acc2 + acc3);
  __shared__ float red[256];
  red[threadIdx.x] = acc;
  __syncthreads();
  if (rl == 0) {
    float t = 0.f;
    for (int q = 0; q < nrl; ++q) t += red[q * K + j];
    partial[(int64_t)blockIdx.x * K + j] = t;
  }
}

// out[j] = (sqrt of) sum_p partial[p][j]: 16 taps x 16 part-lanes per block (256 threads), four independent sums per
// thread, the 16 lanes of a tap added in a fixed order through LDS (one thread walking all the slabs was a chain of
// 1024 dependent loads: 235 us).
__global__ __launch_bounds__(256) void tap_finish_kernel(const float* __restrict__ partial, int nparts, int K,
                                                         float* __restrict__ out, int do_sqrt) {
  __shared__ float red[16][17];
  const int tl = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int j = blockIdx.x * 16 + tl;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (j < K) {
    int p = pl;
    for (; p + 48 < nparts; p += 64) {
      a0 += partial[(int64_t)p * K + j];
      a1 += partial[(int64_t)(p + 16) * K + j];
      a2 += partial[(int64_t)(p + 32) * K + j];
      a3 += partial[(int64_t)(p + 48) * K + j];
    }
    for (; p < nparts; p += 16) a0 += partial[(int64_t)p * K + j];
  }
  red[pl][tl] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (pl == 0 && j < K) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][tl];
    out[j] = do_sqrt ? sqrtf(t) : t;
  }
}

// one block per (row-set): transposes a [R][K] fp32 panel into [K][R] bf16 with scaling.
// forward weights : fixed co_g, R = ci (Cg rows, stride K)        -> wf[co_g][j][ci]
// backward weights: fixed (g, ci), R = co (Cg rows, stride Cg*K)   -> wb[g][ci][K-1-j][co]
__global__ __launch_bounds__(256) void posconv_build_kernel(const float* __restrict__ v,
                                                            const float* __restrict__ gvec,
                                                            const float* __restrict__ norm,
                                                            unsigned short* __restrict__ out,
                                                            int Cg, int K, int mode) {
  extern __shared__ float tile[];  // [Cg][K+1]
  const int blk = blockIdx.x;
  const float* src;
  int64_t rstride;
  if (mode == 0) {
    src = v + (int64_t)blk * Cg * K;
    rstride = K;
  } else {
    const int g = blk / Cg, ci = blk % Cg;
    src = v + ((int64_t)g * Cg * Cg + ci) * K;
    rstride = (int64_t)Cg * K;
  }
  for (int i = threadIdx.x; i < Cg * K; i += 256) {
    const int r = i / K, j = i % K;
    tile[r * (K + 1) + j] = src[r * rstride + j] * (gvec[j] / norm[j]);
  }
  __syncthreads();
  unsigned short* dst = out + (int64_t)blk * K * Cg;
  for (int i = threadIdx.x; i < Cg * K; i += 256) {
    const int jo = i / Cg, r = i % Cg;
    const int j = mode == 0 ? jo : (K - 1 - jo);
    dst[i] = f2bf(tile[r * (K + 1) + j]);
  }
}

extern "C" int ca_posconv_weight(const float* v, const float* g, void* wf, void* wb, float* norm,
                                 float* partial, int32_t d, int32_t Cg, int32_t K, void* stream) {
  CA_CHECK_ARG(v && g && wf && norm && partial, "ca_posconv_weight: null pointer");
  CA_CHECK_ARG(d > 0 && Cg > 0 && (d % Cg) == 0 && K > 0 && K <= 256 && (256 % K) == 0,
               "ca_posconv_weight: K must divide 256");
  hipStream_t s = (hipStream_t)stream;
  const int64_t nrows = (int64_t)d * Cg;
  hipLaunchKernelGGL(tap_dot_kernel, dim3(PCW_SLABS), dim3(256), 0, s, v, v, partial, nrows, K, 0,
                     Cg, d / Cg);
  hipLaunchKernelGGL(tap_finish_kernel, dim3((K + 15) / 16), dim3(256), 0, s, partial, PCW_SLABS, K, norm, 1);
  const size_t lds = (size_t)Cg * (K + 1) * sizeof(float);
  hipLaunchKernelGGL(posconv_build_kernel, dim3(d), dim3(256), lds, s, v, g, norm,
                     (unsigned short*)wf, Cg, K, 0);
  if (wb)
    hipLaunchKernelGGL(posconv_build_kernel, dim3(d), dim3(256), lds, s, v, g, norm,
                       (unsigned short*)wb, Cg, K, 1);
  CA_CHECK_LAUNCH("ca_posconv_weight");
  return CA_OK;
}

// dv[co_g][ci][j] += (g_j/n_j) * (dw - v * dot_j / n_j^2),  dg[j] += dot_j / n_j,
// dot_j = sum_{co,ci} dw*v, with dw[co_g][ci][j] = dwf[co_g][j][ci].
__global__ __launch_bounds__(256) void posconv_wbwd_kernel(const float* __restrict__ dwf,
                                                           const float* __restrict__ v,
                                                           const float* __restrict__ gvec,
                                                           const float* __restrict__ norm,
                                                           const float* __restrict__ dot,
                                                           float* __restrict__ dv, int Cg, int K) {
  extern __shared__ float tile[];  // [K][Cg+1] of dwf for this output channel
  const int co_g = blockIdx.x;
  const float* src = dwf + (int64_t)co_g * K * Cg;
  for (int i = threadIdx.x; i < K * Cg; i += 256) {
    const int j = i / Cg, ci = i % Cg;
    tile[j * (Cg + 1) + ci] = src[i];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < Cg * K; i += 256) {
    const int ci = i / K, j = i % K;
    const float n = norm[j];
    const int64_t idx = ((int64_t)co_g * Cg + ci) * K + j;
    dv[idx] += (gvec[j] / n) * (tile[j * (Cg + 1) + ci] - v[idx] * dot[j] / (n * n));
  }
}
__global__ void posconv_dg_kernel(const float* __restrict__ dot, const float* __restrict__ norm,
                                  float* __restrict__ dg, int K) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < K) dg[j] += dot[j] / norm[j];
}

extern "C" int ca_posconv_weight_bwd(const float* dwf, const float* v, const float* g,
                                     const float* norm, float* dv, float* dg, float* partial,
                                     int32_t d, int32_t Cg, int32_t K, void* stream) {
  CA_CHECK_ARG(dwf && v && g && norm && dv && dg && partial, "ca_posconv_weight_bwd: null pointer");
  CA_CHECK_ARG(K > 0 && K <= 256 && (256 % K) == 0, "ca_posconv_weight_bwd: K must divide 256");
  hipStream_t s = (hipStream_t)stream;
  const int64_t nrows = (int64_t)d * Cg;
  float* dot = partial + (int64_t)PCW_SLABS * K;
  hipLaunchKernelGGL(tap_dot_kernel, dim3(PCW_SLABS), dim3(256), 0, s, v, dwf, partial, nrows, K,
                     1, Cg, d / Cg);
  hipLaunchKernelGGL(tap_finish_kernel, dim3((K + 15) / 16), dim3(256), 0, s, partial, PCW_SLABS, K, dot, 0);
  const size_t lds = (size_t)K * (Cg + 1) * sizeof(float);
  hipLaunchKernelGGL(posconv_wbwd_kernel, dim3(d), dim3(256), lds, s, dwf, v, g, norm, dot, dv,
                     Cg, K);
  hipLaunchKernelGGL(posconv_dg_kernel, dim3(1), dim3(256), 0, s, dot, norm, dg, K);
  CA_CHECK_LAUNCH("ca_posconv_weight_bwd");
  return CA_OK;
}

extern "C" int64_t ca_posconv_partial_floats(int32_t K) { return (int64_t)(PCW_SLABS + 1) * K; }

// ---- casts / reorders -------------------------------------------------------------------------
__global__ void cast_f32_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ y,
                                     int64_t n) {
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    const f32x4_t v = *(const f32x4_t*)(x + i * 4);
    *(u16x4_t*)(y + i * 4) = (u16x4_t){f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (n4 << 2) + threadIdx.x;
    y[i] = f2bf(x[i]);
  }
}
__global__ void cast_bf16_f32_kernel(const unsigned short* __restrict__ x, float* __restrict__ y,
                                     int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    y[i] = bf2f(x[i]);
}
static int ew_grid(int64_t n, int per) {
  int64_t g = (n / per + 255) / 256;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}
extern "C" int ca_cast_f32_bf16(const float* x, void* y, int64_t n, void* stream) {
  CA_CHECK_ARG(x && y && n > 0, "ca_cast_f32_bf16: bad argument");
  CA_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 8) == 0, "ca_cast_f32_bf16: alignment");
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(ew_grid(n, 4)), dim3(256), 0, (hipStream_t)stream,
                     x, (unsigned short*)y, n);
  CA_CHECK_LAUNCH("ca_cast_f32_bf16");
  return CA_OK;
}
extern "C" int ca_cast_bf16_f32(const void* x, float* y, int64_t n, void* stream) {
  CA_CHECK_ARG(x && y && n > 0, "ca_cast_bf16_f32: bad argument");
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(ew_grid(n, 1)), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)x, y, n);
  CA_CHECK_LAUNCH("ca_cast_bf16_f32");
  return CA_OK;
}

// x fp32 [rows][cols] -> y bf16 [cols][rows]; 32x32 LDS tiles
__global__ __launch_bounds__(256) void transpose_f32_bf16_kernel(const float* __restrict__ x,
                                                                 unsigned short* __restrict__ y,
                                                                 int rows, int cols) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int k = ty; k < 32; k += 8) {
    const int r = r0 + k, c = c0 + tx;
    tile[k][tx] = (r < rows && c < cols) ? x[(int64_t)r * cols + c] : 0.f;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int c = c0 + k, r = r0 + tx;
    if (c < cols && r < rows) y[(int64_t)c * rows + r] = f2bf(tile[tx][k]);
  }
}
extern "C" int ca_transpose_f32_bf16(const float* x, void* y, int32_t rows, int32_t cols,
                                     void* stream) {
  CA_CHECK_ARG(x && y && rows > 0 && cols > 0, "ca_transpose_f32_bf16: bad argument");
  hipLaunchKernelGGL(transpose_f32_bf16_kernel, dim3((cols + 31) / 32, (rows + 31) / 32),
                     dim3(256), 0, (hipStream_t)stream, x, (unsigned short*)y, rows, cols);
  CA_CHECK_LAUNCH("ca_transpose_f32_bf16");
  return CA_OK;
}

// w fp32 [Co][Ci][k] -> wr bf16 [Co][k][Ci]
__global__ void conv_w_reorder_kernel(const float* __restrict__ w, unsigned short* __restrict__ wr,
                                      int64_t n, int Ci, int k) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Ci);
    const int64_t r = i / Ci;
    const int j = (int)(r % k);
    const int64_t co = r / k;
    wr[i] = f2bf(w[(co * Ci + ci) * k + j]);
  }
}
extern "C" int ca_conv_weight_reorder(const float* w, void* wr, int32_t Co, int32_t Ci, int32_t k,
                                      void* stream) {
  CA_CHECK_ARG(w && wr && Co > 0 && Ci > 0 && k > 0, "ca_conv_weight_reorder: bad argument");
  const int64_t n = (int64_t)Co * Ci * k;
  hipLaunchKernelGGL(conv_w_reorder_kernel, dim3(ew_grid(n, 1)), dim3(256), 0,
                     (hipStream_t)stream, w, (unsigned short*)wr, n, Ci, k);
  CA_CHECK_LAUNCH("ca_conv_weight_reorder");
  return CA_OK;
}
// dwr fp32 [Co][k][Ci] -> dw fp32 [Co][Ci][k] (+=)
__global__ void conv_w_grad_reorder_kernel(const float* __restrict__ dwr, float* __restrict__ dw,
                                           int64_t n, int Ci, int k) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Ci);
    const int64_t r = i / Ci;
    const int j = (int)(r % k);
    const int64_t co = r / k;
    dw[(co * Ci + ci) * k + j] += dwr[i];
  }
}
extern "C" int ca_conv_weight_grad_reorder(const float* dwr, float* dw, int32_t Co, int32_t Ci,
                                           int32_t k, void* stream) {
  CA_CHECK_ARG(dwr && dw && Co > 0 && Ci > 0 && k > 0, "ca_conv_weight_grad_reorder: bad arg");
  const int64_t n = (int64_t)Co * Ci * k;
  hipLaunchKernelGGL(conv_w_grad_reorder_kernel, dim3(ew_grid(n, 1)), dim3(256), 0,
                     (hipStream_t)stream, dwr, dw, n, Ci, k);
  CA_CHECK_LAUNCH("ca_conv_weight_grad_reorder");
  return CA_OK;
}

// ---- one launch that zeroes a list of ranges -----------------------------------------------------------------
// (what the engines used to do with one torch fill per range: ~70 ATen launches per step at XLS-R-2B)
// ranges: device array of (offset, length) pairs in BYTES, both multiples of 4.  blockIdx.y walks the ranges,
// blockIdx.x strides over one range: a ragged head up to the first 16-byte boundary, 16-byte stores, a ragged tail.
__global__ __launch_bounds__(256) void clear_ranges_kernel(char* __restrict__ base, const int64_t* __restrict__ ranges,
                                                           int nranges) {
  for (int r = blockIdx.y; r < nranges; r += gridDim.y) {
    char* p = base + ranges[2 * r];
    const int64_t nb = ranges[2 * r + 1];
    int64_t head = (16 - ((uintptr_t)p & 15)) & 15;
    head = head < nb ? head : nb;
    const int64_t n16 = (nb - head) >> 4;
    const int64_t tail = nb - head - (n16 << 4);
    if (blockIdx.x == 0) {
      if ((int64_t)threadIdx.x * 4 < head) *(uint32_t*)(p + threadIdx.x * 4) = 0u;
      if ((int64_t)threadIdx.x * 4 < tail) *(uint32_t*)(p + head + (n16 << 4) + threadIdx.x * 4) = 0u;
    }
    uint4* q = (uint4*)(p + head);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256)
      q[i] = make_uint4(0, 0, 0, 0);
  }
}
extern "C" int ca_clear_ranges(void* base, const int64_t* ranges_bytes, int32_t nranges, int64_t max_bytes, void* stream) {
  CA_CHECK_ARG(base && ranges_bytes && nranges > 0 && max_bytes >= 0, "ca_clear_ranges: bad argument");
  int64_t gx = (max_bytes / 16 + 256 * 8 - 1) / (256 * 8);  // ~8 stores per thread
  gx = gx < 1 ? 1 : (gx > 1024 ? 1024 : gx);
  const int gy = nranges < 1024 ? nranges : 1024;
  hipLaunchKernelGGL(clear_ranges_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, (hipStream_t)stream, (char*)base,
                     ranges_bytes, nranges);
  CA_CHECK_LAUNCH("ca_clear_ranges");
  return CA_OK;
}

// ---- gradient norm + fused AdamW ---------------------------------------------------------------
#define SUMSQ_BLOCKS 1024
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n,
                                                    float* __restrict__ partial) {
  __shared__ float red[4];
  float a = 0.f;
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const f32x4_t v = *(const f32x4_t*)(g + i * 4);
    a += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float v = g[(n4 << 2) + threadIdx.x];
    a += v * v;
  }
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void sumsq_finish_kernel(const float* __restrict__ partial,
                                                           int nparts, float* __restrict__ out,
                                                           int accumulate) {
  __shared__ float red[4];
  float a = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) a += partial[i];
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = red[0] + red[1] + red[2] + red[3];
    out[0] = accumulate ? out[0] + t : t;
  }
}
extern "C" int ca_sumsq_f32(const float* g, int64_t n, float* out, int32_t accumulate,
                            float* partial, void* stream) {
  CA_CHECK_ARG(g && out && partial && n > 0, "ca_sumsq_f32: bad argument");
  CA_CHECK_ARG(((uintptr_t)g % 16) == 0, "ca_sumsq_f32: g must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sumsq_kernel, dim3(SUMSQ_BLOCKS), dim3(256), 0, s, g, n, partial);
  hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(256), 0, s, partial, SUMSQ_BLOCKS, out,
                     accumulate);
  CA_CHECK_LAUNCH("ca_sumsq_f32");
  return CA_OK;
}

// out[0] (+)= sum over the listed chunks of g of g^2.  chunks: device array of (offset, length) pairs in floats
// (offsets multiples of 4, lengths any); one block per chunk and pass, partial[chunk] holds its sum.  The trainer lists
// everything of the flat gradient buffer that is NOT a transformer weight matrix this way (those arrive as per-tile
// partials from the weight-gradient GEMMs, CaGemmDesc.c_sumsq).
__global__ __launch_bounds__(256) void sumsq_chunks_kernel(const float* __restrict__ g, const int64_t* __restrict__ chunks,
                                                           int nchunks, float* __restrict__ partial) {
  __shared__ float red[4];
  for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const float* x = g + chunks[2 * c];
    const int64_t n = chunks[2 * c + 1];
    const int64_t n4 = n >> 2;
    float a = 0.f;
    for (int64_t i = threadIdx.x; i < n4; i += 256) {
      const f32x4_t v = *(const f32x4_t*)(x + i * 4);
      a += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (threadIdx.x < (n & 3)) {
      const float v = x[(n4 << 2) + threadIdx.x];
      a += v * v;
    }
    a = wave_sum(a);
    __syncthreads();  // (red is reused by the next chunk)
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) partial[c] = red[0] + red[1] + red[2] + red[3];
  }
}
extern "C" int ca_sumsq_ranges_f32(const float* g, const int64_t* chunks, int32_t nchunks, float* out,
                                   int32_t accumulate, float* partial, void* stream) {
  CA_CHECK_ARG(g && chunks && out && partial && nchunks > 0, "ca_sumsq_ranges_f32: bad argument");
  CA_CHECK_ARG(((uintptr_t)g % 16) == 0, "ca_sumsq_ranges_f32: g must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int grid = nchunks < 2048 ? nchunks : 2048;
  hipLaunchKernelGGL(sumsq_chunks_kernel, dim3(grid), dim3(256), 0, s, g, chunks, nchunks, partial);
  hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(256), 0, s, partial, nchunks, out, accumulate);
  CA_CHECK_LAUNCH("ca_sumsq_ranges_f32");
  return CA_OK;
}
// out[0] (+)= sum x[i] (fixed order: the per-tile partials of CaGemmDesc.c_sumsq)
__global__ __launch_bounds__(256) void sum_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ partial) {
  __shared__ float red[4];
  float a = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) a += x[i];
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
extern "C" int ca_sum_f32(const float* x, int64_t n, float* out, int32_t accumulate, float* partial, void* stream) {
  CA_CHECK_ARG(x && out && partial && n > 0, "ca_sum_f32: bad argument");
  hipStream_t s = (hipStream_t)stream;
  const int grid = (int)((n + 255) / 256 < 256 ? (n + 255) / 256 : 256);
  hipLaunchKernelGGL(sum_kernel, dim3(grid), dim3(256), 0, s, x, n, partial);
  hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(256), 0, s, partial, grid, out, accumulate);
  CA_CHECK_LAUNCH("ca_sum_f32");
  return CA_OK;
}

// torch.optim.AdamW (decoupled decay) with the clip coefficient of clip_grad_norm_ folded in.
// G16: the gradient is a bf16 tensor (ca_adamw_step_g16: weight-matrix gradients kept as the reference's autocast
// produces them) - 2 instead of 4 of the update's 30 bytes per parameter.
template <bool NT, bool G16 = false>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ m,
                                                    float* __restrict__ v,
                                                    const void* __restrict__ gv,
                                                    unsigned short* __restrict__ p16, int64_t n,
                                                    float lr, float b1, float b2, float eps,
                                                    float wd, float bc1, float bc2_sqrt,
                                                    float grad_scale, float max_norm,
                                                    const float* __restrict__ gnorm_sq, int allow_vec) {
  float coef = grad_scale;
  if (gnorm_sq && max_norm > 0.f) {
    const float nrm = sqrtf(gnorm_sq[0]) * grad_scale;
    const float c = max_norm / (nrm + 1e-6f);
    coef *= c < 1.f ? c : 1.f;
  }
  const float step_size = lr / bc1;
  const float decay = 1.f - lr * wd;
  // 16 bytes per lane and array (f32x4; 8 bytes of bf16): the flat buffers and every bucket offset are 32-byte aligned
  const float* g = (const float*)gv;
  const unsigned short* g16 = (const unsigned short*)gv;
  const bool vec = allow_vec && ((((uintptr_t)p | (uintptr_t)m | (uintptr_t)v | (G16 ? 0 : (uintptr_t)gv)) & 15) == 0) &&
                   (!G16 || ((uintptr_t)gv & 7) == 0) && (!p16 || ((uintptr_t)p16 & 7) == 0);
  const int64_t n4 = vec ? (n >> 2) : 0;
  // The 28 bytes per parameter of fp32 state are touched once per step: non-temporal loads and stores keep them from
  // displacing the operand panels of the forward GEMMs this kernel runs beside (trainer.py) from the L2s / Infinity
  // Cache.  The bf16 compute copy is stored with the default policy: the next forward reads it within a millisecond.
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4_t g4;
    if constexpr (G16) {
      const u16x4_t h = NT ? __builtin_nontemporal_load((const u16x4_t*)g16 + i) : ((const u16x4_t*)g16)[i];
      g4 = (f32x4_t){bf2f(h[0]), bf2f(h[1]), bf2f(h[2]), bf2f(h[3])};
    } else {
      g4 = NT ? __builtin_nontemporal_load((const f32x4_t*)g + i) : ((const f32x4_t*)g)[i];
    }
    f32x4_t p4 = NT ? __builtin_nontemporal_load((const f32x4_t*)p + i) : ((const f32x4_t*)p)[i];
    f32x4_t m4 = NT ? __builtin_nontemporal_load((const f32x4_t*)m + i) : ((const f32x4_t*)m)[i];
    f32x4_t v4 = NT ? __builtin_nontemporal_load((const f32x4_t*)v + i) : ((const f32x4_t*)v)[i];
    u16x4_t h4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gi = g4[e] * coef;
      float pi = p4[e] * decay;
      const float mi = b1 * m4[e] + (1.f - b1) * gi;
      const float vi = b2 * v4[e] + (1.f - b2) * gi * gi;
      const float denom = sqrtf(vi) / bc2_sqrt + eps;
      pi -= step_size * mi / denom;
      p4[e] = pi;
      m4[e] = mi;
      v4[e] = vi;
      h4[e] = f2bf(pi);
    }
    if (NT) {
      __builtin_nontemporal_store(p4, (f32x4_t*)p + i);
      __builtin_nontemporal_store(m4, (f32x4_t*)m + i);
      __builtin_nontemporal_store(v4, (f32x4_t*)v + i);
    } else {
      ((f32x4_t*)p)[i] = p4;
      ((f32x4_t*)m)[i] = m4;
      ((f32x4_t*)v)[i] = v4;
    }
    if (p16) ((u16x4_t*)p16)[i] = h4;
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gi = (G16 ? bf2f(g16[i]) : g[i]) * coef;
    float pi = p[i] * decay;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= step_size * mi / denom;
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
    if (p16) p16[i] = f2bf(pi);
  }
}
// max_blocks > 0 caps the grid.  With one 256-thread workgroup per CU (max_blocks = the CU count) a wave of the update
// (60 registers) fits into what the forward GEMMs leave of a SIMD's register file (ca_gemm_kernel_x<0,0>: 2 x 224 of 512,
// ca_gemm_kernel_l: 2 x 160), so the HBM-bound update runs UNDER the next step's forward instead of alternating with it:
// a full-occupancy update holds every register of the CUs, the GEMM's workgroups wait for its waves to drain and the next
// bucket's update for the GEMM's (measured with events on both streams, XLS-R-2B: update 13.0 ms, the forward beside it
// 31.0 instead of 20.9; capped: update 17.0 ms, forward 28.7, step 72.9 -> 70.7 ms).  What remains is the HBM queue
// itself: the forward loses ~8.4 ms to the update's 64.8 GB whatever the update's pace (128 threads per CU: update
// 28 ms, forward 29.5; eight requests in flight per lane instead of four: no better) - tools/archive/dev_opt_timeline.py.
static int adamw_launch(float* p, float* m, float* v, const void* g, bool g16, void* p16, int64_t n, float lr, float beta1,
                        float beta2, float eps, float weight_decay, int32_t step, float grad_scale, float max_norm,
                        const float* gnorm_sq, int32_t max_blocks, void* stream) {
  CA_CHECK_ARG(p && m && v && g && n > 0 && step >= 1 && max_blocks >= 0, "ca_adamw_step: bad argument");
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2 = 1.f - powf(beta2, (float)step);
  // CA_ADAMW_BLOCKS caps the grid of every call (tuning knob)
  static const int cap_env = [] { const char* e = getenv("CA_ADAMW_BLOCKS"); return e ? atoi(e) : 0; }();
  static const int vec = [] { const char* e = getenv("CA_ADAMW_VEC"); return e ? atoi(e) : 1; }();
  static const int nt = [] { const char* e = getenv("CA_ADAMW_NT"); return e ? atoi(e) : 1; }();
  const int cap = cap_env > 0 ? cap_env : max_blocks;
  int grid = ew_grid(n, vec ? 4 : 1);
  if (cap > 0 && grid > cap) grid = cap;
#define ADAMW(NTV, G16V)                                                                                              \
  hipLaunchKernelGGL((adamw_kernel<NTV, G16V>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p, m, v, g,            \
                     (unsigned short*)p16, n, lr, beta1, beta2, eps, weight_decay, bc1, sqrtf(bc2), grad_scale,       \
                     max_norm, gnorm_sq, vec)
  if (nt && g16) ADAMW(true, true);
  else if (nt) ADAMW(true, false);
  else if (g16) ADAMW(false, true);
  else ADAMW(false, false);
#undef ADAMW
  CA_CHECK_LAUNCH("ca_adamw_step");
  return CA_OK;
}
const void* ca_adamw_background_kernel() { return (const void*)adamw_kernel<true, true>; }
extern "C" int ca_adamw_step_ex(float* p, float* m, float* v, const float* g, void* p16, int64_t n,
                                float lr, float beta1, float beta2, float eps, float weight_decay,
                                int32_t step, float grad_scale, float max_norm,
                                const float* gnorm_sq, int32_t max_blocks, void* stream) {
  return adamw_launch(p, m, v, g, false, p16, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, max_norm, gnorm_sq,
                      max_blocks, stream);
}
extern "C" int ca_adamw_step_g16(float* p, float* m, float* v, const void* g_bf16, void* p16, int64_t n,
                                 float lr, float beta1, float beta2, float eps, float weight_decay,
                                 int32_t step, float grad_scale, float max_norm,
                                 const float* gnorm_sq, int32_t max_blocks, void* stream) {
  return adamw_launch(p, m, v, g_bf16, true, p16, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, max_norm,
                      gnorm_sq, max_blocks, stream);
}
extern "C" int ca_adamw_step(float* p, float* m, float* v, const float* g, void* p16, int64_t n,
                             float lr, float beta1, float beta2, float eps, float weight_decay,
                             int32_t step, float grad_scale, float max_norm,
                             const float* gnorm_sq, void* stream) {
  return ca_adamw_step_ex(p, m, v, g, p16, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, max_norm, gnorm_sq, 0,
                          stream);
}

// ---- token + position embedding gather ----------------------------------------------------------
__global__ __launch_bounds__(256) void embed_kernel(const unsigned short* __restrict__ table,
                                                    const unsigned short* __restrict__ pos,
                                                    const int32_t* __restrict__ ids,
                                                    const int32_t* __restrict__ pos_ids,
                                                    unsigned short* __restrict__ y, int64_t rows,
                                                    int C) {
  const int cch = C >> 3;
  const int64_t total = rows * cch;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % cch);
    const int64_t r = i / cch;
    const u16x8_t a = *(const u16x8_t*)(table + (int64_t)ids[r] * C + c8 * 8);
    u16x8_t o;
    if (pos) {
      const u16x8_t b = *(const u16x8_t*)(pos + (int64_t)pos_ids[r] * C + c8 * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = f2bf(bf2f(a[e]) + bf2f(b[e]));
    } else {
      o = a;
    }
    *(u16x8_t*)(y + r * C + c8 * 8) = o;
  }
}
extern "C" int ca_embed_tokens(const void* table, const void* pos, const int32_t* ids,
                               const int32_t* pos_ids, void* y, int64_t rows, int32_t C,
                               void* stream) {
  CA_CHECK_ARG(table && ids && y && rows > 0 && C > 0 && (C % 8) == 0, "ca_embed_tokens: bad arg");
  CA_CHECK_ARG(!pos || pos_ids, "ca_embed_tokens: pos needs pos_ids");
  hipLaunchKernelGGL(embed_kernel, dim3(ew_grid(rows * (C / 8), 1)), dim3(256), 0,
                     (hipStream_t)stream, (const unsigned short*)table, (const unsigned short*)pos,
                     ids, pos_ids, (unsigned short*)y, rows, C);
  CA_CHECK_LAUNCH("ca_embed_tokens");
  return CA_OK;
}

// backward of the embedding gather: dtable[ids[r], :] += dy[r, :], dpos[pos_ids[r], :] += dy[r, :]
// (fp32 atomics: a handful of rows hit the same token; $TF/models/whisper/modeling_whisper.py:676 backward)
__global__ __launch_bounds__(256) void embed_bwd_kernel(const unsigned short* __restrict__ dy,
                                                        const int32_t* __restrict__ ids,
                                                        const int32_t* __restrict__ pos_ids,
                                                        float* __restrict__ dtable, float* __restrict__ dpos,
                                                        int64_t rows, int C) {
  const int cch = C >> 3;
  const int64_t total = rows * cch;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % cch);
    const int64_t r = i / cch;
    const u16x8_t u = *(const u16x8_t*)(dy + r * C + c8 * 8);
    float* t = dtable ? dtable + (int64_t)ids[r] * C + c8 * 8 : nullptr;
    float* p = dpos ? dpos + (int64_t)pos_ids[r] * C + c8 * 8 : nullptr;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = bf2f(u[e]);
      if (t) atomicAdd(t + e, v);
      if (p) atomicAdd(p + e, v);
    }
  }
}
extern "C" int ca_embed_tokens_bwd(const void* dy, const int32_t* ids, const int32_t* pos_ids, float* dtable,
                                   float* dpos, int64_t rows, int32_t C, void* stream) {
  CA_CHECK_ARG(dy && ids && rows > 0 && C > 0 && (C % 8) == 0, "ca_embed_tokens_bwd: bad argument");
  CA_CHECK_ARG(!dpos || pos_ids, "ca_embed_tokens_bwd: dpos needs pos_ids");
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(ew_grid(rows * (C / 8), 1)), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)dy, ids, pos_ids, dtable, dpos, rows, C);
  CA_CHECK_LAUNCH("ca_embed_tokens_bwd");
  return CA_OK;
}
