// LayerNorm (+ optional exact GELU) forward/backward and column sums.  HBM-bound kernels:
// one 64-lane wave per row, 16-byte bf16 vector loads, fp32 statistics via wave shuffles.
// Reference: nn.LayerNorm call sites in $TF/models/wav2vec2/modeling_wav2vec2.py:291-298
// (conv block LN + GELU), :429-434, :611-654, :791.
#include "common.h"
#include <cstdlib>

#define LN_MAXCH 8  // up to 8 chunks of 8 elements per lane -> C <= 4096

// A lane's chunk of 8 consecutive row elements, bf16 (one 16-byte load) or fp32 (two): the conv stack's pre-norm
// tensors and the last conv block's output stay fp32 (ca_layernorm_fwd_ex / ca_layernorm_bwd_ex), as they do under
// the reference's autocast, where nn.LayerNorm and the GELU behind it run in fp32
// ($TF/models/wav2vec2/modeling_wav2vec2.py:291-298,429-434).
template <bool F32>
struct LnRow;
template <>
struct LnRow<false> {
  typedef u16x8_t V;
  static __device__ __forceinline__ V load(const void* base, int64_t idx) {
    return *(const u16x8_t*)((const unsigned short*)base + idx);
  }
  static __device__ __forceinline__ V zero() { return (u16x8_t){0, 0, 0, 0, 0, 0, 0, 0}; }
  static __device__ __forceinline__ float at(const V& v, int e) { return bf2f(v[e]); }
};
typedef __attribute__((ext_vector_type(8))) float f32x8_t;
template <>
struct LnRow<true> {
  typedef f32x8_t V;
  static __device__ __forceinline__ V load(const void* base, int64_t idx) {
    const f32x4_t a = *(const f32x4_t*)((const float*)base + idx), b = *(const f32x4_t*)((const float*)base + idx + 4);
    return (f32x8_t){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  }
  static __device__ __forceinline__ V zero() { return (f32x8_t){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; }
  static __device__ __forceinline__ float at(const V& v, int e) { return v[e]; }
};

template <int NCH, bool XF32 = false, bool YF32 = false>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const void* __restrict__ x,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ beta,
                                                     void* __restrict__ y,
                                                     float* __restrict__ stats, int64_t rows,
                                                     int C, float eps, int act) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int nchunk = C >> 3;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    float v[NCH][8];
    float s = 0.f;
    // short rows (greedy decoding: a handful of rows of d <= 1536): gamma / beta are requested together with the row,
    // so the kernel is one memory round trip instead of two; long-row shapes keep the registers for occupancy (measured
    // again in round 3 at d = 1920: 12.8 us with the early request against 10.0 us without)
    f32x4_t gq[NCH <= 3 ? NCH : 1][2], bq[NCH <= 3 ? NCH : 1][2];
    if (NCH <= 3) {
#pragma unroll
      for (int c = 0; c < (NCH <= 3 ? NCH : 1); ++c) {
        const int ch = lane + c * 64;
        if (ch < nchunk) {
          gq[c][0] = *(const f32x4_t*)(gamma + ch * 8);
          gq[c][1] = *(const f32x4_t*)(gamma + ch * 8 + 4);
          bq[c][0] = *(const f32x4_t*)(beta + ch * 8);
          bq[c][1] = *(const f32x4_t*)(beta + ch * 8 + 4);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + c * 64;
      if (ch < nchunk) {
        const typename LnRow<XF32>::V u = LnRow<XF32>::load(x, row * C + ch * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          v[c][e] = LnRow<XF32>::at(u, e);
          s += v[c][e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[c][e] = 0.f;
      }
    }
    const float mean = wave_sum(s) / (float)C;
    float s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + c * 64;
      if (ch < nchunk) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float dlt = v[c][e] - mean;
          s2 += dlt * dlt;
        }
      }
    }
    const float rstd = rsqrtf(wave_sum(s2) / (float)C + eps);
    if (lane == 0 && stats) {
      stats[row * 2] = mean;
      stats[row * 2 + 1] = rstd;
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + c * 64;
      if (ch < nchunk) {
        f32x4_t g0, g1, b0, b1;
        if (NCH <= 3) {
          g0 = gq[NCH <= 3 ? c : 0][0];
          g1 = gq[NCH <= 3 ? c : 0][1];
          b0 = bq[NCH <= 3 ? c : 0][0];
          b1 = bq[NCH <= 3 ? c : 0][1];
        } else {
          g0 = *(const f32x4_t*)(gamma + ch * 8);
          g1 = *(const f32x4_t*)(gamma + ch * 8 + 4);
          b0 = *(const f32x4_t*)(beta + ch * 8);
          b1 = *(const f32x4_t*)(beta + ch * 8 + 4);
        }
        float o32[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float gm = e < 4 ? g0[e] : g1[e - 4];
          const float bt = e < 4 ? b0[e] : b1[e - 4];
          float u = ln_apply(v[c][e], mean, rstd, gm, bt);
          if (act) u = gelu_erf(u);
          o32[e] = u;
        }
        if (YF32) {
          float* yr = (float*)y + row * C + ch * 8;
          *(f32x4_t*)yr = (f32x4_t){o32[0], o32[1], o32[2], o32[3]};
          *(f32x4_t*)(yr + 4) = (f32x4_t){o32[4], o32[5], o32[6], o32[7]};
        } else {
          u16x8_t o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = f2bf(o32[e]);
          *(u16x8_t*)((unsigned short*)y + row * C + ch * 8) = o;
        }
      }
    }
  }
}

static int ln_grid(int64_t rows) {
  int64_t g = (rows + 3) / 4;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" int ca_layernorm_fwd_ex(const void* x, const float* gamma, const float* beta, void* y,
                                   float* stats, int64_t rows, int32_t C, float eps, int32_t act,
                                   int32_t x_f32, int32_t y_f32, void* stream) {
  CA_CHECK_ARG(x && gamma && beta && y, "ca_layernorm_fwd: null pointer");
  CA_CHECK_ARG(rows > 0 && C > 0 && (C % 8) == 0 && C <= LN_MAXCH * 512,
               "ca_layernorm_fwd: C=%d must be a multiple of 8 and <= %d", C, LN_MAXCH * 512);
  CA_CHECK_ARG((!x_f32 && !y_f32) || C <= 1024, "ca_layernorm_fwd_ex: fp32 rows are served up to C = 1024 (C=%d)", C);
  const int nch = (C / 8 + 63) / 64;
  dim3 grid(ln_grid(rows)), block(256);
  hipStream_t s = (hipStream_t)stream;
#define LN_FWD_(N, XF, YF) \
  hipLaunchKernelGGL((ln_fwd_kernel<N, XF, YF>), grid, block, 0, s, x, gamma, beta, y, stats, rows, C, eps, act)
#define LN_FWD_T(N)                                \
  do {                                             \
    if (x_f32 && y_f32) LN_FWD_(N, true, true);    \
    else if (x_f32) LN_FWD_(N, true, false);       \
    else if (y_f32) LN_FWD_(N, false, true);       \
    else LN_FWD_(N, false, false);                 \
  } while (0)
#define LN_FWD(N) LN_FWD_(N, false, false)
  switch (nch) {
    case 1: LN_FWD_T(1); break;
    case 2: LN_FWD_T(2); break;
    case 3: LN_FWD(3); break;
    case 4: LN_FWD(4); break;
    default: LN_FWD(8); break;
  }
#undef LN_FWD
#undef LN_FWD_T
#undef LN_FWD_
  CA_CHECK_LAUNCH("ca_layernorm_fwd");
  return CA_OK;
}
extern "C" int ca_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y,
                                float* stats, int64_t rows, int32_t C, float eps, int32_t act,
                                void* stream) {
  return ca_layernorm_fwd_ex(x, gamma, beta, y, stats, rows, C, eps, act, 0, 0, stream);
}

// ---- forward with the output also quantised to fp8, one scale per row -------------------------------------------
// The wave that normalises a row holds all of it, so the row's amax and the e4m3 cast cost no extra pass: q = e4m3(y *
// 448 / amax_row) from the bf16-rounded y, row_scale[row] = amax_row / 448 for ca_gemm_fp8's a_row_scale
// (the per-tensor quantiser needs two passes over the activation, which is what the fp8 GEMM saves).
template <int NCH>
__global__ __launch_bounds__(256) void ln_fwd_fp8_kernel(const unsigned short* __restrict__ x,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta,
                                                         unsigned short* __restrict__ y, unsigned int* __restrict__ q,
                                                         float* __restrict__ row_scale, float* __restrict__ stats,
                                                         int64_t rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int nchunk = C >> 3;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    const unsigned short* xr = x + row * C;
    float v[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + c * 64;
      if (ch < nchunk) {
        const u16x8_t u = *(const u16x8_t*)(xr + ch * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          v[c][e] = bf2f(u[e]);
          s += v[c][e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[c][e] = 0.f;
      }
    }
    const float mean = wave_sum(s) / (float)C;
    float s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + c * 64;
      if (ch < nchunk) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float dlt = v[c][e] - mean;
          s2 += dlt * dlt;
        }
      }
    }
    const float rstd = rsqrtf(wave_sum(s2) / (float)C + eps);
    if (lane == 0 && stats) {
      stats[row * 2] = mean;
      stats[row * 2 + 1] = rstd;
    }
    float am = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + c * 64;
      if (ch < nchunk) {
        const f32x4_t g0 = *(const f32x4_t*)(gamma + ch * 8);
        const f32x4_t g1 = *(const f32x4_t*)(gamma + ch * 8 + 4);
        const f32x4_t b0 = *(const f32x4_t*)(beta + ch * 8);
        const f32x4_t b1 = *(const f32x4_t*)(beta + ch * 8 + 4);
        u16x8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          o[e] = f2bf(ln_apply(v[c][e], mean, rstd, e < 4 ? g0[e] : g1[e - 4], e < 4 ? b0[e] : b1[e - 4]));
          v[c][e] = bf2f(o[e]);  // the bf16 value the unquantised path would feed to the GEMM
          am = fmaxf(am, fabsf(v[c][e]));
        }
        if (y) *(u16x8_t*)(y + row * C + ch * 8) = o;
      }
    }
    am = wave_max(am);
    const float scale = am > 0.f ? 448.0f / am : 1.f;
    if (lane == 0) row_scale[row] = am > 0.f ? am / 448.0f : 1.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + c * 64;
      if (ch < nchunk) {
        float t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = fminf(fmaxf(v[c][e] * scale, -448.0f), 448.0f);
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

extern "C" int ca_layernorm_fwd_fp8(const void* x, const float* gamma, const float* beta, void* y, void* q_fp8,
                                    float* row_scale, float* stats, int64_t rows, int32_t C, float eps, void* stream) {
  CA_CHECK_ARG(x && gamma && beta && q_fp8 && row_scale, "ca_layernorm_fwd_fp8: null pointer");
  CA_CHECK_ARG(rows > 0 && C > 0 && (C % 16) == 0 && C <= LN_MAXCH * 512,
               "ca_layernorm_fwd_fp8: C=%d must be a multiple of 16 and <= %d", C, LN_MAXCH * 512);
  const int nch = (C / 8 + 63) / 64;
  dim3 grid(ln_grid(rows)), block(256);
  hipStream_t s = (hipStream_t)stream;
#define LN_FP8(N)                                                                                 \
  hipLaunchKernelGGL((ln_fwd_fp8_kernel<N>), grid, block, 0, s, (const unsigned short*)x, gamma, beta, \
                     (unsigned short*)y, (unsigned int*)q_fp8, row_scale, stats, rows, C, eps)
  switch (nch) {
    case 1: LN_FP8(1); break;
    case 2: LN_FP8(2); break;
    case 3: LN_FP8(3); break;
    case 4: LN_FP8(4); break;
    default: LN_FP8(8); break;
  }
#undef LN_FP8
  CA_CHECK_LAUNCH("ca_layernorm_fwd_fp8");
  return CA_OK;
}

// ---- backward ------------------------------------------------------------------------------
// partial layout: [grid][2][C] (dgamma partials then dbeta partials per block).
// Enough workgroups to keep ~16 MB of row loads in flight (one wave per row, 2 workgroups per CU at C = 1920
// because of the LDS reduction buffer); more only lengthens the partial-sum pass.  Measured at XLS-R-2B:
// 256 -> 29 us, 512 -> 21.5 us, 1024 -> 26.7 us for C = 1920; C = 512 rows prefer 1024.
static int ln_bwd_grid_max(int C) {
  static const int wide = [] { const char* e = getenv("CA_LN_BWD_GRID"); return e ? atoi(e) : 512; }();
  return C >= 1024 ? wide : 1024;
}
static int ln_bwd_grid(int64_t rows, int C) {
  int64_t g = (rows + 3) / 4;
  if (g > ln_bwd_grid_max(C)) g = ln_bwd_grid_max(C);
  if (g < 1) g = 1;
  return (int)g;
}

template <int NCH, bool ACT, bool DPP, bool XF32 = false>
__global__ __launch_bounds__(256, 2) void ln_bwd_kernel(
    const unsigned short* __restrict__ dy, const void* __restrict__ x,
    const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ stats, const unsigned short* __restrict__ dres,
    unsigned short* __restrict__ dx, float* __restrict__ partial, int64_t rows, int C) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* red = (float*)smem_raw;  // [4 waves][2][C]
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int nchunk = C >> 3;
  float dg[NCH][8], db[NCH][8];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int e = 0; e < 8; ++e) dg[c][e] = db[c][e] = 0.f;
  // gamma (and beta for the GELU form) are re-read per row and phase (7.5 KiB at C = 1920: L1 hits) - holding them
  // across the rows costs 32 registers that the row pipeline below needs
  auto load_gamma = [&](int c, float (&g8)[8]) {
    const int ch = lane + c * 64;
    if (ch < nchunk) {
      const f32x4_t g0 = *(const f32x4_t*)(gamma + ch * 8), g1 = *(const f32x4_t*)(gamma + ch * 8 + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) g8[e] = e < 4 ? g0[e] : g1[e - 4];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) g8[e] = 0.f;
    }
  };
  auto load_beta = [&](int c, float (&b8)[8]) {
    const int ch = lane + c * 64;
#pragma unroll
    for (int e = 0; e < 8; ++e) b8[e] = 0.f;
    if (ACT && ch < nchunk) {
      const f32x4_t b0 = *(const f32x4_t*)(beta + ch * 8), b1 = *(const f32x4_t*)(beta + ch * 8 + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) b8[e] = e < 4 ? b0[e] : b1[e - 4];
    }
  };
  // A wave walks its rows with the NEXT row's x / dy / residual-gradient chunks and statistics already requested
  // while it works on the current one (one wave per row and one round trip per phase left the kernel at 47 % of the
  // achievable HBM rate: latency, not bandwidth; with the rows pipelined a smaller grid keeps the same bytes in flight
  // and writes a quarter of the partial sums).
  const int64_t stride = (int64_t)gridDim.x * 4;
  const u16x8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  typedef LnRow<XF32> XR;
  typedef typename XR::V xv_t;
  xv_t cx[NCH];
  u16x8_t cd[NCH], cr[NCH];
  float cmean = 0.f, crstd = 0.f;
  auto request = [&](int64_t row, xv_t (&ux)[NCH], u16x8_t (&ud)[NCH], float& mean, float& rstd) {
    mean = stats[row * 2];
    rstd = stats[row * 2 + 1];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + c * 64;
      ux[c] = XR::zero();
      ud[c] = zero8;
      if (ch < nchunk) {
        ux[c] = XR::load(x, row * C + ch * 8);
        ud[c] = *(const u16x8_t*)(dy + row * C + ch * 8);
      }
    }
  };
  constexpr bool PIPE = NCH <= 4;  // (rows of more than 2048 channels - no model of this path - would spill)
  int64_t row = (int64_t)blockIdx.x * 4 + wave;
  if (PIPE && row < rows) request(row, cx, cd, cmean, crstd);
  while (row < rows) {
    const int64_t nrow = row + stride;
    if (!PIPE) request(row, cx, cd, cmean, crstd);
    // the row's residual gradient: asked for now, used behind the two reductions (it used to be one more exposed
    // round trip after them)
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + c * 64;
      cr[c] = zero8;
      if (dres && ch < nchunk) cr[c] = *(const u16x8_t*)(dres + row * C + ch * 8);
    }
    xv_t nx[PIPE ? NCH : 1];
    u16x8_t nd[PIPE ? NCH : 1];
    float nmean = 0.f, nrstd = 0.f;
    if constexpr (PIPE) {
      if (nrow < rows) request(nrow, nx, nd, nmean, nrstd);
    }
    float du_[ACT ? NCH : 1][8];  // dy through GELU' (the conv form; the plain form re-reads dy from its registers)
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      float g8[8], b8[8];
      load_gamma(c, g8);
      load_beta(c, b8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float h = (XR::at(cx[c], e) - cmean) * crstd;
        float du = bf2f(cd[c][e]);
        if (ACT) {
          du *= dgelu_erf(h * g8[e] + b8[e]);
          du_[ACT ? c : 0][e] = du;
        }
        dg[c][e] += du * h;
        db[c][e] += du;
        const float d = du * g8[e];
        s1 += d;
        s2 += d * h;
      }
    }
    // (chunks beyond the row hold zeros and gamma = 0 there: they add nothing)
    // (two waves per SIMD here: the DPP + readlane reduction of common.h no longer loses to the shuffle butterfly the way
    // it did with eight waves hiding the crossbar latency - 19.6 -> 17.5 us at [3992, 1920]; CA_LN_BWD_DPP=0 selects the butterfly)
    const float m1 = (DPP ? wave_sum_dpp(s1) : wave_sum(s1)) / (float)C;
    const float m2 = (DPP ? wave_sum_dpp(s2) : wave_sum(s2)) / (float)C;
    unsigned short* dxr = dx + row * C;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + c * 64;
      if (ch < nchunk) {
        u16x8_t o;
        float g8[8];
        load_gamma(c, g8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float h = (XR::at(cx[c], e) - cmean) * crstd;  // recomputed: cheaper than 32 more live registers
          const float du = ACT ? du_[ACT ? c : 0][e] : bf2f(cd[c][e]);
          o[e] = f2bf(crstd * (du * g8[e] - m1 - h * m2) + bf2f(cr[c][e]));
        }
        *(u16x8_t*)(dxr + ch * 8) = o;
      }
    }
    if constexpr (PIPE) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        cx[c] = nx[c];
        cd[c] = nd[c];
      }
      cmean = nmean;
      crstd = nrstd;
    }
    row = nrow;
  }
  // block reduction of the parameter gradients over the 4 waves
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = lane + c * 64;
    if (ch < nchunk) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[(wave * 2 + 0) * C + ch * 8 + e] = dg[c][e];
        red[(wave * 2 + 1) * C + ch * 8 + e] = db[c][e];
      }
    }
  }
  __syncthreads();
  float* pout = partial + (int64_t)blockIdx.x * 2 * C;
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    const int which = i / C, ch = i % C;
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) a += red[(w * 2 + which) * C + ch];
    pout[i] = a;
  }
}

// out[i] (+)= sum_p partial[p*stride + i], i < n.  64 columns x 16 partial-lanes per block, four
// independent accumulators per thread: the few hundred partial rows are summed by coalesced,
// parallel loads instead of one serial dependent chain.
__global__ __launch_bounds__(1024) void reduce_partials_kernel(const float* __restrict__ partial,
                                                               int nparts, int64_t stride, int n,
                                                               float* __restrict__ out,
                                                               int accumulate) {
  __shared__ float red[16][64];
  const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + cl;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (i < n) {
    int p = pl;
    for (; p + 48 < nparts; p += 64) {
      a0 += partial[(int64_t)p * stride + i];
      a1 += partial[(int64_t)(p + 16) * stride + i];
      a2 += partial[(int64_t)(p + 32) * stride + i];
      a3 += partial[(int64_t)(p + 48) * stride + i];
    }
    for (; p < nparts; p += 16) a0 += partial[(int64_t)p * stride + i];
  }
  red[pl][cl] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (pl == 0 && i < n) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][cl];
    out[i] = accumulate ? out[i] + t : t;
  }
}

// few parts, long rows (split-K weight gradients): a plain streaming sum, 4 floats per thread, fixed order
__global__ __launch_bounds__(256) void reduce_few_parts_kernel(const float* __restrict__ partial, int nparts,
                                                               int64_t stride, int n, float* __restrict__ out,
                                                               int accumulate) {
  const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= n) return;
  if (i4 + 4 <= n) {
    f32x4_t acc = *(const f32x4_t*)(partial + i4);
    for (int p = 1; p < nparts; ++p) {
      const f32x4_t v = *(const f32x4_t*)(partial + (int64_t)p * stride + i4);
      acc[0] += v[0];
      acc[1] += v[1];
      acc[2] += v[2];
      acc[3] += v[3];
    }
    if (accumulate) {
      const f32x4_t o = *(const f32x4_t*)(out + i4);
      acc[0] += o[0];
      acc[1] += o[1];
      acc[2] += o[2];
      acc[3] += o[3];
    }
    *(f32x4_t*)(out + i4) = acc;
  } else {
    for (int64_t i = i4; i < n; ++i) {
      float a = partial[i];
      for (int p = 1; p < nparts; ++p) a += partial[(int64_t)p * stride + i];
      out[i] = accumulate ? out[i] + a : a;
    }
  }
}

void ca_reduce_partials_launch(const float* partial, int nparts, int64_t stride, int n, float* out,
                               int accumulate, hipStream_t s) {
  const bool aligned = ((uintptr_t)partial % 16) == 0 && ((uintptr_t)out % 16) == 0 && (stride % 4) == 0;
  if (nparts <= 16 && n >= 65536 && aligned) {
    hipLaunchKernelGGL(reduce_few_parts_kernel, dim3((unsigned)(((int64_t)n + 1023) / 1024)), dim3(256), 0, s, partial,
                       nparts, stride, n, out, accumulate);
    return;
  }
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((n + 63) / 64), dim3(1024), 0, s, partial, nparts,
                     stride, n, out, accumulate);
}

extern "C" int64_t ca_layernorm_bwd_partial_floats(int64_t rows, int32_t C) {
  return (int64_t)ln_bwd_grid(rows, C) * 2 * C;
}

extern "C" int ca_layernorm_bwd_ex(const void* dy, const void* x, const float* gamma,
                                   const float* beta, const float* stats, const void* dres,
                                   void* dx, float* dgamma, float* dbeta, float* partial,
                                   int64_t rows, int32_t C, int32_t act, int32_t x_f32, void* stream) {
  CA_CHECK_ARG(dy && x && gamma && stats && dx && partial, "ca_layernorm_bwd: null pointer");
  CA_CHECK_ARG(!x_f32 || C <= 1024, "ca_layernorm_bwd_ex: fp32 rows are served up to C = 1024 (C=%d)", C);
  CA_CHECK_ARG(!act || beta, "ca_layernorm_bwd: act needs beta");
  CA_CHECK_ARG(rows > 0 && C > 0 && (C % 8) == 0 && C <= LN_MAXCH * 512,
               "ca_layernorm_bwd: bad C=%d", C);
  const int nch = (C / 8 + 63) / 64;
  const int g = ln_bwd_grid(rows, C);
  dim3 grid(g), block(256);
  const size_t lds = (size_t)4 * 2 * C * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
#define LN_BWD_(N, A, D)                                                                              \
  do {                                                                                                \
    if (x_f32 && N <= 2)                                                                              \
      hipLaunchKernelGGL((ln_bwd_kernel<(N <= 2 ? N : 1), A, D, true>), grid, block, lds, s,          \
                         (const unsigned short*)dy, x, gamma, beta, stats, (const unsigned short*)dres, \
                         (unsigned short*)dx, partial, rows, C);                                      \
    else                                                                                              \
      hipLaunchKernelGGL((ln_bwd_kernel<N, A, D, false>), grid, block, lds, s,                        \
                         (const unsigned short*)dy, x, gamma, beta, stats, (const unsigned short*)dres, \
                         (unsigned short*)dx, partial, rows, C);                                      \
  } while (0)
  static const int use_dpp = [] { const char* e = getenv("CA_LN_BWD_DPP"); return e ? atoi(e) : 1; }();
#define LN_BWD(N)                \
  do {                           \
    if (act)                     \
      LN_BWD_(N, true, true);    \
    else if (use_dpp)            \
      LN_BWD_(N, false, true);   \
    else                         \
      LN_BWD_(N, false, false);  \
  } while (0)
  switch (nch) {
    case 1: LN_BWD(1); break;
    case 2: LN_BWD(2); break;
    case 3: LN_BWD(3); break;
    case 4: LN_BWD(4); break;
    default: LN_BWD(8); break;
  }
#undef LN_BWD
#undef LN_BWD_
  CA_CHECK_LAUNCH("ca_layernorm_bwd");
  if (dgamma && dbeta == dgamma + C) {  // weight and bias gradients are adjacent in the flat buffer: one launch
    ca_reduce_partials_launch(partial, g, (int64_t)2 * C, 2 * C, dgamma, 1, s);
  } else {
    if (dgamma) ca_reduce_partials_launch(partial, g, (int64_t)2 * C, C, dgamma, 1, s);
    if (dbeta) ca_reduce_partials_launch(partial + C, g, (int64_t)2 * C, C, dbeta, 1, s);
  }
  CA_CHECK_LAUNCH("ca_layernorm_bwd(reduce)");
  return CA_OK;
}
extern "C" int ca_layernorm_bwd(const void* dy, const void* x, const float* gamma,
                                const float* beta, const float* stats, const void* dres,
                                void* dx, float* dgamma, float* dbeta, float* partial,
                                int64_t rows, int32_t C, int32_t act, void* stream) {
  return ca_layernorm_bwd_ex(dy, x, gamma, beta, stats, dres, dx, dgamma, dbeta, partial, rows, C, act, 0, stream);
}

// ---- column sums (bias gradients) ---------------------------------------------------------
// block = 32 column-chunks (256 columns) x 8 row lanes; grid.y slabs of rows.
// (128 until round 4: the conv stack's bias gradients at 64 clips - 200 K rows x 512 channels - ran 1 600 rows per
// workgroup with one load in flight per lane: 173 us for 210 MB)
#define CS_SLAB_MAX 512
static int cs_slabs(int64_t rows) {
  int64_t s = (rows + 63) / 64;
  if (s > CS_SLAB_MAX) s = CS_SLAB_MAX;
  if (s < 1) s = 1;
  return (int)s;
}
__global__ __launch_bounds__(256) void colsum_kernel(const unsigned short* __restrict__ x,
                                                     int64_t ld, int64_t rows, int N,
                                                     const uint8_t* __restrict__ rowmask,
                                                     float* __restrict__ partial) {
  __shared__ float red[8][256 + 8];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int col = (blockIdx.x * 32 + cl) * 8;
  const int nslab = gridDim.y;
  const int64_t per = (rows + nslab - 1) / nslab;
  const int64_t r0 = (int64_t)blockIdx.y * per;
  int64_t r1 = r0 + per;
  if (r1 > rows) r1 = rows;
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (col < N) {
    int64_t r = r0 + rl;
    for (; r + 24 < r1; r += 32) {  // four rows of this lane in flight (added in row order: the same sums as one by one)
      u16x8_t u[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool on = !rowmask || rowmask[r + 8 * k];
        u[k] = on ? *(const u16x8_t*)(x + (r + 8 * k) * ld + col) : (u16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += bf2f(u[k][e]);
    }
    for (; r < r1; r += 8) {
      if (rowmask && !rowmask[r]) continue;
      const u16x8_t u = *(const u16x8_t*)(x + r * ld + col);
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += bf2f(u[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rl][cl * 8 + e] = a[e];
  __syncthreads();
  const int c = threadIdx.x;
  const int gc = blockIdx.x * 256 + c;
  if (gc < N) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) t += red[r][c];
    partial[(int64_t)blockIdx.y * N + gc] = t;
  }
}

extern "C" int64_t ca_colsum_partial_floats(int64_t rows, int32_t N) {
  return (int64_t)cs_slabs(rows) * N;
}

extern "C" int ca_colsum_bf16(const void* x, int64_t ld, int64_t rows, int32_t N,
                              const uint8_t* rowmask, float* out, int32_t accumulate,
                              float* partial, void* stream) {
  CA_CHECK_ARG(x && out && partial, "ca_colsum_bf16: null pointer");
  CA_CHECK_ARG(rows > 0 && N > 0 && (N % 8) == 0 && (ld % 8) == 0,
               "ca_colsum_bf16: N and ld must be multiples of 8");
  const int ns = cs_slabs(rows);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(colsum_kernel, dim3((N + 255) / 256, ns), dim3(256), 0, s,
                     (const unsigned short*)x, ld, rows, N, rowmask, partial);
  ca_reduce_partials_launch(partial, ns, (int64_t)N, N, out, accumulate, s);
  CA_CHECK_LAUNCH("ca_colsum_bf16");
  return CA_OK;
}

// ---- out = dy * gelu'(u) -------------------------------------------------------------------
__global__ void dgelu_mul_kernel(const unsigned short* __restrict__ dy,
                                 const unsigned short* __restrict__ u,
                                 unsigned short* __restrict__ out, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8;
       i += (int64_t)gridDim.x * blockDim.x) {
    const u16x8_t a = *(const u16x8_t*)(dy + i * 8);
    const u16x8_t b = *(const u16x8_t*)(u + i * 8);
    u16x8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(bf2f(a[e]) * dgelu_erf(bf2f(b[e])));
    *(u16x8_t*)(out + i * 8) = o;
  }
}
extern "C" int ca_dgelu_mul(const void* dy, const void* u, void* out, int64_t n, void* stream) {
  CA_CHECK_ARG(dy && u && out && n > 0 && (n % 8) == 0, "ca_dgelu_mul: n must be a multiple of 8");
  int64_t g = (n / 8 + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(dgelu_mul_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)dy, (const unsigned short*)u, (unsigned short*)out,
                     n / 8);
  CA_CHECK_LAUNCH("ca_dgelu_mul");
  return CA_OK;
}

// Hidden-state dropout as an element-wise pass: y = x * keep / (1 - p) over a dense [rows, N] bf16 matrix, with the
// keep decision of element (m, n) taken from (seed, m * N + n) exactly as the GEMM epilogue takes it
// (gemm_epilogue: CA_EPI_RESIDUAL with dropout_p > 0), so the backward regenerates the forward's mask:
// dY = dropout(dH) feeds the data- and weight-gradient GEMMs of the projection whose output was dropped
// ($TF/models/whisper/modeling_whisper.py:398,406,479,493,502,625,763: nn.functional.dropout on the sub-layer output /
// on the embedded inputs).  x == y is allowed (in place).
__global__ void dropout_bf16_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y, int64_t n8,
                                    float p, uint64_t seed) {
  const float ks = 1.f / (1.f - p);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const u16x8_t a = *(const u16x8_t*)(x + i * 8);
    const unsigned int keep = ca_dropout_keep4(seed, (uint64_t)i * 8, p) | (ca_dropout_keep4(seed, (uint64_t)i * 8 + 4, p) << 4);
    u16x8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = ((keep >> e) & 1u) ? f2bf(bf2f(a[e]) * ks) : (unsigned short)0;
    *(u16x8_t*)(y + i * 8) = o;
  }
}
extern "C" int ca_dropout_bf16(const void* x, void* y, int64_t n, float p, uint64_t seed, void* stream) {
  CA_CHECK_ARG(x && y && n > 0 && (n % 8) == 0, "ca_dropout_bf16: n must be a positive multiple of 8");
  CA_CHECK_ARG(p >= 0.f && p < 1.f, "ca_dropout_bf16: bad p");
  int64_t g = (n / 8 + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(dropout_bf16_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x,
                     (unsigned short*)y, n / 8, p, seed);
  CA_CHECK_LAUNCH("ca_dropout_bf16");
  return CA_OK;
}

// Up to CA_REDUCE_MAX independent reductions of the reduce_partials_kernel form in ONE launch: the second stages of a
// layer's backward (fused bias-gradient partials, the d gamma | d beta partials of its two norms) were three launches of
// ~5 us each per layer.  Same arithmetic per reduction as ca_reduce_rows_f32's many-parts form (bit-identical).
struct ReduceMulti {
  CaReduceDesc d[CA_REDUCE_MAX];
  int first[CA_REDUCE_MAX + 1];  // first block of reduction i
  int count;
};
__global__ __launch_bounds__(1024) void reduce_partials_multi_kernel(const ReduceMulti m) {
  __shared__ float red[16][64];
  int which = 0;
#pragma unroll
  for (int i = 1; i < CA_REDUCE_MAX; ++i)
    if (i < m.count && (int)blockIdx.x >= m.first[i]) which = i;
  const CaReduceDesc d = m.d[which];
  const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int i = ((int)blockIdx.x - m.first[which]) * 64 + cl;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (i < d.n) {
    int p = pl;
    for (; p + 48 < d.nparts; p += 64) {
      a0 += d.partial[(int64_t)p * d.stride + i];
      a1 += d.partial[(int64_t)(p + 16) * d.stride + i];
      a2 += d.partial[(int64_t)(p + 32) * d.stride + i];
      a3 += d.partial[(int64_t)(p + 48) * d.stride + i];
    }
    for (; p < d.nparts; p += 16) a0 += d.partial[(int64_t)p * d.stride + i];
  }
  red[pl][cl] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (pl == 0 && i < d.n) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][cl];
    d.out[i] = d.accumulate ? d.out[i] + t : t;
  }
}
extern "C" int ca_reduce_rows_multi(const CaReduceDesc* descs, int32_t count, void* stream) {
  CA_CHECK_ARG(descs && count >= 1 && count <= CA_REDUCE_MAX, "ca_reduce_rows_multi: 1..%d reductions", CA_REDUCE_MAX);
  ReduceMulti m;
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    CA_CHECK_ARG(descs[i].partial && descs[i].out && descs[i].nparts > 0 && descs[i].n > 0 && descs[i].stride >= descs[i].n,
                 "ca_reduce_rows_multi: bad reduction %d", i);
    m.d[i] = descs[i];
    m.first[i] = blocks;
    blocks += (descs[i].n + 63) / 64;
  }
  for (int i = count; i < CA_REDUCE_MAX; ++i) {
    m.d[i] = descs[0];
    m.first[i] = blocks;
  }
  m.first[CA_REDUCE_MAX] = blocks;
  m.count = count;
  hipLaunchKernelGGL(reduce_partials_multi_kernel, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, m);
  CA_CHECK_LAUNCH("ca_reduce_rows_multi");
  return CA_OK;
}

// out[i] (+)= sum_p partial[p*stride + i]: public form of the partial-sum reduction
extern "C" int ca_reduce_rows_f32(const float* partial, int32_t nparts, int64_t stride, int32_t n, float* out,
                                  int32_t accumulate, void* stream) {
  CA_CHECK_ARG(partial && out && nparts > 0 && n > 0 && stride >= n, "ca_reduce_rows_f32: bad argument");
  ca_reduce_partials_launch(partial, nparts, stride, n, out, accumulate, (hipStream_t)stream);
  CA_CHECK_LAUNCH("ca_reduce_rows_f32");
  return CA_OK;
}
