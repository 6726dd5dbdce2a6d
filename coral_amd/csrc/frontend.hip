// Waveform front end of the wav2vec2 path: utterance normalisation, fused feature-encoder
// layer 0 (Conv1d(1,512,k=10,s=5) + LayerNorm + GELU) forward/backward, col2im for the
// strided conv data-gradients.  All HBM-bound; coalesced fp32 PCM reads, 1-KiB row stores.
#include "common.h"

// ---- (x - mean) / sqrt(var + eps) over the valid samples; padding -> 0 ---------------------
// $TF/models/wav2vec2/feature_extraction_wav2vec2.py:77-97.  One 1024-thread block per
// utterance, three passes (the utterance is 640 KB: passes 2 and 3 hit L2).
__global__ __launch_bounds__(1024) void wave_normalize_kernel(const float* __restrict__ x,
                                                              const int32_t* __restrict__ lengths,
                                                              float* __restrict__ y, int64_t N,
                                                              float eps) {
  __shared__ float red[16];
  __shared__ float bc;
  const int b = blockIdx.x;
  const float* xb = x + (int64_t)b * N;
  float* yb = y + (int64_t)b * N;
  int64_t len = lengths ? lengths[b] : N;
  if (len > N) len = N;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < len; i += 1024) s += xb[i];
  s = wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += red[w];
    bc = len > 0 ? t / (float)len : 0.f;
  }
  __syncthreads();
  const float mean = bc;
  float s2 = 0.f;
  for (int64_t i = threadIdx.x; i < len; i += 1024) {
    const float d = xb[i] - mean;
    s2 += d * d;
  }
  s2 = wave_sum(s2);
  __syncthreads();
  if (lane == 0) red[wave] = s2;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += red[w];
    bc = len > 0 ? rsqrtf(t / (float)len + eps) : 0.f;
  }
  __syncthreads();
  const float rstd = bc;
  for (int64_t i = threadIdx.x; i < N; i += 1024) yb[i] = i < len ? (xb[i] - mean) * rstd : 0.f;
}

extern "C" int ca_wave_normalize(const float* x, const int32_t* lengths, float* y, int32_t B,
                                 int64_t N, float eps, void* stream) {
  CA_CHECK_ARG(x && y && B > 0 && N > 0, "ca_wave_normalize: bad argument");
  hipLaunchKernelGGL(wave_normalize_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, x,
                     lengths, y, N, eps);
  CA_CHECK_LAUNCH("ca_wave_normalize");
  return CA_OK;
}

// ---- attention mask -> frames per utterance -----------------------------------------------------
// `_get_feat_extract_output_lengths(attention_mask.sum(-1))` ($TF/models/wav2vec2/modeling_wav2vec2.py:1093-1108):
// the row sum of the int mask pushed through the conv stack's floor((n - k) / s) + 1, one launch instead of the two
// dozen tiny elementwise kernels the same arithmetic takes as tensor ops.
struct ConvStack {
  int32_t n, k[8], s[8];
};
// Two launches: (slices x B) workgroups add up their slice of the mask with 16-byte loads into one counter per utterance
// (integer atomics: the sum does not depend on their order), then one workgroup turns the counts into frame lengths and
// clears the counters for the next call.  (One workgroup per utterance, as before, took 0.41 ms for 8 x 160 000 samples:
// eight workgroups cannot keep enough loads in flight.)
#define FL_MAXB 4096
__device__ int g_fl_count[FL_MAXB];
__global__ __launch_bounds__(256) void frame_count_kernel(const int32_t* __restrict__ mask, int64_t N) {
  __shared__ int red[4];
  const int b = blockIdx.y;
  const int32_t* mb = mask + (int64_t)b * N;
  const int64_t per = ((N + gridDim.x - 1) / gridDim.x + 3) & ~(int64_t)3;
  const int64_t lo = (int64_t)blockIdx.x * per;
  int64_t hi = lo + per;
  if (hi > N) hi = N;
  int s = 0;
  const bool vec = (((uintptr_t)mb) & 15) == 0;  // rows of N % 4 == 0 ints from an aligned base
  if (vec) {
    typedef __attribute__((ext_vector_type(4))) int i32x4_t;
    const int64_t n4 = hi > lo ? (hi - lo) >> 2 : 0;
    for (int64_t i = threadIdx.x; i < n4; i += 256) {
      const i32x4_t v = *(const i32x4_t*)(mb + lo + 4 * i);
      s += v[0] + v[1] + v[2] + v[3];
    }
    for (int64_t i = lo + 4 * n4 + threadIdx.x; i < hi; i += 256) s += mb[i];
  } else {
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) s += mb[i];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&g_fl_count[b], red[0] + red[1] + red[2] + red[3]);
}
__global__ void frame_lengths_finish_kernel(int B, ConvStack cs, int32_t* __restrict__ out) {
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    int64_t n = g_fl_count[b];
    g_fl_count[b] = 0;
    for (int i = 0; i < cs.n; ++i) {
      const int64_t d = n - cs.k[i];
      const int64_t q = d >= 0 ? d / cs.s[i] : -((-d + cs.s[i] - 1) / cs.s[i]);  // floor division
      n = q + 1;
    }
    out[b] = (int32_t)n;
  }
}
extern "C" int ca_frame_lengths(const int32_t* attention_mask, int32_t B, int64_t N, const int32_t* kernels,
                                const int32_t* strides, int32_t nconv, int32_t* out, void* stream) {
  CA_CHECK_ARG(attention_mask && out && kernels && strides && B > 0 && B <= FL_MAXB && N > 0 && nconv >= 0 && nconv <= 8,
               "ca_frame_lengths: bad argument");
  ConvStack cs;
  cs.n = nconv;
  for (int i = 0; i < nconv; ++i) {
    CA_CHECK_ARG(strides[i] > 0, "ca_frame_lengths: stride must be positive");
    cs.k[i] = kernels[i];
    cs.s[i] = strides[i];
  }
  int slices = (int)((N + 8191) / 8192);
  if (slices > 64) slices = 64;
  if (((N & 3) != 0) && B > 1) slices = slices;  // (unaligned rows take the scalar path inside the kernel)
  hipLaunchKernelGGL(frame_count_kernel, dim3(slices, B), dim3(256), 0, (hipStream_t)stream, attention_mask, N);
  hipLaunchKernelGGL(frame_lengths_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, B, cs, out);
  CA_CHECK_LAUNCH("ca_frame_lengths");
  return CA_OK;
}

// ---- raw PCM -> model input (row N1 of SURVEY.md §8f) -------------------------------------------
// One workgroup per utterance: int16 (or fp32) samples -> optional peak normalisation (x / max|x|,
// `ta.PeakNormalization`, R/src/coral/data.py:710) -> zero-mean / unit-variance over the valid samples
// ($TF/models/wav2vec2/feature_extraction_wav2vec2.py:77-97) -> padding 0 + attention mask
// (:99-236 as used by R/src/coral/data_collators.py:72-77).  Three sweeps over the utterance (max and
// sum, centred sum of squares, write), all from L2 after the first.
template <typename T>
__device__ __forceinline__ float pcm_load(const T* p, int64_t i);
template <>
__device__ __forceinline__ float pcm_load<float>(const float* p, int64_t i) { return p[i]; }
template <>
__device__ __forceinline__ float pcm_load<int16_t>(const int16_t* p, int64_t i) { return (float)p[i] * (1.0f / 32768.0f); }

template <typename T>
__global__ __launch_bounds__(1024) void pcm_prepare_kernel(const T* __restrict__ pcm, int64_t ld_in,
                                                           const int32_t* __restrict__ lengths, float* __restrict__ y,
                                                           int32_t* __restrict__ mask, int64_t N, int peak, int znorm,
                                                           float eps) {
  __shared__ float red[2][16];
  __shared__ float bc[2];
  const int b = blockIdx.x;
  const T* xb = pcm + (int64_t)b * ld_in;
  float* yb = y + (int64_t)b * N;
  int64_t len = lengths ? lengths[b] : N;
  if (len > N) len = N;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float s = 0.f, mx = 0.f;
  for (int64_t i = threadIdx.x; i < len; i += 1024) {
    const float v = pcm_load<T>(xb, i);
    s += v;
    mx = fmaxf(mx, fabsf(v));
  }
  s = wave_sum(s);
  mx = wave_max(mx);
  if (lane == 0) {
    red[0][wave] = s;
    red[1][wave] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f, m = 0.f;
    for (int w = 0; w < 16; ++w) {
      t += red[0][w];
      m = fmaxf(m, red[1][w]);
    }
    const float g = (peak && m > 0.f) ? 1.0f / m : 1.0f;  // gain of the peak normalisation
    bc[0] = len > 0 ? t / (float)len * g : 0.f;            // mean of the gained signal
    bc[1] = g;
  }
  __syncthreads();
  const float mean = bc[0], gain = bc[1];
  float rstd = 1.f, shift = 0.f;
  if (znorm) {
    float s2 = 0.f;
    for (int64_t i = threadIdx.x; i < len; i += 1024) {
      const float d = pcm_load<T>(xb, i) * gain - mean;
      s2 += d * d;
    }
    s2 = wave_sum(s2);
    __syncthreads();
    if (lane == 0) red[0][wave] = s2;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int w = 0; w < 16; ++w) t += red[0][w];
      bc[0] = len > 0 ? rsqrtf(t / (float)len + eps) : 0.f;
    }
    __syncthreads();
    rstd = bc[0];
    shift = mean;
  }
  for (int64_t i = threadIdx.x; i < N; i += 1024) {
    yb[i] = i < len ? (pcm_load<T>(xb, i) * gain - shift) * rstd : 0.f;
    if (mask) mask[(int64_t)b * N + i] = i < len ? 1 : 0;
  }
}

extern "C" int ca_pcm_prepare(const void* pcm, int32_t is_int16, int64_t ld_in, const int32_t* lengths, float* y,
                              int32_t* mask, int32_t B, int64_t N, int32_t peak_normalize, int32_t zero_mean_unit_var,
                              float eps, void* stream) {
  CA_CHECK_ARG(pcm && y && B > 0 && N > 0 && ld_in >= 0, "ca_pcm_prepare: bad argument");
  if (is_int16)
    hipLaunchKernelGGL(pcm_prepare_kernel<int16_t>, dim3(B), dim3(1024), 0, (hipStream_t)stream, (const int16_t*)pcm,
                       ld_in, lengths, y, mask, N, peak_normalize, zero_mean_unit_var, eps);
  else
    hipLaunchKernelGGL(pcm_prepare_kernel<float>, dim3(B), dim3(1024), 0, (hipStream_t)stream, (const float*)pcm, ld_in,
                       lengths, y, mask, N, peak_normalize, zero_mean_unit_var, eps);
  CA_CHECK_LAUNCH("ca_pcm_prepare");
  return CA_OK;
}

// ---- feature-encoder layer 0, fused ---------------------------------------------------------
// One wave per output frame: lane owns 8 consecutive channels (C = 512), the k input samples
// are wave-uniform.  LN statistics by wave shuffles; the frame is stored as one 1-KiB row.
#define C0 512
#define K0MAX 16

// Forward.  A workgroup takes CONV0_FPB consecutive frames of one utterance: their input samples (FPB * stride + k
// floats) are staged in LDS once and every frame's taps are wave-uniform LDS reads afterwards (a global load per tap and
// frame left each wave waiting on memory: 881 us for 8 x 160 000 samples against ~55 us of HBM time for the 262 MB it
// writes).  LayerNorm over the channels needs no mean pass: with the weights and the bias centred over the channels
// (w~ = w - mean_c w, b~ = b - mean_c b, once per wave) the convolution yields v - mean directly, and one cross-lane
// reduction (the variance) remains.  Four frames per iteration and wave keep four independent dependency chains in flight; 128 frames per workgroup
// amortise the per-wave weight loads and centring.
#define CONV0_FPB 128
template <int KW>
__global__ __launch_bounds__(256) void conv0_fwd_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ bias,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        unsigned short* __restrict__ y, int B,
                                                        int64_t N, int64_t T0, int stride,
                                                        float eps) {
  extern __shared__ float xs_lds[];  // CONV0_FPB * stride + KW samples
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t b = blockIdx.y;
  const int64_t t0 = (int64_t)blockIdx.x * CONV0_FPB;
  int nf = (int)(T0 - t0 < CONV0_FPB ? T0 - t0 : CONV0_FPB);
  const int nx = (nf - 1) * stride + KW;
  const float* xb = x + b * N + t0 * stride;
  for (int i = threadIdx.x; i < nx; i += 256) xs_lds[i] = xb[i];
  float wc[8][KW], bc[8], gm[8], bt[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = lane * 8 + e;
#pragma unroll
    for (int j = 0; j < KW; ++j) wc[e][j] = w[c * KW + j];
    bc[e] = bias[c];
    gm[e] = gamma[c];
    bt[e] = beta[c];
  }
  {  // centre over the 512 channels
    float sb = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) sb += bc[e];
    sb = wave_sum_dpp(sb) * (1.f / C0);
#pragma unroll
    for (int e = 0; e < 8; ++e) bc[e] -= sb;
#pragma unroll
    for (int j = 0; j < KW; ++j) {
      float sw = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) sw += wc[e][j];
      sw = wave_sum_dpp(sw) * (1.f / C0);
#pragma unroll
      for (int e = 0; e < 8; ++e) wc[e][j] -= sw;
    }
  }
  __syncthreads();
  unsigned short* yb = y + (b * T0 + t0) * C0 + lane * 8;
  constexpr int NF = 4;  // frames per iteration and wave: NF independent dependency chains
  for (int f0 = wave; f0 < nf; f0 += 4 * NF) {
    float v[NF][8];
    const float* xp[NF];
#pragma unroll
    for (int q = 0; q < NF; ++q) {
      const int f = f0 + 4 * q;
      xp[q] = xs_lds + (f < nf ? f : f0) * stride;  // frames beyond the block recompute the first one (not stored)
#pragma unroll
      for (int e = 0; e < 8; ++e) v[q][e] = bc[e];
    }
#pragma unroll
    for (int j = 0; j < KW; ++j) {
      float px[NF];
#pragma unroll
      for (int q = 0; q < NF; ++q) px[q] = xp[q][j];
#pragma unroll
      for (int q = 0; q < NF; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[q][e] = fmaf(wc[e][j], px[q], v[q][e]);
    }
    float s2[NF];
#pragma unroll
    for (int q = 0; q < NF; ++q) {
      s2[q] = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) s2[q] = fmaf(v[q][e], v[q][e], s2[q]);
    }
    float rs[NF];
#pragma unroll
    for (int q = 0; q < NF; ++q) rs[q] = rsqrtf(wave_sum_dpp(s2[q]) * (1.f / C0) + eps);
#pragma unroll
    for (int q = 0; q < NF; ++q) {
      u16x8_t o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = f2bf(gelu_erf(v[q][e] * rs[q] * gm[e] + bt[e]));
      const int f = f0 + 4 * q;
      if (f < nf) *(u16x8_t*)(yb + (int64_t)f * C0) = o;  // wave-uniform
    }
  }
}

extern "C" int ca_conv0_ln_gelu_fwd(const float* x, const float* w, const float* bias,
                                    const float* gamma, const float* beta, void* y, int32_t B,
                                    int64_t N, int32_t C, int32_t k, int32_t stride, float eps,
                                    void* stream) {
  CA_CHECK_ARG(x && w && bias && gamma && beta && y, "ca_conv0_ln_gelu_fwd: null pointer");
  CA_CHECK_ARG(C == C0, "ca_conv0_ln_gelu_fwd: C must be %d (got %d)", C0, C);
  CA_CHECK_ARG(k == 10 && stride >= 1 && stride <= 16 && N >= k, "ca_conv0_ln_gelu_fwd: k must be 10, stride 1..16");
  CA_CHECK_ARG(B > 0 && B <= 65535, "ca_conv0_ln_gelu_fwd: bad batch");
  const int64_t T0 = (N - k) / stride + 1;
  const unsigned gx = (unsigned)((T0 + CONV0_FPB - 1) / CONV0_FPB);
  const size_t lds = (size_t)(CONV0_FPB * stride + k) * sizeof(float);
  hipLaunchKernelGGL((conv0_fwd_kernel<10>), dim3(gx, (unsigned)B), dim3(256), lds, (hipStream_t)stream, x,
                     w, bias, gamma, beta, (unsigned short*)y, B, N, T0, stride, eps);
  CA_CHECK_LAUNCH("ca_conv0_ln_gelu_fwd");
  return CA_OK;
}

// backward: recompute conv + LN from x; accumulate dw[C][k], dbias, dgamma, dbeta per wave in
// registers, reduce over the block's waves in LDS, one partial row per block:
// partial[blk][C*(k+3)] laid out as [dw (C*k) | dbias (C) | dgamma (C) | dbeta (C)].
// Same structure as the forward: blocks of CONV0_FPB frames with their samples staged in LDS, centred weights (no mean
// pass), two frames per iteration, and the next iteration's dy rows requested before the current ones are used.
#define CONV0_BWD_GRID 512
template <int KW>
__global__ __launch_bounds__(256) void conv0_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ gamma, const float* __restrict__ beta,
    const unsigned short* __restrict__ dy, float* __restrict__ partial, int B, int64_t N,
    int64_t T0, int stride, float eps) {
  extern __shared__ float smem_f[];
  float* xs_lds = smem_f;                                          // CONV0_FPB * stride + KW samples
  float(*red)[C0] = (float(*)[C0])(smem_f + CONV0_FPB * 16 + 16);  // [4][C0] cross-wave reduction (stride <= 16)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float wc[8][KW], bc[8], gm[8], bt[8];
  float dw[8][KW], dbs[8], dgm[8], dbt[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = lane * 8 + e;
#pragma unroll
    for (int j = 0; j < KW; ++j) {
      wc[e][j] = w[c * KW + j];
      dw[e][j] = 0.f;
    }
    bc[e] = bias[c];
    gm[e] = gamma[c];
    bt[e] = beta[c];
    dbs[e] = dgm[e] = dbt[e] = 0.f;
  }
  {  // centre weights and bias over the 512 channels: the convolution then yields v - mean
    float sb = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) sb += bc[e];
    sb = wave_sum_dpp(sb) * (1.f / C0);
#pragma unroll
    for (int e = 0; e < 8; ++e) bc[e] -= sb;
#pragma unroll
    for (int j = 0; j < KW; ++j) {
      float sw = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) sw += wc[e][j];
      sw = wave_sum_dpp(sw) * (1.f / C0);
#pragma unroll
      for (int e = 0; e < 8; ++e) wc[e][j] -= sw;
    }
  }
  const int64_t bpu = (T0 + CONV0_FPB - 1) / CONV0_FPB;  // frame blocks per utterance
  const int64_t nblk = bpu * B;
  for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int64_t b = blk / bpu, t0 = (blk % bpu) * CONV0_FPB;
    const int nf = (int)(T0 - t0 < CONV0_FPB ? T0 - t0 : CONV0_FPB);
    const int nx = (nf - 1) * stride + KW;
    const float* xb = x + b * N + t0 * stride;
    __syncthreads();  // the previous block's reads of xs_lds are done
    for (int i = threadIdx.x; i < nx; i += 256) xs_lds[i] = xb[i];
    __syncthreads();
    const unsigned short* dyb = dy + (b * T0 + t0) * C0 + lane * 8;
    constexpr int NF = 2;
    u16x8_t ud[NF], un[NF];
#pragma unroll
    for (int q = 0; q < NF; ++q) {
      const int f = wave + 4 * q;
      ud[q] = *(const u16x8_t*)(dyb + (int64_t)(f < nf ? f : 0) * C0);
    }
    for (int f0 = wave; f0 < nf; f0 += 4 * NF) {
#pragma unroll
      for (int q = 0; q < NF; ++q) {  // next iteration's rows (clamped: unused when beyond the block)
        const int f = f0 + 4 * NF + 4 * q;
        un[q] = *(const u16x8_t*)(dyb + (int64_t)(f < nf ? f : 0) * C0);
      }
      float v[NF][8], px[NF][KW];
#pragma unroll
      for (int q = 0; q < NF; ++q) {
        const int f = f0 + 4 * q;
        const float* xp = xs_lds + (f < nf ? f : f0) * stride;
#pragma unroll
        for (int j = 0; j < KW; ++j) px[q][j] = xp[j];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[q][e] = bc[e];
      }
#pragma unroll
      for (int j = 0; j < KW; ++j)
#pragma unroll
        for (int q = 0; q < NF; ++q)
#pragma unroll
          for (int e = 0; e < 8; ++e) v[q][e] = fmaf(wc[e][j], px[q][j], v[q][e]);
      float rstd[NF];
#pragma unroll
      for (int q = 0; q < NF; ++q) {
        float s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s2 = fmaf(v[q][e], v[q][e], s2);
        rstd[q] = rsqrtf(wave_sum_dpp(s2) * (1.f / C0) + eps);
      }
#pragma unroll
      for (int q = 0; q < NF; ++q) {
        const bool live = f0 + 4 * q < nf;  // wave-uniform
        float h[8], dh[8], a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          h[e] = v[q][e] * rstd[q];
          const float du = live ? bf2f(ud[q][e]) * dgelu_erf(h[e] * gm[e] + bt[e]) : 0.f;
          dgm[e] += du * h[e];
          dbt[e] += du;
          dh[e] = du * gm[e];
          a1 += dh[e];
          a2 += dh[e] * h[e];
        }
        const float m1 = wave_sum_dpp(a1) * (1.f / C0), m2 = wave_sum_dpp(a2) * (1.f / C0);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float dv = rstd[q] * (dh[e] - m1 - h[e] * m2);
          dbs[e] += dv;
#pragma unroll
          for (int j = 0; j < KW; ++j) dw[e][j] = fmaf(dv, px[q][j], dw[e][j]);
        }
      }
#pragma unroll
      for (int q = 0; q < NF; ++q) ud[q] = un[q];
    }
  }
  float* pout = partial + (int64_t)blockIdx.x * (C0 * (KW + 3));
  // reduce each of the KW+3 channel vectors over the 4 waves through LDS
  __syncthreads();
  for (int q = 0; q < KW + 3; ++q) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float val;
      if (q < KW) {
        val = 0.f;
#pragma unroll
        for (int j = 0; j < KW; ++j)
          if (j == q) val = dw[e][j];
      } else if (q == KW) {
        val = dbs[e];
      } else if (q == KW + 1) {
        val = dgm[e];
      } else {
        val = dbt[e];
      }
      red[wave][lane * 8 + e] = val;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C0; c += 256) {
      const float t = red[0][c] + red[1][c] + red[2][c] + red[3][c];
      if (q < KW)
        pout[c * KW + q] = t;
      else
        pout[C0 * KW + (q - KW) * C0 + c] = t;
    }
    __syncthreads();
  }
}

static int conv0_bwd_grid(int32_t B, int64_t T0) {
  int64_t g = (int64_t)B * ((T0 + CONV0_FPB - 1) / CONV0_FPB);
  if (g > CONV0_BWD_GRID) g = CONV0_BWD_GRID;
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" int64_t ca_conv0_bwd_partial_floats(int32_t B, int64_t N, int32_t C, int32_t k,
                                               int32_t stride) {
  const int64_t T0 = (N - k) / stride + 1;
  return (int64_t)conv0_bwd_grid(B, T0) * C * (k + 3);
}

extern "C" int ca_conv0_ln_gelu_bwd(const float* x, const float* w, const float* bias,
                                    const float* gamma, const float* beta, const void* dy,
                                    float* dw, float* dbias, float* dgamma, float* dbeta,
                                    float* partial, int32_t B, int64_t N, int32_t C, int32_t k,
                                    int32_t stride, float eps, void* stream) {
  CA_CHECK_ARG(x && w && bias && gamma && beta && dy && dw && dbias && dgamma && dbeta && partial,
               "ca_conv0_ln_gelu_bwd: null pointer");
  CA_CHECK_ARG(C == C0 && k == 10 && stride >= 1 && stride <= 16, "ca_conv0_ln_gelu_bwd: needs C=512, k=10, stride 1..16");
  const int64_t T0 = (N - k) / stride + 1;
  const int g = conv0_bwd_grid(B, T0);
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = (size_t)(CONV0_FPB * 16 + 16 + 4 * C0) * sizeof(float);
  hipLaunchKernelGGL((conv0_bwd_kernel<10>), dim3(g), dim3(256), lds, s, x, w, bias, gamma, beta,
                     (const unsigned short*)dy, partial, B, N, T0, stride, eps);
  const int64_t st = (int64_t)C0 * (k + 3);
  ca_reduce_partials_launch(partial, g, st, C0 * k, dw, 1, s);
  ca_reduce_partials_launch(partial + C0 * k, g, st, C0, dbias, 1, s);
  ca_reduce_partials_launch(partial + C0 * (k + 1), g, st, C0, dgamma, 1, s);
  ca_reduce_partials_launch(partial + C0 * (k + 2), g, st, C0, dbeta, 1, s);
  CA_CHECK_LAUNCH("ca_conv0_ln_gelu_bwd");
  return CA_OK;
}

// ---- col2im for strided channels-last Conv1d data gradients ---------------------------------
__global__ __launch_bounds__(256) void col2im_kernel(const unsigned short* __restrict__ dcol,
                                                     unsigned short* __restrict__ dx, int B,
                                                     int64_t T, int64_t L, int C, int k,
                                                     int stride) {
  const int cch = C >> 3;
  const int64_t total = (int64_t)B * L * cch;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % cch);
    const int64_t bp = i / cch;
    const int64_t p = bp % L, b = bp / L;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int j = (int)(p % stride); j < k; j += stride) {
      const int64_t t = (p - j) / stride;
      if (p - j < 0) break;
      if (t >= T) continue;
      const u16x8_t u =
          *(const u16x8_t*)(dcol + ((b * T + t) * k + j) * (int64_t)C + c8 * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += bf2f(u[e]);
    }
    u16x8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(a[e]);
    *(u16x8_t*)(dx + (b * L + p) * (int64_t)C + c8 * 8) = o;
  }
}

extern "C" int ca_col2im_1d(const void* dcol, void* dx, int32_t B, int64_t T, int64_t L,
                            int32_t C, int32_t k, int32_t stride, void* stream) {
  CA_CHECK_ARG(dcol && dx && B > 0 && T > 0 && L > 0 && (C % 8) == 0 && k > 0 && stride > 0,
               "ca_col2im_1d: bad argument");
  int64_t g = ((int64_t)B * L * (C / 8) + 255) / 256;
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(col2im_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)dcol, (unsigned short*)dx, B, T, L, C, k, stride);
  CA_CHECK_LAUNCH("ca_col2im_1d");
  return CA_OK;
}
