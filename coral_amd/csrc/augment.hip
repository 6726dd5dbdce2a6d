// On-device waveform augmentation: the GPU side of the chain the reference builds from
// torch_audiomentations at R/src/coral/data.py:708-738 — Gain, AddBackgroundNoise / AddColoredNoise at a drawn
// SNR, and the windowed-sinc (julius-style) low/high/band-pass/band-stop filters.  All random decisions and
// filter designs are made on the host (coral_amd/augment.py); these kernels are deterministic given their
// parameter arrays.  fp32 [B, N] waveforms with per-utterance valid lengths; samples past the length stay 0.
#include "common.h"

// y[b, i] = x[b, i] * scale[b]
__global__ __launch_bounds__(256) void wave_scale_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                         float* __restrict__ y, int64_t N) {
  const int b = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < N) y[(int64_t)b * N + i] = x[(int64_t)b * N + i] * scale[b];
}

// FIR with an odd number of taps centred on the output sample, edge samples replicated (julius pads its input
// with mode="replicate").  mode[b]: 0 copy, 1 y = fir(x), 2 y = x - fir(x) (high-pass from a low-pass design).
// One workgroup = 1024 consecutive outputs; the input window lives in LDS, the taps are broadcast loads.
#define FIR_CHUNK 1024
__global__ __launch_bounds__(256) void fir_kernel(const float* __restrict__ x, const int32_t* __restrict__ lengths,
                                                  const float* __restrict__ taps, const int32_t* __restrict__ ntaps,
                                                  const int32_t* __restrict__ mode, int64_t ld_taps,
                                                  float* __restrict__ y, int64_t N) {
  extern __shared__ float win[];  // FIR_CHUNK + ntaps - 1 samples
  const int b = blockIdx.y;
  const int64_t c0 = (int64_t)blockIdx.x * FIR_CHUNK;
  int64_t len = lengths ? lengths[b] : N;
  if (len > N) len = N;
  const float* xb = x + (int64_t)b * N;
  float* yb = y + (int64_t)b * N;
  const int md = mode ? mode[b] : 1;
  if (md == 0 || c0 >= len) {
    for (int t = threadIdx.x; t < FIR_CHUNK; t += 256)
      if (c0 + t < N) yb[c0 + t] = c0 + t < len ? xb[c0 + t] : 0.f;
    return;
  }
  const int nt = ntaps[b];
  const int half = nt >> 1;
  const float* tp = taps + (int64_t)b * ld_taps;
  for (int t = threadIdx.x; t < FIR_CHUNK + nt - 1; t += 256) {
    int64_t i = c0 + t - half;
    i = i < 0 ? 0 : (i >= len ? len - 1 : i);  // replicate the edges
    win[t] = xb[i];
  }
  __syncthreads();
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int j = 0; j < nt; ++j) {
    const float w = tp[j];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = fmaf(w, win[threadIdx.x + 256 * e + j], acc[e]);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int64_t i = c0 + threadIdx.x + 256 * e;
    if (i < N) {
      const float xv = win[threadIdx.x + 256 * e + half];
      yb[i] = i < len ? (md == 2 ? xv - acc[e] : acc[e]) : 0.f;
    }
  }
}

// y = x + noise * (rms(x) / rms(noise)) * 10^(-snr_db / 20) over the valid samples (torch_audiomentations' SNR
// convention); active[b] == 0 leaves the utterance untouched.  noise rows may be shorter than the utterance:
// they are read cyclically from noise_off[b].
__global__ __launch_bounds__(1024) void mix_noise_kernel(const float* __restrict__ x, const int32_t* __restrict__ lengths,
                                                         const float* __restrict__ noise, int64_t noise_ld,
                                                         int64_t noise_len, const int64_t* __restrict__ noise_off,
                                                         const float* __restrict__ snr_db,
                                                         const int32_t* __restrict__ active, float* __restrict__ y,
                                                         int64_t N) {
  __shared__ float red[2][16];
  __shared__ float bc;
  const int b = blockIdx.x;
  const float* xb = x + (int64_t)b * N;
  float* yb = y + (int64_t)b * N;
  int64_t len = lengths ? lengths[b] : N;
  if (len > N) len = N;
  const bool on = (!active || active[b]) && len > 0;
  const float* nb = noise + (int64_t)b * noise_ld;
  const int64_t off = noise_off ? noise_off[b] : 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float sx = 0.f, sn = 0.f;
  if (on)
    for (int64_t i = threadIdx.x; i < len; i += 1024) {
      const float v = xb[i], n = nb[(off + i) % noise_len];
      sx += v * v;
      sn += n * n;
    }
  sx = wave_sum(sx);
  sn = wave_sum(sn);
  if (lane == 0) {
    red[0][wave] = sx;
    red[1][wave] = sn;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f, c = 0.f;
    for (int w = 0; w < 16; ++w) {
      a += red[0][w];
      c += red[1][w];
    }
    bc = (on && c > 0.f) ? sqrtf(a / c) * exp10f(-snr_db[b] * 0.05f) : 0.f;
  }
  __syncthreads();
  const float g = bc;
  for (int64_t i = threadIdx.x; i < N; i += 1024)
    yb[i] = i < len ? xb[i] + (on ? g * nb[(off + i) % noise_len] : 0.f) : 0.f;
}

// standard normal samples from a counter hash (Box-Muller on two 24-bit uniforms per pair)
__global__ __launch_bounds__(256) void white_noise_kernel(float* __restrict__ out, int64_t n, uint64_t seed) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;  // pair index
  if (2 * p >= n) return;
  unsigned int w0, w1;
  ca_dropout_words(seed, (uint64_t)p, w0, w1);
  const float u1 = ((float)(w0 >> 8) + 1.0f) * (1.0f / 16777216.0f);  // (0, 1]
  const float u2 = (float)(w1 >> 8) * (1.0f / 16777216.0f);
  const float r = sqrtf(-2.0f * __logf(u1));
  float s, c;
  __sincosf(6.28318530717958647692f * u2, &s, &c);
  out[2 * p] = r * c;
  if (2 * p + 1 < n) out[2 * p + 1] = r * s;
}

extern "C" int ca_wave_scale(const float* x, const float* scale, float* y, int32_t B, int64_t N, void* stream) {
  CA_CHECK_ARG(x && scale && y && B > 0 && N > 0, "ca_wave_scale: bad argument");
  hipLaunchKernelGGL(wave_scale_kernel, dim3((unsigned)((N + 255) / 256), B), dim3(256), 0, (hipStream_t)stream, x, scale,
                     y, N);
  CA_CHECK_LAUNCH("ca_wave_scale");
  return CA_OK;
}

extern "C" int ca_fir_filter(const float* x, const int32_t* lengths, const float* taps, const int32_t* ntaps,
                             const int32_t* mode, int64_t ld_taps, int32_t max_taps, float* y, int32_t B, int64_t N,
                             void* stream) {
  CA_CHECK_ARG(x && taps && ntaps && y && B > 0 && N > 0, "ca_fir_filter: bad argument");
  CA_CHECK_ARG(x != y, "ca_fir_filter: in-place filtering is not supported");
  CA_CHECK_ARG(max_taps > 0 && (max_taps & 1) && max_taps <= ld_taps && max_taps <= 16385,
               "ca_fir_filter: max_taps must be odd and at most 16385");
  const size_t lds = (size_t)(FIR_CHUNK + max_taps - 1) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)fir_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (FIR_CHUNK + 16384) * 4);
    attr = true;
  }
  hipLaunchKernelGGL(fir_kernel, dim3((unsigned)((N + FIR_CHUNK - 1) / FIR_CHUNK), B), dim3(256), lds,
                     (hipStream_t)stream, x, lengths, taps, ntaps, mode, ld_taps, y, N);
  CA_CHECK_LAUNCH("ca_fir_filter");
  return CA_OK;
}

extern "C" int ca_mix_noise(const float* x, const int32_t* lengths, const float* noise, int64_t noise_ld,
                            int64_t noise_len, const int64_t* noise_off, const float* snr_db, const int32_t* active,
                            float* y, int32_t B, int64_t N, void* stream) {
  CA_CHECK_ARG(x && noise && snr_db && y && B > 0 && N > 0 && noise_len > 0 && noise_ld >= 0, "ca_mix_noise: bad argument");
  hipLaunchKernelGGL(mix_noise_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, x, lengths, noise, noise_ld,
                     noise_len, noise_off, snr_db, active, y, N);
  CA_CHECK_LAUNCH("ca_mix_noise");
  return CA_OK;
}

extern "C" int ca_white_noise(float* out, int64_t n, uint64_t seed, void* stream) {
  CA_CHECK_ARG(out && n > 0, "ca_white_noise: bad argument");
  hipLaunchKernelGGL(white_noise_kernel, dim3((unsigned)((n / 2 + 256) / 256)), dim3(256), 0, (hipStream_t)stream, out, n,
                     seed);
  CA_CHECK_LAUNCH("ca_white_noise");
  return CA_OK;
}
