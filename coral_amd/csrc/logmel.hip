// Whisper log-mel front end on the GPU.
// Follows $TF/models/whisper/feature_extraction_whisper.py:135-168 (_torch_extract_fbank_features):
//   stft = torch.stft(wave, 400, 160, window=hann(400), return_complex=True)   (center, reflect)
//   mag  = stft[..., :-1].abs() ** 2 ; mel = filters.T @ mag ; log10(clamp(mel, 1e-10))
//   log_spec = max(log_spec, log_spec.max() - 8) ; (log_spec + 4) / 4          (max per clip)
// The 400-point DFT is evaluated exactly (not a zero-padded 512 FFT): each workgroup owns a tile
// of frames, keeps the windowed frames and a 400-entry twiddle table in LDS, one thread per
// frequency bin walks the table with an incremental (k*n mod 400) index.  The mel filterbank
// (201 x n_mels) is applied from LDS-resident power spectra.  Arithmetic is trivial next to the
// encoder; the kernel is written to read PCM once (coalesced) and write log-mel once.
#include "common.h"

#define NFFT 400
#define HOP 160
#define NBINS 201
#define FT 8  // frames per workgroup

__device__ __forceinline__ unsigned int f2ord(float f) {
  const unsigned int u = __builtin_bit_cast(unsigned int, f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned int o) {
  const unsigned int u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return __builtin_bit_cast(float, u);
}

__global__ __launch_bounds__(256) void logmel_kernel(const float* __restrict__ wave,
                                                     const float* __restrict__ filt,
                                                     float* __restrict__ out,
                                                     unsigned int* __restrict__ clipmax,
                                                     int64_t N, int frames, int n_mels) {
  __shared__ float cs[NFFT], sn[NFFT];
  __shared__ float xw[FT][NFFT];
  __shared__ float pw[FT][NBINS + 3];
  const int b = blockIdx.y;
  const int f0 = blockIdx.x * FT;
  const float* w = wave + (int64_t)b * N;
  for (int i = threadIdx.x; i < NFFT; i += 256) {
    float s, c;
    sincospif(2.0f * (float)i / (float)NFFT, &s, &c);
    cs[i] = c;
    sn[i] = s;
  }
  // windowed frames with reflect padding of NFFT/2 at both ends (torch.stft center=True)
  for (int i = threadIdx.x; i < FT * NFFT; i += 256) {
    const int f = i / NFFT, n = i % NFFT;
    const int fr = f0 + f;
    float v = 0.f;
    if (fr < frames) {
      int64_t p = (int64_t)fr * HOP + n - NFFT / 2;
      if (p < 0) p = -p;
      if (p >= N) p = 2 * (N - 1) - p;
      // periodic Hann: 0.5 - 0.5 cos(2 pi n / 400)
      float hs, hc;
      sincospif(2.0f * (float)n / (float)NFFT, &hs, &hc);
      v = w[p] * (0.5f - 0.5f * hc);
    }
    xw[f][n] = v;
  }
  __syncthreads();
  // one thread per frequency bin, all FT frames
  if (threadIdx.x < NBINS) {
    const int k = threadIdx.x;
    float re[FT], im[FT];
#pragma unroll
    for (int f = 0; f < FT; ++f) re[f] = im[f] = 0.f;
    int idx = 0;
    for (int n = 0; n < NFFT; ++n) {
      const float c = cs[idx], s = sn[idx];
#pragma unroll
      for (int f = 0; f < FT; ++f) {
        const float x = xw[f][n];
        re[f] = fmaf(x, c, re[f]);
        im[f] = fmaf(x, s, im[f]);
      }
      idx += k;
      if (idx >= NFFT) idx -= NFFT;
    }
#pragma unroll
    for (int f = 0; f < FT; ++f) pw[f][k] = re[f] * re[f] + im[f] * im[f];
  }
  __syncthreads();
  // mel projection + log10; thread per (mel, frame)
  float lmax = -1e30f;
  for (int i = threadIdx.x; i < n_mels * FT; i += 256) {
    const int m = i / FT, f = i % FT;
    const int fr = f0 + f;
    if (fr >= frames) continue;
    float a = 0.f;
    for (int k = 0; k < NBINS; ++k) a = fmaf(filt[k * n_mels + m], pw[f][k], a);
    const float l = log10f(fmaxf(a, 1e-10f));
    out[((int64_t)b * n_mels + m) * frames + fr] = l;
    lmax = fmaxf(lmax, l);
  }
  lmax = wave_max(lmax);
  if ((threadIdx.x & 63) == 0 && lmax > -1e29f) atomicMax(clipmax + b, f2ord(lmax));
}

__global__ void logmel_finish_kernel(float* __restrict__ out,
                                     const unsigned int* __restrict__ clipmax, int64_t per_clip,
                                     int B) {
  const int64_t total = per_clip * B;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / per_clip);
    const float mx = ord2f(clipmax[b]);
    out[i] = (fmaxf(out[i], mx - 8.0f) + 4.0f) * 0.25f;
  }
}

extern "C" int64_t ca_logmel_workspace_bytes(int32_t B) { return (int64_t)B * 4 + 256; }

extern "C" int ca_logmel(const float* wave, const float* mel_filters, float* out, void* ws,
                         int32_t B, int64_t N, int32_t n_mels, void* stream) {
  CA_CHECK_ARG(wave && mel_filters && out && ws, "ca_logmel: null pointer");
  CA_CHECK_ARG(B > 0 && N >= NFFT && (N % HOP) == 0 && n_mels > 0 && n_mels <= 256,
               "ca_logmel: N must be a multiple of %d", HOP);
  const int frames = (int)(N / HOP);  // stft gives frames+1; the last one is dropped
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(ws, 0, (size_t)B * 4, s) != hipSuccess) {
    ca_set_error("ca_logmel: memset failed");
    return CA_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(logmel_kernel, dim3((frames + FT - 1) / FT, B), dim3(256), 0, s, wave,
                     mel_filters, out, (unsigned int*)ws, N, frames, n_mels);
  const int64_t per = (int64_t)n_mels * frames;
  int64_t g = (per * B + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(logmel_finish_kernel, dim3((int)g), dim3(256), 0, s, out,
                     (const unsigned int*)ws, per, B);
  CA_CHECK_LAUNCH("ca_logmel");
  return CA_OK;
}
