// Attention softmax forward/backward (the middle of SDPA), cross-entropy, masked argmax.
// One 64-lane wave per row; rows are a few KB so the extra passes hit L1/L2, HBM sees one
// read of the scores and one write of the probabilities.
#include "common.h"

#define NEG_INF (-__builtin_inff())

__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ sc,
                                                          unsigned short* __restrict__ pr,
                                                          const int32_t* __restrict__ klen,
                                                          int64_t nrows, int H, int Tq, int Tk,
                                                          int64_t ld, int causal) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < nrows; row += (int64_t)gridDim.x * 4) {
    const int64_t bh = row / Tq;
    const int q = (int)(row % Tq);
    int lim = Tk;
    if (klen) {
      const int kl = klen[bh / H];
      lim = kl < lim ? kl : lim;
    }
    if (causal) lim = (q + 1) < lim ? (q + 1) : lim;
    const float* s = sc + row * ld;
    unsigned short* p = pr + row * ld;
    float mx = NEG_INF;
    for (int c = lane * 4; c < lim; c += 256) {
      const f32x4_t v = *(const f32x4_t*)(s + c);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c + e < lim) mx = fmaxf(mx, v[e]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int c = lane * 4; c < lim; c += 256) {
      const f32x4_t v = *(const f32x4_t*)(s + c);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c + e < lim) sum += __expf(v[e] - mx);
    }
    sum = wave_sum(sum);
    const float inv = sum > 0.f ? 1.f / sum : 0.f;
    for (int c = lane * 4; c < ld; c += 256) {
      u16x4_t o;
      if (c < lim) {
        const f32x4_t v = *(const f32x4_t*)(s + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (c + e < lim) ? f2bf(__expf(v[e] - mx) * inv) : 0;
      } else {
        o = (u16x4_t){0, 0, 0, 0};
      }
      *(u16x4_t*)(p + c) = o;
    }
  }
}

extern "C" int ca_softmax_fwd(const float* scores, void* probs, const int32_t* klen, int32_t BH,
                              int32_t H, int32_t Tq, int32_t Tk, int64_t ld, int32_t causal,
                              void* stream) {
  CA_CHECK_ARG(scores && probs && BH > 0 && H > 0 && Tq > 0 && Tk > 0, "ca_softmax_fwd: bad arg");
  CA_CHECK_ARG(ld >= Tk && (ld % 8) == 0, "ca_softmax_fwd: ld must be >= Tk and a multiple of 8");
  const int64_t nrows = (int64_t)BH * Tq;
  int64_t g = (nrows + 3) / 4;
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(softmax_fwd_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, scores,
                     (unsigned short*)probs, klen, nrows, H, Tq, Tk, ld, causal);
  CA_CHECK_LAUNCH("ca_softmax_fwd");
  return CA_OK;
}

__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ dp,
                                                          const unsigned short* __restrict__ pr,
                                                          unsigned short* __restrict__ ds,
                                                          float scale, int64_t nrows, int Tk,
                                                          int64_t ld) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < nrows; row += (int64_t)gridDim.x * 4) {
    const float* d = dp + row * ld;
    const unsigned short* p = pr + row * ld;
    unsigned short* o = ds + row * ld;
    float dot = 0.f;
    for (int c = lane * 4; c < Tk; c += 256) {
      const f32x4_t v = *(const f32x4_t*)(d + c);
      const u16x4_t u = *(const u16x4_t*)(p + c);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c + e < Tk) dot += v[e] * bf2f(u[e]);
    }
    dot = wave_sum(dot);
    for (int c = lane * 4; c < ld; c += 256) {
      u16x4_t r;
      if (c < Tk) {
        const f32x4_t v = *(const f32x4_t*)(d + c);
        const u16x4_t u = *(const u16x4_t*)(p + c);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          r[e] = (c + e < Tk) ? f2bf(bf2f(u[e]) * (v[e] - dot) * scale) : 0;
      } else {
        r = (u16x4_t){0, 0, 0, 0};
      }
      *(u16x4_t*)(o + c) = r;
    }
  }
}

extern "C" int ca_softmax_bwd(const float* dprobs, const void* probs, void* dscores, float scale,
                              int32_t BH, int32_t Tq, int32_t Tk, int64_t ld, void* stream) {
  CA_CHECK_ARG(dprobs && probs && dscores && BH > 0 && Tq > 0 && Tk > 0, "ca_softmax_bwd: bad arg");
  CA_CHECK_ARG(ld >= Tk && (ld % 8) == 0, "ca_softmax_bwd: ld must be >= Tk and a multiple of 8");
  const int64_t nrows = (int64_t)BH * Tq;
  int64_t g = (nrows + 3) / 4;
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, dprobs,
                     (const unsigned short*)probs, (unsigned short*)dscores, scale, nrows, Tk, ld);
  CA_CHECK_LAUNCH("ca_softmax_bwd");
  return CA_OK;
}

// ---- cross-entropy (Whisper LM loss) ---------------------------------------------------------
// $TF/models/whisper/modeling_whisper.py:1084-1087: CrossEntropyLoss(ignore_index=-100), mean
// taken by the caller (loss_sum / count).  V ~ 51 866.
// One 1024-thread workgroup per row (a wave per row walked 810 dependent loads three times: 0.67 ms for the
// 688 x 51865 logits of a whisper-medium step); 16-byte loads, the row is read from L2 on the second and third sweep.
__global__ __launch_bounds__(1024) void ce_kernel(const float* __restrict__ lg,
                                                  const int32_t* __restrict__ labels,
                                                  float* __restrict__ loss_sum,
                                                  int32_t* __restrict__ count,
                                                  float* __restrict__ grad, int64_t rows, int V,
                                                  int64_t ldv, int ignore) {
  __shared__ float red[16];
  __shared__ float bc;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = blockIdx.x;
  const float* l = lg + row * ldv;
  float* g = grad ? grad + row * ldv : nullptr;
  const int lab = labels[row];
  const int n4 = (int)(ldv >> 2);  // ldv is a multiple of 4 (checked by the launcher)
  if (lab == ignore || lab < 0 || lab >= V) {
    if (g)
      for (int c = threadIdx.x; c < n4; c += 1024) *(f32x4_t*)(g + 4 * c) = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    return;
  }
  float mx = NEG_INF;
  for (int c = threadIdx.x; c < n4; c += 1024) {
    const f32x4_t v = *(const f32x4_t*)(l + 4 * c);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (4 * c + e < V) mx = fmaxf(mx, v[e]);
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = red[0];
    for (int w = 1; w < 16; ++w) m = fmaxf(m, red[w]);
    bc = m;
  }
  __syncthreads();
  mx = bc;
  float sum = 0.f;
  for (int c = threadIdx.x; c < n4; c += 1024) {
    const f32x4_t v = *(const f32x4_t*)(l + 4 * c);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (4 * c + e < V) sum += __expf(v[e] - mx);
  }
  sum = wave_sum(sum);
  __syncthreads();
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < 16; ++w) s += red[w];
    const float lse = mx + __logf(s);
    bc = lse;
    atomicAdd(loss_sum, lse - l[lab]);
    atomicAdd(count, 1);
  }
  __syncthreads();
  const float lse = bc;
  if (g) {
    for (int c = threadIdx.x; c < n4; c += 1024) {
      const f32x4_t v = *(const f32x4_t*)(l + 4 * c);
      f32x4_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int col = 4 * c + e;
        o[e] = col < V ? __expf(v[e] - lse) - (col == lab ? 1.f : 0.f) : 0.f;
      }
      *(f32x4_t*)(g + 4 * c) = o;
    }
  }
}

extern "C" int ca_cross_entropy_fwd_bwd(const float* logits, const int32_t* labels,
                                        float* loss_sum, int32_t* count, float* grad,
                                        int64_t rows, int32_t V, int64_t ldv,
                                        int32_t ignore_index, void* stream) {
  CA_CHECK_ARG(logits && labels && loss_sum && count && rows > 0 && V > 0 && ldv >= V,
               "ca_cross_entropy_fwd_bwd: bad argument");
  CA_CHECK_ARG((ldv % 4) == 0 && ((uintptr_t)logits % 16) == 0 && (!grad || ((uintptr_t)grad % 16) == 0),
               "ca_cross_entropy_fwd_bwd: ldv must be a multiple of 4 and the buffers 16-byte aligned");
  hipLaunchKernelGGL(ce_kernel, dim3((unsigned)rows), dim3(1024), 0, (hipStream_t)stream, logits, labels,
                     loss_sum, count, grad, rows, V, ldv, ignore_index);
  CA_CHECK_LAUNCH("ca_cross_entropy_fwd_bwd");
  return CA_OK;
}

// ---- masked argmax (greedy generation) -------------------------------------------------------
// one 1024-thread workgroup per row (greedy decoding has B rows of ~52 k logits: a wave per row would walk
// 800 dependent loads); ties resolve to the lowest index like torch.argmax.
// adv (ca_argmax_advance): the bookkeeping of a greedy step in the same launch - the row's last thread also records the
// token and moves the row's cursors (eight tiny dependent launches per decoded token otherwise, ~4 us each in a graph)
struct ArgmaxAdvance {
  uint8_t* done;    // [rows] finished flags (a finished row records pad)
  int64_t* ids;     // [rows, ld_ids] generated ids; the token goes to column pos[row] + 1
  int64_t ld_ids;
  int32_t* tok;     // [rows] next input token
  int32_t* pos;     // [rows] position of the token just fed (+1 here)
  int32_t* klen;    // [rows] cached keys (+1 here)
  int32_t pad, eos;
};
__global__ __launch_bounds__(1024) void argmax_kernel(const float* __restrict__ lg,
                                                      const uint8_t* __restrict__ suppress,
                                                      int32_t* __restrict__ out, int64_t rows,
                                                      int V, int64_t ldv, const ArgmaxAdvance adv) {
  __shared__ float sb[16];
  __shared__ int si[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = blockIdx.x;
  const float* l = lg + row * ldv;
  float best = NEG_INF;
  int bi = 0x7fffffff;
  // four logits (and their four suppress bytes) per load where the row is 16-byte aligned; a thread's candidates
  // come in increasing index order, so "strictly greater" keeps the lowest index of equal values
  const bool vec = ((ldv & 3) == 0) && ((((uintptr_t)lg) & 15) == 0) && (!suppress || (((uintptr_t)suppress) & 3) == 0);
  const int V4 = vec ? (V >> 2) : 0;
  for (int q = threadIdx.x; q < V4; q += 1024) {
    const f32x4_t v4 = *(const f32x4_t*)(l + 4 * q);
    const unsigned int sm = suppress ? *(const unsigned int*)(suppress + 4 * q) : 0u;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if ((sm >> (8 * e)) & 0xffu) continue;
      if (v4[e] > best) {
        best = v4[e];
        bi = 4 * q + e;
      }
    }
  }
  for (int c = 4 * V4 + threadIdx.x; c < V; c += 1024) {
    if (suppress && suppress[c]) continue;
    const float v = l[c];
    if (v > best || (v == best && c < bi)) {
      best = v;
      bi = c;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) {
      best = ob;
      bi = oi;
    }
  }
  if (lane == 0) {
    sb[wave] = best;
    si[wave] = bi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w)
      if (sb[w] > best || (sb[w] == best && si[w] < bi)) {
        best = sb[w];
        bi = si[w];
      }
    bi = bi == 0x7fffffff ? 0 : bi;
    out[row] = bi;
    if (adv.ids) {
      const int32_t p = adv.pos[row];
      const int32_t step = adv.done[row] ? adv.pad : bi;
      adv.ids[row * adv.ld_ids + p + 1] = step;
      if (step == adv.eos) adv.done[row] = 1;
      adv.tok[row] = step;
      adv.pos[row] = p + 1;
      adv.klen[row] += 1;
    }
  }
}

extern "C" int ca_argmax_masked(const float* logits, const uint8_t* suppress, int32_t* out,
                                int64_t rows, int32_t V, int64_t ldv, void* stream) {
  CA_CHECK_ARG(logits && out && rows > 0 && V > 0 && ldv >= V, "ca_argmax_masked: bad argument");
  ArgmaxAdvance none = {};
  hipLaunchKernelGGL(argmax_kernel, dim3((unsigned)rows), dim3(1024), 0, (hipStream_t)stream, logits,
                     suppress, out, rows, V, ldv, none);
  CA_CHECK_LAUNCH("ca_argmax_masked");
  return CA_OK;
}
extern "C" int ca_argmax_advance(const float* logits, const uint8_t* suppress, int32_t* out, int64_t rows, int32_t V,
                                 int64_t ldv, uint8_t* done, int64_t* ids, int64_t ld_ids, int32_t* tok, int32_t* pos,
                                 int32_t* klen, int32_t pad_id, int32_t eos_id, void* stream) {
  CA_CHECK_ARG(logits && out && rows > 0 && V > 0 && ldv >= V && done && ids && tok && pos && klen && ld_ids > 0,
               "ca_argmax_advance: bad argument");
  ArgmaxAdvance adv = {done, ids, ld_ids, tok, pos, klen, pad_id, eos_id};
  hipLaunchKernelGGL(argmax_kernel, dim3((unsigned)rows), dim3(1024), 0, (hipStream_t)stream, logits,
                     suppress, out, rows, V, ldv, adv);
  CA_CHECK_LAUNCH("ca_argmax_advance");
  return CA_OK;
}
