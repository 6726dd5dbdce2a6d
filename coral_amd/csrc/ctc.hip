// CTC head: log-softmax (fp32) + CTC loss with gradient wrt the logits, and greedy decode.
// Semantics follow torch.nn.functional.ctc_loss as called at
// $TF/models/wav2vec2/modeling_wav2vec2.py:1717-1728 (blank = pad_token_id, zero_infinity),
// i.e. aten/src/ATen/native/LossCTC.cpp's alpha/beta recursions in log space.
//
// Three launches: log-softmax of every frame (wave per frame, whole chip); the alpha/beta recursions (one
// workgroup per utterance - the T sequential steps are the latency bound: for label sequences up to 127
// tokens one wave per direction keeps 4 states per lane and exchanges neighbours by lane shifts, no barrier
// or LDS inside a step; longer ones fall back to 256 threads per direction with LDS rows); the gradient
// (wave per frame, whole chip) from the alpha/beta tables kept in the workspace.
#include "common.h"

#define NEG_INF (-__builtin_inff())

__device__ __forceinline__ float log_add3(float a, float b, float c) {
  const float m = fmaxf(a, fmaxf(b, c));
  if (m == NEG_INF) return NEG_INF;
  return m + __logf(__expf(a - m) + __expf(b - m) + __expf(c - m));
}
__device__ __forceinline__ float log_add2(float a, float b) {
  const float m = fmaxf(a, b);
  if (m == NEG_INF) return NEG_INF;
  return m + __logf(__expf(a - m) + __expf(b - m));
}

struct CtcWs {
  float* lp;     // [B][T][V]
  float* alpha;  // [B][T][S]
  float* beta;   // [B][T][S]
};

__device__ __host__ inline int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

extern "C" int64_t ca_ctc_workspace_bytes(int32_t B, int32_t T, int32_t Lmax) {
  const int64_t S = 2 * (int64_t)Lmax + 1;
  // V is not known here; reserve 256 columns for lp (V <= 256 enforced at call time)
  return align256((int64_t)B * T * 256 * 4) + 2 * align256((int64_t)B * T * S * 4) + align256((int64_t)B * 4);
}

#define CTC_MAXS 1025

// log-softmax of every frame (wave per frame, the whole chip): lp[b][t][v] = logit - logsumexp
__global__ __launch_bounds__(256) void ctc_logsoftmax_kernel(const float* __restrict__ logits, float* __restrict__ lp,
                                                             int64_t frames, int V, int64_t ldv) {
  const int lane = threadIdx.x & 63;
  const int64_t f = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (f >= frames) return;
  const float* l = logits + f * ldv;
  float x[4];  // V <= 256
  float mx = NEG_INF;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int v = lane + 64 * j;
    x[j] = v < V ? l[v] : NEG_INF;
    mx = fmaxf(mx, x[j]);
  }
  mx = wave_max(mx);
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) s += (lane + 64 * j) < V ? __expf(x[j] - mx) : 0.f;
  s = wave_sum(s);
  const float lse = mx + __logf(s);
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (lane + 64 * j < V) lp[f * V + lane + 64 * j] = x[j] - lse;
}

__global__ __launch_bounds__(512) void ctc_kernel(const int32_t* __restrict__ labels,
                                                  const int32_t* __restrict__ in_len,
                                                  float* __restrict__ nll_out,
                                                  float* __restrict__ raw_nll, const float* __restrict__ lp,
                                                  float* alpha, float* beta, int T, int V,
                                                  int Lmax, int Smax, int blank, int zero_inf, int lp_in_lds) {
  __shared__ int ext[CTC_MAXS];         // extended label sequence l'
  __shared__ float rowa[2][CTC_MAXS];   // alpha ping-pong
  __shared__ float rowb[2][CTC_MAXS];   // beta ping-pong
  __shared__ int sh_L;
  __shared__ float sh_nll;
  extern __shared__ float lds_lp[];  // this utterance's log-probs [T][V] when they fit (lp_in_lds)
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const float* lpb = lp + (int64_t)b * T * V;
  if (lp_in_lds)
    for (int i = tid; i < T * V; i += 512) lds_lp[i] = lpb[i];
  float* ab = alpha + (int64_t)b * T * Smax;
  float* bb = beta + (int64_t)b * T * Smax;
  int Tin = in_len ? in_len[b] : T;
  if (Tin > T) Tin = T;
  if (Tin < 0) Tin = 0;

  // 2. extended labels (targets = labels >= 0, in order)
  if (tid == 0) {
    int L = 0;
    for (int i = 0; i < Lmax; ++i) {
      const int c = labels[(int64_t)b * Lmax + i];
      if (c >= 0) {
        ext[2 * L] = blank;
        ext[2 * L + 1] = c;
        ++L;
      }
    }
    ext[2 * L] = blank;
    sh_L = L;
  }
  __syncthreads();
  const int L = sh_L;
  const int S = 2 * L + 1;
  const bool is_alpha = tid < 256;
  const int st = tid & 255;

  // 3. alpha / beta recursions.
  if (S <= 256) {
    // Fast path (every CoRal label length): one wave per direction, 4 states per lane, neighbours through
    // lane shifts - no barrier and no LDS inside the T sequential steps; the next step's log-probs are
    // fetched while the current step is combined.
    if (wave < 2 && Tin > 0) {
      const bool fwd = wave == 0;
      float* tab = fwd ? ab : bb;
      int c[4];
      bool skip[4], valid[4];
      float a[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int s = 4 * lane + j;
        valid[j] = s < S;
        c[j] = valid[j] ? ext[s] : blank;
        const int s2 = fwd ? s - 2 : s + 2;
        skip[j] = valid[j] && s2 >= 0 && s2 < S && c[j] != blank && ext[s2] != c[j];
      }
      const int t0 = fwd ? 0 : Tin - 1;
      // gathers of log-probs: LDS copy (64-cycle class latency inside the sequential chain) or global
      auto lpv = [&](int tt, int cc) -> float { return lp_in_lds ? lds_lp[tt * V + cc] : lpb[(int64_t)tt * V + cc]; };
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int s = 4 * lane + j;
        const bool init = fwd ? (s == 0 || s == 1) : (s == S - 1 || s == S - 2);
        a[j] = (valid[j] && init) ? lpv(t0, c[j]) : NEG_INF;
        if (valid[j]) tab[(int64_t)t0 * Smax + s] = a[j];
      }
      float nx[4];
      if (Tin > 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) nx[j] = lpv(fwd ? 1 : Tin - 2, c[j]);
      }
      for (int step = 1; step < Tin; ++step) {
        const int tt = fwd ? step : Tin - 1 - step;
        float cur[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) cur[j] = nx[j];
        if (step + 1 < Tin) {
#pragma unroll
          for (int j = 0; j < 4; ++j) nx[j] = lpv(fwd ? step + 1 : Tin - 2 - step, c[j]);
        }
        float n1, n2;  // the two states beyond this lane's block, in the direction of the recursion
        if (fwd) {
          n1 = __shfl_up(a[3], 1, 64);
          n2 = __shfl_up(a[2], 1, 64);
          if (lane == 0) n1 = n2 = NEG_INF;
        } else {
          n1 = __shfl_down(a[0], 1, 64);
          n2 = __shfl_down(a[1], 1, 64);
          if (lane == 63) n1 = n2 = NEG_INF;
        }
        float na[4];
        if (fwd) {
          na[0] = log_add3(a[0], n1, skip[0] ? n2 : NEG_INF);
          na[1] = log_add3(a[1], a[0], skip[1] ? n1 : NEG_INF);
          na[2] = log_add3(a[2], a[1], skip[2] ? a[0] : NEG_INF);
          na[3] = log_add3(a[3], a[2], skip[3] ? a[1] : NEG_INF);
        } else {
          na[3] = log_add3(a[3], n1, skip[3] ? n2 : NEG_INF);
          na[2] = log_add3(a[2], a[3], skip[2] ? n1 : NEG_INF);
          na[1] = log_add3(a[1], a[2], skip[1] ? a[3] : NEG_INF);
          na[0] = log_add3(a[0], a[1], skip[0] ? a[2] : NEG_INF);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          a[j] = valid[j] ? na[j] + cur[j] : NEG_INF;
          if (valid[j]) tab[(int64_t)tt * Smax + 4 * lane + j] = a[j];
        }
      }
      if (fwd) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (valid[j]) rowa[(Tin - 1) & 1][4 * lane + j] = a[j];  // read by the likelihood below
      }
    }
    __syncthreads();
  } else {
    if (Tin > 0) {
      if (is_alpha) {
        for (int s = st; s < S; s += 256) {
          float a = NEG_INF;
          if (s == 0) a = lpb[blank];
          else if (s == 1) a = lpb[ext[1]];
          rowa[0][s] = a;
          ab[s] = a;
        }
      } else {
        const float* lpt = lpb + (int64_t)(Tin - 1) * V;
        for (int s = st; s < S; s += 256) {
          float v = NEG_INF;
          if (s == S - 1) v = lpt[blank];
          else if (s == S - 2) v = lpt[ext[S - 2]];
          rowb[0][s] = v;
          bb[(int64_t)(Tin - 1) * Smax + s] = v;
        }
      }
    }
    __syncthreads();
    for (int step = 1; step < Tin; ++step) {
      const int cur = step & 1, prv = cur ^ 1;
      if (is_alpha) {
        const int t = step;
        const float* lpt = lpb + (int64_t)t * V;
        for (int s = st; s < S; s += 256) {
          const int c = ext[s];
          const float a0 = rowa[prv][s];
          const float a1 = s >= 1 ? rowa[prv][s - 1] : NEG_INF;
          const float a2 = (s >= 2 && c != blank && ext[s - 2] != c) ? rowa[prv][s - 2] : NEG_INF;
          const float a = log_add3(a0, a1, a2) + lpt[c];
          rowa[cur][s] = a;
          ab[(int64_t)t * Smax + s] = a;
        }
      } else {
        const int t = Tin - 1 - step;
        const float* lpt = lpb + (int64_t)t * V;
        for (int s = st; s < S; s += 256) {
          const int c = ext[s];
          const float b0 = rowb[prv][s];
          const float b1 = s + 1 < S ? rowb[prv][s + 1] : NEG_INF;
          const float b2 =
              (s + 2 < S && c != blank && ext[s + 2] != c) ? rowb[prv][s + 2] : NEG_INF;
          const float v = log_add3(b0, b1, b2) + lpt[c];
          rowb[cur][s] = v;
          bb[(int64_t)t * Smax + s] = v;
        }
      }
      __syncthreads();
    }
  }
  // 4. negative log-likelihood
  if (tid == 0) {
    float nll;
    if (Tin <= 0) {
      nll = __builtin_inff();
    } else {
      const int lastrow = (Tin - 1) & 1;
      const float l1 = rowa[lastrow][S - 1];
      const float l2 = S >= 2 ? rowa[lastrow][S - 2] : NEG_INF;
      nll = -log_add2(l1, l2);
    }
    sh_nll = nll;
  }
  __syncthreads();  // also makes this block's global alpha/beta stores visible to itself
  float nll = sh_nll;
  const bool infeasible = !(nll < __builtin_inff());  // inf or nan
  if (tid == 0) {
    nll_out[b] = (infeasible && zero_inf) ? 0.f : nll;
    raw_nll[b] = nll;
  }
}

// gradient wrt the logits, one wave per frame over the whole chip (LossCTC.cpp's collect step):
// res[v] = logsumexp_{s: l'_s = v}(alpha_t(s) + beta_t(s)); grad = exp(lp) - exp(res + nll - lp).
// The per-character log-sum-exp is built in a 2 x 256-float LDS scratch per wave with ds_max_f32 /
// ds_add_f32 (a maximum pass, then a sum of exponentials; the additions inside one character commute
// up to fp32 rounding).
__global__ __launch_bounds__(512) void ctc_grad_kernel(const int32_t* __restrict__ labels,
                                                       const int32_t* __restrict__ in_len,
                                                       const float* __restrict__ raw_nll, float* __restrict__ grad,
                                                       const float* __restrict__ gscale, const float* __restrict__ lp,
                                                       const float* __restrict__ alpha, const float* __restrict__ beta,
                                                       int T, int V, int64_t ldv, int Lmax, int Smax, int blank) {
  __shared__ int ext[CTC_MAXS];
  __shared__ float vm[8][256], vs[8][256];
  __shared__ int sh_L;
  const int b = blockIdx.y;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  if (tid == 0) {
    int L = 0;
    for (int i = 0; i < Lmax; ++i) {
      const int c = labels[(int64_t)b * Lmax + i];
      if (c >= 0) {
        ext[2 * L] = blank;
        ext[2 * L + 1] = c;
        ++L;
      }
    }
    ext[2 * L] = blank;
    sh_L = L;
  }
  __syncthreads();
  const int S = 2 * sh_L + 1;
  int Tin = in_len ? in_len[b] : T;
  if (Tin > T) Tin = T;
  if (Tin < 0) Tin = 0;
  const float nll = raw_nll[b];
  const bool infeasible = !(nll < __builtin_inff());
  const float gs = gscale ? gscale[b] : 1.f;
  const int t = blockIdx.x * 8 + wave;
  if (t >= T) return;
  float* g = grad + ((int64_t)b * T + t) * ldv;
  if (t >= Tin || infeasible) {
    for (int v = lane; v < (int)ldv; v += 64) g[v] = 0.f;
    return;
  }
  float* vmax = vm[wave];
  float* vsum = vs[wave];
  for (int v = lane; v < V; v += 64) {
    vmax[v] = NEG_INF;
    vsum[v] = 0.f;
  }
  const float* at = alpha + ((int64_t)b * T + t) * Smax;
  const float* bt = beta + ((int64_t)b * T + t) * Smax;
  for (int s = lane; s < S; s += 64) {
    const float val = at[s] + bt[s];
    if (val > NEG_INF)
      __builtin_amdgcn_ds_fmaxf((__attribute__((address_space(3))) float*)&vmax[ext[s]], val, 0, 0, false);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  for (int s = lane; s < S; s += 64) {
    const float val = at[s] + bt[s];
    if (val > NEG_INF) {
      const int cc = ext[s];
      __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float*)&vsum[cc], __expf(val - vmax[cc]), 0, 0, false);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const float* lpt = lp + ((int64_t)b * T + t) * V;
  for (int v = lane; v < (int)ldv; v += 64) {
    if (v < V) {
      const float l = lpt[v];
      const float res = vsum[v] > 0.f ? vmax[v] + __logf(vsum[v]) : NEG_INF;
      g[v] = (__expf(l) - __expf(res + nll - l)) * gs;
    } else {
      g[v] = 0.f;
    }
  }
}

extern "C" int ca_ctc_loss_fwd_bwd(const float* logits, const int32_t* labels,
                                   const int32_t* in_len, float* nll, float* grad,
                                   const float* gscale, void* ws, int32_t B, int32_t T, int32_t V,
                                   int64_t ldv, int32_t Lmax, int32_t blank, int32_t zero_infinity,
                                   void* stream) {
  CA_CHECK_ARG(logits && labels && nll && ws, "ca_ctc_loss_fwd_bwd: null pointer");
  CA_CHECK_ARG(B > 0 && T > 0 && V > 0 && V <= 256 && ldv >= V && Lmax >= 0,
               "ca_ctc_loss_fwd_bwd: bad shape (V must be <= 256)");
  CA_CHECK_ARG(2 * Lmax + 1 <= CTC_MAXS, "ca_ctc_loss_fwd_bwd: Lmax %d too long (max %d)", Lmax,
               (CTC_MAXS - 1) / 2);
  CA_CHECK_ARG(blank >= 0 && blank < V, "ca_ctc_loss_fwd_bwd: bad blank");
  const int Smax = 2 * Lmax + 1;
  char* w = (char*)ws;
  float* lp = (float*)w;
  w += align256((int64_t)B * T * 256 * 4);
  float* alpha = (float*)w;
  w += align256((int64_t)B * T * Smax * 4);
  float* beta = (float*)w;
  w += align256((int64_t)B * T * Smax * 4);
  float* raw_nll = (float*)w;
  hipStream_t s = (hipStream_t)stream;
  const int64_t frames = (int64_t)B * T;
  hipLaunchKernelGGL(ctc_logsoftmax_kernel, dim3((unsigned)((frames + 3) / 4)), dim3(256), 0, s, logits, lp, frames, V,
                     ldv);
  const size_t lp_bytes = (size_t)T * V * sizeof(float);
  const int lp_in_lds = lp_bytes <= 120 * 1024 ? 1 : 0;
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)ctc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(ctc_kernel, dim3(B), dim3(512), lp_in_lds ? lp_bytes : 0, s, labels, in_len, nll, raw_nll, lp, alpha,
                     beta, T, V, Lmax, Smax, blank, zero_infinity, lp_in_lds);
  if (grad)
    hipLaunchKernelGGL(ctc_grad_kernel, dim3((unsigned)((T + 7) / 8), B), dim3(512), 0, s, labels, in_len, raw_nll, grad,
                       gscale, lp, alpha, beta, T, V, ldv, Lmax, Smax, blank);
  CA_CHECK_LAUNCH("ca_ctc_loss_fwd_bwd");
  return CA_OK;
}

// ---- greedy decode: argmax, collapse repeats, drop blank -------------------------------------
// R/src/coral/compute_metrics.py:68-69 + $TF/models/wav2vec2/tokenization_wav2vec2.py:311-323.
__global__ __launch_bounds__(1024) void ctc_greedy_kernel(const float* __restrict__ logits,
                                                          const int32_t* __restrict__ in_len,
                                                          int32_t* __restrict__ raw,
                                                          int32_t* __restrict__ ids,
                                                          int32_t* __restrict__ out_len, int T,
                                                          int V, int64_t ldv, int blank) {
  extern __shared__ int sh[];  // [T] keep flags -> inclusive prefix sums
  const int b = blockIdx.x;
  const float* lg = logits + (int64_t)b * T * ldv;
  int32_t* rb = raw + (int64_t)b * T;
  int32_t* ib = ids + (int64_t)b * T;
  int Tin = in_len ? in_len[b] : T;
  if (Tin > T) Tin = T;
  for (int t = threadIdx.x; t < T; t += 1024) {
    const float* l = lg + (int64_t)t * ldv;
    float best = l[0];
    int bi = 0;
    for (int v = 1; v < V; ++v) {
      const float x = l[v];
      if (x > best) {  // first maximum wins, as np.argmax / torch.argmax
        best = x;
        bi = v;
      }
    }
    rb[t] = bi;
    ib[t] = -1;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < T; t += 1024) {
    const int c = rb[t];
    sh[t] = (t < Tin && c != blank && (t == 0 || rb[t - 1] != c)) ? 1 : 0;
  }
  __syncthreads();
  // Hillis-Steele inclusive scan over T <= a few thousand entries
  for (int off = 1; off < T; off <<= 1) {
    int vals[4];
    int n = 0;
    for (int t = threadIdx.x; t < T; t += 1024) vals[n++] = sh[t] + (t >= off ? sh[t - off] : 0);
    __syncthreads();
    n = 0;
    for (int t = threadIdx.x; t < T; t += 1024) sh[t] = vals[n++];
    __syncthreads();
  }
  for (int t = threadIdx.x; t < T; t += 1024) {
    const int incl = sh[t];
    const int prev = t > 0 ? sh[t - 1] : 0;
    if (incl != prev) ib[incl - 1] = rb[t];
  }
  if (threadIdx.x == 0) out_len[b] = T > 0 ? sh[T - 1] : 0;
}

extern "C" int ca_ctc_greedy_decode(const float* logits, const int32_t* in_len, int32_t* raw,
                                    int32_t* ids, int32_t* out_len, int32_t B, int32_t T,
                                    int32_t V, int64_t ldv, int32_t blank, void* stream) {
  CA_CHECK_ARG(logits && raw && ids && out_len, "ca_ctc_greedy_decode: null pointer");
  CA_CHECK_ARG(B > 0 && T > 0 && T <= 4096 && V > 0 && ldv >= V,
               "ca_ctc_greedy_decode: bad shape (T <= 4096)");
  hipLaunchKernelGGL(ctc_greedy_kernel, dim3(B), dim3(1024), (size_t)T * sizeof(int),
                     (hipStream_t)stream, logits, in_len, raw, ids, out_len, T, V, ldv, blank);
  CA_CHECK_LAUNCH("ca_ctc_greedy_decode");
  return CA_OK;
}
