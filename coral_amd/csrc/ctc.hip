// CTC head: log-softmax (fp32) + CTC loss with gradient wrt the logits, and greedy decode.
// Semantics follow torch.nn.functional.ctc_loss as called at
// $TF/models/wav2vec2/modeling_wav2vec2.py:1717-1728 (blank = pad_token_id, zero_infinity),
// i.e. aten/src/ATen/native/LossCTC.cpp's alpha/beta recursions in log space.
//
// One 512-thread workgroup per utterance: waves 0-3 run the alpha recursion while waves 4-7
// run the beta recursion (the T sequential steps are the latency bound of this kernel, so the
// two chains share them); previous rows live in LDS, full alpha/beta tables go to the
// workspace (L2-resident) for the gradient pass, which is parallel over frames.
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
  return align256((int64_t)B * T * 256 * 4) + 2 * align256((int64_t)B * T * S * 4);
}

#define CTC_MAXS 1025

__global__ __launch_bounds__(512) void ctc_kernel(const float* __restrict__ logits,
                                                  const int32_t* __restrict__ labels,
                                                  const int32_t* __restrict__ in_len,
                                                  float* __restrict__ nll_out,
                                                  float* __restrict__ grad,
                                                  const float* __restrict__ gscale, float* lp,
                                                  float* alpha, float* beta, int T, int V,
                                                  int64_t ldv, int Lmax, int Smax, int blank,
                                                  int zero_inf) {
  __shared__ int ext[CTC_MAXS];         // extended label sequence l'
  __shared__ float rowa[2][CTC_MAXS];   // alpha ping-pong
  __shared__ float rowb[2][CTC_MAXS];   // beta ping-pong
  __shared__ int sh_L;
  __shared__ float sh_nll;
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const float* lg = logits + (int64_t)b * T * ldv;
  float* lpb = lp + (int64_t)b * T * V;
  float* ab = alpha + (int64_t)b * T * Smax;
  float* bb = beta + (int64_t)b * T * Smax;
  float* gb = grad ? grad + (int64_t)b * T * ldv : nullptr;
  int Tin = in_len ? in_len[b] : T;
  if (Tin > T) Tin = T;
  if (Tin < 0) Tin = 0;

  // 1. log-softmax rows (one wave per frame)
  for (int t = wave; t < T; t += 8) {
    const float* l = lg + (int64_t)t * ldv;
    float mx = NEG_INF;
    for (int v = lane; v < V; v += 64) mx = fmaxf(mx, l[v]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int v = lane; v < V; v += 64) s += __expf(l[v] - mx);
    s = wave_sum(s);
    const float lse = mx + __logf(s);
    for (int v = lane; v < V; v += 64) lpb[(int64_t)t * V + v] = l[v] - lse;
  }
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

  // 3. alpha (waves 0-3) and beta (waves 4-7) recursions, Tin lock-stepped time steps
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
  if (tid == 0) nll_out[b] = (infeasible && zero_inf) ? 0.f : nll;
  if (!gb) return;

  // 5. gradient wrt logits: thread per frame, serial over the S states (LossCTC.cpp's
  //    collect step): res[v] = logsumexp_{s: l'_s = v}(alpha_t(s) + beta_t(s));
  //    grad = exp(lp) - exp(res + nll - lp).  The grad row itself is the scratch for res.
  const float gs = gscale ? gscale[b] : 1.f;
  for (int t = tid; t < T; t += 512) {
    float* g = gb + (int64_t)t * ldv;
    if (t >= Tin || (infeasible)) {
      for (int v = 0; v < (int)ldv; ++v) g[v] = 0.f;
      continue;
    }
    for (int v = 0; v < V; ++v) g[v] = NEG_INF;
    const float* at = ab + (int64_t)t * Smax;
    const float* bt = bb + (int64_t)t * Smax;
    for (int s = 0; s < S; ++s) {
      const int c = ext[s];
      g[c] = log_add2(g[c], at[s] + bt[s]);
    }
    const float* lpt = lpb + (int64_t)t * V;
    for (int v = 0; v < V; ++v) {
      const float l = lpt[v];
      g[v] = (__expf(l) - __expf(g[v] + nll - l)) * gs;
    }
    for (int v = V; v < (int)ldv; ++v) g[v] = 0.f;
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
  hipLaunchKernelGGL(ctc_kernel, dim3(B), dim3(512), 0, (hipStream_t)stream, logits, labels,
                     in_len, nll, grad, gscale, lp, alpha, beta, T, V, ldv, Lmax, Smax, blank,
                     zero_infinity);
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
