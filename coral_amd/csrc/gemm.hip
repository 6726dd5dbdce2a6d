// MFMA bf16 GEMM family for gfx950 (CDNA4).  See include/coral_amd.h (ca_gemm_bf16).
//
// Design (v1, "128x128x64 / LDS-DMA / 2-buffer"):
//   * 256 threads = 4 waves in a 2x2 grid, each wave owns a 64x64 output tile as 4x4
//     v_mfma_f32_16x16x32_bf16 accumulators (operands swapped so a lane ends up with 4
//     consecutive n of one m: 8-byte bf16 / 16-byte fp32 stores).
//   * both operand tiles are staged HBM -> LDS with global_load_lds_dwordx4 (no VGPR round
//     trip); the LDS image is lane-linear, bank conflicts are removed by permuting the
//     per-lane SOURCE chunk and applying the same XOR on the fragment read.
//   * KMAJOR operands ([rows][64 k], 128-B rows) are read with ds_read_b128;
//     MNMAJOR operands ([64 k][128 mn], 256-B rows) with ds_read_b64_tr_b16 (hardware
//     transpose), so NT / NN / TN / TT all run natively without transposed copies.
//   * out-of-range rows are clamped (results discarded), out-of-range k reads a zero page.
#include "common.h"

#define BM 128
#define BN 128
#define BK 64
#define TILE_BYTES (BM * BK * 2)  // 16 KiB per operand tile
#define STAGE_BYTES (2 * TILE_BYTES)
#define NSTAGE 2
#define EPI_PITCH 68  // floats; epilogue staging row pitch (272 B)
#define LDS_BYTES (4 * 64 * EPI_PITCH * 4)  // 69632 >= NSTAGE*STAGE_BYTES

__device__ __attribute__((aligned(16))) uint32_t g_ca_zero_page[4];

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const void* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}

// ---- per-lane loader state -------------------------------------------------------------
struct KMajorLoader {  // operand stored [row][k], k contiguous
  const char* p[4];    // current source (advanced BK elements per step)
  int kc[4];           // element offset of this lane's chunk inside the k-step
  __device__ __forceinline__ void init(const __bf16* base, int64_t ld, int row0, int nrows,
                                       int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = (wave * 4 + i) * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      int rr = row0 + r;
      rr = rr < nrows ? rr : nrows - 1;
      kc[i] = c * 8;
      p[i] = (const char*)(base + (int64_t)rr * ld + c * 8);
    }
  }
  __device__ __forceinline__ void issue(char* tile, int wave, int k0, int K) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const void* src = (k0 + kc[i] < K) ? (const void*)p[i] : (const void*)g_ca_zero_page;
      glds16(src, tile + (wave * 4 + i) * 1024);
      p[i] += BK * 2;
    }
  }
};

struct MNMajorLoader {  // operand stored [k][mn], mn contiguous; k rows may be segmented
  const char* colp[4];
  int t[4];
  int seg[4];
  int64_t ld, segstride;
  int kseg;
  __device__ __forceinline__ void init(const __bf16* base, int64_t ld_, int kseg_,
                                       int64_t segstride_, int col0, int ncols, int wave,
                                       int lane) {
    ld = ld_;
    kseg = kseg_;
    segstride = segstride_;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kr = (wave * 4 + i) * 4 + (lane >> 4);
      const int swz = (kr & 3) | (((kr >> 3) & 1) << 2);
      const int c = (lane & 15) ^ (swz << 1);
      int cc = col0 + c * 8;
      const int nc8 = (ncols + 7) & ~7;  // rows are readable up to ncols rounded up to 8
      cc = cc <= nc8 - 8 ? cc : nc8 - 8;
      colp[i] = (const char*)(base + cc);
      if (kseg > 0) {
        seg[i] = kr / kseg;
        t[i] = kr % kseg;
      } else {
        seg[i] = 0;
        t[i] = kr;
      }
    }
  }
  __device__ __forceinline__ void issue(char* tile, int wave, int lane, int k0, int K) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kr = (wave * 4 + i) * 4 + (lane >> 4);
      const int64_t off = (int64_t)seg[i] * segstride + (int64_t)t[i] * ld;
      const void* src =
          (k0 + kr < K) ? (const void*)(colp[i] + off * 2) : (const void*)g_ca_zero_page;
      glds16(src, tile + (wave * 4 + i) * 1024);
      t[i] += BK;
      if (kseg > 0) {
        while (t[i] >= kseg) {
          t[i] -= kseg;
          seg[i] += 1;
        }
      }
    }
  }
};

// ---- fragment reads ----------------------------------------------------------------------
// KMAJOR tile: 16 rows starting at rb, k-step s (32 k): lane gets row rb+(lane&15),
// k = 32 s + 8 (lane>>4) .. +7.
__device__ __forceinline__ bf16x8_t frag_kmajor(const char* tile, int rb, int s, int lane) {
  const int r = rb + (lane & 15);
  const int c = (4 * s + (lane >> 4)) ^ ((r >> 1) & 7);
  return *(const bf16x8_t*)(tile + r * 128 + c * 16);
}
// MNMAJOR tile: 16 columns starting at cb (multiple of 16), k-step s: two transposed reads.
__device__ __forceinline__ bf16x8_t frag_mnmajor(const char* tile, int cb, int s, int lane) {
  const int g = lane >> 4;
  const int q = (lane & 15) >> 2;
  const int p = lane & 3;
  const int kr = 32 * s + 8 * g + q;
  const int swz = q | ((g & 1) << 2);
  const int c = ((cb >> 3) + (p >> 1)) ^ (swz << 1);
  const char* a0 = tile + kr * 256 + c * 16 + (p & 1) * 8;
  s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4_t*)(lptr_t)a0);
  s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4_t*)(lptr_t)(a0 + 4 * 256));
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

template <int AL, int BL>
__global__ __launch_bounds__(256) void ca_gemm_kernel(const CaGemmDesc d) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware tile order: consecutive logical tiles (sharing an A row-panel) are dealt to
  // the same XCD (blocks b and b+8 share an L2).  Bijective for any grid size.
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x;
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int ntn = (d.N + BN - 1) / BN;
  const int tm = bid / ntn, tn = bid % ntn;
  const int m0 = tm * BM, n0 = tn * BN;
  const int z = blockIdx.z;
  const int z1 = z / d.batch2, z2 = z % d.batch2;

  const __bf16* A = (const __bf16*)d.A + z1 * d.sA1 + z2 * d.sA2;
  const __bf16* B = (const __bf16*)d.B + z1 * d.sB1 + z2 * d.sB2;

  KMajorLoader la_k, lb_k;
  MNMajorLoader la_m, lb_m;
  if (AL == CA_KMAJOR)
    la_k.init(A, d.lda, m0, d.M, wave, lane);
  else
    la_m.init(A, d.lda, d.a_kseg, d.a_kseg_stride, m0, d.M, wave, lane);
  if (BL == CA_KMAJOR)
    lb_k.init(B, d.ldb, n0, d.N, wave, lane);
  else
    lb_m.init(B, d.ldb, d.b_kseg, d.b_kseg_stride, n0, d.N, wave, lane);

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int K = d.K;
  const int nk = (K + BK - 1) / BK;

  auto issue_stage = [&](int kt) {
    char* st = smem + (kt & 1) * STAGE_BYTES;
    if (AL == CA_KMAJOR)
      la_k.issue(st, wave, kt * BK, K);
    else
      la_m.issue(st, wave, lane, kt * BK, K);
    if (BL == CA_KMAJOR)
      lb_k.issue(st + TILE_BYTES, wave, kt * BK, K);
    else
      lb_m.issue(st + TILE_BYTES, wave, lane, kt * BK, K);
  };

  issue_stage(0);
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) {
      issue_stage(kt + 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // 8 LDS-DMA per stage per wave
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    const char* ta = smem + (kt & 1) * STAGE_BYTES;
    const char* tb = ta + TILE_BYTES;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8_t af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        af[i] = (AL == CA_KMAJOR) ? frag_kmajor(ta, wm * 64 + i * 16, s, lane)
                                  : frag_mnmajor(ta, wm * 64 + i * 16, s, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        bfr[j] = (BL == CA_KMAJOR) ? frag_kmajor(tb, wn * 64 + j * 16, s, lane)
                                   : frag_mnmajor(tb, wn * 64 + j * 16, s, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
    // every wave's fragment reads of this stage have returned before it is re-staged
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

  // ---- epilogue --------------------------------------------------------------------------
  // Each wave parks its 64x64 fp32 tile in LDS (row pitch 68 floats: conflict-free b128
  // writes), then walks it 4 rows x 64 columns at a time in a rolled loop so the generic
  // (runtime-selected) epilogue is emitted once and every row is stored as one contiguous
  // 128-B (bf16) / 256-B (fp32) segment.
  {
    float* wt = (float*)smem + wave * (64 * EPI_PITCH);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *(f32x4_t*)(wt + (i * 16 + (lane & 15)) * EPI_PITCH + j * 16 + 4 * (lane >> 4)) =
            acc[i][j];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private region: no barrier

    const int M = d.M, N = d.N;
    const int64_t zoffC = z1 * d.sC1 + z2 * d.sC2;
    const int64_t zoffR = z1 * d.sR1 + z2 * d.sR2;
    const bool vec_ok = ((d.ldc & 3) == 0) && ((zoffC & 3) == 0);
    const float keep_scale = d.dropout_p > 0.f ? 1.f / (1.f - d.dropout_p) : 1.f;
    const int nb = n0 + wn * 64 + 4 * (lane & 15);
    const int nvalid = (N - nb) < 4 ? (N - nb) : 4;
    float bias4[4] = {0.f, 0.f, 0.f, 0.f};
    if (d.bias) {
      const float* bz = d.bias + z1 * d.sBias1 + z2 * d.sBias2;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (e < nvalid) bias4[e] = bz[nb + e];
    }
#pragma unroll 1
    for (int it = 0; it < 16; ++it) {
      const int ml = it * 4 + (lane >> 4);
      const int m = m0 + wm * 64 + ml;
      if (m >= M || nvalid <= 0) continue;
      const f32x4_t a4 = *(const f32x4_t*)(wt + ml * EPI_PITCH + 4 * (lane & 15));
      float v[4], v2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = a4[e] * d.alpha + bias4[e];
      const int64_t coff = zoffC + (int64_t)m * d.ldc + nb;
      const int64_t roff = zoffR + (int64_t)m * d.ldr + nb;
      if (d.epilogue == CA_EPI_GELU || d.epilogue == CA_EPI_GELU_RESIDUAL) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float g = gelu_erf(v[e]);
          if (d.dropout_p > 0.f) {
            const uint64_t idx = ((uint64_t)z * M + m) * (uint64_t)N + (nb + e);
            g = ca_dropout_keep(d.dropout_seed, idx, d.dropout_p) ? g * keep_scale : 0.f;
          }
          v2[e] = g;
        }
        if (d.epilogue == CA_EPI_GELU_RESIDUAL) {
          const unsigned short* R = (const unsigned short*)d.R + roff;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (e < nvalid) v2[e] += bf2f(R[e]);
        }
      } else if (d.epilogue == CA_EPI_RESIDUAL) {
        const unsigned short* R = (const unsigned short*)d.R + roff;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (e < nvalid) v[e] += bf2f(R[e]);
      } else if (d.epilogue == CA_EPI_DGELU) {
        const unsigned short* R = (const unsigned short*)d.R + roff;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (e < nvalid) {
            float dg = dgelu_erf(bf2f(R[e]));
            if (d.dropout_p > 0.f) {
              const uint64_t idx = ((uint64_t)z * M + m) * (uint64_t)N + (nb + e);
              dg = ca_dropout_keep(d.dropout_seed, idx, d.dropout_p) ? dg * keep_scale : 0.f;
            }
            v[e] *= dg;
          }
      }
      if (d.out_f32) {
        float* C = (float*)d.C + coff;
        if (d.accumulate) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (e < nvalid) v[e] += C[e];
        }
        if (nvalid == 4 && vec_ok) {
          *(f32x4_t*)C = (f32x4_t){v[0], v[1], v[2], v[3]};
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (e < nvalid) C[e] = v[e];
        }
      } else if (d.C) {
        unsigned short* C = (unsigned short*)d.C + coff;
        if (d.accumulate) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (e < nvalid) v[e] += bf2f(C[e]);
        }
        if (nvalid == 4 && vec_ok) {
          *(u16x4_t*)C = (u16x4_t){f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (e < nvalid) C[e] = f2bf(v[e]);
        }
      }
      if ((d.epilogue == CA_EPI_GELU || d.epilogue == CA_EPI_GELU_RESIDUAL) && d.C2) {
        unsigned short* C2 = (unsigned short*)d.C2 + coff;
        if (nvalid == 4 && vec_ok) {
          *(u16x4_t*)C2 = (u16x4_t){f2bf(v2[0]), f2bf(v2[1]), f2bf(v2[2]), f2bf(v2[3])};
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (e < nvalid) C2[e] = f2bf(v2[e]);
        }
      }
    }
  }
}

extern "C" int ca_gemm_bf16(const CaGemmDesc* desc, void* stream) {
  CA_CHECK_ARG(desc != nullptr, "ca_gemm_bf16: null descriptor");
  const CaGemmDesc& d = *desc;
  CA_CHECK_ARG(d.A && d.B &&
                   (d.C || ((d.epilogue == CA_EPI_GELU || d.epilogue == CA_EPI_GELU_RESIDUAL) && d.C2)),
               "ca_gemm_bf16: null operand");
  CA_CHECK_ARG(d.M > 0 && d.N > 0 && d.K > 0, "ca_gemm_bf16: bad shape %d %d %d", d.M, d.N,
               d.K);
  CA_CHECK_ARG(d.batch1 > 0 && d.batch2 > 0, "ca_gemm_bf16: bad batch");
  CA_CHECK_ARG((d.lda % 8) == 0 && (d.ldb % 8) == 0, "ca_gemm_bf16: lda/ldb must be multiples of 8");
  CA_CHECK_ARG((d.sA1 % 8) == 0 && (d.sA2 % 8) == 0 && (d.sB1 % 8) == 0 && (d.sB2 % 8) == 0,
               "ca_gemm_bf16: batch strides must be multiples of 8");
  CA_CHECK_ARG(((uintptr_t)d.A % 16) == 0 && ((uintptr_t)d.B % 16) == 0,
               "ca_gemm_bf16: A/B must be 16-byte aligned");
  CA_CHECK_ARG(d.a_kseg >= 0 && d.b_kseg >= 0, "ca_gemm_bf16: negative kseg");
  if (d.epilogue == CA_EPI_RESIDUAL || d.epilogue == CA_EPI_DGELU || d.epilogue == CA_EPI_GELU_RESIDUAL)
    CA_CHECK_ARG(d.R != nullptr, "ca_gemm_bf16: epilogue needs R");
  CA_CHECK_ARG(d.dropout_p >= 0.f && d.dropout_p < 1.f, "ca_gemm_bf16: bad dropout_p");

  const int ntm = (d.M + BM - 1) / BM, ntn = (d.N + BN - 1) / BN;
  dim3 grid(ntm * ntn, 1, d.batch1 * d.batch2);
  dim3 block(256);
  const size_t lds = LDS_BYTES;
  hipStream_t s = (hipStream_t)stream;
  if (d.a_layout == CA_KMAJOR && d.b_layout == CA_KMAJOR)
    hipLaunchKernelGGL((ca_gemm_kernel<CA_KMAJOR, CA_KMAJOR>), grid, block, lds, s, d);
  else if (d.a_layout == CA_KMAJOR && d.b_layout == CA_MNMAJOR)
    hipLaunchKernelGGL((ca_gemm_kernel<CA_KMAJOR, CA_MNMAJOR>), grid, block, lds, s, d);
  else if (d.a_layout == CA_MNMAJOR && d.b_layout == CA_KMAJOR)
    hipLaunchKernelGGL((ca_gemm_kernel<CA_MNMAJOR, CA_KMAJOR>), grid, block, lds, s, d);
  else if (d.a_layout == CA_MNMAJOR && d.b_layout == CA_MNMAJOR)
    hipLaunchKernelGGL((ca_gemm_kernel<CA_MNMAJOR, CA_MNMAJOR>), grid, block, lds, s, d);
  else {
    ca_set_error("ca_gemm_bf16: bad layout");
    return CA_ERR_ARG;
  }
  CA_CHECK_LAUNCH("ca_gemm_bf16");
  return CA_OK;
}
