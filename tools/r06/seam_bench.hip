// Round 6 calibration: what an all-to-all seam costs inside ONE persistent launch on gfx950 (one 256-thread workgroup per CU).
// Each workgroup publishes `pay` bytes (write-through stores), drains, stores its progress word; wave 0 polls every
// workgroup's word with ONE dwordx4 sc1 load per lane; then all waves gather the whole payload (G * pay bytes) with sc1 loads
// and check every word.  Prints us per seam and the number of stale words seen.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ u32x4 ld16_sc1(const void* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void st16_sc1(void* p, u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st4_sc1(void* p, unsigned v) {
  asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// the wait and the first use of an asm-loaded register must be ordered by a data dependence: the compiler schedules a use
// of the (to it, already defined) register above a bare s_waitcnt asm (guide 5.7)
__device__ __forceinline__ void vm0_tie(u32x4& v) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(v) : : "memory"); }

// mode 0: flags + payload gather; 1: flags only; 2: acquire fence instead of sc1 loads (plain loads)
template <int MODE>
__global__ __launch_bounds__(256) void seam_kernel(unsigned* flags, unsigned* pay0, unsigned* pay1, int pay_words, int nseam,
                                                   unsigned* err, unsigned* tmo, long long* cyc, unsigned* dbg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int G = gridDim.x, w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned bad = 0;
  const long long t0 = wall_clock64();
  for (int s = 1; s <= nseam; ++s) {
    unsigned* pay = (s & 1) ? pay1 : pay0;
    // publish my slice: pay_words words, value = s * 0x10001 + w * 977 + i
    for (int i = tid * 4; i < pay_words; i += 1024) {
      u32x4 v;
      for (int e = 0; e < 4; ++e) v[e] = (unsigned)s * 0x10001u + (unsigned)w * 977u + (unsigned)(i + e);
      st16_sc1(pay + (size_t)w * pay_words + i, v);
    }
    vm0();
    __syncthreads();
    if (tid == 0) st4_sc1(flags + w, (unsigned)s);
    // poll: lane i of wave 0 looks at flags 4i .. 4i+3
    if (wave == 0) {
      unsigned spins = 0;
      for (;;) {
        u32x4 f = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
        if (lane * 4 < G) f = ld16_sc1(flags + lane * 4);
        vm0_tie(f);
        bool ok = true;
        for (int e = 0; e < 4; ++e) ok &= (lane * 4 + e >= G) || f[e] >= (unsigned)s;
        if (__all(ok)) break;
        if (++spins > 2000000u) { if (lane == 0) atomicAdd(tmo, 1u); break; }
        __builtin_amdgcn_s_sleep(2);
      }
      if (MODE == 2) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); vm0(); }
    }
    __syncthreads();
    if (MODE != 1) {
      // gather everybody's payload and check it
      const int total = G * pay_words;
      for (int i0 = tid * 4; i0 < total; i0 += 1024 * 8) {
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = i0 + u * 1024;
          if (i < total) {
            if (MODE == 2) v[u] = *(const u32x4*)(pay + i); else v[u] = ld16_sc1(pay + i);
          }
        }
        vm0();
#pragma unroll
        for (int u = 0; u < 8; ++u) vm0_tie(v[u]);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = i0 + u * 1024;
          if (i < total) {
            const int pw = i / pay_words, pi = i - pw * pay_words;
            for (int e = 0; e < 4; ++e) { const unsigned ex = (unsigned)s * 0x10001u + (unsigned)pw * 977u + (unsigned)(pi + e); if (v[u][e] != ex) { if (!bad && atomicAdd(dbg, 1u) < 8) { unsigned* d = dbg + 8 + 8 * (atomicAdd(dbg + 1, 1u) & 7); d[0] = s; d[1] = w; d[2] = i + e; d[3] = v[u][e]; d[4] = ex; d[5] = tid; } ++bad; } }
          }
        }
      }
    }
  }
  const long long t1 = wall_clock64();
  if (bad) atomicAdd(err, bad);
  if (tid == 0) cyc[w] = t1 - t0;
}

int main(int argc, char** argv) {
  int dev = 0, ncu = 0;
  CHECK(hipGetDevice(&dev));
  CHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
  int wc = 0;
  CHECK(hipDeviceGetAttribute(&wc, hipDeviceAttributeWallClockRate, dev));
  printf("CUs %d  wall clock %d kHz\n", ncu, wc);
  const int G = ncu;
  unsigned *flags, *pay0, *pay1, *err, *tmo;
  long long* cyc; unsigned* dbg;
  const int maxpay = 1024;  // words per workgroup
  CHECK(hipMalloc(&flags, 4096));
  CHECK(hipMalloc(&pay0, (size_t)G * maxpay * 4));
  CHECK(hipMalloc(&pay1, (size_t)G * maxpay * 4));
  CHECK(hipMalloc(&err, 4)); CHECK(hipMalloc(&tmo, 4)); CHECK(hipMalloc(&cyc, G * 8)); CHECK(hipMalloc(&dbg, 4096));
  const int nseam = 2000;
  CHECK(hipFuncSetAttribute((const void*)seam_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  CHECK(hipFuncSetAttribute((const void*)seam_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  CHECK(hipFuncSetAttribute((const void*)seam_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  for (int mode = 0; mode < 3; ++mode)
    for (int pw : {16, 32, 128, 512}) {  // bytes per WG = 4 pw: 64 B .. 2 KB; gathered = G * that: 16 KB .. 512 KB
      if (mode == 1 && pw != 16) continue;
      for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipMemset(flags, 0, 4096)); CHECK(hipMemset(err, 0, 4)); CHECK(hipMemset(tmo, 0, 4)); CHECK(hipMemset(dbg, 0, 4096));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        if (mode == 0) seam_kernel<0><<<G, 256, 96 * 1024>>>(flags, pay0, pay1, pw, nseam, err, tmo, cyc, dbg);
        if (mode == 1) seam_kernel<1><<<G, 256, 96 * 1024>>>(flags, pay0, pay1, pw, nseam, err, tmo, cyc, dbg);
        if (mode == 2) seam_kernel<2><<<G, 256, 96 * 1024>>>(flags, pay0, pay1, pw, nseam, err, tmo, cyc, dbg);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned herr, htmo; CHECK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&htmo, tmo, 4, hipMemcpyDeviceToHost));
        printf("mode %d  pay/WG %5d B  gather %4d KB : %.3f us per seam   stale words %u  timeouts %u\n", mode, pw * 4,
               G * pw * 4 / 1024, ms * 1000.f / nseam, herr, htmo);
        if (herr && rep == 0) { unsigned h[128]; CHECK(hipMemcpy(h, dbg, 512, hipMemcpyDeviceToHost));
          for (int k = 0; k < 4; ++k) printf("   s %u wg %u word %u got %08x expected %08x tid %u\n", h[8+8*k], h[9+8*k], h[10+8*k], h[11+8*k], h[12+8*k], h[13+8*k]); }
      }
    }
  return 0;
}
