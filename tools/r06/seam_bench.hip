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
// LDS-DMA with sc1: 16 bytes per lane from g to lds_base + 16 * lane (polls that land in LDS leave no register in flight)
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ __forceinline__ void glds16_sc1(const void* g, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off sc1"
               :
               : "v"(g), "s"(__builtin_amdgcn_readfirstlane(lds_base))
               : "memory", "m0");
}
template <int N>
__device__ __forceinline__ void vm_keep() {
  if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  if (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  if (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
}
// the wait and the first use of an asm-loaded register must be ordered by a data dependence: the compiler schedules a use
// of the (to it, already defined) register above a bare s_waitcnt asm (guide 5.7)
__device__ __forceinline__ void vm0_tie(u32x4& v) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(v) : : "memory"); }

// mode 0: flags + payload gather; 1: flags only; 2: acquire fence instead of sc1 loads (plain loads)
// POLL: 0 = one sc1 register load at a time; N > 0: N LDS-DMA sc1 polls in flight, landing in N LDS slots (round 6, late)
// MODE 3 (late in round 6): every seam's payload at an address not touched before in this launch (FRESH regions, the launch
// runs FRESH - 1 seams), gathered with PLAIN loads and no fence: a line that no workgroup of the XCD has touched since the
// launch's own acquire cannot be stale in its L2, and after the first miss the XCD's other 31 workgroups hit in L2.
#define FRESH 64
template <int MODE, int POLL = 0>
__global__ __launch_bounds__(256) void seam_kernel(unsigned* flags, unsigned* pay0, unsigned* pay1, int pay_words, int nseam,
                                                   unsigned* err, unsigned* tmo, long long* cyc, unsigned* dbg, int R = 1, int RP = 1, unsigned salt = 0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int G = gridDim.x, w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned bad = 0;
  const long long t0 = wall_clock64();
  for (int s = 1; s <= nseam; ++s) {
    unsigned* pay = (s & 1) ? pay1 : pay0;
    if (MODE == 3) pay = pay0 + (size_t)(s % FRESH) * ((size_t)G * pay_words);  // an address is written once per FRESH seams
    // publish my slice: pay_words words, value = s * 0x10001 + w * 977 + i; RP replicas of the payload (each G * pay_words
    // words), a consumer gathers from replica w % RP: RP x fewer readers per line
    const size_t rstride = (size_t)G * pay_words;
    for (int i = tid * 4; i < pay_words; i += 1024) {
      u32x4 v;
      for (int e = 0; e < 4; ++e) v[e] = (unsigned)s * 0x10001u + (unsigned)w * 977u + (unsigned)(i + e) + salt * 0x9E3779B1u;
      for (int rp = 0; rp < RP; ++rp) st16_sc1(pay + rp * rstride + (size_t)w * pay_words + i, v);
    }
    vm0();
    __syncthreads();
    // R replicas of the progress words (each [G] words, 1 KiB apart): ONE wave instruction, lane i stores replica i's word;
    // a consumer polls replica w % R only (the guide's replicated form: R x fewer readers per line)
    if (tid < R) st4_sc1(flags + tid * 1024 + w, (unsigned)s);
    // poll: lane i of wave 0 looks at flags 4i .. 4i+3
    if (wave == 0 && POLL > 0) {
      // slots: smem[0 .. POLL KiB); poll p lands in slot p % POLL; lanes beyond G / 4 look at flags[0..3] again (harmless)
      const unsigned slot0 = (unsigned)(size_t)(lptr_t)smem;
      const unsigned* src = flags + (lane * 4 < G ? lane * 4 : 0);
      unsigned issued = 0, spins = 0;
      for (; issued < (unsigned)POLL; ++issued) glds16_sc1(src, slot0 + (issued % POLL) * 1024u);
      for (unsigned p = 0;; ++p) {
        vm_keep<POLL - 1>();  // poll p has landed
        const u32x4 f = *(volatile u32x4*)(smem + (p % POLL) * 1024 + lane * 16);
        bool ok = true;
        for (int e = 0; e < 4; ++e) ok &= (lane * 4 + e >= G) || f[e] >= (unsigned)s;
        if (__all(ok)) break;
        if (++spins > 2000000u) { if (lane == 0) atomicAdd(tmo, 1u); break; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        glds16_sc1(src, slot0 + (issued % POLL) * 1024u);  // into the slot just read
        ++issued;
      }
    } else if (wave == 0) {
      unsigned spins = 0;
      const unsigned* myflags = flags + (w % R) * 1024;
      for (;;) {
        u32x4 f = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
        if (lane * 4 < G) f = ld16_sc1(myflags + lane * 4);
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
      pay += (w % RP) * rstride;
      for (int i0 = tid * 4; i0 < total; i0 += 1024 * 8) {
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = i0 + u * 1024;
          if (i < total) {
            if (MODE == 2 || MODE == 3) v[u] = *(const u32x4*)(pay + i); else v[u] = ld16_sc1(pay + i);
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
            for (int e = 0; e < 4; ++e) { const unsigned ex = (unsigned)s * 0x10001u + (unsigned)pw * 977u + (unsigned)(pi + e) + salt * 0x9E3779B1u; if (v[u][e] != ex) { if (!bad && atomicAdd(dbg, 1u) < 8) { unsigned* d = dbg + 8 + 8 * (atomicAdd(dbg + 1, 1u) & 7); d[0] = s; d[1] = w; d[2] = i + e; d[3] = v[u][e]; d[4] = ex; d[5] = tid; } ++bad; } }
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
  CHECK(hipMalloc(&flags, 32 * 4096));
  CHECK(hipMalloc(&pay0, (size_t)G * maxpay * 4 * 64));
  CHECK(hipMalloc(&pay1, (size_t)G * maxpay * 4 * 8));
  CHECK(hipMalloc(&err, 4)); CHECK(hipMalloc(&tmo, 4)); CHECK(hipMalloc(&cyc, G * 8)); CHECK(hipMalloc(&dbg, 4096));
  const int nseam = 2000;
  CHECK(hipFuncSetAttribute((const void*)seam_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  CHECK(hipFuncSetAttribute((const void*)seam_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  CHECK(hipFuncSetAttribute((const void*)seam_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  CHECK(hipFuncSetAttribute((const void*)seam_kernel<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  CHECK(hipFuncSetAttribute((const void*)seam_kernel<0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  CHECK(hipFuncSetAttribute((const void*)seam_kernel<0, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  CHECK(hipFuncSetAttribute((const void*)seam_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  CHECK(hipFuncSetAttribute((const void*)seam_kernel<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  CHECK(hipFuncSetAttribute((const void*)seam_kernel<1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  for (int poll = 1; poll <= 3; ++poll)
    for (int mode = 0; mode < 2; ++mode)
      for (int pw : {16, 128}) {
        if (mode == 1 && pw != 16) continue;
        for (int rep = 0; rep < 2; ++rep) {
          CHECK(hipMemset(flags, 0, 32 * 4096)); CHECK(hipMemset(err, 0, 4)); CHECK(hipMemset(tmo, 0, 4)); CHECK(hipMemset(dbg, 0, 4096));
          hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
          CHECK(hipEventRecord(e0));
#define RUN(M, P) seam_kernel<M, P><<<G, 256, 96 * 1024>>>(flags, pay0, pay1, pw, nseam, err, tmo, cyc, dbg)
          if (mode == 0 && poll == 1) RUN(0, 1);
          if (mode == 0 && poll == 2) RUN(0, 2);
          if (mode == 0 && poll == 3) RUN(0, 3);
          if (mode == 1 && poll == 1) RUN(1, 1);
          if (mode == 1 && poll == 2) RUN(1, 2);
          if (mode == 1 && poll == 3) RUN(1, 3);
          CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
          float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
          unsigned herr, htmo; CHECK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&htmo, tmo, 4, hipMemcpyDeviceToHost));
          printf("LDS-DMA polls, %d in flight: mode %d  pay/WG %5d B : %.3f us per seam   stale words %u  timeouts %u\n", poll, mode,
                 pw * 4, ms * 1000.f / nseam, herr, htmo);
        }
      }
  CHECK(hipFuncSetAttribute((const void*)seam_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  for (int pw : {16, 32, 128, 512})
    for (int rep = 0; rep < 6; ++rep) {  // the same addresses again in every launch, with other values (seed = launch)
      CHECK(hipMemset(flags, 0, 32 * 4096)); CHECK(hipMemset(err, 0, 4)); CHECK(hipMemset(tmo, 0, 4)); CHECK(hipMemset(dbg, 0, 4096));
      hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
      CHECK(hipEventRecord(e0));
      const int ns3 = FRESH - 1;
      seam_kernel<3><<<G, 256, 96 * 1024>>>(flags, pay0, pay1, pw, ns3, err, tmo, cyc, dbg, 8, 1, (unsigned)(rep + 1 + 16 * pw));
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      unsigned herr, htmo; CHECK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&htmo, tmo, 4, hipMemcpyDeviceToHost));
      printf("fresh addresses + plain loads (8 flag replicas): pay/WG %5d B  gather %4d KB : %.3f us per seam   stale words %u  timeouts %u\n",
             pw * 4, G * pw * 4 / 1024, ms * 1000.f / ns3, herr, htmo);
    }
  for (int RP : {1, 2, 4, 8})
    for (int pw : {32, 128}) {
      for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipMemset(flags, 0, 32 * 4096)); CHECK(hipMemset(err, 0, 4)); CHECK(hipMemset(tmo, 0, 4)); CHECK(hipMemset(dbg, 0, 4096));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        seam_kernel<0><<<G, 256, 96 * 1024>>>(flags, pay0, pay1, pw, nseam, err, tmo, cyc, dbg, 8, RP);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned herr, htmo; CHECK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&htmo, tmo, 4, hipMemcpyDeviceToHost));
        printf("8 flag replicas, %d payload replicas: pay/WG %5d B  gather %4d KB : %.3f us per seam   stale words %u  timeouts %u\n", RP,
               pw * 4, G * pw * 4 / 1024, ms * 1000.f / nseam, herr, htmo);
      }
    }
  for (int R : {2, 4, 8, 16, 32})
    for (int mode = 0; mode < 2; ++mode) {
      const int pw = 16;
      for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipMemset(flags, 0, 32 * 4096)); CHECK(hipMemset(err, 0, 4)); CHECK(hipMemset(tmo, 0, 4)); CHECK(hipMemset(dbg, 0, 4096));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        if (mode == 0) seam_kernel<0><<<G, 256, 96 * 1024>>>(flags, pay0, pay1, pw, nseam, err, tmo, cyc, dbg, R);
        if (mode == 1) seam_kernel<1><<<G, 256, 96 * 1024>>>(flags, pay0, pay1, pw, nseam, err, tmo, cyc, dbg, R);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned herr, htmo; CHECK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&htmo, tmo, 4, hipMemcpyDeviceToHost));
        printf("%2d replicas of the progress words: mode %d  pay/WG %5d B : %.3f us per seam   stale words %u  timeouts %u\n", R, mode,
               pw * 4, ms * 1000.f / nseam, herr, htmo);
      }
    }
  for (int mode = 0; mode < 3; ++mode)
    for (int pw : {16, 32, 128, 512}) {  // bytes per WG = 4 pw: 64 B .. 2 KB; gathered = G * that: 16 KB .. 512 KB
      if (mode == 1 && pw != 16) continue;
      for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipMemset(flags, 0, 32 * 4096)); CHECK(hipMemset(err, 0, 4)); CHECK(hipMemset(tmo, 0, 4)); CHECK(hipMemset(dbg, 0, 4096));
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
