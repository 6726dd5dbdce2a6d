// Microbenchmark: how fast does one CU take operand tiles in through LDS-DMA (global_load_lds_dwordx4) as a function of
// the bytes it keeps in flight?  One 512-thread workgroup per CU streams 1-KiB pieces from a buffer that is re-read by
// every workgroup of an XCD (L2 hits, like GEMM operand panels), NINF pieces per wave in flight, no compute.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/ingest.hip -o tools/micro/ingest && tools/micro/ingest
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int NINF>
__global__ __launch_bounds__(512) void ingest_kernel(const char* __restrict__ src, size_t span, int iters, long long* cycles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // every workgroup of an XCD walks the same addresses (offset by wave): L2-resident after the first pass
  const char* base = src + (size_t)wave * 1024 + lane * 16;
  const uint32_t l0 = (uint32_t)(uintptr_t)(lptr_t)smem + wave * (NINF * 1024);
  size_t off = 0;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < NINF; ++i) {  // fill the pipeline
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(base + off), "s"(l0 + i * 1024) : "memory", "m0");
    off = (off + 8192) % span;
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NINF; ++i) {
      asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NINF - 1) : "memory");
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(base + off), "s"(l0 + i * 1024) : "memory", "m0");
      off = (off + 8192) % span;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int NINF>
void run(const char* src, size_t span, long long* dcyc, int ncu) {
  const int iters = 400;
  hipFuncSetAttribute((const void*)ingest_kernel<NINF>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * NINF * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(ingest_kernel<NINF>, dim3(ncu), dim3(512), 8 * NINF * 1024, 0, src, span, iters, dcyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> c(ncu);
  hipMemcpy(c.data(), dcyc, ncu * sizeof(long long), hipMemcpyDeviceToHost);
  double avg = 0;
  for (auto v : c) avg += (double)v;
  avg /= ncu;
  const double bytes = (double)(iters + 1) * NINF * 8 * 1024;  // per workgroup
  printf("in flight %3d KiB/CU (span %4zu MiB): %6.1f us  %5.1f B/clk/CU (s_memtime)  %6.2f TB/s chip\n", NINF * 8, span >> 20,
         ms * 1e3, bytes / avg, bytes * ncu / (ms * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
  const int ncu = 256;
  const size_t cap = (size_t)1 << 30;
  char* src;
  hipMalloc(&src, cap);
  hipMemset(src, 1, cap);
  long long* dcyc;
  hipMalloc(&dcyc, ncu * sizeof(long long));
  for (size_t span : {(size_t)2 << 20, (size_t)32 << 20, (size_t)512 << 20}) {
    run<2>(src, span, dcyc, ncu);
    run<4>(src, span, dcyc, ncu);
    run<8>(src, span, dcyc, ncu);
    run<12>(src, span, dcyc, ncu);
    run<16>(src, span, dcyc, ncu);
  }
  return 0;
}
