// Round 6: what ds_read_b64_tr_b8 returns.  Every lane L of a wave points at its own 8 bytes (value = (L % 16) * 8 + b);
// prints, per lane of the first 16-lane group, the 8 bytes it receives as (source lane, source byte).
// build: hipcc --offload-arch=gfx950 -O2 tools/r06/tr8_test.hip -o /tmp/tr8_test
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void k(uint64_t* out) {
  __shared__ __attribute__((aligned(16))) unsigned char s[64 * 8];
  const int L = threadIdx.x;
  for (int b = 0; b < 8; ++b) s[L * 8 + b] = (unsigned char)((L % 16) * 8 + b);
  __syncthreads();
  const uint32_t a = (uint32_t)(uintptr_t)(lptr_t)(s + L * 8);
  uint64_t v;
  asm volatile("ds_read_b64_tr_b8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a));
  out[L] = v;
}
int main() {
  uint64_t* d;
  hipMalloc(&d, 64 * 8);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  uint64_t h[64];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int L = 0; L < 64; ++L) {
    if (L % 16 == 0) printf("-- lanes %d..%d\n", L, L + 15);
    printf("lane %2d:", L);
    for (int j = 0; j < 8; ++j) {
      const int v = (int)((h[L] >> (8 * j)) & 0xff);
      printf(" (%2d,%d)", v / 8, v % 8);
    }
    printf("\n");
    if (L == 15) { L = 47; }
  }
  return 0;
}
