// Round 6: what s_memtime counts and how fast a lone wave issues dependent VALU work (the persistent decode kernel is
// one wave per SIMD): ticks of s_memtime and of the 100 MHz s_memrealtime around 4096 dependent v_fma_f32.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned long long* out, float* sink, int spin) {
  float x = threadIdx.x * 1e-9f;
  for (int rep = 0; rep < 3; ++rep) {
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int i = 0; i < 4096; ++i) x = __builtin_fmaf(x, 0.999f, 1e-7f);
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[rep * 2] = t1 - t0; out[rep * 2 + 1] = r1 - r0; }
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(10);
  }
  sink[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
int main() {
  unsigned long long* d; float* s; hipMalloc(&d, 64); hipMalloc(&s, 256 * 256 * 4);
  for (int grid : {1, 256}) for (int spin : {0, 2000}) {
    k<<<grid, 256>>>(d, s, spin); hipDeviceSynchronize();
    unsigned long long h[6]; hipMemcpy(h, d, 48, hipMemcpyDeviceToHost);
    printf("grid %3d spin %4d: 4096 dependent fma: %llu / %llu / %llu memtime ticks; %llu / %llu / %llu realtime ticks (10 ns)\n", grid, spin,
           h[0], h[2], h[4], h[1], h[3], h[5]);
  }
  return 0;
}
