// which cross-lane primitive delivers lane ^ o ?  (round 6: the LayerNorm butterfly without ds_bpermute)
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__global__ void k(float* out) {
  const int lane = threadIdx.x;
  const float v = (float)lane;
  float* o = out + lane * 8;
  { unsigned x = __builtin_bit_cast(unsigned, v), y = x; asm volatile("" : "+v"(y));
    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    o[0] = __builtin_bit_cast(float, x); o[1] = __builtin_bit_cast(float, y); }
  { unsigned x = __builtin_bit_cast(unsigned, v), y = x; asm volatile("" : "+v"(y));
    asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    o[2] = __builtin_bit_cast(float, x); o[3] = __builtin_bit_cast(float, y); }
  o[4] = dpp<0x128>(v);
  o[5] = dpp<0x141>(dpp<0x1B>(v));
  o[6] = dpp<0x4E>(v);
  o[7] = dpp<0xB1>(v);
}
int main() {
  float* d; hipMalloc(&d, 64 * 8 * 4);
  k<<<1, 64>>>(d);
  float h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[8] = {"pl32 r0", "pl32 r1", "pl16 r0", "pl16 r1", "ror8", "hmirror(qp3210)", "qp2301", "qp1032"};
  for (int c = 0; c < 8; ++c) { printf("%-16s:", names[c]); for (int l = 0; l < 64; ++l) printf(" %2.0f", h[l * 8 + c]); printf("\n"); }
  return 0;
}
