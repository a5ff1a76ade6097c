// probes/permlane_swap.hip — lane mapping of v_permlane32_swap_b32 on gfx950 (used by the register epilogue of gemm_w8.hip)
//   hipcc --offload-arch=gfx950 -O3 -o probes/permlane_swap probes/permlane_swap.hip && probes/permlane_swap
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* o) {
  unsigned a = 1000 + threadIdx.x, b = 2000 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  o[threadIdx.x] = r[0];
  o[64 + threadIdx.x] = r[1];
}
int main() {
  unsigned* d;
  unsigned h[128];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("first result : lane 0 %u  lane 31 %u  lane 32 %u  lane 63 %u\n", h[0], h[31], h[32], h[63]);
  printf("second result: lane 0 %u  lane 31 %u  lane 32 %u  lane 63 %u\n", h[64], h[95], h[96], h[127]);
  return 0;
}
