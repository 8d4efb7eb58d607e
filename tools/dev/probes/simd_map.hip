// which SIMD does wave w of a workgroup run on?  hipcc --offload-arch=gfx950 -O2 simd_map.hip -o simd_map && ./simd_map
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(unsigned* out) {
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 32 + (threadIdx.x >> 6)] = id;
  __syncthreads();
}
int main() {
  unsigned* d; hipMalloc(&d, 4 * 32 * 4);
  for (int nw : {8, 12, 16}) {
    hipMemset(d, 0xff, 4 * 32 * 4);
    hipLaunchKernelGGL(probe, dim3(4), dim3(nw * 64), 0, 0, d);
    unsigned h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 2; ++b) {
      printf("%2d waves, block %d: simd of wave 0..: ", nw, b);
      for (int w = 0; w < nw; ++w) printf("%u ", (h[b * 32 + w] >> 4) & 3);
      printf(" | wave slot: ");
      for (int w = 0; w < nw; ++w) printf("%u ", h[b * 32 + w] & 15);
      printf(" | cu %u\n", (h[b * 32] >> 8) & 15);
    }
  }
  return 0;
}
