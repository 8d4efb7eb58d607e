// Cost of a workgroup barrier round on gfx950: NW waves loop over `iters` rounds of {s_barrier; `work` dependent VALU ops}.
// hipcc --offload-arch=gfx950 -O3 barrier_cost.hip -o barrier_cost && ./barrier_cost
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int NW, bool SKEW>
__global__ __launch_bounds__(NW * 64) void k(float* out, int iters, int work) {
  extern __shared__ char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float x = threadIdx.x * 1e-3f;
  if (SKEW && wave >= NW / 2) __builtin_amdgcn_s_barrier();
  for (int i = 0; i < iters; ++i) {
    __builtin_amdgcn_s_barrier();
    for (int j = 0; j < work; ++j) x = __builtin_fmaf(x, 1.0001f, 0.5f);
    asm volatile("" : "+v"(x));
  }
  if (SKEW && wave < NW / 2) __builtin_amdgcn_s_barrier();
  if (x == 12345.f) out[threadIdx.x] = x;
}

template <int NW, bool SKEW>
void run(float* d, int lds, int work) {
  hipFuncSetAttribute((const void*)k<NW, SKEW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NW, SKEW>), dim3(256), dim3(NW * 64), lds, 0, d, 100, work);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NW, SKEW>), dim3(256), dim3(NW * 64), lds, 0, d, iters, work);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("waves %2d skew %d lds %6d work %3d: %.1f ns per round\n", NW, (int)SKEW, lds, work, ms * 1e6 / iters);
}

int main() {
  float* d; hipMalloc(&d, 1 << 16);
  for (int work : {0, 16, 64, 256}) {
    run<4, false>(d, 0, work);
    run<8, false>(d, 0, work);
    run<8, true>(d, 0, work);
    run<8, false>(d, 122 * 1024, work);
    run<12, false>(d, 0, work);
    run<16, false>(d, 0, work);
  }
  return 0;
}
