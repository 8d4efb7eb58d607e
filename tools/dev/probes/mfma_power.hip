// Sustained MFMA rate under the package power cap: register-only loops of v_mfma_f32_16x16x32_f16 vs v_mfma_f32_32x32x16_f16 on
// random operands (the data pattern sets the power), 16 waves per CU, ~2 s per variant; sclk / power are read with rocm-smi
// from the wrapper script.  hipcc --offload-arch=gfx950 -O3 mfma_power.hip -o mfma_power && ./mfma_power
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int KIND>
__global__ __launch_bounds__(1024) void k(const h8* __restrict__ in, float* out, int iters) {
  const int t = threadIdx.x + blockIdx.x * 1024;
  h8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = in[(t * 8 + i) & 0xffff]; b[i] = in[(t * 8 + 4 + i) & 0xffff]; }
  float s = 0.f;
  if (KIND == 0) {
    f4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (f4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
  } else {
    f16v acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i + 2 * kk], b[j + 2 * kk], acc[i][j], 0, 0, 0);
    }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][15];
  }
  if (s == 12345.678f) out[t] = s;
}

template <int KIND>
void run(const h8* in, float* out, const char* name, double flop_per_iter) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(1024), 0, 0, in, out, iters);
  hipEventRecord(e0);
  const int reps = 400;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(1024), 0, 0, in, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = flop_per_iter * iters * 16.0 * 256.0 * reps;     // per wave -> 16 waves x 256 CUs
  printf("%s: %.1f ms, %.1f TFLOP/s\n", name, ms, flops / ms / 1e9);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int n = 1 << 16;
  h8* hin = (h8*)malloc(n * sizeof(h8));
  srand(1);
  const bool zeros = argc > 1 && atoi(argv[1]) == 0;
  for (int i = 0; i < n; ++i) for (int e = 0; e < 8; ++e) hin[i][e] = zeros ? (_Float16)0.f : (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.f);
  h8* din; float* dout;
  hipMalloc(&din, n * sizeof(h8)); hipMalloc(&dout, 1 << 22);
  hipMemcpy(din, hin, n * sizeof(h8), hipMemcpyHostToDevice);
  for (int r = 0; r < 2; ++r) {
    run<0>(din, dout, "16x16x32 f16 (16 per iter)", 16 * 2.0 * 16 * 16 * 32);
    run<1>(din, dout, "32x32x16 f16 ( 8 per iter)", 8 * 2.0 * 32 * 32 * 16);
  }
  return 0;
}
