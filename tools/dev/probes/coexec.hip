// Do the MFMAs of one wave and the VALU work of the other wave of a SIMD overlap?  8-wave workgroups, one per CU: waves 0..3
// run role A, waves 4..7 role B (0 idle, 1 MFMA 16x16x32 f16 chain over 16 accumulators, 2 VALU fma/exp mix, 3 both interleaved).
// hipcc --offload-arch=gfx950 -O3 coexec.hip -o coexec && ./coexec
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(4))) float f4;

__device__ __forceinline__ void do_mfma(f4 (&acc)[16], h8 a, h8 b) {
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
}
__device__ __forceinline__ void do_valu(float (&x)[16]) {
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[i], 0.999f, -0.001f));
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(x[i], 1.001f, 0.25f);
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(x[i], 0.5f, 0.125f);
}

__global__ __launch_bounds__(512) void k(float* out, int iters, int roleA, int roleB, int split) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool isA = split == 0 ? wave < 4 : (wave & 1) == 0;     // split 1: even / odd waves instead of low / high
  const int role = isA ? roleA : roleB;
  f4 acc[16];
  float x[16];
  for (int i = 0; i < 16; ++i) { acc[i] = (f4){0, 0, 0, 0}; x[i] = threadIdx.x * 1e-3f + i; }
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 1e-3f); b[i] = (_Float16)1.f; }
  for (int it = 0; it < iters; ++it) {
    if (role == 1) do_mfma(acc, a, b);
    else if (role == 2) do_valu(x);
    else if (role == 3) { do_mfma(acc, a, b); do_valu(x); }
  }
  float sum = 0;
  for (int i = 0; i < 16; ++i) sum += acc[i][0] + x[i];
  if (sum == 12345.f) out[threadIdx.x] = sum;
}

int main() {
  float* d; hipMalloc(&d, 1 << 16);
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int cfg[][3] = {{1, 0, 0}, {0, 1, 0}, {2, 0, 0}, {0, 2, 0}, {1, 1, 0}, {2, 2, 0}, {1, 2, 0}, {2, 1, 0}, {3, 3, 0}, {3, 0, 0},
                        {1, 2, 1}, {1, 1, 1}, {2, 2, 1}};
  for (auto& c : cfg) {
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, 10, c[0], c[1], c[2]);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, iters, c[0], c[1], c[2]);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("roleA %d roleB %d split %d: %.1f ns per iteration (16 MFMAs = 256 cycles at peak; 48 VALU incl. 16 exp)\n", c[0], c[1], c[2], ms * 1e6 / iters);
  }
  return 0;
}
