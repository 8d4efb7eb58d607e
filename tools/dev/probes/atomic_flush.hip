// What the wgrad flush costs by access pattern: 256 workgroups (8 XCDs x 32 tiles of 384 x 192, 12 waves of 96 x 64 each) add
// their fp32 tile into ONE 3072 x 768 output (8 partial sums per element), as the XCD-partitioned wgrad does at its end.
//   A  fp32 atomics in the accumulator layout: a wave instruction = 4 rows x 64 B            (what csrc/wgrad.hip does)
//   B  fp32 atomics, a wave instruction = 1 row x 256 B (two whole 128-B lines)
//   C  plain dword stores in the accumulator layout into per-XCD slabs (no atomics)
//   D  plain 16-B stores into per-XCD slabs, a wave instruction = 4 rows x 256 B
//   E  D + a second kernel that sums the 8 slabs into the output (the whole cost of the slab route)
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics atomic_flush.hip -o atomic_flush && ./atomic_flush
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr int N = 3072, K = 768, TN = 384, TK = 192;

template <int MODE>
__global__ __launch_bounds__(768) void flush(float* out, float* slabs, float seed) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wn = wave / 3, wk = wave % 3;
  const int xcd = blockIdx.x & 7, tile = blockIdx.x >> 3;
  const int tn = tile / 4, tk = tile % 4;
  const int n0 = tn * TN + wn * 96, k0 = tk * TK + wk * 64;
  const int i16 = lane & 15, g4 = lane >> 4;
  float* dst = MODE >= 2 ? slabs + (long)xcd * N * K : out;
  if (MODE == 0 || MODE == 2) {
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* p = dst + (long)(n0 + i * 16 + 4 * g4 + r) * K + k0 + j * 16 + i16;
          const float v = seed + i + j + r;
          if (MODE == 0) atomicAdd(p, v); else *p = v;
        }
  } else if (MODE == 1) {
#pragma unroll 8
    for (int row = 0; row < 96; ++row) atomicAdd(dst + (long)(n0 + row) * K + k0 + lane, seed + row);
  } else {
#pragma unroll 8
    for (int it = 0; it < 24; ++it) {      // 4 rows x 64 columns per instruction, 16 B per lane
      const int row = it * 4 + g4;
      *(f32x4_t*)(dst + (long)(n0 + row) * K + k0 + i16 * 4) = (f32x4_t){seed, seed + 1, seed + 2, seed + row};
    }
  }
}

__global__ __launch_bounds__(256) void reduce8(const float* slabs, float* out) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= (long)N * K) return;
  f32x4_t s = *(const f32x4_t*)(out + i);
#pragma unroll
  for (int x = 0; x < 8; ++x) s += *(const f32x4_t*)(slabs + (long)x * N * K + i);
  *(f32x4_t*)(out + i) = s;
}

template <int MODE>
void run(const char* name, float* out, float* slabs, bool with_reduce = false) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(flush<MODE>, dim3(256), dim3(768), 0, 0, out, slabs, 1.f);
  hipEventRecord(e0);
  const int reps = 50;
  for (int r = 0; r < reps; ++r) {
    hipLaunchKernelGGL(flush<MODE>, dim3(256), dim3(768), 0, 0, out, slabs, 1.f);
    if (with_reduce) hipLaunchKernelGGL(reduce8, dim3(N * K / 1024), dim3(256), 0, 0, slabs, out);
  }
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-78s %7.1f us per launch (%.0f GB/s of partial sums)\n", name, ms / reps * 1e3, 8.0 * N * K * 4 / (ms / reps * 1e-3) / 1e9);
}

int main() {
  float *out, *slabs;
  hipMalloc(&out, sizeof(float) * N * K);
  hipMalloc(&slabs, sizeof(float) * 8 * N * K);
  hipMemset(out, 0, sizeof(float) * N * K);
  run<0>("A fp32 atomics, accumulator layout (4 rows x 64 B per instruction)", out, slabs);
  run<1>("B fp32 atomics, 1 row x 256 B per instruction", out, slabs);
  run<2>("C plain dword stores to per-XCD slabs, accumulator layout", out, slabs);
  run<3>("D plain 16-B stores to per-XCD slabs (4 rows x 256 B per instruction)", out, slabs);
  run<3>("E = D + reduce of the 8 slabs into the output", out, slabs, true);
  return 0;
}
