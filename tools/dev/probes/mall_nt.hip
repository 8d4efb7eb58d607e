// Does a NON-TEMPORAL stream leave older data in the 256 MB Infinity Cache?  A (83 MB, the size of the fp32 residual rows of a layer)
// is read once, then a stream B of 250 MB passes (nothing / plain loads / nt loads / plain stores / nt stores / nt stores + nt loads
// of what was stored), then A is read again and that read is timed.  If an nt stream does not allocate, the second read of A runs
// at cache speed as in the "nothing" case.
// hipcc --offload-arch=gfx950 -O3 mall_nt.hip -o mall_nt && ./mall_nt
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <int MODE>      // 0 plain load, 1 nt load, 2 plain store, 3 nt store
__global__ __launch_bounds__(256) void stream(f32x4_t* p, long n4, float* sink) {
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    if (MODE == 0) acc += p[i];
    else if (MODE == 1) acc += __builtin_nontemporal_load(p + i);
    else if (MODE == 2) p[i] = (f32x4_t){1.f, 2.f, 3.f, (float)i};
    else __builtin_nontemporal_store((f32x4_t){1.f, 2.f, 3.f, (float)i}, p + i);
  }
  if (MODE < 2 && acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) *sink = acc[0];
}

// the timed reader: 8 independent 16-byte loads in flight per lane (the one-load loop above tops out at 5.8 TB/s whatever the source)
__global__ __launch_bounds__(256) void reader(const f32x4_t* p, long n4, float* sink) {
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  for (; i + 7 * stride < n4; i += 8 * stride) {
    f32x4_t v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[i + u * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  for (; i < n4; i += stride) acc += p[i];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) *sink = acc[0];
}

int main() {
  const long NA = 83L << 20, NB = 250L << 20;
  float *A, *B, *sink;
  hipMalloc(&A, NA); hipMalloc(&B, NB); hipMalloc(&sink, 4);
  hipMemset(A, 0, NA); hipMemset(B, 0, NB);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[] = {"nothing between", "plain loads of B", "nt loads of B", "plain stores to B", "nt stores to B", "nt stores to B, then nt loads of B",
                         "plain stores to B, then plain loads of B"};
  for (int mode = 0; mode < 7; ++mode) {
    float best = 1e9f, sum = 0.f;
    for (int rep = 0; rep < 6; ++rep) {
      stream<0><<<2048, 256>>>((f32x4_t*)A, NA / 16, sink);                 // A becomes resident
      if (mode == 1) stream<0><<<2048, 256>>>((f32x4_t*)B, NB / 16, sink);
      if (mode == 2) stream<1><<<2048, 256>>>((f32x4_t*)B, NB / 16, sink);
      if (mode == 3 || mode == 6) stream<2><<<2048, 256>>>((f32x4_t*)B, NB / 16, sink);
      if (mode == 4 || mode == 5) stream<3><<<2048, 256>>>((f32x4_t*)B, NB / 16, sink);
      if (mode == 5) stream<1><<<2048, 256>>>((f32x4_t*)B, NB / 16, sink);
      if (mode == 6) stream<0><<<2048, 256>>>((f32x4_t*)B, NB / 16, sink);
      hipEventRecord(e0);
      reader<<<1024, 256>>>((const f32x4_t*)A, NA / 16, sink);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) { best = ms < best ? ms : best; sum += ms; }
    }
    printf("%-45s second read of A: best %.1f us, mean %.1f us  (%.2f TB/s)\n", names[mode], best * 1e3, sum / 5 * 1e3, NA / (best * 1e-3) / 1e12);
  }
  return 0;
}
