// C-ABI plumbing for libsimvg_hip.so: error reporting, version, and hardware-semantics probes
// (MFMA fragment layout, ds_read_b64_tr_b16, global_load_lds) used by tests/test_hw_probe.py.
#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";

extern "C" void simvg_set_error(const char* msg) {
  strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* simvg_last_error(void) { return g_err; }
extern "C" int simvg_version(void) { return 3; }
// sha256[:32] of csrc/*.hip + csrc/*.h (+ the variant's compiler flags) the library was built from (simvg_amd/build.py);
// `simvg_amd._lib.load()` refuses a library whose hash differs from the sources next to it
#ifndef SIMVG_SOURCE_HASH
#define SIMVG_SOURCE_HASH "unknown"
#endif
extern "C" const char* simvg_source_hash(void) { return SIMVG_SOURCE_HASH; }

namespace {
// out[64][4] = D fragment of one v_mfma_f32_16x16x32_bf16 with A[i][k] = a[i*32+k], B[k][j] = b[k*16+j]
__global__ void probe_mfma_kernel(const lp_t* a, const lp_t* b, float* out) {
  const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
  lpx8_t fa, fb;
  for (int e = 0; e < 8; ++e) {
    fa[e] = (short)a[i * 32 + 8 * g + e];
    fb[e] = (short)b[(8 * g + e) * 16 + i];
  }
  f32x4_t c = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  c = mfma_lp(fa, fb, c);
  for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
}
// LDS filled with lds[e] = e (16-bit); lane l supplies byte address addr[l]; out[l][0..3] = what it got
__global__ void probe_tr16_kernel(const int* addr, short* out) {
  __shared__ __attribute__((aligned(16))) short lds[4096];
  for (int e = threadIdx.x; e < 4096; e += 64) lds[e] = (short)e;
  __syncthreads();
  const lpx4_t v = lds_read_tr16((const char*)lds + addr[threadIdx.x]);
  for (int r = 0; r < 4; ++r) out[threadIdx.x * 4 + r] = v[r];
}
// one wave: global_load_lds 16 B/lane from src + perm[lane]*8 elements; dump LDS linearly
__global__ void probe_glds_kernel(const short* src, const int* perm, short* out) {
  __shared__ __attribute__((aligned(16))) short lds[512];
  __builtin_amdgcn_global_load_lds(GLB_PTR(src + perm[threadIdx.x] * 8), LDS_PTR(lds), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int e = threadIdx.x; e < 512; e += 64) out[e] = lds[e];
}
}  // namespace

extern "C" int simvg_probe_mfma(const void* a, const void* b, float* out, hipStream_t s) {
  hipLaunchKernelGGL(probe_mfma_kernel, dim3(1), dim3(64), 0, s, (const lp_t*)a, (const lp_t*)b, out);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}
extern "C" int simvg_probe_tr16(const int* addr, void* out, hipStream_t s) {
  hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, s, addr, (short*)out);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}
extern "C" int simvg_probe_glds(const void* src, const int* perm, void* out, hipStream_t s) {
  hipLaunchKernelGGL(probe_glds_kernel, dim3(1), dim3(64), 0, s, (const short*)src, perm, (short*)out);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}
