// Shared device/host helpers for libsimvg_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SIMVG_OK 0
#define SIMVG_ERR_ARG (-1)
#define SIMVG_ERR_HIP (-2)

extern "C" void simvg_set_error(const char* msg);

#define SIMVG_CHECK_ARG(cond, msg)            \
  do {                                        \
    if (!(cond)) {                            \
      simvg_set_error(msg);                   \
      return SIMVG_ERR_ARG;                   \
    }                                         \
  } while (0)

#define SIMVG_LAUNCH_CHECK()                               \
  do {                                                     \
    hipError_t e__ = hipGetLastError();                    \
    if (e__ != hipSuccess) {                               \
      simvg_set_error(hipGetErrorString(e__));             \
      return SIMVG_ERR_HIP;                                \
    }                                                      \
  } while (0)

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // MFMA A/B fragment (8 bf16)
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;    // MFMA 16x16 accumulator
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __uint_as_float(((unsigned int)v) << 16);
}
// float -> bf16, round-to-nearest-even (matches torch .to(bfloat16)): native __bf16 conversions, which hipcc lowers to
// the gfx950 hardware v_cvt_pk_bf16_f32 (one instruction per PAIR instead of ~6 integer ops per value)
typedef __bf16 hw_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float hw_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  return __builtin_bit_cast(unsigned short, (__bf16)f);
}
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
  const hw_bf16x2_t v = __builtin_convertvector((hw_f32x2_t){lo, hi}, hw_bf16x2_t);
  return __builtin_bit_cast(unsigned int, v);
}
// exact-erf GELU (torch F.gelu default) with erf from Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, i.e. ~4 orders
// below the bf16 rounding of the stored result) -- one v_exp + one v_rcp instead of the ~50-instruction libm erff;
// the same exp(-x^2/2) also gives the Gaussian density that GELU' needs.
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& pdf) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  // v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division sequence `__frcp_rn` expands to: the relative 6e-8
  // it may add to t is below the 1.5e-7 of the approximation itself
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  const float e = __expf(-0.5f * x * x);                        // exp(-z^2)
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float h = fmaf(-0.5f * poly, e, 0.5f);                   // 0.5 * erf(|x|/sqrt2)
  cdf = 0.5f + copysignf(h, x);
  pdf = 0.39894228040143267794f * e;
}
__device__ __forceinline__ float gelu_erf(float x) {
  float cdf, pdf;
  gelu_parts(x, cdf, pdf);
  return x * cdf;
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
  float cdf, pdf;
  gelu_parts(x, cdf, pdf);
  return fmaf(x, pdf, cdf);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// hardware transpose read: 16-lane group reads a [4 rows][16 cols] bf16 block (each lane supplies
// the address of 4 contiguous elements: lane p -> row p>>2, cols 4*(p&3)..+3) and lane i receives
// column i, rows 0..3.
__device__ __forceinline__ bf16x4_t lds_read_tr16(const void* lds_addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4_t*)(lds_addr));
}
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
