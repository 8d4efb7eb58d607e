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

// ---- the 16-bit operand / storage format ("lp") --------------------------------------------------------------------
// Every MFMA operand and every stored activation of the fast path is a 16-bit float with fp32 accumulation.  Default:
// IEEE fp16 (11 significand bits).  bf16 (8 bits) cannot meet the path's stated parity bound (normalised boxes within
// 1e-3 L1 of the fp32 reference) on trained-scale weights: rounding the WEIGHTS alone already costs 1.5e-3 .. 6e-3
// (tests/precision_emu.py); fp16 runs on the same MFMA rate (v_mfma_f32_16x16x32_f16) with 8x
// finer operand rounding.  Its narrower exponent is handled where it matters: backward tensors carry a power-of-two
// gradient scale (removed when parameter gradients are written) and every store saturates at +-65504 instead of
// producing inf.  -DSIMVG_LOWP_BF16 builds the same kernels on bf16 (for A/B measurements).
typedef uint16_t lp_t;  // raw 16-bit float bits (fp16, or bf16 with SIMVG_LOWP_BF16)
typedef __attribute__((ext_vector_type(8))) short lpx8_t;   // MFMA A/B fragment (8 values)
typedef __attribute__((ext_vector_type(4))) short lpx4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;    // MFMA 16x16 accumulator
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
typedef float hw_f32x2_t __attribute__((ext_vector_type(2)));

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

#if defined(SIMVG_LOWP_BF16)
#define SIMVG_LOWP_FORMAT 2
typedef __bf16 hw_lp_t;
typedef __bf16 hw_lpx2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float lp_to_f32(lp_t v) { return __uint_as_float(((unsigned int)v) << 16); }
__device__ __forceinline__ void unpack_lp2(unsigned int u, float& lo, float& hi) {
  lo = __uint_as_float(u << 16);
  hi = __uint_as_float(u & 0xffff0000u);
}
__device__ __forceinline__ float lp_sat(float f) { return f; }
__device__ __forceinline__ f32x4_t mfma_lp(lpx8_t a, lpx8_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
#else
#define SIMVG_LOWP_FORMAT 1
typedef _Float16 hw_lp_t;
typedef _Float16 hw_lpx2_t __attribute__((ext_vector_type(2)));
typedef _Float16 hw_lpx8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float lp_to_f32(lp_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ void unpack_lp2(unsigned int u, float& lo, float& hi) {
  const hw_lpx2_t h = __builtin_bit_cast(hw_lpx2_t, u);
  lo = (float)h[0];
  hi = (float)h[1];
}
// saturate instead of overflowing to inf (v_med3_f32); NaN passes through
__device__ __forceinline__ float lp_sat(float f) { return __builtin_amdgcn_fmed3f(f, -65504.f, 65504.f); }
__device__ __forceinline__ f32x4_t mfma_lp(lpx8_t a, lpx8_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(hw_lpx8_t, a), __builtin_bit_cast(hw_lpx8_t, b), c, 0, 0, 0);
}
#endif
// float -> 16-bit, round-to-nearest-even (== torch .to(float16 / bfloat16)): native conversions, which hipcc lowers to
// ONE hardware v_cvt_pk_{f16,bf16}_f32 per PAIR
__device__ __forceinline__ lp_t f32_to_lp(float f) {
  return __builtin_bit_cast(unsigned short, (hw_lp_t)lp_sat(f));
}
__device__ __forceinline__ unsigned int pack_lp2(float lo, float hi) {
  const hw_lpx2_t v = __builtin_convertvector((hw_f32x2_t){lp_sat(lo), lp_sat(hi)}, hw_lpx2_t);
  return __builtin_bit_cast(unsigned int, v);
}
// the same without the saturation clamp: for values known to be in range (softmax probabilities)
__device__ __forceinline__ unsigned int pack_lp2_raw(float lo, float hi) {
  const hw_lpx2_t v = __builtin_convertvector((hw_f32x2_t){lo, hi}, hw_lpx2_t);
  return __builtin_bit_cast(unsigned int, v);
}
// exact-erf GELU (torch F.gelu default) with erf from Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, i.e. ~orders
// below the 16-bit rounding of the stored result) -- one v_exp + one v_rcp instead of the ~50-instruction libm erff;
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
// the same on two values at once: every multiply / fma is a packed v_pk_*_f32 (one issue slot for both values); the reciprocal,
// the exponential, |x| and the sign transfer stay per value.  Same operations in the same order as gelu_parts: identical results.
__device__ __forceinline__ void gelu_parts2(hw_f32x2_t x, hw_f32x2_t& cdf, hw_f32x2_t& pdf) {
  hw_f32x2_t t, e;
  t[0] = __builtin_amdgcn_rcpf(fmaf(0.3275911f, fabsf(x[0]) * 0.70710678118654752440f, 1.0f));
  t[1] = __builtin_amdgcn_rcpf(fmaf(0.3275911f, fabsf(x[1]) * 0.70710678118654752440f, 1.0f));
  const hw_f32x2_t a = (-0.5f * x) * x;
  e[0] = __expf(a[0]);
  e[1] = __expf(a[1]);
  const hw_f32x2_t c5 = {1.061405429f, 1.061405429f}, c4 = {-1.453152027f, -1.453152027f}, c3 = {1.421413741f, 1.421413741f},
                   c2 = {-0.284496736f, -0.284496736f}, c1 = {0.254829592f, 0.254829592f}, half = {0.5f, 0.5f};
  hw_f32x2_t poly = __builtin_elementwise_fma(t, c5, c4);
  poly = __builtin_elementwise_fma(t, poly, c3);
  poly = __builtin_elementwise_fma(t, poly, c2);
  poly = __builtin_elementwise_fma(t, poly, c1);
  poly = t * poly;
  const hw_f32x2_t h = __builtin_elementwise_fma(-0.5f * poly, e, half);
  cdf[0] = 0.5f + copysignf(h[0], x[0]);
  cdf[1] = 0.5f + copysignf(h[1], x[1]);
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
__device__ __forceinline__ lpx4_t lds_read_tr16(const void* lds_addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) lpx4_t*)(lds_addr));
}
// ---- LDS reads that must stay invisible to hipcc's wait-count insertion -----------------------------------------------
// hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of an LDS read that follows an LDS-DMA (global_load_lds / buffer_load
// ... lds) in program order whenever it cannot prove that the two do not alias -- always for the ds_read_tr builtin (no
// memory operand), and for plain reads next to the buffer form.  In a ring pipeline that drains every in-flight stage
// right after it was issued.  Inline-asm reads are not modelled by that pass; the caller counts them (lds_wait_all) and
// must not touch their outputs before that (cdna_hip_programming.md 5.7).
template <int OFF>
__device__ __forceinline__ u32x2_t lds_tr16_asm(unsigned addr) {
  u32x2_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int OFF>
__device__ __forceinline__ u32x4_t lds_b128_asm(unsigned addr) {
  u32x4_t v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ void lds_wait_all() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);      // register-only instructions (MFMA) must not be hoisted above the wait
}
__device__ __forceinline__ void lds_pin(u32x2_t& a, u32x2_t& b) { asm volatile("" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ lpx8_t frag8(u32x2_t lo, u32x2_t hi) {
  return __builtin_bit_cast(lpx8_t, (u32x4_t){lo[0], lo[1], hi[0], hi[1]});
}
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)LDS_PTR(p); }

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
// wgrad.hip: XCD-partitioned weight gradient for the ViT-B encoder shapes; false = not handled (generic kernels run)
// include/simvg_hip.h: the second stage of one slab-flushed wgrad (sum of the row partitions' slabs into dW)
struct simvg_wgrad_reduce_desc {
  const float* slabs; float* dW; long dw_group_stride;
  int lddw, N, K, Q, lo0, hi0, lo1, hi1;
  int assign;     // 0: dW += sum of the slabs; 1: dW = sum (dW need not be zeroed, is not read); 2: the same, and dW holds BOTH row
                  // groups: a group without rows is zeroed.  Set by the caller on a deferred description
};
#define SIMVG_WGRAD_REDUCE_MAX 16
bool simvg_wgrad_x(const void* dY, int lddy, const void* X, int ldx, float* dW, long dw_gstride, int lddw, float* db,
                   int db_gstride, int M, int N, int K, int split, float out_scale, float* slabs,
                   simvg_wgrad_reduce_desc* defer, hipStream_t stream);
long simvg_wgrad_x_slab_floats(int M, int N, int K);
int simvg_wgrad_reduce_launch(const simvg_wgrad_reduce_desc* descs, int n, hipStream_t stream);
