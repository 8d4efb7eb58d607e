// Multiway LayerNorm forward / backward (gfx950).  HBM-bound: one wave per row, 16-B vector
// loads, the row lives in registers between the statistics and the normalise pass.
//
// Replaces torch.nn.LayerNorm wrapped by torchscale MultiwayNetwork (reference call sites
// beit3_base.py:136,157,396-397 and torchscale MultiheadAttention.inner_attn_ln /
// FeedForwardNetwork.ffn_layernorm; SURVEY.md §2.3 E5,E13,E18,E20) and the detrex decoder's
// nn.LayerNorm (transformer.py:47-49,119-121).  Rows [0,split) use gamma/beta group 0 ("A",
// vision), rows [split,M) group 1 ("B", text).
#include <stdlib.h>
#include <algorithm>

#include "common.h"

// include/simvg_hip.h: one deferred second stage of the two-stage dgamma / dbeta reduction
struct simvg_ln_reduce_desc {
  const float* partial; float* dgamma; float* dbeta;
  int group_stride, D, blocks0, blocks1;
};
#define SIMVG_LN_REDUCE_MAX 64

namespace {

template <typename T> struct Ld4;
template <> struct Ld4<float> {
  static __device__ __forceinline__ void ld(const float* p, float* v) {
    const f32x4_t t = *(const f32x4_t*)p;
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  }
};
template <> struct Ld4<lp_t> {
  static __device__ __forceinline__ void ld(const lp_t* p, float* v) {
    const u32x2_t t = *(const u32x2_t*)p;
    unpack_lp2(t[0], v[0], v[1]);
    unpack_lp2(t[1], v[2], v[3]);
  }
};
__device__ __forceinline__ void st4_lp(lp_t* p, const float* v) {
  *(u32x2_t*)p = (u32x2_t){pack_lp2(v[0], v[1]), pack_lp2(v[2], v[3])};
}

template <typename TIn, int NIT>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const TIn* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, int gstride, lp_t* __restrict__ y,
                                                     int ldy, float* __restrict__ y32, int ldy32, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int M, int D, int split, float eps,
                                                     int gelu_in) {
  // gelu_in: x holds the fc1 PRE-activation u and the LayerNorm input is gelu(u), recomputed here (the activation is
  // never stored: one [M, F] bf16 write per layer less in the fc1 GEMM, one read less in the backward)
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int g = row >= split;
  const TIn* xr = x + (long)row * ldx;
  float v[NIT][4];
  float s = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = (it * 64 + lane) * 4;
    if (c < D) {
      Ld4<TIn>::ld(xr + c, v[it]);
      if (gelu_in) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[it][k] = gelu_erf(v[it][k]);
      }
      s += (v[it][0] + v[it][1]) + (v[it][2] + v[it][3]);
    } else {
      v[it][0] = v[it][1] = v[it][2] = v[it][3] = 0.f;
    }
  }
  const float mu = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = (it * 64 + lane) * 4;
    if (c < D) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { const float d = v[it][k] - mu; q += d * d; }
    }
  }
  const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
  if (lane == 0) {
    if (mean) mean[row] = mu;
    if (rstd) rstd[row] = rs;
  }
  const float* gm = gamma + (long)g * gstride;
  const float* bt = beta + (long)g * gstride;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = (it * 64 + lane) * 4;
    if (c < D) {
      const f32x4_t gv = *(const f32x4_t*)(gm + c), bv = *(const f32x4_t*)(bt + c);
      float o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = (v[it][k] - mu) * rs * gv[k] + bv[k];
      if (y) st4_lp(y + (long)row * ldy + c, o);
      if (y32) *(f32x4_t*)(y32 + (long)row * ldy32 + c) = (f32x4_t){o[0], o[1], o[2], o[3]};
    }
  }
}


// Narrow rows at training sizes, RW rows per wave: the kernel above keeps one 3-4 KB row per wave in flight, which at full
// occupancy is ~96 KB per CU -- bound by loads in flight (Little), like the backward kernels were.  Here a wave issues the loads of
// RW rows back to back (raw form) and then normalises them one after the other; width (D = NIT * 256) and output (16 bit) are
// compile-time, straight-line code, so hipcc counts its own waits.
template <typename TIn, int NIT, int RW>
__global__ __launch_bounds__(256) void ln_fwd_rows_kernel(const TIn* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int gstride, lp_t* __restrict__ y, int ldy,
                                                          float* __restrict__ mean, float* __restrict__ rstd, int M, int split, float eps) {
  constexpr int D = NIT * 256;
  constexpr bool XF = sizeof(TIn) == 4;
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RW;
  if (row0 >= M) return;
  f32x4_t xf[RW][XF ? NIT : 1];
  u32x2_t xb[RW][XF ? 1 : NIT];
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int row = min(row0 + r, M - 1);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = (it * 64 + lane) * 4;
      if constexpr (XF) xf[r][it] = *(const f32x4_t*)((const float*)x + (long)row * ldx + c);
      else xb[r][it] = *(const u32x2_t*)((const lp_t*)x + (long)row * ldx + c);
    }
  }
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int row = row0 + r;
    if (row >= M) break;
    float v[NIT][4];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      if constexpr (XF) { const f32x4_t t = xf[r][it]; v[it][0] = t[0]; v[it][1] = t[1]; v[it][2] = t[2]; v[it][3] = t[3]; }
      else { unpack_lp2(xb[r][it][0], v[it][0], v[it][1]); unpack_lp2(xb[r][it][1], v[it][2], v[it][3]); }
      s += (v[it][0] + v[it][1]) + (v[it][2] + v[it][3]);
    }
    const float mu = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it)
#pragma unroll
      for (int k = 0; k < 4; ++k) { const float d = v[it][k] - mu; q += d * d; }
    const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
    if (lane == 0) {
      if (mean) mean[row] = mu;
      if (rstd) rstd[row] = rs;
    }
    const int g = row >= split;
    const float* gm = gamma + (long)g * gstride;
    const float* bt = beta + (long)g * gstride;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = (it * 64 + lane) * 4;
      const f32x4_t gv = *(const f32x4_t*)(gm + c), bv = *(const f32x4_t*)(bt + c);
      float o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = (v[it][k] - mu) * rs * gv[k] + bv[k];
      st4_lp(y + (long)row * ldy + c, o);
    }
  }
}

// Wide 16-bit rows (D = NIT8 * 512: the 3072 / 4096-wide ffn_layernorm, whose input is the fc1 pre-activation u and whose
// LayerNorm input gelu(u) is recomputed here).  The generic kernel above guards every 256-column slice with `c < D` and
// tests the run-time gelu flag inside the slice loop; each slice's load then sits in its own exec-masked block, hipcc
// cannot move it above the previous slice's arithmetic, and a wave has ONE 512-B load in flight at a time (measured
// 2.9 TB/s at D = 3072).  Here the width and the activation are compile-time: the row's NIT8 16-B loads are issued
// back-to-back before any arithmetic, each lane owns 8 consecutive columns per slice (16-B loads and stores): 113 -> 85 us.
// What remains above the 66 us of the same kernel without the activation is VALU: the erf GELU + LayerNorm are ~35 VALU
// operations per element (a persistent variant with the next row's loads in flight measured the same 88 us).
#ifndef LN_WIDE_RW
#define LN_WIDE_RW 1
#endif
// RW rows per wave (raw 16-bit rows all requested before the first is normalised).  A wave per row keeps 6-8 KB in flight, ~96 KB
// per CU at 4 waves per SIMD, and the kernel runs at the ~4.2 TB/s that allows; RW = 2 would double it, but the fp32 row (48
// registers) + a second raw row (24) + the activation's temporaries do not fit 128 VGPRs: 90 us with spills against 78 us
// (without the activation 72 vs 74 us), so RW stays 1 here -- the narrow kernel above is where two rows per wave pay
template <int NIT8, bool GELU, int RW>
__global__ __launch_bounds__(256, NIT8 <= 6 ? 4 : 3) void ln_fwd_wide_kernel(const lp_t* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int gstride, lp_t* __restrict__ y,
                                                          int ldy, float* __restrict__ mean, float* __restrict__ rstd, int M,
                                                          int split, float eps) {
  constexpr int D = NIT8 * 512;
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RW;
  if (row0 >= M) return;
  u32x4_t raw[RW][NIT8];
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const lp_t* xr = x + (long)min(row0 + r, M - 1) * ldx + lane * 8;
#pragma unroll
    for (int it = 0; it < NIT8; ++it) raw[r][it] = __builtin_nontemporal_load((const u32x4_t*)(xr + it * 512));
  }
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int row = row0 + r;
    if (row >= M) break;
    const int g = row >= split;
    // two values per register pair: the arithmetic below is packed fp32 (v_pk_fma / v_pk_mul / v_pk_add: one issue slot for two values)
    hw_f32x2_t v[NIT8][4];
    hw_f32x2_t s2 = {0.f, 0.f};
#pragma unroll
    for (int it = 0; it < NIT8; ++it) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float lo, hi;
        unpack_lp2(raw[r][it][k], lo, hi);
        v[it][k] = (hw_f32x2_t){lo, hi};
        if (GELU) {
          hw_f32x2_t cdf, pdf;
          gelu_parts2(v[it][k], cdf, pdf);
          v[it][k] = v[it][k] * cdf;
        }
      }
      s2 += (v[it][0] + v[it][1]) + (v[it][2] + v[it][3]);
    }
    const float mu = wave_sum(s2[0] + s2[1]) * (1.f / (float)D);
    const hw_f32x2_t mu2 = {mu, mu};
    hw_f32x2_t q2 = {0.f, 0.f};
#pragma unroll
    for (int it = 0; it < NIT8; ++it)
#pragma unroll
      for (int k = 0; k < 4; ++k) { const hw_f32x2_t d = v[it][k] - mu2; q2 = __builtin_elementwise_fma(d, d, q2); }
    const float rs = rsqrtf(wave_sum(q2[0] + q2[1]) * (1.f / (float)D) + eps);
    if (lane == 0) {
      if (mean) mean[row] = mu;
      if (rstd) rstd[row] = rs;
    }
    const float* gm = gamma + (long)g * gstride + lane * 8;
    const float* bt = beta + (long)g * gstride + lane * 8;
    lp_t* yr = y + (long)row * ldy + lane * 8;
    const hw_f32x2_t rs2 = {rs, rs};
#pragma unroll
    for (int it = 0; it < NIT8; ++it) {
      const f32x4_t g0 = *(const f32x4_t*)(gm + it * 512), g1 = *(const f32x4_t*)(gm + it * 512 + 4);
      const f32x4_t b0 = *(const f32x4_t*)(bt + it * 512), b1 = *(const f32x4_t*)(bt + it * 512 + 4);
      const hw_f32x2_t gg[4] = {{g0[0], g0[1]}, {g0[2], g0[3]}, {g1[0], g1[1]}, {g1[2], g1[3]}};
      const hw_f32x2_t bb[4] = {{b0[0], b0[1]}, {b0[2], b0[3]}, {b1[0], b1[1]}, {b1[2], b1[3]}};
      unsigned w[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const hw_f32x2_t o = __builtin_elementwise_fma((v[it][k] - mu2) * rs2, gg[k], bb[k]);
        w[k] = pack_lp2(o[0], o[1]);
      }
      *(u32x4_t*)(yr + it * 512) = (u32x4_t){w[0], w[1], w[2], w[3]};
    }
  }
}

// Backward.  Each block owns `rows_per_block` consecutive rows of ONE group; each wave walks its
// rows, accumulating dgamma/dbeta for its columns in registers; one LDS reduction + atomics per block.
//   dx = rstd * (dy*gamma - mean(dy*gamma) - xhat * mean(dy*gamma*xhat))
//   out_bf16 : dx [* gelu'(u)]                       (feeds a dgrad GEMM)
//   out_f32  : dres + dx  (+ bf16 copy * row_scale)   (residual-stream gradient)
template <typename TIn, typename TDy, int NIT>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const TDy* __restrict__ dy, int lddy, const TIn* __restrict__ x, int ldx,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, int gstride,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     lp_t* __restrict__ out_bf16, int ldob, const lp_t* __restrict__ gelu_u, int ldu,
                                                     const float* __restrict__ dres, float* __restrict__ out_f32, int ldof,
                                                     lp_t* __restrict__ out_scaled, int ldos, const float* __restrict__ row_scale,
                                                     int rps0, int rps1, int M, int D, int split, int rows_per_block, int blocks0,
                                                     float* __restrict__ partial, float dy_scale, float pscale) {
  extern __shared__ float red[];  // [2][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // only a bf16 x can be the GELU pre-activation: for fp32 x this is a compile-time false and the GELU' registers vanish
  // (as a runtime flag it cost the residual-stream instantiation 97 -> 126 us through register pressure)
  const bool x_is_u = sizeof(TIn) == 2 && gelu_u != nullptr && (const void*)gelu_u == (const void*)x;
  const int blk = blockIdx.x;
  const int g = blk >= blocks0;
  const int r_begin = g ? split + (blk - blocks0) * rows_per_block : blk * rows_per_block;
  const int r_end = min(r_begin + rows_per_block, g ? M : split);
  const float* gm = gamma + (long)g * gstride;
  float gv[NIT][4], ag[NIT][4], ab[NIT][4];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = (it * 64 + lane) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) { ag[it][k] = 0.f; ab[it][k] = 0.f; gv[it][k] = 0.f; }
    if (c < D) {
      const f32x4_t t = *(const f32x4_t*)(gm + c);
      gv[it][0] = t[0]; gv[it][1] = t[1]; gv[it][2] = t[2]; gv[it][3] = t[3];
    }
  }
  const float invD = 1.f / (float)D;
  for (int row = r_begin + wave; row < r_end; row += 4) {
    const float mu = mean[row], rs = rstd[row];
    float xh[NIT][4], dyv[NIT][4], gp[NIT][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = (it * 64 + lane) * 4;
      if (c < D) {
        float xv[4];
        Ld4<TIn>::ld(x + (long)row * ldx + c, xv);
        Ld4<TDy>::ld(dy + (long)row * lddy + c, dyv[it]);
        if (sizeof(TDy) == 4) {      // fp32 dy: the entry of a scaled backward (gradient scale applied here)
#pragma unroll
          for (int k = 0; k < 4; ++k) dyv[it][k] *= dy_scale;
        }
        if (x_is_u) {     // x is the GELU pre-activation: LN input g = u*Phi(u), and keep GELU'(u) for the output
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float cdf, pdf;
            gelu_parts(xv[k], cdf, pdf);
            gp[it][k] = fmaf(xv[k], pdf, cdf);
            xv[k] *= cdf;
          }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          xh[it][k] = (xv[k] - mu) * rs;
          const float dg = dyv[it][k] * gv[it][k];
          s1 += dg;
          s2 += dg * xh[it][k];
          ag[it][k] += dyv[it][k] * xh[it][k];
          ab[it][k] += dyv[it][k];
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) { xh[it][k] = 0.f; dyv[it][k] = 0.f; }
      }
    }
    const float c1 = wave_sum(s1) * invD, c2 = wave_sum(s2) * invD;
    float scl = 1.f;
    if (out_scaled && row_scale) scl = row_scale[g ? (row - split) / rps1 : row / rps0];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = (it * 64 + lane) * 4;
      if (c < D) {
        float dx[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) dx[k] = rs * (dyv[it][k] * gv[it][k] - c1 - xh[it][k] * c2);
        if (out_bf16) {
          if (x_is_u) {
#pragma unroll
            for (int k = 0; k < 4; ++k) dx[k] *= gp[it][k];
          } else if (gelu_u) {
            float u[4];
            Ld4<lp_t>::ld(gelu_u + (long)row * ldu + c, u);
#pragma unroll
            for (int k = 0; k < 4; ++k) dx[k] *= gelu_erf_grad(u[k]);
          }
          st4_lp(out_bf16 + (long)row * ldob + c, dx);
        }
        if (out_f32) {
          if (dres) {
            const f32x4_t t = *(const f32x4_t*)(dres + (long)row * ldof + c);
#pragma unroll
            for (int k = 0; k < 4; ++k) dx[k] += t[k];
          }
          *(f32x4_t*)(out_f32 + (long)row * ldof + c) = (f32x4_t){dx[0], dx[1], dx[2], dx[3]};
          if (out_scaled) {
            float t[4] = {dx[0] * scl, dx[1] * scl, dx[2] * scl, dx[3] * scl};
            st4_lp(out_scaled + (long)row * ldos + c, t);
          }
        }
      }
    }
  }
  // block reduction of dgamma / dbeta over the 4 waves in LDS (ds_add_f32), then one global
  // atomic per column
  float* rg = red;
  float* rb = red + D;
  for (int c = threadIdx.x; c < 2 * D; c += 256) red[c] = 0.f;
  __syncthreads();
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = (it * 64 + lane) * 4;
    if (c < D) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { atomicAdd(&rg[c + k], ag[it][k] * pscale); atomicAdd(&rb[c + k], ab[it][k] * pscale); }
    }
  }
  __syncthreads();
  if (partial) {          // two-stage reduction: [block][2][D] partials, summed by ln_param_reduce_kernel
    float* pp = partial + (long)blk * 2 * D;
    for (int c = threadIdx.x; c < D; c += 256) { pp[c] = rg[c]; pp[D + c] = rb[c]; }
  } else if (r_begin < r_end) {
    for (int c = threadIdx.x; c < D; c += 256) {
      atomicAdd(dgamma + (long)g * gstride + c, rg[c]);
      atomicAdd(dbeta + (long)g * gstride + c, rb[c]);
    }
  }
}

// Narrow rows (one wave per row) with a software prefetch: the generic kernel above issues a row's loads when it needs
// them and relies on occupancy alone (measured 1.7 - 3.4 TB/s); here every wave keeps its NEXT row in flight in raw
// form (two register slots, rows alternate between them) while it works on the current one.  Instances: the residual
// stream (fp32 x, + dres, fp32 dx and the bf16 copy for the next GEMM) and the attention sub-LayerNorm (bf16 x, bf16 dx).
template <typename TIn, bool RES, int NIT>
__global__ __launch_bounds__(256) void ln_bwd_pf_kernel(const lp_t* __restrict__ dy, int lddy, const TIn* __restrict__ x, int ldx,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma, int gstride,
                                                        float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                        lp_t* __restrict__ out_bf16, int ldob,
                                                        const float* __restrict__ dres, float* __restrict__ out_f32, int ldof,
                                                        lp_t* __restrict__ out_scaled, int ldos, const float* __restrict__ row_scale,
                                                        int rps0, int rps1, int M, int D, int split, int rows_per_block, int blocks0,
                                                        float pscale) {
  extern __shared__ float red[];  // [2][D]
  constexpr bool XF = sizeof(TIn) == 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int blk = blockIdx.x;
  const int g = blk >= blocks0;
  const int r_begin = g ? split + (blk - blocks0) * rows_per_block : blk * rows_per_block;
  const int r_end = min(r_begin + rows_per_block, g ? M : split);
  const float* gm = gamma + (long)g * gstride;
  float gv[NIT][4], ag[NIT][4], ab[NIT][4];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = (it * 64 + lane) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) { ag[it][k] = 0.f; ab[it][k] = 0.f; gv[it][k] = 0.f; }
    if (c < D) {
      const f32x4_t t = *(const f32x4_t*)(gm + c);
      gv[it][0] = t[0]; gv[it][1] = t[1]; gv[it][2] = t[2]; gv[it][3] = t[3];
    }
  }
  const float invD = 1.f / (float)D;
  f32x4_t xf[2][XF ? NIT : 1], rq[2][RES ? NIT : 1];
  u32x2_t xb[2][XF ? 1 : NIT], dq[2][NIT];
  float mu_q[2], rs_q[2];        // the row statistics travel with the prefetch: loaded at the top of the row they showed up
                                 // as 24-31 % of the wave cycles in s_waitcnt lgkmcnt (PMC SQ_WAIT_INST_LDS)
  auto fetch = [&](int row, int slot) {
    mu_q[slot] = mean[row];
    rs_q[slot] = rstd[row];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = (it * 64 + lane) * 4;
      if (c < D) {
        if constexpr (XF) xf[slot][it] = *(const f32x4_t*)((const float*)x + (long)row * ldx + c);
        else xb[slot][it] = *(const u32x2_t*)((const lp_t*)x + (long)row * ldx + c);
        dq[slot][it] = *(const u32x2_t*)(dy + (long)row * lddy + c);
        if constexpr (RES) rq[slot][it] = *(const f32x4_t*)(dres + (long)row * ldof + c);
      }
    }
  };
  auto body = [&](int row, int slot) {
    const float mu = mu_q[slot], rs = rs_q[slot];
    float xh[NIT][4], dyv[NIT][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = (it * 64 + lane) * 4;
      if (c < D) {
        float xv[4];
        if constexpr (XF) { const f32x4_t t = xf[slot][it]; xv[0] = t[0]; xv[1] = t[1]; xv[2] = t[2]; xv[3] = t[3]; }
        else {
          const u32x2_t t = xb[slot][it];
          unpack_lp2(t[0], xv[0], xv[1]);
          unpack_lp2(t[1], xv[2], xv[3]);
        }
        const u32x2_t d = dq[slot][it];
        unpack_lp2(d[0], dyv[it][0], dyv[it][1]);
        unpack_lp2(d[1], dyv[it][2], dyv[it][3]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          xh[it][k] = (xv[k] - mu) * rs;
          const float dg = dyv[it][k] * gv[it][k];
          s1 += dg;
          s2 += dg * xh[it][k];
          ag[it][k] += dyv[it][k] * xh[it][k];
          ab[it][k] += dyv[it][k];
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) { xh[it][k] = 0.f; dyv[it][k] = 0.f; }
      }
    }
    const float c1 = wave_sum(s1) * invD, c2 = wave_sum(s2) * invD;
    float scl = 1.f;
    if (RES && out_scaled && row_scale) scl = row_scale[g ? (row - split) / rps1 : row / rps0];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = (it * 64 + lane) * 4;
      if (c < D) {
        float dx[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) dx[k] = rs * (dyv[it][k] * gv[it][k] - c1 - xh[it][k] * c2);
        if constexpr (RES) {
          const f32x4_t t = rq[slot][it];
#pragma unroll
          for (int k = 0; k < 4; ++k) dx[k] += t[k];
          *(f32x4_t*)(out_f32 + (long)row * ldof + c) = (f32x4_t){dx[0], dx[1], dx[2], dx[3]};
          if (out_scaled) {
            float t2[4] = {dx[0] * scl, dx[1] * scl, dx[2] * scl, dx[3] * scl};
            st4_lp(out_scaled + (long)row * ldos + c, t2);
          }
        } else {
          st4_lp(out_bf16 + (long)row * ldob + c, dx);
        }
      }
    }
    if (row + 8 < r_end) fetch(row + 8, slot);      // this wave's rows are r, r+4, r+8, ...: slots alternate
  };
  int row = r_begin + wave;
  if (row < r_end) fetch(row, 0);
  if (row + 4 < r_end) fetch(row + 4, 1);
  for (; row + 4 < r_end; row += 8) { body(row, 0); body(row + 4, 1); }
  if (row < r_end) body(row, 0);
  float* rg = red;
  float* rb = red + D;
  for (int c = threadIdx.x; c < 2 * D; c += 256) red[c] = 0.f;
  __syncthreads();
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = (it * 64 + lane) * 4;
    if (c < D) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { atomicAdd(&rg[c + k], ag[it][k] * pscale); atomicAdd(&rb[c + k], ab[it][k] * pscale); }
    }
  }
  __syncthreads();
  if (r_begin < r_end) {
    for (int c = threadIdx.x; c < D; c += 256) {
      atomicAdd(dgamma + (long)g * gstride + c, rg[c]);
      atomicAdd(dbeta + (long)g * gstride + c, rb[c]);
    }
  }
}

// Narrow rows, many loads in flight (round 3).  The wave-per-row kernels above keep 12 columns of every array in a lane
// (188 / 148 VGPRs -> 7-12 waves per CU with two rows each in flight: ~100 / 70 KB per CU, and by Little's law 4.1 / 2.1 TB/s
// at the ~6.5 us loaded latency -- the Adam pass reaches 6.8 TB/s with ~300 KB per CU in flight).  Here a ROW is shared by the
// NW = D / 256 waves of a block, each lane owning 4 consecutive columns: a row in flight costs 4 (16-bit x) to 10 (fp32 x +
// dres) registers, so a block keeps 2 R rows in flight in raw form (two batches of R rows: the loads of batch b + 2 are issued
// as soon as batch b is finished) at ~5 blocks per CU.  The two row sums cross the waves through LDS, one barrier per BATCH
// (partials double-buffered by batch parity); dgamma / dbeta partials go to the workspace of the two-stage reduction.
// (The same form for the 3072-wide ffn_layernorm backward -- 12 waves per row, 80 VGPRs, two blocks per CU -- measured 147 us
// against the 142 us of ln_bwd_ffn_kernel below: that kernel is bound by its ~33 VALU operations per element, two of them
// transcendental, not by loads in flight.  Deeper per-wave load rings (4-8 batches, buffer loads, branch-free bodies) did not
// survive hipcc: its wait-count pass closes a ring that crosses the loop's back edge with vmcnt(0) / vmcnt(1), and the
// bodies' register allocation spilled the raw rows.  What carries these kernels is blocks per CU, not ring depth.)
template <typename TIn, bool RES, int NW, int R>
__global__ __launch_bounds__(NW * 64) void ln_bwd_tile_kernel(const lp_t* __restrict__ dy, int lddy, const TIn* __restrict__ x, int ldx,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              const float* __restrict__ gamma, int gstride,
                                                              lp_t* __restrict__ out_bf16, int ldob,
                                                              const float* __restrict__ dres, float* __restrict__ out_f32, int ldof,
                                                              lp_t* __restrict__ out_scaled, int ldos, const float* __restrict__ row_scale,
                                                              int rps0, int rps1, int M, int split, int rows_per_block, int blocks0,
                                                              float* __restrict__ partial, float pscale) {
  constexpr int D = NW * 256;
  constexpr bool XF = sizeof(TIn) == 4;
  __shared__ float part[2][R][NW][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int blk = blockIdx.x;
  const int g = blk >= blocks0;
  const int r_begin = g ? split + (blk - blocks0) * rows_per_block : blk * rows_per_block;
  const int r_end = min(r_begin + rows_per_block, g ? M : split);
  const int c = tid * 4;
  const f32x4_t gv = *(const f32x4_t*)(gamma + (long)g * gstride + c);
  f32x4_t ag = {0.f, 0.f, 0.f, 0.f}, ab = {0.f, 0.f, 0.f, 0.f};
  constexpr float invD = 1.f / (float)D;
  f32x4_t xf[2][XF ? R : 1], rq[2][RES ? R : 1];
  u32x2_t xb[2][XF ? 1 : R], dq[2][R];
  auto fetch = [&](int b, int slot) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int row = r_begin + b * R + r;
      if (row < r_end) {
        if constexpr (XF) xf[slot][r] = *(const f32x4_t*)((const float*)x + (long)row * ldx + c);
        else xb[slot][r] = *(const u32x2_t*)((const lp_t*)x + (long)row * ldx + c);
        dq[slot][r] = *(const u32x2_t*)(dy + (long)row * lddy + c);
        if constexpr (RES) rq[slot][r] = *(const f32x4_t*)(dres + (long)row * ldof + c);
      }
    }
  };
  auto unpack_row = [&](int slot, int r, float mu, float rs, float (&xh)[4], float (&dyv)[4]) {
    float xv[4];
    if constexpr (XF) { const f32x4_t t = xf[slot][r]; xv[0] = t[0]; xv[1] = t[1]; xv[2] = t[2]; xv[3] = t[3]; }
    else { const u32x2_t t = xb[slot][r]; unpack_lp2(t[0], xv[0], xv[1]); unpack_lp2(t[1], xv[2], xv[3]); }
    const u32x2_t d = dq[slot][r];
    unpack_lp2(d[0], dyv[0], dyv[1]);
    unpack_lp2(d[1], dyv[2], dyv[3]);
#pragma unroll
    for (int k = 0; k < 4; ++k) xh[k] = (xv[k] - mu) * rs;
  };
  const int nb = (r_end - r_begin + R - 1) / R;
  if (nb > 0) fetch(0, 0);
  if (nb > 1) fetch(1, 1);
  auto batch = [&](int b, int slot) {
    float mu[R], rs[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int row = r_begin + b * R + r;
      mu[r] = rs[r] = 0.f;
      if (row < r_end) {
        mu[r] = mean[row];
        rs[r] = rstd[row];
        float xh[4], dyv[4];
        unpack_row(slot, r, mu[r], rs[r], xh, dyv);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float dg = dyv[k] * gv[k];
          s1 += dg;
          s2 += dg * xh[k];
          ag[k] += dyv[k] * xh[k];
          ab[k] += dyv[k];
        }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        if (lane == 0) { part[slot][r][wave][0] = s1; part[slot][r][wave][1] = s2; }
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int row = r_begin + b * R + r;
      if (row < r_end) {
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) { c1 += part[slot][r][w][0]; c2 += part[slot][r][w][1]; }
        c1 *= invD;
        c2 *= invD;
        float xh[4], dyv[4], dx[4];
        unpack_row(slot, r, mu[r], rs[r], xh, dyv);
#pragma unroll
        for (int k = 0; k < 4; ++k) dx[k] = rs[r] * (dyv[k] * gv[k] - c1 - xh[k] * c2);
        if constexpr (RES) {
          const f32x4_t t = rq[slot][r];
#pragma unroll
          for (int k = 0; k < 4; ++k) dx[k] += t[k];
          *(f32x4_t*)(out_f32 + (long)row * ldof + c) = (f32x4_t){dx[0], dx[1], dx[2], dx[3]};
          if (out_scaled) {
            const float scl = row_scale ? row_scale[g ? (row - split) / rps1 : row / rps0] : 1.f;
            float t2[4] = {dx[0] * scl, dx[1] * scl, dx[2] * scl, dx[3] * scl};
            st4_lp(out_scaled + (long)row * ldos + c, t2);
          }
        } else {
          st4_lp(out_bf16 + (long)row * ldob + c, dx);
        }
      }
    }
    if (b + 2 < nb) fetch(b + 2, slot);
  };
  int b = 0;
  for (; b + 1 < nb; b += 2) { batch(b, 0); batch(b + 1, 1); }
  if (b < nb) batch(b, 0);
  float* pp = partial + (long)blk * 2 * D;
  *(f32x4_t*)(pp + c) = ag * pscale;
  *(f32x4_t*)(pp + D + c) = ab * pscale;
}

// Wide rows (D >= 2048, e.g. the 3072-wide ffn_layernorm): the 4 waves of a block share ONE row (each lane owns
// D/1024 float4 column groups), so the row costs 12 instead of 48+ live registers per array and the kernel runs at
// full occupancy; the two row statistics cross waves through 32 B of LDS (one barrier per row, double-buffered).
// Each thread owns its columns outright, so dgamma/dbeta need no cross-wave reduction at all.
template <typename TIn, int NITW>
__global__ __launch_bounds__(256) void ln_bwd_wide_kernel(const lp_t* __restrict__ dy, int lddy, const TIn* __restrict__ x, int ldx,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma, int gstride,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                          lp_t* __restrict__ out_bf16, int ldob, const lp_t* __restrict__ gelu_u, int ldu,
                                                          const float* __restrict__ dres, float* __restrict__ out_f32, int ldof,
                                                          lp_t* __restrict__ out_scaled, int ldos, const float* __restrict__ row_scale,
                                                          int rps0, int rps1, int M, int D, int split, int rows_per_block, int blocks0,
                                                          float* __restrict__ partial, float pscale) {
  __shared__ float part[2][4][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool x_is_u = gelu_u != nullptr && (const void*)gelu_u == (const void*)x;
  const int blk = blockIdx.x;
  const int g = blk >= blocks0;
  const int r_begin = g ? split + (blk - blocks0) * rows_per_block : blk * rows_per_block;
  const int r_end = min(r_begin + rows_per_block, g ? M : split);
  const float* gm = gamma + (long)g * gstride;
  float gv[NITW][4], ag[NITW][4], ab[NITW][4];
#pragma unroll
  for (int it = 0; it < NITW; ++it) {
    const int c = (it * 256 + tid) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) { ag[it][k] = 0.f; ab[it][k] = 0.f; gv[it][k] = 0.f; }
    if (c < D) {
      const f32x4_t t = *(const f32x4_t*)(gm + c);
      gv[it][0] = t[0]; gv[it][1] = t[1]; gv[it][2] = t[2]; gv[it][3] = t[3];
    }
  }
  const float invD = 1.f / (float)D;
  int par = 0;
  // software pipeline over rows: the loads of row r+1 (x, dy and the GELU pre-activation) are issued before row r is
  // reduced, so two rows of HBM latency overlap and nothing is loaded behind the barrier
  float xq[NITW][4], dq[NITW][4], uq[NITW][4];
  auto fetch = [&](int row) {
#pragma unroll
    for (int it = 0; it < NITW; ++it) {
      const int c = (it * 256 + tid) * 4;
      if (c < D) {
        if (!x_is_u) Ld4<TIn>::ld(x + (long)row * ldx + c, xq[it]);
        Ld4<lp_t>::ld(dy + (long)row * lddy + c, dq[it]);
        if (gelu_u) Ld4<lp_t>::ld(gelu_u + (long)row * ldu + c, uq[it]);
      }
    }
  };
  if (r_begin < r_end) fetch(r_begin);
  for (int row = r_begin; row < r_end; ++row, par ^= 1) {
    const float mu = mean[row], rs = rstd[row];
    float xh[NITW][4], dyv[NITW][4], uv[NITW][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int it = 0; it < NITW; ++it) {
      const int c = (it * 256 + tid) * 4;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool in = c < D;
        dyv[it][k] = in ? dq[it][k] : 0.f;
        float xval = xq[it][k];
        if (gelu_u) {       // uv <- GELU'(u); with x == u the LayerNorm input g = u*Phi(u) is recomputed as well
          float cdf, pdf;
          gelu_parts(uq[it][k], cdf, pdf);
          uv[it][k] = fmaf(uq[it][k], pdf, cdf);
          if (x_is_u) xval = uq[it][k] * cdf;
        }
        xh[it][k] = in ? (xval - mu) * rs : 0.f;
        const float dg = dyv[it][k] * gv[it][k];
        s1 += dg;
        s2 += dg * xh[it][k];
        ag[it][k] += dyv[it][k] * xh[it][k];
        ab[it][k] += dyv[it][k];
      }
    }
    if (row + 1 < r_end) fetch(row + 1);
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (lane == 0) { part[par][wave][0] = s1; part[par][wave][1] = s2; }
    __syncthreads();
    const float c1 = ((part[par][0][0] + part[par][1][0]) + (part[par][2][0] + part[par][3][0])) * invD;
    const float c2 = ((part[par][0][1] + part[par][1][1]) + (part[par][2][1] + part[par][3][1])) * invD;
    float scl = 1.f;
    if (out_scaled && row_scale) scl = row_scale[g ? (row - split) / rps1 : row / rps0];
#pragma unroll
    for (int it = 0; it < NITW; ++it) {
      const int c = (it * 256 + tid) * 4;
      if (c < D) {
        float dx[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) dx[k] = rs * (dyv[it][k] * gv[it][k] - c1 - xh[it][k] * c2);
        if (out_bf16) {
          if (gelu_u) {
#pragma unroll
            for (int k = 0; k < 4; ++k) dx[k] *= uv[it][k];
          }
          st4_lp(out_bf16 + (long)row * ldob + c, dx);
        }
        if (out_f32) {
          if (dres) {
            const f32x4_t t = *(const f32x4_t*)(dres + (long)row * ldof + c);
#pragma unroll
            for (int k = 0; k < 4; ++k) dx[k] += t[k];
          }
          *(f32x4_t*)(out_f32 + (long)row * ldof + c) = (f32x4_t){dx[0], dx[1], dx[2], dx[3]};
          if (out_scaled) {
            float t[4] = {dx[0] * scl, dx[1] * scl, dx[2] * scl, dx[3] * scl};
            st4_lp(out_scaled + (long)row * ldos + c, t);
          }
        }
      }
    }
  }
  if (partial) {
    float* pp = partial + (long)blk * 2 * D;
#pragma unroll
    for (int it = 0; it < NITW; ++it) {
      const int c = (it * 256 + tid) * 4;
      if (c < D) {
        *(f32x4_t*)(pp + c) = (f32x4_t){ag[it][0], ag[it][1], ag[it][2], ag[it][3]} * pscale;
        *(f32x4_t*)(pp + D + c) = (f32x4_t){ab[it][0], ab[it][1], ab[it][2], ab[it][3]} * pscale;
      }
    }
  } else if (r_begin < r_end && dgamma) {
#pragma unroll
    for (int it = 0; it < NITW; ++it) {
      const int c = (it * 256 + tid) * 4;
      if (c < D) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          atomicAdd(dgamma + (long)g * gstride + c + k, ag[it][k] * pscale);
          atomicAdd(dbeta + (long)g * gstride + c + k, ab[it][k] * pscale);
        }
      }
    }
  }
}

// The hot instance of the wide backward -- ffn_layernorm of the encoder: x == gelu_u (bf16 fc1 pre-activation), bf16 dy,
// bf16 dx, two-stage dgamma / dbeta.  Same 4-waves-per-row scheme as above, but the kernel is latency-bound (one row of
// loads in flight per block is ~12 KB; measured 2.5 TB/s), so rows are prefetched TWO ahead and kept as raw bf16 pairs
// (2 VGPRs per 4 values instead of 4) until they are consumed: twice the bytes in flight for the same registers.
template <int NITW>
__global__ __launch_bounds__(256) void ln_bwd_ffn_kernel(const lp_t* __restrict__ dy, int lddy, const lp_t* __restrict__ u, int ldu,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ gamma, int gstride,
                                                         lp_t* __restrict__ out, int ldo, int M, int D, int split,
                                                         int rows_per_block, int blocks0, float* __restrict__ partial,
                                                         float pscale) {
  // D == NITW * 1024 (launcher).  The arithmetic is packed fp32, two values per instruction: the kernel is bound by VALU issue
  // (GELU, GELU' and the LayerNorm backward are ~33 operations per element, two of them transcendental)
  __shared__ float part[2][4][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int blk = blockIdx.x;
  const int g = blk >= blocks0;
  const int r_begin = g ? split + (blk - blocks0) * rows_per_block : blk * rows_per_block;
  const int r_end = min(r_begin + rows_per_block, g ? M : split);
  const float* gm = gamma + (long)g * gstride;
  hw_f32x2_t gv[NITW][2], ag[NITW][2], ab[NITW][2];
#pragma unroll
  for (int it = 0; it < NITW; ++it) {
    const int c = (it * 256 + tid) * 4;
    const f32x4_t t = *(const f32x4_t*)(gm + c);
    gv[it][0] = (hw_f32x2_t){t[0], t[1]};
    gv[it][1] = (hw_f32x2_t){t[2], t[3]};
#pragma unroll
    for (int h = 0; h < 2; ++h) { ag[it][h] = (hw_f32x2_t){0.f, 0.f}; ab[it][h] = (hw_f32x2_t){0.f, 0.f}; }
  }
  const float invD = 1.f / (float)D;
  u32x2_t dq[2][NITW], uq[2][NITW];
  float mu_q[2], rs_q[2];
  auto fetch = [&](int row, int slot) {
    mu_q[slot] = mean[row];
    rs_q[slot] = rstd[row];
#pragma unroll
    for (int it = 0; it < NITW; ++it) {
      const int c = (it * 256 + tid) * 4;
      dq[slot][it] = *(const u32x2_t*)(dy + (long)row * lddy + c);
      uq[slot][it] = *(const u32x2_t*)(u + (long)row * ldu + c);
    }
  };
  if (r_begin < r_end) fetch(r_begin, 0);
  if (r_begin + 1 < r_end) fetch(r_begin + 1, 1);
  auto row_body = [&](int row, int par) {
    const float mu = mu_q[par], rs = rs_q[par];
    const hw_f32x2_t mu2 = {mu, mu}, rs2 = {rs, rs};
    hw_f32x2_t xh[NITW][2], dg[NITW][2], uv[NITW][2];
    hw_f32x2_t s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
#pragma unroll
    for (int it = 0; it < NITW; ++it) {
      const u32x2_t dr = dq[par][it], ur = uq[par][it];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float d0, d1, u0, u1;
        unpack_lp2(dr[h], d0, d1);
        unpack_lp2(ur[h], u0, u1);
        const hw_f32x2_t dv = {d0, d1}, uu = {u0, u1};
        hw_f32x2_t cdf, pdf;
        gelu_parts2(uu, cdf, pdf);
        uv[it][h] = __builtin_elementwise_fma(uu, pdf, cdf);         // GELU'(u)
        xh[it][h] = (uu * cdf - mu2) * rs2;                            // LayerNorm input g = u * Phi(u), normalised
        dg[it][h] = dv * gv[it][h];
        s1 += dg[it][h];
        s2 = __builtin_elementwise_fma(dg[it][h], xh[it][h], s2);
        ag[it][h] = __builtin_elementwise_fma(dv, xh[it][h], ag[it][h]);
        ab[it][h] += dv;
      }
    }
    if (row + 2 < r_end) fetch(row + 2, par);
    const float t1 = wave_sum(s1[0] + s1[1]), t2 = wave_sum(s2[0] + s2[1]);
    if (lane == 0) { part[par][wave][0] = t1; part[par][wave][1] = t2; }
    __syncthreads();
    const float c1 = ((part[par][0][0] + part[par][1][0]) + (part[par][2][0] + part[par][3][0])) * invD;
    const float c2 = ((part[par][0][1] + part[par][1][1]) + (part[par][2][1] + part[par][3][1])) * invD;
    const hw_f32x2_t c1v = {c1, c1}, nc2 = {-c2, -c2};
#pragma unroll
    for (int it = 0; it < NITW; ++it) {
      const int c = (it * 256 + tid) * 4;
      unsigned w[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const hw_f32x2_t dx = (rs2 * (__builtin_elementwise_fma(xh[it][h], nc2, dg[it][h]) - c1v)) * uv[it][h];
        w[h] = pack_lp2(dx[0], dx[1]);
      }
      *(u32x2_t*)(out + (long)row * ldo + c) = (u32x2_t){w[0], w[1]};
    }
  };
  // two rows per trip so that the prefetch slot is a compile-time index (a runtime-indexed register array would go
  // to scratch)
  int row = r_begin;
  for (; row + 1 < r_end; row += 2) { row_body(row, 0); row_body(row + 1, 1); }
  if (row < r_end) row_body(row, 0);
  float* pp = partial + (long)blk * 2 * D;
#pragma unroll
  for (int it = 0; it < NITW; ++it) {
    const int c = (it * 256 + tid) * 4;
    *(f32x4_t*)(pp + c) = (f32x4_t){ag[it][0][0], ag[it][0][1], ag[it][1][0], ag[it][1][1]} * pscale;
    *(f32x4_t*)(pp + D + c) = (f32x4_t){ab[it][0][0], ab[it][0][1], ab[it][1][0], ab[it][1][1]} * pscale;
  }
}

// second stage: dgamma[g][c] += sum over the blocks of group g of partial[blk][0][c] (same for dbeta)
__global__ __launch_bounds__(1024) void ln_param_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int gstride, int D, int blocks0,
                                                              int blocks1) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int which = blockIdx.y & 1, g = blockIdx.y >> 1;
  const int b0 = g ? blocks0 : 0, b1 = g ? blocks0 + blocks1 : blocks0;
  __shared__ float red[16][64];
  float s = 0.f;
  if (c < D) {
    // 4 independent loads in flight per lane (a serial walk over ~100 partial rows per wave was pure latency: 38 us)
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = b0 + (threadIdx.x >> 6);
    for (; b + 48 < b1; b += 64) {
      s += partial[((long)b * 2 + which) * D + c];
      s1 += partial[((long)(b + 16) * 2 + which) * D + c];
      s2 += partial[((long)(b + 32) * 2 + which) * D + c];
      s3 += partial[((long)(b + 48) * 2 + which) * D + c];
    }
    for (; b < b1; b += 16) s += partial[((long)b * 2 + which) * D + c];
    s += (s1 + s2) + s3;
  }
  red[threadIdx.x >> 6][threadIdx.x & 63] = s;
  __syncthreads();
  if (threadIdx.x < 64 && c < D && b1 > b0) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w][threadIdx.x];
    float* dst = (which ? dbeta : dgamma) + (long)g * gstride + c;
    *dst += t;
  }
}

// the second stage of SEVERAL LayerNorm backward calls in one launch (blockIdx.z = call): a training step's 49 two-stage
// reductions were 49 launches of ~7 us on 48 workgroups each (0.36 ms per step, r03_f_kernel_stats.csv)
struct LnReduceTable { simvg_ln_reduce_desc d[SIMVG_LN_REDUCE_MAX]; };
__global__ __launch_bounds__(1024) void ln_param_reduce_batched_kernel(LnReduceTable t) {
  const simvg_ln_reduce_desc& e = t.d[blockIdx.z];
  const int D = e.D;
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  if (blockIdx.x * 64 >= D) return;
  const int which = blockIdx.y & 1, g = blockIdx.y >> 1;
  const int b0 = g ? e.blocks0 : 0, b1 = g ? e.blocks0 + e.blocks1 : e.blocks0;
  const float* partial = e.partial;
  __shared__ float red[16][64];
  float s = 0.f;
  if (c < D) {
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = b0 + (threadIdx.x >> 6);
    for (; b + 48 < b1; b += 64) {
      s += partial[((long)b * 2 + which) * D + c];
      s1 += partial[((long)(b + 16) * 2 + which) * D + c];
      s2 += partial[((long)(b + 32) * 2 + which) * D + c];
      s3 += partial[((long)(b + 48) * 2 + which) * D + c];
    }
    for (; b < b1; b += 16) s += partial[((long)b * 2 + which) * D + c];
    s += (s1 + s2) + s3;
  }
  red[threadIdx.x >> 6][threadIdx.x & 63] = s;
  __syncthreads();
  if (threadIdx.x < 64 && c < D && b1 > b0) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) v += red[w][threadIdx.x];
    float* dst = (which ? e.dbeta : e.dgamma) + (long)g * e.group_stride + c;
    *dst += v;
  }
}

}  // namespace

#define LN_DISPATCH_NIT(D, CALL)                                     \
  do {                                                               \
    const int nit__ = ((D) + 255) / 256;                             \
    if (nit__ <= 1) { CALL(1); }                                     \
    else if (nit__ <= 3) { CALL(3); }                                \
    else if (nit__ <= 4) { CALL(4); }                                \
    else if (nit__ <= 8) { CALL(8); }                                \
    else if (nit__ <= 12) { CALL(12); }                              \
    else { CALL(16); }                                               \
  } while (0)

#ifndef LN_FWD_RW
#define LN_FWD_RW 2
#endif
static inline bool ln_rows_off() { static const bool off = getenv("SIMVG_LN_ROWS") && atoi(getenv("SIMVG_LN_ROWS")) == 0; return off; }
static inline int ln_wide_rpb(int M) { return std::max(16, cdiv(M, 508)); }
// ln_bwd_tile_kernel: R rows per batch (two batches in flight); rows per block so that the grid is ~ LN_TILE_BLOCKS blocks
// (sweep at M = 26 944, us residual-stream / sub-LN instance incl. the reduction launch, +-5 us run to run: R = 2/4, 3/6, 4/8, 5/10 at
// 1280 blocks: 60.5/41.0, 60.5-68.7/41.3, 72.6/46.0, 72.8/52.9;  R = 3/6 at 1024, 1536, 2048, 2560 blocks: 60.1/41.7, 71.4/42.9,
// 69.9/43.0, 70.6/45.2;  the wave-per-row kernels: 88.3 / 64.5)
#ifndef LN_TILE_R_RES
#define LN_TILE_R_RES 3
#endif
#ifndef LN_TILE_R_SUB
#define LN_TILE_R_SUB 6
#endif
#ifndef LN_TILE_BLOCKS
#define LN_TILE_BLOCKS 1024
#endif
static inline int ln_tile_rpb(int M, int R) { return std::max(2 * R, cdiv(cdiv(M, LN_TILE_BLOCKS), R) * R); }
static inline bool ln_tile_off() { static const bool off = getenv("SIMVG_LN_TILE") && atoi(getenv("SIMVG_LN_TILE")) == 0; return off; }
// narrow rows: 32 rows per block (4 waves x 8 rows) at training sizes; the decoder head's [B * num_queries, 256] problems have
// 64 - 640 rows in all -- 4 rows per block (one per wave) so that they spread over 16 - 160 blocks instead of 2 - 20
static inline int ln_narrow_rpb(int M) { return M <= 2048 ? 4 : 32; }

extern "C" int simvg_ln_fwd(const void* x, int x_is_bf16, int ldx, const float* gamma, const float* beta,
                            int group_stride, void* y_bf16, int ldy, float* y_f32, int ldy32, float* mean,
                            float* rstd, int M, int D, int split, float eps, int x_is_gelu_preact,
                            hipStream_t stream) {
  SIMVG_CHECK_ARG(M > 0 && D > 0 && D % 4 == 0 && D <= 4096, "ln_fwd: D must be a multiple of 4 and <= 4096");
  SIMVG_CHECK_ARG(ldx % 4 == 0 && (y_bf16 == nullptr || ldy % 4 == 0), "ln_fwd: leading dims must be multiples of 4");
  SIMVG_CHECK_ARG(y_bf16 || y_f32, "ln_fwd: no output");
  if (split == 0) split = M;
  const dim3 grid(cdiv(M, 4)), block(256);
  if (x_is_bf16 && y_bf16 && !y_f32 && D % 512 == 0 && D >= 1024 && ldx % 8 == 0 && ldy % 8 == 0 && group_stride % 4 == 0) {
#define WCALL(N_, G_)                                                                                                  \
    hipLaunchKernelGGL((ln_fwd_wide_kernel<N_, G_, LN_WIDE_RW>), dim3(cdiv(M, 4 * LN_WIDE_RW)), block, 0, stream,      \
                       (const lp_t*)x, ldx, gamma, beta, group_stride, (lp_t*)y_bf16, ldy, mean, rstd, M, split, eps)
    const int n8 = D / 512;
    bool done = true;
    if (x_is_gelu_preact) {
      if (n8 == 2) WCALL(2, true); else if (n8 == 4) WCALL(4, true); else if (n8 == 6) WCALL(6, true);
      else if (n8 == 8) WCALL(8, true); else done = false;
    } else {
      if (n8 == 2) WCALL(2, false); else if (n8 == 4) WCALL(4, false); else if (n8 == 6) WCALL(6, false);
      else if (n8 == 8) WCALL(8, false); else done = false;
    }
#undef WCALL
    if (done) {
      SIMVG_LAUNCH_CHECK();
      return SIMVG_OK;
    }
  }
  // (at every M: the same arithmetic for a row whatever the batch it arrives in -- predictions are bit-identical across batch
  // splits, tests/test_properties_gpu.py)
  if (y_bf16 && !y_f32 && !x_is_gelu_preact && (D == 768 || D == 1024) && group_stride % 4 == 0 && !ln_rows_off()) {
    constexpr int RW = LN_FWD_RW;
    const dim3 rgrid(cdiv(M, 4 * RW));
#define RCALL(T_, N_)                                                                                                  \
    hipLaunchKernelGGL((ln_fwd_rows_kernel<T_, N_, RW>), rgrid, block, 0, stream, (const T_*)x, ldx, gamma, beta,      \
                       group_stride, (lp_t*)y_bf16, ldy, mean, rstd, M, split, eps)
    if (x_is_bf16) { if (D == 768) RCALL(lp_t, 3); else RCALL(lp_t, 4); }
    else { if (D == 768) RCALL(float, 3); else RCALL(float, 4); }
#undef RCALL
    SIMVG_LAUNCH_CHECK();
    return SIMVG_OK;
  }
#define CALL(N_)                                                                                                   \
  if (x_is_bf16)                                                                                                   \
    hipLaunchKernelGGL((ln_fwd_kernel<lp_t, N_>), grid, block, 0, stream, (const lp_t*)x, ldx, gamma, beta,    \
                       group_stride, (lp_t*)y_bf16, ldy, y_f32, ldy32, mean, rstd, M, D, split, eps,             \
                       x_is_gelu_preact);                                                                          \
  else                                                                                                             \
    hipLaunchKernelGGL((ln_fwd_kernel<float, N_>), grid, block, 0, stream, (const float*)x, ldx, gamma, beta,      \
                       group_stride, (lp_t*)y_bf16, ldy, y_f32, ldy32, mean, rstd, M, D, split, eps,             \
                       x_is_gelu_preact)
  LN_DISPATCH_NIT(D, CALL);
#undef CALL
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}

// `defer` (optional): the second stage of the two-stage dgamma / dbeta reduction is NOT launched; its description is written to
// *defer for simvg_ln_param_reduce_batched (requires partial_ws; a call that takes the atomic path leaves defer->partial = 0)
static int ln_bwd_impl(const void* dy_bf16, int dy_is_f32, int lddy, const void* x, int x_is_bf16, int ldx, const float* mean,
                       const float* rstd, const float* gamma, int group_stride, float* dgamma, float* dbeta,
                       void* dx_bf16, int lddxb, const void* gelu_u_bf16, int ldu, const float* dres,
                       float* dx_f32, int lddxf, void* dx_scaled_bf16, int lddxs, const float* row_scale,
                       int rows_per_sample0, int rows_per_sample1, int M, int D, int split,
                       float* partial_ws, float dy_scale, float param_scale, simvg_ln_reduce_desc* defer, hipStream_t stream);

extern "C" int simvg_ln_bwd(const void* dy_bf16, int dy_is_f32, int lddy, const void* x, int x_is_bf16, int ldx, const float* mean,
                            const float* rstd, const float* gamma, int group_stride, float* dgamma, float* dbeta,
                            void* dx_bf16, int lddxb, const void* gelu_u_bf16, int ldu, const float* dres,
                            float* dx_f32, int lddxf, void* dx_scaled_bf16, int lddxs, const float* row_scale,
                            int rows_per_sample0, int rows_per_sample1, int M, int D, int split,
                            float* partial_ws, float dy_scale, float param_scale, hipStream_t stream) {
  return ln_bwd_impl(dy_bf16, dy_is_f32, lddy, x, x_is_bf16, ldx, mean, rstd, gamma, group_stride, dgamma, dbeta, dx_bf16, lddxb,
                     gelu_u_bf16, ldu, dres, dx_f32, lddxf, dx_scaled_bf16, lddxs, row_scale, rows_per_sample0, rows_per_sample1,
                     M, D, split, partial_ws, dy_scale, param_scale, nullptr, stream);
}

extern "C" int simvg_ln_bwd_deferred(const void* dy_bf16, int dy_is_f32, int lddy, const void* x, int x_is_bf16, int ldx,
                                     const float* mean, const float* rstd, const float* gamma, int group_stride, float* dgamma,
                                     float* dbeta, void* dx_bf16, int lddxb, const void* gelu_u_bf16, int ldu, const float* dres,
                                     float* dx_f32, int lddxf, void* dx_scaled_bf16, int lddxs, const float* row_scale,
                                     int rows_per_sample0, int rows_per_sample1, int M, int D, int split,
                                     float* partial_ws, float dy_scale, float param_scale, simvg_ln_reduce_desc* desc_out,
                                     hipStream_t stream) {
  SIMVG_CHECK_ARG(desc_out && partial_ws, "ln_bwd_deferred: needs the descriptor to fill and the partial workspace");
  desc_out->partial = nullptr;
  return ln_bwd_impl(dy_bf16, dy_is_f32, lddy, x, x_is_bf16, ldx, mean, rstd, gamma, group_stride, dgamma, dbeta, dx_bf16, lddxb,
                     gelu_u_bf16, ldu, dres, dx_f32, lddxf, dx_scaled_bf16, lddxs, row_scale, rows_per_sample0, rows_per_sample1,
                     M, D, split, partial_ws, dy_scale, param_scale, desc_out, stream);
}

extern "C" int simvg_ln_param_reduce_batched(const simvg_ln_reduce_desc* descs, int n, hipStream_t stream) {
  SIMVG_CHECK_ARG(descs && n > 0 && n <= SIMVG_LN_REDUCE_MAX, "ln_param_reduce_batched: 1 .. SIMVG_LN_REDUCE_MAX descriptors");
  LnReduceTable t;
  int m = 0, dmax = 0;
  for (int i = 0; i < n; ++i) {
    if (!descs[i].partial) continue;          // that call reduced with atomics: nothing left to do
    SIMVG_CHECK_ARG(descs[i].dgamma && descs[i].dbeta && descs[i].D > 0, "ln_param_reduce_batched: incomplete descriptor");
    t.d[m++] = descs[i];
    dmax = descs[i].D > dmax ? descs[i].D : dmax;
  }
  if (m == 0) return SIMVG_OK;
  hipLaunchKernelGGL(ln_param_reduce_batched_kernel, dim3(cdiv(dmax, 64), 4, m), dim3(1024), 0, stream, t);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}

static int ln_bwd_impl(const void* dy_bf16, int dy_is_f32, int lddy, const void* x, int x_is_bf16, int ldx, const float* mean,
                       const float* rstd, const float* gamma, int group_stride, float* dgamma, float* dbeta,
                       void* dx_bf16, int lddxb, const void* gelu_u_bf16, int ldu, const float* dres,
                       float* dx_f32, int lddxf, void* dx_scaled_bf16, int lddxs, const float* row_scale,
                       int rows_per_sample0, int rows_per_sample1, int M, int D, int split,
                       float* partial_ws, float dy_scale, float param_scale, simvg_ln_reduce_desc* defer, hipStream_t stream) {
  // second stage of the two-stage reduction: launched here, or described for the batched launch
  auto second_stage = [&](int b0, int b1) {
    if (defer) {
      *defer = simvg_ln_reduce_desc{partial_ws, dgamma, dbeta, group_stride, D, b0, b1};
    } else {
      hipLaunchKernelGGL(ln_param_reduce_kernel, dim3(cdiv(D, 64), 4), dim3(1024), 0, stream, partial_ws, dgamma, dbeta,
                         group_stride, D, b0, b1);
    }
  };
  SIMVG_CHECK_ARG(M > 0 && D > 0 && D % 4 == 0 && D <= 4096, "ln_bwd: D must be a multiple of 4 and <= 4096");
  SIMVG_CHECK_ARG(dy_is_f32 || dy_scale == 1.0f, "ln_bwd: dy_scale applies to an fp32 dy only (entry of a scaled backward)");
  SIMVG_CHECK_ARG(dx_bf16 || dx_f32, "ln_bwd: no output");
  SIMVG_CHECK_ARG(!dy_is_f32 || !x_is_bf16, "ln_bwd: fp32 dy goes with fp32 x (decoder head, exact-fp32 encoder mode)");
  SIMVG_CHECK_ARG(!(dx_scaled_bf16 && !dx_f32), "ln_bwd: scaled bf16 copy requires the f32 output");
  if (split == 0) split = M;
  // partial_ws (optional, >= simvg_ln_bwd_ws_floats(M, D, split) floats): two-stage dgamma/dbeta reduction instead of
  // global atomics
  // rows per block: the wide two-stage kernel gets its grid into one residency round (2 blocks per CU; >= 16 rows),
  // 32 for the generic wave-per-row kernels (sweeps: profiles/r01_sweeps.md)
  const int rpb = (D >= 2048 && partial_ws) ? ln_wide_rpb(M) : ln_narrow_rpb(M);
  const int blocks0 = cdiv(split, rpb), blocks1 = cdiv(M - split, rpb);
  const dim3 grid(blocks0 + blocks1), block(256);
  const size_t shm = (size_t)2 * D * sizeof(float);
  const int rps0 = rows_per_sample0 > 0 ? rows_per_sample0 : 1, rps1 = rows_per_sample1 > 0 ? rows_per_sample1 : 1;
  if (dy_is_f32) {      // wave-per-row kernel at every width (the exact mode is a parity mode, not a fast one)
#define FCALL(N_)                                                                                                       \
    hipLaunchKernelGGL((ln_bwd_kernel<float, float, N_>), grid, block, shm, stream, (const float*)dy_bf16, lddy,        \
                       (const float*)x, ldx, mean, rstd, gamma, group_stride, dgamma, dbeta, (lp_t*)dx_bf16, lddxb,   \
                       (const lp_t*)gelu_u_bf16, ldu, dres, dx_f32, lddxf, (lp_t*)dx_scaled_bf16, lddxs, row_scale, \
                       rps0, rps1, M, D, split, rpb, blocks0, partial_ws, dy_scale, param_scale)
    LN_DISPATCH_NIT(D, FCALL);
#undef FCALL
    if (partial_ws) second_stage(blocks0, blocks1);
    SIMVG_LAUNCH_CHECK();
    return SIMVG_OK;
  }
  if (D >= 2048) {
    const int nitw = (D + 1023) / 1024;
#define WCALL(T_, N_)                                                                                                   \
    hipLaunchKernelGGL((ln_bwd_wide_kernel<T_, N_>), grid, block, 0, stream, (const lp_t*)dy_bf16, lddy, (const T_*)x, \
                       ldx, mean, rstd, gamma, group_stride, dgamma, dbeta, (lp_t*)dx_bf16, lddxb,                    \
                       (const lp_t*)gelu_u_bf16, ldu, dres, dx_f32, lddxf, (lp_t*)dx_scaled_bf16, lddxs, row_scale, \
                       rps0, rps1, M, D, split, rpb, blocks0, partial_ws, param_scale)
    const bool ffn = x_is_bf16 && gelu_u_bf16 && gelu_u_bf16 == x && dx_bf16 && !dx_f32 && !dres && partial_ws &&
                     (nitw == 3 || nitw == 4) && D == nitw * 1024;
#define FCALL_FFN(N_)                                                                                                   \
    hipLaunchKernelGGL((ln_bwd_ffn_kernel<N_>), grid, block, 0, stream, (const lp_t*)dy_bf16, lddy, (const lp_t*)x,  \
                       ldx, mean, rstd, gamma, group_stride, (lp_t*)dx_bf16, lddxb, M, D, split, rpb, blocks0, partial_ws, param_scale)
    if (ffn) { if (nitw == 3) FCALL_FFN(3); else FCALL_FFN(4); }
    else if (x_is_bf16) { if (nitw <= 2) WCALL(lp_t, 2); else if (nitw == 3) WCALL(lp_t, 3); else WCALL(lp_t, 4); }
    else { if (nitw <= 2) WCALL(float, 2); else if (nitw == 3) WCALL(float, 3); else WCALL(float, 4); }
#undef WCALL
#undef FCALL_FFN
    if (partial_ws) second_stage(blocks0, blocks1);
    SIMVG_LAUNCH_CHECK();
    return SIMVG_OK;
  }
  {
    const bool res = !x_is_bf16 && dres && dx_f32 && !dx_bf16;
    const bool sub = x_is_bf16 && dx_bf16 && !dx_f32 && !dres;
    const int nit_pf = (D + 255) / 256;
    if (dy_bf16 && !gelu_u_bf16 && partial_ws && (D == 768 || D == 1024) && (res || sub) && M >= 1024 && !ln_tile_off()) {
      const int R = res ? LN_TILE_R_RES : LN_TILE_R_SUB;
      const int rpb_t = ln_tile_rpb(M, R);
      const int tb0 = cdiv(split, rpb_t), tb1 = cdiv(M - split, rpb_t);
#define TCALL(T_, RES_, NW_, R_)                                                                                        \
      hipLaunchKernelGGL((ln_bwd_tile_kernel<T_, RES_, NW_, R_>), dim3(tb0 + tb1), dim3(NW_ * 64), 0, stream,          \
                         (const lp_t*)dy_bf16, lddy, (const T_*)x, ldx, mean, rstd, gamma, group_stride, (lp_t*)dx_bf16, lddxb, \
                         dres, dx_f32, lddxf, (lp_t*)dx_scaled_bf16, lddxs, row_scale, rps0, rps1, M, split, rpb_t, tb0,  \
                         partial_ws, param_scale)
      if (res) { if (D == 768) TCALL(float, true, 3, LN_TILE_R_RES); else TCALL(float, true, 4, LN_TILE_R_RES); }
      else { if (D == 768) TCALL(lp_t, false, 3, LN_TILE_R_SUB); else TCALL(lp_t, false, 4, LN_TILE_R_SUB); }
#undef TCALL
      second_stage(tb0, tb1);
      SIMVG_LAUNCH_CHECK();
      return SIMVG_OK;
    }
    if (dy_bf16 && !gelu_u_bf16 && !partial_ws && (nit_pf == 3 || nit_pf == 4) && (res || sub)) {
      // rows per block: the whole grid in ONE residency round (2-3 blocks of 4 waves per CU) -- with 32 rows the 842
      // blocks of a B=64 step took two rounds, the second one a third full (sweep: 32 -> 93.7 / 64.9 us, 53-64 ->
      // 77.4 / 46.6 us for the residual-stream / sub-LN instance)
      const int rpb_pf = std::max(32, (cdiv(M, 480) + 3) / 4 * 4);
      const int pf_blocks0 = cdiv(split, rpb_pf), pf_blocks1 = cdiv(M - split, rpb_pf);
      const dim3 pf_grid(pf_blocks0 + pf_blocks1);
#define PFCALL(T_, R_)                                                                                                  \
      if (nit_pf == 3) PFCALL_N(T_, R_, 3); else PFCALL_N(T_, R_, 4)
#define PFCALL_N(T_, R_, N_)                                                                                            \
      hipLaunchKernelGGL((ln_bwd_pf_kernel<T_, R_, N_>), pf_grid, block, shm, stream, (const lp_t*)dy_bf16, lddy,    \
                         (const T_*)x, ldx, mean, rstd, gamma, group_stride, dgamma, dbeta, (lp_t*)dx_bf16, lddxb,    \
                         dres, dx_f32, lddxf, (lp_t*)dx_scaled_bf16, lddxs, row_scale, rps0, rps1, M, D, split,       \
                         rpb_pf, pf_blocks0, param_scale)
      if (res) { PFCALL(float, true); } else { PFCALL(lp_t, false); }
#undef PFCALL
#undef PFCALL_N
      SIMVG_LAUNCH_CHECK();
      return SIMVG_OK;
    }
  }
#define CALL(N_)                                                                                                        \
  if (x_is_bf16)                                                                                                        \
    hipLaunchKernelGGL((ln_bwd_kernel<lp_t, lp_t, N_>), grid, block, shm, stream, (const lp_t*)dy_bf16, lddy,             \
                       (const lp_t*)x, ldx, mean, rstd, gamma, group_stride, dgamma, dbeta, (lp_t*)dx_bf16, lddxb,  \
                       (const lp_t*)gelu_u_bf16, ldu, dres, dx_f32, lddxf, (lp_t*)dx_scaled_bf16, lddxs, row_scale, \
                       rps0, rps1, M, D, split, rpb, blocks0, partial_ws, 1.0f, param_scale);                                       \
  else                                                                                                                  \
    hipLaunchKernelGGL((ln_bwd_kernel<float, lp_t, N_>), grid, block, shm, stream, (const lp_t*)dy_bf16, lddy,              \
                       (const float*)x, ldx, mean, rstd, gamma, group_stride, dgamma, dbeta, (lp_t*)dx_bf16, lddxb,   \
                       (const lp_t*)gelu_u_bf16, ldu, dres, dx_f32, lddxf, (lp_t*)dx_scaled_bf16, lddxs, row_scale, \
                       rps0, rps1, M, D, split, rpb, blocks0, partial_ws, 1.0f, param_scale)
  LN_DISPATCH_NIT(D, CALL);
#undef CALL
  if (partial_ws) second_stage(blocks0, blocks1);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}

extern "C" long simvg_ln_bwd_ws_floats(int M, int D, int split) {
  if (split == 0) split = M;
  const int rpb = D >= 2048 ? ln_wide_rpb(M) : ln_narrow_rpb(M);
  long blocks = cdiv(split, rpb) + cdiv(M - split, rpb);
  if (D == 768 || D == 1024) {      // ln_bwd_tile_kernel's grid (the smaller R gives the larger one)
    const int rt = ln_tile_rpb(M, std::min(LN_TILE_R_RES, LN_TILE_R_SUB));
    blocks = std::max<long>(blocks, cdiv(split, rt) + cdiv(M - split, rt));
  }
  return blocks * 2 * D;
}
