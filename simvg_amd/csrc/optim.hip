// Fused optimizer step over a FLAT fp32 parameter arena (gfx950): global-norm clip scale + Adam (amsgrad optional) in
// ONE pass -- 5 streams read (p, g, m, v, vmax), 4 written: 36 B per parameter, against ~9 foreach passes / the
// chunked multi-tensor kernel of the framework optimizer (measured 2.2 ms + 0.7 ms clip for the 108 M parameters of
// ViT-B SimVG; this kernel is HBM-bound at ~0.9 ms).
//
// Replaces, for the encoder arena, torch.nn.utils.clip_grad_norm_(model.parameters(), 0.15) + torch.optim.Adam.step()
// of the reference's apis/train.py:81-83 / core/optimizer.py:52-68 (betas (0.9, 0.98), eps 1e-9, amsgrad=True,
// weight_decay 0 in every config).  Same arithmetic as torch's Adam, in fp32:
//   g' = g * min(1, max_norm / (total_norm + 1e-6))            (clip_grad_norm_)
//   g' += wd * p                                                (L2 weight decay, Adam not AdamW)
//   m  = m + (1 - b1) * (g' - m)                                (exp_avg.lerp_)
//   v  = b2 * v + (1 - b2) * g' * g'
//   vmax = max(vmax, v);  denom = sqrt(vmax) / sqrt(1 - b2^t) + eps
//   p  = p - (lr / (1 - b1^t)) * m / denom
#include "common.h"
#include <stdlib.h>

namespace {

template <bool NT> __device__ __forceinline__ f32x4_t ld4(const float* p, long i) {
  if constexpr (NT) return __builtin_nontemporal_load((const f32x4_t*)p + i);
  else return ((const f32x4_t*)p)[i];
}
template <bool NT> __device__ __forceinline__ void st4(float* p, long i, f32x4_t v) {
  if constexpr (NT) __builtin_nontemporal_store(v, (f32x4_t*)p + i);
  else ((f32x4_t*)p)[i] = v;
}

// Two launches, no atomics: every block leaves ONE partial sum, a single wave adds the <= 2048 partials in a fixed order.
// The norm (and with it the clip coefficient, active on almost every step at max_norm 0.15) is therefore bit-identical
// from run to run and between data-parallel replicas that hold identical gradients.
template <bool NT>
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n4, float* __restrict__ partial) {
  float s = 0.f;
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  // 4 independent 16-B loads in flight per thread: one per trip left the read stream latency-bound
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const f32x4_t a = ld4<NT>(g, i), b = ld4<NT>(g, i + stride), c = ld4<NT>(g, i + 2 * stride), d = ld4<NT>(g, i + 3 * stride);
    s += ((a[0] * a[0] + a[1] * a[1]) + (a[2] * a[2] + a[3] * a[3])) + ((b[0] * b[0] + b[1] * b[1]) + (b[2] * b[2] + b[3] * b[3]));
    s += ((c[0] * c[0] + c[1] * c[1]) + (c[2] * c[2] + c[3] * c[3])) + ((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]));
  }
  for (; i < n4; i += stride) {
    const f32x4_t v = ld4<NT>(g, i);
    s += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
  }
  __shared__ float red[4];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(64) void sumsq_finish_kernel(const float* __restrict__ partial, int n, float* __restrict__ out) {
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) s += partial[i];      // lane l: partials l, l + 64, ... in order
  s = wave_sum(s);
  if (threadIdx.x == 0) *out += s;
}

struct AdamArgs {
  float* p; const float* g; float* m; float* v; float* vmax;
  long n4;
  float step_size, bc2_sqrt, beta1, beta2, eps, weight_decay;
  const float* total_norm; float max_norm;     // clip (total_norm == nullptr: no clip)
  int amsgrad;
};

constexpr int STREAM_GRID = 768;      // 3 blocks of 256 threads per CU of an MI355X (256 CUs), all resident

// (non-temporal loads / stores on the moment streams: no gain at 141 M elements, -15 % at 370 M -- default policy)
// U float4 chunks per thread and trip.  PHASED: the g / m / v loads of all U chunks first, then -- only for chunks that are
// not all-zero -- p and vmax (two dependent round trips, 12 B instead of 20 B read for never-touched rows); !PHASED: all five
// streams of all U chunks in flight at once.
template <int U, bool PHASED>
__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
  constexpr bool NT = false;
  float coef = 1.f;
  if (a.total_norm) coef = fminf(a.max_norm / (*a.total_norm + 1e-6f), 1.f);
  const float w1 = 1.f - a.beta1, w2 = 1.f - a.beta2;
  const long stride = (long)gridDim.x * 256;
  for (long i0 = (long)blockIdx.x * 256 + threadIdx.x; i0 < a.n4; i0 += stride * U) {
    f32x4_t g[U], m[U], v[U], p[U], vm[U];
    bool live[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = i0 + u * stride;
      live[u] = i < a.n4;
      if (live[u]) {
        g[u] = ld4<NT>(a.g, i); m[u] = ld4<NT>(a.m, i); v[u] = ld4<NT>(a.v, i);
        if (!PHASED) { p[u] = ld4<NT>(a.p, i); vm[u] = a.amsgrad ? ld4<NT>(a.vmax, i) : v[u]; }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      // never-touched parameters (embedding rows of tokens that have not occurred yet: 49 M of the 223 M elements of ViT-B's arena are
      // the 64 010-row text table): g = m = v = 0 makes the update exactly zero and leaves the state unchanged -> skip the
      // rest of the traffic.  (v = 0 implies vmax = 0; weight decay would move them, so only without it.)
      if (live[u] && a.weight_decay == 0.f && g[u][0] == 0.f && g[u][1] == 0.f && g[u][2] == 0.f && g[u][3] == 0.f &&
          m[u][0] == 0.f && m[u][1] == 0.f && m[u][2] == 0.f && m[u][3] == 0.f && v[u][0] == 0.f && v[u][1] == 0.f &&
          v[u][2] == 0.f && v[u][3] == 0.f)
        live[u] = false;
      if (PHASED && live[u]) {
        const long i = i0 + u * stride;
        p[u] = ld4<NT>(a.p, i);
        vm[u] = a.amsgrad ? ld4<NT>(a.vmax, i) : v[u];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!live[u]) continue;
      const long i = i0 + u * stride;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float gk = g[u][k] * coef;
        if (a.weight_decay != 0.f) gk = fmaf(a.weight_decay, p[u][k], gk);
        m[u][k] = m[u][k] + w1 * (gk - m[u][k]);
        v[u][k] = a.beta2 * v[u][k] + w2 * gk * gk;
        float d2 = v[u][k];
        if (a.amsgrad) { vm[u][k] = fmaxf(vm[u][k], v[u][k]); d2 = vm[u][k]; }
        const float denom = sqrtf(d2) / a.bc2_sqrt + a.eps;
        p[u][k] = p[u][k] - a.step_size * (m[u][k] / denom);
      }
      ((f32x4_t*)a.p)[i] = p[u];       // the parameters are re-read by the weight refresh right after: default policy
      st4<NT>(a.m, i, m[u]);
      st4<NT>(a.v, i, v[u]);
      if (a.amsgrad) st4<NT>(a.vmax, i, vm[u]);
    }
  }
}

}  // namespace

extern "C" int simvg_sumsq(const float* x, long n, float* out_accum, float* partial_ws, hipStream_t stream) {
  SIMVG_CHECK_ARG(x && out_accum && partial_ws && n > 0 && n % 4 == 0,
                  "sumsq: n must be a positive multiple of 4; a 2048-float workspace is required");
  const long n4 = n / 4;
  // grid-stride over 3 resident blocks per CU (768 on MI355X): 84 us for 141 M floats against 106 us with 2048 blocks (the ViT-B arena is 223 M)
  const int grid = (int)((n4 + 255) / 256 < STREAM_GRID ? (n4 + 255) / 256 : STREAM_GRID);
  // read-once stream: non-temporal loads (5.1 -> 5.5 TB/s standalone; 3.7 TB/s before the 4-way unroll)
  hipLaunchKernelGGL(sumsq_kernel<true>, dim3(grid), dim3(256), 0, stream, x, n4, partial_ws);
  hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(64), 0, stream, partial_ws, grid, out_accum);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}

extern "C" int simvg_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* max_exp_avg_sq,
                               long n, float step_size, float bias_correction2_sqrt, float beta1, float beta2, float eps,
                               float weight_decay, const float* total_norm, float max_norm, hipStream_t stream) {
  SIMVG_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n > 0 && n % 4 == 0, "adam_step: n must be a positive multiple of 4");
  SIMVG_CHECK_ARG(bias_correction2_sqrt > 0.f, "adam_step: bias_correction2_sqrt must be > 0");
  AdamArgs a{param, grad, exp_avg, exp_avg_sq, max_exp_avg_sq, n / 4, step_size, bias_correction2_sqrt, beta1, beta2, eps,
             weight_decay, total_norm, max_norm, max_exp_avg_sq != nullptr};
  // 3 resident blocks per CU walking the nine streams in 3 MB strides: 880-980 us for 141 M elements against
  // 1150-1190 us with 4096 blocks (and every power-of-two grid); two chunks per thread in flight, loads phased around the
  // zero-row test (sweep: profiles/r02_sweeps.md)
  const long want = ((a.n4 + 255) / 256 + 1) / 2;
  const int grid = (int)(want < STREAM_GRID ? want : STREAM_GRID);
  hipLaunchKernelGGL((adam_kernel<2, true>), dim3(grid), dim3(256), 0, stream, a);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}
