// Embedding stage of the BEiT-3 encoder and weight preparation (gfx950).  All HBM-bound.
//
//  simvg_im2col      : fp32 NCHW image -> bf16 patch matrix [B*np, 3*P*P] (k = c*P*P + ky*P + kx), the
//                      A operand of the patch-embed GEMM (torchscale VisionEmbedding.proj, a Conv2d with
//                      kernel = stride = P; reference call site beit3_base.py:461)
//  simvg_embed_fwd   : assemble the fp32 residual stream, modality-major:
//                      vision rows [B, 1+np]: cls | patch + posA[t+2];  text rows [B, T]:
//                      (text_embed[id] + posB[j+2]) * (1 - pad)       (beit3_base.py:463-475,317-334,367)
//  simvg_embed_bwd   : the matching gradients (patch grad as bf16 for the wgrad GEMM, cls / position
//                      tables reduced over the batch, text table by atomics)
//  simvg_weight_prep : fp32 master weights -> bf16 compute copies, plain and transposed, batched over a
//                      descriptor table (one launch per optimizer step)
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void im2col_f32_kernel(const float* __restrict__ img, float* __restrict__ cols,
                                                         int B, int S, int P, long total4) {
  const int G = S / P, K = 3 * P * P;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total4; t += (long)gridDim.x * 256) {
    const long e = t * 4;
    const int x = (int)(e % S);
    const long r = e / S;
    const int y = (int)(r % S);
    const long bc = r / S;
    const int c = (int)(bc % 3), b = (int)(bc / 3);
    const int py = y / P, ky = y - py * P, px = x / P, kx = x - px * P;
    *(f32x4_t*)(cols + ((long)b * G * G + py * G + px) * K + c * P * P + ky * P + kx) = *(const f32x4_t*)(img + e);
  }
}

__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ img, lp_t* __restrict__ cols,
                                                     int B, int S, int P, long total4) {
  const int G = S / P, K = 3 * P * P;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total4; t += (long)gridDim.x * 256) {
    const long e = t * 4;
    const int x = (int)(e % S);
    const long r = e / S;
    const int y = (int)(r % S);
    const long bc = r / S;
    const int c = (int)(bc % 3), b = (int)(bc / 3);
    const f32x4_t v = *(const f32x4_t*)(img + e);
    const int py = y / P, ky = y - py * P, px = x / P, kx = x - px * P;
    lp_t* o = cols + ((long)b * G * G + py * G + px) * K + c * P * P + ky * P + kx;
    *(u32x2_t*)o = (u32x2_t){pack_lp2(v[0], v[1]), pack_lp2(v[2], v[3])};
  }
}

struct EmbedArgs {
  const float* patch; int ldp;     // [B*np, D] fp32 (conv output incl. bias)
  const float* cls;                // [D]
  const float* posA;               // [np+3, D]
  const float* posB;               // [1024, D]
  const float* text_embed;         // [V, D]
  const long long* ids;            // [B, T]
  const unsigned char* pad;        // [B, T] or null
  float* x; int ldx;               // [B*(np+1) + B*T, D]
  int B, np, T, D;
};

__global__ __launch_bounds__(256) void embed_fwd_kernel(EmbedArgs a) {
  const int Nv = a.np + 1;
  const long Mv = (long)a.B * Nv;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= Mv + (long)a.B * a.T) return;
  const int lane = threadIdx.x & 63;
  float* xr = a.x + row * a.ldx;
  if (row < Mv) {
    const int b = (int)(row / Nv), t = (int)(row - (long)b * Nv);
    const float* src = t == 0 ? a.cls : a.patch + ((long)b * a.np + (t - 1)) * a.ldp;
    const float* pos = a.posA + (long)(t + 2) * a.D;
    for (int c = lane * 4; c < a.D; c += 256) {
      const f32x4_t s = *(const f32x4_t*)(src + c), p = *(const f32x4_t*)(pos + c);
      *(f32x4_t*)(xr + c) = s + p;
    }
  } else {
    const long r = row - Mv;
    const int b = (int)(r / a.T), jn = (int)(r - (long)b * a.T);
    const float keep = (a.pad && a.pad[r]) ? 0.f : 1.f;
    const float* src = a.text_embed + (long)a.ids[r] * a.D;
    const float* pos = a.posB + (long)(jn + 2) * a.D;
    for (int c = lane * 4; c < a.D; c += 256) {
      const f32x4_t s = *(const f32x4_t*)(src + c), p = *(const f32x4_t*)(pos + c);
      *(f32x4_t*)(xr + c) = (s + p) * keep;
    }
  }
}

struct EmbedBwdArgs {
  const float* dx; int lddx;       // [M, D]
  lp_t* dpatch; int lddp;        // [B*np, D] bf16
  float* dcls;                     // [D]
  float* dposA;                    // [np+3, D]
  float* dposB;                    // [1024, D]
  float* dtext;                    // [V, D]
  const long long* ids;
  const unsigned char* pad;
  int B, np, T, D;
  float scale;                     // on the parameter gradients (1 / gradient scale of dx); dpatch keeps dx's scale
};

// one block per vision token position t (reduction over the batch), then one block per text position j
constexpr int EB = 8;       // samples per batch of embed_bwd_kernel (16: 95.6 us against 62.4)
__global__ __launch_bounds__(256) void embed_bwd_kernel(EmbedBwdArgs a) {
  const int Nv = a.np + 1;
  const long Mv = (long)a.B * Nv;
  const int pos = blockIdx.x;
  if (pos < Nv) {
    const int t = pos;
    for (int c = threadIdx.x * 4; c < a.D; c += 1024) {
      // samples in batches of EB: the batch's loads first, then the sums (in sample order, as before) and the 16-bit copies.  One
      // sample per trip kept every load behind the previous sample's store (they may alias): B memory round trips in a row.
      f32x4_t s = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      for (int b0 = 0; b0 < a.B; b0 += EB) {
        f32x4_t v[EB];
#pragma unroll
        for (int i = 0; i < EB; ++i) v[i] = *(const f32x4_t*)(a.dx + ((long)min(b0 + i, a.B - 1) * Nv + t) * a.lddx + c);
#pragma unroll
        for (int i = 0; i < EB; ++i) {
          if (b0 + i >= a.B) break;
          s += v[i];
          if (t > 0)
            *(u32x2_t*)(a.dpatch + ((long)(b0 + i) * a.np + (t - 1)) * a.lddp + c) =
                (u32x2_t){pack_lp2(v[i][0], v[i][1]), pack_lp2(v[i][2], v[i][3])};
        }
      }
      s *= a.scale;
      float* dp = a.dposA + (long)(t + 2) * a.D + c;
      *(f32x4_t*)dp = *(const f32x4_t*)dp + s;
      if (t == 0) *(f32x4_t*)(a.dcls + c) = *(const f32x4_t*)(a.dcls + c) + s;
    }
  } else {
    const int jn = pos - Nv;
    for (int c = threadIdx.x * 4; c < a.D; c += 1024) {
      f32x4_t s = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      for (int b0 = 0; b0 < a.B; b0 += EB) {
        f32x4_t v[EB];
        long id[EB];
        unsigned char pd[EB];
#pragma unroll
        for (int i = 0; i < EB; ++i) {
          const long r = (long)min(b0 + i, a.B - 1) * a.T + jn;
          pd[i] = a.pad ? a.pad[r] : (unsigned char)0;
          id[i] = a.ids[r];
          v[i] = *(const f32x4_t*)(a.dx + (Mv + r) * a.lddx + c);
        }
#pragma unroll
        for (int i = 0; i < EB; ++i) {
          if (b0 + i >= a.B) break;
          if (pd[i]) continue;
          const f32x4_t u = v[i] * a.scale;
          s += u;
          float* te = a.dtext + id[i] * a.D + c;
          atomicAdd(te + 0, u[0]); atomicAdd(te + 1, u[1]); atomicAdd(te + 2, u[2]); atomicAdd(te + 3, u[3]);
        }
      }
      float* dp = a.dposB + (long)(jn + 2) * a.D + c;
      *(f32x4_t*)dp = *(const f32x4_t*)dp + s;
    }
  }
}

struct WeightDesc {      // mirrored by simvg_amd/_lib.py (ctypes)
  const float* src;      // [rows, cols] fp32
  lp_t* dst;           // [rows, cols] bf16 or null
  lp_t* dst_t;         // [cols, rows] bf16 or null
  int rows, cols;
  int tile_start;        // first 64x64 tile index of this matrix in the launch
  int split_shift;       // > 0: dst rows are 2 * cols long, [lo * 2^shift | hi] (operand of simvg_gemm_nt_split; hi = the plain copy,
                         // lo = the 16-bit rounding of (w - hi) * 2^shift); 0: dst rows are cols long
};

// 64 x 64 tiles, 16-B loads, 8-B stores in both orientations (the 32 x 32 / 2-B-store version ran at 1.3 TB/s: 0.53 ms per
// step for the 340 MB of GEMM weights); tile_start counts 64 x 64 tiles (hip_ops.WeightPrep).
__global__ __launch_bounds__(256) void weight_prep_kernel(const WeightDesc* __restrict__ descs, int n) {
  __shared__ float tile[64][65];
  int lo = 0, hi = n - 1;
  const int bid = blockIdx.x;
  while (lo < hi) {   // last descriptor with tile_start <= bid
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].tile_start <= bid) lo = mid; else hi = mid - 1;
  }
  const WeightDesc d = descs[lo];
  const int tiles_c = (d.cols + 63) >> 6;
  const int tl = bid - d.tile_start;
  const int r0 = (tl / tiles_c) * 64, c0 = (tl % tiles_c) * 64;
  const int tq = threadIdx.x & 15, ty = threadIdx.x >> 4;      // 16 x 4-element groups across, 16 rows down
  const bool vec = (d.cols & 3) == 0 && (d.rows & 3) == 0;
  const bool two = d.split_shift > 0;
  const long ldd = two ? 2L * d.cols : d.cols;           // row length of dst
  const int hi_off = two ? d.cols : 0;                    // where the plain copy sits inside a dst row
  const float lo_mul = two ? (float)(1 << d.split_shift) : 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + 16 * k, c = c0 + 4 * tq;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (r < d.rows) {
      if (vec && c + 3 < d.cols) {
        const f32x4_t t = *(const f32x4_t*)(d.src + (long)r * d.cols + c);
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
        if (d.dst) {
          const u32x2_t h = (u32x2_t){pack_lp2(v[0], v[1]), pack_lp2(v[2], v[3])};
          *(u32x2_t*)(d.dst + (long)r * ldd + hi_off + c) = h;
          if (two) {
            float h0, h1, h2, h3;
            unpack_lp2(h[0], h0, h1);
            unpack_lp2(h[1], h2, h3);
            *(u32x2_t*)(d.dst + (long)r * ldd + c) =
                (u32x2_t){pack_lp2((v[0] - h0) * lo_mul, (v[1] - h1) * lo_mul), pack_lp2((v[2] - h2) * lo_mul, (v[3] - h3) * lo_mul)};
          }
        }
      } else {
        for (int e = 0; e < 4; ++e)
          if (c + e < d.cols) {
            v[e] = d.src[(long)r * d.cols + c + e];
            if (d.dst) {
              const lp_t h = f32_to_lp(v[e]);
              d.dst[(long)r * ldd + hi_off + c + e] = h;
              if (two) d.dst[(long)r * ldd + c + e] = f32_to_lp((v[e] - lp_to_f32(h)) * lo_mul);
            }
          }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[ty + 16 * k][4 * tq + e] = v[e];
  }
  if (!d.dst_t) return;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + 16 * k, r = r0 + 4 * tq;      // one 8-B store of 4 consecutive source rows of column c
    if (c >= d.cols) continue;
    const float t0 = tile[4 * tq][ty + 16 * k], t1 = tile[4 * tq + 1][ty + 16 * k], t2 = tile[4 * tq + 2][ty + 16 * k],
                t3 = tile[4 * tq + 3][ty + 16 * k];
    if (vec && r + 3 < d.rows) {
      *(u32x2_t*)(d.dst_t + (long)c * d.rows + r) = (u32x2_t){pack_lp2(t0, t1), pack_lp2(t2, t3)};
    } else {
      const float tt[4] = {t0, t1, t2, t3};
      for (int e = 0; e < 4; ++e)
        if (r + e < d.rows) d.dst_t[(long)c * d.rows + r + e] = f32_to_lp(tt[e]);
    }
  }
}

__global__ __launch_bounds__(256) void cast_lp_kernel(const float* __restrict__ src, lp_t* __restrict__ dst, long n4, float scale) {
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < n4; t += (long)gridDim.x * 256) {
    const f32x4_t v = *(const f32x4_t*)(src + t * 4) * scale;
    *(u32x2_t*)(dst + t * 4) = (u32x2_t){pack_lp2(v[0], v[1]), pack_lp2(v[2], v[3])};
  }
}

__global__ __launch_bounds__(256) void cast_f32_kernel(const lp_t* __restrict__ src, float* __restrict__ dst, long n4) {
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < n4; t += (long)gridDim.x * 256) {
    const u32x2_t v = *(const u32x2_t*)(src + t * 4);
    float f[4];
    unpack_lp2(v[0], f[0], f[1]);
    unpack_lp2(v[1], f[2], f[3]);
    *(f32x4_t*)(dst + t * 4) = (f32x4_t){f[0], f[1], f[2], f[3]};
  }
}

}  // namespace

extern "C" int simvg_im2col(const float* img, void* cols_bf16, int B, int S, int P, hipStream_t stream) {
  SIMVG_CHECK_ARG(B > 0 && S > 0 && P > 0 && S % P == 0 && P % 4 == 0, "im2col: S must be a multiple of P, P of 4");
  const long total4 = (long)B * 3 * S * S / 4;
  const int grid = (int)((total4 + 255) / 256 < 8192 ? (total4 + 255) / 256 : 8192);
  hipLaunchKernelGGL(im2col_kernel, dim3(grid), dim3(256), 0, stream, img, (lp_t*)cols_bf16, B, S, P, total4);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}

extern "C" int simvg_im2col_f32(const float* img, float* cols, int B, int S, int P, hipStream_t stream) {
  SIMVG_CHECK_ARG(B > 0 && S > 0 && P > 0 && S % P == 0 && P % 4 == 0, "im2col: S must be a multiple of P, P of 4");
  const long total4 = (long)B * 3 * S * S / 4;
  const int grid = (int)((total4 + 255) / 256 < 8192 ? (total4 + 255) / 256 : 8192);
  hipLaunchKernelGGL(im2col_f32_kernel, dim3(grid), dim3(256), 0, stream, img, cols, B, S, P, total4);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}

extern "C" int simvg_embed_fwd(const float* patch, int ldp, const float* cls, const float* posA, const float* posB,
                               const float* text_embed, const long long* ids, const unsigned char* pad, float* x,
                               int ldx, int B, int np, int T, int D, hipStream_t stream) {
  SIMVG_CHECK_ARG(B > 0 && np > 0 && T >= 0 && D % 4 == 0 && ldx % 4 == 0 && ldp % 4 == 0, "embed_fwd: bad geometry");
  SIMVG_CHECK_ARG(T + 2 <= 1024, "embed_fwd: text longer than the position table");
  EmbedArgs a{patch, ldp, cls, posA, posB, text_embed, ids, pad, x, ldx, B, np, T, D};
  const long rows = (long)B * (np + 1 + T);
  hipLaunchKernelGGL(embed_fwd_kernel, dim3((int)((rows + 3) / 4)), dim3(256), 0, stream, a);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}

extern "C" int simvg_embed_bwd(const float* dx, int lddx, void* dpatch_bf16, int lddp, float* dcls, float* dposA,
                               float* dposB, float* dtext, const long long* ids, const unsigned char* pad, int B,
                               int np, int T, int D, float param_scale, hipStream_t stream) {
  SIMVG_CHECK_ARG(B > 0 && np > 0 && T >= 0 && D % 4 == 0 && lddx % 4 == 0 && lddp % 4 == 0, "embed_bwd: bad geometry");
  EmbedBwdArgs a{dx, lddx, (lp_t*)dpatch_bf16, lddp, dcls, dposA, dposB, dtext, ids, pad, B, np, T, D, param_scale};
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(np + 1 + T), dim3(256), 0, stream, a);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}

extern "C" int simvg_weight_prep(const void* descs_dev, int n_desc, int total_tiles, hipStream_t stream) {
  SIMVG_CHECK_ARG(descs_dev && n_desc > 0 && total_tiles > 0, "weight_prep: empty descriptor table");
  hipLaunchKernelGGL(weight_prep_kernel, dim3(total_tiles), dim3(256), 0, stream, (const WeightDesc*)descs_dev, n_desc);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}

extern "C" int simvg_lowp_format(void) { return SIMVG_LOWP_FORMAT; }   // 1 = IEEE fp16, 2 = bfloat16

extern "C" int simvg_cast_f32_to_lp(const float* src, void* dst, long n, float scale, hipStream_t stream) {
  SIMVG_CHECK_ARG(n > 0 && n % 4 == 0, "cast: n must be a positive multiple of 4");
  const long n4 = n / 4;
  const int grid = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  hipLaunchKernelGGL(cast_lp_kernel, dim3(grid), dim3(256), 0, stream, src, (lp_t*)dst, n4, scale);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}

extern "C" int simvg_cast_lp_to_f32(const void* src, float* dst, long n, hipStream_t stream) {
  SIMVG_CHECK_ARG(n > 0 && n % 4 == 0, "cast: n must be a positive multiple of 4");
  const long n4 = n / 4;
  const int grid = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  hipLaunchKernelGGL(cast_f32_kernel, dim3(grid), dim3(256), 0, stream, (const lp_t*)src, dst, n4);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}
