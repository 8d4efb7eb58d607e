// Device-side image pre-processing (gfx950): the pixel work of the reference's CPU pipeline
// (simvg/datasets/pipelines/transforms.py: LargeScaleJitter :221-342, Resize :59-80, Normalize :146-158, Pad :193-205,
// formatting.py DefaultFormatBundle :62-70) on uint8 images that already sit in HBM.  HBM-bound streaming kernels, no MFMA.
//
//  * resize_u8: OpenCV INTER_LINEAR semantics for 8-bit images, restated (mmcv.imresize / imrescale call cv2.resize):
//    half-pixel centres, fx = float((dx + 0.5) * scale - 0.5), border taps clamped, 11-bit fixed-point coefficients
//    rounded half-to-even, int32 horizontal pass, vertical pass ((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
//    exact 2x2 decimation = area average.  A WINDOW of the virtual resized image can be produced directly, which fuses
//    LargeScaleJitter's "rescale, then crop" into one pass that never materialises the rescaled image.
//  * normalize_pad_u8: BGR->RGB swap, (x - mean) * (1 / std) in fp32, zero padding to the padded canvas and the
//    HWC -> CHW transpose of DefaultFormatBundle, one pass: 3 B read, 12 B written per pixel.
#include "common.h"

// mirrors of include/simvg_hip.h (the header is C, its stream type is opaque)
#define SIMVG_PREPROCESS_MAX_JOBS 32
struct simvg_resize_job {
  const void* src; int src_h, src_w; long src_row_bytes;
  void* dst; long dst_row_bytes;
  int out_h, out_w, full_h, full_w, win_y0, win_x0;
};
struct simvg_format_job {
  const void* src; long src_row_bytes; int h, w;
  float* dst_chw; int pad_h, pad_w;
};

namespace {

struct Tap { int s0, s1, a0, a1; };

__device__ __forceinline__ Tap make_tap(int d, int dst_n, int src_n) {
  const double scale = (double)src_n / (double)dst_n;
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) { f = 0.f; s = 0; }
  if (s >= src_n - 1) { f = 0.f; s = src_n - 1; }
  Tap t;
  t.s0 = s;
  t.s1 = min(s + 1, src_n - 1);
  t.a1 = (int)rintf(f * 2048.f);
  t.a0 = (int)rintf((1.f - f) * 2048.f);
  return t;
}

__global__ __launch_bounds__(256) void resize_u8_kernel(const unsigned char* __restrict__ src, int sh, int sw, long src_ld,
                                                        unsigned char* __restrict__ dst, long dst_ld, int out_h, int out_w,
                                                        int full_h, int full_w, int win_y0, int win_x0, int area2) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= out_w || y >= out_h) return;
  const int dx = win_x0 + x, dy = win_y0 + y;
  unsigned char* o = dst + (long)y * dst_ld + 3 * x;
  if (area2) {   // INTER_LINEAR with an exact 2x2 decimation is routed to INTER_AREA by OpenCV
    const unsigned char* p0 = src + (long)(2 * dy) * src_ld + 6 * dx;
    const unsigned char* p1 = p0 + src_ld;
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = (unsigned char)((p0[c] + p0[3 + c] + p1[c] + p1[3 + c] + 2) >> 2);
    return;
  }
  const Tap tx = make_tap(dx, full_w, sw), ty = make_tap(dy, full_h, sh);
  const unsigned char* r0 = src + (long)ty.s0 * src_ld;
  const unsigned char* r1 = src + (long)ty.s1 * src_ld;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int h0 = r0[3 * tx.s0 + c] * tx.a0 + r0[3 * tx.s1 + c] * tx.a1;
    const int h1 = r1[3 * tx.s0 + c] * tx.a0 + r1[3 * tx.s1 + c] * tx.a1;
    int v = (((ty.a0 * (h0 >> 4)) >> 16) + ((ty.a1 * (h1 >> 4)) >> 16) + 2) >> 2;
    o[c] = (unsigned char)min(max(v, 0), 255);
  }
}

__global__ __launch_bounds__(256) void normalize_pad_kernel(const unsigned char* __restrict__ src, long src_ld, int h, int w,
                                                            float* __restrict__ dst, int Hp, int Wp, float m0, float m1,
                                                            float m2, float i0, float i1, float i2, int to_rgb) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= Wp || y >= Hp) return;
  float v0 = 0.f, v1 = 0.f, v2 = 0.f;
  if (x < w && y < h) {
    const unsigned char* p = src + (long)y * src_ld + 3 * x;
    const float c0 = (float)p[to_rgb ? 2 : 0], c1 = (float)p[1], c2 = (float)p[to_rgb ? 0 : 2];
    v0 = (c0 - m0) * i0; v1 = (c1 - m1) * i1; v2 = (c2 - m2) * i2;
  }
  const long plane = (long)Hp * Wp, o = (long)y * Wp + x;
  dst[o] = v0; dst[plane + o] = v1; dst[2 * plane + o] = v2;
}

// ---- batched forms: one launch for up to SIMVG_PREPROCESS_MAX_JOBS frames of DIFFERENT geometry (blockIdx.z = job; the
// grid covers the largest output, smaller jobs exit early).  A loader that runs two resizes and one format pass per frame
// issues 3 launches per 32 frames instead of 96: at 2000 frames/s the per-launch host cost, not the kernels, was the limit.
struct ResizeJobs { int count; simvg_resize_job job[SIMVG_PREPROCESS_MAX_JOBS]; };
struct FormatJobs { int count; float m[3], inv[3]; int to_rgb; simvg_format_job job[SIMVG_PREPROCESS_MAX_JOBS]; };

__global__ __launch_bounds__(256) void resize_u8_batched_kernel(ResizeJobs b) {
  const simvg_resize_job& j = b.job[blockIdx.z];
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= j.out_w || y >= j.out_h) return;
  const unsigned char* src = (const unsigned char*)j.src;
  const int dx = j.win_x0 + x, dy = j.win_y0 + y;
  unsigned char* o = (unsigned char*)j.dst + (long)y * j.dst_row_bytes + 3 * x;
  if (j.src_w == 2 * j.full_w && j.src_h == 2 * j.full_h) {
    const unsigned char* p0 = src + (long)(2 * dy) * j.src_row_bytes + 6 * dx;
    const unsigned char* p1 = p0 + j.src_row_bytes;
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = (unsigned char)((p0[c] + p0[3 + c] + p1[c] + p1[3 + c] + 2) >> 2);
    return;
  }
  const Tap tx = make_tap(dx, j.full_w, j.src_w), ty = make_tap(dy, j.full_h, j.src_h);
  const unsigned char* r0 = src + (long)ty.s0 * j.src_row_bytes;
  const unsigned char* r1 = src + (long)ty.s1 * j.src_row_bytes;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int h0 = r0[3 * tx.s0 + c] * tx.a0 + r0[3 * tx.s1 + c] * tx.a1;
    const int h1 = r1[3 * tx.s0 + c] * tx.a0 + r1[3 * tx.s1 + c] * tx.a1;
    int v = (((ty.a0 * (h0 >> 4)) >> 16) + ((ty.a1 * (h1 >> 4)) >> 16) + 2) >> 2;
    o[c] = (unsigned char)min(max(v, 0), 255);
  }
}

__global__ __launch_bounds__(256) void normalize_pad_batched_kernel(FormatJobs b) {
  const simvg_format_job& j = b.job[blockIdx.z];
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= j.pad_w || y >= j.pad_h) return;
  float v0 = 0.f, v1 = 0.f, v2 = 0.f;
  if (x < j.w && y < j.h) {
    const unsigned char* p = (const unsigned char*)j.src + (long)y * j.src_row_bytes + 3 * x;
    const float c0 = (float)p[b.to_rgb ? 2 : 0], c1 = (float)p[1], c2 = (float)p[b.to_rgb ? 0 : 2];
    v0 = (c0 - b.m[0]) * b.inv[0]; v1 = (c1 - b.m[1]) * b.inv[1]; v2 = (c2 - b.m[2]) * b.inv[2];
  }
  float* dst = j.dst_chw;
  const long plane = (long)j.pad_h * j.pad_w, o = (long)y * j.pad_w + x;
  dst[o] = v0; dst[plane + o] = v1; dst[2 * plane + o] = v2;
}

}  // namespace

extern "C" int simvg_resize_u8_batched(const simvg_resize_job* jobs, int count, hipStream_t stream) {
  SIMVG_CHECK_ARG(jobs && count > 0 && count <= SIMVG_PREPROCESS_MAX_JOBS, "resize_u8_batched: 1..32 jobs per call");
  ResizeJobs b;
  b.count = count;
  int mw = 0, mh = 0;
  for (int i = 0; i < count; ++i) {
    const simvg_resize_job& j = jobs[i];
    SIMVG_CHECK_ARG(j.src && j.dst && j.src_h > 0 && j.src_w > 0 && j.out_h > 0 && j.out_w > 0 && j.full_h > 0 && j.full_w > 0,
                    "resize_u8_batched: empty image");
    SIMVG_CHECK_ARG(j.win_y0 >= 0 && j.win_x0 >= 0 && j.win_y0 + j.out_h <= j.full_h && j.win_x0 + j.out_w <= j.full_w,
                    "resize_u8_batched: the window must lie inside the resized image");
    SIMVG_CHECK_ARG(j.src_row_bytes >= 3L * j.src_w && j.dst_row_bytes >= 3L * j.out_w, "resize_u8_batched: row pitch smaller than a row");
    b.job[i] = j;
    mw = j.out_w > mw ? j.out_w : mw;
    mh = j.out_h > mh ? j.out_h : mh;
  }
  hipLaunchKernelGGL(resize_u8_batched_kernel, dim3(cdiv(mw, 64), cdiv(mh, 4), count), dim3(256), 0, stream, b);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}

extern "C" int simvg_normalize_pad_u8_batched(const simvg_format_job* jobs, int count, const float* mean3_host,
                                              const float* std3_host, int to_rgb, hipStream_t stream) {
  SIMVG_CHECK_ARG(jobs && count > 0 && count <= SIMVG_PREPROCESS_MAX_JOBS, "normalize_pad_batched: 1..32 jobs per call");
  SIMVG_CHECK_ARG(mean3_host && std3_host, "normalize_pad_batched: mean / std");
  FormatJobs b;
  b.count = count;
  b.to_rgb = to_rgb;
  for (int c = 0; c < 3; ++c) {
    b.m[c] = mean3_host[c];
    b.inv[c] = (float)(1.0 / (double)std3_host[c]);
  }
  int mw = 0, mh = 0;
  for (int i = 0; i < count; ++i) {
    const simvg_format_job& j = jobs[i];
    SIMVG_CHECK_ARG(j.src && j.dst_chw && j.h > 0 && j.w > 0 && j.pad_h >= j.h && j.pad_w >= j.w && j.src_row_bytes >= 3L * j.w,
                    "normalize_pad_batched: bad geometry");
    b.job[i] = j;
    mw = j.pad_w > mw ? j.pad_w : mw;
    mh = j.pad_h > mh ? j.pad_h : mh;
  }
  hipLaunchKernelGGL(normalize_pad_batched_kernel, dim3(cdiv(mw, 64), cdiv(mh, 4), count), dim3(256), 0, stream, b);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}

extern "C" int simvg_resize_u8(const void* src, int src_h, int src_w, long src_row_bytes, void* dst, long dst_row_bytes,
                               int out_h, int out_w, int full_h, int full_w, int win_y0, int win_x0, hipStream_t stream) {
  SIMVG_CHECK_ARG(src && dst && src_h > 0 && src_w > 0 && out_h > 0 && out_w > 0 && full_h > 0 && full_w > 0,
                  "resize_u8: empty image");
  SIMVG_CHECK_ARG(win_y0 >= 0 && win_x0 >= 0 && win_y0 + out_h <= full_h && win_x0 + out_w <= full_w,
                  "resize_u8: the window must lie inside the resized image");
  SIMVG_CHECK_ARG(src_row_bytes >= 3L * src_w && dst_row_bytes >= 3L * out_w, "resize_u8: row pitch smaller than a row");
  const int area2 = (src_w == 2 * full_w && src_h == 2 * full_h) ? 1 : 0;
  hipLaunchKernelGGL(resize_u8_kernel, dim3(cdiv(out_w, 64), cdiv(out_h, 4)), dim3(256), 0, stream,
                     (const unsigned char*)src, src_h, src_w, src_row_bytes, (unsigned char*)dst, dst_row_bytes, out_h, out_w,
                     full_h, full_w, win_y0, win_x0, area2);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}

extern "C" int simvg_normalize_pad_u8(const void* src_hwc, long src_row_bytes, int h, int w, float* dst_chw, int pad_h,
                                      int pad_w, const float* mean3_host, const float* std3_host, int to_rgb,
                                      hipStream_t stream) {
  SIMVG_CHECK_ARG(src_hwc && dst_chw && h > 0 && w > 0 && pad_h >= h && pad_w >= w, "normalize_pad: bad geometry");
  SIMVG_CHECK_ARG(mean3_host && std3_host && src_row_bytes >= 3L * w, "normalize_pad: mean / std / pitch");
  float m[3], inv[3];
  for (int c = 0; c < 3; ++c) {
    m[c] = mean3_host[c];
    inv[c] = (float)(1.0 / (double)std3_host[c]);      // mmcv.imnormalize: stdinv = 1 / float64(std)
  }
  hipLaunchKernelGGL(normalize_pad_kernel, dim3(cdiv(pad_w, 64), cdiv(pad_h, 4)), dim3(256), 0, stream,
                     (const unsigned char*)src_hwc, src_row_bytes, h, w, dst_chw, pad_h, pad_w, m[0], m[1], m[2], inv[0],
                     inv[1], inv[2], to_rgb);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}
