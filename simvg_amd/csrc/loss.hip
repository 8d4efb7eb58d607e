// Hungarian matcher + SetCriterion on the GPU (gfx950) -- no host round trip, no .item().
//
//  simvg_match        : detrex HungarianMatcher.forward with ce_cost (cost = 5*L1 - softmax(logit)[label]
//                       - 2*GIoU) followed by the rectangular LSAP that the reference solves with SciPy on
//                       the host (reference call sites core/criterion/criterion.py:239,259 and
//                       heads/tgqs_kd_detr_head/tgqs_kd_detr_head.py:252); one thread per (layer, image),
//                       num_queries <= 16, targets <= 16.
//  simvg_soft_targets : prepare_soft_targets, mode score_iou_weighted (tgqs_kd_detr_head.py:248-264):
//                       matched decoder boxes become the distillation targets, weight = score * IoU;
//                       also emits weights_distill = mean(weight) (:491) and the target counts.
//  simvg_criterion    : SetCriterion.forward (criterion.py:226-271) for all decoder layers at once:
//                       weighted 2-class CE (eos_coef), L1 and (1 - GIoU) on matched pairs / num_boxes,
//                       times weight_dict (tgqs_kd_detr_head.py:340-350), times the branch coefficient
//                       (:486,501,506) -- forward value AND the gradients w.r.t. logits / boxes.
#include "common.h"

namespace {

constexpr int MAXQ = 16, MAXT = 16;
constexpr float BOX_EPS = 1e-6f;   // detrex box_iou / generalized_box_iou epsilons

struct Box { float x1, y1, x2, y2; };
__device__ __forceinline__ Box to_xyxy(const float* c) {
  return Box{c[0] - 0.5f * c[2], c[1] - 0.5f * c[3], c[0] + 0.5f * c[2], c[1] + 0.5f * c[3]};
}
__device__ __forceinline__ float box_iou_only(const Box& a, const Box& b) {
  const float a1 = (a.x2 - a.x1) * (a.y2 - a.y1), a2 = (b.x2 - b.x1) * (b.y2 - b.y1);
  const float iw = fmaxf(fminf(a.x2, b.x2) - fmaxf(a.x1, b.x1), 0.f);
  const float ih = fmaxf(fminf(a.y2, b.y2) - fmaxf(a.y1, b.y1), 0.f);
  const float inter = iw * ih;
  return inter / (a1 + a2 - inter + BOX_EPS);
}
// d(max(a,b))/da with torch's tie rule (half each)
__device__ __forceinline__ float gmax(float a, float b) { return a > b ? 1.f : (a == b ? 0.5f : 0.f); }
__device__ __forceinline__ float gmin(float a, float b) { return a < b ? 1.f : (a == b ? 0.5f : 0.f); }

// GIoU of src (cxcywh) vs tgt (cxcywh) and, if g != nullptr, d giou / d src(cx,cy,w,h)
__device__ float giou_cxcywh(const float* s, const float* t, float* g) {
  const Box a = to_xyxy(s), b = to_xyxy(t);
  const float a1 = (a.x2 - a.x1) * (a.y2 - a.y1), a2 = (b.x2 - b.x1) * (b.y2 - b.y1);
  const float riw = fminf(a.x2, b.x2) - fmaxf(a.x1, b.x1), rih = fminf(a.y2, b.y2) - fmaxf(a.y1, b.y1);
  const float iw = fmaxf(riw, 0.f), ih = fmaxf(rih, 0.f);
  const float inter = iw * ih;
  const float uni = a1 + a2 - inter;
  const float iou = inter / (uni + BOX_EPS);
  const float rcw = fmaxf(a.x2, b.x2) - fminf(a.x1, b.x1), rch = fmaxf(a.y2, b.y2) - fminf(a.y1, b.y1);
  const float cw = fmaxf(rcw, 0.f), ch = fmaxf(rch, 0.f);
  const float hull = cw * ch;
  const float giou = iou - (hull - uni) / (hull + BOX_EPS);
  if (g) {
    const float d_hull = -(uni + BOX_EPS) / ((hull + BOX_EPS) * (hull + BOX_EPS));
    const float d_uni = -inter / ((uni + BOX_EPS) * (uni + BOX_EPS)) + 1.f / (hull + BOX_EPS);
    const float d_inter = 1.f / (uni + BOX_EPS) - d_uni;
    const float d_a1 = d_uni;
    const float d_iw = d_inter * ih * (riw >= 0.f ? 1.f : 0.f), d_ih = d_inter * iw * (rih >= 0.f ? 1.f : 0.f);
    const float d_cw = d_hull * ch * (rcw >= 0.f ? 1.f : 0.f), d_ch = d_hull * cw * (rch >= 0.f ? 1.f : 0.f);
    float dx1 = -d_iw * gmax(a.x1, b.x1) - d_cw * gmin(a.x1, b.x1) - d_a1 * (a.y2 - a.y1);
    float dx2 = d_iw * gmin(a.x2, b.x2) + d_cw * gmax(a.x2, b.x2) + d_a1 * (a.y2 - a.y1);
    float dy1 = -d_ih * gmax(a.y1, b.y1) - d_ch * gmin(a.y1, b.y1) - d_a1 * (a.x2 - a.x1);
    float dy2 = d_ih * gmin(a.y2, b.y2) + d_ch * gmax(a.y2, b.y2) + d_a1 * (a.x2 - a.x1);
    g[0] = dx1 + dx2; g[1] = dy1 + dy2; g[2] = 0.5f * (dx2 - dx1); g[3] = 0.5f * (dy2 - dy1);
  }
  return giou;
}

struct MatchArgs {
  const float* logits;   // [L, B, nq, 2]
  const float* boxes;    // [L, B, nq, 4]
  const float* tboxes;   // [B, TM, 4]
  const int* tlabels;    // [B, TM]
  const int* tcount;     // [B]
  int* match;            // [L, B, nq] -> target index or -1
  int L, B, nq, TM;
  float c_class, c_bbox, c_giou;
};

__global__ void match_kernel(MatchArgs a) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= a.L * a.B) return;
  const int b = id % a.B;
  const int nq = a.nq, k = a.tcount[b];
  const float* lg = a.logits + (long)id * nq * 2;
  const float* bx = a.boxes + (long)id * nq * 4;
  int* out = a.match + (long)id * nq;
  for (int q = 0; q < nq; ++q) out[q] = -1;
  if (k <= 0) return;
  float cost[MAXQ][MAXT];
  for (int q = 0; q < nq; ++q) {
    const float m = fmaxf(lg[2 * q], lg[2 * q + 1]);
    const float e0 = __expf(lg[2 * q] - m), e1 = __expf(lg[2 * q + 1] - m);
    const float pr[2] = {e0 / (e0 + e1), e1 / (e0 + e1)};
    for (int t = 0; t < k; ++t) {
      const float* tb = a.tboxes + ((long)b * a.TM + t) * 4;
      float l1 = 0.f;
      for (int c = 0; c < 4; ++c) l1 += fabsf(bx[4 * q + c] - tb[c]);
      cost[q][t] = a.c_bbox * l1 - a.c_class * pr[a.tlabels[b * a.TM + t]] - a.c_giou * giou_cxcywh(bx + 4 * q, tb, nullptr);
    }
  }
  // Hungarian (Kuhn-Munkres with potentials), rows = the smaller side
  const bool tr = nq > k;
  const int n = tr ? k : nq, m = tr ? nq : k;
  float u[MAXQ + 1], v[MAXQ + 1], minv[MAXQ + 1];
  int p[MAXQ + 1], way[MAXQ + 1];
  bool used[MAXQ + 1];
  for (int j = 0; j <= m; ++j) { v[j] = 0.f; p[j] = 0; way[j] = 0; }
  for (int i = 0; i <= n; ++i) u[i] = 0.f;
  for (int i = 1; i <= n; ++i) {
    p[0] = i;
    int j0 = 0;
    for (int j = 0; j <= m; ++j) { minv[j] = INFINITY; used[j] = false; }
    do {
      used[j0] = true;
      const int i0 = p[j0];
      float delta = INFINITY;
      int j1 = 0;
      for (int j = 1; j <= m; ++j) {
        if (used[j]) continue;
        const float cij = tr ? cost[j - 1][i0 - 1] : cost[i0 - 1][j - 1];
        const float cur = cij - u[i0] - v[j];
        if (cur < minv[j]) { minv[j] = cur; way[j] = j0; }
        if (minv[j] < delta) { delta = minv[j]; j1 = j; }
      }
      for (int j = 0; j <= m; ++j) {
        if (used[j]) { u[p[j]] += delta; v[j] -= delta; }
        else minv[j] -= delta;
      }
      j0 = j1;
    } while (p[j0] != 0);
    do {
      const int j1 = way[j0];
      p[j0] = p[j1];
      j0 = j1;
    } while (j0);
  }
  for (int j = 1; j <= m; ++j) {
    if (p[j] == 0) continue;
    if (tr) out[j - 1] = p[j] - 1;        // column = query, row = target
    else out[p[j] - 1] = j - 1;           // row = query, column = target
  }
}

struct SoftArgs {
  const float* logits;   // [B, nq, 2]   decoder final layer (detached)
  const float* boxes;    // [B, nq, 4]
  const int* match;      // [B, nq]
  const float* tboxes;   // [B, TM, 4]  GT
  const int* tcount;     // [B]
  float* pboxes;         // [B, TM, 4]  out: distillation targets
  int* plabels;          // [B, TM]
  int* pcount;           // [B]
  float* pweight;        // [B, TM]
  float* scal;           // [4] out: weights_distill, n_gt, n_pred, sum_w
  int B, nq, TM;
};

__global__ void soft_targets_kernel(SoftArgs a) {
  __shared__ float s_sum[256];
  __shared__ int s_cnt[256], s_gt[256];
  float sw = 0.f;
  int cnt = 0, ngt = 0;
  for (int b = threadIdx.x; b < a.B; b += blockDim.x) {
    int c = 0;
    ngt += a.tcount[b];
    for (int q = 0; q < a.nq; ++q) {
      const int t = a.match[b * a.nq + q];
      if (t < 0) continue;
      const float* pb = a.boxes + ((long)b * a.nq + q) * 4;
      const float* lg = a.logits + ((long)b * a.nq + q) * 2;
      const float m = fmaxf(lg[0], lg[1]);
      const float e0 = __expf(lg[0] - m), e1 = __expf(lg[1] - m);
      const float score = e0 / (e0 + e1);
      const float iou = box_iou_only(to_xyxy(pb), to_xyxy(a.tboxes + ((long)b * a.TM + t) * 4));
      const float w = score * iou;
      float* o = a.pboxes + ((long)b * a.TM + c) * 4;
      o[0] = pb[0]; o[1] = pb[1]; o[2] = pb[2]; o[3] = pb[3];
      a.plabels[b * a.TM + c] = 0;
      a.pweight[b * a.TM + c] = w;
      sw += w;
      ++c;
    }
    a.pcount[b] = c;
    cnt += c;
  }
  s_sum[threadIdx.x] = sw; s_cnt[threadIdx.x] = cnt; s_gt[threadIdx.x] = ngt;
  __syncthreads();
  if (threadIdx.x == 0) {
    float S = 0.f; int C = 0, G = 0;
    for (int i = 0; i < blockDim.x; ++i) { S += s_sum[i]; C += s_cnt[i]; G += s_gt[i]; }
    a.scal[0] = S / (float)C;      // mean of an empty cat == nan, like torch (quirk Q8)
    a.scal[1] = (float)G;
    a.scal[2] = (float)C;
    a.scal[3] = S;
  }
}

struct CritArgs {
  const float* logits;    // [L, B, nq, 2]
  const float* boxes;     // [L, B, nq, 4]
  const int* match;       // [L, B, nq]
  const float* tboxes;    // [B, TM, 4]
  const int* tlabels;     // [B, TM]
  const float* num_boxes; // device scalar: sum of target counts averaged over ranks (clamped >= 1 here)
  const float* wdist;     // device scalar weights_distill or null
  float* dlogits;         // [L, B, nq, 2]
  float* dboxes;          // [L, B, nq, 4]
  float* out;             // [1 + 3L]: total (incl. coefficient), then per layer class/bbox/giou (weighted by weight_dict)
  int L, B, nq, TM, coef_mode;
  float coef, eos_coef, w_class, w_bbox, w_giou;
};

__global__ __launch_bounds__(256) void criterion_kernel(CritArgs a) {
  __shared__ float red[4][256];
  __shared__ float lay[64][4];   // per layer: Wsum, CEsum, L1sum, GIoUsum
  const int tid = threadIdx.x;
  const int per = a.B * a.nq;
  const float nb = fmaxf(*a.num_boxes, 1.f);
  float coef = a.coef;
  if (a.coef_mode == 1) coef *= (1.f - *a.wdist);
  else if (a.coef_mode == 2) coef *= *a.wdist;
  for (int l = 0; l < a.L; ++l) {
    float ws = 0.f, ce = 0.f, l1 = 0.f, gi = 0.f;
    for (int i = tid; i < per; i += 256) {
      const long idx = (long)l * per + i;
      const int b = i / a.nq;
      const int t = a.match[idx];
      const int cls = t >= 0 ? a.tlabels[b * a.TM + t] : 1;
      const float w = cls == 1 ? a.eos_coef : 1.f;
      const float* lg = a.logits + idx * 2;
      const float m = fmaxf(lg[0], lg[1]);
      const float lse = m + __logf(__expf(lg[0] - m) + __expf(lg[1] - m));
      ws += w;
      ce += w * (lse - lg[cls]);
      if (t >= 0) {
        const float* sb = a.boxes + idx * 4;
        const float* tb = a.tboxes + ((long)b * a.TM + t) * 4;
        for (int c = 0; c < 4; ++c) l1 += fabsf(sb[c] - tb[c]);
        gi += 1.f - giou_cxcywh(sb, tb, nullptr);
      }
    }
    red[0][tid] = ws; red[1][tid] = ce; red[2][tid] = l1; red[3][tid] = gi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (tid < s) for (int c = 0; c < 4; ++c) red[c][tid] += red[c][tid + s];
      __syncthreads();
    }
    if (tid == 0) for (int c = 0; c < 4; ++c) lay[l][c] = red[c][0];
    __syncthreads();
  }
  if (tid == 0) {
    float total = 0.f;
    for (int l = 0; l < a.L; ++l) {
      const float lc = a.w_class * lay[l][1] / lay[l][0];
      const float lb = a.w_bbox * lay[l][2] / nb;
      const float lg = a.w_giou * lay[l][3] / nb;
      a.out[1 + 3 * l] = lc; a.out[2 + 3 * l] = lb; a.out[3 + 3 * l] = lg;
      total += lc + lb + lg;
    }
    a.out[0] = coef * total;
  }
  // gradients
  for (int l = 0; l < a.L; ++l) {
    const float inv_ws = 1.f / lay[l][0];
    for (int i = tid; i < per; i += 256) {
      const long idx = (long)l * per + i;
      const int b = i / a.nq;
      const int t = a.match[idx];
      const int cls = t >= 0 ? a.tlabels[b * a.TM + t] : 1;
      const float w = cls == 1 ? a.eos_coef : 1.f;
      const float* lg = a.logits + idx * 2;
      const float m = fmaxf(lg[0], lg[1]);
      const float e0 = __expf(lg[0] - m), e1 = __expf(lg[1] - m);
      const float p0 = e0 / (e0 + e1), p1 = e1 / (e0 + e1);
      const float k = coef * a.w_class * w * inv_ws;
      a.dlogits[idx * 2 + 0] = k * (p0 - (cls == 0 ? 1.f : 0.f));
      a.dlogits[idx * 2 + 1] = k * (p1 - (cls == 1 ? 1.f : 0.f));
      float g[4] = {0.f, 0.f, 0.f, 0.f};
      if (t >= 0) {
        const float* sb = a.boxes + idx * 4;
        const float* tb = a.tboxes + ((long)b * a.TM + t) * 4;
        float gg[4];
        giou_cxcywh(sb, tb, gg);
        for (int c = 0; c < 4; ++c) {
          const float d = sb[c] - tb[c];
          const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
          g[c] = coef * (a.w_bbox * sgn - a.w_giou * gg[c]) / nb;
        }
      }
      for (int c = 0; c < 4; ++c) a.dboxes[idx * 4 + c] = g[c];
    }
  }
}


// head.inference + detector_postprocess + get_predictions in one launch (tgqs_kd_detr_head.py:577-604, mix_detr_mb.py:127-159
// and detectron2 Boxes.scale / clip / nonempty): per query softmax over the class columns, score / label =
// max over all but the last (no-object) column, cxcywh -> xyxy * (w, h, w, h), clip to the image, keep = non-empty box;
// per image the kept query with the highest score (first on ties; query 0 when none is kept) and its box, divided by the
// scale factor when rescaling.  One 64-lane block per image, lane = query (num_queries <= 16).
__global__ __launch_bounds__(64) void postprocess_kernel(const float* __restrict__ logits, const float* __restrict__ boxes,
                                                          const float* __restrict__ wh, const float* __restrict__ sf,
                                                          float* __restrict__ scores, long long* __restrict__ labels,
                                                          float* __restrict__ xyxy, unsigned char* __restrict__ keep,
                                                          float* __restrict__ best_box, long long* __restrict__ best_label,
                                                          int nq, int ncol) {
  const int b = blockIdx.x, q = threadIdx.x;
  float sc = -2.f;
  int lab = 0;
  float bx[4] = {0.f, 0.f, 0.f, 0.f};
  bool kp = false;
  if (q < nq) {
    const float* lg = logits + ((long)b * nq + q) * ncol;
    float mx = lg[0];
    for (int c = 1; c < ncol; ++c) mx = fmaxf(mx, lg[c]);
    float den = 0.f;
    for (int c = 0; c < ncol; ++c) den += expf(lg[c] - mx);
    float best = -1.f;
    for (int c = 0; c < ncol - 1; ++c) {
      const float p = expf(lg[c] - mx) / den;
      if (p > best) { best = p; lab = c; }
    }
    const float* bp = boxes + ((long)b * nq + q) * 4;
    const float* w4 = wh + (long)b * 4;
    const float cx = bp[0], cy = bp[1], w = bp[2], h = bp[3];
    bx[0] = (cx - 0.5f * w) * w4[0];
    bx[1] = (cy - 0.5f * h) * w4[1];
    bx[2] = (cx + 0.5f * w) * w4[2];
    bx[3] = (cy + 0.5f * h) * w4[3];
#pragma unroll
    for (int k = 0; k < 4; ++k) bx[k] = fminf(fmaxf(bx[k], 0.f), w4[k]);
    kp = (bx[2] - bx[0]) > 0.f && (bx[3] - bx[1]) > 0.f;
    if (sf) {
#pragma unroll
      for (int k = 0; k < 4; ++k) bx[k] = bx[k] / sf[(long)b * 4 + k];
    }
    const long o = (long)b * nq + q;
    scores[o] = best;
    labels[o] = lab;
    keep[o] = kp ? 1 : 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) xyxy[o * 4 + k] = bx[k];
    sc = kp ? best : -1.f;          // torch.where(keep, scores, -1).argmax(1)
  }
  // first maximum over the queries
  float v = sc;
  int idx = q < nq ? q : 1 << 20;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float v2 = __shfl_xor(v, o, 64);
    const int i2 = __shfl_xor(idx, o, 64);
    if (v2 > v || (v2 == v && i2 < idx)) { v = v2; idx = i2; }
  }
  if (q == idx) {
#pragma unroll
    for (int k = 0; k < 4; ++k) best_box[(long)b * 4 + k] = bx[k];
    best_label[b] = lab;
  }
}

}  // namespace

extern "C" int simvg_match(const float* logits, const float* boxes, const float* tboxes, const int* tlabels,
                           const int* tcount, int* match, int L, int B, int nq, int TM, float cost_class,
                           float cost_bbox, float cost_giou, hipStream_t stream) {
  SIMVG_CHECK_ARG(L > 0 && B > 0 && nq > 0 && nq <= MAXQ && TM > 0 && TM <= MAXT, "match: nq and targets must be <= 16");
  MatchArgs a{logits, boxes, tboxes, tlabels, tcount, match, L, B, nq, TM, cost_class, cost_bbox, cost_giou};
  hipLaunchKernelGGL(match_kernel, dim3(cdiv(L * B, 64)), dim3(64), 0, stream, a);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}

extern "C" int simvg_soft_targets(const float* logits, const float* boxes, const int* match, const float* tboxes,
                                  const int* tcount, float* pboxes, int* plabels, int* pcount, float* pweight,
                                  float* scalars4, int B, int nq, int TM, hipStream_t stream) {
  SIMVG_CHECK_ARG(B > 0 && nq > 0 && nq <= MAXQ && TM > 0, "soft_targets: bad geometry");
  SoftArgs a{logits, boxes, match, tboxes, tcount, pboxes, plabels, pcount, pweight, scalars4, B, nq, TM};
  hipLaunchKernelGGL(soft_targets_kernel, dim3(1), dim3(256), 0, stream, a);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}

// ---- target packing (prepare_soft_targets :215-234): xyxy pixel boxes -> normalised cxcywh rows of the [B, max_targets, 4] target
// array, one launch instead of the framework's cat / div / add / sub / stack / index_put / zeros chain.  The host uploads ONE table:
// n rows {source pointer or inline box, image w / h, destination row} followed by the B per-image counts.
namespace {
struct PackRow { const float* src; float box[4]; float w, h; int dst; int pad; };      // 40 bytes
__global__ __launch_bounds__(256) void pack_targets_kernel(const PackRow* __restrict__ rows, int n, float* __restrict__ boxes,
                                                           int* __restrict__ count, int B, int TM) {
  const int* counts_src = (const int*)(rows + n);
  for (int i = threadIdx.x; i < B * TM * 4; i += 256) boxes[i] = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) count[b] = counts_src[b];
  __syncthreads();
  for (int r = threadIdx.x; r < n; r += 256) {
    const PackRow q = rows[r];
    float x0, y0, x1, y1;
    if (q.src) { x0 = q.src[0]; y0 = q.src[1]; x1 = q.src[2]; y1 = q.src[3]; }
    else { x0 = q.box[0]; y0 = q.box[1]; x1 = q.box[2]; y1 = q.box[3]; }
    x0 = x0 / q.w; y0 = y0 / q.h; x1 = x1 / q.w; y1 = y1 / q.h;          // the reference's order: normalise, then convert
    float* o = boxes + (long)q.dst * 4;
    o[0] = (x0 + x1) / 2.f;
    o[1] = (y0 + y1) / 2.f;
    o[2] = x1 - x0;
    o[3] = y1 - y0;
  }
}
}  // namespace

extern "C" int simvg_pack_targets(const void* table, int n_rows, float* boxes, int* count, int B, int max_targets, hipStream_t stream) {
  SIMVG_CHECK_ARG(table && boxes && count && n_rows >= 0 && B > 0 && max_targets > 0, "pack_targets: bad arguments");
  hipLaunchKernelGGL(pack_targets_kernel, dim3(1), dim3(256), 0, stream, (const PackRow*)table, n_rows, boxes, count, B, max_targets);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}

extern "C" int simvg_criterion(const float* logits, const float* boxes, const int* match, const float* tboxes,
                               const int* tlabels, const float* num_boxes, const float* weights_distill,
                               float* dlogits, float* dboxes, float* out, int L, int B, int nq, int TM,
                               int coef_mode, float coef, float eos_coef, float w_class, float w_bbox, float w_giou,
                               hipStream_t stream) {
  SIMVG_CHECK_ARG(L > 0 && L <= 64 && B > 0 && nq > 0 && TM > 0, "criterion: bad geometry");
  SIMVG_CHECK_ARG(coef_mode == 0 || weights_distill != nullptr, "criterion: coef_mode 1/2 needs weights_distill");
  CritArgs a{logits, boxes, match, tboxes, tlabels, num_boxes, weights_distill, dlogits, dboxes, out, L, B, nq, TM,
             coef_mode, coef, eos_coef, w_class, w_bbox, w_giou};
  hipLaunchKernelGGL(criterion_kernel, dim3(1), dim3(256), 0, stream, a);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}

extern "C" int simvg_postprocess(const float* logits, const float* boxes, const float* wh, const float* scale_factor,
                                 float* scores, long long* labels, float* xyxy, unsigned char* keep, float* best_box,
                                 long long* best_label, int B, int num_queries, int num_cols, hipStream_t stream) {
  SIMVG_CHECK_ARG(B > 0 && num_queries > 0 && num_queries <= 16 && num_cols >= 2, "postprocess: num_queries <= 16, >= 2 class columns");
  SIMVG_CHECK_ARG(logits && boxes && wh && scores && labels && xyxy && keep && best_box && best_label, "postprocess: null buffer");
  hipLaunchKernelGGL(postprocess_kernel, dim3(B), dim3(64), 0, stream, logits, boxes, wh, scale_factor, scores, labels, xyxy,
                     keep, best_box, best_label, num_queries, num_cols);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}
