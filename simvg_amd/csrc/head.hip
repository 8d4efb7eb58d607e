// Decoder-head kernels (gfx950): everything that runs on the [B*num_queries, 256] query rows.
//
//  simvg_gemm_f32      : exact-fp32 strided GEMM on v_mfma_f32_16x16x4_f32 (bitwise an fmaf chain) for the
//                        small Linears of the head (M = B*nq .. B*20 rows): forward, dgrad and wgrad are the
//                        same kernel with different strides.  Replaces nn.Linear inside detrex
//                        MultiheadAttention/FFN (transformer.py:106-125), heads/utils.py MLP (:39-46) and
//                        tgqs_kd_detr_head.py:378-379,415-416,427-428.
//  simvg_attn_small_*  : torch.nn.MultiheadAttention core (8 heads x 32) for nq <= 16 queries against
//                        <= 512 keys (self-attn over queries, cross-attn to 20 text keys / 400 patches),
//                        with key_padding_mask and optional attention-dropout mask; fp32, one workgroup per
//                        (sample, head).  (detrex MultiheadAttention)
#include "common.h"
#include <stdlib.h>

// mirror of include/simvg_hip.h
struct simvg_gemm_f32_problem {
  const float* A; long sam, sak;
  const float* B; long sbk, sbn;
  float* C; long ldc;
  const float* bias;
  const float* addend; long ld_addend; int addend_rows;
  int M, N, K, accumulate, act;
  const float* A2; const float* B2;
  const float* mult; long ld_mult;
  const float* gate; long ld_gate;
};

namespace {

struct SGArgs {
  const float* A; long sam, sak;
  const float* B; long sbk, sbn;
  float* C; long ldc;
  const float* bias;
  const float* addend; long lda2;     // optional C += addend[m % add_rows][n]
  int add_rows;
  int M, N, K, accumulate, act;
  const float* A2;                    // optional second A operand with A's strides: the product uses A + A2 (q = (x + pos) W)
  const float* B2;                    // same for B
  const float* mult; long ldm;        // optional elementwise factor after the activation (dropout multipliers)
  const float* gate; long ldg;        // optional gate: the value passes where gate > 0 (ReLU backward on the saved output)
};

// bias, [addend], activation, [* mult], [gate], store.  With `mult` or `gate` the addend is added AFTER them
// (y = dropout(act(x W + b)) + residual); without, before the activation as it always was.
__device__ __forceinline__ void sg_store(const SGArgs& a, float v, int m, int n) {
  if (a.bias) v += a.bias[n];
  const float add = a.addend ? a.addend[(long)(m % a.add_rows) * a.lda2 + n] : 0.f;
  const bool post = a.mult != nullptr || a.gate != nullptr;
  if (!post) v += add;
  if (a.act == 2) v = fmaxf(v, 0.f);
  else if (a.act == 1) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));   // exact mode: libm erf
  else if (a.act == 3) v = 1.0f / (1.0f + expf(-v));                                 // sigmoid (the box heads' last Linear)
  if (a.mult) v *= a.mult[(long)m * a.ldm + n];
  if (a.gate) v = a.gate[(long)m * a.ldg + n] > 0.f ? v : 0.f;
  if (post) v += add;
  float* p = a.C + (long)m * a.ldc + n;
  *p = a.accumulate ? *p + v : v;
}

constexpr int SG_LDS_FLOATS = 2 * 32 * 80;      // two operand images of a 32-deep chunk ([k][68] / [k][80] / [row][36] floats)

__device__ __forceinline__ void gemm_f32_tile64(const SGArgs& a, int bx, int by, float* smem) {
  // 64x64 tile, K-chunks of 32 staged through LDS, next chunk prefetched into registers while the current one
  // feeds v_mfma_f32_16x16x4_f32 (the head's GEMMs are latency-bound: few blocks, long K loops)
  float (*As)[68] = (float (*)[68])smem;
  float (*Bs)[68] = (float (*)[68])(smem + 32 * 68);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = by * 64, n0 = bx * 64;
  f32x4_t acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float ra[8], rb[8];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = tid + 256 * i;
      int m, k;
      if (a.sak == 1) { k = e & 31; m = e >> 5; } else { m = e & 63; k = e >> 6; }
      const long ia = (long)(m0 + m) * a.sam + (long)(k0 + k) * a.sak;
      ra[i] = (m0 + m < a.M && k0 + k < a.K) ? (a.A2 ? a.A[ia] + a.A2[ia] : a.A[ia]) : 0.f;
      int n, kb;
      if (a.sbn == 1) { n = e & 63; kb = e >> 6; } else { kb = e & 31; n = e >> 5; }
      const long ib = (long)(k0 + kb) * a.sbk + (long)(n0 + n) * a.sbn;
      rb[i] = (n0 + n < a.N && k0 + kb < a.K) ? (a.B2 ? a.B[ib] + a.B2[ib] : a.B[ib]) : 0.f;
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < a.K; k0 += 32) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = tid + 256 * i;
      int m, k;
      if (a.sak == 1) { k = e & 31; m = e >> 5; } else { m = e & 63; k = e >> 6; }
      As[k][m] = ra[i];
      int n, kb;
      if (a.sbn == 1) { n = e & 63; kb = e >> 6; } else { kb = e & 31; n = e >> 5; }
      Bs[kb][n] = rb[i];
    }
    __syncthreads();
    if (k0 + 32 < a.K) fetch(k0 + 32);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const float av = As[kk * 4 + (lane >> 4)][wave * 16 + (lane & 15)];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float bv = Bs[kk * 4 + (lane >> 4)][j * 16 + (lane & 15)];
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[j], 0, 0, 0);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + j * 16 + (lane & 15);
    if (n >= a.N) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wave * 16 + 4 * (lane >> 4) + r;
      if (m < a.M) sg_store(a, acc[j][r], m, n);
    }
  }
}

// Mid-size problems (num_queries = 10: M = B * 10 = 640 query rows; weight gradients over those rows): the 64 x 64 tile again, with
//   * 16-byte global loads in either orientation -- an operand is K-contiguous (its 64 x 32 part of a chunk = 8 float4 per row,
//     LDS image [row][36]) or row-contiguous (32 k-lines of 16 float4, LDS image [k][80]); both images take ds_write_b128 and give
//     conflict-free ds_read_b32 MFMA operands (36 m and 16 k are distinct bank offsets over the 2 x 32 lane groups);
//   * split-K: one CU streams operands at ~50 GB/s whatever it does (profiles/r05_sweeps.md section 2), so a workgroup that walks
//     K = 2048 for one tile needs >= 20 us however few tiles there are; `S` workgroups per tile each take K / S and leave their
//     partial tile in a workspace slab [s][M][N]; `gemm_f32_fixup_kernel` adds the slabs in a fixed order and applies the
//     epilogue (deterministic: no atomics, no cross-workgroup handshake).  S = 1: epilogue in place, no second launch.
// Eligible: K % 32 == 0, 16-byte aligned operands and leading strides, row-contiguous operands with M (N) % 4 == 0
// (`sg_v64_ok`); everything else stays on the kernels above.  (num_queries = 10, 81 launches of a training step: 2.31 ms -> see
// profiles/r05_sweeps.md section 6.)
constexpr int SGV_KROW = 36, SGV_NROW = 80, SGV_OP_FLOATS = 32 * SGV_NROW;      // 2560 floats >= 64 * 36
template <bool AK, bool BK>
__device__ __forceinline__ void gemm_f32_tile64v_t(const SGArgs& a, int bx, int by, int k_lo, int k_hi, float* partial, float* smem) {
  float* As = smem;
  float* Bs = smem + SGV_OP_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = by * 64, n0 = bx * 64;
  f32x4_t acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  f32x4_t ra[2], rb[2];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + 256 * i;
      const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
      if (AK) {
        const int m = e >> 3, kq = e & 7;
        const long ia = (long)(m0 + m) * a.sam + k0 + 4 * kq;
        ra[i] = m0 + m < a.M ? (a.A2 ? *(const f32x4_t*)(a.A + ia) + *(const f32x4_t*)(a.A2 + ia) : *(const f32x4_t*)(a.A + ia)) : z;
      } else {
        const int k = e >> 4, mq = e & 15;
        const long ia = (long)(k0 + k) * a.sak + m0 + 4 * mq;
        ra[i] = m0 + 4 * mq < a.M ? (a.A2 ? *(const f32x4_t*)(a.A + ia) + *(const f32x4_t*)(a.A2 + ia) : *(const f32x4_t*)(a.A + ia)) : z;
      }
      if (BK) {
        const int n = e >> 3, kq = e & 7;
        const long ib = (long)(n0 + n) * a.sbn + k0 + 4 * kq;
        rb[i] = n0 + n < a.N ? (a.B2 ? *(const f32x4_t*)(a.B + ib) + *(const f32x4_t*)(a.B2 + ib) : *(const f32x4_t*)(a.B + ib)) : z;
      } else {
        const int k = e >> 4, nq = e & 15;
        const long ib = (long)(k0 + k) * a.sbk + n0 + 4 * nq;
        rb[i] = n0 + 4 * nq < a.N ? (a.B2 ? *(const f32x4_t*)(a.B + ib) + *(const f32x4_t*)(a.B2 + ib) : *(const f32x4_t*)(a.B + ib)) : z;
      }
    }
  };
  fetch(k_lo);
  const int l15 = lane & 15, g = lane >> 4;
  for (int k0 = k_lo; k0 < k_hi; k0 += 32) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + 256 * i;
      if (AK) *(f32x4_t*)(As + (e >> 3) * SGV_KROW + 4 * (e & 7)) = ra[i];
      else *(f32x4_t*)(As + (e >> 4) * SGV_NROW + 4 * (e & 15)) = ra[i];
      if (BK) *(f32x4_t*)(Bs + (e >> 3) * SGV_KROW + 4 * (e & 7)) = rb[i];
      else *(f32x4_t*)(Bs + (e >> 4) * SGV_NROW + 4 * (e & 15)) = rb[i];
    }
    __syncthreads();
    if (k0 + 32 < k_hi) fetch(k0 + 32);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const float av = AK ? As[(wave * 16 + l15) * SGV_KROW + kk * 4 + g] : As[(kk * 4 + g) * SGV_NROW + wave * 16 + l15];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float bv = BK ? Bs[(j * 16 + l15) * SGV_KROW + kk * 4 + g] : Bs[(kk * 4 + g) * SGV_NROW + j * 16 + l15];
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[j], 0, 0, 0);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + j * 16 + l15;
    if (n >= a.N) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wave * 16 + 4 * g + r;
      if (m >= a.M) continue;
      if (partial) partial[(long)m * a.N + n] = acc[j][r];
      else sg_store(a, acc[j][r], m, n);
    }
  }
}
__device__ __forceinline__ void gemm_f32_tile64v(const SGArgs& a, int bx, int by, int k_lo, int k_hi, float* partial, float* smem) {
  if (a.sak == 1) {
    if (a.sbk == 1) gemm_f32_tile64v_t<true, true>(a, bx, by, k_lo, k_hi, partial, smem);
    else gemm_f32_tile64v_t<true, false>(a, bx, by, k_lo, k_hi, partial, smem);
  } else {
    if (a.sbk == 1) gemm_f32_tile64v_t<false, true>(a, bx, by, k_lo, k_hi, partial, smem);
    else gemm_f32_tile64v_t<false, false>(a, bx, by, k_lo, k_hi, partial, smem);
  }
}
// host side: may this problem run on the tile above?  (a K-contiguous operand: element stride 1 along k; otherwise it must be
// contiguous along its rows / columns)
inline bool sg_v64_ok(const SGArgs& a) {
  auto al = [](const void* p) { return (((unsigned long)p) & 15) == 0; };
  if (a.K % 32 != 0 || !al(a.A) || !al(a.B) || (a.A2 && !al(a.A2)) || (a.B2 && !al(a.B2))) return false;
  if (a.sak == 1) { if (a.sam % 4 != 0) return false; }
  else if (a.sam != 1 || a.sak % 4 != 0 || a.M % 4 != 0) return false;
  if (a.sbk == 1) { if (a.sbn % 4 != 0) return false; }
  else if (a.sbn != 1 || a.sbk % 4 != 0 || a.N % 4 != 0) return false;
  return (double)a.M * a.N * a.K >= 40e6;        // below: the small-problem kernel (every shape of num_queries = 1 stays where it was)
}

// Small-problem variant (the head's M = B*num_queries GEMMs: 64x256x256 and friends).  The 64x64 LDS kernel above puts
// such a problem on 4 workgroups that each walk the whole K loop serially (21 us for 64x256x256, pure latency).  Here
// one workgroup owns ONE 16x16 output tile and its 4 waves split K (interleaved 16-k steps); operands go straight from
// global memory into the MFMA lanes -- the 4 "k slots" of v_mfma_f32_16x16x4_f32 may hold ANY 4 k values as long as A
// and B agree, so a lane takes 4 CONSECUTIVE k (one 16-B load when that operand is K-contiguous) and 4 MFMAs eat 16 k.
// Partial tiles are reduced through LDS; same epilogue as above.  64x256x256 -> 64 workgroups x 4 steps per wave.
__device__ __forceinline__ f32x4_t load_k4(const float* base, long sk, bool vec, bool ok, int k, int K) {
  f32x4_t v = {0.f, 0.f, 0.f, 0.f};
  if (!ok) return v;
  if (vec && k + 4 <= K) return *(const f32x4_t*)(base + k);
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (k + i < K) v[i] = base[(long)(k + i) * sk];
  return v;
}
template <int U>      // U 16-k steps of a wave in flight (loaded before the previous U are consumed)
__device__ __forceinline__ void gemm_f32_tile16_u(const SGArgs& a, int bx, int by, float* smem) {
  float (*part)[256] = (float (*)[256])smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int m = by * 16 + r, n = bx * 16 + r;
  const bool mok = m < a.M, nok = n < a.N;
  const float* Ap = a.A + (long)m * a.sam;
  const float* Bp = a.B + (long)n * a.sbn;
  const bool avec = a.sak == 1 && (a.sam & 3) == 0 && (((unsigned long)a.A) & 15) == 0;
  const bool bvec = a.sbk == 1 && (a.sbn & 3) == 0 && (((unsigned long)a.B) & 15) == 0;
  const bool avec2 = avec && (((unsigned long)a.A2) & 15) == 0, bvec2 = bvec && (((unsigned long)a.B2) & 15) == 0;
  const float* Ap2 = a.A2 ? a.A2 + (long)m * a.sam : nullptr;
  const float* Bp2 = a.B2 ? a.B2 + (long)n * a.sbn : nullptr;
  auto ldA = [&](bool ok, int k) {
    f32x4_t v = load_k4(Ap, a.sak, avec2, ok, k, a.K);
    if (Ap2) { const f32x4_t w = load_k4(Ap2, a.sak, avec2, ok, k, a.K); v += w; }
    return v;
  };
  auto ldB = [&](bool ok, int k) {
    f32x4_t v = load_k4(Bp, a.sbk, bvec2, ok, k, a.K);
    if (Bp2) { const f32x4_t w = load_k4(Bp2, a.sbk, bvec2, ok, k, a.K); v += w; }
    return v;
  };
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  const int nsteps = (a.K + 15) >> 4;
  // wave w takes steps w, w+4, ...; U steps in flight (the next U are loaded before the current U are consumed): the loop is a
  // chain of memory latencies, K = 2048 (the FFN's second Linear) is 32 steps per wave -- 16 round trips at U = 2 (33 us), 4 at U = 8
  f32x4_t av[U], bv[U], an[U], bn[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int st = wave + 4 * u;
    av[u] = ldA(mok && st < nsteps, st * 16 + 4 * g);
    bv[u] = ldB(nok && st < nsteps, st * 16 + 4 * g);
  }
  for (int s = wave; s < nsteps; s += 4 * U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int st = s + 4 * U + 4 * u;
      an[u] = ldA(mok && st < nsteps, st * 16 + 4 * g);
      bn[u] = ldB(nok && st < nsteps, st * 16 + 4 * g);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][i], bv[u][i], acc, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < U; ++u) { av[u] = an[u]; bv[u] = bn[u]; }
  }
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) part[wave][(4 * g + rr) * 16 + r] = acc[rr];
  __syncthreads();
  const int ml = tid >> 4, nl = tid & 15;
  const int mo = by * 16 + ml, no = bx * 16 + nl;
  if (mo >= a.M || no >= a.N) return;
  sg_store(a, part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid], mo, no);
}

__device__ __forceinline__ void gemm_f32_tile16(const SGArgs& a, int bx, int by, float* smem) {
  // deep pipelining only where a k4 group is ONE 16-B load (both operands K-contiguous: the forward Linears); the strided
  // operands of a weight gradient take four loads per group, and eight of those in flight per operand ran slower (45 vs 40 us)
  if (a.K >= 1024 && a.sak == 1 && a.sbk == 1) gemm_f32_tile16_u<8>(a, bx, by, smem);
  else gemm_f32_tile16_u<2>(a, bx, by, smem);
}

__global__ __launch_bounds__(256) void gemm_f32_kernel(SGArgs a) {
  __shared__ float smem[SG_LDS_FLOATS];
  gemm_f32_tile64(a, blockIdx.x, blockIdx.y, smem);
}
__global__ __launch_bounds__(256) void gemm_f32_v64_kernel(SGArgs a) {
  __shared__ float smem[SG_LDS_FLOATS];
  gemm_f32_tile64v(a, blockIdx.x, blockIdx.y, 0, a.K, nullptr, smem);
}
__global__ __launch_bounds__(256) void gemm_f32_small_kernel(SGArgs a) {
  __shared__ float smem[4 * 256];
  gemm_f32_tile16(a, blockIdx.x, blockIdx.y, smem);
}

// Several INDEPENDENT problems in one launch (the dgrad / wgrad / bias-gradient GEMMs of one layer, the q|k and v
// projections of an attention ...): the head is a chain of ~5 us launches, so what matters is how many there are.
// blockIdx.x walks the concatenated tile lists; each problem keeps the tile shape the single-problem entry point
// would have picked for it.
constexpr int SG_MAX = 12;
struct SGGroup {
  SGArgs p[SG_MAX];
  int start[SG_MAX + 1];      // first linear block of problem i
  int nx[SG_MAX];             // tiles along N of problem i
  int small[SG_MAX];          // 1: one workgroup per 16 x 16 tile; 0: 64 x 64 tiles; 2: 64 x 64 tiles, 16-byte loads, K split S ways
  int S[SG_MAX];              // kind 2: workgroups per tile (consecutive blocks), each K / S; > 1: partial tiles to `part`
  int kchunks[SG_MAX];        // kind 2: 32-deep chunks per split
  float* part[SG_MAX];        // kind 2, S > 1: [S][M][N] partial sums (gemm_f32_fixup_kernel adds them and applies the epilogue)
  int fix_start[SG_MAX + 1];  // fix-up launch: first block of problem i (256 outputs per block; problems with S == 1 have none)
  int count;
};
__global__ __launch_bounds__(256) void gemm_f32_group_kernel(SGGroup g) {
  __shared__ float smem[SG_LDS_FLOATS];
  int i = 0;
  while (i + 1 < g.count && (int)blockIdx.x >= g.start[i + 1]) ++i;
  int t = blockIdx.x - g.start[i];
  const SGArgs a = g.p[i];
  if (g.small[i] == 2) {
    const int S = g.S[i], s = t % S;
    t /= S;
    const int by = t / g.nx[i], bx = t - by * g.nx[i];
    const int k_lo = s * g.kchunks[i] * 32, k_hi = min(a.K, k_lo + g.kchunks[i] * 32);
    gemm_f32_tile64v(a, bx, by, k_lo, k_hi, S > 1 ? g.part[i] + (long)s * a.M * a.N : nullptr, smem);
    return;
  }
  const int by = t / g.nx[i], bx = t - by * g.nx[i];
  if (g.small[i]) gemm_f32_tile16(a, bx, by, smem);
  else gemm_f32_tile64(a, bx, by, smem);
}
// second stage of the split-K problems of a group: output element e of problem i = sum over its S slabs (in slab order), then the
// problem's epilogue (bias / addend / activation / mult / gate / accumulate)
__global__ __launch_bounds__(256) void gemm_f32_fixup_kernel(SGGroup g) {
  int i = 0;
  while (i + 1 < g.count && (int)blockIdx.x >= g.fix_start[i + 1]) ++i;
  const SGArgs a = g.p[i];
  const long e = (long)(blockIdx.x - g.fix_start[i]) * 256 + threadIdx.x, MN = (long)a.M * a.N;
  if (e >= MN) return;
  float v = 0.f;
  for (int s = 0; s < g.S[i]; ++s) v += g.part[i][s * MN + e];
  sg_store(a, v, (int)(e / a.N), (int)(e % a.N));
}

// ---------------------------------------------------------------------------------------------
struct SAArgs {
  const float* q; int ldq;      // [B*Lq, E] (+ column offset already applied)
  const float* k; int ldk;      // [B*Lk, E]
  const float* v; int ldv;
  float* out; int ldo;          // [B*Lq, E]
  float* P;                     // [B, H, Lq, Lk] softmax probabilities (saved for backward)
  const unsigned char* kpm;     // [B, Lk] 1 = masked, or null
  const float* drop;            // [B, H, Lq, Lk] dropout multipliers (0 or 1/(1-p)), or null
  const float* dout; int lddo;
  float* dq; int lddq;
  float* dk; int lddk;
  float* dv; int lddv;
  int B, H, Lq, Lk, kv_rows;
  float scale;
  const float* kpos; int ldkp, kpos_rows;   // optional key_pos rows [Lk or B*Lk, E] added to K on load (kpos_rows: rows per
                                            // sample, 0 = one set shared by the batch)
};

constexpr int SHD = 32;

__global__ __launch_bounds__(256) void attn_small_fwd_kernel(SAArgs a) {
  extern __shared__ float sm[];
  float* qs = sm;                     // [Lq][32]
  float* sc = sm + a.Lq * SHD;        // [Lq][Lk]
  const int b = blockIdx.x / a.H, h = blockIdx.x - b * a.H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < a.Lq * SHD; e += 256)
    qs[e] = a.q[(long)(b * a.Lq + e / SHD) * a.ldq + h * SHD + (e % SHD)] * a.scale;
  __syncthreads();
  // scores: 8 consecutive lanes share a key -- lane c of the group reads the 16-B chunk c of the key's 128-B head slice, so one
  // wave instruction consumes 8 whole 128-B segments (one thread per key row read its 128 B in 8 instructions whose lanes were
  // 2 KiB apart: every segment was fetched by 8 instructions and lived in L1 in between; the kernel took 40 us for the 52 MB of
  // a 400-key layer at B = 64)
  constexpr int KB = 13;                 // 13 x 32 = 416 keys per pass: ALL of a pass's loads are issued before the first is used
  f32x4_t tv[KB];                        // the V rows of the first pass are requested together with its K rows: their latency
  {                                      // passes under the score / softmax phases (the decoder's 400 keys are one pass)
    const int c4 = (tid & 7) * 4;
    for (int k0 = 0; k0 < a.Lk; k0 += 32 * KB) {      // (one load -> use per trip exposed the memory latency 13 times)
      f32x4_t t[KB];
      if (k0 == 0) {
#pragma unroll
        for (int u = 0; u < KB; ++u) {
          const int kk = u * 32 + (tid >> 3);
          tv[u] = kk < a.Lk ? *(const f32x4_t*)(a.v + (long)(b * a.kv_rows + kk) * a.ldv + h * SHD + c4) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
        }
      }
#pragma unroll
      for (int u = 0; u < KB; ++u) {
        const int kk = k0 + u * 32 + (tid >> 3);
        t[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        if (kk < a.Lk) {
          t[u] = *(const f32x4_t*)(a.k + (long)(b * a.kv_rows + kk) * a.ldk + h * SHD + c4);
          if (a.kpos) t[u] += *(const f32x4_t*)(a.kpos + (long)(b * a.kpos_rows + kk) * a.ldkp + h * SHD + c4);
        }
      }
#pragma unroll
      for (int u = 0; u < KB; ++u) {
        const int kk = k0 + u * 32 + (tid >> 3);
        const bool in = kk < a.Lk;
        const bool masked = in && a.kpm && a.kpm[b * a.Lk + kk];
        for (int qi = 0; qi < a.Lq; ++qi) {
          const f32x4_t q4 = *(const f32x4_t*)(qs + qi * SHD + c4);
          float sv = (q4[0] * t[u][0] + q4[1] * t[u][1]) + (q4[2] * t[u][2] + q4[3] * t[u][3]);
          sv += __shfl_xor(sv, 1, 64);
          sv += __shfl_xor(sv, 2, 64);
          sv += __shfl_xor(sv, 4, 64);
          if (in && c4 == 0) sc[qi * a.Lk + kk] = masked ? -INFINITY : sv;
        }
      }
    }
  }
  __syncthreads();
  for (int qi = wave; qi < a.Lq; qi += 4) {
    float mx = -INFINITY;
    for (int kk = lane; kk < a.Lk; kk += 64) mx = fmaxf(mx, sc[qi * a.Lk + kk]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int kk = lane; kk < a.Lk; kk += 64) {
      const float p = __expf(sc[qi * a.Lk + kk] - mx);
      sc[qi * a.Lk + kk] = p;
      sum += p;
    }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    float* Pg = a.P + ((long)blockIdx.x * a.Lq + qi) * a.Lk;
    const float* dr = a.drop ? a.drop + ((long)blockIdx.x * a.Lq + qi) * a.Lk : nullptr;
    for (int kk = lane; kk < a.Lk; kk += 64) {
      const float p = sc[qi * a.Lk + kk] * inv;
      Pg[kk] = p;
      sc[qi * a.Lk + kk] = dr ? p * dr[kk] : p;
    }
  }
  __syncthreads();
  // out[qi][d] = sum_k P'[qi][k] V[k][d], in the layout of the score phase: 8 lanes share a key, lane c holds 4 of its 32 values
  // (16-B loads, all of a pass issued before the first is used: the earlier one-float-per-lane walk over 50 keys per thread was
  // the longest part of the kernel -- 4-byte loads, 8 in flight).  The 32 key slots of the block meet in LDS.
  float* red = sc + ((a.Lq * a.Lk + 3) & ~3);        // [32][Lq][32]; 16-byte aligned for the f32x4 accesses (Lq * Lk may be odd)
  {
    const int slot = tid >> 3, c4 = (tid & 7) * 4;
    f32x4_t o[16];
#pragma unroll
    for (int qi = 0; qi < 16; ++qi) o[qi] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < a.Lk; k0 += 32 * KB) {
      f32x4_t t[KB];
#pragma unroll
      for (int u = 0; u < KB; ++u) {
        const int kk = k0 + u * 32 + slot;
        if (k0 == 0) t[u] = tv[u];
        else t[u] = kk < a.Lk ? *(const f32x4_t*)(a.v + (long)(b * a.kv_rows + kk) * a.ldv + h * SHD + c4) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < KB; ++u) {
        const int kk = k0 + u * 32 + slot;
        if (kk < a.Lk) {
#pragma unroll
          for (int qi = 0; qi < 16; ++qi)
            if (qi < a.Lq) o[qi] += sc[qi * a.Lk + kk] * t[u];
        }
      }
    }
#pragma unroll
    for (int qi = 0; qi < 16; ++qi)
      if (qi < a.Lq) *(f32x4_t*)(red + (slot * a.Lq + qi) * SHD + c4) = o[qi];
    __syncthreads();
    for (int e = tid; e < a.Lq * SHD; e += 256) {
      float t = 0.f;
#pragma unroll
      for (int p32 = 0; p32 < 32; ++p32) t += red[p32 * a.Lq * SHD + e];
      a.out[(long)(b * a.Lq + e / SHD) * a.ldo + h * SHD + (e % SHD)] = t;
    }
  }
}

__global__ __launch_bounds__(256) void attn_small_bwd_kernel(SAArgs a) {
  // Same thread layout as the forward: 8 consecutive lanes share a key, lane c holds 4 of the 32 values of its head slice (16-B
  // loads and stores, every 128-B segment touched by one instruction; the loads of a pass of 13 x 32 keys are issued together).
  extern __shared__ float sm[];
  float* qs = sm;                           // [Lq][32] (unscaled)
  float* dos = qs + a.Lq * SHD;             // [Lq][32]
  float* ds = dos + a.Lq * SHD;             // [Lq][Lk]  dS
  float* pd = ds + a.Lq * a.Lk;             // [Lq][Lk]  P * dropmask
  float* rs = pd + a.Lq * a.Lk;             // [Lq] rowsum(dP * P)
  const int b = blockIdx.x / a.H, h = blockIdx.x - b * a.H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int slot = tid >> 3, c4 = (tid & 7) * 4;
  constexpr int KB = 13;
  for (int e = tid; e < a.Lq * SHD; e += 256) {
    const long r = (long)(b * a.Lq + e / SHD);
    qs[e] = a.q[r * a.ldq + h * SHD + (e % SHD)];
    dos[e] = a.dout[r * a.lddo + h * SHD + (e % SHD)];
  }
  __syncthreads();
  // dP = (dO . V^T) * drop ; dV = (P*drop)^T dO
  for (int k0 = 0; k0 < a.Lk; k0 += 32 * KB) {
    f32x4_t t[KB];
#pragma unroll
    for (int u = 0; u < KB; ++u) {
      const int kk = k0 + u * 32 + slot;
      t[u] = kk < a.Lk ? *(const f32x4_t*)(a.v + (long)(b * a.kv_rows + kk) * a.ldv + h * SHD + c4) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < KB; ++u) {
      const int kk = k0 + u * 32 + slot;
      const bool in = kk < a.Lk;
      f32x4_t dvr = {0.f, 0.f, 0.f, 0.f};
      for (int qi = 0; qi < a.Lq; ++qi) {
        const long pi = ((long)blockIdx.x * a.Lq + qi) * a.Lk + (in ? kk : 0);
        const float p = a.P[pi];
        const float dm = a.drop ? a.drop[pi] : 1.f;
        const f32x4_t d4 = *(const f32x4_t*)(dos + qi * SHD + c4);
        float dp = (d4[0] * t[u][0] + d4[1] * t[u][1]) + (d4[2] * t[u][2] + d4[3] * t[u][3]);
        dp += __shfl_xor(dp, 1, 64);
        dp += __shfl_xor(dp, 2, 64);
        dp += __shfl_xor(dp, 4, 64);
        dvr += (p * dm) * d4;
        if (in && c4 == 0) {
          ds[qi * a.Lk + kk] = dp * dm;   // dP wrt softmax output
          pd[qi * a.Lk + kk] = p;
        }
      }
      if (in) *(f32x4_t*)(a.dv + (long)(b * a.kv_rows + kk) * a.lddv + h * SHD + c4) = dvr;
    }
  }
  __syncthreads();
  for (int qi = wave; qi < a.Lq; qi += 4) {
    float s = 0.f;
    for (int kk = lane; kk < a.Lk; kk += 64) s += ds[qi * a.Lk + kk] * pd[qi * a.Lk + kk];
    s = wave_sum(s);
    if (lane == 0) rs[qi] = s;
  }
  __syncthreads();
  for (int e = tid; e < a.Lq * a.Lk; e += 256) {
    const int qi = e / a.Lk;
    ds[e] = pd[e] * (ds[e] - rs[qi]) * a.scale;      // dS * scale (masked keys: P = 0 -> 0)
  }
  __syncthreads();
  // dQ[qi][d] = sum_k dS[qi][k] K[k][d] (the 32 key slots of the block meet in LDS);  dK[k][d] = sum_q dS[q][k] Q[q][d]
  {
    float* red = sm + (((int)(rs - sm) + a.Lq + 3) & ~3);      // [32][Lq][32]; 16-byte aligned for the f32x4 accesses (2 Lq Lk + Lq may be odd)
    f32x4_t o[16];
#pragma unroll
    for (int qi = 0; qi < 16; ++qi) o[qi] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < a.Lk; k0 += 32 * KB) {
      f32x4_t t[KB];
#pragma unroll
      for (int u = 0; u < KB; ++u) {
        const int kk = k0 + u * 32 + slot;
        t[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        if (kk < a.Lk) {
          t[u] = *(const f32x4_t*)(a.k + (long)(b * a.kv_rows + kk) * a.ldk + h * SHD + c4);
          if (a.kpos) t[u] += *(const f32x4_t*)(a.kpos + (long)(b * a.kpos_rows + kk) * a.ldkp + h * SHD + c4);
        }
      }
#pragma unroll
      for (int u = 0; u < KB; ++u) {
        const int kk = k0 + u * 32 + slot;
        if (kk < a.Lk) {
          f32x4_t dkr = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int qi = 0; qi < 16; ++qi)
            if (qi < a.Lq) {
              const float sv = ds[qi * a.Lk + kk];
              o[qi] += sv * t[u];
              dkr += sv * *(const f32x4_t*)(qs + qi * SHD + c4);
            }
          *(f32x4_t*)(a.dk + (long)(b * a.kv_rows + kk) * a.lddk + h * SHD + c4) = dkr;
        }
      }
    }
#pragma unroll
    for (int qi = 0; qi < 16; ++qi)
      if (qi < a.Lq) *(f32x4_t*)(red + (slot * a.Lq + qi) * SHD + c4) = o[qi];
    __syncthreads();
    for (int e = tid; e < a.Lq * SHD; e += 256) {
      float t = 0.f;
#pragma unroll
      for (int p32 = 0; p32 < 32; ++p32) t += red[p32 * a.Lq * SHD + e];
      a.dq[(long)(b * a.Lq + e / SHD) * a.lddq + h * SHD + (e % SHD)] = t;
    }
  }
}

}  // namespace

extern "C" int simvg_gemm_f32(const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C,
                              long ldc, const float* bias, const float* addend, long ld_addend, int addend_rows,
                              int M, int N, int K, int accumulate, int act, hipStream_t stream) {
  SIMVG_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm_f32: empty problem");
  SIMVG_CHECK_ARG(act >= 0 && act <= 3, "gemm_f32: act must be 0 (none), 1 (gelu), 2 (relu) or 3 (sigmoid)");
  SGArgs a{A, sam, sak, B, sbk, sbn, C, ldc, bias, addend, ld_addend, addend_rows > 0 ? addend_rows : 1, M, N, K,
           accumulate, act, nullptr, nullptr, nullptr, 0, nullptr, 0};
  // <= 128 tiles of 64x64: the small-M kernel (one workgroup per 16x16 tile, K split over its waves).  32 in round 1 (tuned at
  // num_queries = 1); at num_queries = 10 (M = 640 rows) the K = 2048 FFN problems have 40 - 320 such tiles and ran 27+ us each on
  // the 64x64 kernel: 32 -> 128 is 34.2 -> 32.7 ms per step there, no change at num_queries = 1 (profiles/r02_sweeps.md)
  constexpr int small_env = 128;
  if (sg_v64_ok(a) && cdiv(N, 64) * cdiv(M, 64) > small_env)     // mid-size, enough tiles without a K split: 16-byte loads
    hipLaunchKernelGGL(gemm_f32_v64_kernel, dim3(cdiv(N, 64), cdiv(M, 64)), dim3(256), 0, stream, a);
  else if (cdiv(N, 64) * cdiv(M, 64) <= small_env)   // too few 64x64 tiles to fill the chip: one workgroup per 16x16 tile
    hipLaunchKernelGGL(gemm_f32_small_kernel, dim3(cdiv(N, 16), cdiv(M, 16)), dim3(256), 0, stream, a);
  else
    hipLaunchKernelGGL(gemm_f32_kernel, dim3(cdiv(N, 64), cdiv(M, 64)), dim3(256), 0, stream, a);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}

static int gemm_f32_grouped_impl(const simvg_gemm_f32_problem* problems, int count, float* ws, long ws_floats, hipStream_t stream) {
  SIMVG_CHECK_ARG(problems != nullptr && count > 0 && count <= SG_MAX, "gemm_f32_grouped: 1..12 problems");
  SIMVG_CHECK_ARG(ws_floats >= 0 && (ws != nullptr || ws_floats == 0) && ((((unsigned long)ws) & 15) == 0), "gemm_f32_grouped: workspace must be 16-byte aligned");
  // <= 128 tiles of 64x64: the small-M kernel (one workgroup per 16x16 tile, K split over its waves).  32 in round 1 (tuned at
  // num_queries = 1); at num_queries = 10 (M = 640 rows) the K = 2048 FFN problems have 40 - 320 such tiles and ran 27+ us each on
  // the 64x64 kernel: 32 -> 128 is 34.2 -> 32.7 ms per step there, no change at num_queries = 1 (profiles/r02_sweeps.md).
  // Round 5: problems of >= 40 M multiply-adds with aligned operands take the 16-byte-load tile, K split over `S` workgroups
  // per tile when a workspace is given (target ~320 workgroups, >= 128 k each)
  constexpr int small_env = 128;
  SGGroup g;
  g.count = count;
  int total = 0, fix_total = 0;
  long ws_used = 0;
  for (int i = 0; i < count; ++i) {
    const simvg_gemm_f32_problem& q = problems[i];
    SIMVG_CHECK_ARG(q.M > 0 && q.N > 0 && q.K > 0, "gemm_f32_grouped: empty problem");
    SIMVG_CHECK_ARG(q.act >= 0 && q.act <= 3, "gemm_f32_grouped: act must be 0 (none), 1 (gelu), 2 (relu) or 3 (sigmoid)");
    g.p[i] = SGArgs{q.A, q.sam, q.sak, q.B, q.sbk, q.sbn, q.C, q.ldc, q.bias, q.addend, q.ld_addend,
                    q.addend_rows > 0 ? q.addend_rows : 1, q.M, q.N, q.K, q.accumulate, q.act,
                    q.A2, q.B2, q.mult, q.ld_mult, q.gate, q.ld_gate};
    g.S[i] = 1; g.kchunks[i] = 0; g.part[i] = nullptr;
    g.fix_start[i] = fix_total;
    g.start[i] = total;
    const int tiles = cdiv(q.N, 64) * cdiv(q.M, 64);
    if (sg_v64_ok(g.p[i])) {
      const int chunks = q.K / 32;
      static const int target = getenv("SIMVG_SGV_TARGET") ? atoi(getenv("SIMVG_SGV_TARGET")) : 320;     // (sweep knobs)
      static const int minch = getenv("SIMVG_SGV_MINCH") ? atoi(getenv("SIMVG_SGV_MINCH")) : 4;
      int S = cdiv(target, tiles);
      if (S > chunks / minch) S = chunks / minch;
      const long MN = (long)q.M * q.N;
      while (S > 1 && ws_used + (long)S * MN > ws_floats) --S;       // (no / a small workspace: fewer splits)
      if (S < 1) S = 1;
      int per = cdiv(chunks, S);
      S = cdiv(chunks, per);
      if (S == 1 && tiles < 40 && tiles <= small_env) {               // few tiles, no split possible: the small-problem kernel
        g.small[i] = 1; g.nx[i] = cdiv(q.N, 16);
        total += cdiv(q.N, 16) * cdiv(q.M, 16);
        continue;
      }
      g.small[i] = 2; g.nx[i] = cdiv(q.N, 64); g.S[i] = S; g.kchunks[i] = per;
      if (S > 1) {
        g.part[i] = ws + ws_used;
        ws_used += (S * MN + 3) / 4 * 4;
        fix_total += (int)((MN + 255) / 256);
      }
      total += tiles * S;
      continue;
    }
    const int sm = tiles <= small_env;
    const int t = sm ? 16 : 64;
    g.small[i] = sm;
    g.nx[i] = cdiv(q.N, t);
    total += cdiv(q.N, t) * cdiv(q.M, t);
  }
  for (int i = count; i <= SG_MAX; ++i) { g.start[i] = total; g.fix_start[i] = fix_total; }
  for (int i = count; i < SG_MAX; ++i) { g.p[i] = g.p[0]; g.nx[i] = 1; g.small[i] = 1; g.S[i] = 1; g.kchunks[i] = 0; g.part[i] = nullptr; }
  hipLaunchKernelGGL(gemm_f32_group_kernel, dim3(total), dim3(256), 0, stream, g);
  SIMVG_LAUNCH_CHECK();
  if (fix_total > 0) {
    hipLaunchKernelGGL(gemm_f32_fixup_kernel, dim3(fix_total), dim3(256), 0, stream, g);
    SIMVG_LAUNCH_CHECK();
  }
  return SIMVG_OK;
}

extern "C" int simvg_gemm_f32_grouped(const simvg_gemm_f32_problem* problems, int count, hipStream_t stream) {
  return gemm_f32_grouped_impl(problems, count, nullptr, 0, stream);
}

extern "C" int simvg_gemm_f32_grouped_ws(const simvg_gemm_f32_problem* problems, int count, float* workspace, long workspace_floats,
                                         hipStream_t stream) {
  return gemm_f32_grouped_impl(problems, count, workspace, workspace_floats, stream);
}

extern "C" int simvg_attn_small_fwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                                    float* out, int ldo, float* P, const unsigned char* key_padding_mask,
                                    const float* drop_mult, int B, int H, int Lq, int Lk, int kv_rows_per_batch,
                                    float scale, const float* key_pos, int ld_key_pos, int key_pos_rows_per_batch,
                                    hipStream_t stream) {
  SIMVG_CHECK_ARG(B > 0 && H > 0 && Lq > 0 && Lq <= 16 && Lk > 0 && (size_t)(Lq * SHD + ((Lq * Lk + 3) & ~3) + 32 * Lq * SHD) * sizeof(float) <= 160 * 1024,
                  "attn_small: Lq <= 16 and the [Lq, Lk] score strip must fit the 160 KiB LDS");
  SIMVG_CHECK_ARG(ldk % 4 == 0 && ldv % 4 == 0, "attn_small: K/V rows must be 16-B aligned");
  SAArgs a{q, ldq, k, ldk, v, ldv, out, ldo, P, key_padding_mask, drop_mult, nullptr, 0, nullptr, 0, nullptr, 0,
           nullptr, 0, B, H, Lq, Lk, kv_rows_per_batch > 0 ? kv_rows_per_batch : Lk, scale, key_pos, ld_key_pos,
           key_pos_rows_per_batch};
  SIMVG_CHECK_ARG(!key_pos || ld_key_pos % 4 == 0, "attn_small: key_pos rows must be 16-B aligned");
  const size_t shm = (size_t)(Lq * SHD + ((Lq * Lk + 3) & ~3) + 32 * Lq * SHD) * sizeof(float);
  static bool once = hipFuncSetAttribute((const void*)attn_small_fwd_kernel,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
  (void)once;
  hipLaunchKernelGGL(attn_small_fwd_kernel, dim3(B * H), dim3(256), shm, stream, a);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}

extern "C" int simvg_attn_small_bwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                                    const float* P, const unsigned char* key_padding_mask, const float* drop_mult,
                                    const float* dout, int lddo, float* dq, int lddq, float* dk, int lddk, float* dv,
                                    int lddv, int B, int H, int Lq, int Lk, int kv_rows_per_batch, float scale,
                                    const float* key_pos, int ld_key_pos, int key_pos_rows_per_batch,
                                    hipStream_t stream) {
  SIMVG_CHECK_ARG(B > 0 && H > 0 && Lq > 0 && Lq <= 16 && Lk > 0 &&
                  (size_t)(2 * Lq * SHD + 2 * Lq * Lk + ((Lq + 3) & ~3) + 3 + 32 * Lq * SHD) * sizeof(float) <= 160 * 1024,
                  "attn_small: Lq <= 16 and two [Lq, Lk] strips must fit the 160 KiB LDS");
  SIMVG_CHECK_ARG(ldk % 4 == 0 && ldv % 4 == 0 && lddk % 4 == 0 && lddv % 4 == 0, "attn_small: rows must be 16-B aligned");
  SAArgs a{q, ldq, k, ldk, v, ldv, nullptr, 0, (float*)P, key_padding_mask, drop_mult, dout, lddo, dq, lddq, dk, lddk,
           dv, lddv, B, H, Lq, Lk, kv_rows_per_batch > 0 ? kv_rows_per_batch : Lk, scale, key_pos, ld_key_pos,
           key_pos_rows_per_batch};
  const size_t shm = (size_t)(2 * Lq * SHD + 2 * Lq * Lk + ((Lq + 3) & ~3) + 3 + 32 * Lq * SHD) * sizeof(float);
  static bool once = hipFuncSetAttribute((const void*)attn_small_bwd_kernel,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
  (void)once;
  hipLaunchKernelGGL(attn_small_bwd_kernel, dim3(B * H), dim3(256), shm, stream, a);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}

// ---------------------------------------------------------------------------------------------
// Exact-fp32 encoder attention (precision="fp32" mode: the reference's own arithmetic, fp32 end to end).
// One workgroup per (sample, head, 16-query chunk); scores for the chunk live in LDS; VALU dot products.
// Rows are modality-major like the bf16 kernel.  Forward only (inference parity mode).
// ---------------------------------------------------------------------------------------------
namespace {
struct AF32Args {
  const float* qkv; int ld;   // [M, 3D]
  float* out; int ldo;        // [M, D]
  const unsigned char* pad;   // [B, Nt] or null
  int B, H, Nv, Nt, D;
  float scale;
};
__device__ __forceinline__ long af32_row(const AF32Args& a, int b, int t) {
  return t < a.Nv ? (long)b * a.Nv + t : (long)a.B * a.Nv + (long)b * a.Nt + (t - a.Nv);
}
__global__ __launch_bounds__(256) void attn_f32_fwd_kernel(AF32Args a) {
  extern __shared__ float sm[];
  const int N = a.Nv + a.Nt;
  float* qs = sm;              // [16][64]
  float* sc = sm + 16 * 64;    // [16][N]
  const int bh = blockIdx.x, b = bh / a.H, h = bh - b * a.H;
  const int q0 = blockIdx.y * 16;
  const int nq = min(16, N - q0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < 16 * 64; e += 256) {
    const int qi = e >> 6, d = e & 63;
    qs[e] = qi < nq ? a.qkv[af32_row(a, b, q0 + qi) * a.ld + h * 64 + d] * a.scale : 0.f;
  }
  __syncthreads();
  for (int kk = tid; kk < N; kk += 256) {
    const float* kp = a.qkv + af32_row(a, b, kk) * a.ld + a.D + h * 64;
    float kr[64];
#pragma unroll
    for (int d = 0; d < 64; d += 4) {
      const f32x4_t t = *(const f32x4_t*)(kp + d);
      kr[d] = t[0]; kr[d + 1] = t[1]; kr[d + 2] = t[2]; kr[d + 3] = t[3];
    }
    const bool masked = a.pad && kk >= a.Nv && a.pad[b * a.Nt + (kk - a.Nv)];
    for (int qi = 0; qi < 16; ++qi) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < 64; ++d) s = fmaf(qs[qi * 64 + d], kr[d], s);
      sc[qi * N + kk] = masked ? -INFINITY : s;
    }
  }
  __syncthreads();
  for (int qi = wave; qi < nq; qi += 4) {
    float mx = -INFINITY;
    for (int kk = lane; kk < N; kk += 64) mx = fmaxf(mx, sc[qi * N + kk]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int kk = lane; kk < N; kk += 64) {
      const float p = expf(sc[qi * N + kk] - mx);
      sc[qi * N + kk] = p;
      sum += p;
    }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    for (int kk = lane; kk < N; kk += 64) sc[qi * N + kk] *= inv;
  }
  __syncthreads();
  for (int e = tid; e < nq * 64; e += 256) {
    const int qi = e >> 6, d = e & 63;
    float o = 0.f;
    for (int kk = 0; kk < N; ++kk) o = fmaf(sc[qi * N + kk], a.qkv[af32_row(a, b, kk) * a.ld + 2 * a.D + h * 64 + d], o);
    a.out[af32_row(a, b, q0 + qi) * a.ldo + h * 64 + d] = o;
  }
}

// exact-fp32 attention backward (precision="fp32" training mode): same decomposition as the forward -- one workgroup per
// (sample, head, 16-query chunk).  P is recomputed, dS = P o (dO V^T - rowsum(dO o O)) * scale; dQ rows are owned by the
// workgroup, dK / dV rows are accumulated across the query chunks with fp32 atomics (dqkv's K and V thirds must be
// zero on entry).  VALU dot products, libm expf: the reference's own arithmetic (torchscale MultiheadAttention).
struct AF32BwdArgs {
  const float* qkv; int ld;     // [M, 3D]
  const float* dout; int lddo;  // [M, D]
  float* dqkv; int lddq;        // [M, 3D]
  const unsigned char* pad;
  int B, H, Nv, Nt, D;
  float scale;
};
__device__ __forceinline__ long af32b_row(const AF32BwdArgs& a, int b, int t) {
  return t < a.Nv ? (long)b * a.Nv + t : (long)a.B * a.Nv + (long)b * a.Nt + (t - a.Nv);
}
__global__ __launch_bounds__(256) void attn_f32_bwd_kernel(AF32BwdArgs a) {
  extern __shared__ float sm[];
  const int N = a.Nv + a.Nt;
  float* qs = sm;                 // [16][64]  q * scale
  float* dos = qs + 16 * 64;      // [16][64]  dO
  float* sc = dos + 16 * 64;      // [16][N]   P, then dS
  float* dp = sc + 16 * N;        // [16][N]   dO V^T
  float* rs = dp + 16 * N;        // [16]
  const int bh = blockIdx.x, b = bh / a.H, h = bh - b * a.H;
  const int q0 = blockIdx.y * 16;
  const int nq = min(16, N - q0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < 16 * 64; e += 256) {
    const int qi = e >> 6, d = e & 63;
    const long r = af32b_row(a, b, q0 + (qi < nq ? qi : 0));
    qs[e] = qi < nq ? a.qkv[r * a.ld + h * 64 + d] * a.scale : 0.f;
    dos[e] = qi < nq ? a.dout[r * a.lddo + h * 64 + d] : 0.f;
  }
  __syncthreads();
  for (int kk = tid; kk < N; kk += 256) {
    const long kr_ = af32b_row(a, b, kk);
    const float* kp = a.qkv + kr_ * a.ld + a.D + h * 64;
    const float* vp = a.qkv + kr_ * a.ld + 2 * a.D + h * 64;
    float kr[64], vr[64];
#pragma unroll
    for (int d = 0; d < 64; d += 4) {
      const f32x4_t t = *(const f32x4_t*)(kp + d), u = *(const f32x4_t*)(vp + d);
      kr[d] = t[0]; kr[d + 1] = t[1]; kr[d + 2] = t[2]; kr[d + 3] = t[3];
      vr[d] = u[0]; vr[d + 1] = u[1]; vr[d + 2] = u[2]; vr[d + 3] = u[3];
    }
    const bool masked = a.pad && kk >= a.Nv && a.pad[b * a.Nt + (kk - a.Nv)];
    for (int qi = 0; qi < 16; ++qi) {
      float s = 0.f, t = 0.f;
#pragma unroll
      for (int d = 0; d < 64; ++d) { s = fmaf(qs[qi * 64 + d], kr[d], s); t = fmaf(dos[qi * 64 + d], vr[d], t); }
      sc[qi * N + kk] = masked ? -INFINITY : s;
      dp[qi * N + kk] = t;
    }
  }
  __syncthreads();
  for (int qi = wave; qi < 16; qi += 4) {
    if (qi >= nq) {       // rows beyond the sequence: contribute nothing
      for (int kk = lane; kk < N; kk += 64) sc[qi * N + kk] = 0.f;
      continue;
    }
    float mx = -INFINITY;
    for (int kk = lane; kk < N; kk += 64) mx = fmaxf(mx, sc[qi * N + kk]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int kk = lane; kk < N; kk += 64) {
      const float p = expf(sc[qi * N + kk] - mx);
      sc[qi * N + kk] = p;
      sum += p;
    }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    float r = 0.f;
    for (int kk = lane; kk < N; kk += 64) {
      const float p = sc[qi * N + kk] * inv;
      sc[qi * N + kk] = p;
      r += p * dp[qi * N + kk];
    }
    r = wave_sum(r);
    if (lane == 0) rs[qi] = r;
  }
  __syncthreads();
  // dV[k] += sum_q P[q][k] dO[q]   (before P is overwritten by dS);  then dS = P (dP - rowsum) * scale
  for (int kk = tid; kk < N; kk += 256) {
    float acc[64];
#pragma unroll
    for (int d = 0; d < 64; ++d) acc[d] = 0.f;
    for (int qi = 0; qi < nq; ++qi) {
      const float p = sc[qi * N + kk];
#pragma unroll
      for (int d = 0; d < 64; ++d) acc[d] = fmaf(p, dos[qi * 64 + d], acc[d]);
    }
    float* gv = a.dqkv + af32b_row(a, b, kk) * a.lddq + 2 * a.D + h * 64;
#pragma unroll
    for (int d = 0; d < 64; ++d) atomicAdd(gv + d, acc[d]);
  }
  __syncthreads();
  for (int e = tid; e < 16 * N; e += 256) {
    const int qi = e / N;
    sc[e] = qi < nq ? sc[e] * (dp[e] - rs[qi]) * a.scale : 0.f;       // masked keys: P = 0 -> dS = 0
  }
  __syncthreads();
  // dQ[q][d] = sum_k dS[q][k] K[k][d]
  for (int e = tid; e < nq * 64; e += 256) {
    const int qi = e >> 6, d = e & 63;
    float o = 0.f;
    for (int kk = 0; kk < N; ++kk) o = fmaf(sc[qi * N + kk], a.qkv[af32b_row(a, b, kk) * a.ld + a.D + h * 64 + d], o);
    a.dqkv[af32b_row(a, b, q0 + qi) * a.lddq + h * 64 + d] = o;
  }
  // dK[k][d] += sum_q dS[q][k] Q[q][d]     (qs holds q * scale and dS already carries one scale: use the raw q)
  for (int kk = tid; kk < N; kk += 256) {
    float acc[64];
#pragma unroll
    for (int d = 0; d < 64; ++d) acc[d] = 0.f;
    const float inv_scale = 1.f / a.scale;
    for (int qi = 0; qi < nq; ++qi) {
      const float ds = sc[qi * N + kk] * inv_scale;
#pragma unroll
      for (int d = 0; d < 64; ++d) acc[d] = fmaf(ds, qs[qi * 64 + d], acc[d]);
    }
    float* gk = a.dqkv + af32b_row(a, b, kk) * a.lddq + a.D + h * 64;
#pragma unroll
    for (int d = 0; d < 64; ++d) atomicAdd(gk + d, acc[d]);
  }
}

// exact-erf GELU (libm) forward / backward, elementwise fp32 (precision="fp32" mode keeps u and g separately)
__global__ __launch_bounds__(256) void gelu_f32_kernel(const float* __restrict__ u, const float* __restrict__ dy,
                                                       float* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float x = u[i];
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    if (dy) out[i] = dy[i] * (cdf + x * 0.39894228040143267794f * expf(-0.5f * x * x));
    else out[i] = x * cdf;
  }
}
}  // namespace

extern "C" int simvg_attn_f32_fwd(const float* qkv, int ldqkv, float* out, int ldo, const unsigned char* pad, int B,
                                  int H, int Nv, int Nt, int D, float scale, hipStream_t stream) {
  SIMVG_CHECK_ARG(B > 0 && H > 0 && D == H * 64 && Nv + Nt > 0 && Nv + Nt <= 2048 && ldqkv % 4 == 0,
                  "attn_f32_fwd: need head_dim 64, N <= 2048");
  AF32Args a{qkv, ldqkv, out, ldo, pad, B, H, Nv, Nt, D, scale};
  const int N = Nv + Nt;
  const size_t shm = (size_t)(16 * 64 + 16 * N) * sizeof(float);
  static bool once = hipFuncSetAttribute((const void*)attn_f32_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         160 * 1024) == hipSuccess;
  (void)once;
  hipLaunchKernelGGL(attn_f32_fwd_kernel, dim3(B * H, cdiv(N, 16)), dim3(256), shm, stream, a);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}

extern "C" int simvg_attn_f32_bwd(const float* qkv, int ldqkv, const float* dout, int lddo, float* dqkv, int lddqkv,
                                  const unsigned char* pad, int B, int H, int Nv, int Nt, int D, float scale,
                                  hipStream_t stream) {
  SIMVG_CHECK_ARG(B > 0 && H > 0 && D == H * 64 && Nv + Nt > 0 && Nv + Nt <= 1024 && ldqkv % 4 == 0 && lddqkv % 4 == 0,
                  "attn_f32_bwd: need head_dim 64, N <= 1024");
  AF32BwdArgs a{qkv, ldqkv, dout, lddo, dqkv, lddqkv, pad, B, H, Nv, Nt, D, scale};
  const int N = Nv + Nt;
  const size_t shm = (size_t)(2 * 16 * 64 + 2 * 16 * N + 16) * sizeof(float);
  static bool once = hipFuncSetAttribute((const void*)attn_f32_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         160 * 1024) == hipSuccess;
  (void)once;
  hipLaunchKernelGGL(attn_f32_bwd_kernel, dim3(B * H, cdiv(N, 16)), dim3(256), shm, stream, a);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}

extern "C" int simvg_gelu_f32(const float* u, const float* dy_or_null, float* out, long n, hipStream_t stream) {
  SIMVG_CHECK_ARG(u && out && n > 0, "gelu_f32: empty");
  const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(gelu_f32_kernel, dim3(grid), dim3(256), 0, stream, u, dy_or_null, out, n);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}
