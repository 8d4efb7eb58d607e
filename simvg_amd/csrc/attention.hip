// Fused multi-head self-attention for the BEiT-3 encoder (gfx950), forward and backward.
//
// Replaces torchscale MultiheadAttention's  q*scale -> bmm(q,k^T) -> masked_fill(-inf) ->
// softmax(fp32) -> bmm(p,v) -> head merge  (reference call site beit3_base.py:137-145, SURVEY.md
// §2.3 E7-E12).  The score matrix never goes to HBM.
//
// Geometry of the path: N = 1 + (640/32)^2 + 20 = 421 tokens, head_dim 64.  One workgroup owns one
// (sample, head): the whole K and V (448 x 64 16-bit values each, 128-B swizzled rows) sit in LDS; each wave
// takes 16 query rows at a time, holds the complete S^T = K·Q^T strip (28 tiles of 16x16) in
// registers, does an exact (non-online) fp32 softmax with wavefront shuffles for the row reduce,
// and feeds P straight back as the MFMA B operand of O^T = V^T·P^T (V^T fragments come from
// ds_read_b64_tr_b16).  MFMA is used for the two contractions only.
//
// Token rows are laid out modality-major: vision rows of all samples first ([B*Nv, .]), then text
// rows ([B*Nt, .]); token t of sample b lives at row  t < Nv ? b*Nv + t : B*Nv + b*Nt + (t - Nv).
#include <stdlib.h>
#include <type_traits>

#include "attention.h"

namespace {

__device__ __forceinline__ void fill_key_bias(const AttnArgs& a, int b, int N, int npad, float* bias) {
  for (int k = threadIdx.x; k < npad; k += blockDim.x) {
    bool masked = k >= N;
    if (!masked && a.pad && k >= a.Nv) masked = a.pad[b * a.Nt + (k - a.Nv)] != 0;
    bias[k] = masked ? -INFINITY : 0.f;
  }
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(768) void attn_fwd_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int N = a.Nv + a.Nt;
  const int nkt = (N + 15) >> 4, ns2 = (nkt + 1) >> 1, npad = ns2 * 32;
  char* ldsK = smem;
  char* ldsV = smem + npad * ROWB;
  float* bias = (float*)(smem + 2 * npad * ROWB);
  const int b = blockIdx.x / a.H, h = blockIdx.x - b * a.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int j = lane & 15, g = lane >> 4;

  {
    const HeadSrc hs[2] = {{a.qkv, a.ld, a.D + h * HD, ldsK}, {a.qkv, a.ld, 2 * a.D + h * HD, ldsV}};
    load_heads_to_lds<5, 2>(a, hs, b, N, npad);
  }
  fill_key_bias(a, b, N, npad, bias);
  __syncthreads();

  const float sc2 = a.scale * 1.44269504088896340736f;
  // blockIdx.y: query strips dealt over gridDim.y workgroups of the same head (small B * H, see the host code)
  for (int qb = wave + nwaves * blockIdx.y; qb < nkt; qb += nwaves * gridDim.y) {
    const int tq = qb * 16 + j;
    const lp_t* qp = a.qkv + tok_row(a, b, tq < N ? tq : N - 1) * a.ld + h * HD + 8 * g;
    const lpx8_t q0 = *(const lpx8_t*)qp, q1 = *(const lpx8_t*)(qp + 32);
    f32x4_t s[MAX_KT];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < MAX_KT; ++kt) {
      if (kt < nkt) {
        f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        acc = mfma_lp(lds_frag(ldsK, kt * 16 + j, g), q0, acc);
        acc = mfma_lp(lds_frag(ldsK, kt * 16 + j, 4 + g), q1, acc);
        const f32x4_t kb = *(const f32x4_t*)(bias + kt * 16 + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          acc[r] = acc[r] * sc2 + kb[r];        // log2 domain: scale * log2(e) folded in; kb is 0 or -inf
          mx = fmaxf(mx, acc[r]);
        }
        s[kt] = acc;
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < MAX_KT; ++kt) {
      if (kt < nkt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __builtin_amdgcn_exp2f(s[kt][r] - mx);     // bare v_exp_f32
          s[kt][r] = p;
          sum += p;
        }
      }
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    f32x4_t o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s2 = 0; s2 < MAX_KT / 2; ++s2) {
      if (s2 < ns2) {
        float lo[4], hi[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          lo[r] = s[2 * s2][r];
          hi[r] = (2 * s2 + 1 < nkt) ? s[2 * s2 + 1][r] : 0.f;
        }
        const lpx8_t pf = pack8(lo, hi);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const lpx8_t vf = lds_frag_tr(ldsV, s2 * 32, s2 * 32 + 16, dt * 16, lane);
          o[dt] = mfma_lp(vf, pf, o[dt]);
        }
      }
    }
    if (tq < N) {
      const float inv = 1.f / sum;
      lp_t* op = a.out + tok_row(a, b, tq) * a.ldo + h * HD + 4 * g;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        *(u32x2_t*)(op + dt * 16) = (u32x2_t){pack_lp2(o[dt][0] * inv, o[dt][1] * inv),
                                             pack_lp2(o[dt][2] * inv, o[dt][3] * inv)};
      if (g == 0 && a.lse) a.lse[(long)blockIdx.x * N + tq] = (mx + __log2f(sum)) * 0.69314718055994530942f;   // natural log
    }
  }
}


// ------------------------------------------------------------------------------------------
// forward, compile-time geometry.  NKT key tiles of 16 (N in (16 NKT - 16, 16 NKT]); the first KFULL tiles hold vision
// keys only (never masked), so the -inf key bias is applied to the last NKT - KFULL tiles alone.  Differences to the
// generic kernel above, all aimed at the VALU / scalar work that bounded it (per 16-query strip: 128 v_med3 saturation
// clamps, 96 v_cndmask + ~70 scalar branches of the run-time tile guards, 125 SGPR spill moves, 112 row-sum adds):
//   * no run-time tile guards (every loop bound is a constant);
//   * exp2(s * c - max * c) is ONE fma + v_exp per score (the scale is folded into the subtraction, the running maximum is
//     taken over the raw accumulators);
//   * the row sum comes out of the MFMA pipe: one extra MFMA per 32 keys with an all-ones A operand (the sum of the SAME
//     16-bit-rounded probabilities the PV product uses), no VALU adds, no shuffles;
//   * probabilities are packed without the +-65504 clamp (they are in [0, 1]);
//   * PV MFMAs of key pair s2 are issued right after its 8 exponentials, so the matrix pipe works under the next pair's
//     transcendental VALU instead of after all of it.
// ------------------------------------------------------------------------------------------
// QK_ONLY (measurement: simvg_attn_qk_probe): the same K staging and QK^T contraction, then only the row maximum of the raw scores
// is stored -- no exponentials, no row sums, no PV -- so that the contraction north_star prices can be timed by itself.
// FWD_PROFILE (development build, tools/dev/attn_fwd_profile.py): workgroups 0 and 600 stamp s_memtime per wave into a.delta
#ifdef FWD_PROFILE
#define FWD_T(k_) do { if ((blockIdx.x == 0 || blockIdx.x == 600) && lane == 0 && a.delta) \
    ((unsigned*)a.delta)[((blockIdx.x ? 1 : 0) * 12 + wave) * 16 + (k_)] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define FWD_T(k_) do { } while (0)
#endif
#ifndef ATTN_FWD_G
#define ATTN_FWD_G 2        // key tiles per group of the QK^T loop (round 5 sweep, -DATTN_FWD_G=1 .. 4: isolated 65 - 68 / 70 - 73 / 70 - 72 / 71 - 74 us, but
                            // IN SITU G = 1 loses: 27.75 / 27.77 ms per step against 27.71 / 27.71 with G = 2, alternating runs; profiles/r05_sweeps.md section 12)
#endif
template <int NKT, int KFULL, int NTHREADS, int G = ATTN_FWD_G, bool QK_ONLY = false>
__global__ __launch_bounds__(NTHREADS) void attn_fwd_t_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NS2 = (NKT + 1) / 2, NPAD = NS2 * 32;
  const int N = a.Nv + a.Nt;
  char* ldsK = smem;
  char* ldsV = smem + NPAD * ROWB;
  float* bias = (float*)(smem + 2 * NPAD * ROWB);
  const int b = blockIdx.x / a.H, h = blockIdx.x - b * a.H;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = NTHREADS / 64;
  const int j = lane & 15, g = lane >> 4;

  // One round trip for the whole prologue (round 5): the key-padding byte, the first strip's Q fragments and all K / V chunks are
  // requested before anything is waited for; later strips' Q fragments are requested when the previous strip's contraction has
  // consumed the old ones, i.e. under its softmax / PV phase.
  static_assert(NPAD <= NTHREADS, "one key-bias entry per thread");
  FWD_T(0);
  const int qstep = nwaves * gridDim.y;
  int qb = wave + nwaves * blockIdx.y;
  // 32-bit element offsets from the kernel arguments' (scalar) bases: one VGPR per address instead of two held across the strip
  // loop (the launcher sends problems whose offsets do not fit to the generic kernel)
  auto row32 = [&](int t) { return t < a.Nv ? b * a.Nv + t : a.B * a.Nv + b * a.Nt + (t - a.Nv); };
  auto q_ptr = [&](int qb_) {
    const int tq_ = qb_ * 16 + j;
    return a.qkv + (unsigned)(row32(tq_ < N ? tq_ : N - 1) * a.ld + h * HD + 8 * g);   // beyond N: a valid row, never stored
  };
  lpx8_t q0 = *(const lpx8_t*)q_ptr(qb), q1 = *(const lpx8_t*)(q_ptr(qb) + 32);
  constexpr int UNR = (NPAD * 8 + NTHREADS - 1) / NTHREADS, NSRC = QK_ONLY ? 1 : 2;
  HeadSrc kvs[NSRC];
  kvs[0] = HeadSrc{a.qkv, a.ld, a.D + h * HD, ldsK};
  if constexpr (!QK_ONLY) kvs[1] = HeadSrc{a.qkv, a.ld, 2 * a.D + h * HD, ldsV};
  HeadChunks<UNR, NSRC> kvch;
  heads_issue<UNR, NSRC>(a, kvs, b, N, NPAD * 8, 0, kvch);
  const int kbk = threadIdx.x < NPAD ? (int)threadIdx.x : NPAD - 1;
  const int kbt = min(max(kbk - a.Nv, 0), max(a.Nt - 1, 0));
  const unsigned char padb = *(a.pad ? a.pad + b * a.Nt + kbt : (const unsigned char*)a.qkv);
  FWD_T(1);
  heads_commit<UNR, NSRC>(kvs, N, NPAD * 8, 0, kvch);
  if (threadIdx.x < NPAD)
    bias[threadIdx.x] = (kbk >= N || (a.pad && kbk >= a.Nv && padb != 0)) ? -INFINITY : 0.f;
  FWD_T(2);
  __syncthreads();
  FWD_T(3);
  [[maybe_unused]] int fwd_strip = 0;            // FWD_PROFILE: stamp index

  const float sc2 = a.scale * 1.44269504088896340736f;
  const unsigned int one2 = pack_lp2_raw(1.f, 1.f);
  union { lpx8_t v; unsigned int u[4]; } ones;
  ones.u[0] = ones.u[1] = ones.u[2] = ones.u[3] = one2;

  for (; qb < NKT; qb += qstep) {
    const int tq = qb * 16 + j;
    f32x4_t s[2 * NS2];
    float mx = -INFINITY;
    // K fragments one GROUP of G tiles ahead of their MFMAs (an LDS read takes ~100+ cycles to land, two MFMAs only ~35:
    // one tile ahead left every iteration waiting); within a group the first halves of all tiles are issued before the
    // second halves, so no MFMA follows the one that produces its accumulator.  The scheduling fences keep hipcc from
    // hoisting every LDS read of the unrolled loop to the top (it did: 200 spilled dwords).
    constexpr int NG = (NKT + G - 1) / G;
    lpx8_t kn[G][2];
#pragma unroll
    for (int t = 0; t < G; ++t) { kn[t][0] = lds_frag(ldsK, t * 16 + j, g); kn[t][1] = lds_frag(ldsK, t * 16 + j, 4 + g); }
#pragma unroll
    for (int grp = 0; grp < NG; ++grp) {
      lpx8_t kc[G][2];
#pragma unroll
      for (int t = 0; t < G; ++t) {
        kc[t][0] = kn[t][0]; kc[t][1] = kn[t][1];
        const int kt = (grp + 1) * G + t;
        if (kt < NKT) { kn[t][0] = lds_frag(ldsK, kt * 16 + j, g); kn[t][1] = lds_frag(ldsK, kt * 16 + j, 4 + g); }
      }
      f32x4_t acc[G];
#pragma unroll
      for (int t = 0; t < G; ++t)
        if (grp * G + t < NKT) acc[t] = mfma_lp(kc[t][0], q0, (f32x4_t){0.f, 0.f, 0.f, 0.f});
#pragma unroll
      for (int t = 0; t < G; ++t)
        if (grp * G + t < NKT) acc[t] = mfma_lp(kc[t][1], q1, acc[t]);
#pragma unroll
      for (int t = 0; t < G; ++t) {
        const int kt = grp * G + t;
        if (kt < NKT) {
          if (kt >= KFULL) {
            const f32x4_t kb = *(const f32x4_t*)(bias + kt * 16 + 4 * g);     // 0 or -inf
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t][r] += kb[r];
          }
          mx = fmaxf(fmaxf(mx, acc[t][0]), fmaxf(acc[t][1], fmaxf(acc[t][2], acc[t][3])));
          s[kt] = acc[t];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    {   // maximum over the four 16-lane rows on the VALU (v_permlane16_swap / v_permlane32_swap: no ds_bpermute round trips, no
        // index registers held across the strip loop).  Inline asm: hipcc folds fmaxf over the two results of the builtin forms
        // to the first one (round 5: the QK^T probe's row maxima came out per 16-lane row).  s_nop 1: the wait states hipcc puts
        // between a VALU write and a permlane swap of the same register.
      float ma = mx, mb = mx;
      asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(ma), "+v"(mb));   // {r0 r0 r2 r2}, {r1 r1 r3 r3}
      ma = fmaxf(ma, mb); mb = ma;
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(ma), "+v"(mb));   // {lo lo}, {hi hi}
      mx = fmaxf(ma, mb);
    }
    if constexpr (QK_ONLY) {
      if (tq < N && g == 0 && a.lse) a.lse[(long)blockIdx.x * N + tq] = mx * a.scale;
      const lp_t* qn = q_ptr(qb + qstep);
      q0 = *(const lpx8_t*)qn; q1 = *(const lpx8_t*)(qn + 32);
      continue;
    }
    constexpr int QPF = NS2 / 2;
    FWD_T(4 + 2 * fwd_strip);
    const float mxs = mx * sc2;          // sc2 > 0: max(s) * c == max(s * c)
    f32x4_t o[4], rs = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    lpx8_t vn[4];                                 // V^T fragments one key pair ahead of their MFMAs
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) vn[dt] = lds_frag_tr(ldsV, 0, 16, dt * 16, lane);
#pragma unroll
    for (int s2 = 0; s2 < NS2; ++s2) {
      lpx8_t vc[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        vc[dt] = vn[dt];
        if (s2 + 1 < NS2) vn[dt] = lds_frag_tr(ldsV, (s2 + 1) * 32, (s2 + 1) * 32 + 16, dt * 16, lane);
      }
      union { lpx8_t v; unsigned int u[4]; } pf;
      {
        const f32x4_t t = s[2 * s2];
        pf.u[0] = pack_lp2_raw(__builtin_amdgcn_exp2f(fmaf(t[0], sc2, -mxs)), __builtin_amdgcn_exp2f(fmaf(t[1], sc2, -mxs)));
        pf.u[1] = pack_lp2_raw(__builtin_amdgcn_exp2f(fmaf(t[2], sc2, -mxs)), __builtin_amdgcn_exp2f(fmaf(t[3], sc2, -mxs)));
      }
      if (2 * s2 + 1 < NKT) {
        const f32x4_t t = s[2 * s2 + 1];
        pf.u[2] = pack_lp2_raw(__builtin_amdgcn_exp2f(fmaf(t[0], sc2, -mxs)), __builtin_amdgcn_exp2f(fmaf(t[1], sc2, -mxs)));
        pf.u[3] = pack_lp2_raw(__builtin_amdgcn_exp2f(fmaf(t[2], sc2, -mxs)), __builtin_amdgcn_exp2f(fmaf(t[3], sc2, -mxs)));
      } else {
        pf.u[2] = 0u; pf.u[3] = 0u;
      }
      rs = mfma_lp(ones.v, pf.v, rs);            // every row of the result = sum over these 32 keys, per query column
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o[dt] = mfma_lp(vc[dt], pf.v, o[dt]);
      if (s2 == QPF) {                           // the next strip's Q rows, into registers the consumed score tiles have freed
        const lp_t* qn = q_ptr(qb + qstep);      // (past the last strip: a valid row, not used)
        q0 = *(const lpx8_t*)qn; q1 = *(const lpx8_t*)(qn + 32);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    {
      // a lane holds 4 consecutive hd values of query row j per 16-column tile; one v_permlane16_swap per dword between tiles dt and
      // dt + 1 leaves even-g lanes with 8 consecutive values of tile dt, odd-g lanes with 8 of tile dt + 1: two 16-byte stores
      // instead of four 8-byte ones (MI355X guide T21; the swaps run on every lane, the stores on valid rows)
      const float sum = rs[0];
      const float inv = 1.f / sum;
      unsigned ox[4][2];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        ox[dt][0] = pack_lp2(o[dt][0] * inv, o[dt][1] * inv);
        ox[dt][1] = pack_lp2(o[dt][2] * inv, o[dt][3] * inv);
      }
#pragma unroll
      for (int dt = 0; dt < 4; dt += 2)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const auto r = __builtin_amdgcn_permlane16_swap(ox[dt][hf], ox[dt + 1][hf], false, false);
          ox[dt][hf] = r[0]; ox[dt + 1][hf] = r[1];
        }
      if (tq < N) {
        lp_t* op = a.out + (unsigned)(row32(tq) * a.ldo + h * HD + (g >> 1) * 8 + (g & 1) * 16);
        *(u32x4_t*)op = (u32x4_t){ox[0][0], ox[0][1], ox[1][0], ox[1][1]};
        *(u32x4_t*)(op + 32) = (u32x4_t){ox[2][0], ox[2][1], ox[3][0], ox[3][1]};
        if (g == 0 && a.lse) a.lse[(unsigned)(blockIdx.x * N + tq)] = (mxs + __log2f(sum)) * 0.69314718055994530942f;   // natural log
      }
    }
    FWD_T(5 + 2 * fwd_strip);
    ++fwd_strip;
  }
  if constexpr (!QK_ONLY) {
    // waves 3 .. 7 finish a strip early (they take two, the workgroup's last waves are 6 - 12 k cycles behind them): they fetch the
    // next round's K / V lines into this XCD's L2 (attention.h: l2_touch_rows)
    const int nb = blockIdx.x + a.pf_stride;
    if (a.pf_stride > 0 && gridDim.y == 1 && wave >= 3 && wave < 8 && nb < gridDim.x) {
      const int b2 = nb / a.H, h2 = nb - b2 * a.H;
      l2_touch_rows(a, a.qkv, a.ld, a.D + h2 * HD, b2, N, N, (wave - 3) * 64 + lane, 320);
      l2_touch_rows(a, a.qkv, a.ld, 2 * a.D + h2 * HD, b2, N, N, (wave - 3) * 64 + lane, 320);
    }
  }
}

// ------------------------------------------------------------------------------------------
// backward, part 1: dQ (+ delta = rowsum(dO*O)).  K and V resident in LDS.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(768) void attn_bwd_dq_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int N = a.Nv + a.Nt;
  const int nkt = (N + 15) >> 4, ns2 = (nkt + 1) >> 1, npad = ns2 * 32;
  char* ldsK = smem;
  char* ldsV = smem + npad * ROWB;
  float* bias = (float*)(smem + 2 * npad * ROWB);
  const int b = blockIdx.x / a.H, h = blockIdx.x - b * a.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int j = lane & 15, g = lane >> 4;

  {
    const HeadSrc hs[2] = {{a.qkv, a.ld, a.D + h * HD, ldsK}, {a.qkv, a.ld, 2 * a.D + h * HD, ldsV}};
    load_heads_to_lds<5, 2>(a, hs, b, N, npad);
  }
  fill_key_bias(a, b, N, npad, bias);
  __syncthreads();

  const float sc2 = a.scale * 1.44269504088896340736f;
  for (int qb = wave; qb < nkt; qb += nwaves) {
    const int tq = qb * 16 + j;
    const long row = tok_row(a, b, tq < N ? tq : N - 1);
    const lp_t* qp = a.qkv + row * a.ld + h * HD + 8 * g;
    const lpx8_t q0 = *(const lpx8_t*)qp, q1 = *(const lpx8_t*)(qp + 32);
    const lp_t* dop = a.dout + row * a.lddo + h * HD + 8 * g;
    const lpx8_t d0 = *(const lpx8_t*)dop, d1 = *(const lpx8_t*)(dop + 32);
    const lp_t* op = a.out + row * a.ldo + h * HD + 8 * g;
    const lpx8_t o0 = *(const lpx8_t*)op, o1 = *(const lpx8_t*)(op + 32);
    float dl = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      dl += lp_to_f32((lp_t)d0[e]) * lp_to_f32((lp_t)o0[e]);
      dl += lp_to_f32((lp_t)d1[e]) * lp_to_f32((lp_t)o1[e]);
    }
    dl += __shfl_xor(dl, 16, 64);
    dl += __shfl_xor(dl, 32, 64);
    const float lse2 = a.lse[(long)blockIdx.x * N + (tq < N ? tq : N - 1)] * 1.44269504088896340736f;
    if (tq < N && g == 0) a.delta[(long)blockIdx.x * N + tq] = dl;

    u32x2_t dsb[MAX_KT];
#pragma unroll
    for (int kt = 0; kt < MAX_KT; ++kt) {
      if (kt < nkt) {
        f32x4_t sa = (f32x4_t){0.f, 0.f, 0.f, 0.f}, dp = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        sa = mfma_lp(lds_frag(ldsK, kt * 16 + j, g), q0, sa);
        sa = mfma_lp(lds_frag(ldsK, kt * 16 + j, 4 + g), q1, sa);
        dp = mfma_lp(lds_frag(ldsV, kt * 16 + j, g), d0, dp);
        dp = mfma_lp(lds_frag(ldsV, kt * 16 + j, 4 + g), d1, dp);
        const f32x4_t kb = *(const f32x4_t*)(bias + kt * 16 + 4 * g);
        float ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __builtin_amdgcn_exp2f(sa[r] * sc2 + kb[r] - lse2);
          ds[r] = p * (dp[r] - dl);
        }
        dsb[kt] = (u32x2_t){pack_lp2(ds[0], ds[1]), pack_lp2(ds[2], ds[3])};
      }
    }
    f32x4_t dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s2 = 0; s2 < MAX_KT / 2; ++s2) {
      if (s2 < ns2) {
        union { lpx8_t v; unsigned int u[4]; } pf;
        pf.u[0] = dsb[2 * s2][0]; pf.u[1] = dsb[2 * s2][1];
        if (2 * s2 + 1 < nkt) { pf.u[2] = dsb[2 * s2 + 1][0]; pf.u[3] = dsb[2 * s2 + 1][1]; }
        else { pf.u[2] = 0u; pf.u[3] = 0u; }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const lpx8_t kf = lds_frag_tr(ldsK, s2 * 32, s2 * 32 + 16, dt * 16, lane);
          dq[dt] = mfma_lp(kf, pf.v, dq[dt]);
        }
      }
    }
    if (tq < N) {
      lp_t* gp = a.dqkv + tok_row(a, b, tq) * a.lddq + h * HD + 4 * g;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        *(u32x2_t*)(gp + dt * 16) = (u32x2_t){pack_lp2(dq[dt][0] * a.scale, dq[dt][1] * a.scale),
                                             pack_lp2(dq[dt][2] * a.scale, dq[dt][3] * a.scale)};
    }
  }
}


// backward part 1 with compile-time geometry (same idea as attn_fwd_t_kernel: constant loop bounds, K / V fragments one
// tile ahead of their MFMAs, the key bias only on the tiles that can hold a masked key, exp2 argument as one fma)
template <int NKT, int KFULL, int NTHREADS>
__global__ __launch_bounds__(NTHREADS) void attn_bwd_dq_t_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NS2 = (NKT + 1) / 2, NPAD = NS2 * 32;
  const int N = a.Nv + a.Nt;
  char* ldsK = smem;
  char* ldsV = smem + NPAD * ROWB;
  float* bias = (float*)(smem + 2 * NPAD * ROWB);
  const int b = blockIdx.x / a.H, h = blockIdx.x - b * a.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int j = lane & 15, g = lane >> 4;

  {
    const HeadSrc hs[2] = {{a.qkv, a.ld, a.D + h * HD, ldsK}, {a.qkv, a.ld, 2 * a.D + h * HD, ldsV}};
    load_heads_to_lds<(NPAD * 8 + NTHREADS - 1) / NTHREADS, 2>(a, hs, b, N, NPAD);
  }
  fill_key_bias(a, b, N, NPAD, bias);
  __syncthreads();

  const float sc2 = a.scale * 1.44269504088896340736f;
  for (int qb = wave; qb < NKT; qb += nwaves) {
    const int tq = qb * 16 + j;
    const long row = tok_row(a, b, tq < N ? tq : N - 1);
    const lp_t* qp = a.qkv + row * a.ld + h * HD + 8 * g;
    const lpx8_t q0 = *(const lpx8_t*)qp, q1 = *(const lpx8_t*)(qp + 32);
    const lp_t* dop = a.dout + row * a.lddo + h * HD + 8 * g;
    const lpx8_t d0 = *(const lpx8_t*)dop, d1 = *(const lpx8_t*)(dop + 32);
    const lp_t* op = a.out + row * a.ldo + h * HD + 8 * g;
    const lpx8_t o0 = *(const lpx8_t*)op, o1 = *(const lpx8_t*)(op + 32);
    float dl = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      dl += lp_to_f32((lp_t)d0[e]) * lp_to_f32((lp_t)o0[e]);
      dl += lp_to_f32((lp_t)d1[e]) * lp_to_f32((lp_t)o1[e]);
    }
    dl += __shfl_xor(dl, 16, 64);
    dl += __shfl_xor(dl, 32, 64);
    const float nlse2 = -a.lse[(long)blockIdx.x * N + (tq < N ? tq : N - 1)] * 1.44269504088896340736f;
    if (tq < N && g == 0) a.delta[(long)blockIdx.x * N + tq] = dl;

    u32x2_t dsb[2 * NS2];
    lpx8_t ka = lds_frag(ldsK, j, g), kb2 = lds_frag(ldsK, j, 4 + g), va = lds_frag(ldsV, j, g), vb = lds_frag(ldsV, j, 4 + g);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const lpx8_t cka = ka, ckb = kb2, cva = va, cvb = vb;
      if (kt + 1 < NKT) {
        ka = lds_frag(ldsK, (kt + 1) * 16 + j, g);
        kb2 = lds_frag(ldsK, (kt + 1) * 16 + j, 4 + g);
        va = lds_frag(ldsV, (kt + 1) * 16 + j, g);
        vb = lds_frag(ldsV, (kt + 1) * 16 + j, 4 + g);
      }
      f32x4_t sa = (f32x4_t){0.f, 0.f, 0.f, 0.f}, dp = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      sa = mfma_lp(cka, q0, sa);
      dp = mfma_lp(cva, d0, dp);
      sa = mfma_lp(ckb, q1, sa);
      dp = mfma_lp(cvb, d1, dp);
      float ds[4], pu[4];
      const f32x2_t c2 = {sc2, sc2}, dl2 = {dl, dl};
      if (kt >= KFULL) {
        const f32x4_t kb = *(const f32x4_t*)(bias + kt * 16 + 4 * g);
        ds_pair(sa[0], sa[1], c2, (f32x2_t){kb[0] + nlse2, kb[1] + nlse2}, dp[0], dp[1], dl2, pu[0], pu[1], ds[0], ds[1]);
        ds_pair(sa[2], sa[3], c2, (f32x2_t){kb[2] + nlse2, kb[3] + nlse2}, dp[2], dp[3], dl2, pu[2], pu[3], ds[2], ds[3]);
      } else {
        const f32x2_t nl2 = {nlse2, nlse2};
        ds_pair(sa[0], sa[1], c2, nl2, dp[0], dp[1], dl2, pu[0], pu[1], ds[0], ds[1]);
        ds_pair(sa[2], sa[3], c2, nl2, dp[2], dp[3], dl2, pu[2], pu[3], ds[2], ds[3]);
      }
      dsb[kt] = (u32x2_t){pack_lp2(ds[0], ds[1]), pack_lp2(ds[2], ds[3])};
      __builtin_amdgcn_sched_barrier(0);
    }
    if (NKT & 1) dsb[NKT] = (u32x2_t){0u, 0u};
    f32x4_t dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    lpx8_t kn[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) kn[dt] = lds_frag_tr(ldsK, 0, 16, dt * 16, lane);
#pragma unroll
    for (int s2 = 0; s2 < NS2; ++s2) {
      lpx8_t kc[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        kc[dt] = kn[dt];
        if (s2 + 1 < NS2) kn[dt] = lds_frag_tr(ldsK, (s2 + 1) * 32, (s2 + 1) * 32 + 16, dt * 16, lane);
      }
      union { lpx8_t v; unsigned int u[4]; } pf;
      pf.u[0] = dsb[2 * s2][0]; pf.u[1] = dsb[2 * s2][1];
      pf.u[2] = dsb[2 * s2 + 1][0]; pf.u[3] = dsb[2 * s2 + 1][1];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) dq[dt] = mfma_lp(kc[dt], pf.v, dq[dt]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (tq < N) {
      lp_t* gp = a.dqkv + tok_row(a, b, tq) * a.lddq + h * HD + 4 * g;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        *(u32x2_t*)(gp + dt * 16) = (u32x2_t){pack_lp2(dq[dt][0] * a.scale, dq[dt][1] * a.scale),
                                             pack_lp2(dq[dt][2] * a.scale, dq[dt][3] * a.scale)};
    }
  }
}

// ------------------------------------------------------------------------------------------
// backward, part 2: dK, dV.  Q and dO resident in LDS; each wave owns 16 keys at a time.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(768) void attn_bwd_dkv_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int N = a.Nv + a.Nt;
  const int nkt = (N + 15) >> 4, ns2 = (nkt + 1) >> 1, npad = ns2 * 32;
  char* ldsQ = smem;
  char* ldsDO = smem + npad * ROWB;
  float* lse_s = (float*)(smem + 2 * npad * ROWB);
  float* dl_s = lse_s + npad;
  const int b = blockIdx.x / a.H, h = blockIdx.x - b * a.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int j = lane & 15, g = lane >> 4;

  {
    const HeadSrc hs[2] = {{a.qkv, a.ld, h * HD, ldsQ}, {a.dout, a.lddo, h * HD, ldsDO}};
    load_heads_to_lds<5, 2>(a, hs, b, N, npad);
  }
  for (int q = threadIdx.x; q < npad; q += blockDim.x) {
    lse_s[q] = q < N ? a.lse[(long)blockIdx.x * N + q] * 1.44269504088896340736f : INFINITY;   // log2 domain; exp2(.. - inf) = 0 for pad rows
    dl_s[q] = q < N ? a.delta[(long)blockIdx.x * N + q] : 0.f;
  }
  __syncthreads();

  const float sc2 = a.scale * 1.44269504088896340736f;
  for (int kb = wave; kb < nkt; kb += nwaves) {
    const int tk = kb * 16 + j;
    const long row = tok_row(a, b, tk < N ? tk : N - 1);
    const lp_t* kp = a.qkv + row * a.ld + a.D + h * HD + 8 * g;
    const lpx8_t k0 = *(const lpx8_t*)kp, k1 = *(const lpx8_t*)(kp + 32);
    const lp_t* vp = a.qkv + row * a.ld + 2 * a.D + h * HD + 8 * g;
    const lpx8_t v0 = *(const lpx8_t*)vp, v1 = *(const lpx8_t*)(vp + 32);
    bool masked = tk >= N;
    if (!masked && a.pad && tk >= a.Nv) masked = a.pad[b * a.Nt + (tk - a.Nv)] != 0;
    const float kbias = masked ? -INFINITY : 0.f;

    f32x4_t dk[4], dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dk[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dv[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }

    for (int s2 = 0; s2 < ns2; ++s2) {
      float p[2][4], ds[2][4];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int qt = 2 * s2 + hh;   // rows qt*16.. exist in LDS (zero-filled beyond N)
        f32x4_t sa = (f32x4_t){0.f, 0.f, 0.f, 0.f}, dp = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        sa = mfma_lp(lds_frag(ldsQ, qt * 16 + j, g), k0, sa);
        sa = mfma_lp(lds_frag(ldsQ, qt * 16 + j, 4 + g), k1, sa);
        dp = mfma_lp(lds_frag(ldsDO, qt * 16 + j, g), v0, dp);
        dp = mfma_lp(lds_frag(ldsDO, qt * 16 + j, 4 + g), v1, dp);
        const f32x4_t l4 = *(const f32x4_t*)(lse_s + qt * 16 + 4 * g);
        const f32x4_t d4 = *(const f32x4_t*)(dl_s + qt * 16 + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pr = __builtin_amdgcn_exp2f(sa[r] * sc2 + kbias - l4[r]);
          p[hh][r] = pr;
          ds[hh][r] = pr * (dp[r] - d4[r]);
        }
      }
      const lpx8_t pf = pack8(p[0], p[1]);
      const lpx8_t dsf = pack8(ds[0], ds[1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const lpx8_t dof = lds_frag_tr(ldsDO, s2 * 32, s2 * 32 + 16, dt * 16, lane);
        dv[dt] = mfma_lp(dof, pf, dv[dt]);
        const lpx8_t qf = lds_frag_tr(ldsQ, s2 * 32, s2 * 32 + 16, dt * 16, lane);
        dk[dt] = mfma_lp(qf, dsf, dk[dt]);
      }
    }
    if (tk < N) {
      lp_t* gk = a.dqkv + tok_row(a, b, tk) * a.lddq + a.D + h * HD + 4 * g;
      lp_t* gv = a.dqkv + tok_row(a, b, tk) * a.lddq + 2 * a.D + h * HD + 4 * g;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        *(u32x2_t*)(gk + dt * 16) = (u32x2_t){pack_lp2(dk[dt][0] * a.scale, dk[dt][1] * a.scale),
                                             pack_lp2(dk[dt][2] * a.scale, dk[dt][3] * a.scale)};
        *(u32x2_t*)(gv + dt * 16) = (u32x2_t){pack_lp2(dv[dt][0], dv[dt][1]), pack_lp2(dv[dt][2], dv[dt][3])};
      }
    }
  }
}


// backward part 2 with compile-time geometry.  NQT query tiles (== key tiles); key strips kb < KFULL hold vision keys only.
template <int NQT, bool MASKED, bool AHEAD>
__device__ __forceinline__ void dkv_strip(const char* ldsQ, const char* ldsDO, const float* nlse_s, const float* dl_s,
                                          const lpx8_t k0, const lpx8_t k1, const lpx8_t v0, const lpx8_t v1, float kbias,
                                          float sc2, int lane, f32x4_t (&dk)[4], f32x4_t (&dv)[4]) {
  constexpr int NS2 = (NQT + 1) / 2;
  const int j = lane & 15, g = lane >> 4;
  lpx8_t qn[2][2], dn[2][2];                    // AHEAD: K-major Q / dO fragments of the NEXT query pair (32 registers)
  if (AHEAD) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      qn[hh][0] = lds_frag(ldsQ, hh * 16 + j, g);  qn[hh][1] = lds_frag(ldsQ, hh * 16 + j, 4 + g);
      dn[hh][0] = lds_frag(ldsDO, hh * 16 + j, g); dn[hh][1] = lds_frag(ldsDO, hh * 16 + j, 4 + g);
    }
  }
#pragma unroll
  for (int s2 = 0; s2 < NS2; ++s2) {
    lpx8_t qc[2][2], dc[2][2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      if (AHEAD) {
        qc[hh][0] = qn[hh][0]; qc[hh][1] = qn[hh][1]; dc[hh][0] = dn[hh][0]; dc[hh][1] = dn[hh][1];
      } else {
        const int r0 = (2 * s2 + hh) * 16 + j;
        qc[hh][0] = lds_frag(ldsQ, r0, g);  qc[hh][1] = lds_frag(ldsQ, r0, 4 + g);
        dc[hh][0] = lds_frag(ldsDO, r0, g); dc[hh][1] = lds_frag(ldsDO, r0, 4 + g);
      }
    }
    // transposed fragments of THIS pair (needed after the exponentials) and the K-major ones of the next pair
    lpx8_t dof[4], qf[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      dof[dt] = lds_frag_tr(ldsDO, s2 * 32, s2 * 32 + 16, dt * 16, lane);
      qf[dt] = lds_frag_tr(ldsQ, s2 * 32, s2 * 32 + 16, dt * 16, lane);
    }
    if (AHEAD && s2 + 1 < NS2) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int r0 = (2 * s2 + 2 + hh) * 16 + j;
        qn[hh][0] = lds_frag(ldsQ, r0, g);  qn[hh][1] = lds_frag(ldsQ, r0, 4 + g);
        dn[hh][0] = lds_frag(ldsDO, r0, g); dn[hh][1] = lds_frag(ldsDO, r0, 4 + g);
      }
    }
    float p[2][4], ds[2][4];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int qt = 2 * s2 + hh;   // rows qt*16.. exist in LDS (zero-filled beyond N)
      f32x4_t sa = (f32x4_t){0.f, 0.f, 0.f, 0.f}, dp = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      sa = mfma_lp(qc[hh][0], k0, sa);
      dp = mfma_lp(dc[hh][0], v0, dp);
      sa = mfma_lp(qc[hh][1], k1, sa);
      dp = mfma_lp(dc[hh][1], v1, dp);
      const f32x4_t l4 = *(const f32x4_t*)(nlse_s + qt * 16 + 4 * g);
      const f32x4_t d4 = *(const f32x4_t*)(dl_s + qt * 16 + 4 * g);
      const f32x2_t c2 = {sc2, sc2};
      ds_pair(sa[0], sa[1], c2, (f32x2_t){MASKED ? l4[0] + kbias : l4[0], MASKED ? l4[1] + kbias : l4[1]}, dp[0], dp[1],
              (f32x2_t){d4[0], d4[1]}, p[hh][0], p[hh][1], ds[hh][0], ds[hh][1]);
      ds_pair(sa[2], sa[3], c2, (f32x2_t){MASKED ? l4[2] + kbias : l4[2], MASKED ? l4[3] + kbias : l4[3]}, dp[2], dp[3],
              (f32x2_t){d4[2], d4[3]}, p[hh][2], p[hh][3], ds[hh][2], ds[hh][3]);
    }
    union { lpx8_t v; unsigned int u[4]; } pf;
    pf.u[0] = pack_lp2_raw(p[0][0], p[0][1]); pf.u[1] = pack_lp2_raw(p[0][2], p[0][3]);
    pf.u[2] = pack_lp2_raw(p[1][0], p[1][1]); pf.u[3] = pack_lp2_raw(p[1][2], p[1][3]);
    const lpx8_t dsf = pack8(ds[0], ds[1]);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      dv[dt] = mfma_lp(dof[dt], pf.v, dv[dt]);
      dk[dt] = mfma_lp(qf[dt], dsf, dk[dt]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int NQT, int KFULL, int NTHREADS>
__global__ __launch_bounds__(NTHREADS) void attn_bwd_dkv_t_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NS2 = (NQT + 1) / 2, NPAD = NS2 * 32;
  const int N = a.Nv + a.Nt;
  char* ldsQ = smem;
  char* ldsDO = smem + NPAD * ROWB;
  float* nlse_s = (float*)(smem + 2 * NPAD * ROWB);
  float* dl_s = nlse_s + NPAD;
  const int b = blockIdx.x / a.H, h = blockIdx.x - b * a.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int j = lane & 15, g = lane >> 4;

  {
    const HeadSrc hs[2] = {{a.qkv, a.ld, h * HD, ldsQ}, {a.dout, a.lddo, h * HD, ldsDO}};
    load_heads_to_lds<(NPAD * 8 + NTHREADS - 1) / NTHREADS, 2>(a, hs, b, N, NPAD);
  }
  for (int q = threadIdx.x; q < NPAD; q += blockDim.x) {
    nlse_s[q] = q < N ? -a.lse[(long)blockIdx.x * N + q] * 1.44269504088896340736f : -INFINITY;   // log2 domain, negated; exp2(.. - inf) = 0 for pad rows
    dl_s[q] = q < N ? a.delta[(long)blockIdx.x * N + q] : 0.f;
  }
  __syncthreads();

  const float sc2 = a.scale * 1.44269504088896340736f;
  for (int kb = wave; kb < NQT; kb += nwaves) {
    const int tk = kb * 16 + j;
    const long row = tok_row(a, b, tk < N ? tk : N - 1);
    const lp_t* kp = a.qkv + row * a.ld + a.D + h * HD + 8 * g;
    const lpx8_t k0 = *(const lpx8_t*)kp, k1 = *(const lpx8_t*)(kp + 32);
    const lp_t* vp = a.qkv + row * a.ld + 2 * a.D + h * HD + 8 * g;
    const lpx8_t v0 = *(const lpx8_t*)vp, v1 = *(const lpx8_t*)(vp + 32);
    f32x4_t dk[4], dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dk[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dv[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
    if (kb >= KFULL) {
      bool masked = tk >= N;
      if (!masked && a.pad && tk >= a.Nv) masked = a.pad[b * a.Nt + (tk - a.Nv)] != 0;
      dkv_strip<NQT, true, (NTHREADS <= 512)>(ldsQ, ldsDO, nlse_s, dl_s, k0, k1, v0, v1, masked ? -INFINITY : 0.f, sc2, lane, dk, dv);
    } else {
      dkv_strip<NQT, false, (NTHREADS <= 512)>(ldsQ, ldsDO, nlse_s, dl_s, k0, k1, v0, v1, 0.f, sc2, lane, dk, dv);
    }
    if (tk < N) {
      lp_t* gk = a.dqkv + tok_row(a, b, tk) * a.lddq + a.D + h * HD + 4 * g;
      lp_t* gv = a.dqkv + tok_row(a, b, tk) * a.lddq + 2 * a.D + h * HD + 4 * g;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        *(u32x2_t*)(gk + dt * 16) = (u32x2_t){pack_lp2(dk[dt][0] * a.scale, dk[dt][1] * a.scale),
                                             pack_lp2(dk[dt][2] * a.scale, dk[dt][3] * a.scale)};
        *(u32x2_t*)(gv + dt * 16) = (u32x2_t){pack_lp2(dv[dt][0], dv[dt][1]), pack_lp2(dv[dt][2], dv[dt][3])};
      }
    }
  }
}


// ------------------------------------------------------------------------------------------
// Any token count (N > 448: patch 16, images above 640): K / V (forward, dQ) or Q / dO (dK, dV) are streamed through the
// LDS in blocks of TB rows, the forward with an online softmax over the key blocks (running maximum and sum per query,
// accumulator rescaled when the maximum moves), the backward from the saved LSE.  One workgroup of TW waves per
// (sample, head, 16 TW queries or keys).  Same LDS row layout, fragment reads and masking as the resident kernels above;
// these are the general path, not the tuned one.
// ------------------------------------------------------------------------------------------
constexpr int TB = 256;        // rows per LDS block (2 operands x 256 x 128 B = 64 KiB: two workgroups per CU)
constexpr int TW = 8;          // waves per workgroup: 128 queries (or keys) per workgroup

// rows [r0, r0 + TB) of one head -> LDS block rows [0, TB) (zero beyond N)
__device__ __forceinline__ void load_block_to_lds(const AttnArgs& a, const lp_t* base, int ld, int col0, int b, int N, int r0,
                                                  char* lds) {
  for (int c = threadIdx.x; c < TB * 8; c += blockDim.x) {
    const int row = c >> 3, slot = c & 7;
    u32x4_t v = (u32x4_t){0u, 0u, 0u, 0u};
    if (r0 + row < N) v = *(const u32x4_t*)(base + tok_row(a, b, r0 + row) * ld + col0 + slot * 8);
    *(u32x4_t*)(lds + row * ROWB + lds_slot(row, slot) * 16) = v;
  }
}

__device__ __forceinline__ void fill_key_bias_block(const AttnArgs& a, int b, int N, int r0, float* bias) {
  for (int k = threadIdx.x; k < TB; k += blockDim.x) {
    const int t = r0 + k;
    bool masked = t >= N;
    if (!masked && a.pad && t >= a.Nv) masked = a.pad[b * a.Nt + (t - a.Nv)] != 0;
    bias[k] = masked ? -INFINITY : 0.f;
  }
}

constexpr int TILED_LDS = 2 * TB * ROWB + 2 * TB * (int)sizeof(float);

__global__ __launch_bounds__(TW * 64) void attn_fwd_tiled_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ldsK = smem;
  char* ldsV = smem + TB * ROWB;
  float* bias = (float*)(smem + 2 * TB * ROWB);
  const int N = a.Nv + a.Nt;
  const int b = blockIdx.x / a.H, h = blockIdx.x - b * a.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int tq = (blockIdx.y * TW + wave) * 16 + j;
  const lp_t* qp = a.qkv + tok_row(a, b, tq < N ? tq : N - 1) * a.ld + h * HD + 8 * g;
  const lpx8_t q0 = *(const lpx8_t*)qp, q1 = *(const lpx8_t*)(qp + 32);
  const float sc2 = a.scale * 1.44269504088896340736f;
  float m = -INFINITY, l = 0.f;          // running maximum (raw score units) and sum of this lane's query
  f32x4_t o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  for (int r0 = 0; r0 < N; r0 += TB) {
    __syncthreads();                     // everyone is done with the previous block
    load_block_to_lds(a, a.qkv, a.ld, a.D + h * HD, b, N, r0, ldsK);
    load_block_to_lds(a, a.qkv, a.ld, 2 * a.D + h * HD, b, N, r0, ldsV);
    fill_key_bias_block(a, b, N, r0, bias);
    __syncthreads();
    f32x4_t s[TB / 16];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < TB / 16; ++kt) {
      f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      acc = mfma_lp(lds_frag(ldsK, kt * 16 + j, g), q0, acc);
      acc = mfma_lp(lds_frag(ldsK, kt * 16 + j, 4 + g), q1, acc);
      const f32x4_t kb = *(const f32x4_t*)(bias + kt * 16 + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) { acc[r] += kb[r]; mx = fmaxf(mx, acc[r]); }
      s[kt] = acc;
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m, mx);
    const float m_use = m_new == -INFINITY ? 0.f : m_new;          // a fully masked prefix: nothing to rescale, p = 0
    const float alpha = __builtin_amdgcn_exp2f((m - m_use) * sc2);  // exp2(-inf) = 0 on the first block
    m = m_new;
    l *= alpha;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[dt][r] *= alpha;
    const float mxs = m_use * sc2;
    float sum = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < TB / 32; ++s2) {
      float lo[4], hi[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        lo[r] = __builtin_amdgcn_exp2f(fmaf(s[2 * s2][r], sc2, -mxs));
        hi[r] = __builtin_amdgcn_exp2f(fmaf(s[2 * s2 + 1][r], sc2, -mxs));
        sum += lo[r] + hi[r];
      }
      const lpx8_t pf = pack8(lo, hi);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o[dt] = mfma_lp(lds_frag_tr(ldsV, s2 * 32, s2 * 32 + 16, dt * 16, lane), pf, o[dt]);
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    l += sum;
  }
  if (tq < N) {
    const float inv = 1.f / l;
    lp_t* op = a.out + tok_row(a, b, tq) * a.ldo + h * HD + 4 * g;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      *(u32x2_t*)(op + dt * 16) = (u32x2_t){pack_lp2(o[dt][0] * inv, o[dt][1] * inv), pack_lp2(o[dt][2] * inv, o[dt][3] * inv)};
    if (g == 0 && a.lse) a.lse[(long)blockIdx.x * N + tq] = (m * sc2 + __log2f(l)) * 0.69314718055994530942f;
  }
}

__global__ __launch_bounds__(TW * 64) void attn_bwd_dq_tiled_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ldsK = smem;
  char* ldsV = smem + TB * ROWB;
  float* bias = (float*)(smem + 2 * TB * ROWB);
  const int N = a.Nv + a.Nt;
  const int b = blockIdx.x / a.H, h = blockIdx.x - b * a.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int tq = (blockIdx.y * TW + wave) * 16 + j;
  const long row = tok_row(a, b, tq < N ? tq : N - 1);
  const lp_t* qp = a.qkv + row * a.ld + h * HD + 8 * g;
  const lpx8_t q0 = *(const lpx8_t*)qp, q1 = *(const lpx8_t*)(qp + 32);
  const lp_t* dop = a.dout + row * a.lddo + h * HD + 8 * g;
  const lpx8_t d0 = *(const lpx8_t*)dop, d1 = *(const lpx8_t*)(dop + 32);
  const lp_t* op = a.out + row * a.ldo + h * HD + 8 * g;
  const lpx8_t o0 = *(const lpx8_t*)op, o1 = *(const lpx8_t*)(op + 32);
  float dl = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    dl += lp_to_f32((lp_t)d0[e]) * lp_to_f32((lp_t)o0[e]);
    dl += lp_to_f32((lp_t)d1[e]) * lp_to_f32((lp_t)o1[e]);
  }
  dl += __shfl_xor(dl, 16, 64);
  dl += __shfl_xor(dl, 32, 64);
  const float nlse2 = -a.lse[(long)blockIdx.x * N + (tq < N ? tq : N - 1)] * 1.44269504088896340736f;
  if (tq < N && g == 0) a.delta[(long)blockIdx.x * N + tq] = dl;
  const float sc2 = a.scale * 1.44269504088896340736f;
  f32x4_t dq[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) dq[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  for (int r0 = 0; r0 < N; r0 += TB) {
    __syncthreads();
    load_block_to_lds(a, a.qkv, a.ld, a.D + h * HD, b, N, r0, ldsK);
    load_block_to_lds(a, a.qkv, a.ld, 2 * a.D + h * HD, b, N, r0, ldsV);
    fill_key_bias_block(a, b, N, r0, bias);
    __syncthreads();
#pragma unroll
    for (int s2 = 0; s2 < TB / 32; ++s2) {
      float ds[2][4];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int kt = 2 * s2 + hh;
        f32x4_t sa = (f32x4_t){0.f, 0.f, 0.f, 0.f}, dp = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        sa = mfma_lp(lds_frag(ldsK, kt * 16 + j, g), q0, sa);
        dp = mfma_lp(lds_frag(ldsV, kt * 16 + j, g), d0, dp);
        sa = mfma_lp(lds_frag(ldsK, kt * 16 + j, 4 + g), q1, sa);
        dp = mfma_lp(lds_frag(ldsV, kt * 16 + j, 4 + g), d1, dp);
        const f32x4_t kb = *(const f32x4_t*)(bias + kt * 16 + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) ds[hh][r] = __builtin_amdgcn_exp2f(fmaf(sa[r], sc2, kb[r] + nlse2)) * (dp[r] - dl);
      }
      const lpx8_t dsf = pack8(ds[0], ds[1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) dq[dt] = mfma_lp(lds_frag_tr(ldsK, s2 * 32, s2 * 32 + 16, dt * 16, lane), dsf, dq[dt]);
    }
  }
  if (tq < N) {
    lp_t* gp = a.dqkv + tok_row(a, b, tq) * a.lddq + h * HD + 4 * g;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      *(u32x2_t*)(gp + dt * 16) = (u32x2_t){pack_lp2(dq[dt][0] * a.scale, dq[dt][1] * a.scale),
                                           pack_lp2(dq[dt][2] * a.scale, dq[dt][3] * a.scale)};
  }
}

__global__ __launch_bounds__(TW * 64) void attn_bwd_dkv_tiled_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ldsQ = smem;
  char* ldsDO = smem + TB * ROWB;
  float* nlse_s = (float*)(smem + 2 * TB * ROWB);
  float* dl_s = nlse_s + TB;
  const int N = a.Nv + a.Nt;
  const int b = blockIdx.x / a.H, h = blockIdx.x - b * a.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int tk = (blockIdx.y * TW + wave) * 16 + j;
  const long row = tok_row(a, b, tk < N ? tk : N - 1);
  const lp_t* kp = a.qkv + row * a.ld + a.D + h * HD + 8 * g;
  const lpx8_t k0 = *(const lpx8_t*)kp, k1 = *(const lpx8_t*)(kp + 32);
  const lp_t* vp = a.qkv + row * a.ld + 2 * a.D + h * HD + 8 * g;
  const lpx8_t v0 = *(const lpx8_t*)vp, v1 = *(const lpx8_t*)(vp + 32);
  bool masked = tk >= N;
  if (!masked && a.pad && tk >= a.Nv) masked = a.pad[b * a.Nt + (tk - a.Nv)] != 0;
  const float kbias = masked ? -INFINITY : 0.f;
  const float sc2 = a.scale * 1.44269504088896340736f;
  f32x4_t dk[4], dv[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) { dk[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dv[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }

  for (int r0 = 0; r0 < N; r0 += TB) {
    __syncthreads();
    load_block_to_lds(a, a.qkv, a.ld, h * HD, b, N, r0, ldsQ);
    load_block_to_lds(a, a.dout, a.lddo, h * HD, b, N, r0, ldsDO);
    for (int q = threadIdx.x; q < TB; q += blockDim.x) {
      const int t = r0 + q;
      nlse_s[q] = t < N ? -a.lse[(long)blockIdx.x * N + t] * 1.44269504088896340736f : -INFINITY;   // pad rows: p = 0
      dl_s[q] = t < N ? a.delta[(long)blockIdx.x * N + t] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int s2 = 0; s2 < TB / 32; ++s2) {
      float p[2][4], ds[2][4];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int qt = 2 * s2 + hh;
        f32x4_t sa = (f32x4_t){0.f, 0.f, 0.f, 0.f}, dp = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        sa = mfma_lp(lds_frag(ldsQ, qt * 16 + j, g), k0, sa);
        dp = mfma_lp(lds_frag(ldsDO, qt * 16 + j, g), v0, dp);
        sa = mfma_lp(lds_frag(ldsQ, qt * 16 + j, 4 + g), k1, sa);
        dp = mfma_lp(lds_frag(ldsDO, qt * 16 + j, 4 + g), v1, dp);
        const f32x4_t l4 = *(const f32x4_t*)(nlse_s + qt * 16 + 4 * g);
        const f32x4_t d4 = *(const f32x4_t*)(dl_s + qt * 16 + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pr = __builtin_amdgcn_exp2f(fmaf(sa[r], sc2, l4[r] + kbias));
          p[hh][r] = pr;
          ds[hh][r] = pr * (dp[r] - d4[r]);
        }
      }
      const lpx8_t pf = pack8(p[0], p[1]);
      const lpx8_t dsf = pack8(ds[0], ds[1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        dv[dt] = mfma_lp(lds_frag_tr(ldsDO, s2 * 32, s2 * 32 + 16, dt * 16, lane), pf, dv[dt]);
        dk[dt] = mfma_lp(lds_frag_tr(ldsQ, s2 * 32, s2 * 32 + 16, dt * 16, lane), dsf, dk[dt]);
      }
    }
  }
  if (tk < N) {
    lp_t* gk = a.dqkv + tok_row(a, b, tk) * a.lddq + a.D + h * HD + 4 * g;
    lp_t* gv = a.dqkv + tok_row(a, b, tk) * a.lddq + 2 * a.D + h * HD + 4 * g;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      *(u32x2_t*)(gk + dt * 16) = (u32x2_t){pack_lp2(dk[dt][0] * a.scale, dk[dt][1] * a.scale),
                                           pack_lp2(dk[dt][2] * a.scale, dk[dt][3] * a.scale)};
      *(u32x2_t*)(gv + dt * 16) = (u32x2_t){pack_lp2(dv[dt][0], dv[dt][1]), pack_lp2(dv[dt][2], dv[dt][3])};
    }
  }
}

template <typename K>
int set_lds_limit(K kernel, size_t bytes) {
  return hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
}

}  // namespace

static int attn_check(int B, int H, int Nv, int Nt, int D, int ld) {
  if (B <= 0 || H <= 0 || Nv < 0 || Nt < 0 || Nv + Nt <= 0) return 0;
  if (D != H * HD) return 0;
  if (ld % 8 != 0) return 0;
  return 1;
}

#ifdef FWD_PROFILE
static float* g_fwd_profile = nullptr;
extern "C" void simvg_attn_fwd_profile_buf(float* p) { g_fwd_profile = p; }
#endif
// workgroups per residency round for the L2 prefetch of the next round (attention.h): the CU count when the launch has more
// workgroups than CUs (one workgroup per CU: the kernels take > 80 KB of LDS), else 0.  SIMVG_ATTN_L2PF=0 switches it off.
static int attn_pf_stride(int nwg) {
  static const int cus = [] {
    if (const char* e = getenv("SIMVG_ATTN_L2PF")) if (atoi(e) == 0) return 0;
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return n;
  }();
  return nwg > cus ? cus : 0;
}

extern "C" int simvg_attn_fwd(const void* qkv, int ldqkv, void* out, int ldo, float* lse, const unsigned char* pad,
                              int B, int H, int Nv, int Nt, int D, float scale, hipStream_t stream) {
  SIMVG_CHECK_ARG(attn_check(B, H, Nv, Nt, D, ldqkv) && ldo % 8 == 0,
                  "attn_fwd: need head_dim 64, 16-B aligned rows");
  AttnArgs a{(const lp_t*)qkv, ldqkv, (lp_t*)out, ldo, nullptr, 0, nullptr, 0, lse, nullptr, pad, B, H, Nv, Nt, D, scale};
#ifdef FWD_PROFILE
  a.delta = g_fwd_profile;
#endif
  const int N = Nv + Nt, npad = ((cdiv(N, 16) + 1) / 2) * 32;
  if (N > MAX_KT * 16) {      // K / V do not fit the LDS: stream them in blocks (online softmax)
    static bool oncet = set_lds_limit(attn_fwd_tiled_kernel, TILED_LDS);
    (void)oncet;
    hipLaunchKernelGGL(attn_fwd_tiled_kernel, dim3(B * H, cdiv(N, 16 * TW)), dim3(TW * 64), TILED_LDS, stream, a);
    SIMVG_LAUNCH_CHECK();
    return SIMVG_OK;
  }
  const size_t shm = (size_t)2 * npad * ROWB + npad * sizeof(float);
  static bool once = set_lds_limit(attn_fwd_kernel, 160 * 1024) && set_lds_limit(attn_fwd_t_kernel<27, 25, 768>, 160 * 1024);
  (void)once;
  // few heads in flight (forward_test at B = 1 ... 4: B * H = 12 ... 48 workgroups on 256 CUs): the query strips of a head
  // are dealt over up to 3 workgroups (each loads the head's K / V; one strip per wave instead of three in a row)
  const int qsplit = B * H * 3 <= 256 ? std::min(3, cdiv(cdiv(N, 16), 12)) : 1;
  const bool off32 = (long)B * N * ldqkv < (1L << 31) && (long)B * N * ldo < (1L << 31) && (long)B * H * N < (1L << 31);   // the kernel's 32-bit element offsets
  if (cdiv(N, 16) == 27 && Nv / 16 >= 25 && off32) {   // the path's geometry: 1 + (640/32)^2 vision + 20 text tokens
    a.pf_stride = attn_pf_stride(B * H);
    hipLaunchKernelGGL((attn_fwd_t_kernel<27, 25, 768>), dim3(B * H, qsplit), dim3(768), shm, stream, a);
    SIMVG_LAUNCH_CHECK();
    return SIMVG_OK;
  }
  // 12 waves per workgroup (3 per SIMD): the 27 query strips of a 421-token head take 3 rounds instead of 4 and the
  // MFMA / softmax-VALU / LDS phases of three waves overlap on every SIMD (512 threads: 112 us, 768: 95 us)
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(B * H, qsplit), dim3(768), shm, stream, a);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}

// The encoder attention's QK^T contraction alone (K of a head staged in LDS, Q fragments from global, 16x16x32 MFMAs), for the
// path's geometry (27 key tiles): rowmax[b*H + h][t] = max_k scale * q_t . k_k (key padding -> -inf).  A measurement entry point:
// north_star quotes a target for "the encoder QK^T GEMM", SURVEY 8(d) asks for it beside the fused kernel's figure.
extern "C" int simvg_attn_qk_probe(const void* qkv, int ldqkv, float* rowmax, const unsigned char* pad, int B, int H, int Nv, int Nt, int D,
                                   float scale, hipStream_t stream) {
  SIMVG_CHECK_ARG(attn_check(B, H, Nv, Nt, D, ldqkv) && rowmax, "attn_qk_probe: need head_dim 64, 16-B aligned rows");
  const int N = Nv + Nt, npad = ((cdiv(N, 16) + 1) / 2) * 32;
  SIMVG_CHECK_ARG(cdiv(N, 16) == 27 && Nv / 16 >= 25, "attn_qk_probe: built for the path's geometry (417 .. 432 tokens, >= 400 of them vision)");
  SIMVG_CHECK_ARG((long)B * N * ldqkv < (1L << 31), "attn_qk_probe: 32-bit element offsets");
  AttnArgs a{(const lp_t*)qkv, ldqkv, nullptr, 0, nullptr, 0, nullptr, 0, rowmax, nullptr, pad, B, H, Nv, Nt, D, scale};
  const size_t shm = (size_t)2 * npad * ROWB + npad * sizeof(float);
  static bool once = set_lds_limit(attn_fwd_t_kernel<27, 25, 768, ATTN_FWD_G, true>, 160 * 1024);
  (void)once;
  hipLaunchKernelGGL((attn_fwd_t_kernel<27, 25, 768, ATTN_FWD_G, true>), dim3(B * H, 1), dim3(768), shm, stream, a);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}

extern "C" int simvg_attn_bwd(const void* qkv, int ldqkv, const void* out, int ldo, const void* dout, int lddo,
                              void* dqkv, int lddqkv, const float* lse, float* delta_ws, const unsigned char* pad,
                              int B, int H, int Nv, int Nt, int D, float scale, hipStream_t stream) {
  SIMVG_CHECK_ARG(attn_check(B, H, Nv, Nt, D, ldqkv) && ldo % 8 == 0 && lddo % 8 == 0 && lddqkv % 8 == 0,
                  "attn_bwd: need head_dim 64, 16-B aligned rows");
  SIMVG_CHECK_ARG(lse && delta_ws, "attn_bwd: lse and delta workspace required");
  AttnArgs a{(const lp_t*)qkv, ldqkv, (lp_t*)out, ldo, (const lp_t*)dout, lddo, (lp_t*)dqkv, lddqkv,
             (float*)lse, delta_ws, pad, B, H, Nv, Nt, D, scale};
  const int N = Nv + Nt, npad = ((cdiv(N, 16) + 1) / 2) * 32;
  if (N > MAX_KT * 16) {
    static bool oncet = set_lds_limit(attn_bwd_dq_tiled_kernel, TILED_LDS) && set_lds_limit(attn_bwd_dkv_tiled_kernel, TILED_LDS);
    (void)oncet;
    hipLaunchKernelGGL(attn_bwd_dq_tiled_kernel, dim3(B * H, cdiv(N, 16 * TW)), dim3(TW * 64), TILED_LDS, stream, a);
    hipLaunchKernelGGL(attn_bwd_dkv_tiled_kernel, dim3(B * H, cdiv(N, 16 * TW)), dim3(TW * 64), TILED_LDS, stream, a);
    SIMVG_LAUNCH_CHECK();
    return SIMVG_OK;
  }
  const size_t shm1 = (size_t)2 * npad * ROWB + npad * sizeof(float);
  const size_t shm2 = (size_t)2 * npad * ROWB + 2 * npad * sizeof(float);
  static bool once1 = set_lds_limit(attn_bwd_dq_kernel, 160 * 1024);
  static bool once2 = set_lds_limit(attn_bwd_dkv_kernel, 160 * 1024);
  (void)once1; (void)once2;
  static bool once3 = set_lds_limit(attn_bwd_dq_t_kernel<27, 25, 768>, 160 * 1024) && set_lds_limit(attn_bwd_dkv_t_kernel<27, 25, 512>, 160 * 1024);
  (void)once3;
  // the path's geometry (27 key tiles, vision keys fill the first 24): backward in ONE pass (attention_bwd1.hip), 183 us against
  // 209-217 us for the two kernels below; SIMVG_ATTN_BWD1=0 keeps the two kernels (A/B runs)
  const char* sw = getenv("SIMVG_ATTN_BWD1");          // read per call: the tests flip it
  const bool one_pass = !(sw && atoi(sw) == 0);
  if (one_pass && simvg_attn_bwd_onepass(a, stream)) {
    SIMVG_LAUNCH_CHECK();
    return SIMVG_OK;
  }
  if (cdiv(N, 16) == 27 && Nv / 16 >= 25) {
    // dQ: 12 waves (88 us; 8 waves 94); dK/dV: 8 waves with the next query pair's fragments in flight (115 us; the 12-wave
    // build of the same code spills in its loop: 190 us) -- profiles/r02_sweeps.md
    hipLaunchKernelGGL((attn_bwd_dq_t_kernel<27, 25, 768>), dim3(B * H), dim3(768), shm1, stream, a);
    hipLaunchKernelGGL((attn_bwd_dkv_t_kernel<27, 25, 512>), dim3(B * H), dim3(512), shm2, stream, a);
  } else {
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(B * H), dim3(768), shm1, stream, a);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3(B * H), dim3(768), shm2, stream, a);
  }
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}
