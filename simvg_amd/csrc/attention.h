// Declarations of the encoder attention kernels (attention.hip: K / V resident per head).
#pragma once
#include "common.h"

struct AttnArgs {
  const lp_t* qkv; int ld;    // [M, 3*D]
  lp_t* out; int ldo;         // [M, D]   forward output / saved O in backward
  const lp_t* dout; int lddo; // [M, D]   backward: grad of O
  lp_t* dqkv; int lddq;       // [M, 3*D] backward: grads
  float* lse;                   // [B*H, N]
  float* delta;                 // [B*H, N]
  const unsigned char* pad;     // [B, Nt] 1 = padded text key, or null
  int B, H, Nv, Nt, D;
  float scale;
  int pf_stride;                // workgroups per residency round (= CUs) for the next round's L2 prefetch; 0 = off
};

namespace {

constexpr int HD = 64;          // head dim
constexpr int ROWB = 128;       // LDS row stride in bytes: 64 bf16, no padding -- conflicts are avoided by the slot swizzle below
// Resident K / V / Q / dO rows are read two ways: K-major fragments with ds_read_b128 (lane j -> row j, 16-B slot g) and
// transposed fragments with ds_read_b64_tr_b16 (8 rows x 32 B per LDS cycle).  The hardware serves a b128 read in 16-lane
// groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS), i.e. all 16 rows with slot g for the
// outer rows and g+1 for the middle ones; with the former 144-B padded rows 7 of the 8 middle lanes met an outer lane on
// the same banks, and a transposed read wrapped its 8th row onto the first (PMC: SQ_LDS_BANK_CONFLICT = 42-45 % of the
// LDS cycles of all three kernels).  Physical slot = slot ^ (row & 6) on 128-B rows is conflict-free for both patterns
// (and for the 8-lane groups of the ds_write_b128 fill).
__device__ __forceinline__ int lds_slot(int row, int slot) { return slot ^ (row & 6); }
constexpr int MAX_KT = 28;      // 28 * 16 = 448 >= 421 keys

// L2 prefetch for the workgroup that follows this one on the same XCD.  One workgroup per CU and equal work per workgroup: the
// 256 CUs run their residency rounds in lockstep, and every round opens with each CU pulling its head's K / V (115 KB) through an
// L1 that tracks a few dozen misses -- bound by latency x misses in flight at ~10 B/clk/CU whatever is issued, 12 k of a forward
// workgroup's 39 k cycles with the matrix pipes idle (tools/dev/attn_fwd_profile.py; starting the CUs staggered changed nothing:
// it is not HBM contention).  Workgroup i + pf_stride is dispatched to the same XCD as workgroup i (XCD = id mod 8) one round
// later, so waves of workgroup i that have finished touch one dword of every 128-byte line of that workgroup's K / V rows: the
// lines sit in the XCD's L2 when the next prologue asks for them (forward: prologue 12 k -> 6.5 k cycles from the second round on,
// 71.9 -> 69.1 us isolated).  The one-pass backward does NOT gain: in front of its dK / dV tail the stores' 3.5 MB per XCD push the
// lines out again (+8 us), behind it the wave has to outlive the touches (+4 us; a wave must not end with loads in flight).
// `rows` rows of 128 B starting at column col0, dealt over `nlanes` lanes.
__device__ __forceinline__ void l2_touch_rows(const AttnArgs& a, const lp_t* base, int ld, int col0, int b, int N, int rows,
                                              int lane_id, int nlanes) {
  for (int r = lane_id; r < rows; r += nlanes) {
    const int t = r < N ? r : N - 1;
    const int row = t < a.Nv ? b * a.Nv + t : a.B * a.Nv + b * a.Nt + (t - a.Nv);
    const int v = *(const volatile int*)(base + (unsigned)(row * ld + col0));
    asm volatile("" ::"v"(v));
  }
}

__device__ __forceinline__ long tok_row(const AttnArgs& a, int b, int t) {
  return t < a.Nv ? (long)b * a.Nv + t : (long)a.B * a.Nv + (long)b * a.Nt + (t - a.Nv);
}

// copy rows [0,nrows_pad) x 64 bf16 of NSRC column blocks (K and V, Q and dO ...) of one head into LDS (zero beyond N).
// All UNR x NSRC 16-byte loads of a pass are issued before the first LDS write: the former one-load loop
// (global_load -> s_waitcnt vmcnt(0) -> ds_write per iteration, hipcc does not unroll a run-time trip count) paid one full
// memory round trip per 16 bytes and thread -- 10 in a row in the forward's prologue, 14 in the one-pass backward's, with one
// workgroup per CU and nothing else to run (round 5: the QK^T-only probe spent 3/4 of its time there).  Loads are unconditional
// on clamped indices (a load under a divergent branch is waited for at the branch's end); UNR is chosen per call site so that
// UNR x blockDim covers nrows_pad x 8 chunks in one pass.
struct HeadSrc { const lp_t* base; int ld; int col0; char* lds; };
template <int UNR, int NSRC> struct HeadChunks { u32x4_t v[NSRC][UNR]; };
// the two halves of a pass (chunks c0 + tid + i * blockDim, i < UNR): callers with more loads to put in flight issue, request their
// own, and commit afterwards
template <int UNR, int NSRC>
__device__ __forceinline__ void heads_issue(const AttnArgs& a, const HeadSrc (&src)[NSRC], int b, int N, int total, int c0,
                                            HeadChunks<UNR, NSRC>& ch) {
  const int nt = blockDim.x;
#pragma unroll
  for (int i = 0; i < UNR; ++i) {
    const int c = min(c0 + (int)threadIdx.x + i * nt, total - 1);
    const int row = min(c >> 3, N - 1), slot = c & 7;
    const long r = tok_row(a, b, row);
#pragma unroll
    for (int k = 0; k < NSRC; ++k) ch.v[k][i] = *(const u32x4_t*)(src[k].base + r * src[k].ld + src[k].col0 + slot * 8);
  }
}
template <int UNR, int NSRC>
__device__ __forceinline__ void heads_commit(const HeadSrc (&src)[NSRC], int N, int total, int c0, const HeadChunks<UNR, NSRC>& ch) {
  const int nt = blockDim.x;
#pragma unroll
  for (int i = 0; i < UNR; ++i) {
    const int c = c0 + (int)threadIdx.x + i * nt;
    const int row = c >> 3, slot = c & 7;
    if (c < total) {
#pragma unroll
      for (int k = 0; k < NSRC; ++k)
        *(u32x4_t*)(src[k].lds + row * ROWB + lds_slot(row, slot) * 16) = row < N ? ch.v[k][i] : (u32x4_t){0u, 0u, 0u, 0u};
    }
  }
}
template <int UNR, int NSRC>
__device__ __forceinline__ void load_heads_to_lds(const AttnArgs& a, const HeadSrc (&src)[NSRC], int b, int N, int nrows_pad) {
  const int total = nrows_pad * 8;
  for (int c0 = 0; c0 < total; c0 += UNR * blockDim.x) {
    HeadChunks<UNR, NSRC> ch;
    heads_issue<UNR, NSRC>(a, src, b, N, total, c0, ch);
    heads_commit<UNR, NSRC>(src, N, total, c0, ch);
  }
}
template <int UNR>
__device__ __forceinline__ void load_head_to_lds(const AttnArgs& a, const lp_t* base, int ld, int col0, int b, int N,
                                                 int nrows_pad, char* lds) {
  const HeadSrc src[1] = {{base, ld, col0, lds}};
  load_heads_to_lds<UNR, 1>(a, src, b, N, nrows_pad);
}

__device__ __forceinline__ lpx8_t lds_frag(const char* lds, int row, int slot) {
  return *(const lpx8_t*)(lds + row * ROWB + lds_slot(row, slot) * 16);
}

// transposed fragment for contraction over LDS rows: lane (i = lane&15 -> column c0 + i,
// g = lane>>4); rows rowA+4g..+3 (elements 0..3) and rowB+4g..+3 (elements 4..7)
__device__ __forceinline__ lpx8_t lds_frag_tr(const char* lds, int rowA, int rowB, int c0, int lane) {
  const int i = lane & 15, g = lane >> 4;
  const int col = c0 + 4 * (i & 3), slot = col >> 3, within = (col & 7) * 2;
  const int ra = rowA + 4 * g + (i >> 2), rb = rowB + 4 * g + (i >> 2);
  const lpx4_t lo = lds_read_tr16(lds + ra * ROWB + lds_slot(ra, slot) * 16 + within);
  const lpx4_t hi = lds_read_tr16(lds + rb * ROWB + lds_slot(rb, slot) * 16 + within);
  return (lpx8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

__device__ __forceinline__ lpx8_t pack8(const float* lo, const float* hi) {
  union { lpx8_t v; unsigned int u[4]; } r;
  r.u[0] = pack_lp2(lo[0], lo[1]); r.u[1] = pack_lp2(lo[2], lo[3]);
  r.u[2] = pack_lp2(hi[0], hi[1]); r.u[3] = pack_lp2(hi[2], hi[3]);
  return r.v;
}

// The backward kernels are VALU-issue-bound (a wave64 VALU instruction holds its SIMD's issue port for 4 cycles: 930 VALU against
// 164 MFMAs per 16-query strip of dQ, profiles/r03_sweeps.md): the per-score arithmetic  ds = exp2(s c - lse) (dp - delta)  is
// written on pairs so that hipcc emits v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 (same IEEE operations, half the instructions)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void ds_pair(float s0, float s1, f32x2_t c, f32x2_t nl, float dp0, float dp1, f32x2_t dl, float& p0,
                                        float& p1, float& ds0, float& ds1) {
  const f32x2_t t = __builtin_elementwise_fma((f32x2_t){s0, s1}, c, nl);
  const f32x2_t p = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
  const f32x2_t d = p * ((f32x2_t){dp0, dp1} - dl);
  p0 = p[0]; p1 = p[1]; ds0 = d[0]; ds1 = d[1];
}

}  // namespace

// attention_bwd1.hip: backward in one pass for the path's geometry; false = geometry not handled (the two kernels of attention.hip run)
bool simvg_attn_bwd_onepass(const AttnArgs& a, hipStream_t stream);

