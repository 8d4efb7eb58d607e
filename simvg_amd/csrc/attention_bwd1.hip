// simvg-build-flags: -mllvm -amdgpu-mfma-vgpr-form=1
// (the directive above is read by simvg_amd/build.py: this file only.  With one wave per SIMD the kernel below has 256 VGPRs + 256
// AccVGPRs; left alone hipcc parks the S / dP tiles in the spare AccVGPRs and pays v_accvgpr_read + s_nop for every value the softmax
// touches: 198 us per call against 183 us with MFMA results in VGPR form.)
#include <stdlib.h>
#include <type_traits>

#include "attention.h"

namespace {

// ------------------------------------------------------------------------------------------
// Encoder self-attention backward in ONE pass (round 4): S and dP are formed once per (query pair, key strip) and feed all three
// gradients -- 5 GEMM-equivalents instead of the 7 of the dq + dkv kernels of attention.hip, one read of qkv / o / dO, delta formed on
// the way in.  (The first version, its ablations and what each step below bought: profiles/r04_sweeps.md section 5.)
// 4 waves per (sample, head), one per SIMD; wave w owns key strips 4 s + w (s = 0..6): dK^T / dV^T accumulators (224 AGPRs) and V
// fragments; K resident in LDS; Q / dO / o in a 3-slot ring of 32-query pairs; dS transposed through a wave-private LDS tile;
// the waves' dQ^T partials meet through plain stores + two barriers.  Every LDS read is inline asm with a compile-time offset
// from one of a dozen per-lane bases (the first version let hipcc hoist ~80 loop-invariant addresses into registers).
// ------------------------------------------------------------------------------------------
constexpr int B1_SROW = 72;      // bytes per key row of the dS tile: 32 queries x 2 B + 8
constexpr int B1_DQLD = 68;      // floats per QUERY row of a dQ partial: 64 hd + 4 (16-byte accesses on both sides)

template <int OFF> __device__ __forceinline__ lpx8_t b1_rd(unsigned addr) { return __builtin_bit_cast(lpx8_t, lds_b128_asm<OFF>(addr)); }
template <int OFFA, int OFFB> __device__ __forceinline__ lpx8_t b1_rdt(unsigned addr) {
  const u32x2_t lo = lds_tr16_asm<OFFA>(addr), hi = lds_tr16_asm<OFFB>(addr);
  return __builtin_bit_cast(lpx8_t, (u32x4_t){lo[0], lo[1], hi[0], hi[1]});
}

template <int NQT>
__global__ __launch_bounds__(256) void attn_bwd_one_kernel(AttnArgs a) {
  constexpr int NS2 = (NQT + 1) / 2, NPAD = NS2 * 32, NSTRIP = NPAD / 16, SPW = NSTRIP / 4;
  static_assert(NSTRIP == 28 && SPW == 7, "written for 28 key strips: 4 waves x 7");
  constexpr int STG_W = (SPW * 16 + 16) * B1_SROW;           // + 16 zero rows: the partner of the odd last strip
  constexpr int O_RING = NPAD * ROWB, O_STAT = O_RING + 3 * 8192, O_STG = O_STAT + 2 * 96 * 4, O_DQ = O_STG + 4 * STG_W;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ldsK = smem;
  char* ring = smem + O_RING;                                 // 3 slots x {Q [32][128 B], dO [32][128 B]}
  float* nlse_r = (float*)(smem + O_STAT);
  float* dl_r = nlse_r + 96;
  float* dqb = (float*)(smem + O_DQ);                         // [4 waves][32 queries][B1_DQLD]
  const int N = a.Nv + a.Nt;
  const int b = blockIdx.x / a.H, h = blockIdx.x - b * a.H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int lrow = tid >> 3, lslot = tid & 7;                 // this thread's 16-byte chunk of a pair's Q / dO / o rows
  const float sc2 = a.scale * 1.44269504088896340736f;
  const unsigned lds0 = lds_addr(smem);
  // per-lane bases (bytes from lds0 unless said otherwise)
  const unsigned offA = j * ROWB + ((g ^ (j & 6)) << 4), offB = j * ROWB + (((4 + g) ^ (j & 6)) << 4);          // row fragments
  const int tr_m = ((4 * g + (j >> 2)) & 6) >> 1, tr_x = (j & 3) >> 1;
  unsigned offT[4];                                            // transposed fragments, column tile dt
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) offT[dt] = (4 * g + (j >> 2)) * ROWB + ((2 * (dt ^ tr_m) + tr_x) << 4) + (j & 1) * 8;
  const unsigned kA = lds0 + wave * 2048 + offA, kB = lds0 + wave * 2048 + offB;                                 // + s * 8192
  unsigned kT[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) kT[dt] = lds0 + wave * 2048 + offT[dt];                                          // + s * 8192
  char* st = smem + O_STG + wave * STG_W;
  char* stw = st + j * B1_SROW + 8 * g;                                                                          // + s * 1152 (+ 32)
  const unsigned str = lds_addr(st) + (4 * g + (j >> 2)) * B1_SROW + (j & 3) * 8;                               // + s * 1152 (+ 32)
  float* dqw = dqb + wave * 32 * B1_DQLD + j * B1_DQLD + 4 * g;          // + qt * 16 * B1_DQLD + dt * 16: four hd values
  const float* dqr = dqb + lrow * B1_DQLD + lslot * 8;                  // + w * 32 * B1_DQLD: eight hd values of one query

  auto row32 = [&](int t) { return t < a.Nv ? b * a.Nv + t : a.B * a.Nv + b * a.Nt + (t - a.Nv); };   // tok_row without a branch
  auto fetch = [&](int pair, u32x4_t& vq, u32x4_t& vd, u32x4_t& vo, float& vl) {
    const int q = pair * 32 + lrow;
    const int r = row32(pair < NS2 && q < N ? q : N - 1);                // rows beyond N: a valid row, zeroed at commit
    vq = *(const u32x4_t*)(a.qkv + (unsigned)(r * a.ld + h * HD + lslot * 8));        // (element offsets fit 32 bits)
    vd = *(const u32x4_t*)(a.dout + (unsigned)(r * a.lddo + h * HD + lslot * 8));
    vo = *(const u32x4_t*)(a.out + (unsigned)(r * a.ldo + h * HD + lslot * 8));
    {   // every thread loads some valid element (a load under a divergent branch is waited for right at the branch's end)
      const int ql = pair * 32 + (tid & 31);
      vl = a.lse[(unsigned)(blockIdx.x * N + (pair < NS2 && ql < N ? ql : 0))];
    }
  };
  auto commit = [&](int pair, u32x4_t vq, u32x4_t vd, const u32x4_t& vo, float vl) {
    if (pair >= NS2) return;
    const int slot = pair % 3;
    char* Q = ring + slot * 8192;
    if (pair * 32 + lrow >= N) { vq = (u32x4_t){0u, 0u, 0u, 0u}; vd = vq; }
    *(u32x4_t*)(Q + lrow * ROWB + lds_slot(lrow, lslot) * 16) = vq;
    *(u32x4_t*)(Q + 4096 + lrow * ROWB + lds_slot(lrow, lslot) * 16) = vd;
    float part = 0.f;                                         // delta = rowsum(dO * o)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float d0, d1, o0, o1;
      unpack_lp2(vd[e], d0, d1);
      unpack_lp2(vo[e], o0, o1);
      part += d0 * o0 + d1 * o1;
    }
    // sum over the 8 lanes of a row on the DPP path (quad_perm [1,0,3,2], [2,3,0,1], then row_half_mirror: lane i <-> 7 - i):
    // __shfl_xor goes through ds_bpermute, three dependent LDS round trips
    part += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, part), 0xB1, 0xf, 0xf, true));
    part += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, part), 0x4E, 0xf, 0xf, true));
    part += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, part), 0x141, 0xf, 0xf, true));
    if (lslot == 0) dl_r[slot * 32 + lrow] = part;
    if (tid < 32) nlse_r[slot * 32 + tid] = pair * 32 + tid < N ? -vl * 1.44269504088896340736f : -INFINITY;   // log2 domain, negated
  };

#ifdef B1_PROFILE
#define B1_T(k_) do { if (blockIdx.x == 0 && lane == 0) ((unsigned*)a.delta)[(wave * 16 + p) * 8 + (k_)] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
#define B1_X(k_) do { if (blockIdx.x == 0 && lane == 0) ((unsigned*)a.delta)[(wave * 16 + 14) * 8 + (k_)] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define B1_T(k_) do { } while (0)
#define B1_X(k_) do { } while (0)
#endif
  B1_X(0);
  // ---- prologue: every global load of it (V fragments, the padding byte, the first two pairs' rows, the 14 K chunks per thread)
  // is requested before the first wait -- one memory round trip; the former order (K chunk by chunk, then V, then the pairs) paid 17
  lpx8_t v0[SPW], v1[SPW];
#pragma unroll
  for (int s = 0; s < SPW; ++s) {
    const int tk = (4 * s + wave) * 16 + j;
    const lp_t* vp = a.qkv + (unsigned)(row32(tk < N ? tk : N - 1) * a.ld + 2 * a.D + h * HD + 8 * g);
    v0[s] = *(const lpx8_t*)vp;
    v1[s] = *(const lpx8_t*)(vp + 32);
  }
  u32x4_t pq0, pd0, po0, pq1, pd1, po1;
  float pl0, pl1;
  fetch(0, pq0, pd0, po0, pl0);
  fetch(1, pq1, pd1, po1, pl1);
  static_assert(NPAD * 8 % 256 == 0, "K chunks per thread");
  const HeadSrc ksrc[1] = {{a.qkv, a.ld, a.D + h * HD, ldsK}};
  HeadChunks<NPAD * 8 / 256, 1> kch;
  heads_issue<NPAD * 8 / 256, 1>(a, ksrc, b, N, NPAD * 8, 0, kch);
  const int kb_tk = (4 * (SPW - 1) + wave) * 16 + j;          // the last strip of every wave: text / padding keys
  const unsigned char kb_pad = *(a.pad ? a.pad + b * a.Nt + min(max(kb_tk - a.Nv, 0), max(a.Nt - 1, 0)) : (const unsigned char*)a.qkv);
  heads_commit<NPAD * 8 / 256, 1>(ksrc, N, NPAD * 8, 0, kch);
#pragma unroll
  for (int s = 0; s < SPW; ++s) asm volatile("" : "+v"(v0[s]), "+v"(v1[s]));   // waited for HERE, not by a vmcnt(0) at the loop's head
  const float kbias = (kb_tk >= N || (a.pad && kb_tk >= a.Nv && kb_pad != 0)) ? -INFINITY : 0.f;
  for (int i = lane; i < 16 * B1_SROW / 4; i += 64) ((unsigned int*)(st + SPW * 16 * B1_SROW))[i] = 0u;
  commit(0, pq0, pd0, po0, pl0);
  commit(1, pq1, pd1, po1, pl1);
  // the accumulators are born in the first pair's MFMAs (C = 0 literal): 224 zero-initialised values would sit in VGPRs before
  // hipcc moves them to the AccVGPR file, and push the V fragments out to scratch for the whole launch
  f32x4_t dk[SPW][4], dv[SPW][4];
  B1_X(1);
  __syncthreads();
  B1_X(2);

  // the pair's Q / dO fragments (row form for S / dP, transposed for dV / dK) and the first K fragments: requested right after the
  // barrier that publishes the pair, i.e. under the previous pair's write-out
  lpx8_t qc[2][2], dc[2][2], qf[4], dof[4], kc0, kc1, kn0, kn1;
  f32x4_t l4[2], d4[2];
  const unsigned statA = lds0 + O_STAT + 16 * g;               // + slot * 128 (+ 64: second query tile; + 384: delta)
  // in four chunks of eight reads, so that the dQ phase can interleave them with its own and still wait by count (lgkmcnt <= 15)
  auto request_chunk = [&](int pair, auto chunk_tag) {
    constexpr int CH = decltype(chunk_tag)::value;
    const unsigned Qa = lds0 + O_RING + (pair % 3) * 8192;
    if constexpr (CH == 0) {
      qc[0][0] = b1_rd<0>(Qa + offA);        qc[0][1] = b1_rd<0>(Qa + offB);
      qc[1][0] = b1_rd<2048>(Qa + offA);     qc[1][1] = b1_rd<2048>(Qa + offB);
      dc[0][0] = b1_rd<4096>(Qa + offA);     dc[0][1] = b1_rd<4096>(Qa + offB);
      dc[1][0] = b1_rd<6144>(Qa + offA);     dc[1][1] = b1_rd<6144>(Qa + offB);
    } else if constexpr (CH == 1) {
      qf[0] = b1_rdt<0, 2048>(Qa + offT[0]); qf[1] = b1_rdt<0, 2048>(Qa + offT[1]);
      qf[2] = b1_rdt<0, 2048>(Qa + offT[2]); qf[3] = b1_rdt<0, 2048>(Qa + offT[3]);
    } else if constexpr (CH == 2) {
      dof[0] = b1_rdt<4096, 6144>(Qa + offT[0]); dof[1] = b1_rdt<4096, 6144>(Qa + offT[1]);
      dof[2] = b1_rdt<4096, 6144>(Qa + offT[2]); dof[3] = b1_rdt<4096, 6144>(Qa + offT[3]);
    } else {
      kc0 = b1_rd<0>(kA);    kc1 = b1_rd<0>(kB);
      kn0 = b1_rd<8192>(kA); kn1 = b1_rd<8192>(kB);
      const unsigned sa_ = statA + (pair % 3) * 128;
      l4[0] = __builtin_bit_cast(f32x4_t, lds_b128_asm<0>(sa_));   l4[1] = __builtin_bit_cast(f32x4_t, lds_b128_asm<64>(sa_));
      d4[0] = __builtin_bit_cast(f32x4_t, lds_b128_asm<384>(sa_)); d4[1] = __builtin_bit_cast(f32x4_t, lds_b128_asm<448>(sa_));
    }
  };
  auto request_frags = [&](int pair) {
    request_chunk(pair, std::integral_constant<int, 0>{}); request_chunk(pair, std::integral_constant<int, 1>{});
    request_chunk(pair, std::integral_constant<int, 2>{}); request_chunk(pair, std::integral_constant<int, 3>{});
  };
  // dQ rows of a finished pair: the four waves' partials summed, 8 hd values of one query per thread -> one 16-byte store.  It runs
  // in the MIDDLE of the next pair (after its strip loop), away from the vmcnt waits at the loop's head
  auto write_out = [&](int pp) {
    const int q = pp * 32 + lrow;
    f32x4_t lo[4], hi[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      lo[w] = *(const f32x4_t*)(dqr + w * 32 * B1_DQLD);
      hi[w] = *(const f32x4_t*)(dqr + w * 32 * B1_DQLD + 4);
    }
    const f32x4_t sl = ((lo[0] + lo[1]) + (lo[2] + lo[3])) * a.scale, sh = ((hi[0] + hi[1]) + (hi[2] + hi[3])) * a.scale;
    const float v[8] = {sl[0], sl[1], sl[2], sl[3], sh[0], sh[1], sh[2], sh[3]};
    // UNCONDITIONAL store (rows beyond N land in the first bytes of the delta workspace, which this kernel does not use): behind a
    // branch hipcc can no longer count it and waits vmcnt(0) -- for the store's completion -- at the next use of a prefetched row
    lp_t* dstp = q < N ? a.dqkv + (long)row32(q) * a.lddq + h * HD + lslot * 8 : (lp_t*)a.delta + lslot * 8;
    *(u32x4_t*)dstp = (u32x4_t){pack_lp2(v[0], v[1]), pack_lp2(v[2], v[3]), pack_lp2(v[4], v[5]), pack_lp2(v[6], v[7])};
  };
  request_frags(0);
  lds_wait_all();            // (asm reads: nothing may copy their destination registers before they have landed)
  auto pair_body = [&](auto first_tag, int p) {
    constexpr bool FIRST = decltype(first_tag)::value;
    B1_T(0);
    u32x4_t nq, nd, no;
    float nl;
    fetch(p + 2, nq, nd, no, nl);
    const int slot = p % 3;
    lds_wait_all();
    B1_T(1);
    const f32x2_t c2 = {sc2, sc2};
    const f32x4_t nd4[2] = {-d4[0], -d4[1]};
    // software pipeline over the wave's strips: the S / dP MFMAs of strip s + 1 are issued BEFORE the dV / dK MFMAs of strip s, so
    // that the softmax arithmetic of strip s + 1 (VALU) runs while the matrix pipe works through the eight dV / dK MFMAs of strip s
    f32x4_t sa[2], dp[2];
    union { lpx8_t v; unsigned int u[4]; } pf, dsf;
#define B1_SDP(s)                                                                                                        \
    _Pragma("unroll") for (int hh = 0; hh < 2; ++hh) {                                                                   \
      sa[hh] = mfma_lp(qc[hh][0], kc0, (f32x4_t){0.f, 0.f, 0.f, 0.f});                                                   \
      dp[hh] = mfma_lp(dc[hh][0], v0[s], nd4[hh]);     /* starts at -delta of its query rows: dP - delta for free */   \
      sa[hh] = mfma_lp(qc[hh][1], kc1, sa[hh]);                                                                          \
      dp[hh] = mfma_lp(dc[hh][1], v1[s], dp[hh]);                                                                        \
    }
#define B1_SOFTMAX(s)                                                                                                    \
    {                                                                                                                    \
      float pr[2][4], ds[2][4];                                                                                          \
      _Pragma("unroll") for (int hh = 0; hh < 2; ++hh) {                                                                 \
        const bool m = (s) == SPW - 1;                                                                                   \
        ds_pair(sa[hh][0], sa[hh][1], c2, (f32x2_t){m ? l4[hh][0] + kbias : l4[hh][0], m ? l4[hh][1] + kbias : l4[hh][1]}, \
                dp[hh][0], dp[hh][1], (f32x2_t){0.f, 0.f}, pr[hh][0], pr[hh][1], ds[hh][0], ds[hh][1]);                  \
        ds_pair(sa[hh][2], sa[hh][3], c2, (f32x2_t){m ? l4[hh][2] + kbias : l4[hh][2], m ? l4[hh][3] + kbias : l4[hh][3]}, \
                dp[hh][2], dp[hh][3], (f32x2_t){0.f, 0.f}, pr[hh][2], pr[hh][3], ds[hh][2], ds[hh][3]);                  \
      }                                                                                                                  \
      pf.u[0] = pack_lp2_raw(pr[0][0], pr[0][1]); pf.u[1] = pack_lp2_raw(pr[0][2], pr[0][3]);                            \
      pf.u[2] = pack_lp2_raw(pr[1][0], pr[1][1]); pf.u[3] = pack_lp2_raw(pr[1][2], pr[1][3]);                            \
      dsf.v = pack8(ds[0], ds[1]);                                                                                       \
    }
// (no scheduling fences between the three parts of a strip: left to itself hipcc's order measures 179 us, fenced 182; a three-stage
// form -- S / dP of strip s + 2, softmax of s + 1, dV / dK of s as one region, with or without sched_group_barrier hints -- 186-188)
#define B1_FENCE
#define B1_STRIP(s)                                                                                                      \
    {                                                                                                                    \
      const lpx8_t pfc = pf.v, dsfc = dsf.v;                                                                             \
      const unsigned dsu0 = dsf.u[0], dsu1 = dsf.u[1], dsu2 = dsf.u[2], dsu3 = dsf.u[3];                                 \
      if ((s) + 1 < SPW) {                                                                                               \
        kc0 = kn0; kc1 = kn1;                                                                                            \
        if ((s) + 2 < SPW) { kn0 = b1_rd<((s) + 2 < SPW ? (s) + 2 : 0) * 8192>(kA); kn1 = b1_rd<((s) + 2 < SPW ? (s) + 2 : 0) * 8192>(kB); } \
        B1_SDP((s) + 1 < SPW ? (s) + 1 : 0)                                                                              \
      }                                                                                                                  \
      B1_FENCE                                                                                                           \
      _Pragma("unroll") for (int dt = 0; dt < 4; ++dt) {                                                                 \
        dv[s][dt] = mfma_lp(dof[dt], pfc, FIRST ? (f32x4_t){0.f, 0.f, 0.f, 0.f} : dv[s][dt]);                            \
        dk[s][dt] = mfma_lp(qf[dt], dsfc, FIRST ? (f32x4_t){0.f, 0.f, 0.f, 0.f} : dk[s][dt]);                            \
      }                                                                                                                  \
      *(u32x2_t*)(stw + (s) * 16 * B1_SROW) = (u32x2_t){dsu0, dsu1};                                                     \
      *(u32x2_t*)(stw + (s) * 16 * B1_SROW + 32) = (u32x2_t){dsu2, dsu3};                                                \
      B1_FENCE                                                                                                           \
      if ((s) + 1 < SPW) B1_SOFTMAX((s) + 1)                                                                             \
      /* the K fragments requested at the top of this strip have landed once at most the (younger) tile store is pending */  \
      asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory"); __builtin_amdgcn_sched_barrier(0);                              \
    }
    B1_SDP(0)
    B1_SOFTMAX(0)
    B1_STRIP(0) B1_STRIP(1) B1_STRIP(2) B1_STRIP(3) B1_STRIP(4) B1_STRIP(5) B1_STRIP(6)
#undef B1_STRIP
#undef B1_SOFTMAX
#undef B1_SDP
    B1_T(2);
    // the next pair's fragments: its ring slot was published by the barrier that ended the previous pair, and this pair's are dead
    if (!FIRST) write_out(p - 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // write_out's own reads: nothing of the compiler's is pending below
    // ---- dQ^T [64 hd][32 queries] partial over this wave's 112 keys (the tile is this wave's own: no barrier)
    f32x4_t dq[4][2];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dq[dt][0] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dq[dt][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
    // the fragments of k-step ks + 1 are requested before the MFMAs of k-step ks: 12 reads in flight while waiting by count
#define B1_KS_RD(ks, T0, T1, KT)                                                                                         \
    {                                                                                                                    \
      constexpr int sA = 2 * (ks), sB = 2 * (ks) + 1, kb = sB < SPW ? sB : sA;                                            \
      T0 = b1_rdt<sA * 16 * B1_SROW, sB * 16 * B1_SROW>(str);                                                            \
      T1 = b1_rdt<sA * 16 * B1_SROW + 32, sB * 16 * B1_SROW + 32>(str);                                                  \
      _Pragma("unroll") for (int dt = 0; dt < 4; ++dt) KT[dt] = b1_rdt<sA * 8192, kb * 8192>(kT[dt]);                    \
    }
#define B1_KS_MM(T0, T1, KT)                                                                                             \
    _Pragma("unroll") for (int dt = 0; dt < 4; ++dt) {                                                                   \
      dq[dt][0] = mfma_lp(KT[dt], T0, dq[dt][0]);                                                                        \
      dq[dt][1] = mfma_lp(KT[dt], T1, dq[dt][1]);                                                                        \
    }
    {
      lpx8_t ta0, ta1, ka[4], tb0, tb1, kb4[4];
      B1_KS_RD(0, ta0, ta1, ka)
      B1_KS_RD(1, tb0, tb1, kb4)
      asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
      B1_KS_MM(ta0, ta1, ka)
      __builtin_amdgcn_sched_barrier(0);
      B1_KS_RD(2, ta0, ta1, ka)
      asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
      B1_KS_MM(tb0, tb1, kb4)
      __builtin_amdgcn_sched_barrier(0);
      B1_KS_RD(3, tb0, tb1, kb4)
      asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
      B1_KS_MM(ta0, ta1, ka)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
      B1_KS_MM(tb0, tb1, kb4)
    }
#undef B1_KS_RD
#undef B1_KS_MM
    // the next pair's fragments (this pair's are dead, its ring slot was published by the barrier that ended the previous pair):
    // they land under the barrier wait and the commit below
    if (p + 1 < NS2) request_frags(p + 1);
    B1_T(3);
    __builtin_amdgcn_s_barrier();      // the write-out of the previous pair has read the partials (nothing to publish: no waitcnt)
    B1_T(4);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) *(f32x4_t*)(dqw + qt * 16 * B1_DQLD + dt * 16) = dq[dt][qt];
    commit(p + 2, nq, nd, no, nl);
    B1_T(5);
    __syncthreads();
    B1_T(6);
    lds_wait_all();          // the next pair's fragments have landed before the loop's back edge may copy them
    B1_T(7);
  };
  pair_body(std::true_type{}, 0);
  for (int p = 1; p < NS2; ++p) pair_body(std::false_type{}, p);
  B1_X(3);
  write_out(NS2 - 1);
  B1_X(4);
  // ---- dK^T, dV^T of this wave's strips.  A lane holds 4 consecutive hd values of key row j per 16-column tile (8 bytes); one
  // v_permlane16_swap per dword between the tiles dt and dt + 1 leaves lanes of even g with 8 consecutive values of tile dt and
  // lanes of odd g with 8 of tile dt + 1: 16-byte stores, half as many (the tail was store-ISSUE-bound: 15 k of a workgroup's
  // 126 k cycles for 56 8-byte stores per lane, tools/dev/attn_onepass_profile.py; MI355X guide T21)
  const int col16 = (g >> 1) * 8, odd = g & 1;
#pragma unroll
  for (int s = 0; s < SPW; ++s) {
    const int tk = (4 * s + wave) * 16 + j;
    unsigned kx[4][2], vx[4][2];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      kx[dt][0] = pack_lp2(dk[s][dt][0] * a.scale, dk[s][dt][1] * a.scale);
      kx[dt][1] = pack_lp2(dk[s][dt][2] * a.scale, dk[s][dt][3] * a.scale);
      vx[dt][0] = pack_lp2(dv[s][dt][0], dv[s][dt][1]);
      vx[dt][1] = pack_lp2(dv[s][dt][2], dv[s][dt][3]);
    }
#pragma unroll
    for (int dt = 0; dt < 4; dt += 2)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const auto rk = __builtin_amdgcn_permlane16_swap(kx[dt][hf], kx[dt + 1][hf], false, false);
        kx[dt][hf] = rk[0]; kx[dt + 1][hf] = rk[1];
        const auto rv = __builtin_amdgcn_permlane16_swap(vx[dt][hf], vx[dt + 1][hf], false, false);
        vx[dt][hf] = rv[0]; vx[dt + 1][hf] = rv[1];
      }
    if (tk < N) {
      lp_t* gk = a.dqkv + tok_row(a, b, tk) * a.lddq + a.D + h * HD + col16;
      lp_t* gv = gk + a.D;
#pragma unroll
      for (int dt = 0; dt < 4; dt += 2) {
        const int c = (dt + odd) * 16;
        *(u32x4_t*)(gk + c) = (u32x4_t){kx[dt][0], kx[dt][1], kx[dt + 1][0], kx[dt + 1][1]};
        *(u32x4_t*)(gv + c) = (u32x4_t){vx[dt][0], vx[dt][1], vx[dt + 1][0], vx[dt + 1][1]};
      }
    }
    __builtin_amdgcn_sched_barrier(0);       // one strip's 32 accumulator values in VGPRs at a time
  }
  B1_X(5);
}

}  // namespace

bool simvg_attn_bwd_onepass(const AttnArgs& a, hipStream_t stream) {
  const int N = a.Nv + a.Nt;
  if ((N + 15) / 16 != 27 || a.Nv / 16 < 24 || !a.delta) return false;
  constexpr int SHM1P = 28 * 16 * ROWB + 3 * 8192 + 2 * 96 * 4 + 4 * (7 * 16 + 16) * B1_SROW + 4 * 32 * B1_DQLD * 4;
  static bool once = hipFuncSetAttribute((const void*)attn_bwd_one_kernel<27>, hipFuncAttributeMaxDynamicSharedMemorySize, SHM1P) == hipSuccess;
  (void)once;
  hipLaunchKernelGGL((attn_bwd_one_kernel<27>), dim3(a.B * a.H), dim3(256), SHM1P, stream, a);
  return true;
}
