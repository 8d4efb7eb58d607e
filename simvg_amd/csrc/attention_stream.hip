// Encoder self-attention forward for training-size launches (B * H >= number of CUs), gfx950: PERSISTENT workgroups, K / V
// STREAMED through an LDS-DMA ring, online softmax, the two waves of every SIMD in complementary phases.  Same contract as
// attn_fwd_t_kernel (attention.hip; reference: torchscale MultiheadAttention called at beit3_base.py:137-145):
// out = softmax(q k^T * scale + key mask) v per (sample, head), natural-log LSE saved for the backward, rows modality-major.
//
// Why not the resident kernel (K and V of a head in 116 KB of LDS, one workgroup per CU): its per-head prologue (112 KB
// from HBM) cannot overlap the previous head's compute, a wave owns 16 queries so every 1-KiB K / V fragment read feeds
// ONE pair of MFMAs (LDS array time ~ MFMA time), and 27 query strips over 12 waves leave the last round three quarters
// empty: 66 us per launch against an MFMA floor of 17 us and an HBM floor of 26 us (profiles/r02_*).  Here:
//   * one workgroup per CU walks heads blockIdx.x, + gridDim.x, ...; K / V arrive in UNITS of 32 keys (8 KiB = 8
//     `buffer_load ... lds` wave-instructions, one per wave) through an 8-stage ring that runs 6 units ahead ACROSS head
//     boundaries; the next head's Q (56 KiB) is fetched during the current head -- after the first head no load is
//     exposed; waits are counted (`s_waitcnt vmcnt(n)`: the epilogue's stores are entries of the same FIFO);
//   * a wave owns 3 or 4 query tiles of 16 (27 tiles = 3 x 4 + 5 x 3 over 8 waves; waves w and w + 4 share a SIMD: 7, 7, 7, 6
//     tiles per SIMD), so a K / V fragment read feeds 3-4 MFMA pairs and no tile slot is padding;
//   * a wave alternates an MFMA phase (P V of the previous unit, then Q K^T of the next: 2 x 4 QT MFMAs) and a VALU phase
//     (softmax of that unit), with one workgroup barrier in front of each; waves 4..7 start ONE PHASE LATER than waves 0..3
//     (one extra barrier up front), so on every SIMD one wave is in its matrix phase while the other is in its vector phase.
//     (First version, all eight waves in the same phase behind one barrier per 64 keys: both waves of a SIMD did Q K^T,
//     then both the softmax, then both P V -- 6000 cycles per 64 keys where the MFMAs need 1800; tools/dev/attn_stream_profile.py.)
//   * softmax with a DEFERRED maximum: the running maximum of a query moves only when some score exceeds it by more than 8
//     (log2 domain; P <= 256 fits the 16-bit format, sums and the accumulator are fp32, the result is the same quotient),
//     a wave-uniform branch; otherwise a unit costs an exp2 per score, a packed fma, a pack, a max3 and a dot2 (the row
//     sum, over the rounded P) per two;
//   * every LDS read is inline asm with counted lgkmcnt (hipcc puts `s_waitcnt vmcnt(0)` in front of a compiler-visible LDS
//     read that follows an LDS-DMA, common.h); cross-lane reductions are v_permlane swaps (no LDS crossbar); the kernel must
//     not spill (a scratch access would be an uncounted FIFO entry: the host code refuses the kernel if hipFuncGetAttributes
//     reports a private segment).
// Key masking: keys >= N and padded text keys can only sit in the LAST 64 keys (host check); their 64 pad bytes are fetched
// by one 4-byte LDS-DMA per lane and turned into a wave-uniform 64-bit mask by a ballot.
#include "attention.h"

namespace {

constexpr int SW = 8;                       // waves per workgroup
constexpr int UK = 32;                      // keys per unit (ring stage)
constexpr int USTG = 2 * UK * ROWB;         // 8 KiB: K rows, then V rows
#ifndef SIMVG_STREAM_NSTG
#define SIMVG_STREAM_NSTG 8
#endif
constexpr int NSTG = SIMVG_STREAM_NSTG;     // ring depth in units
constexpr int SQI = 7;                      // Q: wave-instructions per wave (8 rows each)
constexpr int SQROWS = SW * SQI * 8;        // 448 rows
constexpr int SQBYTES = SQROWS * ROWB;      // 56 KiB
constexpr int SPADB = SW * 256;             // one 64-dword slot per wave for the pad bytes of the last 64 keys
constexpr float DEFER_THR = 8.f;            // log2 units

#define VM_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
// -DSIMVG_STREAM_PROFILE (tools/dev/attn_stream_profile.py): every wave of workgroup 0 records s_memtime (shader cycles) at
// its phase boundaries; kept in LDS until the end (a global store would be an uncounted entry of the vmcnt FIFO)
#ifdef SIMVG_STREAM_PROFILE
constexpr int PROF_STEPS = 64, PROF_SLOTS = 4;
constexpr int PROF_BYTES = SW * PROF_STEPS * PROF_SLOTS * 8;
#define PROF(slot)                                                                                                      \
  do {                                                                                                                  \
    if (lane == 0 && w < PROF_STEPS) {                                                                                  \
      const unsigned long long t__ = __builtin_amdgcn_s_memtime();                                                      \
      asm volatile("ds_write_b64 %0, %1" ::"v"(profaddr + (unsigned)((w * PROF_SLOTS + (slot)) * 8)), "v"(t__) : "memory"); \
    }                                                                                                                   \
  } while (0)
#else
constexpr int PROF_BYTES = 0;
#define PROF(slot) do {} while (0)
#endif
constexpr int STREAM_LDS = NSTG * USTG + SQBYTES + SPADB + PROF_BYTES;
#define LDS_WAIT(n)                                            \
  do {                                                         \
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory"); \
    __builtin_amdgcn_sched_barrier(0);                         \
  } while (0)

__device__ __forceinline__ unsigned lds_b32_asm(unsigned addr) {
  unsigned v;
  asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}

// Row addressing without 64-bit lane arithmetic: one buffer descriptor over the whole [M, ld] qkv matrix (the host checks
// that it is smaller than 2 GiB); token t of sample b is matrix row  t + (t < Nv ? b Nv : B Nv + b Nt - Nv).
struct RowMap { int vrow0, trow0; };
__device__ __forceinline__ RowMap row_map(const AttnArgs& a, int b) {
  RowMap r; r.vrow0 = b * a.Nv; r.trow0 = a.B * a.Nv + b * a.Nt - a.Nv; return r;
}

// one 1-KiB wave-instruction: rows t0 + (lane >> 3) (8 rows x 128 B) of one head's operand -> LDS, lane-linear image; the
// 16-B slot swizzle of attention.h (physical = slot ^ (row & 6)) is applied on the SOURCE address.  t0 % 8 == 0.
// `lane` arrives laundered (see issue_*): hipcc would otherwise hoist every lane-constant piece of these addresses out of
// the unit loop and keep (or spill) them for the whole head.
__device__ __forceinline__ void dma_rows8(__amdgpu_buffer_rsrc_t rs, int rowbytes, RowMap rm, int Nv, int N, int t0, int colbytes,
                                          char* lds, int lane) {
  int t = t0 + (lane >> 3);
  t = t < N ? t : N - 1;
  const int row = t + (t < Nv ? rm.vrow0 : rm.trow0);
  const int slot = (lane & 7) ^ ((lane >> 3) & 6);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(lds), 16, row * rowbytes + slot * 16, colbytes, 0, 0);
}

// max / sum over the four 16-lane rows of a wave (the lanes that hold the same query): two swaps, no LDS crossbar.
// v_permlane16_swap exchanges the odd rows of its first operand with the even rows of its second, v_permlane32_swap the upper
// half of the first with the lower half of the second: with both operands = x the two results hold x of both partners.
__device__ __forceinline__ float row4_max(float x) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float row4_sum(float x) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

struct HeadId { int b, h; };
union PF { lpx8_t v; unsigned u[4]; };

// The whole persistent loop of one wave that owns QT query tiles starting at tile qt0.  Every instantiation executes the
// same barriers (the workgroup mixes QT = base and base + 1).
//
// Schedule.  U = units per head, global unit w = i U + u; every wave runs  M(0) V(0) M(1) V(1) ...  with ONE workgroup barrier
// per unit: waves 0..3 in front of M(w), waves 4..7 in front of V(w) -- so between two barriers the early half does
// M(w) V(w) and the late half V(w) M(w + 1): matrix phase beside vector phase on every SIMD.
//   M(w): issue the DMA of unit w + NSTG - 3 into the stage of unit w - 3 (whose last readers waited for their fragments
//         before the previous barrier); at u == 2 also the pad bytes of this head and Q of the NEXT head (this head's Q
//         fragments were consumed before the previous barrier); wait for the fragments prefetched by V(w - 1);
//         Q K^T of unit w, then P V of unit w - 1.
//   V(w): wait (vmcnt) for this wave's piece of unit w + 2 -- published by the next barrier, first read at the end of
//         V(w + 1); at u == 0 the epilogue of the previous head; softmax of unit w; prefetch the fragments of M(w + 1):
//         K of unit w + 1, V^T of unit w [, Q of the next head at u == U - 1].
// FIFO entries issued after the piece of unit w + 2 when V(w) waits for it (program order is the same for both halves):
// the NSTG - 5 later units; the pad piece and the 7 Q pieces of M(iU + 2) while 2 <= u <= NSTG - 3; for i > 0 the QT LSE
// stores of V(iU - 1) while u <= 2 and the 2 QT output stores of V(iU) while 1 <= u <= 3.  (In the first head's first units the true counts are larger: the waits are stricter
// than necessary there.)
template <int QT, bool LSE>
__device__ __forceinline__ void stream_fwd(const AttnArgs& a, char* smem, const int qt0) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int N = a.Nv + a.Nt;
  const int U = ((N + 63) >> 6) * 2;                                   // units per head (whole 64-key blocks)
  const int BH = a.B * a.H;
  const int nh = (BH - 1 - (int)blockIdx.x) / (int)gridDim.x + 1;      // heads of this workgroup
  char* ring = smem;
  char* ldsQ = smem + NSTG * USTG;
  char* padslot = ldsQ + SQBYTES + wave * 256;
  constexpr int NO = 2 * QT, NL = LSE ? QT : 0;                        // global stores per head and wave: output, LSE
  static_assert(NSTG >= 8 && NSTG - 5 + 8 + 3 * 4 <= 63, "vmcnt table: u == 1 < 2 <= NSTG - 6");
  const float sc2 = a.scale * 1.44269504088896340736f;
  const int rowbytes = a.ld * 2;
#if defined(SIMVG_STREAM_PROFILE) || defined(SIMVG_STREAM_ABLATE)
  const int abl = a.lddo;            // ablation switches of the development builds (timing only, results are wrong): see the host code
#else
  constexpr int abl = 0;
#endif
  const __amdgpu_buffer_rsrc_t rs_qkv =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.qkv, 0, (int)((long)a.B * N * rowbytes), 0x00020000);

  // lane-constant LDS byte offsets (relative to a stage / the Q buffer)
  const unsigned ka0 = (unsigned)(j * ROWB + ((g ^ (j & 6)) << 4));    // K-major fragment of row j, 16-B slot g
  const unsigned ka1 = ka0 ^ 64u;                                      // slot 4 + g
  unsigned va[4];                                                      // transposed V fragment, column block dt
  {
    const int ra = 4 * g + (j >> 2);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const int slot = dt * 2 + ((j & 3) >> 1);
      va[dt] = (unsigned)(UK * ROWB + ra * ROWB + ((slot ^ (ra & 6)) << 4) + (j & 1) * 8);
    }
  }
  const unsigned ring0 = lds_addr(ring), q0addr = lds_addr(ldsQ) + (unsigned)qt0 * 16 * ROWB, padaddr = lds_addr(padslot) + lane * 4;
#ifdef SIMVG_STREAM_PROFILE
  const unsigned profaddr = lds_addr(ldsQ + SQBYTES + SPADB) + (unsigned)wave * PROF_STEPS * PROF_SLOTS * 8;
#endif

  auto head_of = [&](int i) {
    const int hid = (int)blockIdx.x + i * (int)gridDim.x;
    HeadId r; r.b = hid / a.H; r.h = hid - r.b * a.H; return r;
  };
  // waves 0..3: K rows, waves 4..7: V rows of the unit, 8 rows each
  auto issue_unit = [&](HeadId hd, int uu, int stage) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int part = wave & 3, isv = wave >> 2;
    dma_rows8(rs_qkv, rowbytes, row_map(a, hd.b), a.Nv, N, uu * UK + part * 8, ((1 + isv) * a.D + hd.h * HD) * 2,
              ring + stage * USTG + isv * (UK * ROWB) + part * 1024, ln);
  };
  auto issue_q = [&](HeadId hd) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const RowMap rm = row_map(a, hd.b);
#pragma unroll
    for (int i = 0; i < SQI; ++i) {
      const int inst = wave * SQI + i;
      dma_rows8(rs_qkv, rowbytes, rm, a.Nv, N, inst * 8, hd.h * HD * 2, ldsQ + inst * 1024, ln);
    }
  };
  // pad bytes of the keys (U - 2) * 32 + lane (the last 64), as the aligned dword that holds the byte
  auto pad_index = [&](HeadId hd, int ln) {
    int k = (U - 2) * UK + ln - a.Nv;
    k = k < 0 ? 0 : (k >= a.Nt ? a.Nt - 1 : k);
    return hd.b * a.Nt + k;
  };
  auto issue_pad = [&](HeadId hd) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const __amdgpu_buffer_rsrc_t rs_pad =
        __builtin_amdgcn_make_buffer_rsrc(a.pad ? (void*)a.pad : (void*)a.qkv, 0, a.pad ? a.B * a.Nt : 256, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_pad, LDS_PTR(padslot), 4, a.pad ? (pad_index(hd, ln) & ~3) : 0, 0, 0, 0);
  };

  lpx8_t q[QT][2];
  f32x4_t o[QT][4], s[2][QT];
  // ref[qt] = the query's softmax reference point (log2 domain): p = exp2(score * scale * log2(e) - ref), one packed fp32 fma
  // per two scores.  It follows the running maximum lazily: thr[qt] = the raw score above which it has to move (DEFER_THR).
  // (Scaling Q once per head instead and seeding the MFMA accumulators with -ref saves that fma, but rounds Q a second time:
  // 3x the output error, 10x at large logits -- measured, dropped.)  l = lane-partial row sum.
  float ref[QT], l[QT];
  bool first = true;                                                   // the next softmax is the head's first: no reference point yet
  PF pf[QT];
  u32x4_t kf[2][2];
  u32x2_t vf[4][2];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    ref[qt] = 0.f; l[qt] = 0.f;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[qt][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }

  auto read_q = [&]() {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      q[qt][0] = __builtin_bit_cast(lpx8_t, lds_b128_asm<0>(q0addr + qt * 16 * ROWB + ka0));
      q[qt][1] = __builtin_bit_cast(lpx8_t, lds_b128_asm<0>(q0addr + qt * 16 * ROWB + ka1));
    }
  };
  auto read_k = [&](int stage) {
    const unsigned sb = ring0 + (unsigned)stage * USTG;
    kf[0][0] = lds_b128_asm<0>(sb + ka0);    kf[0][1] = lds_b128_asm<0>(sb + ka1);
    kf[1][0] = lds_b128_asm<2048>(sb + ka0); kf[1][1] = lds_b128_asm<2048>(sb + ka1);
  };
  auto read_v = [&](int stage) {
    const unsigned sb = ring0 + (unsigned)stage * USTG;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      vf[dt][0] = lds_tr16_asm<0>(sb + va[dt]);
      vf[dt][1] = lds_tr16_asm<2048>(sb + va[dt]);
    }
  };
  // O^T += V^T P^T of one unit: keys 4g..4g+3 of its two key tiles are the MFMA's 8 k-slots of lane group g
  auto mfma_pv = [&]() {
    lpx8_t vfr[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { lds_pin(vf[dt][0], vf[dt][1]); vfr[dt] = frag8(vf[dt][0], vf[dt][1]); }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) o[qt][dt] = mfma_lp(vfr[dt], pf[qt].v, o[qt][dt]);
  };
  // S^T = K Q^T: 16 keys x 16 queries per MFMA pair; first halves of both tiles, then the dependent second halves
  auto mfma_qk = [&]() {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
        s[kt][qt] = mfma_lp(__builtin_bit_cast(lpx8_t, kf[kt][0]), q[qt][0], (f32x4_t){0.f, 0.f, 0.f, 0.f});
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
        s[kt][qt] = mfma_lp(__builtin_bit_cast(lpx8_t, kf[kt][1]), q[qt][1], s[kt][qt]);
  };
  // normalise and store one head's output: 2 QT stores of 16 B per lane (the 8-B pieces of lane rows g and g ^ 1 are
  // exchanged so that every lane owns 16 contiguous bytes of its query's row) + QT LSE stores
  auto epilogue = [&](HeadId hd) {
    const RowMap rm = row_map(a, hd.b);
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const int tq = (qt0 + qt) * 16 + j;
      const float sum = row4_sum(l[qt]);
      const float inv = __builtin_amdgcn_rcpf(sum);                    // 1 ulp; the products are rounded to 11 bits below
      unsigned lo[4], hi[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        lo[dt] = pack_lp2_raw(o[qt][dt][0] * inv, o[qt][dt][1] * inv);     // a convex combination of V rows: in range
        hi[dt] = pack_lp2_raw(o[qt][dt][2] * inv, o[qt][dt][3] * inv);
      }
      const int row = tq + (tq < a.Nv ? rm.vrow0 : rm.trow0);
      lp_t* op = a.out + (long)row * a.ldo + hd.h * HD;
#pragma unroll
      for (int dp = 0; dp < 2; ++dp) {
        const auto rl = __builtin_amdgcn_permlane16_swap(lo[2 * dp], lo[2 * dp + 1], false, false);
        const auto rh = __builtin_amdgcn_permlane16_swap(hi[2 * dp], hi[2 * dp + 1], false, false);
        // even g: own and g + 1's piece of column block 2 dp; odd g: g - 1's and own piece of column block 2 dp + 1
        const int col = (g & 1) ? (2 * dp + 1) * 16 + (g - 1) * 4 : 2 * dp * 16 + g * 4;
        if (tq < N) *(u32x4_t*)(op + col) = (u32x4_t){rl[0], rh[0], rl[1], rh[1]};
      }
    }
  };

  // ---- prologue: Q of the first head, units 0 .. NSTG - 1
  HeadId cur = head_of(0);
  issue_q(cur);
#pragma unroll 1
  for (int v = 0; v < NSTG; ++v) issue_unit(cur, v, v);                // U >= NSTG (host check)
  int ii = 0, iu = NSTG;                                               // issue cursor (head index, unit) of the next unit to fetch
  HeadId ihd = cur;
  if (iu == U) { iu = 0; ii = nh > 1 ? 1 : 0; ihd = head_of(ii); }
  VM_WAIT(NSTG - 2);                                                   // Q, units 0 and 1
  __builtin_amdgcn_s_barrier();
  read_q();
  read_k(0);
  const bool early = wave < 4;                                         // waves 0..3: barrier, M, V; waves 4..7: M, barrier, V

  int st = 0;                                                          // stage of unit w
  int i = 0, u = 0;
  const int V = nh * U;
#pragma unroll 1
  for (int w = 0; w < V; ++w) {
    // =========================== M(w)
    PROF(0);
    if (early && !(abl & 128)) __builtin_amdgcn_s_barrier();
    PROF(1);
    {
      if (w >= 3 && !(abl & 4)) {                                      // unit w + NSTG - 3 into the stage of unit w - 3
        int sti = st - 3; sti = sti < 0 ? sti + NSTG : sti;
        issue_unit(ihd, iu, sti);
        if (++iu == U) { iu = 0; if (ii + 1 < nh) ++ii; ihd = head_of(ii); }
      }
      if (u == 2) { issue_pad(cur); issue_q(head_of(i + 1 < nh ? i + 1 : i)); }   // past the end: reload (never read)
    }
    __builtin_amdgcn_sched_barrier(0);
    LDS_WAIT(0);                                                       // K fragments [and Q] read at the start of V(w - 1)
    if (w > 0 && !(abl & 32)) read_v(st == 0 ? NSTG - 1 : st - 1);     // V^T of unit w - 1: lands under the Q K^T MFMAs
    if (!(abl & 16)) __builtin_amdgcn_s_setprio(1);                    // the matrix phase goes first on its SIMD (-14 % launch time)
    if (!(abl & 2)) {
      if (u == 0) first = true;                                        // new head: no reference point yet
      mfma_qk();                                                       // first: the softmax that follows waits for these
      __builtin_amdgcn_sched_barrier(0);
      LDS_WAIT(0);
      if (w > 0) mfma_pv();
    }
    if (!(abl & 16)) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    // =========================== V(w)
    PROF(2);
    if (!early && !(abl & 128)) __builtin_amdgcn_s_barrier();
    PROF(3);
    if (abl & 8) __builtin_amdgcn_s_setprio(1);
    if (abl & 4) {}
    else if (u == 0) { if (i > 0) VM_WAIT(NSTG - 5 + NL); else VM_WAIT(NSTG - 5); }
    else if (u == 1) { if (i > 0) VM_WAIT(NSTG - 5 + NL + NO); else VM_WAIT(NSTG - 5); }
    else if (u <= NSTG - 6) { if (i > 0) VM_WAIT(NSTG - 5 + 8 + NL + NO); else VM_WAIT(NSTG - 5 + 8); }
    else if (u == NSTG - 5) { if (i > 0) VM_WAIT(NSTG - 5 + 8 + NO); else VM_WAIT(NSTG - 5 + 8); }
    else if (u <= NSTG - 3) VM_WAIT(NSTG - 5 + 8);
    else VM_WAIT(NSTG - 5);
    // ---- K fragments of M(w + 1) [and the next head's Q at u == U - 1]: they land under the softmax arithmetic
    {
      const int stn = st + 1 == NSTG ? 0 : st + 1;
      if (w + 1 < V && !(abl & 32)) {
        read_k(stn);
        if (u == U - 1) read_q();                                      // fetched since u == 2
      }
      st = stn;
    }
    if (u == 0 && i > 0) {
      if (!(abl & 256)) epilogue(head_of(i - 1));
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        l[qt] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[qt][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      }
    }
    if (u >= U - 2) {
      // key mask of the last 64 keys: bit k <-> key (U - 2) * 32 + k; this unit uses bits 32 (u - U + 2) ...
      const unsigned wd = lds_b32_asm(padaddr);
      LDS_WAIT(0);
      const int key = (U - 2) * UK + lane;
      const unsigned byte = (wd >> (8 * (pad_index(cur, lane) & 3))) & 0xffu;
      const bool masked = key >= N || (a.pad != nullptr && key >= a.Nv && byte != 0);
      const unsigned long long mask = __ballot(masked);
      const unsigned m32 = (unsigned)(u == U - 2 ? mask : mask >> 32);
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        const unsigned m4 = (m32 >> (kt * 16)) >> (4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float bias = ((m4 >> r) & 1u) ? -INFINITY : 0.f;
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) s[kt][qt][r] += bias;
        }
      }
    }
    // ---- softmax of unit w.  Deferred maximum: the reference moves only when some raw score of the wave exceeds its
    // query's threshold (a head's first unit: always)
    if (!(abl & 64)) {
      typedef float f32x2_t __attribute__((ext_vector_type(2)));
      float mxl[QT];
      bool need = first;
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        float mx = fmaxf(fmaxf(s[0][qt][0], s[0][qt][1]), s[0][qt][2]);
        mx = fmaxf(fmaxf(mx, s[0][qt][3]), s[1][qt][0]);
        mx = fmaxf(fmaxf(mx, s[1][qt][1]), s[1][qt][2]);
        mx = fmaxf(mx, s[1][qt][3]);
        mxl[qt] = mx;
        need |= fmaf(mx, sc2, -DEFER_THR) > ref[qt];
      }
      if (__ballot(need) != 0ull) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          // finite from a head's first unit on (vision keys are never masked)
          const float rm = row4_max(mxl[qt]) * sc2;
          const float nref = first ? rm : fmaxf(rm, ref[qt]);
          if (!first) {
            const float alpha = __builtin_amdgcn_exp2f(ref[qt] - nref);
            l[qt] *= alpha;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
              for (int r = 0; r < 4; ++r) o[qt][dt][r] *= alpha;
          }
          ref[qt] = nref;
        }
        first = false;
      }
      const f32x2_t sc2v = {sc2, sc2};
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        const f32x2_t nrefv = {-ref[qt], -ref[qt]};
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          const f32x2_t t01 = __builtin_elementwise_fma((f32x2_t){s[kt][qt][0], s[kt][qt][1]}, sc2v, nrefv);
          const f32x2_t t23 = __builtin_elementwise_fma((f32x2_t){s[kt][qt][2], s[kt][qt][3]}, sc2v, nrefv);
          float p[4];
          p[0] = (abl & 1) ? t01[0] : __builtin_amdgcn_exp2f(t01[0]);
          p[1] = (abl & 1) ? t01[1] : __builtin_amdgcn_exp2f(t01[1]);
          p[2] = (abl & 1) ? t23[0] : __builtin_amdgcn_exp2f(t23[0]);
          p[3] = (abl & 1) ? t23[1] : __builtin_amdgcn_exp2f(t23[1]);
          pf[qt].u[kt * 2] = pack_lp2_raw(p[0], p[1]);
          pf[qt].u[kt * 2 + 1] = pack_lp2_raw(p[2], p[3]);
        }
        // row sum of the ROUNDED probabilities (what the P V product uses)
#if SIMVG_LOWP_FORMAT == 1 && !defined(DBG_NO_DOT2)
        const hw_lpx2_t one2 = {(hw_lp_t)1.f, (hw_lp_t)1.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) l[qt] = __builtin_amdgcn_fdot2(__builtin_bit_cast(hw_lpx2_t, pf[qt].u[e]), one2, l[qt], false);
#else
#pragma unroll
        for (int e = 0; e < 4; ++e) { float a0, a1; unpack_lp2(pf[qt].u[e], a0, a1); l[qt] += a0 + a1; }
#endif
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (LSE && u == U - 1 && !(abl & 256)) {                           // the row sums are complete: this head's LSE (natural log)
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        const int tq = (qt0 + qt) * 16 + j;
        const float sum = row4_sum(l[qt]);
        if (g == 0 && tq < N) a.lse[(long)(cur.b * a.H + cur.h) * N + tq] = (__log2f(sum) + ref[qt]) * 0.69314718055994530942f;
      }
    }
    if (++u == U) { u = 0; ++i; cur = head_of(i < nh ? i : nh - 1); }
    if (abl & 8) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  }
  // =========================== the pending P V and the last head
  __builtin_amdgcn_s_barrier();
  read_v(st == 0 ? NSTG - 1 : st - 1);
  LDS_WAIT(0);
  mfma_pv();
  epilogue(head_of(nh - 1));
  VM_WAIT(0);                                                          // the run-ahead loads write this workgroup's LDS
#ifdef SIMVG_STREAM_PROFILE
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (blockIdx.x == 0 && a.delta) {
    const unsigned long long* src = (const unsigned long long*)(ldsQ + SQBYTES + SPADB) + wave * PROF_STEPS * PROF_SLOTS;
    for (int e = lane; e < PROF_STEPS * PROF_SLOTS; e += 64) ((unsigned long long*)a.delta)[wave * PROF_STEPS * PROF_SLOTS + e] = src[e];
  }
#endif
}

template <bool LSE>
__global__ __launch_bounds__(SW * 64, 2) void attn_fwd_stream_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ntiles = (a.Nv + a.Nt + 15) >> 4;
  const int base = ntiles / SW, rem = ntiles - base * SW;              // waves < rem own base + 1 tiles
  const int mine = base + (wave < rem ? 1 : 0);
  const int qt0 = wave * base + (wave < rem ? wave : rem);
  if (mine == 4) stream_fwd<4, LSE>(a, smem, qt0);
  else stream_fwd<3, LSE>(a, smem, qt0);
}

}  // namespace

#ifdef SIMVG_STREAM_PROFILE
static float* g_prof_buffer = nullptr;                                 // development builds only (tools/dev/attn_stream_profile.py)
extern "C" void simvg_stream_profile_buffer(void* p) { g_prof_buffer = (float*)p; }
#endif

bool simvg_attn_fwd_stream(const AttnArgs& a_in, hipStream_t stream) {
  AttnArgs a = a_in;
#ifdef SIMVG_STREAM_PROFILE
  a.delta = g_prof_buffer;
#endif
#if defined(SIMVG_STREAM_PROFILE) || defined(SIMVG_STREAM_ABLATE)
  // SIMVG_STREAM_ABL bits: 1 no exp2, 2 no MFMAs, 4 no unit DMA / vmcnt waits, 8 s_setprio 1 in the VALU phase, 16 in the MFMA
  // phase, 32 no fragment reads, 64 no softmax arithmetic, 128 no barriers in the unit loop, 256 no epilogue
  a.lddo = getenv("SIMVG_STREAM_ABL") ? atoi(getenv("SIMVG_STREAM_ABL")) : 0;
#endif
  const int N = a.Nv + a.Nt, ntiles = (N + 15) / 16, nb = (N + 63) / 64;
  static int ncu = [] { hipDeviceProp_t p; int d = 0; (void)hipGetDevice(&d); (void)hipGetDeviceProperties(&p, d); return p.multiProcessorCount; }();
  // Opt-in (SIMVG_ATTN_STREAM=1): at the end of round 3 this kernel is correct but, at 76 us per launch on the bench geometry,
  // still behind the resident kernel's 66 us (profiles/r03_sweeps.md: where its time goes and what was tried)
  const char* opt = getenv("SIMVG_ATTN_STREAM");
  if (!opt || opt[0] != '1' || getenv("SIMVG_ATTN_RESIDENT")) return false;
  // the counted vmcnt waits assume that the kernel issues no vector memory instruction of its own: refuse a build that spills
  static const bool no_scratch = [] {
    hipFuncAttributes fa, fb;
    return hipFuncGetAttributes(&fa, (const void*)attn_fwd_stream_kernel<true>) == hipSuccess &&
           hipFuncGetAttributes(&fb, (const void*)attn_fwd_stream_kernel<false>) == hipSuccess && fa.localSizeBytes == 0 && fb.localSizeBytes == 0;
  }();
  if (!no_scratch) return false;
  if (a.B * a.H < ncu) return false;                                   // small launches: the resident kernel with its query split
  if (ntiles < 3 * SW || ntiles > 4 * SW || ntiles > SQROWS / 16) return false;   // 3 or 4 query tiles per wave
  if (2 * nb < NSTG + 2 || a.Nv / 64 != nb - 1 || a.Nv < UK) return false;        // masked keys only in the last 64
  if (a.pad && ((uintptr_t)a.pad & 3) != 0) return false;
  if ((long)a.B * N * a.ld * 2 >= (1L << 31)) return false;             // 32-bit buffer offsets
  static bool once = hipFuncSetAttribute((const void*)attn_fwd_stream_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, STREAM_LDS) == hipSuccess &&
                     hipFuncSetAttribute((const void*)attn_fwd_stream_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, STREAM_LDS) == hipSuccess;
  (void)once;
  const int grid = ncu;
  if (a.lse) hipLaunchKernelGGL(attn_fwd_stream_kernel<true>, dim3(grid), dim3(SW * 64), STREAM_LDS, stream, a);
  else hipLaunchKernelGGL(attn_fwd_stream_kernel<false>, dim3(grid), dim3(SW * 64), STREAM_LDS, stream, a);
  return true;
}
