// Encoder self-attention forward for training-size launches (B * H >= number of CUs), gfx950: PERSISTENT workgroups, K / V
// STREAMED through an LDS-DMA ring, online softmax.  Same contract as attn_fwd_t_kernel (attention.hip; reference:
// torchscale MultiheadAttention called at beit3_base.py:137-145): out = softmax(q k^T * scale + key mask) v per
// (sample, head), natural-log LSE saved for the backward, rows modality-major.
//
// Why not the resident kernel (K and V of a head in 116 KB of LDS, one workgroup per CU): its per-head prologue (112 KB
// from HBM) cannot overlap the previous head's compute, a wave owns 16 queries so every 1-KiB K / V fragment read feeds
// ONE pair of MFMAs (LDS array time ~ MFMA time), and 27 query strips over 12 waves leave the last round three quarters
// empty: 66 us per launch against an MFMA floor of 17 us and an HBM floor of 26 us (profiles/r02_*).  Here:
//   * one workgroup per CU walks heads blockIdx.x, + gridDim.x, ...; K / V arrive in 64-key blocks (16 KiB: 8 + 8
//     `global_load_lds` wave-instructions, one of each per wave) through a 3-stage ring that runs two blocks ahead ACROSS
//     head boundaries, the next head's Q (56 KiB) is fetched under the current head's blocks 1..3 -- after the first head
//     no load is exposed; waits are counted (`s_waitcnt vmcnt(n)`: the epilogue's stores are entries of the same FIFO);
//   * a wave owns 3 or 4 query tiles of 16 (27 tiles = 3 x 4 + 5 x 3 over 8 waves; waves w and w + 4 share a SIMD: 7, 7, 7, 6
//     tiles per SIMD), so a K / V fragment read feeds 3-4 MFMA pairs and no tile slot is padding;
//   * scores of a 64-key block live in registers (64 VGPRs at 4 tiles); running maximum per query with two
//     v_permlane swaps (no LDS crossbar op: hipcc would drain the DMA ring in front of it), lane-partial row sums reduced
//     once per head; P is rounded to the 16-bit format and fed back as the MFMA B operand like in the resident kernel;
//   * every LDS read is inline asm with counted lgkmcnt (hipcc puts `s_waitcnt vmcnt(0)` in front of a compiler-visible LDS
//     read that follows an LDS-DMA, common.h), the kernel must not spill (a scratch access would be an uncounted FIFO entry:
//     the build fails the test that reads the kernel's private segment size).
// Key masking: keys >= N and padded text keys can only sit in the LAST 64-key block (host check); its 64 pad bytes are
// fetched by one 4-byte LDS-DMA per lane at the head's first step and turned into a wave-uniform 64-bit mask by a ballot.
#include "attention.h"

namespace {

constexpr int SW = 8;                       // waves per workgroup
constexpr int SKT = 4;                      // key tiles of 16 per ring stage
constexpr int SKR = SKT * 16;               // 64 key rows
constexpr int SSTG = 2 * SKR * ROWB;        // 16 KiB: K rows, then V rows
constexpr int SNST = 3;                     // ring depth
constexpr int SQI = 7;                      // Q: wave-instructions per wave (8 rows each)
constexpr int SQROWS = SW * SQI * 8;        // 448 rows
constexpr int SQBYTES = SQROWS * ROWB;      // 56 KiB
constexpr int SPADB = SW * 256;             // one 64-dword slot per wave for the pad bytes of the masked block

#define VM_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
// -DSIMVG_STREAM_PROFILE (tools/dev/attn_stream_profile.py): workgroup 0 records s_memtime at the phase boundaries of every
// step into AttnArgs::delta (unused by the forward): [wave][step][8] 64-bit ticks
#ifdef SIMVG_STREAM_PROFILE
constexpr int PROF_BYTES = 8 * 64 * 8 * 8;                  // [wave][step < 64][slot < 8] ticks, kept in LDS until the end:
#define PROF(slot)                                          /* a global store would be an uncounted entry of the vmcnt FIFO */ \
  do {                                                                                                              \
    if (lane == 0 && pstep < 64) {                                                                                  \
      const unsigned long long t__ = __builtin_amdgcn_s_memtime();                                                  \
      asm volatile("ds_write_b64 %0, %1" ::"v"(profaddr + (unsigned)((pstep * 8 + (slot)) * 8)), "v"(t__) : "memory"); \
    }                                                                                                               \
  } while (0)
#else
constexpr int PROF_BYTES = 0;
#define PROF(slot) do {} while (0)
#endif
constexpr int STREAM_LDS = SNST * SSTG + SQBYTES + SPADB + PROF_BYTES;
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
// the block loop and keep (or spill) them for the whole head.
__device__ __forceinline__ void dma_rows8(__amdgpu_buffer_rsrc_t rs, int rowbytes, RowMap rm, int Nv, int N, int t0, int colbytes,
                                          char* lds, int lane) {
  int t = t0 + (lane >> 3);
  t = t < N ? t : N - 1;
  const int row = t + (t < Nv ? rm.vrow0 : rm.trow0);
  const int slot = (lane & 7) ^ ((lane >> 3) & 6);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(lds), 16, row * rowbytes + slot * 16, colbytes, 0, 0);
}

// max / sum over the four 16-lane rows of a wave (the lanes that hold the same query): two swaps, no LDS crossbar
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

// The whole persistent loop of one wave that owns QT query tiles starting at tile qt0.  Every instantiation executes the
// same barriers (the workgroup mixes QT = base and base + 1).
template <int QT, bool LSE>
__device__ __forceinline__ void stream_fwd(const AttnArgs& a, char* smem, const int qt0) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int N = a.Nv + a.Nt;
  const int nb = (N + SKR - 1) / SKR;
  const int BH = a.B * a.H;
  const int nh = (BH - 1 - (int)blockIdx.x) / (int)gridDim.x + 1;      // heads of this workgroup
  char* ring = smem;
  char* ldsQ = smem + SNST * SSTG;
  char* padslot = ldsQ + SQBYTES + wave * 256;
  constexpr int NS = 4 * QT + (LSE ? 1 : 0);                           // global stores per head and wave
  const float sc2 = a.scale * 1.44269504088896340736f;
  const int rowbytes = a.ld * 2;
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
      va[dt] = (unsigned)(SKR * ROWB + ra * ROWB + ((slot ^ (ra & 6)) << 4) + (j & 1) * 8);
    }
  }
  const unsigned ring0 = lds_addr(ring), q0addr = lds_addr(ldsQ) + (unsigned)qt0 * 16 * ROWB, padaddr = lds_addr(padslot) + lane * 4;

  auto head_of = [&](int i) {
    const int hid = (int)blockIdx.x + i * (int)gridDim.x;
    HeadId r; r.b = hid / a.H; r.h = hid - r.b * a.H; return r;
  };
  auto issue_block = [&](HeadId hd, int kb, int stage) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    char* st = ring + stage * SSTG;
    const RowMap rm = row_map(a, hd.b);
    dma_rows8(rs_qkv, rowbytes, rm, a.Nv, N, kb * SKR + wave * 8, (a.D + hd.h * HD) * 2, st + wave * 1024, ln);
    dma_rows8(rs_qkv, rowbytes, rm, a.Nv, N, kb * SKR + wave * 8, (2 * a.D + hd.h * HD) * 2, st + SKR * ROWB + wave * 1024, ln);
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
  // pad bytes of the keys (nb - 1) * 64 + lane, as the aligned dword that holds the byte (4 B per lane into this wave's slot)
  auto pad_index = [&](HeadId hd, int ln) {
    int k = (nb - 1) * SKR + ln - a.Nv;
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

  // ---- prologue: Q of the first head, blocks 0 and 1
  HeadId cur = head_of(0);
  issue_q(cur);
  issue_block(cur, 0, 0);
  issue_block(cur, 1, 1);
  // the issue cursor runs two blocks ahead of the compute cursor
  int ii = 0, ikb = 2, istage = 2;
  HeadId ihd = cur;
  int stage = 0;
  int pstep = 0;
  const unsigned profaddr = lds_addr(ldsQ + SQBYTES + SPADB) + (unsigned)wave * 64 * 8 * 8;
  (void)pstep; (void)profaddr;

  for (int i = 0; i < nh; ++i) {
    cur = head_of(i);
    const HeadId nxt = head_of(i + 1 < nh ? i + 1 : i);               // past the end: reload the same head (never read)
    lpx8_t q[QT][2];
    f32x4_t o[QT][4];
    float m[QT], l[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      m[qt] = -INFINITY; l[qt] = 0.f;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o[qt][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }

#pragma unroll 1
    for (int kb = 0; kb < nb; ++kb) {
      // ---- this wave's pieces of block (i, kb) have landed: the count is the number of FIFO entries issued after them
      // (per step 2 block pieces; 1 pad piece before the block at a head's step 0; 7 Q pieces after the block at step 1;
      // NS stores at the end of a head)
      PROF(0);
      if (kb == 0) { if (i == 0) VM_WAIT(2); else VM_WAIT(2 + NS); }
      else if (kb == 1) { if (i == 0) VM_WAIT(3); else VM_WAIT(3 + NS); }
      else if (kb <= 3) VM_WAIT(2 + SQI);
      else VM_WAIT(2);
      PROF(1);
      __builtin_amdgcn_s_barrier();                                    // ... everyone's have, and everyone left the stage refilled below
      PROF(2);
      if (kb == 0) issue_pad(cur);
      issue_block(ihd, ikb, istage);
      if (kb == 1) issue_q(nxt);                                       // every wave read its Q fragments before barrier(1)
      if (++ikb == nb) { ikb = 0; if (ii + 1 < nh) ++ii; ihd = head_of(ii); }
      istage = istage + 1 == SNST ? 0 : istage + 1;
      __builtin_amdgcn_sched_barrier(0);
      PROF(3);

      const unsigned sbase = ring0 + (unsigned)stage * SSTG;
      stage = stage + 1 == SNST ? 0 : stage + 1;
      // ---- [Q fragments at a head's first step,] K fragments of key tiles 0, 1
      if (kb == 0) {
        u32x4_t qf[QT][2];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          qf[qt][0] = lds_b128_asm<0>(q0addr + qt * 16 * ROWB + ka0);
          qf[qt][1] = lds_b128_asm<0>(q0addr + qt * 16 * ROWB + ka1);
        }
        LDS_WAIT(0);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) { q[qt][0] = __builtin_bit_cast(lpx8_t, qf[qt][0]); q[qt][1] = __builtin_bit_cast(lpx8_t, qf[qt][1]); }
      }
      f32x4_t s[SKT][QT];
      {
        u32x4_t kf[2][2], kh[2][2];
        kf[0][0] = lds_b128_asm<0 * 2048>(sbase + ka0); kf[0][1] = lds_b128_asm<0 * 2048>(sbase + ka1);
        kf[1][0] = lds_b128_asm<1 * 2048>(sbase + ka0); kf[1][1] = lds_b128_asm<1 * 2048>(sbase + ka1);
        LDS_WAIT(0);
        kh[0][0] = lds_b128_asm<2 * 2048>(sbase + ka0); kh[0][1] = lds_b128_asm<2 * 2048>(sbase + ka1);
        kh[1][0] = lds_b128_asm<3 * 2048>(sbase + ka0); kh[1][1] = lds_b128_asm<3 * 2048>(sbase + ka1);
        // ---- S^T = K Q^T: 16 keys x 16 queries per MFMA pair; first halves of a tile pair, then the dependent second halves
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
        LDS_WAIT(0);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int qt = 0; qt < QT; ++qt)
            s[2 + kt][qt] = mfma_lp(__builtin_bit_cast(lpx8_t, kh[kt][0]), q[qt][0], (f32x4_t){0.f, 0.f, 0.f, 0.f});
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int qt = 0; qt < QT; ++qt)
            s[2 + kt][qt] = mfma_lp(__builtin_bit_cast(lpx8_t, kh[kt][1]), q[qt][1], s[2 + kt][qt]);
      }
      __builtin_amdgcn_sched_barrier(0);
      PROF(4);
      // V^T fragments of keys 0..31: they land under the softmax arithmetic
      u32x2_t vf[4][2];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        vf[dt][0] = lds_tr16_asm<0>(sbase + va[dt]);
        vf[dt][1] = lds_tr16_asm<2048>(sbase + va[dt]);
      }
      if (kb == nb - 1) {
        // key mask of this block: bit i <-> key kb * 64 + i
        const unsigned w = lds_b32_asm(padaddr);
        LDS_WAIT(0);
        const int key = kb * SKR + lane;
        const unsigned byte = (w >> (8 * (pad_index(cur, lane) & 3))) & 0xffu;
        const bool masked = key >= N || (a.pad != nullptr && key >= a.Nv && byte != 0);
        const unsigned long long mask = __ballot(masked);
#pragma unroll
        for (int kt = 0; kt < SKT; ++kt) {
          const unsigned m16 = (unsigned)(mask >> (kt * 16)) >> (4 * g);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float bias = ((m16 >> r) & 1u) ? -INFINITY : 0.f;
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) s[kt][qt][r] += bias;
          }
        }
      }
      // ---- online softmax (raw-score maximum; exp2 argument = one fma), P packed as the B operand of the PV product
      union PF { lpx8_t v; unsigned u[4]; };
      PF pf[QT][2];
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < SKT; ++kt)
          mx = fmaxf(mx, fmaxf(fmaxf(s[kt][qt][0], s[kt][qt][1]), fmaxf(s[kt][qt][2], s[kt][qt][3])));
        mx = row4_max(mx);
        const float m_new = fmaxf(m[qt], mx);
        const float m_use = m_new == -INFINITY ? 0.f : m_new;          // nothing unmasked so far: p = 0, nothing to rescale
        const float alpha = __builtin_amdgcn_exp2f((m[qt] - m_use) * sc2);
        m[qt] = m_new;
        const float mxs = m_use * sc2;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < SKT; ++kt) {
          float p[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) { p[r] = __builtin_amdgcn_exp2f(fmaf(s[kt][qt][r], sc2, -mxs)); sum += p[r]; }
          pf[qt][kt >> 1].u[(kt & 1) * 2] = pack_lp2_raw(p[0], p[1]);
          pf[qt][kt >> 1].u[(kt & 1) * 2 + 1] = pack_lp2_raw(p[2], p[3]);
        }
        l[qt] = fmaf(l[qt], alpha, sum);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int r = 0; r < 4; ++r) o[qt][dt][r] *= alpha;
      }
      __builtin_amdgcn_sched_barrier(0);
      PROF(5);
      // ---- O^T += V^T P^T, keys 0..31, then keys 32..63 (their fragments are read under the first half's MFMAs)
      LDS_WAIT(0);
      lpx8_t vfr[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) { lds_pin(vf[dt][0], vf[dt][1]); vfr[dt] = frag8(vf[dt][0], vf[dt][1]); }
      u32x2_t vg[4][2];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        vg[dt][0] = lds_tr16_asm<4096>(sbase + va[dt]);
        vg[dt][1] = lds_tr16_asm<4096 + 2048>(sbase + va[dt]);
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) o[qt][dt] = mfma_lp(vfr[dt], pf[qt][0].v, o[qt][dt]);
      LDS_WAIT(0);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) { lds_pin(vg[dt][0], vg[dt][1]); vfr[dt] = frag8(vg[dt][0], vg[dt][1]); }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) o[qt][dt] = mfma_lp(vfr[dt], pf[qt][1].v, o[qt][dt]);
      __builtin_amdgcn_sched_barrier(0);
      PROF(6);
      ++pstep;
    }

    // ---- head epilogue: 4 QT stores of 8 B per lane (+ LSE); every tile of a wave holds at least one real query
    const RowMap rm = row_map(a, cur.b);
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const int tq = (qt0 + qt) * 16 + j;
      const float sum = row4_sum(l[qt]);
      const float inv = __builtin_amdgcn_rcpf(sum);                    // 1 ulp; the products are rounded to 11 bits below
      if (tq < N) {
        const int row = tq + (tq < a.Nv ? rm.vrow0 : rm.trow0);
        lp_t* op = a.out + (long)row * a.ldo + cur.h * HD + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          *(u32x2_t*)(op + dt * 16) = (u32x2_t){pack_lp2(o[qt][dt][0] * inv, o[qt][dt][1] * inv),
                                               pack_lp2(o[qt][dt][2] * inv, o[qt][dt][3] * inv)};
        if (LSE && g == 0)
          a.lse[(long)(cur.b * a.H + cur.h) * N + tq] = (m[qt] * sc2 + __log2f(sum)) * 0.69314718055994530942f;   // natural log
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  VM_WAIT(0);                                                          // the run-ahead loads of the last two steps write this LDS
#ifdef SIMVG_STREAM_PROFILE
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (blockIdx.x == 0 && a.delta) {
    const unsigned long long* src = (const unsigned long long*)(ldsQ + SQBYTES + SPADB) + wave * 64 * 8;
    for (int e = lane; e < 64 * 8; e += 64) ((unsigned long long*)a.delta)[wave * 64 * 8 + e] = src[e];
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
  const int N = a.Nv + a.Nt, ntiles = (N + 15) / 16, nb = (N + SKR - 1) / SKR;
  static int ncu = [] { hipDeviceProp_t p; int d = 0; hipGetDevice(&d); hipGetDeviceProperties(&p, d); return p.multiProcessorCount; }();
  if (getenv("SIMVG_ATTN_RESIDENT")) return false;
  if (a.B * a.H < ncu) return false;                                   // small launches: the resident kernel with its query split
  if (ntiles < 3 * SW || ntiles > 4 * SW || ntiles > SQROWS / 16) return false;   // 3 or 4 query tiles per wave
  if (nb < 4 || a.Nv / SKR != nb - 1) return false;                    // masked keys only in the last block
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
