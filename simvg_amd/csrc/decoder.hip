// Decoder layers of the head as few launches (gfx950): one workgroup OWNS the query rows of one sample and walks the whole
// attention block of a DETR decoder layer -- self-attention, norm, cross-attention, norm (detrex BaseTransformerLayer as the
// reference configures it, simvg/models/heads/tgqs_kd_detr_head/transformer.py:93-131, called at :134-186 and at
// tgqs_kd_detr_head.py:391-399,425-428) -- while the layer's weights stream past it from L2; the FFN of the layer is split over
// its hidden units instead (every workgroup owns a slice of the hidden layer for ALL rows).
//
// Why this shape.  The head works on [B * num_queries, 256] rows (64 rows at the benchmark's batch): its 8 dependent stages per
// layer were 8+ launches whose time is a chain of memory latencies (5 - 15 us each, rocprof).  A workgroup that owns its rows
// needs no other workgroup between the stages, so a layer's attention block is ONE launch; the weights (1.5 MB fp32 per layer)
// are fetched by every workgroup from L2 at the per-CU stream rate, always several 16-byte loads per lane ahead of the MFMAs.
//
// Cross-attention without K / V projections.  With q the projected query of head h (scale folded in):
//     score[k] = q_h . (W_k,h s_k + b_k,h) = (W_k,h^T q_h) . s_k + const          (the constant cancels in the softmax)
//     out_h    = sum_k P'[k] (W_v,h s_k + b_v,h) = W_v,h (sum_k P'[k] s_k) + b_v,h sum_k P'[k]
// so the kernel contracts the 256-wide source rows s_k directly (the image memory in its 16-bit storage format + the sine
// position rows for the keys; the text rows + their 1-D positions for the TGQG layers): no [B*401, 512] fp32 K|V matrix
// (52 MB per layer at B = 64) is written, read back, or differentiated.  The gradient of the source rows comes out of the same
// workgroup (d s_k = sum_h,q dS[k] W_k,h^T q_h + P'[k] W_v,h^T do_h), accumulated over the layers in one fp32 buffer.
//
// All arithmetic fp32 (v_mfma_f32_16x16x4_f32 == an fmaf chain); softmax / LayerNorm as in the reference.
#include <string.h>

#include "common.h"

#ifdef DEC_TIMELINE      // development builds only (SIMVG_EXTRA_FLAGS=-DDEC_TIMELINE): workgroup 0's phase boundaries on the 100 MHz clock
__device__ unsigned long long dec_tl[64];
#define TL(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) dec_tl[i] = wall_clock64(); } while (0)
extern "C" int simvg_dec_timeline(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dec_tl), sizeof(dec_tl)); }
#else
#define TL(i)
#endif

namespace {

constexpr int DE = 256;        // embed_dim of the head (tgqs_kd_detr_head.py: embed_dim=256 in every config)
constexpr int DH = 8;          // heads
constexpr int DHD = 32;        // head dim
constexpr int DLD = DE + 4;    // LDS row stride of a [16][256] fp32 tile: 16 lanes reading one column of 16 rows hit 16 different bank groups
constexpr int DNW = 8;         // waves per workgroup
constexpr int DNT = DNW * 64;

__device__ __forceinline__ f32x4_t f4zero() { return (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ f32x4_t ld4(const float* p) { return *(const f32x4_t*)p; }
__device__ __forceinline__ f32x4_t mfma4(f32x4_t a, f32x4_t b, f32x4_t acc) {
  // four k values per lane: the 4 "k slots" of v_mfma_f32_16x16x4_f32 may carry any 4 k as long as A and B agree
#pragma unroll
  for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[i], acc, 0, 0, 0);
  return acc;
}

// 16-bit storage row -> 4 floats (8-byte load)
__device__ __forceinline__ f32x4_t ld4_lp(const lp_t* p) {
  const u32x2_t u = *(const u32x2_t*)p;
  float a, b, c, d;
  unpack_lp2(u[0], a, b);
  unpack_lp2(u[1], c, d);
  return (f32x4_t){a, b, c, d};
}

// The source rows of a cross-attention (keys AND values): 16-bit rows (the image memory) or fp32 rows (text rows; the image
// memory of the exact-fp32 mode); the keys get `kpos` rows added (sine positions), the values do not.
template <bool S16>       // (compile-time source format: a run-time branch around a load would blur the compiler's count of loads in flight)
struct SrcRows {
  const lp_t* s16; const float* s32; long ld;      // row r of this sample at (s16 | s32) + r * ld
  const float* kpos; long ldkp;                     // key_pos rows of this sample
  __device__ __forceinline__ f32x4_t val(int r, int c) const {
    if constexpr (S16) return ld4_lp(s16 + r * ld + c);
    else return ld4(s32 + r * ld + c);
  }
  __device__ __forceinline__ f32x4_t key(int r, int c) const { return val(r, c) + ld4(kpos + r * ldkp + c); }
};

// ---------------------------------------------------------------------------------------------------------------------
// P1: Y[r][n] = sum_k A(r, k) * W(n, k)  -- the contraction index is contiguous in W's rows.  One call = the n-tiles of ONE
// wave (16 output columns each), streamed as a flat sequence of (tile, 16-k step) pairs with U steps of W in flight.
//   bload(t, k)       -> the wave's B fragment of its t-th tile at contraction offset k: lane (j = lane & 15, g = lane >> 4)
//                        supplies W(n0(t) + j, k + 4g .. +3)
//   aload(t, k, row)  -> A(row, k + 4g .. +3)
//   epi(t, rt, acc)   -> acc[v] = Y[16 rt + 4g + v][n0(t) + j]
// RR = 0: v_mfma_f32_16x16x4_f32 on 16-row tiles (NRT of them).  RR = 1 (NRT = 1): only row 0 exists (num_queries
// = 1: a 16-row MFMA tile would spend 15/16 of its 32 cycles on padding, and the fp32 MFMA rate -- 256 FLOP per cycle and
// CU -- made the whole kernel MFMA-bound): each lane keeps the partial dot products of its quarter of k, the four quarters meet
// through two cross-lane adds at the end of the tile, and the result is handed to `epi` in the MFMA layout (rows >= RR zero).
// (the first U loads of a phase may be issued EARLY -- `stream_nt_prefetch` before the previous phase's barrier: W does not depend
// on the data, and a phase that starts with its loads already in flight does not pay a memory round trip before its first FMA;
// at 14 phases per kernel those round trips were most of the kernel's time)
template <int KS, int U, class BLoad>
__device__ __forceinline__ void stream_nt_prefetch(int ntiles, BLoad bload, f32x4_t (&bq)[U]) {
  const int total = ntiles * KS;
  if (total <= 0) return;
#pragma unroll
  for (int u = 0; u < U; ++u) { const int su = min(u, total - 1); bq[u] = bload(su / KS, (su % KS) * 16); }     // (unconditional: see gemv256)
}
template <int KS, int NRT, int U, int RR, class BLoad, class ALoad, class Epi>
__device__ __forceinline__ void stream_nt(int ntiles, BLoad bload, ALoad aload, Epi epi, f32x4_t (&bq)[U]) {
  static_assert(KS % U == 0 || U % KS == 0, "prefetch groups and tiles nest");
  static_assert(RR == 0 || NRT == 1, "the small-row form has one row tile");
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int total = ntiles * KS;
  f32x4_t bn[U];
  f32x4_t acc[NRT];
  float part[RR > 0 ? RR : 1];
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt) acc[rt] = f4zero();
#pragma unroll
  for (int r = 0; r < (RR > 0 ? RR : 1); ++r) part[r] = 0.f;
  for (int s0 = 0; s0 < total; s0 += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int sn = min(s0 + U + u, total - 1);
      bn[u] = bload(sn / KS, (sn % KS) * 16);
    }
    if constexpr (U > KS) {
      // a prefetch group spans U / KS whole tiles (short contractions: the FFN's 64-unit slices)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int su = s0 + u, t = su / KS, k = (su % KS) * 16;
        if (su < total) {
#pragma unroll
          for (int rt = 0; rt < NRT; ++rt) acc[rt] = mfma4(aload(t, k, 16 * rt + j), bq[u], acc[rt]);
          if ((u + 1) % KS == 0) {
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) { epi(t, rt, acc[rt]); acc[rt] = f4zero(); }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) bq[u] = bn[u];
      continue;
    }
    const int t = s0 / KS, kb = (s0 % KS) * 16;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if constexpr (RR == 0) {
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) acc[rt] = mfma4(aload(t, kb + 16 * u, 16 * rt + j), bq[u], acc[rt]);
      } else {
#pragma unroll
        for (int r = 0; r < RR; ++r) {
          const f32x4_t av = aload(t, kb + 16 * u, r);
          part[r] = fmaf(av[0], bq[u][0], fmaf(av[1], bq[u][1], fmaf(av[2], bq[u][2], fmaf(av[3], bq[u][3], part[r]))));
        }
      }
    }
    if (kb + 16 * U == KS * 16) {
      if constexpr (RR > 0) {
#pragma unroll
        for (int r = 0; r < RR; ++r) {
          float v = part[r];
          v += __shfl_xor(v, 16, 64);
          v += __shfl_xor(v, 32, 64);
          acc[0][r] = g == 0 ? v : 0.f;
          part[r] = 0.f;
        }
      }
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt) { epi(t, rt, acc[rt]); acc[rt] = f4zero(); }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) bq[u] = bn[u];
  }
}

template <int KS, int NRT, int U, int RR, class BLoad, class ALoad, class Epi>
__device__ __forceinline__ void stream_nt(int ntiles, BLoad bload, ALoad aload, Epi epi) {
  f32x4_t bq[U];
  stream_nt_prefetch<KS, U>(ntiles, bload, bq);
  stream_nt<KS, NRT, U, RR>(ntiles, bload, aload, epi, bq);
}

// P2: Y[r][c] = sum_n D(r, n) * W(n, c)  -- the OUTPUT index is contiguous in W's rows (dgrad of a Linear; P' V; dS K).  A wave
// owns NCG groups of 64 output columns; lane (j, g) holds columns cb + 4j .. +3 of its group as FOUR accumulators (one MFMA per
// column offset), so one 16-byte load of W(n0 + g, cb + 4j ..) feeds four MFMAs and a wave instruction reads 4 rows x 256 B.
//   bload(n, cg)   -> W(n, cb(cg) + 4j .. +3) for the lane's n = n0 + g        (n may be out of range: return zeros)
//   aload(n, row)  -> D(row, n)                                                  (the same)
//   acc[rt][cg][c][v] = Y[16 rt + 4g + v][cb(cg) + 4j + c]
// RR as above (rows 0 .. RR-1 on the VALU: lane (j, g) sums its n = n0 + g, n0 + 4 + g, ...; the four g meet at the end).
template <int NCG, int U, class BLoad>
__device__ __forceinline__ void stream_nn_prefetch(int n_begin, int n_end, BLoad bload, f32x4_t (&bq)[U][NCG]) {
  const int g = (threadIdx.x & 63) >> 4;
  const int steps = (n_end - n_begin + 3) >> 2;
  if (steps <= 0) return;
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int cg = 0; cg < NCG; ++cg) bq[u][cg] = bload(n_begin + 4 * min(u, steps - 1) + g, cg);
}
template <int NRT, int NCG, int U, int RR, class BLoad, class ALoad>
__device__ __forceinline__ void stream_nn(int n_begin, int n_end, BLoad bload, ALoad aload, f32x4_t (&acc)[NRT][NCG][4],
                                          f32x4_t (&bq)[U][NCG]) {
  static_assert(RR == 0 || NRT == 1, "the small-row form has one row tile");
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int steps = (n_end - n_begin + 3) >> 2;
  f32x4_t bn[U][NCG];
  f32x4_t part[RR > 0 ? RR : 1][NCG];
#pragma unroll
  for (int r = 0; r < (RR > 0 ? RR : 1); ++r)
#pragma unroll
    for (int cg = 0; cg < NCG; ++cg) part[r][cg] = f4zero();
  for (int s0 = 0; s0 < steps; s0 += U) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int cg = 0; cg < NCG; ++cg) bn[u][cg] = bload(n_begin + 4 * min(s0 + U + u, steps - 1) + g, cg);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (s0 + u < steps) {
        const int n = n_begin + 4 * (s0 + u) + g;
        if constexpr (RR == 0) {
#pragma unroll
          for (int rt = 0; rt < NRT; ++rt) {
            const float a = aload(n, 16 * rt + j);
#pragma unroll
            for (int cg = 0; cg < NCG; ++cg)
#pragma unroll
              for (int c = 0; c < 4; ++c) acc[rt][cg][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bq[u][cg][c], acc[rt][cg][c], 0, 0, 0);
          }
        } else {
#pragma unroll
          for (int r = 0; r < RR; ++r) {
            const float a = aload(n, r);
#pragma unroll
            for (int cg = 0; cg < NCG; ++cg) part[r][cg] += a * bq[u][cg];
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int cg = 0; cg < NCG; ++cg) bq[u][cg] = bn[u][cg];
  }
  if constexpr (RR > 0) {
#pragma unroll
    for (int cg = 0; cg < NCG; ++cg)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < RR; ++r) {
          float v = part[r][cg][c];
          v += __shfl_xor(v, 16, 64);
          v += __shfl_xor(v, 32, 64);
          acc[0][cg][c][r] += g == 0 ? v : 0.f;
        }
  }
}

template <int NRT, int NCG, int U, int RR, class BLoad, class ALoad>
__device__ __forceinline__ void stream_nn(int n_begin, int n_end, BLoad bload, ALoad aload, f32x4_t (&acc)[NRT][NCG][4]) {
  f32x4_t bq[U][NCG];
  stream_nn_prefetch<NCG, U>(n_begin, n_end, bload, bq);
  stream_nn<NRT, NCG, U, RR>(n_begin, n_end, bload, aload, acc, bq);
}

// One query row (num_queries = 1): y[n] = sum_k x(n)[k] W(n, k) for n in [0, N), K = 256, as a matrix-vector product with every wave
// instruction reading 4 whole 256-byte row segments (lane (c = lane & 15, rr = lane >> 4): W(n0 + rr, 64 kc + 4c .. +3)); the 16 lanes
// of a row meet through 4 cross-lane adds.  Row groups n0 = 4 (wave + NW i); U row groups (4 loads of 16 B each per lane) in flight,
// kept in a ring (a group's registers are reloaded the moment it is consumed).  xsel(n0) -> the LDS vector (256 floats) the rows of
// group n0 contract with; out(n, y) is called by ONE lane per n.
template <int NW, int U, class XSel, class Out>
__device__ __forceinline__ void gemv256(const float* W, long ldw, int N, XSel xsel, Out out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, rr = lane >> 4;
  const int groups = N >> 2;
  const int mine = (groups - wave + NW - 1) / NW;
  f32x4_t w[U][4];
  if (mine <= 0) return;
  // (loads past the wave's last group re-read that group: an UNCONDITIONAL load keeps the compiler's vmcnt bookkeeping exact -- with
  // predicated loads it assumed the fewest in flight and its waits drained the queue once per pass)
  auto lw = [&](int i, f32x4_t (&wg)[4]) {
    const float* p = W + (long)(4 * (wave + NW * min(i, mine - 1)) + rr) * ldw + 4 * c;
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) wg[kc] = ld4(p + 64 * kc);
  };
#pragma unroll
  for (int u = 0; u < U; ++u) lw(u, w[u]);
  for (int i0 = 0; i0 < mine; i0 += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (i0 + u < mine) {
        const int n0 = 4 * (wave + NW * (i0 + u));
        const float* x = xsel(n0);
        f32x4_t t = w[u][0] * ld4(x + 4 * c);
#pragma unroll
        for (int kc = 1; kc < 4; ++kc) t += w[u][kc] * ld4(x + 64 * kc + 4 * c);
        lw(i0 + U + u, w[u]);
        float v = (t[0] + t[1]) + (t[2] + t[3]);
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 8, 64);
        if (c == 0) out(n0 + rr, v);
      }
    }
  }
}

// ---- one query per sample: the cross-attention's passes over the source rows on the VALU, every load a whole row -----------------
// (The MFMA forms above fetch 32- / 64-byte pieces of 16 rows per instruction; with 8 (query, head) rows per sample the pieces of a
// row were requested by different instructions far apart, the 32 KB L1 did not hold them in between, and every 128-byte line came
// from L2 two to four times: 15 GB/s.)
// 8 row values per lane at columns 8 c32 .. +7 of source row r: value form (S16: one 16-byte load) or key form (+ the position row)
template <bool S16, bool KEY>
struct Row8 {
  u32x4_t raw; f32x4_t lo, hi, p0, p1;
  __device__ __forceinline__ void load(const SrcRows<S16>& s, int r, int c) {
    if constexpr (S16) raw = *(const u32x4_t*)(s.s16 + r * s.ld + c);
    else { lo = ld4(s.s32 + r * s.ld + c); hi = ld4(s.s32 + r * s.ld + c + 4); }
    if constexpr (KEY) { p0 = ld4(s.kpos + r * s.ldkp + c); p1 = ld4(s.kpos + r * s.ldkp + c + 4); }
  }
  __device__ __forceinline__ void get(f32x4_t& a, f32x4_t& b) const {
    if constexpr (S16) {
      float x0, x1, x2, x3, x4, x5, x6, x7;
      unpack_lp2(raw[0], x0, x1); unpack_lp2(raw[1], x2, x3); unpack_lp2(raw[2], x4, x5); unpack_lp2(raw[3], x6, x7);
      a = (f32x4_t){x0, x1, x2, x3}; b = (f32x4_t){x4, x5, x6, x7};
    } else { a = lo; b = hi; }
    if constexpr (KEY) { a += p0; b += p1; }
  }
};

// out[h][kk] = Q[h] . row(kk) (+ add[h]) for the 8 heads: 32 lanes per row (8 columns each), two rows per wave instruction; the
// 32 partial sums of the 8 heads meet in a butterfly (4 + 2 + 1 + 1 + 1 cross-lane adds: afterwards lane l of the half-wave holds
// head (l >> 2) & 7).  Rows round-robin over the waves in pairs, U pairs in flight.  Q: LDS [8][DLD]; out: LDS [8][LKP].
template <bool S16, bool KEY, int U, class Fin>
__device__ __forceinline__ void rowdot8(const SrcRows<S16>& src, const float* Q, int Lk, Fin fin) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c32 = lane & 31, par = lane >> 5;
  f32x4_t qa[8], qb[8];
#pragma unroll
  for (int h = 0; h < 8; ++h) { qa[h] = ld4(Q + h * DLD + 8 * c32); qb[h] = ld4(Q + h * DLD + 8 * c32 + 4); }
  const int pairs = (Lk + 1) >> 1;
  const int mine = (pairs - wave + DNW - 1) / DNW;
  if (mine <= 0) return;
  Row8<S16, KEY> ring[U];
  auto row_of = [&](int i) { return min(2 * (wave + DNW * min(i, mine - 1)) + par, Lk - 1); };
#pragma unroll
  for (int u = 0; u < U; ++u) ring[u].load(src, row_of(u), 8 * c32);
  for (int i0 = 0; i0 < mine; i0 += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (i0 + u < mine) {
        f32x4_t ka, kb;
        ring[u].get(ka, kb);
        ring[u].load(src, row_of(i0 + U + u), 8 * c32);
        float v[8];
#pragma unroll
        for (int h = 0; h < 8; ++h) {
          const f32x4_t t = qa[h] * ka + qb[h] * kb;
          v[h] = (t[0] + t[1]) + (t[2] + t[3]);
        }
        // butterfly over the 32 lanes of the row: halve the number of live heads at each of the first three steps
        float w4[4], w2[2], w1;
        const bool b4 = c32 & 16, b3 = c32 & 8, b2 = c32 & 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float keep = b4 ? v[4 + k] : v[k], send = b4 ? v[k] : v[4 + k];
          w4[k] = keep + __shfl_xor(send, 16, 64);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const float keep = b3 ? w4[2 + k] : w4[k], send = b3 ? w4[k] : w4[2 + k];
          w2[k] = keep + __shfl_xor(send, 8, 64);
        }
        {
          const float keep = b2 ? w2[1] : w2[0], send = b2 ? w2[0] : w2[1];
          w1 = keep + __shfl_xor(send, 4, 64);
        }
        w1 += __shfl_xor(w1, 2, 64);
        w1 += __shfl_xor(w1, 1, 64);
        const int kk = 2 * (wave + DNW * (i0 + u)) + par;
        if ((c32 & 3) == 0 && kk < Lk) fin((c32 >> 2) & 7, kk, w1);       // head = b4 b3 b2
      }
    }
  }
}

// acc[h][4 columns] += PT[kk][h] * row(kk)[columns 4 lane .. +3] over the wave's rows (kk = wave, wave + 8, ...): one whole row per
// wave instruction, U rows in flight.  PT: LDS [Lk][8] (row kk's 8 head factors).  The waves' partial sums are combined by the caller.
template <bool S16, bool KEY, int U>
__device__ __forceinline__ void rowacc8(const SrcRows<S16>& src, const float* PT, int Lk, f32x4_t (&acc)[8]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int mine = (Lk - wave + DNW - 1) / DNW;
  if (mine <= 0) return;
  f32x4_t ring[U], pring[U];
  u32x2_t ring16[U];
  auto row_of = [&](int i) { return wave + DNW * min(i, mine - 1); };
  auto ld = [&](int u, int i) {
    const int r = row_of(i);
    if constexpr (S16) ring16[u] = *(const u32x2_t*)(src.s16 + r * src.ld + 4 * lane);
    else ring[u] = ld4(src.s32 + r * src.ld + 4 * lane);
    if constexpr (KEY) pring[u] = ld4(src.kpos + r * src.ldkp + 4 * lane);
  };
#pragma unroll
  for (int u = 0; u < U; ++u) ld(u, u);
  for (int i0 = 0; i0 < mine; i0 += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (i0 + u < mine) {
        f32x4_t x;
        if constexpr (S16) { float a0, a1, a2, a3; unpack_lp2(ring16[u][0], a0, a1); unpack_lp2(ring16[u][1], a2, a3); x = (f32x4_t){a0, a1, a2, a3}; }
        else x = ring[u];
        if constexpr (KEY) x += pring[u];
        ld(u, i0 + U + u);
        const int kk = wave + DNW * (i0 + u);
        const f32x4_t f0 = ld4(PT + kk * 8), f1 = ld4(PT + kk * 8 + 4);
#pragma unroll
        for (int h = 0; h < 4; ++h) { acc[h] += f0[h] * x; acc[4 + h] += f1[h] * x; }
      }
    }
  }
}
// the 8 waves' partial [8][256] sums -> their sum in `out` (LDS [8][DLD]) and at out_g (global rows of E floats): waves 4 .. 7 park
// theirs in PART ([4][8][256]), waves 0 .. 3 add their own on top, then every thread sums 4 of the 2048 entries (fixed order)
__device__ __forceinline__ void rowacc8_combine(f32x4_t (&acc)[8], float* PART, float* out, float* out_g) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (wave >= 4) {
#pragma unroll
    for (int h = 0; h < 8; ++h) *(f32x4_t*)(PART + ((wave - 4) * 8 + h) * DE + 4 * lane) = acc[h];
  }
  __syncthreads();
  if (wave < 4) {
#pragma unroll
    for (int h = 0; h < 8; ++h) {
      float* p = PART + (wave * 8 + h) * DE + 4 * lane;
      *(f32x4_t*)p = acc[h] + ld4(p);
    }
  }
  __syncthreads();
  for (int e = tid; e < 8 * (DE / 4); e += DNT) {
    const int h = e / (DE / 4), c = 4 * (e % (DE / 4));
    const f32x4_t v = (ld4(PART + h * DE + c) + ld4(PART + (8 + h) * DE + c)) + (ld4(PART + (16 + h) * DE + c) + ld4(PART + (24 + h) * DE + c));
    *(f32x4_t*)(out + h * DLD + c) = v;
    if (out_g) *(f32x4_t*)(out_g + (long)h * DE + c) = v;
  }
}

// LayerNorm of the R rows of an LDS tile [16][DLD]: y = (x - mean) * rstd * g + b, one wave per row; mean / rstd to global
__device__ __forceinline__ void ln_rows(const float* x, float* y, int R, const float* g, const float* b, float eps,
                                        float* y_g, float* mean_g, float* rstd_g) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = wave; r < R; r += DNW) {
    const f32x4_t v = ld4(x + r * DLD + 4 * lane);
    const float mean = wave_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.f / DE);
    const f32x4_t d = v - mean;
    const float var = wave_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * (1.f / DE);
    const float rstd = rsqrtf(var + eps);
    const f32x4_t o = d * rstd * ld4(g + 4 * lane) + ld4(b + 4 * lane);
    *(f32x4_t*)(y + r * DLD + 4 * lane) = o;
    if (y_g) *(f32x4_t*)(y_g + (long)r * DE + 4 * lane) = o;
    if (lane == 0 && mean_g) { mean_g[r] = mean; rstd_g[r] = rstd; }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
struct DecAttnArgs {
  int B, R, Lk, kv_rows, kv_off;      // R queries per sample; Lk keys per sample = rows kv_off .. kv_off + Lk of its kv_rows source rows
  const float* tgt; const float* qpos; // [B*R, E]
  const float *Ws, *bs, *Wso, *bso, *g0, *b0, *Wc, *bc, *Wco, *bco, *g1, *b1;
  const lp_t* src16; const float* src32; long ldsrc;
  const float* kpos; long ldkp; int kpos_rows;           // kpos_rows: rows per sample (0: one [Lk, E] set for the batch)
  const unsigned char* kpm;                              // [B, Lk] 1 = masked, or null
  const float* dm0; const float* dm1;                    // dropout multipliers [B,H,R,R] / [B,H,R,Lk], or null
  // saved for the backward (and the layer's outputs): rows of this layer's [B*R, .] buffers
  float *qkv, *P0, *o, *r1, *mean1, *rstd1, *t1, *qc, *qk, *P1, *ctx, *sp, *o2, *r2, *mean2, *rstd2, *t2;
  float eps;
};

// LDS map (floats): T, QP, X1 [16][DLD] each, then BIG: { qkv [16][3E+4] + self-attention scores } | { chunk buffers }
// PAR: the layer's biases and LayerNorm parameters + the sample's key mask and sum(P') -- staged ONCE: a global load inside a
// streaming loop (a bias in a tile's epilogue) makes the in-order vmcnt wait drain every weight load in flight behind it
constexpr int P_BS = 0, P_BSO = 3 * DE, P_BC = 4 * DE, P_BCO = 7 * DE, P_G0 = 8 * DE, P_B0 = 9 * DE, P_G1 = 10 * DE, P_B1 = 11 * DE,
              P_SP = 12 * DE, P_KPM = 12 * DE + 16 * DH, P_END = P_KPM + 1024 / 4;
constexpr int L_T = 0, L_QP = 16 * DLD, L_X1 = 2 * 16 * DLD, L_PAR = 3 * 16 * DLD, L_BIG = L_PAR + P_END;
constexpr int QKV_LD = 3 * DE + 4;
constexpr int CR = 32;                // (query, head) rows per cross-attention chunk: 4 queries x 8 heads
__host__ __device__ constexpr int lkp_of(int Lk) { return ((Lk + 15) & ~15) + 4; }
__host__ __device__ constexpr int dec_attn_fwd_lds_floats(int Lk) {
  const int a = 16 * QKV_LD + DH * 16 * 16;                  // qkv + self-attention probabilities
  const int b = CR * DLD + CR * lkp_of(Lk);                  // qk chunk (later the partial ctx of the second key half) + score strip
  return L_BIG + (a > b ? a : b);
}

template <int RR, bool S16>
__global__ __launch_bounds__(DNT) void dec_attn_fwd_kernel(DecAttnArgs a) {
  constexpr int UW = RR ? 16 : 8;     // 16-byte loads of W in flight per lane in the weight phases
  extern __shared__ float sm[];
  float* T = sm + L_T;
  float* QP = sm + L_QP;
  float* X1 = sm + L_X1;
  float* PAR = sm + L_PAR;
  float* SPL = PAR + P_SP;                        // [16][H] sum_k P'
  unsigned char* KPM = (unsigned char*)(PAR + P_KPM);
  float* BIG = sm + L_BIG;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  const int R = a.R;
  const long row0 = (long)b * R;
  const int Lk = a.Lk, LKP = lkp_of(Lk);
  const int ktiles = (Lk + 15) >> 4;
  SrcRows<S16> src;
  src.s16 = a.src16 ? a.src16 + ((long)b * a.kv_rows + a.kv_off) * a.ldsrc : nullptr;
  src.s32 = a.src32 ? a.src32 + ((long)b * a.kv_rows + a.kv_off) * a.ldsrc : nullptr;
  src.ld = a.ldsrc;
  src.kpos = a.kpos + (long)b * a.kpos_rows * a.ldkp;
  src.ldkp = a.ldkp;
  // the B-fragment loaders of every phase (each phase's first loads are issued before the previous phase's barrier)
  auto tile_row = [&](int t) { return 16 * (wave + DNW * t) + j; };
  auto bl_s = [&](int t, int k) { return ld4(a.Ws + (long)tile_row(t) * DE + k + 4 * g); };
  auto bl_so = [&](int t, int k) { return ld4(a.Wso + (long)tile_row(t) * DE + k + 4 * g); };
  auto bl_q = [&](int t, int k) { return ld4(a.Wc + (long)tile_row(t) * DE + k + 4 * g); };
  const float* Wk_h = a.Wc + (long)(DE + wave * DHD) * DE;         // phase 5: wave = head
  auto bl_k = [&](int n, int cg) { return ld4(Wk_h + (long)n * DE + 64 * cg + 4 * j); };
  auto bl_v = [&](int t, int k) { return ld4(a.Wc + (long)(2 * DE + tile_row(t)) * DE + k + 4 * g); };
  auto bl_co = [&](int t, int k) { return ld4(a.Wco + (long)tile_row(t) * DE + k + 4 * g); };
  const int key_mine = (ktiles - wave + DNW - 1) / DNW;
  auto bl_key = [&](int t, int k) { return src.key(min(tile_row(t), Lk - 1), k + 4 * g); };
  const int ccg = wave & 3, chalf = wave >> 2;
  const int kmid = ((ktiles + 1) >> 1) << 4;
  const int cb = chalf ? kmid : 0, ce = chalf ? (ktiles << 4) : kmid;
  auto bl_val = [&](int n, int) { return src.val(min(n, Lk - 1), 64 * ccg + 4 * j); };     // (rows past Lk meet zero probabilities)
  f32x4_t pf_s[UW];
  TL(0);
  stream_nt_prefetch<16, UW>(6, bl_s, pf_s);
  // ---- 0: the sample's rows (rows >= R of the 16-row tiles are zero and stay zero)
  for (int e = tid; e < 16 * (DE / 4); e += DNT) {
    const int r = e / (DE / 4), c = 4 * (e % (DE / 4));
    f32x4_t t = f4zero(), q = f4zero();
    if (r < R) { if (a.tgt) t = ld4(a.tgt + (row0 + r) * DE + c); q = ld4(a.qpos + (row0 + r) * DE + c); }      // tgt == NULL: zeros (a decoder's first layer)
    *(f32x4_t*)(T + r * DLD + c) = t;
    *(f32x4_t*)(QP + r * DLD + c) = q;
    *(f32x4_t*)(X1 + r * DLD + c) = t + q;
  }
  for (int e = tid; e < 3 * DE; e += DNT) { PAR[P_BS + e] = a.bs[e]; PAR[P_BC + e] = a.bc[e]; }
  for (int e = tid; e < DE; e += DNT) {
    PAR[P_BSO + e] = a.bso[e]; PAR[P_BCO + e] = a.bco[e];
    PAR[P_G0 + e] = a.g0[e]; PAR[P_B0 + e] = a.b0[e]; PAR[P_G1 + e] = a.g1[e]; PAR[P_B1 + e] = a.b1[e];
  }
  for (int e = tid; e < Lk; e += DNT) KPM[e] = a.kpm ? a.kpm[(long)b * Lk + e] : 0;
  __syncthreads(); TL(1);
  // ---- 1: self-attention in-projection: q | k from tgt + qpos, v from tgt (48 column tiles, 6 per wave)
  float* QKV = BIG;
  {
    auto aload = [&](int t, int k, int row) { return ld4(((wave + DNW * t) < 32 ? X1 : T) + row * DLD + k + 4 * g); };
    auto epi = [&](int t, int, f32x4_t acc) {
      const int n = tile_row(t);
      const float bias = PAR[P_BS + n];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int r = 4 * g + v;
        const float y = r < R ? acc[v] + bias : 0.f;
        QKV[r * QKV_LD + n] = y;
        if (r < R) a.qkv[(row0 + r) * (3 * DE) + n] = y;
      }
    };
    if constexpr (RR == 1) {
      // one query: the softmax over its single key is 1 whatever q and k are -- only v = tgt Wv^T + bv reaches the output, and
      // q, k get no gradient (dS = 0): their projections (2/3 of this weight) are not computed; the saved q | k are zeros
      (void)pf_s;
      for (int e = tid; e < 2 * DE; e += DNT) { QKV[e] = 0.f; a.qkv[row0 * (3 * DE) + e] = 0.f; }
      gemv256<DNW, 8>(a.Ws + (long)2 * DE * DE, DE, DE, [&](int) { return T; },
                      [&](int n, float y) { y += PAR[P_BS + 2 * DE + n]; QKV[2 * DE + n] = y; a.qkv[row0 * (3 * DE) + 2 * DE + n] = y; });
    } else {
      stream_nt<16, 1, UW, RR>(6, bl_s, aload, epi, pf_s);
    }
  }
  f32x4_t pf_so[UW];
  stream_nt_prefetch<16, UW>(2, bl_so, pf_so);
  __syncthreads(); TL(2);
  // ---- 2: self-attention over the sample's R queries (8 heads x 32): scores, softmax, dropout, P V
  float* S0 = BIG + 16 * QKV_LD;                  // [H][16][16]
  {
    const float scale = 0.17677669529663687f;     // 32^-1/2
    for (int e = tid; e < DH * R * R; e += DNT) {
      const int h = e / (R * R), r = (e / R) % R, r2 = e % R;
      const float* q = QKV + r * QKV_LD + h * DHD;
      const float* k = QKV + r2 * QKV_LD + DE + h * DHD;
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < DHD; ++d) s = fmaf(q[d], k[d], s);
      S0[(h * 16 + r) * 16 + r2] = s * scale;
    }
    __syncthreads(); TL(3);
    for (int e = tid; e < DH * R; e += DNT) {
      const int h = e / R, r = e % R;
      float* s = S0 + (h * 16 + r) * 16;
      float mx = -INFINITY;
      for (int i = 0; i < R; ++i) mx = fmaxf(mx, s[i]);
      float sum = 0.f;
      for (int i = 0; i < R; ++i) { s[i] = expf(s[i] - mx); sum += s[i]; }
      const float inv = 1.f / sum;
      const long pg = (((long)b * DH + h) * R + r) * R;
      for (int i = 0; i < R; ++i) {
        const float p = s[i] * inv;
        a.P0[pg + i] = p;
        s[i] = a.dm0 ? p * a.dm0[pg + i] : p;
      }
    }
    __syncthreads(); TL(4);
    for (int e = tid; e < R * DE; e += DNT) {
      const int r = e / DE, n = e % DE, h = n / DHD;
      const float* p = S0 + (h * 16 + r) * 16;
      float o = 0.f;
      for (int i = 0; i < R; ++i) o = fmaf(p[i], QKV[i * QKV_LD + 2 * DE + n], o);
      X1[r * DLD + n] = o;                        // rows >= R of X1 hold t + q of zero rows = 0
      a.o[(row0 + r) * DE + n] = o;
    }
  }
  __syncthreads(); TL(5);
  // ---- 3: r1 = tgt + o Wso^T + bso ; t1 = LayerNorm(r1)
  float* R1 = BIG;                                // qkv is dead
  {
    auto aload = [&](int, int k, int row) { return ld4(X1 + row * DLD + k + 4 * g); };
    auto epi = [&](int t, int, f32x4_t acc) {
      const int n = tile_row(t);
      const float bias = PAR[P_BSO + n];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int r = 4 * g + v;
        const float y = r < R ? acc[v] + bias + T[r * DLD + n] : 0.f;
        R1[r * DLD + n] = y;
        if (r < R) a.r1[(row0 + r) * DE + n] = y;
      }
    };
    stream_nt<16, 1, UW, RR>(2, bl_so, aload, epi, pf_so);
  }
  f32x4_t pf_q[UW];
  stream_nt_prefetch<16, UW>(2, bl_q, pf_q);
  __syncthreads(); TL(6);
  ln_rows(R1, X1, R, PAR + P_G0, PAR + P_B0, a.eps, a.t1 + row0 * DE, a.mean1 + row0, a.rstd1 + row0);      // X1 = t1
  __syncthreads(); TL(7);
  // ---- 4: cross-attention query: qc = ((t1 + qpos) Wq^T + bq) * 32^-1/2
  float* XQ = BIG;                                 // [16][DLD]
  float* QC = BIG + 16 * DLD;                      // [16][DLD]
  for (int e = tid; e < 16 * (DE / 4); e += DNT) {
    const int r = e / (DE / 4), c = 4 * (e % (DE / 4));
    *(f32x4_t*)(XQ + r * DLD + c) = ld4(X1 + r * DLD + c) + ld4(QP + r * DLD + c);
  }
  __syncthreads(); TL(8);
  {
    auto aload = [&](int, int k, int row) { return ld4(XQ + row * DLD + k + 4 * g); };
    auto epi = [&](int t, int, f32x4_t acc) {
      const int n = tile_row(t);
      const float bias = PAR[P_BC + n];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int r = 4 * g + v;
        const float y = r < R ? (acc[v] + bias) * 0.17677669529663687f : 0.f;
        QC[r * DLD + n] = y;
        if (r < R) a.qc[(row0 + r) * DE + n] = y;
      }
    };
    stream_nt<16, 1, UW, RR>(2, bl_q, aload, epi, pf_q);
  }
  f32x4_t pf_k[4][4];
  stream_nn_prefetch<4, 4>(0, DHD, bl_k, pf_k);
  __syncthreads(); TL(9);
  // ---- 5: qk[r][h][:] = Wk_h^T qc_h[r]  (wave = head: 32 rows of Wk, all 256 columns) -> global [B*R, H, E]
  {
    const int h = wave;
    f32x4_t acc[1][4][4];
#pragma unroll
    for (int cg = 0; cg < 4; ++cg)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[0][cg][c] = f4zero();
    auto aload = [&](int n, int row) { return QC[row * DLD + h * DHD + n]; };
    stream_nn<1, 4, 4, RR>(0, DHD, bl_k, aload, acc, pf_k);
#pragma unroll
    for (int cg = 0; cg < 4; ++cg)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int r = 4 * g + v;
        if (r < R) {
          const f32x4_t o = (f32x4_t){acc[0][cg][0][v], acc[0][cg][1][v], acc[0][cg][2][v], acc[0][cg][3][v]};
          *(f32x4_t*)(a.qk + ((row0 + r) * DH + h) * DE + 64 * cg + 4 * j) = o;
          if constexpr (RR == 1) *(f32x4_t*)(BIG + h * DLD + 64 * cg + 4 * j) = o;      // one query: its 8 (query, head) rows straight into the chunk tile
        }
      }
  }
  f32x4_t pf_key[8];
  if constexpr (RR == 0) stream_nt_prefetch<16, 8>(key_mine, bl_key, pf_key);
  __syncthreads(); TL(10);            // (also orders the qk stores before the chunk loop's loads: same workgroup, write-through L1)
  // ---- 6 (one query): scores, softmax, context on the VALU with whole-row loads
  f32x4_t pf_v[UW];
  float* QKC = BIG;                                // [CR][DLD]   (one query: rows 0 .. 7 = the heads' qk, written by phase 5)
  float* SC = BIG + (RR == 1 ? 8 : CR) * DLD;      // [CR][LKP]   (one query: [8][LKP], then P' transposed and the waves' partial sums)
  if constexpr (RR == 1) {
    float* PT = SC + 8 * LKP;                      // [Lk][8]  P' transposed
    float* PARTS = PT + 8 * LKP;                   // [4][8][256]
    rowdot8<S16, true, 6>(src, QKC, Lk, [&](int h, int kk, float v) { SC[h * LKP + kk] = KPM[kk] ? -INFINITY : v; });
    stream_nt_prefetch<16, UW>(2, bl_v, pf_v);
    __syncthreads();
    {
      const int h = wave;                          // 8 waves = 8 heads
      float* sr = SC + h * LKP;
      float mx = -INFINITY;
      for (int kk = lane; kk < Lk; kk += 64) mx = fmaxf(mx, sr[kk]);
      mx = wave_max(mx);
      float sum = 0.f;
      for (int kk = lane; kk < Lk; kk += 64) { const float p = expf(sr[kk] - mx); sr[kk] = p; sum += p; }
      sum = wave_sum(sum);
      const float inv = 1.f / sum;
      const long pg = ((long)b * DH + h) * Lk;
      float sp = 0.f;
      // keys in chunks of 448 (seven per lane): the chunk's dropout multipliers are requested before its loop -- inside it each
      // load waited behind the previous key's store of P1
      for (int k0 = 0; k0 < Lk; k0 += 448) {
        float dmv[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) dmv[i] = a.dm1 ? a.dm1[pg + min(k0 + lane + 64 * i, Lk - 1)] : 1.f;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          const int kk = k0 + lane + 64 * i;
          if (kk < Lk) {
            const float p = sr[kk] * inv;
            a.P1[pg + kk] = p;
            const float pd = a.dm1 ? p * dmv[i] : p;
            PT[kk * 8 + h] = pd;
            sp += pd;
          }
        }
      }
      sp = wave_sum(sp);
      if (lane == 0) { a.sp[row0 * DH + h] = sp; SPL[h] = sp; }
    }
    __syncthreads();
    f32x4_t cacc[8];
#pragma unroll
    for (int h = 0; h < 8; ++h) cacc[h] = f4zero();
    rowacc8<S16, false, 16>(src, PT, Lk, cacc);
    rowacc8_combine(cacc, PARTS, SC, a.ctx + row0 * DH * DE);          // ctx rows: LDS (SC rows 0 .. 7, stride DLD) + global
    __syncthreads();
  } else {
  for (int q0 = 0; q0 < R; q0 += 4) {
    const int nrho = min(4, R - q0) * DH;          // valid (query, head) rows of this chunk: rho = (r - q0) * H + h
    if constexpr (RR == 0) {
      for (int e = tid; e < CR * (DE / 4); e += DNT) {
        const int rho = e / (DE / 4), c = 4 * (e % (DE / 4));
        *(f32x4_t*)(QKC + rho * DLD + c) = rho < nrho ? ld4(a.qk + ((row0 + q0) * DH + rho) * DE + c) : f4zero();
      }
      __syncthreads();
    } TL(11);
    // scores[rho][kk] = qk[rho] . key[kk]: key tiles round-robin over the waves
    {
      auto aload = [&](int, int k, int row) { return ld4(QKC + row * DLD + k + 4 * g); };
      auto epi = [&](int t, int rt, f32x4_t acc) {
        const int kk = tile_row(t);
        const bool dead = kk >= Lk || KPM[kk];
#pragma unroll
        for (int v = 0; v < 4; ++v) SC[(16 * rt + 4 * g + v) * LKP + kk] = dead ? -INFINITY : acc[v];
      };
      if (nrho > 16) stream_nt<16, 2, 8, 0>(key_mine, bl_key, aload, epi, pf_key);
      else stream_nt<16, 1, 8, 0>(key_mine, bl_key, aload, epi, pf_key);
    }
    f32x4_t pf_val[8][1];
    stream_nn_prefetch<1, 8>(cb, ce, bl_val, pf_val);
    __syncthreads(); TL(12);
    // softmax per (query, head) row; P1 to global, P' = P * dropout in place, sp = sum P'
    for (int rho = wave; rho < nrho; rho += DNW) {
      float* s = SC + rho * LKP;
      const int r = q0 + rho / DH, h = rho % DH;
      float mx = -INFINITY;
      for (int kk = lane; kk < Lk; kk += 64) mx = fmaxf(mx, s[kk]);
      mx = wave_max(mx);
      float sum = 0.f;
      for (int kk = lane; kk < Lk; kk += 64) { const float p = expf(s[kk] - mx); s[kk] = p; sum += p; }
      sum = wave_sum(sum);
      const float inv = 1.f / sum;
      const long pg = (((long)b * DH + h) * R + r) * Lk;
      float sp = 0.f;
      for (int kk = lane; kk < Lk; kk += 64) {
        const float p = s[kk] * inv;
        a.P1[pg + kk] = p;
        const float pd = a.dm1 ? p * a.dm1[pg + kk] : p;
        s[kk] = pd;
        sp += pd;
      }
      for (int kk = Lk + lane; kk < LKP - 4; kk += 64) s[kk] = 0.f;        // the padding keys of the last tile
      sp = wave_sum(sp);
      if (lane == 0) { a.sp[(row0 + r) * DH + h] = sp; SPL[r * DH + h] = sp; }
    }
    for (int rho = nrho + wave; rho < CR; rho += DNW)                       // rows without a query: zero probabilities
      for (int kk = lane; kk < LKP - 4; kk += 64) SC[rho * LKP + kk] = 0.f;
    __syncthreads(); TL(13);
    // ctx[rho][c] = sum_kk P'[rho][kk] val[kk][c]: wave = (64-column group, key half); the halves meet in LDS
    {
      f32x4_t acc[2][1][4];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[rt][0][c] = f4zero();
      auto aload = [&](int n, int row) { return SC[row * LKP + n]; };
      if (nrho > 16) stream_nn<2, 1, 8, 0>(cb, ce, bl_val, aload, acc, pf_val);
      else stream_nn<1, 1, 8, 0>(cb, ce, bl_val, aload, reinterpret_cast<f32x4_t (&)[1][1][4]>(acc), pf_val);
      if (q0 + 4 < R) stream_nt_prefetch<16, 8>(key_mine, bl_key, pf_key);       // the next chunk's first keys
      else stream_nt_prefetch<16, UW>(2, bl_v, pf_v);                            // or the first rows of Wv
      float* PART = QKC;                            // the qk chunk is dead: the second half's partial sums go here
      if (chalf) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int v = 0; v < 4; ++v)
            *(f32x4_t*)(PART + (16 * rt + 4 * g + v) * DLD + 64 * ccg + 4 * j) =
                (f32x4_t){acc[rt][0][0][v], acc[rt][0][1][v], acc[rt][0][2][v], acc[rt][0][3][v]};
      }
      __syncthreads(); TL(14);
      if (!chalf) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int rho = 16 * rt + 4 * g + v;
            if (rho < nrho) {
              const f32x4_t o = (f32x4_t){acc[rt][0][0][v], acc[rt][0][1][v], acc[rt][0][2][v], acc[rt][0][3][v]} +
                                ld4(PART + rho * DLD + 64 * ccg + 4 * j);
              *(f32x4_t*)(a.ctx + ((row0 + q0) * DH + rho) * DE + 64 * ccg + 4 * j) = o;
              if constexpr (RR == 1) *(f32x4_t*)(SC + rho * DLD + 64 * ccg + 4 * j) = o;     // (the score strip is dead: ctx rows for phase 7)
            }
          }
      }
    }
    __syncthreads(); TL(15);
  }
  }
  // ---- 7: o2[r][h*32+d] = Wv_h ctx[r][h] + bv * sp[r][h]   (A fragments straight from the ctx rows this workgroup just wrote)
  float* O2 = T;                                    // [16][DLD]  (tgt is dead since phase 3; BIG still holds the ctx rows of a single query)
  f32x4_t pf_co[UW];
  {
    auto aload = [&](int t, int k, int row) {
      const int h = (wave + DNW * t) >> 1;          // two 16-column tiles per head
      if constexpr (RR == 1) return ld4(SC + h * DLD + k + 4 * g);
      else return row < R ? ld4(a.ctx + ((row0 + row) * DH + h) * DE + k + 4 * g) : f4zero();
    };
    auto epi = [&](int t, int, f32x4_t acc) {
      const int n = tile_row(t), h = n / DHD;
      const float bias = PAR[P_BC + 2 * DE + n];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int r = 4 * g + v;
        const float y = r < R ? acc[v] + bias * SPL[r * DH + h] : 0.f;
        O2[r * DLD + n] = y;
        if (r < R) a.o2[(row0 + r) * DE + n] = y;
      }
    };
    stream_nt<16, 1, UW, RR>(2, bl_v, aload, epi, pf_v);
    stream_nt_prefetch<16, UW>(2, bl_co, pf_co);
  }
  __syncthreads(); TL(16);
  // ---- 8: r2 = t1 + o2 Wco^T + bco ; t2 = LayerNorm(r2)
  float* R2 = BIG + 16 * DLD;
  {
    auto aload = [&](int, int k, int row) { return ld4(O2 + row * DLD + k + 4 * g); };
    auto epi = [&](int t, int, f32x4_t acc) {
      const int n = tile_row(t);
      const float bias = PAR[P_BCO + n];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int r = 4 * g + v;
        const float y = r < R ? acc[v] + bias + X1[r * DLD + n] : 0.f;
        R2[r * DLD + n] = y;
        if (r < R) a.r2[(row0 + r) * DE + n] = y;
      }
    };
    stream_nt<16, 1, UW, RR>(2, bl_co, aload, epi, pf_co);
  }
  __syncthreads(); TL(17);
  ln_rows(R2, T, R, PAR + P_G1, PAR + P_B1, a.eps, a.t2 + row0 * DE, a.mean2 + row0, a.rstd2 + row0);
  TL(40);
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward of the attention block: the same workgroup-per-sample walk in reverse.  It produces the gradients that stay
// row-local (d tgt, d qpos, d source rows) and leaves, per row, the operands of every parameter gradient in the caller's
// buffers; `dec_attn_wgrad_kernel` contracts those over the rows of the whole batch in ONE more launch.
struct DecAttnBwdArgs {
  int B, R, Lk, kv_rows, kv_off;
  const float *Ws, *Wso, *g0, *Wc, *bc, *Wco, *g1;
  const lp_t* src16; const float* src32; long ldsrc;
  const float* kpos; long ldkp; int kpos_rows;
  const float* dm0; const float* dm1;
  // saved by the forward
  const float *qkv, *P0, *r1, *mean1, *rstd1, *qk, *P1, *r2, *mean2, *rstd2;
  // d(t2) = dt2 (or zero) + the sum of nslab slabs [nslab][B*R][E] (the FFN backward's partial sums over its hidden slices)
  const float* dt2; const float* dt2_slabs; int nslab; long slab_stride;
  // row-local gradients
  float* d_tgt; float* d_qpos;
  float* dsrc; long lddsrc; int dsrc_accumulate;       // fp32 gradient of the source rows [B*kv_rows, .]: written (0) or added to (1)
  // per-row operands of the parameter gradients + scratch
  float *dt2sum, *gx2, *d_r2, *d_o2, *dctx, *dqk, *dqpre, *d_t1, *gx1, *d_r1, *dqkv;
};

constexpr int LB_A0 = 0, LB_A1 = 16 * DLD, LB_A2 = 2 * 16 * DLD, LB_A3 = 3 * 16 * DLD, LB_DSP = 4 * 16 * DLD, LB_BIG = LB_DSP + 16 * DH;
constexpr int CRB = 16;               // (query, head) rows per chunk of the cross-attention backward: 2 queries x 8 heads
__host__ __device__ constexpr int dec_attn_bwd_lds_floats(int Lk) {
  const int a = 16 * QKV_LD + 2 * DH * 16 * 16 + 16 * DLD;   // qkv + P0' / dS0 + d(v) of the self-attention
  const int b = 2 * CRB * DLD + 2 * CRB * lkp_of(Lk);        // d(ctx) and qk chunks, dS and P' strips
  return LB_BIG + (a > b ? a : b);
}

// out[r][c] = sum_{n < N} D(r, n) W[n][c] for the 256 output columns: wave = (64-column group, half of the contraction range);
// the second half's partial sums meet the first's through TMP ([16][DLD]); epi(r, c, f32x4 of columns c .. c+3) for rows < 16
template <int RR, class ALoad, class Epi>
__device__ __forceinline__ void dgrad256(const float* W, long ldw, int N, ALoad aload, float* TMP, Epi epi) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
  const int cg = wave & 3, half = wave >> 2;
  const int mid = ((N / 2) + 3) & ~3;
  f32x4_t acc[1][1][4];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[0][0][c] = f4zero();
  auto bload = [&](int n, int) { return ld4(W + (long)min(n, N - 1) * ldw + 64 * cg + 4 * j); };
  auto al = [&](int n, int row) { return n < N ? aload(row, n) : 0.f; };
  stream_nn<1, 1, RR ? 16 : 8, RR>(half ? mid : 0, half ? N : mid, bload, al, acc);
  if (half) {
#pragma unroll
    for (int v = 0; v < 4; ++v)
      *(f32x4_t*)(TMP + (4 * g + v) * DLD + 64 * cg + 4 * j) = (f32x4_t){acc[0][0][0][v], acc[0][0][1][v], acc[0][0][2][v], acc[0][0][3][v]};
  }
  __syncthreads();
  if (!half) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int r = 4 * g + v, c = 64 * cg + 4 * j;
      epi(r, c, (f32x4_t){acc[0][0][0][v], acc[0][0][1][v], acc[0][0][2][v], acc[0][0][3][v]} + ld4(TMP + r * DLD + c));
    }
  }
}

// LayerNorm backward of the R rows of the LDS tile DY ([16][DLD], overwritten with dx): x rows from global; writes dx, dy * xhat
__device__ __forceinline__ void ln_bwd_rows(float* DY, int R, const float* x_g, const float* mean_g, const float* rstd_g, const float* gamma,
                                            float* dx_g, float* gx_g, float* dy_g) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = wave; r < R; r += DNW) {
    const f32x4_t dy = ld4(DY + r * DLD + 4 * lane);
    const float mean = mean_g[r], rstd = rstd_g[r];
    const f32x4_t xh = (ld4(x_g + (long)r * DE + 4 * lane) - mean) * rstd;
    const f32x4_t dyg = dy * ld4(gamma + 4 * lane);
    const float c1 = wave_sum((dyg[0] + dyg[1]) + (dyg[2] + dyg[3])) * (1.f / DE);
    const f32x4_t t = dyg * xh;
    const float c2 = wave_sum((t[0] + t[1]) + (t[2] + t[3])) * (1.f / DE);
    const f32x4_t dx = (dyg - c1 - xh * c2) * rstd;
    *(f32x4_t*)(DY + r * DLD + 4 * lane) = dx;
    *(f32x4_t*)(dx_g + (long)r * DE + 4 * lane) = dx;
    *(f32x4_t*)(gx_g + (long)r * DE + 4 * lane) = dy * xh;
    if (dy_g) *(f32x4_t*)(dy_g + (long)r * DE + 4 * lane) = dy;
  }
}

template <int RR, bool S16>
__global__ __launch_bounds__(DNT) void dec_attn_bwd_kernel(DecAttnBwdArgs a) {
  constexpr int UW = RR ? 16 : 8;
  extern __shared__ float sm[];
  float* A0 = sm + LB_A0;
  float* A1 = sm + LB_A1;
  float* A2 = sm + LB_A2;
  float* A3 = sm + LB_A3;
  float* DSP = sm + LB_DSP;                        // [16][H]  d(sum_k P')
  float* BIG = sm + LB_BIG;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  const int R = a.R;
  const long row0 = (long)b * R;
  const float scale = 0.17677669529663687f;
  // ---- B1: d(t2) of this sample's rows -> LayerNorm-2 backward -> d(r2)
  for (int e = tid; e < 16 * (DE / 4); e += DNT) {
    const int r = e / (DE / 4), c = 4 * (e % (DE / 4));
    f32x4_t v = f4zero();
    if (r < R) {
      if (a.dt2) v = ld4(a.dt2 + (row0 + r) * DE + c);
      // (eight slices' loads in flight, added in slice order: one load per trip was 32 round trips in a row at the head of the
      // kernel -- with one query per sample a single wave does all of them)
      const float* sl = a.dt2_slabs + (row0 + r) * DE + c;
      int s = 0;
      for (; s + 8 <= a.nslab; s += 8) {
        f32x4_t t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = ld4(sl + (long)(s + i) * a.slab_stride);
#pragma unroll
        for (int i = 0; i < 8; ++i) v += t[i];
      }
      for (; s < a.nslab; ++s) v += ld4(sl + (long)s * a.slab_stride);
    }
    *(f32x4_t*)(A0 + r * DLD + c) = v;
    *(f32x4_t*)(A1 + r * DLD + c) = v;
    *(f32x4_t*)(A2 + r * DLD + c) = f4zero();      // rows >= R of every tile start as zeros
    *(f32x4_t*)(A3 + r * DLD + c) = f4zero();
  }
  __syncthreads();
  ln_bwd_rows(A1, R, a.r2 + row0 * DE, a.mean2 + row0, a.rstd2 + row0, a.g1, a.d_r2 + row0 * DE, a.gx2 + row0 * DE, a.dt2sum + row0 * DE);
  __syncthreads();                                 // A1 = d(r2)
  // ---- B2: d(o2) = d(r2) Wco
  dgrad256<RR>(a.Wco, DE, DE, [&](int r, int n) { return A1[r * DLD + n]; }, BIG, [&](int r, int c, f32x4_t v) {
    *(f32x4_t*)(A3 + r * DLD + c) = v;
    if (r < R) *(f32x4_t*)(a.d_o2 + (row0 + r) * DE + c) = v;
  });
  __syncthreads();                                 // A3 = d(o2)
  // ---- B3: d(ctx)[r][h] = Wv_h^T d(o2)_h[r] (wave = head) -> global; d(sp)[r][h] = d(o2)_h[r] . bv_h
  {
    const int h = wave;
    f32x4_t acc[1][4][4];
#pragma unroll
    for (int cg = 0; cg < 4; ++cg)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[0][cg][c] = f4zero();
    const float* Wv = a.Wc + (long)(2 * DE + h * DHD) * DE;
    auto bload = [&](int n, int cg) { return ld4(Wv + (long)n * DE + 64 * cg + 4 * j); };
    auto aload = [&](int n, int row) { return A3[row * DLD + h * DHD + n]; };
    stream_nn<1, 4, 4, RR>(0, DHD, bload, aload, acc);
#pragma unroll
    for (int cg = 0; cg < 4; ++cg)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int r = 4 * g + v;
        if (r < R) {
          const f32x4_t o = (f32x4_t){acc[0][cg][0][v], acc[0][cg][1][v], acc[0][cg][2][v], acc[0][cg][3][v]};
          *(f32x4_t*)(a.dctx + ((row0 + r) * DH + h) * DE + 64 * cg + 4 * j) = o;
          if constexpr (RR == 1) *(f32x4_t*)(BIG + h * DLD + 64 * cg + 4 * j) = o;      // one query: straight into the chunk tile (DCX)
        }
      }
    if (tid < 16 * DH) {
      const int r = tid / DH, hh = tid % DH;
      float s = 0.f;
      for (int d = 0; d < DHD; ++d) s = fmaf(A3[r * DLD + hh * DHD + d], a.bc[2 * DE + hh * DHD + d], s);
      DSP[r * DH + hh] = s;
    }
  }
  __syncthreads();
  // ---- B4: cross-attention backward, 2 queries (16 (query, head) rows) at a time
  const int Lk = a.Lk, LKP = lkp_of(Lk);
  float* DCX = BIG;                                // [CRB][DLD]  d(ctx) chunk
  float* QKC = BIG + CRB * DLD;                    // [CRB][DLD]  qk chunk
  float* DS = BIG + 2 * CRB * DLD;                 // [CRB][LKP]  d(P') then dS
  float* PP = DS + CRB * LKP;                      // [CRB][LKP]  P'
  SrcRows<S16> src;
  src.s16 = a.src16 ? a.src16 + ((long)b * a.kv_rows + a.kv_off) * a.ldsrc : nullptr;
  src.s32 = a.src32 ? a.src32 + ((long)b * a.kv_rows + a.kv_off) * a.ldsrc : nullptr;
  src.ld = a.ldsrc;
  src.kpos = a.kpos + (long)b * a.kpos_rows * a.ldkp;
  src.ldkp = a.ldkp;
  const int ktiles = (Lk + 15) >> 4;
  float* dsrc_b = a.dsrc ? a.dsrc + ((long)b * a.kv_rows + a.kv_off) * a.lddsrc : nullptr;
  if (dsrc_b && !a.dsrc_accumulate) {             // source rows of this sample that are no keys (the CLS row of the image memory)
    for (int e = tid; e < (a.kv_rows - Lk) * (DE / 4); e += DNT) {
      int r = e / (DE / 4);
      const int c = 4 * (e % (DE / 4));
      if (r >= a.kv_off) r += Lk;
      *(f32x4_t*)(a.dsrc + ((long)b * a.kv_rows + r) * a.lddsrc + c) = f4zero();
    }
  }
  if constexpr (RR == 1) {
    // one query: the 8 (query, head) rows are the heads; every pass over the source rows on the VALU with whole-row loads
    float* DC8 = BIG;                              // [8][DLD]  d(ctx)   (written by B3)
    float* QK8 = BIG + 8 * DLD;                    // [8][DLD]  qk
    float* DST = BIG + 16 * DLD;                   // [Lk][8]   dS transposed
    float* PPT = DST + 8 * LKP;                    // [Lk][8]   P' transposed
    float* DS8 = PPT + 8 * LKP;                    // [8][LKP]  d(P'), dead once transposed: the waves' partial sums take its place
    float* PARTS = DS8;                            // [4][8][256]
    for (int e = tid; e < 8 * (DE / 4); e += DNT) {
      const int h = e / (DE / 4), c = 4 * (e % (DE / 4));
      *(f32x4_t*)(QK8 + h * DLD + c) = ld4(a.qk + (row0 * DH + h) * DE + c);
    }
    __syncthreads();
    rowdot8<S16, false, 8>(src, DC8, Lk, [&](int h, int kk, float v) { DS8[h * LKP + kk] = v + DSP[h]; });
    __syncthreads();
    {
      const int h = wave;
      const float* dsr = DS8 + h * LKP;
      const long pg = ((long)b * DH + h) * Lk;
      float rs = 0.f;
      // this lane's probabilities and dropout multipliers (the launcher admits Lk <= 448: seven keys per lane): fourteen loads in
      // flight once, instead of two per trip of both loops
      float pv[7], dmv[7];
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const long o = pg + min(lane + 64 * i, Lk - 1);
        pv[i] = a.P1[o];
        dmv[i] = a.dm1 ? a.dm1[o] : 1.f;
      }
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const int kk = lane + 64 * i;
        if (kk < Lk) rs = fmaf(pv[i], dsr[kk] * dmv[i], rs);
      }
      rs = wave_sum(rs);
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const int kk = lane + 64 * i;
        if (kk < Lk) {
          PPT[kk * 8 + h] = pv[i] * dmv[i];
          DST[kk * 8 + h] = pv[i] * (dsr[kk] * dmv[i] - rs);
        }
      }
    }
    __syncthreads();
    f32x4_t qacc[8];
#pragma unroll
    for (int h = 0; h < 8; ++h) qacc[h] = f4zero();
    rowacc8<S16, true, 8>(src, DST, Lk, qacc);
    // d(src)[kk] = sum_h P'[h][kk] d(ctx)[h] + dS[h][kk] qk[h]: a wave per row, 4 columns per lane (its 16 fragment registers stay)
    if (dsrc_b) {
      f32x4_t fd[8], fq[8];
#pragma unroll
      for (int h = 0; h < 8; ++h) { fd[h] = ld4(DC8 + h * DLD + 4 * lane); fq[h] = ld4(QK8 + h * DLD + 4 * lane); }
      for (int kk = wave; kk < Lk; kk += DNW) {
        float* p = dsrc_b + (long)kk * a.lddsrc + 4 * lane;
        f32x4_t o = a.dsrc_accumulate ? ld4(p) : f4zero();
        const f32x4_t p0 = ld4(PPT + kk * 8), p1 = ld4(PPT + kk * 8 + 4), s0 = ld4(DST + kk * 8), s1 = ld4(DST + kk * 8 + 4);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          o += p0[h] * fd[h] + s0[h] * fq[h];
          o += p1[h] * fd[4 + h] + s1[h] * fq[4 + h];
        }
        *(f32x4_t*)p = o;
      }
    }
    __syncthreads();                               // (DS8 is dead: PARTS may take its place)
    rowacc8_combine(qacc, PARTS, A0, a.dqk + row0 * DH * DE);           // d(qk) rows: LDS (A0 rows 0 .. 7) + global
    __syncthreads();
  } else {
  for (int q0 = 0; q0 < R; q0 += 2) {
    const int nrho = min(2, R - q0) * DH;
    for (int e = tid; e < CRB * (DE / 4); e += DNT) {
      const int rho = e / (DE / 4), c = 4 * (e % (DE / 4));
      const bool ok = rho < nrho;
      if (RR == 0 || !ok) *(f32x4_t*)(DCX + rho * DLD + c) = ok ? ld4(a.dctx + ((row0 + q0) * DH + rho) * DE + c) : f4zero();
      *(f32x4_t*)(QKC + rho * DLD + c) = ok ? ld4(a.qk + ((row0 + q0) * DH + rho) * DE + c) : f4zero();
    }
    __syncthreads();
    // d(P')[rho][kk] = d(ctx)[rho] . val[kk] + d(sp)[rho]
    {
      const int mine = (ktiles - wave + DNW - 1) / DNW;
      auto krow = [&](int t) { return min(16 * (wave + DNW * t) + j, Lk - 1); };
      auto bload = [&](int t, int k) { return src.val(krow(t), k + 4 * g); };
      auto aload = [&](int, int k, int row) { return ld4(DCX + row * DLD + k + 4 * g); };
      auto epi = [&](int t, int, f32x4_t acc) {
        const int kk = 16 * (wave + DNW * t) + j;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int rho = 4 * g + v;
          DS[rho * LKP + kk] = acc[v] + DSP[(q0 + rho / DH) * DH + rho % DH];      // (rows without a query are zeroed below)
        }
      };
      stream_nt<16, 1, 8, 0>(mine, bload, aload, epi);
    }
    __syncthreads();
    for (int rho = wave; rho < CRB; rho += DNW) {
      float* ds = DS + rho * LKP;
      float* pp = PP + rho * LKP;
      if (rho >= nrho) {
        for (int kk = lane; kk < LKP - 4; kk += 64) { ds[kk] = 0.f; pp[kk] = 0.f; }
        continue;
      }
      const int r = q0 + rho / DH, h = rho % DH;
      const long pg = (((long)b * DH + h) * R + r) * Lk;
      float rs = 0.f;
      for (int kk = lane; kk < Lk; kk += 64) {
        const float p = a.P1[pg + kk];
        const float dm = a.dm1 ? a.dm1[pg + kk] : 1.f;
        const float dp = ds[kk] * dm;
        pp[kk] = p * dm;
        ds[kk] = dp;
        rs = fmaf(p, dp, rs);
      }
      rs = wave_sum(rs);
      for (int kk = lane; kk < Lk; kk += 64) {
        const float p = a.P1[pg + kk];
        ds[kk] = p * (ds[kk] - rs);
      }
      for (int kk = Lk + lane; kk < LKP - 4; kk += 64) { ds[kk] = 0.f; pp[kk] = 0.f; }
    }
    __syncthreads();
    // d(qk)[rho][c] = sum_kk dS[rho][kk] key[kk][c]
    {
      const int cg = wave & 3, half = wave >> 2;
      const int kmid = ((ktiles + 1) >> 1) << 4;
      f32x4_t acc[1][1][4];
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[0][0][c] = f4zero();
      auto bload = [&](int n, int) { return src.key(min(n, Lk - 1), 64 * cg + 4 * j); };
      auto aload = [&](int n, int row) { return DS[row * LKP + n]; };
      stream_nn<1, 1, 8, 0>(half ? kmid : 0, half ? (ktiles << 4) : kmid, bload, aload, acc);
      if (half) {
#pragma unroll
        for (int v = 0; v < 4; ++v)
          *(f32x4_t*)(A2 + (4 * g + v) * DLD + 64 * cg + 4 * j) = (f32x4_t){acc[0][0][0][v], acc[0][0][1][v], acc[0][0][2][v], acc[0][0][3][v]};
      }
      __syncthreads();
      if (!half) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int rho = 4 * g + v;
          if (rho < nrho) {
            const f32x4_t o = (f32x4_t){acc[0][0][0][v], acc[0][0][1][v], acc[0][0][2][v], acc[0][0][3][v]} + ld4(A2 + rho * DLD + 64 * cg + 4 * j);
            *(f32x4_t*)(a.dqk + ((row0 + q0) * DH + rho) * DE + 64 * cg + 4 * j) = o;
            if constexpr (RR == 1) *(f32x4_t*)(A0 + rho * DLD + 64 * cg + 4 * j) = o;       // one query: d(qk) rows for B5 (A0 is free here)
          }
        }
      }
    }
    // d(src)[kk][c] += sum_rho P'[rho][kk] d(ctx)[rho][c] + dS[rho][kk] qk[rho][c]: (key tile, 64-column group) pairs over the waves
    if (dsrc_b) {
      const bool add = a.dsrc_accumulate || q0 > 0;
      for (int pr = wave; pr < ktiles * 4; pr += DNW) {
        const int kt = pr >> 2, cg = pr & 3;
        f32x4_t acc[4] = {f4zero(), f4zero(), f4zero(), f4zero()};
#pragma unroll
        for (int s = 0; s < CRB / 4; ++s) {
          const int rho = 4 * s + g;
          const float ap = PP[rho * LKP + 16 * kt + j], as = DS[rho * LKP + 16 * kt + j];
          const f32x4_t bd = ld4(DCX + rho * DLD + 64 * cg + 4 * j), bq = ld4(QKC + rho * DLD + 64 * cg + 4 * j);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap, bd[c], acc[c], 0, 0, 0);
            acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(as, bq[c], acc[c], 0, 0, 0);
          }
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int kk = 16 * kt + 4 * g + v;
          if (kk < Lk) {
            float* p = dsrc_b + (long)kk * a.lddsrc + 64 * cg + 4 * j;
            f32x4_t o = (f32x4_t){acc[0][v], acc[1][v], acc[2][v], acc[3][v]};
            if (add) o += ld4(p);
            *(f32x4_t*)p = o;
          }
        }
      }
    }
    __syncthreads();
  }
  }
  // ---- B5: d(qc)[r][h*32+d] = Wk_h d(qk)[r][h] ; d(q pre-scale) = d(qc) * 32^-1/2
  {
    const float* Wk = a.Wc + (long)DE * DE;
    auto bload = [&](int t, int k) { return ld4(Wk + (long)(16 * (wave + DNW * t) + j) * DE + k + 4 * g); };
    auto aload = [&](int t, int k, int row) {
      const int h = (wave + DNW * t) >> 1;
      if constexpr (RR == 1) return ld4(A0 + h * DLD + k + 4 * g);
      else return row < R ? ld4(a.dqk + ((row0 + row) * DH + h) * DE + k + 4 * g) : f4zero();
    };
    auto epi = [&](int t, int, f32x4_t acc) {
      const int n = 16 * (wave + DNW * t) + j;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int r = 4 * g + v;
        const float y = r < R ? acc[v] * scale : 0.f;
        A2[r * DLD + n] = y;
        if (r < R) a.dqpre[(row0 + r) * DE + n] = y;
      }
    };
    stream_nt<16, 1, UW, RR>(2, bload, aload, epi);
  }
  __syncthreads();                                 // A2 = d(q pre-scale)
  // ---- B6: d(t1 + qpos) = d(q) Wq ; d(t1) = that + d(r2)
  dgrad256<RR>(a.Wc, DE, DE, [&](int r, int n) { return A2[r * DLD + n]; }, BIG, [&](int r, int c, f32x4_t v) {
    *(f32x4_t*)(A3 + r * DLD + c) = v;
    *(f32x4_t*)(A0 + r * DLD + c) = v + ld4(A1 + r * DLD + c);
  });
  __syncthreads();                                 // A3 = d(xq2), A0 = d(t1)
  // ---- B7: LayerNorm-1 backward -> d(r1)
  ln_bwd_rows(A0, R, a.r1 + row0 * DE, a.mean1 + row0, a.rstd1 + row0, a.g0, a.d_r1 + row0 * DE, a.gx1 + row0 * DE, a.d_t1 + row0 * DE);
  __syncthreads();                                 // A0 = d(r1)
  // ---- B8: d(o) = d(r1) Wso
  dgrad256<RR>(a.Wso, DE, DE, [&](int r, int n) { return A0[r * DLD + n]; }, BIG, [&](int r, int c, f32x4_t v) {
    *(f32x4_t*)(A2 + r * DLD + c) = v;
  });
  __syncthreads();                                 // A2 = d(o)
  // ---- B9: self-attention backward on the sample's R x R scores
  float* QKV = BIG;                                // [16][QKV_LD]
  float* PD = BIG + 16 * QKV_LD;                   // [H][16][16]  P0' = P0 * dropout
  float* DS0 = PD + DH * 16 * 16;                  // [H][16][16]  d(P0') then dS0
  float* DV = DS0 + DH * 16 * 16;                  // [16][DLD]    d(v)
  for (int e = tid; e < 16 * (3 * DE / 4); e += DNT) {
    const int r = e / (3 * DE / 4), c = 4 * (e % (3 * DE / 4));
    *(f32x4_t*)(QKV + r * QKV_LD + c) = r < R ? ld4(a.qkv + (row0 + r) * (3 * DE) + c) : f4zero();
  }
  __syncthreads();
  for (int e = tid; e < DH * R * R; e += DNT) {
    const int h = e / (R * R), r = (e / R) % R, r2 = e % R;
    const float* dov = A2 + r * DLD + h * DHD;
    const float* v = QKV + r2 * QKV_LD + 2 * DE + h * DHD;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < DHD; ++d) s = fmaf(dov[d], v[d], s);
    const long pg = (((long)b * DH + h) * R + r) * R + r2;
    const float dm = a.dm0 ? a.dm0[pg] : 1.f;
    PD[(h * 16 + r) * 16 + r2] = a.P0[pg] * dm;
    DS0[(h * 16 + r) * 16 + r2] = s * dm;          // d(P0)
  }
  __syncthreads();
  for (int e = tid; e < DH * R; e += DNT) {
    const int h = e / R, r = e % R;
    float* ds = DS0 + (h * 16 + r) * 16;
    const long pg = (((long)b * DH + h) * R + r) * R;
    float rs = 0.f;
    for (int i = 0; i < R; ++i) rs = fmaf(a.P0[pg + i], ds[i], rs);
    for (int i = 0; i < R; ++i) ds[i] = a.P0[pg + i] * (ds[i] - rs) * scale;
  }
  __syncthreads();
  // d(q)[r][n] = sum_r' dS0[h][r][r'] k[r'][n] -> A0 ; d(k)[r'][n] = sum_r dS0[h][r][r'] q[r][n] -> A1 ; d(v)[r'][n] = sum_r P0'[h][r][r'] d(o)[r][n]
  for (int e = tid; e < R * DE; e += DNT) {
    const int r = e / DE, n = e % DE, h = n / DHD;
    float dq = 0.f, dk = 0.f, dv = 0.f;
    for (int i = 0; i < R; ++i) {
      dq = fmaf(DS0[(h * 16 + r) * 16 + i], QKV[i * QKV_LD + DE + n], dq);
      dk = fmaf(DS0[(h * 16 + i) * 16 + r], QKV[i * QKV_LD + n], dk);
      dv = fmaf(PD[(h * 16 + i) * 16 + r], A2[i * DLD + n], dv);
    }
    A0[r * DLD + n] = dq;
    A1[r * DLD + n] = dk;
    DV[r * DLD + n] = dv;
    float* o = a.dqkv + (row0 + r) * (3 * DE);
    o[n] = dq; o[DE + n] = dk; o[2 * DE + n] = dv;
  }
  for (int e = tid; e < (16 - R) * DE; e += DNT) {
    const int r = R + e / DE, n = e % DE;
    A0[r * DLD + n] = 0.f; A1[r * DLD + n] = 0.f; DV[r * DLD + n] = 0.f;
  }
  __syncthreads();
  // ---- B10: d(tgt + qpos) = [d(q) | d(k)] Ws[0:2E] ; d(tgt) = that + d(v) Ws[2E:] + d(r1) ; d(qpos) = that + d(xq2)
  {
    const int cg = wave & 3, half = wave >> 2;
    f32x4_t axq[1][1][4], av[1][1][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { axq[0][0][c] = f4zero(); av[0][0][c] = f4zero(); }
    auto bload = [&](int n, int) { return ld4(a.Ws + (long)n * DE + 64 * cg + 4 * j); };
    // half 0: d(q) rows of Ws (n in [0, 256)) and the first half of d(v)'s; half 1: d(k) rows and the second half of d(v)'s
    auto aqk = [&](int n, int row) { return n < DE ? A0[row * DLD + n] : A1[row * DLD + n - DE]; };
    auto avl = [&](int n, int row) { return DV[row * DLD + n - 2 * DE]; };
    if constexpr (RR == 0) stream_nn<1, 1, 8, 0>(half ? DE : 0, half ? 2 * DE : DE, bload, aqk, axq);      // (one query: d(q) = d(k) = 0)
    stream_nn<1, 1, RR ? 16 : 8, RR>(2 * DE + (half ? DE / 2 : 0), 2 * DE + (half ? DE : DE / 2), bload, avl, av);
    float* TQ = QKV;                               // the qkv tile is dead (every wave is past the loop above): [16][DLD] x 2
    float* TV = QKV + 16 * DLD;
    __syncthreads();
    if (half) {
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        *(f32x4_t*)(TQ + (4 * g + v) * DLD + 64 * cg + 4 * j) = (f32x4_t){axq[0][0][0][v], axq[0][0][1][v], axq[0][0][2][v], axq[0][0][3][v]};
        *(f32x4_t*)(TV + (4 * g + v) * DLD + 64 * cg + 4 * j) = (f32x4_t){av[0][0][0][v], av[0][0][1][v], av[0][0][2][v], av[0][0][3][v]};
      }
    }
    __syncthreads();
    if (!half) {
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int r = 4 * g + v, c = 64 * cg + 4 * j;
        if (r < R) {
          const f32x4_t xq = (f32x4_t){axq[0][0][0][v], axq[0][0][1][v], axq[0][0][2][v], axq[0][0][3][v]} + ld4(TQ + r * DLD + c);
          const f32x4_t vv = (f32x4_t){av[0][0][0][v], av[0][0][1][v], av[0][0][2][v], av[0][0][3][v]} + ld4(TV + r * DLD + c);
          *(f32x4_t*)(a.d_tgt + (row0 + r) * DE + c) = xq + vv + ld4(a.d_r1 + (row0 + r) * DE + c);
          *(f32x4_t*)(a.d_qpos + (row0 + r) * DE + c) = xq + ld4(A3 + r * DLD + c);
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// Parameter gradients of the attention block: every one is a contraction over the rows of the whole batch,
//     dW[n][k] = sum_row A(row, n) * Bm(row, k)      (bias / LayerNorm gradients: one output row, A = 1 or a per-row factor)
// of operands the row-owning kernel left behind.  38 such problems in ONE launch: a wave owns a 16 x 64 output tile (lane (j, g):
// columns k0 + 4j .. +3 as four accumulators, one 16-byte load of Bm per 4 rows), sums over the rows in a fixed order (the
// gradients are bit-reproducible), four tiles per workgroup.
struct DecWgradArgs {
  int MR;                                     // rows = B * R
  const float *tgt, *qpos, *t1, *o, *o2, *ctx, *sp, *qc;                                        // forward
  const float *dqkv, *d_r1, *gx1, *d_t1, *dqpre, *dqk, *d_o2, *d_r2, *gx2, *dt2sum;             // left by dec_attn_bwd_kernel
  float *dWs, *dbs, *dWso, *dbso, *dg0, *db0, *dWc, *dbc, *dWco, *dbco, *dg1, *db1;
};
constexpr int DWG_NP = 38;
struct WP {
  const float* A; long lda; const float* avec; long savec;      // A(row, n) = A[row * lda + n]; one-row problems: avec[row * savec] or 1
  const float* Bm; const float* B2; long ldb;                    // Bm(row, k) (+ B2(row, k))
  float* C; long ldc; int Nn, Kc; int zero;
};
__host__ __device__ inline WP dec_wgrad_problem(const DecWgradArgs& a, int pid) {
  WP p = {nullptr, 0, nullptr, 0, nullptr, nullptr, 0, nullptr, DE, 1, DE, 0};
  auto mat = [&](const float* A, long lda, const float* Bm, const float* B2, long ldb, float* C, int Nn) {
    p.A = A; p.lda = lda; p.Bm = Bm; p.B2 = B2; p.ldb = ldb; p.C = C; p.Nn = Nn;
  };
  auto vec = [&](const float* Bm, long ldb, float* C, int Kc) { p.Bm = Bm; p.ldb = ldb; p.C = C; p.Kc = Kc; p.Nn = 1; };
  if (pid == 0) mat(a.dqkv, 3 * DE, a.tgt ? a.tgt : a.qpos, a.tgt ? a.qpos : nullptr, DE, a.dWs, 2 * DE);      // (tgt == NULL: zeros)
  else if (pid == 1) { mat(a.dqkv + 2 * DE, 3 * DE, a.tgt, nullptr, DE, a.dWs + 2 * DE * DE, DE); p.zero = a.tgt == nullptr; }
  else if (pid == 2) vec(a.dqkv, 3 * DE, a.dbs, 3 * DE);
  else if (pid == 3) mat(a.d_r1, DE, a.o, nullptr, DE, a.dWso, DE);
  else if (pid == 4) vec(a.d_r1, DE, a.dbso, DE);
  else if (pid == 5) vec(a.gx1, DE, a.dg0, DE);
  else if (pid == 6) vec(a.d_t1, DE, a.db0, DE);
  else if (pid == 7) mat(a.dqpre, DE, a.t1, a.qpos, DE, a.dWc, DE);
  else if (pid == 8) vec(a.dqpre, DE, a.dbc, DE);
  else if (pid < 17) { const int h = pid - 9; mat(a.qc + h * DHD, DE, a.dqk + h * DE, nullptr, DH * DE, a.dWc + (long)(DE + h * DHD) * DE, DHD); }
  else if (pid == 17) { vec(nullptr, 0, a.dbc + DE, DE); p.zero = 1; }      // the key bias moves every score of a row alike: gradient 0
  else if (pid < 26) { const int h = pid - 18; mat(a.d_o2 + h * DHD, DE, a.ctx + h * DE, nullptr, DH * DE, a.dWc + (long)(2 * DE + h * DHD) * DE, DHD); }
  else if (pid < 34) { const int h = pid - 26; vec(a.d_o2 + h * DHD, DE, a.dbc + 2 * DE + h * DHD, DHD); p.avec = a.sp + h; p.savec = DH; }
  else if (pid == 34) mat(a.d_r2, DE, a.o2, nullptr, DE, a.dWco, DE);
  else if (pid == 35) vec(a.d_r2, DE, a.dbco, DE);
  else if (pid == 36) vec(a.gx2, DE, a.dg1, DE);
  else vec(a.dt2sum, DE, a.db1, DE);
  return p;
}
__host__ __device__ inline int wp_tiles(const WP& p) { return ((p.Nn + 15) / 16) * ((p.Kc + 63) / 64); }
struct DecWgradStarts { int start[DWG_NP + 1]; };

__device__ __forceinline__ void wgrad_tile(const WP& p, int tile, int MR) {
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int kts = (p.Kc + 63) / 64;
  const int n0 = 16 * (tile / kts), k0 = 64 * (tile % kts);
  const bool kok = k0 + 4 * j < p.Kc;
  f32x4_t acc[4] = {f4zero(), f4zero(), f4zero(), f4zero()};
  if (!p.zero) {
    const bool one_row = p.Nn == 1;
    const bool nok = one_row ? j == 0 : n0 + j < p.Nn;
    auto la = [&](int r) -> float {
      if (r >= MR || !nok) return 0.f;
      if (one_row) return p.avec ? p.avec[(long)r * p.savec] : 1.f;
      return p.A[(long)r * p.lda + n0 + j];
    };
    auto lb = [&](int r) -> f32x4_t {
      if (r >= MR || !kok) return f4zero();
      f32x4_t v = ld4(p.Bm + (long)r * p.ldb + k0 + 4 * j);
      if (p.B2) v += ld4(p.B2 + (long)r * p.ldb + k0 + 4 * j);
      return v;
    };
    constexpr int U = 8;
    float aq[U], an[U];
    f32x4_t bq[U], bn[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { aq[u] = la(4 * u + g); bq[u] = lb(4 * u + g); }
    for (int r0 = 0; r0 < MR; r0 += 4 * U) {
#pragma unroll
      for (int u = 0; u < U; ++u) { an[u] = la(r0 + 4 * (U + u) + g); bn[u] = lb(r0 + 4 * (U + u) + g); }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[u], bq[u][c], acc[c], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < U; ++u) { aq[u] = an[u]; bq[u] = bn[u]; }
    }
  }
  if (!kok) return;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int n = n0 + 4 * g + v;
    if (n < p.Nn) *(f32x4_t*)(p.C + (long)n * p.ldc + k0 + 4 * j) = (f32x4_t){acc[0][v], acc[1][v], acc[2][v], acc[3][v]};
  }
}

__global__ __launch_bounds__(256) void dec_attn_wgrad_kernel(DecWgradArgs a, DecWgradStarts st) {
  const int f = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (f >= st.start[DWG_NP]) return;
  int pid = 0;
  while (f >= st.start[pid + 1]) ++pid;
  const WP p = dec_wgrad_problem(a, pid);
  wgrad_tile(p, f - st.start[pid], a.MR);
}


// ---------------------------------------------------------------------------------------------------------------------
// The FFN of a decoder layer (Linear, ReLU, dropout, Linear, dropout, + identity, norm; detrex FFN at transformer.py:118-126)
// split over its HIDDEN units: workgroup s owns hidden units [64 s, 64 s + 64) for ALL rows, so it needs 64 rows of W1 and 64
// columns of W2 (128 KB of the layer's 4 MB) and nothing from another workgroup until the sum over the slices:
//   forward : h_s = relu(t2 W1_s^T + b1_s) * m1 ; slab[s] = h_s W2[:, s]^T          (dec_ffn_fwd_kernel, Fd / 64 workgroups)
//             r3 = t2 + m2 * (sum_s slab[s] + b2) ; t3 = LN(r3) ; hs = LN_post(t3)   (dec_ffn_finish_kernel, rows in parallel)
//   backward: d(r3) from d(t3), d(hs) (row-local, recomputed by every workgroup); d(h_s) = (d(r3) m2) W2[:, s] gated by the ReLU;
//             slab[s] = d(h_s) W1_s (the attention block's backward sums the slabs: d(t2) = d(r3) + sum_s slab[s]);
//             dW1_s, db1_s, dW2[:, s] complete inside the workgroup (it sees every row), in a fixed order.
constexpr int FS = 64;             // hidden units per workgroup
constexpr int FLD = FS + 4;
struct DecFfnArgs {
  int M, Fd;
  const float *t2, *W1, *b1, *W2, *m1;      // m1 [M, Fd] dropout multipliers or null
  float* h1d;                               // [M, Fd]   relu(.) * m1, saved for the backward
  float* slabs;                             // [Fd / 64][M][E]
};

__global__ __launch_bounds__(256) void dec_ffn_fwd_kernel(DecFfnArgs a) {
  __shared__ float X[64 * DLD];
  __shared__ float HS[64 * FLD];
  const int s = blockIdx.x, f0 = s * FS, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  for (int m0 = 0; m0 < a.M; m0 += 64) {
    const int rows = min(64, a.M - m0);
    auto bload1 = [&](int, int k) { return ld4(a.W1 + (long)(f0 + 16 * wave + j) * DE + k + 4 * g); };
    auto bload2 = [&](int t, int k) { return ld4(a.W2 + (long)(16 * (wave + 4 * t) + j) * a.Fd + f0 + k + 4 * g); };
    f32x4_t pf1[16], pf2[16];
    stream_nt_prefetch<16, 16>(1, bload1, pf1);      // the wave's whole W1 tile and its four W2 tiles, requested before the rows arrive
    stream_nt_prefetch<4, 16>(4, bload2, pf2);
    // the 64 rows, the bias and the dropout multipliers: every load issued before the first wait (written as one-load loops /
    // inside the tile epilogue they were 16 + 20 memory round trips in a row per wave -- `global_load` / `s_waitcnt vmcnt(0)` /
    // `ds_write` per trip in the ISA -- in a 26 us kernel)
    {
      constexpr int NX = 64 * (DE / 4) / 256;
      f32x4_t xv[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        const int e = tid + 256 * i, r = e / (DE / 4), c = 4 * (e % (DE / 4));
        xv[i] = ld4(a.t2 + (long)(m0 + min(r, rows - 1)) * DE + c);
      }
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        const int e = tid + 256 * i, r = e / (DE / 4), c = 4 * (e % (DE / 4));
        *(f32x4_t*)(X + r * DLD + c) = r < rows ? xv[i] : f4zero();
      }
    }
    const float bias1 = a.b1[f0 + 16 * wave + j];
    float mk[4][4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int v = 0; v < 4; ++v)
        mk[rt][v] = a.m1 ? a.m1[(long)(m0 + min(16 * rt + 4 * g + v, rows - 1)) * a.Fd + f0 + 16 * wave + j] : 1.f;
    __syncthreads();
    {   // h_s: one 16-column tile per wave, 4 row tiles
      auto& bload = bload1;
      auto aload = [&](int, int k, int row) { return ld4(X + row * DLD + k + 4 * g); };
      auto epi = [&](int, int rt, f32x4_t acc) {
        const int f = 16 * wave + j;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int r = 16 * rt + 4 * g + v;
          float y = 0.f;
          if (r < rows) {
            y = fmaxf(acc[v] + bias1, 0.f);
            if (a.m1) y *= mk[rt][v];
            a.h1d[(long)(m0 + r) * a.Fd + f0 + f] = y;
          }
          HS[r * FLD + f] = y;
        }
      };
      stream_nt<16, 4, 16, 0>(1, bload, aload, epi, pf1);
    }
    __syncthreads();
    {   // slab[s][r][n] = sum_f h_s[r][f] W2[n][f0 + f]: 16 column tiles, 4 per wave
      auto& bload = bload2;
      auto aload = [&](int, int k, int row) { return ld4(HS + row * FLD + k + 4 * g); };
      auto epi = [&](int t, int rt, f32x4_t acc) {
        const int n = 16 * (wave + 4 * t) + j;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int r = 16 * rt + 4 * g + v;
          if (r < rows) a.slabs[((long)s * a.M + m0 + r) * DE + n] = acc[v];
        }
      };
      stream_nt<4, 4, 16, 0>(4, bload, aload, epi, pf2);
    }
    __syncthreads();
  }
}

struct DecFfnFinishArgs {
  int M, NS;
  const float *t2, *slabs, *b2, *m2, *g2, *b2n, *gP, *bP;     // m2 [M, E] or null; gP / bP null: no post-norm
  float *r3, *mean3, *rstd3, *t3, *hs, *meanP, *rstdP;
  float eps;
};
__global__ __launch_bounds__(256) void dec_ffn_finish_kernel(DecFfnFinishArgs a) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= a.M) return;
  f32x4_t y = ld4(a.b2 + 4 * lane);
  // (eight slices' loads in flight at a time; the sum keeps the slice order: one load per trip was NS round trips in a row)
  int s = 0;
  for (; s + 8 <= a.NS; s += 8) {
    f32x4_t t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = ld4(a.slabs + ((long)(s + i) * a.M + r) * DE + 4 * lane);
#pragma unroll
    for (int i = 0; i < 8; ++i) y += t[i];
  }
  for (; s < a.NS; ++s) y += ld4(a.slabs + ((long)s * a.M + r) * DE + 4 * lane);
  if (a.m2) y *= ld4(a.m2 + (long)r * DE + 4 * lane);
  y += ld4(a.t2 + (long)r * DE + 4 * lane);
  *(f32x4_t*)(a.r3 + (long)r * DE + 4 * lane) = y;
  auto ln = [&](f32x4_t v, const float* gm, const float* bt, float* mean_o, float* rstd_o) {
    const float mean = wave_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.f / DE);
    const f32x4_t d = v - mean;
    const float var = wave_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * (1.f / DE);
    const float rstd = rsqrtf(var + a.eps);
    if (lane == 0) { mean_o[r] = mean; rstd_o[r] = rstd; }
    return d * rstd * ld4(gm + 4 * lane) + ld4(bt + 4 * lane);
  };
  const f32x4_t t3 = ln(y, a.g2, a.b2n, a.mean3, a.rstd3);
  *(f32x4_t*)(a.t3 + (long)r * DE + 4 * lane) = t3;
  if (a.gP) *(f32x4_t*)(a.hs + (long)r * DE + 4 * lane) = ln(t3, a.gP, a.bP, a.meanP, a.rstdP);
}

struct DecFfnBwdArgs {
  int M, Fd;
  const float *d_t3, *d_hs;                                   // either may be null
  const float *r3, *mean3, *rstd3, *g2, *t3, *meanP, *rstdP, *gP, *m2;
  const float *W1, *W2, *h1d, *m1, *t2;
  float *d_r3, *gx3, *dy3, *gxP, *dr3m;                       // [M, E] row operands (written by slice 0)
  float* slabs;                                               // [Fd / 64][M][E]  d(t2) partial sums
  float *dW1, *db1, *dW2;
};
constexpr int FBR = 64;            // rows per pass of the backward (the benchmark's 64 query rows are ONE pass)

__global__ __launch_bounds__(256) void dec_ffn_bwd_kernel(DecFfnBwdArgs a) {
  __shared__ float DRM[FBR * DLD];      // d(r3) * m2
  __shared__ float HS[FBR * FLD];       // h1d slice
  __shared__ float DH[FBR * FLD];       // d(pre-activation) slice
  const int s = blockIdx.x, f0 = s * FS, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  const bool lead = s == 0;
  // parameter-gradient accumulators, alive over the row passes: dW1 rows [16 wave, +16) x 256 columns (4 groups);
  // dW2 rows n in 4 tiles {wave, wave + 4, ...} x the slice's 64 columns
  f32x4_t aw1[4][4], aw2[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) { aw1[i][c] = f4zero(); aw2[i][c] = f4zero(); }
  float ab1 = 0.f;                      // threads 0 .. 63: db1 of hidden unit f0 + tid
  auto bl_w2 = [&](int n, int) { return ld4(a.W2 + (long)n * a.Fd + f0 + 4 * j); };
  auto bl_w1 = [&](int f, int) { return ld4(a.W1 + (long)(f0 + f) * DE + 64 * wave + 4 * j); };
  for (int m0 = 0; m0 < a.M; m0 += FBR) {
    const int rows = min(FBR, a.M - m0);
    f32x4_t pf_w2[16][1];
    stream_nn_prefetch<1, 16>(0, DE, bl_w2, pf_w2);            // the first W2 rows, requested before the rows' LayerNorm backward
    // ---- row-local: dy3 = d(t3) + LN_post backward(d(hs)); d(r3) = LN3 backward(dy3); DRM = d(r3) * m2
    // Rows in batches of RB: ALL global loads of a batch first (unconditional, on clamped row indices), then the arithmetic and
    // the lead slice's stores.  Written row by row, hipcc kept every row's loads behind the previous row's stores (they may alias)
    // and behind its own branches: sixteen memory round trips in a row per wave (`s_waitcnt vmcnt(0)` five times per row in the
    // ISA) in front of the first MFMA of a 60 us kernel.
#ifndef DEC_FFN_RB
#define DEC_FFN_RB 2       // rows per wave and batch (4: the same 46.8 us against 51.1, with 72 spilled bytes)
#endif
    constexpr int RB = DEC_FFN_RB;
    const f32x4_t gPv = a.d_hs ? ld4(a.gP + 4 * lane) : f4zero(), g2v = ld4(a.g2 + 4 * lane);
#pragma unroll 1
    for (int rb = 0; rb < FBR / 4; rb += RB) {
      f32x4_t Ldy[RB], Ldh[RB], Lt3[RB], Lr3[RB], Lm2[RB];
      float LmP[RB], LrP[RB], Lm3[RB], Lr3s[RB];
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const long r = m0 + min(wave + 4 * (rb + i), rows - 1);
        Ldy[i] = a.d_t3 ? ld4(a.d_t3 + r * DE + 4 * lane) : f4zero();
        if (a.d_hs) {
          Ldh[i] = ld4(a.d_hs + r * DE + 4 * lane);
          Lt3[i] = ld4(a.t3 + r * DE + 4 * lane);
          LmP[i] = a.meanP[r]; LrP[i] = a.rstdP[r];
        }
        Lr3[i] = ld4(a.r3 + r * DE + 4 * lane);
        Lm3[i] = a.mean3[r]; Lr3s[i] = a.rstd3[r];
        if (a.m2) Lm2[i] = ld4(a.m2 + r * DE + 4 * lane);
      }
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const int rr = wave + 4 * (rb + i);
        const long r = m0 + rr;
        f32x4_t drm = f4zero();
        if (rr < rows) {
          f32x4_t dy = Ldy[i];
          if (a.d_hs) {
            const f32x4_t dh = Ldh[i];
            const float mean = LmP[i], rstd = LrP[i];
            const f32x4_t xh = (Lt3[i] - mean) * rstd;
            const f32x4_t dyg = dh * gPv;
            const float c1 = wave_sum((dyg[0] + dyg[1]) + (dyg[2] + dyg[3])) * (1.f / DE);
            const f32x4_t t = dyg * xh;
            const float c2 = wave_sum((t[0] + t[1]) + (t[2] + t[3])) * (1.f / DE);
            dy += (dyg - c1 - xh * c2) * rstd;
            if (lead) *(f32x4_t*)(a.gxP + r * DE + 4 * lane) = dh * xh;
          }
          const float mean = Lm3[i], rstd = Lr3s[i];
          const f32x4_t xh = (Lr3[i] - mean) * rstd;
          const f32x4_t dyg = dy * g2v;
          const float c1 = wave_sum((dyg[0] + dyg[1]) + (dyg[2] + dyg[3])) * (1.f / DE);
          const f32x4_t t = dyg * xh;
          const float c2 = wave_sum((t[0] + t[1]) + (t[2] + t[3])) * (1.f / DE);
          const f32x4_t dr = (dyg - c1 - xh * c2) * rstd;
          drm = a.m2 ? dr * Lm2[i] : dr;
          if (lead) {
            *(f32x4_t*)(a.d_r3 + r * DE + 4 * lane) = dr;
            *(f32x4_t*)(a.gx3 + r * DE + 4 * lane) = dy * xh;
            *(f32x4_t*)(a.dy3 + r * DE + 4 * lane) = dy;
            *(f32x4_t*)(a.dr3m + r * DE + 4 * lane) = drm;
          }
        }
        *(f32x4_t*)(DRM + rr * DLD + 4 * lane) = drm;
      }
    }
    f32x4_t m1v[4];                                  // the dropout multipliers of this lane's four rows, requested with the h1d rows
    {
      constexpr int NH = FBR * (FS / 4) / 256;
      f32x4_t hv[NH];
#pragma unroll
      for (int i = 0; i < NH; ++i) {
        const int e = tid + 256 * i, r = e / (FS / 4), c = 4 * (e % (FS / 4));
        hv[i] = ld4(a.h1d + (long)(m0 + min(r, rows - 1)) * a.Fd + f0 + c);
      }
#pragma unroll
      for (int v = 0; v < 4; ++v)
        m1v[v] = a.m1 ? ld4(a.m1 + (long)(m0 + min(16 * wave + 4 * g + v, rows - 1)) * a.Fd + f0 + 4 * j) : (f32x4_t){1.f, 1.f, 1.f, 1.f};
#pragma unroll
      for (int i = 0; i < NH; ++i) {
        const int e = tid + 256 * i, r = e / (FS / 4), c = 4 * (e % (FS / 4));
        *(f32x4_t*)(HS + r * FLD + c) = r < rows ? hv[i] : f4zero();
      }
    }
    __syncthreads();
    // ---- d(h_s)[r][f] = sum_n DRM[r][n] W2[n][f0 + f], gated by the ReLU: wave w owns row tile w (16 rows) over all n
    {
      f32x4_t acc[1][1][4];
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[0][0][c] = f4zero();
      auto aload = [&](int n, int row) { return DRM[(16 * wave + row) * DLD + n]; };
      stream_nn<1, 1, 16, 0>(0, DE, bl_w2, aload, acc, pf_w2);
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int rr = 16 * wave + 4 * g + v;
        f32x4_t d = (f32x4_t){acc[0][0][0][v], acc[0][0][1][v], acc[0][0][2][v], acc[0][0][3][v]};
        const f32x4_t h = ld4(HS + rr * FLD + 4 * j);
        if (a.m1 && rr < rows) d *= m1v[v];
#pragma unroll
        for (int c = 0; c < 4; ++c) d[c] = h[c] > 0.f ? d[c] : 0.f;
        *(f32x4_t*)(DH + rr * FLD + 4 * j) = d;
      }
    }
    f32x4_t pf_w1[16][1];
    stream_nn_prefetch<1, 16>(0, FS, bl_w1, pf_w1);
    __syncthreads();
    // ---- slab[s][r][k] = sum_f d(pre)[r][f] W1[f0 + f][k]: wave = 64-column group, the four row tiles
    {
      f32x4_t acc[4][1][4];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[rt][0][c] = f4zero();
      auto aload = [&](int f, int row) { return DH[row * FLD + f]; };
      stream_nn<4, 1, 16, 0>(0, FS, bl_w1, aload, acc, pf_w1);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int rr = 16 * rt + 4 * g + v;
          if (rr < rows)
            *(f32x4_t*)(a.slabs + ((long)s * a.M + m0 + rr) * DE + 64 * wave + 4 * j) =
                (f32x4_t){acc[rt][0][0][v], acc[rt][0][1][v], acc[rt][0][2][v], acc[rt][0][3][v]};
        }
    }
    // ---- parameter gradients of this pass's rows (rows >= `rows` are zero in every tile); the t2 fragments come straight from
    // global memory (64 rows x 1 KB, hot in L2): a fourth [64][256] LDS tile would not fit beside the other three
#pragma unroll 4
    for (int st = 0; st < FBR / 4; ++st) {
      const int rr = 4 * st + g;
      const float a1 = DH[rr * FLD + 16 * wave + j];             // dW1 rows f = 16 wave + j
      const float* t2r = a.t2 + (long)(m0 + min(rr, rows - 1)) * DE + 4 * j;
#pragma unroll
      for (int cg = 0; cg < 4; ++cg) {
        const f32x4_t bt = ld4(t2r + 64 * cg);
#pragma unroll
        for (int c = 0; c < 4; ++c) aw1[cg][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bt[c], aw1[cg][c], 0, 0, 0);
      }
      const f32x4_t bh = ld4(HS + rr * FLD + 4 * j);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float a2 = DRM[rr * DLD + 16 * (wave + 4 * t) + j];   // dW2 rows n = 16 (wave + 4 t) + j
#pragma unroll
        for (int c = 0; c < 4; ++c) aw2[t][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, bh[c], aw2[t][c], 0, 0, 0);
      }
    }
    if (tid < FS)
      for (int rr = 0; rr < FBR; ++rr) ab1 += DH[rr * FLD + tid];
    __syncthreads();
  }
#pragma unroll
  for (int cg = 0; cg < 4; ++cg)
#pragma unroll
    for (int v = 0; v < 4; ++v)
      *(f32x4_t*)(a.dW1 + (long)(f0 + 16 * wave + 4 * g + v) * DE + 64 * cg + 4 * j) =
          (f32x4_t){aw1[cg][0][v], aw1[cg][1][v], aw1[cg][2][v], aw1[cg][3][v]};
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int v = 0; v < 4; ++v)
      *(f32x4_t*)(a.dW2 + (long)(16 * (wave + 4 * t) + 4 * g + v) * a.Fd + f0 + 4 * j) =
          (f32x4_t){aw2[t][0][v], aw2[t][1][v], aw2[t][2][v], aw2[t][3][v]};
  if (tid < FS) a.db1[f0 + tid] = ab1;
}

// column sums over the rows of up to 8 [M, 256] operands (LayerNorm / bias gradients of the FFN): out[i][c] = sum_r x_i[r][c];
// one wave per (operand, 64-column group), fixed order
struct DecColsumArgs { int M, n; const float* x[8]; float* out[8]; };
__global__ __launch_bounds__(256) void dec_colsum_kernel(DecColsumArgs a) {
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (w >= a.n * 4) return;
  const int i = w >> 2, c = 64 * (w & 3) + lane;
  const float* x = a.x[i] + c;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int r = 0;
  for (; r + 4 <= a.M; r += 4) {
    s0 += x[(long)r * DE]; s1 += x[(long)(r + 1) * DE]; s2 += x[(long)(r + 2) * DE]; s3 += x[(long)(r + 3) * DE];
  }
  for (; r < a.M; ++r) s0 += x[(long)r * DE];
  a.out[i][c] = (s0 + s1) + (s2 + s3);
}


// ---------------------------------------------------------------------------------------------------------------------
// Query assembly around the TGQG layers (tgqs_kd_detr_head.py:385-411), the element-wise pieces between the fused layers as four
// small launches instead of ~25 framework ones (comparisons, where / maximum and their backward, broadcast adds, reductions):
//   text_filt : has_pad[b] = any(mask[b] == 1); filt[b] = has_pad ? max(text[b, T-1], text[b, T-2]) : text[b, T-1]   (quirk Q1: `~mask`
//               on an int64 mask is a bitwise NOT, which selects exactly these rows); kpm[b][t] = mask[b][t] != 0
//   query_mix : query_embed[b, q] = g[b, q] + filt[b] + qe[q];  tok[b, q] = query_embed[b, q] + cls[b]                 (Q5)
// and their gradients (torch.maximum's convention: a tie splits the gradient in halves).
struct TextFiltArgs { const float* text; const long long* mask; float* filt; unsigned char* kpm; int B, T; };
__global__ __launch_bounds__(256) void text_filt_fwd_kernel(TextFiltArgs a) {
  const int b = blockIdx.x, c = threadIdx.x;
  __shared__ int pad;
  if (c == 0) pad = 0;
  __syncthreads();
  for (int t = c; t < a.T; t += 256) {
    const long long m = a.mask[(long)b * a.T + t];
    a.kpm[(long)b * a.T + t] = m != 0;
    if (m == 1) pad = 1;                      // (benign race: every writer stores 1)
  }
  __syncthreads();
  const float x1 = a.text[((long)b * a.T + a.T - 1) * DE + c], x2 = a.text[((long)b * a.T + a.T - 2) * DE + c];
  a.filt[(long)b * DE + c] = pad ? fmaxf(x1, x2) : x1;
}
struct TextFiltBwdArgs { const float* text; const long long* mask; const float* dfilt; float* dtext; int B, T; };
__global__ __launch_bounds__(256) void text_filt_bwd_kernel(TextFiltBwdArgs a) {
  const int b = blockIdx.x, c = threadIdx.x;
  __shared__ int pad;
  if (c == 0) pad = 0;
  __syncthreads();
  for (int t = c; t < a.T; t += 256)
    if (a.mask[(long)b * a.T + t] == 1) pad = 1;
  __syncthreads();
  const long r1 = ((long)b * a.T + a.T - 1) * DE + c, r2 = ((long)b * a.T + a.T - 2) * DE + c;
  const float x1 = a.text[r1], x2 = a.text[r2], g = a.dfilt[(long)b * DE + c];
  for (int t = 0; t < a.T - 2; ++t) a.dtext[((long)b * a.T + t) * DE + c] = 0.f;
  float g1 = g, g2 = 0.f;
  if (pad) {
    if (x1 == x2) { g1 = 0.5f * g; g2 = 0.5f * g; }
    else if (x1 < x2) { g1 = 0.f; g2 = g; }
  }
  a.dtext[r1] = g1;
  a.dtext[r2] = g2;
}
struct QueryMixArgs { const float *g, *filt, *qe, *cls; float *qeo, *tok; int B, R; };
__global__ __launch_bounds__(256) void query_mix_fwd_kernel(QueryMixArgs a) {
  const int row = blockIdx.x, c = threadIdx.x, b = row / a.R, q = row % a.R;
  const float v = a.g[(long)row * DE + c] + a.filt[(long)b * DE + c] + a.qe[(long)q * DE + c];
  a.qeo[(long)row * DE + c] = v;
  a.tok[(long)row * DE + c] = v + a.cls[(long)b * DE + c];
}
// d(g)[row] = d(qeo)[row] + d(tok)[row]; d(filt)[b] = sum_q that; d(cls)[b] = sum_q d(tok)[b, q]: workgroup per sample;
// d(qe)[q] = sum_b d(g)[b, q]: workgroups B .. B + R - 1, after ... no: a second launch (it reads what the first wrote)
struct QueryMixBwdArgs { const float *dqeo, *dtok; float *dg, *dfilt, *dcls, *dqe; int B, R; };
__global__ __launch_bounds__(256) void query_mix_bwd_kernel(QueryMixBwdArgs a) {
  const int b = blockIdx.x, c = threadIdx.x;
  float sf = 0.f, sc = 0.f;
  for (int q = 0; q < a.R; ++q) {
    const long i = ((long)b * a.R + q) * DE + c;
    const float dt = a.dtok ? a.dtok[i] : 0.f, d = (a.dqeo ? a.dqeo[i] : 0.f) + dt;
    a.dg[i] = d;
    sf += d;
    sc += dt;
  }
  a.dfilt[(long)b * DE + c] = sf;
  a.dcls[(long)b * DE + c] = sc;
}
__global__ __launch_bounds__(256) void query_mix_bwd_qe_kernel(QueryMixBwdArgs a) {
  const int q = blockIdx.x, c = threadIdx.x;
  float s0 = 0.f, s1 = 0.f;
  int b = 0;
  for (; b + 2 <= a.B; b += 2) { s0 += a.dg[((long)b * a.R + q) * DE + c]; s1 += a.dg[((long)(b + 1) * a.R + q) * DE + c]; }
  if (b < a.B) s0 += a.dg[((long)b * a.R + q) * DE + c];
  a.dqe[(long)q * DE + c] = s0 + s1;
}

}  // namespace

// mirror of include/simvg_hip.h
struct simvg_dec_attn_args {
  int B, R, Lk, kv_rows, kv_off;
  const float* tgt; const float* qpos;
  const float *Ws, *bs, *Wso, *bso, *g0, *b0, *Wc, *bc, *Wco, *bco, *g1, *b1;
  const void* src16; const float* src32; long ldsrc;
  const float* kpos; long ldkp; int kpos_rows;
  const unsigned char* kpm;
  const float* dm0; const float* dm1;
  float *qkv, *P0, *o, *r1, *mean1, *rstd1, *t1, *qc, *qk, *P1, *ctx, *sp, *o2, *r2, *mean2, *rstd2, *t2;
  float eps;
};

extern "C" int simvg_dec_attn_fwd(const simvg_dec_attn_args* p, hipStream_t stream) {
  SIMVG_CHECK_ARG(p != nullptr, "dec_attn_fwd: null arguments");
  SIMVG_CHECK_ARG(p->B > 0 && p->R > 0 && p->R <= 16, "dec_attn_fwd: 1..16 queries per sample");
  SIMVG_CHECK_ARG(p->Lk > 0 && p->Lk <= 1024 && p->kv_rows >= p->kv_off + p->Lk, "dec_attn_fwd: 1..1024 keys inside the sample's source rows");
  SIMVG_CHECK_ARG((p->src16 != nullptr) != (p->src32 != nullptr), "dec_attn_fwd: exactly one of src16 / src32");
  SIMVG_CHECK_ARG(p->kpos != nullptr, "dec_attn_fwd: key_pos rows are required (pass zeros for none)");
  SIMVG_CHECK_ARG(p->ldsrc % 4 == 0 && p->ldkp % 4 == 0, "dec_attn_fwd: source / key_pos rows must be 16-byte aligned");
  static_assert(sizeof(simvg_dec_attn_args) == sizeof(DecAttnArgs), "C mirror of DecAttnArgs");
  DecAttnArgs a;
  memcpy(&a, p, sizeof(a));
  const size_t shm = (size_t)dec_attn_fwd_lds_floats(p->Lk) * sizeof(float);
  SIMVG_CHECK_ARG(shm <= 160 * 1024, "dec_attn_fwd: the score strip does not fit the 160 KiB LDS");
  // instantiations: one query per sample (num_queries = 1, the RefCOCO configs: the weight phases on the VALU -- a 16-row MFMA tile
  // would be 15/16 padding) or MFMA (GRefCOCO: 10 queries) x 16-bit or fp32 source rows
  typedef void (*kern_t)(DecAttnArgs);
  static const kern_t kerns[4] = {dec_attn_fwd_kernel<0, false>, dec_attn_fwd_kernel<0, true>, dec_attn_fwd_kernel<1, false>, dec_attn_fwd_kernel<1, true>};
  static bool once = [] {
    bool ok = true;
    for (kern_t k : kerns) ok = ok && hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
    return ok;
  }();
  (void)once;
  hipLaunchKernelGGL(kerns[(p->R == 1 ? 2 : 0) + (p->src16 ? 1 : 0)], dim3(p->B), dim3(DNT), shm, stream, a);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}

struct simvg_dec_attn_bwd_args {
  int B, R, Lk, kv_rows, kv_off;
  const float *Ws, *Wso, *g0, *Wc, *bc, *Wco, *g1;
  const void* src16; const float* src32; long ldsrc;
  const float* kpos; long ldkp; int kpos_rows;
  const float* dm0; const float* dm1;
  const float *qkv, *P0, *r1, *mean1, *rstd1, *qk, *P1, *r2, *mean2, *rstd2;
  const float* dt2; const float* dt2_slabs; int nslab; long slab_stride;
  float* d_tgt; float* d_qpos;
  float* dsrc; long lddsrc; int dsrc_accumulate;
  float *dt2sum, *gx2, *d_r2, *d_o2, *dctx, *dqk, *dqpre, *d_t1, *gx1, *d_r1, *dqkv;
};

extern "C" int simvg_dec_attn_bwd(const simvg_dec_attn_bwd_args* p, hipStream_t stream) {
  SIMVG_CHECK_ARG(p != nullptr, "dec_attn_bwd: null arguments");
  SIMVG_CHECK_ARG(p->B > 0 && p->R > 0 && p->R <= 16, "dec_attn_bwd: 1..16 queries per sample");
  SIMVG_CHECK_ARG(p->Lk > 0 && p->kv_rows >= p->kv_off + p->Lk, "dec_attn_bwd: the keys must lie inside the sample's source rows");
  SIMVG_CHECK_ARG((p->src16 != nullptr) != (p->src32 != nullptr), "dec_attn_bwd: exactly one of src16 / src32");
  SIMVG_CHECK_ARG(p->kpos != nullptr, "dec_attn_bwd: key_pos rows are required (pass zeros for none)");
  SIMVG_CHECK_ARG(p->ldsrc % 4 == 0 && p->ldkp % 4 == 0 && (!p->dsrc || p->lddsrc % 4 == 0),
                  "dec_attn_bwd: source / key_pos / d(source) rows must be 16-byte aligned");
  SIMVG_CHECK_ARG(p->nslab >= 0 && (p->nslab == 0 || p->dt2_slabs != nullptr), "dec_attn_bwd: nslab slabs need a pointer");
  static_assert(sizeof(simvg_dec_attn_bwd_args) == sizeof(DecAttnBwdArgs), "C mirror of DecAttnBwdArgs");
  DecAttnBwdArgs a;
  memcpy(&a, p, sizeof(a));
  const size_t shm = (size_t)dec_attn_bwd_lds_floats(p->Lk) * sizeof(float);
  SIMVG_CHECK_ARG(shm <= 160 * 1024 && p->Lk <= 448, "dec_attn_bwd: the score strips do not fit the 160 KiB LDS (Lk <= 448)");
  // instantiations: one query per sample (num_queries = 1, the RefCOCO configs: the weight phases on the VALU -- a 16-row MFMA tile
  // would be 15/16 padding) or MFMA (GRefCOCO: 10 queries) x 16-bit or fp32 source rows
  typedef void (*kern_t)(DecAttnBwdArgs);
  static const kern_t kerns[4] = {dec_attn_bwd_kernel<0, false>, dec_attn_bwd_kernel<0, true>, dec_attn_bwd_kernel<1, false>, dec_attn_bwd_kernel<1, true>};
  static bool once = [] {
    bool ok = true;
    for (kern_t k : kerns) ok = ok && hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
    return ok;
  }();
  (void)once;
  hipLaunchKernelGGL(kerns[(p->R == 1 ? 2 : 0) + (p->src16 ? 1 : 0)], dim3(p->B), dim3(DNT), shm, stream, a);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}

// the largest Lk simvg_dec_attn_fwd / _bwd hold in LDS (the caller picks the unfused kernels beyond it)
extern "C" int simvg_dec_attn_max_keys(void) {
  int lk = 16;
  while (dec_attn_bwd_lds_floats(lk + 16) * sizeof(float) <= 160 * 1024 && dec_attn_fwd_lds_floats(lk + 16) * sizeof(float) <= 160 * 1024) lk += 16;
  return lk;
}

struct simvg_dec_attn_wgrad_args {
  int MR;
  const float *tgt, *qpos, *t1, *o, *o2, *ctx, *sp, *qc;
  const float *dqkv, *d_r1, *gx1, *d_t1, *dqpre, *dqk, *d_o2, *d_r2, *gx2, *dt2sum;
  float *dWs, *dbs, *dWso, *dbso, *dg0, *db0, *dWc, *dbc, *dWco, *dbco, *dg1, *db1;
};

extern "C" int simvg_dec_attn_wgrad(const simvg_dec_attn_wgrad_args* p, hipStream_t stream) {
  SIMVG_CHECK_ARG(p != nullptr && p->MR > 0, "dec_attn_wgrad: no rows");
  static_assert(sizeof(simvg_dec_attn_wgrad_args) == sizeof(DecWgradArgs), "C mirror of DecWgradArgs");
  DecWgradArgs a;
  memcpy(&a, p, sizeof(a));
  DecWgradStarts st;
  int total = 0;
  for (int i = 0; i < DWG_NP; ++i) {
    st.start[i] = total;
    total += wp_tiles(dec_wgrad_problem(a, i));
  }
  st.start[DWG_NP] = total;
  hipLaunchKernelGGL(dec_attn_wgrad_kernel, dim3((total + 3) / 4), dim3(256), 0, stream, a, st);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}

struct simvg_dec_ffn_args { int M, Fd; const float *t2, *W1, *b1, *W2, *m1; float* h1d; float* slabs; };
extern "C" int simvg_dec_ffn_fwd(const simvg_dec_ffn_args* p, hipStream_t stream) {
  SIMVG_CHECK_ARG(p != nullptr && p->M > 0 && p->Fd > 0 && p->Fd % FS == 0, "dec_ffn_fwd: the hidden width must be a multiple of 64");
  static_assert(sizeof(simvg_dec_ffn_args) == sizeof(DecFfnArgs), "C mirror of DecFfnArgs");
  DecFfnArgs a;
  memcpy(&a, p, sizeof(a));
  hipLaunchKernelGGL(dec_ffn_fwd_kernel, dim3(p->Fd / FS), dim3(256), 0, stream, a);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}

struct simvg_dec_ffn_finish_args {
  int M, NS;
  const float *t2, *slabs, *b2, *m2, *g2, *b2n, *gP, *bP;
  float *r3, *mean3, *rstd3, *t3, *hs, *meanP, *rstdP;
  float eps;
};
extern "C" int simvg_dec_ffn_finish(const simvg_dec_ffn_finish_args* p, hipStream_t stream) {
  SIMVG_CHECK_ARG(p != nullptr && p->M > 0 && p->NS > 0, "dec_ffn_finish: no rows / no slabs");
  static_assert(sizeof(simvg_dec_ffn_finish_args) == sizeof(DecFfnFinishArgs), "C mirror of DecFfnFinishArgs");
  DecFfnFinishArgs a;
  memcpy(&a, p, sizeof(a));
  hipLaunchKernelGGL(dec_ffn_finish_kernel, dim3((p->M + 3) / 4), dim3(256), 0, stream, a);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}

struct simvg_dec_ffn_bwd_args {
  int M, Fd;
  const float *d_t3, *d_hs;
  const float *r3, *mean3, *rstd3, *g2, *t3, *meanP, *rstdP, *gP, *m2;
  const float *W1, *W2, *h1d, *m1, *t2;
  float *d_r3, *gx3, *dy3, *gxP, *dr3m;
  float* slabs;
  float *dW1, *db1, *dW2;
  /* the row sums the slices cannot own: db2 = sum dr3m, dg2 = sum gx3, db2n = sum dy3, dgP = sum gxP, dbP = sum d_hs */
  float *db2, *dg2, *db2n, *dgP, *dbP;
};
extern "C" int simvg_dec_ffn_bwd(const simvg_dec_ffn_bwd_args* p, hipStream_t stream) {
  SIMVG_CHECK_ARG(p != nullptr && p->M > 0 && p->Fd > 0 && p->Fd % FS == 0, "dec_ffn_bwd: the hidden width must be a multiple of 64");
  SIMVG_CHECK_ARG(p->d_t3 != nullptr || p->d_hs != nullptr, "dec_ffn_bwd: no incoming gradient");
  SIMVG_CHECK_ARG(!p->d_hs || (p->gP && p->gxP && p->dgP && p->dbP), "dec_ffn_bwd: d_hs needs the post-norm's operands");
  DecFfnBwdArgs a;
  static_assert(sizeof(DecFfnBwdArgs) + 5 * sizeof(float*) == sizeof(simvg_dec_ffn_bwd_args), "C mirror of DecFfnBwdArgs (+ 5 outputs)");
  memcpy(&a, p, sizeof(a));
  hipLaunchKernelGGL(dec_ffn_bwd_kernel, dim3(p->Fd / FS), dim3(256), 0, stream, a);
  SIMVG_LAUNCH_CHECK();
  DecColsumArgs c;
  c.M = p->M;
  c.n = 0;
  auto add = [&](const float* x, float* o) { c.x[c.n] = x; c.out[c.n] = o; ++c.n; };
  add(p->dr3m, p->db2);
  add(p->gx3, p->dg2);
  add(p->dy3, p->db2n);
  if (p->d_hs) { add(p->gxP, p->dgP); add(p->d_hs, p->dbP); }
  for (int i = c.n; i < 8; ++i) { c.x[i] = nullptr; c.out[i] = nullptr; }
  hipLaunchKernelGGL(dec_colsum_kernel, dim3(c.n), dim3(256), 0, stream, c);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}

extern "C" int simvg_text_filt_fwd(const float* text, const long long* mask, float* filt, unsigned char* kpm, int B, int T, hipStream_t stream) {
  SIMVG_CHECK_ARG(text && mask && filt && kpm && B > 0 && T >= 2, "text_filt_fwd: need at least two text rows per sample");
  hipLaunchKernelGGL(text_filt_fwd_kernel, dim3(B), dim3(256), 0, stream, TextFiltArgs{text, mask, filt, kpm, B, T});
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}
extern "C" int simvg_text_filt_bwd(const float* text, const long long* mask, const float* dfilt, float* dtext, int B, int T, hipStream_t stream) {
  SIMVG_CHECK_ARG(text && mask && dfilt && dtext && B > 0 && T >= 2, "text_filt_bwd: need at least two text rows per sample");
  hipLaunchKernelGGL(text_filt_bwd_kernel, dim3(B), dim3(256), 0, stream, TextFiltBwdArgs{text, mask, dfilt, dtext, B, T});
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}
extern "C" int simvg_query_mix_fwd(const float* g, const float* filt, const float* qe, const float* cls, float* qeo, float* tok, int B, int R,
                                   hipStream_t stream) {
  SIMVG_CHECK_ARG(g && filt && qe && cls && qeo && tok && B > 0 && R > 0, "query_mix_fwd: null operand");
  hipLaunchKernelGGL(query_mix_fwd_kernel, dim3(B * R), dim3(256), 0, stream, QueryMixArgs{g, filt, qe, cls, qeo, tok, B, R});
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}
extern "C" int simvg_query_mix_bwd(const float* dqeo, const float* dtok, float* dg, float* dfilt, float* dcls, float* dqe, int B, int R,
                                   hipStream_t stream) {
  SIMVG_CHECK_ARG((dqeo || dtok) && dg && dfilt && dcls && dqe && B > 0 && R > 0, "query_mix_bwd: null operand");
  const QueryMixBwdArgs a{dqeo, dtok, dg, dfilt, dcls, dqe, B, R};
  hipLaunchKernelGGL(query_mix_bwd_kernel, dim3(B), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(query_mix_bwd_qe_kernel, dim3(R), dim3(256), 0, stream, a);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}
