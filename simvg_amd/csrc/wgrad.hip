// Weight gradient of the encoder's Linears, XCD-partitioned (gfx950):  dW[g][N,K] += s * dY[M,N]^T . X[M,K],
// db[g][N] += s * column sums of dY   (wgrad + bias gradient of torchscale's multiway nn.Linear: autograd of the calls at
// beit3_base.py:137-145,159; s = 1 / gradient scale of dY).
//
// The contraction runs over M = 26 944 token rows while the output is only N x K (2.4 M elements for fc1), so the
// operand panels dominate the traffic.  The first kernel (gemm.hip: 256x128 output tiles, rows cut into ~5 chunks, tiles
// spread over all XCDs) moved 2.4-2.8x its algorithmic bytes across the fabric: every XCD streamed the dY / X panels of
// its tile block over ALL rows, and a separate set of blocks re-read dY for the bias gradient.  Here
//   * each of the 8 XCDs (blockIdx % 8; a speed assumption only) owns a contiguous RANGE OF ROWS and computes the whole
//     N x K output for it: its 32 workgroups (one per CU, all resident, ONE round, no tail) walk those rows in lockstep,
//     so every dY / X row is fetched from HBM once per launch and the re-use happens in the XCD's L2;
//   * 384 x 192 (or 288 x 192 / 192 x 384) output tiles: 128 FLOP per staged byte instead of 85; 12 waves per workgroup
//     (3 per SIMD), each a 96 x 64 block (96 accumulator VGPRs): 20 transposed fragment reads feed 24 MFMAs per stage;
//   * the three waves of a SIMD rotate through LOAD / LOAD / MFMA roles a third of a stage apart (see the kernel);
//   * the bias gradient is taken from the dY stage that is in LDS anyway (each block sums 1/tiles_k of its columns);
//   * the 8 partial results (one per row partition and row group) leave through the LDS ring -- free after the last stage --
//     as 16-B stores into private fp32 slabs, and wgrad_slab_reduce_kernel adds a weight's slabs to dW in a fixed order
//     (batched: one launch for the four weights of an encoder layer, ops.WgradReduceBatch).  Round 3 added them to dW with
//     fp32 atomics: device-scope atomics are memory-side read-modify-writes whose cost follows the count, not the
//     coalescing (47 us of the 166 of the fc1 launch; plain stores of the same bytes 12 us + 16 us for the reduction,
//     profiles/r04_sweeps.md section 3) and whose order made dW differ run to run.  SIMVG_WG_SLABS=0 keeps that path for
//     A/B runs; db (N floats) still meets through atomics.
// LDS image: a stage is 32 contraction rows, row-major, each row padded by 32 B so that the row stride is an ODD multiple
// of 32 B.  A transposed fragment read (ds_read_b64_tr_b16) is served 32 lanes per LDS cycle = 8 rows x 32 B; with lane
// group g reading rows 16h + 4g + (0..3) those are 8 CONSECUTIVE rows, which the odd stride spreads over all 64 banks:
// conflict-free without any address swizzle (the MFMA contraction index <-> stage row assignment is free as long as both
// operands use the same one), so fragment addresses are one per-lane base plus compile-time immediates and the global
// reads stay whole, unpermuted rows.
#include <stdlib.h>

#include "common.h"

namespace {

struct WgradXArgs {
  const lp_t* dY; int lddy;
  const lp_t* X; int ldx;
  float* dW; long dw_gstride; int lddw;
  float* db; int db_gstride;
  int M, N, K, split;
  float out_scale;
  float* slabs;                   // SLABS kernels: fp32 slabs [2 row groups][Q partitions][N][K] of the partial sums
  simvg_wgrad_reduce_desc* defer; // host pointer: where to leave the description of the second stage instead of launching it
  unsigned long long* prof;       // development builds (-DSIMVG_WG_PROFILE): s_memtime at the role boundaries of stages 8..23
  int fv;                         // virtual stages of the second flush (wg_bound)
};
// Row partition q of Q: stages [wg_bound(q), wg_bound(q + 1)) of the launch's stage list (row group 0 first, st0 = its stages).  The
// partition that holds the boundary between the row groups flushes TWICE (one partial sum per group): `fv` virtual stages stand
// at the boundary for that second flush, so that its share of real stages is shorter by what the flush costs (the launch ends
// with its slowest workgroup: without it the straddling partition's workgroups finish a flush -- 12-17 us -- after all others;
// measured, kernel + second stage: 12-wave kernel qkv 128 -> 114 us, fc1 170 -> 160, out-proj 54.6 -> 51.5 at fv = 14; the 256 x 256
// kernel fc1 158 -> 149, fc2 152 -> 147.5 at fv = 12; profiles/r04_sweeps.md section 6).
__host__ __device__ inline int wg_bound(int q, int Q, int ST, int st0, int fv) {
  if (st0 <= 0 || st0 >= ST) fv = 0;
  if (fv > ST / Q / 2) fv = ST / Q / 2;                     // (every partition keeps at least one real stage)
  const int v = (int)((long)q * (ST + fv) / Q);
  return v <= st0 ? v : (v < st0 + fv ? st0 : v - fv);
}
static int wg_fv(int dflt) {                                 // SIMVG_WG_FV: A/B switch for the virtual flush stages (0: even split)
  const char* e = getenv("SIMVG_WG_FV");
  return e ? atoi(e) : dflt;
}
#ifdef SIMVG_WG_PROFILE
#define WG_T(k_) do { if (a.prof && blockIdx.x == 8 && lane == 0 && (wave & 3) == 0 && t >= 8 && t < 24)                 \
    a.prof[((wave >> 2) * 16 + (t - 8)) * 8 + (k_)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WG_T(k_) do { } while (0)
#endif

// The LDS reads of the main loop are inline asm (common.h: lds_tr16_asm / lds_wait_all): the compiler would otherwise
// drain the ring with `s_waitcnt vmcnt(0)` right after every stage is issued, which is what the first wgrad kernel did.

// A x B = 12 waves (along n x along k), each a (16 WI) x (16 WJ) block of the output tile; 4-deep LDS ring.
//
// Schedule.  Measured with s_memtime on the 8-wave predecessor of this kernel (one wave per SIMD issuing): a
// ds_read_b64_tr_b16 costs its wave ~19 cycles of issue, an LDS-DMA piece ~90 (descriptor SALU included), an MFMA ~21 --
// per stage and SIMD 912 + 900 + 1512 cycles, and with the waves of a SIMD in lockstep (or ping-ponging behind two
// barriers per stage with a LOAD segment twice as long as the MFMA segment) those simply added up (MFMA pipe 38 % busy).
// Here the THREE waves of a SIMD (w, w + 4, w + 8: waves are dealt to the SIMDs round-robin) rotate through three
// roles, one barrier-delimited interval each, a third of a stage apart:
//     LOADa(t): issue the fragment reads of stage t          (~20 x 19 cycles)
//     LOADb(t): wait for them, bias-gradient partial sums, issue the LDS-DMA of stage t + 3
//     MFMA(t) : 24 MFMAs at priority 1
// so in every interval exactly one wave per SIMD owns the MFMA pipe while its two partners prepare.
//   group g runs LOADa(t) / LOADb(t) / MFMA(t) in intervals 3t + g, 3t + g + 1, 3t + g + 2.
//   RAW: stage t + 1 is first read in interval 3t + 3, so every wave waits for ITS pieces of stage t + 1 (counted vmcnt)
//        before the barrier that closes interval 3t + 2 -- the end of MFMA(t) / LOADb(t) / LOADa(t) for g = 0 / 1 / 2;
//   WAR: stage t + 3 lands in the buffer of stage t - 1, last read (group 2, LOADb(t - 1)) in interval 3t; the earliest
//        DMA into it is issued in interval 3t + 1 (group 0's LOADb(t)).
// Round-3 timeline (tools/dev/wgrad_profile.py, -DSIMVG_WG_PROFILE, shader cycles per 32-row stage of the fc1 shape): 2850 in all;
// per wave: fragment-read issue 390, LDS-DMA issue 380-460 (3-4 pieces at ~130 each: the CU's L2 -> LDS path takes 1 KiB per ~17
// cycles and four waves issue at once), rest of LOADb 210-450 (the bias column sums live in group 0), 24 MFMAs 510-600, and
// 920 waiting at the three barriers -- the interval is set by LOADb (600-900), not by the MFMA role (384 ideal).  Moving all or
// part of the DMA issue between the MFMAs (60-100 cycles per piece there) made the MFMA role the long one (680-740) and every shape
// 5-6 % slower (fc1 165 -> 173-175 us); what this structure would need is the DMA issue spread evenly over the three roles
// (a WAR hazard forbids it in group 0's LOADa) or dedicated producer waves (the accumulators of 8 consumer waves do not fit).
template <int A, int B, int WI, int WJ, bool SLABS>
__global__ __launch_bounds__(A * B * 64) void wgrad_x_kernel(WgradXArgs a) {
  constexpr int NW = A * B, NST = 4;
  constexpr int TN = 16 * WI * A, TK = 16 * WJ * B;
  constexpr int SN = TN * 2 + 32, SK = TK * 2 + 32;      // padded row strides (bytes), odd multiples of 32
  constexpr int PN = SN / 32, PK = SK / 32, P = PN + PK;  // 1 KiB pieces per stage (32 rows)
  constexpr int STAGE = 32 * (SN + SK);
  constexpr int MAXP = (P + NW - 1) / NW;                  // waves [0, P - (MAXP-1) NW) issue MAXP pieces, the others MAXP - 1
  static_assert((SN / 32) % 2 == 1 && (SK / 32) % 2 == 1, "row strides must be odd multiples of 32 B");
  static_assert(NW == 12, "three waves per SIMD");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave / B, wk = wave % B;
  const int grp3 = wave >> 2;                              // role phase of this wave: 0, 1, 2
  const int tiles_k = a.K / TK;
  const int ntile = (a.N / TN) * tiles_k;
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int tile = local % ntile, sub = local / ntile, nsub = (int)(gridDim.x >> 3) / ntile;
  const int tn = tile / tiles_k, tk = tile - tn * tiles_k;
  const int n0 = tn * TN, k0 = tk * TK;
  // the launch's rows as a list of 32-row stages, group 0 (rows [0, split)) first; this block takes a contiguous share
  const int st0 = (a.split + 31) >> 5, st1 = (a.M - a.split + 31) >> 5, ST = st0 + st1;
  const int q = xcd * nsub + sub, Q = 8 * nsub;
  const int s_begin = wg_bound(q, Q, ST, st0, a.fv), s_end = wg_bound(q + 1, Q, ST, st0, a.fv);

  // ---- this wave's pieces of a stage: LDS piece p holds bytes [p * 1024, +1024) of the stage image.  The loads are
  // buffer loads (`buffer_load_dwordx4 ... offen lds`): the descriptor is rebuilt per stage with its base at the stage's
  // first row and its size at the end of the row group, so rows beyond the group and the 32 padding bytes of every LDS row
  // (voffset = ~0) read as zero by the hardware's range check -- no per-lane pointer selects, one constant VGPR per piece
  unsigned poff[MAXP];
#pragma unroll
  for (int ii = 0; ii < MAXP; ++ii) {
    const int p = wave + ii * NW;
    poff[ii] = 0xffffffffu;
    const bool x = p >= PN;
    const int o = (x ? p - PN : p) * 1024 + lane * 16;
    const int S = x ? SK : SN;
    const int row = o / S, byte = o - row * S;
    if (p < P && byte < S - 32) poff[ii] = (unsigned)(row * (x ? a.ldx : a.lddy) * 2 + (x ? k0 : n0) * 2 + byte);
  }
  const bool full = wave + (MAXP - 1) * NW < P;           // wave-uniform: this wave issues MAXP pieces per stage

  // ---- per-lane fragment bases (stage-relative): lane (i = lane & 15, g = lane >> 4) reads rows 16h + 4g + (i >> 2)
  const int i16 = lane & 15, g4 = lane >> 4;
  const int fb_n = (4 * g4 + (i16 >> 2)) * SN + (wn * (16 * WI) + 4 * (i16 & 3)) * 2;
  const int fb_k = 32 * SN + (4 * g4 + (i16 >> 2)) * SK + (wk * (16 * WJ) + 4 * (i16 & 3)) * 2;
  // ---- bias gradient: this block sums columns [n0 + tk * CS, +CS) of its dY stage, CS = TN / tiles_k
  const int CS = TN / tiles_k, cpr = CS >> 3;               // 16-B chunks per row
  const bool cs_lane = a.db != nullptr && tid < 32 * cpr;
  const int cs_row = cs_lane ? tid / cpr : 0, cs_cc = cs_lane ? tid - cs_row * cpr : 0;
  const int cs_off = cs_row * SN + (tk * CS + cs_cc * 8) * 2;

  const unsigned lds0 = lds_addr(smem);
  for (int grp = 0; grp < 2; ++grp) {
    const int sb = grp ? max(s_begin, st0) : s_begin, se = grp ? s_end : min(s_end, st0);
    if (sb >= se) continue;
    const int r_first = grp ? a.split + (sb - st0) * 32 : sb * 32;      // first row of this segment
    const int m_end = grp ? a.M : a.split;
    const int nt = se - sb;
    f32x4_t acc[WI][WJ];
#pragma unroll
    for (int i = 0; i < WI; ++i)
#pragma unroll
      for (int j = 0; j < WJ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    auto issue = [&](int t) {
      const int row0 = r_first + t * 32;
      char* st = smem + (t % NST) * STAGE;
      const long left = (long)(m_end - row0);             // rows of this group from the stage's first row on
      const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dY + (long)row0 * a.lddy), 0,
                                                                         (int)(left * a.lddy * 2), 0x00020000);
      const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(a.X + (long)row0 * a.ldx), 0,
                                                                         (int)(left * a.ldx * 2), 0x00020000);
#pragma unroll
      for (int ii = 0; ii < MAXP; ++ii) {
        const int p = wave + ii * NW;
        if (p < P) {
          if (p >= PN) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS_PTR(st + p * 1024), 16, (int)poff[ii], 0, 0, 0);
          else __builtin_amdgcn_raw_ptr_buffer_load_lds(ry, LDS_PTR(st + p * 1024), 16, (int)poff[ii], 0, 0, 0);
        }
      }
    };
    // own pieces of the oldest `stages` + 1 outstanding stages: wait for the oldest one
    auto wait_oldest = [&](int later) {                   // later = stages issued after the awaited one (wave-uniform)
      if (later >= 2) {
        if (full) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * MAXP) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (MAXP - 1)) : "memory");
      } else if (later == 1) {
        if (full) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MAXP) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MAXP - 1) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    };
    issue(0);
    if (nt > 1) issue(1);
    if (nt > 2) issue(2);
    wait_oldest(min(nt - 1, 2));
    __builtin_amdgcn_s_barrier();
    for (int k = 0; k < grp3; ++k) __builtin_amdgcn_s_barrier();        // phase shift of this wave's group
    for (int t = 0; t < nt; ++t) {
      // ---------------- LOADa(t): fragment reads
      WG_T(0);
      const unsigned sb_ = lds0 + (unsigned)(t % NST) * STAGE;
      const unsigned an = sb_ + fb_n, ak = sb_ + fb_k;
      u32x2_t ry[WI][2], rx[WJ][2];
#define RD_Y(i_) if constexpr ((i_) < WI) { ry[i_][0] = lds_tr16_asm<(i_) * 32>(an); ry[i_][1] = lds_tr16_asm<16 * SN + (i_) * 32>(an); }
#define RD_X(j_) if constexpr ((j_) < WJ) { rx[j_][0] = lds_tr16_asm<(j_) * 32>(ak); rx[j_][1] = lds_tr16_asm<16 * SK + (j_) * 32>(ak); }
      RD_Y(0) RD_Y(1) RD_Y(2) RD_Y(3) RD_Y(4) RD_Y(5) RD_Y(6) RD_Y(7) RD_Y(8)
      RD_X(0) RD_X(1) RD_X(2) RD_X(3) RD_X(4) RD_X(5)
#undef RD_Y
#undef RD_X
      u32x4_t csv;
      if (cs_lane) csv = lds_b128_asm<0>(sb_ + cs_off);
      WG_T(1);
      if (grp3 == 2 && t + 1 < nt) wait_oldest(min(nt - 2 - t, 1));     // stage t + 1 (t + 3 not issued yet)
      __builtin_amdgcn_s_barrier();
      WG_T(2);
      // ---------------- LOADb(t): reads complete, bias partial sums
      if (t + 3 < nt) issue(t + 3);
      WG_T(3);
      lds_wait_all();
#pragma unroll
      for (int i = 0; i < WI; ++i) lds_pin(ry[i][0], ry[i][1]);
#pragma unroll
      for (int j = 0; j < WJ; ++j) lds_pin(rx[j][0], rx[j][1]);
      lpx8_t fy[WI], fx[WJ];
#pragma unroll
      for (int i = 0; i < WI; ++i) fy[i] = frag8(ry[i][0], ry[i][1]);
#pragma unroll
      for (int j = 0; j < WJ; ++j) fx[j] = frag8(rx[j][0], rx[j][1]);
      if (cs_lane) {
        asm volatile("" : "+v"(csv));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float lo, hi;
          unpack_lp2(csv[e], lo, hi);
          cs[2 * e] += lo;
          cs[2 * e + 1] += hi;
        }
      }
      WG_T(4);
      if (grp3 == 1 && t + 1 < nt) wait_oldest(min(nt - 2 - t, 2));
      __builtin_amdgcn_s_barrier();
      WG_T(5);
      // ---------------- MFMA(t)
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < WI; ++i)
#pragma unroll
        for (int j = 0; j < WJ; ++j) acc[i][j] = mfma_lp(fy[i], fx[j], acc[i][j]);
      __builtin_amdgcn_s_setprio(0);
      WG_T(6);
      if (grp3 == 0 && t + 1 < nt) wait_oldest(min(nt - 2 - t, 2));
      __builtin_amdgcn_s_barrier();
      WG_T(7);
    }
    for (int k = grp3; k < 2; ++k) __builtin_amdgcn_s_barrier();
    // ---- flush: fp32 atomics into dW[grp] (lanes 0-15 of a 16-lane group cover 64 contiguous bytes of one row)
    // SLABS (round 4): the partial sums go to this (row group, partition)'s own slab with PLAIN stores and a second kernel adds
    // the slabs into dW in a fixed order.  The fp32 atomics of the eight partitions were 47 of the 166 us of the fc1 launch
    // (development build without the flush: 119 us; tools/dev/probes/atomic_flush.hip: 60 us for the atomics of this pattern
    // whatever their coalescing, 12 us for the same bytes as plain stores, 30 us with the reduction launch).
    if constexpr (SLABS) {
      // through the (now free) ring, 32 rows of the wave's block at a time, so that a store instruction covers whole 16-B chunks
      // of rows (24 store instructions per lane instead of 96 dword stores on 64-B segments)
      float* sl = a.slabs + ((long)grp * Q + q) * a.N * a.K;
      constexpr int LD = 16 * WJ + 4;                       // floats per staged row
      float* st = (float*)(smem + wave * (32 * LD * 4));    // wave-private: 12 x 12.8 KiB at most
      static_assert(NW * 32 * LD * 4 <= NST * STAGE, "flush staging must fit the ring");
#pragma unroll
      for (int p = 0; p < (WI + 1) / 2; ++p) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          if (2 * p + ii >= WI) continue;
#pragma unroll
          for (int j = 0; j < WJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) st[(ii * 16 + 4 * g4 + r) * LD + j * 16 + i16] = acc[2 * p + ii][j][r] * a.out_scale;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int nb = n0 + wn * (16 * WI) + p * 32, kb = k0 + wk * (16 * WJ);
#pragma unroll
        for (int it = 0; it < 2 * WJ; ++it) {               // 32 rows x 4 WJ chunks of 16 B = 64 lanes x 2 WJ
          const int e = it * 64 + lane, row = e / (4 * WJ), c = (e - row * (4 * WJ)) * 4;
          if (2 * p + (row >> 4) < WI)
            *(f32x4_t*)(sl + (long)(nb + row) * a.K + kb + c) = *(const f32x4_t*)(st + row * LD + c);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the next pass overwrites the slice
      }
      __syncthreads();                                      // before the next segment's loads land in the ring
    } else {
    float* dW = a.dW + (long)grp * a.dw_gstride;
#pragma unroll
    for (int i = 0; i < WI; ++i)
#pragma unroll
      for (int j = 0; j < WJ; ++j) {
        const int k = k0 + wk * (16 * WJ) + j * 16 + i16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = n0 + wn * (16 * WI) + i * 16 + 4 * g4 + r;
          atomicAdd(dW + (long)n * a.lddw + k, acc[i][j][r] * a.out_scale);
        }
      }
    }
    if (a.db) {
      __syncthreads();                                    // the ring is free: reduce the 32 rows' partial sums in LDS
      float* red = (float*)smem;
      for (int c = tid; c < CS; c += NW * 64) red[c] = 0.f;
      __syncthreads();
      if (cs_lane) {
#pragma unroll
        for (int e = 0; e < 8; ++e) atomicAdd(&red[cs_cc * 8 + e], cs[e]);
      }
      __syncthreads();
      for (int c = tid; c < CS; c += NW * 64)
        atomicAdd(a.db + (long)grp * a.db_gstride + n0 + tk * CS + c, red[c] * a.out_scale);
      __syncthreads();                                    // before the next segment's loads land in the ring
    }
  }
}

// dW[g][n][k] += sum of the slabs of the partitions that own rows of group g (blockIdx.y = g, blockIdx.z = which weight of a
// batched launch; fixed order: bit-reproducible, unlike the atomics it replaces)
struct WgradReduceTable { simvg_wgrad_reduce_desc d[SIMVG_WGRAD_REDUCE_MAX]; };
__global__ __launch_bounds__(256) void wgrad_slab_reduce_kernel(WgradReduceTable t) {
  const simvg_wgrad_reduce_desc& d = t.d[blockIdx.z];
  const long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (e >= (long)d.N * d.K) return;
  const int g = blockIdx.y;
  const int lo = g ? d.lo1 : d.lo0, hi = g ? d.hi1 : d.hi0;
  if (lo >= hi && d.assign < 2) return;                     // (assign == 2: dW holds both groups, one without rows gets zeros)
  const int n = (int)(e / d.K), k = (int)(e - (long)n * d.K);
  float* dst = d.dW + (long)g * d.dw_group_stride + (long)n * d.lddw + k;
  f32x4_t s = d.assign ? (f32x4_t){0.f, 0.f, 0.f, 0.f} : *(const f32x4_t*)dst;
  for (int x = lo; x < hi; ++x) s += *(const f32x4_t*)(d.slabs + ((long)g * d.Q + x) * d.N * d.K + e);
  *(f32x4_t*)dst = s;
}


// ------------------------------------------------------------------------------------------
// Round 4, second form: 256 x 256 output tiles, EIGHT waves (2 x 4, each 128 x 64 = 128 accumulator registers, two waves per
// SIMD), 32-row stages in a 4-deep ring, the two waves of a SIMD in PING-PONG phases (one barrier per phase): while one runs its
// 32 MFMAs back-to-back the other issues its DMA and reads its fragments (the loop's comment has the schedule and its hazards).
//   * work items = (row partition q, output tile), Q = 256 / tiles partitions: item x runs on XCD x / 32 (blockIdx % 8 -> XCD is
//     a speed assumption only), so the workgroups that stream the same rows share an L2 and no XCD touches more than two
//     partitions' rows; one workgroup per CU, one round;
//   * same LDS image as above (rows padded to an odd multiple of 32 B: transposed reads conflict-free without a swizzle),
//     same buffer-descriptor zero fill of the rows past a row group's end;
//   * any N, K that are multiples of 256 with K >= 512 (the ViT-L shapes 4096 / 3072 / 1024 included, which the 12-wave tiles do
//     not divide);
//   * bias gradient from the staged dY: the workgroups of one tile row split its 32 16-B column chunks among them;
//   * flush into per-partition slabs through the free ring (wave-private 32 x 68 floats), second stage as above.
// What was measured on the way (kernel + second stage, fc1 shape, us; the 12-wave kernel: 170-174; profiles/r04_sweeps.md section 6):
//   16 waves of 64 x 64, 64-row double buffer, one barrier per stage, everything in phase               177
//   8 waves, every wave software-pipelined on its own (reads of stage t + 1 between the MFMA groups of t)   170
//     ... + L2 prefetch of the stage 4 / 8 / 16 ahead (the loop is not HBM-latency-bound)                   170-173
//     ... + the DMA pieces issued one at a time between the MFMA groups                                      166
//     ablations of that form: MFMAs -50, DMA issue -30, fragment reads -20, bias -7: every component's time ADDS -- nothing
//     a wave issues between its own MFMAs is hidden, and two waves of a SIMD in the same phase do not hide each other's
//   ping-pong phases, all 24 reads + DMA in LOAD                                                             161.5
//   ping-pong, all 24 reads between the MFMA groups, LOAD = DMA only                                         166
//   ping-pong, dY reads in LOAD, X reads behind the MFMAs, bias sums in the MFMAs' shadow (this file)        159
// ------------------------------------------------------------------------------------------
constexpr int SQ_ROWS = 32, SQ_S = 256 * 2 + 32;            // stage rows; padded row stride (bytes) = 17 x 32
constexpr int SQ_PART = SQ_ROWS * SQ_S;                     // one operand's part of a stage: 17 KiB = 17 pieces
constexpr int SQ_STAGE = 2 * SQ_PART, SQ_NST = 4, SQ_LDS = SQ_NST * SQ_STAGE;  // 136 KiB (+ 8 KiB dump area of the out-of-range slots)
constexpr int SQ_NPY = SQ_PART / 1024;
static_assert((SQ_S / 32) % 2 == 1 && SQ_PART % 1024 == 0 && SQ_NPY == 17, "sq stage geometry");

struct WgradSqArgs {
  const lp_t* dY; int lddy;
  const lp_t* X; int ldx;
  float* dW; long dw_gstride; int lddw;
  float* db; int db_gstride;
  int M, N, K, split;
  float out_scale;
  float* slabs;
  int Q, tiles_k, ntile, fv;
  int dbg;                        // development ablations (SIMVG_WG_DBG): 1 no MFMAs, 2 no DMA after the prologue, 4 no fragment reads, 8 no flush, 16 no bias sums
};

template <int N_> __device__ __forceinline__ void sq_lgkm() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N_) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}

template <int DBG>
__global__ __launch_bounds__(512) void wgrad_sq_kernel(WgradSqArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 2, wk = wave & 3;
  const int item = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
  if (item >= a.Q * a.ntile) return;
  const int q = item / a.ntile, tile = item - q * a.ntile;
  const int tn = tile / a.tiles_k, tk = tile - tn * a.tiles_k;
  const int n0 = tn * 256, k0 = tk * 256;
  const int st0 = (a.split + SQ_ROWS - 1) / SQ_ROWS, st1 = (a.M - a.split + SQ_ROWS - 1) / SQ_ROWS, ST = st0 + st1;
  const int s_begin = wg_bound(q, a.Q, ST, st0, a.fv), s_end = wg_bound(q + 1, a.Q, ST, st0, a.fv);

  // this wave's pieces of a stage (piece p = bytes [1024 p, +1024) of the stage image; pieces < 17: dY rows, the others: X rows):
  // two of dY, two of X; waves 0 / 1 also take the dY / X piece left over, the other waves issue that slot out of range into a
  // dump area behind the ring -- every wave issues the same five operations per stage: no branches in the loop, one wait count
  constexpr int SQ_DUMP = SQ_LDS;                           // 8 x 1 KiB (left-over slot)
  const int pidx[4] = {2 * wave, 2 * wave + 1, SQ_NPY + 2 * wave, SQ_NPY + 2 * wave + 1};
  const bool extra = wave < 2, extra_x = wave == 1;
  const int xdst = extra ? (wave == 0 ? 16 : 2 * SQ_NPY - 1) * 1024 : -1;       // stage-relative; -1: the dump area
  unsigned poff[5];
#pragma unroll
  for (int ii = 0; ii < 5; ++ii) {
    const int p = ii < 4 ? pidx[ii] : (wave == 0 ? 16 : 2 * SQ_NPY - 1);
    const bool x = p >= SQ_NPY;
    const int o = (x ? p - SQ_NPY : p) * 1024 + lane * 16;
    const int row = o / SQ_S, byte = o - row * SQ_S;
    poff[ii] = byte < 512 && (ii < 4 || extra) ? (unsigned)(row * (x ? a.ldx : a.lddy) * 2 + (x ? k0 : n0) * 2 + byte) : 0xffffffffu;
  }
  // per-lane fragment bases: lane (i = lane & 15, g = lane >> 4) reads rows 16 h + 4 g + (i >> 2) of the stage
  const int i16 = lane & 15, g4 = lane >> 4;
  const unsigned lds0 = lds_addr(smem);
  const unsigned fb_n = lds0 + (4 * g4 + (i16 >> 2)) * SQ_S + (wn * 128 + 4 * (i16 & 3)) * 2;
  const unsigned fb_k = lds0 + SQ_PART + (4 * g4 + (i16 >> 2)) * SQ_S + (wk * 64 + 4 * (i16 & 3)) * 2;
  // bias gradient: this workgroup sums the 16-B column chunks [c_lo, c_hi) of its dY tile; thread -> (row, chunk) of every stage
  // (tiles_k >= 2: at most 16 chunks x 32 rows = 512 threads); the others read the same way and drop the value (uniform counts)
  const int c_lo = tk * 32 / a.tiles_k, c_hi = (tk + 1) * 32 / a.tiles_k, nch = c_hi - c_lo;
  const bool cs_on = a.db != nullptr && tid < 32 * nch;
  const int cs_c = cs_on ? tid % nch : 0, cs_r0 = cs_on ? tid / nch : 0;
  const unsigned cs_addr = lds0 + cs_r0 * SQ_S + (c_lo + cs_c) * 16;

  char* x_dump = smem + SQ_DUMP + wave * 1024;

  for (int grp = 0; grp < 2; ++grp) {
    const int sb = grp ? max(s_begin, st0) : s_begin, se = grp ? s_end : min(s_end, st0);
    if (sb >= se) continue;
    const int r_first = grp ? a.split + (sb - st0) * SQ_ROWS : sb * SQ_ROWS;
    const int m_end = grp ? a.M : a.split;
    const int nt = se - sb;
    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // DMA of a stage: the descriptors (SGPRs), then this wave's six operations (4 pieces, the left-over slot, the prefetch slot).
    // In the loop they go out ONE AT A TIME between the MFMA groups: the CU's address unit takes a 1-KiB piece per ~17 cycles, and
    // with all 34 pieces of a stage issued behind the barrier every wave sat in that queue (~550 cycles per stage) while the
    // matrix pipe idled (ablations: the DMA issue and the MFMAs each added their full time to the loop).  Past the segment's
    // last stage the descriptors are empty: the operations are still issued (one wait count), as zero fill of a dead buffer.
    auto desc_y = [&](int t, bool real) {                 // !real: a zero-sized descriptor (every lane out of range: zeros into a dead buffer)
      const int row0 = r_first + t * SQ_ROWS;
      return __builtin_amdgcn_make_buffer_rsrc((void*)(a.dY + (long)row0 * a.lddy), 0, real ? (int)((long)(m_end - row0) * a.lddy * 2) : 0, 0x00020000);
    };
    auto desc_x = [&](int t, bool real) {
      const int row0 = r_first + t * SQ_ROWS;
      return __builtin_amdgcn_make_buffer_rsrc((void*)(a.X + (long)row0 * a.ldx), 0, real ? (int)((long)(m_end - row0) * a.ldx * 2) : 0, 0x00020000);
    };
    // this wave's five DMA operations of stage t (past the segment's last stage: empty descriptors, the operations are still
    // issued -- one wait count -- as zero fill of a dead buffer)
    auto issue = [&](int t, bool real) {
      const __amdgpu_buffer_rsrc_t dy = desc_y(t, real), dx = desc_x(t, real);
      char* st = smem + (t & (SQ_NST - 1)) * SQ_STAGE;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(dy, LDS_PTR(st + pidx[0] * 1024), 16, (int)poff[0], 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(dy, LDS_PTR(st + pidx[1] * 1024), 16, (int)poff[1], 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(dx, LDS_PTR(st + pidx[2] * 1024), 16, (int)poff[2], 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(dx, LDS_PTR(st + pidx[3] * 1024), 16, (int)poff[3], 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(extra_x ? dx : dy, LDS_PTR(xdst >= 0 ? st + xdst : x_dump), 16, (int)poff[4], 0, 0, 0);
    };
    u32x2_t rx[4][2], nx[4][2], ry[8][2];
    u32x4_t csv;
    // Phases (one barrier each; `ph` = this wave's group, 0: waves 0-3, 1: waves 4-7 -- one wave of each group per SIMD).  Group 0
    // runs LOAD(s) in phase 2 s - 1 and MFMA(s) in phase 2 s, group 1 LOAD(s) in phase 2 s and MFMA(s) in phase 2 s + 1: in every
    // phase one wave of a SIMD owns the matrix pipe while its partner issues its DMA (MI355X_MICROARCH.md, "Two waves per SIMD":
    // an in-order wave cannot slip MFMAs into the gaps of its own loads, and co-issued MFMAs of the partner come straight out of
    // its stream -- a single-phase software-pipelined form of this loop measured every component's time ADDED: fc1 168 us =
    // 50 MFMA + 30 DMA issue + 20 reads + 68 rest).
    //   MFMA(s): 32 back-to-back MFMAs at priority 1 (the bias sums of stage s in their shadow), then the 8 transposed reads of
    //            stage s + 1's X fragments (into nx), which return under the last MFMAs' execution and the barrier;
    //   LOAD(s): this wave's five DMA operations of stage s + 3, the 16 reads of stage s's dY fragments, its bias chunk, nx -> rx,
    //            lgkmcnt(0) -- MFMA(s) waits for nothing.  (All 24 reads in LOAD: fc1 161.5 us with a ~950-cycle LOAD against
    //            ~550 of MFMA; all of them between the MFMA groups: 166 -- reads in the MFMA stream are not free.)
    //   RAW: stage s + 1 is first read in group 0's MFMA(s): every wave waits for its own pieces of it before the barrier in front
    //        of that phase -- group 0 at the end of LOAD(s) (stages s + 2, s + 3 issued since: vmcnt(10)), group 1 at the end of
    //        MFMA(s - 1) (stage s + 2 issued since: vmcnt(5));
    //   WAR: stage s + 3 is issued in LOAD(s) into the buffer of stage s - 1, last read in LOAD(s - 1) (bias) / MFMA(s - 2).
    const int ph = wave >> 2;
#define SQ_RXA(j_) nx[j_][0] = lds_tr16_asm<(j_) * 32>(ak);
#define SQ_RXB(j_) nx[j_][1] = lds_tr16_asm<16 * SQ_S + (j_) * 32>(ak);
#define SQ_RYA(i_) ry[i_][0] = lds_tr16_asm<(i_) * 32>(an);
#define SQ_RYB(i_) ry[i_][1] = lds_tr16_asm<16 * SQ_S + (i_) * 32>(an);
#define SQ_MMA(i_)                                                                                            \
  if constexpr (!(DBG & 1)) {                                                                                 \
  _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                \
      acc[i_][j] = mfma_lp(frag8(ry[i_][0], ry[i_][1]), frag8(rx[j][0], rx[j][1]), acc[i_][j]);              \
  }                                                                                                           \
  __builtin_amdgcn_sched_barrier(0);
#define SQ_RD(...) if constexpr (!(DBG & 4)) { __VA_ARGS__ } __builtin_amdgcn_sched_barrier(0);
    issue(0, true);
    issue(1, nt > 1);
    issue(2, nt > 2);
    asm volatile("s_waitcnt vmcnt(5)" ::: "memory");      // stages 0 and 1
    __builtin_amdgcn_s_barrier();
    {                                                     // the fragments of stage 0
      const unsigned an = fb_n, ak = fb_k;
      SQ_RXA(0) SQ_RXB(0) SQ_RXA(1) SQ_RXB(1) SQ_RXA(2) SQ_RXB(2) SQ_RXA(3) SQ_RXB(3)
      SQ_RYA(0) SQ_RYB(0) SQ_RYA(1) SQ_RYB(1) SQ_RYA(2) SQ_RYB(2) SQ_RYA(3) SQ_RYB(3)
      SQ_RYA(4) SQ_RYB(4) SQ_RYA(5) SQ_RYB(5) SQ_RYA(6) SQ_RYB(6) SQ_RYA(7) SQ_RYB(7)
    }
    if (ph) __builtin_amdgcn_s_barrier();                 // group 1 starts one phase later
    for (int t = 0; t < nt; ++t) {
      // ---------------- LOAD(t)
      issue(t + 3, t + 3 < nt && !(DBG & 2));
      {
        const unsigned so = (unsigned)(t & (SQ_NST - 1)) * SQ_STAGE;
        const unsigned an = fb_n + so;
        csv = lds_b128_asm<0>(cs_addr + so);
        if (t > 0) { SQ_RD(SQ_RYA(0) SQ_RYB(0) SQ_RYA(1) SQ_RYB(1) SQ_RYA(2) SQ_RYB(2) SQ_RYA(3) SQ_RYB(3)
                           SQ_RYA(4) SQ_RYB(4) SQ_RYA(5) SQ_RYB(5) SQ_RYA(6) SQ_RYB(6) SQ_RYA(7) SQ_RYB(7)) }
      }
      sq_lgkm<0>();
#pragma unroll
      for (int j = 0; j < 4; ++j) { lds_pin(nx[j][0], nx[j][1]); rx[j][0] = nx[j][0]; rx[j][1] = nx[j][1]; }
#pragma unroll
      for (int i = 0; i < 8; ++i) lds_pin(ry[i][0], ry[i][1]);
      asm volatile("" : "+v"(csv));
      if (!ph) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");    // group 0: its pieces of stage t + 1
      __builtin_amdgcn_s_barrier();
      // ---------------- MFMA(t)
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      SQ_MMA(0) SQ_MMA(1) SQ_MMA(2) SQ_MMA(3)
      if (cs_on && !(DBG & 16)) {                         // bias sums of stage t, in the matrix pipe's shadow
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float lo, hi;
          unpack_lp2(csv[e], lo, hi);
          cs[2 * e] += lo;
          cs[2 * e + 1] += hi;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      SQ_MMA(4) SQ_MMA(5) SQ_MMA(6) SQ_MMA(7)
      __builtin_amdgcn_s_setprio(0);
      {                                                   // the X fragments of stage t + 1, under the last MFMAs' execution
        const unsigned ak = fb_k + (unsigned)((t + 1) & (SQ_NST - 1)) * SQ_STAGE;   // (past the last stage: zero fill or stale, unused)
        SQ_RD(SQ_RXA(0) SQ_RXB(0) SQ_RXA(1) SQ_RXB(1) SQ_RXA(2) SQ_RXB(2) SQ_RXA(3) SQ_RXB(3))
      }
      if (ph) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");      // group 1: its pieces of stage t + 2 (read from group 0's MFMA(t + 1) on)
      __builtin_amdgcn_s_barrier();
    }
    if (!ph) __builtin_amdgcn_s_barrier();                // group 0 meets group 1's last phase
#undef SQ_RXA
#undef SQ_RXB
#undef SQ_RYA
#undef SQ_RYB
#undef SQ_MMA
#undef SQ_RD
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (the trailing zero fills too: the flush re-uses the ring)
    __syncthreads();                                      // every wave has left the ring
    if constexpr (DBG & 8) {
    } else if (a.slabs) {
      float* sl = a.slabs + ((long)grp * a.Q + q) * a.N * a.K;
      constexpr int LD = 68;                              // floats per staged row (64 + 4: the 4 row groups of a store land in different banks)
      float* st = (float*)(smem + wave * (32 * LD * 4));  // wave-private, 8.5 KiB each
      static_assert(8 * 32 * LD * 4 <= SQ_LDS, "flush staging must fit the ring");
#pragma unroll
      for (int p = 0; p < 4; ++p) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) st[(ii * 16 + 4 * g4 + r) * LD + j * 16 + i16] = acc[2 * p + ii][j][r] * a.out_scale;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int nb = n0 + wn * 128 + p * 32, kb = k0 + wk * 64;
#pragma unroll
        for (int it = 0; it < 8; ++it) {                  // 32 rows x 16 chunks of 16 B = 64 lanes x 8
          const int e = it * 64 + lane, row = e >> 4, c = (e & 15) * 4;
          *(f32x4_t*)(sl + (long)(nb + row) * a.K + kb + c) = *(const f32x4_t*)(st + row * LD + c);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    } else {
      float* dW = a.dW + (long)grp * a.dw_gstride;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = k0 + wk * 64 + j * 16 + i16;
#pragma unroll
          for (int r = 0; r < 4; ++r) atomicAdd(dW + (long)(n0 + wn * 128 + i * 16 + 4 * g4 + r) * a.lddw + k, acc[i][j][r] * a.out_scale);
        }
    }
    if (a.db) {
      __syncthreads();
      float* red = (float*)smem;
      if (tid < 256) red[tid] = 0.f;
      __syncthreads();
      if (cs_on) {
#pragma unroll
        for (int e = 0; e < 8; ++e) atomicAdd(&red[cs_c * 8 + e], cs[e]);
      }
      __syncthreads();
      if (tid < nch * 8) atomicAdd(a.db + (long)grp * a.db_gstride + n0 + c_lo * 8 + tid, red[tid] * a.out_scale);
    }
    __syncthreads();                                      // before the next segment's loads land in the ring
  }
}

// which kernel takes (M, N, K): 0 = none of this file's, 12 = the 12-wave XCD kernel, 16 = the 256 x 256 kernel;
// Q = row partitions (= slabs per row group), rows = rows per stage of the partition arithmetic
struct WgradPlan { int kind, Q, rows; };
static WgradPlan wgrad_plan(int M, int N, int K) {
  WgradPlan p{0, 0, 32};
  if (M < 4096) return p;
  const char* e = getenv("SIMVG_WGRAD_SQ");                 // read per call (tools/dev/wgrad_ab.py switches it between launches)
  const int sq = e ? atoi(e) : -1;                          // 0: never, 1: wherever it fits, default: by shape
  auto fits = [&](int tn, int tk) { return N % tn == 0 && K % tk == 0 && (N / tn) * (K / tk) == 32 && ((tn / (K / tk)) % 8) == 0; };
  const bool x8 = fits(384, 192) || fits(288, 192) || fits(192, 384);
  const bool x16 = N % 192 == 0 && K % 192 == 0 && (N / 192) * (K / 192) == 16 && ((192 / (K / 192)) % 8) == 0;
  const int tiles = (N % 256 == 0 && K % 256 == 0) ? (N / 256) * (K / 256) : 0;
  const bool sq_ok = tiles > 0 && tiles <= 128 && K / 256 >= 2 && K / 256 <= 32;   // (K = 256: 32 bias chunks x 32 rows > 512 threads)
  // default: the shapes the 12-wave tiles do not divide (ViT-L's) and, of those they do, fc1 / fc2 (36 tiles x 7 partitions:
  // 159 / 155 us against 172 / 165 with the second stage; qkv -- 27 tiles x 9 -- measures equal, out-proj slower: 12-wave)
  const bool want_sq = sq_ok && (sq == 1 || (sq != 0 && ((!x8 && !x16) || tiles >= 36)));
  if (want_sq) {                                            // every partition must own at least one 64-row stage
    p.kind = 16; p.Q = 256 / tiles < M / SQ_ROWS ? 256 / tiles : M / SQ_ROWS; p.rows = SQ_ROWS;
    return p;
  }
  if (x8) { p.kind = 12; p.Q = 8; }
  else if (x16) { p.kind = 12; p.Q = 16; }
  return p;
}

static bool launch_sq(const WgradXArgs& x, const WgradPlan& pl, hipStream_t stream) {
  const int tiles_k = x.K / 256, ntile = (x.N / 256) * tiles_k, Q = pl.Q;
  float* slabs = x.slabs;
  if (x.K % 4 != 0 || x.lddw % 4 != 0 || (x.dw_gstride & 3)) slabs = nullptr;
  WgradSqArgs a{x.dY, x.lddy, x.X, x.ldx, x.dW, x.dw_gstride, x.lddw, x.db, x.db_gstride, x.M, x.N, x.K, x.split, x.out_scale,
                slabs, Q, tiles_k, ntile, wg_fv(12), getenv("SIMVG_WG_DBG") ? atoi(getenv("SIMVG_WG_DBG")) : 0};
#define SQ_LAUNCH(D_) case D_: { static bool once = hipFuncSetAttribute((const void*)wgrad_sq_kernel<D_>, hipFuncAttributeMaxDynamicSharedMemorySize, SQ_LDS + 8192) == hipSuccess; (void)once; \
    hipLaunchKernelGGL(wgrad_sq_kernel<D_>, dim3(256), dim3(512), SQ_LDS + 8192, stream, a); break; }
  switch (a.dbg) {
#ifdef SIMVG_WG_ABLATE
    SQ_LAUNCH(1) SQ_LAUNCH(2) SQ_LAUNCH(4) SQ_LAUNCH(8) SQ_LAUNCH(16) SQ_LAUNCH(7) SQ_LAUNCH(15) SQ_LAUNCH(31) SQ_LAUNCH(6) SQ_LAUNCH(9)
#endif
    default: SQ_LAUNCH(0)
  }
#undef SQ_LAUNCH
  if (slabs) {
    const int st0 = (x.split + SQ_ROWS - 1) / SQ_ROWS, st1 = (x.M - x.split + SQ_ROWS - 1) / SQ_ROWS, ST = st0 + st1;
    int lo0 = Q, hi0 = 0, lo1 = Q, hi1 = 0;
    for (int q = 0; q < Q; ++q) {
      const int sb = wg_bound(q, Q, ST, st0, a.fv), se = wg_bound(q + 1, Q, ST, st0, a.fv);
      if (sb < (se < st0 ? se : st0)) { lo0 = q < lo0 ? q : lo0; hi0 = q + 1; }
      if ((sb > st0 ? sb : st0) < se) { lo1 = q < lo1 ? q : lo1; hi1 = q + 1; }
    }
    const simvg_wgrad_reduce_desc d{slabs, x.dW, x.dw_gstride, x.lddw, x.N, x.K, Q, lo0, hi0, lo1, hi1, 0};
    if (x.defer) {
      *x.defer = d;
    } else {
      WgradReduceTable t;
      t.d[0] = d;
      hipLaunchKernelGGL(wgrad_slab_reduce_kernel, dim3((x.N * x.K / 4 + 255) / 256, 2, 1), dim3(256), 0, stream, t);
    }
  }
  return true;
}

template <int A, int B, int WI, int WJ>
bool launch(const WgradXArgs& a0, hipStream_t stream, int nsub = 1) {
  constexpr int TN = 16 * WI * A, TK = 16 * WJ * B;
  constexpr int LDS = 4 * 32 * (TN * 2 + 32 + TK * 2 + 32);
  static bool once = hipFuncSetAttribute((const void*)wgrad_x_kernel<A, B, WI, WJ, false>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS) == hipSuccess &&
                     hipFuncSetAttribute((const void*)wgrad_x_kernel<A, B, WI, WJ, true>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS) == hipSuccess;
  (void)once;
  WgradXArgs a = a0;
  // which partitions own rows of which group (the kernel's row partition: wg_bound)
  const int st0 = (a.split + 31) >> 5, st1 = (a.M - a.split + 31) >> 5, ST = st0 + st1, Q = 8 * nsub;
  int lo0 = Q, hi0 = 0, lo1 = Q, hi1 = 0;
  for (int q = 0; q < Q; ++q) {
    const int sb = wg_bound(q, Q, ST, st0, a.fv), se = wg_bound(q + 1, Q, ST, st0, a.fv);
    if (sb < (se < st0 ? se : st0)) { lo0 = q < lo0 ? q : lo0; hi0 = q + 1; }
    if ((sb > st0 ? sb : st0) < se) { lo1 = q < lo1 ? q : lo1; hi1 = q + 1; }
  }
  if (a.K % 4 != 0 || a.lddw % 4 != 0 || (a.dw_gstride & 3)) a.slabs = nullptr;
  const int ntile = (a.N / TN) * (a.K / TK);
  if (a.slabs) {
    hipLaunchKernelGGL((wgrad_x_kernel<A, B, WI, WJ, true>), dim3(8 * ntile * nsub), dim3(A * B * 64), LDS, stream, a);
    const simvg_wgrad_reduce_desc d{a.slabs, a.dW, a.dw_gstride, a.lddw, a.N, a.K, Q, lo0, hi0, lo1, hi1, 0};
    if (a.defer) {
      *a.defer = d;                   // the caller batches the second stage (simvg_wgrad_reduce_batched), possibly on another stream
    } else {
      WgradReduceTable t;
      t.d[0] = d;
      hipLaunchKernelGGL(wgrad_slab_reduce_kernel, dim3((a.N * a.K / 4 + 255) / 256, 2, 1), dim3(256), 0, stream, t);
    }
  } else {
    hipLaunchKernelGGL((wgrad_x_kernel<A, B, WI, WJ, false>), dim3(8 * ntile * nsub), dim3(A * B * 64), LDS, stream, a);
  }
  return true;
}

}  // namespace

// -> true when the problem was launched here (ViT-B encoder shapes at training sizes); false: the caller falls back to the
// generic kernels of gemm.hip
// floats of slab workspace the XCD-partitioned kernel wants for this problem (0: the problem is not its)
long simvg_wgrad_x_slab_floats(int M, int N, int K) {
  if (const char* e = getenv("SIMVG_WG_SLABS")) { if (atoi(e) == 0) return 0; }     // the A/B switch: no slabs, no second stage
  const WgradPlan p = wgrad_plan(M, N, K);
  return p.kind ? 2L * p.Q * N * K : 0;
}

int simvg_wgrad_reduce_launch(const simvg_wgrad_reduce_desc* descs, int n, hipStream_t stream) {
  WgradReduceTable t;
  long emax = 0;
  int m = 0;
  for (int i = 0; i < n; ++i) {
    if (!descs[i].slabs) continue;
    t.d[m++] = descs[i];
    const long e = (long)descs[i].N * descs[i].K;
    emax = e > emax ? e : emax;
  }
  if (m) hipLaunchKernelGGL(wgrad_slab_reduce_kernel, dim3((unsigned)((emax / 4 + 255) / 256), 2, m), dim3(256), 0, stream, t);
  return m;
}

bool simvg_wgrad_x(const void* dY, int lddy, const void* X, int ldx, float* dW, long dw_gstride, int lddw, float* db,
                   int db_gstride, int M, int N, int K, int split, float out_scale, float* slabs,
                   simvg_wgrad_reduce_desc* defer, hipStream_t stream) {
  if (defer) defer->slabs = nullptr;
  if (M < 4096 || (lddy & 7) || (ldx & 7)) return false;
  if (const char* e = getenv("SIMVG_WG_SLABS")) { if (atoi(e) == 0) slabs = nullptr; }      // A/B switch (tools/dev/wgrad_ab.py)
  static unsigned long long* prof = getenv("SIMVG_WG_PROF_PTR") ? (unsigned long long*)strtoull(getenv("SIMVG_WG_PROF_PTR"), nullptr, 0) : nullptr;
  WgradXArgs a{(const lp_t*)dY, lddy, (const lp_t*)X, ldx, dW, dw_gstride, lddw, db, db_gstride, M, N, K, split, out_scale, slabs, defer, prof, wg_fv(14)};
  const WgradPlan pl = wgrad_plan(M, N, K);
  if (pl.kind == 16) return launch_sq(a, pl, stream);
  if (pl.kind != 12) return false;
  auto fits = [&](int tn, int tk) {
    return N % tn == 0 && K % tk == 0 && (N / tn) * (K / tk) == 32 && ((tn / (K / tk)) % 8) == 0;
  };
  if (fits(384, 192)) return launch<4, 3, 6, 4>(a, stream);      // fc1: 3072 x 768, waves 4 x 3 of 96 x 64
  if (fits(288, 192)) return launch<3, 4, 6, 3>(a, stream);      // qkv: 2304 x 768, waves 3 x 4 of 96 x 48
  if (fits(192, 384)) return launch<3, 4, 4, 6>(a, stream);      // fc2 / patch embed: 768 x 3072, waves 3 x 4 of 64 x 96
  // out-proj: 768 x 768 = 16 tiles of 192 x 192 (waves 3 x 4 of 64 x 48); every XCD's row range is halved once more so that
  // 32 workgroups per XCD exist (16 row partitions)
  if (pl.Q == 16) return launch<3, 4, 4, 3>(a, stream, 2);
  return false;
}
