// bf16 MFMA GEMMs for the SimVG hot path (gfx950).
//
//  simvg_gemm_nt : C[M,N] = A[M,K] · W[g][N,K]^T (+bias) (+GELU/ReLU) (+residual·row_scale)
//                  "multiway" = two row groups (vision rows | text rows) with their own W/bias,
//                  replacing torchscale MultiwayNetwork's split -> A(x1), B(x2) -> cat
//                  (reference call sites beit3_base.py:137-145,159; SURVEY.md §2.3 E6,E14,E16,E19).
//                  Used for forward and (with the transposed weight copy) for dgrad.
//  simvg_gemm_nt_split : the same with the weight carried as hi + lo 16-bit halves (precise inference forward)
//  simvg_gemm_tn : dW[g][N,K] += dY[M,N]^T · X[M,K]   (wgrad; the encoder shapes go to wgrad.hip, the rest is split over M
//                  here and meets through fp32 atomics)
//
// Kernels by shape (dispatch in gemm_nt_launch; `simvg_gemm_nt_plan` returns the choice without launching): 16-wave 256x256x64
// tiles, persistent with a 2-stage ring (N >= 2304, big M; `_p2`: hi + lo weights); ONE round of 320x256x64 tiles on the same
// hand-managed loop (round 6: N = 768 at M = 26 944 -- `gemm_nt_kernel_tall5_*`; 256 / 224 rows for ViT-L's row count:
// `_tall4_f32`, `_t224_*`); 16-wave 160x256x64 with a 3-stage ring and 224x256 / 256x256 one-tile kernels with compiler-scheduled
// reads (other row counts, epilogues with an activation); 128x128x64 / 256x128x32 / latency variants for the small shapes.  All of
// them: v_mfma_f32_16x16x32, HBM->LDS by LDS-DMA (16 B/lane, LDS image lane-linear, XOR swizzle applied on the SOURCE address and
// on the ds_read address), counted vmcnt, coalesced epilogues.
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile

// ids of simvg_gemm_nt_plan (mirrored in include/simvg_hip.h)
enum { SIMVG_GEMM_PLAN_LAT = 1, SIMVG_GEMM_PLAN_TALL5 = 2, SIMVG_GEMM_PLAN_T224 = 3, SIMVG_GEMM_PLAN_TALL4 = 4, SIMVG_GEMM_PLAN_224 = 5,
       SIMVG_GEMM_PLAN_PERSIST = 6, SIMVG_GEMM_PLAN_PERSIST_SPLIT = 7, SIMVG_GEMM_PLAN_256 = 8, SIMVG_GEMM_PLAN_160 = 9,
       SIMVG_GEMM_PLAN_256K32 = 10, SIMVG_GEMM_PLAN_128 = 11 };

struct GemmNTArgs {
  const lp_t* A; int lda;
  const lp_t* W; long w_gstride; int ldw;
  const float* bias; int bias_gstride;
  void* C; int ldc; int c_f32;
  lp_t* aux; int ldaux;
  const float* res; int ldres;
  const float* row_scale; int rps0, rps1;
  int M, N, K, split, act;
  int gn;     // column-group width of the tile walk (0: rows of all column tiles)
  float alpha; // scalar on the accumulator, before the bias (1/gradient-scale in the head's backward GEMMs)
  // SPLIT weights (simvg_gemm_nt_split, the precise inference forward): W rows are [lo * 2^s | hi] along K (K = 2 ka), A has
  // ka columns and is walked twice; after the lo half the accumulators are multiplied by lo_scale = 2^-s, so that
  // C = A . (hi + lo)^T with the weight carried to ~22 significand bits at twice the MFMA work.  ka == K: plain GEMM.
  int ka; float lo_scale;
  int respf;   // fp32 + residual epilogue: residual rows requested one pass ahead (0: inside the read-out loop, round 3)
};

// k offset of the A operand for k-tile element offset k (A wraps around after ka columns)
__device__ __forceinline__ int a_koff(const GemmNTArgs& a, int k) { return k >= a.ka ? k - a.ka : k; }
// after the k-tile that ends the lo half: acc *= lo_scale (wave-uniform branch, once per tile)
template <int MI_, int NJ_>
__device__ __forceinline__ void split_rescale(const GemmNTArgs& a, f32x4_t (&acc)[MI_][NJ_], int k_done) {
  if (k_done == a.ka && a.ka < a.K) {
#pragma unroll
    for (int i = 0; i < MI_; ++i)
#pragma unroll
      for (int j = 0; j < NJ_; ++j) acc[i][j] *= a.lo_scale;
  }
}

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  // bijective XCD-aware remap: XCD x (= bid % 8) walks a contiguous chunk of the tile space
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// Tile walk in column groups: an XCD runs 32 tiles at a time, consecutive in this order.  With all column tiles of a row
// block next to each other (N = 3072: 12 of them) those 32 tiles touch 3 activation panels and ALL 12 weight panels
// (5.9 MB against 4 MB of L2 per XCD); walking gn = 6 column tiles over ~5 row blocks touches 6 + 6.
__device__ __forceinline__ void tile_order(int bid, int tiles_m, int tiles_n, int gn, int& tm, int& tn) {
  if (gn <= 0 || gn >= tiles_n) { tm = bid / tiles_n; tn = bid - tm * tiles_n; return; }
  const int per_group = tiles_m * gn;
  const int g = bid / per_group;
  const int n_start = g * gn;
  const int w = min(gn, tiles_n - n_start);
  const int l = bid - g * per_group;
  tm = l / w;
  tn = n_start + (l - tm * w);
}

// stage one [128 rows][64 k] bf16 tile: 16 wave-instructions of 1 KiB, 4 per wave
__device__ __forceinline__ void stage_tile_k64(const lp_t* base, int ld, int row0, int row_last, int k0,
                                               char* lds, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int inst = wave * 4 + i;
    const int r = inst * 8 + (lane >> 3);
    int row = row0 + r;
    row = row < row_last ? row : row_last;
    const int lslot = (lane & 7) ^ (r & 7);
    const lp_t* src = base + (long)row * ld + k0 + lslot * 8;
    __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(lds + inst * 1024), 16, 0, 0);
  }
}

__device__ __forceinline__ lpx8_t read_frag_k64(const char* lds, int row, int lslot) {
  return *(const lpx8_t*)(lds + row * 128 + ((lslot ^ (row & 7)) << 4));
}

// shared epilogue: bias, optional pre-activation copy, GELU/ReLU, DropPath-scaled residual add, vector stores.
// acc[i][j][r] = C[row0 + wm*64 + i*16 + (lane&15)][n0 + wn*64 + j*16 + 4*(lane>>4) + r]
template <int MI>
__device__ __forceinline__ void gemm_nt_epilogue(const GemmNTArgs& a, f32x4_t (&acc)[MI][4], int group, int row0,
                                                 int row_end, int n0, int wm, int wn, int lane) {
  const float* bias = a.bias ? a.bias + (long)group * a.bias_gstride : nullptr;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = row0 + wm * (MI * 16) + i * 16 + (lane & 15);
    if (m >= row_end) continue;
    float rs = 1.f;
    if (a.row_scale) {
      const int sample = group ? (m - a.split) / a.rps1 : m / a.rps0;
      rs = a.row_scale[sample];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + 4 * (lane >> 4);
      if (n >= a.N) continue;
      float v[4] = {acc[i][j][0] * a.alpha, acc[i][j][1] * a.alpha, acc[i][j][2] * a.alpha, acc[i][j][3] * a.alpha};
      const bool full = (n + 3 < a.N);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (bias && (full || n + r < a.N)) v[r] += bias[n + r];
      if (a.aux) {
        lp_t* p = a.aux + (long)m * a.ldaux + n;
        if (full) {
          *(u32x2_t*)p = (u32x2_t){pack_lp2(v[0], v[1]), pack_lp2(v[2], v[3])};
        } else {
          for (int r = 0; r < 4 && n + r < a.N; ++r) p[r] = f32_to_lp(v[r]);
        }
      }
      if (a.act == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
      } else if (a.act == 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
      }
      if (a.res) {
        const float* rp = a.res + (long)m * a.ldres + n;
        if (full) {
          const f32x4_t rv = *(const f32x4_t*)rp;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(rs, v[r], rv[r]);
        } else {
          for (int r = 0; r < 4 && n + r < a.N; ++r) v[r] = __builtin_fmaf(rs, v[r], rp[r]);
        }
      } else if (a.row_scale) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= rs;
      }
      if (a.c_f32) {
        float* p = (float*)a.C + (long)m * a.ldc + n;
        if (full) {
          *(f32x4_t*)p = (f32x4_t){v[0], v[1], v[2], v[3]};
        } else {
          for (int r = 0; r < 4 && n + r < a.N; ++r) p[r] = v[r];
        }
      } else {
        lp_t* p = (lp_t*)a.C + (long)m * a.ldc + n;
        if (full) {
          *(u32x2_t*)p = (u32x2_t){pack_lp2(v[0], v[1]), pack_lp2(v[2], v[3])};
        } else {
          for (int r = 0; r < 4 && n + r < a.N; ++r) p[r] = f32_to_lp(v[r]);
        }
      }
    }
  }
}

// Coalesced epilogue: each wave stages (acc + bias) of 32 x 64 outputs at a time through its own 8.5 KiB slice of
// the (finished) LDS ring and writes them back row-wise, 16 B per lane: a store instruction covers 8 rows x 128 B
// (bf16) or 4 rows x 256 B (fp32) instead of 16 rows x 32 B.  Activation, the bf16 pre-activation copy, DropPath
// scale and the residual add are applied at read-out, so GELU/residual operands are read coalesced as well.
// Requires N % 8 == 0 (vector path); callers fall back to gemm_nt_epilogue otherwise.
constexpr int EPI_LD = 68;                       // floats per staged row (64 + 4 pad)
constexpr int EPI_WAVE_BYTES = 32 * EPI_LD * 4;  // 8704 B per wave

// MI x NJ = 16x16 accumulator blocks of one wave (rows x columns); NJ = 4 everywhere except the 16-wave 160x256 kernel
template <int MI, int NJ = 4>
__device__ __forceinline__ void gemm_nt_epilogue_lds(const GemmNTArgs& a, f32x4_t (&acc)[MI][NJ], int group, int row0,
                                                     int row_end, int n0, int wm, int wn, int wave, int lane,
                                                     char* smem) {
  // fp32 + residual epilogue (out-proj, fc2: the residual stream): the residual rows of a 32-row pass are requested one pass
  // AHEAD of their use -- pass 0's before the workgroup's rendezvous, pass p + 1's while pass p is staged and written -- instead
  // of inside the read-out loop, where every pass waited a full memory latency for them (round 4)
  constexpr int LPRF = NJ * 4, RPPF = 64 / LPRF, NITF = 32 / RPPF;
  const bool pre_res = a.c_f32 && a.res && a.respf;
  f32x4_t rnext[NITF];
  auto load_res = [&](int p) {
    const int mb = row0 + wm * (MI * 16) + p * 32, nb = n0 + wn * (NJ * 16);
#pragma unroll
    for (int it = 0; it < NITF; ++it) {
      const int r = it * RPPF + lane / LPRF, c = (lane % LPRF) * 4;
      const int m = mb + r, n = nb + c;
      rnext[it] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      if (m < row_end && n < a.N && 2 * p + (r >> 4) < MI) rnext[it] = *(const f32x4_t*)(a.res + (long)m * a.ldres + n);
    }
  };
  if (pre_res) load_res(0);
  __syncthreads();                                // every wave is done reading the ring
  constexpr int LD = NJ * 16 + 4;                 // floats per staged row (EPI_LD for NJ = 4)
  float* st = (float*)(smem + wave * (32 * LD * 4));   // wave-private slice (EPI_WAVE_BYTES for NJ = 4)
  const float* bias = a.bias ? a.bias + (long)group * a.bias_gstride : nullptr;
  const int nbase = n0 + wn * (NJ * 16);
  float bv[NJ][4];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int n = nbase + j * 16 + 4 * (lane >> 4);
    if (bias && n < a.N) {
      const f32x4_t t = *(const f32x4_t*)(bias + n);
      bv[j][0] = t[0]; bv[j][1] = t[1]; bv[j][2] = t[2]; bv[j][3] = t[3];
    } else {
      bv[j][0] = bv[j][1] = bv[j][2] = bv[j][3] = 0.f;
    }
  }
#pragma unroll
  for (int p = 0; p < (MI + 1) / 2; ++p) {
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
      if (2 * p + ii >= MI) continue;      // odd MI: the last pass stages 16 rows only
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const f32x4_t v = acc[2 * p + ii][j];
        *(f32x4_t*)(st + (ii * 16 + (lane & 15)) * LD + j * 16 + 4 * (lane >> 4)) =
            (f32x4_t){fmaf(v[0], a.alpha, bv[j][0]), fmaf(v[1], a.alpha, bv[j][1]), fmaf(v[2], a.alpha, bv[j][2]),
                      fmaf(v[3], a.alpha, bv[j][3])};
      }
    }
    f32x4_t rcur[NITF];
    if (pre_res) {
#pragma unroll
      for (int it = 0; it < NITF; ++it) rcur[it] = rnext[it];
      if (p + 1 < (MI + 1) / 2) load_res(p + 1);
    }
    // wave-private slice: only this wave's own LDS writes must have landed (no barrier)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int mbase = row0 + wm * (MI * 16) + p * 32;
    if (!a.c_f32) {
      // bf16 output: lane -> 8 consecutive columns, 2*NJ lanes per row (NJ = 4: 8 lanes, 8 rows per pass)
      constexpr int LPR = NJ * 2, RPP = 64 / LPR;
#pragma unroll
      for (int it = 0; it < 32 / RPP; ++it) {
        const int r = it * RPP + lane / LPR, c = (lane % LPR) * 8;
        const int m = mbase + r, n = nbase + c;
        if (m >= row_end || n >= a.N || 2 * p + (r >> 4) >= MI) continue;
        const f32x4_t u0 = *(const f32x4_t*)(st + r * LD + c), u1 = *(const f32x4_t*)(st + r * LD + c + 4);
        float v[8] = {u0[0], u0[1], u0[2], u0[3], u1[0], u1[1], u1[2], u1[3]};
        if (a.aux)
          *(u32x4_t*)(a.aux + (long)m * a.ldaux + n) = (u32x4_t){pack_lp2(v[0], v[1]), pack_lp2(v[2], v[3]),
                                                                 pack_lp2(v[4], v[5]), pack_lp2(v[6], v[7])};
        if (a.act == 1) {
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = gelu_erf(v[k]);
        } else if (a.act == 2) {
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
        }
        if (a.row_scale) {
          const float rs = a.row_scale[group ? (m - a.split) / a.rps1 : m / a.rps0];
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] *= rs;
        }
        *(u32x4_t*)((lp_t*)a.C + (long)m * a.ldc + n) = (u32x4_t){pack_lp2(v[0], v[1]), pack_lp2(v[2], v[3]),
                                                                    pack_lp2(v[4], v[5]), pack_lp2(v[6], v[7])};
      }
    } else {
      // fp32 output (+ residual): lane -> 4 consecutive columns, 4*NJ lanes per row (NJ = 4: 16 lanes, 4 rows per pass)
      constexpr int LPR = NJ * 4, RPP = 64 / LPR;
#pragma unroll
      for (int it = 0; it < 32 / RPP; ++it) {
        const int r = it * RPP + lane / LPR, c = (lane % LPR) * 4;
        const int m = mbase + r, n = nbase + c;
        if (m >= row_end || n >= a.N || 2 * p + (r >> 4) >= MI) continue;
        const f32x4_t u = *(const f32x4_t*)(st + r * LD + c);
        float v[4] = {u[0], u[1], u[2], u[3]};
        if (a.aux) *(u32x2_t*)(a.aux + (long)m * a.ldaux + n) = (u32x2_t){pack_lp2(v[0], v[1]), pack_lp2(v[2], v[3])};
        if (a.act == 1) {
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = gelu_erf(v[k]);
        } else if (a.act == 2) {
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
        }
        float rs = 1.f;
        if (a.row_scale) rs = a.row_scale[group ? (m - a.split) / a.rps1 : m / a.rps0];
        if (a.res) {
          const f32x4_t rv = pre_res ? rcur[it] : *(const f32x4_t*)(a.res + (long)m * a.ldres + n);
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = __builtin_fmaf(rs, v[k], rv[k]);     // (explicit: every kernel's fp32 epilogue is this fma)
        } else if (a.row_scale) {
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] *= rs;
        }
        *(f32x4_t*)((float*)a.C + (long)m * a.ldc + n) = (f32x4_t){v[0], v[1], v[2], v[3]};
      }
    }
    // the next pass overwrites the slice: this wave's reads must have completed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// 128x128x64 tiles, 4 waves (2 x 2, each 64x64), FOUR-stage LDS ring (4 x 32 KiB) with counted vmcnt: the fallback for
// problems with few rows and many columns (M < 512 and more than 800 64x64 tiles) and for N that is no multiple of 256 at
// M < 512.  Round 1's version waited for every k-tile's loads in full; here two younger k-tiles stay in flight.
constexpr int MID_NST = 4;
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmNTArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (a.N + BN - 1) / BN;
  const int tm0 = (a.split + BM - 1) / BM;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
  const int group = tile_m >= tm0;
  const int row0 = group ? a.split + (tile_m - tm0) * BM : tile_m * BM;
  const int row_end = group ? a.M : a.split;
  const int n0 = tile_n * BN;
  const lp_t* W = a.W + (long)group * a.w_gstride;

#define ldsA(c) (smem + (c) * 2 * TILE_BYTES)
#define ldsB(c) (smem + TILE_BYTES + (c) * 2 * TILE_BYTES)
#define ISSUE(t_)                                                                              \
  do {                                                                                         \
    const int st__ = (t_) % MID_NST;                                                           \
    stage_tile_k64(a.A, a.lda, row0, row_end - 1, a_koff(a, (t_) * BK), ldsA(st__), wave, lane); \
    stage_tile_k64(W, a.ldw, n0, a.N - 1, (t_) * BK, ldsB(st__), wave, lane);                  \
  } while (0)

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int nk = a.K / BK;
#pragma unroll
  for (int t = 0; t < MID_NST - 1; ++t)
    if (t < nk) ISSUE(t);
  for (int kt = 0; kt < nk; ++kt) {
    // this wave's 8 pieces of tile kt have landed; up to two younger tiles (8 pieces each) stay in flight
    const int young = min(MID_NST - 2, nk - 1 - kt);
    if (young >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (young == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + MID_NST - 1 < nk) ISSUE(kt + MID_NST - 1);
    const int cur = kt % MID_NST;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      lpx8_t fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i] = read_frag_k64(ldsA(cur), wm * 64 + i * 16 + (lane & 15), s * 4 + (lane >> 4));
        fb[i] = read_frag_k64(ldsB(cur), wn * 64 + i * 16 + (lane & 15), s * 4 + (lane >> 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          // D[n_local][m_local]: lane holds 4 consecutive n for one m -> vector stores
          acc[i][j] = mfma_lp(fb[j], fa[i], acc[i][j]);
    }
    split_rescale(a, acc, (kt + 1) * BK);
  }
#undef ISSUE
  gemm_nt_epilogue<4>(a, acc, group, row0, row_end, n0, wm, wn, lane);
}


// ------------------------------------------------------------------------------------------
// 256x128x64 tile, 8 waves (4 x 2, each 64x64), 3-stage LDS ring (3 x 48 KiB), counted vmcnt + raw s_barrier:
// the global_load_lds of k-tile t+2 are issued right after the barrier of iteration t and stay in flight across
// the next barrier; a wave only waits (vmcnt(6)) for its own share of tile t.  25 % less L2->LDS traffic per FLOP
// than the 128^2 tile and no vmcnt(0) drain in the main loop.
// ------------------------------------------------------------------------------------------
constexpr int BM2 = 256;
constexpr int STAGE2 = (BM2 + BN) * BK * 2;   // 49152 B

__device__ __forceinline__ void stage_tile_k64_n(const lp_t* base, int ld, int row0, int row_last, int k0, char* lds,
                                                 int wave, int lane, int per_wave) {
  for (int i = 0; i < per_wave; ++i) {
    const int inst = wave * per_wave + i;
    const int r = inst * 8 + (lane >> 3);
    int row = row0 + r;
    row = row < row_last ? row : row_last;
    const int lslot = (lane & 7) ^ (r & 7);
    const lp_t* src = base + (long)row * ld + k0 + lslot * 8;
    __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(lds + inst * 1024), 16, 0, 0);
  }
}

// ------------------------------------------------------------------------------------------
// 256x256x64 tile for the wide-N problems (QKV, fc1, dgrad of fc2: N >= 2304): 8 waves as 2 (M) x 4 (N), each
// 128x64 = acc[8][4] (128 accumulator VGPRs), two 64 KiB LDS buffers, one rendezvous per k-tile with the loads of
// tile t+1 in flight under the 64 MFMAs of tile t.  128 FLOP per staged byte (256x128: 85): the L2->LDS path
// (global_load_lds issue ~16+ cycles per KiB) and the LDS reads drop to half of the MFMA time.
// ------------------------------------------------------------------------------------------
constexpr int BNQ = 256;

// [n_inst * 8 rows][64 k] tile, wave w stages instructions w, w+8, ... (n_inst need not be a multiple of 8)
__device__ __forceinline__ void stage_rows_k64(const lp_t* base, int ld, int row0, int row_last, int k0, char* lds,
                                               int wave, int lane, int n_inst, int nwaves = 8) {
  for (int inst = wave; inst < n_inst; inst += nwaves) {
    const int r = inst * 8 + (lane >> 3);
    int row = row0 + r;
    row = row < row_last ? row : row_last;
    const int lslot = (lane & 7) ^ (r & 7);
    const lp_t* src = base + (long)row * ld + k0 + lslot * 8;
    __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(lds + inst * 1024), 16, 0, 0);
  }
}

// ------------------------------------------------------------------------------------------
// Latency kernel for problems that cannot fill the chip with big tiles (forward_test at B = 1 ... 8: M = 421 ... 3368 rows):
// 64x64x64 tiles, 4 waves (each 16 rows x 64 columns), LAT_NST-stage LDS ring of 16 KiB stages, counted vmcnt (a wave waits for
// its own 4 pieces of the oldest k-tile only).  With few workgroups per CU the time of such a GEMM is (k-tiles) x (time per
// k-tile) + launch / prologue / epilogue; the 128x128 kernel of round 1 waited for each k-tile's loads in full (~0.7 us per
// k-tile, one workgroup per CU).  Sweep (tools/dev/gemm_small_bench.py): a THREE-stage ring (48 KiB: three workgroups resident
// per CU, so one workgroup's prologue / epilogue hides behind the others' main loops) beats the six-stage ring (96 KiB, one
// workgroup per CU) at every size -- B = 1: 48.6 vs 61.7 us for the four encoder shapes, B = 8: 128.6 vs 227.1 us.
// ------------------------------------------------------------------------------------------
constexpr int LAT_STAGE = (64 + 64) * BK * 2;      // 16 KiB

template <int LAT_NST>
__global__ __launch_bounds__(256) void gemm_nt_kernel_lat(GemmNTArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (a.N + 63) / 64;
  const int tm0 = (a.split + 63) / 64;
  const int bid = blockIdx.x;
  const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
  const int group = tile_m >= tm0;
  const int row0 = group ? a.split + (tile_m - tm0) * 64 : tile_m * 64;
  const int row_end = group ? a.M : a.split;
  const int n0 = tile_n * 64;
  const lp_t* W = a.W + (long)group * a.w_gstride;
  f32x4_t acc[1][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[0][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int nk = a.K / BK;
#define STA(s_) (smem + (s_) * LAT_STAGE)
#define STB(s_) (smem + (s_) * LAT_STAGE + 64 * BK * 2)
#define ISSUE(t_)                                                                              \
  do {                                                                                         \
    const int st__ = (t_) % LAT_NST;                                                           \
    stage_tile_k64_n(a.A, a.lda, row0, row_end - 1, a_koff(a, (t_) * BK), STA(st__), wave, lane, 2); \
    stage_tile_k64_n(W, a.ldw, n0, a.N - 1, (t_) * BK, STB(st__), wave, lane, 2);              \
  } while (0)
#pragma unroll
  for (int t = 0; t < LAT_NST - 1; ++t)
    if (t < nk) ISSUE(t);
  for (int kt = 0; kt < nk; ++kt) {
    // this wave's 4 pieces of tile kt have landed; up to LAT_NST - 2 younger tiles (4 pieces each) stay in flight
    const int young = min(LAT_NST - 2, nk - 1 - kt);
    if (LAT_NST >= 6 && young >= 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (LAT_NST >= 5 && young == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (young == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (young == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // everyone's pieces have, and everyone is past compute(kt - 1): its slot is free
    if (kt + LAT_NST - 1 < nk) ISSUE(kt + LAT_NST - 1);
    const char* sA = STA(kt % LAT_NST);
    const char* sB = STB(kt % LAT_NST);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      lpx8_t fb[4];
      const lpx8_t fa = read_frag_k64(sA, wave * 16 + (lane & 15), s * 4 + (lane >> 4));
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = read_frag_k64(sB, j * 16 + (lane & 15), s * 4 + (lane >> 4));
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[0][j] = mfma_lp(fb[j], fa, acc[0][j]);
    }
    split_rescale(a, acc, (kt + 1) * BK);
  }
#undef STA
#undef STB
#undef ISSUE
  gemm_nt_epilogue<1>(a, acc, group, row0, row_end, n0, wave, 0, lane);
}

// 256x256x64 tile with SIXTEEN waves (4 x 4, each 64x64; 4 waves per SIMD, 112 VGPRs): more waves cover the LDS-read and
// rendezvous latencies of the one-workgroup-per-CU tile and issue the store-heavy epilogues 2x wider, at the price of 33 %
// more LDS reads per MFMA (still ~50 % of the LDS bandwidth).  fc1 shape 124 vs 134 us, with fp32 residual epilogue
// 212 vs 275 us (same box, warm).
__global__ __launch_bounds__(1024) void gemm_nt_kernel_256sq_w16(GemmNTArgs a) {
  constexpr int MI = 4;
  constexpr int BMQ = 256;
  constexpr int STAGEQ = (BMQ + BNQ) * BK * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;   // 4 x 4 waves
  const int tiles_n = (a.N + BNQ - 1) / BNQ;
  const int tm0 = (a.split + BMQ - 1) / BMQ;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  int tile_m, tile_n;
  tile_order(bid, (int)gridDim.x / tiles_n, tiles_n, a.gn, tile_m, tile_n);
  const int group = tile_m >= tm0;
  const int row0 = group ? a.split + (tile_m - tm0) * BMQ : tile_m * BMQ;
  const int row_end = group ? a.M : a.split;
  const int n0 = tile_n * BNQ;
  const lp_t* W = a.W + (long)group * a.w_gstride;

  f32x4_t acc[MI][4];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int nk = a.K / BK;
#define STA(s_) (smem + (s_) * STAGEQ)
#define STB(s_) (smem + (s_) * STAGEQ + BMQ * BK * 2)
#define ISSUE(t_)                                                                                  \
  do {                                                                                             \
    const int st__ = (t_) & 1;                                                                     \
    stage_tile_k64_n(a.A, a.lda, row0, row_end - 1, a_koff(a, (t_) * BK), STA(st__), wave, lane, 2); \
    stage_tile_k64_n(W, a.ldw, n0, a.N - 1, (t_) * BK, STB(st__), wave, lane, 2);                  \
  } while (0)

  ISSUE(0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's share of tile kt has landed
    __builtin_amdgcn_s_barrier();                        // everyone's has, and everyone is past compute(kt-1)
    if (kt + 1 < nk) ISSUE(kt + 1);
    const char* sA = STA(kt & 1);
    const char* sB = STB(kt & 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      lpx8_t fa[MI], fb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = read_frag_k64(sB, wn * 64 + j * 16 + (lane & 15), s * 4 + (lane >> 4));
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[i] = read_frag_k64(sA, wm * 64 + i * 16 + (lane & 15), s * 4 + (lane >> 4));
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = mfma_lp(fb[j], fa[i], acc[i][j]);
    }
    split_rescale(a, acc, (kt + 1) * BK);
  }
#undef STA
#undef STB
#undef ISSUE
  gemm_nt_epilogue_lds<4>(a, acc, group, row0, row_end, n0, wm, wn, wave, lane, smem);
}


// ------------------------------------------------------------------------------------------
// PERSISTENT form of the 256x256x64 / 16-wave kernel for 16-bit outputs with a plain (+bias) epilogue -- QKV, fc1, the dgrad
// of fc2: launches whose 16-bit output is as large as their operands.  One workgroup per CU walks tiles v, v + G, v + 2G ...
// (same XCD-contiguous order as the hardware dispatch of the one-tile kernel) and never drains its stores:
//   * k-tile 0 of the NEXT tile is put in flight under the last MFMAs of this one, so the next tile starts warm (issuing
//     k-tile 1 as well needs a barrier before the epilogue -- every wave must have left the ring's other stage -- and
//     measured 2 % slower: without it the early waves of a SIMD convert and store while the late ones still own the MFMA pipe);
//   * the epilogue stages through a 32 KiB region of its own (2 KiB per wave, 16-bit, XOR-swizzled) -- the ring stays
//     loadable -- and its 8 stores per wave are left in flight: CDNA4 counts stores in vmcnt, in order with the loads, so the
//     next tile's wait for its first k-tile is counted PAST them (FIFO of a wave at the top of a tile, oldest first:
//     L0 B S -> vmcnt(8); at k-tile 1: S L1 -> vmcnt(0)): the stores have the epilogue plus one MFMA interval to retire;
//   * the bias (B: 4 loads, not tracked by the compiler) is the accumulators' initial value, so the epilogue reads no global memory;
//   * a tile with rows beyond the row group masks stores per lane, which the compiler may branch around: such a tile ends with
//     vmcnt(0) (an uncounted store would make the next waits too permissive).
// ------------------------------------------------------------------------------------------
constexpr int PQ_STAGE = (256 + BNQ) * BK * 2;     // 64 KiB
constexpr int PQ_STG = 2 * PQ_STAGE;                // staging: wave w owns [PQ_STG + 2048 w, +2048)
constexpr int PQ_SMEM = PQ_STG + 16 * 2048;         // 160 KiB
constexpr int PQ_NS = 8;                            // stores per wave and tile

struct PQTile { int row0, row_end, n0, group; };

__device__ __forceinline__ PQTile pq_decode(const GemmNTArgs& a, int v, int ntot, int tiles_n, int tm0) {
  const int bid = xcd_remap(v, ntot);
  int tm, tn;
  tile_order(bid, ntot / tiles_n, tiles_n, a.gn, tm, tn);
  PQTile t;
  t.group = tm >= tm0;
  t.row0 = t.group ? a.split + (tm - tm0) * 256 : tm * 256;
  t.row_end = t.group ? a.M : a.split;
  t.n0 = tn * BNQ;
  return t;
}

template <int N_> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }
template <int N_> __device__ __forceinline__ void lgkm_wait() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N_) : "memory");
  __builtin_amdgcn_sched_barrier(0);         // register-only instructions (MFMA, packs) must not be hoisted above the wait
}
// a load the compiler does not track (its own wait before the first use would be vmcnt(0) across the loop's back edge): the
// caller waits with a counted vmcnt and pins the registers afterwards
__device__ __forceinline__ f32x4_t gload_x4_untracked(const float* p) {
  f32x4_t v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void lds_write_b64_asm(unsigned addr, u32x2_t v) {
  asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ lpx8_t as_frag(u32x4_t v) { return __builtin_bit_cast(lpx8_t, v); }
// makes v opaque at this point (no consumer of v is scheduled above it, no copy of it is folded across).  As a device FUNCTION: a
// kernel TEMPLATE's body is instantiated by the host pass as well, where a bare asm statement with an AMDGPU register constraint is a
// substitution failure (the specialisation silently loses its host stub); calls into device-only functions are not checked there
__device__ __forceinline__ void reg_pin(f32x4_t& v) { asm volatile("" : "+v"(v)); }

// All LDS traffic of this kernel is inline asm (common.h: hipcc drains vmcnt in front of compiler-visible LDS accesses that
// follow a buffer-form LDS-DMA), all DMA addressing is (SGPR descriptor of the tile) + (4 loop-invariant lane offsets).
#ifdef SIMVG_PQ_PROFILE
#define PQ_T(k_) do { if (prof && wave == 0 && lane == 0 && tix < 8) prof[((long)blockIdx.x * 8 + tix) * 8 + (k_)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PQ_T(k_) do { } while (0)
#endif
// TWO: hi + lo weights (round 6: the training forward's qkv, `BEIT3.precise_training`): W rows are [lo * 2^s | hi] (ka = K / 2), A is
// walked twice, the accumulators are multiplied by lo_scale = 2^-s after the lo half -- the bias, which is their start value here,
// enters as bias * 2^s (a power of two: exact).  A kernel of its own (the plain kernel's registers are untouched), its k loop in two
// halves with the rescale between them (a branch inside the loop body cost the registers that made the build spill).
template <bool TWO>
__device__ __forceinline__ void gemm_nt_pq_body(const GemmNTArgs& a, int ntot, unsigned long long* prof, char* smem) {
  constexpr int MI = 4;
  int tix = 0;
  (void)tix; (void)prof;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;   // 4 x 4 waves
  const int tiles_n = a.N / BNQ;
  const int tm0 = (a.split + 255) / 256;
  const int nk = a.K / BK;                   // even, >= 2 (launcher)
  // LDS-DMA: piece p = 2 wave + i holds rows 8 p .. 8 p + 7 of an operand's 256 x 64 k-tile, lane -> row 8 p + (lane >> 3),
  // 16-B slot (lane & 7) ^ (row & 7) of the row (the XOR is on the SOURCE address: the LDS image is lane-linear)
  const int din = lane >> 3, dslot = (lane & 7) ^ din;
  int voffA[2], voffW[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    voffA[i] = ((wave * 2 + i) * 8 + din) * a.lda * 2 + dslot * 16;
    voffW[i] = ((wave * 2 + i) * 8 + din) * a.ldw * 2 + dslot * 16;
  }
  // fragment reads: row (lane & 15) of a 16-row block, slot (4 s + (lane >> 4)) ^ (row & 7)
  const int c0 = (lane >> 4) ^ (lane & 7);
  const unsigned fA0 = lds_addr(smem) + (wm * 64 + (lane & 15)) * 128 + c0 * 16, fA1 = fA0 ^ 64;
  const unsigned fB0 = lds_addr(smem) + 256 * BK * 2 + (wn * 64 + (lane & 15)) * 128 + c0 * 16, fB1 = fB0 ^ 64;

  struct Desc { __amdgpu_buffer_rsrc_t A, W; const float* bias; };
  auto make_desc = [&](const PQTile& t) {
    Desc d;
    d.A = __builtin_amdgcn_make_buffer_rsrc((void*)(a.A + (long)t.row0 * a.lda), 0, (int)((long)(t.row_end - t.row0) * a.lda * 2), 0x00020000);
    d.W = __builtin_amdgcn_make_buffer_rsrc((void*)(a.W + (long)t.group * a.w_gstride + (long)t.n0 * a.ldw), 0, BNQ * a.ldw * 2, 0x00020000);
    d.bias = a.bias ? a.bias + (long)t.group * a.bias_gstride + t.n0 : nullptr;
    return d;
  };
  auto issue = [&](const Desc& d, int kt, int st) {
    char* sA = smem + st * PQ_STAGE + wave * 2048;
    if constexpr (TWO) {
      // the second piece's row offset rides in the scalar offset (two VGPRs less: this variant's spilled SGPRs need them)
      const int ka = (kt >= (nk >> 1) ? kt - (nk >> 1) : kt) * (BK * 2);
#pragma unroll
      for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(d.A, LDS_PTR(sA + i * 1024), 16, voffA[0], ka + i * 8 * a.lda * 2, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(d.W, LDS_PTR(sA + 256 * BK * 2 + i * 1024), 16, voffW[0], kt * (BK * 2) + i * 8 * a.ldw * 2, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(d.A, LDS_PTR(sA + i * 1024), 16, voffA[i], kt * (BK * 2), 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(d.W, LDS_PTR(sA + 256 * BK * 2 + i * 1024), 16, voffW[i], kt * (BK * 2), 0, 0);
    }
  };
  // the bias enters as the accumulators' initial value: bn[j] = bias of columns n0 + wn * 64 + j * 16 + 4 (lane >> 4) .. + 3
  f32x4_t bn[4];
  auto load_bias = [&](const Desc& d) {
    if (d.bias) {
      const float* bp = d.bias + wn * 64 + 4 * (lane >> 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) bn[j] = gload_x4_untracked(bp + j * 16);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) bn[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
  };

  int v = blockIdx.x;
  PQTile cur = pq_decode(a, v, ntot, tiles_n, tm0);
  Desc dcur = make_desc(cur);
  issue(dcur, 0, 0);
  issue(dcur, 1, 1);
  load_bias(dcur);
  bool pend = false;                         // PQ_NS stores of the previous tile are the youngest entries of this wave's FIFO
  bool first = true;
  for (;;) {
    const int vn = v + (int)gridDim.x;
    const bool has_next = vn < ntot;
    PQTile nxt = cur;
    if (has_next) nxt = pq_decode(a, vn, ntot, tiles_n, tm0);
    const Desc dnxt = make_desc(nxt);
    // FIFO, oldest first: L0 [L1] B [S] (L1: first tile only): everything but the stores has to be here
    PQ_T(0);
    if (pend) vm_wait<PQ_NS>(); else vm_wait<0>();
    PQ_T(1);
    asm volatile("" : "+v"(bn[0]), "+v"(bn[1]), "+v"(bn[2]), "+v"(bn[3]));
    if constexpr (TWO) {
      const float inv = 1.f / a.lo_scale;
#pragma unroll
      for (int j = 0; j < 4; ++j) bn[j] *= inv;
    }
    f32x4_t acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[i][j] = bn[j];
        asm volatile("" : "+v"(acc[i][j]));   // copies now: bn's registers are free during the main loop
      }
    auto ktile = [&](int kt) {
      if (kt == 1) PQ_T(2);
      if (kt >= 1) vm_wait<0>();             // k-tile 0: waited for above;  k-tile 1: FIFO = [S] L1, the stores have to be through
      if (kt == 1) PQ_T(3);
      __builtin_amdgcn_s_barrier();          // k-tile kt has landed for everyone, everyone is past the MFMAs of kt - 1
      if (kt >= 1 || !first) {               // (the first tile's k-tile 1 was issued by the prologue)
        if (kt + 1 < nk) issue(dcur, kt + 1, (kt + 1) & 1);
        else if (has_next) issue(dnxt, 0, 0);
      }
      const unsigned so = (kt & 1) * PQ_STAGE;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const unsigned pa = (s ? fA1 : fA0) + so, pb = (s ? fB1 : fB0) + so;
        u32x4_t fb[4], fa[2];
        fb[0] = lds_b128_asm<0>(pb); fb[1] = lds_b128_asm<2048>(pb); fb[2] = lds_b128_asm<4096>(pb); fb[3] = lds_b128_asm<6144>(pb);
        fa[0] = lds_b128_asm<0>(pa);
        fa[1] = lds_b128_asm<2048>(pa);
        lgkm_wait<1>();
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[0][j] = mfma_lp(as_frag(fb[j]), as_frag(fa[0]), acc[0][j]);
        __builtin_amdgcn_sched_barrier(0);
        fa[0] = lds_b128_asm<4096>(pa);
        lgkm_wait<1>();
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[1][j] = mfma_lp(as_frag(fb[j]), as_frag(fa[1]), acc[1][j]);
        __builtin_amdgcn_sched_barrier(0);
        fa[1] = lds_b128_asm<6144>(pa);
        lgkm_wait<1>();
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[2][j] = mfma_lp(as_frag(fb[j]), as_frag(fa[0]), acc[2][j]);
        lgkm_wait<0>();
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[3][j] = mfma_lp(as_frag(fb[j]), as_frag(fa[1]), acc[3][j]);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if constexpr (TWO) {
      for (int kt = 0; kt < (nk >> 1); ++kt) ktile(kt);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] *= a.lo_scale;      // the lo half is complete
      for (int kt = nk >> 1; kt < nk; ++kt) ktile(kt);
    } else {
      for (int kt = 0; kt < nk; ++kt) ktile(kt);
    }
    PQ_T(4);
    if (has_next) load_bias(dnxt);
    PQ_T(5);
    // ---- epilogue of `cur`: 16 bit -> wave-private staging -> row-wise 16-B stores, left in flight
    {
      const unsigned stg = lds_addr(smem) + PQ_STG + wave * 2048;
      const int er = lane & 15, ecg = lane >> 4;         // accumulator layout: row er, columns 4 ecg .. + 3 of each 16 x 16 block
      const int rr = lane >> 3, rq = lane & 7;           // read-out layout: rows rr and rr + 8, 16-B chunk rq
      lp_t* cp = (lp_t*)a.C + (long)(cur.row0 + wm * 64 + rr) * a.ldc + cur.n0 + wn * 64 + rq * 8;
      const int mleft = cur.row_end - (cur.row0 + wm * 64 + rr);
      // 8-B chunk j * 4 + ecg of row er is stored at chunk ^ er (the 16 rows of a store's lane group hit 16 different banks);
      // the 16-B chunk rq of row r is then found at chunk rq ^ (r >> 1), its halves swapped for odd r
      const unsigned wa = stg + er * 128, ra = stg + rr * 128 + ((rq ^ (rr >> 1)) << 4);
#pragma unroll
      for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4_t x = acc[i][j];
          lds_write_b64_asm(wa + (((j * 4 + ecg) ^ er) << 3), (u32x2_t){pack_lp2(x[0], x[1]), pack_lp2(x[2], x[3])});
        }
        lgkm_wait<0>();                                  // wave-private slice: only this wave's writes matter
        u32x4_t d0 = lds_b128_asm<0>(ra);
        u32x4_t d1 = lds_b128_asm<0>(ra ^ 1088);         // row + 8: offset 1024, chunk ^ 4
        lgkm_wait<0>();                                  // (also: the next pass overwrites the slice)
        if (rr & 1) { d0 = (u32x4_t){d0[2], d0[3], d0[0], d0[1]}; d1 = (u32x4_t){d1[2], d1[3], d1[0], d1[1]}; }
        if (i * 16 < mleft) *(u32x4_t*)(cp + (long)(i * 16) * a.ldc) = d0;
        if (i * 16 + 8 < mleft) *(u32x4_t*)(cp + (long)(i * 16 + 8) * a.ldc) = d1;
      }
    }
    PQ_T(6);
    ++tix;
    pend = true;
    if (cur.row0 + 256 > cur.row_end) { vm_wait<0>(); pend = false; }    // lane-masked stores are not countable
    if (!has_next) break;
    cur = nxt;
    dcur = dnxt;
    v = vn;
    first = false;
  }
}
__global__ __launch_bounds__(1024) void gemm_nt_kernel_256sq_p(GemmNTArgs a, int ntot, unsigned long long* prof) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  gemm_nt_pq_body<false>(a, ntot, prof, smem);
}
__global__ __launch_bounds__(1024) void gemm_nt_kernel_256sq_p2(GemmNTArgs a, int ntot, unsigned long long* prof) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  gemm_nt_pq_body<true>(a, ntot, prof, smem);
}

// 160x256x64 tile with sixteen waves: 2 (M) x 8 (N), each 80x32 = acc[5][2] -- the 16-wave layout for the N = 768 problems
// (160 rows do not split over 4 M-waves).  7 fragment reads per 10 MFMAs (8-wave layout: 9 per 20).
// THREE-stage LDS ring (3 x 52 KiB = 156 KiB, the double buffer left 54 KiB of the CU's LDS unused): the
// LDS-DMA of k-tile t+2 is issued at the top of iteration t and has two compute intervals to land.  With one stage of
// look-ahead the ring holds 52 KiB in flight per CU; a CU that consumes a 52 KiB stage every ~0.8 us (75 GB/s) across a
// ~1.5 us L2 / HBM latency needs ~110 KiB in flight (Little) -- the double-buffered kernel waited on `vmcnt(0)` at every
// rendezvous (PMC: 54 % of the wave cycles in waitcnt / barrier).  A wave waits for its OWN pieces of the oldest stage only
// (counted vmcnt: 3 or 4 pieces per wave and stage).
__global__ __launch_bounds__(1024) void gemm_nt_kernel_160x256_r3(GemmNTArgs a) {
  constexpr int BMQ = 160;
  constexpr int STAGEQ = (BMQ + BNQ) * BK * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 3, wn = wave & 7;
  const int tiles_n = (a.N + BNQ - 1) / BNQ;
  const int tm0 = (a.split + BMQ - 1) / BMQ;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
  const int group = tile_m >= tm0;
  const int row0 = group ? a.split + (tile_m - tm0) * BMQ : tile_m * BMQ;
  const int row_end = group ? a.M : a.split;
  const int n0 = tile_n * BNQ;
  const lp_t* W = a.W + (long)group * a.w_gstride;
  f32x4_t acc[5][2];
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int nk = a.K / BK;
  const bool four = wave < 4;            // A: 20 pieces over 16 waves (waves 0-3 take two), B: 32 pieces (two each)
#define STA(s_) (smem + (s_) * STAGEQ)
#define STB(s_) (smem + (s_) * STAGEQ + BMQ * BK * 2)
#define ISSUE(t_)                                                                                      \
  do {                                                                                                 \
    const int st__ = (t_) % 3;                                                                         \
    stage_rows_k64(a.A, a.lda, row0, row_end - 1, a_koff(a, (t_) * BK), STA(st__), wave, lane, BMQ / 8, 16); \
    stage_rows_k64(W, a.ldw, n0, a.N - 1, (t_) * BK, STB(st__), wave, lane, BNQ / 8, 16);              \
  } while (0)
  ISSUE(0);
  if (nk > 1) ISSUE(1);
  for (int kt = 0; kt < nk; ++kt) {
    // this wave's share of tile kt has landed; the pieces of tile kt+1 may still be in flight
    if (kt + 1 < nk) {
      if (four) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                        // everyone's has, and everyone is past compute(kt-1): slot (kt+2)%3 is free
    if (kt + 2 < nk) ISSUE(kt + 2);
    const char* sA = STA(kt % 3);
    const char* sB = STB(kt % 3);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      lpx8_t fa[5], fb[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = read_frag_k64(sB, wn * 32 + j * 16 + (lane & 15), s * 4 + (lane >> 4));
#pragma unroll
      for (int i = 0; i < 5; ++i) fa[i] = read_frag_k64(sA, wm * 80 + i * 16 + (lane & 15), s * 4 + (lane >> 4));
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = mfma_lp(fb[j], fa[i], acc[i][j]);
    }
    split_rescale(a, acc, (kt + 1) * BK);
  }
#undef STA
#undef STB
#undef ISSUE
  gemm_nt_epilogue_lds<5, 2>(a, acc, group, row0, row_end, n0, wm, wn, wave, lane, smem);
}

// ------------------------------------------------------------------------------------------
// TALL tile (round 6): 64 MI rows x 256 columns per workgroup, sixteen waves as 4 (M) x 4 (N), each 16 MI x 64 = acc[MI][4], in the
// hand-managed form of the persistent kernel above (SGPR buffer descriptors + loop-invariant lane offsets for the LDS-DMA, inline-asm
// fragment reads with counted lgkmcnt, a two-entry A-fragment ring).  MI = 5: ViT-B's
// N = 768 launches at 64 pairs per step (M = 26 944 = 25 664 | 1 280) are 85 x 3 = 255 tiles of 320 rows -- ONE round on 256 CUs
// with one prologue / epilogue per CU and 9 fragment reads per 20 MFMAs, instead of two rounds of 160-row tiles (7 per 10).  Round 4's
// 320-row attempt (tools/dev/gemm_320x256_r04.hip.txt) left reads, waits and the LDS-staged epilogue to the compiler and spilled at
// the 128-register budget of sixteen waves; here the main loop holds 80 accumulators + 4 B fragments + 2 A fragments, and the
// epilogues are written for the registers that are left:
//   EPI 0  16-bit output (+bias): 2 KiB of wave-private staging in the (finished) ring, row-wise 16-B stores;
//   EPI 1  fp32 output (+bias) (+ residual * row_scale): straight from the accumulator layout, 16 B per lane (a store covers
//          16 rows x 64 B), the residual rows of block i + 1 requested before block i is finished.
// The accumulators start at zero and the bias is added in the epilogue (as in the other one-tile kernels: bit-identical rows
// whichever kernel a batch size selects).  Split weights (hi + lo, `ka` < K): the accumulators are rescaled after the lo half.
// ------------------------------------------------------------------------------------------
template <int MI> struct TallGeo {
  static constexpr int ROWS = 64 * MI, WROWS = 16 * MI;
  static constexpr int A_BYTES = ROWS * BK * 2, STAGE = (ROWS + BNQ) * BK * 2;
  static constexpr int NPA = ROWS / 8;                 // 1 KiB pieces of an A k-tile (8 rows each)
  static constexpr int NAI = (NPA + 15) / 16;          // per wave: piece wave + 16 i while < NPA
  static constexpr int BIAS_OFF = 2 * STAGE;            // 16 waves x 64 floats of bias behind the ring
  static constexpr int SMEM = 2 * STAGE + 16 * 256;
};

// (a device function template behind plain kernels: the host pass instantiates a KERNEL template's body too, and target builtins /
// register constraints in it are a substitution failure there that silently drops the specialisation's host stub)
template <int MI, int EPI>
__device__ __forceinline__ void gemm_nt_tall_body(const GemmNTArgs& a, char* smem) {
  using G = TallGeo<MI>;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;   // 4 x 4 waves
  const int tiles_n = a.N / BNQ;
  const int tm0 = (a.split + G::ROWS - 1) / G::ROWS;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
  const int group = tile_m >= tm0;
  const int row0 = group ? a.split + (tile_m - tm0) * G::ROWS : tile_m * G::ROWS;
  const int row_end = group ? a.M : a.split;
  const int n0 = tile_n * BNQ;
  const int nk = a.K / BK;
  const bool two = a.ka < a.K;               // hi + lo weights

  // LDS-DMA: piece q holds rows 8 q .. 8 q + 7 of an operand's k-tile, lane -> row 8 q + (lane >> 3), 16-B slot (lane & 7) ^ (row & 7)
  // of the row (the XOR is on the SOURCE address: the LDS image is lane-linear); rows past the row group read as zeros (descriptor)
  const int din = lane >> 3, dslot = (lane & 7) ^ din;
  int voffA[G::NAI], voffW[2];
#pragma unroll
  for (int i = 0; i < G::NAI; ++i) voffA[i] = ((wave + 16 * i) * 8 + din) * a.lda * 2 + dslot * 16;
#pragma unroll
  for (int i = 0; i < 2; ++i) voffW[i] = ((wave * 2 + i) * 8 + din) * a.ldw * 2 + dslot * 16;
  const __amdgpu_buffer_rsrc_t dA =
      __builtin_amdgcn_make_buffer_rsrc((void*)(a.A + (long)row0 * a.lda), 0, (int)((long)(row_end - row0) * a.lda * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t dW =
      __builtin_amdgcn_make_buffer_rsrc((void*)(a.W + (long)group * a.w_gstride + (long)n0 * a.ldw), 0, BNQ * a.ldw * 2, 0x00020000);
  auto issue = [&](int kt, int st) {
    char* sA = smem + st * G::STAGE;
    const int ko = a_koff(a, kt * BK) * 2;
#pragma unroll
    for (int i = 0; i < G::NAI; ++i)
      if (wave + 16 * i < G::NPA)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(dA, LDS_PTR(sA + (wave + 16 * i) * 1024), 16, voffA[i], ko, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(dW, LDS_PTR(sA + G::A_BYTES + (wave * 2 + i) * 1024), 16, voffW[i], kt * (BK * 2), 0, 0);
  };
  // fragment reads: row (lane & 15) of a 16-row block, slot (4 s + (lane >> 4)) ^ (row & 7); 16 MI is a multiple of 8 for every MI
  // that is built (4, 5: 64, 80), so row & 7 == lane & 7
  static_assert(G::WROWS % 8 == 0, "wave row offset must keep the swizzle phase");
  const int c0 = (lane >> 4) ^ (lane & 7);
  const unsigned fA0 = lds_addr(smem) + (wm * G::WROWS + (lane & 15)) * 128 + c0 * 16, fA1 = fA0 ^ 64;
  const unsigned fB0 = lds_addr(smem) + G::A_BYTES + (wn * 64 + (lane & 15)) * 128 + c0 * 16, fB1 = fB0 ^ 64;

  issue(0, 0);
  if (nk > 1) issue(1, 1);
  // this wave's 64 bias values -> 256 B of LDS behind the ring: the epilogues read them 16 B at a time when they need them (held in
  // registers across the epilogue they cost 16 VGPRs next to the 80 accumulators)
  float* lbias = (float*)(smem + G::BIAS_OFF + wave * 256);
  if (lane < 16) {
    const float* bp = a.bias ? a.bias + (long)group * a.bias_gstride + n0 + wn * 64 + 4 * lane : nullptr;
    *(f32x4_t*)(lbias + 4 * lane) = bp ? *(const f32x4_t*)bp : (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }
  // the accumulators start at zero and the bias is added in the epilogue, exactly as the other non-persistent kernels do it
  // (fmaf(acc, 1, bias), then residual + row_scale * that): a row's result is bit-identical whichever kernel its batch size selects
  // (tests/test_properties_gpu.py: batch independence of forward_test)
  f32x4_t acc[MI][4];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      reg_pin(acc[i][j]);                    // opaque zeros: no peeled first k-tile (a second copy of the loop body, spills)
    }

#define TALL_STEP(I_)                                                                                        \
  if constexpr (MI > (I_)) {                                                                                 \
    if constexpr ((I_) + 1 < MI) lgkm_wait<1>(); else lgkm_wait<0>();                                        \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                            \
      acc[(I_)][j] = mfma_lp(as_frag(fb[j]), as_frag(fa[(I_) & 1]), acc[(I_)][j]);                           \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    if constexpr ((I_) + 2 < MI) fa[(I_) & 1] = lds_b128_asm<((I_) + 2) * 2048>(pa);                          \
  }
  const int k_lo = a.ka / BK;                // k-tiles of the lo half (== nk: single weights)
  for (int kt = 0; kt < nk; ++kt) {
    vm_wait<0>();                            // this wave's pieces of k-tile kt have landed
    __builtin_amdgcn_s_barrier();            // everyone's have, and everyone is past the MFMAs of kt - 1: its stage is free
    if (kt >= 1 && kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
    const unsigned so = (kt & 1) * G::STAGE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const unsigned pa = (s ? fA1 : fA0) + so, pb = (s ? fB1 : fB0) + so;
      u32x4_t fb[4], fa[2];
      fb[0] = lds_b128_asm<0>(pb); fb[1] = lds_b128_asm<2048>(pb); fb[2] = lds_b128_asm<4096>(pb); fb[3] = lds_b128_asm<6144>(pb);
      fa[0] = lds_b128_asm<0>(pa);
      if constexpr (MI > 1) fa[1] = lds_b128_asm<2048>(pa);
      TALL_STEP(0) TALL_STEP(1) TALL_STEP(2) TALL_STEP(3) TALL_STEP(4) TALL_STEP(5)
    }
    if (two && kt + 1 == k_lo) {             // wave-uniform, once per tile
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] *= a.lo_scale;
    }
  }
#undef TALL_STEP

  const float* bias = lbias + 4 * (lane >> 4);      // (wave-private LDS: written by this wave before its main loop)
  if constexpr (EPI == 0) {
    // 16 bit -> wave-private 2 KiB of the finished ring -> row-wise 16-B stores (the persistent kernel's epilogue)
    __builtin_amdgcn_s_barrier();            // every wave is done reading the ring
    const unsigned stg = lds_addr(smem) + wave * 2048;
    const int er = lane & 15, ecg = lane >> 4;
    const int rr = lane >> 3, rq = lane & 7;
    lp_t* cp = (lp_t*)a.C + (long)(row0 + wm * G::WROWS + rr) * a.ldc + n0 + wn * 64 + rq * 8;
    const int mleft = row_end - (row0 + wm * G::WROWS + rr);
    const unsigned wa = stg + er * 128, ra = stg + rr * 128 + ((rq ^ (rr >> 1)) << 4);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4_t x = acc[i][j] + *(const f32x4_t*)(bias + j * 16);
        lds_write_b64_asm(wa + (((j * 4 + ecg) ^ er) << 3), (u32x2_t){pack_lp2(x[0], x[1]), pack_lp2(x[2], x[3])});
      }
      lgkm_wait<0>();
      u32x4_t d0 = lds_b128_asm<0>(ra);
      u32x4_t d1 = lds_b128_asm<0>(ra ^ 1088);
      lgkm_wait<0>();
      if (rr & 1) { d0 = (u32x4_t){d0[2], d0[3], d0[0], d0[1]}; d1 = (u32x4_t){d1[2], d1[3], d1[0], d1[1]}; }
      if (i * 16 < mleft) *(u32x4_t*)(cp + (long)(i * 16) * a.ldc) = d0;
      if (i * 16 + 8 < mleft) *(u32x4_t*)(cp + (long)(i * 16 + 8) * a.ldc) = d1;
    }
  } else {
    // fp32 (+ residual * row_scale) straight from the accumulator layout: lane -> row (lane & 15) of block i, columns
    // j * 16 + 4 (lane >> 4) .. + 3 of the wave's 64 (a load / store covers 16 rows x 64 B).  Two passes over column halves (j = 0, 1 |
    // 2, 3) keep the live set at 80 accumulators + 2 bias + 2 x 2 residual vectors; the residual rows of block i + 1 are requested
    // before block i is finished.
    const int mb = row0 + wm * G::WROWS + (lane & 15);
    const int nb = n0 + wn * 64 + 4 * (lane >> 4);
    float* cp = (float*)a.C + (long)mb * a.ldc + nb;
    const float* rp = a.res ? a.res + (long)mb * a.ldres + nb : nullptr;
    float rsv[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = mb + i * 16;
      rsv[i] = (a.row_scale && m < row_end) ? a.row_scale[group ? (m - a.split) / a.rps1 : m / a.rps0] : 1.f;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4_t bh[2];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) bh[jj] = *(const f32x4_t*)(bias + (2 * h + jj) * 16);
      f32x4_t rv[2][2];
      auto load_res = [&](int i, f32x4_t (&r)[2]) {
        const bool ok = mb + i * 16 < row_end;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
          r[jj] = ok ? *(const f32x4_t*)(rp + (long)(i * 16) * a.ldres + (2 * h + jj) * 16) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
      };
      if (rp) load_res(0, rv[0]);
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        if (rp && i + 1 < MI) load_res(i + 1, rv[(i + 1) & 1]);
        if (mb + i * 16 < row_end) {
          const float rs = rsv[i];
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            f32x4_t v = acc[i][2 * h + jj] + bh[jj];
            if (rp) {
#pragma unroll
              for (int k = 0; k < 4; ++k) v[k] = __builtin_fmaf(rs, v[k], rv[i & 1][jj][k]);
            } else if (a.row_scale) {
              v *= rs;
            }
            *(f32x4_t*)(cp + (long)(i * 16) * a.ldc + (2 * h + jj) * 16) = v;
          }
        }
      }
    }
  }
}

__global__ __launch_bounds__(1024) void gemm_nt_kernel_tall5_lp(GemmNTArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  gemm_nt_tall_body<5, 0>(a, smem);
}
__global__ __launch_bounds__(1024) void gemm_nt_kernel_tall5_f32(GemmNTArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  gemm_nt_tall_body<5, 1>(a, smem);
}
// MI = 4 (256 rows, one round): ViT-L's N = 1024 launches at 32 pairs per step (M = 13 472: 54 x 4 = 216 tiles) whose epilogue the
// persistent kernel does not take -- the fp32 + residual forwards of out-proj / fc2, which ran on the compiler-scheduled 224-row kernel
__global__ __launch_bounds__(1024) void gemm_nt_kernel_tall4_f32(GemmNTArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  gemm_nt_tall_body<4, 1>(a, smem);
}

// ------------------------------------------------------------------------------------------
// The same hand-managed form with the sixteen waves as 2 (M) x 8 (N), each 16 MI x 32 = acc[MI][2] (round 6): MI = 7 -> 224 x 256
// tiles, the row extent of ViT-L at 32 pairs per step (M = 13 472 = 12 832 | 640: 61 row tiles, 95 % of the CUs per round where
// 256-row tiles leave 16 % idle) that `gemm_nt_kernel_224x256_w16` serves with compiler-scheduled reads.  All nine fragments of a
// k-half are requested up front (56 accumulators leave the registers for it) and consumed behind counted lgkmcnt waits.
// EPI 0: 16-bit output through 1 KiB of wave-private staging per 16-row block (a store covers 16 rows x 64 B); EPI 1: fp32
// (+ residual * row_scale) straight from the accumulator layout.  Bias in the epilogue, zero start (see gemm_nt_tall_body).
// ------------------------------------------------------------------------------------------
template <int MI> struct Tall28Geo {
  static constexpr int ROWS = 32 * MI, WROWS = 16 * MI;
  static constexpr int A_BYTES = ROWS * BK * 2, STAGE = (ROWS + BNQ) * BK * 2;
  static constexpr int NPA = ROWS / 8, NAI = (NPA + 15) / 16;
  static constexpr int BIAS_OFF = 2 * STAGE;
  static constexpr int SMEM = 2 * STAGE + 16 * 128;       // + 32 bias floats per wave
};

template <int MI, int EPI>
__device__ __forceinline__ void gemm_nt_tall28_body(const GemmNTArgs& a, char* smem) {
  using G = Tall28Geo<MI>;
  static_assert(G::WROWS % 8 == 0, "wave row offset must keep the swizzle phase");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 3, wn = wave & 7;   // 2 x 8 waves
  const int tiles_n = a.N / BNQ;
  const int tm0 = (a.split + G::ROWS - 1) / G::ROWS;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  int tile_m, tile_n;
  tile_order(bid, (int)gridDim.x / tiles_n, tiles_n, a.gn, tile_m, tile_n);
  const int group = tile_m >= tm0;
  const int row0 = group ? a.split + (tile_m - tm0) * G::ROWS : tile_m * G::ROWS;
  const int row_end = group ? a.M : a.split;
  const int n0 = tile_n * BNQ;
  const int nk = a.K / BK;
  const bool two = a.ka < a.K;
  const int din = lane >> 3, dslot = (lane & 7) ^ din;
  int voffA[G::NAI], voffW[2];
#pragma unroll
  for (int i = 0; i < G::NAI; ++i) voffA[i] = ((wave + 16 * i) * 8 + din) * a.lda * 2 + dslot * 16;
#pragma unroll
  for (int i = 0; i < 2; ++i) voffW[i] = ((wave * 2 + i) * 8 + din) * a.ldw * 2 + dslot * 16;
  const __amdgpu_buffer_rsrc_t dA =
      __builtin_amdgcn_make_buffer_rsrc((void*)(a.A + (long)row0 * a.lda), 0, (int)((long)(row_end - row0) * a.lda * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t dW =
      __builtin_amdgcn_make_buffer_rsrc((void*)(a.W + (long)group * a.w_gstride + (long)n0 * a.ldw), 0, BNQ * a.ldw * 2, 0x00020000);
  auto issue = [&](int kt, int st) {
    char* sA = smem + st * G::STAGE;
    const int ko = a_koff(a, kt * BK) * 2;
#pragma unroll
    for (int i = 0; i < G::NAI; ++i)
      if (wave + 16 * i < G::NPA)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(dA, LDS_PTR(sA + (wave + 16 * i) * 1024), 16, voffA[i], ko, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(dW, LDS_PTR(sA + G::A_BYTES + (wave * 2 + i) * 1024), 16, voffW[i], kt * (BK * 2), 0, 0);
  };
  const int c0 = (lane >> 4) ^ (lane & 7);
  const unsigned fA0 = lds_addr(smem) + (wm * G::WROWS + (lane & 15)) * 128 + c0 * 16, fA1 = fA0 ^ 64;
  const unsigned fB0 = lds_addr(smem) + G::A_BYTES + (wn * 32 + (lane & 15)) * 128 + c0 * 16, fB1 = fB0 ^ 64;

  issue(0, 0);
  if (nk > 1) issue(1, 1);
  float* lbias = (float*)(smem + G::BIAS_OFF + wave * 128);
  if (lane < 8) {
    const float* bp = a.bias ? a.bias + (long)group * a.bias_gstride + n0 + wn * 32 + 4 * lane : nullptr;
    *(f32x4_t*)(lbias + 4 * lane) = bp ? *(const f32x4_t*)bp : (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }
  f32x4_t acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      reg_pin(acc[i][j]);
    }
#define T28_READ(I_) if constexpr (MI > (I_)) fa[(I_)] = lds_b128_asm<(I_) * 2048>(pa);
#define T28_STEP(I_)                                                                                   \
  if constexpr (MI > (I_)) {                                                                           \
    lgkm_wait<MI - 1 - (I_)>();                                                                        \
    acc[(I_)][0] = mfma_lp(as_frag(fb[0]), as_frag(fa[(I_)]), acc[(I_)][0]);                           \
    acc[(I_)][1] = mfma_lp(as_frag(fb[1]), as_frag(fa[(I_)]), acc[(I_)][1]);                           \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
  }
  const int k_lo = a.ka / BK;
  for (int kt = 0; kt < nk; ++kt) {
    vm_wait<0>();
    __builtin_amdgcn_s_barrier();
    if (kt >= 1 && kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
    const unsigned so = (kt & 1) * G::STAGE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const unsigned pa = (s ? fA1 : fA0) + so, pb = (s ? fB1 : fB0) + so;
      u32x4_t fb[2], fa[MI];
      fb[0] = lds_b128_asm<0>(pb); fb[1] = lds_b128_asm<2048>(pb);
      T28_READ(0) T28_READ(1) T28_READ(2) T28_READ(3) T28_READ(4) T28_READ(5) T28_READ(6) T28_READ(7)
      T28_STEP(0) T28_STEP(1) T28_STEP(2) T28_STEP(3) T28_STEP(4) T28_STEP(5) T28_STEP(6) T28_STEP(7)
    }
    if (two && kt + 1 == k_lo) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] *= a.lo_scale;
    }
  }
#undef T28_READ
#undef T28_STEP
  const float* bias = lbias + 4 * (lane >> 4);
  if constexpr (EPI == 0) {
    __builtin_amdgcn_s_barrier();            // every wave is done reading the ring
    const unsigned stg = lds_addr(smem) + wave * 1024;
    const int er = lane & 15, ecg = lane >> 4;         // accumulator layout: row er, columns 4 ecg .. + 3 of each 16 x 16 block
    const int rr = lane >> 2, rq = lane & 3;           // read-out layout: row rr, 16-B chunk rq of the wave's 64 B
    lp_t* cp = (lp_t*)a.C + (long)(row0 + wm * G::WROWS + rr) * a.ldc + n0 + wn * 32 + rq * 8;
    const int mleft = row_end - (row0 + wm * G::WROWS + rr);
    // staged row = 64 B; 8-B chunk c (= 4 j + ecg) of row er at chunk c ^ (er & 7): the 16 rows of a write hit distinct bank pairs;
    // the 16-B chunk rq of row rr is then chunks 2 rq and 2 rq + 1 XOR (rr & 7), i.e. 16-B chunk rq ^ ((rr & 7) >> 1), halves swapped
    // for odd rr
    const unsigned wa = stg + er * 64, ra = stg + rr * 64 + ((rq ^ ((rr & 7) >> 1)) << 4);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const f32x4_t x = acc[i][j] + *(const f32x4_t*)(bias + j * 16);
        lds_write_b64_asm(wa + (((j * 4 + ecg) ^ (er & 7)) << 3), (u32x2_t){pack_lp2(x[0], x[1]), pack_lp2(x[2], x[3])});
      }
      lgkm_wait<0>();
      u32x4_t d0 = lds_b128_asm<0>(ra);
      lgkm_wait<0>();
      if (rr & 1) d0 = (u32x4_t){d0[2], d0[3], d0[0], d0[1]};
      if (i * 16 < mleft) *(u32x4_t*)(cp + (long)(i * 16) * a.ldc) = d0;
    }
  } else {
    const int mb = row0 + wm * G::WROWS + (lane & 15);
    const int nb = n0 + wn * 32 + 4 * (lane >> 4);
    float* cp = (float*)a.C + (long)mb * a.ldc + nb;
    const float* rp = a.res ? a.res + (long)mb * a.ldres + nb : nullptr;
    f32x4_t bh[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) bh[j] = *(const f32x4_t*)(bias + j * 16);
    f32x4_t rv[2][2];
    auto load_res = [&](int i, f32x4_t (&r)[2]) {
      const bool ok = mb + i * 16 < row_end;
#pragma unroll
      for (int j = 0; j < 2; ++j) r[j] = ok ? *(const f32x4_t*)(rp + (long)(i * 16) * a.ldres + j * 16) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
    };
    if (rp) load_res(0, rv[0]);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      if (rp && i + 1 < MI) load_res(i + 1, rv[(i + 1) & 1]);
      const int m = mb + i * 16;
      if (m < row_end) {
        const float rs = a.row_scale ? a.row_scale[group ? (m - a.split) / a.rps1 : m / a.rps0] : 1.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x4_t v = acc[i][j] + bh[j];
          if (rp) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = __builtin_fmaf(rs, v[k], rv[i & 1][j][k]);
          } else if (a.row_scale) {
            v *= rs;
          }
          *(f32x4_t*)(cp + (long)(i * 16) * a.ldc + j * 16) = v;
        }
      }
    }
  }
}
__global__ __launch_bounds__(1024) void gemm_nt_kernel_t224_lp(GemmNTArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  gemm_nt_tall28_body<7, 0>(a, smem);
}
__global__ __launch_bounds__(1024) void gemm_nt_kernel_t224_f32(GemmNTArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  gemm_nt_tall28_body<7, 1>(a, smem);
}

// 224x256x64 tile with sixteen waves: 2 (M) x 8 (N), each 112x32 = acc[7][2] (round 4).  For row counts where 256-row tiles
// leave CUs idle: ViT-L at 32 pairs per step has M = 13 472 = 52.6 x 256 -- 53 x 4 = 212 tiles of 256 x 256 for the N = 1024
// launches (out-proj, fc2, three dgrads per layer) on 256 CUs, 61 x 4 = 244 tiles of 224 rows fill 95 % of them in one round of
// 7/8 the length.  Two 60 KiB LDS buffers (three do not fit), one rendezvous per k-tile like the 256x256 kernel; 9 fragment
// reads per 14 MFMAs.
__global__ __launch_bounds__(1024) void gemm_nt_kernel_224x256_w16(GemmNTArgs a) {
  constexpr int BMQ = 224;
  constexpr int STAGEQ = (BMQ + BNQ) * BK * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 3, wn = wave & 7;
  const int tiles_n = (a.N + BNQ - 1) / BNQ;
  const int tm0 = (a.split + BMQ - 1) / BMQ;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  int tile_m, tile_n;
  tile_order(bid, (int)gridDim.x / tiles_n, tiles_n, a.gn, tile_m, tile_n);
  const int group = tile_m >= tm0;
  const int row0 = group ? a.split + (tile_m - tm0) * BMQ : tile_m * BMQ;
  const int row_end = group ? a.M : a.split;
  const int n0 = tile_n * BNQ;
  const lp_t* W = a.W + (long)group * a.w_gstride;
  f32x4_t acc[7][2];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int nk = a.K / BK;
#define STA(s_) (smem + (s_) * STAGEQ)
#define STB(s_) (smem + (s_) * STAGEQ + BMQ * BK * 2)
#define ISSUE(t_)                                                                                      \
  do {                                                                                                 \
    const int st__ = (t_) & 1;                                                                         \
    stage_rows_k64(a.A, a.lda, row0, row_end - 1, a_koff(a, (t_) * BK), STA(st__), wave, lane, BMQ / 8, 16); \
    stage_rows_k64(W, a.ldw, n0, a.N - 1, (t_) * BK, STB(st__), wave, lane, BNQ / 8, 16);              \
  } while (0)
  ISSUE(0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's share of tile kt has landed
    __builtin_amdgcn_s_barrier();                        // everyone's has, and everyone is past compute(kt-1)
    if (kt + 1 < nk) ISSUE(kt + 1);
    const char* sA = STA(kt & 1);
    const char* sB = STB(kt & 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      lpx8_t fa[7], fb[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = read_frag_k64(sB, wn * 32 + j * 16 + (lane & 15), s * 4 + (lane >> 4));
#pragma unroll
      for (int i = 0; i < 7; ++i) fa[i] = read_frag_k64(sA, wm * 112 + i * 16 + (lane & 15), s * 4 + (lane >> 4));
#pragma unroll
      for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = mfma_lp(fb[j], fa[i], acc[i][j]);
    }
    split_rescale(a, acc, (kt + 1) * BK);
  }
#undef STA
#undef STB
#undef ISSUE
  gemm_nt_epilogue_lds<7, 2>(a, acc, group, row0, row_end, n0, wm, wn, wave, lane, smem);
}

// ------------------------------------------------------------------------------------------
// Variant: same 256x128 tile / 8 waves / 3-stage ring, but K-tiles of 32 (24 KiB stages, 72 KiB LDS) so that TWO
// workgroups are resident per CU (16 waves, 4 per SIMD): while one workgroup sits in its wait/barrier the other
// one owns the MFMA pipe.  LDS rows are 64 B (4 slots of 16 B); slot ^= (row >> 2) & 3 keeps ds_read_b128 of 16
// consecutive rows conflict-free.
// ------------------------------------------------------------------------------------------
constexpr int BK3 = 32;
constexpr int STAGE3 = (BM2 + BN) * BK3 * 2;   // 24576 B

__device__ __forceinline__ void stage_tile_k32(const lp_t* base, int ld, int row0, int row_last, int k0, char* lds,
                                               int wave, int lane, int per_wave) {
  for (int i = 0; i < per_wave; ++i) {
    const int inst = wave * per_wave + i;
    const int r = inst * 16 + (lane >> 2);
    int row = row0 + r;
    row = row < row_last ? row : row_last;
    const int lslot = (lane & 3) ^ ((r >> 2) & 3);
    const lp_t* src = base + (long)row * ld + k0 + lslot * 8;
    __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(lds + inst * 1024), 16, 0, 0);
  }
}
__device__ __forceinline__ lpx8_t read_frag_k32(const char* lds, int row, int lslot) {
  return *(const lpx8_t*)(lds + row * 64 + ((lslot ^ ((row >> 2) & 3)) << 4));
}

__global__ __launch_bounds__(512, 2) void gemm_nt_kernel_256k32(GemmNTArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (a.N + BN - 1) / BN;
  const int tm0 = (a.split + BM2 - 1) / BM2;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
  const int group = tile_m >= tm0;
  const int row0 = group ? a.split + (tile_m - tm0) * BM2 : tile_m * BM2;
  const int row_end = group ? a.M : a.split;
  const int n0 = tile_n * BN;
  const lp_t* W = a.W + (long)group * a.w_gstride;
  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int nk = a.K / BK3;
#define STA(s_) (smem + (s_) * STAGE3)
#define STB(s_) (smem + (s_) * STAGE3 + BM2 * BK3 * 2)
#define ISSUE(t_)                                                                         \
  do {                                                                                    \
    const int st__ = (t_) % 3;                                                            \
    stage_tile_k32(a.A, a.lda, row0, row_end - 1, a_koff(a, (t_) * BK3), STA(st__), wave, lane, 2); \
    stage_tile_k32(W, a.ldw, n0, a.N - 1, (t_) * BK3, STB(st__), wave, lane, 1);          \
  } while (0)
  ISSUE(0);
  if (nk > 1) ISSUE(1);
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 2 < nk) ISSUE(kt + 2);
    const char* sA = STA(kt % 3);
    const char* sB = STB(kt % 3);
    lpx8_t fa[4], fb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      fa[i] = read_frag_k32(sA, wm * 64 + i * 16 + (lane & 15), lane >> 4);
      fb[i] = read_frag_k32(sB, wn * 64 + i * 16 + (lane & 15), lane >> 4);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = mfma_lp(fb[j], fa[i], acc[i][j]);
    split_rescale(a, acc, (kt + 1) * BK3);
  }
#undef STA
#undef STB
#undef ISSUE
  if ((a.N & 7) == 0) { gemm_nt_epilogue_lds<4>(a, acc, group, row0, row_end, n0, wm, wn, wave, lane, smem); return; }
  gemm_nt_epilogue<4>(a, acc, group, row0, row_end, n0, wm, wn, lane);
}


// ------------------------------------------------------------------------------------------
// wgrad: dW[g][n][k] += sum_m dY[m][n] * X[m][k]
// ------------------------------------------------------------------------------------------
struct GemmTNArgs {
  const lp_t* dY; int lddy;
  const lp_t* X; int ldx;
  float* dW; long dw_gstride; int lddw;
  int M, N, K, split, rows_per_chunk, chunks0;
  float* db; int db_gstride;     // optional bias gradient db[g][n] += column sums of dY over group g (extra blocks)
  int tiles;                     // GEMM tiles along blockIdx.x; blocks beyond them are the column-sum blocks
  int cs_split;                  // row slices per chunk for the column-sum blocks
  int gk;                        // group width of the tile walk along the faster dimension (0: plain)
  float out_scale;               // scalar on everything this launch adds to dW / db (1 / gradient scale of dY)
};

__device__ __attribute__((aligned(16))) unsigned int g_zero_page[64];  // 256 B of zeros

// XOR swizzle of the 16-B slots of a staged row (rows are a multiple of 256 B, i.e. bank-aligned).  A transposed
// fragment read (`ds_read_b64_tr_b16`) is served 32 lanes per LDS cycle: lanes 0-15 take rows r..r+3, lanes 16-31 rows
// r+8..r+11, each 16 columns wide.  (r & 3) << 1 spreads the 4 rows of a 16-lane group over 8 slots = 32 banks; bit 3
// of the row moves the second group to the OTHER 32 banks -- without it the two groups met on the same banks and every
// LDS access of the wgrad kernels was a 2-way conflict (PMC: SQ_LDS_BANK_CONFLICT = 50 % of SQ_LDS_IDX_ACTIVE).
__device__ __forceinline__ int tn_swz(int r) { return ((r & 3) << 1) ^ (((r >> 3) & 1) << 3); }

// stage [64 m-rows][128 cols] bf16 (256 B rows): 16 wave-instructions (4 rows each), 4 per wave
__device__ __forceinline__ void stage_tile_m64(const lp_t* base, int ld, int m0, int m_end, int c0, int ncols,
                                               char* lds, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int inst = wave * 4 + i;
    const int r = inst * 4 + (lane >> 4);
    const int lslot = (lane & 15) ^ tn_swz(r);
    const int row = m0 + r, col = c0 + lslot * 8;
    const lp_t* src = (row < m_end && col < ncols) ? base + (long)row * ld + col
                                                      : (const lp_t*)g_zero_page;
    __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(lds + inst * 1024), 16, 0, 0);
  }
}

// transposed fragment: lane (i = lane&15 -> column c0+i, g = lane>>4 -> rows ms+8g..+7); raw halves by inline-asm reads
// (common.h: a builtin read behind the LDS-DMA of the next stage would make hipcc drain the ring) -- the caller waits
// (lds_wait_all), pins (lds_pin) and packs (frag8)
__device__ __forceinline__ void read_frag_tr(const char* lds, int ms, int c0, int lane, u32x2_t (&out)[2]) {
  const int i = lane & 15, g = lane >> 4;
  const int col = c0 + 4 * (i & 3);          // this lane supplies 4 contiguous columns of row (i>>2)
  const int lslot = col >> 3, within = (col & 7) * 2;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = ms + 8 * g + 4 * h + (i >> 2);
    out[h] = lds_tr16_asm<0>(lds_addr(lds + row * 256 + ((lslot ^ tn_swz(row)) << 4) + within));
  }
}

// Bias gradient as a horizontal fusion: blocks with blockIdx.x >= a.tiles of the SAME launch each sum 256 columns of dY
// over their row chunk (pure streaming, no MFMA), so the 49 per-step column-sum launches disappear and their HBM
// reads overlap the MFMA blocks.  (Doing it inside the GEMM blocks -- an extra MFMA against a ones fragment on the
// tile_k == 0 blocks -- made those blocks the tail of every launch: wgrad 9 -> 12 ms/step.)
template <int NT>
__device__ __forceinline__ void colsum_block(const GemmTNArgs& a, char* smem, int cb, int group, int m_begin, int m_end) {
  // cb = column block (256 columns) * cs_split + row slice: a GEMM chunk's rows are cut into cs_split slices so that a
  // column-sum block never walks more than ~512 rows (one long serial walk per chunk made these blocks the tail of
  // the out-proj wgrad: 157 us in situ vs 82 us without them); 4 independent 16-B loads in flight per thread
  constexpr int RY = NT / 32;
  float (*red)[256 + 8] = (float (*)[256 + 8])smem;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int col_block = cb / a.cs_split, slice = cb - col_block * a.cs_split;
  const int rows = (m_end - m_begin + a.cs_split - 1) / a.cs_split;
  const int r0 = m_begin + slice * rows, r1 = min(r0 + rows, m_end);
  const int n = col_block * 256 + tx * 8;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (n < a.N) {
    int m = r0 + ty;
    for (; m + 3 * RY < r1; m += 4 * RY) {
      u32x4_t v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = *(const u32x4_t*)(a.dY + (long)(m + q * RY) * a.lddy + n);
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float lo, hi;
          unpack_lp2(v[q][e], lo, hi);
          s[2 * e] += lo;
          s[2 * e + 1] += hi;
        }
    }
    for (; m < r1; m += RY) {
      const u32x4_t v = *(const u32x4_t*)(a.dY + (long)m * a.lddy + n);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float lo, hi;
        unpack_lp2(v[e], lo, hi);
        s[2 * e] += lo;
        s[2 * e + 1] += hi;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) red[ty][tx * 8 + q] = s[q];
  __syncthreads();
  const int c = threadIdx.x;
  if (c < 256) {
    float t = 0.f;
#pragma unroll
    for (int y = 0; y < RY; ++y) t += red[y][c];
    const int nn = col_block * 256 + c;
    if (nn < a.N && r0 < r1) atomicAdd(a.db + (long)group * a.db_gstride + nn, t * a.out_scale);
  }
}

__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTNArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int tiles_k = (a.K + 127) / 128, tiles_n = (a.N + 127) / 128;
  // XCD-aware: each XCD (blockIdx.x % 8) owns a contiguous slab of output tiles along the LARGER of N / K, so its L2
  // only has to hold 1/8 of the bigger operand's column panels (PMC before: 3.5x the algorithmic bytes fetched)
  const int chunk = blockIdx.y;
  const int group = chunk >= a.chunks0;
  const int m_begin = group ? a.split + (chunk - a.chunks0) * a.rows_per_chunk : chunk * a.rows_per_chunk;
  const int g_end = group ? a.M : a.split;
  const int m_end = min(m_begin + a.rows_per_chunk, g_end);
  if (m_begin >= m_end) return;
  if ((int)blockIdx.x >= a.tiles) { colsum_block<256>(a, smem, blockIdx.x - a.tiles, group, m_begin, m_end); return; }
  const int wg = xcd_remap(blockIdx.x, a.tiles);
  int tile_n, tile_k;
  if (tiles_n >= tiles_k) { tile_n = wg / tiles_k; tile_k = wg - tile_n * tiles_k; }
  else { tile_k = wg / tiles_n; tile_n = wg - tile_k * tiles_n; }
  const int n0 = tile_n * 128, k0 = tile_k * 128;

#define ldsY(c) (smem + (c) * 32768)
#define ldsX(c) (smem + 16384 + (c) * 32768)
  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int nt = (m_end - m_begin + 63) / 64;
  stage_tile_m64(a.dY, a.lddy, m_begin, m_end, n0, a.N, ldsY(0), wave, lane);
  stage_tile_m64(a.X, a.ldx, m_begin, m_end, k0, a.K, ldsX(0), wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    if (t + 1 < nt) {
      stage_tile_m64(a.dY, a.lddy, m_begin + (t + 1) * 64, m_end, n0, a.N, ldsY(cur ^ 1), wave, lane);
      stage_tile_m64(a.X, a.ldx, m_begin + (t + 1) * 64, m_end, k0, a.K, ldsX(cur ^ 1), wave, lane);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      u32x2_t ry[4][2], rx[4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        read_frag_tr(ldsY(cur), s * 32, wn * 64 + i * 16, lane, ry[i]);
        read_frag_tr(ldsX(cur), s * 32, wk * 64 + i * 16, lane, rx[i]);
      }
      lds_wait_all();
      lpx8_t fy[4], fx[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        lds_pin(ry[i][0], ry[i][1]);
        lds_pin(rx[i][0], rx[i][1]);
        fy[i] = frag8(ry[i][0], ry[i][1]);
        fx[i] = frag8(rx[i][0], rx[i][1]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = mfma_lp(fy[i], fx[j], acc[i][j]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  float* dW = a.dW + (long)group * a.dw_gstride;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + wk * 64 + j * 16 + (lane & 15);
      if (k >= a.K) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * 64 + i * 16 + 4 * (lane >> 4) + r;
        if (n < a.N) atomicAdd(dW + (long)n * a.lddw + k, acc[i][j][r] * a.out_scale);
      }
    }
}

// wgrad, 256 (n) x 128 (k) output tile: 8 waves as 4 (n) x 2 (k), each 64x64; 32-row stages {dY [32][256], X [32][128]}
// = 24 KiB, 3-stage global_load_lds ring with counted vmcnt + raw barrier, 72 KiB LDS -> two workgroups per CU.
// 87 FLOP per staged byte instead of 65 (the kernel is load-bound: removing the loads saves 40 %).
__device__ __forceinline__ void stage_rows32(const lp_t* base, int ld, int m0, int m_end, int c0, int ncols, int width,
                                             char* lds, int wave, int lane, int per_wave) {
  // width = tile columns (256 or 128); a 1 KiB wave-instruction covers 1024 / (2*width) rows
  const int lanes_per_row = width / 8, rows_per_inst = 64 / lanes_per_row;
  for (int i = 0; i < per_wave; ++i) {
    const int inst = wave * per_wave + i;
    const int r = inst * rows_per_inst + lane / lanes_per_row;
    const int pslot = lane % lanes_per_row;
    const int lslot = pslot ^ tn_swz(r);
    const int row = m0 + r, col = c0 + lslot * 8;
    const lp_t* src = (row < m_end && col < ncols) ? base + (long)row * ld + col : (const lp_t*)g_zero_page;
    __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(lds + inst * 1024), 16, 0, 0);
  }
}
// transposed fragment from a [32 rows][width cols] stage: lane (i = lane&15 -> column c0+i, g = lane>>4 -> rows 8g..8g+7)
__device__ __forceinline__ void read_frag_tr_w(const char* lds, int row_bytes, int c0, int lane, u32x2_t (&out)[2]) {
  const int i = lane & 15, g = lane >> 4;
  const int col = c0 + 4 * (i & 3);
  const int lslot = col >> 3, within = (col & 7) * 2;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = 8 * g + 4 * h + (i >> 2);
    out[h] = lds_tr16_asm<0>(lds_addr(lds + row * row_bytes + ((lslot ^ tn_swz(row)) << 4) + within));
  }
}

__global__ __launch_bounds__(512, 2) void gemm_tn_kernel_256(GemmTNArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 1, wk = wave & 1;
  const int tiles_k = (a.K + 127) / 128, tiles_n = (a.N + 255) / 256;
  const int chunk = blockIdx.y;
  const int group = chunk >= a.chunks0;
  const int m_begin = group ? a.split + (chunk - a.chunks0) * a.rows_per_chunk : chunk * a.rows_per_chunk;
  const int g_end = group ? a.M : a.split;
  const int m_end = min(m_begin + a.rows_per_chunk, g_end);
  if (m_begin >= m_end) return;
  if ((int)blockIdx.x >= a.tiles) { colsum_block<512>(a, smem, blockIdx.x - a.tiles, group, m_begin, m_end); return; }
  const int wg = xcd_remap(blockIdx.x, a.tiles);
  int tile_n, tile_k;
  // an XCD gets ~9 consecutive tiles per row chunk: walk the faster dimension in groups of 3 so that they form a 3 x 3
  // block (3 dY + 3 X panels) instead of 1.5 x 6 (2 dY + 6 X panels; fc1 wgrad: 12 x 6 tiles)
  if (tiles_n >= tiles_k) tile_order(wg, tiles_n, tiles_k, a.gk, tile_n, tile_k);
  else tile_order(wg, tiles_k, tiles_n, a.gk, tile_k, tile_n);
  const int n0 = tile_n * 256, k0 = tile_k * 128;
  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int nt = (m_end - m_begin + 31) / 32;
  constexpr int ST = 24576;
#define SY(s_) (smem + (s_) * ST)
#define SX(s_) (smem + (s_) * ST + 16384)
#define ISSUE(t_)                                                                                          \
  do {                                                                                                     \
    const int st__ = (t_) % 3;                                                                             \
    stage_rows32(a.dY, a.lddy, m_begin + (t_) * 32, m_end, n0, a.N, 256, SY(st__), wave, lane, 2);         \
    stage_rows32(a.X, a.ldx, m_begin + (t_) * 32, m_end, k0, a.K, 128, SX(st__), wave, lane, 1);           \
  } while (0)
  ISSUE(0);
  if (nt > 1) ISSUE(1);
  for (int t = 0; t < nt; ++t) {
    if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + 2 < nt) ISSUE(t + 2);
    const char* sy = SY(t % 3);
    const char* sx = SX(t % 3);
    u32x2_t ry[4][2], rx[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      read_frag_tr_w(sy, 512, wn * 64 + i * 16, lane, ry[i]);
      read_frag_tr_w(sx, 256, wk * 64 + i * 16, lane, rx[i]);
    }
    lds_wait_all();
    lpx8_t fy[4], fx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      lds_pin(ry[i][0], ry[i][1]);
      lds_pin(rx[i][0], rx[i][1]);
      fy[i] = frag8(ry[i][0], ry[i][1]);
      fx[i] = frag8(rx[i][0], rx[i][1]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = mfma_lp(fy[i], fx[j], acc[i][j]);
  }
#undef SY
#undef SX
#undef ISSUE
  float* dW = a.dW + (long)group * a.dw_gstride;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + wk * 64 + j * 16 + (lane & 15);
      if (k >= a.K) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * 64 + i * 16 + 4 * (lane >> 4) + r;
        if (n < a.N) atomicAdd(dW + (long)n * a.lddw + k, acc[i][j][r] * a.out_scale);
      }
    }
}

// column sums by row group: out[g][n] += sum_{m in group g} Y[m][n]   (bias gradients)
__global__ __launch_bounds__(256) void colsum_kernel(const lp_t* Y, int ldy, float* out, int out_gstride,
                                                     int M, int N, int split, int rows_per_chunk, int chunks0) {
  __shared__ float red[8][256 + 8];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int chunk = blockIdx.y;
  const int group = chunk >= chunks0;
  const int m_begin = group ? split + (chunk - chunks0) * rows_per_chunk : chunk * rows_per_chunk;
  const int m_end = min(m_begin + rows_per_chunk, group ? M : split);
  const int n = blockIdx.x * 256 + tx * 8;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (n < N) {
    for (int m = m_begin + ty; m < m_end; m += 8) {
      const u32x4_t v = *(const u32x4_t*)(Y + (long)m * ldy + n);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float lo, hi;
        unpack_lp2(v[q], lo, hi);
        s[2 * q] += lo;
        s[2 * q + 1] += hi;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) red[ty][tx * 8 + q] = s[q];
  __syncthreads();
  const int c = threadIdx.x;
  float t = 0.f;
#pragma unroll
  for (int y = 0; y < 8; ++y) t += red[y][c];
  const int nn = blockIdx.x * 256 + c;
  if (nn < N && m_begin < m_end) atomicAdd(out + (long)group * out_gstride + nn, t);
}

}  // namespace

// the persistent 256x256 kernel: 16-bit output, bias-only epilogue, an even number of k-tiles
static bool persist_ok(const GemmNTArgs& a) {
  static const bool off = getenv("SIMVG_GEMM_PERSIST") && atoi(getenv("SIMVG_GEMM_PERSIST")) == 0;
  // the kernel's counted vmcnt waits assume that its ONLY vector-memory operations are the ones written in the source: a build
  // that spills (scratch loads / stores share the FIFO) would consume k-tile 0 or the bias before they land.  Refuse such a build
  // (the non-persistent 256x256 kernel takes over), as the streamed attention launcher does.
  // (a host without a device -- simvg_gemm_nt_plan in the CPU tests -- cannot run anything: the build is taken as clean there)
  static const bool no_device = [] { int n = 0; return hipGetDeviceCount(&n) != hipSuccess || n == 0; }();
  static const bool no_scratch = no_device || [] {
    hipFuncAttributes fa{};
    return hipFuncGetAttributes(&fa, (const void*)gemm_nt_kernel_256sq_p) == hipSuccess && fa.localSizeBytes == 0;
  }();
  static const bool no_scratch2 = no_device || [] {
    hipFuncAttributes fa{};
    return hipFuncGetAttributes(&fa, (const void*)gemm_nt_kernel_256sq_p2) == hipSuccess && fa.localSizeBytes == 0;
  }();
  if (a.ka < a.K && (2 * a.ka != a.K || (a.K / BK) % 4 != 0)) return false;     // split weights: two halves of an even number of k-tiles
  return !off && (a.ka < a.K ? no_scratch2 : no_scratch) && !a.c_f32 && !a.aux && !a.res && !a.row_scale && a.act == 0 && a.alpha == 1.f && (a.K / BK) % 2 == 0 && a.ldc % 8 == 0;
}

static int gemm_nt_launch(const void* A, int lda, const void* W, long w_gstride, int ldw,
                          const float* bias, int bias_gstride, void* C, int ldc, int c_is_f32,
                          void* aux_preact, int ldaux, const float* residual, int ldres,
                          const float* row_scale, int rows_per_sample0, int rows_per_sample1,
                          int M, int N, int K, int split, int act, float alpha, int ka, float lo_scale, hipStream_t stream,
                          int* plan = nullptr);

extern "C" int simvg_gemm_nt(const void* A, int lda, const void* W, long w_gstride, int ldw,
                             const float* bias, int bias_gstride, void* C, int ldc, int c_is_f32,
                             void* aux_preact, int ldaux, const float* residual, int ldres,
                             const float* row_scale, int rows_per_sample0, int rows_per_sample1,
                             int M, int N, int K, int split, int act, float alpha, hipStream_t stream) {
  return gemm_nt_launch(A, lda, W, w_gstride, ldw, bias, bias_gstride, C, ldc, c_is_f32, aux_preact, ldaux, residual, ldres,
                        row_scale, rows_per_sample0, rows_per_sample1, M, N, K, split, act, alpha, K, 1.f, stream);
}

// C = A[M, K] . (W_hi + W_lo)[g][N, K]^T with the weight held as TWO 16-bit numbers per entry: W rows are 2 K long,
// [lo * 2^s | hi] (lo = the 16-bit rounding of (w - hi) * 2^s, hi = the 16-bit rounding of w), `lo_scale` = 2^-s.  The k loop
// runs over 2 K with A walked twice; the accumulators are multiplied by lo_scale between the halves.  Same epilogues as
// simvg_gemm_nt.  Used by the encoder's inference forward (`BEIT3.precise_inference`): rounding the WEIGHTS to 16 bits is the
// largest single term of the box error on trained-scale weights (tools/dev/token_tail.py), and forward_test has MFMA time to spare.
extern "C" int simvg_gemm_nt_split(const void* A, int lda, const void* W2, long w_gstride, int ldw,
                                   const float* bias, int bias_gstride, void* C, int ldc, int c_is_f32,
                                   const float* residual, int ldres, int M, int N, int K, int split, float lo_scale,
                                   hipStream_t stream) {
  SIMVG_CHECK_ARG(ldw >= 2 * K, "gemm_nt_split: weight rows hold [lo | hi], 2 K entries");
  // the lo half must end on a k-tile boundary of EVERY kernel the dispatcher may pick (BK = 64; the 256k32 variant's 32 divides it):
  // with K % 64 == 32 the rescale between the halves never fires and A wraps in the middle of a k-tile
  SIMVG_CHECK_ARG(K % BK == 0, "gemm_nt_split: K must be a multiple of 64");
  return gemm_nt_launch(A, lda, W2, w_gstride, ldw, bias, bias_gstride, C, ldc, c_is_f32, nullptr, 0, residual, ldres,
                        nullptr, 1, 1, M, N, 2 * K, split, 0, 1.f, K, lo_scale, stream);
}

static int gemm_nt_launch(const void* A, int lda, const void* W, long w_gstride, int ldw,
                          const float* bias, int bias_gstride, void* C, int ldc, int c_is_f32,
                          void* aux_preact, int ldaux, const float* residual, int ldres,
                          const float* row_scale, int rows_per_sample0, int rows_per_sample1,
                          int M, int N, int K, int split, int act, float alpha, int ka, float lo_scale, hipStream_t stream,
                          int* plan) {
  // plan != NULL (simvg_gemm_nt_plan): the dispatcher's choice is written to *plan and nothing is launched
#define PLAN(ID_) do { if (plan) { *plan = (ID_); return SIMVG_OK; } } while (0)
  SIMVG_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm_nt: empty problem");
  SIMVG_CHECK_ARG(K % BK == 0, "gemm_nt: K must be a multiple of 64");
  SIMVG_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0 && ldc % 4 == 0, "gemm_nt: leading dims must keep 16-B alignment");
  SIMVG_CHECK_ARG(split >= 0 && split <= M, "gemm_nt: split out of range");
  SIMVG_CHECK_ARG(act >= 0 && act <= 2, "gemm_nt: act must be 0 (none), 1 (gelu) or 2 (relu)");
  if (split == 0) split = M;  // single group uses group 0 weights
  GemmNTArgs a{(const lp_t*)A, lda, (const lp_t*)W, w_gstride, ldw, bias, bias_gstride, C, ldc, c_is_f32,
               (lp_t*)aux_preact, ldaux, residual, ldres, row_scale,
               rows_per_sample0 > 0 ? rows_per_sample0 : 1, rows_per_sample1 > 0 ? rows_per_sample1 : 1,
               M, N, K, split, act, 0, alpha, ka, lo_scale, 1};
  if (const char* e = getenv("SIMVG_GEMM_RESPF")) a.respf = atoi(e);
  // column-group width of the 256x256 tile walk: 6 of >= 9 column tiles (fc1-shape fetch 246 -> 190 MB per launch)
  a.gn = cdiv(N, BNQ) >= 9 ? 6 : 0;
  // wide-N tiles (N a multiple of 256, big M): the M extent is picked by tile-count quantisation on the 256 CUs,
  // cost = rounds x rows per tile (ViT-B: N >= 2304 -> 256 rows, N = 768 -> 160 rows; ViT-L: N = 1024 at B = 32 -> 256 rows)
  const bool wide_ok = (N % 256) == 0 && M >= 2048;
  auto tile_cost = [&](int bm) {
    const long tiles = (long)(cdiv(split, bm) + cdiv(M - split, bm)) * cdiv(N, BNQ);
    // 160-row tiles stage 23 % more bytes per FLOP: they have to win the quantisation estimate by 20 % to be chosen
    return (double)cdiv((int)tiles, 256) * bm * (bm == 160 ? 1.20 : 1.0);
  };
  // 224-row tiles: where they need fewer rows x rounds than both other extents by 10 % (ViT-L's N = 1024 launches at 32 pairs per
  // step: one round of 244 tiles instead of one of 212 longer ones); SIMVG_GEMM_224 = 0 / 1: never / wherever they divide (A/B)
  auto use224 = [&]() {
    const char* e = getenv("SIMVG_GEMM_224");
    if (e && atoi(e) == 0) return false;
    const long t224 = (long)(cdiv(split, 224) + cdiv(M - split, 224)) * cdiv(N, BNQ);
    if (e && atoi(e) == 1) return true;
    const double c224 = (double)cdiv((int)t224, 256) * 224;
    const double best = tile_cost(256) < tile_cost(160) ? tile_cost(256) : tile_cost(160);
    // one round only, and only epilogues the persistent 256-row kernel does not take (fp32 / residual / activation / copy: measured,
    // tools/dev/gemm_vitl_ab.py, out-proj fwd 53.7 -> 41.9 us, fc2 fwd 127.9 -> 112.2; the 16-bit dgrads 29.5 / 74.3 / 98.0 us on the
    // persistent kernel against 31.6 / 80.4 / 104.6 on this one)
    return c224 * 1.10 <= best && t224 <= 256 && !(tile_cost(256) <= tile_cost(160) && persist_ok(a));
  };
  // problems that cannot fill the chip with the big tiles (forward_test at B <= 8): the latency kernel while its 64x64 tiles
  // fit about one residency round of 3 workgroups per CU (tools/dev/gemm_small_bench.py, us for qkv / out / fc1 / fc2:
  // B = 1: 21.7 19.6 20.6 45.5 -> 10.1 8.6 10.5 19.4;  B = 2: 18.8 17.1 19.1 53.1 -> 11.1 9.6 13.2 20.2;
  // B = 4 out, fc2: 17.5 54.0 -> 10.5 25.6;  B = 8 out, fc2: 18.3 55.5 -> 14.7 38.5; above ~800 tiles the big-tile kernels win)
  const long tiles64 = (long)(cdiv(split, 64) + cdiv(M - split, 64)) * cdiv(N, 64);
  // ONE round of 320-row tiles (round 6, gemm_nt_kernel_tall5_*): where the 320-row tiles of the problem fit the CUs and cost fewer
  // rows x rounds than the other extents (ViT-B's N = 768 launches at 64 pairs per step: 255 tiles against two rounds of 160-row
  // tiles); epilogues: 16-bit (+bias) or fp32 (+bias, + residual * row_scale).  SIMVG_GEMM_TALL = 0 / 1: never / wherever it fits (A/B)
  auto use_tall = [&]() {
    const char* e = getenv("SIMVG_GEMM_TALL");
    if (e && atoi(e) == 0) return false;
    if (!wide_ok || a.aux || a.act != 0 || a.alpha != 1.f) return false;
    if (a.c_f32 ? (a.ldc % 4 != 0 || (a.res && a.ldres % 4 != 0)) : (a.res || a.row_scale || a.ldc % 8 != 0)) return false;
    static const int cus = [] { int dev = 0, n = 256; hipGetDevice(&dev); hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n; }();
    const long t320 = (long)(cdiv(split, 320) + cdiv(M - split, 320)) * cdiv(N, BNQ);
    if (e && atoi(e) == 2) return true;          // (measurement: also where the tiles take several dispatch rounds)
    if (t320 > cus) return false;
    if (e && atoi(e) == 1) return true;
    const double best = tile_cost(256) < tile_cost(160) ? tile_cost(256) : tile_cost(160);
    return 320.0 < best;
  };
  // ONE round of 256-row tiles on the same kernel body (gemm_nt_kernel_tall4_f32) for fp32 epilogues, where it needs fewer rows x
  // rounds than the compiler-scheduled extents (these cost ~1.2 x their rows: 224-row / 160-row kernels).  SIMVG_GEMM_TALL4 = 0: never
  auto use_tall4 = [&]() {
    const char* e = getenv("SIMVG_GEMM_TALL4");
    if (e && atoi(e) == 0) return false;
    if (!wide_ok || !a.c_f32 || a.aux || a.act != 0 || a.alpha != 1.f || a.ldc % 4 != 0 || (a.res && a.ldres % 4 != 0)) return false;
    static const int cus = [] { int dev = 0, n = 256; hipGetDevice(&dev); hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n; }();
    const long t256 = (long)(cdiv(split, 256) + cdiv(M - split, 256)) * cdiv(N, BNQ);
    if (t256 > cus) return false;
    const long t224 = (long)(cdiv(split, 224) + cdiv(M - split, 224)) * cdiv(N, BNQ);
    const double c224 = (double)cdiv((int)t224, 256) * 224 * 1.2;
    const double c160 = tile_cost(160);
    return 256.0 <= (c224 < c160 ? c224 : c160);
  };
  // 224-row tiles on the hand-managed 2 x 8-wave body (gemm_nt_kernel_t224_*): where they need fewer rows x rounds than every other
  // extent (ViT-L at 32 pairs per step: 61 row tiles).  SIMVG_GEMM_T224 = 0: never; 2: also over several dispatch rounds (measurement)
  auto use_t224 = [&]() {
    const char* e = getenv("SIMVG_GEMM_T224");
    if (e && atoi(e) == 0) return false;
    if (!wide_ok || a.aux || a.act != 0 || a.alpha != 1.f) return false;
    if (a.c_f32 ? (a.ldc % 4 != 0 || (a.res && a.ldres % 4 != 0)) : (a.res || a.row_scale || a.ldc % 8 != 0)) return false;
    static const int cus = [] { int dev = 0, n = 256; hipGetDevice(&dev); hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n; }();
    const long t224 = (long)(cdiv(split, 224) + cdiv(M - split, 224)) * cdiv(N, BNQ);
    if (t224 > cus && !(e && atoi(e) == 2)) return false;
    const double c224 = (double)cdiv((int)t224, cus) * 224;
    const long t320 = (long)(cdiv(split, 320) + cdiv(M - split, 320)) * cdiv(N, BNQ);
    double best = tile_cost(256) < tile_cost(160) ? tile_cost(256) : tile_cost(160);
    if (t320 <= cus && 320.0 < best) best = 320.0;
    return c224 < 0.97 * best;
  };
  if (tiles64 > 800 && !use_tall() && use_t224()) {
    PLAN(SIMVG_GEMM_PLAN_T224);
    using G7 = Tall28Geo<7>;
    const int tiles = (cdiv(split, G7::ROWS) + cdiv(M - split, G7::ROWS)) * cdiv(N, BNQ);
    static bool once7 = hipFuncSetAttribute((const void*)gemm_nt_kernel_t224_lp, hipFuncAttributeMaxDynamicSharedMemorySize, G7::SMEM) == hipSuccess &&
                        hipFuncSetAttribute((const void*)gemm_nt_kernel_t224_f32, hipFuncAttributeMaxDynamicSharedMemorySize, G7::SMEM) == hipSuccess;
    (void)once7;
    if (a.c_f32) hipLaunchKernelGGL(gemm_nt_kernel_t224_f32, dim3(tiles), dim3(1024), G7::SMEM, stream, a);
    else hipLaunchKernelGGL(gemm_nt_kernel_t224_lp, dim3(tiles), dim3(1024), G7::SMEM, stream, a);
  } else if (tiles64 > 800 && !use_tall() && use_tall4()) {
    PLAN(SIMVG_GEMM_PLAN_TALL4);
    using G4 = TallGeo<4>;
    const int tiles = (cdiv(split, G4::ROWS) + cdiv(M - split, G4::ROWS)) * cdiv(N, BNQ);
    static bool once4 = hipFuncSetAttribute((const void*)gemm_nt_kernel_tall4_f32, hipFuncAttributeMaxDynamicSharedMemorySize, G4::SMEM) == hipSuccess;
    (void)once4;
    hipLaunchKernelGGL(gemm_nt_kernel_tall4_f32, dim3(tiles), dim3(1024), G4::SMEM, stream, a);
  } else if (tiles64 > 800 && use_tall()) {
    PLAN(SIMVG_GEMM_PLAN_TALL5);
    using G5 = TallGeo<5>;
    const int tiles = (cdiv(split, G5::ROWS) + cdiv(M - split, G5::ROWS)) * cdiv(N, BNQ);
    if (a.c_f32) {
      static bool once1 = hipFuncSetAttribute((const void*)gemm_nt_kernel_tall5_f32, hipFuncAttributeMaxDynamicSharedMemorySize, G5::SMEM) == hipSuccess;
      (void)once1;
      hipLaunchKernelGGL(gemm_nt_kernel_tall5_f32, dim3(tiles), dim3(1024), G5::SMEM, stream, a);
    } else {
      static bool once0 = hipFuncSetAttribute((const void*)gemm_nt_kernel_tall5_lp, hipFuncAttributeMaxDynamicSharedMemorySize, G5::SMEM) == hipSuccess;
      (void)once0;
      hipLaunchKernelGGL(gemm_nt_kernel_tall5_lp, dim3(tiles), dim3(1024), G5::SMEM, stream, a);
    }
  } else if (tiles64 <= 800) {
    PLAN(SIMVG_GEMM_PLAN_LAT);
    static bool oncel = hipFuncSetAttribute((const void*)gemm_nt_kernel_lat<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * LAT_STAGE) == hipSuccess;
    (void)oncel;
    hipLaunchKernelGGL(gemm_nt_kernel_lat<3>, dim3((int)tiles64), dim3(256), 3 * LAT_STAGE, stream, a);
  } else if (wide_ok && use224()) {
    PLAN(SIMVG_GEMM_PLAN_224);
    constexpr int SM224 = 2 * (224 + BNQ) * BK * 2;   // 120 KiB ring; the epilogue staging (16 waves x 32 x 36 x 4 B = 72 KiB) fits
    static bool once224 = hipFuncSetAttribute((const void*)gemm_nt_kernel_224x256_w16, hipFuncAttributeMaxDynamicSharedMemorySize, SM224) == hipSuccess;
    (void)once224;
    const int tiles = (cdiv(split, 224) + cdiv(M - split, 224)) * cdiv(N, BNQ);
    hipLaunchKernelGGL(gemm_nt_kernel_224x256_w16, dim3(tiles), dim3(1024), SM224, stream, a);
  } else if (wide_ok && tile_cost(256) <= tile_cost(160) && persist_ok(a)) {
    PLAN(a.ka < a.K ? SIMVG_GEMM_PLAN_PERSIST_SPLIT : SIMVG_GEMM_PLAN_PERSIST);
    static bool oncep = hipFuncSetAttribute((const void*)gemm_nt_kernel_256sq_p, hipFuncAttributeMaxDynamicSharedMemorySize, PQ_SMEM) == hipSuccess &&
                        hipFuncSetAttribute((const void*)gemm_nt_kernel_256sq_p2, hipFuncAttributeMaxDynamicSharedMemorySize, PQ_SMEM) == hipSuccess;
    (void)oncep;
    const int tiles = (cdiv(split, 256) + cdiv(M - split, 256)) * cdiv(N, BNQ);
    static const int cus = [] { int dev = 0, n = 256; hipGetDevice(&dev); hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n & ~7; }();
    static unsigned long long* prof = getenv("SIMVG_PQ_PROF_PTR") ? (unsigned long long*)strtoull(getenv("SIMVG_PQ_PROF_PTR"), nullptr, 0) : nullptr;
    if (a.ka < a.K) hipLaunchKernelGGL(gemm_nt_kernel_256sq_p2, dim3(tiles < cus ? tiles : cus), dim3(1024), PQ_SMEM, stream, a, tiles, prof);
    else hipLaunchKernelGGL(gemm_nt_kernel_256sq_p, dim3(tiles < cus ? tiles : cus), dim3(1024), PQ_SMEM, stream, a, tiles, prof);
  } else if (wide_ok && tile_cost(256) <= tile_cost(160)) {
    PLAN(SIMVG_GEMM_PLAN_256);
    constexpr int SMW = 160 * 1024;      // ring 128 KiB; the 16-wave epilogue staging needs 136 KiB
    static bool oncew = hipFuncSetAttribute((const void*)gemm_nt_kernel_256sq_w16, hipFuncAttributeMaxDynamicSharedMemorySize, SMW) == hipSuccess;
    (void)oncew;
    const int tiles = (cdiv(split, 256) + cdiv(M - split, 256)) * cdiv(N, BNQ);
    hipLaunchKernelGGL(gemm_nt_kernel_256sq_w16, dim3(tiles), dim3(1024), SMW, stream, a);
  } else if (wide_ok) {
    PLAN(SIMVG_GEMM_PLAN_160);
    constexpr int SM3 = 3 * (160 + BNQ) * BK * 2;     // 156 KiB ring; 16 waves x 32 x 36 x 4 B = 72 KiB of epilogue staging fit
    static bool once3r = hipFuncSetAttribute((const void*)gemm_nt_kernel_160x256_r3, hipFuncAttributeMaxDynamicSharedMemorySize, SM3) == hipSuccess;
    (void)once3r;
    const int tiles = (cdiv(split, 160) + cdiv(M - split, 160)) * cdiv(N, BNQ);
    hipLaunchKernelGGL(gemm_nt_kernel_160x256_r3, dim3(tiles), dim3(1024), SM3, stream, a);
  } else if (M >= 512) {
    PLAN(SIMVG_GEMM_PLAN_256K32);
    static bool once3 = hipFuncSetAttribute((const void*)gemm_nt_kernel_256k32, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            3 * STAGE3) == hipSuccess;
    (void)once3;
    const int tiles = (cdiv(split, BM2) + cdiv(M - split, BM2)) * cdiv(N, BN);
    hipLaunchKernelGGL(gemm_nt_kernel_256k32, dim3(tiles), dim3(512), 3 * STAGE3, stream, a);
  } else {
    PLAN(SIMVG_GEMM_PLAN_128);
    const int tiles = (cdiv(split, BM) + cdiv(M - split, BM)) * cdiv(N, BN);
    static bool oncem = hipFuncSetAttribute((const void*)gemm_nt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MID_NST * 2 * TILE_BYTES) == hipSuccess;
    (void)oncem;
    hipLaunchKernelGGL(gemm_nt_kernel, dim3(tiles), dim3(256), MID_NST * 2 * TILE_BYTES, stream, a);
  }
#undef PLAN
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}

// Which kernel simvg_gemm_nt / simvg_gemm_nt_split would launch for a problem (host logic only: nothing is launched, no device is
// touched beyond the attribute queries the dispatcher makes): the tile-extent cost model is testable without a GPU.
// -> one of SIMVG_GEMM_PLAN_* (include/simvg_hip.h), or a negative error code
extern "C" int simvg_gemm_nt_plan(int M, int N, int K, int split, int c_is_f32, int has_residual, int has_row_scale, int act,
                                  int has_aux, int split_weights) {
  int plan = 0;
  void* p = (void*)(uintptr_t)256;     // (never dereferenced)
  const int kk = split_weights ? 2 * K : K, ld = (kk + 7) / 8 * 8, ldn = (N + 7) / 8 * 8;
  const int rc = gemm_nt_launch(p, (K + 7) / 8 * 8, p, 0, ld, nullptr, 0, p, ldn, c_is_f32, has_aux ? p : nullptr, ldn, has_residual ? (const float*)p : nullptr,
                                ldn, has_row_scale ? (const float*)p : nullptr, 1, 1, M, N, kk, split, act, 1.f, K, split_weights ? 0.5f : 1.f,
                                nullptr, &plan);
  return rc == SIMVG_OK ? plan : rc;
}

static int gemm_tn_impl(const void* dY, int lddy, const void* X, int ldx, float* dW, long dw_gstride,
                        int lddw, float* db, int db_gstride, int M, int N, int K, int split, float out_scale,
                        float* slab_ws, simvg_wgrad_reduce_desc* defer, hipStream_t stream);

extern "C" int simvg_gemm_tn(const void* dY, int lddy, const void* X, int ldx, float* dW, long dw_gstride,
                             int lddw, float* db, int db_gstride, int M, int N, int K, int split, float out_scale,
                             hipStream_t stream) {
  return gemm_tn_impl(dY, lddy, X, ldx, dW, dw_gstride, lddw, db, db_gstride, M, N, K, split, out_scale, nullptr, nullptr, stream);
}

// the same with a caller-owned workspace of simvg_gemm_tn_ws_floats(M, N, K) floats: the XCD-partitioned kernel then leaves the
// partial sums of its eight (sixteen) row partitions in slabs of that workspace (plain stores) and a second stage adds them
// into dW in a fixed order, instead of 8 x N x K fp32 atomics (47 of the 166 us of the fc1 launch).  defer == NULL: the second
// stage is launched here; otherwise its description is written to *defer (defer->slabs == NULL: nothing to do, the problem
// took a kernel that needs no second stage) and simvg_wgrad_reduce_batched runs up to SIMVG_WGRAD_REDUCE_MAX of them in one
// launch on ANY stream that is ordered behind this one -- the reduction is pure HBM streaming and can ride beside the next GEMMs.
extern "C" long simvg_gemm_tn_ws_floats(int M, int N, int K) { return simvg_wgrad_x_slab_floats(M, N, K); }
extern "C" int simvg_gemm_tn_ws(const void* dY, int lddy, const void* X, int ldx, float* dW, long dw_gstride,
                                int lddw, float* db, int db_gstride, int M, int N, int K, int split, float out_scale,
                                float* ws, simvg_wgrad_reduce_desc* defer, hipStream_t stream) {
  return gemm_tn_impl(dY, lddy, X, ldx, dW, dw_gstride, lddw, db, db_gstride, M, N, K, split, out_scale, ws, defer, stream);
}
extern "C" int simvg_wgrad_reduce_batched(const simvg_wgrad_reduce_desc* descs, int n, hipStream_t stream) {
  SIMVG_CHECK_ARG(descs && n > 0 && n <= SIMVG_WGRAD_REDUCE_MAX, "wgrad_reduce_batched: 1 .. SIMVG_WGRAD_REDUCE_MAX descriptors");
  simvg_wgrad_reduce_launch(descs, n, stream);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}

static int gemm_tn_impl(const void* dY, int lddy, const void* X, int ldx, float* dW, long dw_gstride,
                        int lddw, float* db, int db_gstride, int M, int N, int K, int split, float out_scale,
                        float* slab_ws, simvg_wgrad_reduce_desc* defer, hipStream_t stream) {
  SIMVG_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm_tn: empty problem");
  SIMVG_CHECK_ARG(N % 8 == 0 && K % 8 == 0 && lddy % 8 == 0 && ldx % 8 == 0, "gemm_tn: N, K, ld must be multiples of 8");
  SIMVG_CHECK_ARG(split >= 0 && split <= M, "gemm_tn: split out of range");
  if (split == 0) split = M;
  if (defer) defer->slabs = nullptr;
  if (simvg_wgrad_x(dY, lddy, X, ldx, dW, dw_gstride, lddw, db, db_gstride, M, N, K, split, out_scale, slab_ws, defer, stream)) {
    SIMVG_LAUNCH_CHECK();
    return SIMVG_OK;
  }
  // 256x128 ring kernel for the wide problems (qkv / fc1 / fc2 wgrad); the small out-proj stays on the 128x128 kernel
  const bool big = cdiv(N, 256) * cdiv(K, 128) >= 24;
  const int tiles = big ? cdiv(N, 256) * cdiv(K, 128) : cdiv(N, 128) * cdiv(K, 128);
  // split the contraction so that tiles x chunks ~ 2-3 blocks per CU (measured sweep, profiles/r01_sweeps.md)
  const int target_blocks = big ? 384 : (tiles <= 48 ? 384 : 768);
  int want = cdiv(target_blocks, tiles);
  int rpc = cdiv(cdiv(M, want), 64) * 64;
  if (rpc < 256) rpc = 256;   // multiple of 64 (and of the 32-row stages)
  const int chunks0 = cdiv(split, rpc), chunks1 = cdiv(M - split, rpc);
  GemmTNArgs a{(const lp_t*)dY, lddy, (const lp_t*)X, ldx, dW, dw_gstride, lddw, M, N, K, split, rpc, chunks0,
               db, db_gstride, tiles, cdiv(rpc, 512), 3, out_scale};
  const int gx = tiles + (db ? cdiv(N, 256) * a.cs_split : 0);   // + column-sum blocks (bias gradient)
  if (big) {
    static bool once = hipFuncSetAttribute((const void*)gemm_tn_kernel_256, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           3 * 24576) == hipSuccess;
    (void)once;
    hipLaunchKernelGGL(gemm_tn_kernel_256, dim3(gx, chunks0 + chunks1), dim3(512), 3 * 24576, stream, a);
  } else {
    hipLaunchKernelGGL(gemm_tn_kernel, dim3(gx, chunks0 + chunks1), dim3(256), 65536, stream, a);
  }
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}

extern "C" int simvg_colsum(const void* Y, int ldy, float* out, int out_gstride, int M, int N, int split,
                            hipStream_t stream) {
  SIMVG_CHECK_ARG(M > 0 && N > 0 && N % 8 == 0 && ldy % 8 == 0, "colsum: N, ld must be multiples of 8");
  if (split == 0) split = M;
  const int rpc = 512;
  const int chunks0 = cdiv(split, rpc), chunks1 = cdiv(M - split, rpc);
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(N, 256), chunks0 + chunks1), dim3(256), 0, stream,
                     (const lp_t*)Y, ldy, out, out_gstride, M, N, split, rpc, chunks0);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}
