/* libsimvg_hip.so -- C ABI of the MI355X-native (gfx950) SimVG hot path.
 *
 * Drop-in boundary: plain C functions, one per fused op and direction.  All pointers
 * are DEVICE pointers owned by the caller (PyTorch-ROCm allocator: tensor.data_ptr()); the library
 * never allocates; every call only ENQUEUES on the passed hipStream_t (no hidden synchronisation);
 * return 0 on success, <0 on error (message via simvg_last_error()); no global mutable state besides
 * the thread-local error string.  Matrices are row-major with explicit leading dimensions (elements).
 * "lp" = the library's 16-bit operand / storage format as raw bits (uint16_t): IEEE fp16 by default (simvg_lowp_format()
 * == 1), bfloat16 in a -DSIMVG_LOWP_BF16 build (== 2); accumulation is always fp32.  fp16 is the default because bf16
 * operand rounding cannot meet the path's parity bound (boxes within 1e-3 L1 of the fp32 reference) on trained-scale
 * weights; backward tensors therefore carry a caller-chosen power-of-two gradient scale that the
 * `*_scale` arguments below apply / remove, and every 16-bit store saturates instead of producing inf.
 * Weights keep the reference state_dict layout [out, in].
 *
 * Row layout of every activation matrix is MODALITY-MAJOR: the vision tokens of all samples first
 * ([B*Nv] rows), then the text tokens ([B*Nt] rows).  `split` = B*Nv is where the multiway "A"
 * (vision) expert ends and the "B" (text) expert begins; split == 0 means a single expert / group.
 * This replaces torchscale's MultiwayNetwork split/cat (reference beit3_base.py:128-130,362-364).
 *
 * The reference has no FFI for this path (it is eager PyTorch on top of torchscale / detrex; reference
 * setup.py:120 `ext_modules=[]`); each entry point cites the reference call site whose arithmetic it
 * replaces.  Paths are relative to the reference root.
 */
#ifndef SIMVG_HIP_H
#define SIMVG_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* simvg_stream_t; /* == hipStream_t */

/* ---- library ---- */
int simvg_version(void);
/* sha256[:32] of the csrc/ sources (+ variant flags) this library was built from; the Python binding compares it with the
 * sources it finds beside it and refuses a stale library (simvg_amd/build.py, simvg_amd/_lib.py) */
const char* simvg_source_hash(void);
const char* simvg_last_error(void);
int simvg_lowp_format(void); /* 1 = IEEE fp16, 2 = bfloat16 */

/* ---- dense contractions (16-bit MFMA, fp32 accumulate) -------------------------------------------
 * C[M,N] = alpha * A[M,K] . W[g][N,K]^T (+bias[g][N]) (+act) ; optional pre-activation copy (lp) ; optional
 * fused residual: C = residual + row_scale[sample(m)] * (...), the DropPath + residual_connection of
 * simvg/models/vis_encs/beit/beit3_base.py:146-151,166-169.  act: 0 none, 1 exact-erf GELU, 2 ReLU.
 * Replaces the multiway nn.Linear calls of torchscale MultiheadAttention (q/k/v/out_proj) and
 * FeedForwardNetwork (fc1 -> gelu, fc2) invoked at beit3_base.py:137-145,159, the patch-embed Conv2d
 * (beit3_base.py:461, after simvg_im2col) and the head projections
 * simvg/models/heads/tgqs_kd_detr_head/tgqs_kd_detr_head.py:377-379. */
int simvg_gemm_nt(const void* A_lp, int lda, const void* W_lp, long w_group_stride, int ldw,
                  const float* bias, int bias_group_stride, void* C, int ldc, int c_is_f32,
                  void* aux_preact_lp, int ldaux, const float* residual, int ldres,
                  const float* row_scale, int rows_per_sample0, int rows_per_sample1,
                  int M, int N, int K, int split, int act, float alpha, simvg_stream_t stream);
/* The same contraction with the weight held as TWO 16-bit numbers per entry (precise inference forward): rows of W2 are
 * [lo * 2^s | hi] along K (2 K entries, ldw >= 2 K), hi = the 16-bit rounding of w, lo = the 16-bit rounding of (w - hi) * 2^s,
 * lo_scale = 2^-s.  C[M,N] = A[M,K] . (hi + lo)[g][N,K]^T (+bias) (+residual), i.e. the reference's fp32 weight
 * (nn.Linear of beit3_base.py:137-145,159) carried to ~22 significand bits at twice the MFMA work. */
int simvg_gemm_nt_split(const void* A_lp, int lda, const void* W2_lp, long w_group_stride, int ldw,
                        const float* bias, int bias_group_stride, void* C, int ldc, int c_is_f32,
                        const float* residual, int ldres, int M, int N, int K, int split, float lo_scale,
                        simvg_stream_t stream);
/* Which kernel simvg_gemm_nt / simvg_gemm_nt_split would launch for a problem (host logic only, nothing is launched): the
 * dispatcher's tile-extent cost model -- rounds x rows per tile on the CUs of the device (256 without one) -- as data, so that it
 * can be tested without a GPU (tests/test_abi.py).  -> SIMVG_GEMM_PLAN_*, or a negative error code. */
#define SIMVG_GEMM_PLAN_LAT 1            /* 64 x 64 latency kernel (few tiles: forward_test at small batches) */
#define SIMVG_GEMM_PLAN_TALL5 2          /* one round of 320 x 256 tiles, hand-managed loop (ViT-B's N = 768 launches at 64 pairs) */
#define SIMVG_GEMM_PLAN_T224 3           /* one round of 224 x 256 tiles, 2 x 8 waves, hand-managed (ViT-L's N = 1024 launches at 32 pairs) */
#define SIMVG_GEMM_PLAN_TALL4 4          /* one round of 256 x 256 tiles, hand-managed, fp32 epilogues */
#define SIMVG_GEMM_PLAN_224 5            /* 224 x 256, compiler-scheduled (epilogues with an activation) */
#define SIMVG_GEMM_PLAN_PERSIST 6        /* persistent 256 x 256 (16-bit output + bias) */
#define SIMVG_GEMM_PLAN_PERSIST_SPLIT 7  /* ... with hi + lo weights */
#define SIMVG_GEMM_PLAN_256 8            /* 256 x 256, one tile per workgroup */
#define SIMVG_GEMM_PLAN_160 9            /* 160 x 256, three-stage ring */
#define SIMVG_GEMM_PLAN_256K32 10        /* 256 x 128, k-tiles of 32 */
#define SIMVG_GEMM_PLAN_128 11           /* 128 x 128 */
int simvg_gemm_nt_plan(int M, int N, int K, int split, int c_is_f32, int has_residual, int has_row_scale, int act, int has_aux,
                       int split_weights);
/* dW[g][N,K] += out_scale * dY[M,N]^T . X[M,K]  (weight gradient of the same Linears; fp32 accumulate); optional fused bias
 * gradient db[g][N] += out_scale * column sums of dY over the rows of group g (extra streaming blocks of the same
 * launch).  out_scale = 1 / (gradient scale carried by dY). */
int simvg_gemm_tn(const void* dY_lp, int lddy, const void* X_lp, int ldx, float* dW, long dw_group_stride,
                  int lddw, float* db, int db_group_stride, int M, int N, int K, int split, float out_scale,
                  simvg_stream_t stream);
/* The same with a caller-owned workspace of simvg_gemm_tn_ws_floats(M, N, K) floats (0: the shape does not use one): the
 * partial sums of the kernel's row partitions are written to slabs of the workspace (plain stores) and a second stage adds
 * them into dW in a fixed order -- bit-reproducible, and without the 8 x N x K fp32 atomics of simvg_gemm_tn.  defer == NULL:
 * the second stage is launched here.  Otherwise its description is written to *defer (host memory; defer->slabs == NULL
 * afterwards: nothing to reduce) and simvg_wgrad_reduce_batched runs up to SIMVG_WGRAD_REDUCE_MAX second stages in ONE launch,
 * on any stream ordered behind the first stage (the workspace has to stay untouched until then). */
typedef struct simvg_wgrad_reduce_desc {
  const float* slabs; float* dW; long dw_group_stride;
  int lddw, N, K, Q, lo0, hi0, lo1, hi1;
  int assign;   /* written 0 by simvg_gemm_tn_ws; a caller that has NOT zeroed dW sets it to 1 or 2 before simvg_wgrad_reduce_batched:
                 * dW = sum of the slabs instead of dW += sum -- the first accumulation of a step then needs neither the zero
                 * fill of dW nor its read.  1: only row groups that have rows are written (dW may hold one group only); 2: dW
                 * holds both groups, and one without rows (split == M) is zeroed */
} simvg_wgrad_reduce_desc;
#define SIMVG_WGRAD_REDUCE_MAX 16
long simvg_gemm_tn_ws_floats(int M, int N, int K);
int simvg_gemm_tn_ws(const void* dY_lp, int lddy, const void* X_lp, int ldx, float* dW, long dw_group_stride,
                     int lddw, float* db, int db_group_stride, int M, int N, int K, int split, float out_scale,
                     float* ws, simvg_wgrad_reduce_desc* defer, simvg_stream_t stream);
int simvg_wgrad_reduce_batched(const simvg_wgrad_reduce_desc* descs, int n, simvg_stream_t stream);
/* out[g][N] += column sums of Y over the rows of group g (bias gradients) */
int simvg_colsum(const void* Y_lp, int ldy, float* out, int out_group_stride, int M, int N, int split,
                 simvg_stream_t stream);

/* ---- LayerNorm (multiway gamma/beta by row group) -----------------------------------------------
 * torch.nn.LayerNorm under MultiwayWrapper: beit3_base.py:136 (self_attn_layer_norm), :157
 * (final_layer_norm), :396-397 (encoder.layer_norm), torchscale inner_attn_ln / ffn_layernorm; and the
 * decoder norms of heads/tgqs_kd_detr_head/transformer.py:119-132. */
int simvg_ln_fwd(const void* x, int x_is_lp, int ldx, const float* gamma, const float* beta, int group_stride,
                 void* y_lp, int ldy, float* y_f32, int ldy32, float* mean, float* rstd,
                 int M, int D, int split, float eps, int x_is_gelu_preact /* x = fc1 pre-activation u: normalise
                 gelu(u), recomputed in registers -- torchscale FeedForwardNetwork: ffn_layernorm(gelu(fc1(x))) */,
                 simvg_stream_t stream);
/* dx = LN'(dy); outputs: lp dx (optionally * GELU'(u), fusing the activation backward of fc1), and/or
 * fp32 (dres + dx) = the residual-stream gradient, with an optional lp copy * row_scale (DropPath).
 * gelu_u_lp == x (same pointer): x is the pre-activation and the LayerNorm input gelu(u) is recomputed. */
int simvg_ln_bwd(const void* dy, int dy_is_f32, int lddy, const void* x, int x_is_lp, int ldx, const float* mean,
                 const float* rstd, const float* gamma, int group_stride, float* dgamma, float* dbeta,
                 void* dx_lp, int lddxb, const void* gelu_u_lp, int ldu, const float* dres,
                 float* dx_f32, int lddxf, void* dx_scaled_lp, int lddxs, const float* row_scale,
                 int rows_per_sample0, int rows_per_sample1, int M, int D, int split, float* partial_ws /* optional:
                 >= simvg_ln_bwd_ws_floats() floats -> two-stage dgamma/dbeta reduction instead of atomics */,
                 float dy_scale /* multiplies an fp32 dy on load: the entry of a scaled backward (1 otherwise) */,
                 float param_scale /* multiplies what is added to dgamma / dbeta: 1 / gradient scale of dy */,
                 simvg_stream_t stream);
long simvg_ln_bwd_ws_floats(int M, int D, int split);
/* The same backward with the SECOND stage of the two-stage dgamma / dbeta reduction left out: its description is written to
 * *desc_out (host memory), and simvg_ln_param_reduce_batched runs the second stages of up to SIMVG_LN_REDUCE_MAX such calls in
 * ONE launch (each partial workspace has to stay untouched until then).  A training step's 49 LayerNorm backward calls
 * otherwise end in 49 launches of ~7 us on 48 workgroups each.  The sums are formed in the same fixed order either way
 * (bit-identical dgamma / dbeta).  desc_out->partial == NULL afterwards: the call took a kernel that reduces with atomics. */
typedef struct simvg_ln_reduce_desc {
  const float* partial; float* dgamma; float* dbeta;
  int group_stride, D, blocks0, blocks1;
} simvg_ln_reduce_desc;
#define SIMVG_LN_REDUCE_MAX 64
int simvg_ln_bwd_deferred(const void* dy, int dy_is_f32, int lddy, const void* x, int x_is_lp, int ldx, const float* mean,
                          const float* rstd, const float* gamma, int group_stride, float* dgamma, float* dbeta,
                          void* dx_lp, int lddxb, const void* gelu_u_lp, int ldu, const float* dres,
                          float* dx_f32, int lddxf, void* dx_scaled_lp, int lddxs, const float* row_scale,
                          int rows_per_sample0, int rows_per_sample1, int M, int D, int split, float* partial_ws,
                          float dy_scale, float param_scale, simvg_ln_reduce_desc* desc_out, simvg_stream_t stream);
int simvg_ln_param_reduce_batched(const simvg_ln_reduce_desc* descs, int n, simvg_stream_t stream);

/* ---- fused encoder self-attention ----------------------------------------------------------------
 * softmax(scale * Q K^T + key_padding(-inf)) V per (sample, head), head_dim 64.  N = Nv+Nt <= 448: K and V of a
 * head resident in LDS (the path's 421 tokens have compile-time-geometry kernels); larger N (patch 16, images above 640):
 * K / V streamed through LDS in 256-row blocks with an online softmax.
 * torchscale MultiheadAttention.forward as called at beit3_base.py:137-145 (bmm, masked_fill, fp32
 * softmax, bmm, head merge).  qkv: [M, 3D] = q | k | v columns.  pad: [B,Nt] bytes, 1 = padded key.
 * Backward (autograd of the same call): dqkv = d(q | k | v) from dout, the saved out and lse.  delta_ws: [B*H, N] floats of
 * scratch the CALLER owns -- rowsum(dO * O) for the two-kernel form; the one-pass form of the path's geometry (27 key tiles,
 * csrc/attention_bwd1.hip, default; SIMVG_ATTN_BWD1=0 selects the two kernels) only parks the stores of rows beyond N in it. */
int simvg_attn_fwd(const void* qkv_lp, int ldqkv, void* out_lp, int ldo, float* lse, const unsigned char* pad,
                   int B, int H, int Nv, int Nt, int D, float scale, simvg_stream_t stream);
/* Measurement entry point: the QK^T contraction of simvg_attn_fwd alone (same K staging in LDS, same MFMAs; no softmax, no PV) for
 * the path's geometry; rowmax [B*H, N] = max over the keys of the scaled scores.  BASELINE.json's north_star quotes its target on
 * "the encoder QK^T GEMM" (torchscale MultiheadAttention's `torch.bmm(q, k.transpose(1, 2))`, called from beit3_base.py:137-145). */
int simvg_attn_qk_probe(const void* qkv_lp, int ldqkv, float* rowmax, const unsigned char* key_padding_mask, int B, int H, int Nv,
                        int Nt, int D, float scale, simvg_stream_t stream);
int simvg_attn_bwd(const void* qkv_lp, int ldqkv, const void* out_lp, int ldo, const void* dout_lp, int lddo,
                   void* dqkv_lp, int lddqkv, const float* lse, float* delta_ws, const unsigned char* pad,
                   int B, int H, int Nv, int Nt, int D, float scale, simvg_stream_t stream);

/* ---- decoder head: exact-fp32 small GEMM, small multi-head attention --------------------------------
 * C[M,N] (+)= sum_k A(m,k) B(k,n) (+bias[n]) (+addend[m % addend_rows][n]) (+ReLU), arbitrary strides (forward,
 * dgrad and wgrad of the head's nn.Linear layers: detrex MultiheadAttention / FFN built at
 * heads/tgqs_kd_detr_head/transformer.py:106-125, heads/utils.py:39-46 MLP, tgqs_kd_detr_head.py:378-379,
 * 415-416,427-428) on v_mfma_f32_16x16x4_f32. */
int simvg_gemm_f32(const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C, long ldc,
                   const float* bias, const float* addend, long ld_addend, int addend_rows, int M, int N, int K,
                   int accumulate, int act, simvg_stream_t stream);
/* Up to 12 independent simvg_gemm_f32 problems in ONE launch (e.g. the dgrad, wgrad and bias-gradient GEMMs of one
 * nn.Linear backward, or the q|k and v projections of an attention): same arithmetic per problem as simvg_gemm_f32.
 * No problem may read or write another problem's C.  `problems` is a HOST array, consumed before the call returns. */
typedef struct simvg_gemm_f32_problem {
  const float* A; long sam, sak;
  const float* B; long sbk, sbn;
  float* C; long ldc;
  const float* bias;
  const float* addend; long ld_addend; int addend_rows;
  int M, N, K, accumulate, act;
  const float* A2; const float* B2;     /* optional second operands with A's / B's strides: the product uses A + A2 and
                                         * B + B2 (q = (x + query_pos) W without materialising the sum) */
  const float* mult; long ld_mult;      /* optional [M, N] factor applied after the activation (dropout multipliers) */
  const float* gate; long ld_gate;      /* optional [M, N] gate: the value passes where gate > 0 (ReLU backward on the
                                         * saved layer output).  With mult or gate the addend is added AFTER them:
                                         * y = dropout(act(x W + b)) + residual (BaseTransformerLayer FFN,
                                         * heads/tgqs_kd_detr_head/transformer.py:106-125) */
} simvg_gemm_f32_problem;
int simvg_gemm_f32_grouped(const simvg_gemm_f32_problem* problems, int count, simvg_stream_t stream);
/* The same launch with a workspace (16-byte aligned device memory, `workspace_floats` floats, owned by the caller and not used by
 * anything else queued on other streams): problems of the num_queries = 10 size (>= 40 M multiply-adds, few 64 x 64 tiles, long K --
 * the FFN / projection Linears over B * num_queries rows and their weight gradients, transformer.py:106-125,167-186) split K over
 * several workgroups per tile; their partial tiles meet in a second launch that adds them in a fixed order and applies the
 * epilogue (deterministic).  Without a workspace (or a too small one) nothing is split. */
int simvg_gemm_f32_grouped_ws(const simvg_gemm_f32_problem* problems, int count, float* workspace, long workspace_floats,
                              simvg_stream_t stream);
/* torch.nn.MultiheadAttention core (heads of 32) for <= 16 queries: softmax(scale q k^T + key_padding) [* dropout] v
 * (detrex MultiheadAttention wrapper; decoder layers transformer.py:167-186).
 * key_pos (optional): rows [Lk, E] (key_pos_rows_per_batch = 0: shared by the batch) or [B * Lk, E] (= Lk) of the PROJECTED
 * key positional embedding, added to the K rows on load (K = (memory + key_pos) W_k^T = memory W_k^T + key_pos W_k^T); the
 * backward's dk is then also the gradient of those rows. */
int simvg_attn_small_fwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* out, int ldo,
                         float* P, const unsigned char* key_padding_mask, const float* drop_mult, int B, int H, int Lq,
                         int Lk, int kv_rows_per_batch, float scale, const float* key_pos, int ld_key_pos,
                         int key_pos_rows_per_batch, simvg_stream_t stream);
int simvg_attn_small_bwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* P,
                         const unsigned char* key_padding_mask, const float* drop_mult, const float* dout, int lddo,
                         float* dq, int lddq, float* dk, int lddk, float* dv, int lddv, int B, int H, int Lq, int Lk,
                         int kv_rows_per_batch, float scale, const float* key_pos, int ld_key_pos,
                         int key_pos_rows_per_batch, simvg_stream_t stream);

/* ---- a decoder layer's attention block as ONE launch (round 5) ---------------------------------------------------------
 * detrex BaseTransformerLayer as the reference configures it (simvg/models/heads/tgqs_kd_detr_head/transformer.py:93-131;
 * called from the decoder loop :134-186 and for the TGQG layers at tgqs_kd_detr_head.py:391-399, decoder :425-428), operation
 * order (self_attn, norm, cross_attn, norm): one workgroup per sample owns its R = num_queries rows and computes
 *   q|k = (tgt + qpos) Ws[0:2E]^T, v = tgt Ws[2E:]^T; self-attention (8 heads x 32, dropout multipliers dm0);
 *   r1 = tgt + o Wso^T + bso; t1 = LayerNorm(r1; g0, b0);
 *   qc = ((t1 + qpos) Wc[0:E]^T + bc[0:E]) / sqrt(32); cross-attention over the sample's Lk SOURCE rows s_k (16-bit or fp32
 *   rows; key = s_k + kpos_k, value = s_k) WITHOUT materialising K = s Wk^T, V = s Wv^T:
 *       qk[r][h] = Wk_h^T qc_h[r];  P = softmax_k(qk[r][h] . (s_k + kpos_k)) (key_padding_mask -> -inf);  P' = P * dm1
 *       ctx[r][h] = sum_k P'[k] s_k;  o2[r][h] = Wv_h ctx[r][h] + bv_h * sum_k P'[k]
 *   (identical to nn.MultiheadAttention: the key bias adds a per-row constant to the scores, which the softmax removes);
 *   r2 = t1 + o2 Wco^T + bco; t2 = LayerNorm(r2; g1, b1).
 * Every intermediate the backward needs is written to the caller's buffers (rows of [B*R, .] matrices; P0 [B,H,R,R],
 * P1 [B,H,R,Lk], qk / ctx [B*R, H, E], sp [B*R, H]).  E = 256, H = 8, R <= 16, Lk <= 1024. */
typedef struct simvg_dec_attn_args {
  int B, R, Lk, kv_rows, kv_off;       /* source rows of sample b: rows b*kv_rows + kv_off + [0, Lk) */
  const float* tgt; const float* qpos; /* [B*R, E] */
  const float *Ws, *bs, *Wso, *bso, *g0, *b0, *Wc, *bc, *Wco, *bco, *g1, *b1;
  const void* src16; const float* src32; long ldsrc;   /* exactly one of the two; row stride in elements */
  const float* kpos; long ldkp; int kpos_rows;         /* key_pos rows (or NULL); kpos_rows = rows per sample, 0 = one set */
  const unsigned char* kpm;                            /* [B, Lk], 1 = masked key, or NULL */
  const float* dm0; const float* dm1;                  /* dropout multipliers [B,H,R,R] / [B,H,R,Lk], or NULL */
  float *qkv, *P0, *o, *r1, *mean1, *rstd1, *t1, *qc, *qk, *P1, *ctx, *sp, *o2, *r2, *mean2, *rstd2, *t2;
  float eps;
} simvg_dec_attn_args;
int simvg_dec_attn_fwd(const simvg_dec_attn_args* args, simvg_stream_t stream);

/* Backward of simvg_dec_attn_fwd (autograd of the same reference modules), two launches:
 *  simvg_dec_attn_bwd  : workgroup per sample, the forward's walk in reverse.  d(t2) = dt2 (may be NULL) + the sum of `nslab`
 *                        slabs [nslab][B*R][E] (the FFN backward's partial sums, simvg_dec_ffn_bwd).  Writes d_tgt, d_qpos [B*R, E];
 *                        the gradient of the cross-attention's source rows into dsrc [B*kv_rows, lddsrc] (fp32; written, or added to
 *                        when dsrc_accumulate: the image memory is shared by the decoder's layers) -- rows of a sample that are no
 *                        keys are zeroed when writing; and, per row, the operands of the parameter gradients (dt2sum .. dqkv:
 *                        [B*R, E] each, dctx / dqk [B*R, H, E], dqkv [B*R, 3E]).
 *  simvg_dec_attn_wgrad: all 12 parameter gradients of the block (dWs, dbs, dWso, dbso, dg0, db0, dWc, dbc, dWco, dbco, dg1,
 *                        db1: WRITTEN, reference layout) as contractions of those operands over the B*R rows, fixed summation order. */
typedef struct simvg_dec_attn_bwd_args {
  int B, R, Lk, kv_rows, kv_off;
  const float *Ws, *Wso, *g0, *Wc, *bc, *Wco, *g1;
  const void* src16; const float* src32; long ldsrc;
  const float* kpos; long ldkp; int kpos_rows;
  const float* dm0; const float* dm1;
  const float *qkv, *P0, *r1, *mean1, *rstd1, *qk, *P1, *r2, *mean2, *rstd2;      /* saved by simvg_dec_attn_fwd */
  const float* dt2; const float* dt2_slabs; int nslab; long slab_stride;
  float* d_tgt; float* d_qpos;
  float* dsrc; long lddsrc; int dsrc_accumulate;
  float *dt2sum, *gx2, *d_r2, *d_o2, *dctx, *dqk, *dqpre, *d_t1, *gx1, *d_r1, *dqkv;
} simvg_dec_attn_bwd_args;
int simvg_dec_attn_bwd(const simvg_dec_attn_bwd_args* args, simvg_stream_t stream);
typedef struct simvg_dec_attn_wgrad_args {
  int MR;                                                                          /* B * R rows */
  const float *tgt, *qpos, *t1, *o, *o2, *ctx, *sp, *qc;                           /* forward (inputs + saved) */
  const float *dqkv, *d_r1, *gx1, *d_t1, *dqpre, *dqk, *d_o2, *d_r2, *gx2, *dt2sum; /* left by simvg_dec_attn_bwd */
  float *dWs, *dbs, *dWso, *dbso, *dg0, *db0, *dWc, *dbc, *dWco, *dbco, *dg1, *db1;
} simvg_dec_attn_wgrad_args;
int simvg_dec_attn_wgrad(const simvg_dec_attn_wgrad_args* args, simvg_stream_t stream);
/* the largest number of keys per sample the two kernels above hold in LDS */
int simvg_dec_attn_max_keys(void);

/* ---- a decoder layer's FFN split over its hidden units (round 5) ---------------------------------------------------------
 * detrex FFN + norm as configured at transformer.py:118-131: r3 = t2 + dropout(relu(t2 W1^T + b1) dropout W2^T + b2),
 * t3 = LayerNorm(r3), and the decoder's shared post-norm hs = LayerNorm(t3) (transformer.py:176-183).  Workgroup s owns hidden
 * units [64 s, 64 s + 64) for all M rows:
 *   simvg_dec_ffn_fwd    : h1d[:, slice] = relu(.) * m1 (saved), slabs[s] = h_s W2[:, slice]^T      (Fd / 64 workgroups)
 *   simvg_dec_ffn_finish : r3 = t2 + m2 * (sum_s slabs[s] + b2), t3, mean3, rstd3 (+ hs, meanP, rstdP when gP != NULL)
 *   simvg_dec_ffn_bwd    : from d_t3 and / or d_hs: d_r3 (+ the row operands gx3, dy3, gxP, dr3m), slabs[s] = d(h_s) W1_s --
 *                          d(t2) = d_r3 + sum_s slabs[s], which simvg_dec_attn_bwd forms --, dW1, db1, dW2 (written whole, fixed
 *                          summation order) and, in a second small launch, db2, dg2, db2n, dgP, dbP (column sums over the rows).
 * m1 [M, Fd], m2 [M, E]: dropout multipliers or NULL.  E = 256, Fd a multiple of 64. */
typedef struct simvg_dec_ffn_args { int M, Fd; const float *t2, *W1, *b1, *W2, *m1; float* h1d; float* slabs; } simvg_dec_ffn_args;
int simvg_dec_ffn_fwd(const simvg_dec_ffn_args* args, simvg_stream_t stream);
typedef struct simvg_dec_ffn_finish_args {
  int M, NS;
  const float *t2, *slabs, *b2, *m2, *g2, *b2n, *gP, *bP;
  float *r3, *mean3, *rstd3, *t3, *hs, *meanP, *rstdP;
  float eps;
} simvg_dec_ffn_finish_args;
int simvg_dec_ffn_finish(const simvg_dec_ffn_finish_args* args, simvg_stream_t stream);
typedef struct simvg_dec_ffn_bwd_args {
  int M, Fd;
  const float *d_t3, *d_hs;                                            /* either may be NULL */
  const float *r3, *mean3, *rstd3, *g2, *t3, *meanP, *rstdP, *gP, *m2;
  const float *W1, *W2, *h1d, *m1, *t2;
  float *d_r3, *gx3, *dy3, *gxP, *dr3m;                                /* [M, E] each */
  float* slabs;                                                        /* [Fd / 64][M][E] */
  float *dW1, *db1, *dW2;
  float *db2, *dg2, *db2n, *dgP, *dbP;
} simvg_dec_ffn_bwd_args;
int simvg_dec_ffn_bwd(const simvg_dec_ffn_bwd_args* args, simvg_stream_t stream);

/* ---- query assembly around the TGQG layers (tgqs_kd_detr_head.py:385-411), E = 256 ------------------------------------------
 * text_filt: has_pad[b] = any(mask[b] == 1); filt[b] = has_pad ? max(text[b, T-1], text[b, T-2]) : text[b, T-1] -- what the
 *   reference's `text_feat.masked_fill(~mask, -inf).max(1)` selects when `mask` is int64 (quirk Q1) --; kpm[b][t] = mask[b][t] != 0.
 *   The backward writes the whole d(text) [B*T, E] (zeros except the two rows; a tie splits the gradient, as torch.maximum does).
 * query_mix: query_embed[b, q] = g[b, q] + filt[b] + qe[q]; tok[b, q] = query_embed[b, q] + cls[b] (Q5); the backward takes the
 *   gradients of both outputs (either may be NULL) and writes d(g) [B*R, E], d(filt) [B, E], d(cls) [B, E], d(qe) [R, E]. */
int simvg_text_filt_fwd(const float* text, const long long* mask, float* filt, unsigned char* kpm, int B, int T, simvg_stream_t stream);
int simvg_text_filt_bwd(const float* text, const long long* mask, const float* dfilt, float* dtext, int B, int T, simvg_stream_t stream);
int simvg_query_mix_fwd(const float* g, const float* filt, const float* qe, const float* cls, float* query_embed, float* tok, int B, int R,
                        simvg_stream_t stream);
int simvg_query_mix_bwd(const float* d_query_embed, const float* d_tok, float* dg, float* dfilt, float* dcls, float* dqe, int B, int R,
                        simvg_stream_t stream);

/* exact-fp32 forward pieces (precision="fp32" inference mode: the reference computes in fp32, use_fp16=False in all
 * 53 configs): fp32 im2col and an fp32 encoder attention with the same modality-major row layout as simvg_attn_fwd. */
int simvg_im2col_f32(const float* img_nchw, float* cols, int B, int S, int P, simvg_stream_t stream);
int simvg_attn_f32_fwd(const float* qkv, int ldqkv, float* out, int ldo, const unsigned char* pad, int B, int H, int Nv,
                       int Nt, int D, float scale, simvg_stream_t stream);
/* exact-fp32 attention backward: dqkv[:, 0:D] (dQ) is written, dqkv[:, D:3D] (dK, dV) is ACCUMULATED (zero it first);
 * autograd of the reference's torchscale MultiheadAttention core in fp32. */
int simvg_attn_f32_bwd(const float* qkv, int ldqkv, const float* dout, int lddo, float* dqkv, int lddqkv,
                       const unsigned char* pad, int B, int H, int Nv, int Nt, int D, float scale, simvg_stream_t stream);
/* exact-erf GELU, elementwise fp32: out = gelu(u) (dy == NULL) or out = dy * gelu'(u)  (F.gelu of
 * torchscale FeedForwardNetwork in the exact mode, where pre-activation and activation are both kept in fp32) */
int simvg_gelu_f32(const float* u, const float* dy_or_null, float* out, long n, simvg_stream_t stream);

/* ---- matcher + criterion (no host synchronisation) ------------------------------------------------------
 * detrex HungarianMatcher (ce_cost; cost_class 1, cost_bbox 5, cost_giou 2 at tgqs_kd_detr_head.py:132-137) with the
 * LSAP solved on the device instead of SciPy on the host; prepare_soft_targets (tgqs_kd_detr_head.py:207-268,
 * score_iou_weighted); SetCriterion.forward + calc_loss weighting (core/criterion/criterion.py:108-271,
 * tgqs_kd_detr_head.py:340-350,484-507) with analytic gradients.  coef_mode: 0 coef, 1 coef*(1-w), 2 coef*w. */
int simvg_match(const float* logits, const float* boxes, const float* tboxes, const int* tlabels, const int* tcount,
                int* match, int L, int B, int nq, int TM, float cost_class, float cost_bbox, float cost_giou,
                simvg_stream_t stream);
int simvg_soft_targets(const float* logits, const float* boxes, const int* match, const float* tboxes, const int* tcount,
                       float* pboxes, int* plabels, int* pcount, float* pweight, float* scalars4, int B, int nq, int TM,
                       simvg_stream_t stream);
/* GT packing of prepare_soft_targets (tgqs_kd_detr_head.py:215-234: pixel xyxy / (w,h,w,h) -> cxcywh rows of the [B, max_targets, 4]
 * target array, zero elsewhere; per-image counts).  table (device memory, uploaded by the host in one copy): n_rows records
 * {const float* src (device pointer to 4 floats, or NULL: use box), float box[4], float w, float h, int dst_row, int pad} (40 bytes)
 * followed by B int32 counts. */
int simvg_pack_targets(const void* table, int n_rows, float* boxes, int* count, int B, int max_targets, simvg_stream_t stream);
int simvg_criterion(const float* logits, const float* boxes, const int* match, const float* tboxes, const int* tlabels,
                    const float* num_boxes, const float* weights_distill, float* dlogits, float* dboxes, float* out,
                    int L, int B, int nq, int TM, int coef_mode, float coef, float eos_coef, float w_class, float w_bbox,
                    float w_giou, simvg_stream_t stream);
/* head.inference (tgqs_kd_detr_head.py:577-604: softmax, drop the no-object column, cxcywh -> xyxy * (w,h,w,h)) +
 * detectron2 detector_postprocess (clip to the image, Boxes.nonempty) + the per-image selection of
 * MIXDETRMB.get_predictions (mix_detr_mb.py:127-159: best kept query, box / scale_factor when rescaling) in one launch.
 * wh [B,4] = (w,h,w,h) per image; scale_factor [B,4] or null; outputs: scores / labels / keep [B,nq], xyxy [B,nq,4]
 * (clipped, rescaled), best_box [B,4], best_label [B]. */
int simvg_postprocess(const float* logits, const float* boxes, const float* wh, const float* scale_factor, float* scores,
                      long long* labels, float* xyxy, unsigned char* keep, float* best_box, long long* best_label, int B,
                      int num_queries, int num_cols, simvg_stream_t stream);

/* ---- embedding stage -------------------------------------------------------------------------------
 * torchscale VisionEmbedding / TextEmbedding / PositionalEmbedding as wired by BEiT3.forward and
 * Encoder.forward_embedding (beit3_base.py:461-475,317-334) and the pad zeroing at :367. */
int simvg_im2col(const float* img_nchw, void* cols_lp, int B, int S, int P, simvg_stream_t stream);
int simvg_embed_fwd(const float* patch, int ldp, const float* cls, const float* posA, const float* posB,
                    const float* text_embed, const long long* ids, const unsigned char* pad, float* x, int ldx,
                    int B, int np, int T, int D, simvg_stream_t stream);
int simvg_embed_bwd(const float* dx, int lddx, void* dpatch_lp, int lddp, float* dcls, float* dposA, float* dposB,
                    float* dtext, const long long* ids, const unsigned char* pad, int B, int np, int T, int D,
                    float param_scale /* on dcls / dposA / dposB / dtext; dpatch keeps the scale of dx */,
                    simvg_stream_t stream);

/* ---- weight preparation (fp32 master -> 16-bit compute copies, plain + transposed) ---------------- */
typedef struct {
  const float* src; void* dst_lp; void* dst_t_lp; int rows, cols; int tile_start;
  int split_shift;   /* > 0: dst_lp rows are 2 * cols long, [lo * 2^shift | hi] (the operand of simvg_gemm_nt_split); 0: plain */
} simvg_weight_desc;
int simvg_weight_prep(const void* descs_dev, int n_desc, int total_tiles, simvg_stream_t stream);
int simvg_cast_f32_to_lp(const float* src, void* dst_lp, long n, float scale, simvg_stream_t stream);
int simvg_cast_lp_to_f32(const void* src_lp, float* dst, long n, simvg_stream_t stream);

/* ---- device-side image pre-processing (csrc/preprocess.hip) --------------------------------------------------------
 * Replaces the pixel work of simvg/datasets/pipelines/transforms.py (LargeScaleJitter :221-342 -> mmcv.imrescale + crop,
 * Resize :59-80 -> mmcv.imresize, Normalize :146-158 -> mmcv.imnormalize, Pad :193-205 -> mmcv.impad_to_multiple) and the
 * HWC->CHW transpose of formatting.py:62-70, on interleaved 8-bit BGR images resident in HBM.
 * resize_u8 produces the [out_h, out_w] window at (win_y0, win_x0) of the image resized to [full_h, full_w] with OpenCV's
 * 8-bit INTER_LINEAR arithmetic (window == full image for a plain resize; rescale-then-crop in one pass otherwise).
 * normalize_pad_u8 writes fp32 planes [3, pad_h, pad_w]: (x - mean) * (1 / std), optional BGR->RGB, zeros outside. */
int simvg_resize_u8(const void* src_hwc, int src_h, int src_w, long src_row_bytes, void* dst_hwc, long dst_row_bytes,
                    int out_h, int out_w, int full_h, int full_w, int win_y0, int win_x0, simvg_stream_t stream);
int simvg_normalize_pad_u8(const void* src_hwc, long src_row_bytes, int h, int w, float* dst_chw, int pad_h, int pad_w,
                           const float* mean3_host, const float* std3_host, int to_rgb, simvg_stream_t stream);
/* Batched forms: ONE launch for up to SIMVG_PREPROCESS_MAX_JOBS frames of different geometry (a loader issues two resizes
 * and one format pass per frame: 3 launches per 32 frames instead of 96).  `jobs` is a HOST array consumed before the call
 * returns; same arithmetic per job as the single-frame entry points. */
#define SIMVG_PREPROCESS_MAX_JOBS 32
typedef struct simvg_resize_job {
  const void* src; int src_h, src_w; long src_row_bytes;
  void* dst; long dst_row_bytes;
  int out_h, out_w, full_h, full_w, win_y0, win_x0;
} simvg_resize_job;
typedef struct simvg_format_job {
  const void* src; long src_row_bytes; int h, w;
  float* dst_chw; int pad_h, pad_w;
} simvg_format_job;
int simvg_resize_u8_batched(const simvg_resize_job* jobs, int count, simvg_stream_t stream);
int simvg_normalize_pad_u8_batched(const simvg_format_job* jobs, int count, const float* mean3_host, const float* std3_host,
                                   int to_rgb, simvg_stream_t stream);

/* ---- optimizer step over a flat fp32 arena (csrc/optim.hip) ------------------------------------------------------
 * Replaces torch.nn.utils.clip_grad_norm_ + torch.optim.Adam.step of apis/train.py:81-83 (core/optimizer.py:52-68)
 * for the encoder arena: out_accum += sum(x^2) (the arena's share of the global gradient norm), then ONE pass
 *   g' = g * min(1, max_norm / (*total_norm + 1e-6));  Adam (L2 weight decay, amsgrad when max_exp_avg_sq != NULL):
 *   p -= step_size * m / (sqrt(vmax) / bias_correction2_sqrt + eps),  step_size = lr / (1 - beta1^t).
 * total_norm is a DEVICE scalar (no host sync); NULL = no clipping. */
int simvg_sumsq(const float* x, long n, float* out_accum, float* partial_ws /* >= 2048 floats: per-block partial sums, added
                in a fixed order (no atomics: the norm is bit-reproducible) */, simvg_stream_t stream);
int simvg_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* max_exp_avg_sq, long n,
                    float step_size, float bias_correction2_sqrt, float beta1, float beta2, float eps, float weight_decay,
                    const float* total_norm, float max_norm, simvg_stream_t stream);

/* ---- hardware-semantics probes (tests/test_kernels_gpu.py) ---------------------------------------- */
int simvg_probe_mfma(const void* a_lp, const void* b_lp, float* out, simvg_stream_t stream);
int simvg_probe_tr16(const int* byte_addr, void* out_i16, simvg_stream_t stream);
int simvg_probe_glds(const void* src_i16, const int* perm, void* out_i16, simvg_stream_t stream);

/* ---- random multipliers of the training step (csrc/rng.hip) ---------------------------------------------------------------
 * Dropout / DropPath multipliers (0 or 1 / keep) from Philox4x32-10, element i a pure function of (seed, offset, i).
 * Replaces torchscale DropPath (reference beit3_base.py:146-151: per-sample Bernoulli(1 - p) / (1 - p)) and the nn.Dropout /
 * attention-dropout draws of the DETR decoder layers (heads/tgqs_kd_detr_head/transformer.py:106-125).  keep_seg != NULL: device
 * table of keep probabilities, element i uses keep_seg[i / seg] (one DropPath rate per encoder layer).  state != NULL: two
 * device words {epoch, ticket}, zero-initialised by the caller and owned by one stream; the epoch enters the counter and is
 * incremented by the launch itself -- for launches recorded into a hipGraph, whose scalar arguments are frozen. */
int simvg_dropout_mult(float* out, long n, float keep, const float* keep_seg, long seg, unsigned long long seed,
                       unsigned long long offset, unsigned long long* state, simvg_stream_t stream);
/* the same generator evaluated on the host: out[4] = Philox4x32-10(ctr[4], key[2]) (known-answer tests) */
int simvg_philox4x32(const unsigned* ctr, const unsigned* key, unsigned* out);

#ifdef __cplusplus
}
#endif
#endif /* SIMVG_HIP_H */
