/* frameino_hip.h -- C ABI of libframeino_hip.so: the MI355X (gfx950) kernels behind FrameINO's
 * denoising hot path (Wan2.2 / CogVideoX DiT forward, sampler glue, Wan VAE decode).
 *
 * Contract:
 *  - plain pointers + sizes, no tensor objects; the CALLER owns every buffer (inputs, outputs, workspace);
 *  - every call only validates arguments on the host and enqueues kernels on `stream` (a hipStream_t passed
 *    as void*): no allocation, no host sync, no host read of device memory => hipGraph-capturable;
 *  - return 0 on success, <0 on error; fino_last_error() gives the thread-local message;
 *  - `dtype` selects the storage/MFMA-operand type of activations and weights (FINO_BF16 / FINO_F16);
 *    accumulation and the "fp32 islands" of the reference are fp32 inside the kernels.
 *
 * Each entry point cites the reference call site(s) it replaces (paths under /root/reference).
 * The reference itself has no native interface for this path (pure PyTorch), so these are the operator
 * boundaries of its attention-processor / block code.
 */
#ifndef FRAMEINO_HIP_H
#define FRAMEINO_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 101: fino_gemm_split_n / fino_gemm_blocked_a take a per-call tile_m; FINO_TUNE_GEMM_TILE_M is an A/B knob only. */
/* 102: fino_attn_fwd_fp8 takes p_mode (how a softmax weight becomes an e4m3 byte: FINO_FP8_P_EXP2 | FINO_FP8_P_RAMP);
 * fino_attn_fwd_tail / fino_attn_tail_supported, fino_attn_probs / fino_attn_probs_supported added. */
/* 103 (round 5): the Wan VAE can compute like the fp32 the reference app runs it in (app.py:157) -- fino_conv3d_split,
 * fino_split_bf16, fino_rmsnorm_silu_cl_f32, FINO_EPI_F32 / FINO_EPI_F32_RESIDUAL of fino_gemm, dtype FINO_F32 for the VAE's
 * rearrangement kernels; fino_vae_blend_tiles (CogVideoX VAE tiling). */
#define FINO_VERSION 103

/* FINO_F32: fp32 STORAGE of activations -- accepted only where a function says so (the Wan VAE's rearrangement kernels);
 * MFMA operands are always bf16 / fp16. */
enum { FINO_BF16 = 0, FINO_F16 = 1, FINO_F32 = 2 };
enum {
    FINO_OK = 0,
    FINO_ERR_ARG = -1,     /* bad shape / alignment / dtype */
    FINO_ERR_LAUNCH = -2,  /* HIP launch error */
    FINO_ERR_UNSUPPORTED = -3
};

/* FINO_VERSION for a product build.  A library compiled with -DFINO_EXPERIMENT (the only way to enable the kernels'
 * wrong-result timing-experiment switches, csrc/fino_common.h) reports -FINO_VERSION: frameino_amd._lib.load() refuses
 * it unless FINO_ALLOW_EXPERIMENT=1 is set, so such a build can never pass for the product by accident. */
int fino_version(void);
const char* fino_last_error(void);

/* Tuning knobs for A/B timing of kernel variants inside one process (tools/): results never depend on them.
 * value 0 = the built-in default.  FINO_TUNE_GEMM_GROUP_M: tile rows per raster group of the GEMM's XCD-aware tile
 * order.  FINO_TUNE_GEMM_RASTER: 2 = long-K GEMMs (K >= 8192) walk their tile rows first to last as rounds 1 - 5 did (default since round 6: last to first, what the producer wrote last is still in the Infinity Cache); A/B only: 1 = every GEMM last to first, 3 = long-K and wide-N (>= 8192) ones (both measured slower than the default: profiles/r06_ffn_pair_ab.txt).  FINO_TUNE_CONV_LOOP: 1 = the one-barrier conv loop instead of the
 * ping-pong one.  FINO_TUNE_GEMM_TILE_M: 2 .. 7 = one launch of 32 x that many rows per tile, 8 = 256-row tiles only (the round-2 behaviour).
 * FINO_TUNE_ATTN_KERNEL: 1 = the register-staged 8-wave ping-pong kernel everywhere, 2 = the 4-wave one-wave-per-SIMD kernel
 * (head_dim 128), 3 = the free-running kernel (4 waves, two workgroups per CU) everywhere, 4 = the LDS-DMA-staged 8-wave
 * ping-pong kernel everywhere (head_dim 64 too), 5 = the round-3 policy (as 0, but register-staged for Lk > 1024); 0 = the policy:
 * free-running for Lk <= 1024 at head_dim 128 (text cross-attention), LDS-DMA-staged ping-pong for Lk > 1024 at head_dim
 * 128, 4-wave for head_dim 64 with the folded scale, register-staged ping-pong otherwise.
 * 6 = the walking kernel (one workgroup per CU over a run of q-blocks) wherever it can run: head_dim 128, at least two key
 * tiles, whole blocks; by policy it serves Lk <= 1024 when there are at least two q-blocks per CU, the free-running kernel the
 * rest; 7 = the policy without the walking kernel.  FINO_TUNE_ATTN_WALK_GRID: workgroups of the walking kernel (0 = one per CU; tests use small counts to make runs of
 * blocks cross heads and batches on small shapes).
 * FINO_TUNE_ATTN_FP8_KERNEL (fino_attn_fwd_fp8): 1 = the 8-wave ping-pong kernel instead of the free-running 4-wave one. */
enum { FINO_TUNE_GEMM_GROUP_M = 0, FINO_TUNE_GEMM_RASTER = 1, FINO_TUNE_CONV_LOOP = 2, FINO_TUNE_GEMM_TILE_M = 3,
       FINO_TUNE_ATTN_KERNEL = 4, FINO_TUNE_ATTN_FP8_KERNEL = 5, FINO_TUNE_ATTN_WALK_GRID = 6, FINO_TUNE_COUNT = 8 };
int fino_tune_set(int key, int value);
int fino_tune_get(int key);

/* ---- normalisation + modulation (HBM-bound, one wave per token row) ------------------------------------
 * y = T( LN_fp32(x) * (1 + scale[r]) + shift[r] ),  LN without affine, statistics in fp32.
 * scale/shift are fp32 rows of a small modulation table: element (row, c) is  p[sel ? sel[row]*mod_stride : 0][c].
 * Replaces architecture/transformer_wan.py:334, :344-346, :536 (FP32LayerNorm + AdaLN-zero modulate); the
 * table has ONE row per distinct timestep instead of the reference's [1,L,6,D] fp32 tensor. */
int fino_adaln_modulate(const void* x, void* y, int64_t rows, int dim, int64_t ldx, int64_t ldy,
                        const float* shift, const float* scale, int64_t mod_stride, const int32_t* sel,
                        float eps, int dtype, void* stream);

/* y = T( LN_fp32(x) * w + b ), w/b fp32 (may be NULL => no affine).  transformer_wan.py:339 (norm2),
 * also nn.LayerNorm call sites of CogVideoX (cogvideox_transformer_3d.py:531-538). */
int fino_layernorm(const void* x, void* y, int64_t rows, int dim, int64_t ldx, int64_t ldy, const float* w,
                   const float* b, float eps, int dtype, void* stream);

/* out = T( float(x) + float(y) * gate[r] )  (gate fp32 table row as above; gate==NULL => out = T(x + y)).
 * transformer_wan.py:336, :341, :348. out may alias x. */
int fino_gated_residual(const void* x, const void* y, void* out, int64_t rows, int dim, int64_t ldx, int64_t ldy,
                        int64_t ldo, const float* gate, int64_t mod_stride, const int32_t* sel, int dtype,
                        void* stream);

/* out = T( x + T(y * gate[r]) ): the gate multiply rounded to T first (CogVideoX block, all-T arithmetic). */
int fino_gated_residual_staged(const void* x, const void* y, void* out, int64_t rows, int dim, int64_t ldx, int64_t ldy,
                               int64_t ldo, const float* gate, int64_t mod_stride, const int32_t* sel, int dtype,
                               void* stream);

/* y = T( T( T(LN(x)*w + b) * T(1 + scale[r]) ) + shift[r] ): diffusers CogVideoXLayerNormZero / AdaLayerNorm executed in T
 * (cogvideox_transformer_3d.py:134-136, :150-152, :541); w/b fp32 copies of the T affine parameters (NULL => none),
 * scale/shift fp32 copies of the T modulation rows selected by sel (one row per (batch, text|video)). */
int fino_layernorm_zero(const void* x, void* y, int64_t rows, int dim, int64_t ldx, int64_t ldy, const float* w,
                        const float* b, const float* shift, const float* scale, int64_t mod_stride, const int32_t* sel,
                        float eps, int dtype, void* stream);

/* The three LayerNorm forms above with the result emitted as MXFP8 activations -- the bytes fino_quantize_mxfp8 produces from
 * the T-rounded y (e4m3 to q [rows, dim], one e8m0 scale per 32 channels in that function's layout), without writing y:
 * mode 0 = fino_adaln_modulate, 1 = fino_layernorm (w / b may be NULL), 2 = fino_layernorm_zero.  dim % 128 == 0.  The input
 * of the MXFP8 linears that follow an adaLN (transformer.enable_mxfp8_linears()): one pass instead of two. */
int fino_ln_mxfp8(int mode, const void* x, void* q, void* scales, int64_t rows, int dim, int64_t ldx, const float* w,
                  const float* b, const float* shift, const float* scale, int64_t mod_stride, const int32_t* sel,
                  float eps, int dtype, void* stream);

/* In-place q/k preparation: RMSNorm over the whole row (all heads), weight multiply in T, then RoPE on adjacent
 * channel pairs with per-token tables cos/sin [rows, head_dim/2] fp32 (NULL => no RoPE).
 * transformer_wan.py:64-67 (norm_q / norm_k, "rms_norm_across_heads") and :73-90 (apply_rotary_emb).
 * weight == NULL skips the norm (RoPE only). */
int fino_rmsnorm_rope(void* x, int64_t rows, int dim, int64_t ldx, const void* weight, float eps,
                      const float* cos_t, const float* sin_t, int head_dim, int dtype, void* stream);
/* The same with the fp32 result multiplied by out_scale before its ONE rounding to T: q prepared with
 * out_scale = softmax_scale * log2(e) goes to the attention kernels with scale = FINO_ATTN_SCALE_FOLDED. */
int fino_rmsnorm_rope_scaled(void* x, int64_t rows, int dim, int64_t ldx, const void* weight, float eps,
                             const float* cos_t, const float* sin_t, int head_dim, float out_scale, int dtype,
                             void* stream);

/* The same arithmetic, out of place, with the result SCATTERED by head: head hd (channels [hd*head_dim, (hd+1)*head_dim) of a
 * row) is written to out + head_off[hd] + row * head_ld[hd] (element offsets, multiples of 8; the tables live in device
 * memory).  weight == NULL and cos_t == NULL make it a pure scattering copy (the v columns).  What the token-sharded
 * forward's heads exchange uses to write q, k, v of its tokens straight into the all-to-all send buffer [peer][token][q|k|v]
 * [heads of that peer] instead of a permute copy after the in-place kernel (frameino_amd/transformer_wan.py, heads exchange). */
int fino_rmsnorm_rope_scatter(const void* x, int64_t rows, int dim, int64_t ldx, const void* weight, float eps,
                              const float* cos_t, const float* sin_t, int head_dim, float out_scale, void* out,
                              const int64_t* head_off, const int64_t* head_ld, int dtype, void* stream);

/* q and k of a fused q | k | v projection in ONE launch: qkv is [rows, 3*dim] with row stride ldx; columns [0, dim) get
 * fino_rmsnorm_rope_scaled(q_weight, q_eps, q_out_scale), columns [dim, 2 dim) fino_rmsnorm_rope(k_weight, k_eps), both
 * bit for bit.  out == NULL: in place, v untouched.  out != NULL: nothing is modified in place; q, k AND v (a copy) are
 * scattered by head as in fino_rmsnorm_rope_scatter with head_off = [q heads | k heads | v heads] (3 * dim / head_dim
 * entries) and head_ld per head (dim / head_dim entries). */
int fino_qkv_rmsnorm_rope(void* qkv, int64_t rows, int dim, int64_t ldx, const void* q_weight, float q_eps,
                          const void* k_weight, float k_eps, const float* cos_t, const float* sin_t, int head_dim,
                          float q_out_scale, void* out, const int64_t* head_off, const int64_t* head_ld, int dtype,
                          void* stream);

/* Per-head LayerNorm(head_dim, affine) + RoPE on rows >= rope_row0 of every batch element (CogVideoX):
 * architecture/attention_processor.py:2851-2860.  x is [batch, rows, heads*head_dim] with row stride ldx.
 * w/b are T vectors of head_dim (NULL => skip LN). cos/sin are [rows - rope_row0, head_dim] fp32 (full width,
 * as the Cog pipeline builds them), NULL => no RoPE. */
int fino_headnorm_rope(void* x, int batch, int64_t rows, int heads, int head_dim, int64_t ldx, int64_t batch_stride,
                       const void* w, const void* b, float eps, const float* cos_t, const float* sin_t,
                       int64_t rope_row0, int dtype, void* stream);
/* The same with the result multiplied by out_scale before it is stored (rows that get RoPE: one rounding; rows that
 * only get the LayerNorm, i.e. the text tokens, are rounded by the norm first): q for FINO_ATTN_SCALE_FOLDED. */
int fino_headnorm_rope_scaled(void* x, int batch, int64_t rows, int heads, int head_dim, int64_t ldx,
                              int64_t batch_stride, const void* w, const void* b, float eps, const float* cos_t,
                              const float* sin_t, int64_t rope_row0, float out_scale, int dtype, void* stream);

/* ---- attention (MFMA-bound) -----------------------------------------------------------------------------
 * o[b, i, h, :] = softmax_j( scale * q[b,i,h,:].k[b,j,h,:] ) v[b,j,h,:]   non-causal, no mask, no dropout.
 * Element (b, row, h, d) of X lives at X + b*x_bs + row*x_rs + h*x_hs + d  (strides in ELEMENTS; d contiguous),
 * so q/k/v can be column slices of one fused-QKV GEMM output and o is written token-major [L, H*Dh].
 * head_dim in {64, 128}.  Replaces F.scaled_dot_product_attention at transformer_wan.py:108,
 * attention_processor.py:2863, :2934.
 * scale = FINO_ATTN_SCALE_FOLDED (in every fino_attn_* entry point): q already carries softmax_scale * log2(e)
 * (fino_rmsnorm_rope_scaled writes it that way, one rounding instead of two), so the kernels take q.k as the exp2
 * argument as it is -- and the head_dim-128 4-wave kernel folds the running maximum into its MFMAs. */
#define FINO_ATTN_SCALE_FOLDED (-1.0f)
int fino_attn_fwd(const void* q, const void* k, const void* v, void* o, int batch, int heads, int64_t lq,
                  int64_t lk, int head_dim, int64_t q_bs, int64_t q_rs, int64_t q_hs, int64_t k_bs, int64_t k_rs,
                  int64_t k_hs, int64_t v_bs, int64_t v_rs, int64_t v_hs, int64_t o_bs, int64_t o_rs,
                  int64_t o_hs, float scale, int dtype, void* stream);
/* Self-attention with fp8 (OCP e4m3) matrix operands, head_dim 64 (BASELINE config 5's "fp8 MFMA path" for the SDPA at
 * architecture/attention_processor.py:2863) or 128 (the same for transformer_wan.py:108, as two 64-channel sub-heads per
 * head): no reference counterpart, SURVEY F11 -- compared with fp32 SDPA and with this library's bf16 kernel).  q / k / v / o as fino_attn_fwd (bf16 | fp16, head stride = head_dim).  K and V are quantised
 * once per call into `kv_workspace` (fino_attn_fp8_kv_bytes: caller-owned, 16-byte aligned; e4m3 tiles + one e8m0 scale
 * per 32 elements, V transposed and key-permuted for the second product), Q in registers, P = exp2(s - m) to e4m3 with a
 * fixed 2^-6 block scale; both products on v_mfma_scale_f32_32x32x64_f8f6f4 with fp32 accumulation, softmax in fp32.
 * p_mode (an ARGUMENT, not a tune knob: it changes the bits) says how a weight becomes its operand byte:
 *   FINO_FP8_P_EXP2  p = exp2(s - m) on the transcendental unit, then rounded to e4m3 (v_exp_f32 + v_cvt_pk_fp8_f32);
 *   FINO_FP8_P_RAMP  the byte is rne(8 (s - m) + 55.5) written by one v_cvt_pk_u8_f32: e4m3's exponent field counts octaves and
 *                    its 3 mantissa bits interpolate linearly, so this is exp2 with a piecewise-linear mantissa (within +-3 % of
 *                    2^x, the same weight in numerator and denominator, running maximum in whole octaves so rescales and merges
 *                    stay exact) at a third of the vector-instruction time: the kernels are bound by exactly those instructions. */
enum { FINO_FP8_P_EXP2 = 0, FINO_FP8_P_RAMP = 1 };
int64_t fino_attn_fp8_kv_bytes(int batch, int heads, int64_t lk, int head_dim);
int fino_attn_fwd_fp8(const void* q, const void* k, const void* v, void* o, int batch, int heads, int64_t lq, int64_t lk,
                      int head_dim, int64_t q_bs, int64_t q_rs, int64_t k_bs, int64_t k_rs, int64_t v_bs, int64_t v_rs,
                      int64_t o_bs, int64_t o_rs, float scale, int dtype, int p_mode, void* kv_workspace,
                      int64_t kv_workspace_bytes, void* stream);

/* Cross-attention over key sequences whose TAIL is one row repeated -- the zero-padded prompt of
 * pipelines/pipeline_wan_i2v_motion_FrameINO.py:235-238: every padding token of the 512 the text cross-attention
 * (architecture/transformer_wan.py:108 through attn2) attends to yields the SAME K and V row.  Batch element b attends to its first
 * lk_b[b] rows of k / v (host arrays of `batch` entries, copied into the launch: nothing is read later); the last of them stands for
 * tail_mult[b] identical keys, rows from lk_b[b] up to `lk` (the rows every batch element has allocated; finite values) are ignored:
 *     softmax(q.[K; k x M]^T) [V; v x M] = softmax(q.[K; k]^T + [0; ln M]) [V; v]
 * i.e. fino_attn_fwd on the expanded sequences up to the rounding of one weight.  Walking kernel only: head_dim 128, batch <= 4,
 * lk > 64 -- fino_attn_tail_supported() says whether a shape qualifies, FINO_ERR_UNSUPPORTED otherwise. */
int fino_attn_tail_supported(int batch, int heads, int64_t lq, int64_t lk, int head_dim);
int fino_attn_fwd_tail(const void* q, const void* k, const void* v, void* o, int batch, int heads, int64_t lq, int64_t lk,
                       int head_dim, int64_t q_bs, int64_t q_rs, int64_t q_hs, int64_t k_bs, int64_t k_rs, int64_t k_hs,
                       int64_t v_bs, int64_t v_rs, int64_t v_hs, int64_t o_bs, int64_t o_rs, int64_t o_hs, float scale,
                       int dtype, const int* lk_b, const float* tail_mult, void* stream);

/* Attention PROBABILITIES over a short key sequence (head_dim 128, lk <= 128 key rows allocated per sample, batch <= 4):
 * p[b][row][head][0 .. kp) = softmax(scale q.K^T) of that head's keys (kp a multiple of 8, >= every lk_b; columns from lk_b[b] on
 * are zeros; p_bs / p_rs in elements), lk_b / tail_mult as fino_attn_fwd_tail.  For the text cross-attention of
 * architecture/transformer_wan.py:108 + :117 re-associated as P.(V W_o^T): the [rows, heads x kp] matrix is the A operand of a
 * fino_gemm whose weight is the per-prompt constant V_h W_o,h^T -- K = heads x kp instead of 3072. */
int fino_attn_probs_supported(int batch, int heads, int64_t lq, int64_t lk, int head_dim);
int fino_attn_probs(const void* q, const void* k, void* p, int batch, int heads, int64_t lq, int64_t lk, int head_dim,
                    int64_t q_bs, int64_t q_rs, int64_t q_hs, int64_t k_bs, int64_t k_rs, int64_t k_hs, int kp, int64_t p_bs,
                    int64_t p_rs, float scale, int dtype, const int* lk_b, const float* tail_mult, const float* q_rrms,
                    int64_t q_rrms_bs, const void* q_weight, void* stream);
/* q_rrms / q_weight (both or neither): q is the RAW q projection and row i of sample b is RMS-normalised while it is loaded,
 * T(T(q x q_rrms[b * q_rrms_bs + i]) x q_weight[c]) -- fino_rmsnorm_rope's arithmetic and rounding points, without the pass
 * that rewrites q.  fino_row_rrms writes that statistic: rrms[row] = 1 / sqrt(mean(x[row]^2) + eps) with the loads and the
 * summation order of fino_rmsnorm_rope (the same bits). */
int fino_row_rrms(const void* x, int64_t rows, int dim, int64_t ldx, float eps, float* rrms, int dtype, void* stream);

/* Attention over ONE key range of several, for the same queries: fino_attn_partial leaves every (head, 256-row
 * q-block)'s unnormalised O, running max m and sum l in `partial` (fp32, fino_attn_partial_bytes(B, H, Lq, Dh) bytes)
 * instead of storing O; fino_attn_merge combines up to three such partials (disjoint key ranges) into O.  Lets the
 * token-sharded DiT attend to its own K/V chunk while the other ranks' chunks are still arriving (frameino_amd/parallel.py):
 * softmax(q.[K_a; K_b]^T).[V_a; V_b] = merge(partial(K_a, V_a), partial(K_b, V_b)) up to fp32 summation order. */
int64_t fino_attn_partial_bytes(int batch, int heads, int64_t lq, int head_dim);
int fino_attn_partial(const void* q, const void* k, const void* v, int batch, int heads, int64_t lq, int64_t lk,
                      int head_dim, int64_t q_bs, int64_t q_rs, int64_t q_hs, int64_t k_bs, int64_t k_rs, int64_t k_hs,
                      int64_t v_bs, int64_t v_rs, int64_t v_hs, float scale, int dtype, void* partial,
                      int64_t partial_bytes, void* stream);
int fino_attn_merge(void* o, int batch, int heads, int64_t lq, int head_dim, int64_t o_bs, int64_t o_rs, int64_t o_hs,
                    const void* part0, const void* part1, const void* part2, int dtype, void* stream);


/* Same product with a caller-owned workspace that enables the TAIL SPLIT: one workgroup (256 query rows of one
 * head) occupies a CU, so an XCD's 32 CUs take its blocks in rounds; when the last round holds fewer blocks than
 * CUs, each of its blocks is cut along the keys into CUs/blocks ranges whose (O, m, l) partials go to the workspace
 * and are merged by a second small kernel.  fino_attn_workspace_bytes returns the size to allocate (0 = the shape
 * has no tail to split).  workspace NULL / 0 bytes => identical to fino_attn_fwd.  Results differ from the unsplit
 * kernel only by fp32 summation order in the split blocks. */
int64_t fino_attn_workspace_bytes(int batch, int heads, int64_t lq, int64_t lk, int head_dim);
int fino_attn_fwd_ws(const void* q, const void* k, const void* v, void* o, int batch, int heads, int64_t lq,
                     int64_t lk, int head_dim, int64_t q_bs, int64_t q_rs, int64_t q_hs, int64_t k_bs, int64_t k_rs,
                     int64_t k_hs, int64_t v_bs, int64_t v_rs, int64_t v_hs, int64_t o_bs, int64_t o_rs, int64_t o_hs,
                     float scale, int dtype, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- GEMM with fused epilogue (MFMA-bound) --------------------------------------------------------------
 * C[M,N] = epilogue( A[M,K] . W[N,K]^T + bias[N] ),  A/W/C/R of `dtype`, bias of `dtype` (NULL => 0), fp32 accumulate.
 * W is the nn.Linear weight as stored ([out, in], K contiguous).  lda/ldw/ldc/ldr in elements, multiples of 8.
 * The value y = T(acc + bias) is rounded to T first (the reference materialises the Linear output), then:
 *   FINO_EPI_NONE            C = y
 *   FINO_EPI_GELU_TANH       C = T(gelu_tanh(y))                       (FeedForward "gelu-approximate")
 *   FINO_EPI_RESIDUAL        C = T(R + y)                              (transformer_wan.py:341)
 *   FINO_EPI_GATED_RESIDUAL  C = T(float(R) + float(y) * gate[r][n])   (transformer_wan.py:336, :348)
 * gate rows as in fino_adaln_modulate. C may alias R.  Replaces nn.Linear at transformer_wan.py:60-62, :117,
 * diffusers FeedForward (:347), patch_embedding (:486) and proj_out (:537) after patchify.
 *   FINO_EPI_GATED_RESIDUAL_STAGED  C = T(R + T(y * gate[r][n]))    (cogvideox_transformer_3d.py:146-147, :158-159:
 *                                   the CogVideoX block does the gate multiply and the add in T)
 *   FINO_EPI_F32, FINO_EPI_F32_RESIDUAL  C = acc + bias [+ R] WITHOUT the rounding to T: bias, R and C are fp32 (ldc / ldr
 *                                   count floats, N a multiple of 4; M and K as usual).  The closing step of a split-bf16
 *                                   product -- operands expanded by fino_split_bf16 -- with which the Wan VAE computes like
 *                                   the fp32 the reference app runs it in (app.py:157; fino_conv3d_split below). */
enum { FINO_EPI_NONE = 0, FINO_EPI_GELU_TANH = 1, FINO_EPI_RESIDUAL = 2, FINO_EPI_GATED_RESIDUAL = 3,
       FINO_EPI_GATED_RESIDUAL_STAGED = 4, FINO_EPI_F32 = 5, FINO_EPI_F32_RESIDUAL = 6 };
int fino_gemm(const void* a, const void* w, const void* bias, void* c, int64_t m, int64_t n, int64_t k,
              int64_t lda, int64_t ldw, int64_t ldc, int epilogue, const void* r, int64_t ldr, const float* gate,
              int64_t mod_stride, const int32_t* sel, int dtype, void* stream);
/* fino_gemm whose output columns [n_split, N) go to a SECOND buffer c2 (leading dimension ldc2; column n_split of the
 * product = column 0 of c2), columns [0, n_split) to c as usual: the fused q | k | v projection of a token-sharded rank
 * (transformer_wan.py:60-62 as ONE GEMM) leaves q in its own buffer and k | v contiguous in the buffer the K|V all-gather
 * sends -- no copy, and one GEMM of 3 D columns instead of a 2 D and a D one.  n_split a multiple of 256, K of 64; the
 * bias / GELU epilogues only.  c2 == NULL and n_split == 0: plain fino_gemm (any epilogue).
 * tile_m (here and in fino_gemm_blocked_a): the caller's tile height for THIS call -- 0 = planned (fino_gemm_plan), 8 =
 * 256-row tiles only, 2 .. 7 = one launch of 32 x tile_m-row tiles.  Results never depend on it.  A rank that runs a second
 * kernel stream beside its GEMMs (the interleaved multi-GPU plan) passes 8: the other stream fills the CUs a partial round
 * leaves idle, lower tiles would only add operand traffic. */
int fino_gemm_split_n(const void* a, const void* w, const void* bias, void* c, int64_t m, int64_t n, int64_t k, int64_t lda,
                      int64_t ldw, int64_t ldc, int epilogue, const void* r, int64_t ldr, const float* gate,
                      int64_t mod_stride, const int32_t* sel, int dtype, void* c2, int64_t ldc2, int64_t n_split,
                      int tile_m, void* stream);
/* c = r + gate[sel] * (A w^T + bias) (FINO_EPI_GATED_RESIDUAL) with a K-BLOCKED A: K block b (columns [b * a_block_k,
 * (b + 1) * a_block_k) of row i; a_block_k a multiple of 64 dividing K), b = j * a_groups + g, lives at
 * a + g * a_group_stride + j * a_block_stride + i * lda (elements).  That is the layout in which the heads all-to-all
 * returns the attention output to the token owners -- per head group g a buffer [peer j][token][heads of that peer in the
 * group] -- so the out-projection (transformer_wan.py:336) reads it as it arrived, without the permute copy into
 * [token, D].  a_groups = 1: one buffer [peer][token][heads of that peer]. */
int fino_gemm_blocked_a(const void* a, const void* w, const void* bias, void* c, int64_t m, int64_t n, int64_t k,
                        int64_t a_block_k, int64_t a_block_stride, int a_groups, int64_t a_group_stride, int64_t lda,
                        int64_t ldw, int64_t ldc, const void* r, int64_t ldr, const float* gate, int64_t mod_stride,
                        const int32_t* sel, int dtype, int tile_m, void* stream);

/* The tiling fino_gemm uses for an M x N problem on the current device: `rows_256` leading rows run as 256 x 256 tiles
 * (a whole number of rounds of the CUs), the remaining rows as ONE more launch of `tile_rows_rest`-row tiles (64 .. 256
 * in steps of 32; 0 = no second launch) chosen so that they fit (at most) one more round: 3080 rows x 3072 columns are
 * 240 tiles of 160 rows in one round instead of 156 tiles of 256 on 256 CUs; 24640 x 3072 are 4 rounds of 256-row tiles
 * + 216 tiles of 160 rows instead of 4.55 rounds paid as 5.  Results do not depend on the tiling (every output element is
 * one fp32 dot product over K in the same order).  Diagnostics / tests. */
int fino_gemm_plan(int64_t m, int64_t n, int64_t* rows_256, int* tile_rows_rest);
/* Skinny fp32-accurate linear for the conditioning MLPs (M <= 16 rows):
 * y[m,n] = act( sum_k x[m,k] w[n,k] + b[n] ), x/y fp32, w/b fp32 (w_dtype=-1) or T; act: 0 none, 1 SiLU on the INPUT
 * (y = W.silu(x) + b).  transformer_wan.py:175-183 (time_embedder kept fp32 by :393, time_proj). */
int fino_skinny_linear(const float* x, const void* w, const void* b, float* y, int m, int64_t n, int64_t k,
                       int w_dtype, int silu_input, void* stream);

/* ---- patchify / unpatchify (HBM-bound gathers) ------------------------------------------------------------
 * patchify: x [C, F, H, W] (one batch element) -> a [L, C*pt*ph*pw], L = (F/pt)(H/ph)(W/pw), column =
 * ((c*pt+dt)*ph+dh)*pw+dw = the flattening of nn.Conv3d's weight; followed by fino_gemm it is the strided conv of
 * transformer_wan.py:424, :486-487. */
int fino_patchify(const void* x, void* a, int channels, int frames, int height, int width, int pt, int ph, int pw,
                  int64_t lda, int dtype, void* stream);
/* unpatchify: y [L, pt*ph*pw*Cout] -> out [Cout, F, H, W]   (transformer_wan.py:539-543). */
int fino_unpatchify(const void* y, void* out, int cout, int frames, int height, int width, int pt, int ph, int pw,
                    int64_t ldy, int dtype, void* stream);

/* ---- sampler glue (pipelines/pipeline_wan_i2v_motion_FrameINO.py:829-891) -------------------------------- */
/* model input: x[0:C] = T((1-m)*cond + m*lat) for the F_gen generated frames, then the n_id ID frames (:829,:854);
 * x[C:2C] = T(traj) (:858).  lat/cond/id/traj fp32; lat [C,Fg,H,W]; cond [C,1,H,W]; id [C,n_id,H,W];
 * traj [C,Fg+n_id,H,W]; m is 0 on frame 0 and 1 elsewhere (:528-532).  out [2C, Fg+n_id, H, W] of T. */
int fino_wan_model_input(const float* lat, const float* cond, const float* id_lat, const float* traj, void* out,
                         int channels, int gen_frames, int id_frames, int height, int width, int dtype, void* stream);
/* CFG + flow-match Euler update on the generated frames (:882-891):
 *   n = T(u + T(g * T(c - u)))   (the reference combines in the model dtype)   [g <= 1 or u == NULL: n = c]
 *   lat' = lat + dt * n  in fp32, then rounded to T when round_out != 0 (diffusers Euler casts to the model dtype).
 * cond_pred/uncond_pred are [C, Fg+n_id, H, W] of T (ID frames dropped, :886); lat fp32 [C, Fg, H, W], in place. */
int fino_cfg_euler_step(const void* cond_pred, const void* uncond_pred, float* lat, int channels, int gen_frames,
                        int total_frames, int height, int width, float guidance, const float* dt_dev, int round_out,
                        int dtype, void* stream);

/* CFG + UniPC (bh2, predict-x0, flow sigmas, order <= 2) multistep update -- the sampler Wan2.2-TI2V-5B-Diffusers ships
 * (diffusers UniPCMultistepScheduler, third-party; call site pipeline_wan_i2v_motion_FrameINO.py:891).  The step is
 * linear in {x, last_sample, m0, m1, v}; coef_dev = {g, sigma, use_corr, Cx, C0, C1, Ct, Px, P0, P1} (device, computed
 * on the host from the sigma schedule):  v = CFG;  m_t = x - T(sigma*v);  x_c = use_corr ? Cx*last+C0*m0+C1*m1+Ct*m_t : x;
 * x' = Px*x_c + P0*m_t + P1*m0;  then last<-x_c, m1<-m0, m0<-m_t, x<-x'.  x/last/m0/m1 fp32 [C,Fg,H,W] in place. */
int fino_cfg_unipc_step(const void* cond_pred, const void* uncond_pred, float* x, float* last, float* m0, float* m1,
                        int channels, int gen_frames, int total_frames, int height, int width, const float* coef_dev,
                        int dtype, void* stream);

/* CogVideoX sampler step (pipelines/pipeline_cogvideox_i2v_motion_FrameINO.py:893-927, v-prediction DDIM):
 *   v = u + g*(c-u) in fp32;  x0 = T(sa*x) - sb*v;  x' = T(T(ca*x) + cb*x0);  coef_dev = {sa, sb, ca, cb, g} (device).
 * pred [2, batch_stride] of T (uncond, cond; the first n_lat elements of each are the generated frames -- ID frames
 * come after, :901-902), lat [n_lat] of T, updated in place.  has_uncond == 0: v = pred[0]. */
int fino_cfg_vpred_step(const void* pred, void* lat, int64_t n_lat, int64_t batch_stride, const float* coef_dev,
                        int has_uncond, int dtype, void* stream);

/* CogVideoX SDE-DPM-Solver++ step (diffusers CogVideoXDPMScheduler, third-party; call site
 * pipelines/pipeline_cogvideox_i2v_motion_FrameINO.py:915-926; the scheduler the reference's validation builds,
 * train_code/train_cogvideox_motion_FrameINO.py:692), fused with CFG:
 *   v = u + g*(c-u) (fp32);  x0 = T(sa*x) - sb*v;  d = use_old ? m3*x0 - m4*x0_old : x0;
 *   x' = T(T(m1*x) - m2*d + T(mn*noise));  x0_old <- x0.
 * coef_dev = {sa, sb, m1, m2, m3, m4, mn, g, use_old} (device).  pred / lat as fino_cfg_vpred_step; x0_old fp32 [n_lat]
 * (in/out), noise [n_lat] of T: the step's standard-normal draw (the caller owns the RNG stream). */
int fino_cfg_dpm_step(const void* pred, void* lat, float* x0_old, const void* noise, int64_t n_lat, int64_t batch_stride,
                      const float* coef_dev, int has_uncond, int dtype, void* stream);

/* ---- Wan 3D causal VAE (architecture/autoencoder_kl_wan.py) ------------------------------------------------------
 * Activations are channels-last [T, H, W, Cpad] of `dtype`, Cpad = channels zero-padded to a multiple of 64.
 * fino_conv3d: implicit-GEMM convolution (MFMA-bound) -- WanCausalConv3d (:134-176), the Conv2d of WanResample
 * (:244-260) and its nearest-exact 2x upsample (:205-217, fused into the gather when upsample2x != 0).
 *   y[t,h,w,o] = bias[o] + sum_{dt,dh,dw,c} w[o][((dt*kh+dh)*kw+dw)*c_in_pad + c] *
 *                x[t*st+dt-pt, (h*sh+dh-ph) >> up, (w*sw+dw-pw) >> up, c]     (taps outside the input read zeros)
 * Causal convs pass pt = kt-1 (zeros in FRONT of the whole sequence -- the reference's feat_cache streaming is only a
 * schedule of the same causal conv).  epilogue: FINO_EPI_NONE or FINO_EPI_RESIDUAL (y += r, r like y; ResBlock :382).
 * zero_page: >= 128 bytes of zeros in device memory. */
int fino_conv3d(const void* x, const void* w, const void* bias, void* y, int t_in, int h_in, int w_in, int c_in_pad,
                int t_out, int h_out, int w_out, int c_out_pad, int kt, int kh, int kw, int st, int sh, int sw, int pt,
                int ph, int pw, int upsample2x, int epilogue, const void* r, const void* zero_page, int dtype,
                void* stream);
/* WanRMS_norm (:179-202) [+ SiLU]: y = act(x / max(||x||_2, 1e-12) * sqrt(c_valid) * gamma), gamma fp32 [c_pad]. */
int fino_rmsnorm_silu_cl(const void* x, void* y, int64_t rows, int c_valid, int c_pad, const float* gamma, int silu,
                         int dtype, void* stream);
/* in-place softmax over the first n columns of each row of s [rows, ld] of softmax(scale * s) (mid-block attention
 * :402-427: one head of dim C, computed as S = Q.K^T -> softmax -> P.V with the GEMM kernel). */
int fino_softmax_rows(void* s, int64_t rows, int n, int64_t ld, float scale, int dtype, void* stream);
/* out = main + DupUp3D(x) (:90-131; frame 0 keeps only its last temporal copy = the reference's first_chunk). */
int fino_dup_up3d_add(const void* main_in, const void* x, void* out, int t_in, int h_in, int w_in, int c_in,
                      int c_in_pad, int c_out, int c_out_pad, int factor_t, int factor_s, int dtype, void* stream);
/* out = main + AvgDown3D(x) (:37-87; one zero frame in front when t_in is not a multiple of factor_t). */
int fino_avg_down3d_add(const void* main_in, const void* x, void* out, int t_in, int h_in, int w_in, int c_in,
                        int c_in_pad, int c_out, int c_out_pad, int factor_t, int factor_s, int dtype, void* stream);
/* decoder tail: unpatchify (:935-952) + clamp(-1,1) (:1221): y [T,H,W,c_pad] -> out fp32 [channels, T, H*p, W*p]. */
int fino_vae_unpatchify_clamp(const void* y, float* out, int t, int h, int w, int c_pad, int channels, int patch,
                              int dtype, void* stream);
/* encoder head: patchify (:912-932): x fp32 [channels, T, H*p, W*p] -> y [T,H,W,c_pad]. */
int fino_vae_patchify(const float* x, void* y, int t, int h, int w, int c_pad, int channels, int patch, int dtype,
                      void* stream);
/* fino_softmax_rows, fino_dup_up3d_add, fino_avg_down3d_add, fino_vae_unpatchify_clamp and fino_vae_patchify also take
 * dtype = FINO_F32: the activations are fp32 [T, H, W, c_pad] (same index arithmetic, no rounding to 16 bits).
 *
 * ---- the same VAE computing like fp32 (reference app.py:157 loads AutoencoderKLWan in fp32 and decodes in fp32,
 * architecture/autoencoder_kl_wan.py:1198-1227) on the bf16 matrix pipe: SPLIT-BF16 PRODUCTS.  An fp32 value is the sum of
 * three bf16 planes, hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid) (3 x 8 significant bits, every difference exact
 * in fp32); a product of two such sums truncated to the terms >= 2^-16 relative -- (hi,hi) (hi,mid) (hi,lo) (mid,hi) (mid,mid)
 * (lo,hi) -- is accumulated exactly in the MFMA's fp32 accumulator.  Six bf16 MFMAs per product term: 1/6 of the bf16 rate,
 * against 1/16 for the fp32-input MFMA (v_mfma_f32_32x32x2_f32 runs at the vector rate).  Two planes ((hi,hi) (hi,lo)
 * (lo,hi), 2^-16 relative) cost 1/3.
 *
 * fino_split_bf16: out[row][s * cols + c] = plane p_s of x[row][c] for the nseg <= 6 segments s, p_s = (planes_packed >>
 *   2 s) & 3 in {0: hi, 1: mid, 2: lo}; x fp32 [rows, cols] (row stride ldx), out bf16 (row stride ldo >= nseg * cols).  The
 *   layouts in use: the A operand of fino_conv3d_split = the planes side by side (nseg = planes, p = 0, 1[, 2]); the operands of a
 *   plain fino_gemm with FINO_EPI_F32 = one plane per product, A: (0,0,0,1,1,2) / (0,0,1), W: (0,1,2,0,1,0) / (0,1,0).
 * fino_rmsnorm_silu_cl_f32: fino_rmsnorm_silu_cl on fp32 activations, computed in fp32 in the reference's order (x / max(||x||,
 *   1e-12) * sqrt(C) * gamma, SiLU by an IEEE division); y = fp32 [rows, c_pad] (nseg = 0) or the split planes of
 *   fino_split_bf16 (bf16 [rows, nseg * c_pad]) in the same pass.
 * fino_conv3d_split: fino_conv3d on split operands.  x_planes bf16 [T, H, W, planes * c_in_pad] (planes = 2 | 3 side by side),
 *   w_products bf16 [c_out_pad][tap][product][c_in_pad] with the W plane of product s = (0,1,2,0,1,0)[s] / (0,1,0)[s] of the fp32
 *   weight; bias, y and r fp32; the main loop's K walk visits a tap's products in that order and reads A plane (0,0,0,1,1,2)[s]
 *   / (0,0,1)[s].  epilogue FINO_EPI_NONE | FINO_EPI_RESIDUAL (y += r).  Returns FINO_ERR_UNSUPPORTED when one 256-row
 *   tile's input span exceeds 2 GiB (run the layer on fewer frames at a time). */
int fino_split_bf16(const float* x, void* out, int64_t rows, int cols, int64_t ldx, int64_t ldo, int nseg,
                    unsigned planes_packed, void* stream);
int fino_rmsnorm_silu_cl_f32(const float* x, void* y, int64_t rows, int c_valid, int c_pad, const float* gamma, int silu,
                             int nseg, unsigned planes_packed, void* stream);
int fino_conv3d_split(const void* x_planes, const void* w_products, const float* bias, float* y, int t_in, int h_in, int w_in,
                      int c_in_pad, int planes, int t_out, int h_out, int w_out, int c_out_pad, int kt, int kh, int kw, int st,
                      int sh, int sw, int pt, int ph, int pw, int upsample2x, int epilogue, const float* r,
                      const void* zero_page, void* stream);

/* ---- MXFP8 linear layers (BASELINE config 5: "fp8 MFMA path"; no reference counterpart) -------------
 * OCP e4m3 elements with one e8m0 scale per 32 consecutive K-elements (MX), fp32 accumulate
 * (v_mfma_scale_f32_16x16x128_f8f6f4, twice the bf16 MFMA rate), same fused epilogues and output types as fino_gemm.
 * fino_quantize_mxfp8: x [rows, cols] bf16|fp16 (row stride ldx, cols % 128 == 0) -> q [rows, cols] bytes and
 *   `scales`, fino_mxfp8_scale_bytes(rows, cols) bytes laid out [cols/128][ceil(rows/256)][1024], inside a KiB
 *   [K-block (4)][row & 15][row >> 4] (what one 256-row tile needs for one 128-wide K-tile, contiguous).
 *   Scale of a block = 2^e with amax / 2^e in [224, 448].  Quantise weights once, activations per call.
 * fino_gemm_mxfp8: C[M, N] = epilogue(dequant(Aq) . dequant(Wq)^T + bias); Aq [M, K], Wq [N, K] bytes; K % 128 == 0;
 *   bias / residual / C in out_dtype, epilogue arguments as fino_gemm. */
int64_t fino_mxfp8_scale_bytes(int64_t rows, int64_t cols);
int fino_quantize_mxfp8(const void* x, void* q, void* scales, int64_t rows, int64_t cols, int64_t ldx, int dtype,
                        void* stream);
int fino_gemm_mxfp8(const void* aq, const void* a_scales, const void* wq, const void* w_scales, const void* bias,
                    void* c, int64_t m, int64_t n, int64_t k, int64_t ldc, int epilogue, const void* r, int64_t ldr,
                    const float* gate, int64_t mod_stride, const int32_t* sel, int out_dtype, void* stream);
/* Same product with the result QUANTISED in the epilogue (epilogue NONE or GELU_TANH): cq [M, N] e4m3 bytes + c_scales
 * in the fino_quantize_mxfp8 layout, byte-identical to fino_quantize_mxfp8 applied to the bias_dtype-rounded C.  Feeds
 * the next MXFP8 GEMM (FFN up -> FFN down) without the bf16 round trip through HBM.  N % 128 == 0. */
int fino_gemm_mxfp8_q(const void* aq, const void* a_scales, const void* wq, const void* w_scales, const void* bias,
                      void* cq, void* c_scales, int64_t m, int64_t n, int64_t k, int epilogue, int bias_dtype,
                      void* stream);

/* ---- condition builders in front of the path ---------------------------------------------------------
 * Trajectory video of data_loader/video_dataset_motion.py:120-206 (`prepare_traj_tensor`, app.py:616-620).
 * fino_traj_paint: canvas fp32 [frames, 3, H, W] = 255, then for every frame its points, in order, paint the square
 *   [y-r, y+r) x [x-r, x+r) clipped to the canvas with their colour (:146-160; points outside the canvas skipped).
 *   points int32 [n, 4] = (x, y, r | g<<8 | b<<16, 0) sorted by frame; frame_offsets int32 [frames+1] (CSR).
 * fino_traj_blur_quantize: separable `taps`-tap blur with BORDER_REFLECT_101 (the 45x45 isotropic Gaussian of :29 is
 *   the outer product of its normalised 1-D factor; cv2.filter2D :172 is third-party and absent offline), truncation
 *   to uint8 (numpy astype :172) and x/255*2-1 (:39-42).  canvas/scratch/out fp32 [planes, H, W]. */
int fino_traj_paint(const int32_t* points, const int32_t* frame_offsets, float* canvas, int frames, int height, int width,
                    int radius, void* stream);
int fino_traj_blur_quantize(const float* canvas, float* scratch, float* out, const float* taps_dev, int taps, int planes,
                            int height, int width, void* stream);


/* ---- CogVideoX 3D causal VAE (diffusers AutoencoderKLCogVideoX, third-party; call sites
 * pipelines/pipeline_cogvideox_i2v_motion_FrameINO.py:380-396 encode, :426-431 decode, :809-826 condition encodes) ----
 * Channels-last activations [T, H, W, Cpad] like the Wan VAE; convolutions are fino_conv3d.
 * fino_groupnorm_cl: nn.GroupNorm(groups, channels, eps) over one frame batch (statistics per group over
 *   C/G x T x H x W, fp32 partials + fp64 finalisation, deterministic), y = T(gn(x)); with mod_scale / mod_shift
 *   (CogVideoXSpatialNorm3D: the conv_y / conv_b outputs at LATENT resolution [tz, hz, wz, Cpad] of T) it continues
 *   y = T(T(y * scale[z]) + shift[z]) through F.interpolate's nearest index map (first frame of an odd-length batch
 *   mapped on its own); silu != 0 appends T(silu(y)).  workspace: fino_groupnorm_workspace_bytes(c_pad) bytes.
 * fino_avg_pool_time2: CogVideoXDownsample3D's temporal compression: frame pairs averaged, the first frame of an
 *   odd-length batch kept -> [t_in/2 (+1), H, W, Cpad].
 * fino_vae_blend_tiles: diffusers' blend_v (axis 0) / blend_h (axis 1) of AutoencoderKLCogVideoX.tiled_encode / tiled_decode
 *   (`vae.enable_tiling()`, reference test_code/run_cogvideox_FrameIn_mass_evaluation.py:95-96; the loops are those of the
 *   in-tree architecture/autoencoder_kl_wan.py:1254-1268), in place on tile b [t, h_b, w_b, c_pad] from its upper / left
 *   neighbour a [t, h_a, w_a, c_pad]:  b[y] = T(T(a[n_a - e + y] * (1 - y / e)) + T(b[y] * (y / e))), e = min(extent, n_a, n_b). */
int64_t fino_groupnorm_workspace_bytes(int c_pad);
int fino_groupnorm_cl(const void* x, void* y, int t, int h, int w, int channels, int c_pad, int groups,
                      const float* gamma, const float* beta, float eps, const void* mod_scale, const void* mod_shift,
                      int tz, int hz, int wz, int silu, void* workspace, int64_t workspace_bytes, int dtype,
                      void* stream);
int fino_avg_pool_time2(const void* x, void* y, int t_in, int h, int w, int c_pad, int dtype, void* stream);
int fino_vae_blend_tiles(const void* a, void* b, int t, int h_a, int w_a, int h_b, int w_b, int c_pad, int extent, int axis,
                         int dtype, void* stream);

/* Canvas / identity-reference builders of app.py (:270-350 build_canvas, :634-695 ID padding), on the device.
 * fino_resize_area_pad_u8: pixel-area resampling (cv2.INTER_AREA's area relation; OpenCV is third-party and absent
 *   offline: parity unpinned) of src uint8 [src_h, src_w, 3] to region_h x region_w, written at (off_y, off_x) of dst
 *   uint8 [out_h, out_w, 3]; the rest of dst = fill.  build_canvas: region = resized size, offsets = top-left pads,
 *   fill 0 (:303, :309, :322-326); ID reference: region = scaled size, centred, fill 0 (:662-681).
 * fino_u8_hwc_to_chw_unit: uint8 [H, W, 3] -> fp32 [3, H, W] = v / 255 * 2 - 1 (:109-113, :692). */
int fino_resize_area_pad_u8(const void* src, void* dst, int src_h, int src_w, int region_h, int region_w, int out_h,
                            int out_w, int off_y, int off_x, int fill, void* stream);
int fino_u8_hwc_to_chw_unit(const void* src, float* dst, int height, int width, void* stream);

/* ---- diagnostics (tools/ only) --------------------------------------------------------------------------------
 * Dense MFMA rate with nothing else running: `iters` x 16 independent 32x32x16 (kind 0) / 32 independent 16x16x32
 * (kind 1) bf16 MFMAs per wave from registers -- or (kind 2) 16 block-scaled fp8 MFMAs 32x32x64 on e4m3 operands with unit scales,
 * whose 2 x 256 x 32 operand bytes follow the bf16 ones in `scratch`; (kind 3) kind 0's loop on fp16 operands (v_mfma_f32_32x32x16_f16: the
 * same operand bytes read as fp16) --, waves_per_simd in {1, 2} on every SIMD of the device.  *flops = FLOPs
 * launched; time it with events.  What the board sustains under its power cap -- the ceiling of every MFMA-bound
 * kernel here. */
int fino_diag_mfma_peak(int kind, int waves_per_simd, int iters, void* scratch, double* flops, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FRAMEINO_HIP_H */
