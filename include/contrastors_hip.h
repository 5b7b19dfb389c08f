/* contrastors_hip.h -- C-ABI of libcontrastors_hip.so (gfx950 / MI355X only).
 *
 * This is the drop-in boundary of SURVEY.md §8(b): the reference (nomic-ai/contrastors) reaches its native
 * hot path through the python symbol surface of the third-party `flash_attn` package; every entry point
 * below is what that surface binds for ONE op, with the reference call site it replaces.  Rules:
 *   - plain pointers + sizes only; all pointers are DEVICE pointers unless the name ends in `_host`;
 *   - bf16 tensors are `uint16_t*` (raw bfloat16 bits), row-major, leading dimension in ELEMENTS;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); every call only enqueues work;
 *   - nothing allocates or frees: outputs and scratch are caller-owned (torch caching allocator upstream);
 *   - return value: CX_OK (0) or a negative CX_ERR_* code; the library never throws and never exits.
 * Reference paths are relative to /root/reference/src/contrastors (abbreviated sc/).
 */
#ifndef CONTRASTORS_HIP_H
#define CONTRASTORS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CX_OK 0
#define CX_ERR_SHAPE (-1)  /* unsupported shape (e.g. K % 64 != 0, head_dim != 64) */
#define CX_ERR_ALIGN (-2)  /* pointer / leading dimension not aligned as required */
#define CX_ERR_ARG (-3)    /* invalid enum / null pointer */
#define CX_ERR_LAUNCH (-4) /* hipGetLastError() != hipSuccess after the launch */

/* ---- library info -------------------------------------------------------------------------------- */
int cx_abi_version(void);          /* bumped on any signature change */
const char* cx_build_info(void);   /* "gfx950 <date> <compiler>" */
const char* cx_error_string(int code);

/* ---- K9  FusedDense  (flash_attn.ops.fused_dense.FusedDense; sc/layers/attention.py:82-85,112-114,243,
 *          sc/layers/mlp.py:24-28,61-83) ----------------------------------------------------------------
 * Out[m][n] = alpha * sum_k X[m][k] * W[n][k] (+ bias[n])          X:(M,K) ldx   W:(N,K) ldw   Out:(M,N) ldo
 * out_mode 0: Out is bf16;  1: Out is fp32 (overwrite).  Accumulating forms: cx_gemm_bf16_nt_accum / _tn_accum
 * (deterministic split-K); `split_k` is ignored here.  Requirements: K % 64 == 0, N % 4 == 0, ldx/ldw % 8 == 0, ldo % 4 == 0.
 * forward: X=act, W=weight.  dgrad: X=dY, W=W^T.  wgrad: X=dY^T, W=act^T (both via cx_transpose_bf16). */
int cx_gemm_bf16_nt(const uint16_t* X, const uint16_t* W, void* Out, const float* bias, int M, int N, int K, int ldx,
                    int ldw, int ldo, int out_mode, int split_k, float alpha, void* stream);
/* Out:(M,N) fp32 (ld = N) += X W^T with split-K through a caller-owned fp32 workspace `ws` (>= M*N floats; more lets
 * more K slices run concurrently) and a fixed-order reduction: the wgrad form (X = dY^T, W = act^T, K = tokens).
 * Deterministic; no atomics. */
int cx_gemm_bf16_nt_accum(const uint16_t* X, const uint16_t* W, float* Out, float* ws, long ws_floats, int M, int N,
                          int K, int ldx, int ldw, void* stream);
/* wgrad in its natural layout, no transposes: G:(O,I) fp32 (ld = I) += dY:(T,O)^T A:(T,I).  The kernel reads
 * round_up(T,64) token rows of both operands: rows T.. of that range must be ZERO.  O % 256 == 0, I % 256 == 0 (else
 * CX_ERR_SHAPE: transpose both operands with cx_transpose_bf16 and use cx_gemm_bf16_nt_accum). */
int cx_gemm_bf16_tn_accum(const uint16_t* dY, const uint16_t* A, float* G, float* ws, long ws_floats, int T, int O, int I,
                          int ld_dy, int ld_a, void* stream);
/* Sampled per-launch timing of this (dominant) kernel for bench.py's roofline: every `stride`-th launch is bracketed by
 * HIP events on its own stream; collect() synchronises them and returns the summed duration (ms) and algorithmic FLOPs
 * (2*M*N*K) of exactly the sampled launches. */
int cx_prof_gemm_config(int enable, int stride);
int cx_prof_gemm_collect(double* total_ms, double* total_flop, long* launches_timed, long* launches_total);
/* Per-box calibration for bench.py (cx_abi_version >= 9; measurement aids, no product kernel calls them): the pool's boxes differ
 * by +-3 % in sustained matrix-core clock at their power limit, so a roofline fraction is also reported against what THIS box
 * sustains.  cx_calib_mfma_bf16: register-only v_mfma_f32_16x16x32_bf16 loop (the GEMMs' instruction), `nwg` workgroups of one wave per SIMD, each wave
 * `iters` x 16 MFMAs (16 KFLOP each) on fragments from seed (1024 x 16 B, random bf16): FLOPs = nwg * 4 * iters * 8 * 32768; cycles (NULL ok):
 * nwg x 4 s_memtime deltas.  cx_calib_copy: 16 B per lane grid-stride copy, bytes % 16 == 0 (HBM stream rate). */
int cx_calib_mfma_bf16(const void* seed_1024x16B, int iters, int nwg, long long* cycles_nwg_x4, float* sink, void* stream);
int cx_calib_copy(const void* src, void* dst, long bytes, void* stream);

/* Out[c][r] = In[r][c] for r < rows, zero for rows <= r < rows_pad (token padding for the wgrad reduction).
 * In:(rows,cols) ld_in, Out:(cols,rows_pad) ld_out.  cols % 8 == 0. */
int cx_transpose_bf16(const uint16_t* In, uint16_t* Out, int rows, int cols, int ld_in, int ld_out, int rows_pad,
                      void* stream);
/* fp32 master weights -> bf16 shadow (row-major copy) and optional transposed bf16 shadow (for dgrad). */
int cx_cast_f32_to_bf16(const float* In, uint16_t* Out, long n, void* stream);
int cx_cast_transpose_f32_to_bf16(const float* In, uint16_t* OutT, int rows, int cols, void* stream);
/* The same for n_jobs matrices in ONE launch (cx_abi_version >= 8): `jobs` is a DEVICE array; max_tiles = the largest
 * ceil(rows / 64) * ceil(cols / 64) among them.  The optimizer step's refresh of the transposed bf16 weight shadows was 48
 * launches of ~11 us each -- 5 % of BASELINE configs[0]'s 11 ms step. */
typedef struct {
    const float* in;    /* (rows, cols) fp32, row-major */
    uint16_t* out_t;    /* (cols, rows) bf16 */
    int rows, cols;
} CxCastJob;
int cx_cast_transpose_f32_to_bf16_batched(const CxCastJob* jobs, int n_jobs, int max_tiles, void* stream);
int cx_cast_bf16_to_f32(const uint16_t* In, float* Out, long n, void* stream);

/* ---- K5/K6  dropout_add_layer_norm / layer_norm  (flash_attn.ops.layer_norm; sc/layers/block.py:309-319,
 *             422-431,453-462; sc/models/encoder/modeling_nomic_bert.py:534; sc/models/vit/vit.py:253-263) ---
 * z = x0 + residual (residual may be NULL); out = (z-mean)*rstd*gamma + beta.  Statistics in fp32.
 * z_out (may be NULL, may alias x0) receives z in bf16 (kept for backward / the pre-norm residual stream).
 * dropout p == 0 only (all five BASELINE configs train with resid_pdrop = 0).  d in {256,512,768,1024}. */
int cx_layernorm_fwd(const uint16_t* x0, const uint16_t* residual, const float* gamma, const float* beta,
                     uint16_t* out, uint16_t* z_out, float* mean, float* rstd, int rows, int d, float eps,
                     void* stream);
/* dout = dout_a + dout_b (dout_b may be NULL); dz_extra (may be NULL) is added to dz (pre-norm residual grad).
 * dz: bf16 (rows,d) (grad of x0 and of residual); dgamma/dbeta: fp32[d], accumulated (+=).  ws (may be NULL): fp32
 * scratch of ws_floats >= 512*d floats enabling the deterministic two-stage reduction (else atomics are used). */
int cx_layernorm_bwd(const uint16_t* dout_a, const uint16_t* dout_b, const uint16_t* z, const float* gamma,
                     const float* mean, const float* rstd, const uint16_t* dz_extra, uint16_t* dz, float* dgamma,
                     float* dbeta, float* ws, long ws_floats, int rows, int d, void* stream);

/* Backward of the LAST LayerNorm of a pooled encoder with the pooling / normalisation backward folded in
 * (sc/models/biencoder/modeling_biencoder.py:79-90,314-319 backward + the LayerNorm backward of sc/layers/block.py:453-462
 * or sc/models/vit/vit.py:253-263): dout[t] = w(t) * g[seq(t)], g = d(pooled vector) from (demb, emb, norm) exactly as
 * cx_pool_normalize_bwd computes it, w = 1/len (pool_mode 0, mean) or [t first token] (pool_mode 1, cls).  dout stays in
 * fp32 registers: the bf16 rounding of a materialised dout showed in the final LayerNorm's bias gradient (3.7 % vs the
 * fp32 oracle on the reference's GradCache fixture, where bf16-eager is 0.5 % off).  rows = total tokens. */
int cx_layernorm_bwd_pooled(const float* demb, const float* emb, const float* norm, const int32_t* cu_seqlens, int B,
                            int pool_mode, int normalize, const uint16_t* z, const float* gamma, const float* mean,
                            const float* rstd, uint16_t* dz, float* dgamma, float* dbeta, float* dz_colsum, float* ws,
                            long ws_floats, int rows, int d, void* stream);
/* cx_layernorm_bwd that also accumulates dz_colsum[n] += sum_t dz[t][n] (fp32[d]; also a trailing argument of the pooled
 * form above, NULL = off): dz is the gradient of (Linear output + residual), so this IS the bias gradient of that Linear
 * (FusedDense backward's db for fc2 / out_proj of the BERT-base and ViT towers) without a second pass over dz.  Needs
 * the workspace (CX_ERR_ARG otherwise: call cx_layernorm_bwd + cx_bias_grad). */
int cx_layernorm_bwd_colsum(const uint16_t* dout_a, const uint16_t* dout_b, const uint16_t* z, const float* gamma,
                            const float* mean, const float* rstd, const uint16_t* dz_extra, uint16_t* dz, float* dgamma,
                            float* dbeta, float* dz_colsum, float* ws, long ws_floats, int rows, int d, void* stream);

/* dropout p > 0 (flash_attn.ops.layer_norm.dropout_add_layer_norm(p > 0), sc/layers/block.py:422-431,453-462 with
 * resid_pdrop > 0): z = dropout_p(x0) + residual, out = LN(z).  The keep-mask is Philox4x32-10(seed; offset + site,
 * element group) -- a pure function of the torch generator's (seed, offset) the host drew for this chunk, so backward and
 * a GradCache re-forward under RandContext (sc/rand_state.py:6-22) regenerate it; nothing is stored.  bwd returns dz
 * (gradient of the residual) AND dx0 = dz * mask / (1 - p) (gradient of x0).  0 < p < 1.  `site` separates the dropout
 * sites of one chunk (2 per block + the embeddings). */
int cx_dropout_add_layernorm_fwd(const uint16_t* x0, const uint16_t* residual, const float* gamma, const float* beta,
                                 uint16_t* out, uint16_t* z_out, float* mean, float* rstd, int rows, int d, float eps,
                                 float p, unsigned long long seed, unsigned long long offset, unsigned int site,
                                 void* stream);
int cx_dropout_add_layernorm_bwd(const uint16_t* dout_a, const uint16_t* dout_b, const uint16_t* z, const float* gamma,
                                 const float* mean, const float* rstd, uint16_t* dz, uint16_t* dx0, float* dgamma,
                                 float* dbeta, float* ws, long ws_floats, int rows, int d, float p,
                                 unsigned long long seed, unsigned long long offset, unsigned int site, void* stream);
/* The same that also accumulates dx0_colsum[n] += sum_t dx0[t][n] (fp32[d], may be NULL: then identical to the call above) -- dx0 is the
 * gradient of the (dropped) sub-layer output, so these are the bias gradient of the Linear that produced it (cx_abi_version >= 10; needs
 * ws >= 3 * d * 256 floats, CX_ERR_ARG otherwise).  Deterministic two-stage reduction, no atomics. */
int cx_dropout_add_layernorm_bwd_colsum(const uint16_t* dout_a, const uint16_t* dout_b, const uint16_t* z, const float* gamma,
                                        const float* mean, const float* rstd, uint16_t* dz, uint16_t* dx0, float* dgamma, float* dbeta,
                                        float* dx0_colsum, float* ws, long ws_floats, int rows, int d, float p, unsigned long long seed,
                                        unsigned long long offset, unsigned int site, void* stream);
/* x <- x * mask / (1 - p) in place, n % 4 == 0 (embedding dropout, sc/models/encoder/modeling_nomic_bert.py:534-535, and
 * its gradient). */
int cx_dropout_scale(uint16_t* x, long n, float p, unsigned long long seed, unsigned long long offset, unsigned int site,
                     void* stream);

/* The same two ops with a dtype per operand, for the flash_attn.ops.layer_norm python surface (`residual_in_fp32`, fp32
 * inputs on the reference's BERT path: embedding LayerNorm fp32 in / fp32 out, layer-0 residual fp32).
 * flags: bit0 x0 is fp32, bit1 residual is fp32, bit2 out is fp32, bit3 z (the saved sum / prenorm residual output) is
 * fp32; clear = bf16 (statistics always use the unrounded fp32 sum); bit4: RMSNorm (flash_attn.ops.rms_norm, K8): out =
 * z * rsqrt(mean(z^2) + eps) * gamma (+ beta, which may then be NULL), `mean` receives 0.  bwd writes
 * dz to dx0 (x0's dtype) and, if not NULL, to dres (the residual's dtype); dout has out's dtype, dz_extra has z's. */
int cx_layernorm_fwd_mixed(const void* x0, const void* residual, const float* gamma, const float* beta, void* out,
                           void* z_out, float* mean, float* rstd, int rows, int d, float eps, int flags, void* stream);
int cx_layernorm_bwd_mixed(const void* dout, const void* z, const float* gamma, const float* mean, const float* rstd,
                           const void* dz_extra, void* dx0, void* dres, float* dgamma, float* dbeta, int rows, int d,
                           int flags, void* stream);

/* ---- a11 BertEmbeddings + K6 embedding LayerNorm, on the unpadded token stream
 *          (sc/layers/embedding.py:594-615, sc/models/encoder/modeling_nomic_bert.py:531-535, K4 unpad_input
 *          sc/models/encoder/modeling_nomic_bert.py:332-333) -------------------------------------------------
 * token t: flat = indices[t] (= b*S + s of the padded batch); id = input_ids[flat]; pos = flat % S.
 * z = word[id] + type[0] (+ pos_emb[pos] if pos_emb != NULL); out = LN(z) in bf16; mean/rstd saved. */
int cx_embed_ln_fwd(const int64_t* input_ids, const int32_t* indices, const float* word, const float* type0,
                    const float* pos_emb, const float* gamma, const float* beta, uint16_t* out, float* mean,
                    float* rstd, int T, int S, int d, float eps, void* stream);
/* backward: scatter-adds into fp32 grads (word rows except padding_idx, type row 0, pos rows), dgamma/dbeta. */
int cx_embed_ln_bwd(const uint16_t* dout_a, const uint16_t* dout_b, const int64_t* input_ids,
                    const int32_t* indices, const float* word, const float* type0, const float* pos_emb,
                    const float* gamma, const float* mean, const float* rstd, float* dword, float* dtype0,
                    float* dpos, float* dgamma, float* dbeta, int T, int S, int d, int padding_idx, void* stream);

/* The same backward with the word-embedding rows reduced WITHOUT atomics: `sorted_ids` = the chunk's T token ids in
 * ascending order (stable), `perm` = the token index each sorted entry came from (both int32[T], prepared by the host
 * with one device sort per chunk), dz_scratch = (T, d) fp32 (the row gradients are not rounded before they are summed).  One workgroup per vocabulary row sums its tokens in token
 * order: deterministic, and ~10x faster than 100 M fp32 atomics per 131072-token chunk.  dpos / dtype0 / dgamma / dbeta as
 * above. */
int cx_embed_ln_bwd_sorted(const uint16_t* dout_a, const uint16_t* dout_b, const int64_t* input_ids,
                           const int32_t* indices, const float* word, const float* type0, const float* pos_emb,
                           const float* gamma, const float* mean, const float* rstd, float* dword, float* dtype0,
                           float* dpos, float* dgamma, float* dbeta, int T, int S, int d, int padding_idx, int vocab,
                           const int32_t* sorted_ids, const int32_t* perm, float* dz_scratch, void* stream);

/* ---- K10 swiglu (flash_attn.ops.activations.swiglu; sc/layers/mlp.py:75) and GELU(erf) (mlp.py:30-34) ----
 * yg:(T, 2*I) holds y = fc11(x) and gate = fc12(x);  act = silu(gate) * y, fp32 math, one rounding.
 * layout 0: yg = [y | gate] concatenated; layout 1: interleaved in groups of 32 columns
 * ([y 0..31 | gate 0..31 | y 32..63 | ...]) -- the layout of the fused fc1 weight and of cx_gemm_bf16_swiglu. */
int cx_swiglu_fwd(const uint16_t* yg, uint16_t* act, int T, int I, int layout, void* stream);
int cx_swiglu_bwd(const uint16_t* dact, const uint16_t* yg, uint16_t* dyg, int T, int I, int layout, void* stream);
/* K9 + K10 fused: Act:(M,I) = silu(X Wg^T) * (X Wy^T) in one pass, W:(2I,K) rows interleaved by 32 as above; YG (may be
 * NULL):(M,2I) receives the pre-activation pair in the interleaved layout (kept for backward). */
/* Projection whose output is the `x0` of `dropout_add_layer_norm(x0, residual, ...)` (sc/layers/block.py:422-431,
 * 453-462: out_proj and fc2 of every block): Out = bf16(bf16(X W^T + bias) + Residual), i.e. the residual add leaves the
 * LayerNorm kernel (call cx_layernorm_fwd with x0 = Out, residual = NULL).  Residual: (M, N) bf16, leading dim ldr.
 * CX_ERR_SHAPE = shape not covered by the fused kernel (use cx_gemm_bf16_nt and pass the residual to the LayerNorm). */
int cx_gemm_bf16_nt_residual(const uint16_t* X, const uint16_t* W, uint16_t* Out, const float* bias,
                             const uint16_t* Residual, int M, int N, int K, int ldx, int ldw, int ldo, int ldr,
                             void* stream);
/* The same projection for FEW output tiles and a LONG K (sc/trainers at small batches: BASELINE configs[0], B = 32, S = 64 is
 * 2048 token rows -- fc2 is 24 tiles of 256 x 256 with 48 K-tiles each): the K range is split into K / 384 slices (at most
 * 16; a function of K alone, so the summation order does not depend on M), fp32 partial slabs go to `ws` (>= slices * M * N
 * floats) and one fixed-order pass writes Out = bf16(bf16(sum + bias) + Residual) (bias, Residual optional).  Deterministic.
 * CX_ERR_SHAPE = not the case this route is for (more than 64 tiles, K < 1536, no room for the slabs): use
 * cx_gemm_bf16_nt / cx_gemm_bf16_nt_residual. */
int cx_gemm_bf16_nt_splitk(const uint16_t* X, const uint16_t* W, uint16_t* Out, const float* bias, const uint16_t* Residual,
                           float* ws, long ws_floats, int M, int N, int K, int ldx, int ldw, int ldo, int ldr, void* stream);
int cx_gemm_bf16_swiglu(const uint16_t* X, const uint16_t* W, uint16_t* YG, uint16_t* Act, int M, int I, int K, int ldx,
                        int ldw, int ld_yg, int ld_act, void* stream);
/* fc1 of the plain MLP (sc/layers/mlp.py:30-34: fc2(gelu(fc1 x)), erf GELU, block.py:181-189) with bias + GELU fused
 * into the GEMM epilogue: Pre (M,N) bf16 = X W^T + bias (optional: NULL in the no-grad pass), Act (M,N) = gelu(Pre).
 * CX_ERR_SHAPE = shape not covered by the fused kernel (run cx_gemm_bf16_nt + cx_bias_gelu_fwd instead). */
int cx_gemm_bf16_bias_gelu(const uint16_t* X, const uint16_t* W, const float* bias, uint16_t* Pre, uint16_t* Act, int M,
                           int N, int K, int ldx, int ldw, int ld_pre, int ld_act, void* stream);
/* The same with the activation selected at run time (sc/layers/mlp.py:8-34 `activation`): act = 0 exact-erf GELU (BERT-base,
 * HF / timm ViTs), 1 quick_gelu = x * sigmoid(1.702 x) (sc/layers/activations.py:4-5: the OpenAI-CLIP image tower,
 * sc/models/vit/clip.py:14-58). */
int cx_gemm_bf16_bias_act(const uint16_t* X, const uint16_t* W, const float* bias, uint16_t* Pre, uint16_t* Act, int M,
                          int N, int K, int ldx, int ldw, int ld_pre, int ld_act, int act, void* stream);
/* fc2 dgrad of the plain MLP with the GELU / quick_gelu backward fused into the epilogue (cx_abi_version >= 10; the reference gets the
 * pair from flash_attn.ops.fused_dense / FusedMLP, sc/layers/mlp.py:30-34): dPre (M, N) = bf16(bf16(dY W^T) * act'(Pre)), Pre = the
 * biased pre-activation saved by cx_gemm_bf16_bias_act, W: (N, K) row-major (the transposed fc2 weight), act as there.  Bit-identical to
 * cx_gemm_bf16_nt + cx_bias_act_bwd_colsum(bias = NULL).  dbias (may be NULL): fp32 [N] += column sums of the bf16 dPre = the fc1 bias
 * gradient, summed in a fixed order through `ws` (>= ceil(M / 128) * N floats): deterministic, no atomics.
 * CX_ERR_SHAPE = not covered (K % 64, N % 8, ld % 8, workspace too small): run the two calls above instead. */
int cx_gemm_bf16_act_bwd(const uint16_t* dY, const uint16_t* W, const uint16_t* Pre, uint16_t* dPre, float* dbias, float* ws,
                         long ws_floats, int M, int N, int K, int ldx, int ldw, int ld_pre, int ld_dpre, int act, void* stream);
/* fc2 dgrad of the gated MLP with the backward of `swiglu` (flash_attn.ops.activations, sc/layers/mlp.py:75) fused into
 * the epilogue: dYG (M, 2I) = d swiglu(YG) applied to dAct = dY W^T, YG / dYG in the interleaved-by-32 layout of
 * cx_gemm_bf16_swiglu; W: (I, K) row-major (the transposed fc2 weight).  dAct is never written to memory.
 * CX_ERR_SHAPE = shape not covered (needs I % 256 == 0, K % 64 == 0): run cx_gemm_bf16_nt + cx_swiglu_bwd instead. */
int cx_gemm_bf16_swiglu_bwd(const uint16_t* dY, const uint16_t* W, const uint16_t* YG, uint16_t* dYG, int M, int I, int K,
                            int ldx, int ldw, int ld_yg, void* stream);
/* Round 3: the compact save of the gated MLP.  The reference's swiglu keeps y and gate for its backward (sc/layers/mlp.py:75,
 * flash_attn.ops.activations.swiglu) and the activation for fc2's: three (M, I) tensors per layer.  y is redundant:
 * act = y * silu(gate), so   d gate = d * y * silu'(g) = d * act * (1 / g + 1 - sigmoid(g)).   cx_gemm_bf16_swiglu_gate
 * saves G: (M, I) = bf16(X Wg^T) in plain column order (may be NULL) next to Act; cx_gemm_bf16_swiglu_bwd_gate is
 * cx_gemm_bf16_swiglu_bwd reading (Act, G) (both leading dimension ld_ag) and writing dYG (M, 2I) in the interleaved-by-32
 * layout; cx_swiglu_bwd_gate is the standalone form for shapes the fused kernel does not cover (CX_ERR_SHAPE as there).
 * The recovered y carries one bf16 rounding (of act), as the saved bf16 y did. */
int cx_gemm_bf16_swiglu_gate(const uint16_t* X, const uint16_t* W, uint16_t* G, uint16_t* Act, int M, int I, int K, int ldx,
                             int ldw, int ld_g, int ld_act, void* stream);
int cx_gemm_bf16_swiglu_bwd_gate(const uint16_t* dY, const uint16_t* W, const uint16_t* Act, const uint16_t* G, uint16_t* dYG,
                                 int M, int I, int K, int ldx, int ldw, int ld_ag, int ld_dyg, void* stream);
int cx_swiglu_bwd_gate(const uint16_t* dact, const uint16_t* act, const uint16_t* gate, uint16_t* dyg, int T, int I,
                       void* stream);
/* act = gelu_erf(pre + bias); bias fp32[I] may be NULL.  backward: dpre = dact * gelu'(pre + bias). */
int cx_bias_gelu_fwd(const uint16_t* pre, const float* bias, uint16_t* act, int T, int I, void* stream);
int cx_bias_gelu_bwd(const uint16_t* dact, const uint16_t* pre, const float* bias, uint16_t* dpre, int T, int I,
                     void* stream);
/* The same backward with the bias gradient of the same pass: dbias[n] += sum_t dpre[t][n] (fp32 atomics; dbias may be NULL).
 * Replaces cx_bias_gelu_bwd + cx_bias_grad (the second kernel re-read dpre: T x I x 2 bytes). */
int cx_bias_gelu_bwd_colsum(const uint16_t* dact, const uint16_t* pre, const float* bias, uint16_t* dpre, float* dbias, int T,
                            int I, void* stream);
/* ... and both directions with the activation selected at run time (act as in cx_gemm_bf16_bias_act). */
int cx_bias_act_fwd(const uint16_t* pre, const float* bias, uint16_t* act_out, int T, int I, int act, void* stream);
int cx_bias_act_bwd_colsum(const uint16_t* dact, const uint16_t* pre, const float* bias, uint16_t* dpre, float* dbias, int T,
                           int I, int act, void* stream);
/* dbias[n] += sum_t dY[t][n]  (fp32 atomic accumulate; bgrad half of FusedDense backward). */
int cx_bias_grad(const uint16_t* dY, float* dbias, int T, int N, int ld, void* stream);

/* ---- K1/K2 + K11  flash_attn_varlen_qkvpacked_func with rotary fused into the Q/K tile load
 *          (sc/layers/attention.py:122-135,172-182; sc/layers/embedding.py:685-706) -------------------------
 * qkv:(T,3,H,64) bf16 packed, cu_seqlens:(B+1) int32, non-causal, softmax in fp32, dropout 0.
 * rot_cos/rot_sin: fp32 (>= max_seqlen, 32) tables for full non-interleaved rotary, or NULL for no rotary
 * (bert-base / ViT).  out:(T,H,64) bf16.  lse:(H,T) fp32 (natural log of the scaled-score softmax denominator). */
int cx_attn_varlen_fwd(const uint16_t* qkv, const int32_t* cu_seqlens, const float* rot_cos, const float* rot_sin,
                       uint16_t* out, float* lse, int B, int H, int T, int max_seqlen, float softmax_scale,
                       void* stream);
/* delta:(H,T) fp32 scratch (rowsum(dO*O)); dqkv:(T,3,H,64) bf16 fully overwritten (dq,dk un-rotated). */
int cx_attn_varlen_bwd(const uint16_t* dout, const uint16_t* qkv, const uint16_t* out, const float* lse,
                       const int32_t* cu_seqlens, const float* rot_cos, const float* rot_sin, float* delta,
                       uint16_t* dqkv, int B, int H, int T, int max_seqlen, float softmax_scale, void* stream);

/* backward for a qkv whose q and k were rotated IN PLACE beforehand (cx_rotary_qkv_inplace; the forward is then
 * cx_attn_varlen_fwd with NULL tables): nothing is rotated at the loads, dq / dk leave through the inverse rotation.  The
 * engine takes this route for max_seqlen > 128 (the streaming kernels would re-rotate every row once per 128-row block). */
int cx_attn_varlen_bwd_prerotated(const uint16_t* dout, const uint16_t* qkv_rotated, const uint16_t* out, const float* lse,
                                  const int32_t* cu_seqlens, const float* rot_cos, const float* rot_sin, float* delta,
                                  uint16_t* dqkv, int B, int H, int T, int max_seqlen, float softmax_scale, void* stream);

/* attention dropout (flash_attn_varlen_qkvpacked_func(dropout_p > 0), attn_pdrop > 0): O = (P * keep / (1 - p)) V with
 * keep(b, h, q, key) = Philox4x32-10(seed; offset + site, (b * H + h, q, key / 4))[key % 4] >= p * 2^32 -- regenerated by
 * the backward and by a GradCache re-forward under RandContext, never stored.  Same arguments as cx_attn_varlen_fwd/_bwd
 * plus (p, seed, offset, site); 0 < p < 1, any max_seqlen < 2^20, B * H < 2^24. */
int cx_attn_varlen_dropout_fwd(const uint16_t* qkv, const int32_t* cu_seqlens, const float* rot_cos, const float* rot_sin,
                               uint16_t* out, float* lse, int B, int H, int T, int max_seqlen, float softmax_scale,
                               float p_drop, unsigned long long seed, unsigned long long offset, unsigned int site,
                               void* stream);
int cx_attn_varlen_dropout_bwd(const uint16_t* dout, const uint16_t* qkv, const uint16_t* out, const float* lse,
                               const int32_t* cu_seqlens, const float* rot_cos, const float* rot_sin, float* delta,
                               uint16_t* dqkv, int B, int H, int T, int max_seqlen, float softmax_scale, float p_drop,
                               unsigned long long seed, unsigned long long offset, unsigned int site, void* stream);

/* ---- K3  flash_attn_kvpacked_func / flash_attn_varlen_kvpacked_func (cross-attention; sc/layers/attention.py:313-433,
 *          FlashAttentionPooling: one latent query per sequence over the sequence's keys) ---
 * q (Tq, H, 64), kv (Tk, 2, H, 64) bf16, batch entry b: queries cu_seqlens_q[b] .. [b+1), keys cu_seqlens_k[b] .. [b+1);
 * out (Tq, H, 64), lse (H, Tq) fp32.  Non-causal, no dropout, no rotary, H_kv == H (the reference: "we don't really
 * support mqa / gqa").  A query with no keys gets out = 0.  bwd: delta = fp32 scratch (H, Tq); dq (Tq, H, 64),
 * dkv (Tk, 2, H, 64) are overwritten. */
int cx_attn_varlen_kvpacked_fwd(const uint16_t* q, const uint16_t* kv, const int32_t* cu_seqlens_q,
                                const int32_t* cu_seqlens_k, uint16_t* out, float* lse, int B, int H, int Tq,
                                int max_seqlen_q, int max_seqlen_k, float softmax_scale, void* stream);
int cx_attn_varlen_kvpacked_bwd(const uint16_t* dout, const uint16_t* q, const uint16_t* kv, const uint16_t* out,
                                const float* lse, const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k, float* delta,
                                uint16_t* dq, uint16_t* dkv, int B, int H, int Tq, int max_seqlen_q, int max_seqlen_k,
                                float softmax_scale, void* stream);
/* standalone K11 (apply_rotary_emb_func on a packed qkv, in place on q and k; sign=-1 gives the backward). */
int cx_rotary_qkv_inplace(uint16_t* qkv, const int32_t* cu_seqlens, const float* rot_cos, const float* rot_sin,
                          int B, int H, int T, int max_seqlen, int sign, void* stream);

/* apply_rotary_emb_func on one (T,H,64) view with token stride `tok_stride` elements (in place; sign=-1: backward). */
int cx_rotary_apply(uint16_t* x, long tok_stride, const int32_t* cu_seqlens, const float* rot_cos,
                    const float* rot_sin, int B, int H, int T, int max_seqlen, int sign, void* stream);

/* ---- a9 BiEncoder pooling + normalize (sc/models/biencoder/modeling_biencoder.py:79-90,44-49,314-319) -----
 * mode 0 = mean over the sequence's tokens, 1 = cls (first token).  emb:(B,d) fp32 = x / max(||x||, 1e-12);
 * if normalize == 0 the raw pooled vector is written.  norm:(B) fp32 saved for backward. */
int cx_pool_normalize_fwd(const uint16_t* h, const int32_t* cu_seqlens, float* emb, float* norm, int B, int d,
                          int mode, int normalize, void* stream);
int cx_pool_normalize_bwd(const float* demb, const float* emb, const float* norm, const int32_t* cu_seqlens,
                          uint16_t* dh, int B, int d, int mode, int normalize, void* stream);

/* ---- K13 fused InfoNCE (sc/loss.py:76-132 clip_loss; LogitScale sc/models/biencoder/modeling_biencoder.py:30-41)
 * Q:(N,dim) ldq, D:(G,dim) ldd fp32 (D = rank-ordered all-gather of the document embeddings),
 * labels:(N) int64 = (arange(N)+rank*N)*(G/(N*world)) computed by the host exactly as loss.py:108-117.
 * logits = scale * Q D^T are never written: exact-fp32 MFMA tiles feed an online row log-sum-exp.
 * ws: fp32 scratch of cx_infonce_ws_floats(N,G) floats.  Outputs: lse:(N) fp32, loss_rows:(N) fp32
 * (= lse_i - logit_{i,label_i}); the host takes mean * world_size (loss.py:125). */
long cx_infonce_ws_floats(int N, int G);
int cx_infonce_fwd(const float* Q, const float* D, const int64_t* labels, float scale, float* ws, float* lse,
                   float* loss_rows, int N, int G, int dim, int ldq, int ldd, void* stream);
/* The same forward that also returns every row's arg max over the G columns (cx_abi_version >= 9): argmax:(N) int32 = the
 * FIRST column attaining the row maximum, i.e. `similarity.argmax(dim=1)` of the in-batch accuracy the reference logs
 * (sc/loss.py:127-130), read off the tiles the loss walks anyway -- the (N,G) similarity is not written for it either.
 * ws: cx_infonce_argmax_ws_floats(N,G) floats.  argmax == NULL: exactly cx_infonce_fwd. */
long cx_infonce_argmax_ws_floats(int N, int G);
int cx_infonce_fwd_argmax(const float* Q, const float* D, const int64_t* labels, float scale, float* ws, float* lse,
                          float* loss_rows, int32_t* argmax, int N, int G, int dim, int ldq, int ldd, void* stream);
/* backward of  coef * sum_i loss_rows[i]:  Gm[i][j] = coef*scale*(softmax_ij - [j==label_i]) is written to
 * Gmat:(N,G) and GmatT:(G,N) fp32 scratch; dQ:(N,dim) = Gm D, dD:(G,dim) = Gm^T Q (overwritten);
 * dscale_accum (may be NULL): += coef * sum_ij (softmax_ij - y_ij) * (Q D^T)_ij   (d loss / d scale).
 * QT:(dim,N) and DT:(dim,G) fp32 scratch for the K-contiguous operands of the two output GEMMs.
 * Requirements: dim % 4 == 0, N % 4 == 0, G % 4 == 0, ldq/ldd % 4 == 0. */
int cx_infonce_bwd(const float* Q, const float* D, const int64_t* labels, const float* lse, float scale, float coef,
                   float* Gmat, float* GmatT, float* QT, float* DT, float* dQ, float* dD, float* dscale_accum,
                   int N, int G, int dim, int ldq, int ldd, void* stream);
/* ---- the same loss on the fp8 matrix-core path (BASELINE.json configs[4] "fp8 MFMA similarity GEMM"; the reference
 * only carries the `use_fp8` flag, configs/train/contrastive_pretrain.yaml:24).  Rows are quantised to OCP e4m3 with
 * one scale per row, the contraction is v_mfma_scale_f32_32x32x64_f8f6f4 (fp32 accumulate, online fp32 log-sum-exp);
 * the logits are never written.  dim in {256,512,768,1024}.  Scratch, all caller-owned: ws = fp32
 * [cx_infonce_fp8_ws_floats(N,G)]; Q8 (N,dim) / D8 (G,dim) bytes and sq (N) / sd (G) row scales are WRITTEN by fwd and
 * read again by bwd.  bwd writes GmT (G,N) bf16 = coef*scale*(softmax - onehot)^T once and forms dD = GmT Q, dQ = GmT^T D
 * on the bf16 GEMM family (Qb (N,dim), QbT (dim,N), Db (G,dim) bf16 scratch; gws / gws_floats = split-K workspace as for
 * cx_gemm_bf16_tn_accum, >= N*dim floats).  bwd needs N % 256 == 0, G % 64 == 0, dim % 128 == 0 (else CX_ERR_SHAPE:
 * use the exact path).  Tolerance against the fp32 oracle (tests/test_infonce_fp8_gpu.py): e4m3 has 3 mantissa bits. */
long cx_infonce_fp8_ws_floats(int N, int G);
int cx_infonce_fp8_fwd(const float* Q, const float* D, const int64_t* labels, float scale, float* ws, uint8_t* Q8,
                       uint8_t* D8, float* sq, float* sd, float* lse, float* loss_rows, int N, int G, int dim, int ldq,
                       int ldd, void* stream);
int cx_infonce_fp8_bwd(const float* Q, const float* D, const int64_t* labels, const float* lse, float scale, float coef,
                       const uint8_t* Q8, const uint8_t* D8, const float* sq, const float* sd, uint16_t* GmT, uint16_t* Qb,
                       uint16_t* QbT, uint16_t* Db, float* gws, long gws_floats, float* dQ, float* dD,
                       float* dscale_accum, int N, int G, int dim, int ldq, int ldd, void* stream);
/* plain exact-fp32 MFMA GEMM  C[m][n] = sum_k A[m][k] B[n][k]  (K % 4 == 0; lda, ldb % 4 == 0). */
int cx_sgemm_nt(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                void* stream);
int cx_transpose_f32(const float* In, float* Out, int rows, int cols, int ld_in, int ld_out, void* stream);

/* ---- native encoder engine (a10-a17: NomicBertModel / NomicBertEncoder / Block / FlashAttention / (Gated)MLP,
 *      sc/models/encoder/modeling_nomic_bert.py:307-395,515-587; sc/layers/block.py:389-463) -----------------
 * One call enqueues a whole chunk forward (embeddings -> L post-norm blocks -> pool -> normalize) or backward on
 * `stream`.  All memory is described by the caller in CxEncoderDesc / CxChunkBuffers. */
typedef struct CxLayerWeights {
    const uint16_t* Wqkv;   /* (3d, d) bf16 */
    const uint16_t* Wout;   /* (d, d) */
    const uint16_t* Wfc1;   /* gated: (2I, d) fc11/fc12 rows interleaved by 32;  plain MLP: (I, d) */
    const uint16_t* Wfc2;   /* (d, I) */
    const uint16_t* WqkvT;  /* transposed bf16 shadows for dgrad: (d, 3d) */
    const uint16_t* WoutT;  /* (d, d) */
    const uint16_t* Wfc1T;  /* (d, 2I) or (d, I) */
    const uint16_t* Wfc2T;  /* (I, d) */
    const float* bqkv;      /* fp32 biases or NULL (nomic-bert has none) */
    const float* bout;
    const float* bfc1;
    const float* bfc2;
    const float* ln1_g; const float* ln1_b; const float* ln2_g; const float* ln2_b;
    /* fp32 gradient accumulators (same shapes as the fp32 master parameters) */
    float* gWqkv; float* gWout; float* gWfc1; float* gWfc2;
    float* gbqkv; float* gbout; float* gbfc1; float* gbfc2;
    float* gln1_g; float* gln1_b; float* gln2_g; float* gln2_b;
} CxLayerWeights;

typedef struct CxEncoderDesc {
    int n_layer, d, n_head, d_inner, gated; /* gated=1: SwiGLU GatedMLP; 0: GELU(erf) MLP */
    int vocab, max_pos, padding_idx;
    float ln_eps, softmax_scale;
    const float* word_emb; const float* type_emb; const float* pos_emb; /* fp32 masters; pos_emb NULL for rotary */
    const float* emb_ln_g; const float* emb_ln_b;
    float* gword_emb; float* gtype_emb; float* gpos_emb; float* gemb_ln_g; float* gemb_ln_b;
    const float* rot_cos; const float* rot_sin; /* NULL when rotary_emb_fraction == 0 */
    const CxLayerWeights* layers;              /* HOST array of n_layer entries (device pointers inside) */
    int pool_mode, normalize;
    /* ---- pre-norm trunks (sc/layers/block.py:293-388) and the ViT image tower (sc/models/vit/vit.py:176-276) ---- */
    int prenorm;                               /* 0: post-norm blocks (BERT family); 1: pre-norm blocks + final LN */
    const float* lnf_g; const float* lnf_b;    /* final LayerNorm ln_f (prenorm only) */
    float* glnf_g; float* glnf_b;
    const uint16_t* Wpatch;                    /* ViT: bf16 (d, patch_dim) patch projection; NULL for text trunks */
    const float* bpatch;                       /* fp32[d] or NULL */
    const float* cls_token;                    /* fp32[d] */
    const float* vit_pos;                      /* fp32 (n_patch + 1, d) */
    float* gWpatch; float* gbpatch; float* gcls_token; float* gvit_pos;
    int patch_dim;                             /* C * p * p */
    /* dropout (post-norm text trunks; 0 = off, the BASELINE configs): on every sub-layer output before the residual add
     * (resid_pdrop, sc/layers/block.py:422-431,453-462) and on the embedding-LayerNorm output (embd_pdrop).  Applied when
     * CxChunkBuffers.drop_active != 0 (training mode). */
    float resid_pdrop, embd_pdrop;
    float attn_pdrop;   /* dropout on the attention probabilities (attn_pdrop, sc/layers/attention.py:158-182); text trunks */
    /* ---- the OpenAI-CLIP flavour of the image tower (sc/models/vit/clip.py:14-58; round 3) ---- */
    int mlp_act;                                   /* plain MLP activation: 0 exact-erf GELU, 1 quick_gelu (cx_gemm_bf16_bias_act) */
    const float* lnpre_g; const float* lnpre_b;    /* `prepre_layernom` (sc/models/vit/vit.py:128-132,180): LayerNorm on the */
    float* glnpre_g; float* glnpre_b;              /* embeddings ahead of the first block; NULL = none */
} CxEncoderDesc;

/* Per-chunk activation arena (device memory owned by the caller).  save_for_backward = 0 lets every layer reuse
 * the layer-0 slots (GradCache pass 1, sc/loss.py:135-146); = 1 keeps per-layer slots (pass 2, loss.py:149-161). */
typedef struct CxChunkBuffers {
    long T_cap;                 /* capacity in tokens of every per-token buffer (>= round_up(T,128)) */
    uint16_t* h0;               /* (T,d) embedding-LN output */
    float* emb_mean; float* emb_rstd;
    /* per layer, n_layer slots each (slot stride = T_cap * width elements): */
    uint16_t* qkv;              /* (T,3d) */
    uint16_t* ctx;              /* (T,d) attention output (pre out_proj) */
    float* lse;                 /* (H,T) */
    uint16_t* z1;               /* (T,d) attn_out + residual (LN1 input) */
    uint16_t* h1;               /* (T,d) LN1 output */
    float* mean1; float* rstd1;
    uint16_t* yg;               /* (T,I) per slot: what fc1 keeps for backward -- the biased pre-activation (plain MLP) or the
                                 * gate alone (gated MLP, cx_abi_version >= 6; it was the (T,2I) (y, gate) pair before) */
    uint16_t* act;              /* (T,I) */
    uint16_t* z2;               /* (T,d) */
    uint16_t* h2;               /* (T,d) LN2 output = layer output */
    float* mean2; float* rstd2;
    float* pool_norm;           /* (B) */
    /* backward scratch (single slot each): */
    uint16_t* g_a; uint16_t* g_b; uint16_t* g_c;   /* (T,d) gradient ping-pong buffers */
    uint16_t* g_wide;           /* (T, max(3d, 2I)) */
    uint16_t* g_act;            /* (T, I) */
    uint16_t* tr_a; uint16_t* tr_b; /* transposed operands for wgrad: (max(3d,2I), T_cap) each.  Only the fallback for feature
                                     * counts that are not multiples of 256 uses them (cx_gemm_bf16_tn_accum declines those
                                     * shapes); may be NULL otherwise (cx_abi_version >= 8) -- 24.6 KB per token at d = 768 */
    float* delta;               /* (H,T) */
    float* ws_f32;              /* split-K workspace for the wgrad GEMMs */
    long ws_floats;
    /* pre-norm trunks: input of / output of the final LayerNorm (single slot), its statistics */
    uint16_t* zf; uint16_t* hf; float* meanf; float* rstdf;
    /* ViT front end: patchified pixels (Bc*n_patch rounded up to 64 rows, patch_dim) and their projection (.., d);
     * patch_proj doubles as the gradient of the projection in backward */
    uint16_t* patch_in; uint16_t* patch_proj;
    /* activation checkpointing (sc/models/encoder/modeling_nomic_bert.py:339-365 gradient_checkpointing,
     * sc/models/vit/vit.py:200-231): != 0 -> a saving forward keeps ONE (T,d) tensor per block -- `h2` (post-norm: the
     * block's output = the next block's input) or `z1` (pre-norm: the residual stream at LN1) has n_layer slots, every
     * other per-layer buffer has a single slot -- and backward recomputes each block from it before differentiating
     * it.  Results are bit-identical to checkpoint = 0; the arena shrinks from ~25 KB to ~1.5 KB per token and layer. */
    int checkpoint;
    /* dropout state of the chunk, set by the host before the forward and read again by the backward: Philox (seed, offset)
     * drawn from the torch generator (so RandContext replays it), the on/off switch, and one more (T,d) gradient buffer
     * (the LayerNorm backward returns two different gradients once a mask sits between x0 and the sum). */
    int drop_active;
    unsigned long long drop_seed, drop_offset;
    uint16_t* g_d;
    /* Optional (NULL = none): HOST array of n_layer + 1 hipEvent_t.  A backward call records layer_events[l] on `stream`
     * right after the last kernel that writes block l's parameter gradients (blocks run L-1 ... 0), and
     * layer_events[n_layer] after the embedding (text) / patch-projection (ViT) gradients, i.e. at the end of the call.
     * The data-parallel gradient reduction of a step's LAST backward starts on block l's slice of the flat gradient as
     * soon as its event fires, overlapped with the blocks still differentiating -- what DDP's bucket hooks do for the
     * reference (sc/trainers/text_text.py:163-170). */
    void* const* layer_events;
    /* image towers with a pre-LayerNorm (CxEncoderDesc.lnpre_g): its input, (T, d), kept for backward; statistics in
     * emb_mean / emb_rstd (the slots the text trunk's embedding LayerNorm uses) */
    uint16_t* zpre;
    /* selective activation checkpointing (cx_abi_version >= 6; meaningful with checkpoint != 0): the TOP ckpt_keep blocks
     * (l >= n_layer - ckpt_keep) keep all their intermediates -- every per-layer buffer then has 1 + ckpt_keep slots, slot 0
     * shared by the recomputed blocks, slot 1 + l - (n_layer - ckpt_keep) owned by kept block l -- and backward recomputes
     * only the blocks below them.  0 = the reference's behaviour (every block recomputed).  A memory knob like
     * `checkpoint` itself: results are bit-identical for every value (tests/test_checkpoint_gpu.py). */
    int ckpt_keep;
    /* PatchDropout of the image tower (cx_abi_version >= 7; sc/layers/embedding.py:415-418, 519-557): NULL = every patch.
     * patch_keep: (Bc, n_keep) int32 device array, the patch indices (0 .. n_patch - 1) each image keeps, in the order their
     * tokens take in the sequence ([cls] first, then patch_keep[b][0 .. n_keep - 1]); patch_inv: (Bc, n_patch) int32, position
     * of a patch among its image's kept ones or -1.  Only the kept patches are gathered, projected and run through the blocks:
     * sequences have n_keep + 1 tokens, cu_seqlens = multiples of n_keep + 1, cx_vit_backward takes n_patch = n_keep;
     * n_patch_all = (H / patch) * (W / patch), the row count of patch_inv (the backward does not see the image size). */
    const int32_t* patch_keep;
    const int32_t* patch_inv;
    int n_keep;
    int n_patch_all;
} CxChunkBuffers;

/* input_ids:(Bc,S) int64 padded batch rows of this chunk; indices:(T) int32; cu_seqlens:(Bc+1) int32.
 * emb_out:(Bc,d) fp32. */
int cx_encoder_forward(const CxEncoderDesc* enc, const CxChunkBuffers* buf, const int64_t* input_ids,
                       const int32_t* indices, const int32_t* cu_seqlens, int Bc, int S, int T, int max_seqlen,
                       int save_for_backward, float* emb_out, void* stream);
/* demb:(Bc,d) fp32 cached embedding gradient (GradCache surrogate, sc/loss.py:158-161). Accumulates every
 * parameter gradient in enc->g* / layers[i].g*.  Must follow a forward with save_for_backward = 1 on `buf`. */
int cx_encoder_backward(const CxEncoderDesc* enc, const CxChunkBuffers* buf, const int64_t* input_ids,
                        const int32_t* indices, const int32_t* cu_seqlens, int Bc, int S, int T, int max_seqlen,
                        const float* demb, const float* emb_out, const int32_t* sort_ids, const int32_t* sort_perm,
                        void* stream);
/* sort_ids / sort_perm (both int32[T], or both NULL): the chunk's token ids in ascending stable order and the permutation
 * that sorts them -- with them the word-embedding gradient is a deterministic segmented reduction
 * (cx_embed_ln_bwd_sorted), without them fp32 atomics. */

/* Token-level variant for heads that read every position (the MLM head of NomicBertForPreTraining,
 * sc/models/encoder/modeling_nomic_bert.py:590-669): hidden_out / dhidden are (T, d) bf16 in unpadded token order
 * (row t = token indices[t] of the padded batch).  Same arenas, same contracts as the pooled pair above. */
int cx_encoder_forward_hidden(const CxEncoderDesc* enc, const CxChunkBuffers* buf, const int64_t* input_ids,
                              const int32_t* indices, const int32_t* cu_seqlens, int Bc, int S, int T, int max_seqlen,
                              int save_for_backward, uint16_t* hidden_out, void* stream);
int cx_encoder_backward_hidden(const CxEncoderDesc* enc, const CxChunkBuffers* buf, const int64_t* input_ids,
                               const int32_t* indices, const int32_t* cu_seqlens, int Bc, int S, int T, int max_seqlen,
                               const uint16_t* dhidden, const int32_t* sort_ids, const int32_t* sort_perm, void* stream);

/* ---- ViT image tower (sc/models/vit/vit.py:176-276 ViTModel.forward, sc/layers/embedding.py:465-516
 *      PatchEmbedding.forward, sc/models/biencoder/modeling_biencoder.py:287-319 pooling): one call per chunk.
 * pixels: (Bc, C, H, W) fp32 or bf16 (pixels_bf16 != 0) device tensor; n_patch = (H/p)*(W/p); sequences have
 * n_patch + 1 tokens ([cls] first); cu_seqlens: (Bc+1) int32 = multiples of n_patch + 1.  desc->prenorm selects the
 * block order (ViT-B/16: 1), desc->Wpatch etc. must be set.  backward accumulates every parameter gradient. */
int cx_vit_patchify(const void* pixels, int pixels_bf16, uint16_t* patches, int B, int C, int H, int W, int patch,
                    void* stream);
int cx_vit_assemble_fwd(const uint16_t* proj, const float* cls_token, const float* pos_embed, uint16_t* out, int B,
                        int P, int d, void* stream);
int cx_vit_assemble_bwd(const uint16_t* dz, uint16_t* dproj, float* gcls, float* gpos, int B, int P, int d,
                        void* stream);
/* the same three with a patch subset (PatchDropout, see CxChunkBuffers.patch_keep): keep (B, n_keep) / inv (B, P_all) int32
 * device arrays, NULL = the plain forms above; with keep, P = n_keep in the assemble calls */
int cx_vit_patchify_gather(const void* pixels, int pixels_bf16, uint16_t* patches, int B, int C, int H, int W, int patch,
                           const int32_t* keep, int n_keep, void* stream);
int cx_vit_assemble_fwd_gather(const uint16_t* proj, const float* cls_token, const float* pos_embed, uint16_t* out, int B,
                               int P, int d, const int32_t* keep, void* stream);
int cx_vit_assemble_bwd_gather(const uint16_t* dz, uint16_t* dproj, float* gcls, float* gpos, int B, int P, int d,
                               const int32_t* inv, int P_all, void* stream);
int cx_vit_forward(const CxEncoderDesc* enc, const CxChunkBuffers* buf, const void* pixels, int pixels_bf16,
                   const int32_t* cu_seqlens, int Bc, int C, int H, int W, int patch, int save_for_backward,
                   float* emb_out, void* stream);
int cx_vit_backward(const CxEncoderDesc* enc, const CxChunkBuffers* buf, const int32_t* cu_seqlens, int Bc,
                    int n_patch, const float* demb, const float* emb_out, void* stream);
/* Token-level twins (as cx_encoder_forward_hidden / _backward_hidden for the text trunk): hidden_out / dhidden are the
 * (Bc * (n_patch + 1), d) bf16 hidden states after ln_f and their gradient -- for poolers that live above the C-ABI
 * (sc/models/biencoder/modeling_biencoder.py:93-156 MultiHeadAttentionPooling, `pooling: map` of the vision recipes). */
int cx_vit_forward_hidden(const CxEncoderDesc* enc, const CxChunkBuffers* buf, const void* pixels, int pixels_bf16,
                          const int32_t* cu_seqlens, int Bc, int Cc, int H, int W, int patch, int save_for_backward,
                          uint16_t* hidden_out, void* stream);
int cx_vit_backward_hidden(const CxEncoderDesc* enc, const CxChunkBuffers* buf, const int32_t* cu_seqlens, int Bc, int n_patch,
                           const uint16_t* dhidden, void* stream);

/* ---- K12  fused softmax cross-entropy over a vocabulary-sized class axis (flash_attn.losses.cross_entropy.
 *      CrossEntropyLoss, csrc/xentropy; sc/models/encoder/modeling_nomic_bert.py:603-610).  logits: (N, V) bf16
 *      (logits_bf16 != 0) or fp32, row stride ld; labels int64[N] (== ignore_index -> loss 0, zero gradient).
 *      fwd: loss[i] = lse_i - s*logit[i][label_i], lse[i] = log sum_j exp(s*logit[i][j]), s = logit_scale.
 *      bwd: dlogits = dloss[i] * s * (softmax - onehot); dlogits may alias logits (inplace_backward). */
int cx_xent_fwd(const void* logits, int logits_bf16, const int64_t* labels, float* loss, float* lse, int N, int V,
                long ld, float logit_scale, long ignore_index, void* stream);
int cx_xent_bwd(const float* dloss, const void* logits, int logits_bf16, const float* lse, const int64_t* labels,
                void* dlogits, int N, int V, long ld, long ld_d, float logit_scale, long ignore_index, void* stream);

/* ---- fused optimizer tail of training_step (replaces torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW.step as
 *      sc/trainers/base.py:362-385 and sc/optimizer.py:7-47 use them).  fp32, 16-B aligned contiguous buffers.
 *      cx_grad_sq_norm ADDS sum(grad^2) to *sq_norm_accum (device double, zeroed by the caller; call once per gradient
 *      tensor).  cx_adamw_clip_step applies one AdamW step (amsgrad off, decoupled decay, bias correction for the
 *      1-based `step`) to `param` using grad * min(1, max_norm / (sqrt(*sq_norm) + 1e-6)); sq_norm NULL or
 *      max_norm <= 0 = no clipping.  The clip coefficient is derived on the device: no host synchronisation. */
int cx_grad_sq_norm(const float* grad, long n, double* sq_norm_accum, void* stream);
int cx_adamw_clip_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1,
                       float beta2, float eps, float weight_decay, long step, const double* sq_norm, float max_norm,
                       void* stream);
/* EMA copy of the weights (sc/trainers/base.py:387-391 `self.model["ema"].update(model)`): ema = decay * ema + (1 - decay) *
 * param over a flat fp32 buffer, one pass; 16-B aligned, 0 <= decay <= 1. */
int cx_ema_update(float* ema, const float* param, long n, float decay, void* stream);


/* ---- one-shot exchange over xGMI (sc/distributed.py:5-12 gather_with_grad): receive buffers shared between the
 * per-GPU processes by HIP IPC, every rank stores its shard straight into every peer's buffer (all 7 links at once), one
 * flag exchange.  Host protocol: contrastors_amd/distributed.py::OneShotExchange.
 * cx_ipc_alloc: device memory that can be exported (uncached == 1: hipDeviceMallocUncached, for the flags; == 2: mapped
 *   coherent HOST memory, for the error flag the host polls without synchronising);
 * cx_ipc_export / cx_ipc_open: 64-byte hipIpcMemHandle_t in / mapped pointer out.
 * cx_xgmi_push: peer_bufs_dev[p] + dst_offset_bytes <- src[0 .. bytes) for p = 0 .. world-1 (device array of pointers).
 * cx_xgmi_scatter: peer_bufs_dev[p] + slot * slice_bytes <- src + p * slice_bytes (the reduce-scatter's send side).
 * cx_xgmi_signal_wait: peer_flags_dev[p][rank] = epoch for all p, then wait until my_flags[p] >= epoch for all p
 *   (system-scope release / acquire); gives up after max_spins polls and stores 1 + p in *err_flag (uncached memory).
 * cx_sum_slots_f32: out[i] = sum_w slots[w * n + i], n % 4 == 0.  All sizes / offsets multiples of 16 bytes. */
int cx_ipc_alloc(void** ptr, long bytes, int uncached);
int cx_ipc_free(void* ptr);
int cx_ipc_export(void* ptr, unsigned char* handle64);
int cx_ipc_open(const unsigned char* handle64, void** ptr);
int cx_ipc_close(void* ptr);
int cx_xgmi_push(const void* src, void* const* peer_bufs_dev, long dst_offset_bytes, long bytes, int world, void* stream);
int cx_xgmi_scatter(const void* src, void* const* peer_bufs_dev, int slot, long slice_bytes, int world, void* stream);
int cx_xgmi_signal_wait(unsigned int* const* peer_flags_dev, unsigned int* my_flags, int rank, int world, unsigned int epoch,
                        long max_spins, unsigned int* err_flag, void* stream);
int cx_sum_slots_f32(const float* slots, float* out, long n, int world, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CONTRASTORS_HIP_H */
