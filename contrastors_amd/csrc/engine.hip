// engine.hip -- native encoder runtime: one C call enqueues a whole GradCache chunk forward or backward
// (embeddings -> L post-norm or pre-norm transformer blocks -> pooling -> L2 normalise) on a HIP stream; text trunks
// (cx_encoder_*) and the ViT image tower (cx_vit_*) share the block schedule.
//
// Mirrors, as a kernel schedule over a caller-owned activation arena, what the reference expresses as python
// module calls: NomicBertModel.forward (sc/models/encoder/modeling_nomic_bert.py:515-587), NomicBertEncoder.forward
// (:307-395, unpad once / run the blocks on the token stream), Block.forward post-norm branch
// (sc/layers/block.py:389-463), FlashAttention.forward (sc/layers/attention.py:90-245), GatedMLP/MLP
// (sc/layers/mlp.py:30-34,68-83) and BiEncoder pooling (sc/models/biencoder/modeling_biencoder.py:287-319).
// The reference issues ~128 python model invocations per optimizer step; here each is a
// single host call and the kernels queue back-to-back.
#include <hip/hip_runtime.h>
#include "../../include/contrastors_hip.h"

namespace {

#define CX_TRY(expr)              \
    do {                          \
        int rc__ = (expr);        \
        if (rc__ != CX_OK) return rc__; \
    } while (0)

inline long round_up(long v, long m) { return (v + m - 1) / m * m; }

// CxChunkBuffers.layer_events: block l's (or, at index n_layer, the embeddings') parameter gradients are complete
inline int mark_grads_done(const CxChunkBuffers* buf, int idx, void* stream) {
    if (!buf->layer_events || !buf->layer_events[idx]) return CX_OK;
    return hipEventRecord((hipEvent_t)buf->layer_events[idx], (hipStream_t)stream) == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
}

// Slot mapping.  mode 0: every layer shares slot 0 (no-grad pass); 1: one slot per layer (saved for backward);
// 2 (activation checkpointing, sc/models/encoder/modeling_nomic_bert.py:339-365, sc/models/vit/vit.py:200-231): only the
// tensor that carries a block's INPUT keeps one slot per layer -- h2 (the previous block's output) for post-norm trunks,
// z1 (the complete residual stream at LN1) for pre-norm trunks -- everything else lives in slot 0 and is recomputed
// block by block during backward.
// Selective checkpointing (round 3; CxChunkBuffers.ckpt_keep = k): a recipe's `gradient_checkpointing: true` was written
// for 80 GB parts; on 288 GB most of a step's activations fit.  The TOP k blocks (l >= L - k) keep all their
// intermediates in slots 1 .. k exactly as mode 1 would, only the blocks below them share slot 0 and are recomputed:
// k / L of the re-forward disappears, the results stay bit-identical (the same kernels on the same data either way).
struct Slots {
    const CxChunkBuffers* b;
    const CxEncoderDesc* e;
    long T_cap;
    int d, I, wfc1;  // wfc1 = width of the fc1 output (2I gated, I plain) = width of its gradient in g_wide
    int mode;
    int first_kept;  // mode 2: blocks l >= first_kept own slot 1 + l - first_kept; below it: slot 0, recomputed in backward
    bool kept(int s) const { return mode == 1 || (mode == 2 && s >= first_kept); }   // block s's intermediates survive the forward
    int sl(int s) const { return mode == 1 ? s : (mode == 2 && s >= first_kept) ? 1 + s - first_kept : 0; }
    int sl_in(int s) const { return mode == 0 ? 0 : s; }   // the per-layer tensor of the checkpointing mode
    uint16_t* qkv(int s) const { return b->qkv + (size_t)sl(s) * T_cap * 3 * d; }
    uint16_t* ctx(int s) const { return b->ctx + (size_t)sl(s) * T_cap * d; }
    float* lse(int s) const { return b->lse + (size_t)sl(s) * T_cap * e->n_head; }
    uint16_t* z1(int s) const { return b->z1 + (size_t)(e->prenorm ? sl_in(s) : sl(s)) * T_cap * d; }
    uint16_t* h1(int s) const { return b->h1 + (size_t)sl(s) * T_cap * d; }
    float* mean1(int s) const { return b->mean1 + (size_t)sl(s) * T_cap; }
    float* rstd1(int s) const { return b->rstd1 + (size_t)sl(s) * T_cap; }
    // what the MLP's first projection keeps for backward, (T, I) per slot: the biased pre-activation of the plain MLP, the
    // GATE of the gated one (round 3: y is recovered from act = y * silu(gate); cx_gemm_bf16_swiglu_gate)
    uint16_t* yg(int s) const { return b->yg + (size_t)sl(s) * T_cap * I; }
    uint16_t* act(int s) const { return b->act + (size_t)sl(s) * T_cap * I; }
    uint16_t* z2(int s) const { return b->z2 + (size_t)sl(s) * T_cap * d; }
    uint16_t* h2(int s) const { return b->h2 + (size_t)(e->prenorm ? sl(s) : sl_in(s)) * T_cap * d; }
    float* mean2(int s) const { return b->mean2 + (size_t)sl(s) * T_cap; }
    float* rstd2(int s) const { return b->rstd2 + (size_t)sl(s) * T_cap; }
};

// save: 0 no-grad pass, 1 saving forward / backward (the arena's `checkpoint` / `ckpt_keep` pick the slot mode)
Slots make_slots(const CxEncoderDesc* enc, const CxChunkBuffers* buf, int save) {
    const int L = enc->n_layer;
    int keep = buf->ckpt_keep < 0 ? 0 : buf->ckpt_keep > L ? L : buf->ckpt_keep;
    return Slots{buf, enc, buf->T_cap, enc->d, enc->d_inner, enc->gated ? 2 * enc->d_inner : enc->d_inner,
                 save ? (buf->checkpoint ? 2 : 1) : 0, L - keep};
}

int check_desc(const CxEncoderDesc* e, const CxChunkBuffers* b, int T) {
    if (!e || !b || !e->layers) return CX_ERR_ARG;
    if (e->n_layer <= 0 || e->d <= 0 || e->n_head <= 0) return CX_ERR_ARG;
    if (e->d != e->n_head * 64) return CX_ERR_SHAPE;           // head_dim 64 only
    if ((e->d % 64) != 0 || (e->d_inner % 64) != 0) return CX_ERR_SHAPE;
    if (b->T_cap < round_up(T, 64)) return CX_ERR_SHAPE;
    return CX_OK;
}

// split-K factor for a wgrad GEMM: enough workgroups to fill 256 CUs x 2 resident blocks
int wgrad_split(int out_f, int in_f, long Tp) {
    const long tiles = (long)((out_f + 127) / 128) * ((in_f + 127) / 128);
    const long nk = Tp / 64;
    long s = (512 + tiles - 1) / tiles;
    if (s > nk / 2) s = nk / 2;
    if (s < 1) s = 1;
    return (int)s;
}

// gW(out_f,in_f) += dY^T X.  Natural-layout (transposing-read) kernel when the feature counts tile exactly, otherwise
// two explicit transposes (zero padded to Tp) and the NT kernel.
int wgrad(const uint16_t* dY, int out_f, const uint16_t* X, int in_f, float* gW, const CxChunkBuffers* b, int T,
          void* stream) {
    if (!gW) return CX_OK;
    const int rc = cx_gemm_bf16_tn_accum(dY, X, gW, b->ws_f32, b->ws_floats, T, out_f, in_f, out_f, in_f, stream);
    if (rc != CX_ERR_SHAPE) return rc;
    if (!b->tr_a || !b->tr_b) return CX_ERR_ARG;   // (optional since ABI 8: arenas of towers whose feature counts tile by 256 omit them)
    const int Tp = (int)round_up(T, 64);
    CX_TRY(cx_transpose_bf16(dY, b->tr_a, T, out_f, out_f, Tp, Tp, stream));
    CX_TRY(cx_transpose_bf16(X, b->tr_b, T, in_f, in_f, Tp, Tp, stream));
    return cx_gemm_bf16_nt_accum(b->tr_a, b->tr_b, gW, b->ws_f32, b->ws_floats, out_f, in_f, Tp, Tp, Tp, stream);
}

// out = x W^T + bias (+ residual when the fused epilogue covers the shape).  *folded tells the caller whether the
// residual is already in `out` (then the LayerNorm that follows gets residual = NULL).
// `buf` (optional): its split-K workspace serves the few-tile, long-K case (small batches: cx_gemm_bf16_nt_splitk).
int proj_residual(const uint16_t* x, const uint16_t* W, const float* bias, const uint16_t* residual, uint16_t* out, int T,
                  int N, int K, bool* folded, void* stream, const CxChunkBuffers* buf = nullptr) {
    *folded = false;
    if (buf && buf->ws_f32 && (long)((T + 255) / 256) * ((N + 255) / 256) <= 64 && K >= 1536) {
        const int rc = cx_gemm_bf16_nt_splitk(x, W, out, bias, residual, buf->ws_f32, buf->ws_floats, T, N, K, K, K, N, N, stream);
        if (rc != CX_ERR_SHAPE) {
            *folded = residual != nullptr && rc == CX_OK;
            return rc;
        }
    }
    if (residual) {
        const int rc = cx_gemm_bf16_nt_residual(x, W, out, bias, residual, T, N, K, K, K, N, N, stream);
        if (rc != CX_ERR_SHAPE) {
            *folded = rc == CX_OK;
            return rc;
        }
    }
    return cx_gemm_bf16_nt(x, W, out, bias, T, N, K, K, K, N, 0, 1, 1.f, stream);
}

// true when cx_gemm_bf16_bias_gelu covers the fc1 shape (keep in sync with its checks)
bool gelu_fused_shape(int T, int N, int K) {
    return T > 0 && (K % 64) == 0 && (N % 8) == 0;
}

// ---- transformer blocks, forward.  h0: (T,d) input embeddings.  Returns the final hidden states in *h_final. ------
// post-norm (sc/layers/block.py:389-463):  h = LN1(attn(h) + h);  h = LN2(mlp(h) + h)
// pre-norm  (sc/layers/block.py:293-388):  r = x + r;  h = LN1(r);  x = attn(h);  r = x + r;  h = LN2(r);  x = mlp(h);
//           after the last block  h = ln_f(x + r)  (sc/models/vit/vit.py:253-263).
// Buffers in pre-norm mode: z1/z2 hold the residual stream r at the two LayerNorms (the GEMM producing x writes into
// the slot the LayerNorm then completes in place), h1/h2 the normalised inputs of attention / MLP.
struct BlockRunner {
    const CxEncoderDesc* enc;
    const CxChunkBuffers* buf;
    const Slots& s;
    const int32_t* cu_seqlens;
    int Bc, T, max_seqlen;
    void* stream;

    // (`residual`: added to the fc2 / out_proj output in the GEMM epilogue when possible; *folded reports it)
    int mlp_up(const CxLayerWeights& w, const uint16_t* x, int l, bool keep) const {
        const int d = enc->d, I = enc->d_inner;
        if (enc->gated) {
            // fc11 || fc12 + SwiGLU in one kernel; the gate is only written when backward needs it
            return cx_gemm_bf16_swiglu_gate(x, w.Wfc1, keep ? s.yg(l) : nullptr, s.act(l), T, I, d, d, d, I, I, stream);
        }
        // bias + erf-GELU in the GEMM epilogue; yg then holds the biased pre-activation, which backward reads with a
        // NULL bias.  Shapes the fused kernel does not cover take the two-kernel route (yg without the bias).
        const int rc = cx_gemm_bf16_bias_act(x, w.Wfc1, w.bfc1, keep ? s.yg(l) : nullptr, s.act(l), T, s.wfc1, d, d, d, s.wfc1,
                                             I, enc->mlp_act, stream);
        if (rc != CX_ERR_SHAPE) return rc;
        CX_TRY(cx_gemm_bf16_nt(x, w.Wfc1, s.yg(l), nullptr, T, s.wfc1, d, d, d, s.wfc1, 0, 1, 1.f, stream));
        return cx_bias_act_fwd(s.yg(l), w.bfc1, s.act(l), T, I, enc->mlp_act, stream);
    }
    int mlp(const CxLayerWeights& w, const uint16_t* x, int l, uint16_t* out, const uint16_t* residual, bool keep,
            bool* folded) const {
        CX_TRY(mlp_up(w, x, l, keep));
        return proj_residual(s.act(l), w.Wfc2, w.bfc2, residual, out, T, enc->d, enc->d_inner, folded, stream, buf);
    }
    int attn(const CxLayerWeights& w, const uint16_t* x, int l, uint16_t* out, const uint16_t* residual, bool* folded) const {
        const int d = enc->d;
        CX_TRY(cx_gemm_bf16_nt(x, w.Wqkv, s.qkv(l), w.bqkv, T, 3 * d, d, d, d, 3 * d, 0, 1, 1.f, stream));
        if (buf->drop_active && enc->attn_pdrop > 0.f) {  // dropout sites of a chunk: 2l, 2l+1 residual, 2L embeddings, 2L+1+l attention
            CX_TRY(cx_attn_varlen_dropout_fwd(s.qkv(l), cu_seqlens, enc->rot_cos, enc->rot_sin, s.ctx(l), s.lse(l), Bc,
                                              enc->n_head, T, max_seqlen, enc->softmax_scale, enc->attn_pdrop, buf->drop_seed,
                                              buf->drop_offset, (unsigned)(2 * enc->n_layer + 1 + l), stream));
        } else if (enc->rot_cos && max_seqlen > 128) {
            // long sequences: rotate q and k once, in place (the saved qkv is the rotated one; the backward knows), instead
            // of once per 128-row block inside the streaming attention kernels (18-23 % of their time at S = 2048)
            CX_TRY(cx_rotary_qkv_inplace(s.qkv(l), cu_seqlens, enc->rot_cos, enc->rot_sin, Bc, enc->n_head, T, max_seqlen, 1,
                                         stream));
            CX_TRY(cx_attn_varlen_fwd(s.qkv(l), cu_seqlens, nullptr, nullptr, s.ctx(l), s.lse(l), Bc, enc->n_head, T, max_seqlen,
                                      enc->softmax_scale, stream));
        } else {
            CX_TRY(cx_attn_varlen_fwd(s.qkv(l), cu_seqlens, enc->rot_cos, enc->rot_sin, s.ctx(l), s.lse(l), Bc, enc->n_head, T,
                                      max_seqlen, enc->softmax_scale, stream));
        }
        return proj_residual(s.ctx(l), w.Wout, w.bout, residual, out, T, d, d, folded, stream);
    }
    // one post-norm block: h_in -> h2(l).  keep = backward will read this block's intermediates
    int post_block(int l, const uint16_t* h_in, bool keep) const {
        const CxLayerWeights& w = enc->layers[l];
        const int d = enc->d;
        // z = sublayer output + residual: added in the projection's epilogue when the fused kernel covers the shape
        // (then z is complete in place and the LayerNorm reads one stream), else by the LayerNorm kernel, which
        // writes z only when backward needs it.
        bool f1 = false, f2 = false;
        if (drop()) {
            // resid_pdrop > 0 (training): the mask sits between the projection and the residual add, so nothing is folded
            // into the GEMM epilogue -- K5 with p > 0 does dropout + add + LayerNorm in one pass (sites 2l, 2l + 1)
            const float p = enc->resid_pdrop;
            CX_TRY(attn(w, h_in, l, s.z1(l), nullptr, &f1));
            CX_TRY(cx_dropout_add_layernorm_fwd(s.z1(l), h_in, w.ln1_g, w.ln1_b, s.h1(l), keep ? s.z1(l) : nullptr, s.mean1(l),
                                                s.rstd1(l), T, d, enc->ln_eps, p, buf->drop_seed, buf->drop_offset, 2 * l, stream));
            CX_TRY(mlp(w, s.h1(l), l, s.z2(l), nullptr, keep, &f2));
            return cx_dropout_add_layernorm_fwd(s.z2(l), s.h1(l), w.ln2_g, w.ln2_b, s.h2(l), keep ? s.z2(l) : nullptr, s.mean2(l),
                                                s.rstd2(l), T, d, enc->ln_eps, p, buf->drop_seed, buf->drop_offset, 2 * l + 1,
                                                stream);
        }
        CX_TRY(attn(w, h_in, l, s.z1(l), h_in, &f1));
        CX_TRY(cx_layernorm_fwd(s.z1(l), f1 ? nullptr : h_in, w.ln1_g, w.ln1_b, s.h1(l), (keep && !f1) ? s.z1(l) : nullptr,
                                s.mean1(l), s.rstd1(l), T, d, enc->ln_eps, stream));
        CX_TRY(mlp(w, s.h1(l), l, s.z2(l), s.h1(l), keep, &f2));
        return cx_layernorm_fwd(s.z2(l), f2 ? nullptr : s.h1(l), w.ln2_g, w.ln2_b, s.h2(l), (keep && !f2) ? s.z2(l) : nullptr,
                                s.mean2(l), s.rstd2(l), T, d, enc->ln_eps, stream);
    }
    bool drop() const { return buf->drop_active && enc->resid_pdrop > 0.f; }
    // one pre-norm block.  In: x = output of the previous sub-layer, r = residual stream (NULL once folded into x).
    // Out: *x_out (= z1(l+1) or zf), *r_out, *folded_out.  up_only: stop after the MLP's first projection (the
    // recomputation of a checkpointed block needs nothing beyond `act`).
    int pre_block(int l, const uint16_t* x, const uint16_t* r, bool r_folded, bool keep, bool up_only, const uint16_t** x_out,
                  const uint16_t** r_out, bool* folded_out) const {
        const CxLayerWeights& w = enc->layers[l];
        const int d = enc->d, L = enc->n_layer;
        // (when the previous fc2 folded the residual in, x == z1(l) already holds the complete residual stream; for the
        // first block x = h0 is copied into z1 by the kernel)
        CX_TRY(cx_layernorm_fwd(x, r, w.ln1_g, w.ln1_b, s.h1(l), (r_folded && x == s.z1(l)) ? nullptr : s.z1(l), s.mean1(l),
                                s.rstd1(l), T, d, enc->ln_eps, stream));
        bool f1 = false, f2 = false;
        CX_TRY(attn(w, s.h1(l), l, s.z2(l), s.z1(l), &f1));
        CX_TRY(cx_layernorm_fwd(s.z2(l), f1 ? nullptr : s.z1(l), w.ln2_g, w.ln2_b, s.h2(l), f1 ? nullptr : s.z2(l), s.mean2(l),
                                s.rstd2(l), T, d, enc->ln_eps, stream));
        if (up_only) return mlp_up(w, s.h2(l), l, keep);
        // the MLP output lands where the next LayerNorm completes it in place: the next block's z1 slot, or zf
        uint16_t* nxt = (l + 1 < L) ? s.z1(l + 1) : buf->zf;
        CX_TRY(mlp(w, s.h2(l), l, nxt, s.z2(l), keep, &f2));
        *x_out = nxt;
        *r_out = f2 ? nullptr : s.z2(l);   // already folded into x by the fc2 epilogue
        *folded_out = f2;
        return CX_OK;
    }
};

int blocks_forward(const CxEncoderDesc* enc, const CxChunkBuffers* buf, const Slots& s, const uint16_t* h0,
                   const int32_t* cu_seqlens, int Bc, int T, int max_seqlen, int save, const uint16_t** h_final,
                   void* stream) {
    const int d = enc->d, L = enc->n_layer;
    const BlockRunner run{enc, buf, s, cu_seqlens, Bc, T, max_seqlen, stream};
    // (checkpointing: a recomputed block keeps only its input, by the slot mapping; the blocks above first_kept keep all)
    if (!enc->prenorm) {
        const uint16_t* h_in = h0;
        for (int l = 0; l < L; ++l) {
            CX_TRY(run.post_block(l, h_in, save != 0 && s.kept(l)));
            h_in = s.h2(l);
        }
        *h_final = h_in;
        return CX_OK;
    }
    if (!buf->zf || !buf->hf || !buf->meanf || !buf->rstdf || !enc->lnf_g || !enc->lnf_b) return CX_ERR_ARG;
    if (buf->drop_active && enc->resid_pdrop > 0.f) return CX_ERR_ARG;   // dropout: post-norm text trunks only
    const uint16_t* x = h0;        // output of the previous sub-layer (the embeddings for the first block)
    const uint16_t* r = nullptr;   // residual stream
    bool r_folded = false;         // the residual stream was already added into x by a GEMM epilogue
    for (int l = 0; l < L; ++l) {
        CX_TRY(run.pre_block(l, x, r, r_folded, save != 0 && s.kept(l), false, &x, &r, &r_folded));
        // checkpointing keeps ONE tensor per block, the complete residual stream z1(l): needs the fc2 epilogue fold
        if (save == 2 && !r_folded) return CX_ERR_SHAPE;
    }
    CX_TRY(cx_layernorm_fwd(x, r, enc->lnf_g, enc->lnf_b, buf->hf, r_folded ? nullptr : buf->zf, buf->meanf, buf->rstdf, T,
                            d, enc->ln_eps, stream));
    *h_final = buf->hf;
    return CX_OK;
}

int check_bwd_buffers(const CxChunkBuffers* buf) {
    return (!buf->g_a || !buf->g_b || !buf->g_c || !buf->g_wide || !buf->g_act || !buf->delta ||
            !buf->ws_f32) ? CX_ERR_ARG : CX_OK;
}

// The natural-layout wgrad kernel reduces over round_up(T,64) token rows: clear the pad rows of every operand.
int clear_pad_rows(const CxEncoderDesc* enc, const CxChunkBuffers* buf, const Slots& s, int T, void* stream) {
    const int Tp = (int)round_up(T, 64);
    if (Tp == T) return CX_OK;
    const int d = enc->d, I = enc->d_inner, L = enc->n_layer;
    auto clear = [&](uint16_t* base, int width) {
        return hipMemsetAsync(base + (size_t)T * width, 0, (size_t)(Tp - T) * width * sizeof(uint16_t),
                              (hipStream_t)stream) == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
    };
    CX_TRY(clear(buf->g_a, d));
    CX_TRY(clear(buf->g_b, d));
    CX_TRY(clear(buf->g_c, d));
    if (buf->g_d) CX_TRY(clear(buf->g_d, d));
    CX_TRY(clear(buf->g_wide, 3 * d));   // used as (T,3d) and as (T,wfc1): clear for both widths
    CX_TRY(clear(buf->g_wide, s.wfc1));
    CX_TRY(clear(buf->h0, d));
    for (int l = 0; l < L; ++l) {
        CX_TRY(clear(s.act(l), I));
        CX_TRY(clear(s.h1(l), d));
        CX_TRY(clear(s.ctx(l), d));
        CX_TRY(clear(s.h2(l), d));
    }
    return CX_OK;
}

// The gradient of a POOLED encoder's final hidden states is a rank-one pattern, w(t) * g[seq(t)] (mean / cls pooling): when
// `pg` is given, the backward of the last LayerNorm builds it on the fly in fp32 (cx_layernorm_bwd_pooled) instead of reading
// a bf16 copy from buf->g_a -- the rounding of that copy was the largest parity gap of round 2 (final-LayerNorm bias).
struct PooledGrad {
    const float* demb;
    const float* emb;
    const float* norm;
    int pool_mode, normalize;
};

// ---- transformer blocks, backward.  In: gradient of the final hidden states in buf->g_a (or `pg`, see above).  Out: the
// gradient of the input embeddings as the sum of *da and *db (db may come back NULL). ---------------------------------
int blocks_backward(const CxEncoderDesc* enc, const CxChunkBuffers* buf, const Slots& s, const int32_t* cu_seqlens, int Bc,
                    int T, int max_seqlen, const PooledGrad* pg, const uint16_t** da_out, const uint16_t** db_out, void* stream) {
    const int d = enc->d, I = enc->d_inner, H = enc->n_head, L = enc->n_layer;
    // MLP backward: dm -> gradients of fc2 / fc1 parameters, d(mlp input) into buf->g_b
    // `add` (optional): the residual-branch gradient that the following LayerNorm backward would add to this dgrad
    // output; folded into the last GEMM's epilogue when the fused kernel covers the shape (*folded).
    // (bias_done: the LayerNorm backward that produced dm / dx already accumulated its column sums = this bias gradient)
    auto mlp_bwd = [&](const CxLayerWeights& w, int l, const uint16_t* dm, const uint16_t* mlp_in, const uint16_t* add,
                       bool* folded, bool bias_done) -> int {
        if (w.gbfc2 && !bias_done) CX_TRY(cx_bias_grad(dm, w.gbfc2, T, d, d, stream));
        CX_TRY(wgrad(dm, d, s.act(l), I, w.gWfc2, buf, T, stream));
        int fused = CX_ERR_SHAPE;
        if (enc->gated)  // fc2 dgrad + SwiGLU backward in one kernel: d(act) never touches HBM
            fused = cx_gemm_bf16_swiglu_bwd_gate(dm, w.Wfc2T, s.act(l), s.yg(l), buf->g_wide, T, I, d, d, d, I, s.wfc1, stream);
        // plain MLP whose forward took the fused bias + activation kernel (yg = the biased pre-activation): fc2 dgrad, the activation
        // backward and the fc1 bias gradient's partials in ONE kernel (round 6) -- d(act) never touches HBM either
#ifndef CX_AB_R5_ROUTES   // (evidence builds: scripts/build_variant.py r5routes attention.hip -DCX_AB_R5_ROUTES --- engine.hip -DCX_AB_R5_ROUTES)
        if (!enc->gated && gelu_fused_shape(T, s.wfc1, d))
            fused = cx_gemm_bf16_act_bwd(dm, w.Wfc2T, s.yg(l), buf->g_wide, w.gbfc1, buf->ws_f32, buf->ws_floats, T, I, d, d, d, I, I,
                                         enc->mlp_act, stream);
#endif
        if (fused != CX_ERR_SHAPE) {
            CX_TRY(fused);
        } else {
            CX_TRY(cx_gemm_bf16_nt(dm, w.Wfc2T, buf->g_act, nullptr, T, I, d, d, d, I, 0, 1, 1.f, stream));
        }
        if (enc->gated) {
            if (fused == CX_ERR_SHAPE)
                CX_TRY(cx_swiglu_bwd_gate(buf->g_act, s.act(l), s.yg(l), buf->g_wide, T, I, stream));
        } else if (fused == CX_ERR_SHAPE) {
            // (yg holds the biased pre-activation when the forward took the fused kernel: same predicate as there)
            // GELU backward and the fc1 bias gradient in one pass over (dact, pre)
            CX_TRY(cx_bias_act_bwd_colsum(buf->g_act, s.yg(l), gelu_fused_shape(T, s.wfc1, d) ? nullptr : w.bfc1, buf->g_wide,
                                          w.gbfc1, T, I, enc->mlp_act, stream));
        }
        CX_TRY(wgrad(buf->g_wide, s.wfc1, mlp_in, d, w.gWfc1, buf, T, stream));
        return proj_residual(buf->g_wide, w.Wfc1T, nullptr, add, buf->g_b, T, d, s.wfc1, folded, stream, buf);
    };
    // attention backward: dx (grad of the out_proj output) -> parameter gradients, d(attention input) into buf->g_b
    auto attn_bwd = [&](const CxLayerWeights& w, int l, const uint16_t* dx, const uint16_t* attn_in, const uint16_t* add,
                        bool* folded, bool bias_done) -> int {
        if (w.gbout && !bias_done) CX_TRY(cx_bias_grad(dx, w.gbout, T, d, d, stream));
        CX_TRY(wgrad(dx, d, s.ctx(l), d, w.gWout, buf, T, stream));
        CX_TRY(cx_gemm_bf16_nt(dx, w.WoutT, buf->g_b, nullptr, T, d, d, d, d, d, 0, 1, 1.f, stream));
        // attention core (+ inverse rotary)
        if (buf->drop_active && enc->attn_pdrop > 0.f) {
            CX_TRY(cx_attn_varlen_dropout_bwd(buf->g_b, s.qkv(l), s.ctx(l), s.lse(l), cu_seqlens, enc->rot_cos, enc->rot_sin,
                                              buf->delta, buf->g_wide, Bc, H, T, max_seqlen, enc->softmax_scale, enc->attn_pdrop,
                                              buf->drop_seed, buf->drop_offset, (unsigned)(2 * enc->n_layer + 1 + l), stream));
        } else if (enc->rot_cos && max_seqlen > 128) {   // (the forward rotated qkv in place)
            CX_TRY(cx_attn_varlen_bwd_prerotated(buf->g_b, s.qkv(l), s.ctx(l), s.lse(l), cu_seqlens, enc->rot_cos, enc->rot_sin,
                                                 buf->delta, buf->g_wide, Bc, H, T, max_seqlen, enc->softmax_scale, stream));
        } else {
            CX_TRY(cx_attn_varlen_bwd(buf->g_b, s.qkv(l), s.ctx(l), s.lse(l), cu_seqlens, enc->rot_cos, enc->rot_sin,
                                      buf->delta, buf->g_wide, Bc, H, T, max_seqlen, enc->softmax_scale, stream));
        }
        if (w.gbqkv) CX_TRY(cx_bias_grad(buf->g_wide, w.gbqkv, T, 3 * d, 3 * d, stream));
        CX_TRY(wgrad(buf->g_wide, 3 * d, attn_in, d, w.gWqkv, buf, T, stream));
        return proj_residual(buf->g_wide, w.WqkvT, nullptr, add, buf->g_b, T, d, 3 * d, folded, stream, buf);
    };
    // LayerNorm backward whose dz is the gradient of (Linear output + residual): `gbias` (the Linear's bias gradient, may be
    // NULL) rides along as the column sums of dz when the kernel has its workspace; *done says whether it did
    auto ln_bwd = [&](const uint16_t* a, const uint16_t* b2, const uint16_t* z, const float* g, const float* mean,
                      const float* rstd, const uint16_t* dz_extra, uint16_t* dz, float* gg, float* gb, float* gbias,
                      bool* done) -> int {
        *done = false;
        if (gbias) {
            const int rc = cx_layernorm_bwd_colsum(a, b2, z, g, mean, rstd, dz_extra, dz, gg, gb, gbias, buf->ws_f32,
                                                   buf->ws_floats, T, d, stream);
            if (rc != CX_ERR_ARG) {
                *done = rc == CX_OK;
                return rc;
            }
        }
        return cx_layernorm_bwd(a, b2, z, g, mean, rstd, dz_extra, dz, gg, gb, buf->ws_f32, buf->ws_floats, T, d, stream);
    };
    // activation checkpointing (slot mode 2): the block's intermediates are recomputed from its saved input into slot 0
    // right before its backward (bit-identical: every kernel on the path is deterministic)
    const BlockRunner run{enc, buf, s, cu_seqlens, Bc, T, max_seqlen, stream};
    if (!enc->prenorm) {
        const uint16_t* da = buf->g_a;
        const uint16_t* db = nullptr;
        for (int l = L - 1; l >= 0; --l) {
            const CxLayerWeights& w = enc->layers[l];
            const uint16_t* h_in = (l == 0) ? buf->h0 : s.h2(l - 1);
            if (!s.kept(l)) CX_TRY(run.post_block(l, h_in, true));
            if (run.drop()) {
                // dropout between every sub-layer output and the residual add: the LayerNorm backward returns the
                // residual's gradient (dz) and the masked, rescaled gradient of the sub-layer output (dx0, in g_d)
                if (!buf->g_d) return CX_ERR_ARG;
                const float p = enc->resid_pdrop;
                bool f1 = false, f2 = false;
                // (round 6) the masked gradient g_d IS the gradient of the sub-layer's Linear output: its column sums = that Linear's bias
                // gradient ride along in the LayerNorm backward when the arena's workspace holds the third partial vector
#ifdef CX_AB_R5_ROUTES
                const bool cs = false;
#else
                const bool cs = buf->ws_f32 && buf->ws_floats >= 3L * d * 256;
#endif
                CX_TRY(cx_dropout_add_layernorm_bwd_colsum(da, db, s.z2(l), w.ln2_g, s.mean2(l), s.rstd2(l), buf->g_c, buf->g_d, w.gln2_g,
                                                           w.gln2_b, cs ? w.gbfc2 : nullptr, buf->ws_f32, buf->ws_floats, T, d, p,
                                                           buf->drop_seed, buf->drop_offset, 2 * l + 1, stream));
                CX_TRY(mlp_bwd(w, l, buf->g_d, s.h1(l), buf->g_c, &f1, cs));   // -> g_b = d h1 (+ dz2 when folded)
                CX_TRY(cx_dropout_add_layernorm_bwd_colsum(f1 ? buf->g_b : buf->g_c, f1 ? nullptr : buf->g_b, s.z1(l), w.ln1_g, s.mean1(l),
                                                           s.rstd1(l), buf->g_a, buf->g_d, w.gln1_g, w.gln1_b, cs ? w.gbout : nullptr,
                                                           buf->ws_f32, buf->ws_floats, T, d, p, buf->drop_seed, buf->drop_offset, 2 * l,
                                                           stream));
                CX_TRY(attn_bwd(w, l, buf->g_d, h_in, buf->g_a, &f2, cs));
                da = f2 ? buf->g_b : buf->g_a;
                db = f2 ? nullptr : buf->g_b;
                CX_TRY(mark_grads_done(buf, l, stream));
                continue;
            }
            // LN2: dz2 = grad of (mlp_out + h1); its column sums are fc2's bias gradient
            bool b2_done = false, b1_done = false;
            if (pg && l == L - 1) {
                const bool cs = w.gbfc2 && buf->ws_f32 && buf->ws_floats >= 3L * d * 256;
                CX_TRY(cx_layernorm_bwd_pooled(pg->demb, pg->emb, pg->norm, cu_seqlens, Bc, pg->pool_mode, pg->normalize, s.z2(l),
                                               w.ln2_g, s.mean2(l), s.rstd2(l), buf->g_c, w.gln2_g, w.gln2_b,
                                               cs ? w.gbfc2 : nullptr, buf->ws_f32, buf->ws_floats, T, d, stream));
                b2_done = cs;
            } else {
                CX_TRY(ln_bwd(da, db, s.z2(l), w.ln2_g, s.mean2(l), s.rstd2(l), nullptr, buf->g_c, w.gln2_g, w.gln2_b, w.gbfc2,
                              &b2_done));
            }
            bool f1 = false, f2 = false;
            CX_TRY(mlp_bwd(w, l, buf->g_c, s.h1(l), buf->g_c, &f1, b2_done));
            // LN1: dout = dz2 (residual branch) + dh1 from the MLP (already summed in g_b when the fc1 dgrad folded it);
            // the column sums of its dz are out_proj's bias gradient
            CX_TRY(ln_bwd(f1 ? buf->g_b : buf->g_c, f1 ? nullptr : buf->g_b, s.z1(l), w.ln1_g, s.mean1(l), s.rstd1(l), nullptr,
                          buf->g_a, w.gln1_g, w.gln1_b, w.gbout, &b1_done));
            CX_TRY(attn_bwd(w, l, buf->g_a, h_in, buf->g_a, &f2, b1_done));
            da = f2 ? buf->g_b : buf->g_a;  // dz1 (residual branch into h_in) [+ the attention branch when folded]
            db = f2 ? nullptr : buf->g_b;   // attention branch into h_in
            CX_TRY(mark_grads_done(buf, l, stream));
        }
        *da_out = da;
        *db_out = db;
        return CX_OK;
    }
    // pre-norm: the gradient of the residual stream r rides along as dz_extra of every LayerNorm backward.  g_c (= dz of
    // ln_f, then of each block's LN1) is the gradient of the fc2 output of the block BELOW it in the loop, g_a (= dz of LN2)
    // that of the out_proj output of the same block: their column sums are those bias gradients.
    bool bias_c_done = false;   // "the kernel that wrote g_c accumulated layers[l].gbfc2"
    {
        float* gb_top = enc->layers[L - 1].gbfc2;
        if (pg) {
            const bool cs = gb_top && buf->ws_f32 && buf->ws_floats >= 3L * d * 256;
            CX_TRY(cx_layernorm_bwd_pooled(pg->demb, pg->emb, pg->norm, cu_seqlens, Bc, pg->pool_mode, pg->normalize, buf->zf,
                                           enc->lnf_g, buf->meanf, buf->rstdf, buf->g_c, enc->glnf_g, enc->glnf_b,
                                           cs ? gb_top : nullptr, buf->ws_f32, buf->ws_floats, T, d, stream));
            bias_c_done = cs;
        } else {
            CX_TRY(ln_bwd(buf->g_a, nullptr, buf->zf, enc->lnf_g, buf->meanf, buf->rstdf, nullptr, buf->g_c, enc->glnf_g,
                          enc->glnf_b, gb_top, &bias_c_done));
        }
    }
    for (int l = L - 1; l >= 0; --l) {   // invariant: buf->g_c = d(x_l + r_l) = gradient of both the MLP output and r
        const CxLayerWeights& w = enc->layers[l];
        bool unused = false, bias_a_done = false;
        if (!s.kept(l)) {
            const uint16_t* xo = nullptr;
            const uint16_t* ro = nullptr;
            bool fo = false;
            CX_TRY(run.pre_block(l, s.z1(l), nullptr, true, true, /*up_only*/ true, &xo, &ro, &fo));
        }
        CX_TRY(mlp_bwd(w, l, buf->g_c, s.h2(l), nullptr, &unused, bias_c_done));    // -> g_b = d h2
        CX_TRY(ln_bwd(buf->g_b, nullptr, s.z2(l), w.ln2_g, s.mean2(l), s.rstd2(l), /*dz_extra*/ buf->g_c, buf->g_a, w.gln2_g,
                      w.gln2_b, w.gbout, &bias_a_done));
        CX_TRY(attn_bwd(w, l, buf->g_a, s.h1(l), nullptr, &unused, bias_a_done));   // -> g_b = d h1
        CX_TRY(ln_bwd(buf->g_b, nullptr, s.z1(l), w.ln1_g, s.mean1(l), s.rstd1(l), /*dz_extra*/ buf->g_a, buf->g_c, w.gln1_g,
                      w.gln1_b, l > 0 ? enc->layers[l - 1].gbfc2 : nullptr, &bias_c_done));
        CX_TRY(mark_grads_done(buf, l, stream));
    }
    *da_out = buf->g_c;
    *db_out = nullptr;
    return CX_OK;
}

}  // namespace

namespace {
// pooled embeddings (emb_out) or, for poolers that live above the C-ABI (sc MultiHeadAttentionPooling, `pooling: map`),
// the final hidden states themselves (hidden_out, (T, d) bf16 after ln_f)
int vit_forward_impl(const CxEncoderDesc* enc, const CxChunkBuffers* buf, const void* pixels, int pixels_bf16,
                     const int32_t* cu_seqlens, int Bc, int Cc, int H, int W, int patch, int save_for_backward, float* emb_out,
                     uint16_t* hidden_out, void* stream) {
    if (Bc <= 0) return CX_OK;
    if (!enc || !buf || patch <= 0 || (H % patch) || (W % patch)) return CX_ERR_ARG;
    // PatchDropout (CxChunkBuffers.patch_keep): only the kept patches of every image exist from here on
    const int P_all = (H / patch) * (W / patch);
    const int32_t* keep = buf ? buf->patch_keep : nullptr;
    if (keep && (buf->n_keep <= 0 || buf->n_keep > P_all || buf->n_patch_all != P_all)) return CX_ERR_ARG;
    const int P = keep ? buf->n_keep : P_all, S = P + 1, T = Bc * S;
    CX_TRY(check_desc(enc, buf, T));
    if (!enc->Wpatch || !enc->cls_token || !enc->vit_pos || !buf->patch_in || !buf->patch_proj) return CX_ERR_ARG;
    if (enc->patch_dim != Cc * patch * patch || (enc->patch_dim % 64) != 0) return CX_ERR_SHAPE;
    (void)hipGetLastError();
    const int d = enc->d, I = enc->d_inner;
    const Slots s = make_slots(enc, buf, save_for_backward);
    CX_TRY(cx_vit_patchify_gather(pixels, pixels_bf16, buf->patch_in, Bc, Cc, H, W, patch, keep, P, stream));
    CX_TRY(cx_gemm_bf16_nt(buf->patch_in, enc->Wpatch, buf->patch_proj, enc->bpatch, Bc * P, d, enc->patch_dim,
                           enc->patch_dim, enc->patch_dim, d, 0, 1, 1.f, stream));
    if (enc->lnpre_g) {   // CLIP flavour (sc/models/vit/vit.py:180): LayerNorm on [cls | patches] + pos before the first block
        if (!enc->lnpre_b || !buf->zpre) return CX_ERR_ARG;
        CX_TRY(cx_vit_assemble_fwd_gather(buf->patch_proj, enc->cls_token, enc->vit_pos, buf->zpre, Bc, P, d, keep, stream));
        CX_TRY(cx_layernorm_fwd(buf->zpre, nullptr, enc->lnpre_g, enc->lnpre_b, buf->h0, nullptr, buf->emb_mean, buf->emb_rstd, T, d,
                                enc->ln_eps, stream));
    } else {
        CX_TRY(cx_vit_assemble_fwd_gather(buf->patch_proj, enc->cls_token, enc->vit_pos, buf->h0, Bc, P, d, keep, stream));
    }
    const uint16_t* h_final = nullptr;
    CX_TRY(blocks_forward(enc, buf, s, buf->h0, cu_seqlens, Bc, T, S, s.mode, &h_final, stream));
    if (hidden_out)
        return hipMemcpyAsync(hidden_out, h_final, (size_t)T * d * sizeof(uint16_t), hipMemcpyDeviceToDevice,
                              (hipStream_t)stream) == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
    return cx_pool_normalize_fwd(h_final, cu_seqlens, emb_out, buf->pool_norm, Bc, d, enc->pool_mode, enc->normalize,
                                 stream);
}

int vit_backward_impl(const CxEncoderDesc* enc, const CxChunkBuffers* buf, const int32_t* cu_seqlens, int Bc, int n_patch,
                      const float* demb, const float* emb_out, const uint16_t* dhidden, void* stream) {
    if (Bc <= 0) return CX_OK;
    if (!enc || !buf || n_patch <= 0) return CX_ERR_ARG;
    const int P = n_patch, S = P + 1, T = Bc * S;
    CX_TRY(check_desc(enc, buf, T));
    if ((!dhidden && (!demb || !emb_out)) || !buf->patch_in || !buf->patch_proj) return CX_ERR_ARG;
    CX_TRY(check_bwd_buffers(buf));
    (void)hipGetLastError();
    const int d = enc->d, I = enc->d_inner;
    const Slots s = make_slots(enc, buf, 1);
    CX_TRY(clear_pad_rows(enc, buf, s, T, stream));
    const PooledGrad pg{demb, emb_out, buf->pool_norm, enc->pool_mode, enc->normalize};
    const bool fold_pool = !dhidden && enc->prenorm && (enc->pool_mode == 0 || enc->pool_mode == 1);   // (see cx_encoder_backward)
    if (dhidden) {
        if (hipMemcpyAsync(buf->g_a, dhidden, (size_t)T * d * sizeof(uint16_t), hipMemcpyDeviceToDevice,
                           (hipStream_t)stream) != hipSuccess)
            return CX_ERR_LAUNCH;
    } else if (!fold_pool) {
        CX_TRY(cx_pool_normalize_bwd(demb, emb_out, buf->pool_norm, cu_seqlens, buf->g_a, Bc, d, enc->pool_mode,
                                     enc->normalize, stream));
    }
    const uint16_t* da = nullptr;
    const uint16_t* db = nullptr;
    CX_TRY(blocks_backward(enc, buf, s, cu_seqlens, Bc, T, S, fold_pool ? &pg : nullptr, &da, &db, stream));
    if (db) return CX_ERR_ARG;  // (post-norm ViT would need the two branches summed first; no such model family)
    if (enc->lnpre_g) {   // through the pre-LayerNorm: da (= g_c) -> g_a
        if (!buf->zpre) return CX_ERR_ARG;
        CX_TRY(cx_layernorm_bwd(da, nullptr, buf->zpre, enc->lnpre_g, buf->emb_mean, buf->emb_rstd, nullptr, buf->g_a,
                                enc->glnpre_g, enc->glnpre_b, buf->ws_f32, buf->ws_floats, T, d, stream));
        da = buf->g_a;
    }
    // d(embeddings) -> cls / position gradients and the contiguous d(projection) rows; pad rows of both wgrad operands
    // (patch_in was written by the forward of this chunk and is still intact) are cleared for the 64-row reduction
    const int Tp = Bc * P, Tpp = (int)round_up(Tp, 64);
    if (Tpp > Tp) {
        if (hipMemsetAsync(buf->patch_proj + (size_t)Tp * d, 0, (size_t)(Tpp - Tp) * d * 2, (hipStream_t)stream) != hipSuccess ||
            hipMemsetAsync(buf->patch_in + (size_t)Tp * enc->patch_dim, 0, (size_t)(Tpp - Tp) * enc->patch_dim * 2,
                           (hipStream_t)stream) != hipSuccess)
            return CX_ERR_LAUNCH;
    }
    if (buf->patch_keep) {   // PatchDropout: n_patch is the kept count; position gradients land on the ORIGINAL positions
        if (!buf->patch_inv || buf->n_keep != P || buf->n_patch_all < P) return CX_ERR_ARG;
        CX_TRY(cx_vit_assemble_bwd_gather(da, buf->patch_proj, enc->gcls_token, enc->gvit_pos, Bc, P, d, buf->patch_inv,
                                          buf->n_patch_all, stream));
    } else {
        CX_TRY(cx_vit_assemble_bwd(da, buf->patch_proj, enc->gcls_token, enc->gvit_pos, Bc, P, d, stream));
    }
    if (enc->gbpatch) CX_TRY(cx_bias_grad(buf->patch_proj, enc->gbpatch, Tp, d, d, stream));
    CX_TRY(wgrad(buf->patch_proj, d, buf->patch_in, enc->patch_dim, enc->gWpatch, buf, Tp, stream));
    return mark_grads_done(buf, enc->n_layer, stream);
}
}  // namespace

extern "C" {

int cx_encoder_forward(const CxEncoderDesc* enc, const CxChunkBuffers* buf, const int64_t* input_ids,
                       const int32_t* indices, const int32_t* cu_seqlens, int Bc, int S, int T, int max_seqlen,
                       int save_for_backward, float* emb_out, void* stream) {
    if (Bc <= 0 || T <= 0) return CX_OK;
    CX_TRY(check_desc(enc, buf, T));
    (void)hipGetLastError();  // a stale error of some earlier, unrelated runtime call must not fail this launch train
    const int d = enc->d, I = enc->d_inner;
    const Slots s = make_slots(enc, buf, save_for_backward);
    CX_TRY(cx_embed_ln_fwd(input_ids, indices, enc->word_emb, enc->type_emb, enc->pos_emb, enc->emb_ln_g,
                           enc->emb_ln_b, buf->h0, buf->emb_mean, buf->emb_rstd, T, S, d, enc->ln_eps, stream));
    if (buf->drop_active && enc->embd_pdrop > 0.f)  // modeling_nomic_bert.py:534-535: dropout on the embedding-LN output
        CX_TRY(cx_dropout_scale(buf->h0, (long)T * d, enc->embd_pdrop, buf->drop_seed, buf->drop_offset, 2 * enc->n_layer, stream));
    const uint16_t* h_final = nullptr;
    CX_TRY(blocks_forward(enc, buf, s, buf->h0, cu_seqlens, Bc, T, max_seqlen, s.mode, &h_final, stream));
    return cx_pool_normalize_fwd(h_final, cu_seqlens, emb_out, buf->pool_norm, Bc, d, enc->pool_mode, enc->normalize,
                                 stream);
}

int cx_encoder_backward(const CxEncoderDesc* enc, const CxChunkBuffers* buf, const int64_t* input_ids,
                        const int32_t* indices, const int32_t* cu_seqlens, int Bc, int S, int T, int max_seqlen,
                        const float* demb, const float* emb_out, const int32_t* sort_ids, const int32_t* sort_perm,
                        void* stream) {
    if (Bc <= 0 || T <= 0) return CX_OK;
    CX_TRY(check_desc(enc, buf, T));
    if (!demb || !emb_out) return CX_ERR_ARG;
    CX_TRY(check_bwd_buffers(buf));
    (void)hipGetLastError();
    const int d = enc->d, I = enc->d_inner;
    const Slots s = make_slots(enc, buf, 1);
    CX_TRY(clear_pad_rows(enc, buf, s, T, stream));
    // the pooling backward rides inside the last LayerNorm's backward (fp32, never materialised); the dropout schedule
    // keeps the two-kernel form (its LayerNorm backward also returns the masked branch gradient)
    const PooledGrad pg{demb, emb_out, buf->pool_norm, enc->pool_mode, enc->normalize};
    const bool fold_pool = !(buf->drop_active && enc->resid_pdrop > 0.f) && (enc->pool_mode == 0 || enc->pool_mode == 1);
    if (!fold_pool)
        CX_TRY(cx_pool_normalize_bwd(demb, emb_out, buf->pool_norm, cu_seqlens, buf->g_a, Bc, d, enc->pool_mode,
                                     enc->normalize, stream));
    const uint16_t* da = fold_pool ? nullptr : buf->g_a;
    const uint16_t* db = nullptr;
    CX_TRY(blocks_backward(enc, buf, s, cu_seqlens, Bc, T, max_seqlen, fold_pool ? &pg : nullptr, &da, &db, stream));
    if (buf->drop_active && enc->embd_pdrop > 0.f) {  // gradient through the embedding dropout (both branches: linear)
        CX_TRY(cx_dropout_scale(const_cast<uint16_t*>(da), (long)T * d, enc->embd_pdrop, buf->drop_seed, buf->drop_offset,
                                2 * enc->n_layer, stream));
        if (db)
            CX_TRY(cx_dropout_scale(const_cast<uint16_t*>(db), (long)T * d, enc->embd_pdrop, buf->drop_seed, buf->drop_offset,
                                    2 * enc->n_layer, stream));
    }
    // word rows: deterministic segmented reduction when the host supplied the sorted token order (g_wide is free by now:
    // (T, >= 3d) bf16 = room for the (T, d) fp32 row gradients), fp32 atomics otherwise
    if (sort_ids && sort_perm) {
        CX_TRY(cx_embed_ln_bwd_sorted(da, db, input_ids, indices, enc->word_emb, enc->type_emb, enc->pos_emb, enc->emb_ln_g,
                                      buf->emb_mean, buf->emb_rstd, enc->gword_emb, enc->gtype_emb, enc->gpos_emb,
                                      enc->gemb_ln_g, enc->gemb_ln_b, T, S, d, enc->padding_idx, enc->vocab, sort_ids,
                                      sort_perm, reinterpret_cast<float*>(buf->g_wide), stream));
    } else {
        CX_TRY(cx_embed_ln_bwd(da, db, input_ids, indices, enc->word_emb, enc->type_emb, enc->pos_emb, enc->emb_ln_g,
                               buf->emb_mean, buf->emb_rstd, enc->gword_emb, enc->gtype_emb, enc->gpos_emb,
                               enc->gemb_ln_g, enc->gemb_ln_b, T, S, d, enc->padding_idx, stream));
    }
    return mark_grads_done(buf, enc->n_layer, stream);
}

// ---- token-level outputs (MLM head, sc/models/encoder/modeling_nomic_bert.py:590-669): same trunk, no pooling ------
int cx_encoder_forward_hidden(const CxEncoderDesc* enc, const CxChunkBuffers* buf, const int64_t* input_ids,
                              const int32_t* indices, const int32_t* cu_seqlens, int Bc, int S, int T, int max_seqlen,
                              int save_for_backward, uint16_t* hidden_out, void* stream) {
    if (Bc <= 0 || T <= 0) return CX_OK;
    CX_TRY(check_desc(enc, buf, T));
    if (!hidden_out) return CX_ERR_ARG;
    (void)hipGetLastError();
    const int d = enc->d, I = enc->d_inner;
    const Slots s = make_slots(enc, buf, save_for_backward);
    CX_TRY(cx_embed_ln_fwd(input_ids, indices, enc->word_emb, enc->type_emb, enc->pos_emb, enc->emb_ln_g,
                           enc->emb_ln_b, buf->h0, buf->emb_mean, buf->emb_rstd, T, S, d, enc->ln_eps, stream));
    if (buf->drop_active && enc->embd_pdrop > 0.f)  // modeling_nomic_bert.py:534-535: dropout on the embedding-LN output
        CX_TRY(cx_dropout_scale(buf->h0, (long)T * d, enc->embd_pdrop, buf->drop_seed, buf->drop_offset, 2 * enc->n_layer, stream));
    const uint16_t* h_final = nullptr;
    CX_TRY(blocks_forward(enc, buf, s, buf->h0, cu_seqlens, Bc, T, max_seqlen, s.mode, &h_final, stream));
    return hipMemcpyAsync(hidden_out, h_final, (size_t)T * d * sizeof(uint16_t), hipMemcpyDeviceToDevice,
                          (hipStream_t)stream) == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
}

int cx_encoder_backward_hidden(const CxEncoderDesc* enc, const CxChunkBuffers* buf, const int64_t* input_ids,
                               const int32_t* indices, const int32_t* cu_seqlens, int Bc, int S, int T, int max_seqlen,
                               const uint16_t* dhidden, const int32_t* sort_ids, const int32_t* sort_perm, void* stream) {
    if (Bc <= 0 || T <= 0) return CX_OK;
    CX_TRY(check_desc(enc, buf, T));
    if (!dhidden) return CX_ERR_ARG;
    CX_TRY(check_bwd_buffers(buf));
    (void)hipGetLastError();
    const int d = enc->d, I = enc->d_inner;
    const Slots s = make_slots(enc, buf, 1);
    CX_TRY(clear_pad_rows(enc, buf, s, T, stream));
    if (hipMemcpyAsync(buf->g_a, dhidden, (size_t)T * d * sizeof(uint16_t), hipMemcpyDeviceToDevice,
                       (hipStream_t)stream) != hipSuccess)
        return CX_ERR_LAUNCH;
    const uint16_t* da = nullptr;
    const uint16_t* db = nullptr;
    CX_TRY(blocks_backward(enc, buf, s, cu_seqlens, Bc, T, max_seqlen, nullptr, &da, &db, stream));
    if (buf->drop_active && enc->embd_pdrop > 0.f) {  // gradient through the embedding dropout (both branches: linear)
        CX_TRY(cx_dropout_scale(const_cast<uint16_t*>(da), (long)T * d, enc->embd_pdrop, buf->drop_seed, buf->drop_offset,
                                2 * enc->n_layer, stream));
        if (db)
            CX_TRY(cx_dropout_scale(const_cast<uint16_t*>(db), (long)T * d, enc->embd_pdrop, buf->drop_seed, buf->drop_offset,
                                    2 * enc->n_layer, stream));
    }
    // word rows: deterministic segmented reduction when the host supplied the sorted token order (g_wide is free by now:
    // (T, >= 3d) bf16 = room for the (T, d) fp32 row gradients), fp32 atomics otherwise
    if (sort_ids && sort_perm) {
        CX_TRY(cx_embed_ln_bwd_sorted(da, db, input_ids, indices, enc->word_emb, enc->type_emb, enc->pos_emb, enc->emb_ln_g,
                                      buf->emb_mean, buf->emb_rstd, enc->gword_emb, enc->gtype_emb, enc->gpos_emb,
                                      enc->gemb_ln_g, enc->gemb_ln_b, T, S, d, enc->padding_idx, enc->vocab, sort_ids,
                                      sort_perm, reinterpret_cast<float*>(buf->g_wide), stream));
    } else {
        CX_TRY(cx_embed_ln_bwd(da, db, input_ids, indices, enc->word_emb, enc->type_emb, enc->pos_emb, enc->emb_ln_g,
                               buf->emb_mean, buf->emb_rstd, enc->gword_emb, enc->gtype_emb, enc->gpos_emb,
                               enc->gemb_ln_g, enc->gemb_ln_b, T, S, d, enc->padding_idx, stream));
    }
    return mark_grads_done(buf, enc->n_layer, stream);
}

// ---- ViT image tower -------------------------------------------------------------------------------------------------
int cx_vit_forward(const CxEncoderDesc* enc, const CxChunkBuffers* buf, const void* pixels, int pixels_bf16,
                   const int32_t* cu_seqlens, int Bc, int Cc, int H, int W, int patch, int save_for_backward,
                   float* emb_out, void* stream) {
    if (Bc > 0 && !emb_out) return CX_ERR_ARG;
    return vit_forward_impl(enc, buf, pixels, pixels_bf16, cu_seqlens, Bc, Cc, H, W, patch, save_for_backward, emb_out, nullptr,
                            stream);
}

int cx_vit_backward(const CxEncoderDesc* enc, const CxChunkBuffers* buf, const int32_t* cu_seqlens, int Bc, int n_patch,
                    const float* demb, const float* emb_out, void* stream) {
    return vit_backward_impl(enc, buf, cu_seqlens, Bc, n_patch, demb, emb_out, nullptr, stream);
}

// token-level outputs of the image tower (the hidden states after ln_f), for poolers above the C-ABI
int cx_vit_forward_hidden(const CxEncoderDesc* enc, const CxChunkBuffers* buf, const void* pixels, int pixels_bf16,
                          const int32_t* cu_seqlens, int Bc, int Cc, int H, int W, int patch, int save_for_backward,
                          uint16_t* hidden_out, void* stream) {
    if (Bc > 0 && !hidden_out) return CX_ERR_ARG;
    return vit_forward_impl(enc, buf, pixels, pixels_bf16, cu_seqlens, Bc, Cc, H, W, patch, save_for_backward, nullptr, hidden_out,
                            stream);
}

int cx_vit_backward_hidden(const CxEncoderDesc* enc, const CxChunkBuffers* buf, const int32_t* cu_seqlens, int Bc, int n_patch,
                           const uint16_t* dhidden, void* stream) {
    if (Bc > 0 && !dhidden) return CX_ERR_ARG;
    return vit_backward_impl(enc, buf, cu_seqlens, Bc, n_patch, nullptr, nullptr, dhidden, stream);
}

int cx_abi_version(void) { return 10; }  // 10: cx_gemm_bf16_act_bwd; 9: cx_infonce_fwd_argmax; 8: cx_cast_transpose_f32_to_bf16_batched / CxCastJob; 7: CxChunkBuffers.patch_keep / patch_inv / n_keep (PatchDropout);  // 2: CxChunkBuffers.checkpoint; 3: dropout state, sorted embedding backward; 4: CxEncoderDesc.attn_pdrop; 5: CxChunkBuffers.layer_events, cx_layernorm_bwd_pooled; 6: CxChunkBuffers.ckpt_keep
const char* cx_build_info(void) { return "contrastors_hip gfx950 " __DATE__ " " __VERSION__; }
const char* cx_error_string(int code) {
    switch (code) {
        case CX_OK: return "ok";
        case CX_ERR_SHAPE: return "unsupported shape";
        case CX_ERR_ALIGN: return "misaligned pointer or leading dimension";
        case CX_ERR_ARG: return "invalid argument";
        case CX_ERR_LAUNCH: return "kernel launch failed";
        default: return "unknown error";
    }
}

}  // extern "C"
