// engine.hip -- native encoder runtime: one C call enqueues a whole GradCache chunk forward or backward
// (embeddings -> L post-norm transformer blocks -> pooling -> L2 normalise) on a HIP stream.
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

struct Slots {
    const CxChunkBuffers* b;
    const CxEncoderDesc* e;
    long T_cap;
    int d, I, wfc1;  // wfc1 = width of the fc1 output (2I gated, I plain)
    uint16_t* qkv(int s) const { return b->qkv + (size_t)s * T_cap * 3 * d; }
    uint16_t* ctx(int s) const { return b->ctx + (size_t)s * T_cap * d; }
    float* lse(int s) const { return b->lse + (size_t)s * T_cap * e->n_head; }
    uint16_t* z1(int s) const { return b->z1 + (size_t)s * T_cap * d; }
    uint16_t* h1(int s) const { return b->h1 + (size_t)s * T_cap * d; }
    float* mean1(int s) const { return b->mean1 + (size_t)s * T_cap; }
    float* rstd1(int s) const { return b->rstd1 + (size_t)s * T_cap; }
    uint16_t* yg(int s) const { return b->yg + (size_t)s * T_cap * wfc1; }
    uint16_t* act(int s) const { return b->act + (size_t)s * T_cap * I; }
    uint16_t* z2(int s) const { return b->z2 + (size_t)s * T_cap * d; }
    uint16_t* h2(int s) const { return b->h2 + (size_t)s * T_cap * d; }
    float* mean2(int s) const { return b->mean2 + (size_t)s * T_cap; }
    float* rstd2(int s) const { return b->rstd2 + (size_t)s * T_cap; }
};

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
    if ((out_f % 256) == 0 && (in_f % 128) == 0)
        return cx_gemm_bf16_tn_accum(dY, X, gW, b->ws_f32, b->ws_floats, T, out_f, in_f, out_f, in_f, stream);
    const int Tp = (int)round_up(T, 64);
    CX_TRY(cx_transpose_bf16(dY, b->tr_a, T, out_f, out_f, Tp, Tp, stream));
    CX_TRY(cx_transpose_bf16(X, b->tr_b, T, in_f, in_f, Tp, Tp, stream));
    return cx_gemm_bf16_nt_accum(b->tr_a, b->tr_b, gW, b->ws_f32, b->ws_floats, out_f, in_f, Tp, Tp, Tp, stream);
}

}  // namespace

extern "C" {

int cx_encoder_forward(const CxEncoderDesc* enc, const CxChunkBuffers* buf, const int64_t* input_ids,
                       const int32_t* indices, const int32_t* cu_seqlens, int Bc, int S, int T, int max_seqlen,
                       int save_for_backward, float* emb_out, void* stream) {
    if (Bc <= 0 || T <= 0) return CX_OK;
    CX_TRY(check_desc(enc, buf, T));
    (void)hipGetLastError();  // a stale error of some earlier, unrelated runtime call must not fail this launch train
    const int d = enc->d, I = enc->d_inner, H = enc->n_head;
    Slots s{buf, enc, buf->T_cap, d, I, enc->gated ? 2 * I : I};

    CX_TRY(cx_embed_ln_fwd(input_ids, indices, enc->word_emb, enc->type_emb, enc->pos_emb, enc->emb_ln_g,
                           enc->emb_ln_b, buf->h0, buf->emb_mean, buf->emb_rstd, T, S, d, enc->ln_eps, stream));
    const uint16_t* h_in = buf->h0;
    for (int l = 0; l < enc->n_layer; ++l) {
        const CxLayerWeights& w = enc->layers[l];
        const int sl = save_for_backward ? l : 0;
        CX_TRY(cx_gemm_bf16_nt(h_in, w.Wqkv, s.qkv(sl), w.bqkv, T, 3 * d, d, d, d, 3 * d, 0, 1, 1.f, stream));
        CX_TRY(cx_attn_varlen_fwd(s.qkv(sl), cu_seqlens, enc->rot_cos, enc->rot_sin, s.ctx(sl), s.lse(sl), Bc, H, T,
                                  max_seqlen, enc->softmax_scale, stream));
        CX_TRY(cx_gemm_bf16_nt(s.ctx(sl), w.Wout, s.z1(sl), w.bout, T, d, d, d, d, d, 0, 1, 1.f, stream));
        // z (= attn_out + residual) is only kept for backward; the no-grad pass skips that 1/4 of the LN traffic
        CX_TRY(cx_layernorm_fwd(s.z1(sl), h_in, w.ln1_g, w.ln1_b, s.h1(sl), save_for_backward ? s.z1(sl) : nullptr,
                                s.mean1(sl), s.rstd1(sl), T, d, enc->ln_eps, stream));
        if (enc->gated) {
            // fc11 || fc12 + SwiGLU in one kernel; the pre-activation pair is only written when backward needs it
            CX_TRY(cx_gemm_bf16_swiglu(s.h1(sl), w.Wfc1, save_for_backward ? s.yg(sl) : nullptr, s.act(sl), T, I, d, d, d,
                                       s.wfc1, I, stream));
        } else {
            CX_TRY(cx_gemm_bf16_nt(s.h1(sl), w.Wfc1, s.yg(sl), nullptr, T, s.wfc1, d, d, d, s.wfc1, 0, 1, 1.f, stream));
            CX_TRY(cx_bias_gelu_fwd(s.yg(sl), w.bfc1, s.act(sl), T, I, stream));
        }
        CX_TRY(cx_gemm_bf16_nt(s.act(sl), w.Wfc2, s.z2(sl), w.bfc2, T, d, I, I, I, d, 0, 1, 1.f, stream));
        CX_TRY(cx_layernorm_fwd(s.z2(sl), s.h1(sl), w.ln2_g, w.ln2_b, s.h2(sl), save_for_backward ? s.z2(sl) : nullptr,
                                s.mean2(sl), s.rstd2(sl), T, d, enc->ln_eps, stream));
        h_in = s.h2(sl);
    }
    return cx_pool_normalize_fwd(h_in, cu_seqlens, emb_out, buf->pool_norm, Bc, d, enc->pool_mode, enc->normalize,
                                 stream);
}

int cx_encoder_backward(const CxEncoderDesc* enc, const CxChunkBuffers* buf, const int64_t* input_ids,
                        const int32_t* indices, const int32_t* cu_seqlens, int Bc, int S, int T, int max_seqlen,
                        const float* demb, const float* emb_out, void* stream) {
    if (Bc <= 0 || T <= 0) return CX_OK;
    CX_TRY(check_desc(enc, buf, T));
    if (!demb || !emb_out || !buf->g_a || !buf->g_b || !buf->g_c || !buf->g_wide || !buf->g_act || !buf->tr_a ||
        !buf->tr_b || !buf->delta || !buf->ws_f32)
        return CX_ERR_ARG;
    (void)hipGetLastError();
    const int d = enc->d, I = enc->d_inner, H = enc->n_head, L = enc->n_layer;
    Slots s{buf, enc, buf->T_cap, d, I, enc->gated ? 2 * I : I};

    // The natural-layout wgrad kernel reduces over round_up(T,64) token rows: clear the pad rows of every operand.
    const int Tp = (int)round_up(T, 64);
    if (Tp > T) {
        auto clear = [&](uint16_t* base, int width) {
            return hipMemsetAsync(base + (size_t)T * width, 0, (size_t)(Tp - T) * width * sizeof(uint16_t),
                                  (hipStream_t)stream) == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
        };
        CX_TRY(clear(buf->g_a, d));
        CX_TRY(clear(buf->g_b, d));
        CX_TRY(clear(buf->g_c, d));
        CX_TRY(clear(buf->g_wide, 3 * d));   // used as (T,3d) and as (T,wfc1): clear for both widths
        CX_TRY(clear(buf->g_wide, s.wfc1));
        CX_TRY(clear(buf->h0, d));
        for (int l = 0; l < L; ++l) {
            CX_TRY(clear(s.act(l), I));
            CX_TRY(clear(s.h1(l), d));
            CX_TRY(clear(s.ctx(l), d));
            CX_TRY(clear(s.h2(l), d));
        }
    }
    CX_TRY(cx_pool_normalize_bwd(demb, emb_out, buf->pool_norm, cu_seqlens, buf->g_a, Bc, d, enc->pool_mode,
                                 enc->normalize, stream));
    const uint16_t* da = buf->g_a;
    const uint16_t* db = nullptr;
    for (int l = L - 1; l >= 0; --l) {
        const CxLayerWeights& w = enc->layers[l];
        const uint16_t* h_in = (l == 0) ? buf->h0 : s.h2(l - 1);
        // LN2: dz2 = grad of (mlp_out + h1)
        CX_TRY(cx_layernorm_bwd(da, db, s.z2(l), w.ln2_g, s.mean2(l), s.rstd2(l), nullptr, buf->g_c, w.gln2_g,
                                w.gln2_b, buf->ws_f32, buf->ws_floats, T, d, stream));
        // fc2
        if (w.gbfc2) CX_TRY(cx_bias_grad(buf->g_c, w.gbfc2, T, d, d, stream));
        CX_TRY(wgrad(buf->g_c, d, s.act(l), I, w.gWfc2, buf, T, stream));
        CX_TRY(cx_gemm_bf16_nt(buf->g_c, w.Wfc2T, buf->g_act, nullptr, T, I, d, d, d, I, 0, 1, 1.f, stream));
        // activation
        if (enc->gated) {
            CX_TRY(cx_swiglu_bwd(buf->g_act, s.yg(l), buf->g_wide, T, I, /*interleaved*/ 1, stream));
        } else {
            CX_TRY(cx_bias_gelu_bwd(buf->g_act, s.yg(l), w.bfc1, buf->g_wide, T, I, stream));
            if (w.gbfc1) CX_TRY(cx_bias_grad(buf->g_wide, w.gbfc1, T, I, I, stream));
        }
        // fc1
        CX_TRY(wgrad(buf->g_wide, s.wfc1, s.h1(l), d, w.gWfc1, buf, T, stream));
        CX_TRY(cx_gemm_bf16_nt(buf->g_wide, w.Wfc1T, buf->g_b, nullptr, T, d, s.wfc1, s.wfc1, s.wfc1, d, 0, 1, 1.f,
                               stream));
        // LN1: dout = dz2 (residual branch) + dh1 from the MLP
        CX_TRY(cx_layernorm_bwd(buf->g_c, buf->g_b, s.z1(l), w.ln1_g, s.mean1(l), s.rstd1(l), nullptr, buf->g_a,
                                w.gln1_g, w.gln1_b, buf->ws_f32, buf->ws_floats, T, d, stream));
        // out_proj
        if (w.gbout) CX_TRY(cx_bias_grad(buf->g_a, w.gbout, T, d, d, stream));
        CX_TRY(wgrad(buf->g_a, d, s.ctx(l), d, w.gWout, buf, T, stream));
        CX_TRY(cx_gemm_bf16_nt(buf->g_a, w.WoutT, buf->g_b, nullptr, T, d, d, d, d, d, 0, 1, 1.f, stream));
        // attention core (+ inverse rotary)
        CX_TRY(cx_attn_varlen_bwd(buf->g_b, s.qkv(l), s.ctx(l), s.lse(l), cu_seqlens, enc->rot_cos, enc->rot_sin,
                                  buf->delta, buf->g_wide, Bc, H, T, max_seqlen, enc->softmax_scale, stream));
        // Wqkv
        if (w.gbqkv) CX_TRY(cx_bias_grad(buf->g_wide, w.gbqkv, T, 3 * d, 3 * d, stream));
        CX_TRY(wgrad(buf->g_wide, 3 * d, h_in, d, w.gWqkv, buf, T, stream));
        CX_TRY(cx_gemm_bf16_nt(buf->g_wide, w.WqkvT, buf->g_b, nullptr, T, d, 3 * d, 3 * d, 3 * d, d, 0, 1, 1.f,
                               stream));
        da = buf->g_a;  // dz1: residual branch into h_in
        db = buf->g_b;  // attention branch into h_in
    }
    return cx_embed_ln_bwd(da, db, input_ids, indices, enc->word_emb, enc->type_emb, enc->pos_emb, enc->emb_ln_g,
                           buf->emb_mean, buf->emb_rstd, enc->gword_emb, enc->gtype_emb, enc->gpos_emb,
                           enc->gemb_ln_g, enc->gemb_ln_b, T, S, d, enc->padding_idx, stream);
}

int cx_abi_version(void) { return 1; }
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
