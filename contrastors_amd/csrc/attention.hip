// attention.hip -- K1/K2 (+K11 fused) of SURVEY.md §2b: non-causal variable-length self-attention forward and
// backward for head_dim 64 on CDNA4 matrix cores.  Replaces flash_attn_varlen_qkvpacked_func at
// sc/layers/attention.py:172-182 and the rotary pass sc/layers/embedding.py:685-706 (rotation is applied while
// the Q/K tiles are staged into LDS, so q/k never make an extra HBM round trip).
//
// Layout idea (all three kernels): the softmax row index is kept in the LANE dimension of every MFMA result.
//   forward / dQ:  S^T[key][q] = mfma(A=K rows, B=Q rows)   -> lane = q, 16 keys per lane in registers
//                  O^T[d][q]  += mfma(A=V^T rows, B=P)       -> lane = q again: rescale/LSE are lane-local
//   dK/dV:         S[q][key]   = mfma(A=Q rows, B=K rows)    -> lane = key, accumulate dK^T/dV^T[d][key]
// The MFMA accumulator register r of lane-half hi holds row (r&3)+8*(r>>2)+4*hi; feeding registers 8h..8h+7
// straight back as the next MFMA's B fragment only permutes the reduction index, provided the A fragment is
// read with the same permutation: rows {4hi..4hi+3, 8+4hi..8+4hi+3} (+16h) of a TRANSPOSED LDS tile
// (two ds_read_b64).  Transposed tiles are built while staging (pairs of rows packed into 32-bit LDS writes).
#include "cx_common.h"
#include "../../include/contrastors_hip.h"

namespace {

constexpr int DH = 64;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

// v_exp_f32 directly: the libm wrapper adds denormal-range fix-ups (4-5 extra VALU ops per element) that a softmax
// probability (<= 1, flushed to 0 below 2^-126 either way once rounded to bf16) does not need.
CX_DEVICE float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

CX_DEVICE void unpack8(const uint4& v, float (&f)[8]) {
    f[0] = bf16lo_to_f32(v.x); f[1] = bf16hi_to_f32(v.x);
    f[2] = bf16lo_to_f32(v.y); f[3] = bf16hi_to_f32(v.y);
    f[4] = bf16lo_to_f32(v.z); f[5] = bf16hi_to_f32(v.z);
    f[6] = bf16lo_to_f32(v.w); f[7] = bf16hi_to_f32(v.w);
}
CX_DEVICE uint4 pack8(const float (&f)[8]) {
    uint4 v;
    v.x = pack_bf16x2(f[0], f[1]); v.y = pack_bf16x2(f[2], f[3]);
    v.z = pack_bf16x2(f[4], f[5]); v.w = pack_bf16x2(f[6], f[7]);
    return v;
}
CX_DEVICE uint16_t elem16(const uint4& v, int e) {
    const uint32_t w = (e < 2) ? v.x : (e < 4) ? v.y : (e < 6) ? v.z : v.w;
    return (uint16_t)((e & 1) ? (w >> 16) : (w & 0xffffu));
}

// Load chunks cp and cp+4 (8 bf16 each) of one 64-wide head row and apply the non-interleaved rotation at
// position `pos` (fp32 math, one bf16 rounding -- same as the reference op).
CX_DEVICE void load_row_pair(const bf16_t* row, int cp, const float* cosv, const float* sinv, int pos, uint4& lo,
                             uint4& hi) {
    lo = *reinterpret_cast<const uint4*>(row + cp * 8);
    hi = *reinterpret_cast<const uint4*>(row + 32 + cp * 8);
    if (cosv) {
        float x1[8], x2[8], o1[8], o2[8];
        unpack8(lo, x1);
        unpack8(hi, x2);
        const float* c = cosv + (size_t)pos * 32 + cp * 8;
        const float* s = sinv + (size_t)pos * 32 + cp * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o1[e] = x1[e] * c[e] - x2[e] * s[e];
            o2[e] = x2[e] * c[e] + x1[e] * s[e];
        }
        lo = pack8(o1);
        hi = pack8(o2);
    }
}

// The rotation half of load_row_pair for data that was loaded earlier (register prefetch of the next chunk).
CX_DEVICE void rotate_loaded(uint4& lo, uint4& hi, int cp, const float* cosv, const float* sinv, int pos) {
    if (!cosv) return;
    float x1[8], x2[8], o1[8], o2[8];
    unpack8(lo, x1);
    unpack8(hi, x2);
    const float* c = cosv + (size_t)pos * 32 + cp * 8;
    const float* s = sinv + (size_t)pos * 32 + cp * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        o1[e] = x1[e] * c[e] - x2[e] * s[e];
        o2[e] = x2[e] * c[e] + x1[e] * s[e];
    }
    lo = pack8(o1);
    hi = pack8(o2);
}

// accumulator registers 8*half .. 8*half+7 -> bf16 B fragment
CX_DEVICE bf16x8_t pack_frag(const float (&v)[16], int half) {
    union {
        uint32_t u[4];
        bf16x8_t v;
    } x;
#pragma unroll
    for (int e = 0; e < 4; ++e) x.u[e] = pack_bf16x2(v[8 * half + 2 * e], v[8 * half + 2 * e + 1]);
    return x.v;
}

// A-fragment of the TRANSPOSE of a row-major [t][64 f] tile (tile64_off layout) through the transposing LDS read: lane gets
// T[i = f0 + (lane & 31)][k = t0 + 4*hi + {0..3}, t0 + 8 + 4*hi + {0..3}] -- what read_transposed_frag returns from a
// separately staged transposed copy, without staging one (round 2: the streaming kernels kept writing V^T / K^T / Q^T /
// dO^T tiles element by element; the tile they already stage row-major serves both orientations).
typedef __attribute__((address_space(3))) bf16x4_t* lds_b4_t64;
CX_DEVICE bf16x8_t tile64_tr_frag(const char* tile, int f0, int t0, int lane) {
    const int g = lane >> 4, pp = lane & 15;
    const int t = t0 + 4 * (g >> 1) + (pp >> 2);
    const int f = f0 + 16 * (g & 1) + 4 * (pp & 3);
    union { bf16x4_t h[2]; bf16x8_t v; } u;
    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4_t64)(tile + tile64_off(t, f >> 3) + (f & 4) * 2));
    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4_t64)(tile + tile64_off(t + 8, f >> 3) + (f & 4) * 2));
    return u.v;
}

struct AttnParams {
    const bf16_t* qkv;     // (T,3,H,64)
    const int32_t* cu;
    const float* cosv;     // rotary tables: inverse rotation of dq / dk at the stores (and the loads of the S <= 128 kernels)
    const float* sinv;
    const float* lcos;     // general kernels: rotation of q / k rows at the LOADS; NULL when qkv was rotated in place
    const float* lsin;     //   beforehand (long sequences: a K row would otherwise be re-rotated once per 128-query block)
    bf16_t* out;           // (T,H,64)            [fwd out / bwd: forward output]
    float* lse;            // (H,T)
    const bf16_t* dout;    // (T,H,64)
    float* delta;          // (H,T)
    bf16_t* dqkv;          // (T,3,H,64)
    int H, T;
    float scale;
    // kv-packed cross-attention (flash_attn_[varlen_]kvpacked_func, K3): the <true> instantiations of the general kernels
    // read queries from `qkv` as (Tq, H, 64), keys / values from `kv` (Tk, 2, H, 64) with their own cu_seqlens, and write
    // dq to `dqkv` as (Tq, H, 64), dk / dv to `dkv` (Tk, 2, H, 64).  T = Tq (lse / delta are (H, Tq)).  No rotary.
    const bf16_t* kv;
    const int32_t* cu_k;
    bf16_t* dkv;
    // attention dropout (attn_pdrop > 0, flash_attn_*_func(dropout_p > 0)): the <.., true> instantiations of the general
    // kernels.  keep(b, h, q, key) is a pure function of the chunk's Philox (seed, offset), the site and the indices --
    // forward, dQ and dK/dV kernels (and a GradCache re-forward under RandContext) regenerate it, nothing is stored.
    CxDropout drop;
    uint32_t drop_site;
    int prio;   // experiments (dev library, cx_attn_set_prio): 1 = MFMA loops of the S <= 128 backward run at s_setprio 1
};

// keep-scale (0 or 1 / (1 - p)) of P[q][key .. key + 3] of problem `unit` = b * H + h; key % 4 == 0
CX_DEVICE void attn_keep4(const AttnParams& p, uint32_t unit, int q, int key, float (&keep)[4]) {
    dropout_keep4(p.drop, p.drop_site, ((unsigned long long)unit << 40) | ((unsigned long long)(uint32_t)q << 20) | (uint32_t)(key >> 2),
                  keep);
}

// keep-scales of ONE key (`key`, this lane's) for the four consecutive queries q0 .. q0 + 3.  The lanes of a quad (lane & ~3)
// hold the keys 4m .. 4m + 3 of one Philox key group: lane j = lane & 3 draws the four words of query q0 + j (keys 4m .. 4m + 3)
// and the quad transposes the 4 x 4 block with DPP quad permutes: out[e] = word (key & 3) of the lane that drew query q0 + e.
template <int E>
CX_DEVICE float quad_bcast(float x) {   // lane (quad base + E)'s value in every lane of the quad (DPP quad_perm [E, E, E, E])
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), E * 0x55, 0xf, 0xf, true));
}
CX_DEVICE void quad_keep4(const AttnParams& p, uint32_t unit, int q0, int key, int lane, float (&out)[4]) {
    float w[4];
    attn_keep4(p, unit, q0 + (lane & 3), key & ~3, w);
    const int j = key & 3;   // (= lane & 3: the lane's key is wave * 32 + (lane & 31))
    // out[e] = word j of the lane that drew query q0 + e: the word index depends on the READER, so every word of lane e is
    // broadcast and the reader picks its own
#define CX_QK(e) (j == 0 ? quad_bcast<e>(w[0]) : j == 1 ? quad_bcast<e>(w[1]) : j == 2 ? quad_bcast<e>(w[2]) : quad_bcast<e>(w[3]))
    {
        const float a0 = quad_bcast<0>(w[0]), a1 = quad_bcast<0>(w[1]), a2 = quad_bcast<0>(w[2]), a3 = quad_bcast<0>(w[3]);
        out[0] = j == 0 ? a0 : j == 1 ? a1 : j == 2 ? a2 : a3;
    }
    {
        const float a0 = quad_bcast<1>(w[0]), a1 = quad_bcast<1>(w[1]), a2 = quad_bcast<1>(w[2]), a3 = quad_bcast<1>(w[3]);
        out[1] = j == 0 ? a0 : j == 1 ? a1 : j == 2 ? a2 : a3;
    }
    {
        const float a0 = quad_bcast<2>(w[0]), a1 = quad_bcast<2>(w[1]), a2 = quad_bcast<2>(w[2]), a3 = quad_bcast<2>(w[3]);
        out[2] = j == 0 ? a0 : j == 1 ? a1 : j == 2 ? a2 : a3;
    }
    {
        const float a0 = quad_bcast<3>(w[0]), a1 = quad_bcast<3>(w[1]), a2 = quad_bcast<3>(w[2]), a3 = quad_bcast<3>(w[3]);
        out[3] = j == 0 ? a0 : j == 1 ? a1 : j == 2 ? a2 : a3;
    }
#undef CX_QK
}

// Where a (sequence, head) problem's rows live.  X = false: packed qkv, one set of lengths (self-attention).
struct AttnView {
    const bf16_t *q, *k, *v;
    size_t qs, ks;            // row strides (elements) of the query rows and of the key / value rows
    int t0q, lenq, t0k, lenk;
};
template <bool X>
CX_DEVICE AttnView attn_view(const AttnParams& p, int h, int b) {
    AttnView w;
    w.t0q = p.cu[b];
    w.lenq = p.cu[b + 1] - w.t0q;
    if constexpr (X) {
        w.t0k = p.cu_k[b];
        w.lenk = p.cu_k[b + 1] - w.t0k;
        w.q = p.qkv + (size_t)h * DH;
        w.qs = (size_t)p.H * DH;
        w.k = p.kv + (size_t)h * DH;
        w.v = w.k + (size_t)p.H * DH;
        w.ks = (size_t)2 * p.H * DH;
    } else {
        w.t0k = w.t0q;
        w.lenk = w.lenq;
        w.q = p.qkv + (size_t)h * DH;
        w.k = w.q + (size_t)p.H * DH;
        w.v = w.k + (size_t)p.H * DH;
        w.qs = w.ks = (size_t)3 * p.H * DH;
    }
    return w;
}

// ---------------------------------------------------------------------------------------------------- forward
template <bool X, bool DROP = false>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(16))) char smem[16384 + 8192 + 8192];
    char* Qs = smem;
    char* Ks = smem + 16384;
    char* Vs = smem + 16384 + 8192;   // V chunk row-major [64 keys][64 d]; its transpose is read with tile64_tr_frag
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    const int h = blockIdx.y, b = blockIdx.z;
    const AttnView w = attn_view<X>(p, h, b);
    const int t0 = w.t0q, len = w.lenq;        // query side
    const int t0k = w.t0k, lenk = w.lenk;      // key / value side (the same numbers in self-attention)
    const int q0 = blockIdx.x * 128;
    if (q0 >= len) return;
    const size_t tok_stride = w.qs, kv_stride = w.ks;
    const bf16_t* qbase = w.q;
    const bf16_t* kbase = w.k;
    const bf16_t* vbase = w.v;

    // stage (rotated) Q tile
#pragma unroll
    for (int pss = 0; pss < 2; ++pss) {
        const int r = pss * 64 + (tid >> 2), cp = tid & 3;
        int tq = q0 + r;
        tq = tq < len ? tq : len - 1;
        uint4 lo, hi4;
        load_row_pair(qbase + (size_t)(t0 + tq) * tok_stride, cp, p.lcos, p.lsin, tq, lo, hi4);
        *reinterpret_cast<uint4*>(Qs + tile64_off(r, cp)) = lo;
        *reinterpret_cast<uint4*>(Qs + tile64_off(r, cp + 4)) = hi4;
    }
    __syncthreads();
    bf16x8_t qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = lds_read_frag(Qs, tile64_off(wave * 32 + l31, ks * 2 + hi));

    const float sc2 = p.scale * LOG2E;
    float m_run = -1e30f, l_run = 0.f;
    f32x16_t acc_o[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[db][r] = 0.f;

    // The next K / V chunk is loaded into registers while the current one is computed from LDS (the global latency of
    // every chunk used to sit between two barriers).  K item = (row, chunk pair), V item = (key pair, 8-column chunk).
    uint4 k_lo, k_hi, v_r0, v_r1;
    auto prefetch = [&](int kv0) {
        const int r = tid >> 2, cp = tid & 3;
        int tk = kv0 + r;
        tk = tk < lenk ? tk : lenk - 1;
        const bf16_t* krow = kbase + (size_t)(t0k + tk) * kv_stride;
        k_lo = *reinterpret_cast<const uint4*>(krow + cp * 8);
        k_hi = *reinterpret_cast<const uint4*>(krow + 32 + cp * 8);
        const int kp = tid >> 3, c = tid & 7;
        int k0i = kv0 + 2 * kp, k1i = k0i + 1;
        k0i = k0i < lenk ? k0i : lenk - 1;
        k1i = k1i < lenk ? k1i : lenk - 1;
        v_r0 = *reinterpret_cast<const uint4*>(vbase + (size_t)(t0k + k0i) * kv_stride + c * 8);
        v_r1 = *reinterpret_cast<const uint4*>(vbase + (size_t)(t0k + k1i) * kv_stride + c * 8);
    };
    if (lenk > 0) prefetch(0);
    for (int kv0 = 0; kv0 < lenk; kv0 += 64) {
        {
            const int r = tid >> 2, cp = tid & 3;
            int tk = kv0 + r;
            tk = tk < lenk ? tk : lenk - 1;
            rotate_loaded(k_lo, k_hi, cp, p.lcos, p.lsin, tk);
            *reinterpret_cast<uint4*>(Ks + tile64_off(r, cp)) = k_lo;
            *reinterpret_cast<uint4*>(Ks + tile64_off(r, cp + 4)) = k_hi;
            *reinterpret_cast<uint4*>(Vs + tile64_off(2 * (tid >> 3), tid & 7)) = v_r0;
            *reinterpret_cast<uint4*>(Vs + tile64_off(2 * (tid >> 3) + 1, tid & 7)) = v_r1;
        }
        __syncthreads();
        if (kv0 + 64 < lenk) prefetch(kv0 + 64);

        // raw scores; the softmax scale is folded into the exponent (p = exp2(fma(s, c, -m))), and the key mask exists only
        // on a partial last chunk (a wave-uniform branch): the loop is VALU-bound -- 16 exponentials per 32x32 block
        // against 8 MFMAs -- so the 32 multiplies and 64 compare / select pairs per chunk this removes are 1/3 of it
        float s[2][16];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16_t a;
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                a = mfma_bf16_32x32x16(lds_read_frag(Ks, tile64_off(kb * 32 + l31, ks * 2 + hi)), qf[ks], a);
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = a[r];
        }
        if (kv0 + 64 > lenk) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kv0 + kb * 32 + acc_row(r, hi) >= lenk) s[kb][r] = -INFINITY;
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * sc2;   // (sc2 > 0: the maximum commutes with the scale)
        const float m_new = fmaxf(m_run, mx);
        const float alpha = fast_exp2(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[kb][r] = fast_exp2(__builtin_fmaf(s[kb][r], sc2, -m_new));
                psum += s[kb][r];
            }
        l_run = l_run * alpha + psum;
        m_run = m_new;
        if constexpr (DROP) {  // O accumulates P * keep / (1 - p); the normaliser above is the undropped row sum
            const int qi = q0 + wave * 32 + l31;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    float k4[4];
                    attn_keep4(p, (uint32_t)(b * p.H + h), qi, kv0 + kb * 32 + 8 * qd + 4 * hi, k4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) s[kb][4 * qd + e] *= k4[e];
                }
        }
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_o[db][r] *= alpha;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const bf16x8_t pf = pack_frag(s[kb], half);
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    acc_o[db] =
                        mfma_bf16_32x32x16(tile64_tr_frag(Vs, db * 32, kb * 32 + half * 16, lane), pf, acc_o[db]);
            }
        __syncthreads();
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = X ? (l_tot > 0.f ? 1.f / l_tot : 0.f) : 1.f / l_tot;  // (a cross problem may have no keys: out = 0)
    const int q = q0 + wave * 32 + l31;
    // full-row stores through this wave's own Q rows (dead since the fragments were read; see store_unrotated_rows)
    char* stage = Qs + wave * 4096;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            uint2 pk;
            pk.x = pack_bf16x2(acc_o[db][4 * qd] * inv, acc_o[db][4 * qd + 1] * inv);
            pk.y = pack_bf16x2(acc_o[db][4 * qd + 2] * inv, acc_o[db][4 * qd + 3] * inv);
            *reinterpret_cast<uint2*>(stage + tile64_off(l31, db * 4 + qd) + hi * 8) = pk;
        }
    if (q < len && hi == 0) p.lse[(size_t)h * p.T + t0 + q] = (m_run + log2f(l_tot)) * LN2;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = it * 8 + (lane >> 3), c = lane & 7;
        const int qq = q0 + wave * 32 + r;
        const uint4 v = *reinterpret_cast<const uint4*>(stage + tile64_off(r, c));
        if (qq < len) *reinterpret_cast<uint4*>(p.out + ((size_t)(t0 + qq) * p.H + h) * DH + c * 8) = v;
    }
}


// ------------------------------------------------------------------------------------ forward, sequences <= 128
// BERT-style contrastive batches (seq_len 128, the BASELINE metric) fit a whole (sequence, head) problem in one
// workgroup: Q, K and V^T (128 keys) live in LDS at once.  All global loads are issued up front (one memory latency
// instead of three), there is one barrier, and softmax is single-pass (no online rescale).

CX_DEVICE void rot8(const uint4& lo_in, const uint4& hi_in, const float4 (&c)[2], const float4 (&s)[2], uint4& lo,
                    uint4& hi) {
    float x1[8], x2[8], o1[8], o2[8];
    unpack8(lo_in, x1);
    unpack8(hi_in, x2);
    const float cc[8] = {c[0].x, c[0].y, c[0].z, c[0].w, c[1].x, c[1].y, c[1].z, c[1].w};
    const float ss[8] = {s[0].x, s[0].y, s[0].z, s[0].w, s[1].x, s[1].y, s[1].z, s[1].w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        o1[e] = x1[e] * cc[e] - x2[e] * ss[e];
        o2[e] = x2[e] * cc[e] + x1[e] * ss[e];
    }
    lo = pack8(o1);
    hi = pack8(o2);
}




// ------------------------------------------------------------------------------------- delta = rowsum(dO * O)
__global__ __launch_bounds__(256) void attn_delta_kernel(AttnParams p) {
    const long total = (long)p.T * p.H * 8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long th = i >> 3;
        const int c = (int)(i & 7);
        float a[8], bq[8];
        unpack8(*reinterpret_cast<const uint4*>(p.dout + th * DH + c * 8), a);
        unpack8(*reinterpret_cast<const uint4*>(p.out + th * DH + c * 8), bq);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += a[e] * bq[e];
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        if (c == 0) {
            const long t = th / p.H;
            const int hh = (int)(th - t * p.H);
            p.delta[(size_t)hh * p.T + t] = s;
        }
    }
}

// Inverse rotation of a gradient held as acc[0] (d < 32) / acc[1] (d >= 32), then scaled bf16 store of one head row.
CX_DEVICE void store_unrotated(bf16_t* row, const f32x16_t (&acc)[2], float scale, const float* cosv,
                               const float* sinv, int pos, int hi) {
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
        const int d = 8 * qd + 4 * hi;
        float lo[4], hh[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            lo[e] = acc[0][4 * qd + e] * scale;
            hh[e] = acc[1][4 * qd + e] * scale;
        }
        if (cosv) {
            const float4 c = *reinterpret_cast<const float4*>(cosv + (size_t)pos * 32 + d);
            const float4 s = *reinterpret_cast<const float4*>(sinv + (size_t)pos * 32 + d);
            const float cc[4] = {c.x, c.y, c.z, c.w}, ss[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gl = lo[e], gh = hh[e];
                lo[e] = gl * cc[e] + gh * ss[e];
                hh[e] = gh * cc[e] - gl * ss[e];
            }
        }
        uint2 pk;
        pk.x = pack_bf16x2(lo[0], lo[1]);
        pk.y = pack_bf16x2(lo[2], lo[3]);
        *reinterpret_cast<uint2*>(row + d) = pk;
        pk.x = pack_bf16x2(hh[0], hh[1]);
        pk.y = pack_bf16x2(hh[2], hh[3]);
        *reinterpret_cast<uint2*>(row + 32 + d) = pk;
    }
}


// store_unrotated for a wave's 32 consecutive rows with full-row global stores: the accumulator layout gives a lane
// 4 consecutive d of ONE row, so direct stores are 16-byte pieces of 32 different rows per instruction (partial-line
// writes; they cost the S <= 128 forward 17 %).  Staged in `stage` (4 KiB private to the wave, tile64 swizzle) the rows
// leave as 16 B per lane, 8 lanes per 128-B row.  g0 = row 0 of the wave's rows, rows_valid = how many of them exist.
CX_DEVICE void store_unrotated_rows(char* stage, bf16_t* g0, size_t row_stride, int rows_valid, const f32x16_t (&acc)[2],
                                    float scale, const float* cosv, const float* sinv, int pos, int hi, int lane) {
    const int l31 = lane & 31;
    // the table rows are fetched ONE column group ahead of their use (8 more registers): the four global round trips of
    // the first version were serialised, each behind whatever stores the wave had just issued (one in-order vmcnt) --
    // phase timers put the dK / dV store phase of the fused S <= 128 backward at 10.9 k of its 44.5 k cycles per problem
    float4 cn = make_float4(1.f, 1.f, 1.f, 1.f), sn = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cosv) {
        cn = *reinterpret_cast<const float4*>(cosv + (size_t)pos * 32 + 4 * hi);
        sn = *reinterpret_cast<const float4*>(sinv + (size_t)pos * 32 + 4 * hi);
    }
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
        const int d = 8 * qd + 4 * hi;
        float lo[4], hh[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            lo[e] = acc[0][4 * qd + e] * scale;
            hh[e] = acc[1][4 * qd + e] * scale;
        }
        if (cosv) {
            const float4 c = cn, s = sn;
            if (qd < 3) {
                cn = *reinterpret_cast<const float4*>(cosv + (size_t)pos * 32 + d + 8);
                sn = *reinterpret_cast<const float4*>(sinv + (size_t)pos * 32 + d + 8);
            }
            const float cc[4] = {c.x, c.y, c.z, c.w}, ss[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gl = lo[e], gh = hh[e];
                lo[e] = gl * cc[e] + gh * ss[e];
                hh[e] = gh * cc[e] - gl * ss[e];
            }
        }
        uint2 pk;
        pk.x = pack_bf16x2(lo[0], lo[1]);
        pk.y = pack_bf16x2(lo[2], lo[3]);
        *reinterpret_cast<uint2*>(stage + tile64_off(l31, qd) + hi * 8) = pk;
        pk.x = pack_bf16x2(hh[0], hh[1]);
        pk.y = pack_bf16x2(hh[2], hh[3]);
        *reinterpret_cast<uint2*>(stage + tile64_off(l31, 4 + qd) + hi * 8) = pk;
        __builtin_amdgcn_sched_barrier(0);  // one qd at a time: hoisting all four cos/sin fetches costs 32 registers
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = it * 8 + (lane >> 3), c = lane & 7;
        const uint4 v = *reinterpret_cast<const uint4*>(stage + tile64_off(r, c));
        if (r < rows_valid) *reinterpret_cast<uint4*>(g0 + (size_t)r * row_stride + c * 8) = v;
    }
}

// The inverse rotation's table rows of ONE position, all 32 column pairs of this lane's accumulator layout (d = 8 qd + 4 hi + e):
// fetched once per problem by the fused S <= 128 backward (round 5) -- the lane's key row (dK) and its query row (dQ) are the same
// index there, so the 8 float4 serve both store phases and no global round trip sits inside either of them.
struct RotRow {
    float4 c[4], s[4];
};
CX_DEVICE void load_rot_row(const float* cosv, const float* sinv, int pos, int hi, RotRow& o) {
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
        o.c[qd] = *reinterpret_cast<const float4*>(cosv + (size_t)pos * 32 + 8 * qd + 4 * hi);
        o.s[qd] = *reinterpret_cast<const float4*>(sinv + (size_t)pos * 32 + 8 * qd + 4 * hi);
    }
}
// store_unrotated_rows with the table rows in registers (by reference + flag: a pointer would put the struct in scratch memory)
CX_DEVICE void store_unrotated_rows_pre(char* stage, bf16_t* g0, size_t row_stride, int rows_valid, const f32x16_t (&acc)[2],
                                        float scale, const RotRow& rot, bool use_rot, int hi, int lane) {
    const int l31 = lane & 31;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
        float lo[4], hh[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            lo[e] = acc[0][4 * qd + e] * scale;
            hh[e] = acc[1][4 * qd + e] * scale;
        }
        if (use_rot) {
            const float4 c = rot.c[qd], s = rot.s[qd];
            const float cc[4] = {c.x, c.y, c.z, c.w}, ss[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gl = lo[e], gh = hh[e];
                lo[e] = gl * cc[e] + gh * ss[e];
                hh[e] = gh * cc[e] - gl * ss[e];
            }
        }
        uint2 pk;
        pk.x = pack_bf16x2(lo[0], lo[1]);
        pk.y = pack_bf16x2(lo[2], lo[3]);
        *reinterpret_cast<uint2*>(stage + tile64_off(l31, qd) + hi * 8) = pk;
        pk.x = pack_bf16x2(hh[0], hh[1]);
        pk.y = pack_bf16x2(hh[2], hh[3]);
        *reinterpret_cast<uint2*>(stage + tile64_off(l31, 4 + qd) + hi * 8) = pk;
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = it * 8 + (lane >> 3), c = lane & 7;
        const uint4 v = *reinterpret_cast<const uint4*>(stage + tile64_off(r, c));
        if (r < rows_valid) *reinterpret_cast<uint4*>(g0 + (size_t)r * row_stride + c * 8) = v;
    }
}

// ---------------------------------------------------------------------------------------------------- dQ
// CX_ATTN_BWD_PF (round 5): the two streaming backward kernels loaded each 64-row chunk straight into LDS -- global round trip, barrier,
// compute, barrier, per chunk and workgroup, covered only by the CU's other workgroups (three of them at 168 registers).  With the
// switch the NEXT chunk's rows are requested right after the barrier that publishes the current chunk and wait in 16 registers through
// the compute (the forward kernel has done this since round 2); the register budget goes from 168 (3 workgroups per CU) to 256 (2).
template <bool X, bool DROP = false>
__global__ __launch_bounds__(256, 3) void attn_bwd_dq_kernel(AttnParams p) {
    // Qs/dOs are only needed to build the loop-invariant register fragments; the K / V tiles alias them.
    __shared__ __attribute__((aligned(16))) char smem[32768];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    const int h = blockIdx.y, b = blockIdx.z;
    const AttnView w = attn_view<X>(p, h, b);
    const int t0 = w.t0q, len = w.lenq;        // query side
    const int t0k = w.t0k, lenk = w.lenk;      // key / value side
    const int q0 = blockIdx.x * 128;
    if (q0 >= len) return;
    const size_t tok_stride = w.qs, kv_stride = w.ks;
    const size_t o_stride = (size_t)p.H * DH;
    const bf16_t* qbase = w.q;
    const bf16_t* kbase = w.k;
    const bf16_t* vbase = w.v;
    const bf16_t* dobase = p.dout + (size_t)h * DH;

    {
        char* Qs = smem;
        char* dOs = smem + 16384;
#pragma unroll
        for (int pss = 0; pss < 2; ++pss) {
            const int r = pss * 64 + (tid >> 2), cp = tid & 3;
            int tq = q0 + r;
            tq = tq < len ? tq : len - 1;
            uint4 lo, hi4;
            load_row_pair(qbase + (size_t)(t0 + tq) * tok_stride, cp, p.lcos, p.lsin, tq, lo, hi4);
            *reinterpret_cast<uint4*>(Qs + tile64_off(r, cp)) = lo;
            *reinterpret_cast<uint4*>(Qs + tile64_off(r, cp + 4)) = hi4;
            load_row_pair(dobase + (size_t)(t0 + tq) * o_stride, cp, nullptr, nullptr, 0, lo, hi4);
            *reinterpret_cast<uint4*>(dOs + tile64_off(r, cp)) = lo;
            *reinterpret_cast<uint4*>(dOs + tile64_off(r, cp + 4)) = hi4;
        }
    }
    __syncthreads();
    bf16x8_t qf[4], dof[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        qf[ks] = lds_read_frag(smem, tile64_off(wave * 32 + l31, ks * 2 + hi));
        dof[ks] = lds_read_frag(smem + 16384, tile64_off(wave * 32 + l31, ks * 2 + hi));
    }
    __syncthreads();

    char* Ks = smem;
    char* Vs = smem + 8192;
    const int q = q0 + wave * 32 + l31;
    const int qc = q < len ? q : len - 1;
    const float lse2 = p.lse[(size_t)h * p.T + t0 + qc] * LOG2E;
    const float dl = p.delta[(size_t)h * p.T + t0 + qc];
    const float sc2 = p.scale * LOG2E;

    f32x16_t acc_dq[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_dq[db][r] = 0.f;

    for (int kv0 = 0; kv0 < lenk; kv0 += 64) {
        if (wave < 2) {  // K: rotated, row-major (its transpose for dQ comes from tile64_tr_frag).  item = (key pair, chunk pair)
            const int kp = tid >> 2, cp = tid & 3;
            int k0i = kv0 + 2 * kp, k1i = k0i + 1;
            k0i = k0i < lenk ? k0i : lenk - 1;
            k1i = k1i < lenk ? k1i : lenk - 1;
            uint4 a_lo, a_hi, b_lo, b_hi;
            load_row_pair(kbase + (size_t)(t0k + k0i) * kv_stride, cp, p.lcos, p.lsin, k0i, a_lo, a_hi);
            load_row_pair(kbase + (size_t)(t0k + k1i) * kv_stride, cp, p.lcos, p.lsin, k1i, b_lo, b_hi);
            *reinterpret_cast<uint4*>(Ks + tile64_off(2 * kp, cp)) = a_lo;
            *reinterpret_cast<uint4*>(Ks + tile64_off(2 * kp, cp + 4)) = a_hi;
            *reinterpret_cast<uint4*>(Ks + tile64_off(2 * kp + 1, cp)) = b_lo;
            *reinterpret_cast<uint4*>(Ks + tile64_off(2 * kp + 1, cp + 4)) = b_hi;
        } else {  // V row-major: 64 rows x 8 chunks over 128 threads
            const int t2 = tid - 128;
#pragma unroll
            for (int pss = 0; pss < 4; ++pss) {
                const int r = pss * 16 + (t2 >> 3), c = t2 & 7;
                int tk = kv0 + r;
                tk = tk < lenk ? tk : lenk - 1;
                *reinterpret_cast<uint4*>(Vs + tile64_off(r, c)) =
                    *reinterpret_cast<const uint4*>(vbase + (size_t)(t0k + tk) * kv_stride + c * 8);
            }
        }
        __syncthreads();
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16_t a_s, a_dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) a_s[r] = a_dp[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                a_s = mfma_bf16_32x32x16(lds_read_frag(Ks, tile64_off(kb * 32 + l31, ks * 2 + hi)), qf[ks], a_s);
                a_dp = mfma_bf16_32x32x16(lds_read_frag(Vs, tile64_off(kb * 32 + l31, ks * 2 + hi)), dof[ks], a_dp);
            }
            float ds[16];
            if constexpr (DROP) {  // dP reaches P only through the kept entries: dS = P * (dP * keep / (1 - p) - delta)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    float k4[4];
                    attn_keep4(p, (uint32_t)(b * p.H + h), q, kv0 + kb * 32 + 8 * qd + 4 * hi, k4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) a_dp[4 * qd + e] *= k4[e];
                }
            }
            if (kv0 + 64 <= lenk) {  // full chunk (wave-uniform): no key mask -- 16 compare / select pairs per block less
#pragma unroll
                for (int r = 0; r < 16; ++r) ds[r] = fast_exp2(__builtin_fmaf(a_s[r], sc2, -lse2)) * (a_dp[r] - dl);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + kb * 32 + acc_row(r, hi);
                    const float pr = key < lenk ? fast_exp2(__builtin_fmaf(a_s[r], sc2, -lse2)) : 0.f;
                    ds[r] = pr * (a_dp[r] - dl);
                }
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const bf16x8_t dsf = pack_frag(ds, half);
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    acc_dq[db] = mfma_bf16_32x32x16(tile64_tr_frag(Ks, db * 32, kb * 32 + half * 16, lane), dsf, acc_dq[db]);
            }
        }
        __syncthreads();
    }
    // (the loop ends with a barrier: all of LDS is dead, every wave stages its 32 rows in its own 4 KiB)
    store_unrotated_rows(smem + wave * 4096, p.dqkv + (size_t)(t0 + q0 + wave * 32) * tok_stride + (size_t)h * DH,
                         tok_stride, len - (q0 + wave * 32), acc_dq, p.scale, p.cosv, p.sinv, q < len ? q : len - 1, hi,
                         lane);
}

// ---------------------------------------------------------------------------------------------------- dK, dV
template <bool X, bool DROP = false>
__global__ __launch_bounds__(256, DROP ? 2 : 3) void attn_bwd_dkv_kernel(AttnParams p) {
    // prologue: K,V tiles [128][64] (2 x 16 KiB); loop: Qs 8K | dOs 8K | lse[64] | delta[64]
    __shared__ __attribute__((aligned(16))) char smem[32768];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    const int h = blockIdx.y, b = blockIdx.z;
    const AttnView w = attn_view<X>(p, h, b);
    const int t0 = w.t0q, len = w.lenq;        // query side
    const int t0k = w.t0k, lenk = w.lenk;      // key / value side: this workgroup owns keys k0 .. k0 + 127
    const int k0 = blockIdx.x * 128;
    if (k0 >= lenk) return;
    const size_t tok_stride = w.qs, kv_stride = w.ks;
    const size_t o_stride = (size_t)p.H * DH;
    const bf16_t* qbase = w.q;
    const bf16_t* kbase = w.k;
    const bf16_t* vbase = w.v;
    const bf16_t* dobase = p.dout + (size_t)h * DH;

    {
        char* Ks = smem;
        char* Vs = smem + 16384;
#pragma unroll
        for (int pss = 0; pss < 2; ++pss) {
            const int r = pss * 64 + (tid >> 2), cp = tid & 3;
            int tk = k0 + r;
            tk = tk < lenk ? tk : lenk - 1;
            uint4 lo, hi4;
            load_row_pair(kbase + (size_t)(t0k + tk) * kv_stride, cp, p.lcos, p.lsin, tk, lo, hi4);
            *reinterpret_cast<uint4*>(Ks + tile64_off(r, cp)) = lo;
            *reinterpret_cast<uint4*>(Ks + tile64_off(r, cp + 4)) = hi4;
            load_row_pair(vbase + (size_t)(t0k + tk) * kv_stride, cp, nullptr, nullptr, 0, lo, hi4);
            *reinterpret_cast<uint4*>(Vs + tile64_off(r, cp)) = lo;
            *reinterpret_cast<uint4*>(Vs + tile64_off(r, cp + 4)) = hi4;
        }
    }
    __syncthreads();
    bf16x8_t kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        kf[ks] = lds_read_frag(smem, tile64_off(wave * 32 + l31, ks * 2 + hi));
        vf[ks] = lds_read_frag(smem + 16384, tile64_off(wave * 32 + l31, ks * 2 + hi));
    }
    __syncthreads();

    char* Qs = smem;
    char* dOs = smem + 8192;
    float* lse_s = reinterpret_cast<float*>(smem + 16384);   // (the transposes of Q and dO come from tile64_tr_frag)
    float* dl_s = lse_s + 64;

    const int key = k0 + wave * 32 + l31;
    const bool key_ok = key < lenk;
    const bool keys_full = k0 + 128 <= lenk;
    const float sc2 = p.scale * LOG2E;
    f32x16_t acc_dk[2], acc_dv[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_dk[db][r] = acc_dv[db][r] = 0.f;

    // DKV_PF: see CX_ATTN_BWD_PF at the top of the file -- the dropout instantiation only
    constexpr bool DKV_PF = DROP;
    // raw rows of a chunk: waves 0-1 a Q row pair, waves 2-3 a dO row pair (chunks cp, cp + 4 of both rows); threads 0-63 lse, delta
    uint4 pf0 = {}, pf1 = {}, pf2 = {}, pf3 = {};
    float pf_lse = 0.f, pf_dl = 0.f;
    auto issue_chunk = [&](int q0) {
        const int t2 = wave < 2 ? tid : tid - 128;
        const int rp = t2 >> 2, cp = t2 & 3;
        int r0i = q0 + 2 * rp, r1i = r0i + 1;
        r0i = r0i < len ? r0i : len - 1;
        r1i = r1i < len ? r1i : len - 1;
        const bf16_t* ra = wave < 2 ? qbase + (size_t)(t0 + r0i) * tok_stride : dobase + (size_t)(t0 + r0i) * o_stride;
        const bf16_t* rb = wave < 2 ? qbase + (size_t)(t0 + r1i) * tok_stride : dobase + (size_t)(t0 + r1i) * o_stride;
        pf0 = *reinterpret_cast<const uint4*>(ra + cp * 8);
        pf1 = *reinterpret_cast<const uint4*>(ra + 32 + cp * 8);
        pf2 = *reinterpret_cast<const uint4*>(rb + cp * 8);
        pf3 = *reinterpret_cast<const uint4*>(rb + 32 + cp * 8);
        if (tid < 64) {
            int r = q0 + tid;
            const bool ok = r < len;
            r = ok ? r : len - 1;
            pf_lse = ok ? p.lse[(size_t)h * p.T + t0 + r] * LOG2E : INFINITY;   // rows past the end: P = exp2(-inf) = 0
            pf_dl = p.delta[(size_t)h * p.T + t0 + r];
        }
    };
    if constexpr (DKV_PF) {
        if (len > 0) issue_chunk(0);
    }
    for (int q0 = 0; q0 < len; q0 += 64) {
        if constexpr (DKV_PF) {
            const int t2 = wave < 2 ? tid : tid - 128;
            const int rp = t2 >> 2, cp = t2 & 3;
            char* dst = wave < 2 ? Qs : dOs;
            if (wave < 2) {
                int r0i = q0 + 2 * rp, r1i = r0i + 1;
                r0i = r0i < len ? r0i : len - 1;
                r1i = r1i < len ? r1i : len - 1;
                rotate_loaded(pf0, pf1, cp, p.lcos, p.lsin, r0i);
                rotate_loaded(pf2, pf3, cp, p.lcos, p.lsin, r1i);
            }
            *reinterpret_cast<uint4*>(dst + tile64_off(2 * rp, cp)) = pf0;
            *reinterpret_cast<uint4*>(dst + tile64_off(2 * rp, cp + 4)) = pf1;
            *reinterpret_cast<uint4*>(dst + tile64_off(2 * rp + 1, cp)) = pf2;
            *reinterpret_cast<uint4*>(dst + tile64_off(2 * rp + 1, cp + 4)) = pf3;
            if (tid < 64) {
                lse_s[tid] = pf_lse;
                dl_s[tid] = pf_dl;
            }
            __syncthreads();
            if (q0 + 64 < len) issue_chunk(q0 + 64);   // in flight through this chunk's compute
        } else {
        if (wave < 2) {  // Q rotated, row-major; item = (row pair, chunk pair)
            const int rp = tid >> 2, cp = tid & 3;
            int r0i = q0 + 2 * rp, r1i = r0i + 1;
            r0i = r0i < len ? r0i : len - 1;
            r1i = r1i < len ? r1i : len - 1;
            uint4 a_lo, a_hi, b_lo, b_hi;
            load_row_pair(qbase + (size_t)(t0 + r0i) * tok_stride, cp, p.lcos, p.lsin, r0i, a_lo, a_hi);
            load_row_pair(qbase + (size_t)(t0 + r1i) * tok_stride, cp, p.lcos, p.lsin, r1i, b_lo, b_hi);
            *reinterpret_cast<uint4*>(Qs + tile64_off(2 * rp, cp)) = a_lo;
            *reinterpret_cast<uint4*>(Qs + tile64_off(2 * rp, cp + 4)) = a_hi;
            *reinterpret_cast<uint4*>(Qs + tile64_off(2 * rp + 1, cp)) = b_lo;
            *reinterpret_cast<uint4*>(Qs + tile64_off(2 * rp + 1, cp + 4)) = b_hi;
        } else {  // dO: row-major
            const int t2 = tid - 128;
            const int rp = t2 >> 2, cp = t2 & 3;
            int r0i = q0 + 2 * rp, r1i = r0i + 1;
            r0i = r0i < len ? r0i : len - 1;
            r1i = r1i < len ? r1i : len - 1;
            uint4 a_lo, a_hi, b_lo, b_hi;
            load_row_pair(dobase + (size_t)(t0 + r0i) * o_stride, cp, nullptr, nullptr, 0, a_lo, a_hi);
            load_row_pair(dobase + (size_t)(t0 + r1i) * o_stride, cp, nullptr, nullptr, 0, b_lo, b_hi);
            *reinterpret_cast<uint4*>(dOs + tile64_off(2 * rp, cp)) = a_lo;
            *reinterpret_cast<uint4*>(dOs + tile64_off(2 * rp, cp + 4)) = a_hi;
            *reinterpret_cast<uint4*>(dOs + tile64_off(2 * rp + 1, cp)) = b_lo;
            *reinterpret_cast<uint4*>(dOs + tile64_off(2 * rp + 1, cp + 4)) = b_hi;
        }
        if (tid < 64) {
            int r = q0 + tid;
            const bool ok = r < len;
            r = ok ? r : len - 1;
            // rows past the end of the sequence get lse = +inf -> P = exp2(-inf) = 0: they contribute nothing
            lse_s[tid] = ok ? p.lse[(size_t)h * p.T + t0 + r] * LOG2E : INFINITY;
            dl_s[tid] = p.delta[(size_t)h * p.T + t0 + r];
        }
        __syncthreads();
        }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            f32x16_t a_s, a_dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) a_s[r] = a_dp[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                a_s = mfma_bf16_32x32x16(lds_read_frag(Qs, tile64_off(qb * 32 + l31, ks * 2 + hi)), kf[ks], a_s);
                a_dp = mfma_bf16_32x32x16(lds_read_frag(dOs, tile64_off(qb * 32 + l31, ks * 2 + hi)), vf[ks], a_dp);
            }
            float pr[16], ds[16];
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int row = qb * 32 + 8 * qd + 4 * hi;
                const float4 l4 = *reinterpret_cast<const float4*>(lse_s + row);
                const float4 d4 = *reinterpret_cast<const float4*>(dl_s + row);
                const float ll[4] = {l4.x, l4.y, l4.z, l4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
                // DROP: this lane holds ONE key and four consecutive queries.  Rounds 2-4 drew one Philox4x32-10 word quadruple per
                // ELEMENT here (the four lanes of a key group each computed the same quadruple and kept one word) --
                // the streaming backward under dropout ran 2.4-3.2 x slower than without.  As in the fused S <= 128 backward: lane j
                // of a quad draws the quadruple of query row + j and the 4 x 4 block is transposed across the quad with DPP
                // (quad_keep4): one call per four mask values, the same masks bit for bit.
                float kq[4] = {1.f, 1.f, 1.f, 1.f};
                if constexpr (DROP) quad_keep4(p, (uint32_t)(b * p.H + h), q0 + row, key, lane, kq);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * qd + e;
                    // (keys_full: every key of this workgroup's 128 exists -- wave-uniform, no per-element select)
                    const float pe = fast_exp2(__builtin_fmaf(a_s[r], sc2, -ll[e]));
                    const float pv = (keys_full || key_ok) ? pe : 0.f;
                    if constexpr (DROP) {
                        const float kp = kq[e];
                        pr[r] = pv * kp;                          // dV = (P * keep / (1 - p))^T dO
                        ds[r] = pv * (a_dp[r] * kp - dd[e]);
                    } else {
                        pr[r] = pv;
                        ds[r] = pv * (a_dp[r] - dd[e]);
                    }
                }
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const bf16x8_t pf = pack_frag(pr, half), dsf = pack_frag(ds, half);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    acc_dv[db] = mfma_bf16_32x32x16(tile64_tr_frag(dOs, db * 32, qb * 32 + half * 16, lane), pf, acc_dv[db]);
                    acc_dk[db] = mfma_bf16_32x32x16(tile64_tr_frag(Qs, db * 32, qb * 32 + half * 16, lane), dsf, acc_dk[db]);
                }
            }
        }
        __syncthreads();
    }
    {
        bf16_t* kr0 = (X ? p.dkv + (size_t)h * DH : p.dqkv + (size_t)(p.H + h) * DH) + (size_t)(t0k + k0 + wave * 32) * kv_stride;
        const int valid = lenk - (k0 + wave * 32);
        store_unrotated_rows(smem + wave * 4096, kr0, kv_stride, valid, acc_dk, p.scale, p.cosv, p.sinv,
                             key_ok ? key : lenk - 1, hi, lane);
        store_unrotated_rows(smem + 16384 + wave * 4096, kr0 + (size_t)p.H * DH, kv_stride, valid, acc_dv, 1.f, nullptr,
                             nullptr, 0, hi, lane);
    }
}


// ------------------------------------------------------------------------------- backward, sequences <= 128
// Same idea as attn_fwd_s128_kernel: one workgroup owns a whole (sequence, head) problem, EVERY global load is issued
// before anything is staged (one HBM latency per workgroup instead of three serial load -> LDS -> compute phases), and
// the whole key / query range is processed from LDS in one pass.  Thread (kp = tid >> 2, cp = tid & 3) owns the row pair
// (2kp, 2kp+1) and the 16-B chunks cp and cp+4 of Q, K, V and dO: the same cos/sin values rotate its Q and K rows.
struct RowPairLoads {
    uint4 lo[2], hi[2];  // [row of the pair]
};
CX_DEVICE void load_pair_raw(const bf16_t* base, size_t stride, int t0, int ra, int rb, int cp, RowPairLoads& o) {
    const bf16_t* a = base + (size_t)(t0 + ra) * stride;
    const bf16_t* b = base + (size_t)(t0 + rb) * stride;
    o.lo[0] = *reinterpret_cast<const uint4*>(a + cp * 8);
    o.hi[0] = *reinterpret_cast<const uint4*>(a + 32 + cp * 8);
    o.lo[1] = *reinterpret_cast<const uint4*>(b + cp * 8);
    o.hi[1] = *reinterpret_cast<const uint4*>(b + 32 + cp * 8);
}
struct CosSin {
    float4 c[2][2], s[2][2];  // [row of the pair][half of the 8 columns]
};
CX_DEVICE void load_cossin(const float* cosv, const float* sinv, int ra, int rb, int cp, CosSin& o) {
    const int rows[2] = {ra, rb};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float* c = cosv + (size_t)rows[i] * 32 + cp * 8;
        const float* sn = sinv + (size_t)rows[i] * 32 + cp * 8;
        o.c[i][0] = *reinterpret_cast<const float4*>(c);
        o.c[i][1] = *reinterpret_cast<const float4*>(c + 4);
        o.s[i][0] = *reinterpret_cast<const float4*>(sn);
        o.s[i][1] = *reinterpret_cast<const float4*>(sn + 4);
    }
}
CX_DEVICE void rotate_pair(RowPairLoads& x, const CosSin& cs) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        uint4 lo, hi;
        rot8(x.lo[i], x.hi[i], cs.c[i], cs.s[i], lo, hi);
        x.lo[i] = lo;
        x.hi[i] = hi;
    }
}
CX_DEVICE void stage_rows(char* tile, int kp, int cp, const RowPairLoads& x) {  // row-major [128][64], tile64 swizzle
    *reinterpret_cast<uint4*>(tile + tile64_off(2 * kp, cp)) = x.lo[0];
    *reinterpret_cast<uint4*>(tile + tile64_off(2 * kp, cp + 4)) = x.hi[0];
    *reinterpret_cast<uint4*>(tile + tile64_off(2 * kp + 1, cp)) = x.lo[1];
    *reinterpret_cast<uint4*>(tile + tile64_off(2 * kp + 1, cp + 4)) = x.hi[1];
}



// ----------------------------------------------------------------- backward, sequences <= 128, ONE fused kernel
// dQ, dK, dV (and delta) of a whole (sequence, head) problem in one workgroup, reading Q, K, V, dO, O from HBM once
// (the two-kernel form reads qkv and dO twice and runs a separate delta pass: 2.2 GB vs 1.4 GB per 131072-token call).
// S and dP are computed once, in the key-owner orientation (lane = key); dV and dK follow from registers as in
// attn_bwd_dkv; dS goes through LDS as a [key][query] tile and comes back through the transposing LDS read
// (ds_read_b64_tr_b16) as the B operand of dQ^T = K^T dS -- the re-orientation that otherwise costs a second S / dP.
// 116 KiB of LDS -> one workgroup (4 waves, one per SIMD, 512 registers each) per CU: the workgroup is persistent and
// L2-prefetches the NEXT problem while the current one is computed.
typedef __attribute__((address_space(3))) bf16x4_t* lds_b4_attn;
// 16-B chunk swizzle of the dS tile: 4 row bits permute the 16 chunks of a row, so that the writers (32 keys x 8 B per
// instruction) and the transposing readers (4 rows x 32 B per 16-lane group) are both spread over the banks
CX_DEVICE int ds_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }


// ------------------------------------------------ backward, sequences <= 128, fused, TWO workgroups per CU (80 KiB)
// Same algorithm as attn_bwd_fused_s128_kernel with an LDS diet so that two workgroups share a CU (two waves per SIMD:
// one workgroup's VALU / LDS / wait phases overlap the other's MFMA phases).  Every tile is [row][128 idx] bf16 with a
// 256-B row stride and ONE swizzle (16-B chunk ^ ds_swz(row)) that serves all access patterns:
//   Qt, dOt [64 d][128 q]: S / dP A-operands through the transposing read, dK / dV A-operands as 2 x 8 B per row
//   Kt      [64 d][128 k]: dQ A-operand, one 16-B read per row       dS [128 k][128 q]: dQ B-operand, transposing read
// LDS map: R0 Qt | R1 dOt | R2 lse, delta (phase D) then Kt (dQ) | R3 K, V row-major (fragments) then dS.
constexpr int SWROW = 256;
CX_DEVICE int sw_off(int row, int idx) { return row * SWROW + ((((idx >> 3) ^ ds_swz(row)) << 4)) + (idx & 7) * 2; }
CX_DEVICE void stage_transposed_sw(char* tile, int kp, int cp, const RowPairLoads& x) {  // rows (2kp, 2kp+1) -> T[d][idx]
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const uint32_t wl = (uint32_t)elem16(x.lo[0], e) | ((uint32_t)elem16(x.lo[1], e) << 16);
        const uint32_t wh = (uint32_t)elem16(x.hi[0], e) | ((uint32_t)elem16(x.hi[1], e) << 16);
        *reinterpret_cast<uint32_t*>(tile + sw_off(cp * 8 + e, 2 * kp)) = wl;
        *reinterpret_cast<uint32_t*>(tile + sw_off(32 + cp * 8 + e, 2 * kp)) = wh;
    }
}
// fragment [k = t0 + 8*(lane>>5) + e][j = f0 + (lane&31)] of a swizzled [t][f] tile through the transposing read
CX_DEVICE bf16x8_t sw_tr_frag(const char* tile, int f0, int t0, int lane) {
    const int g = lane >> 4, pp = lane & 15;
    const int t = t0 + 8 * (g >> 1) + (pp >> 2);
    const int f = f0 + 16 * (g & 1) + 4 * (pp & 3);
    union { bf16x4_t h[2]; bf16x8_t v; } u;
    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4_attn)(tile + sw_off(t, f)));
    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4_attn)(tile + sw_off(t + 4, f)));
    return u.v;
}
// A[i = row][k in accumulator-register order: i0 + {0..3}, i0 + 8 + {0..3}], i0 = blk16*16 + 4*hi  (pairs with pack_frag)
CX_DEVICE bf16x8_t sw_perm_frag(const char* tile, int row, int blk16, int hi) {
    const int i0 = blk16 * 16 + 4 * hi;
    union { uint2 u[2]; bf16x8_t v; } x;
    x.u[0] = *reinterpret_cast<const uint2*>(tile + sw_off(row, i0));
    x.u[1] = *reinterpret_cast<const uint2*>(tile + sw_off(row, i0 + 8));
    return x.v;
}
// A[i = row][k = k0 + 8*hi + e]: one 16-B chunk
CX_DEVICE bf16x8_t sw_linear_frag(const char* tile, int row, int k0, int hi) {
    return *reinterpret_cast<const bf16x8_t*>(tile + sw_off(row, k0 + 8 * hi));
}

constexpr int FUSED2_LDS = 16384 * 3 + 32768;  // 80 KiB
// CX_ATTN_TRACE (variant builds of the dev library only, scripts/attn_trace.py): wave 0 of every workgroup stamps s_memtime
// at the phase boundaries of its first 16 problems into p.delta (unused by this kernel otherwise); vmcnt / lgkmcnt are
// drained at the stamps so that a phase owns its own latency.
#ifdef CX_ATTN_TRACE
#define CX_STAMP(i)                                                                                          \
    do {                                                                                                     \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                          \
        if (tid == 0 && it < 16) tr[((size_t)blockIdx.x * 16 + it) * 16 + (i)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define CX_STAMP(i) do {} while (0)
#endif

// DROP: dP reaches P only through the kept entries -- dV = (P * keep / (1 - p))^T dO, dS = P * (dP * keep / (1 - p) - delta);
// a lane holds ONE key and four consecutive queries per accumulator quad.  The four lanes of a key group (keys 4m .. 4m + 3)
// would each draw the same four Philox words per query; instead lane j draws the word quadruple of query q0 + j and the 4 x 4
// block is transposed across the quad with DPP (one Philox call per four mask values, as in the forward).
// (Rounds 4-5 also built two pipelined forms -- the next problem's rows requested ahead of the dQ store into registers, or by LDS-DMA into
// the dead tiles: bit-identical, 1.0-1.39 x slower, profiles/r4_attn_s128_experiments.txt, r5_attn_bwd_s128_ab.txt; removed in round 6.)
template <bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_fused2_s128_kernel(AttnParams p, int B) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Qt = smem;
    char* dOt = smem + 16384;
    char* R2 = smem + 32768;                 // lse | delta during phase D, Kt afterwards
    char* R3 = smem + 49152;                 // K | V row-major (tile64) for the fragments, dS afterwards
    float* lse_s = reinterpret_cast<float*>(R2);
    float* dl_s = lse_s + 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    const int kp = tid >> 2, cp = tid & 3;
    const size_t tok_stride = (size_t)3 * p.H * DH, o_stride = (size_t)p.H * DH;
    const int n_units = B * p.H;
#ifdef CX_ATTN_TRACE
    long long* tr = reinterpret_cast<long long*>(p.delta);
    int it = -1;
#endif
    for (int u = blockIdx.x; u < n_units; u += gridDim.x) {
        const int b = u / p.H, h = u - b * p.H;
        const int t0 = p.cu[b], len = p.cu[b + 1] - t0;
        if (len <= 0) continue;  // (uniform per workgroup)
#ifdef CX_ATTN_TRACE
        ++it;
#endif
        CX_STAMP(0);
        RowPairLoads ql, kl, vl, dOl, ol;
        CosSin csl;
        float lsel = 0.f;
        {   // (the serial form, exactly as it shipped in round 3: its register allocation is a fragile optimum)
            int ra = 2 * kp, rb = ra + 1;
            ra = ra < len ? ra : len - 1;
            rb = rb < len ? rb : len - 1;
            const bf16_t* qbase = p.qkv + (size_t)h * DH;
            // the rotation table rows FIRST: the K rows are the first thing staged and they are rotated on the way -- with the table
            // requested last (rounds 3-4) the first use waited for every row of the problem (loads return in order)
            if (p.cosv) load_cossin(p.cosv, p.sinv, ra, rb, cp, csl);
            load_pair_raw(qbase + (size_t)p.H * DH, tok_stride, t0, ra, rb, cp, kl);
            load_pair_raw(qbase + 2 * (size_t)p.H * DH, tok_stride, t0, ra, rb, cp, vl);
            load_pair_raw(qbase, tok_stride, t0, ra, rb, cp, ql);
            load_pair_raw(p.dout + (size_t)h * DH, o_stride, t0, ra, rb, cp, dOl);
            load_pair_raw(p.out + (size_t)h * DH, o_stride, t0, ra, rb, cp, ol);
            if (tid < 128) {
                const bool ok = tid < len;
                // rows past the end of the sequence get lse = +inf -> P = exp2(-inf) = 0: they contribute nothing
                lsel = ok ? p.lse[(size_t)h * p.T + t0 + tid] * LOG2E : INFINITY;
            }
        }
        RowPairLoads &q = ql, &k = kl, &v = vl, &dO = dOl, &o = ol;
        CosSin& cs = csl;
        const float lse_v = lsel;
        CX_STAMP(1);  // loads landed
        // ---- K, V row-major (for this wave's key fragments), then everything that depends on dO / O / Q ----
        if (p.cosv) rotate_pair(k, cs);
        stage_rows(R3, kp, cp, k);
        stage_rows(R3 + 16384, kp, cp, v);
        float dpart[2] = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 2; ++i) {  // delta = rowsum(dO * O): 16 of the 64 columns of rows (2kp, 2kp+1) per thread
            float a[8], c[8], acc = 0.f;
            unpack8(dO.lo[i], a); unpack8(o.lo[i], c);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc += a[e] * c[e];
            unpack8(dO.hi[i], a); unpack8(o.hi[i], c);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc += a[e] * c[e];
            acc += __shfl_xor(acc, 1, 64);
            acc += __shfl_xor(acc, 2, 64);
            dpart[i] = acc;
        }
        if (p.cosv) rotate_pair(q, cs);
        // (round 5 measured the row-major alternative -- Q / dO staged with 16-B writes, S / dP operands as plain row reads, dK / dV
        // operands through the transposing read: bit-identical, -1.9 % alone, but +3.7 % on top of the table prefetch below;
        // profiles/r5_attn_bwd_s128_ab.txt -- not kept)
        stage_transposed_sw(Qt, kp, cp, q);
        stage_transposed_sw(dOt, kp, cp, dO);
        if (cp == 0) {
            dl_s[2 * kp] = dpart[0];
            dl_s[2 * kp + 1] = dpart[1];
        }
        if (tid < 128) lse_s[tid] = lse_v;
        CX_STAMP(2);  // staged
        __syncthreads();
        CX_STAMP(3);  // barrier
        const int row = wave * 32 + l31;  // this lane's key (dK, dV) and later its query (dQ)
        const bool row_ok = row < len;
        bf16x8_t kf[4], vf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            kf[ks] = lds_read_frag(R3, tile64_off(row, ks * 2 + hi));
            vf[ks] = lds_read_frag(R3 + 16384, tile64_off(row, ks * 2 + hi));
        }
        __syncthreads();  // R3 becomes the dS tile
        CX_STAMP(4);  // fragments + barrier

        const float sc2 = p.scale * LOG2E;
        f32x16_t acc_dk[2], acc_dv[2];
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_dk[db][r] = acc_dv[db][r] = 0.f;
        if (p.prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll 1  // (rolled: unrolling makes the compiler hoist ~100 loop-invariant LDS addresses and spill)
        for (int qb = 0; qb < 4; ++qb) {
            // DROP: the 16 keep decisions of this (key, query block) are drawn BEFORE the S / dP products and carried as one 16-bit word:
            // the generator's temporaries are live while the two accumulator blocks (32 registers) are not
            uint32_t kbits = 0;
            if constexpr (DROP) {
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    float kq0[4] = {0.f, 1.f, 1.f, 1.f};
                    quad_keep4(p, (uint32_t)u, qb * 32 + 8 * qd + 4 * hi, row, lane, kq0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) kbits |= (kq0[e] != 0.f ? 1u : 0u) << (4 * qd + e);
                }
            }
            const float keep_inv = DROP ? 1.f / (1.f - p.drop.p) : 1.f;
            f32x16_t a_s, a_dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) a_s[r] = a_dp[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                a_s = mfma_bf16_32x32x16(sw_tr_frag(Qt, qb * 32, ks * 16, lane), kf[ks], a_s);
                a_dp = mfma_bf16_32x32x16(sw_tr_frag(dOt, qb * 32, ks * 16, lane), vf[ks], a_dp);
            }
            float pr[16], ds[16];
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int qrow = qb * 32 + 8 * qd + 4 * hi;
                const float4 l4 = *reinterpret_cast<const float4*>(lse_s + qrow);
                const float4 d4 = *reinterpret_cast<const float4*>(dl_s + qrow);
                const float ll[4] = {l4.x, l4.y, l4.z, l4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
                float kq[4] = {1.f, 1.f, 1.f, 1.f};
                if constexpr (DROP) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) kq[e] = ((kbits >> (4 * qd + e)) & 1u) ? keep_inv : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * qd + e;
                    const float pv = row_ok ? fast_exp2(a_s[r] * sc2 - ll[e]) : 0.f;
                    if constexpr (DROP) {
                        pr[r] = pv * kq[e];
                        ds[r] = pv * (a_dp[r] * kq[e] - dd[e]);
                    } else {
                        pr[r] = pv;
                        ds[r] = pv * (a_dp[r] - dd[e]);
                    }
                }
                uint2 pk;  // dS[key = row][queries qrow .. qrow+3]
                pk.x = pack_bf16x2(ds[4 * qd], ds[4 * qd + 1]);
                pk.y = pack_bf16x2(ds[4 * qd + 2], ds[4 * qd + 3]);
                *reinterpret_cast<uint2*>(R3 + sw_off(row, qrow)) = pk;
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const bf16x8_t pf = pack_frag(pr, half), dsf = pack_frag(ds, half);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    acc_dv[db] = mfma_bf16_32x32x16(sw_perm_frag(dOt, db * 32 + l31, qb * 2 + half, hi), pf, acc_dv[db]);
                    acc_dk[db] = mfma_bf16_32x32x16(sw_perm_frag(Qt, db * 32 + l31, qb * 2 + half, hi), dsf, acc_dk[db]);
                }
            }
        }
        if (p.prio) __builtin_amdgcn_s_setprio(0);
        CX_STAMP(5);  // main loop
        // the inverse rotation's table rows of this lane's position (its key for dK, its query for dQ: the same index), requested
        // BEFORE the barrier into registers the main loop has just freed: their round trip hides behind the barrier wait instead
        // of sitting inside the dK and dQ store phases (rounds 3-4 fetched them there, one column group ahead)
        RotRow rot = {};
        if (p.cosv) load_rot_row(p.cosv, p.sinv, row_ok ? row : len - 1, hi, rot);
        __syncthreads();  // the dS tile is complete; lse / delta, Q^T and dO^T are dead: R2 becomes Kt
        CX_STAMP(6);  // barrier
        {   // dK, dV of this wave's 32 keys leave as full rows through the wave's slices of the dead Q^T / dO^T tiles
            bf16_t* k0 = p.dqkv + (size_t)(t0 + wave * 32) * tok_stride + (size_t)(p.H + h) * DH;
            // dV first: its stores are in flight while the table rows land
            store_unrotated_rows_pre(dOt + wave * 4096, k0 + (size_t)p.H * DH, tok_stride, len - wave * 32, acc_dv, 1.f, rot, false, hi, lane);
            store_unrotated_rows_pre(Qt + wave * 4096, k0, tok_stride, len - wave * 32, acc_dk, p.scale, rot, p.cosv != nullptr, hi, lane);
        }
        CX_STAMP(7);  // dK, dV stored
        stage_transposed_sw(R2, kp, cp, k);
        CX_STAMP(8);  // Kt staged
        __syncthreads();
        CX_STAMP(9);  // barrier

        // ---- dQ^T[d][q] = sum_k K^T[d][k] dS[k][q] for this wave's 32 queries ----
        f32x16_t acc_dq[2];
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_dq[db][r] = 0.f;
#pragma unroll 2
        for (int kc = 0; kc < 8; ++kc) {
            const bf16x8_t dsf = sw_tr_frag(R3, wave * 32, kc * 16, lane);
#pragma unroll
            for (int db = 0; db < 2; ++db)
                acc_dq[db] = mfma_bf16_32x32x16(sw_linear_frag(R2, db * 32 + l31, kc * 16, hi), dsf, acc_dq[db]);
        }
        CX_STAMP(10);  // dQ products
        store_unrotated_rows_pre(Qt + wave * 4096, p.dqkv + (size_t)(t0 + wave * 32) * tok_stride + (size_t)h * DH, tok_stride,
                                 len - wave * 32, acc_dq, p.scale, rot, p.cosv != nullptr, hi, lane);
        CX_STAMP(11);  // dQ stored
        __syncthreads();  // LDS is restaged by the next problem
        CX_STAMP(12);  // barrier
    }
}

#ifndef CX_PRODUCT
__global__ void attn_keep_mask_kernel(AttnParams p, unsigned char* keep, int B, int S) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = (long)B * p.H * S * (S / 4);
    if (i >= n) return;
    const int kg = (int)(i % (S / 4));
    const int q = (int)((i / (S / 4)) % S);
    const uint32_t unit = (uint32_t)(i / ((long)S * (S / 4)));
    float k4[4];
    attn_keep4(p, unit, q, kg * 4, k4);
    unsigned char* o = keep + ((long)unit * S + q) * S + kg * 4;
    o[0] = k4[0] != 0.f; o[1] = k4[1] != 0.f; o[2] = k4[2] != 0.f; o[3] = k4[3] != 0.f;
}
#endif

inline int done() { return hipGetLastError() == hipSuccess ? CX_OK : CX_ERR_LAUNCH; }

// ------------------------------------------------------------------------ forward, sequences <= 128, lean-VALU form
// attn_fwd_s128_kernel issues ~950 VALU instructions per problem and wave against 32 MFMAs and stores its output as
// 16-byte pieces of 32 different rows per instruction.  Same data flow (default, cx_attn_set_fwd_s128(2); results
// within 1 bf16 ulp of mode 0) with the avoidable VALU work (~37 %) removed and the output staged through LDS so that
// it leaves as full 128-byte rows.  At T = 131072: the VALU diet alone 195-202 us vs 202-204 us (nothing), with the
// full-row stores 167-173 us -- the partial-line writes were the bound, not load latency (mode 1) or VALU issue.
//  * V is staged row-major ([key][64 d], 16-B stores, XOR-swizzled so that both the 16-B writers and the transposing
//    readers are bank-conflict free) and its fragments come from ds_read_b64_tr_b16 in the accumulator-register key
//    order pack_frag produces -- no per-element repacking into a transposed tile;
//  * full-length sequences (len == 128, the metric) skip the key mask;
//  * the softmax scale is folded into the exponent: p = exp2(fma(s, c, -max(s) * c)).
CX_DEVICE int v128_off(int key, int d) { return key * 128 + ((((d >> 3) ^ (((key >> 1) & 1) << 2)) << 4)) + (d & 7) * 2; }
// A[i = d0 + (lane&31)][k]: keys kbase + 4*hi + {0..3}, kbase + 8 + 4*hi + {0..3}  (pairs with pack_frag(s, half),
// kbase = 32*kb + 16*half)
CX_DEVICE bf16x8_t v128_tr_frag(const char* tile, int d0, int kbase, int lane) {
    const int g = lane >> 4, pp = lane & 15;
    const int t = kbase + 4 * (g >> 1) + (pp >> 2);
    const int f = d0 + 16 * (g & 1) + 4 * (pp & 3);
    union { bf16x4_t h[2]; bf16x8_t v; } u;
    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4_attn)(tile + v128_off(t, f)));
    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4_attn)(tile + v128_off(t + 8, f)));
    return u.v;
}

//  * DROP (attn_pdrop > 0, the reference's bert-base-uncased recipes: sc/models/encoder/bert.py:19-21 carries the hub config's
//    attention_probs_dropout_prob = 0.1): O accumulates P * keep / (1 - p), the normaliser is the undropped row sum; the lane
//    holds 4 consecutive keys of ONE query per accumulator quad = one Philox call (round 4; before, p > 0 fell to the general
//    streaming kernels at 1.6 x the attention time).
template <bool DROP>
__global__ __launch_bounds__(256, 2) void attn_fwd_s128v_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(16))) char smem[16384 * 3];
    char* Qs = smem;
    char* Ks = smem + 16384;
    char* Vs = smem + 32768;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    const int h = blockIdx.x, b = blockIdx.y;
    const int t0 = p.cu[b], len = p.cu[b + 1] - t0;
    if (len <= 0) return;
    const size_t tok_stride = (size_t)3 * p.H * DH;
    const bf16_t* qbase = p.qkv + (size_t)h * DH;
    const bf16_t* kbase = qbase + (size_t)p.H * DH;
    const bf16_t* vbase = kbase + (size_t)p.H * DH;
    uint4 qlo[2], qhi[2], klo[2], khi[2], vv[4];
    float4 cs[2][2], sn[2][2];
    const int cp = tid & 3;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        int r = it * 64 + (tid >> 2);
        r = r < len ? r : len - 1;
        const bf16_t* qrow = qbase + (size_t)(t0 + r) * tok_stride;
        const bf16_t* krow = kbase + (size_t)(t0 + r) * tok_stride;
        qlo[it] = *reinterpret_cast<const uint4*>(qrow + cp * 8);
        qhi[it] = *reinterpret_cast<const uint4*>(qrow + 32 + cp * 8);
        klo[it] = *reinterpret_cast<const uint4*>(krow + cp * 8);
        khi[it] = *reinterpret_cast<const uint4*>(krow + 32 + cp * 8);
        if (p.cosv) {
            const float* c = p.cosv + (size_t)r * 32 + cp * 8;
            const float* sp = p.sinv + (size_t)r * 32 + cp * 8;
            cs[it][0] = *reinterpret_cast<const float4*>(c);
            cs[it][1] = *reinterpret_cast<const float4*>(c + 4);
            sn[it][0] = *reinterpret_cast<const float4*>(sp);
            sn[it][1] = *reinterpret_cast<const float4*>(sp + 4);
        }
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {  // V: key row it*32 + tid/8, 16-B chunk tid%8
        int key = it * 32 + (tid >> 3);
        key = key < len ? key : len - 1;
        vv[it] = *reinterpret_cast<const uint4*>(vbase + (size_t)(t0 + key) * tok_stride + (tid & 7) * 8);
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int r = it * 64 + (tid >> 2);
        uint4 a_lo = qlo[it], a_hi = qhi[it], b_lo = klo[it], b_hi = khi[it];
        if (p.cosv) {
            rot8(qlo[it], qhi[it], cs[it], sn[it], a_lo, a_hi);
            rot8(klo[it], khi[it], cs[it], sn[it], b_lo, b_hi);
        }
        *reinterpret_cast<uint4*>(Qs + tile64_off(r, cp)) = a_lo;
        *reinterpret_cast<uint4*>(Qs + tile64_off(r, cp + 4)) = a_hi;
        *reinterpret_cast<uint4*>(Ks + tile64_off(r, cp)) = b_lo;
        *reinterpret_cast<uint4*>(Ks + tile64_off(r, cp + 4)) = b_hi;
    }
#pragma unroll
    for (int it = 0; it < 4; ++it)
        *reinterpret_cast<uint4*>(Vs + v128_off(it * 32 + (tid >> 3), (tid & 7) * 8)) = vv[it];
    __syncthreads();

    bf16x8_t qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = lds_read_frag(Qs, tile64_off(wave * 32 + l31, ks * 2 + hi));
    const float sc2 = p.scale * LOG2E;
    float s[4][16];
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        f32x16_t a;
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            a = mfma_bf16_32x32x16(lds_read_frag(Ks, tile64_off(kb * 32 + l31, ks * 2 + hi)), qf[ks], a);
        if (len < 128) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = kb * 32 + acc_row(r, hi) < len ? a[r] : -INFINITY;
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = a[r];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mxs = mx * sc2;  // softmax_scale > 0: the maximum commutes with the scale
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[kb][r] = fast_exp2(__builtin_fmaf(s[kb][r], sc2, -mxs));
            psum += s[kb][r];
        }
    if constexpr (DROP) {
        const int qi = wave * 32 + l31;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                float k4[4];
                attn_keep4(p, (uint32_t)(b * p.H + h), qi, kb * 32 + 8 * qd + 4 * hi, k4);
#pragma unroll
                for (int e = 0; e < 4; ++e) s[kb][4 * qd + e] *= k4[e];
            }
    }
    f32x16_t acc_o[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[db][r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const bf16x8_t pf = pack_frag(s[kb], half);
#pragma unroll
            for (int db = 0; db < 2; ++db)
                acc_o[db] = mfma_bf16_32x32x16(v128_tr_frag(Vs, db * 32, kb * 32 + 16 * half, lane), pf, acc_o[db]);
        }
    const float l_tot = psum + __shfl_xor(psum, 32, 64);
    const float inv = 1.f / l_tot;
    // Output through LDS: the accumulator layout gives a lane 4 consecutive d of ONE query row (8 B), so direct stores
    // touch 32 rows x 16 B per instruction; staged in this wave's own (now dead) Q rows they leave as 16 B per lane,
    // 8 lanes per 128-B row.  Only own-wave rows are touched: no barrier, the write -> read order is the wave's own.
    {
        const int qrow = wave * 32 + l31;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                uint2 pk;
                pk.x = pack_bf16x2(acc_o[db][4 * qd] * inv, acc_o[db][4 * qd + 1] * inv);
                pk.y = pack_bf16x2(acc_o[db][4 * qd + 2] * inv, acc_o[db][4 * qd + 3] * inv);
                *reinterpret_cast<uint2*>(Qs + tile64_off(qrow, db * 4 + qd) + hi * 8) = pk;
            }
        if (qrow < len && hi == 0) p.lse[(size_t)h * p.T + t0 + qrow] = (mxs + log2f(l_tot)) * LN2;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int q = wave * 32 + it * 8 + (lane >> 3), chunk = lane & 7;
            const uint4 v = *reinterpret_cast<const uint4*>(Qs + tile64_off(q, chunk));
            if (q < len) *reinterpret_cast<uint4*>(p.out + ((size_t)(t0 + q) * p.H + h) * DH + chunk * 8) = v;
        }
    }
}

#include "attn_s256.inc"
#include "attn_long.inc"

#ifndef CX_PRODUCT
int g_fwd_s128 = 2;  // cx_attn_set_fwd_s128: non-zero = the single-pass kernels for max_seqlen <= 128 / <= 256 (default), 0 = the general streaming kernel (A/B, tests)
int g_bwd_long = 1;  // cx_attn_set_bwd_long: 1 = attn_bwd_dq_long / attn_bwd_dkv_long for max_seqlen > 128 without rotate-on-load (default), 0 = round 1's pair
int g_fwd_long = 1;  // cx_attn_set_fwd_long: 1 = attn_fwd_long_kernel for max_seqlen > 256 without rotate-on-load (default), 0 = attn_fwd_kernel
int g_bwd_s128 = 3;  // cx_attn_set_bwd_s128: max_seqlen <= 128 -> 3 = the fused persistent kernel (default), 0 = the general streaming kernels (A/B, tests)
#endif

}  // namespace

extern "C" {

#ifndef CX_PRODUCT
int g_attn_prio = 0;
void cx_attn_set_prio(int on) { g_attn_prio = on ? 1 : 0; }
void cx_attn_set_bwd_s128(int mode) { g_bwd_s128 = mode == 0 ? 0 : 3; }
void cx_attn_set_fwd_s128(int mode) { g_fwd_s128 = mode == 0 ? 0 : 2; }
void cx_attn_set_fwd_long(int on) { g_fwd_long = on ? 1 : 0; }
void cx_attn_set_bwd_long(int on) { g_bwd_long = on ? 1 : 0; }
#endif

int cx_attn_varlen_fwd(const uint16_t* qkv, const int32_t* cu_seqlens, const float* rot_cos, const float* rot_sin,
                       uint16_t* out, float* lse, int B, int H, int T, int max_seqlen, float softmax_scale,
                       void* stream) {
    if (B <= 0 || T <= 0 || max_seqlen <= 0) return CX_OK;
    if (!qkv || !cu_seqlens || !out || !lse) return CX_ERR_ARG;
    if ((rot_cos == nullptr) != (rot_sin == nullptr)) return CX_ERR_ARG;
    AttnParams p = {};
    p.qkv = qkv; p.cu = cu_seqlens; p.cosv = rot_cos; p.sinv = rot_sin; p.lcos = rot_cos; p.lsin = rot_sin; p.out = out; p.lse = lse;
    p.H = H; p.T = T; p.scale = softmax_scale;
#ifndef CX_PRODUCT
    const bool fwd_long = g_fwd_long != 0;   // (dev library: cx_attn_set_fwd_long(0) keeps round 1's streaming kernel for A/B)
    const bool single_pass = g_fwd_s128 != 0;
#elif defined(CX_AB_R5_ROUTES)   // evidence builds only (scripts/build_variant.py r5routes ...): round 5's kernel routing for a same-box A/B of the legs
    constexpr bool fwd_long = false, single_pass = true;
#else
    constexpr bool fwd_long = true, single_pass = true;
#endif
    if (max_seqlen <= 128 && single_pass) {  // one workgroup per (sequence, head) problem, single pass
        hipLaunchKernelGGL(attn_fwd_s128v_kernel<false>, dim3(H, B), dim3(256), 0, (hipStream_t)stream, p);
#ifndef CX_AB_R5_ROUTES
    } else if (max_seqlen <= 256 && single_pass) {  // the same with K / V of up to 256 rows resident (round 6: the ViT's 197 tokens)
        static CxLdsOptIn lds_f256;
        if (!lds_f256.ensure(reinterpret_cast<const void*>(&attn_fwd_s256_kernel<false>), S256_LDS_FWD)) return CX_ERR_LAUNCH;
        hipLaunchKernelGGL(attn_fwd_s256_kernel<false>, dim3(H, B), dim3(256), S256_LDS_FWD, (hipStream_t)stream, p);
#endif
    } else if (max_seqlen > 128 && !rot_cos && fwd_long) {  // long sequences, q / k already rotated (or no rotary at all): 64 rows per wave, K / V by LDS-DMA (round 6)
        hipLaunchKernelGGL(attn_fwd_long_kernel<false>, dim3((max_seqlen + 255) / 256, H, B), dim3(256), LONG_LDS, (hipStream_t)stream, p);
    } else {
        dim3 grid((max_seqlen + 127) / 128, H, B);
        hipLaunchKernelGGL(attn_fwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, p);
    }
    return done();
}

int cx_attn_varlen_bwd(const uint16_t* dout, const uint16_t* qkv, const uint16_t* out, const float* lse,
                       const int32_t* cu_seqlens, const float* rot_cos, const float* rot_sin, float* delta,
                       uint16_t* dqkv, int B, int H, int T, int max_seqlen, float softmax_scale, void* stream) {
    if (B <= 0 || T <= 0 || max_seqlen <= 0) return CX_OK;
    if (!dout || !qkv || !out || !lse || !cu_seqlens || !delta || !dqkv) return CX_ERR_ARG;
    if ((rot_cos == nullptr) != (rot_sin == nullptr)) return CX_ERR_ARG;
    AttnParams p = {};
    p.qkv = qkv; p.cu = cu_seqlens; p.cosv = rot_cos; p.sinv = rot_sin; p.lcos = rot_cos; p.lsin = rot_sin;
    p.out = const_cast<uint16_t*>(out); p.lse = const_cast<float*>(lse);
    p.dout = dout; p.delta = delta; p.dqkv = dqkv;
    p.H = H; p.T = T; p.scale = softmax_scale;
#ifndef CX_PRODUCT
    p.prio = g_attn_prio;
    const int bwd_mode = g_bwd_s128;
    const bool bwd_long = g_bwd_long != 0;
#elif defined(CX_AB_R5_ROUTES)
    constexpr int bwd_mode = 3;
    constexpr bool bwd_long = false;
#else
    constexpr int bwd_mode = 3;
    constexpr bool bwd_long = true;
#endif
    if (max_seqlen <= 128 && bwd_mode == 3) {  // fused persistent kernel, 80 KiB LDS: two workgroups per CU
        static CxLdsOptIn lds2;
        if (!lds2.ensure(reinterpret_cast<const void*>(&attn_bwd_fused2_s128_kernel<false>), FUSED2_LDS)) return CX_ERR_LAUNCH;
        const int n_units = B * H;
        hipLaunchKernelGGL(attn_bwd_fused2_s128_kernel<false>, dim3(n_units < 512 ? n_units : 512), dim3(256), FUSED2_LDS,
                           (hipStream_t)stream, p, B);
        return done();
    }
    if (max_seqlen > 128 && !p.lcos && bwd_long) {   // second-generation streaming kernels (round 6): no delta pass
        hipLaunchKernelGGL(attn_bwd_dq_long_kernel<false>, dim3((max_seqlen + 255) / 256, H, B), dim3(256), LONG_LDS, (hipStream_t)stream, p);
        hipLaunchKernelGGL(attn_bwd_dkv_long_kernel<false>, dim3((max_seqlen + 127) / 128, H, B), dim3(256), LONG_LDS_DKV, (hipStream_t)stream, p);
        return done();
    }
    long nthreads = (long)T * H * 8;
    int g = (int)((nthreads + 255) / 256);
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(attn_delta_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, p);
    dim3 grid((max_seqlen + 127) / 128, H, B);
    hipLaunchKernelGGL(attn_bwd_dq_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, p);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, p);
    return done();
}

// Long sequences with rotary: the engine rotates q and k in place once (cx_rotary_qkv_inplace), runs the forward with no
// tables at all, and calls this backward: the streaming kernels read the saved (rotated) q / k as they are -- they used to
// re-rotate every K row once per 128-query block and every Q row once per 128-key block, 18-23 % of their time at S = 2048
// -- and only the gradients' inverse rotation at the stores uses the tables.  Any length (always the general kernels).
int cx_attn_varlen_bwd_prerotated(const uint16_t* dout, const uint16_t* qkv_rotated, const uint16_t* out, const float* lse,
                                  const int32_t* cu_seqlens, const float* rot_cos, const float* rot_sin, float* delta,
                                  uint16_t* dqkv, int B, int H, int T, int max_seqlen, float softmax_scale, void* stream) {
    if (B <= 0 || T <= 0 || max_seqlen <= 0) return CX_OK;
    if (!dout || !qkv_rotated || !out || !lse || !cu_seqlens || !delta || !dqkv || !rot_cos || !rot_sin) return CX_ERR_ARG;
    AttnParams p = {};
    p.qkv = qkv_rotated; p.cu = cu_seqlens; p.cosv = rot_cos; p.sinv = rot_sin; p.lcos = nullptr; p.lsin = nullptr;
    p.out = const_cast<uint16_t*>(out); p.lse = const_cast<float*>(lse);
    p.dout = dout; p.delta = delta; p.dqkv = dqkv;
    p.H = H; p.T = T; p.scale = softmax_scale;
#ifndef CX_PRODUCT
    const bool bwd_long = g_bwd_long != 0;
#elif defined(CX_AB_R5_ROUTES)
    constexpr bool bwd_long = false;
#else
    constexpr bool bwd_long = true;
#endif
    if (bwd_long) {   // second-generation streaming kernels (round 6): delta inside the dQ kernel
        hipLaunchKernelGGL(attn_bwd_dq_long_kernel<false>, dim3((max_seqlen + 255) / 256, H, B), dim3(256), LONG_LDS, (hipStream_t)stream, p);
        hipLaunchKernelGGL(attn_bwd_dkv_long_kernel<false>, dim3((max_seqlen + 127) / 128, H, B), dim3(256), LONG_LDS_DKV, (hipStream_t)stream, p);
        return done();
    }
    long nthreads = (long)T * H * 8;
    int g = (int)((nthreads + 255) / 256);
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(attn_delta_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, p);
    dim3 grid((max_seqlen + 127) / 128, H, B);
    hipLaunchKernelGGL(attn_bwd_dq_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, p);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, p);
    return done();
}

// attention dropout > 0 (the reference's bert-base-uncased recipes train with attention_probs_dropout_prob = 0.1,
// sc/models/encoder/bert.py:19-21): max_seqlen <= 128 runs the <DROP> instantiations of the single-pass forward and of the
// fused persistent backward (round 4: separate instantiations, the p = 0 hot path keeps its registers), longer sequences the
// general streaming kernels.
int cx_attn_varlen_dropout_fwd(const uint16_t* qkv, const int32_t* cu_seqlens, const float* rot_cos, const float* rot_sin,
                               uint16_t* out, float* lse, int B, int H, int T, int max_seqlen, float softmax_scale, float p_drop,
                               unsigned long long seed, unsigned long long offset, unsigned int site, void* stream) {
    if (B <= 0 || T <= 0 || max_seqlen <= 0) return CX_OK;
    if (!qkv || !cu_seqlens || !out || !lse) return CX_ERR_ARG;
    if ((rot_cos == nullptr) != (rot_sin == nullptr)) return CX_ERR_ARG;
    if (!(p_drop > 0.f) || p_drop >= 1.f || max_seqlen >= (1 << 20) || (long)B * H >= (1L << 24)) return CX_ERR_ARG;
    AttnParams p = {};
    p.qkv = qkv; p.cu = cu_seqlens; p.cosv = rot_cos; p.sinv = rot_sin; p.lcos = rot_cos; p.lsin = rot_sin; p.out = out; p.lse = lse;
    p.H = H; p.T = T; p.scale = softmax_scale;
    p.drop = CxDropout{p_drop, seed, offset}; p.drop_site = site;
#ifndef CX_PRODUCT
    const bool single_pass = g_fwd_s128 != 0;   // (dev library: cx_attn_set_fwd_s128(0) keeps the general kernel for A/B)
#else
    constexpr bool single_pass = true;
#endif
    if (max_seqlen <= 128 && single_pass) {   // the single-pass kernel with the mask (round 4)
        hipLaunchKernelGGL(attn_fwd_s128v_kernel<true>, dim3(H, B), dim3(256), 0, (hipStream_t)stream, p);
        return done();
    }
#ifndef CX_AB_R5_ROUTES
    if (max_seqlen <= 256 && single_pass) {
        static CxLdsOptIn lds_f256d;
        if (!lds_f256d.ensure(reinterpret_cast<const void*>(&attn_fwd_s256_kernel<true>), S256_LDS_FWD)) return CX_ERR_LAUNCH;
        hipLaunchKernelGGL(attn_fwd_s256_kernel<true>, dim3(H, B), dim3(256), S256_LDS_FWD, (hipStream_t)stream, p);
        return done();
    }
    if (max_seqlen > 256 && !rot_cos && single_pass) {
        hipLaunchKernelGGL(attn_fwd_long_kernel<true>, dim3((max_seqlen + 255) / 256, H, B), dim3(256), LONG_LDS, (hipStream_t)stream, p);
        return done();
    }
#endif
    dim3 grid((max_seqlen + 127) / 128, H, B);
    hipLaunchKernelGGL((attn_fwd_kernel<false, true>), grid, dim3(256), 0, (hipStream_t)stream, p);
    return done();
}

int cx_attn_varlen_dropout_bwd(const uint16_t* dout, const uint16_t* qkv, const uint16_t* out, const float* lse,
                               const int32_t* cu_seqlens, const float* rot_cos, const float* rot_sin, float* delta,
                               uint16_t* dqkv, int B, int H, int T, int max_seqlen, float softmax_scale, float p_drop,
                               unsigned long long seed, unsigned long long offset, unsigned int site, void* stream) {
    if (B <= 0 || T <= 0 || max_seqlen <= 0) return CX_OK;
    if (!dout || !qkv || !out || !lse || !cu_seqlens || !delta || !dqkv) return CX_ERR_ARG;
    if ((rot_cos == nullptr) != (rot_sin == nullptr)) return CX_ERR_ARG;
    if (!(p_drop > 0.f) || p_drop >= 1.f || max_seqlen >= (1 << 20) || (long)B * H >= (1L << 24)) return CX_ERR_ARG;
    AttnParams p = {};
    p.qkv = qkv; p.cu = cu_seqlens; p.cosv = rot_cos; p.sinv = rot_sin; p.lcos = rot_cos; p.lsin = rot_sin;
    p.out = const_cast<uint16_t*>(out); p.lse = const_cast<float*>(lse);
    p.dout = dout; p.delta = delta; p.dqkv = dqkv;
    p.H = H; p.T = T; p.scale = softmax_scale;
    p.drop = CxDropout{p_drop, seed, offset}; p.drop_site = site;
#ifndef CX_PRODUCT
    const bool fused = g_bwd_s128 != 0;         // (dev library: cx_attn_set_bwd_s128(0) keeps the general kernels for A/B)
#else
    constexpr bool fused = true;
#endif
    if (max_seqlen <= 128 && fused) {   // fused persistent kernel with the mask (delta inline: `delta` is not written)
        static CxLdsOptIn lds2d;
        if (!lds2d.ensure(reinterpret_cast<const void*>(&attn_bwd_fused2_s128_kernel<true>), FUSED2_LDS)) return CX_ERR_LAUNCH;
        const int n_units = B * H;
        hipLaunchKernelGGL(attn_bwd_fused2_s128_kernel<true>, dim3(n_units < 512 ? n_units : 512), dim3(256), FUSED2_LDS,
                           (hipStream_t)stream, p, B);
        return done();
    }
#ifndef CX_AB_R5_ROUTES
    if (max_seqlen > 128 && !p.lcos && fused) {   // second-generation streaming kernels with the mask (round 6)
        hipLaunchKernelGGL(attn_bwd_dq_long_kernel<true>, dim3((max_seqlen + 255) / 256, H, B), dim3(256), LONG_LDS, (hipStream_t)stream, p);
        hipLaunchKernelGGL(attn_bwd_dkv_long_kernel<true>, dim3((max_seqlen + 127) / 128, H, B), dim3(256), LONG_LDS_DKV, (hipStream_t)stream, p);
        return done();
    }
#endif
    long nthreads = (long)T * H * 8;
    int g = (int)((nthreads + 255) / 256);
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(attn_delta_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, p);
    dim3 grid((max_seqlen + 127) / 128, H, B);
    hipLaunchKernelGGL((attn_bwd_dq_kernel<false, true>), grid, dim3(256), 0, (hipStream_t)stream, p);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<false, true>), grid, dim3(256), 0, (hipStream_t)stream, p);
    return done();
}

#ifndef CX_PRODUCT
// test helper (dev library): keep[b][h][q][key] in {0, 1} of the mask the kernels above apply
int cx_attn_dropout_keep_mask(unsigned char* keep, int B, int H, int S, float p_drop, unsigned long long seed,
                              unsigned long long offset, unsigned int site, void* stream) {
    if (!keep || S % 4 != 0) return CX_ERR_ARG;
    AttnParams p = {};
    p.H = H;
    p.drop = CxDropout{p_drop, seed, offset}; p.drop_site = site;
    const long n = (long)B * H * S * (S / 4);
    hipLaunchKernelGGL(attn_keep_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, keep, B, S);
    return done();
}
#endif

// kv-packed cross-attention (K3): queries (Tq, H, 64) attend to keys / values (Tk, 2, H, 64) of the same batch entry;
// the general (streaming, any length) kernels with separate query and key views.  Work is tiled 128 queries x 64 keys:
// the reference's use (FlashAttentionPooling, ONE latent query per sequence) fills 1/128 of a query tile -- it is a
// bandwidth-sized op there (it reads kv once) and runs at that speed, not at MFMA speed.
int cx_attn_varlen_kvpacked_fwd(const uint16_t* q, const uint16_t* kv, const int32_t* cu_seqlens_q,
                                const int32_t* cu_seqlens_k, uint16_t* out, float* lse, int B, int H, int Tq,
                                int max_seqlen_q, int max_seqlen_k, float softmax_scale, void* stream) {
    if (B <= 0 || Tq <= 0 || max_seqlen_q <= 0) return CX_OK;
    if (!q || !kv || !cu_seqlens_q || !cu_seqlens_k || !out || !lse || max_seqlen_k < 0) return CX_ERR_ARG;
    AttnParams p = {};
    p.qkv = q; p.kv = kv; p.cu = cu_seqlens_q; p.cu_k = cu_seqlens_k; p.out = out; p.lse = lse;
    p.H = H; p.T = Tq; p.scale = softmax_scale;
    dim3 grid((max_seqlen_q + 127) / 128, H, B);
    hipLaunchKernelGGL(attn_fwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, p);
    return done();
}

int cx_attn_varlen_kvpacked_bwd(const uint16_t* dout, const uint16_t* q, const uint16_t* kv, const uint16_t* out,
                                const float* lse, const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k, float* delta,
                                uint16_t* dq, uint16_t* dkv, int B, int H, int Tq, int max_seqlen_q, int max_seqlen_k,
                                float softmax_scale, void* stream) {
    if (B <= 0) return CX_OK;
    if (!dout || !q || !kv || !out || !lse || !cu_seqlens_q || !cu_seqlens_k || !delta || !dq || !dkv) return CX_ERR_ARG;
    AttnParams p = {};
    p.qkv = q; p.kv = kv; p.cu = cu_seqlens_q; p.cu_k = cu_seqlens_k;
    p.out = const_cast<uint16_t*>(out); p.lse = const_cast<float*>(lse);
    p.dout = dout; p.delta = delta; p.dqkv = dq; p.dkv = dkv;
    p.H = H; p.T = Tq; p.scale = softmax_scale;
    if (Tq > 0) {
        long nthreads = (long)Tq * H * 8;
        int g = (int)((nthreads + 255) / 256);
        if (g > 2048) g = 2048;
        hipLaunchKernelGGL(attn_delta_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, p);
        if (max_seqlen_q > 0)
            hipLaunchKernelGGL(attn_bwd_dq_kernel<true>, dim3((max_seqlen_q + 127) / 128, H, B), dim3(256), 0,
                               (hipStream_t)stream, p);
    }
    if (max_seqlen_k > 0)
        hipLaunchKernelGGL(attn_bwd_dkv_kernel<true>, dim3((max_seqlen_k + 127) / 128, H, B), dim3(256), 0,
                           (hipStream_t)stream, p);
    return done();
}

}  // extern "C"
