// layernorm.hip -- K5/K6 of SURVEY.md §2b: fused (residual add +) LayerNorm forward/backward, and the
// BertEmbeddings gather fused with the embedding LayerNorm (a11).  HBM-bound: one wave per row, every lane owns
// 4 consecutive columns per 256-column chunk (8-B bf16 / 16-B fp32 accesses), statistics in fp32, one pass over
// the row held in registers.  Replaces flash_attn.ops.layer_norm.{dropout_add_layer_norm, layer_norm} at
// sc/layers/block.py:422-431,453-462 and sc/models/encoder/modeling_nomic_bert.py:534 (dropout p = 0).
#include "cx_common.h"
#include "../../include/contrastors_hip.h"

namespace {

CX_DEVICE void load4_bf16(const bf16_t* p, float (&v)[4]) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    v[0] = bf16lo_to_f32(u.x); v[1] = bf16hi_to_f32(u.x);
    v[2] = bf16lo_to_f32(u.y); v[3] = bf16hi_to_f32(u.y);
}
CX_DEVICE void store4_bf16(bf16_t* p, const float (&v)[4]) {
    uint2 u;
    u.x = pack_bf16x2(v[0], v[1]);
    u.y = pack_bf16x2(v[2], v[3]);
    *reinterpret_cast<uint2*>(p) = u;
}
CX_DEVICE void load4_f32(const float* p, float (&v)[4]) {
    const float4 u = *reinterpret_cast<const float4*>(p);
    v[0] = u.x; v[1] = u.y; v[2] = u.z; v[3] = u.w;
}

CX_DEVICE void unpack4_bf16(const uint2 u, float (&v)[4]) {
    v[0] = bf16lo_to_f32(u.x); v[1] = bf16hi_to_f32(u.x);
    v[2] = bf16lo_to_f32(u.y); v[3] = bf16hi_to_f32(u.y);
}
// Raw (packed bf16) row fetch, issued one row ahead of its use: with one 1.5 KiB row per wave and ~16 waves per CU the
// kernels had only ~24-36 KiB in flight per CU, well short of what latency x HBM bandwidth asks for (~60 KiB).
template <int NCH>
CX_DEVICE void fetch_row(const bf16_t* __restrict__ base, int row, int lane, uint2 (&raw)[NCH]) {
#pragma unroll
    for (int i = 0; i < NCH; ++i)
        raw[i] = *reinterpret_cast<const uint2*>(base + (size_t)row * (NCH * 256) + (i * 64 + lane) * 4);
}

template <int NCH>
CX_DEVICE void row_stats(const float (&z)[NCH][4], int d, float eps, float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) s += z[i][e];
    mean = wave_sum(s) / (float)d;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float c = z[i][e] - mean;
            v += c * c;
        }
    rstd = rsqrtf(wave_sum(v) / (float)d + eps);
}

template <int NCH>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const bf16_t* __restrict__ x0, const bf16_t* __restrict__ res,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, bf16_t* __restrict__ out,
                                                     bf16_t* z_out, float* __restrict__ mean_o,
                                                     float* __restrict__ rstd_o, int rows, float eps) {
    constexpr int D = NCH * 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float g[NCH][4], b[NCH][4];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        load4_f32(gamma + (i * 64 + lane) * 4, g[i]);
        load4_f32(beta + (i * 64 + lane) * 4, b[i]);
    }
    const int stride = gridDim.x * 4;
    int row = blockIdx.x * 4 + wave;
    uint2 nx[NCH], nr[NCH];
    if (row < rows) {
        fetch_row<NCH>(x0, row, lane, nx);
        if (res) fetch_row<NCH>(res, row, lane, nr);
    }
    for (; row < rows; row += stride) {
        float z[NCH][4];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            unpack4_bf16(nx[i], z[i]);
            if (res) {
                float r[4];
                unpack4_bf16(nr[i], r);
#pragma unroll
                for (int e = 0; e < 4; ++e) z[i][e] += r[e];
            }
        }
        if (row + stride < rows) {  // next row's loads fly under this row's statistics and stores
            fetch_row<NCH>(x0, row + stride, lane, nx);
            if (res) fetch_row<NCH>(res, row + stride, lane, nr);
        }
        float mean, rstd;
        row_stats<NCH>(z, D, eps, mean, rstd);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const size_t off = (size_t)row * D + (i * 64 + lane) * 4;
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (z[i][e] - mean) * rstd * g[i][e] + b[i][e];
            store4_bf16(out + off, o);
            if (z_out) store4_bf16(z_out + off, z[i]);
        }
        if (lane == 0) {
            mean_o[row] = mean;
            rstd_o[row] = rstd;
        }
    }
}

// Shared tail of both backward kernels: fold the per-wave dgamma/dbeta partials of one block and atomically add.
// Two-stage variant used when the caller provides a workspace: block `blk` writes its column partials to
// part[blk][0..D) (dgamma) and part[blk][D..2D) (dbeta) with plain stores; ln_param_reduce_kernel sums the blocks.
template <int NCH>
CX_DEVICE void store_param_partials(float (&dg)[NCH][4], float (&db)[NCH][4], float* part, float* smem) {
    constexpr int D = NCH * 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            smem[(0 * 4 + wave) * D + (i * 64 + lane) * 4 + e] = dg[i][e];
            smem[(1 * 4 + wave) * D + (i * 64 + lane) * 4 + e] = db[i][e];
        }
    __syncthreads();
    float* mine = part + (size_t)blockIdx.x * 2 * D;
    for (int c = threadIdx.x; c < 2 * D; c += 256) {
        const int which = c / D, col = c - which * D;
        float sacc = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) sacc += smem[(which * 4 + w) * D + col];
        mine[c] = sacc;
    }
}
// third vector of a backward that also returns the column sums of its dz (the bias gradient of the Linear whose output
// the LayerNorm consumed): block partials in part3[blk][D], folded by ln_param_reduce_kernel like the other two
template <int NCH>
CX_DEVICE void store_colsum_partials(float (&dc)[NCH][4], float* part3, float* smem) {
    constexpr int D = NCH * 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) smem[wave * D + (i * 64 + lane) * 4 + e] = dc[i][e];
    __syncthreads();
    float* mine = part3 + (size_t)blockIdx.x * D;
    for (int c = threadIdx.x; c < D; c += 256) mine[c] = smem[c] + smem[D + c] + smem[2 * D + c] + smem[3 * D + c];
}
__global__ __launch_bounds__(256) void colsum_reduce_kernel(const float* __restrict__ part3, float* dst, int nblocks, int D) {
    __shared__ float red[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
    float sacc = 0.f;
    if (col < D) {
#pragma unroll 4
        for (int b = grp; b < nblocks; b += 4) sacc += part3[(size_t)b * D + col];
    }
    red[grp][threadIdx.x & 63] = sacc;
    __syncthreads();
    if (grp == 0 && col < D) dst[col] += red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// 64 columns x 4 block-groups per workgroup: each thread sums every 4th block's partial for its column (independent,
// unrolled loads), then the 4 groups are folded through LDS.
__global__ __launch_bounds__(256) void ln_param_reduce_kernel(const float* __restrict__ part, float* dgamma, float* dbeta,
                                                              int nblocks, int D) {
    __shared__ float red[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
    float sacc = 0.f;
    if (col < 2 * D) {
#pragma unroll 4
        for (int b = grp; b < nblocks; b += 4) sacc += part[(size_t)b * 2 * D + col];
    }
    red[grp][threadIdx.x & 63] = sacc;
    __syncthreads();
    if (grp == 0 && col < 2 * D) {
        const float tot = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        float* dst = col < D ? dgamma : dbeta;
        if (dst) dst[col < D ? col : col - D] += tot;
    }
}

template <int NCH>
CX_DEVICE void flush_param_grads(float (&dg)[NCH][4], float (&db)[NCH][4], float* dgamma, float* dbeta,
                                 float* smem /* [2][4][D] */) {
    constexpr int D = NCH * 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            smem[(0 * 4 + wave) * D + (i * 64 + lane) * 4 + e] = dg[i][e];
            smem[(1 * 4 + wave) * D + (i * 64 + lane) * 4 + e] = db[i][e];
        }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
        float sg = 0.f, sb = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            sg += smem[(0 * 4 + w) * D + c];
            sb += smem[(1 * 4 + w) * D + c];
        }
        if (dgamma) unsafeAtomicAdd(dgamma + c, sg);
        if (dbeta) unsafeAtomicAdd(dbeta + c, sb);
    }
}

// CS (round 3): also accumulate the column sums of the dz this kernel writes -- dz is the gradient of (Linear output +
// residual), so its column sums ARE the bias gradient of that Linear (fc2 / out_proj; sc FusedDense backward's `db`).
// The standalone cx_bias_grad passes they replace re-read dz: 2 x T x d x 2 bytes per block, 6.5 % of the CLIP step.
// The sums take what is STORED (the bf16-rounded dz), like the standalone kernel did.
template <int NCH, bool CS>
__global__ __launch_bounds__(256, CS ? 3 : 4) void ln_bwd_kernel(const bf16_t* __restrict__ da, const bf16_t* __restrict__ dbb,
                                                     const bf16_t* __restrict__ z, const float* __restrict__ gamma,
                                                     const float* __restrict__ mean_i,
                                                     const float* __restrict__ rstd_i,
                                                     const bf16_t* __restrict__ dz_extra, bf16_t* __restrict__ dz,
                                                     float* dgamma, float* dbeta, float* part, float* part3, int rows) {
    constexpr int D = NCH * 256;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float g[NCH][4], dg[NCH][4], db[NCH][4], dc[CS ? NCH : 1][4];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        load4_f32(gamma + (i * 64 + lane) * 4, g[i]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            dg[i][e] = db[i][e] = 0.f;
            if (CS) dc[i][e] = 0.f;
        }
    }
    const int stride = gridDim.x * 4;
    int row = blockIdx.x * 4 + wave;
    uint2 na[NCH], nb[NCH], nz[NCH];
    float nmean = 0.f, nrstd = 0.f;
    if (row < rows) {
        fetch_row<NCH>(da, row, lane, na);
        if (dbb) fetch_row<NCH>(dbb, row, lane, nb);
        fetch_row<NCH>(z, row, lane, nz);
        nmean = mean_i[row];
        nrstd = rstd_i[row];
    }
    for (; row < rows; row += stride) {
        const float mean = nmean, rstd = nrstd;
        float dy[NCH][4], xh[NCH][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            unpack4_bf16(na[i], dy[i]);
            if (dbb) {
                float t[4];
                unpack4_bf16(nb[i], t);
#pragma unroll
                for (int e = 0; e < 4; ++e) dy[i][e] += t[e];
            }
            float zz[4];
            unpack4_bf16(nz[i], zz);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xh[i][e] = (zz[e] - mean) * rstd;
                const float wdy = g[i][e] * dy[i][e];
                s1 += wdy * xh[i][e];
                s2 += wdy;
                dg[i][e] += dy[i][e] * xh[i][e];
                db[i][e] += dy[i][e];
            }
        }
        if (row + stride < rows) {  // next row's loads fly under this row's reductions and stores
            fetch_row<NCH>(da, row + stride, lane, na);
            if (dbb) fetch_row<NCH>(dbb, row + stride, lane, nb);
            fetch_row<NCH>(z, row + stride, lane, nz);
            nmean = mean_i[row + stride];
            nrstd = rstd_i[row + stride];
        }
        s1 = wave_sum(s1) / (float)D;
        s2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const size_t off = (size_t)row * D + (i * 64 + lane) * 4;
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (g[i][e] * dy[i][e] - s1 * xh[i][e] - s2) * rstd;
            if (dz_extra) {
                float t[4];
                load4_bf16(dz_extra + off, t);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] += t[e];
            }
            if constexpr (CS) {
                uint2 u;
                u.x = pack_bf16x2(o[0], o[1]);
                u.y = pack_bf16x2(o[2], o[3]);
                *reinterpret_cast<uint2*>(dz + off) = u;
                float r[4];
                unpack4_bf16(u, r);
#pragma unroll
                for (int e = 0; e < 4; ++e) dc[i][e] += r[e];
            } else {
                store4_bf16(dz + off, o);
            }
        }
    }
    if (part) {
        store_param_partials<NCH>(dg, db, part, smem);
    } else {
        flush_param_grads<NCH>(dg, db, dgamma, dbeta, smem);
    }
    if constexpr (CS) store_colsum_partials<NCH>(dc, part3, smem);
}

// Backward of the LAST LayerNorm of a pooled encoder with the pooling backward folded in (round 3).  The gradient of the
// final hidden states is  dout[t] = w(t) * g[seq(t)]  with g = the (B, d) fp32 gradient of the pooled vector (the
// normalisation's backward applied to d(embedding)) and w = 1 / len (mean pooling) or [t is the sequence's first token]
// (cls pooling).  Materialising dout in bf16 first -- cx_pool_normalize_bwd, then ln_bwd_kernel -- rounded every row of
// it, and because the rows of one sequence carry the SAME vector the rounding errors of dbeta = sum_t dout[t] do not
// average out: on the reference's GradCache fixture the final LayerNorm's bias gradient was 3.7 % off the fp32 oracle
// where bf16-eager (fp32 LayerNorm) is 0.5 % off, ln_f.bias of ViT-B/16 0.18 % vs 0.04 %.  Here dout never leaves fp32
// and never touches HBM.  One workgroup per sequence (grid-stride), one wave per row; parameter gradients through the
// deterministic two-stage reduction.
template <int NCH, bool CS>
__global__ __launch_bounds__(256, CS ? 3 : 4) void ln_bwd_pooled_kernel(const float* __restrict__ demb, const float* __restrict__ emb,
                                                            const float* __restrict__ norm, const int32_t* __restrict__ cu,
                                                            int B, int mode, int normalize, const bf16_t* __restrict__ z,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean_i,
                                                            const float* __restrict__ rstd_i, bf16_t* __restrict__ dz,
                                                            float* dgamma, float* dbeta, float* part, float* part3) {
    constexpr int D = NCH * 256;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [8 * D] (parameter partials; the first D floats hold g) + 4
    float* red = smem + 8 * D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float g[NCH][4], dg[NCH][4], db[NCH][4], dc[CS ? NCH : 1][4];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        load4_f32(gamma + (i * 64 + lane) * 4, g[i]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            dg[i][e] = db[i][e] = 0.f;
            if (CS) dc[i][e] = 0.f;
        }
    }
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        const int t0 = cu[b], len = cu[b + 1] - t0;
        if (len <= 0) continue;  // (uniform per workgroup)
        // g = d(pooled vector), the arithmetic of pool_normalize_bwd_kernel
        float dot = 0.f;
        if (normalize) {
            for (int c = tid; c < D; c += 256) dot += demb[(size_t)b * D + c] * emb[(size_t)b * D + c];
            dot = wave_sum(dot);
            if (lane == 0) red[wave] = dot;
            __syncthreads();
            dot = red[0] + red[1] + red[2] + red[3];
        }
        const float inv_n = normalize ? 1.f / fmaxf(norm[b], 1e-12f) : 1.f;
        const float inv_len = (mode == 1) ? 1.f : 1.f / (float)len;
        for (int c = tid; c < D; c += 256) {
            const float gg = demb[(size_t)b * D + c];
            smem[c] = (normalize ? (gg - emb[(size_t)b * D + c] * dot) * inv_n : gg) * inv_len;
        }
        __syncthreads();
        float dy[NCH][4];
#pragma unroll
        for (int i = 0; i < NCH; ++i) load4_f32(smem + (i * 64 + lane) * 4, dy[i]);
        for (int r = wave; r < len; r += 4) {
            const int row = t0 + r;
            if (mode == 1 && r != 0) {   // cls pooling: only the first token has a gradient
                const float zero[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < NCH; ++i) store4_bf16(dz + (size_t)row * D + (i * 64 + lane) * 4, zero);
                continue;
            }
            const float mean = mean_i[row], rstd = rstd_i[row];
            float xh[NCH][4];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                float zz[4];
                load4_bf16(z + (size_t)row * D + (i * 64 + lane) * 4, zz);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xh[i][e] = (zz[e] - mean) * rstd;
                    const float wdy = g[i][e] * dy[i][e];
                    s1 += wdy * xh[i][e];
                    s2 += wdy;
                    dg[i][e] += dy[i][e] * xh[i][e];
                    db[i][e] += dy[i][e];
                }
            }
            s1 = wave_sum(s1) / (float)D;
            s2 = wave_sum(s2) / (float)D;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (g[i][e] * dy[i][e] - s1 * xh[i][e] - s2) * rstd;
                uint2 u;
                u.x = pack_bf16x2(o[0], o[1]);
                u.y = pack_bf16x2(o[2], o[3]);
                *reinterpret_cast<uint2*>(dz + (size_t)row * D + (i * 64 + lane) * 4) = u;
                if (CS) {
                    float r[4];
                    unpack4_bf16(u, r);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dc[i][e] += r[e];
                }
            }
        }
        __syncthreads();  // g and red are rewritten for the next sequence
    }
    if (part) {
        store_param_partials<NCH>(dg, db, part, smem);
    } else {
        flush_param_grads<NCH>(dg, db, dgamma, dbeta, smem);
    }
    if constexpr (CS) store_colsum_partials<NCH>(dc, part3, smem);
}

template <int NCH>
__global__ __launch_bounds__(256) void embed_ln_fwd_kernel(const int64_t* __restrict__ ids,
                                                           const int32_t* __restrict__ indices,
                                                           const float* __restrict__ word,
                                                           const float* __restrict__ type0,
                                                           const float* __restrict__ pos_emb,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, bf16_t* __restrict__ out,
                                                           float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                           int T, int S, float eps) {
    constexpr int D = NCH * 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float g[NCH][4], b[NCH][4], ty[NCH][4];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        load4_f32(gamma + (i * 64 + lane) * 4, g[i]);
        load4_f32(beta + (i * 64 + lane) * 4, b[i]);
        load4_f32(type0 + (i * 64 + lane) * 4, ty[i]);
    }
    for (int t = blockIdx.x * 4 + wave; t < T; t += gridDim.x * 4) {
        const int flat = indices[t];
        const long id = ids[flat];
        const int pos = flat % S;
        float z[NCH][4];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int col = (i * 64 + lane) * 4;
            load4_f32(word + (size_t)id * D + col, z[i]);
            // reference order: word + pos + type (sc/layers/embedding.py:601-614)
            if (pos_emb) {
                float pe[4];
                load4_f32(pos_emb + (size_t)pos * D + col, pe);
#pragma unroll
                for (int e = 0; e < 4; ++e) z[i][e] += pe[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) z[i][e] += ty[i][e];
        }
        float mean, rstd;
        row_stats<NCH>(z, D, eps, mean, rstd);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (z[i][e] - mean) * rstd * g[i][e] + b[i][e];
            store4_bf16(out + (size_t)t * D + (i * 64 + lane) * 4, o);
        }
        if (lane == 0) {
            mean_o[t] = mean;
            rstd_o[t] = rstd;
        }
    }
}

template <int NCH>
__global__ __launch_bounds__(256) void embed_ln_bwd_kernel(
    const bf16_t* __restrict__ da, const bf16_t* __restrict__ dbb, const int64_t* __restrict__ ids,
    const int32_t* __restrict__ indices, const float* __restrict__ word, const float* __restrict__ type0,
    const float* __restrict__ pos_emb, const float* __restrict__ gamma, const float* __restrict__ mean_i,
    const float* __restrict__ rstd_i, float* dword, float* dtype0, float* dpos, float* dgamma, float* dbeta, int T,
    int S, int padding_idx, float* __restrict__ dz_out) {
    constexpr int D = NCH * 256;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float g[NCH][4], ty[NCH][4], dg[NCH][4], db[NCH][4], dty[NCH][4];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        load4_f32(gamma + (i * 64 + lane) * 4, g[i]);
        load4_f32(type0 + (i * 64 + lane) * 4, ty[i]);
#pragma unroll
        for (int e = 0; e < 4; ++e) dg[i][e] = db[i][e] = dty[i][e] = 0.f;
    }
    for (int t = blockIdx.x * 4 + wave; t < T; t += gridDim.x * 4) {
        const int flat = indices[t];
        const long id = ids[flat];
        const int pos = flat % S;
        const float mean = mean_i[t], rstd = rstd_i[t];
        float dy[NCH][4], xh[NCH][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int col = (i * 64 + lane) * 4;
            load4_bf16(da + (size_t)t * D + col, dy[i]);
            if (dbb) {
                float tt[4];
                load4_bf16(dbb + (size_t)t * D + col, tt);
#pragma unroll
                for (int e = 0; e < 4; ++e) dy[i][e] += tt[e];
            }
            float zz[4];
            load4_f32(word + (size_t)id * D + col, zz);
            if (pos_emb) {
                float pe[4];
                load4_f32(pos_emb + (size_t)pos * D + col, pe);
#pragma unroll
                for (int e = 0; e < 4; ++e) zz[e] += pe[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                zz[e] += ty[i][e];
                xh[i][e] = (zz[e] - mean) * rstd;
                const float wdy = g[i][e] * dy[i][e];
                s1 += wdy * xh[i][e];
                s2 += wdy;
                dg[i][e] += dy[i][e] * xh[i][e];
                db[i][e] += dy[i][e];
            }
        }
        s1 = wave_sum(s1) / (float)D;
        s2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int col = (i * 64 + lane) * 4;
            float ov[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float o = (g[i][e] * dy[i][e] - s1 * xh[i][e] - s2) * rstd;
                ov[e] = o;
                dty[i][e] += o;
                // nn.Embedding(padding_idx=...) gives that row no gradient (sc/layers/embedding.py:581)
                if (!dz_out && dword && id != padding_idx) unsafeAtomicAdd(dword + (size_t)id * D + col + e, o);
                if (dpos) unsafeAtomicAdd(dpos + (size_t)pos * D + col + e, o);
            }
            // sorted path: the row gradient goes to scratch, embed_scatter_sorted_kernel sums it per vocabulary row
            // fp32 row gradients for the sorted reduction (round 3, ADVICE r2: a bf16 scratch cost every addend 8 bits, and
            // frequent tokens -- [CLS], [SEP] -- sum thousands of them)
            if (dz_out) *reinterpret_cast<float4*>(dz_out + (size_t)t * D + col) = make_float4(ov[0], ov[1], ov[2], ov[3]);
        }
    }
    flush_param_grads<NCH>(dg, db, dgamma, dbeta, smem);
    __syncthreads();
    // type-embedding row 0 receives the sum over every token: reuse the same block fold.
    float zero[NCH][4];
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) zero[i][e] = 0.f;
    flush_param_grads<NCH>(dty, zero, dtype0, nullptr, smem);
}

// Word-embedding gradient without atomics (the scatter of 100 M fp32 atomics cost 1.4 ms per 131072-token chunk and made the
// result depend on arrival order).  The host passes the chunk's token ids sorted (stable) with the permutation; workgroup v
// owns vocabulary row v: it binary-searches its run [lo, hi) in the sorted ids, its four waves sum the rows
// dz[perm[lo + w]], dz[perm[lo + w + 4]], ... (each in token order), the four partials are folded in wave order and the
// row is read-modify-written once, by its only owner.  Bit-reproducible.
template <int NCH>
__global__ __launch_bounds__(256) void embed_scatter_sorted_kernel(const float* __restrict__ dz, const int32_t* __restrict__ sorted_ids,
                                                                   const int32_t* __restrict__ perm, float* __restrict__ dword, int T,
                                                                   int vocab, int padding_idx) {
    constexpr int D = NCH * 256;
    __shared__ float red[4][D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int v = blockIdx.x; v < vocab; v += gridDim.x) {
        if (v == padding_idx) continue;
        // lower bounds of v and v + 1 in sorted_ids (wave-uniform scalar loop)
        int lo = 0, hi = T;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (sorted_ids[mid] < v) lo = mid + 1; else hi = mid;
        }
        int lo2 = lo, hi2 = T;
        while (lo2 < hi2) {
            const int mid = (lo2 + hi2) >> 1;
            if (sorted_ids[mid] <= v) lo2 = mid + 1; else hi2 = mid;
        }
        const int n = lo2 - lo;
        if (n == 0) continue;     // (uniform over the workgroup: no barrier is skipped by part of it)
        float acc[NCH][4];
#pragma unroll
        for (int i = 0; i < NCH; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
        for (int k = wave; k < n; k += 4) {
            const int t = perm[lo + k];
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                float r[4];
                load4_f32(dz + (size_t)t * D + (i * 64 + lane) * 4, r);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][e] += r[e];
            }
        }
        float* row = dword + (size_t)v * D;
        if (n <= 1) {               // the common case for a large vocabulary: one token, one wave, no fold
            if (wave == 0) {
#pragma unroll
                for (int i = 0; i < NCH; ++i) {
                    float4* p4 = reinterpret_cast<float4*>(row + (i * 64 + lane) * 4);
                    float4 cur = *p4;
                    cur.x += acc[i][0]; cur.y += acc[i][1]; cur.z += acc[i][2]; cur.w += acc[i][3];
                    *p4 = cur;
                }
            }
            continue;
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[wave][(i * 64 + lane) * 4 + e] = acc[i][e];
        __syncthreads();
        for (int c = threadIdx.x; c < D; c += 256) row[c] += ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
        __syncthreads();
    }
}

// ---- mixed-dtype forms for the flash_attn.ops.layer_norm surface: on the reference's BERT path the
// embedding LayerNorm sees fp32 in / fp32 out and layer 0's residual is fp32 (`residual_in_fp32`, or simply an fp32
// residual); every operand carries its own dtype flag.  One wave per row, statistics in fp32; not the engine's hot path
// (the engine keeps the bf16 kernels above), so no prefetch pipelining here.
CX_DEVICE void load4_any(const void* base, bool f32, size_t off, float (&v)[4]) {
    if (f32) load4_f32(reinterpret_cast<const float*>(base) + off, v);
    else load4_bf16(reinterpret_cast<const bf16_t*>(base) + off, v);
}
CX_DEVICE void store4_any(void* base, bool f32, size_t off, const float (&v)[4]) {
    if (f32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + off) = make_float4(v[0], v[1], v[2], v[3]);
    else store4_bf16(reinterpret_cast<bf16_t*>(base) + off, v);
}

enum { LNF_X0_F32 = 1, LNF_RES_F32 = 2, LNF_OUT_F32 = 4, LNF_Z_F32 = 8, LNF_RMS = 16 };  // RMS: no mean subtraction (K8)

template <int NCH>
__global__ __launch_bounds__(256) void ln_fwd_mixed_kernel(const void* __restrict__ x0, const void* __restrict__ res,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           void* __restrict__ out, void* z_out, float* __restrict__ mean_o,
                                                           float* __restrict__ rstd_o, int rows, float eps, int flags) {
    constexpr int D = NCH * 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        float z[NCH][4];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const size_t off = (size_t)row * D + (i * 64 + lane) * 4;
            load4_any(x0, flags & LNF_X0_F32, off, z[i]);
            if (res) {
                float r[4];
                load4_any(res, flags & LNF_RES_F32, off, r);
#pragma unroll
                for (int e = 0; e < 4; ++e) z[i][e] += r[e];
            }
        }
        float mean, rstd;
        if (flags & LNF_RMS) {  // RMSNorm (flash_attn.ops.rms_norm): z * rsqrt(mean(z^2) + eps) * gamma (+ beta if given)
            float v = 0.f;
#pragma unroll
            for (int i = 0; i < NCH; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) v += z[i][e] * z[i][e];
            mean = 0.f;
            rstd = rsqrtf(wave_sum(v) / (float)D + eps);
        } else {
            row_stats<NCH>(z, D, eps, mean, rstd);
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const size_t off = (size_t)row * D + (i * 64 + lane) * 4;
            float g[4], b[4] = {0.f, 0.f, 0.f, 0.f}, o[4];
            load4_f32(gamma + (i * 64 + lane) * 4, g);
            if (beta) load4_f32(beta + (i * 64 + lane) * 4, b);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (z[i][e] - mean) * rstd * g[e] + b[e];
            store4_any(out, flags & LNF_OUT_F32, off, o);
            if (z_out) store4_any(z_out, flags & LNF_Z_F32, off, z[i]);  // statistics use the unrounded fp32 sum (as upstream)
        }
        if (lane == 0) {
            mean_o[row] = mean;
            rstd_o[row] = rstd;
        }
    }
}

// dz = LN-backward(dout) (+ dz_extra);  dx0 (dtype of x0) and dres (dtype of the residual, may be NULL) both receive dz.
template <int NCH>
__global__ __launch_bounds__(256) void ln_bwd_mixed_kernel(const void* __restrict__ dout, const void* __restrict__ z,
                                                           const float* __restrict__ gamma, const float* __restrict__ mean_i,
                                                           const float* __restrict__ rstd_i, const void* __restrict__ dz_extra,
                                                           void* __restrict__ dx0, void* __restrict__ dres, float* dgamma,
                                                           float* dbeta, int rows, int flags) {
    constexpr int D = NCH * 256;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float g[NCH][4], dg[NCH][4], db[NCH][4];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        load4_f32(gamma + (i * 64 + lane) * 4, g[i]);
#pragma unroll
        for (int e = 0; e < 4; ++e) dg[i][e] = db[i][e] = 0.f;
    }
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const float mean = mean_i[row], rstd = rstd_i[row];
        float dy[NCH][4], xh[NCH][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const size_t off = (size_t)row * D + (i * 64 + lane) * 4;
            load4_any(dout, flags & LNF_OUT_F32, off, dy[i]);
            float zz[4];
            load4_any(z, flags & LNF_Z_F32, off, zz);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xh[i][e] = (zz[e] - mean) * rstd;
                const float wdy = g[i][e] * dy[i][e];
                s1 += wdy * xh[i][e];
                s2 += wdy;
                dg[i][e] += dy[i][e] * xh[i][e];
                db[i][e] += dy[i][e];
            }
        }
        s1 = wave_sum(s1) / (float)D;
        s2 = (flags & LNF_RMS) ? 0.f : wave_sum(s2) / (float)D;   // RMSNorm has no mean to differentiate through (mean_i = 0)
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const size_t off = (size_t)row * D + (i * 64 + lane) * 4;
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (g[i][e] * dy[i][e] - s1 * xh[i][e] - s2) * rstd;
            if (dz_extra) {
                float t[4];
                load4_any(dz_extra, flags & LNF_Z_F32, off, t);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] += t[e];
            }
            store4_any(dx0, flags & LNF_X0_F32, off, o);
            if (dres) store4_any(dres, flags & LNF_RES_F32, off, o);
        }
    }
    flush_param_grads<NCH>(dg, db, dgamma, dbeta, smem);
}

// ---- dropout > 0 (K5 dropout_add_layer_norm(p > 0), BertEmbeddings dropout): z = dropout_p(x0) + residual.  The mask is
// regenerated from Philox (cx_common.h), never stored.  Backward returns BOTH gradients: dz (the residual's) and
// dx0 = dz * mask / (1 - p) (the sub-layer's).  One wave per row.
template <int NCH>
__global__ __launch_bounds__(256) void ln_fwd_drop_kernel(const bf16_t* __restrict__ x0, const bf16_t* __restrict__ res,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          bf16_t* __restrict__ out, bf16_t* z_out, float* __restrict__ mean_o,
                                                          float* __restrict__ rstd_o, int rows, float eps, CxDropout dr,
                                                          uint32_t site) {
    constexpr int D = NCH * 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        float z[NCH][4];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const size_t off = (size_t)row * D + (i * 64 + lane) * 4;
            float keep[4], r[4] = {0.f, 0.f, 0.f, 0.f};
            load4_bf16(x0 + off, z[i]);
            dropout_keep4(dr, site, off >> 2, keep);
            if (res) load4_bf16(res + off, r);
#pragma unroll
            for (int e = 0; e < 4; ++e) z[i][e] = z[i][e] * keep[e] + r[e];
        }
        float mean, rstd;
        row_stats<NCH>(z, D, eps, mean, rstd);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const size_t off = (size_t)row * D + (i * 64 + lane) * 4;
            float g[4], b[4], o[4];
            load4_f32(gamma + (i * 64 + lane) * 4, g);
            load4_f32(beta + (i * 64 + lane) * 4, b);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (z[i][e] - mean) * rstd * g[e] + b[e];
            store4_bf16(out + off, o);
            if (z_out) store4_bf16(z_out + off, z[i]);
        }
        if (lane == 0) {
            mean_o[row] = mean;
            rstd_o[row] = rstd;
        }
    }
}

// CS (round 6): the column sums of the bf16 dx0 written -- dx0 is the gradient of the sub-layer's OUTPUT (out_proj / fc2 of the block), so
// they are that Linear's bias gradient, as in ln_bwd_kernel<., CS>: the dropout recipes (bert-base-uncased: cfg 1, the LiT / CLIP text towers)
// ran a standalone colsum_kernel launch per bias (36 per step at cfg 1: 4.9 % of it, profiles/r6_kernel_summary_cfg1.txt).
template <int NCH, bool CS = false>
__global__ __launch_bounds__(256) void ln_bwd_drop_kernel(const bf16_t* __restrict__ da, const bf16_t* __restrict__ dbb,
                                                          const bf16_t* __restrict__ z, const float* __restrict__ gamma,
                                                          const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                          bf16_t* __restrict__ dz, bf16_t* __restrict__ dx0, float* dgamma,
                                                          float* dbeta, float* part, float* part3, int rows, CxDropout dr, uint32_t site) {
    constexpr int D = NCH * 256;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float g[NCH][4], dg[NCH][4], db[NCH][4], dc[CS ? NCH : 1][4];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        load4_f32(gamma + (i * 64 + lane) * 4, g[i]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            dg[i][e] = db[i][e] = 0.f;
            if (CS) dc[i][e] = 0.f;
        }
    }
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const float mean = mean_i[row], rstd = rstd_i[row];
        float dy[NCH][4], xh[NCH][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const size_t off = (size_t)row * D + (i * 64 + lane) * 4;
            load4_bf16(da + off, dy[i]);
            if (dbb) {
                float t[4];
                load4_bf16(dbb + off, t);
#pragma unroll
                for (int e = 0; e < 4; ++e) dy[i][e] += t[e];
            }
            float zz[4];
            load4_bf16(z + off, zz);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xh[i][e] = (zz[e] - mean) * rstd;
                const float wdy = g[i][e] * dy[i][e];
                s1 += wdy * xh[i][e];
                s2 += wdy;
                dg[i][e] += dy[i][e] * xh[i][e];
                db[i][e] += dy[i][e];
            }
        }
        s1 = wave_sum(s1) / (float)D;
        s2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const size_t off = (size_t)row * D + (i * 64 + lane) * 4;
            float o[4], keep[4], m[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (g[i][e] * dy[i][e] - s1 * xh[i][e] - s2) * rstd;
            dropout_keep4(dr, site, off >> 2, keep);
#pragma unroll
            for (int e = 0; e < 4; ++e) m[e] = o[e] * keep[e];
            store4_bf16(dz + off, o);
            if constexpr (CS) {
                uint2 u;
                u.x = pack_bf16x2(m[0], m[1]);
                u.y = pack_bf16x2(m[2], m[3]);
                *reinterpret_cast<uint2*>(dx0 + off) = u;
                float r[4];
                unpack4_bf16(u, r);   // (the bias gradient sums what the next kernels read: the bf16 dx0)
#pragma unroll
                for (int e = 0; e < 4; ++e) dc[i][e] += r[e];
            } else {
                store4_bf16(dx0 + off, m);
            }
        }
    }
    if (part) {
        store_param_partials<NCH>(dg, db, part, smem);
    } else {
        flush_param_grads<NCH>(dg, db, dgamma, dbeta, smem);
    }
    if constexpr (CS) store_colsum_partials<NCH>(dc, part3, smem);
}

// x <- x * mask / (1 - p) in place (the embedding dropout on the embedding-LayerNorm output, and on its incoming gradient)
__global__ __launch_bounds__(256) void dropout_scale_kernel(bf16_t* __restrict__ x, long n4, CxDropout dr, uint32_t site) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float v[4], keep[4];
        load4_bf16(x + i * 4, v);
        dropout_keep4(dr, site, (unsigned long long)i, keep);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= keep[e];
        store4_bf16(x + i * 4, v);
    }
}

// backward kernels end with 2*d device-scope atomics per block (they serialise at the memory fabric): one block per CU
inline int ln_grid_bwd(int rows) {
    int g = (rows + 3) / 4;
    if (g > 256) g = 256;
    if (g < 1) g = 1;
    return g;
}
inline int ln_grid(int rows) {
    int g = (rows + 3) / 4;
    if (g > 256 * 8) g = 256 * 8;
    if (g < 1) g = 1;
    return g;
}
inline int done() { return hipGetLastError() == hipSuccess ? CX_OK : CX_ERR_LAUNCH; }

#define CX_LN_DISPATCH(d, CALL)                     \
    switch (d) {                                    \
        case 256: { constexpr int NCH = 1; CALL; break; }  \
        case 512: { constexpr int NCH = 2; CALL; break; }  \
        case 768: { constexpr int NCH = 3; CALL; break; }  \
        case 1024: { constexpr int NCH = 4; CALL; break; } \
        default: return CX_ERR_SHAPE;               \
    }

}  // namespace

extern "C" {

int cx_layernorm_fwd(const uint16_t* x0, const uint16_t* residual, const float* gamma, const float* beta,
                     uint16_t* out, uint16_t* z_out, float* mean, float* rstd, int rows, int d, float eps,
                     void* stream) {
    if (rows <= 0) return CX_OK;
    if (!x0 || !gamma || !beta || !out || !mean || !rstd) return CX_ERR_ARG;
    CX_LN_DISPATCH(d, hipLaunchKernelGGL((ln_fwd_kernel<NCH>), dim3(ln_grid(rows)), dim3(256), 0,
                                         (hipStream_t)stream, x0, residual, gamma, beta, out, z_out, mean, rstd,
                                         rows, eps));
    return done();
}

namespace {
// Shared launcher.  dz_colsum != NULL: fp32[d] += column sums of the dz written (needs the workspace: the sums go through
// the deterministic two-stage reduction like dgamma / dbeta).
int ln_bwd_launch(const uint16_t* dout_a, const uint16_t* dout_b, const uint16_t* z, const float* gamma, const float* mean,
                  const float* rstd, const uint16_t* dz_extra, uint16_t* dz, float* dgamma, float* dbeta, float* dz_colsum,
                  float* ws, long ws_floats, int rows, int d, void* stream) {
    if (rows <= 0) return CX_OK;
    if (!dout_a || !z || !gamma || !mean || !rstd || !dz) return CX_ERR_ARG;
    const size_t smem = (size_t)8 * d * sizeof(float);
    // with a workspace: many blocks (bandwidth) + deterministic two-stage parameter-gradient reduction;
    // without: one block per CU and 2*d atomics per block
    int grid = ln_grid_bwd(rows);
    float* part = nullptr;
    const long per_block = (dz_colsum ? 3L : 2L) * d;
    if (ws && ws_floats >= per_block * 256) {
        long cap = ws_floats / per_block;
        // 3 blocks per CU = the kernel's occupancy (130 VGPRs): one resident round, few partials; small inputs get at least 32
        // rows per block (round 4: 8192 rows used to leave 768 partials of ~10 rows each, and the fold below -- 19 us for 768
        // partials whatever the input -- was 4 % of a literal chunk_size-64 step)
        grid = (rows + 31) / 32;
        if (grid > 768) grid = 768;
        if (grid > cap) grid = (int)cap;
        part = ws;
    }
    if (dz_colsum && !part) return CX_ERR_ARG;
    float* part3 = dz_colsum ? part + (size_t)grid * 2 * d : nullptr;
    if (dz_colsum) {
        CX_LN_DISPATCH(d, hipLaunchKernelGGL((ln_bwd_kernel<NCH, true>), dim3(grid), dim3(256), smem, (hipStream_t)stream, dout_a,
                                             dout_b, z, gamma, mean, rstd, dz_extra, dz, dgamma, dbeta, part, part3, rows));
    } else {
        CX_LN_DISPATCH(d, hipLaunchKernelGGL((ln_bwd_kernel<NCH, false>), dim3(grid), dim3(256), smem, (hipStream_t)stream, dout_a,
                                             dout_b, z, gamma, mean, rstd, dz_extra, dz, dgamma, dbeta, part, part3, rows));
    }
    if (part && (dgamma || dbeta))
        hipLaunchKernelGGL(ln_param_reduce_kernel, dim3((2 * d + 63) / 64), dim3(256), 0, (hipStream_t)stream, part,
                           dgamma, dbeta, grid, d);
    if (dz_colsum)
        hipLaunchKernelGGL(colsum_reduce_kernel, dim3((d + 63) / 64), dim3(256), 0, (hipStream_t)stream, part3, dz_colsum, grid, d);
    return done();
}
}  // namespace

int cx_layernorm_bwd(const uint16_t* dout_a, const uint16_t* dout_b, const uint16_t* z, const float* gamma,
                     const float* mean, const float* rstd, const uint16_t* dz_extra, uint16_t* dz, float* dgamma,
                     float* dbeta, float* ws, long ws_floats, int rows, int d, void* stream) {
    return ln_bwd_launch(dout_a, dout_b, z, gamma, mean, rstd, dz_extra, dz, dgamma, dbeta, nullptr, ws, ws_floats, rows, d, stream);
}

int cx_layernorm_bwd_colsum(const uint16_t* dout_a, const uint16_t* dout_b, const uint16_t* z, const float* gamma,
                            const float* mean, const float* rstd, const uint16_t* dz_extra, uint16_t* dz, float* dgamma,
                            float* dbeta, float* dz_colsum, float* ws, long ws_floats, int rows, int d, void* stream) {
    return ln_bwd_launch(dout_a, dout_b, z, gamma, mean, rstd, dz_extra, dz, dgamma, dbeta, dz_colsum, ws, ws_floats, rows, d, stream);
}

int cx_layernorm_bwd_pooled(const float* demb, const float* emb, const float* norm, const int32_t* cu_seqlens, int B,
                            int pool_mode, int normalize, const uint16_t* z, const float* gamma, const float* mean,
                            const float* rstd, uint16_t* dz, float* dgamma, float* dbeta, float* dz_colsum, float* ws,
                            long ws_floats, int rows, int d, void* stream) {
    if (rows <= 0 || B <= 0) return CX_OK;
    if (!demb || !emb || !norm || !cu_seqlens || !z || !gamma || !mean || !rstd || !dz) return CX_ERR_ARG;
    const size_t smem = ((size_t)8 * d + 8) * sizeof(float);
    int grid = B < 256 ? B : 256;
    float* part = nullptr;
    const long per_block = (dz_colsum ? 3L : 2L) * d;
    if (ws && ws_floats >= per_block * 256) {
        const long cap = ws_floats / per_block;
        grid = B < 768 ? B : 768;
        if (grid > cap) grid = (int)cap;
        part = ws;
    }
    if (dz_colsum && !part) return CX_ERR_ARG;
    float* part3 = dz_colsum ? part + (size_t)grid * 2 * d : nullptr;
    if (dz_colsum) {
        CX_LN_DISPATCH(d, hipLaunchKernelGGL((ln_bwd_pooled_kernel<NCH, true>), dim3(grid), dim3(256), smem, (hipStream_t)stream,
                                             demb, emb, norm, cu_seqlens, B, pool_mode, normalize, z, gamma, mean, rstd, dz, dgamma,
                                             dbeta, part, part3));
    } else {
        CX_LN_DISPATCH(d, hipLaunchKernelGGL((ln_bwd_pooled_kernel<NCH, false>), dim3(grid), dim3(256), smem, (hipStream_t)stream,
                                             demb, emb, norm, cu_seqlens, B, pool_mode, normalize, z, gamma, mean, rstd, dz, dgamma,
                                             dbeta, part, part3));
    }
    if (part && (dgamma || dbeta))
        hipLaunchKernelGGL(ln_param_reduce_kernel, dim3((2 * d + 63) / 64), dim3(256), 0, (hipStream_t)stream, part,
                           dgamma, dbeta, grid, d);
    if (dz_colsum)
        hipLaunchKernelGGL(colsum_reduce_kernel, dim3((d + 63) / 64), dim3(256), 0, (hipStream_t)stream, part3, dz_colsum, grid, d);
    return done();
}

int cx_dropout_add_layernorm_fwd(const uint16_t* x0, const uint16_t* residual, const float* gamma, const float* beta,
                                 uint16_t* out, uint16_t* z_out, float* mean, float* rstd, int rows, int d, float eps, float p,
                                 unsigned long long seed, unsigned long long offset, unsigned int site, void* stream) {
    if (rows <= 0) return CX_OK;
    if (!x0 || !gamma || !beta || !out || !mean || !rstd) return CX_ERR_ARG;
    if (!(p > 0.f) || p >= 1.f) return CX_ERR_ARG;
    const CxDropout dr{p, seed, offset};
    CX_LN_DISPATCH(d, hipLaunchKernelGGL((ln_fwd_drop_kernel<NCH>), dim3(ln_grid(rows)), dim3(256), 0, (hipStream_t)stream, x0,
                                         residual, gamma, beta, out, z_out, mean, rstd, rows, eps, dr, site));
    return done();
}

int cx_dropout_add_layernorm_bwd(const uint16_t* dout_a, const uint16_t* dout_b, const uint16_t* z, const float* gamma,
                                 const float* mean, const float* rstd, uint16_t* dz, uint16_t* dx0, float* dgamma, float* dbeta,
                                 float* ws, long ws_floats, int rows, int d, float p, unsigned long long seed,
                                 unsigned long long offset, unsigned int site, void* stream) {
    return cx_dropout_add_layernorm_bwd_colsum(dout_a, dout_b, z, gamma, mean, rstd, dz, dx0, dgamma, dbeta, nullptr, ws, ws_floats, rows, d,
                                               p, seed, offset, site, stream);
}

// The same with dx0_colsum (may be NULL): fp32[d] += column sums of the bf16 dx0 written = the bias gradient of the Linear whose (dropped)
// output the LayerNorm consumed; needs the workspace (>= 3 * d * 256 floats), CX_ERR_ARG without it.
int cx_dropout_add_layernorm_bwd_colsum(const uint16_t* dout_a, const uint16_t* dout_b, const uint16_t* z, const float* gamma,
                                        const float* mean, const float* rstd, uint16_t* dz, uint16_t* dx0, float* dgamma, float* dbeta,
                                        float* dx0_colsum, float* ws, long ws_floats, int rows, int d, float p, unsigned long long seed,
                                        unsigned long long offset, unsigned int site, void* stream) {
    if (rows <= 0) return CX_OK;
    if (!dout_a || !z || !gamma || !mean || !rstd || !dz || !dx0) return CX_ERR_ARG;
    if (!(p > 0.f) || p >= 1.f) return CX_ERR_ARG;
    const CxDropout dr{p, seed, offset};
    const size_t smem = (size_t)8 * d * sizeof(float);
    int grid = ln_grid_bwd(rows);
    float* part = nullptr;
    const long per_block = (dx0_colsum ? 3L : 2L) * d;
    if (ws && ws_floats >= per_block * 256) {
        long cap = ws_floats / per_block;
        grid = (rows + 31) / 32;   // (as in ln_bwd_launch)
        if (grid > 768) grid = 768;
        if (grid > cap) grid = (int)cap;
        part = ws;
    }
    if (dx0_colsum && !part) return CX_ERR_ARG;
    float* part3 = dx0_colsum ? part + (size_t)grid * 2 * d : nullptr;
    if (dx0_colsum) {
        CX_LN_DISPATCH(d, hipLaunchKernelGGL((ln_bwd_drop_kernel<NCH, true>), dim3(grid), dim3(256), smem, (hipStream_t)stream, dout_a,
                                             dout_b, z, gamma, mean, rstd, dz, dx0, dgamma, dbeta, part, part3, rows, dr, site));
    } else {
        CX_LN_DISPATCH(d, hipLaunchKernelGGL((ln_bwd_drop_kernel<NCH, false>), dim3(grid), dim3(256), smem, (hipStream_t)stream, dout_a,
                                             dout_b, z, gamma, mean, rstd, dz, dx0, dgamma, dbeta, part, part3, rows, dr, site));
    }
    if (part && (dgamma || dbeta))
        hipLaunchKernelGGL(ln_param_reduce_kernel, dim3((2 * d + 63) / 64), dim3(256), 0, (hipStream_t)stream, part, dgamma, dbeta,
                           grid, d);
    if (dx0_colsum)
        hipLaunchKernelGGL(colsum_reduce_kernel, dim3((d + 63) / 64), dim3(256), 0, (hipStream_t)stream, part3, dx0_colsum, grid, d);
    return done();
}

int cx_dropout_scale(uint16_t* x, long n, float p, unsigned long long seed, unsigned long long offset, unsigned int site,
                     void* stream) {
    if (n <= 0) return CX_OK;
    if (!x || (n % 4) != 0) return CX_ERR_ARG;
    if (!(p > 0.f) || p >= 1.f) return CX_ERR_ARG;
    long g = (n / 4 + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(dropout_scale_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, x, n / 4, CxDropout{p, seed, offset}, site);
    return done();
}

int cx_layernorm_fwd_mixed(const void* x0, const void* residual, const float* gamma, const float* beta, void* out, void* z_out,
                           float* mean, float* rstd, int rows, int d, float eps, int flags, void* stream) {
    if (rows <= 0) return CX_OK;
    if (!x0 || !gamma || (!beta && !(flags & LNF_RMS)) || !out || !mean || !rstd) return CX_ERR_ARG;
    CX_LN_DISPATCH(d, hipLaunchKernelGGL((ln_fwd_mixed_kernel<NCH>), dim3(ln_grid(rows)), dim3(256), 0, (hipStream_t)stream, x0,
                                         residual, gamma, beta, out, z_out, mean, rstd, rows, eps, flags));
    return done();
}

int cx_layernorm_bwd_mixed(const void* dout, const void* z, const float* gamma, const float* mean, const float* rstd,
                           const void* dz_extra, void* dx0, void* dres, float* dgamma, float* dbeta, int rows, int d, int flags,
                           void* stream) {
    if (rows <= 0) return CX_OK;
    if (!dout || !z || !gamma || !mean || !rstd || !dx0) return CX_ERR_ARG;
    const size_t smem = (size_t)8 * d * sizeof(float);
    CX_LN_DISPATCH(d, hipLaunchKernelGGL((ln_bwd_mixed_kernel<NCH>), dim3(ln_grid_bwd(rows)), dim3(256), smem, (hipStream_t)stream,
                                         dout, z, gamma, mean, rstd, dz_extra, dx0, dres, dgamma, dbeta, rows, flags));
    return done();
}

int cx_embed_ln_fwd(const int64_t* input_ids, const int32_t* indices, const float* word, const float* type0,
                    const float* pos_emb, const float* gamma, const float* beta, uint16_t* out, float* mean,
                    float* rstd, int T, int S, int d, float eps, void* stream) {
    if (T <= 0) return CX_OK;
    if (!input_ids || !indices || !word || !type0 || !gamma || !beta || !out) return CX_ERR_ARG;
    CX_LN_DISPATCH(d, hipLaunchKernelGGL((embed_ln_fwd_kernel<NCH>), dim3(ln_grid(T)), dim3(256), 0,
                                         (hipStream_t)stream, input_ids, indices, word, type0, pos_emb, gamma, beta,
                                         out, mean, rstd, T, S, eps));
    return done();
}

int cx_embed_ln_bwd(const uint16_t* dout_a, const uint16_t* dout_b, const int64_t* input_ids,
                    const int32_t* indices, const float* word, const float* type0, const float* pos_emb,
                    const float* gamma, const float* mean, const float* rstd, float* dword, float* dtype0,
                    float* dpos, float* dgamma, float* dbeta, int T, int S, int d, int padding_idx, void* stream) {
    if (T <= 0) return CX_OK;
    if (!dout_a || !input_ids || !indices || !word || !type0 || !gamma || !mean || !rstd) return CX_ERR_ARG;
    const size_t smem = (size_t)8 * d * sizeof(float);
    CX_LN_DISPATCH(d, hipLaunchKernelGGL((embed_ln_bwd_kernel<NCH>), dim3(ln_grid_bwd(T)), dim3(256), smem,
                                         (hipStream_t)stream, dout_a, dout_b, input_ids, indices, word, type0,
                                         pos_emb, gamma, mean, rstd, dword, dtype0, dpos, dgamma, dbeta, T, S,
                                         padding_idx, (float*)nullptr));
    return done();
}

int cx_embed_ln_bwd_sorted(const uint16_t* dout_a, const uint16_t* dout_b, const int64_t* input_ids,
                           const int32_t* indices, const float* word, const float* type0, const float* pos_emb,
                           const float* gamma, const float* mean, const float* rstd, float* dword, float* dtype0,
                           float* dpos, float* dgamma, float* dbeta, int T, int S, int d, int padding_idx, int vocab,
                           const int32_t* sorted_ids, const int32_t* perm, float* dz_scratch, void* stream) {
    if (T <= 0) return CX_OK;
    if (!dout_a || !input_ids || !indices || !word || !type0 || !gamma || !mean || !rstd) return CX_ERR_ARG;
    if (!sorted_ids || !perm || !dz_scratch || vocab <= 0) return CX_ERR_ARG;
    const size_t smem = (size_t)8 * d * sizeof(float);
    // many blocks here: without the word-row atomics this kernel is a plain streaming LayerNorm backward
    int grid = (T + 3) / 4;
    if (grid > 1024) grid = 1024;
    CX_LN_DISPATCH(d, hipLaunchKernelGGL((embed_ln_bwd_kernel<NCH>), dim3(grid), dim3(256), smem, (hipStream_t)stream, dout_a,
                                         dout_b, input_ids, indices, word, type0, pos_emb, gamma, mean, rstd, dword, dtype0,
                                         dpos, dgamma, dbeta, T, S, padding_idx, dz_scratch));
    if (dword) {
        int sg = vocab < 8192 ? vocab : 8192;
        CX_LN_DISPATCH(d, hipLaunchKernelGGL((embed_scatter_sorted_kernel<NCH>), dim3(sg), dim3(256), 0, (hipStream_t)stream,
                                             dz_scratch, sorted_ids, perm, dword, T, vocab, padding_idx));
    }
    return done();
}

}  // extern "C"
