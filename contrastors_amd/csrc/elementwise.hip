// elementwise.hip -- the HBM-bound glue ops of the hot path: transposes and
// casts feeding the MFMA GEMMs, SwiGLU / GELU, bias gradients, mean/cls pooling + L2 normalisation, standalone
// rotary.  All loads/stores are 16 B per lane where the layout allows (guide G13).
#include "cx_common.h"
#include "../../include/contrastors_hip.h"

namespace {

constexpr int EW_BLOCK = 256;

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

// ------------------------------------------------------------------------------------------ transposes
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out,
                                                             int rows, int cols, int ld_in, int ld_out,
                                                             int rows_pad) {
    __shared__ bf16_t tile[64][66];
    const int tid = threadIdx.x;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int r = p * 32 + (tid >> 3), ch = tid & 7;
        const int gr = r0 + r, gc = c0 + ch * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (gr < rows && gc < cols) v = *reinterpret_cast<const uint4*>(in + (size_t)gr * ld_in + gc);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&tile[r][ch * 8]);
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int c = p * 32 + (tid >> 3), rch = tid & 7;
        const int gc = c0 + c, gr = r0 + rch * 8;
        if (gc < cols && gr < rows_pad) {
            uint32_t w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                w[e] = (uint32_t)tile[rch * 8 + 2 * e][c] | ((uint32_t)tile[rch * 8 + 2 * e + 1][c] << 16);
            *reinterpret_cast<uint4*>(out + (size_t)gc * ld_out + gr) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
}

__global__ __launch_bounds__(256) void cast_transpose_f32_bf16_kernel(const float* __restrict__ in,
                                                                      bf16_t* __restrict__ out, int rows, int cols) {
    __shared__ float tile[64][65];
    const int tid = threadIdx.x;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    for (int i = tid; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        const int gr = r0 + r, gc = c0 + c;
        tile[r][c] = (gr < rows && gc < cols) ? in[(size_t)gr * cols + gc] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < 64 * 64; i += 256) {
        const int c = i >> 6, r = i & 63;
        const int gr = r0 + r, gc = c0 + c;
        if (gr < rows && gc < cols) out[(size_t)gc * rows + gr] = f32_to_bf16(tile[r][c]);
    }
}

// blockIdx.y = matrix, blockIdx.x = 64 x 64 tile of it (blocks past a matrix's tile count leave at once)
__global__ __launch_bounds__(256) void cast_transpose_batched_kernel(const CxCastJob* __restrict__ jobs) {
    __shared__ float tile[64][65];
    const CxCastJob j = jobs[blockIdx.y];
    const int tiles_c = (j.cols + 63) / 64, tiles_r = (j.rows + 63) / 64;
    if ((int)blockIdx.x >= tiles_c * tiles_r) return;
    const int tid = threadIdx.x;
    const int r0 = (blockIdx.x / tiles_c) * 64, c0 = (blockIdx.x % tiles_c) * 64;
    for (int i = tid; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        const int gr = r0 + r, gc = c0 + c;
        tile[r][c] = (gr < j.rows && gc < j.cols) ? j.in[(size_t)gr * j.cols + gc] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < 64 * 64; i += 256) {
        const int c = i >> 6, r = i & 63;
        const int gr = r0 + r, gc = c0 + c;
        if (gr < j.rows && gc < j.cols) j.out_t[(size_t)gc * j.rows + gr] = f32_to_bf16(tile[r][c]);
    }
}

__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                            int rows, int cols, int ld_in, int ld_out) {
    __shared__ float tile[64][65];
    const int tid = threadIdx.x;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    for (int i = tid; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        const int gr = r0 + r, gc = c0 + c;
        tile[r][c] = (gr < rows && gc < cols) ? in[(size_t)gr * ld_in + gc] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < 64 * 64; i += 256) {
        const int c = i >> 6, r = i & 63;
        const int gr = r0 + r, gc = c0 + c;
        if (gr < rows && gc < cols) out[(size_t)gc * ld_out + gr] = tile[r][c];
    }
}

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out,
                                                            long n) {
    const long stride = (long)gridDim.x * blockDim.x * 4;
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
            const float4 v = *reinterpret_cast<const float4*>(in + i);
            uint2 pk;
            pk.x = pack_bf16x2(v.x, v.y);
            pk.y = pack_bf16x2(v.z, v.w);
            *reinterpret_cast<uint2*>(out + i) = pk;
        } else {
            for (long j = i; j < n; ++j) out[j] = f32_to_bf16(in[j]);
        }
    }
}

__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const bf16_t* __restrict__ in, float* __restrict__ out,
                                                            long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = bf16_to_f32(in[i]);
}

// ------------------------------------------------------------------------------------------ activations
// v_exp_f32 + v_rcp_f32 (1 ulp each): an IEEE division costs ~10 VALU ops per element, which made the SwiGLU backward
// kernel VALU-bound before it was HBM-bound
CX_DEVICE float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }

// column of y / gate for activation column c: concatenated [y | gate] (layout 0) or interleaved in groups of 32
// ([y 0..31 | gate 0..31 | y 32..63 | ...], layout 1 = what the fused GEMM epilogue and its weight use)
CX_DEVICE int ycol(int c, int I, int layout) { return layout ? ((c >> 5) << 6) + (c & 31) : c; }
CX_DEVICE int gcol(int c, int I, int layout) { return layout ? ((c >> 5) << 6) + 32 + (c & 31) : I + c; }

__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const bf16_t* __restrict__ yg, bf16_t* __restrict__ act,
                                                         long T, int I, int layout) {
    const int chunks = I >> 3;
    const long total = T * chunks;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long t = i / chunks;
        const int c = (int)(i - t * chunks) * 8;
        const bf16_t* row = yg + t * (2L * I);
        float y[8], g[8], o[8];
        unpack8(*reinterpret_cast<const uint4*>(row + ycol(c, I, layout)), y);
        unpack8(*reinterpret_cast<const uint4*>(row + gcol(c, I, layout)), g);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = g[e] * sigmoidf_(g[e]) * y[e];
        *reinterpret_cast<uint4*>(act + t * (long)I + c) = pack8(o);
    }
}

__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const bf16_t* __restrict__ dact,
                                                         const bf16_t* __restrict__ yg, bf16_t* __restrict__ dyg,
                                                         long T, int I, int layout) {
    // flat grid-stride over (row, 8-column chunk): consecutive lanes stream consecutive 16-B chunks (a 2-D row-strided
    // mapping measured 8-14 % slower at the same byte count)
    const int chunks = I >> 3;
    const long total = T * chunks;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long t = i / chunks;
        const int c = (int)(i - t * chunks) * 8;
        const bf16_t* row = yg + t * (2L * I);
        float y[8], g[8], d[8], dy[8], dg[8];
        unpack8(*reinterpret_cast<const uint4*>(row + ycol(c, I, layout)), y);
        unpack8(*reinterpret_cast<const uint4*>(row + gcol(c, I, layout)), g);
        unpack8(*reinterpret_cast<const uint4*>(dact + t * (long)I + c), d);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float s = sigmoidf_(g[e]);
            const float gs = g[e] * s;
            dy[e] = gs * d[e];
            dg[e] = (s + gs * (1.f - s)) * d[e] * y[e];
        }
        bf16_t* orow = dyg + t * (2L * I);
        *reinterpret_cast<uint4*>(orow + ycol(c, I, layout)) = pack8(dy);
        *reinterpret_cast<uint4*>(orow + gcol(c, I, layout)) = pack8(dg);
    }
}

// SwiGLU backward from (act, gate): the forward saved the gate alone next to the activation (cx_gemm_bf16_swiglu_gate);
// y is recovered inside the derivative, d gate = d * y * silu'(g) with y = act / silu(g) = d * act * (1 / g + 1 - sigmoid(g)).
// dyg: (T, 2I) in the interleaved-by-32 layout the fc1 dgrad / wgrad GEMMs consume.
__global__ __launch_bounds__(256) void swiglu_bwd_gate_kernel(const bf16_t* __restrict__ dact, const bf16_t* __restrict__ act,
                                                              const bf16_t* __restrict__ gate, bf16_t* __restrict__ dyg,
                                                              long T, int I) {
    const int chunks = I >> 3;
    const long total = T * chunks;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long t = i / chunks;
        const int c = (int)(i - t * chunks) * 8;
        float a[8], g[8], d[8], dy[8], dg[8];
        unpack8(*reinterpret_cast<const uint4*>(act + t * (long)I + c), a);
        unpack8(*reinterpret_cast<const uint4*>(gate + t * (long)I + c), g);
        unpack8(*reinterpret_cast<const uint4*>(dact + t * (long)I + c), d);
#pragma unroll
        for (int e = 0; e < 8; ++e) swiglu_bwd_from_act(d[e], a[e], g[e], dy[e], dg[e]);   // (g = 0 <=> act = 0 -> d gate = 0)
        bf16_t* orow = dyg + t * (2L * I);
        *reinterpret_cast<uint4*>(orow + ycol(c, I, 1)) = pack8(dy);
        *reinterpret_cast<uint4*>(orow + gcol(c, I, 1)) = pack8(dg);
    }
}

template <int kind>
__global__ __launch_bounds__(256) void bias_gelu_fwd_kernel(const bf16_t* __restrict__ pre,
                                                            const float* __restrict__ bias,
                                                            bf16_t* __restrict__ act, long T, int I) {
    const int chunks = I >> 3;
    const long total = T * chunks;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long t = i / chunks;
        const int c = (int)(i - t * chunks) * 8;
        float x[8], o[8];
        unpack8(*reinterpret_cast<const uint4*>(pre + t * (long)I + c), x);
        float bb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (bias) {  // two 16-B loads, not eight scalar ones (the kernel was VMEM-issue-bound on them)
            const float4 b0 = *reinterpret_cast<const float4*>(bias + c), b1 = *reinterpret_cast<const float4*>(bias + c + 4);
            bb[0] = b0.x; bb[1] = b0.y; bb[2] = b0.z; bb[3] = b0.w; bb[4] = b1.x; bb[5] = b1.y; bb[6] = b1.z; bb[7] = b1.w;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o[e] = act_val(x[e] + bb[e], kind);
        }
        *reinterpret_cast<uint4*>(act + t * (long)I + c) = pack8(o);
    }
}

__global__ __launch_bounds__(256) void bias_gelu_bwd_kernel(const bf16_t* __restrict__ dact,
                                                            const bf16_t* __restrict__ pre,
                                                            const float* __restrict__ bias,
                                                            bf16_t* __restrict__ dpre, long T, int I) {
    const int chunks = I >> 3;
    const long total = T * chunks;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long t = i / chunks;
        const int c = (int)(i - t * chunks) * 8;
        float x[8], d[8], o[8];
        unpack8(*reinterpret_cast<const uint4*>(pre + t * (long)I + c), x);
        unpack8(*reinterpret_cast<const uint4*>(dact + t * (long)I + c), d);
        float bb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (bias) {
            const float4 b0 = *reinterpret_cast<const float4*>(bias + c), b1 = *reinterpret_cast<const float4*>(bias + c + 4);
            bb[0] = b0.x; bb[1] = b0.y; bb[2] = b0.z; bb[3] = b0.w; bb[4] = b1.x; bb[5] = b1.y; bb[6] = b1.z; bb[7] = b1.w;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = x[e] + bb[e];
            float gauss;
            const float cdf = gelu_cdf(v, gauss);
            o[e] = d[e] * (cdf + v * 0.3989422804014327f * gauss);
        }
        *reinterpret_cast<uint4*>(dpre + t * (long)I + c) = pack8(o);
    }
}

// dbias[n] += sum_t dY[t][n].  Block = 32 column-chunks (256 columns) x 8 row lanes; every thread keeps four 16-byte loads
// in flight (round 3: one load per iteration and at most 192 blocks left this HBM-bound reduction at 1.2-4 TB/s -- 6.5 % of
// the CLIP step, profiles/r3_kernel_summary_clip_before_bias_fusion.txt).  GELU = true (cx_bias_gelu_bwd_colsum) is the
// backward of bias + erf-GELU with the bias gradient of the SAME pass: dpre = dact * gelu'(pre + bias) is written and its
// column sums accumulated, instead of a second kernel reading dpre back.
template <bool GELU, int kind = 0>
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ dY, const bf16_t* __restrict__ pre,
                                                     const float* __restrict__ bias, bf16_t* __restrict__ dpre,
                                                     float* __restrict__ dbias, int T, int N, int ld) {
    __shared__ float red[8][256];
    const int tid = threadIdx.x;
    const int cch = tid & 31, rl = tid >> 5;
    const int col = blockIdx.x * 256 + cch * 8;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (col < N) {
        float bb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (GELU && bias) {
            const float4 b0 = *reinterpret_cast<const float4*>(bias + col), b1 = *reinterpret_cast<const float4*>(bias + col + 4);
            bb[0] = b0.x; bb[1] = b0.y; bb[2] = b0.z; bb[3] = b0.w; bb[4] = b1.x; bb[5] = b1.y; bb[6] = b1.z; bb[7] = b1.w;
        }
        const int step = gridDim.y * 8;
        constexpr int U = GELU ? 2 : 4;
        for (int t0 = blockIdx.y * 8 + rl; t0 < T; t0 += U * step) {
            uint4 a[U], x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = t0 + u * step;
                const int tc = t < T ? t : T - 1;
                a[u] = *reinterpret_cast<const uint4*>(dY + (size_t)tc * ld + col);
                if (GELU) x[u] = *reinterpret_cast<const uint4*>(pre + (size_t)tc * ld + col);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = t0 + u * step;
                if (t >= T) break;
                float v[8];
                unpack8(a[u], v);
                if (GELU) {
                    float xv[8], o[8];
                    unpack8(x[u], xv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        o[e] = v[e] * act_grad(xv[e] + bb[e], kind);
                    }
                    const uint4 pk = pack8(o);
                    *reinterpret_cast<uint4*>(dpre + (size_t)t * ld + col) = pk;
                    unpack8(pk, v);   // the bias gradient sums what the next kernels read: the bf16 dpre
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] += v[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rl][cch * 8 + e] = s[e];
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) tot += red[r][tid];
    const int c = blockIdx.x * 256 + tid;
    if (c < N && dbias) unsafeAtomicAdd(dbias + c, tot);
}

// ------------------------------------------------------------------------------------------ pooling
// One block per sequence.  thread -> (8-column chunk, row group).
__global__ __launch_bounds__(256) void pool_normalize_fwd_kernel(const bf16_t* __restrict__ h,
                                                                 const int32_t* __restrict__ cu,
                                                                 float* __restrict__ emb, float* __restrict__ norm,
                                                                 int d, int mode, int normalize) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // [ngroups][d] + 256
    const int b = blockIdx.x, tid = threadIdx.x;
    const int t0 = cu[b], t1 = cu[b + 1];
    const int len = t1 - t0;
    const int chunks = d >> 3;
    const int ngroups = 256 / chunks > 0 ? 256 / chunks : 1;
    const int ch = tid % chunks, grp = tid / chunks;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (grp < ngroups && tid < chunks * ngroups) {
        const int rend = (mode == 1) ? (len > 0 ? t0 + 1 : t0) : t1;
        for (int t = t0 + grp; t < rend; t += ngroups) {
            float v[8];
            unpack8(*reinterpret_cast<const uint4*>(h + (size_t)t * d + ch * 8), v);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += v[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) sm[grp * d + ch * 8 + e] = s[e];
    }
    __syncthreads();
    float* red = sm + ngroups * d;
    // reference divides by the mask sum with no clamp (modeling_biencoder.py:88-90): len==0 -> NaN, as there.
    const float inv_len = (mode == 1) ? 1.f : 1.f / (float)len;
    float ssq = 0.f;
    for (int c = tid; c < d; c += 256) {
        float v = 0.f;
        for (int g = 0; g < ngroups; ++g) v += sm[g * d + c];
        v *= inv_len;
        sm[c] = v;  // group 0 slot now holds the pooled vector (each c touched by exactly one thread)
        ssq += v * v;
    }
    ssq = wave_sum(ssq);
    if ((tid & 63) == 0) red[tid >> 6] = ssq;
    __syncthreads();
    const float nrm = sqrtf(red[0] + red[1] + red[2] + red[3]);
    const float denom = fmaxf(nrm, 1e-12f);  // F.normalize eps
    if (tid == 0) norm[b] = nrm;
    for (int c = tid; c < d; c += 256) emb[(size_t)b * d + c] = normalize ? sm[c] / denom : sm[c];
}

__global__ __launch_bounds__(256) void pool_normalize_bwd_kernel(const float* __restrict__ demb,
                                                                 const float* __restrict__ emb,
                                                                 const float* __restrict__ norm,
                                                                 const int32_t* __restrict__ cu,
                                                                 bf16_t* __restrict__ dh, int d, int mode,
                                                                 int normalize) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // [d] + 4
    const int b = blockIdx.x, tid = threadIdx.x;
    const int t0 = cu[b], t1 = cu[b + 1];
    const int len = t1 - t0;
    float* red = sm + d;
    float dot = 0.f;
    if (normalize) {
        for (int c = tid; c < d; c += 256) dot += demb[(size_t)b * d + c] * emb[(size_t)b * d + c];
        dot = wave_sum(dot);
        if ((tid & 63) == 0) red[tid >> 6] = dot;
        __syncthreads();
        dot = red[0] + red[1] + red[2] + red[3];
    }
    const float inv_n = normalize ? 1.f / fmaxf(norm[b], 1e-12f) : 1.f;
    const float inv_len = (mode == 1) ? 1.f : 1.f / (float)len;
    for (int c = tid; c < d; c += 256) {
        const float g = demb[(size_t)b * d + c];
        const float dx = normalize ? (g - emb[(size_t)b * d + c] * dot) * inv_n : g;
        sm[c] = dx * inv_len;
    }
    __syncthreads();
    const int chunks = d >> 3;
    const int total = len * chunks;
    for (int i = tid; i < total; i += 256) {
        const int r = i / chunks, ch = i - r * chunks;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (mode == 1 && r != 0) ? 0.f : sm[ch * 8 + e];
        *reinterpret_cast<uint4*>(dh + (size_t)(t0 + r) * d + ch * 8) = pack8(v);
    }
}

// ------------------------------------------------------------------------------------------ rotary
// In-place non-interleaved rotary on `nwhich` (T,H,64) slices of a token-major tensor (q and k of a packed qkv: nwhich=2,
// which_stride=H*64, tok_stride=3*H*64; a standalone (T,H,64) tensor: nwhich=1).  One thread per (token, slice, head,
// chunk pair).  Position of token t = t - cu_seqlens[seq(t)] (sc/layers/embedding.py:685-706).
__global__ __launch_bounds__(256) void rotary_kernel(bf16_t* __restrict__ x, long tok_stride, long which_stride,
                                                     int nwhich, const int32_t* __restrict__ cu,
                                                     const float* __restrict__ cosv, const float* __restrict__ sinv,
                                                     int H, float sign) {
    const int b = blockIdx.y;
    const int t0 = cu[b], len = cu[b + 1] - t0;
    const long per_tok = (long)nwhich * H * 4;
    const long total = (long)len * per_tok;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int pos = (int)(i / per_tok);
        int rem = (int)(i - pos * per_tok);
        const int which = rem / (H * 4);
        rem -= which * H * 4;
        const int hh = rem >> 2, ch = rem & 3;
        bf16_t* base = x + (size_t)(t0 + pos) * tok_stride + (size_t)which * which_stride + hh * 64;
        float x1[8], x2[8], o1[8], o2[8];
        unpack8(*reinterpret_cast<const uint4*>(base + ch * 8), x1);
        unpack8(*reinterpret_cast<const uint4*>(base + 32 + ch * 8), x2);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float c = cosv[pos * 32 + ch * 8 + e], sn = sign * sinv[pos * 32 + ch * 8 + e];
            o1[e] = x1[e] * c - x2[e] * sn;
            o2[e] = x2[e] * c + x1[e] * sn;
        }
        *reinterpret_cast<uint4*>(base + ch * 8) = pack8(o1);
        *reinterpret_cast<uint4*>(base + 32 + ch * 8) = pack8(o2);
    }
}

inline int grid_for(long total_threads) {
    long g = (total_threads + EW_BLOCK - 1) / EW_BLOCK;
    if (g > 256 * 8) g = 256 * 8;  // 256 CUs x 8 blocks, grid-stride the rest (guide G11)
    if (g < 1) g = 1;
    return (int)g;
}
inline int done(hipError_t e = hipGetLastError()) { return e == hipSuccess ? CX_OK : CX_ERR_LAUNCH; }

}  // namespace

extern "C" {

int cx_transpose_bf16(const uint16_t* In, uint16_t* Out, int rows, int cols, int ld_in, int ld_out, int rows_pad,
                      void* stream) {
    if (rows_pad <= 0 || cols <= 0) return CX_OK;
    if ((cols % 8) || (ld_in % 8) || (ld_out % 8) || (rows_pad % 8) || rows_pad < rows) return CX_ERR_ALIGN;
    dim3 grid((cols + 63) / 64, (rows_pad + 63) / 64);
    hipLaunchKernelGGL(transpose_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, In, Out, rows, cols, ld_in,
                       ld_out, rows_pad);
    return done();
}

int cx_cast_f32_to_bf16(const float* In, uint16_t* Out, long n, void* stream) {
    if (n <= 0) return CX_OK;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for((n + 3) / 4)), dim3(EW_BLOCK), 0, (hipStream_t)stream, In,
                       Out, n);
    return done();
}

int cx_cast_bf16_to_f32(const uint16_t* In, float* Out, long n, void* stream) {
    if (n <= 0) return CX_OK;
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid_for(n)), dim3(EW_BLOCK), 0, (hipStream_t)stream, In, Out, n);
    return done();
}

int cx_cast_transpose_f32_to_bf16(const float* In, uint16_t* OutT, int rows, int cols, void* stream) {
    if (rows <= 0 || cols <= 0) return CX_OK;
    dim3 grid((cols + 63) / 64, (rows + 63) / 64);
    hipLaunchKernelGGL(cast_transpose_f32_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, In, OutT, rows, cols);
    return done();
}

int cx_cast_transpose_f32_to_bf16_batched(const CxCastJob* jobs, int n_jobs, int max_tiles, void* stream) {
    if (n_jobs <= 0 || max_tiles <= 0) return CX_OK;
    if (!jobs) return CX_ERR_ARG;
    if (n_jobs > 65535) return CX_ERR_SHAPE;
    hipLaunchKernelGGL(cast_transpose_batched_kernel, dim3(max_tiles, n_jobs), dim3(256), 0, (hipStream_t)stream, jobs);
    return done();
}

int cx_transpose_f32(const float* In, float* Out, int rows, int cols, int ld_in, int ld_out, void* stream) {
    if (rows <= 0 || cols <= 0) return CX_OK;
    dim3 grid((cols + 63) / 64, (rows + 63) / 64);
    hipLaunchKernelGGL(transpose_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, In, Out, rows, cols, ld_in,
                       ld_out);
    return done();
}

int cx_swiglu_fwd(const uint16_t* yg, uint16_t* act, int T, int I, int layout, void* stream) {
    if (T <= 0) return CX_OK;
    if ((I % 8) || (layout && (I % 32))) return CX_ERR_SHAPE;
    hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(grid_for((long)T * (I / 8))), dim3(EW_BLOCK), 0, (hipStream_t)stream,
                       yg, act, (long)T, I, layout);
    return done();
}

int cx_swiglu_bwd(const uint16_t* dact, const uint16_t* yg, uint16_t* dyg, int T, int I, int layout, void* stream) {
    if (T <= 0) return CX_OK;
    if ((I % 8) || (layout && (I % 32))) return CX_ERR_SHAPE;
    hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(grid_for((long)T * (I / 8))), dim3(EW_BLOCK), 0, (hipStream_t)stream,
                       dact, yg, dyg, (long)T, I, layout);
    return done();
}

int cx_swiglu_bwd_gate(const uint16_t* dact, const uint16_t* act, const uint16_t* gate, uint16_t* dyg, int T, int I,
                       void* stream) {
    if (T <= 0) return CX_OK;
    if (!dact || !act || !gate || !dyg) return CX_ERR_ARG;
    if (I % 32) return CX_ERR_SHAPE;
    hipLaunchKernelGGL(swiglu_bwd_gate_kernel, dim3(grid_for((long)T * (I / 8))), dim3(EW_BLOCK), 0, (hipStream_t)stream,
                       dact, act, gate, dyg, (long)T, I);
    return done();
}

int cx_bias_gelu_fwd(const uint16_t* pre, const float* bias, uint16_t* act, int T, int I, void* stream) {
    return cx_bias_act_fwd(pre, bias, act, T, I, CX_ACT_GELU, stream);
}

int cx_bias_act_fwd(const uint16_t* pre, const float* bias, uint16_t* act_out, int T, int I, int act, void* stream) {
    if (T <= 0) return CX_OK;
    if (I % 8) return CX_ERR_SHAPE;
    if (act != CX_ACT_GELU && act != CX_ACT_QUICK_GELU) return CX_ERR_ARG;
    if (act == CX_ACT_QUICK_GELU)
        hipLaunchKernelGGL(bias_gelu_fwd_kernel<CX_ACT_QUICK_GELU>, dim3(grid_for((long)T * (I / 8))), dim3(EW_BLOCK), 0,
                           (hipStream_t)stream, pre, bias, act_out, (long)T, I);
    else
        hipLaunchKernelGGL(bias_gelu_fwd_kernel<CX_ACT_GELU>, dim3(grid_for((long)T * (I / 8))), dim3(EW_BLOCK), 0,
                           (hipStream_t)stream, pre, bias, act_out, (long)T, I);
    return done();
}

int cx_bias_gelu_bwd(const uint16_t* dact, const uint16_t* pre, const float* bias, uint16_t* dpre, int T, int I,
                     void* stream) {
    if (T <= 0) return CX_OK;
    if (I % 8) return CX_ERR_SHAPE;
    hipLaunchKernelGGL(bias_gelu_bwd_kernel, dim3(grid_for((long)T * (I / 8))), dim3(EW_BLOCK), 0,
                       (hipStream_t)stream, dact, pre, bias, dpre, (long)T, I);
    return done();
}

// row groups per 256-column block: enough blocks for ~4 per CU, never more than one per 256 rows (a small problem is ONE
// block per column block: a single atomic per column, i.e. bit-reproducible -- tests/test_checkpoint_gpu.py relies on it)
static int colsum_rows_grid(int T, int N) {
    const int colblocks = (N + 255) / 256;
    int gy = (1024 + colblocks - 1) / colblocks;
    const int cap = (T + 255) / 256;
    if (gy > cap) gy = cap;
    if (gy > 512) gy = 512;
    return gy < 1 ? 1 : gy;
}

int cx_bias_grad(const uint16_t* dY, float* dbias, int T, int N, int ld, void* stream) {
    if (T <= 0 || N <= 0) return CX_OK;
    if ((N % 8) || (ld % 8)) return CX_ERR_ALIGN;
    dim3 grid((N + 255) / 256, colsum_rows_grid(T, N));
    hipLaunchKernelGGL(colsum_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, dY, (const bf16_t*)nullptr,
                       (const float*)nullptr, (bf16_t*)nullptr, dbias, T, N, ld);
    return done();
}

int cx_bias_gelu_bwd_colsum(const uint16_t* dact, const uint16_t* pre, const float* bias, uint16_t* dpre, float* dbias, int T,
                            int I, void* stream) {
    return cx_bias_act_bwd_colsum(dact, pre, bias, dpre, dbias, T, I, CX_ACT_GELU, stream);
}

int cx_bias_act_bwd_colsum(const uint16_t* dact, const uint16_t* pre, const float* bias, uint16_t* dpre, float* dbias, int T,
                           int I, int act, void* stream) {
    if (T <= 0) return CX_OK;
    if (I % 8) return CX_ERR_SHAPE;
    if (!dact || !pre || !dpre || (act != CX_ACT_GELU && act != CX_ACT_QUICK_GELU)) return CX_ERR_ARG;
    dim3 grid((I + 255) / 256, colsum_rows_grid(T, I));
    if (act == CX_ACT_QUICK_GELU)
        hipLaunchKernelGGL((colsum_kernel<true, CX_ACT_QUICK_GELU>), grid, dim3(256), 0, (hipStream_t)stream, dact, pre, bias, dpre,
                           dbias, T, I, I);
    else
        hipLaunchKernelGGL((colsum_kernel<true, CX_ACT_GELU>), grid, dim3(256), 0, (hipStream_t)stream, dact, pre, bias, dpre, dbias,
                           T, I, I);
    return done();
}

int cx_pool_normalize_fwd(const uint16_t* h, const int32_t* cu_seqlens, float* emb, float* norm, int B, int d,
                          int mode, int normalize, void* stream) {
    if (B <= 0) return CX_OK;
    if ((d % 8) || d > 2048 || d < 8) return CX_ERR_SHAPE;
    const int chunks = d / 8;
    const int ngroups = 256 / chunks > 0 ? 256 / chunks : 1;
    const size_t smem = ((size_t)ngroups * d + 256) * sizeof(float);
    hipLaunchKernelGGL(pool_normalize_fwd_kernel, dim3(B), dim3(256), smem, (hipStream_t)stream, h, cu_seqlens, emb,
                       norm, d, mode, normalize);
    return done();
}

int cx_pool_normalize_bwd(const float* demb, const float* emb, const float* norm, const int32_t* cu_seqlens,
                          uint16_t* dh, int B, int d, int mode, int normalize, void* stream) {
    if (B <= 0) return CX_OK;
    if ((d % 8) || d > 2048) return CX_ERR_SHAPE;
    const size_t smem = ((size_t)d + 8) * sizeof(float);
    hipLaunchKernelGGL(pool_normalize_bwd_kernel, dim3(B), dim3(256), smem, (hipStream_t)stream, demb, emb, norm,
                       cu_seqlens, dh, d, mode, normalize);
    return done();
}

static int rotary_launch(uint16_t* x, long tok_stride, long which_stride, int nwhich, const int32_t* cu,
                         const float* rot_cos, const float* rot_sin, int B, int H, int max_seqlen, int sign, void* stream) {
    if (!rot_cos || !rot_sin || !x || !cu) return CX_ERR_ARG;
    if ((tok_stride % 8) || (which_stride % 8)) return CX_ERR_ALIGN;
    long per_seq = (long)max_seqlen * nwhich * H * 4;
    int gx = (int)((per_seq + 255) / 256);
    if (gx > 64) gx = 64;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(rotary_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, x, tok_stride, which_stride, nwhich,
                       cu, rot_cos, rot_sin, H, sign >= 0 ? 1.f : -1.f);
    return done();
}

int cx_rotary_qkv_inplace(uint16_t* qkv, const int32_t* cu_seqlens, const float* rot_cos, const float* rot_sin,
                          int B, int H, int T, int max_seqlen, int sign, void* stream) {
    if (B <= 0 || T <= 0) return CX_OK;
    return rotary_launch(qkv, 3L * H * 64, (long)H * 64, 2, cu_seqlens, rot_cos, rot_sin, B, H, max_seqlen, sign, stream);
}

int cx_rotary_apply(uint16_t* x, long tok_stride, const int32_t* cu_seqlens, const float* rot_cos,
                    const float* rot_sin, int B, int H, int T, int max_seqlen, int sign, void* stream) {
    if (B <= 0 || T <= 0) return CX_OK;
    return rotary_launch(x, tok_stride, 0, 1, cu_seqlens, rot_cos, rot_sin, B, H, max_seqlen, sign, stream);
}

}  // extern "C"
