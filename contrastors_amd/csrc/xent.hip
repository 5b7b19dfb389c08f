// xent.hip -- K12: fused softmax cross-entropy over a vocabulary-sized class axis, forward and backward
// (flash_attn.losses.cross_entropy.CrossEntropyLoss, csrc/xentropy in the reference's dependency; call site
// sc/models/encoder/modeling_nomic_bert.py:603-610, `partial(CrossEntropyLoss, inplace_backward=True)`).
// HBM-bound: the forward reads every logit once (online max / sum-exp, one workgroup per row), the backward reads it
// once more and writes the gradient once -- in place when asked to (the 30528-way MLM logits are the largest tensor of
// that path).  Logits may be bf16 or fp32; statistics are fp32.
#include "cx_common.h"
#include "../../include/contrastors_hip.h"

namespace {

constexpr int XB = 256;

template <typename T>
CX_DEVICE float ld_logit(const T* p, long i) {
    if constexpr (sizeof(T) == 4) return p[i];
    else return bf16_to_f32(p[i]);
}
template <typename T>
CX_DEVICE void st_logit(T* p, long i, float v) {
    if constexpr (sizeof(T) == 4) p[i] = v;
    else p[i] = f32_to_bf16(v);
}

CX_DEVICE void online_merge(float& m, float& s, float m2, float s2) {
    const float mm = fmaxf(m, m2);
    s = (mm == -INFINITY) ? 0.f : s * __expf(m - mm) + s2 * __expf(m2 - mm);
    m = mm;
}

// 16-byte vector access: 8 bf16 or 4 fp32 per lane when V and the row stride allow it (VEC elements), else scalar.
template <typename T, int VEC>
CX_DEVICE void ld_vec(const T* p, long i, float (&v)[VEC]) {
    if constexpr (VEC == 1) {
        v[0] = ld_logit(p, i);
    } else if constexpr (sizeof(T) == 4) {
        const float4 u = *reinterpret_cast<const float4*>(p + i);
        v[0] = u.x; v[1] = u.y; v[2] = u.z; v[3] = u.w;
    } else {
        const uint4 u = *reinterpret_cast<const uint4*>(p + i);
        v[0] = bf16lo_to_f32(u.x); v[1] = bf16hi_to_f32(u.x); v[2] = bf16lo_to_f32(u.y); v[3] = bf16hi_to_f32(u.y);
        v[4] = bf16lo_to_f32(u.z); v[5] = bf16hi_to_f32(u.z); v[6] = bf16lo_to_f32(u.w); v[7] = bf16hi_to_f32(u.w);
    }
}
template <typename T, int VEC>
CX_DEVICE void st_vec(T* p, long i, const float (&v)[VEC]) {
    if constexpr (VEC == 1) {
        st_logit(p, i, v[0]);
    } else if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<float4*>(p + i) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        uint4 u;
        u.x = pack_bf16x2(v[0], v[1]); u.y = pack_bf16x2(v[2], v[3]);
        u.z = pack_bf16x2(v[4], v[5]); u.w = pack_bf16x2(v[6], v[7]);
        *reinterpret_cast<uint4*>(p + i) = u;
    }
}

// loss[row] = lse - scale * logit[label]  (0 for ignored rows);  lse[row] = log sum_j exp(scale * logit_j)
template <typename T, int VEC>
__global__ __launch_bounds__(XB) void xent_fwd_kernel(const T* __restrict__ logits, const int64_t* __restrict__ labels,
                                                      float* __restrict__ loss, float* __restrict__ lse_out, int V,
                                                      long ld, float scale, long ignore_index) {
    const int row = blockIdx.x;
    const T* x = logits + (size_t)row * ld;
    float m = -INFINITY, s = 0.f;
    for (int j = threadIdx.x * VEC; j < V; j += XB * VEC) {
        float v[VEC];
        ld_vec<T, VEC>(x, j, v);
        float mm = m;
#pragma unroll
        for (int e = 0; e < VEC; ++e) { v[e] *= scale; mm = fmaxf(mm, v[e]); }
        float acc = s * __expf(m - mm);
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc += __expf(v[e] - mm);
        s = acc;
        m = mm;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
        online_merge(m, s, m2, s2);
    }
    __shared__ float sm[XB / 64], ss[XB / 64];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sm[wave] = m; ss[wave] = s; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float M = sm[0], S = ss[0];
#pragma unroll
        for (int w = 1; w < XB / 64; ++w) online_merge(M, S, sm[w], ss[w]);
        const float lse = M + __logf(S);
        lse_out[row] = lse;
        const long lab = labels[row];
        loss[row] = (lab == ignore_index || lab < 0 || lab >= V) ? 0.f : lse - ld_logit(x, lab) * scale;
    }
}

// dlogits[row][j] = dloss[row] * scale * (exp(scale * logit_j - lse) - [j == label]);  0 for ignored rows
template <typename T, int VEC>
__global__ __launch_bounds__(XB) void xent_bwd_kernel(const float* __restrict__ dloss, const T* logits,
                                                      const float* __restrict__ lse, const int64_t* __restrict__ labels,
                                                      T* dlogits, int V, long ld, long ldd, float scale,
                                                      long ignore_index) {
    const int row = blockIdx.x;
    const T* x = logits + (size_t)row * ld;
    T* dx = dlogits + (size_t)row * ldd;
    const long lab = labels[row];
    const bool ignored = (lab == ignore_index || lab < 0 || lab >= V);
    const float g = ignored ? 0.f : dloss[row] * scale;
    const float l = lse[row];
    for (int j = threadIdx.x * VEC; j < V; j += XB * VEC) {
        float v[VEC];
        if (ignored) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] = 0.f;
        } else {
            ld_vec<T, VEC>(x, j, v);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                v[e] = __expf(v[e] * scale - l);
                if (j + e == lab) v[e] -= 1.f;
                v[e] *= g;
            }
        }
        st_vec<T, VEC>(dx, j, v);
    }
}

inline int done() { return hipGetLastError() == hipSuccess ? CX_OK : CX_ERR_LAUNCH; }

}  // namespace

extern "C" {

int cx_xent_fwd(const void* logits, int logits_bf16, const int64_t* labels, float* loss, float* lse, int N, int V,
                long ld, float logit_scale, long ignore_index, void* stream) {
    if (N <= 0) return CX_OK;
    if (!logits || !labels || !loss || !lse) return CX_ERR_ARG;
    if (V <= 0 || ld < V) return CX_ERR_SHAPE;
    const bool vec16 = ((uintptr_t)logits % 16) == 0;
    if (logits_bf16) {
        if (vec16 && (V % 8) == 0 && (ld % 8) == 0)
            hipLaunchKernelGGL((xent_fwd_kernel<bf16_t, 8>), dim3(N), dim3(XB), 0, (hipStream_t)stream, (const bf16_t*)logits,
                               labels, loss, lse, V, ld, logit_scale, ignore_index);
        else
            hipLaunchKernelGGL((xent_fwd_kernel<bf16_t, 1>), dim3(N), dim3(XB), 0, (hipStream_t)stream, (const bf16_t*)logits,
                               labels, loss, lse, V, ld, logit_scale, ignore_index);
    } else {
        if (vec16 && (V % 4) == 0 && (ld % 4) == 0)
            hipLaunchKernelGGL((xent_fwd_kernel<float, 4>), dim3(N), dim3(XB), 0, (hipStream_t)stream, (const float*)logits,
                               labels, loss, lse, V, ld, logit_scale, ignore_index);
        else
            hipLaunchKernelGGL((xent_fwd_kernel<float, 1>), dim3(N), dim3(XB), 0, (hipStream_t)stream, (const float*)logits,
                               labels, loss, lse, V, ld, logit_scale, ignore_index);
    }
    return done();
}

int cx_xent_bwd(const float* dloss, const void* logits, int logits_bf16, const float* lse, const int64_t* labels,
                void* dlogits, int N, int V, long ld, long ld_d, float logit_scale, long ignore_index, void* stream) {
    if (N <= 0) return CX_OK;
    if (!dloss || !logits || !lse || !labels || !dlogits) return CX_ERR_ARG;
    if (V <= 0 || ld < V || ld_d < V) return CX_ERR_SHAPE;
    const bool vec16 = ((uintptr_t)logits % 16) == 0 && ((uintptr_t)dlogits % 16) == 0;
    if (logits_bf16) {
        if (vec16 && (V % 8) == 0 && (ld % 8) == 0 && (ld_d % 8) == 0)
            hipLaunchKernelGGL((xent_bwd_kernel<bf16_t, 8>), dim3(N), dim3(XB), 0, (hipStream_t)stream, dloss,
                               (const bf16_t*)logits, lse, labels, (bf16_t*)dlogits, V, ld, ld_d, logit_scale, ignore_index);
        else
            hipLaunchKernelGGL((xent_bwd_kernel<bf16_t, 1>), dim3(N), dim3(XB), 0, (hipStream_t)stream, dloss,
                               (const bf16_t*)logits, lse, labels, (bf16_t*)dlogits, V, ld, ld_d, logit_scale, ignore_index);
    } else {
        if (vec16 && (V % 4) == 0 && (ld % 4) == 0 && (ld_d % 4) == 0)
            hipLaunchKernelGGL((xent_bwd_kernel<float, 4>), dim3(N), dim3(XB), 0, (hipStream_t)stream, dloss,
                               (const float*)logits, lse, labels, (float*)dlogits, V, ld, ld_d, logit_scale, ignore_index);
        else
            hipLaunchKernelGGL((xent_bwd_kernel<float, 1>), dim3(N), dim3(XB), 0, (hipStream_t)stream, dloss,
                               (const float*)logits, lse, labels, (float*)dlogits, V, ld, ld_d, logit_scale, ignore_index);
    }
    return done();
}

}  // extern "C"
