// optimizer.hip -- the tail of training_step (sc/trainers/base.py:362-385, sc/optimizer.py:7-47) as two HBM-bound
// kernels over the flat fp32 parameter / gradient buffers:
//   1. grad_sq_norm: sum of squares of every gradient into ONE device double (clip_grad_norm_'s total norm),
//   2. adamw_clip_step: reads that double, derives torch's clip coefficient min(1, max_norm / (norm + 1e-6)) on the
//      device (no host sync), and applies decoupled-weight-decay Adam (torch.optim.AdamW, amsgrad=False) in one pass:
//      16 B read + 12 B written per parameter (p, g, m, v -> p, m, v), the scaled gradient is never written back.
// torch's foreach path makes ~7 passes over the 547 MB of state for the same update.
#include "cx_common.h"
#include "../../include/contrastors_hip.h"

namespace {

__global__ __launch_bounds__(256) void grad_sq_norm_kernel(const float* __restrict__ g, long n, double* __restrict__ out) {
    const long n4 = n / 4;
    const long stride = (long)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4*>(g)[i];
        acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float v = g[n4 * 4 + threadIdx.x];
        acc += v * v;
    }
    double d = (double)acc;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

struct AdamArgs {
    float lr, beta1, beta2, eps, weight_decay, bias_c1, bias_c2_sqrt, max_norm;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamArgs& a, float coef) {
    g *= coef;
    p *= 1.f - a.lr * a.weight_decay;                 // decoupled decay first (torch._single_tensor_adamw order)
    m += (g - m) * (1.f - a.beta1);                   // exp_avg.lerp_(grad, 1 - beta1)
    v = v * a.beta2 + (1.f - a.beta2) * g * g;        // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    const float denom = sqrtf(v) / a.bias_c2_sqrt + a.eps;
    p -= (a.lr / a.bias_c1) * (m / denom);
}

__global__ __launch_bounds__(256) void adamw_clip_step_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                              float* __restrict__ m, float* __restrict__ v, long n,
                                                              AdamArgs a, const double* __restrict__ sq_norm) {
    float coef = 1.f;
    if (sq_norm && a.max_norm > 0.f) {
        const float norm = (float)sqrt(*sq_norm);
        coef = fminf(1.f, a.max_norm / (norm + 1e-6f));  // torch.nn.utils.clip_grad_norm_
    }
    const long n4 = n / 4;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        adam_one(pp.x, gg.x, mm.x, vv.x, a, coef);
        adam_one(pp.y, gg.y, mm.y, vv.y, a, coef);
        adam_one(pp.z, gg.z, mm.z, vv.z, a, coef);
        adam_one(pp.w, gg.w, mm.w, vv.w, a, coef);
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long i = n4 * 4 + threadIdx.x;
        float pp = p[i], mm = m[i], vv = v[i];
        adam_one(pp, g[i], mm, vv, a, coef);
        p[i] = pp; m[i] = mm; v[i] = vv;
    }
}

// ema = decay * ema + (1 - decay) * param over a flat fp32 buffer (sc/trainers/base.py:387-391: `self.model["ema"].update`)
__global__ __launch_bounds__(256) void ema_update_kernel(float* __restrict__ ema, const float* __restrict__ p, long n, float decay) {
    const long n4 = n / 4;
    const long stride = (long)gridDim.x * blockDim.x;
    const float w = 1.f - decay;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 e = reinterpret_cast<float4*>(ema)[i];
        const float4 q = reinterpret_cast<const float4*>(p)[i];
        e.x = decay * e.x + w * q.x; e.y = decay * e.y + w * q.y; e.z = decay * e.z + w * q.z; e.w = decay * e.w + w * q.w;
        reinterpret_cast<float4*>(ema)[i] = e;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long i = n4 * 4 + threadIdx.x;
        ema[i] = decay * ema[i] + w * p[i];
    }
}

inline int grid_for(long n) {
    long blocks = (n / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 16) blocks = 256 * 16;  // grid-stride: 16 workgroups per CU keep the HBM queues full
    return (int)blocks;
}

inline bool aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

extern "C" {

int cx_grad_sq_norm(const float* grad, long n, double* sq_norm_accum, void* stream) {
    if (n <= 0) return CX_OK;
    if (!grad || !sq_norm_accum) return CX_ERR_ARG;
    if (!aligned16(grad)) return CX_ERR_SHAPE;
    hipLaunchKernelGGL(grad_sq_norm_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, grad, n, sq_norm_accum);
    return hipGetLastError() == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
}

int cx_adamw_clip_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1,
                       float beta2, float eps, float weight_decay, long step, const double* sq_norm, float max_norm,
                       void* stream) {
    if (n <= 0) return CX_OK;
    if (!param || !grad || !exp_avg || !exp_avg_sq || step < 1) return CX_ERR_ARG;
    if (!aligned16(param) || !aligned16(grad) || !aligned16(exp_avg) || !aligned16(exp_avg_sq)) return CX_ERR_SHAPE;
    AdamArgs a;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay;
    a.bias_c1 = (float)(1.0 - pow((double)beta1, (double)step));
    a.bias_c2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    a.max_norm = max_norm;
    hipLaunchKernelGGL(adamw_clip_step_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_avg_sq, n, a, sq_norm);
    return hipGetLastError() == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
}

int cx_ema_update(float* ema, const float* param, long n, float decay, void* stream) {
    if (n <= 0) return CX_OK;
    if (!ema || !param || !(decay >= 0.f) || decay > 1.f) return CX_ERR_ARG;
    if (!aligned16(ema) || !aligned16(param)) return CX_ERR_SHAPE;
    hipLaunchKernelGGL(ema_update_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, ema, param, n, decay);
    return hipGetLastError() == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
}

}  // extern "C"
