// probe.hip -- hardware self-checks run by tests/test_probe_gpu.py: they pin, on the actual gfx950 silicon, the two
// layout facts every MFMA kernel in this library relies on (accumulator row/column mapping of v_mfma_f32_32x32x16_bf16
// and the cross-lane pattern of ds_read_b64_tr_b16).
#include "cx_common.h"
#include "../../include/contrastors_hip.h"

namespace {

// D[i][j] = (i+1) + 64*(j+1) from A[i][0]=i+1, A[i][1]=1, B[0][j]=1, B[1][j]=64*(j+1)  (all exact in bf16).
__global__ void probe_mfma_kernel(float* out) {
    const int lane = threadIdx.x & 63, hi = lane >> 5, l31 = lane & 31;
    union { bf16_t h[8]; bf16x8_t v; } a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) a.h[e] = b.h[e] = 0;
    if (hi == 0) {
        a.h[0] = f32_to_bf16((float)(l31 + 1));
        a.h[1] = f32_to_bf16(1.f);
        b.h[0] = f32_to_bf16(1.f);
        b.h[1] = f32_to_bf16(64.f * (l31 + 1));
    }
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = mfma_bf16_32x32x16(a.v, b.v, acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) out[acc_row(r, hi) * 32 + l31] = acc[r];
}

__global__ void probe_tr16_kernel(const uint16_t* in, uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[256];
    const int lane = threadIdx.x;
    for (int e = 0; e < 4; ++e) lds[lane * 4 + e] = in[lane * 4 + e];
    __syncthreads();
    typedef __attribute__((address_space(3))) bf16x4_t* lds_b4_ptr;
    const bf16x4_t t = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4_ptr)(lds + lane * 4));
    union { bf16x4_t v; uint16_t h[4]; } u;
    u.v = t;
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = u.h[e];
}

// ---- global -> LDS DMA throughput probe ---------------------------------------------------------------------------
// 512-thread workgroups, one per CU.  Each iteration every wave issues `per_wave` global_load_lds_dwordx4 (1 KiB
// each), waits, and hits a barrier -- the skeleton of the GEMM main loop without MFMA.  Address pattern per
// instruction: 8 rows x 128 B with `row_stride` bytes between rows (row_stride = 128 -> one contiguous KiB).
// `span` bytes per workgroup are walked cyclically (small span = L2/L1 resident, huge span = HBM stream).
typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* glb_vp;

__global__ __launch_bounds__(512, 2) void probe_dma_kernel(const char* __restrict__ src, long wg_stride, long span,
                                                           long row_stride, int per_wave, int iters, int depth,
                                                           float* sink) {
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = src + (long)blockIdx.x * wg_stride;
    const long instr_bytes = (row_stride == 128) ? 1024 : 8 * row_stride;  // address range one instruction spans
    long off = (long)wave * per_wave * instr_bytes;
    const long lane_off = (long)(lane >> 3) * row_stride + (lane & 7) * 16;
    for (int it = 0; it < iters; ++it) {
        for (int j = 0; j < per_wave; ++j) {
            const char* g = base + (off % span) + lane_off;
            __builtin_amdgcn_global_load_lds((glb_vp)g, (lds_vp)(dsm + ((it % depth) * 8 * per_wave + wave * per_wave + j) * 1024),
                                             16, 0, 0);
            off += instr_bytes;
        }
        off += (long)7 * per_wave * instr_bytes;  // the other 7 waves' share
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if (threadIdx.x == 0 && sink) sink[blockIdx.x] = reinterpret_cast<float*>(dsm)[0];
}

// ---- MFMA issue-rate probe: `waves` waves per workgroup (one workgroup per CU), each looping over 8 independent
// 32x32x16 bf16 MFMAs on fragments taken from `seed` (random data = realistic switching power). out[wg] = cycles.
typedef __attribute__((ext_vector_type(8))) __bf16 pb_bf16x8;
typedef __attribute__((ext_vector_type(16))) float pb_f32x16;
__global__ __launch_bounds__(512) void probe_mfma_rate_kernel(const uint4* __restrict__ seed, int iters, long long* cyc,
                                                              float* sink) {
    const uint4 a0 = seed[threadIdx.x], a1 = seed[threadIdx.x + 512], b0 = seed[threadIdx.x + 1024],
                b1 = seed[threadIdx.x + 1536];
    pb_f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const pb_bf16x8 fa0 = __builtin_bit_cast(pb_bf16x8, a0), fa1 = __builtin_bit_cast(pb_bf16x8, a1);
    const pb_bf16x8 fb0 = __builtin_bit_cast(pb_bf16x8, b0), fb1 = __builtin_bit_cast(pb_bf16x8, b1);
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((i & 1) ? fa1 : fa0, (i & 2) ? fb1 : fb0, acc[i], 0, 0, 0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

// the same FLOPs per iteration from the 16x16x32 form (16 instructions of 4 passes instead of 8 of 8 passes): half the
// accumulator-register traffic per FLOP -- does the part sustain a higher rate at its power limit with it?
typedef __attribute__((ext_vector_type(4))) float pb_f32x4;
__global__ __launch_bounds__(512) void probe_mfma_rate16_kernel(const uint4* __restrict__ seed, int iters, long long* cyc,
                                                                float* sink) {
    const uint4 a0 = seed[threadIdx.x], a1 = seed[threadIdx.x + 512], b0 = seed[threadIdx.x + 1024],
                b1 = seed[threadIdx.x + 1536];
    pb_f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    const pb_bf16x8 fa0 = __builtin_bit_cast(pb_bf16x8, a0), fa1 = __builtin_bit_cast(pb_bf16x8, a1);
    const pb_bf16x8 fb0 = __builtin_bit_cast(pb_bf16x8, b0), fb1 = __builtin_bit_cast(pb_bf16x8, b1);
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16((i & 1) ? fa1 : fa0, (i & 2) ? fb1 : fb0, acc[i], 0, 0, 0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += acc[i][r];
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

// ---- dQ read-modify-write probe (round 5): what the accumulation traffic of a single-owner fused long-sequence
// attention backward costs BY ITSELF.  One workgroup owns a (sequence, head) problem; with dK / dV in registers and the query
// blocks in the inner loop, every (key block, query block) pair adds a 128 x 64 fp32 partial to the problem's dQ scratch: per
// problem `sweeps` (= key blocks) passes of load + add + store over `floats` (= S x 64) floats that only this workgroup touches.
// No MFMA, no softmax: the floor the fused form pays on top of its five GEMMs.
__global__ __launch_bounds__(256) void probe_rmw_kernel(float* __restrict__ buf, long floats, int sweeps, int n_problems) {
    for (int pr = blockIdx.x; pr < n_problems; pr += gridDim.x) {
        float4* base = reinterpret_cast<float4*>(buf + (size_t)pr * floats);
        const long n4 = floats / 4;
        for (int sw = 0; sw < sweeps; ++sw) {
            const float add = 1.0f + sw;
            for (long i = threadIdx.x; i < n4; i += 256) {
                float4 v = base[i];
                v.x += add; v.y += add; v.z += add; v.w += add;
                base[i] = v;
            }
            __syncthreads();
        }
    }
}

}  // namespace

extern "C" {
int cx_probe_rmw(float* buf, long floats_per_problem, int sweeps, int n_problems, int nwg, void* stream) {
    if (!buf || floats_per_problem <= 0 || (floats_per_problem & 3) || sweeps <= 0 || n_problems <= 0 || nwg <= 0) return CX_ERR_ARG;
    hipLaunchKernelGGL(probe_rmw_kernel, dim3(nwg), dim3(256), 0, (hipStream_t)stream, buf, floats_per_problem, sweeps, n_problems);
    return hipGetLastError() == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
}
int cx_probe_mfma_rate16(const void* seed_2048x16B, int waves, int iters, int nwg, long long* cycles_nwg_x8, float* sink,
                         void* stream) {
    if (waves < 1 || waves > 8) return CX_ERR_SHAPE;
    hipLaunchKernelGGL(probe_mfma_rate16_kernel, dim3(nwg), dim3(64 * waves), 0, (hipStream_t)stream,
                       (const uint4*)seed_2048x16B, iters, cycles_nwg_x8, sink);
    return hipGetLastError() == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
}
int cx_probe_mfma_rate(const void* seed_2048x16B, int waves, int iters, int nwg, long long* cycles_nwg_x8, float* sink,
                       void* stream) {
    if (waves < 1 || waves > 8) return CX_ERR_SHAPE;
    hipLaunchKernelGGL(probe_mfma_rate_kernel, dim3(nwg), dim3(64 * waves), 0, (hipStream_t)stream,
                       (const uint4*)seed_2048x16B, iters, cycles_nwg_x8, sink);
    return hipGetLastError() == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
}
int cx_probe_dma_bw(const void* src, long wg_stride, long span, long row_stride, int per_wave, int iters, int depth,
                    int nwg, float* sink, void* stream) {
    const int lds = depth * 8 * per_wave * 1024;
    if (lds > 160 * 1024) return CX_ERR_SHAPE;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_dma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            lds) != hipSuccess)
        return CX_ERR_LAUNCH;
    hipLaunchKernelGGL(probe_dma_kernel, dim3(nwg), dim3(512), lds, (hipStream_t)stream, (const char*)src, wg_stride, span,
                       row_stride, per_wave, iters, depth, sink);
    return hipGetLastError() == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
}
int cx_probe_mfma_layout(float* out_32x32, void* stream) {
    hipLaunchKernelGGL(probe_mfma_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out_32x32);
    return hipGetLastError() == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
}
int cx_probe_ds_read_tr16(const uint16_t* in_64x4, uint16_t* out_64x4, void* stream) {
    hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, in_64x4, out_64x4);
    return hipGetLastError() == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
}
}
