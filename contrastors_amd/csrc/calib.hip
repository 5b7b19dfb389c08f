// calib.hip -- per-box calibration kernels for bench.py: the same code measured 4209 / 4143 / 4127 / 3966
// pairs/s on four boxes of the pool -- the sustained matrix-core clock under load is set by each package's power budget, not by
// the instruction stream.  These two kernels measure what THIS box can do right before the headline leg:
//   cx_calib_mfma_bf16   register-only v_mfma_f32_16x16x32_bf16 loop (the GEMMs' instruction since round 6; 32x32x16 before: ~10 % fewer
//                        FLOP per joule) on caller-provided (random) fragments, one wave per SIMD on
//                        every CU: no LDS, no memory -- the ceiling any bf16 GEMM main loop can reach on this package at its
//                        power limit (1.56 PFLOP/s on the round-2 box against the 2.5 PFLOP/s the data sheet prices)
//   cx_calib_copy        16-byte-per-lane grid-stride copy: the achievable HBM stream rate (read + write)
// Measurement aids like cx_prof_gemm_*: no product kernel calls them.
#include "cx_common.h"
#include "../../include/contrastors_hip.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 cb_bf16x8;
typedef __attribute__((ext_vector_type(4))) float cb_f32x4;

__global__ __launch_bounds__(256) void calib_mfma_kernel(const uint4* __restrict__ seed, int iters, long long* cyc, float* sink) {
    const uint4 a0 = seed[threadIdx.x], a1 = seed[threadIdx.x + 256], b0 = seed[threadIdx.x + 512], b1 = seed[threadIdx.x + 768];
    cb_f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    const cb_bf16x8 fa0 = __builtin_bit_cast(cb_bf16x8, a0), fa1 = __builtin_bit_cast(cb_bf16x8, a1);
    const cb_bf16x8 fb0 = __builtin_bit_cast(cb_bf16x8, b0), fb1 = __builtin_bit_cast(cb_bf16x8, b1);
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i)   // 16 x 16 KFLOP = the 8 x 32 KFLOP of the 32x32x16 form this probe used until round 5
            // (inline asm, accumulator tied in place: through the builtin the compiler rotates the 16 small accumulators between
            // iterations -- register copies and s_nop between the MFMAs, 27 cycles each instead of 17)
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"((i & 1) ? fa1 : fa0), "v"((i & 2) ? fb1 : fb0));
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");   // MFMA results read by VALU below: the hazard recogniser does not see into the asm
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += acc[i][r];
    if (s == 12345.678f) sink[0] = s;
    if (cyc && (threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

__global__ __launch_bounds__(256) void calib_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, long n16) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}

}  // namespace

extern "C" {

// seed: 1024 x 16 B of (random) bf16 fragments; nwg workgroups of 4 waves (one per SIMD) run `iters` x 16 MFMAs of 16x16x32 each:
// FLOPs = nwg * 4 * iters * 8 * 2 * 32 * 32 * 16 (unchanged: 16 x 16 KFLOP = 8 x 32 KFLOP per iteration).  cycles (may be NULL): nwg x 4 s_memtime deltas.
int cx_calib_mfma_bf16(const void* seed_1024x16B, int iters, int nwg, long long* cycles_nwg_x4, float* sink, void* stream) {
    if (!seed_1024x16B || !sink || iters <= 0 || nwg <= 0) return CX_ERR_ARG;
    hipLaunchKernelGGL(calib_mfma_kernel, dim3(nwg), dim3(256), 0, (hipStream_t)stream, (const uint4*)seed_1024x16B, iters,
                       cycles_nwg_x4, sink);
    return hipGetLastError() == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
}

int cx_calib_copy(const void* src, void* dst, long bytes, void* stream) {
    if (!src || !dst || bytes <= 0 || (bytes & 15)) return CX_ERR_ARG;
    hipLaunchKernelGGL(calib_copy_kernel, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, (uint4*)dst, bytes / 16);
    return hipGetLastError() == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
}

}  // extern "C"
