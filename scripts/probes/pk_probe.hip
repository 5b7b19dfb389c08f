// Lone-wave-per-SIMD VALU issue probe: is v_pk_fma_f32 (2 FMAs per lane) cheaper than two v_fma_f32 for an epilogue that is
// bound by ONE wave's issue rate?   hipcc --offload-arch=gfx950 -O3 -o pk_probe pk_probe.hip && ./pk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
    const float m = 1.0001f, c = 0.5f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        } else if (MODE == 1) {
            f2* p = reinterpret_cast<f2*>(a);
            f2 mm = {m, m}, cc = {c, c};
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(mm), "v"(cc));
        } else if (MODE == 2) {   // transcendental mix: exp + rcp per element
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0\n\tv_rcp_f32 %0, %0" : "+v"(a[i]));
        } else {                  // v_pk_mul + v_pk_add
            f2* p = reinterpret_cast<f2*>(a);
            f2 mm = {m, m}, cc = {c, c};
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %2" : "+v"(p[i]) : "v"(mm), "v"(cc));
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
double run(float* d, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3;
}
int main() {
    float* d; hipMalloc(&d, 256 * 256 * 4);
    const int iters = 20000;
    const double n_elem_ops = (double)iters * 64;   // per lane: 64 scalar FMAs (mode 0), 64 FMAs as 32 pk (mode 1), ...
    double t0 = run<0>(d, iters), t1 = run<1>(d, iters), t2 = run<2>(d, iters), t3 = run<3>(d, iters);
    // one wave per SIMD, 256 CUs; cycles at an assumed 2.4 GHz are only indicative (clock floats): compare the ratios
    printf("per wave-instruction ns:  v_fma_f32 %.3f   v_pk_fma_f32 %.3f (2 FMAs)   v_exp+v_rcp pair %.3f   v_pk_mul+v_pk_add pair %.3f\n",
           t0 / (iters * 64.0) * 1e9, t1 / (iters * 32.0) * 1e9, t2 / (iters * 64.0) * 1e9, t3 / (iters * 32.0) * 1e9);
    printf("time for the same 64 FMAs per lane:  scalar %.3f ms   packed %.3f ms  (ratio %.2f)\n", t0 * 1e3, t1 * 1e3, t1 / t0);
    (void)n_elem_ops;
    return 0;
}
