// gemm_api.hip -- C-ABI entry points of the bf16 MFMA GEMM family (include/contrastors_hip.h, K9 FusedDense and its fused
// forms) for the PRODUCT library.  Two kernel files sit behind it:
//   gemm_bf16_v6.hip  one wave per SIMD, persistent 256x256x64 tiles: every bf16-output NT form (plain / bias / alpha /
//                     residual, fused SwiGLU, fused bias+GELU, fc2-dgrad + SwiGLU backward) and the natural-layout wgrad;
//   gemm_bf16_v5.hip  256x256x64, one tile per workgroup: fp32-output and fp32-partial NT forms, bf16 forms whose N or
//                     leading dimension is not a multiple of 8.
// The earlier kernel generations (v1-v4, the 8-wave persistent v5p), the variant switch and every debug setter live in
// the DEV library only (libcontrastors_hip_dev.so: gemm_bf16.hip, include/contrastors_hip_dev.h).  No mutable globals
// here except the opt-in launch profiler used by bench.py, which is mutex-guarded.
#include "cx_common.h"
#include "../../include/contrastors_hip.h"
#include "gemm_params.h"
#include <mutex>

namespace {

constexpr int BK = 64;

// fixed-order reduction of split-K fp32 partial slabs into Out (+=): deterministic, no atomics
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, long n4,
                                                            long slab, int splits) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 a = reinterpret_cast<const float4*>(out)[i];
        for (int s = 0; s < splits; ++s) {
            const float4 v = reinterpret_cast<const float4*>(part + s * slab)[i];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        reinterpret_cast<float4*>(out)[i] = a;
    }
}

int reduce_slabs(const float* ws, float* out, long slab, int split, hipStream_t s) {
    const long n4 = slab / 4;
    long g = (n4 + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((int)g), dim3(256), 0, s, ws, out, n4, slab, split);
    return hipGetLastError() == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
}

// split-K factor for `tiles` output tiles on 256 CUs (one resident workgroup per CU): maximise the fill of whole rounds,
// lightly penalising extra partial slabs (each costs one more pass of the reduction)
long choose_split(long tiles, long max_split) {
    if (max_split < 1) max_split = 1;
    long best = 1;
    double best_score = -1.0;
    for (long s = 1; s <= max_split; ++s) {
        const long wgs = tiles * s;
        const long rounds = (wgs + 255) / 256;
        const double score = (double)wgs / (double)(rounds * 256) - 0.006 * (double)s;
        if (score > best_score + 1e-9) {
            best_score = score;
            best = s;
        }
    }
    return best;
}

// ---- sampled per-launch timing (bench.py's live roofline measurement): every `stride`-th launch is bracketed by two
// HIP events recorded on the launch stream.  Off by default; a mutex keeps concurrent callers consistent.
struct GemmProf {
    std::mutex mu;
    bool enabled = false;
    int stride = 1;
    long launches = 0;
    static constexpr int CAP = 8192;
    hipEvent_t ev0[CAP], ev1[CAP];
    double flop[CAP];
    int created = 0, used = 0;
} g_prof;

struct ProfScope {
    int slot = -1;
    hipStream_t s;
    ProfScope(double flop, hipStream_t stream) : s(stream) {
        if (!g_prof.enabled) return;   // (racy read of a bool that only bench.py flips between steps: benign)
        std::lock_guard<std::mutex> lk(g_prof.mu);
        if ((g_prof.launches % g_prof.stride) == 0 && g_prof.used < GemmProf::CAP) {
            const int sl = g_prof.used;
            if (sl >= g_prof.created) {
                if (hipEventCreate(&g_prof.ev0[sl]) != hipSuccess || hipEventCreate(&g_prof.ev1[sl]) != hipSuccess) return;
                g_prof.created = sl + 1;
            }
            slot = g_prof.used++;
            g_prof.flop[slot] = flop;
            (void)hipEventRecord(g_prof.ev0[slot], s);
        }
        ++g_prof.launches;
    }
    ~ProfScope() {
        if (slot >= 0) (void)hipEventRecord(g_prof.ev1[slot], s);
    }
};

GemmParams base_params(const uint16_t* X, const uint16_t* W, void* Out, const float* bias, int M, int N, int K, int ldx,
                       int ldw, int ldo) {
    GemmParams p = {};
    p.X = X; p.W = W; p.Out = Out; p.bias = bias;
    p.M = M; p.N = N; p.K = K; p.ldx = ldx; p.ldw = ldw; p.ldo = ldo;
    p.split_k = 1; p.alpha = 1.f;
    return p;
}

}  // namespace

extern "C" {

int cx_gemm_bf16_nt(const uint16_t* X, const uint16_t* W, void* Out, const float* bias, int M, int N, int K, int ldx,
                    int ldw, int ldo, int out_mode, int split_k, float alpha, void* stream) {
    if (M <= 0 || N <= 0) return CX_OK;
    if (K <= 0 || (K % BK) != 0 || (N % 4) != 0) return CX_ERR_SHAPE;
    if ((ldx % 8) != 0 || (ldw % 8) != 0 || (ldo % 4) != 0) return CX_ERR_ALIGN;
    if (out_mode < 0 || out_mode > 1) return CX_ERR_ARG;   // accumulation: cx_gemm_bf16_nt_accum (deterministic split-K)
    (void)split_k;
    GemmParams p = base_params(X, W, Out, bias, M, N, K, ldx, ldw, ldo);
    p.alpha = alpha;
    ProfScope prof(2.0 * (double)M * (double)N * (double)K, (hipStream_t)stream);
    hipError_t e;
    if (out_mode == 0 && (N % 8) == 0 && (ldo % 8) == 0)
        e = cx_launch_gemm_v6(p, GEMM_EPI_NONE, (hipStream_t)stream);
    else
        e = cx_launch_gemm_v5(p, 0, out_mode == 0 ? GEMM_OUT_BF16 : GEMM_OUT_F32, GEMM_EPI_NONE, (hipStream_t)stream);
    return e == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
}

// Out(M,N) fp32 += X W^T with the K range split over `split` workgroup groups whose fp32 partial tiles go to `ws` with
// plain stores, followed by one fixed-order reduction pass (device-scope atomics serialise at the memory fabric because
// the XCD L2s are not coherent: 235 us floor per launch measured in round 1).
int cx_gemm_bf16_nt_accum(const uint16_t* X, const uint16_t* W, float* Out, float* ws, long ws_floats, int M, int N, int K,
                          int ldx, int ldw, void* stream) {
    if (M <= 0 || N <= 0) return CX_OK;
    if (K <= 0 || (K % BK) != 0 || (N % 4) != 0) return CX_ERR_SHAPE;
    if ((ldx % 8) != 0 || (ldw % 8) != 0) return CX_ERR_ALIGN;
    if (!ws || !Out) return CX_ERR_ARG;
    const long slab = (long)M * N;
    if (ws_floats < slab) return CX_ERR_SHAPE;
    const long tiles = (long)((M + 255) / 256) * ((N + 255) / 256), nk = K / BK;
    long max_split = nk / 4;                               // keep >= 4 K-tiles per slice (pipeline depth)
    if (max_split > ws_floats / slab) max_split = ws_floats / slab;
    if (max_split > 16) max_split = 16;
    GemmParams p = base_params(X, W, ws, nullptr, M, N, K, ldx, ldw, N);
    p.split_k = (int)choose_split(tiles, max_split);
    if (p.split_k > nk) p.split_k = (int)nk;
    if (cx_launch_gemm_v5(p, 0, GEMM_OUT_F32_PARTIAL, GEMM_EPI_NONE, (hipStream_t)stream) != hipSuccess) return CX_ERR_LAUNCH;
    return reduce_slabs(ws, Out, slab, p.split_k, (hipStream_t)stream);
}

// wgrad without transposes: G(O,I) fp32 += dY(T,O)^T A(T,I).  Tp = round_up(T,64) rows of both operands are read; rows
// T..Tp-1 MUST be zero.  O % 256 == 0 and I % 256 == 0 (all BASELINE encoder shapes) -- otherwise CX_ERR_SHAPE: transpose
// the operands (cx_transpose_bf16) and use cx_gemm_bf16_nt_accum.
int cx_gemm_bf16_tn_accum(const uint16_t* dY, const uint16_t* A, float* G, float* ws, long ws_floats, int T, int O, int I,
                          int ld_dy, int ld_a, void* stream) {
    if (T <= 0 || O <= 0 || I <= 0) return CX_OK;
    if ((O % 256) != 0 || (I % 256) != 0) return CX_ERR_SHAPE;
    if ((ld_dy % 8) != 0 || (ld_a % 8) != 0) return CX_ERR_ALIGN;
    if (!ws || !G) return CX_ERR_ARG;
    const long slab = (long)O * I;
    if (ws_floats < slab) return CX_ERR_SHAPE;
    const int Tp = (T + BK - 1) / BK * BK;
    GemmParams p = base_params(dY, A, ws, nullptr, O, I, Tp, ld_dy, ld_a, I);
    const long tiles = (long)(O / 256) * (I / 256), nk = Tp / BK;
    long max_split = nk / 4;
    if (max_split > ws_floats / slab) max_split = ws_floats / slab;
    if (max_split > 32) max_split = 32;
    p.split_k = (int)choose_split(tiles, max_split);
    {
        ProfScope prof(2.0 * (double)T * (double)O * (double)I, (hipStream_t)stream);
        if (cx_launch_gemm_v6_tn(p, (hipStream_t)stream) != hipSuccess) return CX_ERR_LAUNCH;
    }
    return reduce_slabs(ws, G, slab, p.split_k, (hipStream_t)stream);
}

// fc1 of the gated MLP with SwiGLU fused into the epilogue (K9 + K10).  W:(2I, K) holds fc11/fc12 rows interleaved
// in groups of 32 ([y rows 0..31 | gate rows 0..31 | y rows 32..63 | ...]); YG (optional, may be NULL):(M, 2I) in the
// same interleaved column layout; Act:(M, I) = silu(gate) * y.
int cx_gemm_bf16_swiglu(const uint16_t* X, const uint16_t* W, uint16_t* YG, uint16_t* Act, int M, int I, int K, int ldx,
                        int ldw, int ld_yg, int ld_act, void* stream) {
    if (M <= 0 || I <= 0) return CX_OK;
    if (K <= 0 || (K % BK) != 0 || (I % 32) != 0) return CX_ERR_SHAPE;
    if ((ldx % 8) != 0 || (ldw % 8) != 0 || (ld_yg % 4) != 0 || (ld_act % 4) != 0) return CX_ERR_ALIGN;
    if (!Act) return CX_ERR_ARG;
    GemmParams p = base_params(X, W, YG, nullptr, M, 2 * I, K, ldx, ldw, ld_yg);
    p.Out2 = Act; p.ldo2 = ld_act;
    ProfScope prof(2.0 * (double)M * (double)(2 * I) * (double)K, (hipStream_t)stream);
    const hipError_t e = ((ld_yg % 8) == 0 && (ld_act % 8) == 0)
                             ? cx_launch_gemm_v6(p, GEMM_EPI_SWIGLU, (hipStream_t)stream)
                             : cx_launch_gemm_v5(p, 0, GEMM_OUT_BF16, GEMM_EPI_SWIGLU, (hipStream_t)stream);
    return e == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
}

// The same GEMM whose optional save is the GATE alone: G (may be NULL): (M, I) bf16 = bf16(X Wg^T), plain column order.
// With the activation (which fc2 needs anyway) it determines the SwiGLU backward (cx_gemm_bf16_swiglu_bwd_gate): the saving
// forward writes 2 (M, I) tensors instead of 3, and the activation arena loses a quarter of its bytes per token and layer.
int cx_gemm_bf16_swiglu_gate(const uint16_t* X, const uint16_t* W, uint16_t* G, uint16_t* Act, int M, int I, int K, int ldx,
                             int ldw, int ld_g, int ld_act, void* stream) {
    if (M <= 0 || I <= 0) return CX_OK;
    if (K <= 0 || (K % BK) != 0 || (I % 32) != 0 || (ld_g % 8) != 0 || (ld_act % 8) != 0) return CX_ERR_SHAPE;
    if ((ldx % 8) != 0 || (ldw % 8) != 0) return CX_ERR_ALIGN;
    if (!Act) return CX_ERR_ARG;
    GemmParams p = base_params(X, W, G, nullptr, M, 2 * I, K, ldx, ldw, ld_g);
    p.Out2 = Act; p.ldo2 = ld_act;
    ProfScope prof(2.0 * (double)M * (double)(2 * I) * (double)K, (hipStream_t)stream);
    return cx_launch_gemm_v6(p, GEMM_EPI_SWIGLU_G, (hipStream_t)stream) == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
}

// Out (M,N) bf16 = bf16(bf16(X W^T + bias) + Residual): a projection whose output feeds `x0 + residual -> LayerNorm`
// (out_proj and fc2 of every block).  CX_ERR_SHAPE when the one-wave-per-SIMD kernel does not cover the shape.
int cx_gemm_bf16_nt_residual(const uint16_t* X, const uint16_t* W, uint16_t* Out, const float* bias, const uint16_t* Residual,
                             int M, int N, int K, int ldx, int ldw, int ldo, int ldr, void* stream) {
    if (M <= 0 || N <= 0) return CX_OK;
    if (!Residual || !Out) return CX_ERR_ARG;
    if (K <= 0 || (K % 64) != 0 || (N % 8) != 0 || (ldo % 8) != 0 || (ldr % 8) != 0) return CX_ERR_SHAPE;
    if ((ldx % 8) != 0 || (ldw % 8) != 0) return CX_ERR_ALIGN;
    GemmParams p = base_params(X, W, Out, bias, M, N, K, ldx, ldw, ldo);
    p.Out2 = const_cast<uint16_t*>(Residual); p.ldo2 = ldr;
    ProfScope prof(2.0 * (double)M * (double)N * (double)K, (hipStream_t)stream);
    return cx_launch_gemm_v6(p, GEMM_EPI_NONE, (hipStream_t)stream) == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
}

// fc1 of the plain (GELU) MLP with bias + erf-GELU fused into the epilogue.  Pre: (M, N) bf16 pre-activation = X W^T +
// bias (optional, may be NULL: the no-grad pass), Act: (M, N) bf16 = gelu(Pre).  Returns CX_ERR_SHAPE when the fused
// kernel does not cover the shape (caller then runs GEMM + cx_bias_gelu_fwd).
int cx_gemm_bf16_bias_gelu(const uint16_t* X, const uint16_t* W, const float* bias, uint16_t* Pre, uint16_t* Act, int M, int N,
                           int K, int ldx, int ldw, int ld_pre, int ld_act, void* stream) {
    return cx_gemm_bf16_bias_act(X, W, bias, Pre, Act, M, N, K, ldx, ldw, ld_pre, ld_act, 0, stream);
}

int cx_gemm_bf16_bias_act(const uint16_t* X, const uint16_t* W, const float* bias, uint16_t* Pre, uint16_t* Act, int M, int N,
                          int K, int ldx, int ldw, int ld_pre, int ld_act, int act, void* stream) {
    if (M <= 0 || N <= 0) return CX_OK;
    if (!Act || (act != 0 && act != 1)) return CX_ERR_ARG;
    if (K <= 0 || (K % 64) != 0 || (N % 8) != 0 || (ld_act % 8) != 0 || (Pre && (ld_pre % 8) != 0)) return CX_ERR_SHAPE;
    if ((ldx % 8) != 0 || (ldw % 8) != 0) return CX_ERR_ALIGN;
    GemmParams p = base_params(X, W, Pre, bias, M, N, K, ldx, ldw, ld_pre);
    p.Out2 = Act; p.ldo2 = ld_act;
    p.act = act;
    ProfScope prof(2.0 * (double)M * (double)N * (double)K, (hipStream_t)stream);
    return cx_launch_gemm_v6(p, act == 1 ? GEMM_EPI_QGELU : GEMM_EPI_GELU, (hipStream_t)stream) == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
}

// fc2 dgrad of the gated MLP with the SwiGLU backward fused into the epilogue: dYG (M, 2I) = swiglu'(YG) * (dY W^T), where
// W: (I, K) is the transposed fc2 shadow (the NT operand of the dgrad GEMM).  d(act) is never materialised.  Returns
// CX_ERR_SHAPE when the fused kernel does not cover the shape (I % 256, K % 64) -> run the GEMM and cx_swiglu_bwd.
int cx_gemm_bf16_swiglu_bwd(const uint16_t* dY, const uint16_t* W, const uint16_t* YG, uint16_t* dYG, int M, int I, int K,
                            int ldx, int ldw, int ld_yg, void* stream) {
    if (M <= 0 || I <= 0) return CX_OK;
    if (!dY || !W || !YG || !dYG) return CX_ERR_ARG;
    if (K <= 0 || (K % 64) != 0 || (I % 256) != 0 || (ld_yg % 8) != 0 || ld_yg < 2 * I) return CX_ERR_SHAPE;
    if ((ldx % 8) != 0 || (ldw % 8) != 0) return CX_ERR_ALIGN;
    GemmParams p = base_params(dY, W, dYG, nullptr, M, I, K, ldx, ldw, ld_yg);
    p.Out2 = const_cast<uint16_t*>(YG); p.ldo2 = ld_yg;
    ProfScope prof(2.0 * (double)M * (double)I * (double)K, (hipStream_t)stream);
    return cx_launch_gemm_v6(p, GEMM_EPI_SWIGLU_BWD, (hipStream_t)stream) == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
}

// The same from the (activation, gate) pair cx_gemm_bf16_swiglu_gate leaves: Act, G: (M, I) bf16, leading dimension ld_ag.
int cx_gemm_bf16_swiglu_bwd_gate(const uint16_t* dY, const uint16_t* W, const uint16_t* Act, const uint16_t* G, uint16_t* dYG,
                                 int M, int I, int K, int ldx, int ldw, int ld_ag, int ld_dyg, void* stream) {
    if (M <= 0 || I <= 0) return CX_OK;
    if (!dY || !W || !Act || !G || !dYG) return CX_ERR_ARG;
    if (K <= 0 || (K % 64) != 0 || (I % 256) != 0 || (ld_dyg % 8) != 0 || ld_dyg < 2 * I || (ld_ag % 8) != 0 || ld_ag < I)
        return CX_ERR_SHAPE;
    if ((ldx % 8) != 0 || (ldw % 8) != 0) return CX_ERR_ALIGN;
    GemmParams p = base_params(dY, W, dYG, nullptr, M, I, K, ldx, ldw, ld_dyg);
    p.Out2 = const_cast<uint16_t*>(Act); p.ldo2 = ld_ag;
    p.In3 = G;
    ProfScope prof(2.0 * (double)M * (double)I * (double)K, (hipStream_t)stream);
    return cx_launch_gemm_v6(p, GEMM_EPI_SWIGLU_BWD_AG, (hipStream_t)stream) == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
}

// fc2 dgrad of the plain (GELU / quick_gelu) MLP with the activation backward fused into the epilogue (round 6; the reference gets the
// pair from flash_attn.ops.fused_dense.FusedMLP, sc/layers/mlp.py:30-34): dPre (M, N) bf16 = bf16(bf16(dY W^T) * act'(Pre)), Pre: the
// biased pre-activation the forward saved (cx_gemm_bf16_bias_act), W: (N, K) the transposed fc2 shadow.  Bit-identical to
// cx_gemm_bf16_nt + cx_bias_act_bwd_colsum(bias = NULL) without the (M, N) d(act) round trip.  dbias (optional): fp32 [N] += column sums
// of the bf16 dPre (the fc1 bias gradient), through `ws` (>= ceil(M / 128) * N floats) and a fixed-order reduction: deterministic.
// CX_ERR_SHAPE = not covered (K % 64, N % 8, leading dimensions % 8, workspace too small): run the two kernels instead.
int cx_gemm_bf16_act_bwd(const uint16_t* dY, const uint16_t* W, const uint16_t* Pre, uint16_t* dPre, float* dbias, float* ws,
                         long ws_floats, int M, int N, int K, int ldx, int ldw, int ld_pre, int ld_dpre, int act, void* stream) {
    if (M <= 0 || N <= 0) return CX_OK;
    if (!dY || !W || !Pre || !dPre || (act != 0 && act != 1)) return CX_ERR_ARG;
    if (K <= 0 || (K % 64) != 0 || (N % 8) != 0 || (ld_pre % 8) != 0 || (ld_dpre % 8) != 0) return CX_ERR_SHAPE;
    if ((ldx % 8) != 0 || (ldw % 8) != 0) return CX_ERR_ALIGN;
    const int nblocks = (M + 127) / 128;
    if (dbias && (!ws || ws_floats < (long)nblocks * N)) return CX_ERR_SHAPE;
    GemmParams p = base_params(dY, W, dPre, nullptr, M, N, K, ldx, ldw, ld_dpre);
    p.Out2 = const_cast<uint16_t*>(Pre); p.ldo2 = ld_pre;
    p.act = act;
    p.colsum_part = dbias ? ws : nullptr;
    {
        ProfScope prof(2.0 * (double)M * (double)N * (double)K, (hipStream_t)stream);
        if (cx_launch_gemm_v6(p, GEMM_EPI_ACT_BWD, (hipStream_t)stream) != hipSuccess) return CX_ERR_LAUNCH;
    }
    if (dbias && cx_launch_colsum_part_reduce(ws, dbias, nblocks, N, (hipStream_t)stream) != hipSuccess) return CX_ERR_LAUNCH;
    return CX_OK;
}

int cx_prof_gemm_config(int enable, int stride) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.enabled = enable != 0;
    g_prof.stride = stride > 0 ? stride : 1;
    g_prof.launches = 0;
    g_prof.used = 0;
    return CX_OK;
}

int cx_prof_gemm_collect(double* total_ms, double* total_flop, long* launches_timed, long* launches_total) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    double ms = 0.0, fl = 0.0;
    for (int i = 0; i < g_prof.used; ++i) {
        if (hipEventSynchronize(g_prof.ev1[i]) != hipSuccess) return CX_ERR_LAUNCH;
        float t = 0.f;
        if (hipEventElapsedTime(&t, g_prof.ev0[i], g_prof.ev1[i]) != hipSuccess) return CX_ERR_LAUNCH;
        ms += t;
        fl += g_prof.flop[i];
    }
    if (total_ms) *total_ms = ms;
    if (total_flop) *total_flop = fl;
    if (launches_timed) *launches_timed = g_prof.used;
    if (launches_total) *launches_total = g_prof.launches;
    return CX_OK;
}

}  // extern "C"

#include "gemm_splitk_small.inc"
