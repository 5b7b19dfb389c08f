// infonce_fp8.hip -- the InfoNCE similarity GEMM on the fp8 matrix-core path (BASELINE.json configs[4]: CLIP-style
// joint ViT-B/16 + BERT-base, global batch 32768, "fp8 MFMA similarity GEMM"; the reference only carries the flag,
// configs/train/contrastive_pretrain.yaml:24 `use_fp8`, no code -- parity is against the fp32 oracle with a stated
// fp8 tolerance).
//
//   logits[i][j] = scale * <q_i, d_j>,   q, d L2-normalised fp32 rows of width dim (sc/loss.py:100-117)
//
// Rows are quantised to OCP e4m3 with ONE scale per row (amax / 448), the contraction runs on
// v_mfma_scale_f32_32x32x64_f8f6f4 (2x the bf16 rate) with all block scales 2^0, and the two row scales, the logit
// scale and log2(e) are applied to the fp32 accumulator.  MFMA is issued as (A := document rows, B := query rows), so
// a lane owns ONE query and the online log-sum-exp over documents is lane-local, as in the exact-fp32 kernel
// (infonce.hip).  The (N x G) logits are never written.  Backward recomputes the tile, turns it into
// coef * scale * (softmax - onehot) and writes it ONCE, transposed and in bf16 (GmT: G x N); the two output products
// run on the bf16 GEMM family: dD = GmT Q (NT form) and dQ = GmT^T D (the natural-layout wgrad form, no transposes).
#include "cx_common.h"
#include "../../include/contrastors_hip.h"

namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8_t;

constexpr float LOG2E = 1.4426950408889634f;
constexpr float FP8_MAX = 448.f;      // OCP e4m3fn
constexpr int TD = 64;                // documents per tile (two 32-row MFMA blocks)
constexpr int QW = 32;                // queries per wave
constexpr int QB = 128;               // queries per workgroup (4 waves)

enum Mode { MODE_LSE = 0, MODE_GRAD = 1 };

// ---- row quantisation: one wave per row, dim % 256 == 0, dim <= 1024 -------------------------------------------------
template <int NCH>  // dim = NCH * 256: every lane converts 4 consecutive values per 256-column chunk
__global__ __launch_bounds__(256) void quantize_rows_fp8_kernel(const float* __restrict__ X, int ldx, uint8_t* __restrict__ X8,
                                                                float* __restrict__ scale, uint16_t* __restrict__ Xb, int rows) {
    constexpr int D = NCH * 256;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v[NCH][4];
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const float4 u = *reinterpret_cast<const float4*>(X + (size_t)row * ldx + (i * 64 + lane) * 4);
        v[i][0] = u.x; v[i][1] = u.y; v[i][2] = u.z; v[i][3] = u.w;
#pragma unroll
        for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fabsf(v[i][e]));
    }
    amax = wave_max(amax);
    const float sc = amax > 0.f ? amax / FP8_MAX : 1.f;
    const float inv = 1.f / sc;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        int w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][0] * inv, v[i][1] * inv, w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][2] * inv, v[i][3] * inv, w, true);
        *reinterpret_cast<int*>(X8 + (size_t)row * D + (i * 64 + lane) * 4) = w;
        if (Xb) {  // bf16 copy for the backward's output GEMMs
            uint2 pk;
            pk.x = pack_bf16x2(v[i][0], v[i][1]);
            pk.y = pack_bf16x2(v[i][2], v[i][3]);
            *reinterpret_cast<uint2*>(Xb + (size_t)row * D + (i * 64 + lane) * 4) = pk;
        }
    }
    if (lane == 0) scale[row] = sc;
}

struct Fp8Params {
    const uint8_t* Q8; const uint8_t* D8;   // (N, dim), (G, dim) e4m3
    const float* sq; const float* sd;       // row scales
    const int64_t* labels;
    int N, G, dim;
    int nsplit;                             // column splits (grid.y); each covers tiles_per_split document tiles
    int tiles_per_split;
    float scale;
    // MODE_LSE
    float* pmax; float* psum; float* lab;
    // MODE_GRAD
    const float* lse; float coef; uint16_t* GmT; float* dscale;
};

// LDS document tile: [TD rows][dim + 16 bytes]; the 16-byte pad puts consecutive rows 4 banks apart, which makes the
// 16-lane service groups of ds_read_b128 conflict free (rows distinct mod 16 inside a group).
template <int KS>  // KS = dim / 64 k-steps
__global__ __launch_bounds__(256, 1) void infonce_fp8_kernel(Fp8Params p, int mode) {
    constexpr int DIM = KS * 64;
    constexpr int PITCH = DIM + 16;
    constexpr int CH = DIM / 16;                   // 16-byte chunks per row
    constexpr int LOADS = TD * CH / 256;           // chunks per thread and tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    auto tile_at = [&](int buf) { return smem + buf * (TD * PITCH); };
    float* sd_lds = reinterpret_cast<float*>(smem + 2 * TD * PITCH);  // [2][TD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    const int q0 = blockIdx.x * QB + wave * QW;
    const int qi = q0 + l31;
    const int qc = qi < p.N ? qi : p.N - 1;
    const int t_begin = blockIdx.y * p.tiles_per_split;
    int t_end = t_begin + p.tiles_per_split;
    const int ntiles = (p.G + TD - 1) / TD;
    if (t_end > ntiles) t_end = ntiles;
    if (t_begin >= t_end) return;

    // ---- the wave's query fragments stay in registers: lane (l31, hi) holds bytes [64 ks + 32 hi, +32) of query l31 -----
    i32x8_t qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const uint4 a = *reinterpret_cast<const uint4*>(p.Q8 + (size_t)qc * DIM + ks * 64 + hi * 32);
        const uint4 b = *reinterpret_cast<const uint4*>(p.Q8 + (size_t)qc * DIM + ks * 64 + hi * 32 + 16);
        qf[ks] = i32x8_t{(int)a.x, (int)a.y, (int)a.z, (int)a.w, (int)b.x, (int)b.y, (int)b.z, (int)b.w};
    }
    const float qs2 = p.sq[qc] * p.scale * LOG2E;   // accumulator -> log2-domain logit (times the document scale)
    const float qs1 = p.sq[qc];                      // accumulator -> cosine (times the document scale)
    const long label = p.labels[qc];
    const float lse2 = mode == MODE_GRAD ? p.lse[qc] * LOG2E : 0.f;

    // ---- tile staging: global -> registers -> LDS --------------------------------------------------------------------
    // (named registers + macros, not an array behind lambdas: that one is left in scratch memory by the compiler)
    uint4 s0 = {}, s1 = {}, s2 = {}, s3 = {}, s4 = {}, s5 = {}, s6 = {}, s7 = {}, s8 = {}, s9 = {}, s10 = {}, s11 = {}, s12 = {},
          s13 = {}, s14 = {}, s15 = {};
    float sdv = 0.f;
#define CX_F1(i, t_)                                                                                     \
    if constexpr ((i) < LOADS) {                                                                         \
        const int item_ = (i) * 256 + tid;                                                               \
        const int r_ = item_ / CH, c_ = item_ - r_ * CH;                                                 \
        int g_ = (t_) * TD + r_;                                                                         \
        g_ = g_ < p.G ? g_ : p.G - 1;                                                                    \
        s##i = *reinterpret_cast<const uint4*>(p.D8 + (size_t)g_ * DIM + c_ * 16);                       \
    }
#define CX_FETCH(t_)                                                                                     \
    {                                                                                                    \
        CX_F1(0, t_) CX_F1(1, t_) CX_F1(2, t_) CX_F1(3, t_) CX_F1(4, t_) CX_F1(5, t_) CX_F1(6, t_) CX_F1(7, t_)          \
        CX_F1(8, t_) CX_F1(9, t_) CX_F1(10, t_) CX_F1(11, t_) CX_F1(12, t_) CX_F1(13, t_) CX_F1(14, t_) CX_F1(15, t_)    \
        if (tid < TD) {                                                                                  \
            const int g_ = (t_) * TD + tid;                                                              \
            sdv = g_ < p.G ? p.sd[g_] : 0.f;                                                             \
        }                                                                                                \
    }
#define CX_C1(i, buf_)                                                                                   \
    if constexpr ((i) < LOADS) {                                                                         \
        const int item_ = (i) * 256 + tid;                                                               \
        const int r_ = item_ / CH, c_ = item_ - r_ * CH;                                                 \
        *reinterpret_cast<uint4*>(tile_at(buf_) + r_ * PITCH + c_ * 16) = s##i;                          \
    }
#define CX_COMMIT(buf_)                                                                                  \
    {                                                                                                    \
        CX_C1(0, buf_) CX_C1(1, buf_) CX_C1(2, buf_) CX_C1(3, buf_) CX_C1(4, buf_) CX_C1(5, buf_) CX_C1(6, buf_) CX_C1(7, buf_)      \
        CX_C1(8, buf_) CX_C1(9, buf_) CX_C1(10, buf_) CX_C1(11, buf_) CX_C1(12, buf_) CX_C1(13, buf_) CX_C1(14, buf_) CX_C1(15, buf_) \
        if (tid < TD) sd_lds[(buf_) * TD + tid] = sdv;                                                   \
    }

    float run_max = -INFINITY, run_sum = 0.f, lab_logit = 0.f, dsc = 0.f;
    bool have_lab = false;
    CX_FETCH(t_begin)
    CX_COMMIT(0)
    __syncthreads();
    for (int t = t_begin; t < t_end; ++t) {
        const int buf = (t - t_begin) & 1;
        if (t + 1 < t_end) CX_FETCH(t + 1)
        f32x16_t acc[2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
            const char* rowp = tile_at(buf) + (rb * 32 + l31) * PITCH + hi * 32;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const uint4 a = *reinterpret_cast<const uint4*>(rowp + ks * 64);
                const uint4 b = *reinterpret_cast<const uint4*>(rowp + ks * 64 + 16);
                const i32x8_t df = {(int)a.x, (int)a.y, (int)a.z, (int)a.w, (int)b.x, (int)b.y, (int)b.z, (int)b.w};
                // A = documents (rows of the result), B = queries (columns = lanes); e4m3 x e4m3, block scales 2^0
                acc[rb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(df, qf[ks], acc[rb], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            }
        }
        // acc[rb][r]: document t*TD + rb*32 + acc_row(r, hi), query q0 + l31
        const float* sdt = sd_lds + buf * TD;
        if (mode == MODE_LSE) {
            float tmax = -INFINITY;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 s4 = *reinterpret_cast<const float4*>(sdt + rb * 32 + 8 * q + 4 * hi);
                    const float sv[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int doc = t * TD + rb * 32 + 8 * q + 4 * hi + e;
                        const float v = doc < p.G ? acc[rb][4 * q + e] * sv[e] * qs2 : -INFINITY;
                        if (doc == label) { lab_logit = v; have_lab = true; }
                        acc[rb][4 * q + e] = v;
                        tmax = fmaxf(tmax, v);
                    }
                }
            const float nmax = fmaxf(run_max, tmax);
            if (nmax > -INFINITY) {
                float s = run_sum * exp2f(run_max - nmax);
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s += exp2f(acc[rb][r] - nmax);
                run_sum = s;
                run_max = nmax;
            }
        } else {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 s4 = *reinterpret_cast<const float4*>(sdt + rb * 32 + 8 * q + 4 * hi);
                    const float sv[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int doc = t * TD + rb * 32 + 8 * q + 4 * hi + e;
                        const float cosv = acc[rb][4 * q + e] * sv[e] * qs1;
                        const float pr = exp2f(acc[rb][4 * q + e] * sv[e] * qs2 - lse2) - (doc == label ? 1.f : 0.f);
                        const bool ok = doc < p.G && qi < p.N;
                        if (ok) {
                            dsc += pr * cosv * p.coef;
                            // GmT[doc][query]: the 32 lanes of a half-wave write 64 contiguous bytes
                            p.GmT[(size_t)doc * p.N + qi] = f32_to_bf16(pr * p.coef * p.scale);
                        }
                    }
                }
        }
        if (t + 1 < t_end) CX_COMMIT(buf ^ 1)
        __syncthreads();
    }
    if (mode == MODE_LSE) {
        // the two half-waves hold disjoint document subsets of the same query: fold them
        const float omax = __shfl_xor(run_max, 32, 64), osum = __shfl_xor(run_sum, 32, 64);
        const float m = fmaxf(run_max, omax);
        float s = 0.f;
        if (m > -INFINITY) s = run_sum * exp2f(run_max - m) + osum * exp2f(omax - m);
        if (qi < p.N) {
            if (hi == 0) {
                p.pmax[(size_t)qi * p.nsplit + blockIdx.y] = m;   // log2 units, as lse_combine expects
                p.psum[(size_t)qi * p.nsplit + blockIdx.y] = s;
            }
            if (have_lab) p.lab[qi] = lab_logit * (1.f / LOG2E);  // natural-log units: scale * cos
        }
    } else if (p.dscale) {
        dsc = wave_sum(dsc);
        if (lane == 0) unsafeAtomicAdd(p.dscale, dsc);
    }
}

// one wave per row: fold the per-split partials into lse (natural log) and the per-row loss (same contract as
// lse_combine_kernel of infonce.hip)
__global__ __launch_bounds__(256) void lse_combine_fp8_kernel(const float* __restrict__ pmax, const float* __restrict__ psum,
                                                              const float* __restrict__ lab, float* __restrict__ lse,
                                                              float* __restrict__ loss_rows, int N, int nparts) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    float mx = -INFINITY;
    for (int i = lane; i < nparts; i += 64) mx = fmaxf(mx, pmax[(size_t)row * nparts + i]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int i = lane; i < nparts; i += 64) {
        const float pm = pmax[(size_t)row * nparts + i];
        if (pm > -INFINITY) s += psum[(size_t)row * nparts + i] * exp2f(pm - mx);
    }
    s = wave_sum(s);
    if (lane == 0) {
        const float l = (mx + log2f(s)) * 0.6931471805599453f;
        lse[row] = l;
        loss_rows[row] = l - lab[row];
    }
}

// column splits: ~2 workgroups per CU, every split non-empty (each writes its partial)
void splits_for(int N, int G, int* nsplit, int* tiles_per_split) {
    const int qblocks = (N + QB - 1) / QB, ntiles = (G + TD - 1) / TD;
    int ns = (512 + qblocks - 1) / qblocks;
    if (ns > ntiles) ns = ntiles;
    if (ns > 64) ns = 64;
    if (ns < 1) ns = 1;
    const int per = (ntiles + ns - 1) / ns;
    *tiles_per_split = per;
    *nsplit = (ntiles + per - 1) / per;
}

// strided fp32 rows -> contiguous bf16 rows (operands of the backward's output GEMMs)
__global__ __launch_bounds__(256) void cast_rows_bf16_kernel(const float* __restrict__ X, int ldx, uint16_t* __restrict__ Xb,
                                                             int rows, int dim) {
    const int per_row = dim / 4;
    const long n = (long)rows * per_row;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int r = (int)(i / per_row), c = (int)(i - (long)r * per_row);
        const float4 u = *reinterpret_cast<const float4*>(X + (size_t)r * ldx + c * 4);
        uint2 pk;
        pk.x = pack_bf16x2(u.x, u.y);
        pk.y = pack_bf16x2(u.z, u.w);
        *reinterpret_cast<uint2*>(Xb + (size_t)r * dim + c * 4) = pk;
    }
}
int cast_rows(const float* X, int ldx, uint16_t* Xb, int rows, int dim, hipStream_t s) {
    long g = ((long)rows * (dim / 4) + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(cast_rows_bf16_kernel, dim3((int)g), dim3(256), 0, s, X, ldx, Xb, rows, dim);
    return hipGetLastError() == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
}

int quantize(const float* X, int ldx, uint8_t* X8, float* sc, uint16_t* Xb, int rows, int dim, hipStream_t s) {
    const dim3 grid((rows + 3) / 4), block(256);
    switch (dim) {
        case 256: hipLaunchKernelGGL((quantize_rows_fp8_kernel<1>), grid, block, 0, s, X, ldx, X8, sc, Xb, rows); break;
        case 512: hipLaunchKernelGGL((quantize_rows_fp8_kernel<2>), grid, block, 0, s, X, ldx, X8, sc, Xb, rows); break;
        case 768: hipLaunchKernelGGL((quantize_rows_fp8_kernel<3>), grid, block, 0, s, X, ldx, X8, sc, Xb, rows); break;
        case 1024: hipLaunchKernelGGL((quantize_rows_fp8_kernel<4>), grid, block, 0, s, X, ldx, X8, sc, Xb, rows); break;
        default: return CX_ERR_SHAPE;
    }
    return hipGetLastError() == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
}

template <int KS>
int launch_main(const Fp8Params& p, int mode, hipStream_t s) {
    constexpr int LDS = 2 * TD * (KS * 64 + 16) + 2 * TD * 4;
    static CxLdsOptIn lds;
    if (!lds.ensure(reinterpret_cast<const void*>(&infonce_fp8_kernel<KS>), LDS)) return CX_ERR_LAUNCH;
    hipLaunchKernelGGL((infonce_fp8_kernel<KS>), dim3((p.N + QB - 1) / QB, p.nsplit), dim3(256), LDS, s, p, mode);
    return hipGetLastError() == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
}

int launch_dim(const Fp8Params& p, int mode, hipStream_t s) {
    switch (p.dim) {
        case 256: return launch_main<4>(p, mode, s);
        case 512: return launch_main<8>(p, mode, s);
        case 768: return launch_main<12>(p, mode, s);
        case 1024: return launch_main<16>(p, mode, s);
        default: return CX_ERR_SHAPE;
    }
}

}  // namespace

extern "C" {

long cx_infonce_fp8_ws_floats(int N, int G) {
    int ns, per;
    splits_for(N, G, &ns, &per);
    return (long)N * (2L * ns + 1);
}

int cx_infonce_fp8_fwd(const float* Q, const float* D, const int64_t* labels, float scale, float* ws, uint8_t* Q8, uint8_t* D8,
                       float* sq, float* sd, float* lse, float* loss_rows, int N, int G, int dim, int ldq, int ldd,
                       void* stream) {
    if (N <= 0 || G <= 0) return CX_OK;
    if (!Q || !D || !labels || !ws || !Q8 || !D8 || !sq || !sd || !lse || !loss_rows) return CX_ERR_ARG;
    if ((ldq % 4) != 0 || (ldd % 4) != 0) return CX_ERR_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    int rc = quantize(Q, ldq, Q8, sq, nullptr, N, dim, s);
    if (rc != CX_OK) return rc;
    rc = quantize(D, ldd, D8, sd, nullptr, G, dim, s);
    if (rc != CX_OK) return rc;
    Fp8Params p = {};
    p.Q8 = Q8; p.D8 = D8; p.sq = sq; p.sd = sd; p.labels = labels; p.N = N; p.G = G; p.dim = dim; p.scale = scale;
    splits_for(N, G, &p.nsplit, &p.tiles_per_split);
    p.pmax = ws;
    p.psum = ws + (size_t)N * p.nsplit;
    p.lab = ws + (size_t)2 * N * p.nsplit;
    rc = launch_dim(p, MODE_LSE, s);
    if (rc != CX_OK) return rc;
    hipLaunchKernelGGL(lse_combine_fp8_kernel, dim3((N + 3) / 4), dim3(256), 0, s, p.pmax, p.psum, p.lab, lse, loss_rows, N,
                       p.nsplit);
    return hipGetLastError() == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
}

int cx_infonce_fp8_bwd(const float* Q, const float* D, const int64_t* labels, const float* lse, float scale, float coef,
                       const uint8_t* Q8, const uint8_t* D8, const float* sq, const float* sd, uint16_t* GmT, uint16_t* Qb,
                       uint16_t* QbT, uint16_t* Db, float* ws, long ws_floats, float* dQ, float* dD, float* dscale_accum,
                       int N, int G, int dim, int ldq, int ldd, void* stream) {
    if (N <= 0 || G <= 0) return CX_OK;
    if (!Q || !D || !labels || !lse || !Q8 || !D8 || !sq || !sd || !GmT || !Qb || !QbT || !Db || !ws || !dQ || !dD)
        return CX_ERR_ARG;
    // the two output products run on the bf16 GEMM family: dD = GmT Q needs K = N % 64 == 0; dQ = GmT^T D is the
    // natural-layout wgrad form (O = N % 256 == 0, I = dim % 256 == 0, K = G rows zero-padded to a multiple of 64)
    if ((N % 256) != 0 || (dim % 256) != 0 || (G % 64) != 0) return CX_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    Fp8Params p = {};
    p.Q8 = Q8; p.D8 = D8; p.sq = sq; p.sd = sd; p.labels = labels; p.N = N; p.G = G; p.dim = dim; p.scale = scale;
    splits_for(N, G, &p.nsplit, &p.tiles_per_split);
    p.lse = lse; p.coef = coef; p.GmT = GmT; p.dscale = dscale_accum;
    int rc = launch_dim(p, MODE_GRAD, s);
    if (rc != CX_OK) return rc;
    // bf16 copies of the fp32 embeddings for the output GEMMs (fp32 accumulation; the quantised rows are NOT reused here)
    rc = cast_rows(Q, ldq, Qb, N, dim, s);
    if (rc != CX_OK) return rc;
    rc = cast_rows(D, ldd, Db, G, dim, s);
    if (rc != CX_OK) return rc;
    rc = cx_transpose_bf16(Qb, QbT, N, dim, dim, N, N, stream);          // (dim, N)
    if (rc != CX_OK) return rc;
    rc = cx_gemm_bf16_nt(GmT, QbT, dD, nullptr, G, dim, N, N, N, dim, /*fp32 out*/ 1, 1, 1.f, stream);   // dD = GmT Q
    if (rc != CX_OK) return rc;
    if (hipMemsetAsync(dQ, 0, (size_t)N * dim * sizeof(float), s) != hipSuccess) return CX_ERR_LAUNCH;
    return cx_gemm_bf16_tn_accum(GmT, Db, dQ, ws, ws_floats, G, N, dim, N, dim, stream);                  // dQ = GmT^T D
}

}  // extern "C"
