// infonce.hip -- K13 of SURVEY.md §2b: the InfoNCE / CLIP loss of sc/loss.py:76-132 as ONE similarity GEMM fused
// with the row-softmax cross-entropy; the (N x G) logits are never written to memory in the forward pass.
//
// The contraction runs on the exact-fp32 matrix-core path (v_mfma_f32_32x32x2_f32: bit-for-bit an fmaf chain,
// guide §3) because the loss is < 0.1 % of the step's FLOPs -- spending 1/16 of the bf16 rate
// here buys parity with the fp32 oracle at the 1e-6 level instead of bf16-logit noise (quirk 5, Appendix A).
// MFMA is issued as (A := document rows, B := query rows) so a lane owns ONE query row: the online
// log-sum-exp over documents is lane-local.
#include "cx_common.h"
#include "../../include/contrastors_hip.h"

namespace {

constexpr int SBM = 128, SBN = 128, SBK = 16;
constexpr int STILE = 128 * SBK * 4;  // 8 KiB
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

// [128 rows][16 f32]: 4 chunks of 16 B per 64-B row; chunk c of row r at r*64 + ((c ^ ((r>>2)&3)) << 4)
CX_DEVICE int stile_off(int r, int c) { return r * 64 + ((c ^ ((r >> 2) & 3)) << 4); }

enum Epi { EPI_STORE = 0, EPI_LSE = 1, EPI_GRAD = 2 };

struct SgemmParams {
    const float* A;  // (M,K) lda  -- "query side": lane dimension of the result
    const float* B;  // (N,K) ldb  -- "document side": register dimension of the result
    int M, N, K, lda, ldb;
    int tiles_m, tiles_n;
    // EPI_STORE
    float* C; int ldc;
    // EPI_LSE / EPI_GRAD
    const int64_t* labels;
    float scale;
    float* pmax; float* psum; float* lab;  // workspace views (EPI_LSE)
    int* pidx;                             // EPI_LSE, optional: first column of each 64-column slice that attains its maximum
    int nparts;
    const float* lse; float coef; float* Gm; float* GmT; float* dscale;  // EPI_GRAD
};

struct SRegs { float4 r0, r1; };  // named members, not an array (keeps them in VGPRs)

CX_DEVICE float4 sload(const float* __restrict__ base, int ld, int row0, int nrows, int k0, int K, int item) {
    const int r = item >> 2, c = item & 3;
    int gr = row0 + r;
    gr = gr < nrows ? gr : nrows - 1;
    const int k = k0 + c * 4;
    if (k >= K) return make_float4(0.f, 0.f, 0.f, 0.f);  // K % 4 == 0: a chunk is all-in or all-out
    return *reinterpret_cast<const float4*>(base + (size_t)gr * ld + k);
}
CX_DEVICE void sstage(const float* __restrict__ base, int ld, int row0, int nrows, int k0, int K, int tid,
                      SRegs& regs) {
    regs.r0 = sload(base, ld, row0, nrows, k0, K, tid);
    regs.r1 = sload(base, ld, row0, nrows, k0, K, 256 + tid);
}
CX_DEVICE void scommit(char* tile, int tid, const SRegs& regs) {
    *reinterpret_cast<float4*>(tile + stile_off(tid >> 2, tid & 3)) = regs.r0;
    *reinterpret_cast<float4*>(tile + stile_off((256 + tid) >> 2, tid & 3)) = regs.r1;
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void sgemm_nt_kernel(SgemmParams p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * STILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const int nwg = p.tiles_m * p.tiles_n;
    const int lid = xcd_remap(blockIdx.x, nwg);
    const int tn = lid % p.tiles_n, tm = lid / p.tiles_n;
    const int m0 = tm * SBM, n0 = tn * SBN;
    const int nk = (p.K + SBK - 1) / SBK;

    f32x16_t acc[2][2];  // [n-block a][m-block b]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    SRegs ar = {}, br = {};
    sstage(p.A, p.lda, m0, p.M, 0, p.K, tid, ar);
    sstage(p.B, p.ldb, n0, p.N, 0, p.K, tid, br);
    scommit(smem, tid, ar);
    scommit(smem + STILE, tid, br);
    __syncthreads();

    for (int it = 0; it < nk; ++it) {
        char* cur = smem + (it & 1) * 2 * STILE;
        char* nxt = smem + ((it + 1) & 1) * 2 * STILE;
        const bool more = (it + 1) < nk;
        if (more) {
            sstage(p.A, p.lda, m0, p.M, (it + 1) * SBK, p.K, tid, ar);
            sstage(p.B, p.ldb, n0, p.N, (it + 1) * SBK, p.K, tid, br);
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            // lane-half hi reads k = 8g + 4hi + {0..3}; MFMA e consumes element e of both operands, i.e. the
            // reduction pair {8g+e, 8g+4+e}: a permutation of k shared by A and B.
            float4 af[2], bf[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                af[b] = *reinterpret_cast<const float4*>(cur + stile_off(wm * 64 + b * 32 + l31, 2 * g + hi));
                bf[b] = *reinterpret_cast<const float4*>(cur + STILE + stile_off(wn * 64 + b * 32 + l31, 2 * g + hi));
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[a].x, af[b].x, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[a].y, af[b].y, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[a].z, af[b].z, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[a].w, af[b].w, acc[a][b], 0, 0, 0);
                }
        }
        if (more) {
            scommit(nxt, tid, ar);
            scommit(nxt + STILE, tid, br);
        }
        __syncthreads();
    }

    // acc[a][b][r] = sum_k A[m][k] B[n][k],  m = m0 + wm*64 + b*32 + l31,  n = n0 + wn*64 + a*32 + acc_row(r,hi)
    if constexpr (EPI == EPI_STORE) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int m = m0 + wm * 64 + b * 32 + l31;
            if (m >= p.M) continue;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = n0 + wn * 64 + a * 32 + 8 * q + 4 * hi;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < p.N) p.C[(size_t)m * p.ldc + n + e] = acc[a][b][4 * q + e];
                }
        }
    } else if constexpr (EPI == EPI_LSE) {
        const float sc2 = p.scale * LOG2E;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int m = m0 + wm * 64 + b * 32 + l31;
            const int mc = m < p.M ? m : p.M - 1;
            const long label = p.labels[mc];
            float mx = -INFINITY;
            int amx = 0x7fffffff;   // this lane's first column with v == mx (n grows with a, then r: strict > keeps the first)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = n0 + wn * 64 + a * 32 + acc_row(r, hi);
                    const float v = n < p.N ? acc[a][b][r] * sc2 : -INFINITY;
                    if (n == label && m < p.M) p.lab[m] = acc[a][b][r] * p.scale;  // exactly one lane owns it
                    acc[a][b][r] = v;
                    if (v > mx) amx = n;
                    mx = fmaxf(mx, v);
                }
            {
                const float omx = __shfl_xor(mx, 32, 64);
                const int oamx = __shfl_xor(amx, 32, 64);
                if (omx > mx || (omx == mx && oamx < amx)) amx = oamx;   // torch.argmax: the first index among equal maxima
                mx = fmaxf(mx, omx);
            }
            float sm = 0.f;
            if (mx > -INFINITY) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sm += exp2f(acc[a][b][r] - mx);
            }
            sm += __shfl_xor(sm, 32, 64);
            if (hi == 0 && m < p.M) {
                const int part = tn * 2 + wn;
                p.pmax[(size_t)m * p.nparts + part] = mx;   // log2 units
                p.psum[(size_t)m * p.nparts + part] = sm;
                if (p.pidx) p.pidx[(size_t)m * p.nparts + part] = amx;
            }
        }
    } else {
        const float sc2 = p.scale * LOG2E;
        float dsc = 0.f;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int m = m0 + wm * 64 + b * 32 + l31;
            const bool m_ok = m < p.M;
            const int mc = m_ok ? m : p.M - 1;
            const long label = p.labels[mc];
            const float lse2 = p.lse[mc] * LOG2E;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = n0 + wn * 64 + a * 32 + 8 * q + 4 * hi;
                    float g[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float s = acc[a][b][4 * q + e];
                        const float pr = exp2f(s * sc2 - lse2) - ((n + e) == label ? 1.f : 0.f);
                        const bool ok = m_ok && (n + e) < p.N;
                        g[e] = ok ? pr * p.coef * p.scale : 0.f;
                        dsc += ok ? pr * s * p.coef : 0.f;
                        if (ok) p.GmT[(size_t)(n + e) * p.M + m] = g[e];
                    }
                    if (m_ok) {
                        if (n + 3 < p.N) {
                            *reinterpret_cast<float4*>(p.Gm + (size_t)m * p.N + n) = make_float4(g[0], g[1], g[2], g[3]);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (n + e < p.N) p.Gm[(size_t)m * p.N + n + e] = g[e];
                        }
                    }
                }
        }
        if (p.dscale) {
            dsc = wave_sum(dsc);
            if (lane == 0) unsafeAtomicAdd(p.dscale, dsc);
        }
    }
}

// one wave per row: fold the per-column-slice partials into lse (natural log) and the per-row loss
// pidx / argmax (optional): the row's arg max over all G columns = the smallest recorded column among the slices whose maximum
// is the row maximum (slices are disjoint column ranges, each recorded its FIRST maximal column) -- what
// `similarity.argmax(dim=1)` of sc/loss.py:127-130 returns, without the similarity matrix.
__global__ __launch_bounds__(256) void lse_combine_kernel(const float* __restrict__ pmax,
                                                          const float* __restrict__ psum,
                                                          const float* __restrict__ lab, float* __restrict__ lse,
                                                          float* __restrict__ loss_rows, int N, int nparts,
                                                          const int* __restrict__ pidx, int* __restrict__ argmax) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    float mx = -INFINITY;
    for (int i = lane; i < nparts; i += 64) mx = fmaxf(mx, pmax[(size_t)row * nparts + i]);
    mx = wave_max(mx);
    if (pidx && argmax) {
        int am = 0x7fffffff;
        for (int i = lane; i < nparts; i += 64)
            if (pmax[(size_t)row * nparts + i] == mx) am = min(am, pidx[(size_t)row * nparts + i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) am = min(am, __shfl_xor(am, o, 64));
        if (lane == 0) argmax[row] = am;
    }
    float s = 0.f;
    for (int i = lane; i < nparts; i += 64) {
        const float pm = pmax[(size_t)row * nparts + i];
        if (pm > -INFINITY) s += psum[(size_t)row * nparts + i] * exp2f(pm - mx);
    }
    s = wave_sum(s);
    if (lane == 0) {
        const float l = (mx + log2f(s)) * LN2;
        lse[row] = l;
        loss_rows[row] = l - lab[row];
    }
}

template <int EPI>
int launch(const SgemmParams& p, hipStream_t s) {
    hipLaunchKernelGGL((sgemm_nt_kernel<EPI>), dim3(p.tiles_m * p.tiles_n), dim3(256), 0, s, p);
    return hipGetLastError() == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
}

int check_common(int M, int N, int K, int lda, int ldb) {
    if (K <= 0 || (K % 4) != 0) return CX_ERR_SHAPE;
    if ((lda % 4) != 0 || (ldb % 4) != 0) return CX_ERR_ALIGN;
    (void)M; (void)N;
    return CX_OK;
}

}  // namespace

extern "C" {

int cx_sgemm_nt(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                void* stream) {
    if (M <= 0 || N <= 0) return CX_OK;
    int rc = check_common(M, N, K, lda, ldb);
    if (rc != CX_OK) return rc;
    SgemmParams p = {};
    p.A = A; p.B = B; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb;
    p.tiles_m = (M + SBM - 1) / SBM; p.tiles_n = (N + SBN - 1) / SBN;
    p.C = C; p.ldc = ldc;
    return launch<EPI_STORE>(p, (hipStream_t)stream);
}

long cx_infonce_ws_floats(int N, int G) {
    const long nparts = 2L * ((G + SBN - 1) / SBN);
    return (long)N * (2 * nparts + 1);
}
long cx_infonce_argmax_ws_floats(int N, int G) {
    const long nparts = 2L * ((G + SBN - 1) / SBN);
    return (long)N * (3 * nparts + 1);
}

int cx_infonce_fwd(const float* Q, const float* D, const int64_t* labels, float scale, float* ws, float* lse,
                   float* loss_rows, int N, int G, int dim, int ldq, int ldd, void* stream) {
    return cx_infonce_fwd_argmax(Q, D, labels, scale, ws, lse, loss_rows, nullptr, N, G, dim, ldq, ldd, stream);
}

int cx_infonce_fwd_argmax(const float* Q, const float* D, const int64_t* labels, float scale, float* ws, float* lse,
                          float* loss_rows, int32_t* argmax, int N, int G, int dim, int ldq, int ldd, void* stream) {
    if (N <= 0 || G <= 0) return CX_OK;
    if (!Q || !D || !labels || !ws || !lse || !loss_rows) return CX_ERR_ARG;
    int rc = check_common(N, G, dim, ldq, ldd);
    if (rc != CX_OK) return rc;
    SgemmParams p = {};
    p.A = Q; p.B = D; p.M = N; p.N = G; p.K = dim; p.lda = ldq; p.ldb = ldd;
    p.tiles_m = (N + SBM - 1) / SBM; p.tiles_n = (G + SBN - 1) / SBN;
    p.labels = labels; p.scale = scale;
    p.nparts = 2 * p.tiles_n;
    p.pmax = ws;
    p.psum = ws + (size_t)N * p.nparts;
    p.lab = ws + (size_t)2 * N * p.nparts;
    p.pidx = argmax ? reinterpret_cast<int*>(ws + (size_t)2 * N * p.nparts + N) : nullptr;   // (ws: cx_infonce_argmax_ws_floats)
    rc = launch<EPI_LSE>(p, (hipStream_t)stream);
    if (rc != CX_OK) return rc;
    hipLaunchKernelGGL(lse_combine_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, p.pmax, p.psum,
                       p.lab, lse, loss_rows, N, p.nparts, p.pidx, argmax);
    return hipGetLastError() == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
}

int cx_infonce_bwd(const float* Q, const float* D, const int64_t* labels, const float* lse, float scale, float coef,
                   float* Gmat, float* GmatT, float* QT, float* DT, float* dQ, float* dD, float* dscale_accum,
                   int N, int G, int dim, int ldq, int ldd, void* stream) {
    if (N <= 0 || G <= 0) return CX_OK;
    if (!Q || !D || !labels || !lse || !Gmat || !GmatT || !QT || !DT || !dQ || !dD) return CX_ERR_ARG;
    int rc = check_common(N, G, dim, ldq, ldd);
    if (rc != CX_OK) return rc;
    if ((N % 4) != 0 || (G % 4) != 0) return CX_ERR_SHAPE;  // they are the K (and ld) of the two output GEMMs
    SgemmParams p = {};
    p.A = Q; p.B = D; p.M = N; p.N = G; p.K = dim; p.lda = ldq; p.ldb = ldd;
    p.tiles_m = (N + SBM - 1) / SBM; p.tiles_n = (G + SBN - 1) / SBN;
    p.labels = labels; p.scale = scale; p.lse = lse; p.coef = coef;
    p.Gm = Gmat; p.GmT = GmatT; p.dscale = dscale_accum;
    rc = launch<EPI_GRAD>(p, (hipStream_t)stream);
    if (rc != CX_OK) return rc;
    rc = cx_transpose_f32(Q, QT, N, dim, ldq, N, stream);
    if (rc != CX_OK) return rc;
    rc = cx_transpose_f32(D, DT, G, dim, ldd, G, stream);
    if (rc != CX_OK) return rc;
    rc = cx_sgemm_nt(Gmat, DT, dQ, N, dim, G, G, G, dim, stream);   // dQ[m][c] = sum_n Gm[m][n] D[n][c]
    if (rc != CX_OK) return rc;
    return cx_sgemm_nt(GmatT, QT, dD, G, dim, N, N, N, dim, stream); // dD[n][c] = sum_m Gm[m][n] Q[m][c]
}

}  // extern "C"
