// gemm_bf16_v5.hip -- 256x256x64 bf16 MFMA GEMM (K9), the tile shape the round-1 ablations asked for.
//
// Measured on v2 (256x128x64): the K loop is limited by the global->LDS request stream, which saturates near
// 45 GB/s per CU whatever the prefetch depth (DMA-only loop == full loop; 64-B segments of v3 gained nothing) -- i.e.
// by BYTES (128-B line requests) PER FLOP of the tile shape.  256x256x64 moves 64 KiB per 8.4 MFLOP instead of
// 48 KiB per 4.2 MFLOP: 0.67x the requests per FLOP, always as whole 128-B lines.
//
// Structure: 8 waves as 2(M) x 4(N), 128x64 per wave = 4x2 MFMA 32x32x16 accumulators (128 VGPRs); two 64 KiB LDS
// stages filled by LDS-DMA (global_load_lds, swizzle on the source address); the DMA of K-tile t+1 is issued in two
// halves between the MFMA groups of iteration t; fragments are read one k-step ahead; one raw s_barrier per K-tile.  Epilogues: LDS-staged coalesced bf16 rows / fp32 / fp32 split-K partial slabs / fused SwiGLU.
// Three operand forms share the main loop:
//   NT      Out[m][n] = sum_k X[m][k] W[n][k]          (forward, dgrad)
//   SWIGLU  NT + silu(gate)*y epilogue                  (fc11 || fc12, rows interleaved by 32)
//   TN      G[o][i]   = sum_t dY[t][o] A[t][i]          (wgrad in its natural layout, ds_read_b64_tr_b16 fragments)
#include "cx_common.h"
#include "../../include/contrastors_hip.h"
#include "gemm_params.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* glb_void_ptr;
typedef __attribute__((address_space(3))) bf16x4_t* lds_b4_ptr;

constexpr int BM5 = 256, BN5 = 256, BK5 = 64;
constexpr int XB5 = BM5 * BK5 * 2;           // 32 KiB
constexpr int ST5 = (BM5 + BN5) * BK5 * 2;   // 64 KiB
constexpr int NST5 = 2;
constexpr int TNROW5 = 512;                  // bytes per token row of a [64 t][256 f] TN tile

enum Form { FORM_NT = 0, FORM_TN = 1 };

// TN fragment: see gemm_bf16.hip (tn_frag) -- lane (g = lane>>4, p = lane&15) -> feature f0 + 16*(g&1) + p,
// tokens t0 + 8*(g>>1) + {0..7}; 16-B chunk index XORed with (t&3)<<2 (conflict-free, verified by PMC).
CX_DEVICE bf16x8_t tn_frag5(const char* tile, int f0, int t0, int lane) {
    const int g = lane >> 4, p = lane & 15;
    const int t = t0 + 8 * (g >> 1) + (p >> 2);
    const int f = f0 + 16 * (g & 1) + 4 * (p & 3);
    union { bf16x4_t h[2]; bf16x8_t v; } u;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int tt = t + 4 * half;
        const int chunk = (f >> 3) ^ ((tt & 3) << 2);
        u.h[half] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4_ptr)(tile + tt * TNROW5 + chunk * 16 + (f & 4) * 2));
    }
    return u.v;
}

template <int FORM, int OUT_MODE, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_bf16_v5_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 2, wn = wave & 3;

    const int nwg = p.tiles_m * p.tiles_n * p.split_k;
    int lid = xcd_remap(blockIdx.x, nwg);
    const int tn = lid % p.tiles_n;
    lid /= p.tiles_n;
    const int tm = lid % p.tiles_m;
    const int sk = lid / p.tiles_m;
    const int m0 = tm * BM5, n0 = tn * BN5;
    const int nk_total = p.K / BK5;
    const int kt_begin = (int)(((long)nk_total * sk) / p.split_k);
    const int kt_end = (int)(((long)nk_total * (sk + 1)) / p.split_k);
    const int nk = kt_end - kt_begin;

    // ---- DMA sources: 64 instructions of 1 KiB per stage (32 X + 32 W), 4 + 4 per wave -------------------------
    const bf16_t* xsrc[4];
    const bf16_t* wsrc[4];
    size_t xstep, wstep;
    if constexpr (FORM == FORM_NT) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // instruction q = j*8 + wave covers tile rows 8q..8q+7 (128 B each)
            const int r = (j * 8 + wave) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            int gx = m0 + r, gw = n0 + r;
            gx = gx < p.M ? gx : p.M - 1;
            gw = gw < p.N ? gw : p.N - 1;
            xsrc[j] = p.X + (size_t)gx * p.ldx + (size_t)kt_begin * BK5 + c * 8;
            wsrc[j] = p.W + (size_t)gw * p.ldw + (size_t)kt_begin * BK5 + c * 8;
        }
        xstep = wstep = BK5;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // instruction q covers token rows 2q, 2q+1 of a [64 t][256 f] tile (512 B each)
            const int t = (j * 8 + wave) * 2 + (lane >> 5);
            const int c = (lane & 31) ^ ((t & 3) << 2);
            xsrc[j] = p.X + ((size_t)kt_begin * BK5 + t) * p.ldx + m0 + c * 8;
            wsrc[j] = p.W + ((size_t)kt_begin * BK5 + t) * p.ldw + n0 + c * 8;
        }
        xstep = (size_t)BK5 * p.ldx;
        wstep = (size_t)BK5 * p.ldw;
    }
    // One K-tile = 8 DMA instructions per wave; issued as an X half and a W half between MFMA groups (back-to-back DMA
    // instructions stall the wave at issue, and that stall would serialise with the MFMA phase).
    auto issue_x = [&](int stage) {
        char* base = dsm + stage * ST5;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __builtin_amdgcn_global_load_lds((glb_void_ptr)xsrc[j], (lds_void_ptr)(base + (j * 8 + wave) * 1024), 16, 0, 0);
            xsrc[j] += xstep;
        }
    };
    auto issue_w = [&](int stage) {
        char* base = dsm + stage * ST5 + XB5;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __builtin_amdgcn_global_load_lds((glb_void_ptr)wsrc[j], (lds_void_ptr)(base + (j * 8 + wave) * 1024), 16, 0, 0);
            wsrc[j] += wstep;
        }
    };

    f32x16_t acc[2][4];  // [n-block a][m-block b]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // Fragments are double-buffered and read one k-step ahead of the MFMAs that consume them (across the K-tile boundary
    // too); the reads of the next k-step are issued after the first MFMA of the current one, so the lgkmcnt wait in
    // front of that MFMA never covers reads that were only just issued.
    struct Frags { bf16x8_t w[2], x[4]; };
    Frags F0, F1;
    auto read_frags = [&](Frags& f, int stage, int ks) {
        const char* xs = dsm + stage * ST5;
        const char* ws = xs + XB5;
        if constexpr (FORM == FORM_NT) {
#pragma unroll
            for (int a = 0; a < 2; ++a) f.w[a] = lds_read_frag(ws, tile64_off(wn * 64 + a * 32 + l31, ks * 2 + hi));
#pragma unroll
            for (int b = 0; b < 4; ++b) f.x[b] = lds_read_frag(xs, tile64_off(wm * 128 + b * 32 + l31, ks * 2 + hi));
        } else {
#pragma unroll
            for (int a = 0; a < 2; ++a) f.w[a] = tn_frag5(ws, wn * 64 + a * 32, ks * 16, lane);
#pragma unroll
            for (int b = 0; b < 4; ++b) f.x[b] = tn_frag5(xs, wm * 128 + b * 32, ks * 16, lane);
        }
    };
    auto mma_head = [&](const Frags& f) { acc[0][0] = mfma_bf16_32x32x16(f.w[0], f.x[0], acc[0][0]); };
    auto mma_tail = [&](const Frags& f) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (a + b) acc[a][b] = mfma_bf16_32x32x16(f.w[a], f.x[b], acc[a][b]);
    };

    if (nk > 0) {
        issue_x(0);
        issue_w(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        read_frags(F0, 0, 0);
    }
#pragma unroll 1
    for (int t = 0; t < nk; ++t) {
        const bool more = t + 1 < nk;
        const int cur = t & 1, nxt = cur ^ 1;  // stage nxt was consumed in iteration t-1 (barrier at its end)
        __builtin_amdgcn_s_setprio(1);
        mma_head(F0);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(F1, cur, 1);
        if (more) issue_x(nxt);
        mma_tail(F0);
        mma_head(F1);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(F0, cur, 2);
        if (more) issue_w(nxt);
        mma_tail(F1);
        mma_head(F0);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(F1, cur, 3);
        mma_tail(F0);
        __builtin_amdgcn_s_setprio(0);
        // this wave's reads of stage cur are complete and its share of K-tile t+1 has landed ...
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // ... everyone's
        __builtin_amdgcn_s_setprio(1);
        mma_head(F1);
        __builtin_amdgcn_sched_barrier(0);
        if (more) read_frags(F0, nxt, 0);
        mma_tail(F1);
        __builtin_amdgcn_s_setprio(0);
    }

    // ---- epilogue.  acc[a][b][r]: m = m0 + wm*128 + b*32 + l31,  n = n0 + wn*64 + a*32 + acc_row(r,hi) --------
    const bool add_bias = (p.bias != nullptr) && (sk == 0);
    constexpr int ROWB = 144, SLOT = 64 * ROWB;  // staging slot of one wave: 64 rows x 64 bf16 (+16 B pad)
    if constexpr (OUT_MODE == GEMM_OUT_BF16 && EPI == GEMM_EPI_NONE) {
        if ((p.ldo & 7) == 0) {
            char* my = dsm + wave * SLOT;
            __builtin_amdgcn_s_barrier();  // every wave is done reading the operand stages; slots are wave-private after
#pragma unroll
            for (int half = 0; half < 2; ++half) {  // 64 of the wave's 128 rows per pass (8 x 9 KiB = 72 KiB of LDS)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    const int b = half * 2 + bb;
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int nl = a * 32 + 8 * q + 4 * hi;
                            float v[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = acc[a][b][4 * q + e] * p.alpha;
                            if (add_bias) {
                                const int n = n0 + wn * 64 + nl;
                                if (n < p.N) {
                                    const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
                                    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                                }
                            }
                            uint2 pk;
                            pk.x = pack_bf16x2(v[0], v[1]);
                            pk.y = pack_bf16x2(v[2], v[3]);
                            *reinterpret_cast<uint2*>(my + (bb * 32 + l31) * ROWB + nl * 2) = pk;
                        }
                }
#pragma unroll
                for (int ps = 0; ps < 8; ++ps) {  // in-order LDS: this wave reads back its own slot
                    const int row = ps * 8 + (lane >> 3), ch = lane & 7;
                    const int m = m0 + wm * 128 + half * 64 + row, n = n0 + wn * 64 + ch * 8;
                    const uint4 vv = *reinterpret_cast<const uint4*>(my + row * ROWB + ch * 16);
                    if (m < p.M) {
                        bf16_t* dst = reinterpret_cast<bf16_t*>(p.Out) + (size_t)m * p.ldo + n;
                        if (n + 8 <= p.N) {
                            *reinterpret_cast<uint4*>(dst) = vv;
                        } else if (n < p.N) {
                            *reinterpret_cast<uint2*>(dst) = make_uint2(vv.x, vv.y);
                        }
                    }
                }
            }
            return;
        }
    }
    if constexpr (EPI == GEMM_EPI_SWIGLU) {
        // a = 0: y rows, a = 1: gate rows of the same 32 activation columns (weight rows interleaved by 32), so the
        // wave's 64 fused columns [y 0..31 | gate 0..31] are exactly one 128-B line per row of the (M, 2I) pre-activation
        // tensor.  Both outputs leave through LDS staging as whole-line / half-line coalesced stores.
        __builtin_amdgcn_s_barrier();  // operand stages are free; staging slots are wave-private from here on
        if (p.Out) {
            char* my = dsm + wave * SLOT;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    const int b = half * 2 + bb;
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            uint2 pk;
                            pk.x = pack_bf16x2(acc[a][b][4 * q], acc[a][b][4 * q + 1]);
                            pk.y = pack_bf16x2(acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]);
                            *reinterpret_cast<uint2*>(my + (bb * 32 + l31) * ROWB + (a * 32 + 8 * q + 4 * hi) * 2) = pk;
                        }
                }
#pragma unroll
                for (int ps = 0; ps < 8; ++ps) {
                    const int row = ps * 8 + (lane >> 3), ch = lane & 7;
                    const int m = m0 + wm * 128 + half * 64 + row, n = n0 + wn * 64 + ch * 8;
                    const uint4 vv = *reinterpret_cast<const uint4*>(my + row * ROWB + ch * 16);
                    if (m < p.M && n < p.N)
                        *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.Out) + (size_t)m * p.ldo + n) = vv;
                }
            }
        }
        constexpr int AROWB = 80;  // 64 B of activations per row + 16 B pad
        char* mya = dsm + wave * (128 * AROWB);
        if (p.Out) __builtin_amdgcn_s_barrier();  // the activation slots overlap other waves' pre-activation slots
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {  // the standalone op sees bf16 y / gate (FusedDense outputs)
                    const float yy = bf16_to_f32(f32_to_bf16(acc[0][b][4 * q + e]));
                    const float gg = bf16_to_f32(f32_to_bf16(acc[1][b][4 * q + e]));
                    o[e] = gg / (1.f + __expf(-gg)) * yy;
                }
                uint2 pk;
                pk.x = pack_bf16x2(o[0], o[1]); pk.y = pack_bf16x2(o[2], o[3]);
                *reinterpret_cast<uint2*>(mya + (b * 32 + l31) * AROWB + (8 * q + 4 * hi) * 2) = pk;
            }
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {  // 16 rows x 4 chunks of 16 B per pass
            const int row = ps * 16 + (lane >> 2), ch = lane & 3;
            const int m = m0 + wm * 128 + row;
            const int col = ((n0 + wn * 64) >> 1) + ch * 8;
            const uint4 vv = *reinterpret_cast<const uint4*>(mya + row * AROWB + ch * 16);
            if (m < p.M && 2 * col < p.N)
                *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.Out2) + (size_t)m * p.ldo2 + col) = vv;
        }
        return;
    }
    float* part = nullptr;
    if constexpr (OUT_MODE == GEMM_OUT_F32_PARTIAL) part = reinterpret_cast<float*>(p.Out) + (size_t)sk * p.M * p.ldo;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int m = m0 + wm * 128 + b * 32 + l31;
        if (m >= p.M) continue;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wn * 64 + a * 32 + 8 * q + 4 * hi;
                if (n >= p.N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[a][b][4 * q + e] * p.alpha;
                if (add_bias) {
                    const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
                    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                }
                if constexpr (OUT_MODE == GEMM_OUT_BF16) {
                    uint2 pk;
                    pk.x = pack_bf16x2(v[0], v[1]);
                    pk.y = pack_bf16x2(v[2], v[3]);
                    *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.Out) + (size_t)m * p.ldo + n) = pk;
                } else if constexpr (OUT_MODE == GEMM_OUT_F32) {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.Out) + (size_t)m * p.ldo + n) =
                        make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    *reinterpret_cast<float4*>(part + (size_t)m * p.ldo + n) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
    }
}

template <int FORM, int OUT_MODE, int EPI>
hipError_t launch5(const GemmParams& p, hipStream_t stream) {
    static CxLdsOptIn lds;
    if (!lds.ensure(reinterpret_cast<const void*>(&gemm_bf16_v5_kernel<FORM, OUT_MODE, EPI>), NST5 * ST5)) return hipErrorInvalidValue;
    const int nwg = p.tiles_m * p.tiles_n * p.split_k;
    hipLaunchKernelGGL((gemm_bf16_v5_kernel<FORM, OUT_MODE, EPI>), dim3(nwg), dim3(512), NST5 * ST5, stream, p);
    return hipGetLastError();
}


#ifndef CX_PRODUCT  // the 8-wave persistent kernel (superseded by gemm_bf16_v6.hip) and its switches: dev library only
// =====================================================================================================================
// Persistent walk of the same 256x256x64 tile (NT forms with bf16 output): 256 workgroups, each loops over its tiles
// and keeps the two-stage DMA pipeline running ACROSS tile boundaries -- the first K-tile of the next output tile is
// already in flight while the current tile's epilogue runs, and the epilogue's global stores drain underneath the next
// tile's first K-step.  vmcnt is shared by DMA loads and stores, so the kernel never counts across them: at a tile end
// it waits (vmcnt(0)) for the already-issued first DMA group of the next tile BEFORE issuing the epilogue stores (that
// group has had a whole compute phase to land), and skips the wait of the following iteration; one iteration later the
// ordinary vmcnt(0) also covers the stores, which have had a full K-step to drain.
// The epilogue stages through the stage consumed last (64 KiB), a quarter (32 rows per wave) at a time.
// =====================================================================================================================
// LDS map of the persistent kernel (all 160 KiB): three 32 KiB X slots (ring of 3) + two 32 KiB W slots (ring of 2).
// X runs TWO K-tiles ahead, W one: 96 KiB of DMA in flight per CU instead of 64 KiB (the two-stage kernel is latency-
// bound at 64 KiB / ~2 us = 32 GB/s/CU; the request pipe sustains ~45 GB/s/CU).  Issue order per iteration is
// [W of t+1][X of t+2], so "everything but the newest X group" = s_waitcnt vmcnt(4) -- loads retire in order among
// loads, and any still-pending epilogue store only makes that wait longer, never shorter.
constexpr int XS5 = 32768;
constexpr int PERS_LDS = 5 * XS5;

template <int EPI, bool TRACE>
__global__ __launch_bounds__(512, 2) void gemm_bf16_v5p_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 2, wn = wave & 3;
    const int ntiles = p.tiles_m * p.tiles_n;
    const int nk = p.K / BK5;
    // tile order: workgroup b runs on XCD b%8; per round an XCD takes 32 consecutive tiles (n fastest: shared X panel)
    const int per_xcd = gridDim.x >> 3;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    auto tile_of = [&](int round) { return (round * 8 + xcd) * per_xcd + idx; };

    // ---- two independent DMA cursors (X two K-tiles ahead, W one) ------------------------------------------------
    const bf16_t* xsrc[4];
    const bf16_t* wsrc[4];
    int lx_round = 0, lx_kt = 0, lx_slot = 0;
    int lw_round = 0, lw_kt = 0, lw_slot = 0;
    int x_issued = 0, g = 0;  // X groups issued so far / index of the current iteration
    bool lx_live, lw_live;
    auto x_setup = [&](int tile) {
        const int tm = tile / p.tiles_n;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = (j * 8 + wave) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            int gx = tm * BM5 + r;
            gx = gx < p.M ? gx : p.M - 1;
            xsrc[j] = p.X + (size_t)gx * p.ldx + c * 8;
        }
    };
    auto w_setup = [&](int tile) {
        const int tn = tile % p.tiles_n;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = (j * 8 + wave) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            int gw = tn * BN5 + r;
            gw = gw < p.N ? gw : p.N - 1;
            wsrc[j] = p.W + (size_t)gw * p.ldw + c * 8;
        }
    };
    const bool no_dma = TRACE && (p.dbg & 1);  // experiment (trace build only): main loop without DMA, results wrong
    // A K-tile of either operand is 4 DMA instructions per wave.  They are issued in two halves so that the main loop
    // can spread them between MFMA groups: back-to-back DMA instructions stall the wave at issue (the vector-memory
    // front end accepts ~1 KiB-instruction per ~30 clk per CU), and that stall would serialise with the MFMA phase.
    auto issue_x_half = [&](int h) {  // next X K-tile -> X slot ring
        char* base = dsm + lx_slot * XS5;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int j = 2 * h + jj;
            if (!no_dma)
                __builtin_amdgcn_global_load_lds((glb_void_ptr)xsrc[j], (lds_void_ptr)(base + (j * 8 + wave) * 1024), 16, 0, 0);
            xsrc[j] += BK5;
        }
        if (h == 1) {
            lx_slot = lx_slot == 2 ? 0 : lx_slot + 1;
            ++x_issued;
            if (++lx_kt == nk) {
                lx_kt = 0;
                const int t = tile_of(++lx_round);
                lx_live = t < ntiles;
                if (lx_live) x_setup(t);
            }
        }
    };
    auto issue_w_half = [&](int h) {
        char* base = dsm + 3 * XS5 + lw_slot * XS5;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int j = 2 * h + jj;
            if (!no_dma)
                __builtin_amdgcn_global_load_lds((glb_void_ptr)wsrc[j], (lds_void_ptr)(base + (j * 8 + wave) * 1024), 16, 0, 0);
            wsrc[j] += BK5;
        }
        if (h == 1) {
            lw_slot ^= 1;
            if (++lw_kt == nk) {
                lw_kt = 0;
                const int t = tile_of(++lw_round);
                lw_live = t < ntiles;
                if (lw_live) w_setup(t);
            }
        }
    };
    auto issue_x = [&]() { issue_x_half(0); issue_x_half(1); };
    auto issue_w = [&]() { issue_w_half(0); issue_w_half(1); };

    f32x16_t acc[2][4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    };
    zero_acc();

    int cp_round = 0, cp_kt = 0;
    int cp_tile = tile_of(0);
    lx_live = lw_live = cp_tile < ntiles;
    if (lx_live) {
        x_setup(cp_tile);
        w_setup(cp_tile);
        issue_x();               // X of iteration 0
        issue_w();               // W of iteration 0
        if (lx_live) issue_x();  // X of iteration 1
    }
    int xs_slot = 0, ws_slot = 0;
    constexpr int ROWB = 144, QSLOT = 32 * ROWB;  // quarter staging slot of one wave: 32 rows x 64 bf16 (+16 B pad)

    // Fragments are double-buffered (F0/F1) and read one k-step ahead of the MFMAs that consume them, across the
    // iteration boundary too: the LDS round trip never sits between two MFMA groups of the same wave.
    struct Frags { bf16x8_t w[2], x[4]; };
    Frags F0, F1;
    auto read_frags = [&](Frags& f, const char* xs, const char* ws, int ks) {
#pragma unroll
        for (int a = 0; a < 2; ++a) f.w[a] = lds_read_frag(ws, tile64_off(wn * 64 + a * 32 + l31, ks * 2 + hi));
#pragma unroll
        for (int b = 0; b < 4; ++b) f.x[b] = lds_read_frag(xs, tile64_off(wm * 128 + b * 32 + l31, ks * 2 + hi));
    };
    // One k-step = 8 MFMAs.  The reads of the NEXT k-step are issued after the first of them: the lgkmcnt wait the
    // compiler places in front of that first MFMA then never covers reads that were only just issued.
    auto mma_head = [&](const Frags& f) { acc[0][0] = mfma_bf16_32x32x16(f.w[0], f.x[0], acc[0][0]); };
    auto mma_tail = [&](const Frags& f) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (a + b) acc[a][b] = mfma_bf16_32x32x16(f.w[a], f.x[b], acc[a][b]);
    };

    constexpr bool tracing = TRACE;  // separate instantiation: the timers' SMEM reads would force lgkmcnt(0) waits
    long long t_wait = 0, t_comp = 0, t_epi = 0, t_epi_wait = 0, t0 = 0, t1 = 0;
    if (cp_tile < ntiles) {
        // operands of iteration 0 (X_0, W_0); the only younger group is X_1 (4 ops), if issued
        if (x_issued > 1) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        read_frags(F0, dsm, dsm + 3 * XS5, 0);
    }
    if constexpr (tracing) t1 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    while (cp_tile < ntiles) {
        // DMA of this iteration: W of iteration +1 (k-steps 0,1) then X of iteration +2 (k-steps 2,3); both target
        // slots consumed in iteration -1.  W before X keeps the vmcnt accounting below: at the end of iteration g the
        // operands of g+1 are everything except the 4 newest ops (X_{g+2}).
        const bool w_go = lw_live, x_go = lx_live;
        const char* xs = dsm + xs_slot * XS5;
        const char* ws = dsm + 3 * XS5 + ws_slot * XS5;
        const int nxs_slot = xs_slot == 2 ? 0 : xs_slot + 1, nws_slot = ws_slot ^ 1;
        const bool tile_end = cp_kt + 1 == nk;
        __builtin_amdgcn_s_setprio(1);
        mma_head(F0);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(F1, xs, ws, 1);
        if (w_go) issue_w_half(0);
        mma_tail(F0);
        mma_head(F1);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(F0, xs, ws, 2);
        if (w_go) issue_w_half(1);
        mma_tail(F1);
        mma_head(F0);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(F1, xs, ws, 3);
        if (x_go) issue_x_half(0);
        mma_tail(F0);
        if (x_go) issue_x_half(1);
        __builtin_amdgcn_s_setprio(0);
        if constexpr (tracing) { t0 = __builtin_amdgcn_s_memtime(); t_comp += t0 - t1; }
        // this wave's reads of the current slots are complete (F1 has landed) ...
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // ... and so are its DMA writes of the next iteration's operands (everything, at a tile end: the epilogue's
        // global stores must not be mixed into the DMA accounting)
        if (tracing && (p.dbg & 16) && !tile_end) {
            // experiment (trace build only): no DMA wait -> what is left of t_wait is barrier skew; results are wrong
        } else if (x_go && !tile_end) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if constexpr (tracing) { t1 = __builtin_amdgcn_s_memtime(); t_wait += t1 - t0; }
        __builtin_amdgcn_s_setprio(1);
        mma_head(F1);
        __builtin_amdgcn_sched_barrier(0);
        if (!tile_end) read_frags(F0, dsm + nxs_slot * XS5, dsm + 3 * XS5 + nws_slot * XS5, 0);
        mma_tail(F1);
        __builtin_amdgcn_s_setprio(0);

        if (tile_end) {
            // ---- epilogue.  Free LDS: the X slot and the W slot just consumed (32 KiB each); waves 0-3 stage in the
            // former, waves 4-7 in the latter.  Every in-flight DMA group targets other slots.
            const int tn = cp_tile % p.tiles_n, tm = cp_tile / p.tiles_n;
            const int m0 = tm * BM5, n0 = tn * BN5;
            char* area = (wave < 4) ? dsm + xs_slot * XS5 : dsm + 3 * XS5 + ws_slot * XS5;
            char* my = area + (wave & 3) * QSLOT;
            if constexpr (tracing) t0 = __builtin_amdgcn_s_memtime();
            if constexpr (EPI == GEMM_EPI_NONE) {
                const bool add_bias = p.bias != nullptr;
#pragma unroll
                for (int b = 0; b < 4; ++b) {  // 32 rows of the wave's 128 per pass
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int nl = a * 32 + 8 * q + 4 * hi;
                            float v[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = acc[a][b][4 * q + e] * p.alpha;
                            if (add_bias) {
                                const int n = n0 + wn * 64 + nl;
                                if (n < p.N) {
                                    const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
                                    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                                }
                            }
                            uint2 pk;
                            pk.x = pack_bf16x2(v[0], v[1]);
                            pk.y = pack_bf16x2(v[2], v[3]);
                            *reinterpret_cast<uint2*>(my + l31 * ROWB + nl * 2) = pk;
                        }
#pragma unroll
                    for (int ps = 0; ps < 4; ++ps) {
                        const int row = ps * 8 + (lane >> 3), ch = lane & 7;
                        const int m = m0 + wm * 128 + b * 32 + row, n = n0 + wn * 64 + ch * 8;
                        const uint4 vv = *reinterpret_cast<const uint4*>(my + row * ROWB + ch * 16);
                        if (m < p.M && n + 8 <= p.N)
                            *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.Out) + (size_t)m * p.ldo + n) = vv;
                    }
                }
            } else {
                // SwiGLU: optional pre-activation pair (whole 128-B lines), then the activation (64-B row segments)
                if (p.Out) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
#pragma unroll
                        for (int a = 0; a < 2; ++a)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                uint2 pk;
                                pk.x = pack_bf16x2(acc[a][b][4 * q], acc[a][b][4 * q + 1]);
                                pk.y = pack_bf16x2(acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]);
                                *reinterpret_cast<uint2*>(my + l31 * ROWB + (a * 32 + 8 * q + 4 * hi) * 2) = pk;
                            }
#pragma unroll
                        for (int ps = 0; ps < 4; ++ps) {
                            const int row = ps * 8 + (lane >> 3), ch = lane & 7;
                            const int m = m0 + wm * 128 + b * 32 + row, n = n0 + wn * 64 + ch * 8;
                            const uint4 vv = *reinterpret_cast<const uint4*>(my + row * ROWB + ch * 16);
                            if (m < p.M && n < p.N)
                                *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.Out) + (size_t)m * p.ldo + n) = vv;
                        }
                    }
                }
                constexpr int AROWB = 80, HSLOT = 64 * AROWB;  // activation staging: 64 rows x 32 cols per wave per pass
                char* mya = area + (wave & 3) * HSLOT;         // 4 x 5 KiB = 20 KiB <= 32 KiB per area
                if (p.Out) __builtin_amdgcn_s_barrier();       // activation slots overlap neighbours' quarter slots
#pragma unroll
                for (int half = 0; half < 2; ++half) {
#pragma unroll
                    for (int bb = 0; bb < 2; ++bb) {
                        const int b = half * 2 + bb;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            float o[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float yy = bf16_to_f32(f32_to_bf16(acc[0][b][4 * q + e]));
                                const float gg = bf16_to_f32(f32_to_bf16(acc[1][b][4 * q + e]));
                                o[e] = gg / (1.f + __expf(-gg)) * yy;
                            }
                            uint2 pk;
                            pk.x = pack_bf16x2(o[0], o[1]); pk.y = pack_bf16x2(o[2], o[3]);
                            *reinterpret_cast<uint2*>(mya + (bb * 32 + l31) * AROWB + (8 * q + 4 * hi) * 2) = pk;
                        }
                    }
#pragma unroll
                    for (int ps = 0; ps < 4; ++ps) {
                        const int row = ps * 16 + (lane >> 2), ch = lane & 3;
                        const int m = m0 + wm * 128 + half * 64 + row;
                        const int col = ((n0 + wn * 64) >> 1) + ch * 8;
                        const uint4 vv = *reinterpret_cast<const uint4*>(mya + row * AROWB + ch * 16);
                        if (m < p.M && 2 * col < p.N)
                            *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.Out2) + (size_t)m * p.ldo2 + col) = vv;
                    }
                }
            }
            zero_acc();
            cp_kt = -1;
            cp_tile = tile_of(++cp_round);
            // the staging areas are the DMA targets of the next iteration: nobody may still be reading them
            __builtin_amdgcn_s_barrier();
            if (cp_tile < ntiles) read_frags(F0, dsm + nxs_slot * XS5, dsm + 3 * XS5 + nws_slot * XS5, 0);
            if constexpr (tracing) { t1 = __builtin_amdgcn_s_memtime(); t_epi += t1 - t0; }
        }
        ++cp_kt;
        xs_slot = nxs_slot;
        ws_slot = nws_slot;
        ++g;
    }
    if (tracing && lane == 0) {
        long long* o = p.trace + ((size_t)blockIdx.x * 8 + wave) * 8;
        o[0] = t_wait; o[1] = t_comp; o[2] = t_epi; o[3] = g; o[4] = t_epi_wait;
    }
}

long long* g_v5_trace = nullptr;

template <int EPI, bool TRACE>
hipError_t launch5p_t(const GemmParams& q, int grid, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_v5p_kernel<EPI, TRACE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, PERS_LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_bf16_v5p_kernel<EPI, TRACE>), dim3(grid), dim3(512), PERS_LDS, stream, q);
    return hipGetLastError();
}

template <int EPI>
hipError_t launch5p(const GemmParams& p, hipStream_t stream) {
    const int ntiles = p.tiles_m * p.tiles_n;
    const int grid = ntiles < 256 ? (ntiles + 7) / 8 * 8 : 256;
    GemmParams q = p;
    q.trace = g_v5_trace;
    return q.trace ? launch5p_t<EPI, true>(q, grid, stream) : launch5p_t<EPI, false>(q, grid, stream);
}

bool g_v5_persistent = true;
bool g_v5_use_v6 = true;  // default: persistent NT forms run on the one-wave-per-SIMD kernel (gemm_bf16_v6.hip)
#endif  // !CX_PRODUCT

}  // namespace

#ifndef CX_PRODUCT
void cx_gemm_v5_set_use_v6(bool on) { g_v5_use_v6 = on; }
bool cx_gemm_v5_get_use_v6(void) { return g_v5_use_v6; }

void cx_gemm_v5_set_persistent(bool on) { g_v5_persistent = on; }
void cx_gemm_v5_set_trace(long long* buf) { g_v5_trace = buf; }
#endif

// form 0 = NT, 1 = TN (p.X = dY (T,M), p.W = A (T,N), p.K = tokens; M % 256 == 0 and N % 256 == 0 required).
hipError_t cx_launch_gemm_v5(GemmParams p, int form, int out_mode, int epi, hipStream_t stream) {
    if ((p.K % BK5) != 0) return hipErrorInvalidValue;
    p.tiles_m = (p.M + BM5 - 1) / BM5;
    p.tiles_n = (p.N + BN5 - 1) / BN5;
    if (p.split_k < 1) p.split_k = 1;
    if (p.split_k > p.K / BK5) p.split_k = p.K / BK5;
#ifdef CX_PRODUCT
    // product library: the persistent forms are gemm_bf16_v6.hip's (called directly by gemm_api.hip); what is left here is
    // the one-tile-per-workgroup kernel for fp32 / fp32-partial outputs and bf16 shapes with N or ldo not a multiple of 8
    if (form == 1) return hipErrorInvalidValue;
    if (epi == GEMM_EPI_SWIGLU)
        return out_mode == GEMM_OUT_BF16 ? launch5<FORM_NT, GEMM_OUT_BF16, GEMM_EPI_SWIGLU>(p, stream) : hipErrorInvalidValue;
#else
    if (form == 1) {
        if ((p.M % BM5) != 0 || (p.N % BN5) != 0 || out_mode != GEMM_OUT_F32_PARTIAL) return hipErrorInvalidValue;
        if (g_v5_use_v6) return cx_launch_gemm_v6_tn(p, stream);
        return launch5<FORM_TN, GEMM_OUT_F32_PARTIAL, GEMM_EPI_NONE>(p, stream);
    }
    const bool pers = g_v5_persistent && p.split_k == 1 && (p.N % 8) == 0;
    if (epi == GEMM_EPI_SWIGLU) {
        if (out_mode != GEMM_OUT_BF16) return hipErrorInvalidValue;
        if (pers && (p.ldo % 8) == 0 && (p.ldo2 % 8) == 0)
            return (g_v5_use_v6 && !g_v5_trace) ? cx_launch_gemm_v6(p, GEMM_EPI_SWIGLU, stream) : launch5p<GEMM_EPI_SWIGLU>(p, stream);
        return launch5<FORM_NT, GEMM_OUT_BF16, GEMM_EPI_SWIGLU>(p, stream);
    }
    if (out_mode == GEMM_OUT_BF16 && pers && (p.ldo % 8) == 0)
        return (g_v5_use_v6 && !g_v5_trace) ? cx_launch_gemm_v6(p, GEMM_EPI_NONE, stream) : launch5p<GEMM_EPI_NONE>(p, stream);
#endif
    switch (out_mode) {
        case GEMM_OUT_BF16: return launch5<FORM_NT, GEMM_OUT_BF16, GEMM_EPI_NONE>(p, stream);
        case GEMM_OUT_F32: return launch5<FORM_NT, GEMM_OUT_F32, GEMM_EPI_NONE>(p, stream);
        case GEMM_OUT_F32_PARTIAL: return launch5<FORM_NT, GEMM_OUT_F32_PARTIAL, GEMM_EPI_NONE>(p, stream);
        default: return hipErrorInvalidValue;
    }
}
