// gemm_bf16.hip -- K9 of SURVEY.md §2b: the FusedDense contraction  Out[m][n] = sum_k X[m][k] * W[n][k]
// (+ bias[n]) on CDNA4 matrix cores.  Replaces flash_attn.ops.fused_dense (cuBLASLt) at the call sites
// contrastors/layers/attention.py:82-85,112-114,243 and contrastors/layers/mlp.py:24-28,61-83.
//
// One kernel serves forward (X=activations, W=weight), dgrad (X=dY, W=W^T shadow) and wgrad
// (X=dY^T, W=act^T, fp32 atomic accumulate, split-K over tokens): both operands are always
// K-contiguous ("NT"), which is the layout MFMA fragments want (8 consecutive k per lane = one 16-B load).
//
// Structure (v1, "step-3" of the guide's ladder): 128x128x64 block tile, 4 waves in 2x2, each wave a 64x64
// sub-tile = 2x2 MFMA 32x32x16 accumulators; LDS double buffer (64 KiB -> 2 blocks/CU); operand tiles either
// DMA'd straight into LDS (global_load_lds, 16 B/lane, swizzle applied on the SOURCE address) or staged through
// registers; XCD-aware tile order.  The MFMA is issued with A := W-fragment and B := X-fragment so that each
// lane ends up holding 4 consecutive n for one m -> 8/16-byte epilogue stores.
#include "cx_common.h"
#include "../../include/contrastors_hip.h"
#include "gemm_params.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * BK * 2;  // 16 KiB per operand tile

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* glb_void_ptr;

enum OutMode { OUT_BF16 = GEMM_OUT_BF16, OUT_F32 = GEMM_OUT_F32, OUT_F32_ATOMIC = GEMM_OUT_F32_ATOMIC };

// Staging registers as a plain struct of named members: an indexed array here is not scalarised by the compiler
// (it lands in scratch / promoted LDS).
struct StageRegs { uint4 r0, r1, r2, r3; };

CX_DEVICE uint4 stage_load(const bf16_t* __restrict__ base, int ld, int row0, int nrows, int k0, int tid, int p) {
    const int r = p * 32 + (tid >> 3);
    int gr = row0 + r;
    gr = gr < nrows ? gr : nrows - 1;
    return *reinterpret_cast<const uint4*>(base + (size_t)gr * ld + k0 + (tid & 7) * 8);
}

template <bool GLDS>
CX_DEVICE void stage_tile(const bf16_t* __restrict__ base, int ld, int row0, int nrows, int k0, char* lds_tile,
                          int tid, StageRegs& regs) {
    if constexpr (GLDS) {
        // LDS destination of one wave-instruction is  M0 + lane*16 : 1 KiB = 8 tile rows.  The swizzle of
        // tile64_off() is an involution on the chunk index within a row, so fetching chunk (s ^ key(r)) into
        // linear slot s produces exactly the swizzled image (guide §5.4 rule 21).
        const int lane = tid & 63;
        const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = j * 32 + w * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            int gr = row0 + r;
            gr = gr < nrows ? gr : nrows - 1;
            const bf16_t* g = base + (size_t)gr * ld + k0 + c * 8;
            __builtin_amdgcn_global_load_lds((glb_void_ptr)g, (lds_void_ptr)(lds_tile + (j * 4 + w) * 1024), 16, 0, 0);
        }
    } else {
        regs.r0 = stage_load(base, ld, row0, nrows, k0, tid, 0);
        regs.r1 = stage_load(base, ld, row0, nrows, k0, tid, 1);
        regs.r2 = stage_load(base, ld, row0, nrows, k0, tid, 2);
        regs.r3 = stage_load(base, ld, row0, nrows, k0, tid, 3);
    }
}

CX_DEVICE void commit_tile(char* lds_tile, int tid, const StageRegs& regs) {
    const int r = tid >> 3, c = tid & 7;
    *reinterpret_cast<uint4*>(lds_tile + tile64_off(r, c)) = regs.r0;
    *reinterpret_cast<uint4*>(lds_tile + tile64_off(r + 32, c)) = regs.r1;
    *reinterpret_cast<uint4*>(lds_tile + tile64_off(r + 64, c)) = regs.r2;
    *reinterpret_cast<uint4*>(lds_tile + tile64_off(r + 96, c)) = regs.r3;
}

template <bool GLDS, int OUT_MODE>
__global__ __launch_bounds__(256, 2) void gemm_bf16_nt_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];  // [buf][X|W]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;

    const int nwg = p.tiles_m * p.tiles_n * p.split_k;
    int lid = xcd_remap(blockIdx.x, nwg);
    const int tn = lid % p.tiles_n;
    lid /= p.tiles_n;
    const int tm = lid % p.tiles_m;
    const int sk = lid / p.tiles_m;

    const int m0 = tm * BM, n0 = tn * BN;
    const int nk_total = p.K / BK;
    const int kt_begin = (int)(((long)nk_total * sk) / p.split_k);
    const int kt_end = (int)(((long)nk_total * (sk + 1)) / p.split_k);
    const int nk = kt_end - kt_begin;

    f32x16_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    StageRegs xr = {}, wr = {};
    if (nk > 0) {
        stage_tile<GLDS>(p.X, p.ldx, m0, p.M, kt_begin * BK, smem, tid, xr);
        stage_tile<GLDS>(p.W, p.ldw, n0, p.N, kt_begin * BK, smem + TILE_BYTES, tid, wr);
        if constexpr (!GLDS) {
            commit_tile(smem, tid, xr);
            commit_tile(smem + TILE_BYTES, tid, wr);
        }
    }
    __syncthreads();

    for (int it = 0; it < nk; ++it) {
        char* cur = smem + (it & 1) * 2 * TILE_BYTES;
        char* nxt = smem + ((it + 1) & 1) * 2 * TILE_BYTES;
        const bool more = (it + 1) < nk;
        if (more) {
            const int k0 = (kt_begin + it + 1) * BK;
            stage_tile<GLDS>(p.X, p.ldx, m0, p.M, k0, nxt, tid, xr);
            stage_tile<GLDS>(p.W, p.ldw, n0, p.N, k0, nxt + TILE_BYTES, tid, wr);
        }
        const char* xs = cur;
        const char* ws = cur + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8_t wf[2], xf[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                wf[b] = lds_read_frag(ws, tile64_off(wn * 64 + b * 32 + l31, ks * 2 + hi));
                xf[b] = lds_read_frag(xs, tile64_off(wm * 64 + b * 32 + l31, ks * 2 + hi));
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = mfma_bf16_32x32x16(wf[a], xf[b], acc[a][b]);
        }
        if constexpr (!GLDS) {
            if (more) {
                commit_tile(nxt, tid, xr);
                commit_tile(nxt + TILE_BYTES, tid, wr);
            }
        }
        __syncthreads();
    }

    // Epilogue.  acc[a][b][r] = Out[m][n] with m = m0 + wm*64 + b*32 + l31, n = n0 + wn*64 + a*32 + acc_row(r,hi)
    const bool add_bias = (p.bias != nullptr) && (sk == 0);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int m = m0 + wm * 64 + b * 32 + l31;
        if (m >= p.M) continue;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wn * 64 + a * 32 + 8 * q + 4 * hi;
                if (n >= p.N) continue;  // N % 4 == 0 (launcher): a quad is all-in or all-out
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[a][b][4 * q + e] * p.alpha;
                if (add_bias) {
                    const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
                    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                }
                if constexpr (OUT_MODE == OUT_BF16) {
                    uint2 pk;
                    pk.x = pack_bf16x2(v[0], v[1]);
                    pk.y = pack_bf16x2(v[2], v[3]);
                    *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.Out) + (size_t)m * p.ldo + n) = pk;
                } else if constexpr (OUT_MODE == OUT_F32) {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.Out) + (size_t)m * p.ldo + n) =
                        make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    float* o = reinterpret_cast<float*>(p.Out) + (size_t)m * p.ldo + n;
#pragma unroll
                    for (int e = 0; e < 4; ++e) unsafeAtomicAdd(o + e, v[e]);
                }
            }
        }
    }
}


// =====================================================================================================================
// v2: 256x128x64 tile, 8 waves (4 along M x 2 along N, 64x64 per wave), 3-stage LDS ring filled by LDS-DMA.
// One raw s_barrier per K-tile; the DMA of tile t+2 is issued right after the barrier of iteration t and only has to
// land two iterations later (counted vmcnt keeps tile t+1's six DMA instructions in flight ACROSS the barrier -- guide
// §5 "Pipelining across barriers").  All LDS lives in ONE dynamic array (a second __shared__ object makes hipcc drain
// vmcnt(0) in front of every ds_read).
// =====================================================================================================================
constexpr int V2_BM = 256, V2_BN = 128;
constexpr int V2_STAGE = (V2_BM + V2_BN) * BK * 2;  // 48 KiB
constexpr int V2_NSTAGE = 3;
enum { OUT_F32_PARTIAL = GEMM_OUT_F32_PARTIAL };

template <int OUT_MODE>
__global__ __launch_bounds__(512, 2) void gemm_bf16_nt_v2_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;

    const int nwg = p.tiles_m * p.tiles_n * p.split_k;
    int lid = xcd_remap(blockIdx.x, nwg);
    const int per_k = p.tiles_m * p.tiles_n;
    const int sk = lid / per_k;
    lid -= sk * per_k;
    int tm, tn;
    if (p.sup_m > 0) {
        // super-tile-major order: the ~32 workgroups an XCD runs concurrently cover sup_m x sup_n tiles that share
        // sup_m X panels and sup_n W panels in that XCD's 4 MiB L2 (sup_* divide tiles_* exactly: a bijection)
        const int per_sup = p.sup_m * p.sup_n;
        const int nsup_n = p.tiles_n / p.sup_n;
        const int sup = lid / per_sup, in = lid - sup * per_sup;
        const int sm_i = sup / nsup_n, sn_i = sup - sm_i * nsup_n;
        tm = sm_i * p.sup_m + in / p.sup_n;
        tn = sn_i * p.sup_n + in % p.sup_n;
    } else {
        tn = lid % p.tiles_n;
        tm = lid / p.tiles_n;
    }
    const int m0 = tm * V2_BM, n0 = tn * V2_BN;
    const int nk_total = p.K / BK;
    const int kt_begin = (int)(((long)nk_total * sk) / p.split_k);
    const int kt_end = (int)(((long)nk_total * (sk + 1)) / p.split_k);
    const int nk = kt_end - kt_begin;

    // per-lane DMA source pointers (advance by one K-tile = 128 B per issue) and wave-uniform LDS slots
    const bf16_t* xsrc[4];
    const bf16_t* wsrc[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (j * 8 + wave) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int gr = m0 + r;
        gr = gr < p.M ? gr : p.M - 1;
        xsrc[j] = p.X + (size_t)gr * p.ldx + (size_t)kt_begin * BK + c * 8;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (j * 8 + wave) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int gr = n0 + r;
        gr = gr < p.N ? gr : p.N - 1;
        wsrc[j] = p.W + (size_t)gr * p.ldw + (size_t)kt_begin * BK + c * 8;
    }
    auto issue = [&](int stage) {
        char* base = dsm + stage * V2_STAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __builtin_amdgcn_global_load_lds((glb_void_ptr)xsrc[j], (lds_void_ptr)(base + (j * 8 + wave) * 1024), 16, 0, 0);
            xsrc[j] += BK;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            __builtin_amdgcn_global_load_lds((glb_void_ptr)wsrc[j],
                                             (lds_void_ptr)(base + V2_BM * BK * 2 + (j * 8 + wave) * 1024), 16, 0, 0);
            wsrc[j] += BK;
        }
    };

    f32x16_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    if (nk > 0) issue(0);
    if (nk > 1) issue(1);
    int st_cur = 0, st_fill = 2;
    for (int t = 0; t < nk; ++t) {
        if (t + 1 < nk) {
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // tile t landed; tile t+1 (6 DMA ops) may still fly
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();  // everyone's share of tile t is in LDS; everyone is done reading stage st_fill
        if (t + 2 < nk && !(p.dbg & 1)) issue(st_fill);
        const char* xs = dsm + st_cur * V2_STAGE;
        const char* ws = xs + V2_BM * BK * 2;
        if (!(p.dbg & 2)) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8_t wf[2], xf[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                wf[b] = lds_read_frag(ws, tile64_off(wn * 64 + b * 32 + l31, ks * 2 + hi));
                xf[b] = lds_read_frag(xs, tile64_off(wm * 64 + b * 32 + l31, ks * 2 + hi));
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = mfma_bf16_32x32x16(wf[a], xf[b], acc[a][b]);
        }
        __builtin_amdgcn_s_setprio(0);
        }
        st_cur = (st_cur == V2_NSTAGE - 1) ? 0 : st_cur + 1;
        st_fill = (st_fill == V2_NSTAGE - 1) ? 0 : st_fill + 1;
    }

    if constexpr (OUT_MODE == OUT_BF16) {
        if ((p.ldo & 7) == 0) {
            // LDS-staged epilogue: every 128-B output line leaves the CU as ONE coalesced 8-lane x 16-B store instead of
            // eight 16-B fragments issued by eight different instructions (partial-line writes cost an L2 request each:
            // ~4 us per 256-tile round measured in round 1).  Each wave stages its own 64x64 tile; rows are padded to
            // 144 B so the column-strided 8-B writes spread over the banks and the 16-B reads stay aligned.
            constexpr int ROWB = 144;
            const bool add_bias2 = (p.bias != nullptr) && (sk == 0);
            __builtin_amdgcn_s_barrier();  // every wave is done reading the operand stages
            char* my = dsm + wave * (64 * ROWB);
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int nl = a * 32 + 8 * q + 4 * hi;
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[a][b][4 * q + e] * p.alpha;
                        if (add_bias2) {
                            const int n = n0 + wn * 64 + nl;
                            if (n < p.N) {
                                const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
                                v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                            }
                        }
                        uint2 pk;
                        pk.x = pack_bf16x2(v[0], v[1]);
                        pk.y = pack_bf16x2(v[2], v[3]);
                        *reinterpret_cast<uint2*>(my + (b * 32 + l31) * ROWB + nl * 2) = pk;
                    }
            // LDS is in-order per wave: the reads below see this wave's own writes without a barrier
#pragma unroll
            for (int ps = 0; ps < 8; ++ps) {
                const int row = ps * 8 + (lane >> 3), ch = lane & 7;
                const int m = m0 + wm * 64 + row, n = n0 + wn * 64 + ch * 8;
                const uint4 vv = *reinterpret_cast<const uint4*>(my + row * ROWB + ch * 16);
                if (m < p.M) {
                    bf16_t* dst = reinterpret_cast<bf16_t*>(p.Out) + (size_t)m * p.ldo + n;
                    if (n + 8 <= p.N) {
                        *reinterpret_cast<uint4*>(dst) = vv;
                    } else if (n < p.N) {  // ragged N tail (N % 4 == 0): first half of the chunk only
                        *reinterpret_cast<uint2*>(dst) = make_uint2(vv.x, vv.y);
                    }
                }
            }
            return;
        }
    }
    const bool add_bias = (p.bias != nullptr) && (sk == 0);
    float* part = nullptr;
    if constexpr (OUT_MODE == OUT_F32_PARTIAL) part = reinterpret_cast<float*>(p.Out) + (size_t)sk * p.M * p.ldo;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int m = m0 + wm * 64 + b * 32 + l31;
        if (m >= p.M) continue;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
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
                if constexpr (OUT_MODE == OUT_BF16) {
                    uint2 pk;
                    pk.x = pack_bf16x2(v[0], v[1]);
                    pk.y = pack_bf16x2(v[2], v[3]);
                    *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.Out) + (size_t)m * p.ldo + n) = pk;
                } else if constexpr (OUT_MODE == OUT_F32) {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.Out) + (size_t)m * p.ldo + n) =
                        make_float4(v[0], v[1], v[2], v[3]);
                } else if constexpr (OUT_MODE == OUT_F32_PARTIAL) {
                    *reinterpret_cast<float4*>(part + (size_t)m * p.ldo + n) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    float* o = reinterpret_cast<float*>(p.Out) + (size_t)m * p.ldo + n;
#pragma unroll
                    for (int e = 0; e < 4; ++e) unsafeAtomicAdd(o + e, v[e]);
                }
            }
        }
    }
}

// =====================================================================================================================
// wgrad in its natural "TN" form:  G[o][i] (+)= sum_t dY[t][o] * A[t][i]   with dY:(T,O) and A:(T,I) row-major, i.e. the
// reduction index t is the SLOW index of both operands.  Same 256x128x64 / 8-wave / 3-stage LDS-DMA pipeline as v2, but
//   * a K-tile is 64 token rows: [64][256] bf16 of dY (512-B rows) + [64][128] of A (256-B rows), DMA'd with fully
//     coalesced row segments (no explicit transposes, no transposed scratch);
//   * MFMA fragments (8 consecutive t for one feature) are built with ds_read_b64_tr_b16: within a 16-lane group, lane p
//     points at [t0 + p/4][f0 + 4*(p%4) .. +3] and receives column f0 + p, rows t0..t0+3 (pattern pinned on hardware by
//     cx_probe_ds_read_tr16); two reads give the 8 k-values of a 32x32x16 operand;
//   * rows t..t+3 of one 64-B column group would share banks, so the 16-B chunk index is XORed with (t&3)<<2 -- applied
//     on the DMA source address, undone in the read address (guide §5.4 rule 21).
// Tokens T..Tp-1 (Tp = round_up(T,64)) must be zero in both operands (the engine clears those rows).
// =====================================================================================================================
constexpr int TN_XROW = V2_BM * 2;  // 512 B per token row of the dY tile
constexpr int TN_WROW = V2_BN * 2;  // 256 B per token row of the A tile

typedef __attribute__((address_space(3))) bf16x4_t* lds_b4_ptr;

// A/B fragment for feature block starting at column f0 (32 features) and token block t0 (16 tokens) of a tile whose
// token rows are ROWB bytes: lane (g = lane>>4, p = lane&15) -> features f0 + 16*(g&1) + p, tokens t0 + 8*(g>>1) + {0..7}
template <int ROWB>
CX_DEVICE bf16x8_t tn_frag(const char* tile, int f0, int t0, int lane) {
    const int g = lane >> 4, p = lane & 15;
    const int t = t0 + 8 * (g >> 1) + (p >> 2);
    const int f = f0 + 16 * (g & 1) + 4 * (p & 3);
    union { bf16x4_t h[2]; bf16x8_t v; } u;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int tt = t + 4 * half;
        const int chunk = (f >> 3) ^ ((tt & 3) << 2);
        const char* addr = tile + tt * ROWB + chunk * 16 + (f & 4) * 2;
        u.h[half] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4_ptr)addr);
    }
    return u.v;
}

__global__ __launch_bounds__(512, 2) void gemm_bf16_tn_kernel(GemmParams p) {
    // p.X = dY (T, M=O) ldx, p.W = A (T, N=I) ldw, p.K = Tp tokens, Out = fp32 partial slabs [split][M][N]
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;

    const int nwg = p.tiles_m * p.tiles_n * p.split_k;
    int lid = xcd_remap(blockIdx.x, nwg);
    const int tn = lid % p.tiles_n;
    lid /= p.tiles_n;
    const int tm = lid % p.tiles_m;
    const int sk = lid / p.tiles_m;
    const int m0 = tm * V2_BM, n0 = tn * V2_BN;
    const int nk_total = p.K / BK;
    const int kt_begin = (int)(((long)nk_total * sk) / p.split_k);
    const int kt_end = (int)(((long)nk_total * (sk + 1)) / p.split_k);
    const int nk = kt_end - kt_begin;

    // DMA sources.  dY tile: instruction q (0..31) covers token rows 2q, 2q+1 (32 lanes x 16 B each);
    // A tile: instruction q (0..15) covers token rows 4q..4q+3 (16 lanes each).
    const bf16_t* xsrc[4];
    const bf16_t* wsrc[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int t = (j * 8 + wave) * 2 + (lane >> 5);
        int c = (lane & 31) ^ ((t & 3) << 2);
        int col = m0 + c * 8;
        col = col + 8 <= p.M ? col : (p.M >= 8 ? p.M - 8 : 0);  // ragged feature tail: any in-bounds chunk (masked at store)
        xsrc[j] = p.X + ((size_t)kt_begin * BK + t) * p.ldx + col;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int t = (j * 8 + wave) * 4 + (lane >> 4);
        int c = (lane & 15) ^ ((t & 3) << 2);
        int col = n0 + c * 8;
        col = col + 8 <= p.N ? col : (p.N >= 8 ? p.N - 8 : 0);
        wsrc[j] = p.W + ((size_t)kt_begin * BK + t) * p.ldw + col;
    }
    const size_t xstep = (size_t)BK * p.ldx, wstep = (size_t)BK * p.ldw;
    auto issue = [&](int stage) {
        char* base = dsm + stage * V2_STAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __builtin_amdgcn_global_load_lds((glb_void_ptr)xsrc[j], (lds_void_ptr)(base + (j * 8 + wave) * 1024), 16, 0, 0);
            xsrc[j] += xstep;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            __builtin_amdgcn_global_load_lds((glb_void_ptr)wsrc[j],
                                             (lds_void_ptr)(base + BK * TN_XROW + (j * 8 + wave) * 1024), 16, 0, 0);
            wsrc[j] += wstep;
        }
    };

    f32x16_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    if (nk > 0) issue(0);
    if (nk > 1) issue(1);
    int st_cur = 0, st_fill = 2;
    for (int t = 0; t < nk; ++t) {
        if (t + 1 < nk) {
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (t + 2 < nk) issue(st_fill);
        const char* xs = dsm + st_cur * V2_STAGE;
        const char* ws = xs + BK * TN_XROW;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8_t wf[2], xf[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                wf[b] = tn_frag<TN_WROW>(ws, wn * 64 + b * 32, ks * 16, lane);
                xf[b] = tn_frag<TN_XROW>(xs, wm * 64 + b * 32, ks * 16, lane);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = mfma_bf16_32x32x16(wf[a], xf[b], acc[a][b]);
        }
        __builtin_amdgcn_s_setprio(0);
        st_cur = (st_cur == V2_NSTAGE - 1) ? 0 : st_cur + 1;
        st_fill = (st_fill == V2_NSTAGE - 1) ? 0 : st_fill + 1;
    }

    // acc[a][b][r]: m (out feature) = m0 + wm*64 + b*32 + l31, n (in feature) = n0 + wn*64 + a*32 + acc_row(r,hi)
    float* part = reinterpret_cast<float*>(p.Out) + (size_t)sk * p.M * p.ldo;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int m = m0 + wm * 64 + b * 32 + l31;
        if (m >= p.M) continue;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wn * 64 + a * 32 + 8 * q + 4 * hi;
                if (n >= p.N) continue;
                *reinterpret_cast<float4*>(part + (size_t)m * p.ldo + n) =
                    make_float4(acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]);
            }
    }
}

// out[i] += sum_s part[s][i]   (float4 per thread; fixed summation order -> deterministic wgrad)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                            long n4, long slab, int splits) {
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

template <int OUT_MODE>
hipError_t launch_v2_mode(const GemmParams& p, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_nt_v2_kernel<OUT_MODE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, V2_NSTAGE * V2_STAGE);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int nwg = p.tiles_m * p.tiles_n * p.split_k;
    hipLaunchKernelGGL((gemm_bf16_nt_v2_kernel<OUT_MODE>), dim3(nwg), dim3(512), V2_NSTAGE * V2_STAGE, stream, p);
    return hipGetLastError();
}

// split-K factor for `tiles` output tiles on 256 CUs (one resident workgroup per CU): maximise the fill of whole
// rounds, lightly penalising extra partial slabs (each costs one more pass of the reduction kernel).
long choose_split(long tiles, long max_split, int slots = 256) {
    if (max_split < 1) max_split = 1;
    long best = 1;
    double best_score = -1.0;
    for (long s = 1; s <= max_split; ++s) {
        const long wgs = tiles * s;
        const long rounds = (wgs + slots - 1) / slots;
        const double score = (double)wgs / (double)(rounds * slots) - 0.006 * (double)s;
        if (score > best_score + 1e-9) {
            best_score = score;
            best = s;
        }
    }
    return best;
}

int largest_divisor_le(int n, int cap) {
    for (int d = cap < n ? cap : n; d >= 1; --d)
        if (n % d == 0) return d;
    return 1;
}

hipError_t launch_v2(const GemmParams& p_in, int out_mode, hipStream_t stream) {
    GemmParams p = p_in;
    p.sup_n = largest_divisor_le(p.tiles_n, 8);
    p.sup_m = largest_divisor_le(p.tiles_m, 32 / p.sup_n);
    if (p.sup_m * p.sup_n < 8) p.sup_m = p.sup_n = 0;  // awkward factorisation: keep the row-major order
    switch (out_mode) {
        case OUT_BF16: return launch_v2_mode<OUT_BF16>(p, stream);
        case OUT_F32: return launch_v2_mode<OUT_F32>(p, stream);
        case OUT_F32_PARTIAL: return launch_v2_mode<OUT_F32_PARTIAL>(p, stream);
        default: return launch_v2_mode<OUT_F32_ATOMIC>(p, stream);
    }
}

int g_dbg = 0;
int g_variant = 5;  // 5: v5 256x256x64 (default; 3 and 4 -- generations deleted in round 6 -- alias to it);  2: v2 (256x128, 3-stage ring);  1: v1 (128x128, 2-stage)

int g_use_glds = 1;

// ---- sampled per-launch timing (bench.py's live roofline measurement) --------------------------------------
// Every `stride`-th launch is bracketed by two HIP events recorded on the launch stream; cx_prof_gemm_collect()
// turns them into (sum of durations, sum of algorithmic FLOPs) for exactly the sampled launches.
struct GemmProf {
    bool enabled = false;
    int stride = 1;
    long launches = 0;
    static constexpr int CAP = 8192;
    hipEvent_t ev0[CAP], ev1[CAP];
    double flop[CAP];
    int created = 0, used = 0;
} g_prof;

template <bool GLDS>
hipError_t launch_mode(const GemmParams& p, int out_mode, hipStream_t stream) {
    const int nwg = p.tiles_m * p.tiles_n * p.split_k;
    dim3 grid(nwg), block(256);
    switch (out_mode) {
        case OUT_BF16: hipLaunchKernelGGL((gemm_bf16_nt_kernel<GLDS, OUT_BF16>), grid, block, 0, stream, p); break;
        case OUT_F32: hipLaunchKernelGGL((gemm_bf16_nt_kernel<GLDS, OUT_F32>), grid, block, 0, stream, p); break;
        default: hipLaunchKernelGGL((gemm_bf16_nt_kernel<GLDS, OUT_F32_ATOMIC>), grid, block, 0, stream, p); break;
    }
    return hipGetLastError();
}

}  // namespace

extern "C" {

void cx_gemm_set_trace(void* buf) { cx_gemm_v5_set_trace(static_cast<long long*>(buf)); }
void cx_gemm_v6_trace(void* buf) { cx_gemm_v6_set_trace(static_cast<long long*>(buf)); }
void cx_gemm_v6_ablate(int mask) { cx_gemm_v6_set_ablate(mask); }
void cx_gemm_v7_mode(int mode) { cx_gemm_v7_set_mode(mode); }
void cx_gemm_v7_trace(void* buf) { cx_gemm_v7_set_trace(static_cast<long long*>(buf)); }
int cx_gemm_v7_occupancy(void) { return cx_gemm_v7_occupancy_query(); }
void cx_gemm_v7_ablate(int mask) { cx_gemm_v7_set_ablate(mask); }
void cx_gemm_v7_flags(int f) { cx_gemm_v7_set_flags(f); }
void cx_gemm_v7_period(int cycles) { cx_gemm_v7_set_period(cycles); }
void cx_gemm_set_debug(int d) {
    g_dbg = d;
    cx_gemm_v5_set_persistent((d & 4) == 0);
    cx_gemm_v6_force_groups((d >> 8) & 15);  // bits 8..11: force the v6 XCD-grid N-group count (1, 2, 4, 8)  // bit2: run the 256x256 kernel one-tile-per-workgroup (A/B of the persistent walk)
}
void cx_gemm_set_variant(int v) {  // 6 = variant 5 with the one-wave-per-SIMD kernel (v6) for the persistent NT forms
    cx_gemm_v5_set_use_v6(v == 6);
    g_variant = (v >= 1 && v <= 5) ? v : 5;  // out-of-range (incl. 6) -> the default family
}
int cx_gemm_get_variant(void) { return (g_variant == 5 && cx_gemm_v5_get_use_v6()) ? 6 : g_variant; }
void cx_gemm_set_glds(int enable) { g_use_glds = enable ? 1 : 0; }
int cx_gemm_get_glds(void) { return g_use_glds; }

int cx_gemm_bf16_nt(const uint16_t* X, const uint16_t* W, void* Out, const float* bias, int M, int N, int K, int ldx,
                    int ldw, int ldo, int out_mode, int split_k, float alpha, void* stream) {
    if (M <= 0 || N <= 0) return CX_OK;
    if (K <= 0 || (K % BK) != 0 || (N % 4) != 0) return CX_ERR_SHAPE;
    if ((ldx % 8) != 0 || (ldw % 8) != 0) return CX_ERR_ALIGN;
    if (out_mode < 0 || out_mode > 2) return CX_ERR_ARG;
    if (out_mode == OUT_BF16 && (ldo % 4) != 0) return CX_ERR_ALIGN;
    if (out_mode != OUT_BF16 && (ldo % 4) != 0) return CX_ERR_ALIGN;
    GemmParams p;
    p.X = X; p.W = W; p.Out = Out; p.bias = bias;
    p.M = M; p.N = N; p.K = K; p.ldx = ldx; p.ldw = ldw; p.ldo = ldo;
    const bool v5 = (g_variant >= 3) && g_use_glds && out_mode != OUT_F32_ATOMIC;
    const bool v2 = ((g_variant == 2) || (g_variant >= 3 && !v5)) && g_use_glds;
    p.tiles_m = v2 ? (M + V2_BM - 1) / V2_BM : (M + BM - 1) / BM;
    p.tiles_n = v2 ? (N + V2_BN - 1) / V2_BN : (N + BN - 1) / BN;
    const int nk = K / BK;
    if (split_k < 1) split_k = 1;
    if (split_k > nk) split_k = nk;
    if (out_mode != OUT_F32_ATOMIC) split_k = 1;  // only the accumulating epilogue can combine K slices
    p.split_k = split_k;
    p.alpha = alpha;
    p.dbg = g_dbg;
    p.Out2 = nullptr;
    p.ldo2 = 0;
    p.sup_m = p.sup_n = 0;
    int slot = -1;
    if (g_prof.enabled) {
        if ((g_prof.launches % g_prof.stride) == 0 && g_prof.used < GemmProf::CAP) {
            slot = g_prof.used++;
            if (slot >= g_prof.created) {
                if (hipEventCreate(&g_prof.ev0[slot]) != hipSuccess || hipEventCreate(&g_prof.ev1[slot]) != hipSuccess)
                    return CX_ERR_LAUNCH;
                g_prof.created = slot + 1;
            }
            g_prof.flop[slot] = 2.0 * (double)M * (double)N * (double)K;
            (void)hipEventRecord(g_prof.ev0[slot], (hipStream_t)stream);
        }
        ++g_prof.launches;
    }
    hipError_t e = v5 ? cx_launch_gemm_v5(p, 0, out_mode, GEMM_EPI_NONE, (hipStream_t)stream)
                   : v2 ? launch_v2(p, out_mode, (hipStream_t)stream)
                      : (g_use_glds ? launch_mode<true>(p, out_mode, (hipStream_t)stream)
                                    : launch_mode<false>(p, out_mode, (hipStream_t)stream));
    if (slot >= 0) (void)hipEventRecord(g_prof.ev1[slot], (hipStream_t)stream);
    return e == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
}

// Out(M,N) fp32 += X W^T with the K range split over `split_k` workgroup groups whose fp32 partial tiles go to `ws`
// with plain stores, followed by one fixed-order reduction pass (no device-scope atomics: those serialise at the
// memory fabric because the XCD L2s are not coherent -- 235 us floor per launch measured in round 1).
int cx_gemm_bf16_nt_accum(const uint16_t* X, const uint16_t* W, float* Out, float* ws, long ws_floats, int M, int N,
                          int K, int ldx, int ldw, void* stream) {
    if (M <= 0 || N <= 0) return CX_OK;
    if (K <= 0 || (K % BK) != 0 || (N % 4) != 0) return CX_ERR_SHAPE;
    if (!ws || !Out) return CX_ERR_ARG;
    const long slab = (long)M * N;
    if (ws_floats < slab) return CX_ERR_SHAPE;
    const bool v5 = (g_variant >= 3) && g_use_glds;
    const bool v2 = (g_variant == 2) && g_use_glds;
    const long tiles = v5 ? (long)((M + 255) / 256) * ((N + 255) / 256)
                     : v2 ? (long)((M + V2_BM - 1) / V2_BM) * ((N + V2_BN - 1) / V2_BN)
                          : (long)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    const long nk = K / BK;
    long max_split = nk / 4;                               // keep >= 4 K-tiles per slice (pipeline depth)
    if (max_split > ws_floats / slab) max_split = ws_floats / slab;
    if (max_split > 16) max_split = 16;
    long split = choose_split(tiles, max_split, (v2 || v5) ? 256 : 512);
    int rc;
    if (v5) {
        GemmParams p;
        p.X = X; p.W = W; p.Out = ws; p.bias = nullptr;
        p.M = M; p.N = N; p.K = K; p.ldx = ldx; p.ldw = ldw; p.ldo = N;
        p.tiles_m = p.tiles_n = 0; p.split_k = (int)split; p.alpha = 1.f; p.dbg = g_dbg; p.Out2 = nullptr; p.ldo2 = 0; p.sup_m = p.sup_n = 0;
        split = p.split_k;
        const hipError_t he = cx_launch_gemm_v5(p, 0, GEMM_OUT_F32_PARTIAL, GEMM_EPI_NONE, (hipStream_t)stream);
        rc = he == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
    } else if (v2) {
        GemmParams p;
        p.X = X; p.W = W; p.Out = ws; p.bias = nullptr;
        p.M = M; p.N = N; p.K = K; p.ldx = ldx; p.ldw = ldw; p.ldo = N;
        p.tiles_m = (M + V2_BM - 1) / V2_BM; p.tiles_n = (N + V2_BN - 1) / V2_BN;
        p.split_k = (int)split; p.alpha = 1.f; p.dbg = g_dbg; p.Out2 = nullptr; p.ldo2 = 0; p.sup_m = p.sup_n = 0;
        rc = launch_v2(p, OUT_F32_PARTIAL, (hipStream_t)stream) == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
    } else {
        // v1 has no partial epilogue: one slice at a time into its slab via the overwrite mode
        rc = CX_OK;
        for (long s = 0; s < split && rc == CX_OK; ++s) {
            const long k0 = nk * s / split * BK, k1 = nk * (s + 1) / split * BK;
            rc = cx_gemm_bf16_nt(X + k0, W + k0, ws + s * slab, nullptr, M, N, (int)(k1 - k0), ldx, ldw, N, OUT_F32, 1,
                                 1.f, stream);
        }
    }
    if (rc != CX_OK) return rc;
    const long n4 = slab / 4;
    long g = (n4 + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, ws, Out, n4, slab,
                       (int)split);
    return hipGetLastError() == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
}

// wgrad without transposes: G(O,I) fp32 += dY(T,O)^T A(T,I).  Tp = round_up(T,64) rows of both operands are read; rows
// T..Tp-1 MUST be zero.  O % 256 == 0 and I % 128 == 0 (all encoder shapes) -- otherwise use cx_gemm_bf16_nt_accum.
int cx_gemm_bf16_tn_accum(const uint16_t* dY, const uint16_t* A, float* G, float* ws, long ws_floats, int T, int O, int I,
                          int ld_dy, int ld_a, void* stream) {
    if (T <= 0 || O <= 0 || I <= 0) return CX_OK;
    if ((O % V2_BM) != 0 || (I % V2_BN) != 0) return CX_ERR_SHAPE;
    const bool use_v5 = (g_variant == 5) && (I % 256) == 0;
    if ((ld_dy % 8) != 0 || (ld_a % 8) != 0) return CX_ERR_ALIGN;
    if (!ws || !G) return CX_ERR_ARG;
    const long slab = (long)O * I;
    if (ws_floats < slab) return CX_ERR_SHAPE;
    const int Tp = (T + BK - 1) / BK * BK;
    GemmParams p;
    p.X = dY; p.W = A; p.Out = ws; p.bias = nullptr;
    p.M = O; p.N = I; p.K = Tp; p.ldx = ld_dy; p.ldw = ld_a; p.ldo = I;
    p.tiles_m = O / V2_BM; p.tiles_n = use_v5 ? I / 256 : I / V2_BN;
    const long tiles = (long)p.tiles_m * p.tiles_n, nk = Tp / BK;
    long max_split = nk / 4;
    if (max_split > ws_floats / slab) max_split = ws_floats / slab;
    if (max_split > 32) max_split = 32;
    const long split = choose_split(tiles, max_split);
    p.split_k = (int)split; p.alpha = 1.f; p.dbg = g_dbg; p.Out2 = nullptr; p.ldo2 = 0; p.sup_m = p.sup_n = 0;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_tn_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, V2_NSTAGE * V2_STAGE) != hipSuccess)
            return CX_ERR_LAUNCH;
        attr_set = true;
    }
    int slot = -1;
    if (g_prof.enabled) {
        if ((g_prof.launches % g_prof.stride) == 0 && g_prof.used < GemmProf::CAP) {
            slot = g_prof.used++;
            if (slot >= g_prof.created) {
                if (hipEventCreate(&g_prof.ev0[slot]) != hipSuccess || hipEventCreate(&g_prof.ev1[slot]) != hipSuccess)
                    return CX_ERR_LAUNCH;
                g_prof.created = slot + 1;
            }
            g_prof.flop[slot] = 2.0 * (double)T * (double)O * (double)I;
            (void)hipEventRecord(g_prof.ev0[slot], (hipStream_t)stream);
        }
        ++g_prof.launches;
    }
    if (use_v5) {
        if (cx_launch_gemm_v5(p, 1, GEMM_OUT_F32_PARTIAL, GEMM_EPI_NONE, (hipStream_t)stream) != hipSuccess) return CX_ERR_LAUNCH;
    } else {
        hipLaunchKernelGGL(gemm_bf16_tn_kernel, dim3((int)(tiles * split)), dim3(512), V2_NSTAGE * V2_STAGE,
                           (hipStream_t)stream, p);
    }
    if (slot >= 0) (void)hipEventRecord(g_prof.ev1[slot], (hipStream_t)stream);
    if (hipGetLastError() != hipSuccess) return CX_ERR_LAUNCH;
    const long n4 = slab / 4;
    long g = (n4 + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, ws, G, n4, slab, (int)split);
    return hipGetLastError() == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
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
    GemmParams p;
    p.X = X; p.W = W; p.Out = YG; p.bias = nullptr;
    p.M = M; p.N = 2 * I; p.K = K; p.ldx = ldx; p.ldw = ldw; p.ldo = ld_yg;
    p.tiles_m = p.tiles_n = 0; p.split_k = 1; p.alpha = 1.f; p.dbg = g_dbg;
    p.Out2 = Act; p.ldo2 = ld_act; p.sup_m = p.sup_n = 0;
    int slot = -1;
    if (g_prof.enabled) {
        if ((g_prof.launches % g_prof.stride) == 0 && g_prof.used < GemmProf::CAP) {
            slot = g_prof.used++;
            if (slot >= g_prof.created) {
                if (hipEventCreate(&g_prof.ev0[slot]) != hipSuccess || hipEventCreate(&g_prof.ev1[slot]) != hipSuccess)
                    return CX_ERR_LAUNCH;
                g_prof.created = slot + 1;
            }
            g_prof.flop[slot] = 2.0 * (double)M * (double)(2 * I) * (double)K;
            (void)hipEventRecord(g_prof.ev0[slot], (hipStream_t)stream);
        }
        ++g_prof.launches;
    }
    hipError_t e = cx_launch_gemm_v5(p, 0, GEMM_OUT_BF16, GEMM_EPI_SWIGLU, (hipStream_t)stream);
    if (slot >= 0) (void)hipEventRecord(g_prof.ev1[slot], (hipStream_t)stream);
    return e == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
}

// Out (M,N) bf16 = bf16(bf16(X W^T + bias) + Residual): a projection whose output feeds `x0 + residual -> LayerNorm`
// (out_proj and fc2 of every block).  CX_ERR_SHAPE when the one-wave-per-SIMD kernel does not cover the shape.
int cx_gemm_bf16_nt_residual(const uint16_t* X, const uint16_t* W, uint16_t* Out, const float* bias,
                             const uint16_t* Residual, int M, int N, int K, int ldx, int ldw, int ldo, int ldr,
                             void* stream) {
    if (M <= 0 || N <= 0) return CX_OK;
    if (!Residual || !Out) return CX_ERR_ARG;
    if (K <= 0 || (K % 64) != 0 || (N % 8) != 0 || (ldo % 8) != 0 || (ldr % 8) != 0) return CX_ERR_SHAPE;
    if ((ldx % 8) != 0 || (ldw % 8) != 0) return CX_ERR_ALIGN;
    if (cx_gemm_get_variant() != 6) return CX_ERR_SHAPE;
    GemmParams p;
    p.X = X; p.W = W; p.Out = Out; p.bias = bias;
    p.M = M; p.N = N; p.K = K; p.ldx = ldx; p.ldw = ldw; p.ldo = ldo;
    p.tiles_m = p.tiles_n = 0; p.split_k = 1; p.alpha = 1.f; p.dbg = g_dbg;
    p.Out2 = const_cast<uint16_t*>(Residual); p.ldo2 = ldr; p.sup_m = p.sup_n = 0; p.trace = nullptr;
    int slot = -1;
    if (g_prof.enabled) {
        if ((g_prof.launches % g_prof.stride) == 0 && g_prof.used < GemmProf::CAP) {
            slot = g_prof.used++;
            if (slot >= g_prof.created) {
                if (hipEventCreate(&g_prof.ev0[slot]) != hipSuccess || hipEventCreate(&g_prof.ev1[slot]) != hipSuccess)
                    return CX_ERR_LAUNCH;
                g_prof.created = slot + 1;
            }
            g_prof.flop[slot] = 2.0 * (double)M * (double)N * (double)K;
            (void)hipEventRecord(g_prof.ev0[slot], (hipStream_t)stream);
        }
        ++g_prof.launches;
    }
    const hipError_t e = cx_launch_gemm_v6(p, GEMM_EPI_NONE, (hipStream_t)stream);
    if (slot >= 0) (void)hipEventRecord(g_prof.ev1[slot], (hipStream_t)stream);
    return e == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
}

// fc1 of the plain (GELU) MLP with bias + erf-GELU fused into the epilogue.  Pre: (M, N) bf16 pre-activation = X W^T +
// bias (optional, may be NULL: the no-grad pass), Act: (M, N) bf16 = gelu(Pre).  Returns CX_ERR_SHAPE when the fused
// kernel does not cover the shape (caller then runs GEMM + cx_bias_gelu_fwd).
int cx_gemm_bf16_bias_gelu(const uint16_t* X, const uint16_t* W, const float* bias, uint16_t* Pre, uint16_t* Act, int M,
                           int N, int K, int ldx, int ldw, int ld_pre, int ld_act, void* stream) {
    return cx_gemm_bf16_bias_act(X, W, bias, Pre, Act, M, N, K, ldx, ldw, ld_pre, ld_act, 0, stream);
}

int cx_gemm_bf16_bias_act(const uint16_t* X, const uint16_t* W, const float* bias, uint16_t* Pre, uint16_t* Act, int M,
                          int N, int K, int ldx, int ldw, int ld_pre, int ld_act, int act, void* stream) {
    if (M <= 0 || N <= 0) return CX_OK;
    if (!Act || (act != 0 && act != 1)) return CX_ERR_ARG;
    if (K <= 0 || (K % 64) != 0 || (N % 8) != 0 || (ld_act % 8) != 0 || (Pre && (ld_pre % 8) != 0)) return CX_ERR_SHAPE;
    if ((ldx % 8) != 0 || (ldw % 8) != 0) return CX_ERR_ALIGN;
    if (cx_gemm_get_variant() != 6) return CX_ERR_SHAPE;
    GemmParams p;
    p.X = X; p.W = W; p.Out = Pre; p.bias = bias;
    p.M = M; p.N = N; p.K = K; p.ldx = ldx; p.ldw = ldw; p.ldo = ld_pre;
    p.tiles_m = p.tiles_n = 0; p.split_k = 1; p.alpha = 1.f; p.dbg = g_dbg;
    p.Out2 = Act; p.ldo2 = ld_act; p.sup_m = p.sup_n = 0; p.trace = nullptr;
    p.act = act;
    int slot = -1;
    if (g_prof.enabled) {
        if ((g_prof.launches % g_prof.stride) == 0 && g_prof.used < GemmProf::CAP) {
            slot = g_prof.used++;
            if (slot >= g_prof.created) {
                if (hipEventCreate(&g_prof.ev0[slot]) != hipSuccess || hipEventCreate(&g_prof.ev1[slot]) != hipSuccess)
                    return CX_ERR_LAUNCH;
                g_prof.created = slot + 1;
            }
            g_prof.flop[slot] = 2.0 * (double)M * (double)N * (double)K;
            (void)hipEventRecord(g_prof.ev0[slot], (hipStream_t)stream);
        }
        ++g_prof.launches;
    }
    const hipError_t e = cx_launch_gemm_v6(p, act == 1 ? GEMM_EPI_QGELU : GEMM_EPI_GELU, (hipStream_t)stream);
    if (slot >= 0) (void)hipEventRecord(g_prof.ev1[slot], (hipStream_t)stream);
    return e == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
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
    if (cx_gemm_get_variant() != 6) return CX_ERR_SHAPE;
    GemmParams p;
    p.X = dY; p.W = W; p.Out = dYG; p.bias = nullptr;
    p.M = M; p.N = I; p.K = K; p.ldx = ldx; p.ldw = ldw; p.ldo = ld_yg;
    p.tiles_m = p.tiles_n = 0; p.split_k = 1; p.alpha = 1.f; p.dbg = g_dbg;
    p.Out2 = const_cast<uint16_t*>(YG); p.ldo2 = ld_yg; p.sup_m = p.sup_n = 0; p.trace = nullptr;
    int slot = -1;
    if (g_prof.enabled) {
        if ((g_prof.launches % g_prof.stride) == 0 && g_prof.used < GemmProf::CAP) {
            slot = g_prof.used++;
            if (slot >= g_prof.created) {
                if (hipEventCreate(&g_prof.ev0[slot]) != hipSuccess || hipEventCreate(&g_prof.ev1[slot]) != hipSuccess)
                    return CX_ERR_LAUNCH;
                g_prof.created = slot + 1;
            }
            g_prof.flop[slot] = 2.0 * (double)M * (double)I * (double)K;
            (void)hipEventRecord(g_prof.ev0[slot], (hipStream_t)stream);
        }
        ++g_prof.launches;
    }
    const hipError_t e = cx_launch_gemm_v6(p, GEMM_EPI_SWIGLU_BWD, (hipStream_t)stream);
    if (slot >= 0) (void)hipEventRecord(g_prof.ev1[slot], (hipStream_t)stream);
    return e == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
}

int cx_prof_gemm_config(int enable, int stride) {
    g_prof.enabled = enable != 0;
    g_prof.stride = stride > 0 ? stride : 1;
    g_prof.launches = 0;
    g_prof.used = 0;
    return CX_OK;
}

int cx_prof_gemm_collect(double* total_ms, double* total_flop, long* launches_timed, long* launches_total) {
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

// ---- round 3: compact save of the gated MLP (see gemm_api.hip for the product's entry points).  The dev library serves
// them on the v6 kernel only (there is no A/B generation of these epilogues), without the launch profiler.
int cx_gemm_bf16_swiglu_gate(const uint16_t* X, const uint16_t* W, uint16_t* G, uint16_t* Act, int M, int I, int K, int ldx,
                             int ldw, int ld_g, int ld_act, void* stream) {
    if (M <= 0 || I <= 0) return CX_OK;
    if (K <= 0 || (K % BK) != 0 || (I % 32) != 0 || (ld_g % 8) != 0 || (ld_act % 8) != 0) return CX_ERR_SHAPE;
    if ((ldx % 8) != 0 || (ldw % 8) != 0) return CX_ERR_ALIGN;
    if (!Act) return CX_ERR_ARG;
    GemmParams p;
    p.X = X; p.W = W; p.Out = G; p.bias = nullptr;
    p.M = M; p.N = 2 * I; p.K = K; p.ldx = ldx; p.ldw = ldw; p.ldo = ld_g;
    p.tiles_m = p.tiles_n = 0; p.split_k = 1; p.alpha = 1.f; p.dbg = 0; p.act = 0;
    p.Out2 = Act; p.ldo2 = ld_act; p.In3 = nullptr; p.sup_m = p.sup_n = 0; p.trace = nullptr;
    return cx_launch_gemm_v6(p, GEMM_EPI_SWIGLU_G, (hipStream_t)stream) == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
}

int cx_gemm_bf16_swiglu_bwd_gate(const uint16_t* dY, const uint16_t* W, const uint16_t* Act, const uint16_t* G, uint16_t* dYG,
                                 int M, int I, int K, int ldx, int ldw, int ld_ag, int ld_dyg, void* stream) {
    if (M <= 0 || I <= 0) return CX_OK;
    if (!dY || !W || !Act || !G || !dYG) return CX_ERR_ARG;
    if (K <= 0 || (K % 64) != 0 || (I % 256) != 0 || (ld_dyg % 8) != 0 || ld_dyg < 2 * I || (ld_ag % 8) != 0 || ld_ag < I)
        return CX_ERR_SHAPE;
    if ((ldx % 8) != 0 || (ldw % 8) != 0) return CX_ERR_ALIGN;
    GemmParams p;
    p.X = dY; p.W = W; p.Out = dYG; p.bias = nullptr;
    p.M = M; p.N = I; p.K = K; p.ldx = ldx; p.ldw = ldw; p.ldo = ld_dyg;
    p.tiles_m = p.tiles_n = 0; p.split_k = 1; p.alpha = 1.f; p.dbg = 0; p.act = 0;
    p.Out2 = const_cast<uint16_t*>(Act); p.ldo2 = ld_ag; p.In3 = G; p.sup_m = p.sup_n = 0; p.trace = nullptr;
    return cx_launch_gemm_v6(p, GEMM_EPI_SWIGLU_BWD_AG, (hipStream_t)stream) == hipSuccess ? CX_OK : CX_ERR_LAUNCH;
}

// dev-library twin of gemm_api.hip's cx_gemm_bf16_act_bwd (fc2 dgrad + GELU / quick_gelu backward + bias-gradient partials; v6 only)
int cx_gemm_bf16_act_bwd(const uint16_t* dY, const uint16_t* W, const uint16_t* Pre, uint16_t* dPre, float* dbias, float* ws,
                         long ws_floats, int M, int N, int K, int ldx, int ldw, int ld_pre, int ld_dpre, int act, void* stream) {
    if (M <= 0 || N <= 0) return CX_OK;
    if (!dY || !W || !Pre || !dPre || (act != 0 && act != 1)) return CX_ERR_ARG;
    if (K <= 0 || (K % 64) != 0 || (N % 8) != 0 || (ld_pre % 8) != 0 || (ld_dpre % 8) != 0) return CX_ERR_SHAPE;
    if ((ldx % 8) != 0 || (ldw % 8) != 0) return CX_ERR_ALIGN;
    if (cx_gemm_get_variant() != 6) return CX_ERR_SHAPE;
    const int nblocks = (M + 127) / 128;
    if (dbias && (!ws || ws_floats < (long)nblocks * N)) return CX_ERR_SHAPE;
    GemmParams p = {};
    p.X = dY; p.W = W; p.Out = dPre; p.bias = nullptr;
    p.M = M; p.N = N; p.K = K; p.ldx = ldx; p.ldw = ldw; p.ldo = ld_dpre;
    p.split_k = 1; p.alpha = 1.f; p.act = act;
    p.Out2 = const_cast<uint16_t*>(Pre); p.ldo2 = ld_pre;
    p.colsum_part = dbias ? ws : nullptr;
    if (cx_launch_gemm_v6(p, GEMM_EPI_ACT_BWD, (hipStream_t)stream) != hipSuccess) return CX_ERR_LAUNCH;
    if (dbias && cx_launch_colsum_part_reduce(ws, dbias, nblocks, N, (hipStream_t)stream) != hipSuccess) return CX_ERR_LAUNCH;
    return CX_OK;
}

#include "gemm_splitk_small.inc"
