// gemm_bf16_v3.hip -- persistent bf16 MFMA GEMM for the encoder's FusedDense shapes (K9), third generation.
//
// What round-1 measurements said about v2 (256x128x64 tile, one workgroup per tile, 3-stage LDS-DMA ring):
//   * with the K loop's DMA removed the loop runs at ~75 % of the MFMA bound, with the MFMA removed the DMA alone
//     takes as long as the full kernel -> L2->LDS bytes per FLOP are the first limiter;
//   * K = 768 tiles have only 12 K-steps: pipeline fill + epilogue drain cost ~35 % at one block per CU.
// v3 therefore (a) uses a 256x256 output tile (8 waves as 2x4, 128x64 per wave): 32 KiB of operand bytes per 16
// MFMAs/wave instead of 48 KiB; (b) is PERSISTENT: 256 workgroups walk the tile list and the 4-stage LDS ring keeps
// prefetching across tile boundaries, so the next tile's first K-steps are already in flight while the epilogue
// stores drain; (c) orders tiles in 8x4 super-tiles per XCD so the 32 CUs sharing an L2 share operand panels;
// (d) optionally fuses SwiGLU into the epilogue (fc11/fc12 rows interleaved by 32 in the fused weight).
#include "cx_common.h"
#include "../../include/contrastors_hip.h"
#include "gemm_params.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* glb_void_ptr;

constexpr int TBM = 256, TBN = 256, TBK = 32;
constexpr int X_BYTES = TBM * TBK * 2;           // 16 KiB
constexpr int ST_BYTES = (TBM + TBN) * TBK * 2;  // 32 KiB
constexpr int NST = 4;

// [rows][32 bf16] tile: 4 chunks of 16 B per 64-B row, chunk c of row r at r*64 + ((c ^ ((r>>2)&3)) << 4)
// (ds_read_b128 16-lane groups cover rows distinct mod 16 -> 16 distinct 16-B slots of the 256-B bank row).
CX_DEVICE int t32_off(int r, int c) { return r * 64 + ((c ^ ((r >> 2) & 3)) << 4); }

struct Item {
    int m0, n0, sk, kt0, nk;
};

// Work list: (super-tile-major tile order) x split_k.  Position `pos` -> tile; positions outside the matrix are
// skipped (ragged super-tiles).  Returns false when pos is past the end.
struct Sched {
    int tiles_m, tiles_n, split_k, nk_total;
    int sm, sn, nsup_n, per_super, total_pos;
    __device__ bool decode(long work, Item& it, bool& valid) const {
        const long total = (long)total_pos * split_k;
        if (work >= total) return false;
        const int sk = (int)(work / total_pos);
        const int pos = (int)(work - (long)sk * total_pos);
        const int sup = pos / per_super, in = pos - sup * per_super;
        const int sup_m = sup / nsup_n, sup_n = sup - sup_m * nsup_n;
        const int tm = sup_m * sm + in / sn, tn = sup_n * sn + in % sn;
        valid = (tm < tiles_m) && (tn < tiles_n);
        it.m0 = tm * TBM;
        it.n0 = tn * TBN;
        it.sk = sk;
        it.kt0 = (int)(((long)nk_total * sk) / split_k);
        it.nk = (int)(((long)nk_total * (sk + 1)) / split_k) - it.kt0;
        if (it.nk <= 0) valid = false;
        return true;
    }
};

template <int OUT_MODE, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_bf16_nt_v3_kernel(GemmParams p, Sched sc) {
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 2, wn = wave & 3;

    // XCD-aware walk: block b runs on XCD b%8; the 32 blocks of an XCD take 32 consecutive positions per round.
    const int per_xcd = gridDim.x >> 3;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const long stride = (long)per_xcd * 8;
    auto work_of = [&](long round) { return (round * 8 + xcd) * per_xcd + idx; };

    // ---- load cursor -------------------------------------------------------------------------------------------
    long ld_round = 0;
    Item ld_it;
    int ld_kt = 0;
    bool ld_live = false;
    const bf16_t* xsrc[2];
    const bf16_t* wsrc[2];
    auto ld_setup = [&]() {  // find the next valid item for the load cursor and build its per-lane DMA pointers
        ld_live = false;
        for (;;) {
            bool valid = false;
            if (!sc.decode(work_of(ld_round), ld_it, valid)) return;
            if (valid) break;
            ++ld_round;
        }
        ld_live = true;
        ld_kt = 0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = (j * 8 + wave) * 16 + (lane >> 2);
            const int c = (lane & 3) ^ ((r >> 2) & 3);
            int gx = ld_it.m0 + r, gw = ld_it.n0 + r;
            gx = gx < p.M ? gx : p.M - 1;
            gw = gw < p.N ? gw : p.N - 1;
            xsrc[j] = p.X + (size_t)gx * p.ldx + (size_t)ld_it.kt0 * TBK + c * 8;
            wsrc[j] = p.W + (size_t)gw * p.ldw + (size_t)ld_it.kt0 * TBK + c * 8;
        }
    };
    auto issue = [&](int stage) {  // 4 LDS-DMA instructions per wave: 2 KiB of X and 2 KiB of W
        char* base = dsm + stage * ST_BYTES;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            __builtin_amdgcn_global_load_lds((glb_void_ptr)xsrc[j], (lds_void_ptr)(base + (j * 8 + wave) * 1024), 16, 0, 0);
            xsrc[j] += TBK;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            __builtin_amdgcn_global_load_lds((glb_void_ptr)wsrc[j],
                                             (lds_void_ptr)(base + X_BYTES + (j * 8 + wave) * 1024), 16, 0, 0);
            wsrc[j] += TBK;
        }
        if (++ld_kt == ld_it.nk) {
            ++ld_round;
            ld_setup();
        }
    };

    // ---- compute cursor ----------------------------------------------------------------------------------------
    long cp_round = 0;
    Item cp_it;
    bool cp_live = false;
    auto cp_setup = [&]() {
        cp_live = false;
        for (;;) {
            bool valid = false;
            if (!sc.decode(work_of(cp_round), cp_it, valid)) return;
            if (valid) break;
            ++cp_round;
        }
        cp_live = true;
    };

    f32x16_t acc[2][4];  // [n-block a][m-block b]
    auto zero_acc = [&]() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    };
    zero_acc();

    ld_setup();
    cp_setup();
    int inflight = 0;  // DMA groups issued and not yet consumed
#pragma unroll 1
    for (int i = 0; i < NST - 1; ++i)
        if (ld_live) {
            issue(i);
            ++inflight;
        }
    int st_cur = 0, st_fill = NST - 1;
    int cp_kt = 0;
    bool drained = false;  // true right after epilogue stores were issued: the next wait must be vmcnt(0)

#pragma unroll 1
    while (cp_live) {
        // wait until the oldest in-flight DMA group (the one we are about to read) has landed
        if (drained || inflight <= 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (inflight == 2) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
        drained = false;
        __builtin_amdgcn_s_barrier();
        if (ld_live) {
            issue(st_fill);
            ++inflight;
        }
        const char* xs = dsm + st_cur * ST_BYTES;
        const char* ws = xs + X_BYTES;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t wf[2], xf[4];
#pragma unroll
            for (int a = 0; a < 2; ++a) wf[a] = lds_read_frag(ws, t32_off(wn * 64 + a * 32 + l31, ks * 2 + hi));
#pragma unroll
            for (int b = 0; b < 4; ++b) xf[b] = lds_read_frag(xs, t32_off(wm * 128 + b * 32 + l31, ks * 2 + hi));
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = mfma_bf16_32x32x16(wf[a], xf[b], acc[a][b]);
        }
        __builtin_amdgcn_s_setprio(0);
        --inflight;
        st_cur = (st_cur == NST - 1) ? 0 : st_cur + 1;
        st_fill = (st_fill == NST - 1) ? 0 : st_fill + 1;

        if (++cp_kt == cp_it.nk) {
            // ---- epilogue of this tile (the DMA of the next tiles' first K-steps is already in flight) ----------
            const bool add_bias = (p.bias != nullptr) && (cp_it.sk == 0);
            float* part = nullptr;
            if constexpr (OUT_MODE == GEMM_OUT_F32_PARTIAL)
                part = reinterpret_cast<float*>(p.Out) + (size_t)cp_it.sk * p.M * p.ldo;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int m = cp_it.m0 + wm * 128 + b * 32 + l31;
                if (m < p.M) {
                    if constexpr (EPI == GEMM_EPI_SWIGLU) {
                        // a = 0: y rows, a = 1: gate rows of the same 32 activation columns (interleaved weight)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int n = cp_it.n0 + wn * 64 + 8 * q + 4 * hi;  // column of y in the fused output
                            if (n < p.N) {
                                float y[4], g[4], o[4];
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    y[e] = acc[0][b][4 * q + e];
                                    g[e] = acc[1][b][4 * q + e];
                                }
                                if (p.Out) {
                                    bf16_t* row = reinterpret_cast<bf16_t*>(p.Out) + (size_t)m * p.ldo;
                                    uint2 pk;
                                    pk.x = pack_bf16x2(y[0], y[1]); pk.y = pack_bf16x2(y[2], y[3]);
                                    *reinterpret_cast<uint2*>(row + n) = pk;
                                    pk.x = pack_bf16x2(g[0], g[1]); pk.y = pack_bf16x2(g[2], g[3]);
                                    *reinterpret_cast<uint2*>(row + n + 32) = pk;
                                }
                                // the standalone op rounds y and gate to bf16 first (they are FusedDense outputs)
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const float yy = bf16_to_f32(f32_to_bf16(y[e])), gg = bf16_to_f32(f32_to_bf16(g[e]));
                                    o[e] = gg / (1.f + __expf(-gg)) * yy;
                                }
                                const int col = ((cp_it.n0 + wn * 64) >> 1) + 8 * q + 4 * hi;
                                uint2 pk;
                                pk.x = pack_bf16x2(o[0], o[1]); pk.y = pack_bf16x2(o[2], o[3]);
                                *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.Out2) + (size_t)m * p.ldo2 + col) = pk;
                            }
                        }
                    } else {
#pragma unroll
                        for (int a = 0; a < 2; ++a)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int n = cp_it.n0 + wn * 64 + a * 32 + 8 * q + 4 * hi;
                                if (n < p.N) {
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
                                        *reinterpret_cast<float4*>(part + (size_t)m * p.ldo + n) =
                                            make_float4(v[0], v[1], v[2], v[3]);
                                    }
                                }
                            }
                    }
                }
            }
            zero_acc();
            cp_kt = 0;
            ++cp_round;
            cp_setup();
            drained = true;  // stores share vmcnt with the DMA groups and may retire out of order w.r.t. them
        }
    }
}

template <int OUT_MODE, int EPI>
hipError_t launch_one(const GemmParams& p, const Sched& sc, int grid, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_nt_v3_kernel<OUT_MODE, EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, NST * ST_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_bf16_nt_v3_kernel<OUT_MODE, EPI>), dim3(grid), dim3(512), NST * ST_BYTES, stream, p, sc);
    return hipGetLastError();
}

}  // namespace

// Called by gemm_bf16.hip.  p.tiles_* / split_k are recomputed here for the 256x256 tile.
hipError_t cx_launch_gemm_v3(GemmParams p, int out_mode, int epi, hipStream_t stream) {
    if ((p.K % TBK) != 0) return hipErrorInvalidValue;
    Sched sc;
    sc.tiles_m = (p.M + TBM - 1) / TBM;
    sc.tiles_n = (p.N + TBN - 1) / TBN;
    sc.split_k = p.split_k < 1 ? 1 : p.split_k;
    sc.nk_total = p.K / TBK;
    if (sc.split_k > sc.nk_total) sc.split_k = sc.nk_total;
    sc.sn = sc.tiles_n < 4 ? sc.tiles_n : 4;
    sc.sm = 32 / sc.sn;
    if (sc.sm > sc.tiles_m) sc.sm = sc.tiles_m;
    sc.nsup_n = (sc.tiles_n + sc.sn - 1) / sc.sn;
    sc.per_super = sc.sm * sc.sn;
    const int nsup_m = (sc.tiles_m + sc.sm - 1) / sc.sm;
    sc.total_pos = nsup_m * sc.nsup_n * sc.per_super;
    p.split_k = sc.split_k;
    const long work = (long)sc.total_pos * sc.split_k;
    int grid = 256;                       // one persistent workgroup per CU (128 KiB of LDS each)
    if (work < grid) grid = (int)((work + 7) / 8 * 8);
    if (grid < 8) grid = 8;
    if (epi == GEMM_EPI_SWIGLU) {
        if (out_mode != GEMM_OUT_BF16) return hipErrorInvalidValue;
        return launch_one<GEMM_OUT_BF16, GEMM_EPI_SWIGLU>(p, sc, grid, stream);
    }
    switch (out_mode) {
        case GEMM_OUT_BF16: return launch_one<GEMM_OUT_BF16, GEMM_EPI_NONE>(p, sc, grid, stream);
        case GEMM_OUT_F32: return launch_one<GEMM_OUT_F32, GEMM_EPI_NONE>(p, sc, grid, stream);
        case GEMM_OUT_F32_PARTIAL: return launch_one<GEMM_OUT_F32_PARTIAL, GEMM_EPI_NONE>(p, sc, grid, stream);
        default: return hipErrorInvalidValue;
    }
}
