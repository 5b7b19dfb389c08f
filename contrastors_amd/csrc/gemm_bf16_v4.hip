// gemm_bf16_v4.hip -- EXPERIMENTAL (cx_gemm_set_variant(4); not the default): persistent bf16 MFMA GEMM, fourth generation: v3's cross-tile prefetch pipeline on v2's tile
// geometry (256x128x64, full 128-B DMA segments, 3-stage ring) with loader/storer wave specialisation.  Round-1 measurement:
// parity-green but no faster than v2 (636 vs 691 TFLOP/s over the encoder shapes at M=32768), so v2 stays the default;
// kept as the starting point for the 8-phase schedule planned next.  See gemm_bf16_v3.hip for the persistent walk.
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

constexpr int TBM = 256, TBN = 128, TBK = 64;
constexpr int X_BYTES = TBM * TBK * 2;           // 32 KiB
constexpr int ST_BYTES = (TBM + TBN) * TBK * 2;  // 48 KiB
constexpr int NST = 3;


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
__global__ __launch_bounds__(512, 2) void gemm_bf16_nt_v4_kernel(GemmParams p, Sched sc) {
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    // Wave specialisation for memory traffic (all 8 waves do MFMA work):
    //   waves 4-7 ("loaders") issue every LDS-DMA instruction and are the only ones that ever wait on vmcnt;
    //   waves 0-3 ("storers") issue every global store of the bf16 epilogues (from the LDS staging area).
    // vmcnt is per wave and stores may retire out of order w.r.t. loads, so keeping the two kinds on different waves
    // makes the loaders' counted waits exact and lets the output stream drain underneath the next tile's K loop.
    const bool is_loader = wave >= 4;
    const int lw = wave & 3;

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
    const bf16_t* xsrc[8];
    const bf16_t* wsrc[4];
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
        if (is_loader) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {  // X tile: 32 KiB = 32 DMA instructions, 8 per loader wave
                const int r = (j * 4 + lw) * 8 + (lane >> 3);
                const int c = (lane & 7) ^ ((r >> 1) & 7);
                int gx = ld_it.m0 + r;
                gx = gx < p.M ? gx : p.M - 1;
                xsrc[j] = p.X + (size_t)gx * p.ldx + (size_t)ld_it.kt0 * TBK + c * 8;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {  // W tile: 16 KiB = 16 DMA instructions, 4 per loader wave
                const int r = (j * 4 + lw) * 8 + (lane >> 3);
                const int c = (lane & 7) ^ ((r >> 1) & 7);
                int gw = ld_it.n0 + r;
                gw = gw < p.N ? gw : p.N - 1;
                wsrc[j] = p.W + (size_t)gw * p.ldw + (size_t)ld_it.kt0 * TBK + c * 8;
            }
        }
    };
    auto issue = [&](int stage) {  // 12 LDS-DMA instructions per LOADER wave: 8 KiB of X and 4 KiB of W
        char* base = dsm + stage * ST_BYTES;
        if (is_loader) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                __builtin_amdgcn_global_load_lds((glb_void_ptr)xsrc[j], (lds_void_ptr)(base + (j * 4 + lw) * 1024), 16, 0, 0);
                xsrc[j] += TBK;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                __builtin_amdgcn_global_load_lds((glb_void_ptr)wsrc[j],
                                                 (lds_void_ptr)(base + X_BYTES + (j * 4 + lw) * 1024), 16, 0, 0);
                wsrc[j] += TBK;
            }
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

    f32x16_t acc[2][2];  // [n-block a][m-block b]
    auto zero_acc = [&]() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
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
    // Epilogue stores share vmcnt with the DMA groups and may retire out of order w.r.t. them.  So at a tile end we
    // first wait vmcnt(0) (the in-flight groups are >= 1 iteration old: cheap), remember how many groups are thereby
    // KNOWN to have landed, issue the stores, and skip the counted wait for that many iterations.  Later counted waits
    // are conservative in the presence of still-pending stores (loads retire in order among loads).
    int landed = 0;

#pragma unroll 1
    while (cp_live) {
        // wait until the oldest in-flight DMA group (the one we are about to read) has landed
        if (is_loader) {
            if (landed > 0) {
                --landed;
            } else if (inflight <= 1) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(12)" ::: "memory");  // the newer group (12 DMA ops) may still fly
            }
        }
        __builtin_amdgcn_s_barrier();
        if (ld_live) {
            issue(st_fill);
            ++inflight;
        }
        const char* xs = dsm + st_cur * ST_BYTES;
        const char* ws = xs + X_BYTES;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8_t wf[2], xf[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) wf[a] = lds_read_frag(ws, tile64_off(wn * 64 + a * 32 + l31, ks * 2 + hi));
#pragma unroll
            for (int b = 0; b < 2; ++b) xf[b] = lds_read_frag(xs, tile64_off(wm * 64 + b * 32 + l31, ks * 2 + hi));
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = mfma_bf16_32x32x16(wf[a], xf[b], acc[a][b]);
        }
        __builtin_amdgcn_s_setprio(0);
        --inflight;
        st_cur = (st_cur == NST - 1) ? 0 : st_cur + 1;
        st_fill = (st_fill == NST - 1) ? 0 : st_fill + 1;

        if (++cp_kt == cp_it.nk) {
            // ---- epilogue of this tile (the DMA of the next tiles' first K-steps is already in flight) ----------
            const bool add_bias = (p.bias != nullptr) && (cp_it.sk == 0);
            if constexpr (OUT_MODE == GEMM_OUT_BF16 && EPI == GEMM_EPI_NONE) {
                if ((p.ldo & 7) == 0) {
                    // Staged through the stage consumed last (st_fill after rotation; 48 KiB): per half b every wave parks
                    // 32 rows x 64 cols (144-B padded rows) in its slot, then storer wave w streams the slots of waves 2w and
                    // 2w+1 to memory as whole 128-B lines.
                    constexpr int ROWB = 144, SLOT = 32 * ROWB;
                    char* area = dsm + st_fill * ST_BYTES;
                    __builtin_amdgcn_s_barrier();  // every wave is done reading this stage
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        char* my = area + wave * SLOT;
#pragma unroll
                        for (int a = 0; a < 2; ++a)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int nl = a * 32 + 8 * q + 4 * hi;
                                float v[4];
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = acc[a][b][4 * q + e] * p.alpha;
                                if (add_bias) {
                                    const int n = cp_it.n0 + wn * 64 + nl;
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
                        __builtin_amdgcn_s_barrier();  // slots complete
                        if (!is_loader) {
#pragma unroll
                            for (int sw = 0; sw < 2; ++sw) {
                                const int src_wave = wave * 2 + sw;
                                const int swm = src_wave >> 1, swn = src_wave & 1;
                                const char* slot = area + src_wave * SLOT;
#pragma unroll
                                for (int ps = 0; ps < 4; ++ps) {
                                    const int row = ps * 8 + (lane >> 3), ch = lane & 7;
                                    const int m = cp_it.m0 + swm * 64 + b * 32 + row, n = cp_it.n0 + swn * 64 + ch * 8;
                                    const uint4 vv = *reinterpret_cast<const uint4*>(slot + row * ROWB + ch * 16);
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
                        }
                        if (b == 0) __builtin_amdgcn_s_barrier();  // slots drained into registers before half 1 overwrites
                    }
                    zero_acc();
                    cp_kt = 0;
                    ++cp_round;
                    cp_setup();
                    continue;
                }
            }
            // direct-store epilogues (fp32 / SwiGLU / unaligned ld): every wave stores, so the loaders first make sure
            // their in-flight DMA groups have landed and then skip the counted wait for that many iterations
            if (is_loader) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                landed = inflight;
            }
            float* part = nullptr;
            if constexpr (OUT_MODE == GEMM_OUT_F32_PARTIAL)
                part = reinterpret_cast<float*>(p.Out) + (size_t)cp_it.sk * p.M * p.ldo;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int m = cp_it.m0 + wm * 64 + b * 32 + l31;
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
        }
    }
}

template <int OUT_MODE, int EPI>
hipError_t launch_one(const GemmParams& p, const Sched& sc, int grid, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_nt_v4_kernel<OUT_MODE, EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, NST * ST_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_bf16_nt_v4_kernel<OUT_MODE, EPI>), dim3(grid), dim3(512), NST * ST_BYTES, stream, p, sc);
    return hipGetLastError();
}

}  // namespace

// Called by gemm_bf16.hip.  p.tiles_* / split_k are recomputed here for the 256x128 tile.
hipError_t cx_launch_gemm_v4(GemmParams p, int out_mode, int epi, hipStream_t stream) {
    if ((p.K % TBK) != 0) return hipErrorInvalidValue;
    Sched sc;
    sc.tiles_m = (p.M + TBM - 1) / TBM;
    sc.tiles_n = (p.N + TBN - 1) / TBN;
    sc.split_k = p.split_k < 1 ? 1 : p.split_k;
    sc.nk_total = p.K / TBK;
    if (sc.split_k > sc.nk_total) sc.split_k = sc.nk_total;
    sc.sn = sc.tiles_n < 8 ? sc.tiles_n : 8;
    sc.sm = 32 / sc.sn;
    if (sc.sm > sc.tiles_m) sc.sm = sc.tiles_m;
    sc.nsup_n = (sc.tiles_n + sc.sn - 1) / sc.sn;
    sc.per_super = sc.sm * sc.sn;
    const int nsup_m = (sc.tiles_m + sc.sm - 1) / sc.sm;
    sc.total_pos = nsup_m * sc.nsup_n * sc.per_super;
    p.split_k = sc.split_k;
    const long work = (long)sc.total_pos * sc.split_k;
    int grid = 256;                       // one persistent workgroup per CU (144 KiB of LDS each)
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
