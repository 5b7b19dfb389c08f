// gemm_bf16_v7.hip -- persistent 256x128x64 bf16 MFMA GEMM with TWO resident workgroups per CU (round 4).
//
// Why a second structure next to gemm_bf16_v6.hip: v6 gives every SIMD ONE in-order wave that owns all 256 accumulators of
// a 128x128 sub-tile.  Its K = 768 launches with an HBM-sized epilogue (fc2-dgrad + SwiGLU backward: 512 KB of side streams
// per 256x256 tile; fc1 + SwiGLU with the gate save) run main loop and epilogue strictly one after the other -- the
// measured launch time is the SUM of the MFMA-bound and the HBM-bound time (profiles/r3_kernel_summary_gb16384.txt,
// VERDICT r3) -- because nothing else is resident on the SIMD while the lone wave sits in its epilogue.
//
// Here a workgroup is 4 waves as 4(M) x 1(N) on a 256x128 tile: each wave owns a 64x128 sub-tile = 2x4 MFMA 32x32x16
// accumulators = 128 AGPRs, i.e. 256 registers per wave and two workgroups per CU (two waves per SIMD).  The two
// workgroups are INDEPENDENT (own tiles, own 80 KiB of LDS, own barriers): while one streams its epilogue the other's
// main loop has the matrix pipe to itself, and the hardware interleaves the two instruction streams.
//
// LDS map per workgroup (80 KiB):
//   [ 0, 32K)  X, PRIVATE per wave: wave w's 64 token rows x 128 B at w * 8 KiB, ONE slot.  A wave reads the 8 fragments
//              of a K-tile into registers at the start of the K-tile (32 VGPRs), which frees the slot: the DMA of the next
//              K-tile is issued right behind the reads and has the whole K-tile to land.  No barrier is ever needed for X.
//   [32K, 48K) W slot 0   [48K, 64K) spare   [64K, 80K) W slot 1 -- W (128 rows x 128 B) is shared by the 4 waves, two
//              slots, fragments read one k-step ahead, ONE s_barrier per K-tile (behind k-step 2, as in v6).
//   The epilogue stages through the W slot consumed last + the spare region, which are adjacent either way (8 KiB per
//   wave: 32 output rows x 256 B, 16-B chunks XOR-swizzled by the row).
// Operand movement per FLOP is 1.5 x v6's (0.75 fragment reads per MFMA, 48 KiB of LDS-DMA per 256x128x64 step): the plain
// long-K launches stay on v6; this kernel takes the launches whose epilogue v6 cannot hide (routing: cx_launch_gemm_v6).
// Accumulation order per output element is the same k-ascending chain of 32x32x16 MFMAs as v6: results are bit-identical.
#include "cx_common.h"
#include "../../include/contrastors_hip.h"
#include "gemm_params.h"
#include <type_traits>

namespace {

#ifndef CX_V7_NT
#define CX_V7_NT 3   // bit 0: non-temporal epilogue stores, bit 1: non-temporal epilogue side-stream loads (as CX_V6_NT)
#endif

template <int IMM>
__device__ __forceinline__ void v7_dma_m0(uint32_t base) {
    asm volatile("s_add_u32 m0, %0, %1" ::"s"(base), "n"(IMM) : "memory", "scc");
}
__device__ __forceinline__ void v7_dma_ld(uint32_t off, const bf16_t* base) {
    asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base) : "memory");
}
template <int IMM>
__device__ __forceinline__ void v7_dma_full(uint32_t slot_base, uint32_t off, const bf16_t* base) {
    asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3" ::"s"(slot_base), "n"(IMM), "v"(off), "s"(base)
                 : "memory", "scc");
}
// Epilogue global traffic goes through buffer instructions: a wave-uniform descriptor (4 SGPRs) + a 32-bit lane offset + a
// uniform SGPR offset per row group -- no 64-bit per-row address registers (the epilogues have 128 VGPRs in all), and the
// descriptor's size does the row predication of a partial last M-panel in hardware (out-of-range loads return 0, stores are
// dropped).  aux = 2 is `nt` on gfx950.
typedef unsigned int v4u7 __attribute__((ext_vector_type(4)));
// (the descriptor inputs go through readfirstlane: they ARE wave-uniform, but unless the compiler can prove it, it wraps every
// buffer instruction in a waterfall loop -- 4 v_readfirstlane + 2 v_cmp + s_and_saveexec per memory operation)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc7(const void* base, uint32_t bytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    const uint32_t nb = __builtin_amdgcn_readfirstlane(bytes);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, (int)nb, 0x00020000);
}
__device__ __forceinline__ uint4 bld7(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    const v4u7 t = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, (CX_V7_NT & 2) ? 2 : 0);
    return make_uint4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ float4 bldf7(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {   // (cached: the bias is re-read by every tile)
    const v4u7 t = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w));
}
__device__ __forceinline__ void bst7(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, uint4 v) {
    const v4u7 t = {v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(t, r, voff, soff, (CX_V7_NT & 1) ? 2 : 0);
}
// a ^ IMM, opaque to the optimiser: the swizzled staging addresses are ONE lane-constant base XOR a compile-time constant;
// left to itself the compiler precomputes every variant at the top of the epilogue (16-24 registers it does not have)
template <uint32_t IMM>
__device__ __forceinline__ uint32_t vxor7(uint32_t a) {
    uint32_t r;
    asm volatile("v_xor_b32 %0, %2, %1" : "=v"(r) : "v"(a), "i"(IMM));
    return r;
}

typedef __attribute__((address_space(3))) void* lds_void_ptr7;

constexpr int BM7 = 256, BN7 = 128, BK7 = 64;
constexpr int XW7 = 8192;                                   // a wave's private X slot: 64 rows x 128 B
constexpr int W0_OFF7 = 32768, SP_OFF7 = 49152, W1_OFF7 = 65536;
constexpr int LDS7 = 81920;

// Accumulators: block i = 4b + a (b: m-block 0..1, a: n-block 0..3) is the f32x16 variable acc[i], tied to the MFMA asm
// through an "a" (AGPR) constraint.  Unlike v6 (whose 256 accumulators are physical registers hidden from the compiler)
// they must be VISIBLE here: at 128 VGPRs the register allocator parks live values in whatever AGPR it believes free, and
// eight live f32x16 values own all 128 AGPRs of the wave's budget.  The MFMAs stay inline asm (AGPR placement forced, issue
// order pinned); the compiler reads the results in the epilogue with its own v_accvgpr_read.
#define V7_MFMA(i, w, x) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[(i)]) : "v"(w), "v"(x))
#define V7_MFMA_Z(i, w, x) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(acc[(i)]) : "v"(w), "v"(x))

struct XFrags7 {
    bf16x8_t f[2][4];   // [m-block][k-step]: the wave's X operand of one whole K-tile
};
struct WFrags7 {
    bf16x8_t f[4];      // [n-block]: W operand of one k-step
};

// DBG (ablation instantiations, dev library only; results are garbage, timing is the point): bit0 no LDS-DMA in the K loop,
// bit1 no barrier, bit2 no W fragment reads, bit3 no MFMA, bit4 no epilogue, bit5 no X fragment reads, bit6 no vmcnt waits.
template <int EPI, int DBG = 0>
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_bf16_v7_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    constexpr bool IS_BWD = EPI == GEMM_EPI_SWIGLU_BWD_AG;
    constexpr bool IS_SWIGLU = EPI == GEMM_EPI_SWIGLU_G;
    static_assert(EPI == GEMM_EPI_NONE || IS_BWD || IS_SWIGLU, "v7 serves the plain / residual, SwiGLU (gate save) and SwiGLU-backward (act, gate) forms");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifndef CX_PRODUCT
    long long t_begin = 0;
    if (p.trace) t_begin = (long long)__builtin_amdgcn_s_memtime();
#endif
    const int hi_m = lane >> 5, l31_m = lane & 31;   // (main loop; the epilogue derives its own from an opaque copy)
    const int nk = p.K / BK7;
    // Tile order: as v6 (XCD (xi, xj) of a gm x gn grid owns M-panels [m_lo, m_hi) x N-tiles [n_lo, n_lo + n_wd), walked
    // n-fastest, per_xcd consecutive tiles per round) with 64 workgroups per XCD and 128-wide N-tiles.
    const int per_xcd = gridDim.x >> 3;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int gn = p.sup_n > 0 ? p.sup_n : 1, gm = 8 / gn;
    const int xi = xcd / gn, xj = xcd - xi * gn;
    const int m_lo = p.tiles_m * xi / gm, m_hi = p.tiles_m * (xi + 1) / gm;
    const int n_lo = p.tiles_n * xj / gn, n_wd = p.tiles_n * (xj + 1) / gn - n_lo;
    const int nloc = (m_hi - m_lo) * n_wd;
    auto tile_of = [&](int round) {   // (tm << 8) | tn, -1 = none
        const int l = round * per_xcd + idx;
        if (l >= nloc) return -1;
        const int q = l / n_wd;
        return ((m_lo + q) << 8) | (n_lo + (l - q * n_wd));
    };

    // ---- DMA cursors (see gemm_bf16_v6.hip): SGPR base advanced by 128 B per K-tile + one constant VGPR byte offset per
    // instruction.  X: 8 instructions of 8 rows x 128 B per K-tile, all of them this wave's own rows (64 w .. 64 w + 63);
    // W: 16 instructions per K-tile, instruction q = 4 j + wave covers W rows 8 q .. 8 q + 7 (j = 0..3).  Past its last
    // tile a cursor re-walks the workgroup's first tile (dummy traffic into slots nobody reads) instead of branching.
    uint32_t xoff[8], woff[8];   // (woff: 4 used; sized 8 so that the discarded branch of V7_LD indexes inside the array)
    const bf16_t* xbase = p.X;
    const bf16_t* wbase = p.W;
    int lx_round = 0, lx_kt = 0;
    int lw_round = 0, lw_kt = 0, lw_slot = 0;
    const int first_tile = tile_of(0);
    bool x_clamped = true, w_clamped = true;
    auto x_setup = [&](int tile) {
        const int tm = tile >> 8;
        xbase = p.X + (size_t)tm * BM7 * p.ldx;
        const int rows = p.M - tm * BM7;
        if (rows < BM7 || x_clamped) {
            x_clamped = rows < BM7;
            int lz = lane;                                      // (opaque: nothing of this is hoisted to kernel entry and spilled)
            asm volatile("" : "+v"(lz));
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int rl = j * 8 + (lz >> 3);              // row inside the wave's slot
                const int c = (lz & 7) ^ ((rl >> 1) & 7);      // source chunk that lands in slot chunk (lane & 7): tile64_off
                const int r = wave * 64 + rl;
                const int rr = r < rows ? r : rows - 1;
                xoff[j] = (uint32_t)rr * (uint32_t)p.ldx * 2u + c * 16;
            }
        }
    };
    auto w_setup = [&](int tile) {
        const int tn = tile & 255;
        wbase = p.W + (size_t)tn * BN7 * p.ldw;
        const int rows = p.N - tn * BN7;
        if (rows < BN7 || w_clamped) {
            w_clamped = rows < BN7;
            int lz = lane;
            asm volatile("" : "+v"(lz));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = (j * 4 + wave) * 8 + (lz >> 3);
                const int c = (lz & 7) ^ ((r >> 1) & 7);
                const int rr = r < rows ? r : rows - 1;
                woff[j] = (uint32_t)rr * (uint32_t)p.ldw * 2u + c * 16;
            }
        }
    };
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_ptr7)dsm;
    const uint32_t x_m0 = lds0 + wave * XW7;                 // X instruction j lands at x_m0 + j * 1024
    const uint32_t w_lane = lds0 + wave * 1024;              // W instruction j lands at slot + (4 j + wave) * 1024
    uint32_t w_m0 = w_lane + W0_OFF7;
    auto x_advance = [&]() {
        xbase += BK7;
        if (++lx_kt == nk) {
            lx_kt = 0;
            const int t = tile_of(++lx_round);
            x_setup(t >= 0 ? t : first_tile);
        }
    };
    auto w_advance = [&]() {
        wbase += BK7;
        lw_slot ^= 1;
        w_m0 = w_lane + (lw_slot ? W1_OFF7 : W0_OFF7);
        if (++lw_kt == nk) {
            lw_kt = 0;
            const int t = tile_of(++lw_round);
            w_setup(t >= 0 ? t : first_tile);
        }
    };

    int cp_round = 0;
    int cp_tile = first_tile;
    if (cp_tile < 0) return;   // (grid larger than the tile count: never with launch7's grid)
    // Start stagger.  Two workgroups that enter their K loops together share the matrix pipe half-and-half, reach their
    // epilogues together and stay in lock-step for the whole launch (equal tiles, equal rates): main loop and epilogue then
    // never overlap -- measured: launch time = no-epilogue time + epilogue time, profiles/r4_gemm_v7_ablate.txt.  So the
    // workgroups start spread over one tile period (p.sup_m = period / 64 in shader cycles, chosen by the launcher): the two
    // workgroups the dispatcher is observed to co-locate (b and b + grid / 2) half a period apart, the CUs spread over the
    // period so that the chip's epilogue streams do not burst together either.  Speed only: any placement stays correct.
    if (p.sup_m > 0) {
        const int half = gridDim.x >> 1;
        const int b = blockIdx.x;
        const int slot = (((b >= half ? b - half : b) * 37) + (b >= half ? 32 : 0)) & 63;
        for (int i = 0; i < slot * p.sup_m; i += 1024) __builtin_amdgcn_s_sleep(16);   // (s_sleep 16 = 1024 cycles)
    }
    x_setup(cp_tile);
    w_setup(cp_tile);
    // prologue: X of K-tile 0 (own rows) and W of K-tile 0 (slot 0)
    v7_dma_full<0>(x_m0, xoff[0], xbase); v7_dma_full<1024>(x_m0, xoff[1], xbase); v7_dma_full<2048>(x_m0, xoff[2], xbase);
    v7_dma_full<3072>(x_m0, xoff[3], xbase); v7_dma_full<4096>(x_m0, xoff[4], xbase); v7_dma_full<5120>(x_m0, xoff[5], xbase);
    v7_dma_full<6144>(x_m0, xoff[6], xbase); v7_dma_full<7168>(x_m0, xoff[7], xbase);
    x_advance();
    v7_dma_full<0>(w_m0, woff[0], wbase); v7_dma_full<4096>(w_m0, woff[1], wbase); v7_dma_full<8192>(w_m0, woff[2], wbase);
    v7_dma_full<12288>(w_m0, woff[3], wbase);
    w_advance();

    const char* xs = dsm + wave * XW7;
    XFrags7 XF;
    WFrags7 W0, W1;
    f32x16_t acc[8];
    auto read_w = [&](WFrags7& f, const char* ws, int ks, int a) {
        if constexpr ((DBG & 4) != 0) return;
        f.f[a] = lds_read_frag(ws, tile64_off(a * 32 + l31_m, ks * 2 + hi_m));
    };
    auto read_x = [&](int b, int ks) {
        if constexpr ((DBG & 32) != 0) return;
        XF.f[b][ks] = lds_read_frag(xs, tile64_off(b * 32 + l31_m, ks * 2 + hi_m));
    };
    if constexpr ((DBG & (4 | 32 | 8)) != 0) {   // (ablations: the fragments / accumulators are never defined otherwise)
#pragma unroll
        for (int a = 0; a < 4; ++a) { W0.f[a] = bf16x8_t{}; W1.f[a] = bf16x8_t{}; }
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int k = 0; k < 4; ++k) XF.f[b][k] = bf16x8_t{};
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = f32x16_t{};
    }
    int ws_slot = 0;   // W slot of the K-tile being computed
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int a = 0; a < 4; ++a) read_w(W0, dsm + W0_OFF7, 0, a);

    // One MFMA slot of a k-step: [M0 write of a DMA] MFMA [its load] [one fragment read]; order pinned.
#define V7_M0(kind, J)                                                    \
    do {                                                                  \
        if constexpr ((kind) == 1 && (DBG & 1) == 0) v7_dma_m0<(J) * 4096>(w_m0);           \
        if constexpr ((kind) == 2 && (DBG & 1) == 0) v7_dma_m0<(J) * 1024>(x_m0);           \
    } while (0)
#define V7_LD(kind, J)                                                    \
    do {                                                                  \
        if constexpr ((kind) == 1 && (DBG & 1) == 0) v7_dma_ld(woff[(J)], wbase);           \
        if constexpr ((kind) == 2 && (DBG & 1) == 0) v7_dma_ld(xoff[(J)], xbase);           \
    } while (0)
#define V7_SLOT(MMA, i, WC, ks, kind, J, RD)                              \
    do {                                                                  \
        V7_M0(kind, J);                                                   \
        if constexpr ((DBG & 8) == 0) MMA((i), WC.f[(i) & 3], XF.f[(i) >> 2][(ks)]);   \
        V7_LD(kind, J);                                                   \
        RD;                                                               \
        __builtin_amdgcn_sched_barrier(0);                                \
    } while (0)
#define V7_NORD ((void)0)

    // One K-tile.  On entry: W0 = W fragments of (this K-tile, k-step 0); X of this K-tile is in flight or landed (own DMA);
    // the W slot of the NEXT K-tile is free (barrier of the previous K-tile / epilogue barrier).
    auto kt_body = [&](auto first) {
        const char* ws = dsm + (ws_slot ? W1_OFF7 : W0_OFF7);
        const char* nws = dsm + (ws_slot ? W0_OFF7 : W1_OFF7);
        // X of this K-tile has landed: nothing younger than its 8 instructions has been issued by this wave (the W DMA of
        // the next K-tile follows below; epilogue stores are older and retire in order)
        if constexpr ((DBG & 64) == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        read_x(0, 0); read_x(1, 0);
        __builtin_amdgcn_sched_barrier(0);   // (k-step 0's fragments first: the LDS returns a wave's reads in issue order)
        read_x(0, 1); read_x(1, 1); read_x(0, 2); read_x(1, 2); read_x(0, 3); read_x(1, 3);
        __builtin_amdgcn_sched_barrier(0);
        // k-step 0: W fragments of k-step 1; W DMA of the next K-tile (4); then -- once the 8 X reads above have returned
        // (everything but the 4 W reads just issued: lgkmcnt(4)) the private X slot is free -- the first X DMAs of the next
        if constexpr (decltype(first)::value) {
            V7_SLOT(V7_MFMA_Z, 0, W0, 0, 0, 0, read_w(W1, ws, 1, 0));
            V7_SLOT(V7_MFMA_Z, 1, W0, 0, 1, 0, read_w(W1, ws, 1, 1));
            V7_SLOT(V7_MFMA_Z, 2, W0, 0, 1, 1, read_w(W1, ws, 1, 2));
            V7_SLOT(V7_MFMA_Z, 3, W0, 0, 1, 2, read_w(W1, ws, 1, 3));
            V7_SLOT(V7_MFMA_Z, 4, W0, 0, 1, 3, V7_NORD);
            asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            V7_SLOT(V7_MFMA_Z, 5, W0, 0, 2, 0, V7_NORD);
            V7_SLOT(V7_MFMA_Z, 6, W0, 0, 2, 1, V7_NORD);
            V7_SLOT(V7_MFMA_Z, 7, W0, 0, 2, 2, V7_NORD);
        } else {
            V7_SLOT(V7_MFMA, 0, W0, 0, 0, 0, read_w(W1, ws, 1, 0));
            V7_SLOT(V7_MFMA, 1, W0, 0, 1, 0, read_w(W1, ws, 1, 1));
            V7_SLOT(V7_MFMA, 2, W0, 0, 1, 1, read_w(W1, ws, 1, 2));
            V7_SLOT(V7_MFMA, 3, W0, 0, 1, 2, read_w(W1, ws, 1, 3));
            V7_SLOT(V7_MFMA, 4, W0, 0, 1, 3, V7_NORD);
            asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            V7_SLOT(V7_MFMA, 5, W0, 0, 2, 0, V7_NORD);
            V7_SLOT(V7_MFMA, 6, W0, 0, 2, 1, V7_NORD);
            V7_SLOT(V7_MFMA, 7, W0, 0, 2, 2, V7_NORD);
        }
        w_advance();
        // k-step 1: the other 5 X DMAs, W fragments of k-step 2
        V7_SLOT(V7_MFMA, 0, W1, 1, 2, 3, read_w(W0, ws, 2, 0));
        V7_SLOT(V7_MFMA, 1, W1, 1, 2, 4, read_w(W0, ws, 2, 1));
        V7_SLOT(V7_MFMA, 2, W1, 1, 2, 5, read_w(W0, ws, 2, 2));
        V7_SLOT(V7_MFMA, 3, W1, 1, 2, 6, read_w(W0, ws, 2, 3));
        V7_SLOT(V7_MFMA, 4, W1, 1, 2, 7, V7_NORD);
        V7_SLOT(V7_MFMA, 5, W1, 1, 0, 0, V7_NORD);
        V7_SLOT(V7_MFMA, 6, W1, 1, 0, 0, V7_NORD);
        V7_SLOT(V7_MFMA, 7, W1, 1, 0, 0, V7_NORD);
        x_advance();
        // k-step 2: W fragments of k-step 3 (the last reads of this K-tile's W slot)
        V7_SLOT(V7_MFMA, 0, W0, 2, 0, 0, read_w(W1, ws, 3, 0));
        V7_SLOT(V7_MFMA, 1, W0, 2, 0, 0, read_w(W1, ws, 3, 1));
        V7_SLOT(V7_MFMA, 2, W0, 2, 0, 0, read_w(W1, ws, 3, 2));
        V7_SLOT(V7_MFMA, 3, W0, 2, 0, 0, read_w(W1, ws, 3, 3));
        V7_SLOT(V7_MFMA, 4, W0, 2, 0, 0, V7_NORD);
        V7_SLOT(V7_MFMA, 5, W0, 2, 0, 0, V7_NORD);
        V7_SLOT(V7_MFMA, 6, W0, 2, 0, 0, V7_NORD);
        V7_SLOT(V7_MFMA, 7, W0, 2, 0, 0, V7_NORD);
        // this wave's reads of the current W slot are complete, and so are its 4 W DMAs of the next K-tile (everything but
        // the 8 X instructions issued behind them)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr ((DBG & 64) == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        if constexpr ((DBG & 2) == 0) __builtin_amdgcn_s_barrier();
        // k-step 3: W fragments of (next K-tile, k-step 0) from the other slot
        V7_SLOT(V7_MFMA, 0, W1, 3, 0, 0, read_w(W0, nws, 0, 0));
        V7_SLOT(V7_MFMA, 1, W1, 3, 0, 0, read_w(W0, nws, 0, 1));
        V7_SLOT(V7_MFMA, 2, W1, 3, 0, 0, read_w(W0, nws, 0, 2));
        V7_SLOT(V7_MFMA, 3, W1, 3, 0, 0, read_w(W0, nws, 0, 3));
        V7_SLOT(V7_MFMA, 4, W1, 3, 0, 0, V7_NORD);
        V7_SLOT(V7_MFMA, 5, W1, 3, 0, 0, V7_NORD);
        V7_SLOT(V7_MFMA, 6, W1, 3, 0, 0, V7_NORD);
        V7_SLOT(V7_MFMA, 7, W1, 3, 0, 0, V7_NORD);
        ws_slot ^= 1;
    };

#pragma unroll 1
    while (cp_tile >= 0) {
        // (p.dbg bit 0, experiments: the K loop runs at wave priority 1, the epilogue at 0 -- the SIMD's other wave is the other
        // workgroup's; when it is in its epilogue its VALU / VMEM stream then yields issue slots to this wave's MFMA stream)
        if (p.dbg & 1) __builtin_amdgcn_s_setprio(1);
        kt_body(std::true_type{});
#pragma unroll 1
        for (int kt = 1; kt < nk; ++kt) kt_body(std::false_type{});

        // ---- epilogue.  Staging: the W slot consumed last + the spare region (adjacent), 8 KiB per wave.
        // MFMA results -> VALU reads: the hazard recogniser does not see through the inline asm.  The accumulators are
        // operands of the nops so that no accumulator read can be scheduled above them.
        if (p.dbg & 1) __builtin_amdgcn_s_setprio(0);
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]), "+a"(acc[5]), "+a"(acc[6]), "+a"(acc[7]));
        // Every lane-dependent address of the epilogue is derived from `le` (= lane, made opaque per tile): loop-invariant
        // to the compiler they would all be computed at kernel entry and kept live -- i.e. spilled -- across the main loop,
        // whose scratch reloads put s_waitcnt vmcnt(0) in front of the LDS-DMA ring.
        int le = lane;
        asm volatile("" : "+v"(le));
        const int hi = le >> 5, l31 = le & 31;
        const int tn = cp_tile & 255, tm = cp_tile >> 8;
        const int m0 = tm * BM7 + wave * 64, n0 = tn * BN7;
        const int rows_left = p.M - m0;                              // rows of this wave that exist (<= 0: none)
        const uint32_t rows_here = rows_left >= 64 ? 64u : (rows_left > 0 ? (uint32_t)rows_left : 0u);
        // staging region of this wave as a byte offset into the dynamic LDS (ws_slot now names the NEXT K-tile's slot; the
        // one consumed last is the other).  A multiple of 256, so it can be folded into the XOR bases below.
        const uint32_t my = (uint32_t)((ws_slot ? W0_OFF7 : SP_OFF7) + wave * 8192);
        // Staged block [32 rows][256 B]: 16-B chunk c of row r lives at r * 256 + ((c ^ (r & 15)) << 4).
        //   lane (l31, hi) owns row l31, the 8 bytes at half `hi` of chunk c:     dsm + (cb ^ (c << 4))
        //   lane (lrow, lch) moves 16-B chunk lch of row 4 i + lrow:              dsm + (rb ^ ((i & 3) << 6)) + 1024 i
        const uint32_t cb = my + (uint32_t)l31 * 256 + ((l31 & 15) << 4) + hi * 8;
        const int lrow = le >> 4, lch = le & 15;
        const uint32_t rb = my + (uint32_t)lrow * 256 + ((lch ^ lrow) << 4);
#define V7_CELL(c) (dsm + vxor7<((c) << 4)>(cb))
#define V7_ROW(i) (dsm + vxor7<(((i) & 3) << 6)>(rb) + 1024 * (i))
        if constexpr ((DBG & 16) != 0) {
            // ablation: no epilogue
        } else if constexpr (EPI == GEMM_EPI_NONE) {
            // Out = bf16(acc) [+ residual, added in fp32 and rounded once more: the x0 + residual of dropout_add_layer_norm].
            // optional fp32 bias; alpha == 1 (the launcher sends everything else to v6).
            const bf16_t* resid = reinterpret_cast<const bf16_t*>(p.Out2);
            const __amdgpu_buffer_rsrc_t rs_out = rsrc7(reinterpret_cast<bf16_t*>(p.Out) + (size_t)m0 * p.ldo + n0, rows_here * (uint32_t)p.ldo * 2u);
            const uint32_t vo = ((uint32_t)lrow * (uint32_t)p.ldo + lch * 8) * 2u;
            const uint32_t so = (uint32_t)p.ldo * 8u;   // 4 rows of the output, bytes
            auto add_res = [&](uint4& v, const uint4& r) {
                v.x = pack_bf16x2(bf16lo_to_f32(v.x) + bf16lo_to_f32(r.x), bf16hi_to_f32(v.x) + bf16hi_to_f32(r.x));
                v.y = pack_bf16x2(bf16lo_to_f32(v.y) + bf16lo_to_f32(r.y), bf16hi_to_f32(v.y) + bf16hi_to_f32(r.y));
                v.z = pack_bf16x2(bf16lo_to_f32(v.z) + bf16lo_to_f32(r.z), bf16hi_to_f32(v.z) + bf16hi_to_f32(r.z));
                v.w = pack_bf16x2(bf16lo_to_f32(v.w) + bf16lo_to_f32(r.w), bf16hi_to_f32(v.w) + bf16hi_to_f32(r.w));
            };
            auto tile_epi = [&](auto with_resid, auto with_bias) {
                constexpr bool RES = decltype(with_resid)::value;
                constexpr bool BIAS = decltype(with_bias)::value;
                // bias (fp32[N], FusedDense of the biased towers): added in fp32 before the bf16 rounding, as v6 does
                const __amdgpu_buffer_rsrc_t rs_bias = rsrc7(BIAS ? p.bias + n0 : nullptr, BIAS ? (uint32_t)BN7 * 4u : 0u);
                const __amdgpu_buffer_rsrc_t rs_res = rsrc7(RES ? resid + (size_t)m0 * p.ldo2 + n0 : nullptr, RES ? rows_here * (uint32_t)p.ldo2 * 2u : 0u);
                const uint32_t vr = ((uint32_t)lrow * (uint32_t)p.ldo2 + lch * 8) * 2u;
                const uint32_t sr = (uint32_t)p.ldo2 * 8u;
                uint4 r0 = {}, r1 = {}, r2 = {}, r3 = {}, r4 = {}, r5 = {}, r6 = {}, r7 = {};
#define V7_RES_ROWS(b_)                                                                                                     \
    r0 = bld7(rs_res, vr, sr * (8 * (b_) + 0)); r1 = bld7(rs_res, vr, sr * (8 * (b_) + 1)); r2 = bld7(rs_res, vr, sr * (8 * (b_) + 2)); \
    r3 = bld7(rs_res, vr, sr * (8 * (b_) + 3)); r4 = bld7(rs_res, vr, sr * (8 * (b_) + 4)); r5 = bld7(rs_res, vr, sr * (8 * (b_) + 5)); \
    r6 = bld7(rs_res, vr, sr * (8 * (b_) + 6)); r7 = bld7(rs_res, vr, sr * (8 * (b_) + 7));
                auto stage_pass = [&](auto bc) {
                    constexpr int b = decltype(bc)::value;
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        uint2 pk[4];
                        float4 bq[4] = {};
                        if constexpr (BIAS) {   // columns a * 32 + 8 q + 4 hi .. + 3 of the tile
#pragma unroll
                            for (int q = 0; q < 4; ++q) bq[q] = bldf7(rs_bias, (uint32_t)hi * 16u, (uint32_t)(a * 128 + q * 32));
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            pk[q].x = pack_bf16x2(acc[4 * b + a][4 * q] + bq[q].x, acc[4 * b + a][4 * q + 1] + bq[q].y);
                            pk[q].y = pack_bf16x2(acc[4 * b + a][4 * q + 2] + bq[q].z, acc[4 * b + a][4 * q + 3] + bq[q].w);
                        }
                        if (a == 0) { *reinterpret_cast<uint2*>(V7_CELL(0)) = pk[0]; *reinterpret_cast<uint2*>(V7_CELL(1)) = pk[1]; *reinterpret_cast<uint2*>(V7_CELL(2)) = pk[2]; *reinterpret_cast<uint2*>(V7_CELL(3)) = pk[3]; }
                        if (a == 1) { *reinterpret_cast<uint2*>(V7_CELL(4)) = pk[0]; *reinterpret_cast<uint2*>(V7_CELL(5)) = pk[1]; *reinterpret_cast<uint2*>(V7_CELL(6)) = pk[2]; *reinterpret_cast<uint2*>(V7_CELL(7)) = pk[3]; }
                        if (a == 2) { *reinterpret_cast<uint2*>(V7_CELL(8)) = pk[0]; *reinterpret_cast<uint2*>(V7_CELL(9)) = pk[1]; *reinterpret_cast<uint2*>(V7_CELL(10)) = pk[2]; *reinterpret_cast<uint2*>(V7_CELL(11)) = pk[3]; }
                        if (a == 3) { *reinterpret_cast<uint2*>(V7_CELL(12)) = pk[0]; *reinterpret_cast<uint2*>(V7_CELL(13)) = pk[1]; *reinterpret_cast<uint2*>(V7_CELL(14)) = pk[2]; *reinterpret_cast<uint2*>(V7_CELL(15)) = pk[3]; }
                        __builtin_amdgcn_sched_barrier(0);   // one accumulator block at a time (register pressure)
                    }
                };
                auto one_pass = [&](auto bc) {
                    constexpr int b = decltype(bc)::value;
                    uint4 v0 = *reinterpret_cast<const uint4*>(V7_ROW(0)), v1 = *reinterpret_cast<const uint4*>(V7_ROW(1)),
                          v2 = *reinterpret_cast<const uint4*>(V7_ROW(2)), v3 = *reinterpret_cast<const uint4*>(V7_ROW(3)),
                          v4 = *reinterpret_cast<const uint4*>(V7_ROW(4)), v5 = *reinterpret_cast<const uint4*>(V7_ROW(5)),
                          v6 = *reinterpret_cast<const uint4*>(V7_ROW(6)), v7 = *reinterpret_cast<const uint4*>(V7_ROW(7));
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (RES) {
                        add_res(v0, r0); add_res(v1, r1); add_res(v2, r2); add_res(v3, r3);
                        add_res(v4, r4); add_res(v5, r5); add_res(v6, r6); add_res(v7, r7);
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (b == 0) {   // the next pass's residual rows: issued BEFORE this pass's stores (one in-order vmcnt)
                            V7_RES_ROWS(1)
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    if constexpr (b == 0) stage_pass(std::integral_constant<int, 1>{});   // (LDS ops of a wave execute in order: these follow the row reads)
                    bst7(rs_out, vo, so * (8 * b + 0), v0); bst7(rs_out, vo, so * (8 * b + 1), v1);
                    bst7(rs_out, vo, so * (8 * b + 2), v2); bst7(rs_out, vo, so * (8 * b + 3), v3);
                    bst7(rs_out, vo, so * (8 * b + 4), v4); bst7(rs_out, vo, so * (8 * b + 5), v5);
                    bst7(rs_out, vo, so * (8 * b + 6), v6); bst7(rs_out, vo, so * (8 * b + 7), v7);
                    __builtin_amdgcn_sched_barrier(0);
                };
                if constexpr (RES) { V7_RES_ROWS(0) }
                __builtin_amdgcn_sched_barrier(0);
                stage_pass(std::integral_constant<int, 0>{});
                one_pass(std::integral_constant<int, 0>{});
                one_pass(std::integral_constant<int, 1>{});
#undef V7_RES_ROWS
            };
            if (p.bias) {
                if (resid) tile_epi(std::true_type{}, std::true_type{});
                else tile_epi(std::false_type{}, std::true_type{});
            } else {
                if (resid) tile_epi(std::true_type{}, std::false_type{});
                else tile_epi(std::false_type{}, std::false_type{});
            }
        } else if constexpr (IS_SWIGLU) {
            // fc1 + SwiGLU: weight rows interleaved by 32, so the tile's 128 fused columns are [y0 | g0 | y1 | g1] = 64
            // activation columns (one 128-B line per row).  Act = silu(g) * y on the bf16-rounded y / g (what the standalone
            // op sees); the optional save (p.Out) is the GATE alone, (M, N/2), plain column order.
            // staging: activation rows in the first 4 KiB ([32 rows][128 B], chunk ^ (row & 7)), gate rows in the second;
            // lane (l31, hi): dsm + (ab ^ (c << 4)); row moves (arow = le >> 3, ach = le & 7, rows 8 k + arow): arb + 1024 k
            const uint32_t ab = my + (uint32_t)l31 * 128 + ((l31 & 7) << 4) + hi * 8;
            const int arow = le >> 3, ach = le & 7;
            const uint32_t arb = my + (uint32_t)arow * 128 + ((ach ^ arow) << 4);
            const int ncol = (n0 >> 1);
            const __amdgpu_buffer_rsrc_t rs_act = rsrc7(reinterpret_cast<bf16_t*>(p.Out2) + (size_t)m0 * p.ldo2 + ncol, rows_here * (uint32_t)p.ldo2 * 2u);
            const uint32_t va = ((uint32_t)arow * (uint32_t)p.ldo2 + ach * 8) * 2u;
            const uint32_t sa8 = (uint32_t)p.ldo2 * 16u;   // 8 rows, bytes
            auto swiglu_epi = [&](auto save_c) {
                constexpr bool SAVE = decltype(save_c)::value;
                const __amdgpu_buffer_rsrc_t rs_g = rsrc7(SAVE ? reinterpret_cast<bf16_t*>(p.Out) + (size_t)m0 * p.ldo + ncol : nullptr, SAVE ? rows_here * (uint32_t)p.ldo * 2u : 0u);
                const uint32_t vg = ((uint32_t)arow * (uint32_t)p.ldo + ach * 8) * 2u;
                const uint32_t sg8 = (uint32_t)p.ldo * 16u;
                auto stage_q = [&](auto bc, auto prc, auto qc) {
                    constexpr int b = decltype(bc)::value, pr = decltype(prc)::value, q = decltype(qc)::value;
                    uint2 py, pg;
                    py.x = pack_bf16x2(acc[4 * b + 2 * pr][4 * q], acc[4 * b + 2 * pr][4 * q + 1]);
                    py.y = pack_bf16x2(acc[4 * b + 2 * pr][4 * q + 2], acc[4 * b + 2 * pr][4 * q + 3]);
                    pg.x = pack_bf16x2(acc[4 * b + 2 * pr + 1][4 * q], acc[4 * b + 2 * pr + 1][4 * q + 1]);
                    pg.y = pack_bf16x2(acc[4 * b + 2 * pr + 1][4 * q + 2], acc[4 * b + 2 * pr + 1][4 * q + 3]);
                    const float yy[4] = {bf16lo_to_f32(py.x), bf16hi_to_f32(py.x), bf16lo_to_f32(py.y), bf16hi_to_f32(py.y)};
                    const float gg[4] = {bf16lo_to_f32(pg.x), bf16hi_to_f32(pg.x), bf16lo_to_f32(pg.y), bf16hi_to_f32(pg.y)};
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = gg[e] * yy[e] * __builtin_amdgcn_rcpf(1.f + __expf(-gg[e]));
                    uint2 pa;
                    pa.x = pack_bf16x2(o[0], o[1]);
                    pa.y = pack_bf16x2(o[2], o[3]);
                    const uint32_t ca = vxor7<((pr * 4 + q) << 4)>(ab);
                    *reinterpret_cast<uint2*>(dsm + ca) = pa;
                    if constexpr (SAVE) *reinterpret_cast<uint2*>(dsm + ca + 4096) = pg;
                };
                auto stage_pass = [&](auto bc) {
                    stage_q(bc, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}); stage_q(bc, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
                    stage_q(bc, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{}); stage_q(bc, std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{});
                    __builtin_amdgcn_sched_barrier(0);
                    stage_q(bc, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}); stage_q(bc, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
                    stage_q(bc, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{}); stage_q(bc, std::integral_constant<int, 1>{}, std::integral_constant<int, 3>{});
                    __builtin_amdgcn_sched_barrier(0);
                };
                auto one_pass = [&](auto bc) {
                    constexpr int b = decltype(bc)::value;
                    const char* sa = dsm + arb;
                    const uint4 a0 = *reinterpret_cast<const uint4*>(sa), a1 = *reinterpret_cast<const uint4*>(sa + 1024),
                                a2 = *reinterpret_cast<const uint4*>(sa + 2048), a3 = *reinterpret_cast<const uint4*>(sa + 3072);
                    uint4 g0 = {}, g1 = {}, g2 = {}, g3 = {};
                    if constexpr (SAVE) {
                        g0 = *reinterpret_cast<const uint4*>(sa + 4096); g1 = *reinterpret_cast<const uint4*>(sa + 5120);
                        g2 = *reinterpret_cast<const uint4*>(sa + 6144); g3 = *reinterpret_cast<const uint4*>(sa + 7168);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (b == 0) stage_pass(std::integral_constant<int, 1>{});
                    bst7(rs_act, va, sa8 * (4 * b + 0), a0); bst7(rs_act, va, sa8 * (4 * b + 1), a1);
                    bst7(rs_act, va, sa8 * (4 * b + 2), a2); bst7(rs_act, va, sa8 * (4 * b + 3), a3);
                    if constexpr (SAVE) {
                        bst7(rs_g, vg, sg8 * (4 * b + 0), g0); bst7(rs_g, vg, sg8 * (4 * b + 1), g1);
                        bst7(rs_g, vg, sg8 * (4 * b + 2), g2); bst7(rs_g, vg, sg8 * (4 * b + 3), g3);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                };
                stage_pass(std::integral_constant<int, 0>{});
                one_pass(std::integral_constant<int, 0>{});
                one_pass(std::integral_constant<int, 1>{});
            };
            if (p.Out) swiglu_epi(std::true_type{});
            else swiglu_epi(std::false_type{});
        } else {
            // fc2 dgrad + SwiGLU backward from the saved (act, gate) (see GEMM_EPI_SWIGLU_BWD_AG in gemm_bf16_v6.hip): the tile is
            // d(act) for 128 activation columns = 256 columns [y0|g0|y1|g1|y2|g2|y3|g3] of dYG.  A pass covers one m-block
            // (32 rows) x one PAIR of 32-column groups (ap) = 128 dYG columns (256 B per staged row):
            //   4 + 4 row loads per lane (a 128-B line of Act / of G per row, 8 rows per instruction) -> LDS block [y|g|y|g]
            //   -> every lane turns its cells into (dy, dg) in place -> 8 row reads + 8 coalesced 16-B stores.
            // The loads of pass i + 1 are issued before the stores of pass i (one in-order vmcnt).
            //   d y = silu(g) d,   d gate = d * act * (1 / g + 1 - sigmoid(g))
            const __amdgpu_buffer_rsrc_t rs_a = rsrc7(reinterpret_cast<const bf16_t*>(p.Out2) + (size_t)m0 * p.ldo2 + n0, rows_here * (uint32_t)p.ldo2 * 2u);
            const __amdgpu_buffer_rsrc_t rs_g = rsrc7(p.In3 + (size_t)m0 * p.ldo2 + n0, rows_here * (uint32_t)p.ldo2 * 2u);
            const __amdgpu_buffer_rsrc_t rs_d = rsrc7(reinterpret_cast<bf16_t*>(p.Out) + (size_t)m0 * p.ldo + 2 * n0, rows_here * (uint32_t)p.ldo * 2u);
            // source lanes: (r8, c8) = (le >> 3, le & 7): 16-B chunk c8 of the 128-B line (64 columns of pair ap) of row 8 j + r8
            const int r8 = le >> 3, c8 = le & 7;
            const uint32_t vs = ((uint32_t)r8 * (uint32_t)p.ldo2 + c8 * 8) * 2u;
            const uint32_t ss8 = (uint32_t)p.ldo2 * 16u;   // 8 rows, bytes
            // staged position of that chunk: logical chunk La = (c8 >> 2) * 8 + (c8 & 3) (act), La + 4 (gate) of row 8 j + r8:
            //   dsm + (sb ^ ((j & 1) << 7)) + 2048 j   [+ 64 XORed in for the gate]
            const uint32_t sb = my + (uint32_t)r8 * 256 + (((((c8 >> 2) << 3) | (c8 & 3)) ^ r8) << 4);
            const uint32_t vd = ((uint32_t)lrow * (uint32_t)p.ldo + lch * 8) * 2u;
            const uint32_t sd4 = (uint32_t)p.ldo * 8u;     // 4 rows of dYG, bytes
            uint4 ta0, ta1, ta2, ta3, tg0, tg1, tg2, tg3;
#define V7_LOAD(b_, ap_)                                                                                                                   \
    ta0 = bld7(rs_a, vs + (ap_) * 128, ss8 * (4 * (b_) + 0)); tg0 = bld7(rs_g, vs + (ap_) * 128, ss8 * (4 * (b_) + 0));                    \
    ta1 = bld7(rs_a, vs + (ap_) * 128, ss8 * (4 * (b_) + 1)); tg1 = bld7(rs_g, vs + (ap_) * 128, ss8 * (4 * (b_) + 1));                    \
    ta2 = bld7(rs_a, vs + (ap_) * 128, ss8 * (4 * (b_) + 2)); tg2 = bld7(rs_g, vs + (ap_) * 128, ss8 * (4 * (b_) + 2));                    \
    ta3 = bld7(rs_a, vs + (ap_) * 128, ss8 * (4 * (b_) + 3)); tg3 = bld7(rs_g, vs + (ap_) * 128, ss8 * (4 * (b_) + 3));
            auto cell_q = [&](auto blkc, auto aac, auto qc) {
                constexpr int blk = decltype(blkc)::value, aa = decltype(aac)::value, q = decltype(qc)::value;
                const uint32_t oy = vxor7<((aa * 8 + q) << 4)>(cb);   // act (-> d y) cell of this lane
                const uint32_t og = vxor7<64>(oy);                     // gate (-> d gate) cell: chunk + 4 <=> offset ^ 64
                char* py = dsm + oy;
                char* pg = dsm + og;
                const uint2 yy = *reinterpret_cast<const uint2*>(py);
                const uint2 gg = *reinterpret_cast<const uint2*>(pg);
                const float y[4] = {bf16lo_to_f32(yy.x), bf16hi_to_f32(yy.x), bf16lo_to_f32(yy.y), bf16hi_to_f32(yy.y)};
                const float g[4] = {bf16lo_to_f32(gg.x), bf16hi_to_f32(gg.x), bf16lo_to_f32(gg.y), bf16hi_to_f32(gg.y)};
                // (d(act) stays in fp32: as in gemm_bf16_v6.hip, round 4)
                const float d[4] = {acc[blk + aa][4 * q], acc[blk + aa][4 * q + 1], acc[blk + aa][4 * q + 2], acc[blk + aa][4 * q + 3]};
                float dy[4], dg[4];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    cx_f2 y2, g2;
                    swiglu_bwd_from_act2(cx_f2{d[2 * h], d[2 * h + 1]}, cx_f2{y[2 * h], y[2 * h + 1]}, cx_f2{g[2 * h], g[2 * h + 1]}, y2, g2);
                    dy[2 * h] = y2.x; dy[2 * h + 1] = y2.y;
                    dg[2 * h] = g2.x; dg[2 * h + 1] = g2.y;
                }
                uint2 o;
                o.x = pack_bf16x2(dy[0], dy[1]); o.y = pack_bf16x2(dy[2], dy[3]);
                *reinterpret_cast<uint2*>(py) = o;
                o.x = pack_bf16x2(dg[0], dg[1]); o.y = pack_bf16x2(dg[2], dg[3]);
                *reinterpret_cast<uint2*>(pg) = o;
            };
            auto one_pass = [&](auto pc) {
                constexpr int ps = decltype(pc)::value;   // 0..3: (b, ap) = (ps >> 1, ps & 1)
                constexpr int b = ps >> 1, ap = ps & 1;
                {
                    const uint32_t s0 = sb, s1 = vxor7<128>(sb);
                    *reinterpret_cast<uint4*>(dsm + s0) = ta0; *reinterpret_cast<uint4*>(dsm + (s0 ^ 64)) = tg0;
                    *reinterpret_cast<uint4*>(dsm + s1 + 2048) = ta1; *reinterpret_cast<uint4*>(dsm + (s1 ^ 64) + 2048) = tg1;
                    *reinterpret_cast<uint4*>(dsm + s0 + 4096) = ta2; *reinterpret_cast<uint4*>(dsm + (s0 ^ 64) + 4096) = tg2;
                    *reinterpret_cast<uint4*>(dsm + s1 + 6144) = ta3; *reinterpret_cast<uint4*>(dsm + (s1 ^ 64) + 6144) = tg3;
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (ps < 3) { V7_LOAD((ps + 1) >> 1, (ps + 1) & 1) }   // the next pass's rows fly under the arithmetic
                __builtin_amdgcn_sched_barrier(0);
                constexpr int blk = 4 * b + 2 * ap;
                cell_q(std::integral_constant<int, blk>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
                cell_q(std::integral_constant<int, blk>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
                __builtin_amdgcn_sched_barrier(0);
                cell_q(std::integral_constant<int, blk>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
                cell_q(std::integral_constant<int, blk>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{});
                __builtin_amdgcn_sched_barrier(0);
                cell_q(std::integral_constant<int, blk>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
                cell_q(std::integral_constant<int, blk>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
                __builtin_amdgcn_sched_barrier(0);
                cell_q(std::integral_constant<int, blk>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
                cell_q(std::integral_constant<int, blk>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 3>{});
                __builtin_amdgcn_sched_barrier(0);
                auto store_half = [&](auto hc) {   // two halves of 4 row groups: 16 registers of row data at a time
                    constexpr int h = decltype(hc)::value;
                    const uint4 v0 = *reinterpret_cast<const uint4*>(V7_ROW(4 * h + 0)), v1 = *reinterpret_cast<const uint4*>(V7_ROW(4 * h + 1)),
                                v2 = *reinterpret_cast<const uint4*>(V7_ROW(4 * h + 2)), v3 = *reinterpret_cast<const uint4*>(V7_ROW(4 * h + 3));
                    bst7(rs_d, vd + ap * 256, sd4 * (8 * b + 4 * h + 0), v0); bst7(rs_d, vd + ap * 256, sd4 * (8 * b + 4 * h + 1), v1);
                    bst7(rs_d, vd + ap * 256, sd4 * (8 * b + 4 * h + 2), v2); bst7(rs_d, vd + ap * 256, sd4 * (8 * b + 4 * h + 3), v3);
                    __builtin_amdgcn_sched_barrier(0);
                };
                store_half(std::integral_constant<int, 0>{});
                store_half(std::integral_constant<int, 1>{});
            };
            V7_LOAD(0, 0)
            __builtin_amdgcn_sched_barrier(0);
            one_pass(std::integral_constant<int, 0>{});
            one_pass(std::integral_constant<int, 1>{});
            one_pass(std::integral_constant<int, 2>{});
            one_pass(std::integral_constant<int, 3>{});
#undef V7_LOAD
        }
#undef V7_CELL
#undef V7_ROW
        cp_tile = tile_of(++cp_round);
        // Register relief: the first W fragments of the next tile (read into W0 during the last k-step) are fetched again
        // here -- their slot is not part of the staging space -- so W0 is dead across the epilogue.
        asm volatile("" : "=v"(W0.f[0]), "=v"(W0.f[1]), "=v"(W0.f[2]), "=v"(W0.f[3]));
        // ... and so are the 12 DMA cursor offsets: rebuilt from (round, K-tile) of the two cursors
        asm volatile("" : "=v"(xoff[0]), "=v"(xoff[1]), "=v"(xoff[2]), "=v"(xoff[3]), "=v"(xoff[4]), "=v"(xoff[5]), "=v"(xoff[6]), "=v"(xoff[7]));
        asm volatile("" : "=v"(woff[0]), "=v"(woff[1]), "=v"(woff[2]), "=v"(woff[3]));
        {
            x_clamped = w_clamped = true;   // (forces the re-computation)
            const int tx = tile_of(lx_round), tw = tile_of(lw_round);
            x_setup(tx >= 0 ? tx : first_tile);
            xbase += lx_kt * BK7;
            w_setup(tw >= 0 ? tw : first_tile);
            wbase += lw_kt * BK7;
        }
        if (cp_tile >= 0) {
            const char* nws = dsm + (ws_slot ? W1_OFF7 : W0_OFF7);
#pragma unroll
            for (int a = 0; a < 4; ++a) read_w(W0, nws, 0, a);
        }
        // the staging area is the DMA target of the next K-tile's W: nobody may still be reading it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
#undef V7_SLOT
#undef V7_M0
#undef V7_LD
#undef V7_NORD
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the cursors' dummy DMAs must land before the LDS is handed on
#ifndef CX_PRODUCT
    if (p.trace && tid == 0) {   // residency census (scripts/gemm_v7_census.py): {start, end, HW_ID, XCC_ID} per workgroup
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        p.trace[4 * blockIdx.x] = t_begin;
        p.trace[4 * blockIdx.x + 1] = (long long)__builtin_amdgcn_s_memtime();
        p.trace[4 * blockIdx.x + 2] = hw;
        p.trace[4 * blockIdx.x + 3] = xcc;
    }
#endif
}

template <int EPI, int DBG = 0>
hipError_t launch7(const GemmParams& p, hipStream_t stream) {
    static CxLdsOptIn lds;
    if (!lds.ensure(reinterpret_cast<const void*>(&gemm_bf16_v7_kernel<EPI, DBG>), LDS7)) return hipErrorInvalidValue;
    const int ntiles = p.tiles_m * p.tiles_n;
    const int grid = ntiles < 512 ? (ntiles + 7) / 8 * 8 : 512;
    hipLaunchKernelGGL((gemm_bf16_v7_kernel<EPI, DBG>), dim3(grid), dim3(256), LDS7, stream, p);
    return hipGetLastError();
}

// N-groups of the XCD grid (see cx_gemm_v6_groups): gn * |X| + (8 / gn) * |W| when an XCD's W slice stays L2-resident.
int v7_groups(int tiles_m, int tiles_n, int K) {
    const double xb = (double)tiles_m * BM7 * K * 2, wb = (double)tiles_n * BN7 * K * 2;
    int best = 1;
    double best_cost = 0;
    for (int gn = 1; gn <= 8; gn *= 2) {
        const int gm = 8 / gn;
        if ((tiles_n % gn) != 0 || (tiles_m % gm) != 0) continue;
        const double slice = wb / gn;
        const double rounds = (double)tiles_m / gm * (tiles_n / gn) / 64.0;
        const double wcost = slice <= 2.5 * 1048576 ? gm * wb : (rounds < 1 ? 1 : rounds) * slice * 8;
        const double cost = gn * xb + wcost;
        if (gn == 1 || cost < best_cost) { best = gn; best_cost = cost; }
    }
    return best;
}

}  // namespace

// Shapes the two-workgroups-per-CU kernel covers: N a multiple of 128, K of 64, 16-B aligned leading dimensions (checked by
// the C entry points), alpha == 1 for the plain / bias / residual form.
bool cx_gemm_v7_covers(const GemmParams& p, int epi) {
    if (epi != GEMM_EPI_NONE && epi != GEMM_EPI_SWIGLU_G && epi != GEMM_EPI_SWIGLU_BWD_AG) return false;
    if ((p.N % BN7) != 0 || (p.K % BK7) != 0 || p.K < 2 * BK7) return false;
    if (epi == GEMM_EPI_NONE && p.alpha != 1.f) return false;
    if ((p.N / BN7) > 256) return false;
    return true;
}

#ifndef CX_PRODUCT
static int g_v7_period = -1;   // cx_gemm_v7_set_period: tile period in cycles for the start stagger; -1 = estimate, 0 = no stagger
void cx_gemm_v7_set_period(int cycles) { g_v7_period = cycles; }
#else
constexpr int g_v7_period = -1;
#endif
// Start stagger unit (tile period / 64, shader cycles; 0 = none).  The period is an estimate -- K-tiles at the shared-pipe
// rate plus the epilogue of the form -- and needs no precision: what matters is that co-located workgroups do NOT start
// together.  Launches with fewer than two tiles per workgroup are not staggered (nothing to overlap, the delay would be pure tail).
static int v7_stagger_unit(const GemmParams& p, int epi) {
    if (g_v7_period == 0) return 0;
    if ((long)p.tiles_m * p.tiles_n < 1024) return 0;
    const int epi_cycles = epi == GEMM_EPI_SWIGLU_BWD_AG ? 24000 : epi == GEMM_EPI_SWIGLU_G ? 12000 : 9000;
    const int period = g_v7_period > 0 ? g_v7_period : (p.K / BK7) * 2400 + epi_cycles;
    return period / 64;
}

#ifndef CX_PRODUCT
static long long* g_v7_trace = nullptr;
static int g_v7_dbg = 0;
static int g_v7_flags = 1;   // bit 0: K loop at s_setprio 1 (measured +1 .. 2.5 %, profiles/r4_gemm_v7_ab.txt)
void cx_gemm_v7_set_flags(int f) { g_v7_flags = f; }
void cx_gemm_v7_set_trace(long long* buf) { g_v7_trace = buf; }
void cx_gemm_v7_set_ablate(int mask) { g_v7_dbg = mask; }
int cx_gemm_v7_occupancy_query(void) {
    int n = -1;
    static CxLdsOptIn lds;
    if (!lds.ensure(reinterpret_cast<const void*>(&gemm_bf16_v7_kernel<GEMM_EPI_SWIGLU_BWD_AG>), LDS7)) return -2;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_bf16_v7_kernel<GEMM_EPI_SWIGLU_BWD_AG>, 256, LDS7) != hipSuccess) return -3;
    return n;
}
#endif

hipError_t cx_launch_gemm_v7(GemmParams p, int epi, int force_gn, hipStream_t stream) {
#ifndef CX_PRODUCT
    p.trace = g_v7_trace;
    p.dbg = g_v7_flags;
#else
    p.trace = nullptr;
    p.dbg = 1;   // the K loop runs at s_setprio 1 (see the kernel)
#endif
#ifndef CX_PRODUCT
    if (g_v7_dbg) {   // ablation builds (scripts/gemm_v7_ablate.py): timing only
        p.tiles_m = (p.M + BM7 - 1) / BM7;
        p.tiles_n = p.N / BN7;
        p.sup_n = force_gn > 0 ? force_gn : v7_groups(p.tiles_m, p.tiles_n, p.K);
        p.sup_m = v7_stagger_unit(p, epi);
        const bool bwd = epi == GEMM_EPI_SWIGLU_BWD_AG;
        switch (g_v7_dbg) {
#define V7_ABL(m) case m: return bwd ? launch7<GEMM_EPI_SWIGLU_BWD_AG, m>(p, stream) : launch7<GEMM_EPI_NONE, m>(p, stream);
            V7_ABL(1) V7_ABL(2) V7_ABL(16) V7_ABL(17) V7_ABL(19) V7_ABL(64) V7_ABL(65) V7_ABL(81) V7_ABL(83) V7_ABL(87) V7_ABL(119) V7_ABL(127) V7_ABL(111)
#undef V7_ABL
            default: break;
        }
    }
#endif
    p.tiles_m = (p.M + BM7 - 1) / BM7;
    p.tiles_n = p.N / BN7;
    p.sup_n = force_gn > 0 ? force_gn : v7_groups(p.tiles_m, p.tiles_n, p.K);
    p.sup_m = v7_stagger_unit(p, epi);
    return epi == GEMM_EPI_SWIGLU_G        ? launch7<GEMM_EPI_SWIGLU_G>(p, stream)
           : epi == GEMM_EPI_SWIGLU_BWD_AG ? launch7<GEMM_EPI_SWIGLU_BWD_AG>(p, stream)
                                           : launch7<GEMM_EPI_NONE>(p, stream);
}
