// gemm_bf16_v6.hip -- persistent 256x256x64 bf16 MFMA GEMM with ONE wave per SIMD (K10).
//
// What the phase trace of the 8-wave kernel (gemm_bf16_v5.hip) showed: with two waves per SIMD sharing one MFMA pipe
// and one barrier per K-tile, the pipe is busy ~64% of the steady-state loop even with the DMA removed -- the two waves
// fall out of phase (one waits ~600 cycles at every barrier) and the 64x128 wave tile needs 192 ds_read_b128 per
// K-tile, which together with the 64 KiB of LDS-DMA writes keeps the LDS array busy for most of the iteration.
//
// Here a workgroup is 4 waves as 2(M) x 2(N), each owning a 128x128 sub-tile = 256 fp32 accumulator registers, which live
// in the AGPR half of the unified 512-entry register file (one wave per SIMD).  Per K-tile a wave reads 32 ds_read_b128
// (every fragment row once) and there is no second wave to fall out of phase with.  Everything is software-pipelined in
// the instruction stream of that one wave: fragments are read one k-step ahead, the 16 DMA instructions of the next
// K-tiles are spread between MFMAs, and the only synchronisation per K-tile is one raw s_barrier across the 4 waves.
//
// Round 6: the matrix instruction is v_mfma_f32_16x16x32_bf16 (8 x 8 blocks of 16 x 16 per wave, 128 per K-tile), not
// 32x32x16 (4 x 4 blocks, 64 per K-tile).  Same FLOPs, same LDS reads, one more issue cycle per 16 K FLOP -- and ~5 %
// more work per joule: every launch of this kernel runs at the 1400 W cap, where throughput is set by energy per FLOP,
// and the 16 x 16 form moves half the accumulator bytes per FLOP through the register file (4 registers read + written
// per 16 K FLOP against 16 per 32 K).  Measured in situ before the rewrite (results discarded, 1-s windows at the cap,
// profiles/r6_gemm_mfma16_in_situ.txt): 1546 -> 1768 MHz and +4.5 ... +6.4 % on the seven shapes of a block.  The
// epilogues keep thinking in 32 x 32 regions of four blocks (gemm_v6_acc.inc); only the lane -> element map changed.
//
// LDS map (160 KiB): three 32 KiB X slots (X runs two K-tiles ahead), two 32 KiB W slots (one ahead), issue order per
// K-tile [W of t+1 in k-steps 0,1][X of t+2 in k-steps 2,3]; "operands of t+1 landed" == s_waitcnt vmcnt(4) in front of the
// one barrier of the K-tile (everything but the 4 X instructions just issued; the epilogue's stores are older and retire in
// order with them on gfx9's single vmcnt).  Epilogue staged through the two slots consumed last, 32 output rows per wave
// per pass.  Round 2: an LDS-DMA is two instructions between two MFMAs (see the cursors), tiles are named (tm << 8) | tn.
#include "cx_common.h"
#include "../../include/contrastors_hip.h"
#include "gemm_params.h"
#include <type_traits>

namespace {

// Epilogue traffic is touched once: outputs are not re-read by this launch and the streamed epilogue inputs (residual,
// saved (y, gate)) are read once.  CX_V6_NT (bit 0 stores, bit 1 loads) marks them non-temporal so that they do not evict
// the operand panels the XCD's other workgroups are about to re-use from L2.
// SwiGLU-backward epilogue (round 5).  HI_EARLY: all 16 (act, gate) row loads of pass b + 1 are issued right after pass b's rows
// have been staged (1), instead of rows 16..31 only after the pass's arithmetic (0: they then have ~700 cycles -- the row reads and
// store issue of pass b -- to land before pass b + 1 stages them).  That needs 32 more registers through the arithmetic, which
// the kernel (256 VGPRs) did not have: OPAQUE passes the lane index the staging addresses derive from through an empty asm per
// tile (1) / per pass (2), so that the compiler re-derives the ~50 LDS addresses (a few VALU each) instead of keeping them live.
// Same-box A/B at the metric's fc2-dgrad launch (T = 262144; profiles/r5_gemm_swiglu_bwd_hi_early_ab.txt, interleaved rounds, outputs
// bit-identical): HI_EARLY 0 / OPAQUE 0 1840 us | 1 / 1 1802 us (-2.1 %) | 1 / 2 1852 us | 0 / 2 1884 us (what re-deriving the
// addresses per pass costs: +2.4 %; issuing the loads early buys 3 % of it back).  Shipped: 1 / 1.
#ifndef CX_V6_HI_EARLY
#define CX_V6_HI_EARLY 1
#endif
#ifndef CX_V6_OPAQUE
#define CX_V6_OPAQUE 1
#endif
#ifndef CX_V6_NT
#define CX_V6_NT 3   // measured on the whole step (scripts/gpu_variant_bench.sh): 0 -> 3861..3873, 1 -> 3895, 3 -> 3907 pairs/s
#endif
// LDS-DMA issue (see the cursor comments in the kernels): M0 = slot base + J * 4 KiB, then the load; split in two so that
// an MFMA can sit between the M0 write and its use (the wait state the hardware asks for), or fused with an s_nop.
template <int J>
__device__ __forceinline__ void v6_dma_m0(uint32_t slot_base) {
    asm volatile("s_add_u32 m0, %0, %1" ::"s"(slot_base), "n"(J * 4096) : "memory", "scc");
}
__device__ __forceinline__ void v6_dma_ld(uint32_t off, const bf16_t* base) {
    asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base) : "memory");
}
template <int J>
__device__ __forceinline__ void v6_dma_full(uint32_t slot_base, uint32_t off, const bf16_t* base) {
    asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3" ::"s"(slot_base), "n"(J * 4096), "v"(off),
                 "s"(base) : "memory", "scc");
}
typedef unsigned int cx_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void gst_nt(void* ptr, uint4 v) {
    cx_u32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<cx_u32x4*>(ptr));
}
__device__ __forceinline__ uint4 gld_nt(const void* ptr) {
    const cx_u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const cx_u32x4*>(ptr));
    return make_uint4(t.x, t.y, t.z, t.w);
}
#if CX_V6_NT & 1
#define gst(ptr, v) gst_nt((ptr), (v))
#else
#define gst(ptr, v) (*reinterpret_cast<uint4*>(ptr) = (v))
#endif
#if CX_V6_NT & 2
#define gld(ptr) gld_nt(ptr)
#else
#define gld(ptr) (*reinterpret_cast<const uint4*>(ptr))
#endif

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* glb_void_ptr;

constexpr int BM6 = 256, BN6 = 256, BK6 = 64;
constexpr int XS6 = 32768;          // one operand K-tile: 256 rows x 128 B
constexpr int LDS6 = 5 * XS6;       // 3 X slots + 2 W slots

struct Frags6 {   // one k-step (32 of K): 8 W fragments (16 output columns each) and 8 X fragments (16 output rows each)
    bf16x8_t w[8], x[8];
};

#include "gemm_v6_acc.inc"

// f(integral_constant<I0>) ... f(integral_constant<I1 - 1>): the accumulator accessors are switches over a literal block index -- a
// `#pragma unroll` loop the compiler decides not to unroll (39 MFMAs in a row) turns them into jump tables over fragments in scratch
template <int I0, int I1, class F>
CX_DEVICE void v6_static_for(F&& f) {
    if constexpr (I0 < I1) {
        f(std::integral_constant<int, I0>{});
        v6_static_for<I0 + 1, I1>(f);
    }
}

// DBG (ablation builds only, scripts/gemm_ablate.py; results are garbage, timing is the point): bit0 no DMA in the main
// loop, bit1 no barrier, bit2 no fragment reads, bit3 no MFMA, bit4 no epilogue, bit5 no vmcnt wait.  With p.trace set,
// wave 0 of every workgroup stores its s_memtime span and K-tile count.
template <int EPI, int DBG = 0>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_bf16_v6_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    constexpr bool IS_SWIGLU_BWD = EPI == GEMM_EPI_SWIGLU_BWD || EPI == GEMM_EPI_SWIGLU_BWD_AG;
    constexpr bool AG = EPI == GEMM_EPI_SWIGLU_BWD_AG;      // the saved pair is (act, gate) instead of (y, gate)
    constexpr bool SAVEG = EPI == GEMM_EPI_SWIGLU_G;        // the forward's optional save is the gate alone
    long long t_begin = 0, n_ktiles = 0;
    if constexpr (DBG != 0) t_begin = (long long)__builtin_amdgcn_s_memtime();
    // DBG & 128 (trace builds): s_memtime phase sums of wave 0 over all of the workgroup's tiles -> p.trace[16 b + 2 ..]: 2 K loop, 3 epilogue
    // prologue (first pass's arithmetic + staging), 4 row-read issue, 5 next pass's arithmetic, 6 its staging writes, 7 stores, 8 tile tail
    // (fragment re-read, barrier).  An in-order wave's stamp between two groups is the issue time of the first, stalls included.
    long long ph[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, t_last = t_begin, kt0_cyc = 0;
#define CX_PH(i_)                                                              \
    do {                                                                       \
        if constexpr ((DBG & 128) != 0) {                                      \
            const long long now_ = (long long)__builtin_amdgcn_s_memtime();    \
            ph[(i_)] += now_ - t_last;                                         \
            t_last = now_;                                                     \
        }                                                                      \
    } while (0)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g4 = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int nk = p.K / BK6;
    // Tile order.  Workgroup b runs on XCD b%8 (own 4 MiB L2).  The 8 XCDs form a gm x gn grid over the tile matrix
    // (gn = p.sup_n N-groups): XCD (xi, xj) owns M-panels [m_lo, m_hi) x N-tiles [n_lo, n_hi) and its workgroups walk
    // that block n-fastest, 32 (= workgroups per XCD) consecutive tiles per round.  With gn > 1 an XCD touches only its
    // slice of W, which then stays L2-resident across rounds instead of being re-streamed for every M-panel.
    const int per_xcd = gridDim.x >> 3;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int gn = p.sup_n > 0 ? p.sup_n : 1, gm = 8 / gn;
    const int xi = xcd / gn, xj = xcd - xi * gn;
    const int m_lo = p.tiles_m * xi / gm, m_hi = p.tiles_m * (xi + 1) / gm;
    const int n_lo = p.tiles_n * xj / gn, n_wd = p.tiles_n * (xj + 1) / gn - n_lo;
    const int nloc = (m_hi - m_lo) * n_wd;
    // a tile is named (tm << 8) | tn (tiles_n <= 256, checked by the launcher), -1 = none: the cursors and the epilogue take
    // tm / tn apart with a shift and a mask (integer division is ~20 SALU instructions, and it sits between two MFMAs)
    auto tile_of = [&](int round) {
        const int l = round * per_xcd + idx;
        if (l >= nloc) return -1;
        const int q = l / n_wd;
        return ((m_lo + q) << 8) | (n_lo + (l - q * n_wd));
    };

    // ---- DMA cursors.  A K-tile of one operand = 32 instructions of 1 KiB (8 rows x 128 B), 8 per wave.  A cursor is a
    // wave-uniform 64-bit base (SGPR pair: the panel of the output tile, advanced by 128 B per K-tile with two SALU
    // instructions) + one CONSTANT 32-bit byte offset per instruction (VGPR, set when the cursor enters an output tile).
    // With one wave per SIMD the main loop is issue-bound: an LDS-DMA used to cost nine instructions between two MFMAs
    // (a branch on "is there another tile", an SGPR reload through v_readlane for the LDS address, three SALU, s_nop,
    // the load, a VALU offset add); it is two now -- `s_add_u32 m0, <slot base>, <imm>` in front of an MFMA (which is the
    // wait state the M0 write needs) and the load behind it.  There is no "live" branch: past its last tile a cursor
    // re-walks the workgroup's first tile into ring slots nobody reads (a few KiB of dummy traffic per workgroup).
    uint32_t xoff[8], woff[8];
    const bf16_t* xbase = p.X;
    const bf16_t* wbase = p.W;
    int lx_round = 0, lx_kt = 0, lx_slot = 0;
    int lw_round = 0, lw_kt = 0, lw_slot = 0;
    bool lx_live, lw_live;
    const int first_tile = tile_of(0);
    // per-lane offsets of a FULL 256-row panel do not depend on the tile: they are computed once, and re-computed (with the
    // row clamp) only when a cursor enters or leaves a partial last panel
    bool x_clamped = true, w_clamped = true;  // "offsets are not the generic ones": forces the first computation
    auto x_setup = [&](int tile) {
        const int tm = tile >> 8;
        xbase = p.X + (size_t)tm * BM6 * p.ldx;
        const int rows = p.M - tm * BM6;  // >= 1: rows of this panel that exist (others re-read the last one)
        if (rows < BM6 || x_clamped) {
            x_clamped = rows < BM6;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = (j * 4 + wave) * 8 + (lane >> 3);
                const int c = (lane & 7) ^ ((r >> 1) & 7);
                const int rr = r < rows ? r : rows - 1;
                xoff[j] = (uint32_t)rr * (uint32_t)p.ldx * 2u + c * 16;
            }
        }
    };
    auto w_setup = [&](int tile) {
        const int tn = tile & 255;
        wbase = p.W + (size_t)tn * BN6 * p.ldw;
        const int rows = p.N - tn * BN6;
        if (rows < BN6 || w_clamped) {
            w_clamped = rows < BN6;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = (j * 4 + wave) * 8 + (lane >> 3);
                const int c = (lane & 7) ^ ((r >> 1) & 7);
                const int rr = r < rows ? r : rows - 1;
                woff[j] = (uint32_t)rr * (uint32_t)p.ldw * 2u + c * 16;
            }
        }
    };
    // LDS-DMA through inline asm, not the builtin: the compiler, knowing that VMEM writes LDS, guards LDS accesses it
    // cannot disambiguate with s_waitcnt vmcnt(0) -- in front of the first staging write of every epilogue here (a full
    // DMA round trip per tile), in front of every transposing read in the TN kernel below.  All DMA waits in this file
    // are explicit counted s_waitcnt + barrier.  (M0 = the wave's LDS destination; nothing else here uses M0.)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_ptr)dsm;
    const uint32_t lds_wave = lds0 + wave * 1024;      // instruction j of a K-tile lands at slot + (j * 4 + wave) * 1024
    uint32_t x_m0 = lds_wave, w_m0 = lds_wave + 3 * XS6;
    auto issue_x_tile = [&]() {  // a whole K-tile at once (prologue: no MFMA to put between the M0 write and the load)
        if constexpr ((DBG & 1) != 0) return;
        v6_dma_full<0>(x_m0, xoff[0], xbase); v6_dma_full<1>(x_m0, xoff[1], xbase); v6_dma_full<2>(x_m0, xoff[2], xbase);
        v6_dma_full<3>(x_m0, xoff[3], xbase); v6_dma_full<4>(x_m0, xoff[4], xbase); v6_dma_full<5>(x_m0, xoff[5], xbase);
        v6_dma_full<6>(x_m0, xoff[6], xbase); v6_dma_full<7>(x_m0, xoff[7], xbase);
    };
    auto issue_w_tile = [&]() {
        if constexpr ((DBG & 1) != 0) return;
        v6_dma_full<0>(w_m0, woff[0], wbase); v6_dma_full<1>(w_m0, woff[1], wbase); v6_dma_full<2>(w_m0, woff[2], wbase);
        v6_dma_full<3>(w_m0, woff[3], wbase); v6_dma_full<4>(w_m0, woff[4], wbase); v6_dma_full<5>(w_m0, woff[5], wbase);
        v6_dma_full<6>(w_m0, woff[6], wbase); v6_dma_full<7>(w_m0, woff[7], wbase);
    };
    auto x_advance = [&]() {
        xbase += BK6;
        lx_slot = lx_slot == 2 ? 0 : lx_slot + 1;
        x_m0 = lds_wave + lx_slot * XS6;
        if (++lx_kt == nk) {
            lx_kt = 0;
            const int t = tile_of(++lx_round);
            lx_live = t >= 0;
            x_setup(lx_live ? t : first_tile);
        }
    };
    auto w_advance = [&]() {
        wbase += BK6;
        lw_slot ^= 1;
        w_m0 = lds_wave + (3 + lw_slot) * XS6;
        if (++lw_kt == nk) {
            lw_kt = 0;
            const int t = tile_of(++lw_round);
            lw_live = t >= 0;
            w_setup(lw_live ? t : first_tile);
        }
    };

    // Accumulators: block (a = n-block, b = m-block) is index i = 4b + a of gemm_v6_acc.inc, pinned to physical AGPRs
    // a[16i : 16i+15] and invisible to the compiler: written only by the MFMA asm (the first k-step of a tile uses the
    // C = 0 form, so nothing is ever zeroed), read only by v6_read_block in the epilogue.

    int cp_round = 0;
    int cp_tile = first_tile;
    lx_live = lw_live = cp_tile >= 0;
    if (cp_tile < 0) return;  // (only when the grid is larger than the tile count: never with launch6's grid)
    x_setup(cp_tile);
    w_setup(cp_tile);
    issue_x_tile();  // X of iteration 0
    x_advance();
    issue_w_tile();  // W of iteration 0
    w_advance();
    issue_x_tile();  // X of iteration 1
    x_advance();
    int xs_slot = 0, ws_slot = 0;

    Frags6 F0, F1;
    // fragment i of k-step kk (32 of the K-tile's 64): i = 0..7 -> W blocks (16 output columns), 8..15 -> X blocks (16 output rows).
    // Lane (l15, g4) reads row l15 of the block, 16-byte chunk 4 kk + g4 of the 128-byte K-tile row: 16 lanes x 16 rows cover every
    // (row parity, swizzled chunk) pair once = all 64 banks (tile64_off XORs the chunk with (row >> 1) & 7).
    auto read_one = [&](Frags6& f, const char* xs, const char* ws, int kk, int i) {
        if constexpr ((DBG & 4) != 0) return;
        if (i < 8)
            f.w[i] = lds_read_frag(ws, tile64_off(wn * 128 + i * 16 + l15, kk * 4 + g4));
        else
            f.x[i - 8] = lds_read_frag(xs, tile64_off(wm * 128 + (i - 8) * 16 + l15, kk * 4 + g4));
    };
    // i-th MFMA of a k-step: nb = i & 7, mb = i >> 3 (gemm_v6_acc.inc)
    auto mma1 = [&](const Frags6& f, int i) {
        if constexpr ((DBG & 8) != 0) return;
        v6_mfma(i, f.w[i & 7], f.x[i >> 3]);
    };
    auto mma1z = [&](const Frags6& f, int i) {
        if constexpr ((DBG & 8) != 0) return;
        v6_mfma_z(i, f.w[i & 7], f.x[i >> 3]);
    };

    // A K-tile's instruction stream: 64 MFMAs of 16 cycles per k-step, 128 per K-tile; a fragment read or an LDS-DMA (M0 write | MFMA |
    // load, see the cursors) rides with an MFMA, nothing else sits between two MFMAs.
#ifndef CX_V6_DMA_SPREAD
#define CX_V6_DMA_SPREAD 4   // MFMAs per LDS-DMA (1 = back to back)
#endif
#ifndef CX_V6_RD_SPREAD
#define CX_V6_RD_SPREAD 2    // MFMAs per fragment read (1: the 16 reads of a k-step behind 16 consecutive MFMAs)
#endif
// CX_SEG: MFMAs i0 .. i1 - 1 of a k-step on fragments f.  Fragment read r (0 .. 15) of k-step `rkk` into `nxt` rides in front of MFMA
// R0 + RSP * r (RSP = 0: no reads); DMA instruction J0 + d (d = 0 .. ND - 1) of operand `kind` brackets MFMA D0 + DSP * d (M0 | MFMA |
// load).  The four waves leave every barrier in step and share one LDS and one vector-memory front end: requests of the same kind
// issued behind CONSECUTIVE MFMAs by all four queue up (a 1-KiB LDS-DMA keeps the front end busy for ~16 cycles = one MFMA, so four
// of them fill four MFMA slots), and an in-order wave stalls behind its own queued request -- 300 cycles of a 2640-cycle K-tile for
// the 16 DMAs back to back, 62 one per four MFMAs (scripts/gemm_ablate.py, profiles/r6_gemm_v6_m16_ablation.txt).
#define CX_SEG(MMA, f, i0, i1, nxt, rxs, rws, rkk, R0, RSP, kind, D0, DSP, J0, ND)                     \
    do {                                                                                              \
        v6_static_for<(i0), (i1)>([&](auto ic_) __attribute__((always_inline)) {                      \
            constexpr int I_ = decltype(ic_)::value;                                                  \
            constexpr bool RD_ = (RSP) > 0 && I_ >= (R0) && (I_ - (R0)) % ((RSP) > 0 ? (RSP) : 1) == 0 && (I_ - (R0)) / ((RSP) > 0 ? (RSP) : 1) < 16; \
            constexpr bool DM_ = (ND) > 0 && I_ >= (D0) && (I_ - (D0)) % (DSP) == 0 && (I_ - (D0)) / (DSP) < (ND); \
            constexpr int J_ = DM_ ? (J0) + (I_ - (D0)) / (DSP) : 0;                                  \
            if constexpr (RD_) read_one(nxt, rxs, rws, rkk, (I_ - (R0)) / ((RSP) > 0 ? (RSP) : 1));   \
            if constexpr (DM_) CX_DMA_M0(kind, J_);                                                   \
            MMA(f, I_);                                                                               \
            if constexpr (DM_) CX_DMA_LD(kind, J_);                                                   \
            __builtin_amdgcn_sched_barrier(0);                                                        \
        });                                                                                           \
    } while (0)
#define CX_DMA_M0(kind, J)                                                       \
    do {                                                                         \
        if constexpr ((kind) == 1 && (DBG & 1) == 0) v6_dma_m0<(J)>(w_m0);        \
        if constexpr ((kind) == 2 && (DBG & 1) == 0) v6_dma_m0<(J)>(x_m0);        \
    } while (0)
#define CX_DMA_LD(kind, J)                                                       \
    do {                                                                         \
        if constexpr ((kind) == 1 && (DBG & 1) == 0) v6_dma_ld(woff[(J)], wbase); \
        if constexpr ((kind) == 2 && (DBG & 1) == 0) v6_dma_ld(xoff[(J)], xbase); \
    } while (0)

    {
        // operands of iteration 0 (X_0, W_0); the only younger group is X_1 (8 ops)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < 16; ++i) read_one(F0, dsm, dsm + 3 * XS6, 0, i);
    }

    // One K-tile of the current output tile.  `first` (compile time) selects the C = 0 MFMA form for its first k-step:
    // the first K-tile of every output tile is a peeled copy of this body, so the accumulators are DEFINED there and
    // only ever updated in place afterwards (no zeroing, no conditional definitions for the register allocator).
    int cp_kt = 0;
    int pxs_slot = 0, pws_slot = 0;  // slots consumed by the K-tile just finished (the epilogue's staging space)
    auto kt_body = [&](auto first) {
        // DMA of this iteration: W of iteration +1 (k-step 0), X of iteration +2 (k-step 1, half before and half after the barrier);
        // both target slots consumed in iteration -1.
        if constexpr (DBG != 0) ++n_ktiles;
        constexpr int RS_ = CX_V6_RD_SPREAD, DS_ = CX_V6_DMA_SPREAD;
        const char* xs = dsm + xs_slot * XS6;
        const char* ws = dsm + (3 + ws_slot) * XS6;
        const int nxs_slot = xs_slot == 2 ? 0 : xs_slot + 1, nws_slot = ws_slot ^ 1;
        // k-step 0 on F0: the 16 fragments of k-step 1 -> F1, then the 8 W instructions
        if constexpr (decltype(first)::value) {
            CX_SEG(mma1z, F0, 0, 64, F1, xs, ws, 1, RS_ == 2 ? 0 : 1, RS_, 1, RS_ == 2 ? 1 : 17, DS_, 0, 8);
        } else {
            CX_SEG(mma1, F0, 0, 64, F1, xs, ws, 1, RS_ == 2 ? 0 : 1, RS_, 1, RS_ == 2 ? 1 : 17, DS_, 0, 8);
        }
        w_advance();
        // k-step 1 on F1, first half: 4 X instructions
        CX_SEG(mma1, F1, 0, 32, F0, xs, ws, 0, 0, 0, 2, 0, RS_ == 2 ? 8 : DS_, 0, 4);
        // this wave's reads of the current slots are complete (F1 has landed) ...
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // ... and so are its DMA writes of the next iteration's operands: everything but the 4 X instructions just issued (the
        // other 4 of that K-tile follow below).  "All but the newest 4" stays correct across a tile end: the epilogue's global
        // stores are older than the next iteration's newest 4 and retire in order with them (gfx9 has one in-order vmcnt for
        // loads and stores), so they can only make that wait stronger, never weaker.
        if constexpr ((DBG & 128) != 0) {   // trace builds: the counted DMA wait of a tile's FIRST K-tile (ph[0]: it also waits for the previous
            // tile's stores, older on gfx9's one in-order vmcnt) and of the others (ph[1]), separately; their sum stays inside ph[2]
            const long long w0_ = (long long)__builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            const long long w1_ = (long long)__builtin_amdgcn_s_memtime();
            ph[decltype(first)::value ? 0 : 1] += w1_ - w0_;
        } else if constexpr ((DBG & 32) == 0) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
        if constexpr ((DBG & 2) == 0) __builtin_amdgcn_s_barrier();
        const char* nxs = dsm + nxs_slot * XS6;
        const char* nws = dsm + (3 + nws_slot) * XS6;
        // second half: the fragments of the next K-tile's k-step 0 -> F0 (at a tile end these are the first fragments of the
        // next tile: its operands have landed too), then the other 4 X instructions
        CX_SEG(mma1, F1, 32, 64, F0, nxs, nws, 0, RS_ == 2 ? 32 : 33, RS_, 2, RS_ == 2 ? 33 : 49, RS_ == 2 ? 8 : DS_, 4, 4);
        x_advance();
        if constexpr ((DBG & 128) != 0 && decltype(first)::value) {   // trace builds: the tile's first K-tile on its own (ph[0] reused: see below)
            const long long now_ = (long long)__builtin_amdgcn_s_memtime();
            kt0_cyc += now_ - t_last;
        }
        ++cp_kt;
        pxs_slot = xs_slot;
        pws_slot = ws_slot;
        xs_slot = nxs_slot;
        ws_slot = nws_slot;
    };

#pragma unroll 1
    while (cp_tile >= 0) {
        cp_kt = 0;
        kt_body(std::true_type{});
#pragma unroll 1
        while (cp_kt < nk) kt_body(std::false_type{});

        {
            CX_PH(2);
            // MFMA results are read by VALU below; the hazard recogniser does not see through the inline asm
            asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
            // ---- epilogue.  Free LDS: the X slot and the W slot just consumed (32 KiB each); waves 0,1 stage in the
            // former, waves 2,3 in the latter (16 KiB per wave).  In-flight DMA targets other slots.
            const int tn = cp_tile & 255, tm = cp_tile >> 8;
            const int m0 = tm * BM6 + wm * 128, n0 = tn * BN6 + wn * 128;
            char* my = ((wave < 2) ? dsm + pxs_slot * XS6 : dsm + (3 + pws_slot) * XS6) + (wave & 1) * 16384;
            constexpr int ROWB = 272;  // 32 staged rows x 128 bf16 (+16 B pad) = 8704 B
            // Where a lane's accumulator registers sit in a 32-row pass (gemm_v6_acc.inc): piece q (4 consecutive columns, registers
            // 4q .. 4q + 3 of v6_read_block) of the 32-column region a is row prow(q), columns a * 32 + pcol(q) .. + 3.  In a staging
            // area of `rowb` bytes per row that is base(prow(0), pcol(0)) + a * 64 + qoff(q, rowb) bytes -- one address register and
            // immediates.  32 lanes of an 8-byte staging write cover 16 rows x 16 bytes at a row stride of 68 / 36 dwords: 64 banks once.
            auto prow = [&](int q) { return (q >> 1) * 16 + l15; };
            auto pcol = [&](int q) { return (q & 1) * 16 + 4 * g4; };
            auto qoff = [](int q, int rowb) { return (q >> 1) * 16 * rowb + (q & 1) * 32; };
            if constexpr ((DBG & 16) != 0) {
                // ablation: no epilogue
            } else if constexpr (EPI == GEMM_EPI_NONE || EPI == GEMM_EPI_ACT_BWD || EPI == GEMM_EPI_QACT_BWD) {
                // ACT_BWD (round 6): the same epilogue with the residual slot holding the saved pre-activation of the plain MLP and
                // the combine step Out = bf16(bf16(acc) * act'(pre)) instead of an add -- the arithmetic of the standalone
                // cx_bias_act_bwd_colsum pass on the bf16 d(act) this GEMM used to write (bit-identical), without the (M, N) round
                // trip; the column sums of the bf16 result (the fc1 bias gradient) leave as one fp32 partial row per 128-row block.
                constexpr bool ACTB = EPI == GEMM_EPI_ACT_BWD || EPI == GEMM_EPI_QACT_BWD;
                float cs0 = 0.f, cs1 = 0.f, cs2 = 0.f, cs3 = 0.f, cs4 = 0.f, cs5 = 0.f, cs6 = 0.f, cs7 = 0.f;
                constexpr int act_kind = EPI == GEMM_EPI_QACT_BWD ? CX_ACT_QUICK_GELU : CX_ACT_GELU;
                auto mul_grad = [&](uint4& v, const uint4& r) {
#define CX_MG(w_, ca, cb)                                                                              \
    {                                                                                                 \
        const float o0 = bf16lo_to_f32(v.w_) * act_grad(bf16lo_to_f32(r.w_), act_kind);               \
        const float o1 = bf16hi_to_f32(v.w_) * act_grad(bf16hi_to_f32(r.w_), act_kind);               \
        v.w_ = pack_bf16x2(o0, o1);                                                                   \
        ca += bf16lo_to_f32(v.w_);                                                                    \
        cb += bf16hi_to_f32(v.w_);                                                                    \
    }
                    CX_MG(x, cs0, cs1) CX_MG(y, cs2, cs3) CX_MG(z, cs4, cs5) CX_MG(w, cs6, cs7)
#undef CX_MG
                };
                // column sums of this wave's 128 rows: fold the four row groups (lane >> 4), lanes 0..15 own 8 columns each
                auto flush_colsum = [&]() {
                    if constexpr (ACTB) {
                        float c[8] = {cs0, cs1, cs2, cs3, cs4, cs5, cs6, cs7};
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            c[e] += __shfl_xor(c[e], 16, 64);
                            c[e] += __shfl_xor(c[e], 32, 64);
                        }
                        const int n = n0 + (lane & 15) * 8;
                        if (p.colsum_part && lane < 16 && m0 < p.M && n + 8 <= p.N) {
                            float* dst = p.colsum_part + (size_t)(m0 >> 7) * p.N + n;
                            *reinterpret_cast<float4*>(dst) = make_float4(c[0], c[1], c[2], c[3]);
                            *reinterpret_cast<float4*>(dst + 4) = make_float4(c[4], c[5], c[6], c[7]);
                        }
                    }
                };
                const bool add_bias = p.bias != nullptr;
                // `plain` (compile time): alpha == 1 and no bias -- the forward / dgrad launches of bias-free models;
                // saves 256 multiplies per lane per tile in an epilogue that is VALU-issue-bound (one wave per SIMD)
                // optional residual (p.Out2, same (M, N) bf16 layout as the output): Out = bf16(bf16(result) + residual) --
                // the `x0 + residual` of dropout_add_layer_norm (sc/layers/block.py:422-431) moves from the HBM-bound
                // LayerNorm kernel into this epilogue, where it overlaps other workgroups' MFMA phases.
                const bf16_t* resid = reinterpret_cast<const bf16_t*>(p.Out2);
                // ---- fast path: interior tile, alpha == 1, no bias (every launch of the bias-free trunk).  What the
                // first version lost (13 k cycles per tile = 29 % of a K = 768 tile, scripts/gemm_ablate.py): each
                // 16-byte row piece went LDS read -> wait -> predicated store one at a time (8 LDS round trips per pass),
                // and the residual loads of pass b + 1, issued after the stores of pass b, could only be waited for
                // together with those stores (gfx9 counts loads and stores on one in-order vmcnt) -- the wave sat
                // through a store round trip per pass.  Here a pass is software-pipelined: all 8 row reads are issued
                // at once, the accumulator read + convert of the NEXT pass runs under their latency, the residual loads
                // of the next pass are issued BEFORE this pass's stores, and nothing is predicated.
                auto fast_tile = [&](auto with_resid) {
                    constexpr bool RES = decltype(with_resid)::value;
                    const int rrow = lane >> 4, rch = lane & 15;
                    bf16_t* outp = reinterpret_cast<bf16_t*>(p.Out) + (size_t)(m0 + rrow) * p.ldo + n0 + rch * 8;
                    const bf16_t* resp = RES ? resid + (size_t)(m0 + rrow) * p.ldo2 + n0 + rch * 8 : nullptr;
                    const char* rd = my + rrow * ROWB + rch * 16;
                    char* wr = my + prow(0) * ROWB + pcol(0) * 2;
                    uint2 pk[16];
                    auto pack_pass = [&](int b) {
#pragma unroll
                        for (int a = 0; a < 4; ++a) {
                            float blk[16];
                            v6_read_block(4 * b + a, blk);
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                pk[a * 4 + q].x = pack_bf16x2(blk[4 * q], blk[4 * q + 1]);
                                pk[a * 4 + q].y = pack_bf16x2(blk[4 * q + 2], blk[4 * q + 3]);
                            }
                        }
                    };
                    auto stage = [&]() {
#pragma unroll
                        for (int a = 0; a < 4; ++a)
#pragma unroll
                            for (int q = 0; q < 4; ++q) *reinterpret_cast<uint2*>(wr + a * 64 + qoff(q, ROWB)) = pk[a * 4 + q];
                    };
                    // (named registers and explicit pass constants: a `uint4 rr[8]` carried across the passes of an
                    // unrolled loop is left in scratch memory by the compiler)
                    uint4 r0 = {}, r1 = {}, r2 = {}, r3 = {}, r4 = {}, r5 = {}, r6 = {}, r7 = {};
#define CX_RES_ROWS(b_)                                                                                   \
    r0 = gld(resp + (size_t)((b_) * 32 + 0) * p.ldo2);                          \
    r1 = gld(resp + (size_t)((b_) * 32 + 4) * p.ldo2);                          \
    r2 = gld(resp + (size_t)((b_) * 32 + 8) * p.ldo2);                          \
    r3 = gld(resp + (size_t)((b_) * 32 + 12) * p.ldo2);                         \
    r4 = gld(resp + (size_t)((b_) * 32 + 16) * p.ldo2);                         \
    r5 = gld(resp + (size_t)((b_) * 32 + 20) * p.ldo2);                         \
    r6 = gld(resp + (size_t)((b_) * 32 + 24) * p.ldo2);                         \
    r7 = gld(resp + (size_t)((b_) * 32 + 28) * p.ldo2);
                    auto add_res = [&](uint4& v, const uint4& r) {
                        v.x = pack_bf16x2(bf16lo_to_f32(v.x) + bf16lo_to_f32(r.x), bf16hi_to_f32(v.x) + bf16hi_to_f32(r.x));
                        v.y = pack_bf16x2(bf16lo_to_f32(v.y) + bf16lo_to_f32(r.y), bf16hi_to_f32(v.y) + bf16hi_to_f32(r.y));
                        v.z = pack_bf16x2(bf16lo_to_f32(v.z) + bf16lo_to_f32(r.z), bf16hi_to_f32(v.z) + bf16hi_to_f32(r.z));
                        v.w = pack_bf16x2(bf16lo_to_f32(v.w) + bf16lo_to_f32(r.w), bf16hi_to_f32(v.w) + bf16hi_to_f32(r.w));
                    };
                    auto one_pass = [&](auto bc) {
                        constexpr int b = decltype(bc)::value;
                        uint4 v0 = *reinterpret_cast<const uint4*>(rd + 0 * ROWB), v1 = *reinterpret_cast<const uint4*>(rd + 4 * ROWB),
                              v2 = *reinterpret_cast<const uint4*>(rd + 8 * ROWB), v3 = *reinterpret_cast<const uint4*>(rd + 12 * ROWB),
                              v4 = *reinterpret_cast<const uint4*>(rd + 16 * ROWB), v5 = *reinterpret_cast<const uint4*>(rd + 20 * ROWB),
                              v6 = *reinterpret_cast<const uint4*>(rd + 24 * ROWB), v7 = *reinterpret_cast<const uint4*>(rd + 28 * ROWB);
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (b < 3) pack_pass(b + 1);
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (RES) {
                            add_res(v0, r0); add_res(v1, r1); add_res(v2, r2); add_res(v3, r3);
                            add_res(v4, r4); add_res(v5, r5); add_res(v6, r6); add_res(v7, r7);
                            if constexpr (b < 3) {  // next pass's residual rows: issued BEFORE this pass's stores (one in-order vmcnt)
                                __builtin_amdgcn_sched_barrier(0);
                                CX_RES_ROWS(b + 1)
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                        if constexpr (b < 3) stage();  // the LDS executes a wave's operations in order: these follow the row reads
                        bf16_t* o = outp + (size_t)(b * 32) * p.ldo;
                        if constexpr ((DBG & 64) != 0) {  // ablation: everything but the stores
                            const uint4 vs[8] = {v0, v1, v2, v3, v4, v5, v6, v7};
#pragma unroll
                            for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(vs[i].x), "v"(vs[i].y), "v"(vs[i].z), "v"(vs[i].w));
                            asm volatile("" ::"v"(o));
                            return;
                        }
                        gst(o, v0);
                        gst(o + (size_t)4 * p.ldo, v1);
                        gst(o + (size_t)8 * p.ldo, v2);
                        gst(o + (size_t)12 * p.ldo, v3);
                        gst(o + (size_t)16 * p.ldo, v4);
                        gst(o + (size_t)20 * p.ldo, v5);
                        gst(o + (size_t)24 * p.ldo, v6);
                        gst(o + (size_t)28 * p.ldo, v7);
                    };
                    if constexpr (RES) { CX_RES_ROWS(0) }
                    pack_pass(0);
                    stage();
                    one_pass(std::integral_constant<int, 0>{});
                    one_pass(std::integral_constant<int, 1>{});
                    one_pass(std::integral_constant<int, 2>{});
                    one_pass(std::integral_constant<int, 3>{});
#undef CX_RES_ROWS
                };
                // ---- ACT_BWD, interior tiles: the plain fast path keeps 8 staged rows + 8 residual rows + the next pass's 16 packed pairs in
                // registers (it sits at 256 VGPRs); with the activation derivative's temporaries on top the compiler parked values in the
                // AGPRs -- inside the accumulators it cannot see (build.py's audit).  Here a pass is staged block by block (4 registers at a
                // time) and leaves in two halves of 4 rows per lane: 4 staged + 4 pre-activation rows live, the next half's pre-activation
                // rows requested BEFORE this half's stores (one in-order vmcnt).  The epilogue is VALU-bound either way (~20 instructions
                // per element for the erf form): what it hides is the (M, N) d(act) write + read and the pre-activation's second read.
                auto fast_tile_actb = [&]() {
                    const int rrow = lane >> 4, rch = lane & 15;
                    bf16_t* outp = reinterpret_cast<bf16_t*>(p.Out) + (size_t)(m0 + rrow) * p.ldo + n0 + rch * 8;
                    const bf16_t* resp = resid + (size_t)(m0 + rrow) * p.ldo2 + n0 + rch * 8;
                    const char* rd = my + rrow * ROWB + rch * 16;
                    char* wr = my + prow(0) * ROWB + pcol(0) * 2;
                    auto stage_pass = [&](int b) {
#pragma unroll
                        for (int a = 0; a < 4; ++a) {
                            float blk[16];
                            v6_read_block(4 * b + a, blk);
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                uint2 pk2;
                                pk2.x = pack_bf16x2(blk[4 * q], blk[4 * q + 1]);
                                pk2.y = pack_bf16x2(blk[4 * q + 2], blk[4 * q + 3]);
                                *reinterpret_cast<uint2*>(wr + a * 64 + qoff(q, ROWB)) = pk2;
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    };
                    uint4 r0, r1, r2, r3;
#define CX_PRE_ROWS(hh_)                                                        \
    r0 = gld(resp + (size_t)((hh_) * 16 + 0) * p.ldo2);                         \
    r1 = gld(resp + (size_t)((hh_) * 16 + 4) * p.ldo2);                         \
    r2 = gld(resp + (size_t)((hh_) * 16 + 8) * p.ldo2);                         \
    r3 = gld(resp + (size_t)((hh_) * 16 + 12) * p.ldo2);
                    auto half_pass = [&](auto hc) {
                        constexpr int hh = decltype(hc)::value;   // half-pass index 0..7: rows hh * 16 + {0, 4, 8, 12} + rrow of the wave's 128
                        constexpr int h = hh & 1;
                        uint4 v0 = *reinterpret_cast<const uint4*>(rd + (h * 16 + 0) * ROWB), v1 = *reinterpret_cast<const uint4*>(rd + (h * 16 + 4) * ROWB),
                              v2 = *reinterpret_cast<const uint4*>(rd + (h * 16 + 8) * ROWB), v3 = *reinterpret_cast<const uint4*>(rd + (h * 16 + 12) * ROWB);
                        __builtin_amdgcn_sched_barrier(0);
                        mul_grad(v0, r0); __builtin_amdgcn_sched_barrier(0);
                        mul_grad(v1, r1); __builtin_amdgcn_sched_barrier(0);
                        mul_grad(v2, r2); __builtin_amdgcn_sched_barrier(0);
                        mul_grad(v3, r3); __builtin_amdgcn_sched_barrier(0);
                        if constexpr (hh < 7) { CX_PRE_ROWS(hh + 1) }
                        __builtin_amdgcn_sched_barrier(0);
                        bf16_t* o = outp + (size_t)(hh * 16) * p.ldo;
                        gst(o, v0);
                        gst(o + (size_t)4 * p.ldo, v1);
                        gst(o + (size_t)8 * p.ldo, v2);
                        gst(o + (size_t)12 * p.ldo, v3);
                        if constexpr (h == 1 && hh < 7) stage_pass((hh >> 1) + 1);   // (the LDS executes a wave's operations in order: after this pass's row reads)
                    };
                    CX_PRE_ROWS(0)
                    stage_pass(0);
                    half_pass(std::integral_constant<int, 0>{}); half_pass(std::integral_constant<int, 1>{});
                    half_pass(std::integral_constant<int, 2>{}); half_pass(std::integral_constant<int, 3>{});
                    half_pass(std::integral_constant<int, 4>{}); half_pass(std::integral_constant<int, 5>{});
                    half_pass(std::integral_constant<int, 6>{}); half_pass(std::integral_constant<int, 7>{});
#undef CX_PRE_ROWS
                };
                const bool fast = p.alpha == 1.f && !add_bias && m0 + 128 <= p.M && n0 + 128 <= p.N;
                auto store_tile = [&](auto plain) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) {  // 32 rows of the wave's 128 per pass
                        uint4 r0 = {}, r1 = {}, r2 = {}, r3 = {}, r4 = {}, r5 = {}, r6 = {}, r7 = {};
                        if (resid) {  // issued before the conversion work below; named registers (see SWIGLU_BWD)
#define CX_RES_LOAD(ps, dst)                                                                              \
    {                                                                                                 \
        int m_ = m0 + b * 32 + (ps) * 4 + (lane >> 4);                                                \
        m_ = m_ < p.M ? m_ : p.M - 1;                                                                 \
        int n_ = n0 + (lane & 15) * 8;                                                                \
        n_ = n_ + 8 <= p.N ? n_ : 0;                                                                  \
        dst = gld(resid + (size_t)m_ * p.ldo2 + n_);                      \
    }
                            CX_RES_LOAD(0, r0) CX_RES_LOAD(1, r1) CX_RES_LOAD(2, r2) CX_RES_LOAD(3, r3)
                            CX_RES_LOAD(4, r4) CX_RES_LOAD(5, r5) CX_RES_LOAD(6, r6) CX_RES_LOAD(7, r7)
#undef CX_RES_LOAD
                        }
#pragma unroll
                        for (int a = 0; a < 4; ++a) {
                            float blk[16];
                            v6_read_block(4 * b + a, blk);
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int nl = a * 32 + pcol(q);
                                float v[4];
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = blk[4 * q + e];
                                if constexpr (!decltype(plain)::value) {
#pragma unroll
                                    for (int e = 0; e < 4; ++e) v[e] *= p.alpha;
                                    if (add_bias) {
                                        const int n = n0 + nl;
                                        if (n < p.N) {
                                            const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
                                            v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                                        }
                                    }
                                }
                                uint2 pk;
                                pk.x = pack_bf16x2(v[0], v[1]);
                                pk.y = pack_bf16x2(v[2], v[3]);
                                *reinterpret_cast<uint2*>(my + prow(q) * ROWB + nl * 2) = pk;
                            }
                        }
#pragma unroll
                        for (int ps = 0; ps < 8; ++ps) {
                            const int row = ps * 4 + (lane >> 4), ch = lane & 15;
                            const int m = m0 + b * 32 + row, n = n0 + ch * 8;
                            uint4 vv = *reinterpret_cast<const uint4*>(my + row * ROWB + ch * 16);
                            if (resid && ACTB) {
                                const uint4 rr = ps == 0 ? r0 : ps == 1 ? r1 : ps == 2 ? r2 : ps == 3 ? r3 : ps == 4 ? r4 : ps == 5 ? r5
                                                 : ps == 6 ? r6 : r7;
                                if (m < p.M && n + 8 <= p.N) mul_grad(vv, rr);   // (only rows / columns that exist enter the column sums)
                            } else if (resid) {
                                const uint4 rr = ps == 0 ? r0 : ps == 1 ? r1 : ps == 2 ? r2 : ps == 3 ? r3 : ps == 4 ? r4 : ps == 5 ? r5
                                                 : ps == 6 ? r6 : r7;
                                vv.x = pack_bf16x2(bf16lo_to_f32(vv.x) + bf16lo_to_f32(rr.x), bf16hi_to_f32(vv.x) + bf16hi_to_f32(rr.x));
                                vv.y = pack_bf16x2(bf16lo_to_f32(vv.y) + bf16lo_to_f32(rr.y), bf16hi_to_f32(vv.y) + bf16hi_to_f32(rr.y));
                                vv.z = pack_bf16x2(bf16lo_to_f32(vv.z) + bf16lo_to_f32(rr.z), bf16hi_to_f32(vv.z) + bf16hi_to_f32(rr.z));
                                vv.w = pack_bf16x2(bf16lo_to_f32(vv.w) + bf16lo_to_f32(rr.w), bf16hi_to_f32(vv.w) + bf16hi_to_f32(rr.w));
                            }
                            if (m < p.M && n + 8 <= p.N)
                                gst(reinterpret_cast<bf16_t*>(p.Out) + (size_t)m * p.ldo + n, vv);
                        }
                    }
                };
                if constexpr (ACTB) {   // (the launcher guarantees Pre, alpha == 1 and no bias)
                    if (fast) {
                        fast_tile_actb();
                    } else {
                        store_tile(std::true_type{});
                    }
                    flush_colsum();
                } else if (fast) {
                    if (resid) {
                        fast_tile(std::true_type{});
                    } else {
                        fast_tile(std::false_type{});
                    }
                } else if (p.alpha == 1.f && !add_bias) {
                    store_tile(std::true_type{});
                } else {
                    store_tile(std::false_type{});
                }
            } else if constexpr (IS_SWIGLU_BWD) {
                // fc2 dgrad with the SwiGLU backward in the epilogue (sc/layers/mlp.py:75 swiglu, its autograd): the tile
                // is d(act) for 128 activation columns of this wave = 256 columns [y0|g0|y1|g1|y2|g2|y3|g3] of the
                // pre-activation tensor YG (p.Out2, input) and of its gradient dYG (p.Out).  d(act) is never written; the
                // HBM-bound elementwise pass (read YG + d(act), write dYG) now overlaps other workgroups' MFMA phases.
                // Per pass of 32 rows: YG block -> LDS (coalesced 16-B loads), every lane updates its (row, 4 columns)
                // cells in place, dYG block LDS -> HBM (coalesced).  [32 rows][512 B], 16-B chunk index ^ row.
                // AG form (round 3): the forward saved the gate alone, (M, I), next to the activation it writes anyway; the
                // (y, gate) block is assembled from the two tensors -- y slots take act, gate slots take gate, the same 16
                // row loads per pass, each lane picks its tensor -- and y is recovered inside the derivative:
                //   d gate = d * y * silu'(g),  y = act / silu(g)   =>   d gate = d * act * (1 / g + 1 - sigmoid(g))
                // (act is y * silu(g) rounded once to bf16, so the recovered y carries the same 2^-9 the saved bf16 y had).
                const bf16_t* yg_in = reinterpret_cast<const bf16_t*>(p.Out2);
                const bf16_t* g_in = reinterpret_cast<const bf16_t*>(p.In3);
                bf16_t* dyg = reinterpret_cast<bf16_t*>(p.Out);
                const int c0 = 2 * n0;  // first pre-activation column of this wave
                // d(gate) coefficient per element: silu'(g) * y from (y, g)  |  act * (1/g + 1 - sg) from (act, g)
                // one element: d(act) dv (fp32, straight from the accumulator -- round 4: the bf16 round trip the standalone op
                // implies cost 1.5 instructions per element of an epilogue that is VALU-issue-bound, and fp32 is the better
                // number), saved (y | act) yv and gate gv -> d y, d gate
                auto bwd_elem = [&](float yv, float gv, float dv, float& dyo, float& dgo) {
                    if constexpr (AG) {
                        swiglu_bwd_from_act(dv, yv, gv, dyo, dgo);
                    } else {
                        const float sg = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * gv));
                        const float gs = gv * sg;
                        dyo = gs * dv;
                        dgo = (sg + gs * (1.f - sg)) * dv * yv;
                    }
                };
                // four elements of a lane's (row, 4 columns) cell: the (act, gate) form on explicit pairs
                auto bwd_quad = [&](const float (&yv)[4], const float (&gv)[4], const float* dv, float (&dyo)[4], float (&dgo)[4]) {
                    if constexpr (AG) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            cx_f2 y2, g2;
                            swiglu_bwd_from_act2(cx_f2{dv[2 * h], dv[2 * h + 1]}, cx_f2{yv[2 * h], yv[2 * h + 1]},
                                                 cx_f2{gv[2 * h], gv[2 * h + 1]}, y2, g2);
                            dyo[2 * h] = y2.x; dyo[2 * h + 1] = y2.y;
                            dgo[2 * h] = g2.x; dgo[2 * h + 1] = g2.y;
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) bwd_elem(yv[e], gv[e], dv[e], dyo[e], dgo[e]);
                    }
                };
                auto cell = [&](int row, int colbyte) { return my + row * 512 + ((((colbyte >> 4) ^ row) & 31) << 4) + (colbyte & 15); };
                // ---- fast path (interior tiles).  The first version ran its four 32-row passes strictly one after the
                // other: the (y, gate) loads of pass b + 1 were issued after the dYG stores of pass b, and on gfx9 a load
                // can only be waited for together with every older store (one in-order vmcnt) -- each pass sat through a
                // full store round trip before its data was even requested.  Here the 16 row loads of pass b + 1 are
                // issued right after pass b's rows have been staged (their registers are free from then on) and BEFORE
                // pass b's stores; they land while the wave does the pass's sigmoid arithmetic.
                const bool fast_bwd = m0 + 128 <= p.M && n0 + 128 <= p.N;
                if (fast_bwd) {
                    int lane_e = lane;
#if CX_V6_OPAQUE >= 1
                    asm volatile("" : "+v"(lane_e));
#endif
                    const int lrow = lane_e >> 5, lch = lane_e & 31;
                    // (AG: chunk lch of the 512-B row = columns [8 lch, 8 lch + 8) of [y0|g0|y1|g1|y2|g2|y3|g3]: 32-column group
                    // lch >> 3 of this wave's 128 activation columns, from Act (y slot) or G (gate slot))
                    const bf16_t* src = AG ? ((lch & 4) ? g_in : yg_in) + (size_t)(m0 + lrow) * p.ldo2 + n0 + (lch >> 3) * 32 + (lch & 3) * 8
                                           : yg_in + (size_t)(m0 + lrow) * p.ldo2 + c0 + lch * 8;
                    bf16_t* dst = dyg + (size_t)(m0 + lrow) * p.ldo + c0 + lch * 8;
                    uint4 t0, t1, t2, t3, t4, t5, t6, t7, t8, t9, t10, t11, t12, t13, t14, t15;
#define CX_L(i, b_) t##i = gld(src + (size_t)((b_) * 32 + (i) * 2) * p.ldo2);
#define CX_LOAD_LO(b_) CX_L(0, b_) CX_L(1, b_) CX_L(2, b_) CX_L(3, b_) CX_L(4, b_) CX_L(5, b_) CX_L(6, b_) CX_L(7, b_)
#define CX_LOAD_HI(b_) CX_L(8, b_) CX_L(9, b_) CX_L(10, b_) CX_L(11, b_) CX_L(12, b_) CX_L(13, b_) CX_L(14, b_) CX_L(15, b_)
#define CX_S(i) *reinterpret_cast<uint4*>(cell((i) * 2 + lrow, lch * 16)) = t##i;
#define CX_STAGE_ALL CX_S(0) CX_S(1) CX_S(2) CX_S(3) CX_S(4) CX_S(5) CX_S(6) CX_S(7) CX_S(8) CX_S(9) CX_S(10) CX_S(11) CX_S(12) CX_S(13) CX_S(14) CX_S(15)
                    auto one_pass = [&](auto bc) {
                        constexpr int b = decltype(bc)::value;
                        int lane_p = lane_e;
#if CX_V6_OPAQUE >= 2
                        asm volatile("" : "+v"(lane_p));
#endif
                        const int lrow = lane_p >> 5, lch = lane_p & 31, l15 = lane_p & 15, g4 = lane_p >> 4;   // (shadow the kernel's)
                        CX_STAGE_ALL
                        __builtin_amdgcn_sched_barrier(0);
                        CX_PH(4);
                        if constexpr (b < 3) { CX_LOAD_LO(b + 1) }  // rows 0..15 of the next pass fly under the arithmetic
#if CX_V6_HI_EARLY
                        if constexpr (b < 3) { CX_LOAD_HI(b + 1) }  // ... and rows 16..31 with them (round 5)
#endif
                        __builtin_amdgcn_sched_barrier(0);
                        CX_PH(5);
#pragma unroll
                        for (int a = 0; a < 4; ++a) {
                            float da[16];
                            v6_read_block(4 * b + a, da);
                            // (round 6: the block's 8 cells are read up front -- a lane reads and writes only its own cells, so nothing orders
                            // a quad's reads behind the previous quad's writes but the compiler's alias analysis; read / wait / compute / write
                            // per quad exposed an LDS round trip 16 times per pass: ~30 % of this phase, scripts/gemm_swiglu_trace.py)
                            uint2 yy[4], gg[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                yy[q] = *reinterpret_cast<const uint2*>(cell((q >> 1) * 16 + l15, (a * 64 + (q & 1) * 16 + 4 * g4) * 2));
                                gg[q] = *reinterpret_cast<const uint2*>(cell((q >> 1) * 16 + l15, (a * 64 + 32 + (q & 1) * 16 + 4 * g4) * 2));
                            }
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                char* py = cell((q >> 1) * 16 + l15, (a * 64 + (q & 1) * 16 + 4 * g4) * 2);
                                char* pg = cell((q >> 1) * 16 + l15, (a * 64 + 32 + (q & 1) * 16 + 4 * g4) * 2);
                                const float y[4] = {bf16lo_to_f32(yy[q].x), bf16hi_to_f32(yy[q].x), bf16lo_to_f32(yy[q].y), bf16hi_to_f32(yy[q].y)};
                                const float g[4] = {bf16lo_to_f32(gg[q].x), bf16hi_to_f32(gg[q].x), bf16lo_to_f32(gg[q].y), bf16hi_to_f32(gg[q].y)};
                                float dy[4], dg[4];
                                bwd_quad(y, g, &da[4 * q], dy, dg);
                                uint2 o;
                                o.x = pack_bf16x2(dy[0], dy[1]); o.y = pack_bf16x2(dy[2], dy[3]);
                                *reinterpret_cast<uint2*>(py) = o;
                                o.x = pack_bf16x2(dg[0], dg[1]); o.y = pack_bf16x2(dg[2], dg[3]);
                                *reinterpret_cast<uint2*>(pg) = o;
                            }
                            __builtin_amdgcn_sched_barrier(0);  // one accumulator block at a time (register pressure)
                        }
#if !CX_V6_HI_EARLY
                        if constexpr (b < 3) { CX_LOAD_HI(b + 1) }  // rows 16..31: still ahead of this pass's stores
#endif
                        __builtin_amdgcn_sched_barrier(0);
                        CX_PH(6);
#pragma unroll
                        for (int h = 0; h < 4; ++h) {  // (named registers: a small uint4 array lands in scratch memory here)
                            const uint4 v0 = *reinterpret_cast<const uint4*>(cell((h * 4 + 0) * 2 + lrow, lch * 16));
                            const uint4 v1 = *reinterpret_cast<const uint4*>(cell((h * 4 + 1) * 2 + lrow, lch * 16));
                            const uint4 v2 = *reinterpret_cast<const uint4*>(cell((h * 4 + 2) * 2 + lrow, lch * 16));
                            const uint4 v3 = *reinterpret_cast<const uint4*>(cell((h * 4 + 3) * 2 + lrow, lch * 16));
                            bf16_t* o = dst + (size_t)(b * 32 + h * 8) * p.ldo;
                            gst(o, v0);
                            gst(o + (size_t)2 * p.ldo, v1);
                            gst(o + (size_t)4 * p.ldo, v2);
                            gst(o + (size_t)6 * p.ldo, v3);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        CX_PH(7);
                    };
                    CX_LOAD_LO(0) CX_LOAD_HI(0)
                    __builtin_amdgcn_sched_barrier(0);
                    CX_PH(3);
                    one_pass(std::integral_constant<int, 0>{});
                    one_pass(std::integral_constant<int, 1>{});
                    one_pass(std::integral_constant<int, 2>{});
                    one_pass(std::integral_constant<int, 3>{});
#undef CX_L
#undef CX_LOAD_LO
#undef CX_LOAD_HI
#undef CX_S
#undef CX_STAGE_ALL
                } else
#pragma unroll 1
                for (int b = 0; b < 4; ++b) {
                    // 2 x 8 loads in flight, held in NAMED registers (an L2 prefetch during the last K-tile and 4 x 4 groups measured slower) (a `uint4 in[16]` array lands in scratch memory here)
                    // (edge tiles only: the lane index is made opaque so that this path's ~50 staging addresses are not hoisted out of the
                    // tile loop, where they would sit in -- or spill from -- the registers of every interior tile's K loop)
                    int lane_c = lane;
                    asm volatile("" : "+v"(lane_c));
                    const int lrow = lane_c >> 5, lch = lane_c & 31, l15 = lane_c & 15, g4 = lane_c >> 4;
#define CX_YG_LOAD(i)                                                                                   \
    uint4 in##i;                                                                                      \
    {                                                                                                 \
        int m_ = m0 + b * 32 + (i) * 2 + lrow;                                                        \
        m_ = m_ < p.M ? m_ : p.M - 1;                                                                 \
        in##i = AG ? *reinterpret_cast<const uint4*>(((lch & 4) ? g_in : yg_in) + (size_t)m_ * p.ldo2 + n0 + (lch >> 3) * 32 + (lch & 3) * 8) \
                   : *reinterpret_cast<const uint4*>(yg_in + (size_t)m_ * p.ldo2 + c0 + lch * 8);     \
    }
#define CX_YG_STAGE(i) *reinterpret_cast<uint4*>(cell((i) * 2 + lrow, lch * 16)) = in##i;
                    {
                        CX_YG_LOAD(0) CX_YG_LOAD(1) CX_YG_LOAD(2) CX_YG_LOAD(3) CX_YG_LOAD(4) CX_YG_LOAD(5) CX_YG_LOAD(6)
                        CX_YG_LOAD(7)
                        CX_YG_STAGE(0) CX_YG_STAGE(1) CX_YG_STAGE(2) CX_YG_STAGE(3) CX_YG_STAGE(4) CX_YG_STAGE(5)
                        CX_YG_STAGE(6) CX_YG_STAGE(7)
                    }
                    {
                        CX_YG_LOAD(8) CX_YG_LOAD(9) CX_YG_LOAD(10) CX_YG_LOAD(11) CX_YG_LOAD(12) CX_YG_LOAD(13)
                        CX_YG_LOAD(14) CX_YG_LOAD(15)
                        CX_YG_STAGE(8) CX_YG_STAGE(9) CX_YG_STAGE(10) CX_YG_STAGE(11) CX_YG_STAGE(12) CX_YG_STAGE(13)
                        CX_YG_STAGE(14) CX_YG_STAGE(15)
                    }
#undef CX_YG_LOAD
#undef CX_YG_STAGE
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        float da[16];
                        v6_read_block(4 * b + a, da);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            char* py = cell((q >> 1) * 16 + l15, (a * 64 + (q & 1) * 16 + 4 * g4) * 2);
                            char* pg = cell((q >> 1) * 16 + l15, (a * 64 + 32 + (q & 1) * 16 + 4 * g4) * 2);
                            const uint2 yy = *reinterpret_cast<const uint2*>(py);
                            const uint2 gg = *reinterpret_cast<const uint2*>(pg);
                            const float y[4] = {bf16lo_to_f32(yy.x), bf16hi_to_f32(yy.x), bf16lo_to_f32(yy.y), bf16hi_to_f32(yy.y)};
                            const float g[4] = {bf16lo_to_f32(gg.x), bf16hi_to_f32(gg.x), bf16lo_to_f32(gg.y), bf16hi_to_f32(gg.y)};
                            float dy[4], dg[4];
                            bwd_quad(y, g, &da[4 * q], dy, dg);
                            uint2 o;
                            o.x = pack_bf16x2(dy[0], dy[1]); o.y = pack_bf16x2(dy[2], dy[3]);
                            *reinterpret_cast<uint2*>(py) = o;
                            o.x = pack_bf16x2(dg[0], dg[1]); o.y = pack_bf16x2(dg[2], dg[3]);
                            *reinterpret_cast<uint2*>(pg) = o;
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int row = i * 2 + lrow, ch = lch;
                        const int m = m0 + b * 32 + row;
                        const uint4 vv = *reinterpret_cast<const uint4*>(cell(row, ch * 16));
                        if (m < p.M) gst(dyg + (size_t)m * p.ldo + c0 + ch * 8, vv);
                    }
                }
            } else if constexpr (EPI == GEMM_EPI_GELU || EPI == GEMM_EPI_QGELU) {
                // fc1 of the plain MLP (sc/layers/mlp.py:30-34): pre = acc + bias (bf16, kept for backward when p.Out is
                // set), act = gelu_erf(pre).  The standalone op sees the bf16-rounded pre-activation; so does this one.
                constexpr int act_kind = EPI == GEMM_EPI_QGELU ? CX_ACT_QUICK_GELU : CX_ACT_GELU;
                // ---- fast path (interior tiles; round 6).  The first version below costs the launch +39 ... +49 % over the plain GEMM of the same
                // shape (profiles/r6_microbench_gemm2048.txt): per pass it re-loads its 16 bias quads from global memory (the compiler cannot
                // carry them across the stores), and each of its two outputs goes stage -> wait -> 8 x (row read -> predicated store) with the
                // LDS round trip exposed.  Here: the bias in registers once per tile (a lane has only 8 distinct quads: piece q and q + 2 share
                // their columns), the outputs as half-passes through ONE staging region -- the rows of a half-pass are read, the next half-pass's
                // arithmetic runs under the read latency (the erf arithmetic of this pass under the pre-activation rows; the next pass's
                // accumulator read + bias + rounding under the activation rows), its staging writes queue behind the reads, stores unpredicated.
                auto fast_gelu = [&](auto save_c) {
                    constexpr bool SAVE = decltype(save_c)::value;
                    const int rrow = lane >> 4, rch = lane & 15;
                    bf16_t* prep = SAVE ? reinterpret_cast<bf16_t*>(p.Out) + (size_t)(m0 + rrow) * p.ldo + n0 + rch * 8 : nullptr;
                    bf16_t* actp = reinterpret_cast<bf16_t*>(p.Out2) + (size_t)(m0 + rrow) * p.ldo2 + n0 + rch * 8;
                    const char* rd = my + rrow * ROWB + rch * 16;
                    char* wr = my + prow(0) * ROWB + pcol(0) * 2;
                    float4 bq[8];   // [2 a + (q & 1)]
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int h = 0; h < 2; ++h)
                            bq[2 * a + h] = p.bias ? *reinterpret_cast<const float4*>(p.bias + n0 + a * 32 + pcol(h)) : make_float4(0.f, 0.f, 0.f, 0.f);
                    uint2 pkp[16], pka[16];
                    // part A of a pass: accumulators + bias, rounded to bf16 (the packed words ARE the pre-activation that is stored)
                    auto part_a = [&](int b) {
#pragma unroll
                        for (int a = 0; a < 4; ++a) {
                            float blk[16];
                            v6_read_block(4 * b + a, blk);
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float4 bv = bq[2 * a + (q & 1)];
                                pkp[a * 4 + q].x = pack_bf16x2(blk[4 * q] + bv.x, blk[4 * q + 1] + bv.y);
                                pkp[a * 4 + q].y = pack_bf16x2(blk[4 * q + 2] + bv.z, blk[4 * q + 3] + bv.w);
                            }
                        }
                    };
                    // part B: the activation of the rounded pre-activation
                    auto part_b = [&]() {
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const float v0 = bf16lo_to_f32(pkp[i].x), v1 = bf16hi_to_f32(pkp[i].x), v2 = bf16lo_to_f32(pkp[i].y), v3 = bf16hi_to_f32(pkp[i].y);
                            pka[i].x = pack_bf16x2(act_val(v0, act_kind), act_val(v1, act_kind));
                            pka[i].y = pack_bf16x2(act_val(v2, act_kind), act_val(v3, act_kind));
                        }
                    };
                    auto stage = [&](const uint2 (&pk)[16]) {
#pragma unroll
                        for (int a = 0; a < 4; ++a)
#pragma unroll
                            for (int q = 0; q < 4; ++q) *reinterpret_cast<uint2*>(wr + a * 64 + qoff(q, ROWB)) = pk[a * 4 + q];
                    };
#define CX_ROWS8                                                                                                      \
    uint4 v0 = *reinterpret_cast<const uint4*>(rd + 0 * ROWB), v1 = *reinterpret_cast<const uint4*>(rd + 4 * ROWB),    \
          v2 = *reinterpret_cast<const uint4*>(rd + 8 * ROWB), v3 = *reinterpret_cast<const uint4*>(rd + 12 * ROWB),   \
          v4 = *reinterpret_cast<const uint4*>(rd + 16 * ROWB), v5 = *reinterpret_cast<const uint4*>(rd + 20 * ROWB),  \
          v6 = *reinterpret_cast<const uint4*>(rd + 24 * ROWB), v7 = *reinterpret_cast<const uint4*>(rd + 28 * ROWB);  \
    __builtin_amdgcn_sched_barrier(0);
#define CX_STORE8(o_, ld_)                                                                                            \
    gst(o_, v0); gst(o_ + (size_t)4 * (ld_), v1); gst(o_ + (size_t)8 * (ld_), v2); gst(o_ + (size_t)12 * (ld_), v3);   \
    gst(o_ + (size_t)16 * (ld_), v4); gst(o_ + (size_t)20 * (ld_), v5); gst(o_ + (size_t)24 * (ld_), v6); gst(o_ + (size_t)28 * (ld_), v7);
                    auto one_pass = [&](auto bc) {
                        constexpr int b = decltype(bc)::value;
                        if constexpr (SAVE) {   // on entry: the pre-activation rows of pass b are staged, pkp holds them
                            {
                                CX_ROWS8
                                part_b();
                                __builtin_amdgcn_sched_barrier(0);
                                stage(pka);   // (the LDS executes a wave's operations in order: behind the row reads)
                                bf16_t* o = prep + (size_t)(b * 32) * p.ldo;
                                CX_STORE8(o, p.ldo)
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            {
                                CX_ROWS8
                                if constexpr (b < 3) part_a(b + 1);
                                __builtin_amdgcn_sched_barrier(0);
                                if constexpr (b < 3) stage(pkp);
                                bf16_t* o = actp + (size_t)(b * 32) * p.ldo2;
                                CX_STORE8(o, p.ldo2)
                            }
                        } else {                // on entry: the activation rows of pass b are staged
                            CX_ROWS8
                            if constexpr (b < 3) { part_a(b + 1); part_b(); }
                            __builtin_amdgcn_sched_barrier(0);
                            if constexpr (b < 3) stage(pka);
                            bf16_t* o = actp + (size_t)(b * 32) * p.ldo2;
                            CX_STORE8(o, p.ldo2)
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    };
                    part_a(0);
                    if constexpr (SAVE) {
                        stage(pkp);
                    } else {
                        part_b();
                        stage(pka);
                    }
                    one_pass(std::integral_constant<int, 0>{});
                    one_pass(std::integral_constant<int, 1>{});
                    one_pass(std::integral_constant<int, 2>{});
                    one_pass(std::integral_constant<int, 3>{});
#undef CX_ROWS8
#undef CX_STORE8
                };
                if (m0 + 128 <= p.M && n0 + 128 <= p.N) {
                    if (p.Out) {
                        fast_gelu(std::true_type{});
                    } else {
                        fast_gelu(std::false_type{});
                    }
                } else
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    float pre[4][16];
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        v6_read_block(4 * b + a, pre[a]);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int n = n0 + a * 32 + pcol(q);
                            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                            if (p.bias && n < p.N) bv = *reinterpret_cast<const float4*>(p.bias + n);
                            pre[a][4 * q] = bf16_to_f32(f32_to_bf16(pre[a][4 * q] + bv.x));
                            pre[a][4 * q + 1] = bf16_to_f32(f32_to_bf16(pre[a][4 * q + 1] + bv.y));
                            pre[a][4 * q + 2] = bf16_to_f32(f32_to_bf16(pre[a][4 * q + 2] + bv.z));
                            pre[a][4 * q + 3] = bf16_to_f32(f32_to_bf16(pre[a][4 * q + 3] + bv.w));
                        }
                    }
#pragma unroll
                    for (int which = 0; which < 2; ++which) {  // 0: pre-activation (optional), 1: activation
                        bf16_t* outp = reinterpret_cast<bf16_t*>(which == 0 ? p.Out : p.Out2);
                        const int ldo = which == 0 ? p.ldo : p.ldo2;
                        if (!outp) continue;
#pragma unroll
                        for (int a = 0; a < 4; ++a)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                float v[4];
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    v[e] = pre[a][4 * q + e];
                                    // (the activation is a compile-time property of the instantiation: a run-time branch
                                    // here cost the erf kernel 24 %, profiles/r3_kernel_summary_clip.txt history)
                                    if (which == 1) v[e] = act_val(v[e], EPI == GEMM_EPI_QGELU ? CX_ACT_QUICK_GELU : CX_ACT_GELU);
                                }
                                uint2 pk;
                                pk.x = pack_bf16x2(v[0], v[1]);
                                pk.y = pack_bf16x2(v[2], v[3]);
                                *reinterpret_cast<uint2*>(my + prow(q) * ROWB + (a * 32 + pcol(q)) * 2) = pk;
                            }
#pragma unroll
                        for (int ps = 0; ps < 8; ++ps) {
                            const int row = ps * 4 + (lane >> 4), ch = lane & 15;
                            const int m = m0 + b * 32 + row, n = n0 + ch * 8;
                            const uint4 vv = *reinterpret_cast<const uint4*>(my + row * ROWB + ch * 16);
                            if (m < p.M && n + 8 <= p.N)
                                gst(outp + (size_t)m * ldo + n, vv);
                        }
                    }
                }
            } else {
                // SwiGLU: weight rows interleaved by 32, so the wave's 128 fused columns are [y0 | g0 | y1 | g1] (32 each)
                // = one 256-B run per row of the (M, 2I) pre-activation tensor and 64 activation columns.
                constexpr int AROWB = 144;      // 32 staged rows x 64 bf16 (+16 B pad) = 4608 B
                char* mya = my + 32 * ROWB;     // separate region: no hazard against the pre-activation staging
                // ---- fast path (interior tiles): the passes are software-pipelined as in the plain epilogue -- all row
                // reads of a pass issued together, the accumulator read + SiLU arithmetic of the next pass under their
                // latency, unpredicated 16-byte stores -- instead of 12 serial LDS-read -> wait -> predicated-store steps.
                auto fast_swiglu = [&](auto save_c) {
                    constexpr bool SAVE = decltype(save_c)::value;
                    constexpr bool SAVE_YG = SAVE && !SAVEG;   // the interleaved (y, gate) pair, (M, 2I)
                    constexpr bool SAVE_G = SAVE && SAVEG;     // the gate alone, (M, I): staged and stored like the activation
                    const int rrow = lane >> 4, rch = lane & 15;     // pre-activation rows: 16 lanes x 16 B
                    const int arow = lane >> 3, ach = lane & 7;      // activation rows: 8 lanes x 16 B
                    bf16_t* ygp = SAVE_YG ? reinterpret_cast<bf16_t*>(p.Out) + (size_t)(m0 + rrow) * p.ldo + n0 + rch * 8 : nullptr;
                    bf16_t* gp = SAVE_G ? reinterpret_cast<bf16_t*>(p.Out) + (size_t)(m0 + arow) * p.ldo + (n0 >> 1) + ach * 8 : nullptr;
                    bf16_t* actp = reinterpret_cast<bf16_t*>(p.Out2) + (size_t)(m0 + arow) * p.ldo2 + (n0 >> 1) + ach * 8;
                    const char* rdy = my + rrow * ROWB + rch * 16;
                    const char* rda = mya + arow * AROWB + ach * 16;
                    const char* rdg = my + arow * AROWB + ach * 16;   // (SAVE_G: the (y, gate) staging region holds the gate rows)
                    char* wry = my + prow(0) * ROWB + pcol(0) * 2;
                    char* wra = mya + prow(0) * AROWB + pcol(0) * 2;
                    char* wrg = my + prow(0) * AROWB + pcol(0) * 2;
                    uint2 pky[SAVE_YG ? 16 : 1], pkg[SAVE_G ? 8 : 1], pka[8];
                    auto compute_pass = [&](int b) {
#pragma unroll
                        for (int pr = 0; pr < 2; ++pr) {
                            float yb[16], gb[16];
                            v6_read_block(4 * b + 2 * pr, yb);
                            v6_read_block(4 * b + 2 * pr + 1, gb);
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                // the standalone op sees bf16 y / gate (FusedDense outputs): round pairwise, reuse the packed words
                                uint2 py, pg;
                                py.x = pack_bf16x2(yb[4 * q], yb[4 * q + 1]);
                                py.y = pack_bf16x2(yb[4 * q + 2], yb[4 * q + 3]);
                                pg.x = pack_bf16x2(gb[4 * q], gb[4 * q + 1]);
                                pg.y = pack_bf16x2(gb[4 * q + 2], gb[4 * q + 3]);
                                if constexpr (SAVE_YG) {
                                    pky[(2 * pr) * 4 + q] = py;
                                    pky[(2 * pr + 1) * 4 + q] = pg;
                                }
                                if constexpr (SAVE_G) pkg[pr * 4 + q] = pg;
                                const float yy[4] = {bf16lo_to_f32(py.x), bf16hi_to_f32(py.x), bf16lo_to_f32(py.y), bf16hi_to_f32(py.y)};
                                const float gg[4] = {bf16lo_to_f32(pg.x), bf16hi_to_f32(pg.x), bf16lo_to_f32(pg.y), bf16hi_to_f32(pg.y)};
                                float o[4];
#pragma unroll
                                for (int e = 0; e < 4; ++e) o[e] = gg[e] * yy[e] * __builtin_amdgcn_rcpf(1.f + __expf(-gg[e]));
                                pka[pr * 4 + q].x = pack_bf16x2(o[0], o[1]);
                                pka[pr * 4 + q].y = pack_bf16x2(o[2], o[3]);
                            }
                        }
                    };
                    auto stage = [&]() {
                        if constexpr (SAVE_YG) {
#pragma unroll
                            for (int a = 0; a < 4; ++a)
#pragma unroll
                                for (int q = 0; q < 4; ++q) *reinterpret_cast<uint2*>(wry + a * 64 + qoff(q, ROWB)) = pky[a * 4 + q];
                        }
                        if constexpr (SAVE_G) {
#pragma unroll
                            for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                                for (int q = 0; q < 4; ++q) *reinterpret_cast<uint2*>(wrg + pr * 64 + qoff(q, AROWB)) = pkg[pr * 4 + q];
                        }
#pragma unroll
                        for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                            for (int q = 0; q < 4; ++q) *reinterpret_cast<uint2*>(wra + pr * 64 + qoff(q, AROWB)) = pka[pr * 4 + q];
                    };
                    auto one_pass = [&](auto bc) {
                        constexpr int b = decltype(bc)::value;
                        uint4 v0 = {}, v1 = {}, v2 = {}, v3 = {}, v4 = {}, v5 = {}, v6 = {}, v7 = {};
                        if constexpr (SAVE_YG) {
                            v0 = *reinterpret_cast<const uint4*>(rdy + 0 * ROWB); v1 = *reinterpret_cast<const uint4*>(rdy + 4 * ROWB);
                            v2 = *reinterpret_cast<const uint4*>(rdy + 8 * ROWB); v3 = *reinterpret_cast<const uint4*>(rdy + 12 * ROWB);
                            v4 = *reinterpret_cast<const uint4*>(rdy + 16 * ROWB); v5 = *reinterpret_cast<const uint4*>(rdy + 20 * ROWB);
                            v6 = *reinterpret_cast<const uint4*>(rdy + 24 * ROWB); v7 = *reinterpret_cast<const uint4*>(rdy + 28 * ROWB);
                        }
                        if constexpr (SAVE_G) {
                            v0 = *reinterpret_cast<const uint4*>(rdg + 0 * AROWB); v1 = *reinterpret_cast<const uint4*>(rdg + 8 * AROWB);
                            v2 = *reinterpret_cast<const uint4*>(rdg + 16 * AROWB); v3 = *reinterpret_cast<const uint4*>(rdg + 24 * AROWB);
                        }
                        const uint4 a0 = *reinterpret_cast<const uint4*>(rda + 0 * AROWB), a1 = *reinterpret_cast<const uint4*>(rda + 8 * AROWB),
                                    a2 = *reinterpret_cast<const uint4*>(rda + 16 * AROWB), a3 = *reinterpret_cast<const uint4*>(rda + 24 * AROWB);
                        __builtin_amdgcn_sched_barrier(0);
                        CX_PH(4);
                        if constexpr (b < 3) compute_pass(b + 1);
                        __builtin_amdgcn_sched_barrier(0);
                        CX_PH(5);
                        if constexpr (b < 3) stage();  // the LDS executes a wave's operations in order: these follow the row reads
                        __builtin_amdgcn_sched_barrier(0);
                        CX_PH(6);
                        if constexpr (SAVE_YG) {
                            bf16_t* o = ygp + (size_t)(b * 32) * p.ldo;
                            gst(o, v0);
                            gst(o + (size_t)4 * p.ldo, v1);
                            gst(o + (size_t)8 * p.ldo, v2);
                            gst(o + (size_t)12 * p.ldo, v3);
                            gst(o + (size_t)16 * p.ldo, v4);
                            gst(o + (size_t)20 * p.ldo, v5);
                            gst(o + (size_t)24 * p.ldo, v6);
                            gst(o + (size_t)28 * p.ldo, v7);
                        }
                        if constexpr (SAVE_G) {
                            bf16_t* o = gp + (size_t)(b * 32) * p.ldo;
                            gst(o, v0);
                            gst(o + (size_t)8 * p.ldo, v1);
                            gst(o + (size_t)16 * p.ldo, v2);
                            gst(o + (size_t)24 * p.ldo, v3);
                        }
                        bf16_t* oa = actp + (size_t)(b * 32) * p.ldo2;
                        gst(oa, a0);
                        gst(oa + (size_t)8 * p.ldo2, a1);
                        gst(oa + (size_t)16 * p.ldo2, a2);
                        gst(oa + (size_t)24 * p.ldo2, a3);
                        __builtin_amdgcn_sched_barrier(0);
                        CX_PH(7);
                    };
                    compute_pass(0);
                    stage();
                    __builtin_amdgcn_sched_barrier(0);
                    CX_PH(3);
                    one_pass(std::integral_constant<int, 0>{});
                    one_pass(std::integral_constant<int, 1>{});
                    one_pass(std::integral_constant<int, 2>{});
                    one_pass(std::integral_constant<int, 3>{});
                };
                if (m0 + 128 <= p.M && n0 + 128 <= p.N) {
                    if (p.Out) {
                        fast_swiglu(std::true_type{});
                    } else {
                        fast_swiglu(std::false_type{});
                    }
                } else
#pragma unroll
                for (int b = 0; b < 4; ++b) {
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {  // (y, gate) block pair -> 32 activation columns
                        float yb[16], gb[16];
                        v6_read_block(4 * b + 2 * pr, yb);
                        v6_read_block(4 * b + 2 * pr + 1, gb);
                        if (p.Out) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                uint2 pk;
                                if constexpr (SAVEG) {   // the gate alone, staged in the activation's row layout
                                    pk.x = pack_bf16x2(gb[4 * q], gb[4 * q + 1]);
                                    pk.y = pack_bf16x2(gb[4 * q + 2], gb[4 * q + 3]);
                                    *reinterpret_cast<uint2*>(my + prow(q) * AROWB + (pr * 32 + pcol(q)) * 2) = pk;
                                } else {
                                    pk.x = pack_bf16x2(yb[4 * q], yb[4 * q + 1]);
                                    pk.y = pack_bf16x2(yb[4 * q + 2], yb[4 * q + 3]);
                                    *reinterpret_cast<uint2*>(my + prow(q) * ROWB + (2 * pr * 32 + pcol(q)) * 2) = pk;
                                    pk.x = pack_bf16x2(gb[4 * q], gb[4 * q + 1]);
                                    pk.y = pack_bf16x2(gb[4 * q + 2], gb[4 * q + 3]);
                                    *reinterpret_cast<uint2*>(my + prow(q) * ROWB + ((2 * pr + 1) * 32 + pcol(q)) * 2) = pk;
                                }
                            }
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            float o[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {  // the standalone op sees bf16 y / gate (FusedDense outputs)
                                const float yy = bf16_to_f32(f32_to_bf16(yb[4 * q + e]));
                                const float gg = bf16_to_f32(f32_to_bf16(gb[4 * q + e]));
                                o[e] = gg * yy * __builtin_amdgcn_rcpf(1.f + __expf(-gg));  // 1-ulp rcp: far below bf16
                            }
                            uint2 pk;
                            pk.x = pack_bf16x2(o[0], o[1]); pk.y = pack_bf16x2(o[2], o[3]);
                            *reinterpret_cast<uint2*>(mya + prow(q) * AROWB + (pr * 32 + pcol(q)) * 2) = pk;
                        }
                    }
                    if (p.Out) {
                        if constexpr (SAVEG) {
#pragma unroll
                            for (int ps = 0; ps < 4; ++ps) {
                                const int row = ps * 8 + (lane >> 3), ch = lane & 7;
                                const int m = m0 + b * 32 + row;
                                const int col = (n0 >> 1) + ch * 8;
                                const uint4 vv = *reinterpret_cast<const uint4*>(my + row * AROWB + ch * 16);
                                if (m < p.M && 2 * col < p.N)
                                    gst(reinterpret_cast<bf16_t*>(p.Out) + (size_t)m * p.ldo + col, vv);
                            }
                        } else {
#pragma unroll
                            for (int ps = 0; ps < 8; ++ps) {
                                const int row = ps * 4 + (lane >> 4), ch = lane & 15;
                                const int m = m0 + b * 32 + row, n = n0 + ch * 8;
                                const uint4 vv = *reinterpret_cast<const uint4*>(my + row * ROWB + ch * 16);
                                if (m < p.M && n < p.N)
                                    gst(reinterpret_cast<bf16_t*>(p.Out) + (size_t)m * p.ldo + n, vv);
                            }
                        }
                    }
#pragma unroll
                    for (int ps = 0; ps < 4; ++ps) {
                        const int row = ps * 8 + (lane >> 3), ch = lane & 7;
                        const int m = m0 + b * 32 + row;
                        const int col = (n0 >> 1) + ch * 8;
                        const uint4 vv = *reinterpret_cast<const uint4*>(mya + row * AROWB + ch * 16);
                        if (m < p.M && 2 * col < p.N)
                            gst(reinterpret_cast<bf16_t*>(p.Out2) + (size_t)m * p.ldo2 + col, vv);
                    }
                }
            }
            cp_tile = tile_of(++cp_round);
            // Register relief for the two epilogues that need it (the plain / residual one sat at 256 VGPRs and the compiler
            // parked live values in a0..a3, i.e. INSIDE accumulator block 0, which it cannot see; build.py audits the
            // generated code for exactly that).  The first fragments of the next tile (read into F0 during the last k-step
            // above) are fetched again here -- their slots are untouched by the staging -- so F0 is dead across the epilogue
            // (~150 cycles of exposed LDS latency per tile against 32 registers); the SwiGLU backward additionally rebuilds
            // its 16 DMA cursor offsets from (round, K-tile).
            {
                asm volatile("" : "=v"(F0.w[0]), "=v"(F0.w[1]), "=v"(F0.w[2]), "=v"(F0.w[3]), "=v"(F0.w[4]), "=v"(F0.w[5]), "=v"(F0.w[6]), "=v"(F0.w[7]));
                asm volatile("" : "=v"(F0.x[0]), "=v"(F0.x[1]), "=v"(F0.x[2]), "=v"(F0.x[3]), "=v"(F0.x[4]), "=v"(F0.x[5]), "=v"(F0.x[6]), "=v"(F0.x[7]));
                if constexpr (IS_SWIGLU_BWD) {
                    asm volatile("" : "=v"(xoff[0]), "=v"(xoff[1]), "=v"(xoff[2]), "=v"(xoff[3]), "=v"(xoff[4]), "=v"(xoff[5]), "=v"(xoff[6]), "=v"(xoff[7]));
                    asm volatile("" : "=v"(woff[0]), "=v"(woff[1]), "=v"(woff[2]), "=v"(woff[3]), "=v"(woff[4]), "=v"(woff[5]), "=v"(woff[6]), "=v"(woff[7]));
                    x_clamped = w_clamped = true;  // (forces the re-computation)
                    x_setup(lx_live ? tile_of(lx_round) : first_tile);
                    xbase += lx_kt * BK6;
                    w_setup(lw_live ? tile_of(lw_round) : first_tile);
                    wbase += lw_kt * BK6;
                }
                if (cp_tile >= 0) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) read_one(F0, dsm + xs_slot * XS6, dsm + (3 + ws_slot) * XS6, 0, i);
                }
            }
            // the staging areas are the DMA targets of the next iteration: nobody may still be reading them
            __builtin_amdgcn_s_barrier();
            CX_PH(8);
        }
    }
#undef CX_PH
#undef CX_DMA_M0
#undef CX_DMA_LD
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the cursors' dummy DMAs must land before the LDS is handed on
    if constexpr (DBG != 0) {
        if (p.trace && tid == 0) {
            p.trace[16 * blockIdx.x] = (long long)__builtin_amdgcn_s_memtime() - t_begin;
            p.trace[16 * blockIdx.x + 1] = n_ktiles;
            if constexpr ((DBG & 128) != 0) {
#pragma unroll
                for (int i = 2; i < 9; ++i) p.trace[16 * blockIdx.x + i] = ph[i];
                p.trace[16 * blockIdx.x + 9] = ph[0];
                p.trace[16 * blockIdx.x + 10] = ph[1];
                p.trace[16 * blockIdx.x + 11] = kt0_cyc;   // the first K-tile of every tile (part of ph[2])
            }
        }
    }
}

template <int EPI, int DBG = 0>
hipError_t launch6(const GemmParams& p, hipStream_t stream) {
    static CxLdsOptIn lds;
    if (!lds.ensure(reinterpret_cast<const void*>(&gemm_bf16_v6_kernel<EPI, DBG>), LDS6)) return hipErrorInvalidValue;
    const int ntiles = p.tiles_m * p.tiles_n;
    const int grid = ntiles < 256 ? (ntiles + 7) / 8 * 8 : 256;
    hipLaunchKernelGGL((gemm_bf16_v6_kernel<EPI, DBG>), dim3(grid), dim3(256), LDS6, stream, p);
    return hipGetLastError();
}

// =====================================================================================================================
// TN form (wgrad): G[o][i] = sum_t dY[t][o] * A[t][i], both operands in their natural (tokens, features) layout, K =
// tokens.  Same one-wave-per-SIMD structure; one (output tile, K slice) unit per workgroup (a unit is hundreds of
// K-tiles long, so nothing is gained by walking units persistently), the same 3 X + 2 W slot ring, fragments through
// ds_read_b64_tr_b16 (transposing read; conflict-free 16-B-chunk XOR, see gemm_bf16_v5.hip tn_frag5), fp32 split-K
// partial slabs stored straight from the AGPRs (global_store_dwordx4 a[..]) -- no VGPR copy, no LDS staging.
// =====================================================================================================================
typedef __attribute__((address_space(3))) bf16x4_t* lds_b4_ptr6;
constexpr int TNROW6 = 512;      // bytes per token row of a [64 t][256 f] tile

// 16-byte chunk c of token row t of a tile sits at chunk position c ^ tn_swz(t).  A 16-lane group of a transposing read covers
// 4 token rows x 32 bytes and the two groups of a 32-lane half sit 8 token rows apart: bits 2..3 of the XOR separate the 4 rows
// (512 B apart = same banks), bit 1 the two groups.
CX_DEVICE int tn_swz(int t) { return ((t & 3) << 2) ^ (((t >> 3) & 1) << 1); }

__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_bf16_v6tn_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g4 = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int nwg = p.tiles_m * p.tiles_n * p.split_k;
    // An XCD runs a contiguous range of logical ids (xcd_remap), normally one K slice's tiles: the index that varies FASTEST is
    // the one with fewer tiles, so that the workgroups sharing a panel of the BIGGER operand are neighbours and land on one
    // XCD's L2.  fc2's wgrad (3 x 12 tiles, the activation operand four times the size of dY) walked tn fastest and put the
    // three sharers of an activation panel 12 ids apart -- across an XCD boundary for a good part of them: 3.60 GB of L2 misses
    // per launch against 2.01 GB algorithmic (profiles/r4_wgrad_traffic_by_shape.txt).
    int lid = xcd_remap(blockIdx.x, nwg);
    int tm, tn;
    if (p.tiles_m < p.tiles_n) {
        tm = lid % p.tiles_m;
        lid /= p.tiles_m;
        tn = lid % p.tiles_n;
        lid /= p.tiles_n;
    } else {
        tn = lid % p.tiles_n;
        lid /= p.tiles_n;
        tm = lid % p.tiles_m;
        lid /= p.tiles_m;
    }
    const int sk = lid;
    const int m0 = tm * BM6, n0 = tn * BN6;
    const int nk_total = p.K / BK6;
    const int kt_begin = (int)(((long)nk_total * sk) / p.split_k);
    const int kt_end = (int)(((long)nk_total * (sk + 1)) / p.split_k);
    const int nk = kt_end - kt_begin;
    if (nk <= 0) return;  // (split_k <= K tiles is enforced by the launcher)

    // DMA: instruction q = j*4 + wave covers token rows 2q, 2q+1 of a [64 t][256 f] tile (512 B each).  Ring as in the
    // NT kernel: three X slots (dY runs two K-tiles ahead), two W slots (A one ahead); cursors as in the NT kernel (SGPR
    // base advanced per K-tile, constant per-lane offsets, M0 write | MFMA | load).  Past the last K-tile of the split a
    // cursor stops advancing and re-loads that tile into a slot nobody reads: no "is there another tile" branch per DMA.
    uint32_t xo[8], wo[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int t = (j * 4 + wave) * 2 + (lane >> 5);
        const int c = (lane & 31) ^ tn_swz(t);
        xo[j] = (uint32_t)t * (uint32_t)p.ldx * 2u + c * 16;
        wo[j] = (uint32_t)t * (uint32_t)p.ldw * 2u + c * 16;
    }
    const bf16_t* xb = p.X + (size_t)kt_begin * BK6 * p.ldx + m0;
    const bf16_t* wb = p.W + (size_t)kt_begin * BK6 * p.ldw + n0;
    const size_t xstep = (size_t)BK6 * p.ldx, wstep = (size_t)BK6 * p.ldw;
    int lx_slot = 0, lw_slot = 0;
    // LDS-DMA through inline asm (see the NT kernel): with the builtin the compiler puts s_waitcnt vmcnt(0) in front of
    // the first ds_read_b64_tr_b16 of every iteration (the intrinsic carries no alias information), serialising the ring.
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_ptr)dsm;
    const uint32_t lds_wave = lds0 + wave * 1024;
    uint32_t x_m0 = lds_wave, w_m0 = lds_wave + 3 * XS6;
    auto issue_x_tile = [&]() {
        v6_dma_full<0>(x_m0, xo[0], xb); v6_dma_full<1>(x_m0, xo[1], xb); v6_dma_full<2>(x_m0, xo[2], xb); v6_dma_full<3>(x_m0, xo[3], xb);
        v6_dma_full<4>(x_m0, xo[4], xb); v6_dma_full<5>(x_m0, xo[5], xb); v6_dma_full<6>(x_m0, xo[6], xb); v6_dma_full<7>(x_m0, xo[7], xb);
    };
    auto issue_w_tile = [&]() {
        v6_dma_full<0>(w_m0, wo[0], wb); v6_dma_full<1>(w_m0, wo[1], wb); v6_dma_full<2>(w_m0, wo[2], wb); v6_dma_full<3>(w_m0, wo[3], wb);
        v6_dma_full<4>(w_m0, wo[4], wb); v6_dma_full<5>(w_m0, wo[5], wb); v6_dma_full<6>(w_m0, wo[6], wb); v6_dma_full<7>(w_m0, wo[7], wb);
    };
    // the cursor moves on to K-tile `next` of this split if it exists (else it stays: dummy re-load) and to the next slot
    auto x_advance = [&](int next) {
        xb += next < nk ? xstep : 0;
        lx_slot = lx_slot == 2 ? 0 : lx_slot + 1;
        x_m0 = lds_wave + lx_slot * XS6;
    };
    auto w_advance = [&](int next) {
        wb += next < nk ? wstep : 0;
        lw_slot ^= 1;
        w_m0 = lds_wave + (3 + lw_slot) * XS6;
    };

    // (An L2 prefetch of the K-tiles 3-4 ahead -- one dword touch per 128-B line -- paid in round 1; with the two-instruction
    // DMA issue of round 2 it costs 5-16 %: removed.  profiles/r2_vendor_blas_calibration.txt has the before / after.)
    // Fragment addressing: fragment i = features 16 i .. 16 i + 15 of the wave's 128, k-step kk = tokens 32 kk .. 32 kk + 31.  Lane
    // (p = l15, g = g4) must end up with feature p, tokens 8 g .. 8 g + 7: two transposing reads of 4 tokens each, in which the lane
    // ADDRESSES token tl = 8 g + (p >> 2) (+ 4 for the second) and features fl = 4 (p & 3) .. + 3 -- the 16 lanes of a group cover
    // 4 tokens x 16 features and the read hands every lane its feature's 4 tokens.  With one wave per SIMD the instruction stream is
    // issue-bound, so a fragment costs one VALU add + two reads: everything lane-dependent is hoisted into one VGPR per fragment.
    //   byte offset inside a tile = tt * 512 + ((f >> 3) ^ tn_swz(tt)) * 16 + (f & 4) * 2,  tt = 32 kk + 4 half + tl, tn_swz(tt) == tn_swz(tl)
    //   lane part = tl * 512 + (((f0 + fl) >> 3) ^ tn_swz(tl)) * 16 + (fl & 4) * 2;   immediate = kk * 16384 + half * 2048
    Frags6 F0, F1;
    uint32_t woff[8], xoff[8];
    {
        const int tl = 8 * g4 + (l15 >> 2), fl = 4 * (l15 & 3);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int fw = wn * 128 + i * 16 + fl, fx = wm * 128 + i * 16 + fl;
            woff[i] = tl * TNROW6 + (((fw >> 3) ^ tn_swz(tl)) << 4) + (fw & 4) * 2;
            xoff[i] = tl * TNROW6 + (((fx >> 3) ^ tn_swz(tl)) << 4) + (fx & 4) * 2;
        }
    }
    auto tr_pair = [&](const char* base, int kk) -> bf16x8_t {
        union { bf16x4_t h[2]; bf16x8_t v; } u;
        u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4_ptr6)(base + kk * 16384));
        u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4_ptr6)(base + kk * 16384 + 2048));
        return u.v;
    };
    auto read_one = [&](Frags6& f, int xslot, int wslot, int kk, int i) {
        if (i < 8)
            f.w[i] = tr_pair(dsm + (3 + wslot) * XS6 + woff[i], kk);
        else
            f.x[i - 8] = tr_pair(dsm + xslot * XS6 + xoff[i - 8], kk);
    };
    auto mma1 = [&](const Frags6& f, int i) { v6_mfma(i, f.w[i & 7], f.x[i >> 3]); };
    auto mma1z = [&](const Frags6& f, int i) { v6_mfma_z(i, f.w[i & 7], f.x[i >> 3]); };
    // (k-step pieces: CX_SEG of the NT kernel; a fragment read is two transposing reads here)
#define CX_DMA_M0(kind, J)                                  \
    do {                                                    \
        if constexpr ((kind) == 1) v6_dma_m0<(J)>(w_m0);     \
        if constexpr ((kind) == 2) v6_dma_m0<(J)>(x_m0);     \
    } while (0)
#define CX_DMA_LD(kind, J)                                  \
    do {                                                    \
        if constexpr ((kind) == 1) v6_dma_ld(wo[(J)], wb);   \
        if constexpr ((kind) == 2) v6_dma_ld(xo[(J)], xb);   \
    } while (0)

    issue_x_tile();  // X of K-tile 0
    x_advance(1);
    issue_w_tile();  // W of K-tile 0
    w_advance(1);
    issue_x_tile();  // X of K-tile 1 (or K-tile 0 again)
    x_advance(2);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 16; ++i) read_one(F0, 0, 0, 0, i);

    int xs_slot = 0, ws_slot = 0;
    // K-tile body (see the NT kernel): DMA of this iteration = W of K-tile t+1 (k-step 0) and X of K-tile t+2 (k-step 1, half
    // before and half after the barrier); `first` selects the C = 0 MFMA form.
    auto kt_body = [&](auto first, int t) {
        constexpr int RS_ = CX_V6_RD_SPREAD, DS_ = CX_V6_DMA_SPREAD;
        const int nxs_slot = xs_slot == 2 ? 0 : xs_slot + 1, nws_slot = ws_slot ^ 1;
        if constexpr (decltype(first)::value) {
            CX_SEG(mma1z, F0, 0, 64, F1, xs_slot, ws_slot, 1, RS_ == 2 ? 0 : 1, RS_, 1, RS_ == 2 ? 1 : 17, DS_, 0, 8);
        } else {
            CX_SEG(mma1, F0, 0, 64, F1, xs_slot, ws_slot, 1, RS_ == 2 ? 0 : 1, RS_, 1, RS_ == 2 ? 1 : 17, DS_, 0, 8);
        }
        w_advance(t + 2);
        CX_SEG(mma1, F1, 0, 32, F0, xs_slot, ws_slot, 0, 0, 0, 2, 0, RS_ == 2 ? 8 : DS_, 0, 4);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's reads of the current slots are complete ...
        // ... and so are its DMA writes of K-tile t+1: everything but the 4 X instructions just issued
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // (after the last K-tile these reads fetch garbage from a landed slot; F0 is not used again)
        CX_SEG(mma1, F1, 32, 64, F0, nxs_slot, nws_slot, 0, RS_ == 2 ? 32 : 33, RS_, 2, RS_ == 2 ? 33 : 49, RS_ == 2 ? 8 : DS_, 4, 4);
        x_advance(t + 3);
        xs_slot = nxs_slot;
        ws_slot = nws_slot;
    };
    kt_body(std::true_type{}, 0);
#pragma unroll 1
    for (int t = 1; t < nk; ++t) kt_body(std::false_type{}, t);
#undef CX_SEG
#undef CX_DMA_M0
#undef CX_DMA_LD

    // ---- epilogue: fp32 partial slab of this K slice, straight from the AGPRs.  Region (a, b), piece q: row m0 + wm*128 + b*32
    // + 16 (q >> 1) + l15, columns n0 + wn*128 + a*32 + 16 (q & 1) + 4 g4 (+0..3): a store writes 16 rows x 64 contiguous bytes
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");  // MFMA -> VMEM read of the accumulators
    float* part = reinterpret_cast<float*>(p.Out) + (size_t)sk * p.M * p.ldo;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        float* rowp = part + (size_t)(m0 + wm * 128 + b * 32 + l15) * p.ldo + n0 + wn * 128 + 4 * g4;
        float* rowq = rowp + (size_t)16 * p.ldo;
#pragma unroll
        for (int a = 0; a < 4; ++a) v6_store_block_f32(4 * b + a, rowp + a * 32, rowq + a * 32);
    }
}

// Routing between this kernel and the two-workgroups-per-CU kernel (gemm_bf16_v7.hip).  CX_V7_POLICY (build-time, for whole-
// step A/B builds through CX_EXTRA_HIPCC_FLAGS): 0 = never, 1 = the shipped policy below, 2 = every launch v7 covers.
#ifndef CX_V7_POLICY
#define CX_V7_POLICY 1
#endif
#ifndef CX_PRODUCT
int g_v7_mode = -1;            // cx_gemm_v7_set_mode: -1 = CX_V7_POLICY, 0 = never, 1 = every launch v7 covers
#else
constexpr int g_v7_mode = -1;
#endif
bool v7_takes(const GemmParams& p, int epi) {
    const int mode = g_v7_mode >= 0 ? (g_v7_mode ? 2 : 0) : CX_V7_POLICY;
    if (mode == 0 || !cx_gemm_v7_covers(p, epi)) return false;
    if (mode == 2) return true;
    // shipped policy (round 4 A/B, profiles/r4_gemm_v7_ab.txt, r4_gemm_v7_small.txt): v7's half-size tiles and two resident
    // workgroups per CU win 7 - 37 % while the launch is at most ~1.5 rounds of v6's 256 x 256 tiles on 256 CUs (literal
    // chunk_size 64: 8192 token rows; cfg 1: 2048) -- there a launch lasts as long as its slowest tile and half the CUs idle --
    // and lose 1 - 36 % on the metric-sized launches (1.5 x the operand traffic per FLOP), fc1 + SwiGLU at every size measured.
    if (epi == GEMM_EPI_SWIGLU_G) return false;
    const long tiles256 = (long)((p.M + BM6 - 1) / BM6) * ((p.N + BN6 - 1) / BN6);
    return tiles256 <= 400;
}

#ifndef CX_PRODUCT
int g_v6_dbg = 0;              // ablation mask (cx_gemm_v6_ablate)
long long* g_v6_trace = nullptr;  // ablation builds only: 2 x int64 per workgroup {cycles, K-tiles}
int g_v6_force_gn = 0;  // experiments: 1, 2, 4 or 8 forces the N-group count; 0 = heuristic
#else
constexpr int g_v6_force_gn = 0;
#endif

// N-groups of the XCD grid: the estimated L2-miss traffic is gn * |X| (every X panel is fetched by the gn XCDs of its
// grid row) + (8 / gn) * |W| when an XCD's W slice (tiles_n / gn row-blocks of 256 x K) can stay L2-resident, and
// rounds * slice * 8 otherwise; even splits only (no load imbalance between XCDs).
int cx_gemm_v6_groups(int tiles_m, int tiles_n, int K) {
    if (g_v6_force_gn) return g_v6_force_gn;
    const double xb = (double)tiles_m * 256 * K * 2, wb = (double)tiles_n * 256 * K * 2;
    int best = 1;
    double best_cost = 0;
    for (int gn = 1; gn <= 8; gn *= 2) {
        const int gm = 8 / gn;
        if ((tiles_n % gn) != 0 || (tiles_m % gm) != 0) continue;
        const double slice = wb / gn;
        const double rounds = (double)tiles_m / gm * (tiles_n / gn) / 32.0;
        const double wcost = slice <= 2.5 * 1048576 ? gm * wb : (rounds < 1 ? 1 : rounds) * slice * 8;
        const double cost = gn * xb + wcost;
        if (gn == 1 || cost < best_cost) { best = gn; best_cost = cost; }
    }
    return best;
}

}  // namespace

#ifndef CX_PRODUCT
void cx_gemm_v6_set_trace(long long* buf) { g_v6_trace = buf; }
void cx_gemm_v6_set_ablate(int mask) { g_v6_dbg = mask; }
void cx_gemm_v6_force_groups(int gn) { g_v6_force_gn = (gn == 1 || gn == 2 || gn == 4 || gn == 8) ? gn : 0; }
void cx_gemm_v7_set_mode(int mode) { g_v7_mode = mode < 0 ? -1 : (mode ? 1 : 0); }
#endif

// TN wgrad form: p.X = dY (T, M), p.W = A (T, N), p.K = tokens, p.Out = fp32 partial slabs [split_k][M][ldo];
// M % 256 == 0, N % 256 == 0, K % 64 == 0, 1 <= split_k <= K / 64 (checked by the caller).
hipError_t cx_launch_gemm_v6_tn(GemmParams p, hipStream_t stream) {
    static CxLdsOptIn lds;
    if (!lds.ensure(reinterpret_cast<const void*>(&gemm_bf16_v6tn_kernel), LDS6)) return hipErrorInvalidValue;
    p.tiles_m = p.M / BM6;
    p.tiles_n = p.N / BN6;
    hipLaunchKernelGGL(gemm_bf16_v6tn_kernel, dim3(p.tiles_m * p.tiles_n * p.split_k), dim3(256), LDS6, stream, p);
    return hipGetLastError();
}

namespace {
// 64 columns x 4 block groups per workgroup; each thread sums every 4th block's partial of its column, the groups fold through LDS
__global__ __launch_bounds__(256) void colsum_part_reduce_kernel(const float* __restrict__ part, float* dst, int nblocks, int N) {
    __shared__ float red[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
    float sacc = 0.f;
    if (col < N) {
#pragma unroll 4
        for (int b = grp; b < nblocks; b += 4) sacc += part[(size_t)b * N + col];
    }
    red[grp][threadIdx.x & 63] = sacc;
    __syncthreads();
    if (grp == 0 && col < N) dst[col] += (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
}  // namespace
hipError_t cx_launch_colsum_part_reduce(const float* part, float* dst, int nblocks, int N, hipStream_t stream) {
    hipLaunchKernelGGL(colsum_part_reduce_kernel, dim3((N + 63) / 64), dim3(256), 0, stream, part, dst, nblocks, N);
    return hipGetLastError();
}

// NT forms with bf16 output (plain / bias / alpha, or fused SwiGLU), K % 64 == 0, N % 8 == 0, split_k == 1.
hipError_t cx_launch_gemm_v6(GemmParams p, int epi, hipStream_t stream) {
    if (v7_takes(p, epi)) return cx_launch_gemm_v7(p, epi, g_v6_force_gn, stream);
    p.tiles_m = (p.M + BM6 - 1) / BM6;
    p.tiles_n = (p.N + BN6 - 1) / BN6;
    if (p.tiles_n > 256) return hipErrorInvalidValue;  // tile ids are (tm << 8) | tn: N <= 65536
    p.sup_n = cx_gemm_v6_groups(p.tiles_m, p.tiles_n, p.K);
#ifndef CX_PRODUCT
    if (epi == GEMM_EPI_NONE && g_v6_dbg) {  // ablation builds (scripts/gemm_ablate.py)
        p.trace = g_v6_trace;
        switch (g_v6_dbg) {
            case 1: return launch6<GEMM_EPI_NONE, 1>(p, stream);
            case 2: return launch6<GEMM_EPI_NONE, 2>(p, stream);
            case 4: return launch6<GEMM_EPI_NONE, 4>(p, stream);
            case 16: return launch6<GEMM_EPI_NONE, 16>(p, stream);
            case 32: return launch6<GEMM_EPI_NONE, 32>(p, stream);
            case 33: return launch6<GEMM_EPI_NONE, 33>(p, stream);
            case 35: return launch6<GEMM_EPI_NONE, 35>(p, stream);
            case 39: return launch6<GEMM_EPI_NONE, 39>(p, stream);
            case 55: return launch6<GEMM_EPI_NONE, 55>(p, stream);
            case 59: return launch6<GEMM_EPI_NONE, 59>(p, stream);  // everything but the fragment reads
            case 64: return launch6<GEMM_EPI_NONE, 64>(p, stream);    // epilogue without its global stores
            case 128: return launch6<GEMM_EPI_NONE, 128>(p, stream);  // trace only
            default: break;
        }
    }
#endif
#ifndef CX_PRODUCT
    if (g_v6_dbg == 128 && (epi == GEMM_EPI_SWIGLU_G || epi == GEMM_EPI_SWIGLU || epi == GEMM_EPI_SWIGLU_BWD_AG)) {   // phase traces of the fused SwiGLU epilogues
        p.trace = g_v6_trace;
        return epi == GEMM_EPI_SWIGLU_G ? launch6<GEMM_EPI_SWIGLU_G, 128>(p, stream)
               : epi == GEMM_EPI_SWIGLU ? launch6<GEMM_EPI_SWIGLU, 128>(p, stream) : launch6<GEMM_EPI_SWIGLU_BWD_AG, 128>(p, stream);
    }
#endif
    return epi == GEMM_EPI_SWIGLU ? launch6<GEMM_EPI_SWIGLU>(p, stream)
           : epi == GEMM_EPI_SWIGLU_G ? launch6<GEMM_EPI_SWIGLU_G>(p, stream)
           : epi == GEMM_EPI_SWIGLU_BWD_AG ? launch6<GEMM_EPI_SWIGLU_BWD_AG>(p, stream)
           : epi == GEMM_EPI_GELU ? launch6<GEMM_EPI_GELU>(p, stream)
           : epi == GEMM_EPI_QGELU ? launch6<GEMM_EPI_QGELU>(p, stream)
           : epi == GEMM_EPI_SWIGLU_BWD ? launch6<GEMM_EPI_SWIGLU_BWD>(p, stream)
           : epi == GEMM_EPI_ACT_BWD ? (p.act == CX_ACT_QUICK_GELU ? launch6<GEMM_EPI_QACT_BWD>(p, stream) : launch6<GEMM_EPI_ACT_BWD>(p, stream))
                                  : launch6<GEMM_EPI_NONE>(p, stream);
}
