// gemm_params.h -- launch descriptor shared by the bf16 GEMM generations (gemm_bf16.hip, gemm_bf16_v3.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

enum { GEMM_OUT_BF16 = 0, GEMM_OUT_F32 = 1, GEMM_OUT_F32_ATOMIC = 2, GEMM_OUT_F32_PARTIAL = 3 };
enum { GEMM_EPI_NONE = 0, GEMM_EPI_SWIGLU = 1, GEMM_EPI_GELU = 2, GEMM_EPI_SWIGLU_BWD = 3,
       GEMM_EPI_QGELU = 4,   // GELU / QGELU: Out2 = act(acc + bias), Out (optional) = acc + bias; act = erf GELU / quick_gelu
       GEMM_EPI_SWIGLU_G = 5,        // SWIGLU whose optional save is the GATE alone: Out = G (M, N/2), Out2 = Act (M, N/2)
       GEMM_EPI_SWIGLU_BWD_AG = 6,   // SWIGLU_BWD from (Act, G): Out2 = Act, In3 = G (INPUTS, (M, N) each, ld = ldo2), Out = dYG
       GEMM_EPI_QACT_BWD = 8,        // ACT_BWD with quick_gelu (a compile-time twin: a run-time branch inside the derivative cost a register too many)
       GEMM_EPI_ACT_BWD = 7 };       // fc2 dgrad of the plain MLP + GELU / quick_gelu backward: Out = bf16(bf16(acc) * act'(Pre)), Pre = Out2 (INPUT,
                                     // (M, N) bf16, ld = ldo2); colsum_part (optional): fp32 [ceil(M / 128)][N] column sums of the bf16 Out per 128-row block
// SWIGLU_BWD: acc = d(act); Out2 = YG (INPUT, (M, 2N) interleaved by 32), Out = dYG (same layout)

struct GemmParams {
    const uint16_t* X;
    const uint16_t* W;
    void* Out;
    const float* bias;  // fp32[N] or nullptr
    int M, N, K;
    int ldx, ldw, ldo;
    int tiles_m, tiles_n, split_k;
    float alpha;
    int dbg;    // experiments only: bit0 = no DMA in the main loop, bit1 = no LDS reads / MFMA
    int act;    // cx_gemm_bf16_bias_act: CX_ACT_GELU (erf, default 0) or CX_ACT_QUICK_GELU -> GEMM_EPI_GELU / GEMM_EPI_QGELU
    void* Out2; // SwiGLU epilogue: activation output (M, N/2) bf16
    int ldo2;
    const uint16_t* In3;  // SWIGLU_BWD_AG: the saved gate (M, N) bf16, leading dimension ldo2
    float* colsum_part;   // ACT_BWD: per-128-row-block column sums of Out (the fc1 bias gradient's partials), or nullptr
    int sup_m, sup_n;  // v2: L2 super-tile (sup_m x sup_n tiles walked together); 0 = plain row-major order
    long long* trace;  // v5p only (set by its launcher): per-workgroup phase cycle counters, or nullptr
};

hipError_t cx_launch_gemm_v5(GemmParams p, int form, int out_mode, int epi, hipStream_t stream);
hipError_t cx_launch_gemm_v6(GemmParams p, int epi, hipStream_t stream);
void cx_gemm_v6_force_groups(int gn);
void cx_gemm_v6_set_trace(long long* buf);
void cx_gemm_v6_set_ablate(int mask);
hipError_t cx_launch_gemm_v6_tn(GemmParams p, hipStream_t stream);
// dst[c] += sum_b part[b][c] (fixed order: deterministic), part: fp32 [nblocks][N] -- the ACT_BWD epilogue's bias-gradient partials
hipError_t cx_launch_colsum_part_reduce(const float* part, float* dst, int nblocks, int N, hipStream_t stream);
// gemm_bf16_v7.hip: two resident workgroups per CU, 256x128x64 tiles (plain / residual, SwiGLU with gate save, SwiGLU backward
// from (act, gate)); cx_launch_gemm_v6 routes to it (policy there).  force_gn: 0 = heuristic XCD grid.
bool cx_gemm_v7_covers(const GemmParams& p, int epi);
hipError_t cx_launch_gemm_v7(GemmParams p, int epi, int force_gn, hipStream_t stream);
void cx_gemm_v7_set_trace(long long* buf);   // dev: int64[grid][4] = {start, end, HW_ID, XCC_ID} per workgroup
int cx_gemm_v7_occupancy_query(void);
void cx_gemm_v7_set_ablate(int mask);
void cx_gemm_v7_set_flags(int f);   // dev: bit 0 = K loop at s_setprio 1
void cx_gemm_v7_set_period(int cycles);   // dev: tile period for the start stagger (-1 estimate, 0 off)
void cx_gemm_v7_set_mode(int mode);   // dev library only: -1 = shipped policy, 0 = never, 1 = every launch v7 covers
void cx_gemm_v5_set_persistent(bool on);
void cx_gemm_v5_set_use_v6(bool on);
bool cx_gemm_v5_get_use_v6(void);
void cx_gemm_v5_set_trace(long long* buf);  // 4 counters per workgroup: wait, compute, epilogue cycles, iterations
