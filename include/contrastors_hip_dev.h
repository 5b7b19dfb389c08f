/* contrastors_hip_dev.h -- DEVELOPMENT-ONLY entry points of libcontrastors_hip_dev.so (gfx950).
 *
 * The product library (libcontrastors_hip.so, include/contrastors_hip.h) carries ONE kernel per op and no mutable
 * switches.  This library is the same code built WITHOUT -DCX_PRODUCT: it additionally contains the earlier GEMM
 * generations (gemm_bf16.hip v1/v2, gemm_bf16_v3.hip, gemm_bf16_v4.hip, the 8-wave persistent v5p), the A/B attention
 * kernels, the hardware probes and the process-global switches below.  Used by scripts/ (microbenchmarks, ablations) and
 * by the A/B parity tests; never by contrastors_amd's product path.  Everything in contrastors_hip.h is exported here too. */
#ifndef CONTRASTORS_HIP_DEV_H
#define CONTRASTORS_HIP_DEV_H

#include "contrastors_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

void cx_gemm_set_variant(int v); /* 5 (default): 256x256x64 2-stage; 2: 256x128x64 3-stage ring; 3/4: persistent experiments; 1: 128x128 */
int cx_gemm_get_variant(void);

void cx_gemm_set_debug(int bits); /* experiments only (0 = normal): bit0/bit1 ablate the v2 main loop; bit2 = non-persistent v5 */

/* in-kernel phase timers of the persistent 256x256x64 kernel: buf = int64[grid*8 waves*8] {wait, compute, epilogue
   cycles, iterations, epilogue DMA-wait cycles, -, -, -} per wave, or NULL to disable (scripts/gemm_trace.py) */
void cx_gemm_set_trace(void* buf);

/* one-wave-per-SIMD kernel: ablation builds (mask bits: 1 no DMA, 2 no barrier, 4 no fragment reads, 8 no MFMA, 16 no
   epilogue, 32 no DMA wait, 128 trace only; results are garbage, timing is the point) and their per-workgroup trace
   buffer int64[grid][2] = {s_memtime span, K-tiles} (scripts/gemm_ablate.py) */
void cx_gemm_v6_ablate(int mask);
void cx_gemm_v6_trace(void* buf);
/* deferred register stores of the one-wave-per-SIMD kernel's plain form (gemm_bf16_v6.hip, DEFER): -1 = the shipped policy
   (default), 0 = never, 1 = every launch the form covers (A/B, parity tests) */

/* routing between the one-wave-per-SIMD kernel (gemm_bf16_v6.hip) and the two-workgroups-per-CU kernel
   (gemm_bf16_v7.hip): -1 = the shipped policy (default), 0 = never v7, 1 = every launch v7 covers (A/B, parity tests) */
void cx_gemm_v7_mode(int mode);
/* residency census of the v7 kernel: buf = int64[grid][4] {s_memtime at entry, at exit, HW_REG_HW_ID, HW_REG_XCC_ID} per
   workgroup (NULL = off); cx_gemm_v7_occupancy = the occupancy API's workgroups per CU at its LDS / register budget */
void cx_gemm_v7_trace(void* buf);
int cx_gemm_v7_occupancy(void);
/* ablation instantiations of the v7 kernel (plain and SwiGLU-backward forms; mask bits: 1 no LDS-DMA in the K loop, 2 no
   barrier, 4 no W fragment reads, 8 no MFMA, 16 no epilogue, 32 no X fragment reads, 64 no vmcnt waits; a mask that is not
   instantiated runs the real kernel; results are garbage) */
void cx_gemm_v7_ablate(int mask);
void cx_gemm_v7_flags(int flags); /* experiments: bit 0 = the K loop runs at s_setprio 1, the epilogue at 0 */
/* start stagger of the v7 kernel: tile period in shader cycles over which the workgroups' starts are spread (-1 = the
   launcher's estimate, 0 = no stagger: every workgroup starts at once) */
void cx_gemm_v7_period(int cycles);

void cx_gemm_set_glds(int enable); /* 1 (default): operand tiles via global_load_lds DMA; 0: register staging */
int cx_gemm_get_glds(void);

/* A/B switches of the attention dispatch (benchmarks, tests; process-global, dev library only).  The product library has none:
 * max_seqlen <= 128 -> attn_fwd_s128v / attn_bwd_fused2_s128 (single pass, one workgroup per problem); 128 < max_seqlen <= 256 ->
 * attn_fwd_s256 (K / V resident); longer, or any length beyond 128 in the backward -> the second-generation streaming kernels
 * (attn_fwd_long / attn_bwd_dq_long / attn_bwd_dkv_long) unless tables ask for rotate-on-load or the call is kv-packed, which keep round
 * 1's streaming kernels. */
void cx_attn_set_fwd_s128(int mode);   /* 0: the general streaming forward also for max_seqlen <= 256; non-zero (default): single pass */
void cx_attn_set_bwd_s128(int mode);   /* 0: the general streaming backward also for max_seqlen <= 128; non-zero (default): fused persistent kernel */
void cx_attn_set_fwd_long(int on);     /* 0: round 1's attn_fwd_kernel for max_seqlen > 256; 1 (default): attn_fwd_long_kernel */
void cx_attn_set_bwd_long(int on);     /* 0: round 1's delta + dQ + dK/dV kernels for max_seqlen > 128; 1 (default): attn_bwd_dq_long / _dkv_long */
void cx_attn_set_prio(int on);         /* experiments: the MFMA loop of the fused S <= 128 backward at s_setprio 1 */

/* ---- hardware self-checks used by tests (MFMA fragment layout, transpose-read semantics) ------------------- */
int cx_probe_mfma_layout(float* out_32x32, void* stream);           /* D = A*B with A[i][k]=i+1 (k==0), asymmetric B */
int cx_probe_ds_read_tr16(const uint16_t* in_64x4, uint16_t* out_64x4, void* stream);
/* MFMA issue-rate probe (scripts/mfma_probe.py): see probe.hip */
/* keep[b][h][q][key] in {0, 1}: the mask cx_attn_varlen_dropout_fwd/_bwd apply (tests) */
int cx_attn_dropout_keep_mask(unsigned char* keep, int B, int H, int S, float p_drop, unsigned long long seed,
                              unsigned long long offset, unsigned int site, void* stream);
/* what the dQ accumulation of a single-owner fused long-sequence attention backward costs by itself: every workgroup walks its
 * problems (floats_per_problem fp32 each, contiguous), `sweeps` load + add + store passes over each (scripts/dq_rmw_probe.py) */
int cx_probe_rmw(float* buf, long floats_per_problem, int sweeps, int n_problems, int nwg, void* stream);
int cx_probe_mfma_rate16(const void* seed_2048x16B, int waves, int iters, int nwg, long long* cycles_nwg_x8, float* sink,
                         void* stream);  /* the same loop from v_mfma_f32_16x16x32_bf16 */
int cx_probe_mfma_rate(const void* seed_2048x16B, int waves, int iters, int nwg, long long* cycles_nwg_x8, float* sink,
                       void* stream);
/* global->LDS DMA throughput probe (scripts/dma_probe.py): see probe.hip */
int cx_probe_dma_bw(const void* src, long wg_stride, long span, long row_stride, int per_wave, int iters, int depth,
                    int nwg, float* sink, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CONTRASTORS_HIP_DEV_H */
