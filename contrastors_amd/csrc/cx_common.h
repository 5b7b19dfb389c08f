// cx_common.h -- device-side helpers shared by every gfx950 kernel in this library.
//
// Everything here is written for CDNA4 only (wave64, MFMA 32x32x16 bf16, 64-bank LDS);
// there is deliberately no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bfloat16 bits; all bf16 tensors cross the C-ABI as uint16_t*

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

#define CX_DEVICE __device__ __forceinline__

// ---------------------------------------------------------------------------------------------
// bf16 <-> f32
// ---------------------------------------------------------------------------------------------
CX_DEVICE float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
CX_DEVICE float bf16lo_to_f32(uint32_t packed) { return __uint_as_float(packed << 16); }
CX_DEVICE float bf16hi_to_f32(uint32_t packed) { return __uint_as_float(packed & 0xffff0000u); }

// fp32 -> bf16, round-to-nearest-even: gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32, NaN stays NaN).
typedef float cx_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 cx_bf16x2 __attribute__((ext_vector_type(2)));
CX_DEVICE uint32_t pack_bf16x2(float lo, float hi) {
    const cx_f32x2 v = {lo, hi};
    const cx_bf16x2 r = __builtin_convertvector(v, cx_bf16x2);
    return __builtin_bit_cast(uint32_t, r);
}
CX_DEVICE bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }

// Backward of act = y * silu(g) from the saved (act, gate) pair (round 3's compact save: y = act / silu(g) is recovered
// inside the derivative):  d y = d * g * s,  d gate = d * y * silu'(g) = d * act * (1 / g + 1 - s),  s = sigmoid(g).
// An exactly-zero gate (or a silu that underflowed) means act = 0, y is not recoverable and d gate comes out 0.  1 / g is
// clamped to +-1e30 (one v_med3_f32: round 5): 0 * (clamped 1/0) = 0 with an ordinary multiply, and a denormal gate
// next to a non-zero (denormal) act gives a finite d gate instead of inf * x -- round 4's legacy multiply covered act == 0 only.
// No compare / select per element (the fc2-dgrad + SwiGLU-backward epilogue spends ~3200 VALU instructions per tile and wave in
// the one wave that also issues the MFMAs).  Every |g| >= 1e-30 is untouched by the clamp: same bits as round 4 there.
// NaN: v_med3_f32 returns a finite bound for a NaN 1/g, so a NaN GATE yields a finite (garbage) d gate here -- but d y = g * s * d is NaN for the
// same element, and both leave in the same (T, 2I) gradient tensor that feeds the fc1 wgrad / dgrad: the non-finite value still reaches the
// gradient norm the trainer checks.
CX_DEVICE float rcp_clamped(float g) { return __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(g), -1e30f, 1e30f); }
CX_DEVICE void swiglu_bwd_from_act(float d, float act, float g, float& dy, float& dg) {
    const float s = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * g));
    dy = g * s * d;
    dg = (d * act) * (rcp_clamped(g) + 1.f - s);
}
// The same on element PAIRS (v_pk_mul_f32 / v_pk_add_f32: two lanes' worth of fp32 per instruction; the transcendental and
// median instructions have no packed form).  Same operations in the same order per element: bit-identical to the
// scalar form.
typedef float cx_f2 __attribute__((ext_vector_type(2)));
CX_DEVICE void swiglu_bwd_from_act2(cx_f2 d, cx_f2 act, cx_f2 g, cx_f2& dy, cx_f2& dg) {
    const cx_f2 m = g * -1.4426950408889634f;
    const cx_f2 t = cx_f2{__builtin_amdgcn_exp2f(m.x), __builtin_amdgcn_exp2f(m.y)} + 1.f;
    const cx_f2 s = {__builtin_amdgcn_rcpf(t.x), __builtin_amdgcn_rcpf(t.y)};
    dy = g * s * d;
    const cx_f2 w = cx_f2{rcp_clamped(g.x), rcp_clamped(g.y)} + 1.f - s;
    dg = (d * act) * w;
}

// erf-GELU pieces on v_exp_f32 / v_rcp_f32: erf(x) = sign(x) (1 - poly(t) exp(-x^2)), t = 1 / (1 + 0.3275911 |x|)
// (Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7 -- two orders below a bf16 ulp).  libm's erff costs ~40 VALU ops per
// element and made both GELU kernels VALU-bound at a third of HBM speed.  For gelu the argument is v / sqrt(2), so
// exp(-x^2) = exp(-v^2 / 2) is also the Gaussian of the derivative: `gauss` is returned for reuse.
CX_DEVICE float gelu_cdf(float v, float& gauss) {
    const float av = fabsf(v);
    const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * 0.70710678118654752f * av);
    const float poly =
        t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    gauss = __builtin_amdgcn_exp2f(-0.72134752044448170f * v * v);  // exp(-v^2 / 2)
    const float erf_abs = 1.f - poly * gauss;
    return 0.5f * (1.f + copysignf(erf_abs, v));
}

// MLP activations of the plain (non-gated) MLP, selected at run time (sc/layers/mlp.py:8-34 `activation`; block.py:45-52):
//   CX_ACT_GELU        exact-erf GELU (BERT-base, HF / timm ViTs)
//   CX_ACT_QUICK_GELU  x * sigmoid(1.702 x) (sc/layers/activations.py:4-5; the OpenAI-CLIP image tower, sc/models/vit/clip.py)
// act_val: activation value; act_grad: d act / d v.
enum { CX_ACT_GELU = 0, CX_ACT_QUICK_GELU = 1 };
CX_DEVICE float act_val(float v, int act) {
    if (act == CX_ACT_QUICK_GELU) return v * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * v));
    float gauss;
    return v * gelu_cdf(v, gauss);
}
CX_DEVICE float act_grad(float v, int act) {
    if (act == CX_ACT_QUICK_GELU) {
        const float s = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * v));
        return s * (1.f + 1.702f * v * (1.f - s));
    }
    float gauss;
    const float cdf = gelu_cdf(v, gauss);
    return cdf + v * 0.3989422804014327f * gauss;
}

// ---------------------------------------------------------------------------------------------
// wave / block reductions (wave = 64 lanes)
// ---------------------------------------------------------------------------------------------
CX_DEVICE float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
CX_DEVICE float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---------------------------------------------------------------------------------------------
// LDS tile layout used by every MFMA kernel for K-contiguous operands:
//   a tile is [rows][64 bf16] = 128 B per row = 8 chunks of 16 B.  Chunk c of row r lives at
//   byte  r*128 + ((c ^ ((r>>1)&7)) << 4).
// With this XOR the 16-lane service groups of ds_read_b128 (rows distinct mod 16, same c) hit 16
// distinct 16-B slots of the 256-B bank row: conflict free (see DESIGN.md "LDS layouts").
// ---------------------------------------------------------------------------------------------
CX_DEVICE int tile64_off(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

// MFMA wrappers ---------------------------------------------------------------------------------
// D(32x32) += A(32x16) * B(16x32).  Lane l supplies A[i=l&31][k=8*(l>>5)+e] and B[k=8*(l>>5)+e][j=l&31],
// e=0..7.  Result: lane l holds D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31] in register r.
CX_DEVICE f32x16_t mfma_bf16_32x32x16(bf16x8_t a, bf16x8_t b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// Row index held by accumulator register r of a 32x32 MFMA result for lane-half hi.
CX_DEVICE int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

CX_DEVICE bf16x8_t lds_read_frag(const char* lds, int byte_off) {
    return *reinterpret_cast<const bf16x8_t*>(lds + byte_off);
}

// XCD-aware remap of a linear workgroup id: the dispatcher places block b on XCD b%8, so give each
// XCD a contiguous run of logical tiles (shared operand panels then hit the same 4 MiB L2).
// Bijective for any nwg (guide T1, bijective variant).
CX_DEVICE int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}


// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., the counter-based generator behind torch's CUDA/HIP generator): dropout masks are a pure
// function of (seed, offset, element index), so the backward -- and a GradCache re-forward under the reference's
// RandContext (sc/rand_state.py:6-22), which restores the generator's (seed, offset) -- regenerates them bit for bit
// instead of storing them.
// ---------------------------------------------------------------------------------------------
CX_DEVICE uint4 philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32 x 32 -> 64-bit product per multiplier (v_mad_u64_u32): written as __umulhi + a 32-bit multiply the compiler emitted
        // v_mul_hi_u32 AND v_mul_lo_u32 -- four quarter-rate instructions per round instead of two (round 5)
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c.x, p1 = (unsigned long long)0xCD9E8D57u * c.z;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += 0x9E3779B9u;
        k.y += 0xBB67AE85u;
    }
    return c;
}
// keep-mask of 4 consecutive elements: element group `g` (64-bit) of dropout site `site` under (seed, offset)
struct CxDropout {
    float p;                       // drop probability; 0 = off
    unsigned long long seed, offset;
};
CX_DEVICE void dropout_keep4(const CxDropout& d, uint32_t site, unsigned long long g, float (&keep)[4]) {
    const unsigned long long off = d.offset + site;
    const uint4 r = philox4x32_10(make_uint4((uint32_t)g, (uint32_t)(g >> 32), (uint32_t)off, (uint32_t)(off >> 32)),
                                  make_uint2((uint32_t)d.seed, (uint32_t)(d.seed >> 32)));
    const uint32_t thr = (uint32_t)fminf(d.p * 4294967296.f, 4294967040.f);
    const float inv = 1.f / (1.f - d.p);
    keep[0] = r.x >= thr ? inv : 0.f;
    keep[1] = r.y >= thr ? inv : 0.f;
    keep[2] = r.z >= thr ? inv : 0.f;
    keep[3] = r.w >= thr ? inv : 0.f;
}

// ---------------------------------------------------------------------------------------------
// host side: opt a kernel into > 64 KiB of dynamic LDS.  The attribute is per device, so it is remembered per device
// index (the deployment is one process per GPU, but nothing in the library may silently depend on that).
// ---------------------------------------------------------------------------------------------
struct CxLdsOptIn {
    bool done[32] = {};
    bool ensure(const void* fn, int bytes) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) dev = 0;
        if (done[dev]) return true;
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
        done[dev] = true;
        return true;
    }
};
