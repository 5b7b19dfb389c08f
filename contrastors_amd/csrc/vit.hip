// vit.hip -- ViT front end of the image tower: patchify, [cls | patch projections] + position embeddings, and the
// backward of that assembly.  Replaces PatchEmbedding.forward (sc/layers/embedding.py:465-516: rearrange
// "b c (h p1) (w p2) -> b h w (c p1 p2)", Linear, cls token, + pos_embed) around the patch-projection GEMM, which runs
// on the bf16 MFMA GEMM like every other Linear.  All HBM-bound, 16 B per lane where the layout allows.
#include "cx_common.h"
#include "../../include/contrastors_hip.h"

namespace {

// pixels (B, C, H, W) fp32 or bf16 -> patches (B * hp * wp, C * p * p) bf16, feature order (c, p1, p2) as the
// reference's rearrange; one thread per 4 consecutive p2 pixels (p % 4 == 0).
// keep (PatchDropout, sc/layers/embedding.py:519-557; NULL = every patch): (B, K) patch indices per image -- output row
// b * K + j holds patch keep[b][j], i.e. only the kept patches are gathered, projected and run through the blocks.
template <typename T>
__global__ void patchify_kernel(const T* __restrict__ pix, bf16_t* __restrict__ out, int B, int Cc, int H, int W, int p,
                                long total4, const int32_t* __restrict__ keep, int K) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total4) return;
    const int p4 = p / 4;
    long r = i;
    const int q2 = (int)(r % p4); r /= p4;     // which group of 4 along p2
    const int p1 = (int)(r % p); r /= p;
    const int c = (int)(r % Cc); r /= Cc;
    const int wp = W / p, hp = H / p;
    int pw, ph, b;
    size_t row;
    if (keep) {
        const int j = (int)(r % K); r /= K;
        b = (int)r;
        const int pi = keep[(size_t)b * K + j];
        ph = pi / wp;
        pw = pi - ph * wp;
        row = (size_t)b * K + j;
    } else {
        pw = (int)(r % wp); r /= wp;
        ph = (int)(r % hp); r /= hp;
        b = (int)r;
        row = ((size_t)b * hp + ph) * wp + pw;
    }
    const T* src = pix + (((size_t)b * Cc + c) * H + (size_t)ph * p + p1) * W + (size_t)pw * p + q2 * 4;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if constexpr (sizeof(T) == 4) v[e] = src[e];
        else v[e] = bf16_to_f32(src[e]);
    }
    uint2 pk;
    pk.x = pack_bf16x2(v[0], v[1]);
    pk.y = pack_bf16x2(v[2], v[3]);
    *reinterpret_cast<uint2*>(out + row * ((size_t)Cc * p * p) + ((size_t)c * p + p1) * p + q2 * 4) = pk;
}

// out[b, 0, :] = cls + pos[0];  out[b, 1 + j, :] = proj[b * P + j, :] + pos[1 + j]   (fp32 add, bf16 store)
// keep != NULL: P = kept patches per image, sequence position s >= 1 of image b is patch keep[b][s - 1] -> pos[1 + that]
__global__ void vit_assemble_fwd_kernel(const bf16_t* __restrict__ proj, const float* __restrict__ cls,
                                        const float* __restrict__ pos, bf16_t* __restrict__ out, int P, int d,
                                        long total8, const int32_t* __restrict__ keep) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total8) return;
    const int d8 = d / 8;
    const int c8 = (int)(i % d8);
    const long row = i / d8;                 // b * (P+1) + s
    const int s = (int)(row % (P + 1));
    const long b = row / (P + 1);
    float v[8];
    if (s == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = cls[c8 * 8 + e];
    } else {
        const uint4 u = *reinterpret_cast<const uint4*>(proj + ((size_t)b * P + (s - 1)) * d + c8 * 8);
        v[0] = bf16lo_to_f32(u.x); v[1] = bf16hi_to_f32(u.x); v[2] = bf16lo_to_f32(u.y); v[3] = bf16hi_to_f32(u.y);
        v[4] = bf16lo_to_f32(u.z); v[5] = bf16hi_to_f32(u.z); v[6] = bf16lo_to_f32(u.w); v[7] = bf16hi_to_f32(u.w);
    }
    const int ps = (s > 0 && keep) ? 1 + keep[(size_t)b * P + (s - 1)] : s;
    const float* pp = pos + (size_t)ps * d + c8 * 8;
    uint4 o;
    o.x = pack_bf16x2(v[0] + pp[0], v[1] + pp[1]);
    o.y = pack_bf16x2(v[2] + pp[2], v[3] + pp[3]);
    o.z = pack_bf16x2(v[4] + pp[4], v[5] + pp[5]);
    o.w = pack_bf16x2(v[6] + pp[6], v[7] + pp[7]);
    *reinterpret_cast<uint4*>(out + (size_t)row * d + c8 * 8) = o;
}

// Backward of the assembly.  Grid (d/8 column groups, P+1 sequence positions); each thread walks the batch:
//   dproj[b*P + s-1, :] = dz[b, s, :]            (s >= 1; the contiguous operand of the patch-projection wgrad)
//   gpos[s, :] += sum_b dz[b, s, :]               gcls[:] += sum_b dz[b, 0, :]
// One owner per (s, column) -> deterministic, no atomics.
// inv != NULL (PatchDropout): the grid walks the P_all + 1 ORIGINAL positions; inv[b][patch] = position of that patch among image
// b's K kept ones, or -1: the owner of (original position, column) still sums over the batch in order -- deterministic.
__global__ void vit_assemble_bwd_kernel(const bf16_t* __restrict__ dz, bf16_t* __restrict__ dproj,
                                        float* __restrict__ gcls, float* __restrict__ gpos, int B, int P, int d,
                                        const int32_t* __restrict__ inv, int P_all) {
    const int c8 = blockIdx.x * blockDim.x + threadIdx.x;
    const int s = blockIdx.y;
    if (c8 * 8 >= d) return;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < B; ++b) {
        int sk = s;                               // position in the (kept) sequence of image b
        if (inv && s > 0) {
            const int j = inv[(size_t)b * P_all + (s - 1)];
            if (j < 0) continue;                  // this patch was dropped from image b
            sk = 1 + j;
        }
        const uint4 u = *reinterpret_cast<const uint4*>(dz + ((size_t)b * (P + 1) + sk) * d + c8 * 8);
        if (s > 0) *reinterpret_cast<uint4*>(dproj + ((size_t)b * P + (sk - 1)) * d + c8 * 8) = u;
        acc[0] += bf16lo_to_f32(u.x); acc[1] += bf16hi_to_f32(u.x); acc[2] += bf16lo_to_f32(u.y); acc[3] += bf16hi_to_f32(u.y);
        acc[4] += bf16lo_to_f32(u.z); acc[5] += bf16hi_to_f32(u.z); acc[6] += bf16lo_to_f32(u.w); acc[7] += bf16hi_to_f32(u.w);
    }
    if (gpos) {
#pragma unroll
        for (int e = 0; e < 8; ++e) gpos[(size_t)s * d + c8 * 8 + e] += acc[e];
    }
    if (s == 0 && gcls) {
#pragma unroll
        for (int e = 0; e < 8; ++e) gcls[c8 * 8 + e] += acc[e];
    }
}

inline int done() { return hipGetLastError() == hipSuccess ? CX_OK : CX_ERR_LAUNCH; }

}  // namespace

extern "C" {

int cx_vit_patchify_gather(const void* pixels, int pixels_bf16, uint16_t* patches, int B, int Cc, int H, int W, int patch,
                           const int32_t* keep, int n_keep, void* stream) {
    if (B <= 0) return CX_OK;
    if (!pixels || !patches) return CX_ERR_ARG;
    if (patch <= 0 || (patch % 4) || (H % patch) || (W % patch)) return CX_ERR_SHAPE;
    const int P_all = (H / patch) * (W / patch);
    if (keep && (n_keep <= 0 || n_keep > P_all)) return CX_ERR_ARG;
    const long total4 = keep ? (long)B * n_keep * Cc * patch * patch / 4 : (long)B * Cc * H * W / 4;
    const int grid = (int)((total4 + 255) / 256);
    if (pixels_bf16)
        hipLaunchKernelGGL(patchify_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)pixels, patches, B, Cc, H, W, patch, total4, keep, n_keep);
    else
        hipLaunchKernelGGL(patchify_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)pixels,
                           patches, B, Cc, H, W, patch, total4, keep, n_keep);
    return done();
}

int cx_vit_patchify(const void* pixels, int pixels_bf16, uint16_t* patches, int B, int Cc, int H, int W, int patch,
                    void* stream) {
    return cx_vit_patchify_gather(pixels, pixels_bf16, patches, B, Cc, H, W, patch, nullptr, 0, stream);
}

int cx_vit_assemble_fwd_gather(const uint16_t* proj, const float* cls_token, const float* pos_embed, uint16_t* out, int B,
                               int P, int d, const int32_t* keep, void* stream) {
    if (B <= 0) return CX_OK;
    if (!proj || !cls_token || !pos_embed || !out) return CX_ERR_ARG;
    if (d % 8) return CX_ERR_SHAPE;
    const long total8 = (long)B * (P + 1) * (d / 8);
    hipLaunchKernelGGL(vit_assemble_fwd_kernel, dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       proj, cls_token, pos_embed, out, P, d, total8, keep);
    return done();
}

int cx_vit_assemble_fwd(const uint16_t* proj, const float* cls_token, const float* pos_embed, uint16_t* out, int B,
                        int P, int d, void* stream) {
    return cx_vit_assemble_fwd_gather(proj, cls_token, pos_embed, out, B, P, d, nullptr, stream);
}

int cx_vit_assemble_bwd_gather(const uint16_t* dz, uint16_t* dproj, float* gcls, float* gpos, int B, int P, int d,
                               const int32_t* inv, int P_all, void* stream) {
    if (B <= 0) return CX_OK;
    if (!dz || !dproj) return CX_ERR_ARG;
    if (d % 8 || (inv && P_all < P)) return CX_ERR_SHAPE;
    const int d8 = d / 8;
    const int bx = d8 < 64 ? d8 : 64;
    hipLaunchKernelGGL(vit_assemble_bwd_kernel, dim3((d8 + bx - 1) / bx, (inv ? P_all : P) + 1), dim3(bx), 0, (hipStream_t)stream, dz,
                       dproj, gcls, gpos, B, P, d, inv, P_all);
    return done();
}

int cx_vit_assemble_bwd(const uint16_t* dz, uint16_t* dproj, float* gcls, float* gpos, int B, int P, int d,
                        void* stream) {
    return cx_vit_assemble_bwd_gather(dz, dproj, gcls, gpos, B, P, d, nullptr, P, stream);
}

}  // extern "C"
