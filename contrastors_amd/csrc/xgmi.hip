// One-shot exchange over xGMI for the loss path's single exchange step.
//
// The embedding all-gather of sc/distributed.py:5-12 moves 2048 x 768 fp32 = 6.3 MB per rank: far too small for a ring
// (7 serial hops, each latency-bound).  MI355X nodes are fully connected -- every GPU has a direct xGMI link to each of the
// other seven -- so the natural schedule is ONE step: every rank stores its shard straight into the other ranks' receive
// buffers, all seven links of every GPU busy at once, then one flag exchange.  Buffers are device memory shared between
// the per-GPU processes with HIP IPC handles (dmabuf; HSA_ENABLE_IPC_MODE_LEGACY=0); receive buffers and flags are
// allocated uncached (fine-grained, hipDeviceMallocUncached -- the host side's choice, cx_ipc_alloc) so that a peer's
// stores can never hide behind a stale line of the reader's L2; flags are written / polled with system-scope atomics.  The reduce-scatter of the backward is the same step in the
// other direction followed by a local sum of the W received slices.
//
// Host protocol (contrastors_amd/distributed.py::OneShotExchange): receive buffers are used round-robin (N_BUF >= 2), so a
// rank that runs ahead writes buffer (k+1) % N while a slower peer still reads buffer k % N, and it cannot reach collective
// k + 2 before that peer has signalled k + 1 -- which the peer enqueues behind its copy-out of k (results are copied out of
// the receive buffer by default; read in place they stay valid for N - 1 further collectives).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/contrastors_hip.h"

namespace {

inline int ok(hipError_t e) { return e == hipSuccess ? CX_OK : CX_ERR_LAUNCH; }

// dst[p] + dst_off <- src for every peer p: 16 bytes per lane, blockIdx.y = peer
__global__ __launch_bounds__(256) void xgmi_push_kernel(const uint4* __restrict__ src, char* const* __restrict__ dst, long dst_off,
                                                        long n16) {
    uint4* out = reinterpret_cast<uint4*>(dst[blockIdx.y] + dst_off);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) out[i] = src[i];
}
// peer p receives slice p of src (W slices of n16 x 16 B) at dst[p] + slot * slice bytes: the reduce-scatter's send side
__global__ __launch_bounds__(256) void xgmi_scatter_kernel(const uint4* __restrict__ src, char* const* __restrict__ dst, int slot,
                                                           long n16) {
    const uint4* in = src + (long)blockIdx.y * n16;
    uint4* out = reinterpret_cast<uint4*>(dst[blockIdx.y]) + (long)slot * n16;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) out[i] = in[i];
}
// thread p: peer_flags[p][rank] = epoch (release, system scope: everything this stream wrote before is visible first),
// then wait until my_flags[p] >= epoch (acquire).  Bounded: after max_spins polls the kernel gives up and sets *err.
__global__ void xgmi_signal_wait_kernel(uint32_t* const* __restrict__ peer_flags, uint32_t* __restrict__ my_flags, int rank, int world,
                                        uint32_t epoch, long max_spins, uint32_t* __restrict__ err) {
    const int p = threadIdx.x;
    if (p >= world) return;
    __threadfence_system();
    __hip_atomic_store(peer_flags[p] + rank, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    long spins = 0;
    // (signed distance: the epoch counter may wrap)
    while ((int32_t)(__hip_atomic_load(my_flags + p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
        if (++spins > max_spins) {
            __hip_atomic_store(err, 1u + (uint32_t)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
        __builtin_amdgcn_s_sleep(8);
    }
    __threadfence_system();
}
// out[i] = sum over W slots (fp32, fixed order: deterministic)
__global__ __launch_bounds__(256) void sum_slots_kernel(const float4* __restrict__ slots, float4* __restrict__ out, long n4, int world) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 a = slots[i];
        for (int w = 1; w < world; ++w) {
            const float4 b = slots[(long)w * n4 + i];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        out[i] = a;
    }
}

}  // namespace

extern "C" {

int cx_ipc_alloc(void** ptr, long bytes, int uncached) {
    if (!ptr || bytes <= 0) return CX_ERR_ARG;
    if (uncached == 2)  // host memory the device can write and the host can poll without a stream synchronisation (error flag)
        return ok(hipHostMalloc(ptr, (size_t)bytes, hipHostMallocMapped | hipHostMallocCoherent));
    if (uncached) return ok(hipExtMallocWithFlags(ptr, (size_t)bytes, hipDeviceMallocUncached));
    return ok(hipMalloc(ptr, (size_t)bytes));
}
int cx_ipc_free(void* ptr) {
    if (!ptr) return CX_OK;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, ptr) == hipSuccess && a.type == hipMemoryTypeHost) return ok(hipHostFree(ptr));
    return ok(hipFree(ptr));
}
int cx_ipc_export(void* ptr, unsigned char* handle64) {
    if (!ptr || !handle64) return CX_ERR_ARG;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
    return ok(hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(handle64), ptr));
}
int cx_ipc_open(const unsigned char* handle64, void** ptr) {
    if (!ptr || !handle64) return CX_ERR_ARG;
    hipIpcMemHandle_t h;
    __builtin_memcpy(&h, handle64, 64);
    return ok(hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess));
}
int cx_ipc_close(void* ptr) { return ptr ? ok(hipIpcCloseMemHandle(ptr)) : CX_OK; }

int cx_xgmi_push(const void* src, void* const* peer_bufs_dev, long dst_offset_bytes, long bytes, int world, void* stream) {
    if (!src || !peer_bufs_dev || world < 1 || bytes < 0 || (bytes % 16) || (dst_offset_bytes % 16)) return CX_ERR_ARG;
    if (bytes == 0) return CX_OK;
    const long n16 = bytes / 16;
    long bx = (n16 + 255) / 256;
    if (bx > 128) bx = 128;   // per peer: 8 peers x 128 workgroups fill the chip; a link is saturated by far fewer
    hipLaunchKernelGGL(xgmi_push_kernel, dim3((unsigned)bx, world), dim3(256), 0, (hipStream_t)stream, (const uint4*)src,
                       (char* const*)peer_bufs_dev, dst_offset_bytes, n16);
    return ok(hipGetLastError());
}
int cx_xgmi_scatter(const void* src, void* const* peer_bufs_dev, int slot, long slice_bytes, int world, void* stream) {
    if (!src || !peer_bufs_dev || world < 1 || slice_bytes < 0 || (slice_bytes % 16) || slot < 0) return CX_ERR_ARG;
    if (slice_bytes == 0) return CX_OK;
    const long n16 = slice_bytes / 16;
    long bx = (n16 + 255) / 256;
    if (bx > 128) bx = 128;
    hipLaunchKernelGGL(xgmi_scatter_kernel, dim3((unsigned)bx, world), dim3(256), 0, (hipStream_t)stream, (const uint4*)src,
                       (char* const*)peer_bufs_dev, slot, n16);
    return ok(hipGetLastError());
}
int cx_xgmi_signal_wait(unsigned int* const* peer_flags_dev, unsigned int* my_flags, int rank, int world, unsigned int epoch,
                        long max_spins, unsigned int* err_flag, void* stream) {
    if (!peer_flags_dev || !my_flags || !err_flag || world < 1 || world > 64 || rank < 0 || rank >= world) return CX_ERR_ARG;
    hipLaunchKernelGGL(xgmi_signal_wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (uint32_t* const*)peer_flags_dev,
                       (uint32_t*)my_flags, rank, world, epoch, max_spins, (uint32_t*)err_flag);
    return ok(hipGetLastError());
}
int cx_sum_slots_f32(const float* slots, float* out, long n, int world, void* stream) {
    if (!slots || !out || n < 0 || (n % 4) || world < 1) return CX_ERR_ARG;
    if (n == 0) return CX_OK;
    long g = (n / 4 + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(sum_slots_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const float4*)slots, (float4*)out,
                       n / 4, world);
    return ok(hipGetLastError());
}

}  // extern "C"
