// Shared host/device helpers of liborbgpu (MI355X / gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <mutex>

#include "orbgpu.h"

// The whole library is compiled with -ffp-contract=off; this pragma makes the intent explicit at
// file scope as well: every float expression that mirrors the reference (cvRound(x*b + y*a),
// fastAtan2's polynomial, back-projection) must round after every operation.
#pragma clang fp contract(off)

namespace orbgpu {

void set_error(const char *fmt, ...);

#define ORBGPU_HIP_TRY(expr)                                                                                \
    do {                                                                                                    \
        hipError_t e__ = (expr);                                                                            \
        if (e__ != hipSuccess) {                                                                            \
            ::orbgpu::set_error("%s: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__);       \
            return ORBGPU_EHIP;                                                                             \
        }                                                                                                   \
    } while (0)

#define ORBGPU_REQUIRE(cond, ...)                                                                           \
    do {                                                                                                    \
        if (!(cond)) {                                                                                      \
            ::orbgpu::set_error(__VA_ARGS__);                                                               \
            return ORBGPU_EINVAL;                                                                           \
        }                                                                                                   \
    } while (0)

// Simple owning device buffer (grow-only).
// XCD-aware block placement for (items, frames) grids.  Workgroups are dealt round-robin over the 8 XCDs (each
// with a private 4 MiB L2), so in launch order the blocks of ONE frame land on all eight L2s and every L2 fetches
// that frame's rows from HBM.  The remap gives the blocks that share an XCD (linear id % 8) a contiguous range of
// the frame-major virtual order, i.e. whole frames, so the halo rows / overlapping patches that neighbouring
// blocks of a frame read are L2 hits.  Bijective for any grid size (cdna_hip_programming.md, "XCD swizzle must be
// bijective"); placement is a speed matter only -- results never depend on it.
#ifdef __HIPCC__
__device__ __forceinline__ void xcd_frame_block(int &bx, int &frame)
{
    const unsigned gx = gridDim.x, nwg = gridDim.x * gridDim.y;
    const unsigned orig = blockIdx.x + gx * blockIdx.y;
    const unsigned q = nwg >> 3, r = nwg & 7u, xcd = orig & 7u;
    const unsigned v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    frame = (int)(v / gx);
    bx = (int)(v - (unsigned)frame * gx);
}
// Serpentine frame order.  The stages of a batched extraction follow each other on one stream, each over all frames: by
// the time stage k + 1 reaches a frame, the planes stage k read or wrote for it have long left the 256 MB Infinity Cache
// (a 512-frame batch moves > 1 GB per stage) -- except the frames stage k touched LAST.  Kernels that call this walk
// the frames in descending order, their neighbours in the chain in ascending order, so a stage starts on what is still
// cached: blur (descending) -> FAST (ascending) -> orientation (descending).  Measured at 512 frames per launch:
// k_orient 0.211 -> 0.201 ms, k_fast_detect 0.642 -> 0.634 ms.  Results never depend on the order.
__device__ __forceinline__ int serpentine(int frame) { return (int)gridDim.y - 1 - frame; }
#endif

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int reserve(size_t n)
    {
        if (n <= bytes)
            return ORBGPU_OK;
        if (p) {
            (void)hipFree(p);
            p = nullptr;
            bytes = 0;
        }
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) {
            set_error("hipMalloc(%zu): %s", n, hipGetErrorString(e));
            p = nullptr;
            return ORBGPU_ENOMEM;
        }
        bytes = n;
        return ORBGPU_OK;
    }
    void release()
    {
        if (p)
            (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

int select_device(int device_id);

// Creation and destruction of handles take this lock: they are the calls that allocate / free device and pinned memory,
// create / destroy streams, events and graphs, and synchronise the device -- the runtime has aborted once (1 of ~9 runs of
// the GPU suite) with two threads inside orbgpu_extractor_destroy at the same moment.  Nothing on a per-frame path takes it.
std::mutex &lifecycle_mutex();

// ---- device helpers -------------------------------------------------------------------------
__device__ __forceinline__ int wave_reduce_add(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_down(v, off, 64);
    return v;
}

// 256-bit Hamming distance of two descriptors held as 4 x u64.
__device__ __forceinline__ int hamming256(const uint64_t a[4], const uint64_t b[4])
{
    return __popcll(a[0] ^ b[0]) + __popcll(a[1] ^ b[1]) + __popcll(a[2] ^ b[2]) + __popcll(a[3] ^ b[3]);
}

} // namespace orbgpu
