// Shared host/device helpers of liborbgpu (MI355X / gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

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
