// Error reporting, device selection and the ABI's bookkeeping entry points.
#include "common.h"
#include "workspace.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace orbgpu {

// see workspace.h: from here on per-thread workspaces are dropped without HIP calls
__attribute__((constructor)) static void register_exit_flag()
{
    std::atexit([] { process_exiting().store(true); });
}

static thread_local char t_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_err, sizeof(t_err), fmt, ap);
    va_end(ap);
}

std::mutex &lifecycle_mutex()
{
    static std::mutex m;
    return m;
}

int select_device(int device_id)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device available (%s); liborbgpu has no CPU fallback",
                  e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
        return ORBGPU_EHIP;
    }
    if (device_id < 0 || device_id >= n) {
        set_error("device_id %d out of range (%d devices)", device_id, n);
        return ORBGPU_EINVAL;
    }
    e = hipSetDevice(device_id);
    if (e != hipSuccess) {
        set_error("hipSetDevice(%d): %s", device_id, hipGetErrorString(e));
        return ORBGPU_EHIP;
    }
    return ORBGPU_OK;
}

// plain streaming copies, 16 bytes per lane and access: the practical HBM roofline on this box.  Three
// shapes are timed and the best is reported: one access per thread, a grid-stride loop with four loads in flight
// per lane, and the one-access shape with non-temporal loads / stores.
__global__ __launch_bounds__(256) void k_copy16(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a;
        dst[i + stride] = b;
        dst[i + 2 * stride] = c;
        dst[i + 3 * stride] = d;
    }
    for (; i < n16; i += stride)
        dst[i] = src[i];
}
__global__ __launch_bounds__(256) void k_copy16_flat(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16)
        dst[i] = src[i];
}
__global__ __launch_bounds__(256) void k_copy16_nt(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) {
        typedef unsigned v4 __attribute__((ext_vector_type(4)));
        const v4 v = __builtin_nontemporal_load(reinterpret_cast<const v4 *>(src) + i);
        __builtin_nontemporal_store(v, reinterpret_cast<v4 *>(dst) + i);
    }
}

} // namespace orbgpu

extern "C" {

int orbgpu_measure_copy_bandwidth(size_t bytes, int32_t reps, int32_t device_id, float *gbs)
{
    using namespace orbgpu;
    ORBGPU_REQUIRE(gbs && bytes >= (1u << 20) && reps >= 1, "bad arguments");
    int rc = select_device(device_id);
    if (rc != ORBGPU_OK)
        return rc;
    DevBuf a, b;
    if ((rc = a.reserve(bytes)) != ORBGPU_OK || (rc = b.reserve(bytes)) != ORBGPU_OK) {
        a.release();
        b.release();
        return rc;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t he = hipMemset(a.p, 1, bytes);
    if (he == hipSuccess)
        he = hipEventCreate(&e0);
    if (he == hipSuccess)
        he = hipEventCreate(&e1);
    float best = 0.f;
    if (he == hipSuccess) {
        const size_t n16 = bytes / 16;
        const dim3 flat((unsigned)((n16 + 255) / 256));
        for (int variant = 0; variant < 3 && he == hipSuccess; variant++) {
            auto launch = [&](const uint4 *src, uint4 *dst) {
                if (variant == 0)
                    hipLaunchKernelGGL(k_copy16_flat, flat, dim3(256), 0, nullptr, src, dst, n16);
                else if (variant == 1)
                    hipLaunchKernelGGL(k_copy16, dim3(256 * 8), dim3(256), 0, nullptr, src, dst, n16);
                else
                    hipLaunchKernelGGL(k_copy16_nt, flat, dim3(256), 0, nullptr, src, dst, n16);
            };
            launch(a.as<uint4>(), b.as<uint4>());  // warm-up
            (void)hipEventRecord(e0, nullptr);
            for (int r = 0; r < reps; r++)
                launch((r & 1) ? b.as<uint4>() : a.as<uint4>(), (r & 1) ? a.as<uint4>() : b.as<uint4>());
            (void)hipEventRecord(e1, nullptr);
            he = hipEventSynchronize(e1);
            float ms = 0.f;
            if (he == hipSuccess)
                he = hipEventElapsedTime(&ms, e0, e1);
            if (he == hipSuccess && ms > 0.f)
                best = std::max(best, (float)(2.0 * (double)(n16 * 16) * reps / ((double)ms * 1e-3) / 1e9));  // read + write
        }
    }
    if (e0)
        (void)hipEventDestroy(e0);
    if (e1)
        (void)hipEventDestroy(e1);
    a.release();
    b.release();
    if (he != hipSuccess) {
        set_error("copy bandwidth: %s", hipGetErrorString(he));
        return ORBGPU_EHIP;
    }
    *gbs = best;
    return ORBGPU_OK;
}

const char *orbgpu_last_error_string(void) { return orbgpu::t_err; }
int orbgpu_abi_version(void) { return ORBGPU_ABI_VERSION; }
int orbgpu_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

} // extern "C"
