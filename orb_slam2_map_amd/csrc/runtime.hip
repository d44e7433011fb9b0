// Error reporting, device selection and the ABI's bookkeeping entry points.
#include "common.h"

#include <cstring>

namespace orbgpu {

static thread_local char t_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_err, sizeof(t_err), fmt, ap);
    va_end(ap);
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

} // namespace orbgpu

extern "C" {

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
