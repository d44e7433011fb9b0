// The host libm's cosf / sinf on the device (see trig_base.h): the exception table and the process-wide mode switch.
//
// computeOrbDescriptor (reference src/ORBextractor.cc:112-113) calls std::cos(float) / std::sin(float) of the machine it
// runs on.  The table makes k_trig return exactly those values: every float x in [0, ORBGPU_TRIG_MAX] is evaluated once
// with the host's cosf / sinf and with orbgpu_sincos_base(); where they differ (1.47 M of 1.09 G arguments with glibc
// 2.35 on an FMA-capable x86-64) the host's pair is recorded.  The scan runs once per process, at the first
// extractor creation in ORBGPU_TRIG_HOST_LIBM mode, on all CPUs the process may use (about 15 core-seconds).
#include "common.h"
#include "trig_base.h"

#include <sched.h>

#include <chrono>
#include <cmath>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

namespace orbgpu {

namespace {

struct HostTable {
    std::vector<uint32_t> bucket, key;
    std::vector<float2> val;
    double build_ms = 0;
    bool built = false;
};
HostTable g_host;
std::mutex g_mu;
int g_mode = ORBGPU_TRIG_HOST_LIBM;

struct DevTable {
    DevBuf bucket, key, val;
};
std::map<int, DevTable *> g_dev;  // per device, released with the process

uint32_t bits_of(float f)
{
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}

int scan_threads()
{
    cpu_set_t set;
    int n = 0;
    if (sched_getaffinity(0, sizeof(set), &set) == 0)
        n = CPU_COUNT(&set);
    if (n <= 0)
        n = (int)std::thread::hardware_concurrency();
    return std::max(1, std::min(n, 32));
}

// cosf / sinf through pointers the optimiser cannot see through: the calls stay two separate libm calls, as written in
// the reference (a compiler may merge the pair into sincosf; tests/test_trig.py checks that this libm returns the same
// bits either way)
float (*volatile p_cosf)(float) = cosf;
float (*volatile p_sinf)(float) = sinf;

void build_host_table()
{
    const auto t0 = std::chrono::steady_clock::now();
    const uint32_t last = bits_of(ORBGPU_TRIG_MAX);
    const uint32_t nb = (last >> ORBGPU_TRIG_BUCKET_SHIFT) + 1;
    const int nt = scan_threads();
    struct Part {
        std::vector<uint32_t> key, cnt;
        std::vector<float2> val;
    };
    std::vector<Part> parts((size_t)nt);
    std::vector<std::thread> th;
    // the low binades hold almost no exceptions and cost the same per value: interleave buckets over the threads in
    // chunks so that every thread gets a share of every binade
    const uint32_t chunk = 512;  // buckets per work item (1 M floats)
    for (int t = 0; t < nt; t++)
        th.emplace_back([&, t] {
            Part &P = parts[(size_t)t];
            float (*const fc)(float) = p_cosf;
            float (*const fs)(float) = p_sinf;
            for (uint32_t b0 = (uint32_t)t * chunk; b0 < nb; b0 += (uint32_t)nt * chunk)
                for (uint32_t b = b0; b < std::min(b0 + chunk, nb); b++) {
                    uint32_t c = 0;
                    const uint32_t u0 = b << ORBGPU_TRIG_BUCKET_SHIFT;
                    const uint32_t u1 = std::min(u0 + (1u << ORBGPU_TRIG_BUCKET_SHIFT) - 1u, last);
                    for (uint32_t u = u0; u <= u1; u++) {
                        float x, bc, bs;
                        std::memcpy(&x, &u, 4);
                        orbgpu_sincos_base(x, &bc, &bs);
                        const float hc = fc(x), hs = fs(x);
                        if (bits_of(hc) != bits_of(bc) || bits_of(hs) != bits_of(bs)) {
                            P.key.push_back(u);
                            P.val.push_back(make_float2(hc, hs));
                            c++;
                        }
                    }
                    P.cnt.push_back(c);
                }
        });
    for (auto &x : th)
        x.join();
    // stitch the parts back together in bucket order
    g_host.bucket.assign((size_t)nb + 1, 0u);
    std::vector<size_t> pos((size_t)nt, 0), cpos((size_t)nt, 0);
    size_t total = 0;
    for (auto &P : parts)
        total += P.key.size();
    g_host.key.reserve(total);
    g_host.val.reserve(total);
    for (uint32_t b = 0; b < nb; b++) {
        const int t = (int)((b / chunk) % (uint32_t)nt);
        Part &P = parts[(size_t)t];
        const uint32_t c = P.cnt[cpos[(size_t)t]++];
        g_host.bucket[b] = (uint32_t)g_host.key.size();
        g_host.key.insert(g_host.key.end(), P.key.begin() + (long)pos[(size_t)t], P.key.begin() + (long)(pos[(size_t)t] + c));
        g_host.val.insert(g_host.val.end(), P.val.begin() + (long)pos[(size_t)t], P.val.begin() + (long)(pos[(size_t)t] + c));
        pos[(size_t)t] += c;
    }
    g_host.bucket[nb] = (uint32_t)g_host.key.size();
    g_host.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    g_host.built = true;
}

} // namespace

// The table of the calling thread's current device (created on first use); nbuckets = 0 in ORBGPU_TRIG_ROUNDED_DOUBLE mode.
int trig_table_for_device(int device_id, TrigTable *out)
{
    std::lock_guard<std::mutex> lk(g_mu);
    *out = TrigTable{nullptr, nullptr, nullptr, 0u};
    if (g_mode == ORBGPU_TRIG_ROUNDED_DOUBLE)
        return ORBGPU_OK;
    if (!g_host.built)
        build_host_table();
    DevTable *&d = g_dev[device_id];
    if (!d) {
        DevTable *n = new (std::nothrow) DevTable();
        if (!n) {
            set_error("out of host memory");
            return ORBGPU_ENOMEM;
        }
        const size_t ne = std::max<size_t>(g_host.key.size(), 1);
        int rc = n->bucket.reserve(g_host.bucket.size() * 4);
        if (rc == ORBGPU_OK)
            rc = n->key.reserve(ne * 4);
        if (rc == ORBGPU_OK)
            rc = n->val.reserve(ne * 8);
        hipError_t he = hipSuccess;
        if (rc == ORBGPU_OK)
            he = hipMemcpy(n->bucket.p, g_host.bucket.data(), g_host.bucket.size() * 4, hipMemcpyHostToDevice);
        if (rc == ORBGPU_OK && he == hipSuccess && !g_host.key.empty()) {
            he = hipMemcpy(n->key.p, g_host.key.data(), g_host.key.size() * 4, hipMemcpyHostToDevice);
            if (he == hipSuccess)
                he = hipMemcpy(n->val.p, g_host.val.data(), g_host.val.size() * 8, hipMemcpyHostToDevice);
        }
        if (rc == ORBGPU_OK && he == hipSuccess)
            he = hipStreamSynchronize(nullptr);  // (the kernels that read the table run on non-blocking streams)
        if (rc != ORBGPU_OK || he != hipSuccess) {
            if (rc == ORBGPU_OK) {
                set_error("trig table upload: %s", hipGetErrorString(he));
                rc = ORBGPU_EHIP;
            }
            n->bucket.release();
            n->key.release();
            n->val.release();
            delete n;
            return rc;
        }
        d = n;
    }
    out->bucket = d->bucket.as<uint32_t>();
    out->key = d->key.as<uint32_t>();
    out->val = d->val.as<float2>();
    out->nbuckets = (uint32_t)g_host.bucket.size() - 1u;
    return ORBGPU_OK;
}

__global__ __launch_bounds__(256) void k_trig_eval(const float *__restrict__ x, int n, float *__restrict__ c, float *__restrict__ s,
                                                   TrigTable tt)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n)
        orbgpu_trig_device(x[i], tt, c + i, s + i);
}

} // namespace orbgpu

extern "C" {

/* test aid: the device's cos / sin (the function k_trig calls) of n host floats, in the current mode */
int orbgpu_trig_eval(const float *x, int32_t n, float *c, float *s, int32_t device_id)
{
    using namespace orbgpu;
    ORBGPU_REQUIRE(x && c && s && n >= 0, "bad arguments");
    if (n == 0)
        return ORBGPU_OK;
    int rc = select_device(device_id);
    if (rc != ORBGPU_OK)
        return rc;
    TrigTable tt;
    rc = trig_table_for_device(device_id, &tt);
    if (rc != ORBGPU_OK)
        return rc;
    DevBuf dx, dc, ds;
    rc = dx.reserve((size_t)n * 4);
    if (rc == ORBGPU_OK)
        rc = dc.reserve((size_t)n * 4);
    if (rc == ORBGPU_OK)
        rc = ds.reserve((size_t)n * 4);
    hipError_t he = hipSuccess;
    if (rc == ORBGPU_OK) {
        he = hipMemcpy(dx.p, x, (size_t)n * 4, hipMemcpyHostToDevice);
        if (he == hipSuccess) {
            hipLaunchKernelGGL(k_trig_eval, dim3((n + 255) / 256), dim3(256), 0, 0, dx.as<float>(), n, dc.as<float>(), ds.as<float>(), tt);
            he = hipGetLastError();
        }
        if (he == hipSuccess)
            he = hipMemcpy(c, dc.p, (size_t)n * 4, hipMemcpyDeviceToHost);
        if (he == hipSuccess)
            he = hipMemcpy(s, ds.p, (size_t)n * 4, hipMemcpyDeviceToHost);
        if (he != hipSuccess) {
            set_error("orbgpu_trig_eval: %s", hipGetErrorString(he));
            rc = ORBGPU_EHIP;
        }
    }
    dx.release();
    dc.release();
    ds.release();
    return rc;
}

int orbgpu_set_trig_mode(int32_t mode)
{
    using namespace orbgpu;
    ORBGPU_REQUIRE(mode == ORBGPU_TRIG_HOST_LIBM || mode == ORBGPU_TRIG_ROUNDED_DOUBLE, "unknown trig mode %d", mode);
    std::lock_guard<std::mutex> lk(g_mu);
    g_mode = mode;
    return ORBGPU_OK;
}

int orbgpu_get_trig_mode(void)
{
    std::lock_guard<std::mutex> lk(orbgpu::g_mu);
    return orbgpu::g_mode;
}

int orbgpu_trig_table_info(int64_t *entries, double *build_ms)
{
    using namespace orbgpu;
    std::lock_guard<std::mutex> lk(g_mu);
    if (entries)
        *entries = g_host.built ? (int64_t)g_host.key.size() : -1;
    if (build_ms)
        *build_ms = g_host.built ? g_host.build_ms : 0.0;
    return ORBGPU_OK;
}

/* test aid (no device needed): builds the host table if necessary and returns the host's cosf / sinf of x as the device
 * would deliver them -- base value, or the table's entry where there is one */
int orbgpu_trig_host_eval(float x, float *c, float *s, int32_t *from_table)
{
    using namespace orbgpu;
    ORBGPU_REQUIRE(c && s, "null argument");
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_host.built)
        build_host_table();
    orbgpu_sincos_base(x, c, s);
    if (from_table)
        *from_table = 0;
    const uint32_t u = bits_of(x), b = u >> ORBGPU_TRIG_BUCKET_SHIFT;
    if (b + 1 < g_host.bucket.size()) {
        uint32_t lo = g_host.bucket[b], hi = g_host.bucket[b + 1];
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (g_host.key[mid] < u)
                lo = mid + 1;
            else
                hi = mid;
        }
        if (lo < g_host.bucket[b + 1] && g_host.key[lo] == u) {
            *c = g_host.val[lo].x;
            *s = g_host.val[lo].y;
            if (from_table)
                *from_table = 1;
        }
    }
    return ORBGPU_OK;
}

} // extern "C"
