// cos / sin of computeOrbDescriptor's float angle (reference src/ORBextractor.cc:112-113), reproducible on host and device.
//
// The reference writes  float a = (float)cos(angle), b = (float)sin(angle);  with `using namespace std;` in force
// (:66-67): overload resolution picks std::cos(float), i.e. libm's cosf / sinf.  cosf is not correctly rounded and its
// last bit depends on the libm build (on x86-64 glibc even on the CPU: an FMA variant is selected at load time), so "the
// value the reference computes" is a property of the host the reference runs on.  The device therefore evaluates
//   (1) orbgpu_sincos_base(): a fixed sequence of IEEE double operations (no libm call, no contraction) that both the
//       host compiler and the device compiler turn into the same bits, rounded to float, and
//   (2) a table of the arguments for which the HOST's cosf / sinf differ from (1), holding the host's values, built by
//       scanning every float of the argument range once per process (csrc/trig.hip).
// base + exceptions == the host's cosf / sinf for every float in [0, ORBGPU_TRIG_MAX], by construction.
#pragma once

#include <cstdint>
#include <cstring>

#ifdef __HIPCC__
#define ORBGPU_HD __host__ __device__ __forceinline__
#else
#define ORBGPU_HD inline
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#endif  // gcc: build with -ffp-contract=off

namespace orbgpu {

// kpt.angle is in [0, 360] (cv::fastAtan2), angle = kpt.angle * (float)(CV_PI / 180.f) <= 6.2831855f; the table covers
// a little more
constexpr float ORBGPU_TRIG_MAX = 6.2832f;
constexpr int ORBGPU_TRIG_BUCKET_SHIFT = 11;  // 2048 consecutive floats per bucket of the exception table: 0.53 M buckets (2.1 MB, L2-resident);
                                               // ~50 keys per bucket where angles fall (the top binades), so a search touches <= 4 cache lines

// x in [0, 6.3]: cos and sin with an error of a few 1e-17 (Cody-Waite reduction by pi/2 in two pieces, Taylor series
// to r^18 / r^17 on |r| <= pi/4), rounded to float.  Every operation is a single IEEE double operation.
ORBGPU_HD void orbgpu_sincos_base(float xf, float *c, float *s)
{
    const double x = (double)xf;
    const int k = (int)(x * 0.63661977236758138 + 0.5);  // nearest multiple of pi/2 (k = 0..4)
    const double kd = (double)k;
    // pi/2 = 1.57079632673412561417 (33 significant bits: kd * hi is exact) + 6.07710050650619224932e-11
    const double r = (x - kd * 1.57079632673412561417e+00) - kd * 6.07710050650619224932e-11;
    const double z = r * r;
    double p = 1.0 / 355687428096000.0;                // 1/17!
    p = -1.0 / 1307674368000.0 + z * p;                // 1/15!
    p = 1.0 / 6227020800.0 + z * p;                    // 1/13!
    p = -1.0 / 39916800.0 + z * p;                     // 1/11!
    p = 1.0 / 362880.0 + z * p;                        // 1/9!
    p = -1.0 / 5040.0 + z * p;                         // 1/7!
    p = 1.0 / 120.0 + z * p;                           // 1/5!
    p = -1.0 / 6.0 + z * p;                            // 1/3!
    const double sr = r + r * (z * p);
    double q = -1.0 / 6402373705728000.0;              // 1/18!
    q = 1.0 / 20922789888000.0 + z * q;                // 1/16!
    q = -1.0 / 87178291200.0 + z * q;                  // 1/14!
    q = 1.0 / 479001600.0 + z * q;                     // 1/12!
    q = -1.0 / 3628800.0 + z * q;                      // 1/10!
    q = 1.0 / 40320.0 + z * q;                         // 1/8!
    q = -1.0 / 720.0 + z * q;                          // 1/6!
    q = 1.0 / 24.0 + z * q;                            // 1/4!
    const double cr = (1.0 - 0.5 * z) + (z * z) * q;
    double cc, ss;
    switch (k & 3) {
    case 0: cc = cr, ss = sr; break;
    case 1: cc = -sr, ss = cr; break;
    case 2: cc = -cr, ss = -sr; break;
    default: cc = sr, ss = -cr; break;
    }
    *c = (float)cc;
    *s = (float)ss;
}

#ifdef __HIPCC__
// the exception table as the device sees it: bucket b holds the entries whose argument bits >> BUCKET_SHIFT == b
struct TrigTable {
    const uint32_t *bucket;  // [nbuckets + 1] first entry of each bucket
    const uint32_t *key;     // [n] argument bits, ascending
    const float2 *val;       // [n] (cosf, sinf) of the host
    uint32_t nbuckets;       // 0: no table (ORBGPU_TRIG_ROUNDED_DOUBLE)
};

// cos / sin of `angle` (radians) as the table's host computes cosf / sinf: the base value unless the table lists the
// argument (one load for the bucket bounds, issued before the double-precision evaluation, then a binary search)
__device__ __forceinline__ void orbgpu_trig_device(float angle, const TrigTable &tt, float *c, float *s)
{
    const uint32_t u = __float_as_uint(angle), b = u >> ORBGPU_TRIG_BUCKET_SHIFT;
    uint32_t lo = 0, end = 0;
    if (b < tt.nbuckets) {
        lo = tt.bucket[b];
        end = tt.bucket[b + 1];
    }
    orbgpu_sincos_base(angle, c, s);
    uint32_t hi = end;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (tt.key[mid] < u)
            lo = mid + 1;
        else
            hi = mid;
    }
    if (lo < end && tt.key[lo] == u) {
        const float2 v = tt.val[lo];
        *c = v.x;
        *s = v.y;
    }
}
#endif

} // namespace orbgpu
