// CPU checks around computeOrbDescriptor's cos / sin (reference src/ORBextractor.cc:112-113); driven by tests/test_trig.py.
//   (1) overload resolution: with `using namespace std;` in force, cos(float) IS std::cos(float), i.e. float-valued;
//   (2) exhaustive scan of [0, ORBGPU_TRIG_MAX]: this host's cosf / sinf against the fixed double sequence of
//       csrc/trig_base.h (compiled here by g++, in the library by clang: the counts must agree), against the libm double
//       functions rounded to float, and against sincosf (a compiler may merge the reference's two calls into it).
#define _GNU_SOURCE 1
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <type_traits>
#include <vector>

#include "trig_base.h"

using namespace std;  // as ORBextractor.cc:67

static float resolved_cos(float angle) { return cos(angle); }  // the expression of :113
static_assert(is_same<decltype(cos(1.0f)), float>::value, "cos(float) resolves to the float overload, std::cos(float)");
static_assert(is_same<decltype(sin(1.0f)), float>::value, "sin(float) resolves to the float overload, std::sin(float)");

int main()
{
    uint32_t last;
    const float hi = orbgpu::ORBGPU_TRIG_MAX;
    memcpy(&last, &hi, 4);
    const int nt = (int)max(1u, min(thread::hardware_concurrency(), 16u));
    struct Acc {
        uint64_t entries = 0, cos_vs_base = 0, sin_vs_base = 0, base_vs_rounded = 0, sincos = 0, resolved = 0;
    };
    vector<Acc> acc((size_t)nt);
    vector<thread> th;
    for (int t = 0; t < nt; t++)
        th.emplace_back([&, t] {
            Acc a;
            for (uint64_t u = (uint64_t)t; u <= last; u += (uint64_t)nt) {
                const uint32_t uu = (uint32_t)u;
                float x, bc, bs, sc, ss;
                memcpy(&x, &uu, 4);
                orbgpu::orbgpu_sincos_base(x, &bc, &bs);
                const float hc = cosf(x), hs = sinf(x);
                sincosf(x, &ss, &sc);
                a.entries += (hc != bc) || (hs != bs);
                a.cos_vs_base += hc != bc;
                a.sin_vs_base += hs != bs;
                a.base_vs_rounded += (bc != (float)::cos((double)x)) + (bs != (float)::sin((double)x));
                a.sincos += (sc != hc) || (ss != hs);
                a.resolved += resolved_cos(x) != hc;
            }
            acc[(size_t)t] = a;
        });
    for (auto &x : th)
        x.join();
    Acc s;
    for (auto &a : acc) {
        s.entries += a.entries, s.cos_vs_base += a.cos_vs_base, s.sin_vs_base += a.sin_vs_base;
        s.base_vs_rounded += a.base_vs_rounded, s.sincos += a.sincos, s.resolved += a.resolved;
    }
    printf("{\"values\": %llu, \"entries\": %llu, \"cos_differs\": %llu, \"sin_differs\": %llu, \"base_vs_rounded_double\": %llu, "
           "\"sincosf_differs\": %llu, \"resolved_call_differs_from_cosf\": %llu}\n",
           (unsigned long long)last + 1ull, (unsigned long long)s.entries, (unsigned long long)s.cos_vs_base,
           (unsigned long long)s.sin_vs_base, (unsigned long long)s.base_vs_rounded, (unsigned long long)s.sincos,
           (unsigned long long)s.resolved);
    return 0;
}
