// Harness of tests/test_math_exact.py: wurblpt_amd/csrc/wpt_math.h against the C library of the machine it runs on
// (build container: glibc 2.35, x86-64 with FMA), exhaustively for the one-argument functions.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <omp.h>

#include "../wurblpt_amd/csrc/wpt_math.h"

static inline float fromBits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline bool same(float a, float b) { return memcmp(&a, &b, 4) == 0 || (a != a && b != b); }

template<class F, class G> static unsigned long long allFloats(const char* name, F libm, G mine)
{
    unsigned long long bad = 0;
#pragma omp parallel for schedule(static) reduction(+ : bad)
    for (long long i = 0; i < (1ll << 32); i++) {
        volatile float x = fromBits((uint32_t)i); /* volatile: the compiler must call the library, not fold */
        if (!same(libm(x), mine(x)))
            bad++;
    }
    printf("%s all 2^32 arguments: %llu differences\n", name, bad);
    fflush(stdout);
    return bad;
}

static inline uint64_t next(uint64_t& s)
{
    uint64_t z = (s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

template<class F, class G> static unsigned long long pairs(const char* name, F libm, G mine, long long count)
{
    std::vector<float> v = { 0.0f, -0.0f, 1.0f, -1.0f, 2.0f, -2.0f, 0.5f, -0.5f, 3.0f, -3.0f, 1e-45f, -1e-45f, 1.17549435e-38f, -1.17549435e-38f,
        1e-40f, 3.4028235e38f, -3.4028235e38f, fromBits(0x7f800000u), fromBits(0xff800000u), fromBits(0x7fc00000u), 1e30f, -1e30f, 1e-30f,
        16777216.0f, -16777216.0f, 16777217.0f, 8388608.0f, 8388609.0f, 0.99999994f, 1.0000001f, 127.0f, 128.0f, -149.0f, -150.0f, 1024.0f, 2.4f,
        0.33333334f, 1.5f, -1.5f, 2.5f, 1e10f, 1e-10f, 88.0f, -88.0f, 100.0f };
    for (int e = -149; e <= 127; e += 3) {
        v.push_back(ldexpf(1.0f, e));
        v.push_back(-ldexpf(1.5f, e));
    }
    unsigned long long bad = 0;
    for (float x : v)
        for (float y : v)
            if (!same(libm(x, y), mine(x, y)))
                bad++;
#pragma omp parallel reduction(+ : bad)
    {
        uint64_t s = 12345 + 7919 * omp_get_thread_num();
#pragma omp for schedule(static)
        for (long long i = 0; i < count; i++) {
            const uint64_t r = next(s);
            float x, y;
            if (i & 1) { /* any two bit patterns */
                x = fromBits((uint32_t)r);
                y = fromBits((uint32_t)(r >> 32));
            } else { /* what the renderer asks for: bases in (0, 1], exponents in [0, 4096) */
                x = (float)((r & 0xffffff) + 1) * 0x1p-24f;
                y = (float)((r >> 24) & 0xffffff) * 0x1p-12f;
            }
            volatile float xv = x, yv = y;
            if (!same(libm(xv, yv), mine(x, y)))
                bad++;
        }
    }
    printf("%s special values and %lld argument pairs: %llu differences\n", name, count, bad);
    fflush(stdout);
    return bad;
}

int main(int argc, char** argv)
{
    const long long count = argc > 1 ? atoll(argv[1]) : 200000000ll;
    unsigned long long bad = 0;
    bad += allFloats("sinf", [](float x) { return sinf(x); }, [](float x) { return wptm::sinf_(x); });
    bad += allFloats("cosf", [](float x) { return cosf(x); }, [](float x) { return wptm::cosf_(x); });
    bad += allFloats("expf", [](float x) { return expf(x); }, [](float x) { return wptm::expf_(x); });
    bad += allFloats("asinf", [](float x) { return asinf(x); }, [](float x) { return wptm::asinf_(x); });
    bad += allFloats("acosf", [](float x) { return acosf(x); }, [](float x) { return wptm::acosf_(x); });
    bad += allFloats("atanf", [](float x) { return atanf(x); }, [](float x) { return wptm::atanf_(x); });
    bad += allFloats("2 * asin (double, rounded to float)", [](float x) { volatile double d = x; return (float)(2.0 * asin(d)); },
            [](float x) { return (x >= -1.0f && x <= 1.0f) ? (float)(2.0 * wptm::asin_d((double)x)) : (float)(2.0 * asin((double)x)); });
    bad += pairs("powf", [](float x, float y) { return powf(x, y); }, [](float x, float y) { return wptm::powf_(x, y); }, count);
    bad += pairs("atan2f", [](float y, float x) { return atan2f(y, x); }, [](float y, float x) { return wptm::atan2f_(y, x); }, count);
    printf("total: %llu differences\n", bad);
    return bad == 0 ? 0 : 1;
}
