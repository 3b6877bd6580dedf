#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "wurblpt/rgl.hpp"
#include "../wurblpt_amd/csrc/wpt_rgl.h"
struct HostMath {
    static float sin(float x) { return std::sin(x); }
    static float cos(float x) { return std::cos(x); }
    static float atan2(float y, float x) { return std::atan2(y, x); }
    static float acos(float x) { return std::acos(x); }
    static float sqrt(float x) { return std::sqrt(x); }
    static float twiceAsin(float x) { return float(2.0 * ::asin(double(x))); }
};
int main(int argc, char** argv)
{
    int bad = 0;
    for (int f = 1; f < argc; f++) {
        std::vector<float> pool;
        wpt_rgl_brdf b;
        std::string err;
        if (!WurblPT::buildRglBrdf(argv[f], pool, b, err)) { printf("load failed: %s\n", err.c_str()); return 2; }
        if (!wptrgl::rglInterleavable(b)) { printf("%s: not interleavable\n", argv[f]); return 3; }
        const size_t size = size_t(b.rgb.size_x) * b.rgb.size_y, slices = size_t(b.luminance.param_size[0]) * b.luminance.param_size[1];
        const size_t at = (pool.size() + 3) & ~size_t(3);
        std::vector<float> p2(pool);
        p2.resize(at + slices * size * 4);
        for (size_t sl = 0; sl < slices; sl++)
            for (size_t e = 0; e < size; e++) {
                for (size_t c = 0; c < 3; c++) p2[at + (sl * size + e) * 4 + c] = pool[b.rgb.data + (sl * 3 + c) * size + e];
                p2[at + (sl * size + e) * 4 + 3] = pool[b.luminance.data + sl * size + e];
            }
        std::mt19937 rng(7);
        std::uniform_real_distribution<float> U(0.0f, 1.0f);
        for (int i = 0; i < 200000; i++) {
            auto dir = [&]() { float z = U(rng), ph = 6.2831853f * U(rng), r = std::sqrt(std::max(0.0f, 1 - z * z)); wptrgl::V3 v { r * std::cos(ph), r * std::sin(ph), z }; return v; };
            wptrgl::V3 wi = dir(), wo = dir();
            wptrgl::V2 u { U(rng), U(rng) };
            const wptrgl::RglIncident a = wptrgl::rglIncident<HostMath>(b, pool.data(), wi), c = wptrgl::rglIncident<HostMath>(b, p2.data(), wi, uint32_t(at));
            wptrgl::V3 woA, woC, frA, frC; float pA, pC, qA, qC;
            wptrgl::V3 sA = wptrgl::rglSampleWith<HostMath>(b, pool.data(), a, u, wi, woA, pA), sC = wptrgl::rglSampleWith<HostMath>(b, p2.data(), c, u, wi, woC, pC);
            wptrgl::rglEvalPdfWith<HostMath>(b, pool.data(), a, wi, wo, frA, qA);
            wptrgl::rglEvalPdfWith<HostMath>(b, p2.data(), c, wi, wo, frC, qC);
            float x[12] = { sA.x, sA.y, sA.z, woA.x, woA.y, woA.z, pA, frA.x, frA.y, frA.z, qA, 0 }, y[12] = { sC.x, sC.y, sC.z, woC.x, woC.y, woC.z, pC, frC.x, frC.y, frC.z, qC, 0 };
            if (memcmp(x, y, sizeof(x)) != 0) { if (bad < 5) printf("%s record %d differs\n", argv[f], i); bad++; }
        }
    }
    printf("mismatches: %d\n", bad);
    return bad != 0;
}
