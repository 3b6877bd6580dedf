/*
 * prng.hpp -- the per-pixel random number generator on the host (reference prng.hpp:47-101), for
 * applications that build their scenes with it (wurblpt-rtiow.cpp:39-70 places its spheres with Prng(17)).
 *
 * xoshiro128+ (Blackman and Vigna, public domain) seeded through splitmix64 from pixelIndex + 42, floats from the
 * upper 24 bits: the stream the kernels draw per pixel (wurblpt_amd/csrc/wpt_device.h, prngSeed / in01), bit for
 * bit; the golden vectors of tests/golden/ref_golden.json hold the reference's own streams.
 */
#pragma once

#include <cstdint>

#include "gvm.hpp"

namespace WurblPT {

class Prng
{
private:
    uint32_t _state[4];

    static uint64_t mix(uint64_t x)
    {
        x += 0x9e3779b97f4a7c15ull;
        x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
        x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
        return x ^ (x >> 31);
    }

public:
    Prng(unsigned int pixelIndex)
    {
        const uint64_t first = mix(uint64_t(pixelIndex) + 42u);
        const uint64_t second = mix(first);
        _state[0] = uint32_t(first >> 32);
        _state[1] = uint32_t(first);
        _state[2] = uint32_t(second >> 32);
        _state[3] = uint32_t(second);
    }

    float in01()
    {
        const uint32_t sum = _state[0] + _state[3];
        const uint32_t shifted = _state[1] << 9;
        _state[2] ^= _state[0];
        _state[3] ^= _state[1];
        _state[1] ^= _state[2];
        _state[0] ^= _state[3];
        _state[2] ^= shifted;
        _state[3] = (_state[3] << 11) | (_state[3] >> 21);
        return float(sum >> 8) * 0x1.0p-24f;
    }

    /* two draws; which of them lands in x is the compiler's choice in the reference (prng.hpp:97-100: the draws are
     * function arguments), g++ fills y first and the kernels follow that */
    vec2 in01x2()
    {
        const float second = in01();
        const float first = in01();
        return vec2(first, second);
    }
};

}
