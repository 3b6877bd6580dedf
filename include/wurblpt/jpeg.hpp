/*
 * jpeg.hpp -- JPEG decoder for texture files (baseline, extended sequential and progressive Huffman
 * JPEG, 8 bits per sample, grey or YCbCr / RGB; restart intervals; any sampling factors whose ratios
 * are whole numbers).
 *
 * The reference reads images through libtgd, which decodes JPEG with libjpeg at its default settings.
 * Texture values enter the renderer bit for bit, so this decoder reproduces the default pipeline of
 * the libjpeg the distributions ship (libjpeg-turbo): the accurate integer inverse DCT ("islow",
 * 13 bit constants, two passes), "fancy" triangle-filter upsampling for 2:1 horizontal, 2:1 vertical
 * and 2x2 chroma, and its fixed-point YCbCr -> RGB tables.  tests/test_import.py holds files and the
 * pixels that library produces for them.
 */
#pragma once

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "array.hpp"

namespace WurblPT {

namespace jpegdetail {

static const unsigned char zigzag[64] = { 0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20,
    13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54,
    47, 55, 62, 63 };

struct HuffmanTable {
    bool defined = false;
    unsigned char vals[256];
    int mincode[17], maxcode[18], valptr[17];
    uint16_t look[512]; /* 9 bit lookahead: (length << 8) | value, 0 = longer code */

    /* false for lengths that do not form a prefix code (more codes of a length than there is room for) */
    bool build(const unsigned char* bits /* [1..16] */, const unsigned char* values, int count)
    {
        memcpy(vals, values, count);
        int code = 0, k = 0;
        memset(look, 0, sizeof(look));
        for (int l = 1; l <= 16; l++) {
            valptr[l] = k;
            mincode[l] = code;
            if (code + bits[l] > (1 << l))
                return false;
            for (int i = 0; i < bits[l]; i++, k++, code++) {
                if (l <= 9) {
                    const int first = code << (9 - l);
                    for (int f = 0; f < (1 << (9 - l)); f++)
                        look[first + f] = uint16_t((l << 8) | vals[k]);
                }
            }
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        defined = true;
        return true;
    }
};

struct BitReader {
    const unsigned char* p;
    const unsigned char* end;
    uint32_t acc = 0;
    int n = 0;
    int marker = 0; /* a marker was met: zeros are fed from here on */

    BitReader(const unsigned char* p, const unsigned char* end) : p(p), end(end) {}

    void fill()
    {
        while (n <= 24) {
            unsigned int byte = 0;
            if (!marker && p < end) {
                byte = *p++;
                if (byte == 0xff) {
                    unsigned int next = p < end ? *p : 0xd9;
                    if (next == 0) {
                        p++;
                    } else {
                        marker = int(next);
                        p--; /* stay on the marker */
                        byte = 0;
                    }
                }
            }
            acc |= byte << (24 - n);
            n += 8;
        }
    }
    int peek9()
    {
        if (n < 16)
            fill();
        return int(acc >> 23);
    }
    void skip(int k)
    {
        acc <<= k;
        n -= k;
    }
    int bit()
    {
        if (n < 1)
            fill();
        const int b = int(acc >> 31);
        skip(1);
        return b;
    }
    int receive(int s)
    {
        if (s == 0)
            return 0;
        if (n < s)
            fill();
        const int v = int(acc >> (32 - s));
        skip(s);
        return v;
    }
    int decode(const HuffmanTable& t)
    {
        const int e = t.look[peek9()];
        if (e) {
            skip(e >> 8);
            return e & 0xff;
        }
        int code = receive(9);
        for (int l = 10; l <= 16; l++) {
            code = (code << 1) | bit();
            if (code <= t.maxcode[l])
                return t.vals[(t.valptr[l] + code - t.mincode[l]) & 0xff];
        }
        return 0; /* corrupt data */
    }
    /* restart marker: drop the bits of the current byte, take the marker */
    bool restart()
    {
        acc = 0;
        n = 0;
        if (!marker) {
            /* markers may be preceded by fill bytes */
            while (p + 1 < end && !(p[0] == 0xff && p[1] != 0 && p[1] != 0xff))
                p++;
            if (p + 1 < end)
                marker = p[1];
        }
        if (marker < 0xd0 || marker > 0xd7)
            return false;
        p += 2;
        marker = 0;
        return true;
    }
};

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int td = 0, ta = 0;
    int width = 0, height = 0;       /* downsampled_width, downsampled_height */
    int blocksW = 0, blocksH = 0;    /* padded to whole MCUs */
    int pred = 0;
    std::vector<int16_t> coef;       /* blocksW * blocksH * 64, natural order */
    std::vector<unsigned char> plane; /* blocksW * 8 by blocksH * 8 samples */
};

/* jidctint.c: accurate integer inverse DCT; `q` in natural order */
inline void idctIslow(const int16_t* in, const uint16_t* q, unsigned char* out, int stride)
{
    constexpr int CONST_BITS = 13, PASS1_BITS = 2;
    constexpr int64_t F_0_298631336 = 2446, F_0_390180644 = 3196, F_0_541196100 = 4433, F_0_765366865 = 6270, F_0_899976223 = 7373,
                      F_1_175875602 = 9633, F_1_501321110 = 12299, F_1_847759065 = 15137, F_1_961570560 = 16069, F_2_053119869 = 16819,
                      F_2_562915447 = 20995, F_3_072711026 = 25172;
    auto descale = [](int64_t x, int n) { return (x + (int64_t(1) << (n - 1))) >> n; };
    int64_t ws[64];
    for (int pass = 0; pass < 2; pass++) {
        for (int i = 0; i < 8; i++) {
            int64_t d[8];
            if (pass == 0) {
                for (int k = 0; k < 8; k++)
                    d[k] = int64_t(in[8 * k + i]) * q[8 * k + i];
            } else {
                for (int k = 0; k < 8; k++)
                    d[k] = ws[8 * i + k];
            }
            int64_t z2 = d[2], z3 = d[6];
            int64_t z1 = (z2 + z3) * F_0_541196100;
            int64_t tmp2 = z1 + z3 * (-F_1_847759065);
            int64_t tmp3 = z1 + z2 * F_0_765366865;
            z2 = d[0];
            z3 = d[4];
            int64_t tmp0 = (z2 + z3) * (int64_t(1) << CONST_BITS);
            int64_t tmp1 = (z2 - z3) * (int64_t(1) << CONST_BITS);
            const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
            tmp0 = d[7];
            tmp1 = d[5];
            tmp2 = d[3];
            tmp3 = d[1];
            z1 = tmp0 + tmp3;
            z2 = tmp1 + tmp2;
            z3 = tmp0 + tmp2;
            int64_t z4 = tmp1 + tmp3;
            const int64_t z5 = (z3 + z4) * F_1_175875602;
            tmp0 *= F_0_298631336;
            tmp1 *= F_2_053119869;
            tmp2 *= F_3_072711026;
            tmp3 *= F_1_501321110;
            z1 *= -F_0_899976223;
            z2 *= -F_2_562915447;
            z3 *= -F_1_961570560;
            z4 *= -F_0_390180644;
            z3 += z5;
            z4 += z5;
            tmp0 += z1 + z3;
            tmp1 += z2 + z4;
            tmp2 += z2 + z3;
            tmp3 += z1 + z4;
            const int64_t r[8] = { tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0, tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2,
                tmp10 - tmp3 };
            if (pass == 0) {
                for (int k = 0; k < 8; k++)
                    ws[8 * k + i] = descale(r[k], CONST_BITS - PASS1_BITS);
            } else {
                for (int k = 0; k < 8; k++) {
                    /* range_limit[x & RANGE_MASK] of jdmaster.c, centred on 128 */
                    const int x = int(descale(r[k], CONST_BITS + PASS1_BITS + 3) & 1023);
                    out[i * stride + k] = (unsigned char)(x < 128 ? x + 128 : x < 512 ? 255 : x < 896 ? 0 : x - 896);
                }
            }
        }
    }
}

struct Decoder {
    const unsigned char* data;
    size_t size;
    std::string error;
    int width = 0, height = 0, ncomp = 0;
    bool progressive = false;
    int hmax = 1, vmax = 1, mcusX = 0, mcusY = 0;
    Component comp[4];
    uint16_t quant[4][64]; /* natural order */
    bool quantDefined[4] = { false, false, false, false };
    HuffmanTable dc[4], ac[4];
    int restartInterval = 0;
    int adobeTransform = -1;
    bool jfif = false;

    Decoder(const unsigned char* d, size_t n) : data(d), size(n) {}

    bool fail(const char* msg)
    {
        error = msg;
        return false;
    }
    static int be16(const unsigned char* p) { return (p[0] << 8) | p[1]; }

    bool parseFrame(const unsigned char* p, int len)
    {
        if (len < 6 || p[0] != 8)
            return fail("only 8 bit JPEG is decoded");
        height = be16(p + 1);
        width = be16(p + 3);
        ncomp = p[5];
        if (width == 0 || height == 0 || (ncomp != 1 && ncomp != 3) || len < 6 + 3 * ncomp)
            return fail("JPEG with this number of components is not decoded");
        for (int c = 0; c < ncomp; c++) {
            comp[c].id = p[6 + 3 * c];
            comp[c].h = p[7 + 3 * c] >> 4;
            comp[c].v = p[7 + 3 * c] & 15;
            comp[c].tq = p[8 + 3 * c] & 3;
            if (comp[c].h < 1 || comp[c].h > 4 || comp[c].v < 1 || comp[c].v > 4)
                return fail("bad JPEG sampling factors");
            hmax = comp[c].h > hmax ? comp[c].h : hmax;
            vmax = comp[c].v > vmax ? comp[c].v : vmax;
        }
        if (ncomp == 1) /* a single component is never interleaved: its sampling factors mean nothing */
            comp[0].h = comp[0].v = hmax = vmax = 1;
        mcusX = (width + 8 * hmax - 1) / (8 * hmax);
        mcusY = (height + 8 * vmax - 1) / (8 * vmax);
        /* every block costs at least one bit in every scan that touches it: a damaged size must not make us allocate
         * what the file cannot fill */
        if (uint64_t(mcusX) * mcusY > 8ull * size)
            return fail("truncated JPEG data");
        for (int c = 0; c < ncomp; c++) {
            Component& k = comp[c];
            if (hmax % k.h != 0 || vmax % k.v != 0)
                return fail("JPEG sampling factors with fractional ratios are not decoded");
            k.width = (width * k.h + hmax - 1) / hmax;
            k.height = (height * k.v + vmax - 1) / vmax;
            k.blocksW = mcusX * k.h;
            k.blocksH = mcusY * k.v;
            k.coef.assign(size_t(k.blocksW) * k.blocksH * 64, 0);
        }
        return true;
    }

    bool parseHuffman(const unsigned char* p, int len)
    {
        while (len > 0) {
            if (len < 17)
                return fail("bad Huffman table");
            const int tc = p[0] >> 4, th = p[0] & 15;
            unsigned char bits[17];
            bits[0] = 0;
            int count = 0;
            for (int i = 1; i <= 16; i++) {
                bits[i] = p[i];
                count += p[i];
            }
            if (tc > 1 || th > 3 || count > 256 || len < 17 + count)
                return fail("bad Huffman table");
            if (!(tc == 0 ? dc[th] : ac[th]).build(bits, p + 17, count))
                return fail("bad Huffman table");
            p += 17 + count;
            len -= 17 + count;
        }
        return true;
    }

    bool parseQuant(const unsigned char* p, int len)
    {
        while (len > 0) {
            const int pq = p[0] >> 4, tq = p[0] & 15;
            const int need = 1 + 64 * (pq ? 2 : 1);
            if (tq > 3 || len < need)
                return fail("bad quantization table");
            for (int i = 0; i < 64; i++)
                quant[tq][zigzag[i]] = uint16_t(pq ? be16(p + 1 + 2 * i) : p[1 + i]);
            quantDefined[tq] = true;
            p += need;
            len -= need;
        }
        return true;
    }

    /* one scan: entropy-coded data starts at `p`; returns where it ended */
    const unsigned char* decodeScan(const unsigned char* hdr, int len, const unsigned char* p)
    {
        const int ns = hdr[0];
        if (ns < 1 || ns > ncomp || len < 1 + 2 * ns + 3) {
            fail("bad scan header");
            return nullptr;
        }
        Component* sc[4];
        for (int i = 0; i < ns; i++) {
            sc[i] = nullptr;
            for (int c = 0; c < ncomp; c++)
                if (comp[c].id == hdr[1 + 2 * i])
                    sc[i] = &comp[c];
            if (!sc[i]) {
                fail("scan refers to an unknown component");
                return nullptr;
            }
            sc[i]->td = hdr[2 + 2 * i] >> 4;
            sc[i]->ta = hdr[2 + 2 * i] & 15;
            if (sc[i]->td > 3 || sc[i]->ta > 3) {
                fail("bad Huffman table index");
                return nullptr;
            }
        }
        const int Ss = hdr[1 + 2 * ns], Se = hdr[2 + 2 * ns], Ah = hdr[3 + 2 * ns] >> 4, Al = hdr[3 + 2 * ns] & 15;
        if (progressive ? (Ss > Se || Se > 63 || (Ss == 0 && Se != 0) || (Ss > 0 && ns != 1)) : false) {
            fail("bad progressive scan parameters");
            return nullptr;
        }
        for (int i = 0; i < ns; i++) {
            const bool needDc = !progressive || (Ss == 0 && Ah == 0);
            const bool needAc = !progressive || Ss > 0;
            if ((needDc && !dc[sc[i]->td].defined) || (needAc && !ac[sc[i]->ta].defined)) {
                fail("scan uses a Huffman table that was not defined");
                return nullptr;
            }
            sc[i]->pred = 0;
        }
        BitReader br(p, data + size);
        int eobrun = 0;
        /* a scan of one component walks that component's own blocks, an interleaved one walks MCUs */
        const bool interleaved = ns > 1;
        const int unitsX = interleaved ? mcusX : (sc[0]->width + 7) / 8;
        const int unitsY = interleaved ? mcusY : (sc[0]->height + 7) / 8;
        int untilRestart = restartInterval;

        auto block = [&](Component& k, int16_t* b) {
            if (!progressive) {
                const int s = br.decode(dc[k.td]);
                k.pred += s ? extend(br.receive(s), s) : 0;
                b[0] = int16_t(k.pred);
                for (int kk = 1; kk < 64;) {
                    const int rs = br.decode(ac[k.ta]);
                    const int r = rs >> 4, s2 = rs & 15;
                    if (s2 == 0) {
                        if (r != 15)
                            break;
                        kk += 16;
                        continue;
                    }
                    kk += r;
                    if (kk > 63)
                        break;
                    b[zigzag[kk]] = int16_t(extend(br.receive(s2), s2));
                    kk++;
                }
            } else if (Ss == 0) {
                if (Ah == 0) {
                    const int s = br.decode(dc[k.td]);
                    k.pred += s ? extend(br.receive(s), s) : 0;
                    b[0] = int16_t(k.pred * (1 << Al));
                } else if (br.bit()) {
                    b[0] |= int16_t(1 << Al);
                }
            } else if (Ah == 0) {
                if (eobrun > 0) {
                    eobrun--;
                    return;
                }
                for (int kk = Ss; kk <= Se; kk++) {
                    const int rs = br.decode(ac[k.ta]);
                    const int r = rs >> 4, s = rs & 15;
                    if (s) {
                        kk += r;
                        if (kk > 63)
                            break;
                        b[zigzag[kk]] = int16_t(extend(br.receive(s), s) * (1 << Al));
                    } else if (r == 15) {
                        kk += 15;
                    } else {
                        eobrun = 1 << r;
                        if (r)
                            eobrun += br.receive(r);
                        eobrun--;
                        break;
                    }
                }
            } else {
                const int p1 = 1 << Al, m1 = -(1 << Al);
                int kk = Ss;
                auto refine = [&](int16_t& c) {
                    if (br.bit() && (c & p1) == 0)
                        c = int16_t(c + (c >= 0 ? p1 : m1));
                };
                if (eobrun == 0) {
                    for (; kk <= Se; kk++) {
                        const int rs = br.decode(ac[k.ta]);
                        int r = rs >> 4, s = rs & 15;
                        if (s) {
                            s = br.bit() ? p1 : m1;
                        } else if (r != 15) {
                            eobrun = 1 << r;
                            if (r)
                                eobrun += br.receive(r);
                            break;
                        }
                        /* pass over coefficients with history and over r still-zero ones */
                        do {
                            int16_t& c = b[zigzag[kk]];
                            if (c != 0) {
                                refine(c);
                            } else if (--r < 0) {
                                break;
                            }
                            kk++;
                        } while (kk <= Se);
                        if (s && kk <= 63)
                            b[zigzag[kk]] = int16_t(s);
                    }
                }
                if (eobrun > 0) {
                    for (; kk <= Se; kk++) {
                        int16_t& c = b[zigzag[kk]];
                        if (c != 0)
                            refine(c);
                    }
                    eobrun--;
                }
            }
        };

        for (int uy = 0; uy < unitsY; uy++) {
            for (int ux = 0; ux < unitsX; ux++) {
                if (restartInterval && untilRestart == 0) {
                    if (!br.restart()) {
                        fail("missing restart marker");
                        return nullptr;
                    }
                    for (int i = 0; i < ns; i++)
                        sc[i]->pred = 0;
                    eobrun = 0;
                    untilRestart = restartInterval;
                }
                if (interleaved) {
                    for (int i = 0; i < ns; i++) {
                        Component& k = *sc[i];
                        for (int by = 0; by < k.v; by++)
                            for (int bx = 0; bx < k.h; bx++)
                                block(k, k.coef.data() + (size_t(uy * k.v + by) * k.blocksW + ux * k.h + bx) * 64);
                    }
                } else {
                    block(*sc[0], sc[0]->coef.data() + (size_t(uy) * sc[0]->blocksW + ux) * 64);
                }
                untilRestart--;
            }
        }
        /* the next marker follows the entropy-coded segment */
        const unsigned char* q = br.p;
        while (q + 1 < data + size && !(q[0] == 0xff && q[1] != 0 && q[1] != 0xff && !(q[1] >= 0xd0 && q[1] <= 0xd7)))
            q++;
        return q;
    }

    bool decodeCoefficients()
    {
        if (size < 4 || data[0] != 0xff || data[1] != 0xd8)
            return fail("not a JPEG file");
        const unsigned char* p = data + 2;
        const unsigned char* end = data + size;
        bool haveFrame = false, haveScan = false;
        while (p + 4 <= end) {
            if (p[0] != 0xff) {
                p++;
                continue;
            }
            const int m = p[1];
            if (m == 0xff) {
                p++;
                continue;
            }
            if (m == 0xd9)
                break;
            if (m == 0x01 || (m >= 0xd0 && m <= 0xd7)) {
                p += 2;
                continue;
            }
            const int len = be16(p + 2);
            if (len < 2 || p + 2 + len > end)
                return fail("truncated JPEG segment");
            const unsigned char* body = p + 4;
            const int blen = len - 2;
            if (m == 0xc0 || m == 0xc1 || m == 0xc2) {
                if (haveFrame)
                    return fail("more than one frame");
                progressive = m == 0xc2;
                if (!parseFrame(body, blen))
                    return false;
                haveFrame = true;
            } else if (m == 0xc3 || (m >= 0xc5 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc)) {
                return fail("lossless, hierarchical and arithmetic coded JPEG are not decoded");
            } else if (m == 0xc4) {
                if (!parseHuffman(body, blen))
                    return false;
            } else if (m == 0xdb) {
                if (!parseQuant(body, blen))
                    return false;
            } else if (m == 0xdd) {
                if (blen >= 2)
                    restartInterval = be16(body);
            } else if (m == 0xe0) {
                if (blen >= 5 && !memcmp(body, "JFIF", 5))
                    jfif = true;
            } else if (m == 0xee) {
                if (blen >= 12 && !memcmp(body, "Adobe", 5))
                    adobeTransform = body[11];
            } else if (m == 0xda) {
                if (!haveFrame)
                    return fail("scan before frame header");
                p = decodeScan(body, blen, p + 2 + len);
                if (!p)
                    return false;
                haveScan = true;
                continue;
            }
            p += 2 + len;
        }
        if (!haveFrame || !haveScan)
            return fail("JPEG without image data");
        return true;
    }

    void inverseTransform()
    {
        for (int c = 0; c < ncomp; c++) {
            Component& k = comp[c];
            k.plane.assign(size_t(k.blocksW) * 8 * k.blocksH * 8, 0);
            const int stride = k.blocksW * 8;
            for (int by = 0; by < k.blocksH; by++)
                for (int bx = 0; bx < k.blocksW; bx++)
                    idctIslow(k.coef.data() + (size_t(by) * k.blocksW + bx) * 64, quant[k.tq], k.plane.data() + size_t(by) * 8 * stride + bx * 8,
                            stride);
            k.coef.clear();
            k.coef.shrink_to_fit();
        }
    }

    /* jdsample.c: component plane -> full resolution (width x height) */
    std::vector<unsigned char> upsample(const Component& k) const
    {
        const int hx = hmax / k.h, vx = vmax / k.v;
        const int inStride = k.blocksW * 8;
        const int iw = k.width, ih = k.height;
        const int ow = iw * hx, oh = ih * vx; /* >= width, height */
        std::vector<unsigned char> out(size_t(ow) * oh);
        auto in = [&](int x, int y) -> int { return k.plane[size_t(y) * inStride + x]; };
        if (hx == 1 && vx == 1) {
            for (int y = 0; y < ih; y++)
                memcpy(out.data() + size_t(y) * ow, k.plane.data() + size_t(y) * inStride, iw);
        } else if (hx == 2 && vx == 1 && iw > 2) {
            /* h2v1_fancy_upsample: 3/4 nearer + 1/4 further, biased rounding that alternates */
            for (int y = 0; y < ih; y++) {
                unsigned char* o = out.data() + size_t(y) * ow;
                o[0] = (unsigned char)in(0, y);
                o[1] = (unsigned char)((in(0, y) * 3 + in(1, y) + 2) >> 2);
                for (int x = 1; x < iw - 1; x++) {
                    const int v = in(x, y) * 3;
                    o[2 * x] = (unsigned char)((v + in(x - 1, y) + 1) >> 2);
                    o[2 * x + 1] = (unsigned char)((v + in(x + 1, y) + 2) >> 2);
                }
                o[2 * iw - 2] = (unsigned char)((in(iw - 1, y) * 3 + in(iw - 2, y) + 1) >> 2);
                o[2 * iw - 1] = (unsigned char)in(iw - 1, y);
            }
        } else if (hx == 1 && vx == 2) {
            /* h1v2_fancy_upsample; rows above the first / below the last real row repeat it */
            for (int y = 0; y < ih; y++) {
                for (int v = 0; v < 2; v++) {
                    const int yn = v == 0 ? (y > 0 ? y - 1 : 0) : (y < ih - 1 ? y + 1 : ih - 1);
                    const int bias = v == 0 ? 1 : 2;
                    unsigned char* o = out.data() + size_t(2 * y + v) * ow;
                    for (int x = 0; x < iw; x++)
                        o[x] = (unsigned char)((in(x, y) * 3 + in(x, yn) + bias) >> 2);
                }
            }
        } else if (hx == 2 && vx == 2 && iw > 2) {
            /* h2v2_fancy_upsample: 9/16, 3/16, 3/16, 1/16 */
            for (int y = 0; y < ih; y++) {
                for (int v = 0; v < 2; v++) {
                    const int yn = v == 0 ? (y > 0 ? y - 1 : 0) : (y < ih - 1 ? y + 1 : ih - 1);
                    unsigned char* o = out.data() + size_t(2 * y + v) * ow;
                    int thiscolsum = in(0, y) * 3 + in(0, yn);
                    int nextcolsum = in(1, y) * 3 + in(1, yn);
                    o[0] = (unsigned char)((thiscolsum * 4 + 8) >> 4);
                    o[1] = (unsigned char)((thiscolsum * 3 + nextcolsum + 7) >> 4);
                    int lastcolsum = thiscolsum;
                    thiscolsum = nextcolsum;
                    for (int x = 1; x < iw - 1; x++) {
                        nextcolsum = in(x + 1, y) * 3 + in(x + 1, yn);
                        o[2 * x] = (unsigned char)((thiscolsum * 3 + lastcolsum + 8) >> 4);
                        o[2 * x + 1] = (unsigned char)((thiscolsum * 3 + nextcolsum + 7) >> 4);
                        lastcolsum = thiscolsum;
                        thiscolsum = nextcolsum;
                    }
                    o[2 * iw - 2] = (unsigned char)((thiscolsum * 3 + lastcolsum + 8) >> 4);
                    o[2 * iw - 1] = (unsigned char)((thiscolsum * 4 + 7) >> 4);
                }
            }
        } else {
            /* int_upsample / h2v1_upsample / h2v2_upsample: replication */
            for (int y = 0; y < oh; y++)
                for (int x = 0; x < ow; x++)
                    out[size_t(y) * ow + x] = (unsigned char)in(x / hx, y / vx);
        }
        if (ow != width || oh != height) {
            std::vector<unsigned char> cropped(size_t(width) * height);
            for (int y = 0; y < height; y++)
                memcpy(cropped.data() + size_t(y) * width, out.data() + size_t(y) * ow, width);
            return cropped;
        }
        return out;
    }

    bool decode(ArrayContainer& img)
    {
        if (!decodeCoefficients())
            return false;
        for (int c = 0; c < ncomp; c++)
            if (!quantDefined[comp[c].tq])
                return fail("quantization table missing");
        inverseTransform();
        img = ArrayContainer(width, height, ncomp, uint8);
        unsigned char* dst = static_cast<unsigned char*>(img.data());
        std::vector<unsigned char> full[3];
        for (int c = 0; c < ncomp; c++)
            full[c] = upsample(comp[c]);
        /* jdapimin.c default_decompress_parms: three components are YCbCr unless an Adobe marker says
         * "no transform" or, without JFIF / Adobe markers, the component ids spell R G B */
        bool ycc = ncomp == 3;
        if (ncomp == 3 && !jfif) {
            if (adobeTransform >= 0)
                ycc = adobeTransform != 0;
            else if (comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B')
                ycc = false;
        }
        auto clamp255 = [](int x) { return (unsigned char)(x < 0 ? 0 : x > 255 ? 255 : x); };
        for (int y = 0; y < height; y++) {
            unsigned char* row = dst + size_t(height - 1 - y) * width * ncomp; /* row 0 of the array is the bottom one */
            for (int x = 0; x < width; x++) {
                const size_t i = size_t(y) * width + x;
                if (ncomp == 1) {
                    row[x] = full[0][i];
                } else if (!ycc) {
                    row[3 * x] = full[0][i];
                    row[3 * x + 1] = full[1][i];
                    row[3 * x + 2] = full[2][i];
                } else {
                    /* jdcolor.c build_ycc_rgb_table / ycc_rgb_convert: 16 bit fixed point */
                    const int Y = full[0][i], cb = int(full[1][i]) - 128, cr = int(full[2][i]) - 128;
                    const int r = Y + int((int64_t(91881) * cr + 32768) >> 16);
                    const int g = Y + int((int64_t(-22554) * cb + 32768 + int64_t(-46802) * cr) >> 16);
                    const int b = Y + int((int64_t(116130) * cb + 32768) >> 16);
                    row[3 * x] = clamp255(r);
                    row[3 * x + 1] = clamp255(g);
                    row[3 * x + 2] = clamp255(b);
                }
            }
        }
        return true;
    }
};

}

/* Decodes a JPEG file in memory; row 0 of the result is the bottom row, as for the other decoders. */
inline bool loadJpeg(const std::vector<unsigned char>& bytes, ArrayContainer& img, std::string& error)
{
    jpegdetail::Decoder d(bytes.data(), bytes.size());
    if (!d.decode(img)) {
        error = d.error;
        return false;
    }
    return true;
}

}
