/*
 * imageio.hpp -- image file decoders for the importer: what the reference gets from libtgd's
 * TGD::load() (import.hpp:288-299, texture_image.hpp:352-372), an external library that is not
 * part of this build.  Decoded are the formats the scenes of the reference's examples use for
 * textures and environment maps and that need no further library: PNG (8 and 16 bit, grey, grey +
 * alpha, RGB, RGBA, palette; plain and Adam7 interlaced), TGA (types 2, 3, 10, 11; 8/24/32 bit), binary PNM
 * (P5, P6; 8 and 16 bit), PFM (Pf, PF), Radiance HDR (RLE and flat), JPEG (jpeg.hpp) and OpenEXR
 * scanline files (exr.hpp).  A file that cannot be decoded is reported and the importer substitutes
 * its dummy texture, as the reference does for any file libtgd cannot load (import.hpp:131-134).
 * saveImage() writes PNG, PGM/PPM, PFM, PFS, OpenEXR and this framework's raw array file (*.tgd).
 *
 * Convention: the returned array has row 0 at the BOTTOM of the picture (texture coordinate
 * v = 0, texture_image.hpp:85-212), i.e. formats that store the top row first are flipped.
 */
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "array.hpp"
#include "jpeg.hpp"
#include "exr.hpp"

namespace WurblPT {

namespace imagedetail {

inline bool readFile(const std::string& filename, std::vector<unsigned char>& bytes)
{
    FILE* f = fopen(filename.c_str(), "rb");
    if (!f)
        return false;
    unsigned char buf[65536];
    size_t got;
    while ((got = fread(buf, 1, sizeof(buf), f)) > 0)
        bytes.insert(bytes.end(), buf, buf + got);
    fclose(f);
    return true;
}

/* ---- zlib / deflate (RFC 1950, 1951) ---- */
struct BitReader {
    const unsigned char* p;
    size_t n, pos = 0;
    uint32_t bitBuf = 0;
    int bitCount = 0;
    bool bad = false;
    uint32_t bits(int count)
    {
        while (bitCount < count) {
            if (pos >= n) {
                bad = true;
                return 0;
            }
            bitBuf |= uint32_t(p[pos++]) << bitCount;
            bitCount += 8;
        }
        uint32_t v = bitBuf & ((count == 32) ? 0xffffffffu : ((1u << count) - 1u));
        bitBuf >>= count;
        bitCount -= count;
        return v;
    }
};

struct Huffman {
    uint16_t count[16], symbol[288];
    void build(const uint8_t* lengths, int n)
    {
        memset(count, 0, sizeof(count));
        for (int i = 0; i < n; i++)
            count[lengths[i]]++;
        count[0] = 0;
        uint16_t offs[16];
        offs[1] = 0;
        for (int i = 1; i < 15; i++)
            offs[i + 1] = offs[i] + count[i];
        for (int i = 0; i < n; i++)
            if (lengths[i] != 0)
                symbol[offs[lengths[i]]++] = uint16_t(i);
    }
    int decode(BitReader& br) const
    {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len <= 15; len++) {
            code |= int(br.bits(1));
            if (br.bad)
                return -1;
            int c = count[len];
            if (code - c < first)
                return symbol[index + (code - first)];
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
        }
        return -1;
    }
};

inline bool inflate(const unsigned char* src, size_t n, std::vector<unsigned char>& out)
{
    if (n < 2 || (src[0] & 0x0f) != 8 || ((src[0] << 8) | src[1]) % 31 != 0 || (src[1] & 0x20))
        return false;
    BitReader br { src + 2, n - 2 };
    static const uint16_t lenBase[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
    static const uint16_t lenExtra[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
    static const uint16_t distBase[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
    static const uint16_t distExtra[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };
    for (;;) {
        const uint32_t last = br.bits(1), type = br.bits(2);
        if (br.bad)
            return false;
        if (type == 0) {
            br.bitBuf = 0;
            br.bitCount = 0;
            if (br.pos + 4 > br.n)
                return false;
            const uint32_t len = br.p[br.pos] | (br.p[br.pos + 1] << 8), nlen = br.p[br.pos + 2] | (br.p[br.pos + 3] << 8);
            br.pos += 4;
            if ((len ^ 0xffffu) != nlen || br.pos + len > br.n)
                return false;
            out.insert(out.end(), br.p + br.pos, br.p + br.pos + len);
            br.pos += len;
        } else if (type == 1 || type == 2) {
            Huffman lit, dist;
            uint8_t lengths[320];
            if (type == 1) {
                for (int i = 0; i < 288; i++)
                    lengths[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
                lit.build(lengths, 288);
                for (int i = 0; i < 30; i++)
                    lengths[i] = 5;
                dist.build(lengths, 30);
            } else {
                const int nlen = int(br.bits(5)) + 257, ndist = int(br.bits(5)) + 1, ncode = int(br.bits(4)) + 4;
                static const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
                if (br.bad || nlen > 286 || ndist > 30)
                    return false;
                memset(lengths, 0, sizeof(lengths));
                for (int i = 0; i < ncode; i++)
                    lengths[order[i]] = uint8_t(br.bits(3));
                Huffman lencode;
                lencode.build(lengths, 19);
                int index = 0;
                while (index < nlen + ndist) {
                    int sym = lencode.decode(br);
                    if (sym < 0)
                        return false;
                    if (sym < 16) {
                        lengths[index++] = uint8_t(sym);
                    } else {
                        int prev = 0, rep;
                        if (sym == 16) {
                            if (index == 0)
                                return false;
                            prev = lengths[index - 1];
                            rep = 3 + int(br.bits(2));
                        } else if (sym == 17) {
                            rep = 3 + int(br.bits(3));
                        } else {
                            rep = 11 + int(br.bits(7));
                        }
                        if (br.bad || index + rep > nlen + ndist)
                            return false;
                        while (rep--)
                            lengths[index++] = uint8_t(prev);
                    }
                }
                lit.build(lengths, nlen);
                dist.build(lengths + nlen, ndist);
            }
            for (;;) {
                int sym = lit.decode(br);
                if (sym < 0)
                    return false;
                if (sym < 256) {
                    out.push_back((unsigned char)sym);
                } else if (sym == 256) {
                    break;
                } else {
                    sym -= 257;
                    if (sym >= 29)
                        return false;
                    const size_t len = lenBase[sym] + br.bits(lenExtra[sym]);
                    const int ds = dist.decode(br);
                    if (ds < 0 || ds >= 30)
                        return false;
                    const size_t d = distBase[ds] + br.bits(distExtra[ds]);
                    if (br.bad || d > out.size())
                        return false;
                    const size_t start = out.size() - d;
                    for (size_t i = 0; i < len; i++)
                        out.push_back(out[start + i]);
                }
            }
        } else {
            return false;
        }
        if (last)
            return true;
    }
}

inline uint32_t be32(const unsigned char* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }

/* ---- PNG ---- */
inline bool loadPng(const std::vector<unsigned char>& b, ArrayContainer& img, std::string& error)
{
    size_t pos = 8;
    uint32_t w = 0, h = 0;
    int depth = 0, colorType = 0, interlace = 0;
    std::vector<unsigned char> idat, palette, trns;
    while (pos + 12 <= b.size()) {
        const uint32_t len = be32(b.data() + pos);
        const unsigned char* type = b.data() + pos + 4;
        const unsigned char* data = b.data() + pos + 8;
        if (pos + 12 + len > b.size()) {
            error = "truncated PNG chunk";
            return false;
        }
        if (!memcmp(type, "IHDR", 4) && len >= 13) {
            w = be32(data);
            h = be32(data + 4);
            depth = data[8];
            colorType = data[9];
            interlace = data[12];
        } else if (!memcmp(type, "PLTE", 4)) {
            palette.assign(data, data + len);
        } else if (!memcmp(type, "tRNS", 4)) {
            trns.assign(data, data + len);
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + len;
    }
    if (w == 0 || h == 0 || interlace > 1 || !(depth == 8 || depth == 16 || (colorType == 3 && depth <= 8) || (colorType == 0 && depth < 8))) {
        error = "PNG variant not handled (unusual bit depth or interlace method)";
        return false;
    }
    const int channels = colorType == 0 ? 1 : colorType == 2 ? 3 : colorType == 3 ? 1 : colorType == 4 ? 2 : colorType == 6 ? 4 : 0;
    if (channels == 0) {
        error = "PNG colour type not handled";
        return false;
    }
    std::vector<unsigned char> raw;
    if (!inflate(idat.data(), idat.size(), raw)) {
        error = "PNG data cannot be inflated";
        return false;
    }
    const size_t bpp = size_t(channels * depth + 7) / 8; /* filter unit in bytes, at least 1 */
    const bool sixteen = depth == 16;
    const int outComps = colorType == 3 ? (trns.empty() ? 3 : 4) : channels;
    /* a damaged header must not make us allocate what the data cannot fill: every scanline has its filter byte */
    if (uint64_t(h) > raw.size() || (uint64_t(w) * channels * depth + 7) / 8 * uint64_t(h) > uint64_t(raw.size())) {
        error = "PNG data too short";
        return false;
    }
    img = ArrayContainer(w, h, outComps, sixteen ? uint16 : uint8);
    /* One reduced image: passW x passH pixels that go to (x0 + i dx, y0 + j dy).  A plain file is one such image over
     * all pixels, an interlaced one (Adam7) seven of them, each filtered on its own. */
    size_t at = 0;
    auto pass = [&](uint32_t passW, uint32_t passH, uint32_t x0, uint32_t dx, uint32_t y0, uint32_t dy) {
        if (passW == 0 || passH == 0)
            return true;
        const size_t stride = (size_t(passW) * channels * depth + 7) / 8; /* bytes per scanline */
        if (raw.size() < at + (stride + 1) * passH)
            return false;
        std::vector<unsigned char> prev(stride, 0), cur(stride);
        for (uint32_t j = 0; j < passH; j++) {
            const unsigned char* line = raw.data() + at + (stride + 1) * j;
            const int filter = line[0];
            for (size_t i = 0; i < stride; i++) {
                const int a = i >= bpp ? cur[i - bpp] : 0, bb = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
                int pred = 0;
                switch (filter) {
                case 1: pred = a; break;
                case 2: pred = bb; break;
                case 3: pred = (a + bb) >> 1; break;
                case 4: {
                    const int p = a + bb - c, pa = std::abs(p - a), pb = std::abs(p - bb), pc = std::abs(p - c);
                    pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? bb : c);
                    break;
                }
                default: break;
                }
                cur[i] = (unsigned char)(line[1 + i] + pred);
            }
            const size_t dstRow = h - 1 - (y0 + size_t(j) * dy); /* PNG stores the top row first */
            for (uint32_t i = 0; i < passW; i++) {
                const size_t x = x0 + size_t(i) * dx;
                if (sixteen) {
                    uint16_t* dst = img.get<uint16_t>(x, dstRow);
                    for (int c = 0; c < channels; c++)
                        dst[c] = uint16_t((cur[(size_t(i) * channels + c) * 2] << 8) | cur[(size_t(i) * channels + c) * 2 + 1]);
                } else if (colorType == 3) {
                    const unsigned int index = depth == 8 ? cur[i] : (cur[(size_t(i) * depth) / 8] >> (8 - depth - (i * depth) % 8)) & ((1u << depth) - 1u);
                    uint8_t* dst = img.get<uint8_t>(x, dstRow);
                    for (int c = 0; c < 3; c++)
                        dst[c] = 3 * index + c < palette.size() ? palette[3 * index + c] : 0;
                    if (outComps == 4)
                        dst[3] = index < trns.size() ? trns[index] : 255;
                } else if (depth < 8) {
                    const unsigned int v = (cur[(size_t(i) * depth) / 8] >> (8 - depth - (i * depth) % 8)) & ((1u << depth) - 1u);
                    img.get<uint8_t>(x, dstRow)[0] = uint8_t(v * 255u / ((1u << depth) - 1u));
                } else {
                    memcpy(img.get<uint8_t>(x, dstRow), cur.data() + size_t(i) * channels, channels);
                }
            }
            prev.swap(cur);
        }
        at += (stride + 1) * passH;
        return true;
    };
    bool ok = true;
    if (interlace == 0) {
        ok = pass(w, h, 0, 1, 0, 1);
    } else {
        static const uint32_t x0[7] = { 0, 4, 0, 2, 0, 1, 0 }, y0[7] = { 0, 0, 4, 0, 2, 0, 1 }, dx[7] = { 8, 8, 4, 4, 2, 2, 1 },
                              dy[7] = { 8, 8, 8, 4, 4, 2, 2 };
        for (int k = 0; k < 7 && ok; k++)
            ok = pass(w > x0[k] ? (w - x0[k] + dx[k] - 1) / dx[k] : 0, h > y0[k] ? (h - y0[k] + dy[k] - 1) / dy[k] : 0, x0[k], dx[k], y0[k], dy[k]);
    }
    if (!ok) {
        error = "PNG data too short";
        return false;
    }
    return true;
}

/* ---- TGA ---- */
inline bool loadTga(const std::vector<unsigned char>& b, ArrayContainer& img, std::string& error)
{
    if (b.size() < 18) {
        error = "truncated TGA header";
        return false;
    }
    const int idLen = b[0], cmapType = b[1], type = b[2], bits = b[16], desc = b[17];
    const uint32_t w = b[12] | (b[13] << 8), h = b[14] | (b[15] << 8);
    if (cmapType != 0 || !(type == 2 || type == 3 || type == 10 || type == 11) || !(bits == 8 || bits == 24 || bits == 32) || w == 0 || h == 0) {
        error = "TGA variant not handled";
        return false;
    }
    const int bytes = bits / 8;
    if (uint64_t(w) * h * bytes > 130ull * b.size()) { /* run-length packets expand at most 128-fold */
        error = "truncated TGA data";
        return false;
    }
    std::vector<unsigned char> px(size_t(w) * h * bytes);
    size_t pos = 18 + size_t(idLen);
    if (type == 2 || type == 3) {
        if (pos + px.size() > b.size()) {
            error = "truncated TGA data";
            return false;
        }
        memcpy(px.data(), b.data() + pos, px.size());
    } else {
        size_t o = 0;
        while (o < px.size()) {
            if (pos >= b.size()) {
                error = "truncated TGA data";
                return false;
            }
            const int head = b[pos++];
            const size_t count = size_t(head & 0x7f) + 1;
            if (head & 0x80) {
                if (pos + bytes > b.size() || o + count * bytes > px.size()) {
                    error = "bad TGA run";
                    return false;
                }
                for (size_t i = 0; i < count; i++, o += bytes)
                    memcpy(px.data() + o, b.data() + pos, bytes);
                pos += bytes;
            } else {
                if (pos + count * bytes > b.size() || o + count * bytes > px.size()) {
                    error = "bad TGA packet";
                    return false;
                }
                memcpy(px.data() + o, b.data() + pos, count * bytes);
                pos += count * bytes;
                o += count * bytes;
            }
        }
    }
    const bool topFirst = (desc & 0x20) != 0, rightFirst = (desc & 0x10) != 0;
    img = ArrayContainer(w, h, bytes, uint8);
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            const unsigned char* s = px.data() + (size_t(y) * w + x) * bytes;
            uint8_t* d = img.get<uint8_t>(rightFirst ? w - 1 - x : x, topFirst ? h - 1 - y : y);
            if (bytes == 1) {
                d[0] = s[0];
            } else { /* stored blue, green, red[, alpha] */
                d[0] = s[2];
                d[1] = s[1];
                d[2] = s[0];
                if (bytes == 4)
                    d[3] = s[3];
            }
        }
    return true;
}

/* token of a PNM / PFM header: skips white space and # comments */
inline bool headerToken(const std::vector<unsigned char>& b, size_t& pos, std::string& tok)
{
    tok.clear();
    for (;;) {
        while (pos < b.size() && (b[pos] == ' ' || b[pos] == '\t' || b[pos] == '\n' || b[pos] == '\r'))
            pos++;
        if (pos < b.size() && b[pos] == '#') {
            while (pos < b.size() && b[pos] != '\n')
                pos++;
            continue;
        }
        break;
    }
    while (pos < b.size() && !(b[pos] == ' ' || b[pos] == '\t' || b[pos] == '\n' || b[pos] == '\r'))
        tok += char(b[pos++]);
    return !tok.empty();
}

/* ---- binary PGM / PPM and PFM ---- */
inline bool loadPnm(const std::vector<unsigned char>& b, ArrayContainer& img, std::string& error)
{
    size_t pos = 0;
    std::string magic, sw, sh, smax;
    if (!headerToken(b, pos, magic) || !headerToken(b, pos, sw) || !headerToken(b, pos, sh) || !headerToken(b, pos, smax)) {
        error = "bad PNM header";
        return false;
    }
    pos++; /* the single white space after the header */
    const long w = atol(sw.c_str()), h = atol(sh.c_str());
    if (w <= 0 || h <= 0) {
        error = "bad PNM size";
        return false;
    }
    if (magic == "P5" || magic == "P6") {
        const int comps = magic == "P5" ? 1 : 3;
        const long maxval = atol(smax.c_str());
        const bool sixteen = maxval > 255;
        const size_t need = size_t(w) * h * comps * (sixteen ? 2 : 1);
        if (maxval <= 0 || maxval > 65535 || pos + need > b.size()) {
            error = "truncated PNM data";
            return false;
        }
        img = ArrayContainer(w, h, comps, sixteen ? uint16 : uint8);
        for (long y = 0; y < h; y++) {
            const unsigned char* s = b.data() + pos + size_t(y) * w * comps * (sixteen ? 2 : 1);
            const size_t dstRow = h - 1 - y; /* top row first in the file */
            if (sixteen) {
                uint16_t* d = img.get<uint16_t>(0, dstRow);
                for (long i = 0; i < w * comps; i++)
                    d[i] = uint16_t((s[2 * i] << 8) | s[2 * i + 1]);
            } else {
                memcpy(img.get<uint8_t>(0, dstRow), s, size_t(w) * comps);
            }
        }
        return true;
    }
    if (magic == "Pf" || magic == "PF") {
        const int comps = magic == "Pf" ? 1 : 3;
        const double scale = atof(smax.c_str());
        const size_t need = size_t(w) * h * comps * 4;
        if (scale == 0.0 || pos + need > b.size()) {
            error = "truncated PFM data";
            return false;
        }
        const bool little = scale < 0.0;
        img = ArrayContainer(w, h, comps, float32);
        float* d = img.get<float>(0); /* PFM stores the bottom row first */
        for (size_t i = 0; i < size_t(w) * h * comps; i++) {
            const unsigned char* s = b.data() + pos + 4 * i;
            uint32_t u = little ? (uint32_t(s[0]) | (uint32_t(s[1]) << 8) | (uint32_t(s[2]) << 16) | (uint32_t(s[3]) << 24)) : be32(s);
            memcpy(d + i, &u, 4);
        }
        return true;
    }
    error = "PNM variant not handled (only the binary P5, P6 and Pf, PF)";
    return false;
}

/* ---- Radiance HDR (RGBE) ---- */
inline bool loadHdr(const std::vector<unsigned char>& b, ArrayContainer& img, std::string& error)
{
    size_t pos = 0;
    auto line = [&](std::string& s) {
        s.clear();
        while (pos < b.size() && b[pos] != '\n')
            s += char(b[pos++]);
        pos++;
        return pos <= b.size();
    };
    std::string s;
    line(s);
    if (s.compare(0, 2, "#?") != 0) {
        error = "not a Radiance HDR file";
        return false;
    }
    while (line(s) && !s.empty())
        ;
    line(s);
    int w = 0, h = 0;
    if (sscanf(s.c_str(), "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0) {
        error = "HDR orientation not handled";
        return false;
    }
    if (uint64_t(w) * uint64_t(h) > 130ull * b.size()) { /* run-length coding expands at most 64-fold */
        error = "truncated HDR data";
        return false;
    }
    img = ArrayContainer(w, h, 3, float32);
    std::vector<unsigned char> scan(size_t(w) * 4);
    for (int y = 0; y < h; y++) {
        if (pos + 4 > b.size()) {
            error = "truncated HDR data";
            return false;
        }
        if (w >= 8 && w < 32768 && b[pos] == 2 && b[pos + 1] == 2 && (b[pos + 2] & 0x80) == 0) {
            pos += 4;
            for (int c = 0; c < 4; c++) {
                int x = 0;
                while (x < w) {
                    if (pos >= b.size()) {
                        error = "truncated HDR data";
                        return false;
                    }
                    int count = b[pos++];
                    if (count > 128) {
                        count -= 128;
                        if (pos >= b.size() || x + count > w) {
                            error = "bad HDR run";
                            return false;
                        }
                        const unsigned char v = b[pos++];
                        while (count--)
                            scan[size_t(x++) * 4 + c] = v;
                    } else {
                        if (count == 0 || pos + count > b.size() || x + count > w) {
                            error = "bad HDR packet";
                            return false;
                        }
                        while (count--)
                            scan[size_t(x++) * 4 + c] = b[pos++];
                    }
                }
            }
        } else {
            if (pos + size_t(w) * 4 > b.size()) {
                error = "truncated HDR data";
                return false;
            }
            memcpy(scan.data(), b.data() + pos, size_t(w) * 4);
            pos += size_t(w) * 4;
        }
        float* d = img.get<float>(0, size_t(h - 1 - y)); /* top scanline first in the file */
        for (int x = 0; x < w; x++) {
            const unsigned char* p = scan.data() + size_t(x) * 4;
            if (p[3] == 0) {
                d[3 * x] = d[3 * x + 1] = d[3 * x + 2] = 0.0f;
            } else {
                const float f = std::ldexp(1.0f, int(p[3]) - (128 + 8));
                d[3 * x] = (p[0] + 0.5f) * f;
                d[3 * x + 1] = (p[1] + 0.5f) * f;
                d[3 * x + 2] = (p[2] + 0.5f) * f;
            }
        }
    }
    return true;
}

}

namespace imagedetail {

inline uint32_t crc32(uint32_t crc, const unsigned char* p, size_t n)
{
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++)
                c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        ready = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; i++)
        crc = table[(crc ^ p[i]) & 0xffu] ^ (crc >> 8);
    return ~crc;
}

inline void putBe32(std::vector<unsigned char>& out, uint32_t v)
{
    out.push_back(v >> 24);
    out.push_back((v >> 16) & 0xffu);
    out.push_back((v >> 8) & 0xffu);
    out.push_back(v & 0xffu);
}

inline void pngChunk(std::vector<unsigned char>& out, const char* type, const std::vector<unsigned char>& payload)
{
    putBe32(out, uint32_t(payload.size()));
    const size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), payload.begin(), payload.end());
    putBe32(out, crc32(0, out.data() + start, out.size() - start));
}

/* PNG with stored (uncompressed) deflate blocks: every decoder reads it, no compressor needed. */
inline bool savePng(const ArrayContainer& img, std::vector<unsigned char>& out, std::string& error)
{
    const size_t w = img.dimension(0), h = img.dimension(1), comps = img.componentCount();
    if (w == 0 || h == 0 || comps < 1 || comps > 4 || (img.componentType() != uint8 && img.componentType() != uint16)) {
        error = "PNG takes 1-4 components of uint8 or uint16";
        return false;
    }
    const size_t cs = img.componentSize();
    static const unsigned char colorType[5] = { 0, 0, 4, 2, 6 };
    out.assign({ 0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n' });
    std::vector<unsigned char> ihdr;
    putBe32(ihdr, uint32_t(w));
    putBe32(ihdr, uint32_t(h));
    ihdr.push_back(cs == 1 ? 8 : 16);
    ihdr.push_back(colorType[comps]);
    ihdr.push_back(0);
    ihdr.push_back(0);
    ihdr.push_back(0);
    pngChunk(out, "IHDR", ihdr);
    /* scanlines top to bottom (array row 0 is the bottom one), filter type 0, samples big endian */
    const size_t rowBytes = w * comps * cs;
    std::vector<unsigned char> raw;
    raw.reserve((rowBytes + 1) * h);
    for (size_t y = 0; y < h; y++) {
        const unsigned char* row = static_cast<const unsigned char*>(img.data()) + (h - 1 - y) * rowBytes;
        raw.push_back(0);
        if (cs == 1) {
            raw.insert(raw.end(), row, row + rowBytes);
        } else {
            for (size_t i = 0; i < rowBytes; i += 2) {
                uint16_t v;
                memcpy(&v, row + i, 2);
                raw.push_back(v >> 8);
                raw.push_back(v & 0xffu);
            }
        }
    }
    std::vector<unsigned char> z;
    z.push_back(0x78);
    z.push_back(0x01);
    uint32_t s1 = 1, s2 = 0;
    for (size_t pos = 0; pos < raw.size() || pos == 0;) {
        const size_t n = raw.size() - pos < 65535 ? raw.size() - pos : 65535;
        z.push_back(pos + n == raw.size() ? 1 : 0);
        z.push_back(n & 0xffu);
        z.push_back(n >> 8);
        z.push_back(~n & 0xffu);
        z.push_back((~n >> 8) & 0xffu);
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        for (size_t i = 0; i < n; i++) {
            s1 = (s1 + raw[pos + i]) % 65521u;
            s2 = (s2 + s1) % 65521u;
        }
        pos += n;
        if (n == 0)
            break;
    }
    putBe32(z, (s2 << 16) | s1);
    pngChunk(out, "IDAT", z);
    pngChunk(out, "IEND", std::vector<unsigned char>());
    return true;
}

/* PGM/PPM for uint8/uint16 (1 or 3 components), PFM for float (1 or 3 components; bottom row first, little endian) */
inline bool savePnm(const ArrayContainer& img, std::vector<unsigned char>& out, std::string& error)
{
    const size_t w = img.dimension(0), h = img.dimension(1), comps = img.componentCount();
    if (w == 0 || h == 0 || (comps != 1 && comps != 3) || img.componentType() == int32) {
        error = "PNM/PFM takes 1 or 3 components of uint8, uint16 or float";
        return false;
    }
    const size_t rowBytes = w * img.elementSize();
    const unsigned char* data = static_cast<const unsigned char*>(img.data());
    char header[96];
    if (img.componentType() == float32) {
        snprintf(header, sizeof(header), "%s\n%zu %zu\n-1.0\n", comps == 3 ? "PF" : "Pf", w, h);
        out.assign(header, header + strlen(header));
        out.insert(out.end(), data, data + rowBytes * h);
        return true;
    }
    const bool wide = img.componentType() == uint16;
    snprintf(header, sizeof(header), "%s\n%zu %zu\n%d\n", comps == 3 ? "P6" : "P5", w, h, wide ? 65535 : 255);
    out.assign(header, header + strlen(header));
    for (size_t y = 0; y < h; y++) {
        const unsigned char* row = data + (h - 1 - y) * rowBytes;
        if (!wide) {
            out.insert(out.end(), row, row + rowBytes);
        } else {
            for (size_t i = 0; i < rowBytes; i += 2) {
                uint16_t v;
                memcpy(&v, row + i, 2);
                out.push_back(v >> 8);
                out.push_back(v & 0xffu);
            }
        }
    }
    return true;
}

/* Portable Floating-point Streams (pfstools), one frame: text header, then every channel as width x height floats,
 * top row first; channels are named X Y Z (three components), Y (one) or C0, C1, ... */
inline bool savePfs(const ArrayContainer& img, std::vector<unsigned char>& out, std::string& error)
{
    if (img.componentType() != float32 || img.componentCount() < 1) {
        error = "PFS takes float data";
        return false;
    }
    const size_t w = img.dimension(0), h = img.dimension(1), comps = img.componentCount();
    std::string header = "PFS1\n" + std::to_string(w) + " " + std::to_string(h) + "\n" + std::to_string(comps) + "\n0\n";
    static const char* xyz[3] = { "X", "Y", "Z" };
    for (size_t c = 0; c < comps; c++)
        header += (comps == 3 ? std::string(xyz[c]) : comps == 1 ? std::string("Y") : "C" + std::to_string(c)) + "\n0\n";
    header += "ENDH";
    out.assign(header.begin(), header.end());
    const float* data = static_cast<const float*>(img.data());
    for (size_t c = 0; c < comps; c++)
        for (size_t y = 0; y < h; y++)
            for (size_t x = 0; x < w; x++) {
                const float v = data[((h - 1 - y) * w + x) * comps + c];
                const unsigned char* p = reinterpret_cast<const unsigned char*>(&v);
                out.insert(out.end(), p, p + 4);
            }
    return true;
}

/* This framework's raw array file: the line "WPTARRAY1 <width> <height> <components> <type>\n" (type 0 uint8, 1 uint16,
 * 2 float32, 3 int32), then the data as it lies in memory (row 0 first, little endian) */
inline bool saveRaw(const ArrayContainer& img, std::vector<unsigned char>& out, std::string&)
{
    const std::string header = "WPTARRAY1 " + std::to_string(img.dimension(0)) + " " + std::to_string(img.dimension(1)) + " "
        + std::to_string(img.componentCount()) + " " + std::to_string(int(img.componentType())) + "\n";
    out.assign(header.begin(), header.end());
    const unsigned char* data = static_cast<const unsigned char*>(img.data());
    out.insert(out.end(), data, data + img.dataSize());
    return true;
}
inline bool loadRaw(const std::vector<unsigned char>& b, ArrayContainer& img, std::string& error)
{
    size_t eol = 0;
    while (eol < b.size() && eol < 128 && b[eol] != '\n')
        eol++;
    unsigned long long w = 0, h = 0, comps = 0;
    int type = -1;
    const std::string line(b.begin(), b.begin() + eol);
    if (eol >= b.size() || sscanf(line.c_str(), "WPTARRAY1 %llu %llu %llu %d", &w, &h, &comps, &type) != 4 || type < 0 || type > 3 || comps < 1
            || comps > 64 || w == 0 || h == 0 || w > (1ull << 24) || h > (1ull << 24)) {
        error = "damaged array file header";
        return false;
    }
    const size_t bytes = size_t(w) * size_t(h) * size_t(comps) * componentTypeSize(ComponentType(type));
    if (b.size() - (eol + 1) < bytes) {
        error = "array file is shorter than its header says";
        return false;
    }
    img = ArrayContainer(w, h, comps, ComponentType(type));
    memcpy(img.data(), b.data() + eol + 1, bytes);
    return true;
}

}

/* Saves an image by file name extension: .png (uint8/uint16), .ppm/.pgm/.pnm (uint8/uint16), .pfm / .pfs / .exr (float),
 * .tgd (any type, this framework's raw layout).
 * The counterpart of the reference's TGD::save() calls for the formats this build writes. */
inline bool saveImage(const ArrayContainer& img, const std::string& filename, std::string* error = nullptr)
{
    using namespace imagedetail;
    const size_t dot = filename.find_last_of('.');
    std::string ext = dot == std::string::npos ? "" : filename.substr(dot + 1);
    for (char& c : ext)
        c = char(tolower(c));
    std::vector<unsigned char> bytes;
    std::string err;
    bool ok = false;
    if (ext == "png")
        ok = savePng(img, bytes, err);
    else if (ext == "exr")
        ok = saveExr(img, bytes, err);
    else if (ext == "pfs")
        ok = savePfs(img, bytes, err);
    else if (ext == "tgd")
        ok = saveRaw(img, bytes, err);
    else if (ext == "pfm" && img.componentType() != float32)
        err = "PFM takes float data";
    else if ((ext == "ppm" || ext == "pgm" || ext == "pnm") && img.componentType() == float32)
        err = "PPM/PGM take uint8 or uint16 data";
    else if (ext == "pfm" || ext == "ppm" || ext == "pgm" || ext == "pnm")
        ok = savePnm(img, bytes, err);
    else
        err = "no writer for this file type";
    if (ok) {
        FILE* f = fopen(filename.c_str(), "wb");
        ok = f && fwrite(bytes.data(), 1, bytes.size(), f) == bytes.size();
        if (f && fclose(f) != 0)
            ok = false;
        if (!ok)
            err = "cannot write file";
    }
    if (!ok && error)
        *error = filename + ": " + err;
    return ok;
}

/* Loads an image file; an empty array (elementCount() == 0) and a message on failure. */
inline ArrayContainer loadImage(const std::string& filename, std::string* error = nullptr)
{
    using namespace imagedetail;
    std::string err;
    ArrayContainer img;
    std::vector<unsigned char> b;
    bool ok = false;
    try {
    if (!readFile(filename, b)) {
        err = "cannot open file";
    } else if (b.size() >= 8 && !memcmp(b.data(), "\x89PNG\r\n\x1a\n", 8)) {
        ok = loadPng(b, img, err);
    } else if (b.size() >= 2 && b[0] == 'P' && (b[1] == '5' || b[1] == '6' || b[1] == 'f' || b[1] == 'F')) {
        ok = loadPnm(b, img, err);
    } else if (b.size() >= 2 && b[0] == '#' && b[1] == '?') {
        ok = loadHdr(b, img, err);
    } else if (b.size() >= 4 && b[0] == 0x76 && b[1] == 0x2f && b[2] == 0x31 && b[3] == 0x01) {
        ok = loadExr(b, img, err);
    } else if (b.size() >= 3 && b[0] == 0xff && b[1] == 0xd8) {
        ok = loadJpeg(b, img, err);
    } else if (b.size() >= 10 && !memcmp(b.data(), "WPTARRAY1 ", 10)) {
        ok = loadRaw(b, img, err);
    } else {
        const size_t dot = filename.find_last_of('.');
        std::string ext = dot == std::string::npos ? "" : filename.substr(dot + 1);
        for (char& c : ext)
            c = char(tolower(c));
        if (ext == "tga")
            ok = loadTga(b, img, err);
        else
            err = "unknown image format";
    }
    } catch (const std::exception& e) {
        /* a damaged header can ask for more memory than there is */
        ok = false;
        err = std::string("cannot decode: ") + e.what();
    }
    if (!ok) {
        if (error)
            *error = filename + ": " + err;
        return ArrayContainer();
    }
    return img;
}

}
