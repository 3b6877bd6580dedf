/*
 * exr.hpp -- OpenEXR scanline files: the float frames the reference's applications save through
 * TGD::save("*.exr") and the HDR images they load.
 *
 * Read: single-part scanline images, channels of type HALF / FLOAT / UINT (converted to float),
 * compression NONE, RLE, ZIPS and ZIP; channels R G B (A) or Y become the components of the array
 * in that order, other layouts keep the file's (alphabetical) channel order.  Tiled, multi-part,
 * deep and PIZ / PXR24 / B44 / DWA files are refused with a message.
 * Write: FLOAT channels, no compression (every reader accepts it, values are kept bit for bit).
 *
 * Row 0 of an array is the bottom row of the picture; EXR stores the top scanline first.
 */
#pragma once

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "array.hpp"

namespace WurblPT {

namespace imagedetail {

inline bool inflate(const unsigned char* src, size_t n, std::vector<unsigned char>& out); /* imageio.hpp */

inline float halfToFloat(uint16_t h)
{
    const uint32_t sign = uint32_t(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu, man = h & 0x3ffu, bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else { /* subnormal: normalise */
            exp = 127 - 15 + 1;
            while (!(man & 0x400u)) {
                man <<= 1;
                exp--;
            }
            bits = sign | (exp << 23) | ((man & 0x3ffu) << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7f800000u | (man << 13);
    } else {
        bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

inline uint32_t le32(const unsigned char* p) { return uint32_t(p[0]) | (uint32_t(p[1]) << 8) | (uint32_t(p[2]) << 16) | (uint32_t(p[3]) << 24); }
inline uint64_t le64(const unsigned char* p) { return uint64_t(le32(p)) | (uint64_t(le32(p + 4)) << 32); }

struct ExrChannel {
    std::string name;
    int type; /* 0 UINT, 1 HALF, 2 FLOAT */
};

inline bool loadExr(const std::vector<unsigned char>& b, ArrayContainer& img, std::string& error)
{
    auto fail = [&](const char* msg) {
        error = msg;
        return false;
    };
    if (b.size() < 12 || le32(b.data()) != 20000630u)
        return fail("not an OpenEXR file");
    const uint32_t version = le32(b.data() + 4);
    if ((version & 0xffu) != 2 || (version & (0x200u | 0x800u | 0x1000u)))
        return fail("only single-part scanline OpenEXR files are decoded");
    size_t pos = 8;
    std::vector<ExrChannel> channels;
    int compression = -1;
    int32_t win[4] = { 0, 0, -1, -1 };
    int lineOrder = 0;
    auto cstring = [&](std::string& s) {
        const size_t start = pos;
        while (pos < b.size() && b[pos] != 0)
            pos++;
        if (pos >= b.size())
            return false;
        s.assign(reinterpret_cast<const char*>(b.data() + start), pos - start);
        pos++;
        return true;
    };
    for (;;) {
        if (pos >= b.size())
            return fail("truncated OpenEXR header");
        if (b[pos] == 0) {
            pos++;
            break;
        }
        std::string name, type;
        if (!cstring(name) || !cstring(type) || pos + 4 > b.size())
            return fail("truncated OpenEXR header");
        const uint32_t size = le32(b.data() + pos);
        pos += 4;
        if (pos + size > b.size())
            return fail("truncated OpenEXR header");
        const unsigned char* v = b.data() + pos;
        if (name == "channels") {
            size_t q = 0;
            while (q < size && v[q] != 0) {
                ExrChannel c;
                while (q < size && v[q] != 0)
                    c.name.push_back(char(v[q++]));
                q++;
                if (q + 16 > size)
                    return fail("bad OpenEXR channel list");
                c.type = int(le32(v + q));
                if (le32(v + q + 8) != 1 || le32(v + q + 12) != 1)
                    return fail("subsampled OpenEXR channels are not decoded");
                q += 16;
                if (c.type < 0 || c.type > 2)
                    return fail("bad OpenEXR channel type");
                channels.push_back(c);
            }
        } else if (name == "compression" && size >= 1) {
            compression = v[0];
        } else if (name == "dataWindow" && size >= 16) {
            for (int i = 0; i < 4; i++)
                win[i] = int32_t(le32(v + 4 * i));
        } else if (name == "lineOrder" && size >= 1) {
            lineOrder = v[0];
        }
        pos += size;
    }
    if (channels.empty() || channels.size() > 64 || compression < 0 || win[2] < win[0] || win[3] < win[1])
        return fail("incomplete OpenEXR header");
    if (compression > 3)
        return fail("this OpenEXR compression (PIZ, PXR24, B44, DWA) is not decoded; use NONE, RLE, ZIPS or ZIP");
    (void)lineOrder; /* chunks carry their y coordinate, so the order in the file does not matter */
    const size_t width = size_t(int64_t(win[2]) - win[0] + 1), height = size_t(int64_t(win[3]) - win[1] + 1);
    if (width > 65536 || height > 65536)
        return fail("OpenEXR image too large");
    /* channel -> component */
    std::vector<int> componentOf(channels.size(), -1);
    auto find = [&](const char* n) {
        for (size_t i = 0; i < channels.size(); i++)
            if (channels[i].name == n)
                return int(i);
        return -1;
    };
    size_t comps = 0;
    const int r = find("R"), g = find("G"), bl = find("B"), a = find("A"), y = find("Y");
    if (r >= 0 && g >= 0 && bl >= 0) {
        componentOf[r] = 0;
        componentOf[g] = 1;
        componentOf[bl] = 2;
        comps = 3;
        if (a >= 0)
            componentOf[a] = int(comps++);
    } else if (y >= 0) {
        componentOf[y] = 0;
        comps = 1;
        if (a >= 0)
            componentOf[a] = int(comps++);
    } else {
        for (size_t i = 0; i < channels.size() && i < 4; i++)
            componentOf[i] = int(comps++);
    }
    size_t bytesPerLine = 0;
    for (const ExrChannel& c : channels)
        bytesPerLine += width * (c.type == 1 ? 2 : 4);
    const size_t linesPerChunk = compression == 3 ? 16 : 1;
    const size_t chunks = (height + linesPerChunk - 1) / linesPerChunk;
    if (pos + chunks * 8 > b.size())
        return fail("truncated OpenEXR offset table");
    if (uint64_t(bytesPerLine) * height > 1100ull * b.size() + 65536ull) /* zlib expands at most 1032-fold */
        return fail("truncated OpenEXR data");
    img = ArrayContainer(width, height, comps, float32);
    float* dst = static_cast<float*>(img.data());
    std::vector<unsigned char> raw, tmp;
    for (size_t chunk = 0; chunk < chunks; chunk++) {
        const uint64_t off = le64(b.data() + pos + 8 * chunk);
        if (off + 8 > b.size())
            return fail("bad OpenEXR chunk offset");
        const int64_t y0 = int64_t(int32_t(le32(b.data() + off))) - win[1];
        const size_t dataSize = le32(b.data() + off + 4);
        if (y0 < 0 || size_t(y0) >= height || off + 8 + dataSize > b.size())
            return fail("bad OpenEXR chunk");
        const size_t lines = size_t(y0) + linesPerChunk <= height ? linesPerChunk : height - size_t(y0);
        const size_t expect = lines * bytesPerLine;
        const unsigned char* src = b.data() + off + 8;
        if (compression == 0 || dataSize == expect) {
            if (dataSize != expect)
                return fail("bad OpenEXR chunk size");
            raw.assign(src, src + dataSize);
        } else {
            tmp.clear();
            if (compression == 1) {
                for (size_t i = 0; i < dataSize;) {
                    const int count = int(int8_t(src[i++]));
                    if (count < 0) {
                        if (i + size_t(-count) > dataSize)
                            return fail("bad OpenEXR run");
                        tmp.insert(tmp.end(), src + i, src + i + size_t(-count));
                        i += size_t(-count);
                    } else {
                        if (i >= dataSize)
                            return fail("bad OpenEXR run");
                        tmp.insert(tmp.end(), size_t(count) + 1, src[i++]);
                    }
                }
            } else if (!inflate(src, dataSize, tmp)) {
                return fail("bad OpenEXR zlib data");
            }
            if (tmp.size() != expect)
                return fail("OpenEXR chunk decompresses to the wrong size");
            /* undo the predictor, then the split into even and odd bytes */
            for (size_t i = 1; i < tmp.size(); i++)
                tmp[i] = (unsigned char)(tmp[i - 1] + tmp[i] - 128);
            raw.resize(expect);
            const size_t half = (expect + 1) / 2;
            for (size_t i = 0; i < expect; i++)
                raw[i] = (i & 1) ? tmp[half + i / 2] : tmp[i / 2];
        }
        const unsigned char* p = raw.data();
        for (size_t line = 0; line < lines; line++) {
            float* row = dst + (height - 1 - (size_t(y0) + line)) * width * comps;
            for (size_t c = 0; c < channels.size(); c++) {
                const int k = componentOf[c];
                const int type = channels[c].type;
                if (k >= 0) {
                    for (size_t x = 0; x < width; x++) {
                        float f;
                        if (type == 1) {
                            f = halfToFloat(uint16_t(p[2 * x] | (p[2 * x + 1] << 8)));
                        } else if (type == 2) {
                            const uint32_t u = le32(p + 4 * x);
                            memcpy(&f, &u, 4);
                        } else {
                            f = float(le32(p + 4 * x));
                        }
                        row[x * comps + size_t(k)] = f;
                    }
                }
                p += width * (type == 1 ? 2 : 4);
            }
        }
    }
    return true;
}

inline void exrAttribute(std::vector<unsigned char>& out, const char* name, const char* type, const void* data, uint32_t size)
{
    out.insert(out.end(), name, name + strlen(name) + 1);
    out.insert(out.end(), type, type + strlen(type) + 1);
    for (int i = 0; i < 4; i++)
        out.push_back((size >> (8 * i)) & 0xffu);
    const unsigned char* p = static_cast<const unsigned char*>(data);
    out.insert(out.end(), p, p + size);
}

/* 1 component -> Y, 2 -> Y A, 3 -> R G B, 4 -> R G B A; FLOAT, uncompressed, top scanline first.
 * (The host is little endian, as every machine this framework runs on.) */
inline bool saveExr(const ArrayContainer& img, std::vector<unsigned char>& out, std::string& error)
{
    const size_t w = img.dimension(0), h = img.dimension(1), comps = img.componentCount();
    if (w == 0 || h == 0 || comps < 1 || comps > 4 || img.componentType() != float32) {
        error = "OpenEXR output takes 1-4 float components";
        return false;
    }
    static const char* names[4][4] = { { "Y" }, { "A", "Y" }, { "B", "G", "R" }, { "A", "B", "G", "R" } };   /* alphabetical: the file order */
    static const int source[4][4] = { { 0 }, { 1, 0 }, { 2, 1, 0 }, { 3, 2, 1, 0 } };
    out.clear();
    const uint32_t magic = 20000630u, version = 2;
    out.insert(out.end(), reinterpret_cast<const unsigned char*>(&magic), reinterpret_cast<const unsigned char*>(&magic) + 4);
    out.insert(out.end(), reinterpret_cast<const unsigned char*>(&version), reinterpret_cast<const unsigned char*>(&version) + 4);
    std::vector<unsigned char> chlist;
    for (size_t c = 0; c < comps; c++) {
        const char* n = names[comps - 1][c];
        chlist.insert(chlist.end(), n, n + strlen(n) + 1);
        const uint32_t fields[4] = { 2u /* FLOAT */, 0u /* pLinear + reserved */, 1u, 1u };
        const unsigned char* f = reinterpret_cast<const unsigned char*>(fields);
        chlist.insert(chlist.end(), f, f + 16);
    }
    chlist.push_back(0);
    exrAttribute(out, "channels", "chlist", chlist.data(), uint32_t(chlist.size()));
    const unsigned char none = 0;
    exrAttribute(out, "compression", "compression", &none, 1);
    const int32_t window[4] = { 0, 0, int32_t(w) - 1, int32_t(h) - 1 };
    exrAttribute(out, "dataWindow", "box2i", window, 16);
    exrAttribute(out, "displayWindow", "box2i", window, 16);
    exrAttribute(out, "lineOrder", "lineOrder", &none, 1);
    const float one = 1.0f, centre[2] = { 0.0f, 0.0f };
    exrAttribute(out, "pixelAspectRatio", "float", &one, 4);
    exrAttribute(out, "screenWindowCenter", "v2f", centre, 8);
    exrAttribute(out, "screenWindowWidth", "float", &one, 4);
    out.push_back(0);
    const size_t lineBytes = w * comps * 4;
    const uint64_t first = out.size() + 8 * h;
    for (size_t y = 0; y < h; y++) {
        const uint64_t off = first + y * (8 + lineBytes);
        out.insert(out.end(), reinterpret_cast<const unsigned char*>(&off), reinterpret_cast<const unsigned char*>(&off) + 8);
    }
    std::vector<float> line(w * comps);
    for (size_t y = 0; y < h; y++) {
        const float* row = static_cast<const float*>(img.data()) + (h - 1 - y) * w * comps;
        for (size_t c = 0; c < comps; c++)
            for (size_t x = 0; x < w; x++)
                line[c * w + x] = row[x * comps + size_t(source[comps - 1][c])];
        const int32_t yy = int32_t(y), size = int32_t(lineBytes);
        out.insert(out.end(), reinterpret_cast<const unsigned char*>(&yy), reinterpret_cast<const unsigned char*>(&yy) + 4);
        out.insert(out.end(), reinterpret_cast<const unsigned char*>(&size), reinterpret_cast<const unsigned char*>(&size) + 4);
        out.insert(out.end(), reinterpret_cast<const unsigned char*>(line.data()), reinterpret_cast<const unsigned char*>(line.data()) + lineBytes);
    }
    return true;
}

}

}
