/*
 * objreader.hpp -- Wavefront OBJ / MTL reader for the importer (import.hpp).
 *
 * The reference parses OBJ files with the tinyobjloader it vendors (tiny_obj_loader.h, version
 * 2.0.0rc) and the hitable order of an imported scene -- hence the BVH and the rendered bits --
 * follows from what that parser produces: which shapes exist, how polygons are split into
 * triangles, what a decimal string becomes as a float.  This reader produces the same arrays for
 * the statements the importer consumes (v, vn, vt, f, g, o, usemtl, mtllib; newmtl, Ka Kd Ks Ke
 * Kt/Tf Ni Ns d Tr illum, map_Kd map_Ks map_Ns map_d map_Ke, map_bump / bump, norm, with the
 * texture options), including its number parser (tiny_obj_loader.h:897-1028: decimal digits
 * accumulated in double, exponent applied as 5^e * 2^e), its triangulation (quads along the shorter
 * diagonal, :1510-1620; larger polygons by its ear clipping in the dominant plane, :1740-1975) and
 * its shape rules (a shape ends at `g` and `o`; `usemtl` changes the material inside a shape,
 * :2877-2996).  tests/test_import.py compares it with the vendored parser itself
 * (oracle/ref_probe.cpp) on the fixture files of tests/golden/obj.
 *
 * The behaviour reproduced here (number parsing, triangulation, shape and material rules) is tinyobjloader's, which
 * is distributed under the MIT licence: Copyright (c) 2012-Present, Syoyo Fujita and many contributors; the permission
 * notice is reproduced in the LICENSE file of this repository.  The code is this repository's own.
 */
#pragma once

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <map>
#include <set>
#include <sstream>
#include <string>
#include <vector>

namespace WurblPT {

struct ObjIndex {
    int vertex = -1, normal = -1, texcoord = -1;
};

struct ObjShape {
    std::string name;
    std::vector<ObjIndex> indices; /* three per triangle */
    std::vector<int> materialIds;  /* one per triangle, -1 = none */
};

struct ObjTexOpt {
    float scale[3] = { 1.0f, 1.0f, 1.0f };
    float originOffset[3] = { 0.0f, 0.0f, 0.0f };
    float bumpMultiplier = 1.0f;
};

struct ObjMaterial {
    std::string name;
    float ambient[3] = { 0, 0, 0 }, diffuse[3] = { 0, 0, 0 }, specular[3] = { 0, 0, 0 };
    float transmittance[3] = { 0, 0, 0 }, emission[3] = { 0, 0, 0 };
    float shininess = 1.0f, ior = 1.0f, dissolve = 1.0f;
    int illum = 0;
    std::string diffuseTex, specularTex, shininessTex, bumpTex, alphaTex, emissiveTex, normalTex;
    ObjTexOpt diffuseOpt, specularOpt, shininessOpt, bumpOpt, alphaOpt, emissiveOpt, normalOpt;
};

struct ObjData {
    std::vector<float> vertices, normals, texcoords;
    std::vector<ObjShape> shapes;
    std::vector<ObjMaterial> materials;
    std::string warning, error;
};

namespace objdetail {

inline bool isSpace(char c) { return c == ' ' || c == '\t'; }
inline bool isDigit(char c) { return (unsigned int)(c - '0') < 10u; }
inline bool isNewLine(char c) { return c == '\r' || c == '\n' || c == '\0'; }

/* tiny_obj_loader.h:897-1028 */
inline bool tryParseDouble(const char* s, const char* sEnd, double* result)
{
    if (s >= sEnd)
        return false;
    double mantissa = 0.0;
    int exponent = 0;
    char sign = '+', expSign = '+';
    const char* curr = s;
    int read = 0;
    bool endNotReached = false, leadingDot = false;
    if (*curr == '+' || *curr == '-') {
        sign = *curr;
        curr++;
        if (curr != sEnd && *curr == '.')
            leadingDot = true;
    } else if (isDigit(*curr)) {
    } else if (*curr == '.') {
        leadingDot = true;
    } else {
        return false;
    }
    endNotReached = (curr != sEnd);
    if (!leadingDot) {
        while (endNotReached && isDigit(*curr)) {
            mantissa *= 10;
            mantissa += int(*curr - 0x30);
            curr++;
            read++;
            endNotReached = (curr != sEnd);
        }
        if (read == 0)
            return false;
    }
    bool haveExponent = false;
    if (endNotReached) {
        bool goOn = true;
        if (*curr == '.') {
            curr++;
            read = 1;
            endNotReached = (curr != sEnd);
            static const double powLut[] = { 1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001 };
            const int lutEntries = sizeof(powLut) / sizeof(powLut[0]);
            while (endNotReached && isDigit(*curr)) {
                mantissa += int(*curr - 0x30) * (read < lutEntries ? powLut[read] : std::pow(10.0, -read));
                read++;
                curr++;
                endNotReached = (curr != sEnd);
            }
        } else if (*curr == 'e' || *curr == 'E') {
        } else {
            goOn = false;
        }
        if (goOn && endNotReached && (*curr == 'e' || *curr == 'E')) {
            curr++;
            endNotReached = (curr != sEnd);
            if (endNotReached && (*curr == '+' || *curr == '-')) {
                expSign = *curr;
                curr++;
            } else if (isDigit(*curr)) {
            } else {
                return false;
            }
            read = 0;
            endNotReached = (curr != sEnd);
            while (endNotReached && isDigit(*curr)) {
                if (exponent > (2147483647 / 10))
                    return false;
                exponent *= 10;
                exponent += int(*curr - 0x30);
                curr++;
                read++;
                endNotReached = (curr != sEnd);
            }
            exponent *= (expSign == '+' ? 1 : -1);
            if (read == 0)
                return false;
            haveExponent = true;
        }
    }
    (void)haveExponent;
    *result = (sign == '+' ? 1 : -1) * (exponent ? std::ldexp(mantissa * std::pow(5.0, exponent), exponent) : mantissa);
    return true;
}

inline float parseReal(const char** token, double defaultValue = 0.0)
{
    (*token) += strspn((*token), " \t");
    const char* end = (*token) + strcspn((*token), " \t\r");
    double val = defaultValue;
    tryParseDouble((*token), end, &val);
    (*token) = end;
    return float(val);
}

inline int parseInt(const char** token)
{
    (*token) += strspn((*token), " \t");
    int i = atoi((*token));
    (*token) += strcspn((*token), " \t\r");
    return i;
}

inline std::string parseString(const char** token)
{
    (*token) += strspn((*token), " \t");
    size_t e = strcspn((*token), " \t\r");
    std::string s((*token), (*token) + e);
    (*token) += e;
    return s;
}

/* tiny_obj_loader.h:819-850: one-based -> zero-based, negative = relative */
inline bool fixIndex(int idx, int n, int* ret, bool allowZero)
{
    if (idx > 0) {
        *ret = idx - 1;
        return true;
    }
    if (idx == 0) {
        *ret = idx - 1;
        return allowZero;
    }
    *ret = n + idx;
    return *ret >= 0;
}

struct VertexIndex {
    int v = -1, vt = -1, vn = -1;
};

/* i, i/j, i//k, i/j/k (tiny_obj_loader.h:1188-1239) */
inline bool parseTriple(const char** token, int vsize, int vnsize, int vtsize, VertexIndex* ret)
{
    VertexIndex vi;
    if (!fixIndex(atoi((*token)), vsize, &vi.v, false))
        return false;
    (*token) += strcspn((*token), "/ \t\r");
    if ((*token)[0] != '/') {
        *ret = vi;
        return true;
    }
    (*token)++;
    if ((*token)[0] == '/') {
        (*token)++;
        if (!fixIndex(atoi((*token)), vnsize, &vi.vn, true))
            return false;
        (*token) += strcspn((*token), "/ \t\r");
        *ret = vi;
        return true;
    }
    if (!fixIndex(atoi((*token)), vtsize, &vi.vt, true))
        return false;
    (*token) += strcspn((*token), "/ \t\r");
    if ((*token)[0] != '/') {
        *ret = vi;
        return true;
    }
    (*token)++;
    if (!fixIndex(atoi((*token)), vnsize, &vi.vn, true))
        return false;
    (*token) += strcspn((*token), "/ \t\r");
    *ret = vi;
    return true;
}

/* a line without its terminator; "\r\n", "\n" and a lone "\r" end a line (tiny_obj_loader.h:767-799) */
inline bool getLine(std::istream& is, std::string& t)
{
    t.clear();
    if (is.peek() == std::char_traits<char>::eof())
        return false;
    std::streambuf* sb = is.rdbuf();
    for (;;) {
        int c = sb->sbumpc();
        if (c == '\n')
            return true;
        if (c == '\r') {
            if (sb->sgetc() == '\n')
                sb->sbumpc();
            return true;
        }
        if (c == std::char_traits<char>::eof()) {
            if (t.empty())
                is.setstate(std::ios::eofbit);
            return true;
        }
        t += char(c);
    }
}

inline ObjIndex toIndex(const VertexIndex& vi)
{
    ObjIndex i;
    i.vertex = vi.v;
    i.normal = vi.vn;
    i.texcoord = vi.vt;
    return i;
}

/* pnpoly (tiny_obj_loader.h:1438-1450) */
inline int pnpoly(int nvert, const float* vertx, const float* verty, float testx, float testy)
{
    int i, j, c = 0;
    for (i = 0, j = nvert - 1; i < nvert; j = i++) {
        if (((verty[i] > testy) != (verty[j] > testy))
                && (testx < (vertx[j] - vertx[i]) * (testy - verty[i]) / (verty[j] - verty[i]) + vertx[i]))
            c = !c;
    }
    return c;
}

/* exportGroupsToShape, triangulating (tiny_obj_loader.h:1481-1990) */
inline bool exportFaces(ObjShape& shape, std::vector<std::vector<VertexIndex>>& faces, int materialId, const std::string& name,
        const std::vector<float>& v, std::string& warn)
{
    if (faces.empty())
        return false;
    shape.name = name;
    auto push = [&](const VertexIndex& a, const VertexIndex& b, const VertexIndex& c) {
        shape.indices.push_back(toIndex(a));
        shape.indices.push_back(toIndex(b));
        shape.indices.push_back(toIndex(c));
        shape.materialIds.push_back(materialId);
    };
    for (const std::vector<VertexIndex>& face : faces) {
        size_t npolys = face.size();
        if (npolys < 3) {
            warn += "Degenerated face found\n.";
            continue;
        }
        if (npolys == 3) {
            push(face[0], face[1], face[2]);
        } else if (npolys == 4) {
            const size_t vi0 = size_t(face[0].v), vi1 = size_t(face[1].v), vi2 = size_t(face[2].v), vi3 = size_t(face[3].v);
            if ((3 * vi0 + 2) >= v.size() || (3 * vi1 + 2) >= v.size() || (3 * vi2 + 2) >= v.size() || (3 * vi3 + 2) >= v.size()) {
                warn += "Face with invalid vertex index found.\n";
                continue;
            }
            const float e02x = v[vi2 * 3 + 0] - v[vi0 * 3 + 0], e02y = v[vi2 * 3 + 1] - v[vi0 * 3 + 1], e02z = v[vi2 * 3 + 2] - v[vi0 * 3 + 2];
            const float e13x = v[vi3 * 3 + 0] - v[vi1 * 3 + 0], e13y = v[vi3 * 3 + 1] - v[vi1 * 3 + 1], e13z = v[vi3 * 3 + 2] - v[vi1 * 3 + 2];
            const float sqr02 = e02x * e02x + e02y * e02y + e02z * e02z;
            const float sqr13 = e13x * e13x + e13y * e13y + e13z * e13z;
            if (sqr02 < sqr13) {
                push(face[0], face[1], face[2]);
                push(face[0], face[2], face[3]);
            } else {
                push(face[0], face[1], face[3]);
                push(face[1], face[2], face[3]);
            }
        } else {
            /* the two axes of the plane to work in: from the first corner that is not degenerate */
            size_t axes[2] = { 1, 2 };
            for (size_t k = 0; k < npolys; ++k) {
                const size_t vi0 = size_t(face[(k + 0) % npolys].v), vi1 = size_t(face[(k + 1) % npolys].v), vi2 = size_t(face[(k + 2) % npolys].v);
                if ((3 * vi0 + 2) >= v.size() || (3 * vi1 + 2) >= v.size() || (3 * vi2 + 2) >= v.size())
                    continue;
                const float e0x = v[vi1 * 3 + 0] - v[vi0 * 3 + 0], e0y = v[vi1 * 3 + 1] - v[vi0 * 3 + 1], e0z = v[vi1 * 3 + 2] - v[vi0 * 3 + 2];
                const float e1x = v[vi2 * 3 + 0] - v[vi1 * 3 + 0], e1y = v[vi2 * 3 + 1] - v[vi1 * 3 + 1], e1z = v[vi2 * 3 + 2] - v[vi1 * 3 + 2];
                const float cx = std::fabs(e0y * e1z - e0z * e1y);
                const float cy = std::fabs(e0z * e1x - e0x * e1z);
                const float cz = std::fabs(e0x * e1y - e0y * e1x);
                const float epsilon = std::numeric_limits<float>::epsilon();
                if (cx > epsilon || cy > epsilon || cz > epsilon) {
                    if (!(cx > cy && cx > cz)) {
                        axes[0] = 0;
                        if (cz > cx && cz > cy)
                            axes[1] = 1;
                    }
                    break;
                }
            }
            std::vector<VertexIndex> remaining = face;
            size_t guessVert = 0;
            VertexIndex ind[3];
            float vx[3], vy[3];
            size_t remainingIterations = face.size();
            size_t previousRemainingVertices = remaining.size();
            while (remaining.size() > 3 && remainingIterations > 0) {
                npolys = remaining.size();
                if (guessVert >= npolys)
                    guessVert -= npolys;
                if (previousRemainingVertices != npolys) {
                    previousRemainingVertices = npolys;
                    remainingIterations = npolys;
                } else {
                    remainingIterations--;
                }
                for (size_t k = 0; k < 3; k++) {
                    ind[k] = remaining[(guessVert + k) % npolys];
                    const size_t vi = size_t(ind[k].v);
                    if ((vi * 3 + axes[0]) >= v.size() || (vi * 3 + axes[1]) >= v.size()) {
                        vx[k] = 0.0f;
                        vy[k] = 0.0f;
                    } else {
                        vx[k] = v[vi * 3 + axes[0]];
                        vy[k] = v[vi * 3 + axes[1]];
                    }
                }
                const float e0x = vx[1] - vx[0], e0y = vy[1] - vy[0], e1x = vx[2] - vx[1], e1y = vy[2] - vy[1];
                const float cross = e0x * e1y - e0y * e1x;
                const float area = (vx[0] * vy[1] - vy[0] * vx[1]) * 0.5f;
                if (cross * area < 0.0f) { /* an internal angle */
                    guessVert += 1;
                    continue;
                }
                bool overlap = false;
                for (size_t otherVert = 3; otherVert < npolys; ++otherVert) {
                    const size_t idx = (guessVert + otherVert) % npolys;
                    if (idx >= remaining.size())
                        continue;
                    const size_t ovi = size_t(remaining[idx].v);
                    if ((ovi * 3 + axes[0]) >= v.size() || (ovi * 3 + axes[1]) >= v.size())
                        continue;
                    if (pnpoly(3, vx, vy, v[ovi * 3 + axes[0]], v[ovi * 3 + axes[1]])) {
                        overlap = true;
                        break;
                    }
                }
                if (overlap) {
                    guessVert += 1;
                    continue;
                }
                push(ind[0], ind[1], ind[2]); /* an ear */
                size_t removed = (guessVert + 1) % npolys;
                while (removed + 1 < npolys) {
                    remaining[removed] = remaining[removed + 1];
                    removed += 1;
                }
                remaining.pop_back();
            }
            if (remaining.size() == 3)
                push(remaining[0], remaining[1], remaining[2]);
        }
    }
    return true;
}

/* ParseTextureNameAndOption (tiny_obj_loader.h:1274-1358) */
inline bool parseTexture(std::string& texname, ObjTexOpt& opt, const char* linebuf)
{
    bool found = false;
    std::string name;
    const char* token = linebuf;
    auto parseOnOff = [](const char** t) {
        (*t) += strspn((*t), " \t");
        (*t) += strcspn((*t), " \t\r");
    };
    while (!isNewLine(*token)) {
        token += strspn(token, " \t");
        if (0 == strncmp(token, "-blendu", 7) && isSpace(token[7])) {
            token += 8;
            parseOnOff(&token);
        } else if (0 == strncmp(token, "-blendv", 7) && isSpace(token[7])) {
            token += 8;
            parseOnOff(&token);
        } else if (0 == strncmp(token, "-clamp", 6) && isSpace(token[6])) {
            token += 7;
            parseOnOff(&token);
        } else if (0 == strncmp(token, "-boost", 6) && isSpace(token[6])) {
            token += 7;
            parseReal(&token, 1.0);
        } else if (0 == strncmp(token, "-bm", 3) && isSpace(token[3])) {
            token += 4;
            opt.bumpMultiplier = parseReal(&token, 1.0);
        } else if (0 == strncmp(token, "-o", 2) && isSpace(token[2])) {
            token += 3;
            opt.originOffset[0] = parseReal(&token);
            opt.originOffset[1] = parseReal(&token);
            opt.originOffset[2] = parseReal(&token);
        } else if (0 == strncmp(token, "-s", 2) && isSpace(token[2])) {
            token += 3;
            opt.scale[0] = parseReal(&token, 1.0);
            opt.scale[1] = parseReal(&token, 1.0);
            opt.scale[2] = parseReal(&token, 1.0);
        } else if (0 == strncmp(token, "-t", 2) && isSpace(token[2])) {
            token += 3;
            parseReal(&token);
            parseReal(&token);
            parseReal(&token);
        } else if (0 == strncmp(token, "-type", 5) && isSpace(token[5])) {
            token += 5;
            parseString(&token);
        } else if (0 == strncmp(token, "-texres", 7) && isSpace(token[7])) {
            token += 7;
            parseInt(&token);
        } else if (0 == strncmp(token, "-imfchan", 8) && isSpace(token[8])) {
            token += 9;
            token += strspn(token, " \t");
            token += strcspn(token, " \t\r");
        } else if (0 == strncmp(token, "-mm", 3) && isSpace(token[3])) {
            token += 4;
            parseReal(&token, 0.0);
            parseReal(&token, 1.0);
        } else if (0 == strncmp(token, "-colorspace", 11) && isSpace(token[11])) {
            token += 12;
            parseString(&token);
        } else {
            name = std::string(token); /* the rest of the line: file names may contain spaces */
            token += name.length();
            found = true;
        }
    }
    if (found)
        texname = name;
    return found;
}

/* LoadMtl (tiny_obj_loader.h:2068-2467) */
inline void loadMtl(std::map<std::string, int>& materialMap, std::vector<ObjMaterial>& materials, std::istream& in, std::string& warning)
{
    ObjMaterial material;
    bool hasD = false, hasTr = false, hasKd = false;
    std::string linebuf;
    size_t lineNo = 0;
    while (getLine(in, linebuf)) {
        lineNo++;
        if (linebuf.size() > 0)
            linebuf = linebuf.substr(0, linebuf.find_last_not_of(" \t") + 1);
        if (linebuf.empty())
            continue;
        const char* token = linebuf.c_str();
        token += strspn(token, " \t");
        if (token[0] == '\0' || token[0] == '#')
            continue;
        auto real3 = [&](float* dst) {
            dst[0] = parseReal(&token);
            dst[1] = parseReal(&token);
            dst[2] = parseReal(&token);
        };
        if (0 == strncmp(token, "newmtl", 6) && isSpace(token[6])) {
            if (!material.name.empty()) {
                materialMap.insert(std::pair<std::string, int>(material.name, int(materials.size())));
                materials.push_back(material);
            }
            material = ObjMaterial();
            hasD = hasTr = hasKd = false;
            token += 7;
            material.name = parseString(&token);
            if (material.name.empty())
                warning += "empty material name in `newmtl`\n";
        } else if (token[0] == 'K' && token[1] == 'a' && isSpace(token[2])) {
            token += 2;
            real3(material.ambient);
        } else if (token[0] == 'K' && token[1] == 'd' && isSpace(token[2])) {
            token += 2;
            real3(material.diffuse);
            hasKd = true;
        } else if (token[0] == 'K' && token[1] == 's' && isSpace(token[2])) {
            token += 2;
            real3(material.specular);
        } else if ((token[0] == 'K' && token[1] == 't' && isSpace(token[2])) || (token[0] == 'T' && token[1] == 'f' && isSpace(token[2]))) {
            token += 2;
            real3(material.transmittance);
        } else if (token[0] == 'N' && token[1] == 'i' && isSpace(token[2])) {
            token += 2;
            material.ior = parseReal(&token);
        } else if (token[0] == 'K' && token[1] == 'e' && isSpace(token[2])) {
            token += 2;
            real3(material.emission);
        } else if (token[0] == 'N' && token[1] == 's' && isSpace(token[2])) {
            token += 2;
            material.shininess = parseReal(&token);
        } else if (0 == strncmp(token, "illum", 5) && isSpace(token[5])) {
            token += 6;
            material.illum = parseInt(&token);
        } else if (token[0] == 'd' && isSpace(token[1])) {
            token += 1;
            material.dissolve = parseReal(&token);
            if (hasTr)
                warning += "Both `d` and `Tr` parameters defined for \"" + material.name + "\". Use the value of `d` for dissolve (line " + std::to_string(lineNo) + " in .mtl.)\n";
            hasD = true;
        } else if (token[0] == 'T' && token[1] == 'r' && isSpace(token[2])) {
            token += 2;
            if (hasD)
                warning += "Both `d` and `Tr` parameters defined for \"" + material.name + "\". Use the value of `d` for dissolve (line " + std::to_string(lineNo) + " in .mtl.)\n";
            else
                material.dissolve = 1.0f - parseReal(&token);
            hasTr = true;
        } else if (0 == strncmp(token, "map_Kd", 6) && isSpace(token[6])) {
            token += 7;
            parseTexture(material.diffuseTex, material.diffuseOpt, token);
            if (!hasKd)
                material.diffuse[0] = material.diffuse[1] = material.diffuse[2] = 0.6f;
        } else if (0 == strncmp(token, "map_Ks", 6) && isSpace(token[6])) {
            token += 7;
            parseTexture(material.specularTex, material.specularOpt, token);
        } else if (0 == strncmp(token, "map_Ns", 6) && isSpace(token[6])) {
            token += 7;
            parseTexture(material.shininessTex, material.shininessOpt, token);
        } else if ((0 == strncmp(token, "map_bump", 8) || 0 == strncmp(token, "map_Bump", 8)) && isSpace(token[8])) {
            token += 9;
            parseTexture(material.bumpTex, material.bumpOpt, token);
        } else if (0 == strncmp(token, "bump", 4) && isSpace(token[4])) {
            token += 5;
            parseTexture(material.bumpTex, material.bumpOpt, token);
        } else if (0 == strncmp(token, "map_d", 5) && isSpace(token[5])) {
            token += 6;
            material.alphaTex = token;
            parseTexture(material.alphaTex, material.alphaOpt, token);
        } else if (0 == strncmp(token, "map_Ke", 6) && isSpace(token[6])) {
            token += 7;
            parseTexture(material.emissiveTex, material.emissiveOpt, token);
        } else if (0 == strncmp(token, "norm", 4) && isSpace(token[4])) {
            token += 5;
            parseTexture(material.normalTex, material.normalOpt, token);
        }
        /* everything else (Pr, Pm, map_Ka, disp, refl, ...) is not used by the importer */
    }
    materialMap.insert(std::pair<std::string, int>(material.name, int(materials.size())));
    materials.push_back(material);
}

inline void splitString(const std::string& s, char delim, char escape, std::vector<std::string>& elems)
{
    std::string token;
    bool escaping = false;
    for (size_t i = 0; i < s.size(); ++i) {
        char ch = s[i];
        if (escaping) {
            escaping = false;
        } else if (ch == escape) {
            escaping = true;
            continue;
        } else if (ch == delim) {
            if (!token.empty())
                elems.push_back(token);
            token.clear();
            continue;
        }
        token += ch;
    }
    elems.push_back(token);
}

}

/* ObjReader::ParseFromFile with triangulate = true, vertex_color = false (import.hpp:214-218).
 * Returns false (message in out.error) when the file cannot be read or a face cannot be parsed. */
inline bool loadObj(const std::string& filename, ObjData& out)
{
    using namespace objdetail;
    out = ObjData();
    std::string baseDir;
    size_t pos = filename.find_last_of("/\\");
    if (pos != std::string::npos)
        baseDir = filename.substr(0, pos);
    std::ifstream ifs(filename.c_str());
    if (!ifs) {
        out.error += "Cannot open file [" + filename + "]\n";
        return false;
    }
    std::vector<float>&v = out.vertices, &vn = out.normals, &vt = out.texcoords;
    std::vector<std::vector<VertexIndex>> faceGroup;
    std::string name;
    std::set<std::string> materialFilenames;
    std::map<std::string, int> materialMap;
    int material = -1;
    ObjShape shape;
    size_t lineNum = 0;
    std::string linebuf;
    while (getLine(ifs, linebuf)) {
        lineNum++;
        if (linebuf.empty())
            continue;
        const char* token = linebuf.c_str();
        token += strspn(token, " \t");
        if (token[0] == '\0' || token[0] == '#')
            continue;
        if (token[0] == 'v' && isSpace(token[1])) {
            token += 2;
            v.push_back(parseReal(&token));
            v.push_back(parseReal(&token));
            v.push_back(parseReal(&token));
            continue;
        }
        if (token[0] == 'v' && token[1] == 'n' && isSpace(token[2])) {
            token += 3;
            vn.push_back(parseReal(&token));
            vn.push_back(parseReal(&token));
            vn.push_back(parseReal(&token));
            continue;
        }
        if (token[0] == 'v' && token[1] == 't' && isSpace(token[2])) {
            token += 3;
            vt.push_back(parseReal(&token));
            vt.push_back(parseReal(&token));
            continue;
        }
        if (token[0] == 'f' && isSpace(token[1])) {
            token += 2;
            token += strspn(token, " \t");
            std::vector<VertexIndex> face;
            while (!isNewLine(token[0]) && token[0] != '#') {
                VertexIndex vi;
                if (!parseTriple(&token, int(v.size() / 3), int(vn.size() / 3), int(vt.size() / 2), &vi)) {
                    out.error += "Failed to parse `f' line (e.g. a zero value for vertex index or invalid relative vertex index). Line "
                        + std::to_string(lineNum) + ").\n";
                    return false;
                }
                face.push_back(vi);
                token += strspn(token, " \t\r");
            }
            faceGroup.push_back(face);
            continue;
        }
        if (0 == strncmp(token, "usemtl", 6)) {
            token += 6;
            std::string namebuf = parseString(&token);
            int newMaterialId = -1;
            auto it = materialMap.find(namebuf);
            if (it != materialMap.end())
                newMaterialId = it->second;
            else
                out.warning += "material [ '" + namebuf + "' ] not found in .mtl\n";
            if (newMaterialId != material) {
                exportFaces(shape, faceGroup, material, name, v, out.warning);
                faceGroup.clear();
                material = newMaterialId;
            }
            continue;
        }
        if (0 == strncmp(token, "mtllib", 6) && isSpace(token[6])) {
            token += 7;
            std::vector<std::string> filenames;
            splitString(std::string(token), ' ', '\\', filenames);
            bool found = false;
            for (size_t s = 0; s < filenames.size(); s++) {
                if (materialFilenames.count(filenames[s]) > 0) {
                    found = true;
                    continue;
                }
                std::string path = baseDir.empty() ? filenames[s] : (*baseDir.rbegin() != '/' ? baseDir + "/" + filenames[s] : baseDir + filenames[s]);
                std::ifstream mtl(path.c_str());
                if (mtl) {
                    std::string w;
                    loadMtl(materialMap, out.materials, mtl, w);
                    out.warning += w;
                    found = true;
                    materialFilenames.insert(filenames[s]);
                    break;
                }
                out.warning += "Material file [ " + filenames[s] + " ] not found in a path : " + baseDir + "\n";
            }
            if (!found)
                out.warning += "Failed to load material file(s). Use default material.\n";
            continue;
        }
        if (token[0] == 'g' && isSpace(token[1])) {
            exportFaces(shape, faceGroup, material, name, v, out.warning);
            if (shape.indices.size() > 0)
                out.shapes.push_back(shape);
            shape = ObjShape();
            faceGroup.clear();
            std::vector<std::string> names;
            while (!isNewLine(token[0]) && token[0] != '#') {
                names.push_back(parseString(&token));
                token += strspn(token, " \t\r");
            }
            /* names[0] is the "g" itself */
            if (names.size() < 2) {
                out.warning += "Empty group name. line: " + std::to_string(lineNum) + "\n";
                name = "";
            } else {
                name = names[1];
                for (size_t i = 2; i < names.size(); i++)
                    name += " " + names[i];
            }
            continue;
        }
        if (token[0] == 'o' && isSpace(token[1])) {
            exportFaces(shape, faceGroup, material, name, v, out.warning);
            if (shape.indices.size() > 0)
                out.shapes.push_back(shape);
            faceGroup.clear();
            shape = ObjShape();
            token += 2;
            name = std::string(token);
            continue;
        }
        /* l, p, t, s, vw and unknown statements do not concern the importer */
    }
    bool ret = exportFaces(shape, faceGroup, material, name, v, out.warning);
    if (ret || shape.indices.size())
        out.shapes.push_back(shape);
    return true;
}

}
