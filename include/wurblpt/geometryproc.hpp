/*
 * geometryproc.hpp -- vertex normal / tangent generation on the host.
 *
 * Every textured Mesh gets tangents (reference mesh.hpp:108-111), and they end up in the
 * shading frame of every hit, so computeTangents() follows the reference's arithmetic
 * (geometryproc.hpp:181-226): per-triangle tangent from the uv determinant, summed per
 * vertex, Gram-Schmidt against the normal.  computeNormals() follows :58-177
 * (angle-weighted face normals by default).
 */
#pragma once

#include <vector>

#include "gvm.hpp"

namespace WurblPT {

typedef enum { NormalsFromFirstFace, NormalsFromFaceAverage, NormalsFromWeightedFaceAverage } NormalSource;

inline std::vector<vec3> computeNormals(const std::vector<vec3>& positions, const std::vector<unsigned int> indices,
        NormalSource normalSource = NormalsFromWeightedFaceAverage)
{
    const size_t nv = positions.size();
    const size_t nt = indices.size() / 3;
    std::vector<vec3> faceNormals(nt);
    std::vector<std::vector<unsigned int>> facesOfVertex(nv);
    for (size_t f = 0; f < nt; f++) {
        unsigned int i0 = indices[3 * f + 0], i1 = indices[3 * f + 1], i2 = indices[3 * f + 2];
        vec3 e0 = positions[i1] - positions[i0];
        vec3 e1 = positions[i2] - positions[i0];
        vec3 e2 = e1 - e0;
        vec3 fn(0.0f, 0.0f, 1.0f);
        if (dot(e0, e0) > 0.0f && dot(e1, e1) > 0.0f && dot(e2, e2) > 0.0f) {
            vec3 c = cross(e0, e1);
            if (dot(c, c) > 0.0f)
                fn = normalize(c);
        }
        faceNormals[f] = fn;
        facesOfVertex[i0].push_back(f);
        facesOfVertex[i1].push_back(f);
        facesOfVertex[i2].push_back(f);
    }
    std::vector<vec3> normals(nv);
    for (size_t v = 0; v < nv; v++) {
        const std::vector<unsigned int>& faces = facesOfVertex[v];
        vec3 n(0.0f);
        if (faces.size() == 0) {
            n = vec3(0.0f, 0.0f, 1.0f);
        } else if (faces.size() == 1) {
            n = faceNormals[faces[0]];
        } else {
            if (normalSource == NormalsFromWeightedFaceAverage) {
                for (size_t j = 0; j < faces.size(); j++) {
                    unsigned int f = faces[j];
                    unsigned int fi[3] = { indices[3 * f + 0], indices[3 * f + 1], indices[3 * f + 2] };
                    vec3 e0, e1;
                    if (v == fi[0]) {
                        e0 = positions[fi[1]];
                        e1 = positions[fi[2]];
                    } else if (v == fi[1]) {
                        e0 = positions[fi[2]];
                        e1 = positions[fi[0]];
                    } else {
                        e0 = positions[fi[0]];
                        e1 = positions[fi[1]];
                    }
                    e0 = e0 - positions[v];
                    e1 = e1 - positions[v];
                    if (dot(e0, e0) <= 0.0f || dot(e1, e1) <= 0.0f)
                        continue;
                    float x = dot(normalize(e0), normalize(e1));
                    float alpha = acos(clamp(x, -1.0f, +1.0f));
                    n += alpha * faceNormals[f];
                }
            }
            if (normalSource == NormalsFromFaceAverage || (normalSource == NormalsFromWeightedFaceAverage && dot(n, n) <= 0.0f)) {
                n = vec3(0.0f);
                for (size_t j = 0; j < faces.size(); j++)
                    n += faceNormals[faces[j]];
            }
            if (normalSource == NormalsFromFirstFace || dot(n, n) <= 0.0f)
                n = faceNormals[faces[0]];
            n = normalize(n);
        }
        normals[v] = n;
    }
    return normals;
}

inline std::vector<vec3> computeTangents(const std::vector<vec3>& positions, const std::vector<vec3>& normals,
        const std::vector<vec2>& texcoords, const std::vector<unsigned int> indices)
{
    const size_t nv = positions.size();
    const size_t nt = indices.size() / 3;
    std::vector<vec3> tangents(nv, vec3(0.0f));
    for (size_t f = 0; f < nt; f++) {
        unsigned int i0 = indices[3 * f + 0], i1 = indices[3 * f + 1], i2 = indices[3 * f + 2];
        vec3 e1 = positions[i1] - positions[i0];
        vec3 e2 = positions[i2] - positions[i0];
        float s1 = texcoords[i1].x() - texcoords[i0].x();
        float t1 = texcoords[i1].y() - texcoords[i0].y();
        float s2 = texcoords[i2].x() - texcoords[i0].x();
        float t2 = texcoords[i2].y() - texcoords[i0].y();
        float det = (s1 * t2 - s2 * t1);
        if (abs(det) > epsilon) {
            vec3 tp = 1.0f / det * (t2 * e1 - t1 * e2);
            tangents[i0] += tp;
            tangents[i1] += tp;
            tangents[i2] += tp;
        }
    }
    for (size_t v = 0; v < nv; v++) {
        const vec3 n = normals[v];
        const vec3 tp = tangents[v];
        vec3 t(1.0f, 0.0f, 0.0f);
        if (dot(tp, tp) > 0.0f)
            t = normalize(tp - dot(n, tp) * n);
        tangents[v] = t;
    }
    return tangents;
}

}
