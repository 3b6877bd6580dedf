/*
 * mesh.hpp -- triangle mesh storage with the reference's layout (mesh.hpp:39-189):
 * interleaved float vertices pos3, nrm3, [tc2], [tan3] (6 / 8 / 11 floats) and uint indices;
 * MeshInstance = mesh + material + transformation.
 *
 * Interface (class, member and function names, argument order) and the arithmetic that the bit-parity contract fixes
 * follow marlam/wurblpt, which is distributed under the MIT licence: Copyright (c) 2023 Martin Lambers
 * <marlam@marlam.de>; the permission notice is reproduced in the LICENSE file of this repository.  The implementation
 * below is this repository's own.
 */
#pragma once

#include <cassert>
#include <vector>

#include "geometryproc.hpp"
#include "material.hpp"
#include "scene_component.hpp"
#include "transformation.hpp"

namespace WurblPT {

class Mesh
{
public:
    constexpr static size_t positionOffset = 0;
    constexpr static size_t normalOffset = 3;
    constexpr static size_t texcoordOffset = 6;
    constexpr static size_t tangentOffset = 8;
    bool haveTexCoords;
    bool haveTangents;
    std::vector<float> vertices;
    std::vector<unsigned int> indices;

    static size_t vertexSize(bool haveTexCoords, bool haveTangents)
    {
        return (haveTexCoords && haveTangents ? 11 : haveTexCoords ? 8 : 6);
    }
    size_t vertexSize() const { return vertexSize(haveTexCoords, haveTangents); }
    size_t vertexCount() const { return vertices.size() / vertexSize(); }
    size_t triangleCount() const { return indices.size() / 3; }
    vec3 position(unsigned int i) const { return vec3(vertices.data() + i * vertexSize() + positionOffset); }
    vec3 normal(unsigned int i) const { return vec3(vertices.data() + i * vertexSize() + normalOffset); }
    vec2 texcoord(unsigned int i) const { return vec2(vertices.data() + i * vertexSize() + texcoordOffset); }
    vec3 tangent(unsigned int i) const { return vec3(vertices.data() + i * vertexSize() + tangentOffset); }

private:
    static bool anyNonZero(const std::vector<vec2>& coordinates)
    {
        for (const vec2& c : coordinates)
            if (c.x() != 0.0f || c.y() != 0.0f)
                return true;
        return false;
    }
    static void put3(float* destination, const vec3& v)
    {
        destination[0] = v.x();
        destination[1] = v.y();
        destination[2] = v.z();
    }

public:
    /* Positions, normals and indices are required.  Texture coordinates may be missing or all zero: the mesh then
     * stores none, and no tangents either (there is nothing to derive them from).  T is applied to the vertices here,
     * once: positions through its 4x4 matrix, normals and tangents through its normal matrix. */
    Mesh(const std::vector<vec3>& pos, const std::vector<vec3>& nrm, const std::vector<vec2>& tc,
            const std::vector<unsigned int>& ind, const Transformation& T = Transformation(), bool wantTangents = true) :
        haveTexCoords(anyNonZero(tc)), haveTangents(wantTangents && haveTexCoords), indices(ind)
    {
        assert(!pos.empty() && nrm.size() == pos.size() && (tc.empty() || tc.size() == pos.size()));
        assert(!ind.empty() && ind.size() % 3 == 0);
        const std::vector<vec3> tangents = haveTangents ? computeTangents(pos, nrm, tc, ind) : std::vector<vec3>();
        const bool baked = !T.isIdentity();
        const mat4 toWorld = baked ? T.toMat4() : mat4(1.0f);
        const mat3 normalToWorld = baked ? T.toNormalMatrix() : mat3(1.0f);
        const size_t stride = vertexSize();
        vertices.resize(pos.size() * stride);
        for (size_t i = 0; i < pos.size(); i++) {
            float* vertex = vertices.data() + i * stride;
            put3(vertex + positionOffset, baked ? (toWorld * vec4(pos[i], 1.0f)).xyz() : pos[i]);
            put3(vertex + normalOffset, baked ? normalToWorld * nrm[i] : nrm[i]);
            if (haveTexCoords) {
                vertex[texcoordOffset] = tc[i].x();
                vertex[texcoordOffset + 1] = tc[i].y();
            }
            if (haveTangents)
                put3(vertex + tangentOffset, baked ? normalToWorld * tangents[i] : tangents[i]);
        }
    }
};

class MeshInstance : public SceneComponent
{
public:
    const Mesh* mesh;
    const Material* material;
    const Transformation transformation;
    const int animationIndex;
    const mat4 transformationM;
    const mat3 transformationN;

    MeshInstance(const Mesh* mesh, const Material* m, const Transformation& t, int ai = -1) :
        mesh(mesh), material(m), transformation(t), animationIndex(ai),
        transformationM(transformation.toMat4()), transformationN(transformation.toNormalMatrix())
    {
    }
    MeshInstance(const Mesh* mesh, const Material* m, int ai = -1) : MeshInstance(mesh, m, Transformation(), ai) {}
};

}
