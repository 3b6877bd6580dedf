/*
 * mesh.hpp -- triangle mesh storage with the reference's layout (mesh.hpp:39-189):
 * interleaved float vertices pos3, nrm3, [tc2], [tan3] (6 / 8 / 11 floats) and uint indices;
 * MeshInstance = mesh + material + transformation.
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

    /* Positions, normals and indices are required; texture coordinates may be empty (or all
     * zero), in which case there are no tangents either.  T is baked into the vertices. */
    Mesh(const std::vector<vec3>& pos, const std::vector<vec3>& nrm, const std::vector<vec2>& tc,
            const std::vector<unsigned int>& ind, const Transformation& T = Transformation(), bool wantTangents = true) :
        indices(ind)
    {
        assert(pos.size() > 0 && pos.size() == nrm.size());
        assert(pos.size() == tc.size() || tc.size() == 0);
        assert(ind.size() > 0 && ind.size() % 3 == 0);
        bool allTexCoordsAreZero = true;
        for (size_t i = 0; i < tc.size(); i++) {
            if (tc[i] != vec2(0.0f, 0.0f)) {
                allTexCoordsAreZero = false;
                break;
            }
        }
        if (allTexCoordsAreZero)
            wantTangents = false;
        std::vector<vec3> tng;
        if (wantTangents)
            tng = computeTangents(pos, nrm, tc, ind);
        haveTexCoords = !allTexCoordsAreZero;
        haveTangents = wantTangents;
        vertices.resize(pos.size() * vertexSize());
        const bool transform = !T.isIdentity();
        mat4 M(1.0f);
        mat3 N(1.0f);
        if (transform) {
            M = T.toMat4();
            N = T.toNormalMatrix();
        }
        for (size_t i = 0; i < pos.size(); i++) {
            float* d = vertices.data() + i * vertexSize();
            vec3 p = pos[i];
            vec3 n = nrm[i];
            if (transform) {
                p = (M * vec4(p, 1.0f)).xyz();
                n = N * n;
            }
            d[0] = p.x(); d[1] = p.y(); d[2] = p.z();
            d[3] = n.x(); d[4] = n.y(); d[5] = n.z();
            if (haveTexCoords) {
                d[6] = tc[i][0];
                d[7] = tc[i][1];
            }
            if (haveTangents) {
                vec3 t = tng[i];
                if (transform)
                    t = N * t;
                d[8] = t.x(); d[9] = t.y(); d[10] = t.z();
            }
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
