/*
 * generator.hpp -- procedural meshes with the reference's vertex output
 * (generator.hpp:39-70 quad, :72-171 cube, :272-304 disk, :311-347 sphere, :350-441 cylinder).
 *
 * All shapes fill [-1,+1]^3 around the origin and bake the given Transformation.  Vertex
 * order, index order and texture coordinates match the reference, because they decide the
 * hitable order seen by the BVH builder and the tangents computed per vertex.  The shapes are
 * expressed here as parametric (row, column) grids with a pluggable vertex function.
 */
#pragma once

#include <functional>
#include <vector>

#include "mesh.hpp"

namespace WurblPT {

struct GeneratorVertex {
    vec3 position, normal;
    vec2 texcoord;
};

class MeshAssembler
{
public:
    std::vector<vec3> positions, normals;
    std::vector<vec2> texcoords;
    std::vector<unsigned int> indices;

    /* cell winding patterns over the corners a=(i,j) b=(i,j+1) c=(i+1,j) d=(i+1,j+1) */
    enum Winding { ABC_BDC, ACB_BCD };

    /* Appends a (rows+1) x (cols+1) vertex grid; vertex(i, j) is emitted row by row and the
     * two triangles of each cell directly after the cell's first vertex. */
    void grid(int rows, int cols, Winding winding, const std::function<GeneratorVertex(int, int)>& vertex)
    {
        unsigned int base = positions.size();
        for (int i = 0; i <= rows; i++) {
            for (int j = 0; j <= cols; j++) {
                GeneratorVertex v = vertex(i, j);
                positions.push_back(v.position);
                normals.push_back(v.normal);
                texcoords.push_back(v.texcoord);
                if (i < rows && j < cols) {
                    unsigned int a = base + (i + 0) * (cols + 1) + (j + 0);
                    unsigned int b = base + (i + 0) * (cols + 1) + (j + 1);
                    unsigned int c = base + (i + 1) * (cols + 1) + (j + 0);
                    unsigned int d = base + (i + 1) * (cols + 1) + (j + 1);
                    switch (winding) {
                    case ABC_BDC: {
                        unsigned int t[6] = { a, b, c, b, d, c };
                        indices.insert(indices.end(), t, t + 6);
                        break;
                    }
                    case ACB_BCD: {
                        unsigned int t[6] = { a, c, b, b, c, d };
                        indices.insert(indices.end(), t, t + 6);
                        break;
                    }
                    }
                }
            }
        }
    }

    Mesh* finish(const Transformation& T) const { return new Mesh(positions, normals, texcoords, indices, T); }
};

inline Mesh* generateQuad(const Transformation& T = Transformation(), int slices = 1)
{
    MeshAssembler m;
    m.grid(slices, slices, MeshAssembler::ABC_BDC, [slices](int i, int j) {
        float ty = i / (slices / 2.0f);
        float tx = j / (slices / 2.0f);
        return GeneratorVertex { vec3(-1.0f + tx, -1.0f + ty, 0.0f), vec3(0.0f, 0.0f, 1.0f), 0.5f * vec2(tx, ty) };
    });
    return m.finish(T);
}

inline Mesh* generateCube(const Transformation& T = Transformation(), int slices = 1)
{
    MeshAssembler m;
    /* sides: front, back, left, right, top, bottom */
    static const float nrm[6][3] = { { 0, 0, 1 }, { 0, 0, -1 }, { -1, 0, 0 }, { 1, 0, 0 }, { 0, 1, 0 }, { 0, -1, 0 } };
    for (int side = 0; side < 6; side++) {
        m.grid(slices, slices, MeshAssembler::ABC_BDC, [slices, side](int i, int j) {
            float ty = i / (slices / 2.0f);
            float tx = j / (slices / 2.0f);
            vec3 p;
            switch (side) {
            case 0: p = vec3(-1.0f + tx, -1.0f + ty, 1.0f); break;
            case 1: p = vec3(1.0f - tx, -1.0f + ty, -1.0f); break;
            case 2: p = vec3(-1.0f, -1.0f + ty, -1.0f + tx); break;
            case 3: p = vec3(1.0f, -1.0f + ty, 1.0f - tx); break;
            case 4: p = vec3(-1.0f + ty, 1.0f, -1.0f + tx); break;
            default: p = vec3(1.0f - ty, -1.0f, -1.0f + tx); break;
            }
            return GeneratorVertex { p, vec3(nrm[side][0], nrm[side][1], nrm[side][2]), 0.5f * vec2(tx, ty) };
        });
    }
    return m.finish(T);
}

inline Mesh* generateDisk(const Transformation& T = Transformation(), float innerRadius = 0.0f, int slices = 40)
{
    MeshAssembler m;
    m.grid(1, slices, MeshAssembler::ACB_BCD, [slices, innerRadius](int i, int j) {
        float ty = static_cast<float>(i) / 1;
        float r = innerRadius + ty * (1.0f - innerRadius);
        float tx = static_cast<float>(j) / slices;
        float alpha = tx * (2.0f * pi) + pi_2;
        return GeneratorVertex { vec3(r * cos(alpha), r * sin(alpha), 0.0f), vec3(0.0f, 0.0f, 1.0f), vec2(1.0f - tx, ty) };
    });
    return m.finish(T);
}

inline Mesh* generateSphere(const Transformation& T = Transformation(), int slices = 40, int stacks = 20)
{
    MeshAssembler m;
    m.grid(stacks, slices, MeshAssembler::ABC_BDC, [slices, stacks](int i, int j) {
        float ty = static_cast<float>(i) / stacks;
        float lat = ty * pi;
        float tx = static_cast<float>(j) / slices;
        float lon = tx * (2.0f * pi) - pi_2;
        float sinlat = sin(lat), coslat = cos(lat), sinlon = sin(lon), coslon = cos(lon);
        vec3 p(sinlat * coslon, coslat, sinlat * sinlon);
        return GeneratorVertex { p, p, vec2(1.0f - tx, 1.0f - ty) };
    });
    return m.finish(T);
}

inline Mesh* generateCylinder(bool closed, const Transformation& T, int slices)
{
    MeshAssembler m;
    m.grid(1, slices, MeshAssembler::ABC_BDC, [slices](int i, int j) {
        float ty = static_cast<float>(i) / 1;
        float tx = static_cast<float>(j) / slices;
        float alpha = tx * (2.0f * pi) - pi_2;
        float x = cos(alpha);
        float y = -(ty * 2.0f - 1.0f);
        float z = sin(alpha);
        return GeneratorVertex { vec3(x, y, z), vec3(x, 0.0f, z), vec2(1.0f - tx, 1.0f - ty) };
    });
    if (closed) {
        for (int side = 0; side < 2; side++) {
            const float y = side == 0 ? +1.0f : -1.0f;
            m.grid(1, slices, side == 0 ? MeshAssembler::ABC_BDC : MeshAssembler::ACB_BCD, [slices, y](int i, int j) {
                float ty = static_cast<float>(i) / 1;
                float r = 0.0f + ty * (1.0f - 0.0f);
                float tx = static_cast<float>(j) / slices;
                float alpha = tx * (2.0f * pi) + pi_2;
                return GeneratorVertex { vec3(r * cos(alpha), y, r * sin(alpha)), vec3(0.0f, y, 0.0f), vec2(1.0f - tx, ty) };
            });
        }
    }
    return m.finish(T);
}

inline Mesh* generateCylinder(const Transformation& T = Transformation(), int slices = 40) { return generateCylinder(false, T, slices); }
inline Mesh* generateClosedCylinder(const Transformation& T = Transformation(), int slices = 40) { return generateCylinder(true, T, slices); }

}
