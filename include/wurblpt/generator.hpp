/*
 * generator.hpp -- procedural meshes with the reference's vertex output
 * (generator.hpp:39-70 quad, :72-171 cube, :173-265 one cube side, :272-304 disk, :311-347 sphere, :350-441 cylinder,
 * :444-526 cone, :528-580 torus, :583-735 tetrahedron, octahedron and icosahedron).
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

    /* one flat triangle with its own three vertices and the face normal (generator.hpp:583-610) */
    void face(const vec3& v0, const vec3& v1, const vec3& v2, const vec2& tc0, const vec2& tc1, const vec2& tc2)
    {
        const vec3 n = normalize(cross(v1 - v0, v2 - v0));
        const vec3 v[3] = { v0, v1, v2 };
        const vec2 tc[3] = { tc0, tc1, tc2 };
        for (int k = 0; k < 3; k++) {
            indices.push_back(positions.size());
            positions.push_back(v[k]);
            normals.push_back(n);
            texcoords.push_back(tc[k]);
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

/* One side of the cube; the sides count +x -x +y -y +z -z here (the cube itself emits them in another order) */
inline Mesh* generateCubeSide(int side, const Transformation& T = Transformation(), int slices = 1)
{
    MeshAssembler m;
    static const float nrm[6][3] = { { 1, 0, 0 }, { -1, 0, 0 }, { 0, 1, 0 }, { 0, -1, 0 }, { 0, 0, 1 }, { 0, 0, -1 } };
    if (side < 0 || side > 5)
        side = 0;
    m.grid(slices, slices, MeshAssembler::ABC_BDC, [slices, side](int i, int j) {
        float ty = i / (slices / 2.0f);
        float tx = j / (slices / 2.0f);
        vec3 p;
        switch (side) {
        case 4: p = vec3(-1.0f + tx, -1.0f + ty, 1.0f); break;
        case 5: p = vec3(1.0f - tx, -1.0f + ty, -1.0f); break;
        case 1: p = vec3(-1.0f, -1.0f + ty, -1.0f + tx); break;
        case 2: p = vec3(-1.0f + ty, 1.0f, -1.0f + tx); break;
        case 3: p = vec3(1.0f - ty, -1.0f, -1.0f + tx); break;
        default: p = vec3(1.0f, -1.0f + ty, 1.0f - tx); break;
        }
        return GeneratorVertex { p, vec3(nrm[side][0], nrm[side][1], nrm[side][2]), 0.5f * vec2(tx, ty) };
    });
    return m.finish(T);
}

/* Tip at y = +1, base circle of radius 1 at y = -1 */
inline Mesh* generateCone(bool closed, const Transformation& T, int slices = 40, int stacks = 20)
{
    MeshAssembler m;
    m.grid(stacks, slices, MeshAssembler::ABC_BDC, [slices, stacks](int i, int j) {
        float ty = static_cast<float>(i) / stacks;
        float tx = static_cast<float>(j) / slices;
        float alpha = tx * (2.0f * pi) - pi_2;
        float x = ty * cos(alpha);
        float y = -(ty * 2.0f - 1.0f);
        float z = ty * sin(alpha);
        float nx = x;
        float ny = 0.5f;
        float nz = z;
        float nl = sqrt(nx * nx + ny * ny + nz * nz);
        return GeneratorVertex { vec3(x, y, z), vec3(nx, ny, nz) / nl, vec2(1.0f - tx, 1.0f - ty) };
    });
    if (closed) {
        m.grid(1, slices, MeshAssembler::ACB_BCD, [slices](int i, int j) {
            float ty = static_cast<float>(i) / 1;
            float r = 0.0f + ty * (1.0f - 0.0f);
            float tx = static_cast<float>(j) / slices;
            float alpha = tx * (2.0f * pi) + pi_2;
            return GeneratorVertex { vec3(r * cos(alpha), -1.0f, r * sin(alpha)), vec3(0.0f, -1.0f, 0.0f), vec2(1.0f - tx, ty) };
        });
    }
    return m.finish(T);
}
inline Mesh* generateCone(const Transformation& T = Transformation(), int slices = 40, int stacks = 20) { return generateCone(false, T, slices, stacks); }
inline Mesh* generateClosedCone(const Transformation& T = Transformation(), int slices = 40, int stacks = 20) { return generateCone(true, T, slices, stacks); }

/* A torus around the z axis: tube centres on a circle of radius innerRadius + (1 - innerRadius) / 2 */
inline Mesh* generateTorus(const Transformation& T = Transformation(), float innerRadius = 0.4f, int sides = 40, int rings = 40)
{
    MeshAssembler m;
    const float ringradius = (1.0f - innerRadius) / 2.0f;
    const float ringcenter = innerRadius + ringradius;
    m.grid(sides, rings, MeshAssembler::ABC_BDC, [sides, rings, ringradius, ringcenter](int i, int j) {
        float ty = static_cast<float>(i) / sides;
        float alpha = ty * (2.0f * pi) - pi_2;
        float c = cos(alpha);
        float s = sin(alpha);
        float tx = static_cast<float>(j) / rings;
        float beta = tx * (2.0f * pi) - pi;
        float x = ringcenter + ringradius * cos(beta);
        float y = 0.0f;
        float z = ringradius * sin(beta);
        float rx = c * x + s * y;
        float ry = c * y - s * x;
        float rz = z;
        float rcx = c * ringcenter;
        float rcy = -s * ringcenter;
        float rcz = 0.0f;
        float nx = rx - rcx;
        float ny = ry - rcy;
        float nz = rz - rcz;
        float nl = sqrt(nx * nx + ny * ny + nz * nz);
        return GeneratorVertex { vec3(rx, ry, rz), vec3(nx, ny, nz) / nl, vec2(1.0f - tx, 1.0f - ty) };
    });
    return m.finish(T);
}

/* Platonic solids: flat faces, every face with the texture coordinates (0,0) (1,0) (0.5,1) */
inline Mesh* generateTetrahedron(const Transformation& T = Transformation())
{
    MeshAssembler m;
    const float a = 1.0f / 3.0f;
    const float b = sqrt(8.0f / 9.0f);
    const float c = sqrt(2.0f / 9.0f);
    const float d = sqrt(2.0f / 3.0f);
    const vec3 v0(-c, -a, d), v1(b, -a, 0.0f), v2(-d, -a, -d), v3(0.0f, 1.0f, 0.0f);
    const vec2 tc0(0.0f, 0.0f), tc1(1.0f, 0.0f), tc2(0.5f, 1.0f);
    m.face(v0, v1, v3, tc0, tc1, tc2);
    m.face(v1, v2, v3, tc0, tc1, tc2);
    m.face(v2, v0, v3, tc0, tc1, tc2);
    m.face(v2, v1, v0, tc0, tc1, tc2);
    return m.finish(T);
}

inline Mesh* generateOctahedron(const Transformation& T = Transformation())
{
    MeshAssembler m;
    const vec3 v0(0.0f, -1.0f, 0.0f), v1(0.0f, 0.0f, -1.0f), v2(+1.0f, 0.0f, 0.0f), v3(0.0f, 0.0f, +1.0f), v4(-1.0f, 0.0f, 0.0f),
          v5(0.0f, +1.0f, 0.0f);
    const vec2 tc0(0.0f, 0.0f), tc1(1.0f, 0.0f), tc2(0.5f, 1.0f);
    m.face(v1, v2, v0, tc0, tc1, tc2);
    m.face(v2, v3, v0, tc0, tc1, tc2);
    m.face(v3, v4, v0, tc0, tc1, tc2);
    m.face(v4, v1, v0, tc0, tc1, tc2);
    m.face(v1, v5, v2, tc1, tc2, tc0);
    m.face(v2, v5, v3, tc1, tc2, tc0);
    m.face(v3, v5, v4, tc1, tc2, tc0);
    m.face(v4, v5, v1, tc1, tc2, tc0);
    return m.finish(T);
}

inline Mesh* generateIcosahedron(const Transformation& T = Transformation())
{
    MeshAssembler m;
    const float r = 2.0f / (1 + sqrt(5.0f)); /* 1 / golden ratio */
    const vec3 v[12] = { vec3(0.0f, +r, -1.0f), vec3(+r, +1.0f, 0.0f), vec3(-r, +1.0f, 0.0f), vec3(0.0f, +r, +1.0f), vec3(0.0f, -r, +1.0f),
        vec3(-1.0f, 0.0f, +r), vec3(0.0f, -r, -1.0f), vec3(+1.0f, 0.0f, -r), vec3(+1.0f, 0.0f, +r), vec3(-1.0f, 0.0f, -r),
        vec3(+r, -1.0f, 0.0f), vec3(-r, -1.0f, 0.0f) };
    /* the twenty faces in the reference's order (it decides the hitable order) */
    static const int f[20][3] = { { 2, 1, 0 }, { 1, 2, 3 }, { 5, 4, 3 }, { 4, 8, 3 }, { 7, 6, 0 }, { 6, 9, 0 }, { 11, 10, 4 }, { 10, 11, 6 },
        { 9, 5, 2 }, { 5, 9, 11 }, { 8, 7, 1 }, { 7, 8, 10 }, { 2, 5, 3 }, { 8, 1, 3 }, { 9, 2, 0 }, { 1, 7, 0 }, { 11, 9, 6 }, { 7, 10, 6 },
        { 5, 11, 4 }, { 10, 8, 4 } };
    const vec2 tc0(0.0f, 0.0f), tc1(1.0f, 0.0f), tc2(0.5f, 1.0f);
    for (int k = 0; k < 20; k++)
        m.face(v[f[k][0]], v[f[k][1]], v[f[k][2]], tc0, tc1, tc2);
    return m.finish(T);
}

}
