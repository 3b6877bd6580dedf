"""Writes the import fixture: scene.obj / scene.mtl and its textures (a small room that exercises
every branch of importIntoScene: Lambertian / ModPhong / emissive / transparent materials, diffuse
texture with and without alpha, bump map -> normal map, faces without normals or texture
coordinates, a polygon, an unknown material, two groups sharing a material).

usage: python tests/golden/obj/make_scene_fixture.py    (rewrites the files next to it)"""
import os
import struct
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.RandomState(7)


def write_ppm(name, img):
    h, w, c = img.shape
    open(os.path.join(HERE, name), "wb").write((b"P6" if c == 3 else b"P5") + b"\n%d %d\n255\n" % (w, h) + img.astype(np.uint8).tobytes())


def write_png(name, img):
    h, w, c = img.shape
    raw = b"".join(b"\0" + img[y].astype(np.uint8).tobytes() for y in range(h))

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    data = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, {1: 0, 2: 4, 3: 2, 4: 6}[c], 0, 0, 0))
    open(os.path.join(HERE, name), "wb").write(data + chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b""))


n = 32
yy, xx = np.mgrid[0:n, 0:n]
tiles = (((xx // 8) + (yy // 8)) % 2).astype(np.float32)
floor = np.stack([120 + 90 * tiles + rng.randint(0, 20, (n, n)), 100 + 80 * tiles + rng.randint(0, 20, (n, n)), 80 + 60 * tiles], -1)
write_ppm("floor.ppm", floor)
bump = (127 + 100 * np.sin(xx * 0.8) * np.cos(yy * 0.5)).clip(0, 255)[..., None]
write_ppm("bump.pgm", bump)
leaf = np.zeros((n, n, 4))
leaf[..., 0] = 40 + 3 * xx
leaf[..., 1] = 150 + 2 * yy
leaf[..., 2] = 30
leaf[..., 3] = 255 * (((xx - 16) ** 2 + (yy - 16) ** 2) < 14 ** 2)
write_png("leaf.png", leaf)

open(os.path.join(HERE, "scene.mtl"), "w").write("""# materials of the import fixture (Tf 1 1 1 marks an opaque filter, as exporters write it; the importer
# reads a missing Tf as 0 0 0, i.e. fully transparent, import.hpp:305-311 -- `broken` below shows that)
newmtl floor
Kd 0.8 0.8 0.8
Tf 1 1 1
map_Kd -s 4 4 1 floor.ppm
bump -bm 6 bump.pgm

newmtl wall
Kd 0.7 0.65 0.6
Tf 1 1 1

newmtl shiny
Kd 0.2 0.3 0.6
Tf 1 1 1
Ks 0.6 0.6 0.6
Ns 80

newmtl lamp
Kd 0 0 0
Tf 1 1 1
Ke 18 17 15

newmtl leaf
Kd 1 1 1
Tf 1 1 1
map_Kd leaf.png
Ks 0.05 0.05 0.05
Ns 20

newmtl pane
Kd 0.9 0.95 1.0
Tf 1 1 1
d 0.35
Ni 1.5

newmtl broken
Kd 0.5 0.1 0.1
map_Kd does_not_exist.png
""")

open(os.path.join(HERE, "scene.obj"), "w").write("""# import fixture: a small room
mtllib scene.mtl
# floor (quad with normals and texture coordinates) and back wall (no normals, no texture coordinates)
v -3 0 -3
v 3 0 -3
v 3 0 3
v -3 0 3
v -3 4 -3
v 3 4 -3
vt 0 0
vt 1 0
vt 1 1
vt 0 1
vn 0 1 0
g room
usemtl floor
f 1/1/1 4/4/1 3/3/1 2/2/1
usemtl wall
f 1 2 6 5
# a box made of quads (unit normals per face are left to the importer)
v -1.5 0 -1
v -0.5 0 -1
v -0.5 0 0
v -1.5 0 0
v -1.5 1.2 -1
v -0.5 1.2 -1
v -0.5 1.2 0
v -1.5 1.2 0
g box
usemtl shiny
f 11 12 13 14
f 7 8 12 11
f 8 9 13 12
f 9 10 14 13
f 10 7 11 14
# lamp: a quad near the ceiling facing down
v -0.6 3.8 -0.6
v 0.6 3.8 -0.6
v 0.6 3.8 0.6
v -0.6 3.8 0.6
vn 0 -1 0
g lamp
usemtl lamp
f 15//2 16//2 17//2 18//2
# a leaf card with an alpha texture, a transparent pane, a pentagon of the wall material
v 0.8 0.2 0.5
v 2.0 0.2 0.2
v 2.0 1.6 0.2
v 0.8 1.6 0.5
g cards
usemtl leaf
f 19/1 20/2 21/3 22/4
v 0.2 0.1 1.5
v 1.4 0.1 1.8
v 1.4 1.3 1.8
v 0.2 1.3 1.5
usemtl pane
f 23 24 25 26
v -2.6 0.01 1.0
v -1.6 0.01 1.2
v -1.3 0.01 2.0
v -2.1 0.01 2.6
v -2.9 0.01 1.9
usemtl wall
f 27 31 30 29 28
usemtl broken
f 27/1 28/2 29/3
usemtl undefined_material
f 29 30 31
""")
print("wrote", sorted(os.listdir(HERE)))
