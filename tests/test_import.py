"""The importer's building blocks (scope row f1): OBJ / MTL reader, image decoders, importIntoScene."""
import json
import os

import numpy as np
import pytest

from wurblpt_amd import host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ_DIR = os.path.join(ROOT, "tests", "golden", "obj")


def test_obj_reader_matches_the_vendored_tinyobjloader(golden, tmp_path):
    """include/wurblpt/objreader.hpp against what the reference's own parser (tiny_obj_loader.h, compiled
    by oracle/ref_probe.cpp with the importer's configuration) produces for tests/golden/obj/cases.obj:
    every float bit (its number parser is restated), index forms incl. relative indices, triangulation
    of quads (shorter diagonal) and of a pentagon, a concave hexagon and a heptagon in the xz plane
    (its ear clipping), shape boundaries at `g` / `o`, material switches inside a shape, an unknown
    material, MTL values incl. d / Tr precedence and texture options."""
    out = str(tmp_path / "obj.json")
    assert host.lib().wpt_host_obj_dump(os.path.join(OBJ_DIR, "cases.obj").encode(), out.encode()) == 0
    mine = json.load(open(out))
    assert len(mine) == 10
    for key, value in mine.items():
        assert value == golden.raw[key], key
    assert golden.raw["obj_shape_index_counts"] == [21, 12, 9, 15, 18, 3]


# ---- image decoders: files written here with numpy / zlib, decoded by include/wurblpt/imageio.hpp ----

def _png_rows(img, depth, rng):
    """the filtered scanlines of one (reduced) image; a random filter type per scanline exercises all five predictors"""
    h, w, c = img.shape
    bpp = max(1, c * depth // 8)
    rows = []
    prev = np.zeros(w * c * depth // 8, np.int32)
    for y in range(h):
        if depth == 16:
            cur = np.stack([img[y].astype(np.uint16) >> 8, img[y].astype(np.uint16) & 255], axis=-1).reshape(-1).astype(np.int32)
        else:
            cur = img[y].reshape(-1).astype(np.int32)
        f = int(rng.randint(0, 5))
        a = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        cc = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        if f == 0:
            pred = np.zeros_like(cur)
        elif f == 1:
            pred = a
        elif f == 2:
            pred = prev
        elif f == 3:
            pred = (a + prev) >> 1
        else:
            p = a + prev - cc
            pa, pb, pc = np.abs(p - a), np.abs(p - prev), np.abs(p - cc)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, cc))
        rows.append(bytes([f]) + ((cur - pred) & 255).astype(np.uint8).tobytes())
        prev = cur
    return rows


def _png_bytes(img, depth, palette=None, trns=None, interlace=False):
    """img: [h, w, c] top row first; plain, or the seven reduced images of Adam7 one after the other"""
    import struct
    import zlib
    h, w, c = img.shape
    color_type = {1: 0, 2: 4, 3: 2, 4: 6}[c] if palette is None else 3
    rng = np.random.RandomState(w * 131 + h)
    if interlace:
        rows = []
        for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
            sub = img[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                rows += _png_rows(sub, depth, rng)
    else:
        rows = _png_rows(img, depth, rng)

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color_type, 0, 0, 1 if interlace else 0))
    if palette is not None:
        out += chunk(b"PLTE", palette.astype(np.uint8).tobytes())
        if trns is not None:
            out += chunk(b"tRNS", trns.astype(np.uint8).tobytes())
    data = zlib.compress(b"".join(rows), 6)
    out += chunk(b"IDAT", data[:len(data) // 2]) + chunk(b"IDAT", data[len(data) // 2:]) + chunk(b"IEND", b"")
    return out


def test_png_decoder(tmp_path):
    rng = np.random.RandomState(1)
    for comps in (1, 2, 3, 4):
        for depth in (8, 16):
            img = rng.randint(0, 256 if depth == 8 else 65536, (13, 17, comps)).astype(np.uint16 if depth == 16 else np.uint8)
            img[3:9, 2:12] = img[3, 2]  # runs, so that deflate emits matches
            f = tmp_path / ("t_%d_%d.png" % (comps, depth))
            f.write_bytes(_png_bytes(img, depth))
            got = host.image_load(str(f))
            assert got is not None and got.dtype == img.dtype and np.array_equal(got, img[::-1]), (comps, depth)
    pal = rng.randint(0, 256, (16, 3))
    idx = rng.randint(0, 16, (9, 11, 1)).astype(np.uint8)
    f = tmp_path / "pal.png"
    f.write_bytes(_png_bytes(idx, 8, palette=pal, trns=np.arange(10) * 20))
    got = host.image_load(str(f))
    want = np.concatenate([pal[idx[..., 0]], np.where(idx < 10, idx * 20, 255)], axis=-1).astype(np.uint8)
    assert np.array_equal(got, want[::-1])


def test_png_decoder_adam7(tmp_path):
    """Interlaced PNG: seven reduced images, each filtered on its own; sizes from 1 x 1 (six empty passes) to ones
    that are no multiple of 8.  Where Pillow is present it reads the test's own files the same way."""
    rng = np.random.RandomState(4)
    try:
        from PIL import Image
    except ImportError:
        Image = None
    for (h, w) in ((1, 1), (2, 3), (5, 1), (8, 8), (9, 17), (23, 31), (40, 16)):
        for comps, depth in ((1, 8), (3, 8), (4, 8), (2, 16), (3, 16)):
            img = rng.randint(0, 256 if depth == 8 else 65536, (h, w, comps)).astype(np.uint16 if depth == 16 else np.uint8)
            f = tmp_path / "i.png"
            f.write_bytes(_png_bytes(img, depth, interlace=True))
            got = host.image_load(str(f))
            assert got is not None and got.dtype == img.dtype and np.array_equal(got, img[::-1]), (h, w, comps, depth)
            if Image is not None and depth == 8 and comps != 2:
                ref = np.asarray(Image.open(str(f)))
                assert np.array_equal(ref.reshape(img.shape), img), "the test's writer"
    pal = rng.randint(0, 256, (16, 3))
    idx = rng.randint(0, 16, (19, 21, 1)).astype(np.uint8)
    f = tmp_path / "ipal.png"
    f.write_bytes(_png_bytes(idx, 8, palette=pal, interlace=True))
    assert np.array_equal(host.image_load(str(f)), pal[idx[..., 0]].astype(np.uint8)[::-1])


def test_tga_pnm_pfm_hdr_decoders(tmp_path):
    import struct
    rng = np.random.RandomState(2)
    # TGA: flat and run-length encoded, bottom-up and top-down, grey / BGR / BGRA
    for comps in (1, 3, 4):
        img = rng.randint(0, 256, (10, 12, comps)).astype(np.uint8)
        img[2:6, 1:9] = img[2, 1]
        stored = img if comps == 1 else img[..., [2, 1, 0] + ([3] if comps == 4 else [])]
        for rle in (False, True):
            for top in (False, True):
                rows = stored[::-1] if top else stored  # file order; array row 0 = bottom
                if rle:
                    body = bytearray()
                    flat = rows.reshape(-1, comps)
                    i = 0
                    while i < len(flat):
                        run = 1
                        while i + run < len(flat) and run < 128 and np.array_equal(flat[i + run], flat[i]):
                            run += 1
                        if run > 1:
                            body += bytes([0x80 | (run - 1)]) + flat[i].tobytes()
                            i += run
                        else:
                            n = 1
                            while i + n < len(flat) and n < 128 and not np.array_equal(flat[i + n], flat[i + n - 1]):
                                n += 1
                            body += bytes([n - 1]) + flat[i:i + n].tobytes()
                            i += n
                    body = bytes(body)
                else:
                    body = rows.tobytes()
                hdr = struct.pack("<BBBHHBHHHHBB", 0, 0, (11 if comps == 1 else 10) if rle else (3 if comps == 1 else 2), 0, 0, 0, 0, 0, 12, 10,
                                  8 * comps, 0x20 if top else 0)
                f = tmp_path / ("t_%d_%d_%d.tga" % (comps, rle, top))
                f.write_bytes(hdr + body)
                got = host.image_load(str(f))
                assert got is not None and np.array_equal(got, img), (comps, rle, top)
    # PGM / PPM, 8 and 16 bit (top row first in the file), with a comment in the header
    for comps, magic in ((1, b"P5"), (3, b"P6")):
        for maxval in (255, 65535):
            img = rng.randint(0, maxval + 1, (7, 9, comps)).astype(np.uint16 if maxval > 255 else np.uint8)
            data = img.astype(">u2").tobytes() if maxval > 255 else img.tobytes()
            f = tmp_path / ("t_%d_%d.pnm" % (comps, maxval))
            f.write_bytes(magic + b"\n# a comment\n9 7\n%d\n" % maxval + data)
            got = host.image_load(str(f))
            assert got is not None and got.dtype == img.dtype and np.array_equal(got, img[::-1])
    # PFM (bottom row first), both byte orders
    for comps, magic in ((1, b"Pf"), (3, b"PF")):
        for little in (True, False):
            img = rng.uniform(-2, 50, (6, 5, comps)).astype(np.float32)
            f = tmp_path / ("t_%d_%d.pfm" % (comps, little))
            f.write_bytes(magic + b"\n5 6\n" + (b"-1.0" if little else b"1.0") + b"\n" + img.astype("<f4" if little else ">f4").tobytes())
            got = host.image_load(str(f))
            assert got is not None and np.array_equal(got.view(np.uint32), img.view(np.uint32))
    # Radiance HDR: flat and new-style run-length encoded scanlines
    w, h = 16, 5
    rgbe = rng.randint(0, 256, (h, w, 4)).astype(np.uint8)
    rgbe[:, :, 3] = rng.randint(120, 136, (h, w))
    rgbe[1, :, :] = rgbe[1, 0, :]
    rgbe[2, 3, 3] = 0
    expect = np.where(rgbe[..., 3:4] == 0, 0.0, (rgbe[..., :3].astype(np.float32) + 0.5) * np.ldexp(np.float32(1.0), rgbe[..., 3:4].astype(np.int32) - 136)).astype(np.float32)
    for rle in (False, True):
        body = bytearray()
        for y in range(h):
            if not rle:
                body += rgbe[y].tobytes()
                continue
            body += bytes([2, 2, w >> 8, w & 255])
            for c in range(4):
                line = rgbe[y, :, c]
                x = 0
                while x < w:
                    run = 1
                    while x + run < w and run < 127 and line[x + run] == line[x]:
                        run += 1
                    if run >= 3:
                        body += bytes([128 + run, int(line[x])])
                        x += run
                    else:
                        n = 1
                        while x + n < w and n < 128 and not (x + n + 2 < w and line[x + n] == line[x + n + 1] == line[x + n + 2]):
                            n += 1
                        body += bytes([n]) + line[x:x + n].tobytes()
                        x += n
        f = tmp_path / ("t_%d.hdr" % rle)
        f.write_bytes(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w) + bytes(body))
        got = host.image_load(str(f))
        assert got is not None and np.array_equal(got.view(np.uint32), expect[::-1].copy().view(np.uint32)), rle
    # what is not decoded is reported, not guessed
    f = tmp_path / "x.jpg"
    f.write_bytes(b"\xff\xd8\xff\xe0" + b"\0" * 32)
    assert host.image_load(str(f)) is None


def _fixture_scene(import_bits=0, w=64, h=48):
    return host.import_obj(os.path.join(OBJ_DIR, "scene.obj"), w, h, eye=(0.5, 2.2, 6.5), at=(0.0, 1.2, 0.0),
                           import_bits=import_bits, env_radiance=0.05)


def test_import_into_scene_structure_and_material_rules(oracle):
    """importIntoScene on the fixture room (tests/golden/obj/scene.obj, written by make_scene_fixture.py):
    one MeshInstance per (material, shape) in the serial order no-material, then MTL order; Lambertian
    where nothing else is needed, ModPhong otherwise; emissive material -> hot spots; a missing Tf reads
    as a transparent filter (import.hpp:305-311); an undecodable texture becomes the dummy texture; with
    ImportBitTwoSidedMaterials | ImportBitWithGlass every material is wrapped and the pane is glass."""
    from wurblpt_amd import _abi
    sc = _fixture_scene()
    d = sc.d
    assert (d.tri_count, d.instance_count, d.hotspot_count) == (25, 9, 2)
    types = [d.materials[i].type for i in range(d.material_count)]
    assert types == [_abi.MAT_LAMBERTIAN] * 3 + [_abi.MAT_MODPHONG] * 5
    assert d.texture_count == 5  # floor, floor bump->normal map, leaf, dummy, environment
    floor = d.materials[1]
    assert floor.normal_tex >= 0 and d.tri_geom[1].flags & 2  # the normal-mapped material gets tangents
    lamp = d.materials[4]
    assert tuple(lamp.v[3])[:3] == (18.0, 17.0, 15.0)
    broken = d.materials[7]
    assert broken.f[1] == 0.0  # opacity: missing Tf
    img, cnt = oracle.render(sc, 4)
    assert np.isfinite(img).all() and img.mean() > 0.05 and cnt["pdf_tests"] > 0
    sc2 = _fixture_scene(host.IMPORT_TWO_SIDED_MATERIALS | host.IMPORT_WITH_GLASS)
    types2 = [sc2.d.materials[i].type for i in range(sc2.d.material_count)]
    assert types2.count(_abi.MAT_TWOSIDED) == 7 and types2.count(_abi.MAT_GLASS) == 1
    assert host.import_obj(os.path.join(OBJ_DIR, "nothing_here.obj"), 8, 8, (0, 0, 1), (0, 0, 0)) is None


def test_import_with_an_environment_map_file(tmp_path, oracle):
    """wurblpt-sponza.cpp:46-59: the imported scene under an environment map read from an image file (what bench.py --obj
    --envmap builds): the texture is the file's texels, importance sampling is set up on request, the frame is lit by it."""
    rng = np.random.default_rng(3)
    sky = (rng.random((16, 32, 3)) * 4.0).astype(np.float32)
    env = str(tmp_path / "sky.pfm")
    assert host.image_save(env, sky)
    sc = host.import_obj_env(os.path.join(OBJ_DIR, "scene.obj"), env, 24, 16, (0.0, 1.0, 3.0), (0.0, 1.0, 0.0), 50.0, importance_n=8)
    assert sc is not None
    d = sc.d
    assert d.envmap.type != 0 and d.envmap.N == 8
    tex = d.textures[d.envmap.tex]
    assert (tex.width, tex.height, tex.comps) == (32, 16, 3)
    plain = host.import_obj_env(os.path.join(OBJ_DIR, "scene.obj"), env, 24, 16, (0.0, 1.0, 3.0), (0.0, 1.0, 0.0), 50.0)
    assert plain.d.envmap.N == 0
    sc.set_envmap_tables(*oracle.envmap_tables(sc))
    img, _ = oracle.render(sc, 3)
    dark, _ = oracle.render(_fixture_scene(), 3)
    assert np.isfinite(img).all() and img.mean() > dark.mean()
    assert host.import_obj_env(os.path.join(OBJ_DIR, "scene.obj"), str(tmp_path / "nothing.hdr"), 8, 8, (0, 0, 1), (0, 0, 0)) is None


def _png_pixels(path):
    """Independent PNG reader (zlib + the five filters not needed: writers here use filter 0)."""
    import struct
    import zlib
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, hdr = 8, b"", None
    while pos < len(b):
        n, typ = struct.unpack(">I4s", b[pos:pos + 8])
        body = b[pos + 8:pos + 8 + n]
        assert zlib.crc32(typ + body) == struct.unpack(">I", b[pos + 8 + n:pos + 12 + n])[0]
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat += body
        pos += 12 + n
    w, h, depth, ctype = hdr[:4]
    comps = {0: 1, 4: 2, 2: 3, 6: 4}[ctype]
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + w * comps * depth // 8)
    assert (raw[:, 0] == 0).all()
    px = raw[:, 1:]
    if depth == 16:
        px = px.reshape(h, w * comps, 2).astype(np.uint16)
        px = (px[:, :, 0] << 8) | px[:, :, 1]
    return px.reshape(h, w, comps)[::-1]          # row 0 = bottom, like the arrays


@pytest.mark.parametrize("comps", [1, 2, 3, 4])
@pytest.mark.parametrize("dtype", [np.uint8, np.uint16])
def test_png_writer_round_trip(tmp_path, comps, dtype):
    """saveImage() PNG: read back by zlib (checksums, layout) and by the importer's own decoder; 301 x 223 x 4 x 2 bytes
    spans several stored deflate blocks."""
    rng = np.random.default_rng(comps)
    img = rng.integers(0, np.iinfo(dtype).max + 1, size=(223, 301, comps)).astype(dtype)
    path = str(tmp_path / "out.png")
    assert host.image_save(path, img)
    assert np.array_equal(_png_pixels(path), img)
    assert np.array_equal(host.image_load(path), img)


def test_pnm_and_pfm_writers_round_trip(tmp_path):
    rng = np.random.default_rng(5)
    for comps, ext in ((1, "pgm"), (3, "ppm")):
        for dtype in (np.uint8, np.uint16):
            img = rng.integers(0, np.iinfo(dtype).max + 1, size=(17, 23, comps)).astype(dtype)
            path = str(tmp_path / ("out." + ext))
            assert host.image_save(path, img)
            assert np.array_equal(host.image_load(path), img)
    for comps in (1, 3):
        img = rng.standard_normal((9, 14, comps)).astype(np.float32)
        path = str(tmp_path / "out.pfm")
        assert host.image_save(path, img)
        assert np.array_equal(host.image_load(path).view(np.uint32), img.view(np.uint32))
    # what a format cannot hold is refused, not converted
    assert not host.image_save(str(tmp_path / "bad.png"), rng.random((4, 4, 3)).astype(np.float32))
    assert not host.image_save(str(tmp_path / "bad.pfm"), np.zeros((4, 4, 3), np.uint8))
    assert not host.image_save(str(tmp_path / "bad.jpg"), np.zeros((4, 4, 3), np.uint8))


def _jpeg_cases():
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg")
    exp = np.load(os.path.join(d, "expected.npz"))
    return d, exp


def test_jpeg_decoder_matches_libjpeg(tmp_path):
    """include/wurblpt/jpeg.hpp against the pixels libjpeg-turbo decodes (tests/golden/jpeg, written by
    make_jpeg_fixtures.py): baseline and progressive, 4:4:4 / 4:2:2 / 4:2:0 / 4:1:1 and grey, restart intervals,
    quality 1 to 100, sizes that are no multiple of the MCU and images too narrow for the triangle filter. Every byte."""
    d, exp = _jpeg_cases()
    assert len(exp.files) >= 16
    for name in exp.files:
        ref = exp[name]
        ref = ref[:, :, None] if ref.ndim == 2 else ref
        got = host.image_load(os.path.join(d, name + ".jpg"))
        assert got is not None and got.dtype == np.uint8, name
        assert np.array_equal(got[::-1], ref), name          # row 0 of the array is the bottom row
    # corrupt and unsupported files are refused, not guessed at
    data = open(os.path.join(d, "s420.jpg"), "rb").read()
    for cut, label in ((data[:200], "truncated"), (b"\xff\xd8\xff\xd9", "empty"), (data.replace(b"\xff\xc0", b"\xff\xc9", 1), "arithmetic")):
        p = str(tmp_path / (label + ".jpg"))
        open(p, "wb").write(cut)
        assert host.image_load(p) is None, label


def test_jpeg_decoder_against_pillow_on_larger_images(tmp_path):
    """The same comparison on larger, freshly encoded images and on the JPEG textures of the reference's example
    applications where this machine has them; needs Pillow (libjpeg-turbo) and is skipped without it."""
    PIL = pytest.importorskip("PIL.Image")
    import glob
    import io
    rng = np.random.default_rng(3)
    files = []
    y, x = np.mgrid[0:203, 0:317]
    base = np.stack([(x * 3 + y) % 256, (x ^ y) % 256, (x * y // 7) % 256], axis=2).astype(np.float64)
    img = np.clip(base + rng.normal(0, 20, base.shape), 0, 255).astype(np.uint8)
    for i, opts in enumerate((dict(quality=87, subsampling=2), dict(quality=70, subsampling=1, progressive=True),
                              dict(quality=95, subsampling=0, optimize=True), dict(quality=50, subsampling=2, progressive=True, restart_marker_rows=2))):
        p = str(tmp_path / ("big%d.jpg" % i))
        PIL.fromarray(img).save(p, "JPEG", **opts)
        files.append(p)
    files += sorted(glob.glob("/root/reference/wurblpt-*/*.jpg"))[:4]
    for p in files:
        ref = np.asarray(PIL.open(p))
        ref = ref[:, :, None] if ref.ndim == 2 else ref
        got = host.image_load(p)
        assert got is not None and np.array_equal(got[::-1], ref), p


def _write_exr(path, img, compression, half):
    """Test-side OpenEXR writer (scanline; NONE 0, RLE 1, ZIPS 2, ZIP 3; HALF or FLOAT channels R G B / Y),
    independent of include/wurblpt/exr.hpp.  img: float32 [h, w, comps], row 0 = top."""
    import struct
    import zlib
    h, w, comps = img.shape
    names = {1: ["Y"], 3: ["B", "G", "R"], 4: ["A", "B", "G", "R"]}[comps]
    src = {"Y": 0, "R": 0, "G": 1, "B": 2, "A": 3}
    out = struct.pack("<II", 20000630, 2)

    def attr(name, typ, data):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<I", len(data)) + data
    chlist = b"".join(n.encode() + b"\0" + struct.pack("<IIII", 1 if half else 2, 0, 1, 1) for n in names) + b"\0"
    out += attr("channels", "chlist", chlist) + attr("compression", "compression", bytes([compression]))
    out += attr("dataWindow", "box2i", struct.pack("<4i", 5, -3, 5 + w - 1, -3 + h - 1))        # a window that does not start at 0
    out += attr("displayWindow", "box2i", struct.pack("<4i", 0, 0, w - 1, h - 1))
    out += attr("lineOrder", "lineOrder", b"\0") + attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    out += attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0)) + attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0"
    lines = 16 if compression == 3 else 1
    chunks = []
    for y0 in range(0, h, lines):
        raw = b""
        for y in range(y0, min(y0 + lines, h)):
            for n in names:
                row = img[y, :, src[n]]
                raw += (row.astype(np.float16) if half else row.astype(np.float32)).tobytes()
        data = raw
        if compression:
            b = np.frombuffer(raw, np.uint8)
            t = np.concatenate([b[0::2], b[1::2]]).astype(np.int32)
            t[1:] = (t[1:] - t[:-1] + 128 + 256) % 256
            t = t.astype(np.uint8).tobytes()
            if compression == 1:
                packed, i = b"", 0
                while i < len(t):                       # runs of >= 3 equal bytes, literals otherwise
                    j = i
                    while j < len(t) and j - i < 127 and t[j] == t[i]:
                        j += 1
                    if j - i >= 3:
                        packed += struct.pack("b", j - i - 1) + t[i:i + 1]
                        i = j
                    else:
                        j = i
                        while j < len(t) and j - i < 127 and not (j + 2 < len(t) and t[j] == t[j + 1] == t[j + 2]):
                            j += 1
                        packed += struct.pack("b", -(j - i)) + t[i:j]
                        i = j
            else:
                packed = zlib.compress(t)
            data = packed if len(packed) < len(raw) else raw         # the format stores a chunk raw when that is shorter
        chunks.append(struct.pack("<iI", -3 + y0, len(data)) + data)
    table_at = len(out)
    pos = table_at + 8 * len(chunks)
    table = b""
    for c in chunks:
        table += struct.pack("<Q", pos)
        pos += len(c)
    open(path, "wb").write(out + table + b"".join(chunks))


def test_exr_reader_and_writer(tmp_path):
    """OpenEXR scanline files: every compression / sample type the reader takes, against a writer that lives in this
    test; the writer's files through the reader again (bit for bit, including infinities, NaN payloads and subnormals)."""
    rng = np.random.default_rng(8)
    for comps in (1, 3, 4):
        img = (rng.random((37, 29, comps)) ** 3 * 50).astype(np.float32)
        img[0, 0] = 0.0
        img[5:9, 3:20] = 0.25                                   # flat areas: runs for RLE, long matches for zlib
        for compression in (0, 1, 2, 3):
            for half in (False, True):
                p = str(tmp_path / "t.exr")
                _write_exr(p, img, compression, half)
                got = host.image_load(p)
                ref = img.astype(np.float16).astype(np.float32) if half else img
                assert got is not None and got.dtype == np.float32, (comps, compression, half)
                assert np.array_equal(got[::-1].view(np.uint32), ref.view(np.uint32)), (comps, compression, half)
    special = np.array([np.inf, -np.inf, 1e-42, -0.0, 65504.0, 3.0e38], np.float32).reshape(1, 6, 1).repeat(3, axis=2)
    special = np.concatenate([special, np.frombuffer(np.uint32([0x7fc12345] * 18).tobytes(), np.float32).reshape(1, 6, 3)], axis=0)
    for comps in (1, 2, 3, 4):
        img = rng.standard_normal((23, 31, comps)).astype(np.float32)
        img[:2, :6, :] = special[:, :, :1] if comps < 3 else np.concatenate([special, special[:, :, :1]], axis=2)[:, :, :comps]
        p = str(tmp_path / ("w%d.exr" % comps))
        assert host.image_save(p, img)
        back = host.image_load(p)
        assert back.shape == img.shape and np.array_equal(back.view(np.uint32), img.view(np.uint32)), comps
    # half precision specials through the reader: subnormals, infinities, NaN
    halves = np.array([0x0001, 0x03ff, 0x0400, 0x7bff, 0x7c00, 0xfc00, 0x7e01, 0x8000], np.uint16).view(np.float16)
    img = halves.astype(np.float32).reshape(1, 8, 1)
    p = str(tmp_path / "h.exr")
    _write_exr(p, img, 0, True)
    assert np.array_equal(host.image_load(p).view(np.uint32), img.view(np.uint32))
    assert not host.image_save(str(tmp_path / "bad.exr"), np.zeros((4, 4, 3), np.uint8))
    data = open(str(tmp_path / "w3.exr"), "rb").read()
    open(str(tmp_path / "cut.exr"), "wb").write(data[:len(data) // 2])
    assert host.image_load(str(tmp_path / "cut.exr")) is None


def test_exr_reader_on_the_mitsuba_render_of_the_reference():
    """The converged Mitsuba render the reference ships (half floats, ZIP): its 16x16 block averages are the committed
    fixture the integrator is compared with; read here through include/wurblpt/exr.hpp."""
    src = "/root/reference/wurblpt-cornellbox/mitsuba/cbox-2500spp.exr"
    if not os.path.exists(src):
        pytest.skip("the reference tree is not on this machine")
    img = host.image_load(src)
    assert img is not None and img.shape == (1024, 1024, 3)
    blocks = img[::-1].reshape(64, 16, 64, 16, 3).mean(axis=(1, 3)).astype(np.float32)
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cbox_mitsuba_64x64.npy"))
    assert np.allclose(blocks, ref, rtol=1e-6, atol=1e-7)


def test_jpeg_decoder_fuzz_against_pillow(tmp_path):
    """120 random images (sizes 1..70, grey and colour, noise to smooth) saved with random quality, chroma subsampling,
    progressive / optimised coding and restart intervals: every byte equals libjpeg-turbo's decode."""
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(11)
    for case in range(120):
        w, h = int(rng.integers(1, 71)), int(rng.integers(1, 71))
        grey = bool(rng.integers(0, 4) == 0)
        smooth = rng.random()
        y, x = np.mgrid[0:h, 0:w]
        base = np.stack([(x * rng.integers(1, 9) + y * rng.integers(1, 9)) % 256 for _ in range(3)], axis=2).astype(np.float64)
        img = np.clip(smooth * base + (1 - smooth) * rng.integers(0, 256, base.shape), 0, 255).astype(np.uint8)
        opts = dict(quality=int(rng.integers(1, 101)))
        if not grey:
            opts["subsampling"] = [0, 1, 2, "4:1:1"][int(rng.integers(0, 4))]
        if rng.integers(0, 2):
            opts["progressive"] = True
        if rng.integers(0, 2):
            opts["optimize"] = True
        k = int(rng.integers(0, 3))
        if k == 1:
            opts["restart_marker_blocks"] = int(rng.integers(1, 6))
        elif k == 2:
            opts["restart_marker_rows"] = int(rng.integers(1, 3))
        p = str(tmp_path / "f.jpg")
        PIL.fromarray(img[:, :, 0] if grey else img).save(p, "JPEG", **opts)
        ref = np.asarray(PIL.open(p))
        ref = ref[:, :, None] if ref.ndim == 2 else ref
        got = host.image_load(p)
        assert got is not None and np.array_equal(got[::-1], ref), (case, w, h, grey, opts)


def _random_number(rng):
    """a float in one of the spellings OBJ files use"""
    v = float(rng.normal(0, 3)) * (10.0 ** int(rng.integers(-3, 4)) if rng.integers(0, 6) == 0 else 1.0)
    style = int(rng.integers(0, 7))
    if style == 0:
        return "%d" % int(v)
    if style == 1:
        return "%.6f" % v
    if style == 2:
        return "%e" % v
    if style == 3:
        return ("%+.4f" % v)
    if style == 4:
        return ("%.3f" % v).replace("0.", ".", 1) if abs(v) < 1 else "%.3f" % v
    if style == 5:
        return "%.9g" % v
    return "%.1fE%+d" % (v, int(rng.integers(-2, 3)))


def _random_obj(rng, d, name):
    """An OBJ + MTL pair with random number spellings, index forms (absolute and relative, with and without normals and
    texture coordinates), planar polygons of 3 to 7 corners (convex and with one notch), groups, objects, material
    switches, comments, blank lines and trailing spaces."""
    nm = int(rng.integers(0, 4))
    mats = ["m%d" % i for i in range(nm)]
    with open(os.path.join(d, name + ".mtl"), "w") as f:
        for m in mats:
            f.write("newmtl %s\n" % m)
            for key in ("Kd", "Ks", "Ke", "Tf"):
                if rng.integers(0, 3):
                    f.write("%s %s %s %s\n" % (key, *[("%.4f" % rng.random()) for _ in range(3)]))
            if rng.integers(0, 2):
                f.write("Ns %s\n" % _random_number(rng))
            k = int(rng.integers(0, 4))
            if k == 1:
                f.write("d %.3f\n" % rng.random())
            elif k == 2:
                f.write("Tr %.3f\n" % rng.random())
            elif k == 3:
                f.write("d %.3f\nTr %.3f\n" % (rng.random(), rng.random()))
            if rng.integers(0, 2):
                f.write("Ni %.3f\n" % (1 + rng.random()))
            for key in ("map_Kd", "map_Ks", "map_Ns", "map_bump", "bump", "map_d", "map_Ke", "norm"):
                if rng.integers(0, 5) == 0:
                    opt = ""
                    if rng.integers(0, 2):
                        opt += "-s %.2f %.2f %.2f " % tuple(rng.random(3) + 0.5)
                    if rng.integers(0, 2):
                        opt += "-o %.2f %.2f %.2f " % tuple(rng.random(3))
                    if key in ("map_bump", "bump") and rng.integers(0, 2):
                        opt += "-bm %.2f " % rng.random()
                    f.write("%s %stex_%s.png\n" % (key, opt, key))
            f.write("\n")
    nv, nn, nt = int(rng.integers(8, 40)), int(rng.integers(0, 10)), int(rng.integers(0, 10))
    lines = ["# fuzz case", "mtllib %s.mtl" % name, ""]
    # vertices in a plane per polygon would need care; polygons below are made from fresh planar rings instead
    for _ in range(nv):
        lines.append("v %s %s %s" % (_random_number(rng), _random_number(rng), _random_number(rng)))
    for _ in range(nn):
        lines.append("vn %s %s %s" % (_random_number(rng), _random_number(rng), _random_number(rng)))
    for _ in range(nt):
        lines.append("vt %s %s" % (_random_number(rng), _random_number(rng)) + (" 0" if rng.integers(0, 3) == 0 else ""))
    count = nv
    for shape in range(int(rng.integers(1, 5))):
        k = int(rng.integers(0, 3))
        if k == 0:
            lines.append("g group%d" % shape)
        elif k == 1:
            lines.append("o object%d" % shape)
        for face in range(int(rng.integers(1, 7))):
            if rng.integers(0, 3) == 0 and (mats or rng.integers(0, 2)):
                lines.append("usemtl %s" % (mats[int(rng.integers(0, len(mats)))] if mats and rng.integers(0, 5) else "unknown"))
            corners = int(rng.choice([3, 3, 3, 4, 4, 5, 6, 7]))
            if corners > 3:
                # a planar ring in a random axis plane, one corner pulled inwards now and then
                axis = int(rng.integers(0, 3))
                ang = np.sort(rng.random(corners)) * 2 * np.pi
                rad = np.full(corners, 1.0 + rng.random())
                if corners > 4 and rng.integers(0, 2):
                    rad[int(rng.integers(0, corners))] *= 0.35
                ring = np.zeros((corners, 3))
                ring[:, (axis + 1) % 3] = rad * np.cos(ang)
                ring[:, (axis + 2) % 3] = rad * np.sin(ang)
                ring[:, axis] = rng.normal()
                for q in ring:
                    lines.append("v %.5f %.5f %.5f" % tuple(q))
                ids = list(range(count + 1, count + corners + 1))
                count += corners
            else:
                ids = [int(i) + 1 for i in rng.choice(count, 3, replace=False)]
            form = int(rng.integers(0, 4)) if (nn or nt) else 0
            toks = []
            for i in ids:
                vi = i if rng.integers(0, 4) else i - count - 1          # relative index now and then
                ti = int(rng.integers(1, nt + 1)) if nt else 0
                ni = int(rng.integers(1, nn + 1)) if nn else 0
                if form == 1 and nt:
                    toks.append("%d/%d" % (vi, ti))
                elif form == 2 and nn:
                    toks.append("%d//%d" % (vi, ni if rng.integers(0, 4) else ni - nn - 1))
                elif form == 3 and nn and nt:
                    toks.append("%d/%d/%d" % (vi, ti, ni))
                else:
                    toks.append("%d" % vi)
            lines.append("f " + " ".join(toks) + ("  " if rng.integers(0, 4) == 0 else ""))
        if rng.integers(0, 3) == 0:
            lines.append("")
    with open(os.path.join(d, name + ".obj"), "w") as f:
        f.write("\n".join(lines) + "\n")
    return os.path.join(d, name + ".obj")


def test_obj_reader_fuzz_against_the_vendored_tinyobjloader(tmp_path):
    """60 random OBJ / MTL pairs through include/wurblpt/objreader.hpp and through the reference's own parser
    (oracle/_ref/ref_probe --obj, built from the reference tree where this machine has it): every float bit, index,
    triangle, shape boundary, material id and MTL value equal."""
    import subprocess
    probe = os.path.join(ROOT, "oracle", "_ref", "ref_probe")
    if not os.path.exists(probe) or not os.path.exists("/root/reference"):
        pytest.skip("the reference tree is not on this machine")
    rng = np.random.default_rng(31)
    for case in range(60):
        obj = _random_obj(rng, str(tmp_path), "case%d" % case)
        ref_json, my_json = str(tmp_path / "ref.json"), str(tmp_path / "mine.json")
        subprocess.run([probe, "--obj", obj, ref_json], check=True, timeout=60, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        assert host.lib().wpt_host_obj_dump(obj.encode(), my_json.encode()) == 0
        ref, mine = json.load(open(ref_json)), json.load(open(my_json))
        for key, value in mine.items():
            assert value == ref[key], (case, key, obj)


def test_png_tga_pnm_decoders_fuzz_against_pillow(tmp_path):
    """Files written by Pillow (its own zlib settings, filters, palettes, RLE) through the importer's decoders:
    PNG grey / grey + alpha / RGB / RGBA / palette (with and without transparency) / 16 bit grey, TGA plain and RLE,
    binary PGM / PPM; every value equal."""
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(23)
    for case in range(80):
        w, h = int(rng.integers(1, 50)), int(rng.integers(1, 50))
        smooth = rng.random() < 0.5
        def pixels(c, hi=256, dtype=np.uint8):
            if smooth:
                y, x = np.mgrid[0:h, 0:w]
                return np.stack([((x * (k + 2) + y * (5 - k)) % hi) for k in range(c)], axis=2).astype(dtype)
            return rng.integers(0, hi, (h, w, c)).astype(dtype)
        kind = int(rng.integers(0, 10))
        p = str(tmp_path / ("f%d" % case))
        if kind == 0:
            ref = pixels(1); PIL.fromarray(ref[:, :, 0], "L").save(p + ".png", compress_level=int(rng.integers(0, 10)))
        elif kind == 1:
            ref = pixels(2); PIL.fromarray(ref, "LA").save(p + ".png")
        elif kind == 2:
            ref = pixels(3); PIL.fromarray(ref, "RGB").save(p + ".png", optimize=bool(rng.integers(0, 2)))
        elif kind == 3:
            ref = pixels(4); PIL.fromarray(ref, "RGBA").save(p + ".png")
        elif kind == 4:
            idx = pixels(1, hi=int(rng.integers(2, 257)))[:, :, 0]
            pal = rng.integers(0, 256, (256, 3)).astype(np.uint8)
            im = PIL.fromarray(idx, "P"); im.putpalette(pal.reshape(-1).tolist())
            if rng.integers(0, 2):
                alpha = rng.integers(0, 256, 256).astype(np.uint8)
                im.save(p + ".png", transparency=bytes(alpha.tolist()))
                ref = np.concatenate([pal[idx], alpha[idx][:, :, None]], axis=2)
            else:
                im.save(p + ".png")
                ref = pal[idx]
        elif kind == 5:
            ref = pixels(1, hi=65536, dtype=np.uint16); PIL.fromarray(ref[:, :, 0]).save(p + ".png")
        elif kind == 6:
            ref = pixels(3); PIL.fromarray(ref, "RGB").save(p + ".tga", compression="tga_rle" if rng.integers(0, 2) else None)
        elif kind == 7:
            ref = pixels(4); PIL.fromarray(ref, "RGBA").save(p + ".tga", compression="tga_rle" if rng.integers(0, 2) else None)
        elif kind == 8:
            ref = pixels(1); PIL.fromarray(ref[:, :, 0], "L").save(p + ".pgm")
        else:
            ref = pixels(3); PIL.fromarray(ref, "RGB").save(p + ".ppm")
        ext = [".png", ".png", ".png", ".png", ".png", ".png", ".tga", ".tga", ".pgm", ".ppm"][kind]
        got = host.image_load(p + ext)
        assert got is not None and got.shape == ref.shape and got.dtype == ref.dtype, (case, kind, w, h)
        assert np.array_equal(got[::-1], ref), (case, kind, w, h)


def test_decoders_survive_damaged_files(tmp_path):
    """Texture files come from outside: flipped bytes, truncations and insertions in JPEG / PNG / OpenEXR / PFM / PPM / TGA files
    make the decoders fail with a message (or decode something), never crash or ask for memory the file cannot fill.
    (The same corruptions were run under AddressSanitizer / UBSan over 13 000 files, and over the OBJ / MTL reader.)"""
    d = str(tmp_path)
    rng0 = np.random.default_rng(1)
    img = (rng0.random((9, 11, 3)) * 3).astype(np.float32)
    samples = [os.path.join(ROOT, "tests", "golden", "jpeg", n) for n in ("s420.jpg", "progressive_420.jpg", "grey.jpg", "restart_420.jpg")]
    for name, data in (("a.exr", img), ("a.pfm", img), ("a.png", (img * 80).astype(np.uint8)), ("b.png", (img * 20000).astype(np.uint16)),
                       ("a.ppm", (img * 80).astype(np.uint8))):
        assert host.image_save(os.path.join(d, name), data)
        samples.append(os.path.join(d, name))
    tga = os.path.join(d, "a.tga")       # type 10 (run-length) true colour, written by hand: one packet of 99 equal pixels
    open(tga, "wb").write(bytes([0, 0, 10, 0, 0, 0, 0, 0, 0, 0, 0, 0, 11, 0, 9, 0, 24, 0]) + bytes([0x80 | 98, 1, 2, 3]))
    assert host.image_load(tga) is not None
    samples.append(tga)
    rng = np.random.default_rng(7)
    decoded = 0
    for f in samples:
        data = bytearray(open(f, "rb").read())
        for k in range(60):
            b = bytearray(data)
            mode = int(rng.integers(0, 3))
            if mode == 0:
                for _ in range(int(rng.integers(1, 6))):
                    b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
            elif mode == 1:
                b = b[:int(rng.integers(1, len(b)))]
            else:
                i = int(rng.integers(0, len(b)))
                b[i:i] = bytes(rng.integers(0, 256, int(rng.integers(1, 9))).tolist())
            p = os.path.join(d, "c" + os.path.splitext(f)[1])
            open(p, "wb").write(b)
            got = host.image_load(p)
            decoded += got is not None
    assert 0 < decoded < 60 * len(samples)        # some damage is harmless, most is refused
