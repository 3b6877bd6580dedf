"""Generates tests/golden/cbox_mitsuba_64x64.npy from the converged Mitsuba render that ships
with the reference (wurblpt-cornellbox/mitsuba/cbox-2500spp.exr: 1024x1024, half float
B,G,R channels, ZIP scanline blocks, stored top-down): 16x16 block averages, RGB order,
float32, 48 KiB.  Run in the build container (needs /root/reference)."""
import struct
import sys
import zlib

import numpy as np

SRC = "/root/reference/wurblpt-cornellbox/mitsuba/cbox-2500spp.exr"


def read_exr(path):
    data = open(path, "rb").read()
    assert struct.unpack("<I", data[:4])[0] == 20000630
    pos = 8
    attrs = {}
    while data[pos] != 0:
        e = data.index(b"\0", pos); name = data[pos:e].decode(); pos = e + 1
        e = data.index(b"\0", pos); typ = data[pos:e].decode(); pos = e + 1
        size = struct.unpack("<I", data[pos:pos + 4])[0]; pos += 4
        attrs[name] = (typ, data[pos:pos + size]); pos += size
    pos += 1
    chans = []
    c = attrs["channels"][1]
    p = 0
    while c[p] != 0:
        e = c.index(b"\0", p); chans.append((c[p:e].decode(), struct.unpack("<I", c[e + 1:e + 5])[0])); p = e + 17
    xmin, ymin, xmax, ymax = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = xmax - xmin + 1, ymax - ymin + 1
    comp = attrs["compression"][1][0]
    assert comp in (2, 3), comp  # ZIPS (1 line) / ZIP (16 lines)
    lines = 1 if comp == 2 else 16
    nblocks = (h + lines - 1) // lines
    offsets = struct.unpack("<%dQ" % nblocks, data[pos:pos + 8 * nblocks])
    assert all(t == 1 for _, t in chans)  # half
    img = np.zeros((h, w, len(chans)), np.float32)
    for off in offsets:
        y, size = struct.unpack("<iI", data[off:off + 8])
        raw = zlib.decompress(data[off + 8:off + 8 + size])
        b = np.frombuffer(raw, np.uint8).astype(np.int32)
        b = np.cumsum(np.concatenate(([b[0]], b[1:] - 128))).astype(np.uint8)  # EXR predictor
        half = (len(b) + 1) // 2
        out = np.empty(len(b), np.uint8)
        out[0::2] = b[:half]
        out[1::2] = b[half:]
        n = min(lines, h - (y - ymin))
        arr = out.view(np.float16).reshape(n, len(chans), w)
        img[y - ymin:y - ymin + n] = arr.transpose(0, 2, 1)
    return img, [c for c, _ in chans]


if __name__ == "__main__":
    img, chans = read_exr(SRC)
    rgb = np.stack([img[:, :, chans.index(c)] for c in ("R", "G", "B")], axis=2)
    h, w, _ = rgb.shape
    blocks = rgb.reshape(64, h // 64, 64, w // 64, 3).mean(axis=(1, 3)).astype(np.float32)
    np.save(sys.argv[1] if len(sys.argv) > 1 else "tests/golden/cbox_mitsuba_64x64.npy", blocks)
    print("channel means", rgb.reshape(-1, 3).mean(0), "->", blocks.shape)
