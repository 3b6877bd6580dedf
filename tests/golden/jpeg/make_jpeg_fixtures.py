"""Writes the JPEG fixtures of tests/test_import.py: small files in every coding mode the decoder
(include/wurblpt/jpeg.hpp) handles, and the pixels libjpeg-turbo (through Pillow) decodes from them.

    python tests/golden/jpeg/make_jpeg_fixtures.py        # needs Pillow; rewrites *.jpg and expected.npz
"""
import io
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))


def picture(w, h, seed, grey=False):
    """Smooth gradients, hard edges and noise: exercises DC prediction, long zero runs and dense blocks."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.zeros((h, w, 3))
    img[..., 0] = 127 + 120 * np.sin(x / 5.0 + seed) * np.cos(y / 7.0)
    img[..., 1] = (x * 255 / max(w - 1, 1) + y * 40) % 256
    img[..., 2] = 255 * (((x // 6) + (y // 4)) % 2)
    img += rng.normal(0, 12, img.shape)
    img = np.clip(img, 0, 255).astype(np.uint8)
    return img[..., 0] if grey else img


CASES = {
    # name: (width, height, grey, save options)
    "s444": (37, 29, False, dict(quality=90, subsampling=0)),
    "s422": (37, 29, False, dict(quality=85, subsampling=1)),
    "s420": (37, 29, False, dict(quality=80, subsampling=2)),
    "s411": (45, 19, False, dict(quality=88, subsampling="4:1:1")),
    "grey": (33, 18, True, dict(quality=75)),
    "progressive_420": (53, 41, False, dict(quality=82, subsampling=2, progressive=True)),
    "progressive_444_optimized": (40, 24, False, dict(quality=93, subsampling=0, progressive=True, optimize=True)),
    "progressive_grey": (31, 31, True, dict(quality=60, progressive=True)),
    "restart_420": (64, 48, False, dict(quality=85, subsampling=2, restart_marker_blocks=3)),
    "restart_progressive": (50, 34, False, dict(quality=85, subsampling=1, progressive=True, restart_marker_rows=1)),
    "low_quality": (48, 48, False, dict(quality=8, subsampling=2)),
    "max_quality": (24, 16, False, dict(quality=100, subsampling=0)),
    "one_pixel": (1, 1, False, dict(quality=90, subsampling=2)),
    "two_wide_420": (2, 5, False, dict(quality=90, subsampling=2)),
    "three_wide_422": (3, 2, False, dict(quality=90, subsampling=1)),
    "quality_1": (32, 32, False, dict(quality=1, subsampling=0)),
}


def main():
    expected = {}
    for seed, (name, (w, h, grey, opts)) in enumerate(sorted(CASES.items())):
        buf = io.BytesIO()
        Image.fromarray(picture(w, h, seed, grey)).save(buf, "JPEG", **opts)
        data = buf.getvalue()
        with open(os.path.join(HERE, name + ".jpg"), "wb") as f:
            f.write(data)
        expected[name] = np.asarray(Image.open(io.BytesIO(data)))
        print(name, len(data), "bytes", expected[name].shape)
    np.savez_compressed(os.path.join(HERE, "expected.npz"), **expected)


if __name__ == "__main__":
    main()
