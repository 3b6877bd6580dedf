"""Writes small synthetic BRDF files in the tensor-file container that the RGL material database
uses (header "tensor_file", version 1.0, fields with name / rank / dtype / offset / shape; the
layout that powitacq_rgb.inl:728-803 of the reference reads).  The values are smooth positive
random fields, not measurements: they exercise every code path of the model (isotropic and
anisotropic parameterisation, sampling, inversion, evaluation), which is all a parity fixture
needs.  The real *.bsdf files of the database are not available offline.

usage: python tests/golden/make_rgl_fixture.py    (rewrites tests/golden/synthetic_*.bsdf)"""
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DTYPE = {np.dtype(np.uint8): 1, np.dtype(np.float32): 10}


def write_tensor_file(path, fields):
    """fields: list of (name, ndarray)"""
    header = b"tensor_file\0" + bytes([1, 0]) + struct.pack("<I", len(fields))
    table = 0
    for name, a in fields:
        table += 2 + len(name) + 2 + 1 + 8 + 8 * a.ndim
    offset = len(header) + table
    out = bytearray(header)
    blobs = []
    for name, a in fields:
        a = np.ascontiguousarray(a)
        offset = (offset + 63) // 64 * 64
        out += struct.pack("<H", len(name)) + name.encode() + struct.pack("<H", a.ndim) + bytes([DTYPE[a.dtype]]) + struct.pack("<Q", offset)
        for s in a.shape:
            out += struct.pack("<Q", s)
        blobs.append((offset, a.tobytes()))
        offset += a.nbytes
    for off, b in blobs:
        out += b"\0" * (off - len(out))
        out += b
    open(path, "wb").write(bytes(out))


def smooth(rng, shape, lo, hi):
    """positive, smooth along the last two axes"""
    a = rng.uniform(lo, hi, shape).astype(np.float64)
    for axis in (-1, -2):
        a = (np.roll(a, 1, axis) + 2 * a + np.roll(a, -1, axis)) / 4
    return a.astype(np.float32)


def make(path, seed, n_phi, n_theta, res, ndf_res, jacobian):
    rng = np.random.default_rng(seed)
    theta_i = np.linspace(0.0, np.pi / 2, n_theta).astype(np.float32)
    if n_phi == 1:
        phi_i = np.zeros(1, np.float32)
    else:
        phi_i = np.linspace(-np.pi, np.pi, n_phi).astype(np.float32)
    # a lobe around the pole of the unit square plus noise, so that densities vary by two orders of magnitude
    y, x = np.meshgrid(np.linspace(0, 1, res), np.linspace(0, 1, res), indexing="ij")
    lobe = (0.05 + np.exp(-6.0 * x)).astype(np.float32)
    vndf = smooth(rng, (n_phi, n_theta, res, res), 0.3, 1.0) * lobe
    luminance = smooth(rng, (n_phi, n_theta, res, res), 0.2, 1.0)
    rgb = smooth(rng, (n_phi, n_theta, 3, res, res), 0.05, 0.9)
    rgb[0, 0, 1, 0, 0] = -0.02  # an out-of-gamut value: the model clips it
    yy, xx = np.meshgrid(np.linspace(0, 1, ndf_res), np.linspace(0, 1, ndf_res), indexing="ij")
    ndf = (smooth(rng, (ndf_res, ndf_res), 0.5, 1.0) * (0.02 + 4.0 * np.exp(-5.0 * xx))).astype(np.float32)
    sigma = smooth(rng, (ndf_res + 1, ndf_res + 1), 0.4, 1.0)
    fields = [("description", np.frombuffer(b"synthetic parity fixture", dtype=np.uint8)),
              ("theta_i", theta_i), ("phi_i", phi_i), ("ndf", ndf), ("sigma", sigma), ("vndf", vndf),
              ("luminance", luminance), ("rgb", rgb), ("jacobian", np.array([jacobian], np.uint8))]
    write_tensor_file(path, fields)


if __name__ == "__main__":
    make(os.path.join(HERE, "synthetic_iso.bsdf"), 11, 1, 5, 8, 9, 1)
    make(os.path.join(HERE, "synthetic_aniso.bsdf"), 12, 5, 4, 6, 7, 0)
    for f in ("synthetic_iso.bsdf", "synthetic_aniso.bsdf"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
