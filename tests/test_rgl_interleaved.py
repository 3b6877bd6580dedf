"""The device keeps, per measured BRDF, the colour and luminance warps' values interleaved (red, green, blue, luminance of a grid
point in one 16-byte record: wpt_capi.hip builds the table at upload, wpt_rgl.h::rglColourInterleaved reads it) so that an evaluation's
look-ups into those two warps come from a quarter of the cache lines.  The reference's arithmetic is untouched: on the host,
BRDF::sample / eval / pdf (powitacq_rgb.inl:1016-1183 as restated in wpt_rgl.h, whose plain path ref_probe's golden vectors pin to the
reference) give the same bits through either table, for an isotropic and an anisotropic tensor file."""
import importlib.util
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_interleaved_table_gives_the_same_bits(tmp_path):
    spec = importlib.util.spec_from_file_location("make_rgl_fixture", os.path.join(ROOT, "tests", "golden", "make_rgl_fixture.py"))
    fx = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fx)
    iso, aniso = str(tmp_path / "iso.bsdf"), str(tmp_path / "aniso.bsdf")
    fx.make(iso, 21, 1, 8, 32, 64, 1)     # the shapes bench.py's Bistro-class workload uses
    fx.make(aniso, 22, 8, 8, 32, 64, 1)
    exe = str(tmp_path / "rgl_interleaved")
    subprocess.run(["g++", "-std=c++20", "-O2", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "rgl_interleaved.cpp"),
                    "-o", exe], check=True, timeout=600)
    r = subprocess.run([exe, iso, aniso], capture_output=True, timeout=600)
    assert r.returncode == 0 and b"mismatches: 0" in r.stdout, r.stdout.decode() + r.stderr.decode()
