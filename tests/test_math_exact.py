"""wurblpt_amd/csrc/wpt_math.h against the C library the reference's golden vectors were computed with (this image's
glibc 2.35 on an x86-64 with FMA): the same bits for EVERY float argument of sinf, cosf, expf, asinf, acosf, atanf and of the
measured-BRDF model's float(2 * asin(double)), and for 2 x 10^8 argument pairs plus the special values of powf and atan2f.
That removes the hop between "GPU == restatement" and "restatement with libm == reference" (DESIGN.md section 2): the
kernels and the oracle's default back end evaluate this header, the oracle's libm back end calls the library, and the
two now are the same function.  Runs where the library is this one (skipped elsewhere: other C libraries, or an x86-64
without FMA, select other builds of these functions)."""
import os
import platform
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def this_is_the_library():
    if platform.machine() != "x86_64" or "fma" not in open("/proc/cpuinfo").read():
        return False
    try:
        return platform.libc_ver()[0] == "glibc" and tuple(int(p) for p in platform.libc_ver()[1].split(".")[:2]) == (2, 35)
    except Exception:
        return False


@pytest.mark.skipif(not this_is_the_library(), reason="not glibc 2.35 on x86-64 with FMA")
def test_math_header_equals_the_c_library_bit_for_bit(tmp_path):
    exe = str(tmp_path / "math_exact")
    subprocess.run(["g++", "-O2", "-fopenmp", "-mfma", "-ffp-contract=off", "-fno-builtin", os.path.join(ROOT, "tests", "math_exact.cpp"),
                    "-o", exe, "-lm"], check=True, timeout=600)
    r = subprocess.run([exe, "200000000"], capture_output=True, timeout=1500)
    out = r.stdout.decode()
    assert r.returncode == 0, out
    assert "total: 0 differences" in out
    assert out.count(" 0 differences") == 10, out


def test_oracle_back_ends_agree_bit_for_bit(oracle, oracle_libm):
    """The restatement with this header and the restatement that calls the C library render the same frames."""
    import numpy as np
    from wurblpt_amd import host
    for sc in (host.cornell(48, 48, 1, 2), host.spheres(40, 30, 0), host.sponza_like(40, 24, detail=0.05, tex_size=32, env_width=64, importance_n=16),
               host.rgl_scene(32, 24, 1), host.animated(32, 24, 8, 0.0, 1.0)):
        params = host.default_params()
        if "animated" in sc.name:
            params.t0, params.t1 = 0.0, 1.0
        if sc.d.envmap.type != 0 and sc.d.envmap.N > 0:
            ta, tb = oracle.envmap_tables(sc), oracle_libm.envmap_tables(sc)
            assert all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(ta, tb))
            sc.set_envmap_tables(*ta)
        fa, ca = oracle.render(sc, 3, params=params)
        fb, cb = oracle_libm.render(sc, 3, params=params)
        assert np.array_equal(fa.view(np.uint32), fb.view(np.uint32)) and ca == cb, sc.name
