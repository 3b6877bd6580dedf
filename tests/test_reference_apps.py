"""The drop-in boundary at source level: the reference's own example applications, compiled from where they lie under
/root/reference (nothing of them is copied or shipped), unmodified, against include/ and linked to libwurblpt_hip.so.

Runs in the build container only (the reference does not travel to the GPU box).  What it pins: the public classes,
constructors, take() overloads, mcpt / getGroundTruth signatures, the post-processing functions and the TGD:: spellings
(include/tgd/) those applications use all exist with the reference's names and argument lists.  The applications whose
features are out of scope (participating media, time of flight, noise textures, user-defined
Texture subclasses evaluated on the CPU) are listed with the reason and must keep failing for that reason only."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree exists in the build container only")

# north_star's configurations (cornellbox, sponza, san-miguel, bistro), the reference's two statistical tests
# (furnace-test, mis-test), and every other application that stays inside the hot path's feature set
APPS = ["cornellbox", "sponza", "san-miguel", "bistro", "furnace-test", "mis-test", "envmap", "material-comparison",
        "material-playground", "normalmap", "rtiow", "rungholt"]

# application -> what it needs that this framework deliberately does not have
OUT_OF_SCOPE = {
    "participating-media": "Medium",                 # participating media (out of scope)
    "tof-example": "SensorTofAmcw",                  # time-of-flight sensor and lights
    "tof-hcibox": "SensorTofAmcw",
    "noise-textures": "value",                       # evaluates textures on the CPU (Texture::value), noise textures
    "rolling-marbles": "TextureGradientNoise",       # noise textures, a user-defined Texture subclass
    "stagelights": "TextureGradientNoise",
    "rttnw": "TexturePerlinNoise",
    "animations": "override",                        # a user-defined Texture subclass (its value() runs on the CPU)
    "toomuch": "override",
}


def compile_app(app, tmp_path, link=True):
    src = os.path.join(REF, "wurblpt-" + app, "wurblpt-" + app + ".cpp")
    lib = os.path.join(ROOT, "wurblpt_amd", "lib")
    exe = str(tmp_path / app)
    cmd = ["g++", "-std=c++20", "-O0", "-fopenmp", "-I" + os.path.join(ROOT, "include"), src]
    cmd += ["-L" + lib, "-lwurblpt_hip", "-Wl,-rpath," + lib, "-o", exe] if link else ["-fsyntax-only"]
    return subprocess.run(cmd, capture_output=True, timeout=600), exe


@pytest.mark.parametrize("app", APPS)
def test_reference_application_compiles_and_links_unchanged(app, tmp_path):
    r, exe = compile_app(app, tmp_path)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    assert os.path.exists(exe)
    # the only library of this framework it needs at run time is the HIP one (no oracle, no CPU path)
    needed = subprocess.check_output(["readelf", "-d", exe]).decode()
    assert "libwurblpt_hip.so" in needed and "oracle" not in needed


def test_reference_cornellbox_fails_loudly_without_a_device(tmp_path):
    """wurblpt-cornellbox as built above: scene construction and the BVH build run (host side), mcpt() then stops with
    the device error -- nothing is rendered on the CPU behind the application's back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r, exe = compile_app("cornellbox", tmp_path)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    run = subprocess.run([exe], capture_output=True, timeout=300, cwd=str(tmp_path))
    assert run.returncode != 0
    assert b"Linearized bounding volume hierarchy with 71 nodes" in run.stderr and b"no HIP device" in run.stderr
    assert not os.path.exists(str(tmp_path / "image.exr"))


@pytest.mark.parametrize("app", sorted(OUT_OF_SCOPE))
def test_out_of_scope_applications_fail_for_the_stated_reason(app, tmp_path):
    r, _ = compile_app(app, tmp_path, link=False)
    assert r.returncode != 0
    assert OUT_OF_SCOPE[app] in r.stderr.decode()
