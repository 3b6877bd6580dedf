"""The C ABI: struct layouts of the ctypes mirror equal the header's, and the HIP library
loads and exports every symbol include/wurblpt_hip.h declares (no compute without a GPU)."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

from wurblpt_amd import _abi, device

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "wurblpt_hip.h")


def test_struct_sizes_match_the_header():
    names = sorted(_abi.STRUCT_SIZES)
    body = "\n".join('printf("%s %%zu\\n", sizeof(%s));' % (n, n) for n in names + ["wpt_scene_desc", "wpt_envmap"])
    src = '#include <stdio.h>\n#include "%s"\nint main(void){%s return 0;}\n' % (HEADER, body)
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", c, "-o", exe])  # the header is plain C
        out = dict(line.split() for line in subprocess.check_output([exe]).decode().splitlines())
    for n in names:
        cls, expected = _abi.STRUCT_SIZES[n]
        assert int(out[n]) == expected == C.sizeof(cls), (n, out[n], expected, C.sizeof(cls))
    assert int(out["wpt_scene_desc"]) == C.sizeof(_abi.SceneDesc)
    assert int(out["wpt_envmap"]) == C.sizeof(_abi.Envmap)


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(wpt_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    fns = declared_functions()
    assert "wpt_scene_upload" in fns and "wpt_render_block_device" in fns and "wpt_render_block" in fns
    lib = device.lib()
    for name in fns:
        assert hasattr(lib, name), "libwurblpt_hip.so does not export %s" % name


def test_no_device_is_reported_not_faked():
    """Without a GPU the library must say so; there is no CPU fallback behind the ABI."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert device.device_count() == 0
    from wurblpt_amd import host
    sc = host.cornell(16, 16)
    with pytest.raises(RuntimeError, match="no HIP device|no ROCm"):
        device.DeviceScene(sc)


def _upload_status(sc):
    handle = C.c_void_p()
    st = device.lib().wpt_scene_upload(sc.desc, C.byref(handle))
    assert not handle.value or st == 0
    if st == 0 and handle.value:        # only where a GPU is present
        device.lib().wpt_scene_free(handle)
    return st, device.lib().wpt_last_error().decode()


def _corruptions():
    """(scene maker, name, function that damages one index or size of the description, word expected in the message)"""
    from wurblpt_amd import host
    cornell = lambda: host.cornell(16, 16, 1, 2)
    sponza = lambda: host.sponza_like(16, 16, detail=0.04, tex_size=16, env_width=32, importance_n=0)
    spheres = lambda: host.spheres(16, 16, 0)
    cube = lambda: host.spheres(16, 16, 1)
    animated = lambda: host.animated(16, 16, 8, 0.0, 1.0)
    rgl = lambda: host.rgl_scene(16, 16, 1)

    def first(d, array, count, pred):
        for i in range(count):
            if pred(array[i]):
                return array[i]
        raise AssertionError("fixture has no such element")
    cases = [
        (cornell, "abi version", lambda d: setattr(d, "abi_version", d.abi_version + 1), "ABI"),
        (cornell, "inner node link", lambda d: setattr(d.nodes[0], "link", 10 ** 6), "BVH"),
        (cornell, "shared child (both children of the root are node 1)", lambda d: setattr(d.nodes[0], "link", 1), "depth-first"),
        (cornell, "unreachable node (the root's second child starts one node late)", lambda d: setattr(d.nodes[0], "link", d.nodes[0].link + 1), "depth-first"),
        (cornell, "leaf triangle", lambda d: setattr(first(d, d.nodes, d.node_count, lambda n: n.kind == 1), "link", d.tri_count), "BVH"),
        (cornell, "node kind", lambda d: setattr(d.nodes[1], "kind", 9), "BVH"),
        (cornell, "triangle instance", lambda d: setattr(d.tri_geom[3], "instance", d.instance_count), "instance"),
        (cornell, "triangle material", lambda d: setattr(d.tri_geom[3], "material", d.material_count), "material"),
        (cornell, "hot spot triangle", lambda d: setattr(d.hotspots[0], "prim", d.tri_count), "hot spot"),
        (cornell, "hot spot kind", lambda d: setattr(d.hotspots[0], "kind", 7), "hot spot"),
        (cornell, "material type", lambda d: setattr(d.materials[0], "type", 99), "material"),
        (sponza, "material texture", lambda d: first(d, d.materials, d.material_count, lambda m: m.type != 7 and m.tex[0] >= 0).tex.__setitem__(0, d.texture_count), "texture"),
        (sponza, "normal map", lambda d: setattr(first(d, d.materials, d.material_count, lambda m: m.normal_tex >= 0), "normal_tex", d.texture_count + 5), "normal map"),
        (sponza, "two-sided child", lambda d: first(d, d.materials, d.material_count, lambda m: m.type == 7).tex.__setitem__(1, d.material_count), "two-sided"),
        (sponza, "texel range", lambda d: setattr(first(d, d.textures, d.texture_count, lambda t: t.type == 2), "texel_offset", d.texel_bytes), "texel"),
        (sponza, "image size", lambda d: setattr(first(d, d.textures, d.texture_count, lambda t: t.type == 2), "width", 1 << 20), "texel"),
        (sponza, "texture type", lambda d: setattr(d.textures[0], "type", 42), "texture type"),
        (sponza, "environment texture", lambda d: setattr(d.envmap, "tex", d.texture_count), "environment"),
        (sponza, "environment type", lambda d: setattr(d.envmap, "type", 17), "environment"),
        (spheres, "leaf sphere", lambda d: setattr(first(d, d.nodes, d.node_count, lambda n: n.kind == 2), "link", d.sphere_count), "BVH"),
        (spheres, "sphere material", lambda d: setattr(d.spheres[0], "material", d.material_count), "sphere"),
        (spheres, "hot spot sphere", lambda d: setattr(first(d, d.hotspots, d.hotspot_count, lambda h: h.kind == 1), "prim", d.sphere_count), "hot spot"),
        (cube, "cube face texture", lambda d: d.envmap.cube_tex.__setitem__(4, d.texture_count), "cube"),
        (animated, "instance animation", lambda d: setattr(first(d, d.instances, d.instance_count, lambda i: i.animation >= 0), "animation", d.animation_count), "animation"),
        (animated, "animated triangle of a still instance", lambda d: setattr(first(d, d.instances, d.instance_count, lambda i: i.animation >= 0), "animation", -1), "anim"),
        (animated, "sphere animation", lambda d: setattr(d.spheres[0], "animation", d.animation_count + 3), "animation"),
        (animated, "hot spot animation", lambda d: setattr(d.hotspots[0], "animation", d.animation_count), "animation"),
        (animated, "key frame range", lambda d: setattr(d.animations[1], "keyframe_count", d.keyframe_count + 1), "key frames"),
        (animated, "key frame order", lambda d: setattr(d.keyframes[d.animations[0].first_keyframe + 1], "t", -5.0), "sorted"),
        (rgl, "measured BRDF table", lambda d: setattr(d.rgl_brdfs[0].vndf, "data", d.rgl_data_count), "measured BRDF"),
        (rgl, "measured BRDF cdf", lambda d: setattr(d.rgl_brdfs[0].luminance, "conditional_cdf", d.rgl_data_count - 3), "measured BRDF"),
        (rgl, "measured BRDF shape", lambda d: setattr(d.rgl_brdfs[0].rgb, "dims", 2), "measured BRDF"),
        (rgl, "measured BRDF grid", lambda d: d.rgl_brdfs[0].vndf.param_values.__setitem__(0, d.rgl_data_count), "measured BRDF"),
    ]
    return cases


@pytest.mark.parametrize("case", range(33))
def test_upload_validation_rejects_bad_scene(case):
    """Every index and size the kernels follow is checked on the host before anything reaches the GPU: a description
    with one of them out of range is refused with a message that names the table (no device needed to find out)."""
    maker, name, damage, word = _corruptions()[case]
    sc = maker()
    damage(sc.d)
    st, message = _upload_status(sc)
    assert st in (1, 4), (name, st, message)        # WPT_ERR_INVALID_ARGUMENT or WPT_ERR_UNSUPPORTED
    assert word.lower() in message.lower(), (name, message)


def test_upload_validation_accepts_the_fixtures():
    """The undamaged descriptions pass validation (they fail later only for want of a device, where there is none)."""
    seen = set()
    for maker, _, _, _ in _corruptions():
        if maker.__code__.co_code in seen:
            continue
        seen.add(maker.__code__.co_code)
    from wurblpt_amd import host
    for sc in (host.cornell(16, 16, 1, 2), host.spheres(16, 16, 0), host.animated(16, 16, 8, 0.0, 1.0), host.rgl_scene(16, 16, 1)):
        st, message = _upload_status(sc)
        assert st == 0 or "no HIP device" in message or "no ROCm" in message, message


def test_product_does_not_touch_the_oracle():
    """The product libraries must not link, load or mention anything under oracle/."""
    for lib in ("libwurblpt_hip.so", "libwurblpt_host.so"):
        path = os.path.join(ROOT, "wurblpt_amd", "lib", lib)
        needed = subprocess.check_output(["readelf", "-d", path]).decode()
        assert "oracle" not in needed
    for dirpath, _, files in os.walk(os.path.join(ROOT, "wurblpt_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle_loader" not in text and "liboracle" not in text, f
    for dirpath, _, files in os.walk(os.path.join(ROOT, "include")):
        for f in files:
            assert "liboracle" not in open(os.path.join(dirpath, f)).read()


def test_product_library_reads_no_environment_variable():
    """One product, separate experiments: nothing an inherited environment variable could switch lives in the library that ships
    (measurement settings are arguments of the wpt_set_* hooks)."""
    path = os.path.join(ROOT, "wurblpt_amd", "lib", "libwurblpt_hip.so")
    symbols = subprocess.check_output(["nm", "-D", path]).decode()
    assert "getenv" not in symbols, [l for l in symbols.splitlines() if "getenv" in l]
    for dirpath, _, files in os.walk(os.path.join(ROOT, "wurblpt_amd", "csrc")):
        if os.path.basename(dirpath).startswith("build"):
            continue
        for f in files:
            if f.endswith((".h", ".hip")):
                assert "getenv" not in open(os.path.join(dirpath, f)).read(), f


def test_header_is_plain_c(tmp_path):
    """include/wurblpt_hip.h is the boundary for any language with a C FFI: it compiles as C99 with -pedantic."""
    src = tmp_path / "cabi.c"
    src.write_text('#include "wurblpt_hip.h"\nint main(void) { wpt_params p; wpt_camera c; (void)p; (void)c; return (int)wpt_gt_components[0] - 3; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"), "-c", str(src), "-o",
                    str(tmp_path / "cabi.o")], check=True, timeout=120)
