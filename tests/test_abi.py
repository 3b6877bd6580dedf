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


def test_upload_validation_rejects_bad_scene():
    """Index validation happens on the host before anything reaches the GPU."""
    from wurblpt_amd import host
    sc = host.cornell(16, 16)
    d = sc.d
    saved = d.nodes[0].link
    d.nodes[0].link = 10 ** 6
    handle = C.c_void_p()
    st = device.lib().wpt_scene_upload(sc.desc, C.byref(handle))
    d.nodes[0].link = saved
    assert st == 1  # WPT_ERR_INVALID_ARGUMENT
    assert b"BVH" in device.lib().wpt_last_error()


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
