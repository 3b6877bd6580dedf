"""examples/room.cpp: an application written against include/wurblpt alone (scene, mcpt, postproc, getGroundTruth,
saveImage), compiled with g++ and linked to libwurblpt_hip.so -- the way a user of the reference switches over."""
import os
import subprocess

import numpy as np
import pytest

from wurblpt_amd import host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(tmp_path):
    exe = str(tmp_path / "room")
    lib = os.path.join(ROOT, "wurblpt_amd", "lib")
    cmd = ["g++", "-std=c++20", "-O1", "-fopenmp", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "room.cpp"), "-L" + lib, "-lwurblpt_hip", "-Wl,-rpath," + lib, "-o", exe]
    subprocess.run(cmd, check=True, timeout=600)
    return exe


def test_example_application_builds_against_the_public_headers(tmp_path):
    """One include (<wurblpt/wurblpt.hpp>) and one library are enough; without a GPU the program says so and stops."""
    import torch
    exe = build(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the run is covered by the gpu test")
    r = subprocess.run([exe, "16", "12", "1", str(tmp_path)], capture_output=True, timeout=120)
    assert r.returncode != 0 and b"no HIP device" in r.stderr
    assert not os.path.exists(str(tmp_path / "room.png"))       # nothing is faked on the CPU


@pytest.mark.gpu
def test_example_application_renders(tmp_path):
    exe = build(tmp_path)
    out = []
    for run in range(2):
        d = tmp_path / ("run%d" % run)
        d.mkdir()
        r = subprocess.run([exe, "96", "72", "4", str(d)], capture_output=True, timeout=300)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        out.append(d)
    frame = host.image_load(str(out[0] / "room.pfm"))
    assert frame.shape == (72, 96, 3) and np.isfinite(frame).all() and frame.mean() > 0.01
    assert np.array_equal(frame, host.image_load(str(out[1] / "room.pfm")))          # deterministic
    png = host.image_load(str(out[0] / "room.png"))
    assert png.dtype == np.uint8 and png.shape == (72, 96, 3) and 20 < png.mean() < 235
    depth = host.image_load(str(out[0] / "room-depth.pfm"))[:, :, 0]
    # the camera stands 3.4 in front of the room's centre and looks at the back wall (z = -1): depth 4.4 there
    assert abs(depth[50, 40] - 4.4) < 1e-3 and b'material "' in r.stderr
    # the run's record: device model, count, seconds and the kernels' compiler are in the frame's tags
    log = r.stderr.decode()
    assert "WURBLPT/DEVICE_MODEL = " in log and "gfx950" in log and "WURBLPT/DEVICE_COUNT = 1" in log
    assert "WURBLPT/COMPILER = hipcc / clang" in log and "WURBLPT/SAMPLES_PER_PIXEL = 16" in log
    seconds = float(log.split("WURBLPT/DEVICE_SECONDS = ")[1].split()[0])
    assert 0.0 < seconds < 120.0
    assert depth[depth > 0].min() > 1.5 and depth.max() < 4.41 and (depth > 0).mean() > 0.5      # 0 where the view passes the room
    blur = host.image_load(str(out[0] / "room-blur.png"))
    assert blur.shape == png.shape and not np.array_equal(blur, png)
    # the panel swings: its edges are sharp in the still and smeared over the exposure
    diff = np.abs(blur.astype(int) - png.astype(int)).mean(axis=2)
    assert diff.max() > 40
