"""Loads the CPU restatement under oracle/ -- TEST INFRASTRUCTURE ONLY.

Allowed importers: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (wurblpt_amd/) never imports this module."""
import ctypes as C
import os
import subprocess

import numpy as np

from wurblpt_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_CACHE = {}


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle.so", "liboracle_libm.so"])


class Oracle:
    def __init__(self, backend):
        name = "liboracle.so" if backend == "portable" else "liboracle_libm.so"
        path = os.path.join(ROOT, "oracle", name)
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.wpt_oracle_backend.restype = C.c_char_p
        assert L.wpt_oracle_backend().decode() == backend
        L.wpt_oracle_render.restype = C.c_int
        L.wpt_oracle_render.argtypes = [C.POINTER(_abi.SceneDesc), C.POINTER(_abi.Camera), C.POINTER(_abi.Params),
                                        C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                        C.c_void_p, C.POINTER(_abi.Counters), C.c_int]
        self.L = L
        self.backend = backend

    # -- the integrator -------------------------------------------------
    def render(self, scene, samples_sqrt, params=None, block=None, threads=0, width=None, height=None):
        """Returns (frame float32 [h, w, 3] with zeros outside the block, counters dict)."""
        from wurblpt_amd import host
        w = width or scene.width
        h = height or scene.height
        p = params if params is not None else host.default_params()
        start, size = block if block is not None else (0, w * h)
        frame = np.zeros((h, w, 3), dtype=np.float32)
        cnt = _abi.Counters()
        rc = self.L.wpt_oracle_render(scene.desc, scene.camera, C.byref(p), w, h, samples_sqrt, start, size,
                                      frame.ctypes.data, C.byref(cnt), threads)
        if rc != 0:
            raise RuntimeError("oracle render failed: %d" % rc)
        return frame, cnt.as_dict()

    def ground_truth(self, scene, bits=(1 << 20) - 1, camera_prev=None, camera_next=None, params=None, width=None, height=None, times=None):
        """getGroundTruth restated: dict name -> numpy array, as wurblpt_amd.device.ground_truth returns it."""
        from wurblpt_amd import device, host
        w = width or scene.width
        h = height or scene.height
        p = params if params is not None else host.default_params()
        arrays, ptrs = device.gt_arrays(w, h, bits)
        self.L.wpt_oracle_ground_truth.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        tm = (C.c_float * 3)(*times) if times is not None else None
        rc = self.L.wpt_oracle_ground_truth(C.cast(scene.desc, C.c_void_p), C.cast(scene.camera, C.c_void_p),
                                            C.addressof(camera_prev) if camera_prev is not None else None,
                                            C.addressof(camera_next) if camera_next is not None else None, tm, C.addressof(p), w, h, ptrs)
        if rc != 0:
            raise RuntimeError("oracle ground truth failed: %d" % rc)
        return {device.GT_NAMES[k]: a for k, a in enumerate(arrays) if a is not None}

    def envmap_tables(self, scene, n=None):
        """Importance tables of the scene's environment map (M, Ms, Mcs) as numpy arrays."""
        n = n or scene.d.envmap.N
        M = np.zeros(n * n, np.float32)
        Ms = np.zeros(n * n, np.int32)
        Mcs = np.zeros(n * n, np.float32)
        rc = self.L.wpt_oracle_envmap_tables(scene.desc, C.c_int(n), C.c_void_p(M.ctypes.data), C.c_void_p(Ms.ctypes.data), C.c_void_p(Mcs.ctypes.data))
        assert rc == 0
        return M, Ms, Mcs

    # -- per-function probes ---------------------------------------------
    def _call(self, fn, *args):
        getattr(self.L, fn)(*args)

    def prng(self, pixel, n):
        out = np.zeros(n, np.float32)
        self.L.wpt_oracle_prng(C.c_uint32(pixel), C.c_int(n), C.c_void_p(out.ctypes.data))
        return out

    def prng_x2(self, pixel, n):
        out = np.zeros(2 * n, np.float32)
        self.L.wpt_oracle_prng_x2(C.c_uint32(pixel), C.c_int(n), C.c_void_p(out.ctypes.data))
        return out

    def sampler(self, which, u):
        u = np.ascontiguousarray(u, np.float32)
        n = u.size // 2
        out = np.zeros(n * (2 if which == 0 else 3), np.float32)
        self.L.wpt_oracle_sampler(C.c_int(which), C.c_int(n), C.c_void_p(u.ctypes.data), C.c_void_p(out.ctypes.data))
        return out

    def simple(self, fn, n, out_per, *inputs, out_dtype=np.float32):
        ins = [np.ascontiguousarray(i, np.float32) for i in inputs]
        out = np.zeros(n * out_per, out_dtype)
        getattr(self.L, fn)(C.c_int(n), *[C.c_void_p(i.ctypes.data) for i in ins], C.c_void_p(out.ctypes.data))
        return out

    def camera_rays(self, cam, pq):
        pq = np.ascontiguousarray(pq, np.float32)
        n = pq.size // 2
        out = np.zeros(6 * n, np.float32)
        self.L.wpt_oracle_camera_rays(C.byref(cam), C.c_int(n), C.c_void_p(pq.ctypes.data), C.c_void_p(out.ctypes.data))
        return out

    def bvh_walk(self, nodes_u32, ray8, leaf_a, max_log=1 << 20):
        nodes = np.ascontiguousarray(nodes_u32, np.uint32)
        ray8 = np.ascontiguousarray(ray8, np.float32)
        leaf_a = np.ascontiguousarray(leaf_a, np.float32)
        log = np.zeros(max_log, np.int64)
        n = C.c_int64(0)
        fin = C.c_int64(0)
        fa = C.c_float(0)
        self.L.wpt_oracle_bvh_walk(C.c_void_p(nodes.ctypes.data), C.c_void_p(ray8.ctypes.data),
                                   C.c_void_p(leaf_a.ctypes.data), C.c_void_p(log.ctypes.data), C.c_int64(max_log),
                                   C.byref(n), C.byref(fin), C.byref(fa))
        return log[:n.value].copy(), fin.value, np.float32(fa.value)

    def bvh_hits(self, scene, rays8):
        rays8 = np.ascontiguousarray(rays8, np.float32)
        n = rays8.size // 8
        out = np.zeros((n, 15), np.float32)
        cnt = _abi.Counters()
        self.L.wpt_oracle_bvh_hits(scene.desc, C.c_int(n), C.c_void_p(rays8.ctypes.data), C.c_void_p(out.ctypes.data), C.byref(cnt))
        return out, cnt.as_dict()

    def wide_walk_check(self, scene, rays8):
        """The collapsed four-wide walk of DESIGN.md section 7.1 next to BVH::hit: (rays that differ, statistics)."""
        rays8 = np.ascontiguousarray(rays8, np.float32)
        n = rays8.size // 8
        stats = (C.c_uint64 * 10)()
        self.L.wpt_oracle_wide_walk_check.restype = C.c_int
        differ = self.L.wpt_oracle_wide_walk_check(scene.desc, C.c_int(n), C.c_void_p(rays8.ctypes.data), stats)
        names = ("rays", "binary_visits", "wide_steps", "wide_box_tests", "leaf_tests", "nan_fallbacks", "revalidations_failed", "max_pending", "admission_disagrees", "parent_disagrees")
        return int(differ), dict(zip(names, [int(x) for x in stats]))

    def math(self, op, a, b=None):
        a = np.ascontiguousarray(a, np.float32)
        b = np.ascontiguousarray(b if b is not None else a, np.float32)
        out = np.zeros(a.size, np.float32)
        self.L.wpt_oracle_math(C.c_int(op), C.c_int(a.size), C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data), C.c_void_p(out.ctypes.data))
        return out


def load(backend="portable"):
    if backend not in _CACHE:
        _CACHE[backend] = Oracle(backend)
    return _CACHE[backend]
