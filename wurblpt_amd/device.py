"""Python binding of the C ABI in include/wurblpt_hip.h (libwurblpt_hip.so).

torch supplies device memory (frame buffers are torch tensors whose data_ptr() crosses the
C ABI) and streams; nothing here computes.  There is no CPU fallback: loading fails loudly
when the HIP library is missing, and rendering fails when no GPU is present."""
import ctypes as C
import os

from . import _abi

_LIB = None


def lib_path():
    # WPT_LIB_DIR: a second build of the pair of libraries (wurblpt_amd/csrc/Makefile, LIB=...), for experiments
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("WPT_LIB_DIR", "lib"), "libwurblpt_hip.so")


WALK_WIDE, WALK_FULL_SHADOW, WALK_COUNT_PRODUCT, WALK_TRIANGLES_AS_GIVEN = 1, 2, 4, 8  # wpt_set_walk (include/wurblpt_hip.h)

EXPORTS = ["wpt_device_count", "wpt_select_device", "wpt_current_device", "wpt_scene_upload", "wpt_scene_free", "wpt_scene_check",
           "wpt_postproc_to_srgb", "wpt_postproc_max_luminance", "wpt_postproc_uniform_rational_quantization",
           "wpt_postproc_scale_luminance", "wpt_postproc_host", "wpt_ground_truth_device", "wpt_ground_truth", "wpt_render_bands_device", "wpt_render_bands",
           "wpt_render_block_device", "wpt_render_block", "wpt_set_launch_config", "wpt_set_top_nodes", "wpt_set_walk", "wpt_set_wavefront", "wpt_kernel_name", "wpt_device_name", "wpt_build_info", "wpt_last_render_passes",
           "wpt_last_error"]


def lib():
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise RuntimeError("%s is missing: the HIP extension must be built (python -c 'import __graft_entry__ as g; g.build()'); "
                               "there is no CPU fallback" % path)
        try:
            # the HIP runtime this process uses must be one: PyTorch ships its own libamdhip64, and when the system's copy
            # gets loaded first (this library links it) the two runtimes do not both see the device
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(path)
        L.wpt_device_count.restype = C.c_int
        L.wpt_device_name.restype = C.c_char_p
        L.wpt_device_name.argtypes = [C.c_int]
        L.wpt_build_info.restype = C.c_char_p
        L.wpt_last_render_passes.restype = C.c_uint32
        L.wpt_select_device.argtypes = [C.c_int]
        L.wpt_scene_upload.argtypes = [C.POINTER(_abi.SceneDesc), C.POINTER(C.c_void_p)]
        L.wpt_scene_free.argtypes = [C.c_void_p]
        L.wpt_scene_check.argtypes = [C.c_void_p]
        L.wpt_render_block_device.argtypes = [C.c_void_p, C.POINTER(_abi.Camera), C.POINTER(_abi.Params),
                                              C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                              C.c_void_p, C.c_void_p, C.c_void_p]
        L.wpt_render_block.argtypes = [C.c_void_p, C.POINTER(_abi.Camera), C.POINTER(_abi.Params),
                                       C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.wpt_set_launch_config.argtypes = [C.c_uint32, C.c_uint32]
        L.wpt_set_wavefront.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.wpt_set_walk.argtypes = [C.c_uint32]
        L.wpt_set_top_nodes.argtypes = [C.c_uint32]
        L.wpt_kernel_name.restype = C.c_char_p
        L.wpt_last_error.restype = C.c_char_p
        L.wpt_selftest_math.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.wpt_scene_get_envmap_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


def _check(status):
    if status != _abi.WPT_OK:
        raise RuntimeError("wurblpt_hip: %s (status %d)" % (lib().wpt_last_error().decode(), status))


def device_count():
    return lib().wpt_device_count()


class DeviceScene:
    """A flattened scene resident in HBM on the current device."""

    def __init__(self, host_scene):
        self._handle = C.c_void_p()
        _check(lib().wpt_scene_upload(host_scene.desc, C.byref(self._handle)))
        self.host = host_scene

    def close(self):
        if self._handle:
            lib().wpt_scene_free(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self):
        """Synchronises and raises if a launch on this scene aborted."""
        _check(lib().wpt_scene_check(self._handle))

    def render_block_into(self, frame, samples_sqrt, block=None, params=None, counters=None, stream=None,
                          width=None, height=None):
        """Asynchronously renders pixels [start, start+size) into `frame`, a CUDA float32 tensor
        [h, w, 3] (full frame).  `counters`: optional CUDA int64 tensor [6] that is added to."""
        from . import host
        w = width or self.host.width
        h = height or self.host.height
        assert frame.is_cuda and frame.is_contiguous() and frame.numel() == w * h * 3
        p = params if params is not None else host.default_params()
        start, size = block if block is not None else (0, w * h)
        sptr = C.c_void_p(stream.cuda_stream) if stream is not None else None
        cptr = C.c_void_p(counters.data_ptr()) if counters is not None else None
        _check(lib().wpt_render_block_device(self._handle, self.host.camera, C.byref(p), w, h, samples_sqrt,
                                              start, size, C.c_void_p(frame.data_ptr()), cptr, sptr))

    def render_bands_into(self, frame, samples_sqrt, band_rows, first_band, band_stride, params=None, counters=None, stream=None,
                          width=None, height=None):
        """Asynchronously renders bands first_band, first_band + band_stride, ... of `band_rows` rows each into `frame`
        (one launch: a rank's interleaved share of the frame)."""
        from . import host
        w = width or self.host.width
        h = height or self.host.height
        assert frame.is_cuda and frame.is_contiguous() and frame.numel() == w * h * 3
        p = params if params is not None else host.default_params()
        sptr = C.c_void_p(stream.cuda_stream) if stream is not None else None
        cptr = C.c_void_p(counters.data_ptr()) if counters is not None else None
        L = lib()
        L.wpt_render_bands_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                              C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        _check(L.wpt_render_bands_device(self._handle, C.cast(self.host.camera, C.c_void_p), C.addressof(p), w, h, samples_sqrt,
                                         band_rows, first_band, band_stride, C.c_void_p(frame.data_ptr()), cptr, sptr))

    def render(self, samples_sqrt, block=None, params=None, with_counters=False, width=None, height=None):
        """Synchronous convenience: returns (frame as numpy [h, w, 3], counters dict or None)."""
        import torch
        w = width or self.host.width
        h = height or self.host.height
        frame = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
        counters = torch.zeros(6, dtype=torch.int64, device="cuda") if with_counters else None
        stream = torch.cuda.current_stream()
        self.render_block_into(frame, samples_sqrt, block, params, counters, stream, w, h)
        torch.cuda.synchronize()
        self.check()
        cnt = None
        if with_counters:
            names = ("samples", "rays", "node_visits", "leaf_tests", "pdf_tests", "scatters")
            cnt = dict(zip(names, [int(x) for x in counters.cpu().tolist()]))
        return frame.cpu().numpy(), cnt

    def render_block_host(self, samples_sqrt, block, params=None, width=None, height=None):
        """wpt_render_block: MPICoordinator::submitBlock semantics, host buffer of size*3 floats."""
        import numpy as np
        from . import host
        w = width or self.host.width
        h = height or self.host.height
        p = params if params is not None else host.default_params()
        start, size = block
        out = np.zeros((size, 3), dtype=np.float32)
        _check(lib().wpt_render_block(self._handle, self.host.camera, C.byref(p), w, h, samples_sqrt, start, size,
                                       C.c_void_p(out.ctypes.data)))
        return out


GT_NAMES = ("world_space_positions", "world_space_geometry_normals", "world_space_geometry_tangents",
            "world_space_material_normals", "world_space_material_tangents", "camera_space_positions",
            "camera_space_geometry_normals", "camera_space_geometry_tangents", "camera_space_material_normals",
            "camera_space_material_tangents", "camera_space_depths", "camera_space_distances", "texcoords",
            "world_space_offset_to_prev", "world_space_offset_to_next", "camera_space_offset_to_prev",
            "camera_space_offset_to_next", "pixel_space_offset_to_prev", "pixel_space_offset_to_next", "materials")
GT_COMPONENTS = (3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 1, 1, 2, 3, 3, 3, 3, 2, 2, 1)
GT_ALL = (1 << 20) - 1


def gt_arrays(width, height, bits=GT_ALL):
    """Host arrays for a ground truth call: (list of numpy arrays or None, ctypes pointer array)."""
    import numpy as np
    arrays = [np.zeros((height, width, GT_COMPONENTS[k]), np.int32 if k == 19 else np.float32) if bits & (1 << k) else None
              for k in range(20)]
    ptrs = (C.c_void_p * 20)(*[a.ctypes.data if a is not None else None for a in arrays])
    return arrays, ptrs


def ground_truth(scene, bits=GT_ALL, camera_prev=None, camera_next=None, params=None, width=None, height=None, times=None):
    """wpt_ground_truth on a DeviceScene: dict name -> numpy array [h, w, comps] of the requested GroundTruth bits.
    times = (t0, tPrev, tNext) for animated instances."""
    from . import host
    w = width or scene.host.width
    h = height or scene.host.height
    p = params if params is not None else host.default_params()
    arrays, ptrs = gt_arrays(w, h, bits)
    L = lib()
    L.wpt_ground_truth.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    tm = (C.c_float * 3)(*times) if times is not None else None
    _check(L.wpt_ground_truth(scene._handle, C.cast(scene.host.camera, C.c_void_p),
                              C.addressof(camera_prev) if camera_prev is not None else None,
                              C.addressof(camera_next) if camera_next is not None else None, tm, C.addressof(p), w, h, ptrs))
    return {GT_NAMES[k]: a for k, a in enumerate(arrays) if a is not None}


def postproc(op, rgb, a=0.0, b=0.0):
    """Output-side operations on a frame (float32 [..., 3]) through wpt_postproc_host: op "srgb" -> uint8 frame,
    "urq" (a = max_val, b = brightness) and "scale" (a = factor, b = clamp) -> float frames, "maxlum" -> float."""
    import numpy as np
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    pixels = rgb.size // 3
    code = {"srgb": 0, "urq": 1, "scale": 2, "maxlum": 3}[op]
    out = np.zeros(rgb.shape, np.uint8) if code == 0 else (np.zeros(1, np.float32) if code == 3 else np.zeros(rgb.shape, np.float32))
    L = lib()
    L.wpt_postproc_host.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_float, C.c_float]
    _check(L.wpt_postproc_host(code, C.c_void_p(rgb.ctypes.data), C.c_void_p(out.ctypes.data), pixels, a, b))
    return float(out[0]) if code == 3 else out


def selftest_aabb(boxes, rays):
    """AABB::mayHit as the kernels evaluate it: boxes (n,6) lo hi, rays (n,8) origin dir amin amax."""
    import numpy as np
    import torch
    tb = torch.as_tensor(np.ascontiguousarray(boxes, dtype=np.float32).reshape(-1, 6), device="cuda")
    tr = torch.as_tensor(np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8), device="cuda")
    assert tb.shape[0] == tr.shape[0]
    out = torch.empty(tb.shape[0], dtype=torch.int32, device="cuda")
    L = lib()
    L.wpt_selftest_aabb.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    _check(L.wpt_selftest_aabb(tb.shape[0], C.c_void_p(tb.data_ptr()), C.c_void_p(tr.data_ptr()), C.c_void_p(out.data_ptr())))
    return out.cpu().numpy()


def selftest_math(op, a, b=None):
    """Evaluates one arithmetic primitive of the kernel on the GPU (see wpt_selftest_kernel)."""
    import torch
    ta = torch.as_tensor(a, dtype=torch.float32, device="cuda").contiguous()
    tb = torch.as_tensor(b if b is not None else a, dtype=torch.float32, device="cuda").contiguous()
    out = torch.empty_like(ta)
    _check(lib().wpt_selftest_math(op, ta.numel(), C.c_void_p(ta.data_ptr()), C.c_void_p(tb.data_ptr()), C.c_void_p(out.data_ptr())))
    return out.cpu().numpy()
