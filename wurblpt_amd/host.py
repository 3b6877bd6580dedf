"""Host side of the framework for Python callers (tests, bench): scene construction and
flattening through libwurblpt_host.so, which is compiled from the C++ API under
include/wurblpt/ (the mirror of the reference's Scene / Mesh / Material / Camera classes).
CPU only; no HIP dependency."""
import ctypes as C
import os

import numpy as np

from . import _abi

_LIB = None


def lib_path():
    # WPT_LIB_DIR: a second build of the pair of libraries (wurblpt_amd/csrc/Makefile, LIB=...), for experiments
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("WPT_LIB_DIR", "lib"), "libwurblpt_host.so")


def lib():
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise RuntimeError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`" % path)
        try:
            # the HIP runtime this process uses must be one: PyTorch ships its own libamdhip64, and when the system's copy
            # gets loaded first (this library links it) the two runtimes do not both see the device
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(path)
        L.wpt_host_cornell.restype = C.c_void_p
        L.wpt_host_cornell.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint, C.c_uint]
        L.wpt_host_random_triangles.restype = C.c_void_p
        L.wpt_host_random_triangles.argtypes = [C.c_uint, C.c_uint, C.c_int, C.c_uint, C.c_uint, C.c_float]
        L.wpt_host_sponza_like.restype = C.c_void_p
        L.wpt_host_sponza_like.argtypes = [C.c_uint, C.c_float, C.c_uint, C.c_uint, C.c_int, C.c_uint, C.c_uint]
        L.wpt_host_furnace.restype = C.c_void_p
        L.wpt_host_furnace.argtypes = [C.c_int, C.c_int, C.c_uint, C.c_uint]
        L.wpt_host_spheres.restype = C.c_void_p
        L.wpt_host_spheres.argtypes = [C.c_int, C.c_uint, C.c_uint]
        L.wpt_host_rgl_scene.restype = C.c_void_p
        L.wpt_host_rgl_scene.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_uint, C.c_uint]
        L.wpt_host_rgl_build.restype = C.c_ulonglong
        L.wpt_host_rgl_build.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_ulonglong]
        L.wpt_host_measured_like.restype = C.c_void_p
        L.wpt_host_measured_like.argtypes = [C.c_uint, C.c_float, C.c_uint, C.c_uint, C.c_int, C.c_char_p, C.c_char_p, C.c_uint, C.c_uint]
        L.wpt_host_courtyard_like.restype = C.c_void_p
        L.wpt_host_courtyard_like.argtypes = [C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint]
        L.wpt_host_scene_desc.restype = C.POINTER(_abi.SceneDesc)
        L.wpt_host_scene_desc.argtypes = [C.c_void_p]
        L.wpt_host_scene_camera.restype = C.POINTER(_abi.Camera)
        L.wpt_host_scene_camera.argtypes = [C.c_void_p]
        L.wpt_host_scene_bvh_levels.restype = C.c_uint
        L.wpt_host_scene_bvh_levels.argtypes = [C.c_void_p]
        L.wpt_host_scene_free.argtypes = [C.c_void_p]
        L.wpt_host_default_params.argtypes = [C.POINTER(_abi.Params)]
        L.wpt_host_bvh_build.restype = C.c_uint
        L.wpt_host_bvh_build.argtypes = [C.c_uint, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint)]
        _LIB = L
    return _LIB


def default_params():
    """Parameters() of the reference (wurblpt.hpp:89-95) plus the default SensorRGB gates."""
    p = _abi.Params()
    lib().wpt_host_default_params(C.byref(p))
    return p


class HostScene:
    """A built and flattened scene with its camera; owns the host buffers."""

    def __init__(self, handle, width, height, name):
        if not handle:
            raise RuntimeError("scene construction failed: %s" % name)
        self._handle = handle
        self.width, self.height, self.name = width, height, name
        self.desc = lib().wpt_host_scene_desc(handle)
        self.camera = lib().wpt_host_scene_camera(handle)
        self._handle_for_camera = handle
        self.bvh_levels = lib().wpt_host_scene_bvh_levels(handle)

    def __del__(self):
        if getattr(self, "_handle", None):
            try:
                lib().wpt_host_scene_free(self._handle)
            except TypeError:   # interpreter shutdown: the module globals are gone already
                pass
            self._handle = None

    @property
    def d(self):
        return self.desc.contents

    def set_envmap_tables(self, M, Ms, Mcs):
        """Attaches importance tables (numpy arrays, kept alive here) to the scene description;
        without them wpt_scene_upload builds them on the device."""
        self._env_tables = (M, Ms, Mcs)
        e = self.d.envmap
        e.M, e.Ms, e.Mcs = M.ctypes.data, Ms.ctypes.data, Mcs.ctypes.data

    def nodes_array(self):
        n = self.d.node_count
        return np.ctypeslib.as_array(C.cast(self.d.nodes, C.POINTER(C.c_uint32)), shape=(n, 8)).copy()


def set_distortion(scene, model, k1=0.0, k2=0.0, k3=0.0, p1=0.0, p2=0.0):
    """Lens distortion for the scene's camera: model 1 RadialAndPlanar(k1,k2,p1,p2), 2 RadialOnly(k1,k2,k3),
    3 OpenCV(k1,k2,k3,p1,p2), 0 none."""
    L = lib()
    L.wpt_host_scene_set_distortion.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float]
    L.wpt_host_scene_set_distortion.restype = None
    L.wpt_host_scene_set_distortion(scene._handle_for_camera, model, k1, k2, k3, p1, p2)


def set_camera_mode(scene, surround_mode=0, stereoscopic_distance=0.0):
    """Camera::surroundMode (0 off, 1 = 180 degrees, 2 = 360 degrees) and ::stereoscopicDistance."""
    L = lib()
    L.wpt_host_scene_set_camera_mode.argtypes = [C.c_void_p, C.c_int, C.c_float]
    L.wpt_host_scene_set_camera_mode.restype = None
    L.wpt_host_scene_set_camera_mode(scene._handle_for_camera, surround_mode, stereoscopic_distance)


def cornell(width, height, tall_box_material=0, short_object_material=0):
    """Cornell box of wurblpt-cornellbox.cpp: tall box 0 = white / 1 = GGX metal,
    short box 0 = white / 2 = glass."""
    h = lib().wpt_host_cornell(tall_box_material, 0, short_object_material, width, height)
    return HostScene(h, width, height, "cornell(tall=%d,short=%d)" % (tall_box_material, short_object_material))


def random_triangles(n, seed, width, height, with_texcoords=True, aperture=0.0):
    h = lib().wpt_host_random_triangles(n, seed, 1 if with_texcoords else 0, width, height, aperture)
    return HostScene(h, width, height, "random_triangles(%d,%d)" % (n, seed))


def sponza_like(width, height, seed=1, detail=1.0, tex_size=1024, env_width=2048, importance_n=512):
    """BASELINE config 3 stand-in: seeded Sponza-class courtyard (about 262 k triangles at
    detail 1.0), textured Lambertian / ModPhong / two-sided / GGX / mirror materials, normal
    maps, procedural sun + sky environment map with importance sampling."""
    h = lib().wpt_host_sponza_like(seed, detail, tex_size, env_width, importance_n, width, height)
    return HostScene(h, width, height, "sponza_like(seed=%d,detail=%g)" % (seed, detail))


def animated(width, height, variant=0, t0=0.0, t1=1.0):
    """A room with a turning cube, a swinging panel, a sliding light and a moving camera, bounded for the exposure
    interval [t0, t1] (pass the same t0, t1 in the render parameters).  variant bit 0: thin lens; bit 1: static
    camera; bit 2: static instances."""
    L = lib()
    L.wpt_host_animated.restype = C.c_void_p
    L.wpt_host_animated.argtypes = [C.c_int, C.c_float, C.c_float, C.c_uint, C.c_uint]
    return HostScene(L.wpt_host_animated(variant, t0, t1, width, height), width, height, "animated(variant=%d)" % variant)


def courtyard_like(width, height, seed=2, triangles=10_000_000, tex_size=1024):
    """BASELINE config 4 stand-in: seeded San-Miguel-class courtyard dominated by foliage (clouds
    of small two-sided leaf quads), every material two-sided, constant environment without
    importance sampling (wurblpt-san-miguel.cpp:36-44).  `triangles` is the approximate total."""
    h = lib().wpt_host_courtyard_like(seed, triangles, tex_size, width, height)
    return HostScene(h, width, height, "courtyard_like(seed=%d,triangles=%d)" % (seed, triangles))


def furnace(width, height, material=0, slices=64):
    """wurblpt-furnace-test.cpp with a tessellated sphere: one material in a constant environment
    of radiance 1.  material: 0 Lambertian 0.42, 1 Lambertian 1, 2 ModPhong(1,0), 3 ModPhong(0,1),
    4 ModPhong(.5,.5), 5 GGX albedo 1 roughness 0.5; on an analytic sphere: 6 clear glass (no absorption, index 1.5),
    7 perfect mirror."""
    h = lib().wpt_host_furnace(material, slices, width, height)
    return HostScene(h, width, height, "furnace(material=%d)" % material)


def mis_test(width, height, with_hot_spots=True, light_mask=15):
    """The scene of wurblpt-mis-test.cpp: four GGX plates under four sphere lights in a white room; the lights are hot
    spots (next-event estimation with MIS) or not (material sampling alone).  light_mask: bit i = light i exists."""
    L = lib()
    L.wpt_host_mis_test.restype = C.c_void_p
    L.wpt_host_mis_test.argtypes = [C.c_int, C.c_uint, C.c_uint, C.c_uint]
    h = L.wpt_host_mis_test(1 if with_hot_spots else 0, light_mask, width, height)
    return HostScene(h, width, height, "mis_test(hot_spots=%d,lights=%d)" % (with_hot_spots, light_mask))


def texture_probe(width, height, compat=0):
    """A textured quad light in front of a float environment map, for checking what a camera ray sees directly
    (wpt_host_texture_probe); compat 0 = Mitsuba, 1 = surround video orientation of the environment."""
    L = lib()
    L.wpt_host_texture_probe.restype = C.c_void_p
    L.wpt_host_texture_probe.argtypes = [C.c_int, C.c_uint, C.c_uint]
    return HostScene(L.wpt_host_texture_probe(compat, width, height), width, height, "texture_probe(compat=%d)" % compat)


def spheres(width, height, variant=0):
    """Scenes with analytic spheres (HitableSphere): 0 = textured / GGX / glass / mirror spheres lit by
    a sphere light and a quad light (both hot spots), 1 = the same under a cube environment map,
    2 = wurblpt-furnace-test.cpp as written (every sphere pixel is exactly 0.42), 3 = inside a large
    emitting sphere that is a hot spot."""
    h = lib().wpt_host_spheres(variant, width, height)
    return HostScene(h, width, height, "spheres(variant=%d)" % variant)


def rgl_fixture(name):
    """path of a synthetic measured-BRDF file committed under tests/golden (iso / aniso)"""
    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "synthetic_%s.bsdf" % name)


def rgl_build(filename):
    """The model's tables for a BRDF file, as MaterialRGL builds them: (_abi.RglBrdf, float32 pool)."""
    n = lib().wpt_host_rgl_build(filename.encode(), None, None, 0)
    if n == 0:
        raise RuntimeError("cannot read %s as a measured BRDF" % filename)
    brdf = _abi.RglBrdf()
    pool = np.zeros(n, np.float32)
    lib().wpt_host_rgl_build(filename.encode(), C.byref(brdf), C.c_void_p(pool.ctypes.data), n)
    return brdf, pool


def rgl_scene(width, height, variant=0, file0=None, file1=None):
    """Scenes with MaterialRGL: 0 = furnace test, 1 = two measured spheres under a quad light."""
    f0 = file0 or rgl_fixture("iso")
    f1 = file1 or rgl_fixture("aniso")
    h = lib().wpt_host_rgl_scene(variant, f0.encode(), f1.encode(), width, height)
    return HostScene(h, width, height, "rgl_scene(variant=%d)" % variant)


def measured_like(width, height, rgl0, rgl1, seed=3, detail=1.0, tex_size=1024, env_width=2048, importance_n=512):
    """BASELINE config 5 stand-in: the Sponza-class architecture with measured BRDFs (MaterialRGL, two
    tensor files) and normal maps on walls, columns, beams and vases; environment importance sampling."""
    h = lib().wpt_host_measured_like(seed, detail, tex_size, env_width, importance_n, rgl0.encode(), rgl1.encode(), width, height)
    return HostScene(h, width, height, "measured_like(seed=%d,detail=%g)" % (seed, detail))


IMPORT_DISABLE_LIGHT_SOURCES, IMPORT_DISABLE_HOT_SPOTS, IMPORT_TWO_SIDED_MATERIALS, IMPORT_INVERTED_TF, IMPORT_WITH_GLASS = 1, 2, 4, 8, 16


def import_obj(filename, width, height, eye, at, vfov_degrees=45.0, import_bits=0, scale=1.0, rotate_y_degrees=0.0, env_radiance=0.0):
    """importIntoScene (include/wurblpt/import.hpp) of an OBJ file, an optional constant environment and a
    look-at camera; None if the file cannot be imported."""
    L = lib()
    L.wpt_host_import_obj.restype = C.c_void_p
    L.wpt_host_import_obj.argtypes = [C.c_char_p, C.c_uint, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_float, C.c_uint, C.c_uint]
    e = (C.c_float * 3)(*eye)
    a = (C.c_float * 3)(*at)
    h = L.wpt_host_import_obj(filename.encode(), import_bits, scale, rotate_y_degrees, env_radiance, e, a, vfov_degrees, width, height)
    return HostScene(h, width, height, "import_obj(%s)" % os.path.basename(filename)) if h else None


def import_obj_env(filename, envmap, width, height, eye, at, vfov_degrees=45.0, import_bits=0, scale=1.0, rotate_y_degrees=0.0, importance_n=0):
    """importIntoScene of an OBJ file under an environment map read from an image file, as wurblpt-sponza.cpp:46-59 sets
    its scene up; importance_n > 0: initializeImportanceSampling(importance_n).  None if a file cannot be read."""
    L = lib()
    L.wpt_host_import_obj_env.restype = C.c_void_p
    L.wpt_host_import_obj_env.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_uint, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_float, C.c_uint, C.c_uint]
    e = (C.c_float * 3)(*eye)
    a = (C.c_float * 3)(*at)
    h = L.wpt_host_import_obj_env(filename.encode(), envmap.encode(), importance_n, import_bits, scale, rotate_y_degrees, e, a, vfov_degrees, width, height)
    return HostScene(h, width, height, "import_obj_env(%s, %s)" % (os.path.basename(filename), os.path.basename(envmap))) if h else None


def image_load(filename):
    """Decodes an image file with the importer's decoders: numpy array [h, w, comps], row 0 = bottom."""
    L = lib()
    L.wpt_host_image_load.restype = C.c_ulonglong
    L.wpt_host_image_load.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_ulonglong]
    info = (C.c_uint * 4)()
    n = L.wpt_host_image_load(filename.encode(), info, None, 0)
    if n == 0:
        return None
    dtype = [np.uint8, np.uint16, np.float32][info[3]]
    out = np.zeros((info[1], info[0], info[2]), dtype=dtype)
    L.wpt_host_image_load(filename.encode(), info, C.c_void_p(out.ctypes.data), n)
    return out


def mcpt(scene, samples_sqrt, t0=0.0, t1=0.0, width=None, height=None, workers=1):
    """mcpt() of include/wurblpt/wurblpt.hpp (needs a device): the scene, its camera and a SensorRGB through the C++
    host API an application uses; returns the frame [h, w, 3]."""
    w = width or scene.width
    h = height or scene.height
    frame = np.zeros((h, w, 3), np.float32)
    L = lib()
    L.wpt_host_mcpt.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.c_float, C.c_float, C.c_void_p, C.c_uint]
    if not L.wpt_host_mcpt(scene._handle, w, h, samples_sqrt, t0, t1, C.c_void_p(frame.ctypes.data), workers):
        raise RuntimeError("mcpt failed")
    return frame


MESH_KINDS = {"quad": 0, "cube": 1, "cube_side": 2, "disk": 3, "sphere": 4, "cylinder": 5, "closed_cylinder": 6, "cone": 7,
              "closed_cone": 8, "torus": 9, "tetrahedron": 10, "octahedron": 11, "icosahedron": 12}


def generate_mesh(kind, a=1, b=1, f=0.0):
    """A mesh of include/wurblpt/generator.hpp: (vertices float32 [n, 11] = position, normal, texcoord, tangent; indices
    uint32 [triangles, 3])."""
    L = lib()
    L.wpt_host_generate_mesh.restype = C.c_uint
    L.wpt_host_generate_mesh.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_uint, C.POINTER(C.c_uint)]
    cap = 1 << 16
    v = np.zeros((cap, 11), np.float32)
    ind = np.zeros(3 * cap, np.uint32)
    ni = C.c_uint(0)
    n = L.wpt_host_generate_mesh(MESH_KINDS[kind], a, b, f, C.c_void_p(v.ctypes.data), C.c_void_p(ind.ctypes.data), cap, C.byref(ni))
    assert n <= cap and ni.value <= 3 * cap
    return v[:n].copy(), ind[:ni.value].reshape(-1, 3).copy()


def material_scene_index(scene):
    """Scene::materialIndex() of every flattened material (what getGroundTruth reports for it)."""
    out = np.zeros(scene.d.material_count, np.int32)
    L = lib()
    L.wpt_host_scene_material_scene_index.argtypes = [C.c_void_p, C.c_void_p]
    L.wpt_host_scene_material_scene_index.restype = None
    L.wpt_host_scene_material_scene_index(scene._handle, C.c_void_p(out.ctypes.data))
    return out


def get_ground_truth(scene, bits=(1 << 20) - 1, prev_from_at=None, next_from_at=None, width=None, height=None, times=None):
    """getGroundTruth() of include/wurblpt/wurblpt.hpp (needs a device) for the scene's look-at camera without
    lens; prev_from_at / next_from_at: 6 floats, eye and target of the camera at tPrev / tNext."""
    from . import device
    w = width or scene.width
    h = height or scene.height
    arrays, ptrs = device.gt_arrays(w, h, bits)
    L = lib()
    L.wpt_host_get_ground_truth.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    tm = (C.c_float * 3)(*times) if times is not None else None
    pf = np.ascontiguousarray(prev_from_at, np.float32) if prev_from_at is not None else None
    nf = np.ascontiguousarray(next_from_at, np.float32) if next_from_at is not None else None
    if not L.wpt_host_get_ground_truth(scene._handle, w, h, C.c_void_p(pf.ctypes.data) if pf is not None else None,
                                       C.c_void_p(nf.ctypes.data) if nf is not None else None, tm, ptrs):
        raise RuntimeError("getGroundTruth failed")
    return {device.GT_NAMES[k]: a for k, a in enumerate(arrays) if a is not None}


def image_save(filename, img):
    """Writes a numpy image [h, w, comps] (uint8, uint16 or float32; row 0 = bottom) by file name extension."""
    img = np.ascontiguousarray(img)
    if img.ndim == 2:
        img = img[:, :, None]
    code = {np.dtype(np.uint8): 0, np.dtype(np.uint16): 1, np.dtype(np.float32): 2}[img.dtype]
    L = lib()
    L.wpt_host_image_save.argtypes = [C.c_char_p, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_void_p]
    return bool(L.wpt_host_image_save(filename.encode(), img.shape[1], img.shape[0], img.shape[2], code,
                                      C.c_void_p(img.ctypes.data)))


def postproc(op, img, a=0.0, b=0.0):
    """The output side through include/wurblpt/postproc.hpp (needs a device): "srgb" -> toSRGB, "urq" ->
    uniformRationalQuantization(a=maxVal, b=brightness), "scale" -> scaleLuminance(a=factor, b=clamp),
    "maxlum" -> maxLuminance.  img: float32 [h, w, comps >= 3]."""
    img = np.ascontiguousarray(img, dtype=np.float32)
    code = {"srgb": 0, "urq": 1, "scale": 2, "maxlum": 3}[op]
    out = (np.zeros(img.shape[:2] + (3,), np.uint8) if code == 0 else
           np.zeros(1, np.float32) if code == 3 else np.zeros(img.shape, np.float32))
    L = lib()
    L.wpt_host_postproc.argtypes = [C.c_int, C.c_uint, C.c_uint, C.c_uint, C.c_void_p, C.c_void_p, C.c_float, C.c_float]
    if not L.wpt_host_postproc(code, img.shape[1], img.shape[0], img.shape[2], C.c_void_p(img.ctypes.data),
                               C.c_void_p(out.ctypes.data), a, b):
        raise RuntimeError("post-processing failed")
    return float(out[0]) if code == 3 else out


def bvh_build(boxes):
    """boxes: float32 [n, 6] (lo, hi) -> (uint32 [nodes, 8] raw node words, levels)."""
    boxes = np.ascontiguousarray(boxes, dtype=np.float32)
    n = boxes.shape[0]
    out = np.zeros((max(2 * n - 1, 1), 8), dtype=np.uint32)
    levels = C.c_uint(0)
    cnt = lib().wpt_host_bvh_build(n, boxes.ctypes.data, out.ctypes.data, C.byref(levels))
    return out[:cnt], levels.value
