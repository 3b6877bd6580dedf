"""ctypes mirror of include/wurblpt_hip.h (the C ABI of the device library).

Field order and types must match the header exactly; tests/test_abi.py checks the sizes
against values compiled from the header.
"""
import ctypes as C

WPT_ABI_VERSION = 5
WPT_OK = 0

NODE_INNER, NODE_TRIANGLE, NODE_SPHERE, NODE_EMPTY = 0, 1, 2, 3
MAT_NONE, MAT_LAMBERTIAN, MAT_LIGHT_DIFFUSE, MAT_MIRROR, MAT_GGX, MAT_GLASS, MAT_MODPHONG, MAT_TWOSIDED, MAT_RGL = range(9)


class BvhNode(C.Structure):
    _fields_ = [("lo", C.c_float * 3), ("hi", C.c_float * 3), ("link", C.c_uint32), ("kind", C.c_uint32)]


class TriGeom(C.Structure):
    _fields_ = [("v0", C.c_float * 3), ("instance", C.c_uint32), ("v1", C.c_float * 3), ("material", C.c_uint32),
                ("v2", C.c_float * 3), ("flags", C.c_uint32)]


class TriAttr(C.Structure):
    _fields_ = [("n0", C.c_float * 3), ("n1", C.c_float * 3), ("n2", C.c_float * 3),
                ("tc0", C.c_float * 2), ("tc1", C.c_float * 2), ("tc2", C.c_float * 2),
                ("t0", C.c_float * 3), ("t1", C.c_float * 3), ("t2", C.c_float * 3)]


class Instance(C.Structure):
    _fields_ = [("N", C.c_float * 9), ("material", C.c_uint32), ("flags", C.c_uint32), ("animation", C.c_int32)]


class Keyframe(C.Structure):
    _fields_ = [("t", C.c_float), ("translation", C.c_float * 3), ("rotation", C.c_float * 4), ("scaling", C.c_float * 3)]


class Animation(C.Structure):
    _fields_ = [("first_keyframe", C.c_uint32), ("keyframe_count", C.c_uint32)]


class Sphere(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("radius", C.c_float), ("rotation", C.c_float * 4),
                ("material", C.c_uint32), ("animation", C.c_int32), ("reserved", C.c_uint32 * 2)]


class Hotspot(C.Structure):
    _fields_ = [("prim", C.c_uint32), ("transform", C.c_uint32), ("kind", C.c_uint32), ("animation", C.c_int32),
                ("p0", C.c_float * 3), ("p1", C.c_float * 3), ("p2", C.c_float * 3), ("M", C.c_float * 16)]


class Material(C.Structure):
    _fields_ = [("type", C.c_uint32), ("flags", C.c_uint32), ("normal_tex", C.c_int32), ("tex", C.c_int32 * 5),
                ("v", (C.c_float * 4) * 5), ("f", C.c_float * 4)]


class Texture(C.Structure):
    _fields_ = [("type", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32), ("comps", C.c_uint32),
                ("texel_type", C.c_uint32), ("linearize_srgb", C.c_uint32), ("child", C.c_int32),
                ("reserved", C.c_uint32), ("texel_offset", C.c_uint64), ("coord_factor", C.c_float * 2),
                ("coord_offset", C.c_float * 2), ("a", C.c_float * 4), ("b", C.c_float * 4)]


class RglWarp(C.Structure):
    _fields_ = [("size_x", C.c_uint32), ("size_y", C.c_uint32), ("dims", C.c_uint32), ("param_size", C.c_uint32 * 3),
                ("param_stride", C.c_uint32 * 3), ("param_values", C.c_uint32 * 3), ("data", C.c_uint32),
                ("marginal_cdf", C.c_uint32), ("conditional_cdf", C.c_uint32), ("patch_size", C.c_float * 2),
                ("inv_patch_size", C.c_float * 2)]


class RglBrdf(C.Structure):
    _fields_ = [("ndf", RglWarp), ("sigma", RglWarp), ("vndf", RglWarp), ("luminance", RglWarp), ("rgb", RglWarp),
                ("isotropic", C.c_uint32), ("jacobian", C.c_uint32)]


class Envmap(C.Structure):
    _fields_ = [("type", C.c_uint32), ("compat", C.c_uint32), ("tex", C.c_int32), ("N", C.c_int32),
                ("M", C.c_void_p), ("Ms", C.c_void_p), ("Mcs", C.c_void_p), ("cube_tex", C.c_int32 * 6)]


class SceneDesc(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("node_count", C.c_uint32), ("tri_count", C.c_uint32),
                ("instance_count", C.c_uint32), ("material_count", C.c_uint32), ("texture_count", C.c_uint32),
                ("hotspot_count", C.c_uint32), ("sphere_count", C.c_uint32), ("texel_bytes", C.c_uint64),
                ("nodes", C.POINTER(BvhNode)), ("tri_geom", C.POINTER(TriGeom)), ("tri_attr", C.POINTER(TriAttr)),
                ("instances", C.POINTER(Instance)), ("materials", C.POINTER(Material)),
                ("textures", C.POINTER(Texture)), ("texels", C.c_void_p), ("hotspots", C.POINTER(Hotspot)),
                ("envmap", Envmap), ("spheres", C.POINTER(Sphere)), ("rgl_count", C.c_uint32), ("reserved", C.c_uint32),
                ("rgl_data_count", C.c_uint64), ("rgl_brdfs", C.POINTER(RglBrdf)), ("rgl_data", C.POINTER(C.c_float)),
                ("animation_count", C.c_uint32), ("keyframe_count", C.c_uint32), ("animations", C.POINTER(Animation)),
                ("keyframes", C.POINTER(Keyframe))]


class Camera(C.Structure):
    _fields_ = [("l", C.c_float), ("r", C.c_float), ("b", C.c_float), ("t", C.c_float),
                ("translation", C.c_float * 3), ("rotation", C.c_float * 4), ("scaling", C.c_float * 3),
                ("lens_radius", C.c_float), ("focus_dist", C.c_float),
                ("distortion_type", C.c_uint32), ("k1", C.c_float), ("k2", C.c_float), ("k3", C.c_float), ("p1", C.c_float),
                ("p2", C.c_float), ("b1", C.c_float), ("b2", C.c_float), ("b3", C.c_float), ("b4", C.c_float),
                ("dist_center", C.c_float * 2), ("dist_focal_length", C.c_float * 2), ("dist_inverse_focal_length", C.c_float * 2),
                ("surround_mode", C.c_uint32), ("stereoscopic_distance", C.c_float), ("animation", C.c_int32)]


class Params(C.Structure):
    _fields_ = [("max_path_components", C.c_uint32), ("rr_threshold", C.c_float),
                ("randomize_ray_over_pixel", C.c_uint32), ("min_hit_distance", C.c_float),
                ("min_dist_to_light", C.c_float), ("max_dist_to_light", C.c_float),
                ("min_path_len", C.c_float), ("max_path_len", C.c_float), ("t0", C.c_float), ("t1", C.c_float)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("samples", "rays", "node_visits", "leaf_tests", "pdf_tests", "scatters")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


STRUCT_SIZES = {
    "wpt_bvh_node": (BvhNode, 32), "wpt_tri_geom": (TriGeom, 48), "wpt_tri_attr": (TriAttr, 96),
    "wpt_instance": (Instance, 48), "wpt_sphere": (Sphere, 48), "wpt_hotspot": (Hotspot, 116), "wpt_material": (Material, 128),
    "wpt_texture": (Texture, 88), "wpt_rgl_warp": (RglWarp, 76), "wpt_rgl_brdf": (RglBrdf, 388), "wpt_camera": (Camera, 140), "wpt_params": (Params, 40),
    "wpt_keyframe": (Keyframe, 44), "wpt_animation": (Animation, 8),
    "wpt_counters": (Counters, 48),
}
