/*
 * wurblpt_hip.h -- C ABI of the MI355X (gfx950) path-tracing core.
 *
 * This is the drop-in boundary for the one hot path of WurblPT: the per-pixel
 * Monte Carlo integrator.  The reference has no FFI of its own; everything is
 * inlined from headers into the applications (reference libwurblpt/wurblpt.hpp:279-449).
 * The entry points below are what the reference-side `mcpt()` binds instead of
 * running its OpenMP pixel loop (wurblpt.hpp:335-381):
 *
 *   wpt_scene_upload()   replaces the data that `mcpt` borrows from `Scene`:
 *                        scene.bvh() (bvh.hpp:217-225,277-311), the HitableTriangle objects
 *                        (hitable_triangle.hpp:46-143), Mesh vertex data (mesh.hpp:39-66),
 *                        Material / Texture objects (material*.hpp, texture*.hpp),
 *                        scene.hotSpots() and scene.environmentMap() (scene.hpp:173-191)
 *   wpt_render_block()   replaces one iteration of the block loop: the OpenMP pixel loop,
 *                        Prng(pixel), the sample loop, Camera::getRay, tracePath and
 *                        Sensor::finishPixel (wurblpt.hpp:319-383, sensor_rgb.hpp:63-87)
 *   MPICoordinator::getBlock/submitBlock (mpi.hpp:241-262) stay on the caller's side: blocks are
 *   plain (start, size) arguments (include/wurblpt/mpi.hpp, wurblpt_amd/blocks.py)
 *
 * All structs are plain C PODs; the caller keeps ownership of everything it
 * passes in (the callee copies during wpt_scene_upload).  No C++ types, no
 * exceptions and no torch types cross this boundary.  Functions return an
 * integer status; wpt_last_error() gives a thread-local message.
 *
 * Thread compatibility: a wpt_scene belongs to the device that was current
 * when it was uploaded; concurrent wpt_render_block* calls on one scene are
 * allowed when they use different streams and disjoint pixel ranges.
 */
#ifndef WURBLPT_HIP_H
#define WURBLPT_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WPT_ABI_VERSION 5u

typedef enum {
    WPT_OK = 0,
    WPT_ERR_INVALID_ARGUMENT = 1,
    WPT_ERR_NO_DEVICE = 2,
    WPT_ERR_HIP = 3,
    WPT_ERR_UNSUPPORTED = 4,
    WPT_ERR_OUT_OF_MEMORY = 5
} wpt_status;

/* ---- flattened scene ------------------------------------------------- */

/* One node of the depth-first linearized BVH; 32 bytes like BVHNodeLinear
 * (bvh.hpp:217-225).  Child 1 of an inner node is the next node in the array
 * (bvh.hpp:301), child 2 is `link`.  For a leaf, `link` is the index of the triangle or sphere. */
enum { WPT_NODE_INNER = 0, WPT_NODE_TRIANGLE = 1, WPT_NODE_SPHERE = 2, WPT_NODE_EMPTY = 3 };
typedef struct wpt_bvh_node {
    float lo[3];
    float hi[3];
    uint32_t link;
    uint32_t kind;
} wpt_bvh_node;

/* Triangle flags; mirror the HitableTriangle template arguments (hitable_triangle.hpp:36) */
enum { WPT_TRI_HAVE_TEXCOORDS = 1, WPT_TRI_HAVE_TANGENTS = 2, WPT_TRI_TRANSFORM = 4, WPT_TRI_ANIMATE = 8 };

/* Intersection stream: world-space positions of one triangle (48 bytes).  With
 * WPT_TRI_TRANSFORM the positions are the instance's mat4 applied on the host with
 * the arithmetic of hitable_triangle.hpp:203-206, which is the value hit() recomputes per call.
 * With WPT_TRI_ANIMATE the instance's animation at the ray's time is applied to them in the
 * kernel (hitable_triangle.hpp:209-218). */
typedef struct wpt_tri_geom {
    float v0[3];
    uint32_t instance;
    float v1[3];
    uint32_t material;
    float v2[3];
    uint32_t flags;
} wpt_tri_geom;

/* Shading stream: de-indexed vertex attributes of one triangle (96 bytes), read once per
 * final hit (hitable_triangle.hpp:289-322).  Unused members are zero. */
typedef struct wpt_tri_attr {
    float n0[3], n1[3], n2[3];
    float tc0[2], tc1[2], tc2[2];
    float t0[3], t1[3], t2[3];
} wpt_tri_attr;

/* MeshInstance data needed at hit time (mesh.hpp:159-189): the normal matrix (column major). */
typedef struct wpt_instance {
    float N[9];
    uint32_t material;
    uint32_t flags;
    int32_t animation; /* MeshInstance::animationIndex: index into wpt_scene_desc::animations, -1 = none */
} wpt_instance;

/* Animations (animation_keyframes.hpp): key frames sorted by time; the transformation at a time is the
 * first / last key frame outside their range and mix() of the two neighbours inside (translation and
 * scaling linear, rotation by slerp).  The pool also holds the camera's animation (wpt_camera::animation). */
typedef struct wpt_keyframe {
    float t;
    float translation[3];
    float rotation[4]; /* quaternion x, y, z, w */
    float scaling[3];
} wpt_keyframe;
typedef struct wpt_animation {
    uint32_t first_keyframe;
    uint32_t keyframe_count;
} wpt_animation;

/* A sphere (HitableSphere, hitable_sphere.hpp:32-76): centre = T.translation, radius =
 * max(T.scaling), rotation = T.rotation (turns the normal into texture space). 48 bytes. */
typedef struct wpt_sphere {
    float center[3];
    float radius;
    float rotation[4]; /* quaternion x, y, z, w */
    uint32_t material;
    int32_t animation; /* HitableSphere::_animationIndex: index into wpt_scene_desc::animations, -1 = none */
    uint32_t reserved[2];
} wpt_sphere;

/* A hot spot (scene.hpp:113-125).  WPT_HOTSPOT_TRIANGLE: `prim` is the triangle, plus what
 * HitableTriangle::direction() needs (hitable_triangle.hpp:425-443): untransformed positions
 * and the instance mat4.  WPT_HOTSPOT_SPHERE: `prim` is the sphere; nothing else is used
 * (HitableSphere::pdfValue/direction, hitable_sphere.hpp:149-220, read the sphere record). */
enum { WPT_HOTSPOT_TRIANGLE = 0, WPT_HOTSPOT_SPHERE = 1 };
typedef struct wpt_hotspot {
    uint32_t prim;
    uint32_t transform;
    uint32_t kind;
    int32_t animation; /* of the triangle's instance, -1 = none */
    float p0[3], p1[3], p2[3];
    float M[16];
} wpt_hotspot;

enum {
    WPT_MAT_NONE = 0,          /* base Material: no scattering, no emission (material.hpp:158-185) */
    WPT_MAT_LAMBERTIAN = 1,    /* material_lambertian.hpp */
    WPT_MAT_LIGHT_DIFFUSE = 2, /* light_diffuse.hpp */
    WPT_MAT_MIRROR = 3,        /* material_mirror.hpp */
    WPT_MAT_GGX = 4,           /* material_ggx.hpp */
    WPT_MAT_GLASS = 5,         /* material_glass.hpp */
    WPT_MAT_MODPHONG = 6,      /* material_modphong.hpp */
    WPT_MAT_TWOSIDED = 7,      /* material.hpp:273-334 */
    WPT_MAT_RGL = 8            /* material_rgl.hpp:46-102, measured BRDF (powitacq_rgb) */
};
enum {
    WPT_MATF_HAVE_NIR = 1,
    WPT_MATF_CHROMATIC_DISPERSION = 2,
    WPT_MATF_DIFFUSE_TEX_HAS_ALPHA = 4,
    WPT_MATF_SPECULAR_TEX_HAS_ALPHA = 8
};
/* Tagged material record (128 bytes).  Member use per type:
 *  LAMBERTIAN     v[0]=albedo                      tex[0]=albedo
 *  LIGHT_DIFFUSE  v[0]=emit                        tex[0]=emit
 *  MIRROR         v[0]=color                       tex[0]=color
 *  GGX            v[0]=albedo f[0..1]=roughness    tex[0]=albedo tex[1]=roughness
 *  GLASS          v[0]=absorption v[1]=RI material v[2]=RI surrounding
 *  MODPHONG       v[0]=diffuse v[1]=specular v[2]=transmissive v[3]=emissive
 *                 f[0]=shininess f[1]=opacity f[2]=indexOfRefraction
 *                 tex[0]=diffuse tex[1]=specular tex[2]=shininess tex[3]=opacity tex[4]=emissive
 *  TWOSIDED       tex[0]=front material index, tex[1]=back material index
 *  RGL            tex[0]=index into wpt_scene_desc::rgl_brdfs
 * Texture indices are -1 when absent. */
typedef struct wpt_material {
    uint32_t type;
    uint32_t flags;
    int32_t normal_tex;
    int32_t tex[5];
    float v[5][4];
    float f[4];
} wpt_material;

/* One interpolant / sample warp of the measured-BRDF model (Marginal2D<Dimension> of
 * powitacq_rgb.inl:183-640): a size_x x size_y grid of bilinear patches per parameter slice.
 * All arrays live in wpt_scene_desc::rgl_data at the given offsets (in floats) and hold what the
 * model's constructor computes (powitacq_rgb.inl:213-310): `data` normalised, and for the warps
 * that are sampled the marginal and conditional CDFs (WPT_RGL_NONE otherwise). */
#define WPT_RGL_NONE 0xffffffffu
typedef struct wpt_rgl_warp {
    uint32_t size_x, size_y;
    uint32_t dims;            /* 0, 2 or 3 parameters */
    uint32_t param_size[3];
    uint32_t param_stride[3];
    uint32_t param_values[3]; /* offsets of the parameter grids */
    uint32_t data, marginal_cdf, conditional_cdf;
    float patch_size[2], inv_patch_size[2];
} wpt_rgl_warp;

/* powitacq_rgb::BRDF::Data (powitacq_rgb.inl:856-864) */
typedef struct wpt_rgl_brdf {
    wpt_rgl_warp ndf, sigma, vndf, luminance, rgb;
    uint32_t isotropic;
    uint32_t jacobian;
} wpt_rgl_brdf;

enum { WPT_TEX_CONSTANT = 0, WPT_TEX_CHECKER = 1, WPT_TEX_IMAGE = 2, WPT_TEX_TRANSFORMER = 3 };
enum { WPT_TEXEL_U8 = 0, WPT_TEXEL_U16 = 1, WPT_TEXEL_F32 = 2 };
/* Texture record (texture.hpp:160-246, texture_image.hpp:39-233).
 *  CONSTANT     a = color
 *  CHECKER      a = color0, b = color1, width = horiz, height = vert
 *  IMAGE        width/height/comps/texel_type/linearize_srgb, texel_offset (bytes into the
 *               texel pool, row 0 = v 0, x fastest, components interleaved),
 *               coord_factor/coord_offset, a = valFactor, b = valOffset
 *  TRANSFORMER  child, coord_factor/coord_offset, a = valFactor, b = valOffset */
typedef struct wpt_texture {
    uint32_t type;
    uint32_t width, height;
    uint32_t comps;
    uint32_t texel_type;
    uint32_t linearize_srgb;
    int32_t child;
    uint32_t reserved;
    uint64_t texel_offset;
    float coord_factor[2];
    float coord_offset[2];
    float a[4];
    float b[4];
} wpt_texture;

enum { WPT_ENV_NONE = 0, WPT_ENV_EQUIRECT = 1, WPT_ENV_CUBE = 2 };
enum { WPT_ENV_COMPAT_MITSUBA = 0, WPT_ENV_COMPAT_SURROUND_VIDEO = 1 };
/* Environment map (envmap.hpp): texture(s) + host-built importance tables (envmap.hpp:121-158).
 * N == 0 means no importance sampling support.  EQUIRECT uses `tex` (envmap.hpp:213-247),
 * CUBE uses `cube_tex` in the order +x -x +y -y +z -z (envmap.hpp:250-285). */
typedef struct wpt_envmap {
    uint32_t type;
    uint32_t compat;
    int32_t tex;
    int32_t N;
    const float* M;
    const int32_t* Ms;
    const float* Mcs;
    int32_t cube_tex[6];
} wpt_envmap;

typedef struct wpt_scene_desc {
    uint32_t abi_version; /* WPT_ABI_VERSION */
    uint32_t node_count;
    uint32_t tri_count;
    uint32_t instance_count;
    uint32_t material_count;
    uint32_t texture_count;
    uint32_t hotspot_count;
    uint32_t sphere_count;
    uint64_t texel_bytes;
    const wpt_bvh_node* nodes;
    const wpt_tri_geom* tri_geom;
    const wpt_tri_attr* tri_attr;
    const wpt_instance* instances;
    const wpt_material* materials;
    const wpt_texture* textures;
    const uint8_t* texels;
    const wpt_hotspot* hotspots;
    wpt_envmap envmap;
    const wpt_sphere* spheres;
    uint32_t rgl_count;      /* measured BRDFs (WPT_MAT_RGL) */
    uint32_t reserved;
    uint64_t rgl_data_count; /* floats in rgl_data */
    const wpt_rgl_brdf* rgl_brdfs;
    const float* rgl_data;
    uint32_t animation_count;
    uint32_t keyframe_count;
    const wpt_animation* animations;
    const wpt_keyframe* keyframes;
} wpt_scene_desc;

/* ---- camera, parameters ---------------------------------------------- */

enum { WPT_SURROUND_OFF = 0, WPT_SURROUND_180 = 1, WPT_SURROUND_360 = 2 };
enum { WPT_DISTORTION_NONE = 0, WPT_DISTORTION_RADIAL_AND_PLANAR = 1, WPT_DISTORTION_RADIAL_ONLY = 2, WPT_DISTORTION_OPENCV = 3 };
/* What Camera::getRay needs for a static pinhole / thin lens camera
 * (camera.hpp:123-185, optics.hpp:37-69,311-334, transformation.hpp:48-83). */
typedef struct wpt_camera {
    float l, r, b, t;        /* Projection frustum at near = 1 */
    float translation[3];
    float rotation[4];       /* quaternion x, y, z, w */
    float scaling[3];
    float lens_radius;
    float focus_dist;
    /* LensDistortion (optics.hpp:112-309) and its per-frame helper (:203-212, from the projection):
     * undistort() maps the distorted image coordinates of a sample to the ones a ray is made from */
    uint32_t distortion_type; /* WPT_DISTORTION_* */
    float k1, k2, k3, p1, p2;
    float b1, b2, b3, b4;     /* RadialOnly: coefficients of the exact inverse (:176-180) */
    float dist_center[2], dist_focal_length[2], dist_inverse_focal_length[2];
    /* Camera::surroundMode and ::stereoscopicDistance (camera.hpp:45-52,128-170): 180 / 360 degree
     * cameras ignore the optics; a stereoscopic camera renders the left view into the upper half */
    uint32_t surround_mode; /* WPT_SURROUND_* */
    float stereoscopic_distance;
    /* Camera::animation: index into the scene's animation pool or -1.  translation / rotation / scaling
     * above hold Camera::at(t0); with t0 != t1 a ray takes the animation at its own time (camera.hpp:175-180). */
    int32_t animation;
} wpt_camera;

/* Parameters (wurblpt.hpp:79-96) plus the SensorRGB gates (sensor_rgb.hpp:41-51). */
typedef struct wpt_params {
    uint32_t max_path_components;
    float rr_threshold;
    uint32_t randomize_ray_over_pixel;
    float min_hit_distance;
    float min_dist_to_light, max_dist_to_light;
    float min_path_len, max_path_len;
    float t0, t1; /* mcpt()'s exposure interval: with t0 != t1 every camera ray draws its time in it (motion blur) */
} wpt_params;

/* Work counters of one render call; the roofline denominator. */
typedef struct wpt_counters {
    uint64_t samples;
    uint64_t rays;          /* BVH::hit calls */
    uint64_t node_visits;   /* nodes fetched in BVH::hit */
    uint64_t leaf_tests;    /* triangle tests in BVH::hit */
    uint64_t pdf_tests;     /* hot-spot pdfValue triangle tests */
    uint64_t scatters;      /* Material::scatter calls */
} wpt_counters;

typedef struct wpt_scene wpt_scene;

/* ---- entry points ---------------------------------------------------- */

/* Number of HIP devices (0 if none); selects the device for this thread. */
int wpt_device_count(void);
wpt_status wpt_select_device(int device);
wpt_status wpt_current_device(int* device); /* the calling thread's HIP device */

/* Copies the flattened scene to the current device. */
wpt_status wpt_scene_upload(const wpt_scene_desc* desc, wpt_scene** out_scene);
void wpt_scene_free(wpt_scene* scene);

/* Renders pixels [block_start, block_start + block_size) of a width x height frame with
 * samples_sqrt^2 samples per pixel, asynchronously on `hip_stream` (NULL = default stream).
 * `frame_device` is a device pointer to the FULL frame, float[height][width][3], row 0 =
 * bottom row (camera.hpp:146-149); only the block's pixels are written.
 * `counters_device` may be NULL; otherwise a device pointer to one wpt_counters that the
 * kernel atomically adds to. */
wpt_status wpt_render_block_device(wpt_scene* scene, const wpt_camera* camera,
        const wpt_params* params, uint32_t width, uint32_t height, uint32_t samples_sqrt,
        uint32_t block_start, uint32_t block_size,
        float* frame_device, wpt_counters* counters_device, void* hip_stream);

/* One rank's interleaved share of the frame in ONE launch: the frame is cut into bands of `band_rows` rows and this
 * call renders bands first_band, first_band + band_stride, first_band + 2 band_stride, ... (rank r of N: first_band = r,
 * band_stride = N).  Same results as rendering those bands as blocks; a GPU keeps all of the rank's pixels resident
 * without one stream per block.  band_rows a multiple of 8 (and width too) keeps the 8x8 pixel tiles per wave. */
wpt_status wpt_render_bands_device(wpt_scene* scene, const wpt_camera* camera,
        const wpt_params* params, uint32_t width, uint32_t height, uint32_t samples_sqrt,
        uint32_t band_rows, uint32_t first_band, uint32_t band_stride,
        float* frame_device, wpt_counters* counters_device, void* hip_stream);

/* Synchronous form for host frames: renders the same bands and writes their pixels into `frame_host`, the FULL frame
 * float[height][width][3] in host memory; the other bands are left as they are (several devices fill one frame). */
wpt_status wpt_render_bands(wpt_scene* scene, const wpt_camera* camera, const wpt_params* params,
        uint32_t width, uint32_t height, uint32_t samples_sqrt, uint32_t band_rows, uint32_t first_band, uint32_t band_stride,
        float* frame_host);

/* Synchronous form with MPICoordinator::submitBlock semantics (mpi.hpp:256-262):
 * writes block_size*3 floats for the block's pixels to host memory `block_rgb`. */
wpt_status wpt_render_block(wpt_scene* scene, const wpt_camera* camera,
        const wpt_params* params, uint32_t width, uint32_t height, uint32_t samples_sqrt,
        uint32_t block_start, uint32_t block_size, float* block_rgb);

/* Waits for the device; WPT_ERR_HIP if a launch since the last call failed (the kernels have no waits that could run
 * out: every loop of theirs ends with its work). */
wpt_status wpt_scene_check(wpt_scene* scene);

/* ---- ground truth (GroundTruth / getGroundTruth, wurblpt.hpp:453-769) ----
 * One ray through the centre of every pixel, without pixel jitter or lens sampling; arrays of what its
 * first hit is (zero where nothing is hit, material -1).  Array k is the reference's GroundTruth bit k. */
enum {
    WPT_GT_WORLD_SPACE_POSITIONS = 0,          /* 3 floats per pixel */
    WPT_GT_WORLD_SPACE_GEOMETRY_NORMALS = 1,   /* 3 */
    WPT_GT_WORLD_SPACE_GEOMETRY_TANGENTS = 2,  /* 3 */
    WPT_GT_WORLD_SPACE_MATERIAL_NORMALS = 3,   /* 3: after the material's normal map */
    WPT_GT_WORLD_SPACE_MATERIAL_TANGENTS = 4,  /* 3 */
    WPT_GT_CAMERA_SPACE_POSITIONS = 5,         /* 3 */
    WPT_GT_CAMERA_SPACE_GEOMETRY_NORMALS = 6,  /* 3 */
    WPT_GT_CAMERA_SPACE_GEOMETRY_TANGENTS = 7, /* 3 */
    WPT_GT_CAMERA_SPACE_MATERIAL_NORMALS = 8,  /* 3 */
    WPT_GT_CAMERA_SPACE_MATERIAL_TANGENTS = 9, /* 3 */
    WPT_GT_CAMERA_SPACE_DEPTHS = 10,           /* 1: -z of the camera space position */
    WPT_GT_CAMERA_SPACE_DISTANCES = 11,        /* 1: its length */
    WPT_GT_TEXCOORDS = 12,                     /* 2 */
    WPT_GT_WORLD_SPACE_OFFSET_TO_PREV = 13,    /* 3: where the hit point of an animated instance is at tPrev / tNext, minus where it is */
    WPT_GT_WORLD_SPACE_OFFSET_TO_NEXT = 14,    /* 3 */
    WPT_GT_CAMERA_SPACE_OFFSET_TO_PREV = 15,   /* 3: from the camera at tPrev / tNext */
    WPT_GT_CAMERA_SPACE_OFFSET_TO_NEXT = 16,   /* 3 */
    WPT_GT_PIXEL_SPACE_OFFSET_TO_PREV = 17,    /* 2: Surround_Off, non-stereoscopic cameras only (camera.hpp:207-208) */
    WPT_GT_PIXEL_SPACE_OFFSET_TO_NEXT = 18,    /* 2 */
    WPT_GT_MATERIALS = 19,                     /* 1 int32: index into wpt_scene_desc::materials of the hitable's material */
    WPT_GT_ARRAY_COUNT = 20
};
/* components per pixel of array k */
static const uint32_t wpt_gt_components[WPT_GT_ARRAY_COUNT] = { 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 1, 1, 2, 3, 3, 3, 3, 2, 2, 1 };

/* arrays_device[k]: device array of width * height * wpt_gt_components[k] elements (row 0 = bottom), or
 * NULL for an array that is not wanted.  camera_prev / camera_next: the camera at tPrev / tNext (only the
 * transformation is read); NULL = the camera itself.  times: t0, tPrev, tNext for the scene's animated
 * instances (NULL = all zero); the picture is taken at t0.  Asynchronous on `hip_stream`. */
wpt_status wpt_ground_truth_device(wpt_scene* scene, const wpt_camera* camera, const wpt_camera* camera_prev,
        const wpt_camera* camera_next, const float times[3], const wpt_params* params, uint32_t width, uint32_t height,
        void* const arrays_device[WPT_GT_ARRAY_COUNT], void* hip_stream);
/* The same into host arrays (synchronous). */
wpt_status wpt_ground_truth(wpt_scene* scene, const wpt_camera* camera, const wpt_camera* camera_prev,
        const wpt_camera* camera_next, const float times[3], const wpt_params* params, uint32_t width, uint32_t height,
        void* const arrays_host[WPT_GT_ARRAY_COUNT]);

/* ---- output side (postproc.hpp:44-108): per-pixel operations on a rendered frame ----
 * Device forms work on `pixels` RGB triples in device memory on `hip_stream`.
 *   to_srgb                          linear RGB float -> sRGB uint8 (values above 1 clipped), toSRGB()
 *   max_luminance                    largest CIE Y of the frame, maxLuminance() (synchronises)
 *   uniform_rational_quantization    Schlick's operator on Y, chromaticity kept
 *   scale_luminance                  Y * factor, clamped to 100 * clamp if clamp > 0 */
wpt_status wpt_postproc_to_srgb(const float* rgb_device, uint8_t* srgb_device, uint64_t pixels, void* hip_stream);
wpt_status wpt_postproc_max_luminance(const float* rgb_device, uint64_t pixels, float* result_host, void* hip_stream);
wpt_status wpt_postproc_uniform_rational_quantization(const float* rgb_device, float* out_device, uint64_t pixels,
        float max_val, float brightness, void* hip_stream);
wpt_status wpt_postproc_scale_luminance(const float* rgb_device, float* out_device, uint64_t pixels, float factor,
        float clamp, void* hip_stream);
/* The same for host buffers (upload, run, download): op 0 = to_srgb (out: uint8), 1 = uniform rational
 * quantization (a = max_val, b = brightness; out: float), 2 = scale luminance (a = factor, b = clamp; out:
 * float), 3 = max luminance (out: one float). */
wpt_status wpt_postproc_host(int op, const float* rgb_host, void* out_host, uint64_t pixels, float a, float b);

/* Kernel launch geometry knobs (0 = default); for benchmarking only, results do not change.  wpt_set_launch_config,
 * wpt_set_top_nodes and wpt_set_wavefront are PROCESS-GLOBAL hooks for tests and measurements: set them before rendering
 * starts, not while other threads render (MPICoordinator's worker threads read them).
 * variant, byte 0: 0x01 scene from HBM even if it fits LDS, 0x02 all-features kernel, 0x20 separate SHADE / NEE-END / NEW rounds, 0x10 no pixel pool (every lane renders the one pixel it was launched for), 0x80 material records from HBM even where they fit into LDS next to the scene, 0x40 never two passes over a frame (timed first row of strata, then the rest with the longest tiles first; scenes fetched from HBM), bits 0x0c: a kind of material with few lanes in a long round stands back once (0 = fewer than 6 lanes, 0x04 = never, 0x08 = fewer than 3, 0x0c = fewer than 12; kernels without textures / spheres / environment only); byte 1: leave threshold of the traversal loop in eighths + 1; byte 2: lanes a long round needs + 1; byte 3:
 * leaf bias (DESIGN.md section 4 has what each was measured to do). */
wpt_status wpt_set_launch_config(uint32_t threads_per_group, uint32_t variant);
/* Storage order of the BVH nodes in HBM for scenes uploaded from now on: the first `nodes` nodes of a tree that is
 * larger than an L2 slice are stored level by level in front of the array, the subtrees below them depth-first
 * (0 = the whole tree depth-first, the reference's own array order; default 65536 = 2 MiB).  Visiting order and results do
 * not depend on it. */
wpt_status wpt_set_top_nodes(uint32_t nodes);
/* How rays walk the tree (results do not depend on it; process-global like the hooks above, set it before uploading and
 * rendering, not while other threads render):
 *   WPT_WALK_WIDE           scenes uploaded from now on also get the tree collapsed by one level (128-byte nodes that hold the
 *                           boxes of a node's four grandchildren), and product launches that fetch the scene from HBM walk that:
 *                           four box tests per fetch, leaf tests in BVH::hit's order (bvh.hpp:277-311), the same hits bit for bit
 *                           (wpt_pathtrace.inc.h says why; rays for which the argument does not hold walk the binary tree).
 *                           Trees with a non-finite box, a child's box outside its parent's, or a worst case of more than 96
 *                           waiting entries have no wide form and are walked as before.
 *   WPT_WALK_FULL_SHADOW    light rays towards the environment walk the tree to the end like the reference's (product launches
 *                           end such a walk at its first accepted hit: the answer the ray is traced for is known there)
 *   WPT_WALK_COUNT_PRODUCT  counting launches, which otherwise walk like the reference so that their counters are its
 *                           counters, count the product's shortened walks instead
 *   WPT_WALK_TRIANGLES_AS_GIVEN  scenes uploaded from now on keep their triangle records in the caller's order (measurements; by
 *                           default the records are stored in the order of their leaves in the tree, so that a subtree's
 *                           triangles share cache lines) */
#define WPT_WALK_WIDE 1u
#define WPT_WALK_FULL_SHADOW 2u
#define WPT_WALK_COUNT_PRODUCT 4u
#define WPT_WALK_TRIANGLES_AS_GIVEN 8u
wpt_status wpt_set_walk(uint32_t flags);
/* Which form of the path tracer renders frames whose scene is fetched from HBM (results do not depend on it):
 * mode 0 = the library decides per launch (default), 1 = the wavefront form wherever it exists (trace and shade as two
 * kernels that hand rays through HBM, wpt_wavefront.inc.h: everything but counting launches and moving scenes), 2 = never.
 * groups, bits 0-7: groups of lanes that iterate on streams of their own (0 = default), bits 8-15 (measurements): workgroups
 * of the trace kernel per compute unit (0 = what fits), bit 16 (measurements): one shade launch per kind of material, so that a
 * kernel trace tells the kinds apart; chunk: queue entries a wave of the trace
 * takes per atomic (0 = default); flags bit 0: the shade walks the ray queue in its own order instead of by kind of
 * material, bits 1-7: nodes in front of the node array that the trace walks from LDS, in units of 128 (0 = default, 0x7f =
 * none), bits 8-13: lanes of a wave that must have finished before the trace deals it new rays (0 = default), bits 16-31: node
 * steps a ray takes per launch of the trace before its walk is suspended until the next (0 = default, 0xffff = no limit).
 * Process-global like wpt_set_launch_config and wpt_set_top_nodes: a hook for tests and measurements, set it before
 * rendering starts, not while other threads render. */
wpt_status wpt_set_wavefront(uint32_t mode, uint32_t groups, uint32_t chunk, uint32_t flags);
/* The library reads no environment variable. */

/* Profiling hook: `stats_device` (device pointer to WPT_SCHED_STATS uint64, or NULL to switch off) receives
 * the wave scheduler's statistics of launches that also count work (counters_device != NULL):
 * [0] NODE rounds [1] NODE loop iterations [2] sum of lanes active in them, then (rounds, lanes)
 * for LEAF [3,4], SHADE [5,6], NEE-END [7,8], NEW [9,10]; shader-clock ticks a wave spent in
 * traversal [11], SHADE [12], NEE-END [13], NEW [14]; [15] unused; [16..23] shader-clock ticks summed
 * over lanes per section of the SHADE block (hit record, scatter, emission, light pdf 1, light sample,
 * light pdf 2, evaluation towards the light, environment sampling / continuation); [24 .. 47] executions of each stretch of
 * the kernel's code by waves (at least one lane ran it) and [48 .. 71] by lanes, in the order of wpt_blocks.h's SEC_* (node step,
 * leaf test, ray start, ...: tools/instruction_budget.py multiplies them with the stretches' instruction counts). */
#define WPT_SCHED_STATS 72
wpt_status wpt_set_scheduler_stats(unsigned long long* stats_device);

/* Kernel family of the process's most recent render call (for profile matching). */
const char* wpt_kernel_name(void);
/* What the reference records about a run for the CPU (wurblpt.hpp:393-400,425-435: COMPILER, CPU_MODEL), for the device:
 * marketing name and architecture of HIP device `device` ("AMD Instinct MI355X (gfx950:...)", or "" if there is none), and
 * the compiler and options the kernels were built with.  The strings live until the next call from the same thread. */
const char* wpt_device_name(int device);
/* Kernel launches the process's most recent render call took for its pixels: 1, or 2 when the frame was rendered in two
 * passes (timed first row of strata, then the rest with the longest tiles first), or the hundreds of trace + shade
 * launches of the wavefront form. Profilers see that many kernel launches per frame; what is rendered does not depend on it. */
uint32_t wpt_last_render_passes(void);
const char* wpt_build_info(void);

const char* wpt_last_error(void);

#ifdef __cplusplus
}
#endif

#endif
