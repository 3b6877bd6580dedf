/*
 * wpt_capi.hip -- the C ABI of include/wurblpt_hip.h: scene upload (validation, conversion of
 * the BVH to the device's stackless node form, environment importance tables), kernel
 * selection and launch.  The kernel itself is in wpt_pathtrace.inc.h.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/wurblpt_hip.h"
#include "wpt_pathtrace.inc.h"
#include "wpt_wavefront.inc.h"
#include "wpt_postproc.h"

using namespace wptd;
using namespace wptk;

namespace {

/* Bit-parity self test of the arithmetic the kernel relies on: ops 0..5 are the
 * transcendentals of wpt_math.h, 6 = IEEE division, 7 = IEEE square root, 10 acos, 11 atan2(x, 1), 12 float(2 * asin(double)). */
__global__ void wpt_selftest_kernel(int op, int n, const float* a, const float* b, float* out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    float x = a[i], y = b[i], r;
    switch (op) {
    case 0: r = wptm::sinf_(x); break;
    case 1: r = wptm::cosf_(x); break;
    case 2: r = wptm::expf_(x); break;
    case 3: r = wptm::powf_(x, y); break;
    case 4: r = wptm::asinf_(x); break;
    case 5: r = wptm::atan2f_(x, y); break;
    case 6: r = x / y; break;
    case 7: r = __builtin_sqrtf(x); break;
    case 8: r = x * y + x; break; /* must stay unfused */
    case 10: r = wptm::acosf_(x); break;
    case 11: r = wptm::atan2f_(x, 1.0f); break;
    case 12: r = (float)(2.0 * wptm::asin_d((double)x)); break;
    default: r = 1.0f / x; break;
    }
    out[i] = r;
}

/* AABB::mayHit as the kernels evaluate it; same argument layout as the oracle's probe:
 * boxes lo(3) hi(3); rays origin(3) dir(3) amin amax */
__global__ void wpt_selftest_aabb_kernel(int n, const float* boxes, const float* rays, int32_t* out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const RayAux h = rayAux(ld3(rays + 8 * i + 3));
    out[i] = boxTest(ld3(boxes + 6 * i), ld3(boxes + 6 * i + 3), ld3(rays + 8 * i), h.inv, rays[8 * i + 6], rays[8 * i + 7]) ? 1 : 0;
}

/* per-bin importance of the environment map (envmap.hpp:128-140) */
__global__ void wpt_env_importance_kernel(SceneView sv, int N, float* importance)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * N)
        return;
    int x = i % N, y = i / N;
    f2 uv;
    uv.y = ((float)y + 0.5f) / (float)N;
    uv.x = ((float)x + 0.5f) / (float)N;
    f4 L = envL(sv, envInvM(uv));
    importance[i] = L.x + L.y + L.z + L.w;
}

/* per triangle hot spot: what its pdf needs of the corners alone (wpt_blocks.h hotSpotFace; the corners are the world-space
 * positions the walk tests, tri_geom) */
__global__ void wpt_hotspot_face_kernel(SceneView sv, float4* face)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= sv.hotspotCount)
        return;
    float4 f = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (sv.hotspots[i].kind != WPT_HOTSPOT_SPHERE) {
        const uint32_t p = sv.hotspots[i].prim;
        const float4 g0 = sv.triGeom[3 * p], g1 = sv.triGeom[3 * p + 1], g2 = sv.triGeom[3 * p + 2];
        f = wptk::hotSpotFace(mk3(g0.x, g0.y, g0.z), mk3(g1.x, g1.y, g1.z), mk3(g2.x, g2.y, g2.z));
    }
    face[i] = f;
}

/* decodes one image texture into the RGBA float4 pool (see imageTexelDecode) */
__global__ void wpt_expand_texels_kernel(const uint8_t* pool, const wpt_texture t, float4* out)
{
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= size_t(t.width) * t.height)
        return;
    const f4 v = imageTexelDecode(pool, t, i % t.width, i / t.width);
    out[i] = make_float4(v.x, v.y, v.z, v.w);
}

/* ---- output side: one thread per pixel (wpt_postproc.h) ---- */
struct DevicePow {
    static __device__ __forceinline__ float pow(float x, float y) { return wptm::powf_(x, y); }
};
__global__ void wpt_postproc_kernel(int op, const float* in, void* out, uint64_t pixels, float a, float b, uint32_t* maxBits)
{
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= pixels)
        return;
    wptpp::V3 rgb;
    rgb.x = in[3 * i];
    rgb.y = in[3 * i + 1];
    rgb.z = in[3 * i + 2];
    if (op == 0) {
        uint8_t* o = static_cast<uint8_t*>(out) + 3 * i;
        o[0] = wptpp::toSrgbByte<DevicePow>(rgb.x);
        o[1] = wptpp::toSrgbByte<DevicePow>(rgb.y);
        o[2] = wptpp::toSrgbByte<DevicePow>(rgb.z);
    } else if (op == 3) {
        /* maximum of non-negative floats = maximum of their bit patterns; NaN and negative values never win,
         * as in the sequential `if (y > lum)` starting from 0 */
        const float y = wptpp::luminance(rgb);
        if (y > 0.0f)
            atomicMax(maxBits, __float_as_uint(y));
    } else {
        const wptpp::V3 r = op == 1 ? wptpp::uniformRationalQuantization(rgb, a, b) : wptpp::scaleLuminance(rgb, a, b);
        float* o = static_cast<float*>(out) + 3 * i;
        o[0] = r.x;
        o[1] = r.y;
        o[2] = r.z;
    }
}

/* ---- host side of the C ABI ---- */

thread_local std::string g_error;

wpt_status fail(wpt_status s, const std::string& msg)
{
    g_error = msg;
    return s;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(WPT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));           \
    } while (0)

} /* namespace */

struct wpt_scene {
    int device;
    SceneView view;
    uint32_t features;
    uint32_t nodeCount, triCount;
    uint32_t animationCount;
    std::vector<void*> allocations;
    int cuCount;
    std::vector<float> envM, envMcs;
    std::vector<int32_t> envMs;
};

namespace {

uint32_t g_threadsPerGroup = WG;
uint32_t g_variant = 0;
uint32_t g_leaveEighths = 0; /* 0 = default: chosen per scene size (single-role kernel) / patience 8 rounds (ray-pool kernel) */
uint32_t g_heavyMin = 0; /* 0 = chosen per scene size at launch */
uint32_t g_leafBias = 0;
/* what the process's most recent render call ran: kernel launches it took for its pixels (wpt_last_render_passes) and which
 * kernel family (wpt_kernel_name).  Process-wide, so that a caller whose worker threads render (MPICoordinator, bench.py's
 * block queue) reads on its main thread what the workers ran. */
std::atomic<uint32_t> g_lastPasses{1};
std::atomic<const char*> g_kernelName{nullptr};
uint32_t g_topNodes = 65536; /* nodes of a large tree that are stored level by level in front (wpt_set_top_nodes) */
unsigned long long* g_schedStats = nullptr;
/* wpt_set_wavefront: 0 = the library decides, 1 = wavefront wherever it exists, 2 = never; launch geometry (0 = defaults) */
uint32_t g_wfMode = 0;
wptk::WfConfig g_wfConfig = { 0, 0, 0, 1, 0, 0, 0, 0, 0 };
/* wpt_set_walk: WPT_WALK_* bits */
uint32_t g_walk = 0;

template<typename T> wpt_status uploadArray(wpt_scene* s, const T* src, size_t count, const T** dst)
{
    *dst = nullptr;
    size_t bytes = count * sizeof(T);
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes > 0 ? bytes : 16);
    if (e != hipSuccess)
        return fail(e == hipErrorOutOfMemory ? WPT_ERR_OUT_OF_MEMORY : WPT_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
    s->allocations.push_back(p);
    if (bytes > 0)
        HIP_TRY(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
    *dst = static_cast<const T*>(p);
    return WPT_OK;
}

uint32_t sceneFeatures(const wpt_scene_desc* d)
{
    uint32_t f = 0;
    for (uint32_t i = 0; i < d->material_count; i++) {
        const wpt_material& m = d->materials[i];
        if (m.type == WPT_MAT_MODPHONG)
            f |= FEAT_MODPHONG;
        if (m.type == WPT_MAT_TWOSIDED)
            f |= FEAT_TWOSIDED;
        if (m.type == WPT_MAT_GGX)
            f |= FEAT_GGX;
        if (m.type == WPT_MAT_GLASS || m.type == WPT_MAT_MIRROR)
            f |= FEAT_GLASS;
        bool tex = m.normal_tex >= 0;
        if (m.type != WPT_MAT_TWOSIDED)
            for (int k = 0; k < 5; k++)
                tex = tex || m.tex[k] >= 0;
        if (tex)
            f |= FEAT_TEXTURES;
    }
    if (d->envmap.type != WPT_ENV_NONE)
        f |= FEAT_ENVMAP | FEAT_TEXTURES;
    if (d->sphere_count > 0)
        f |= FEAT_SPHERES;
    for (uint32_t i = 0; i < d->material_count; i++)
        if (d->materials[i].type == WPT_MAT_RGL)
            f |= FEAT_RGL;
    for (uint32_t i = 0; i < d->instance_count; i++)
        if (d->instances[i].animation >= 0)
            f |= FEAT_ANIM;
    for (uint32_t i = 0; i < d->sphere_count; i++)
        if (d->spheres[i].animation >= 0)
            f |= FEAT_ANIM;
    return f;
}

wpt_status validate(const wpt_scene_desc* d)
{
    if (!d)
        return fail(WPT_ERR_INVALID_ARGUMENT, "scene description is NULL");
    if (d->abi_version != WPT_ABI_VERSION)
        return fail(WPT_ERR_INVALID_ARGUMENT, "scene description has a different ABI version");
    if (d->node_count == 0 || !d->nodes)
        return fail(WPT_ERR_INVALID_ARGUMENT, "scene has no BVH nodes (run Scene::updateBVH)");
    /* every index the kernel will follow must stay inside its array: a bad index would be an
     * out-of-bounds access on the GPU */
    for (uint32_t i = 0; i < d->node_count; i++) {
        const wpt_bvh_node& n = d->nodes[i];
        if (n.kind == WPT_NODE_INNER) {
            if (n.link >= d->node_count || n.link <= i || i + 1 >= d->node_count)
                return fail(WPT_ERR_INVALID_ARGUMENT, "BVH inner node links outside the node array");
        } else if (n.kind == WPT_NODE_TRIANGLE) {
            if (n.link >= d->tri_count || n.link >= PRIM_SPHERE)
                return fail(WPT_ERR_INVALID_ARGUMENT, "BVH leaf references a triangle outside the array");
        } else if (n.kind == WPT_NODE_SPHERE) {
            if (n.link >= d->sphere_count || n.link >= (NODE_CHILD & ~PRIM_SPHERE))
                return fail(WPT_ERR_INVALID_ARGUMENT, "BVH leaf references a sphere outside the array");
        } else if (n.kind != WPT_NODE_EMPTY) {
            return fail(WPT_ERR_UNSUPPORTED, "BVH node kind is not known to the kernel");
        }
    }
    {
        /* ... and the links must describe ONE depth-first tree over all nodes: the first child of an inner node is the
         * next node, its second child (link) starts where the first child's subtree ends.  Links that are merely in
         * range could share children (the device form would grow without bound) or leave nodes unreachable (the walk
         * would run into records nobody wrote).  One reverse pass: end[i] = first node behind the subtree of node i. */
        std::vector<uint32_t> end(d->node_count);
        for (uint32_t i = d->node_count; i-- > 0;) {
            const wpt_bvh_node& n = d->nodes[i];
            if (n.kind == WPT_NODE_INNER) {
                if (n.link != end[i + 1])
                    return fail(WPT_ERR_INVALID_ARGUMENT, "BVH nodes are not one depth-first tree (second child does not follow the first child's subtree)");
                end[i] = end[n.link];
            } else {
                end[i] = i + 1;
            }
        }
        if (end[0] != d->node_count)
            return fail(WPT_ERR_INVALID_ARGUMENT, "BVH nodes are not one depth-first tree (nodes behind the root's subtree)");
    }
    for (uint32_t i = 0; i < d->tri_count; i++) {
        if (d->tri_geom[i].instance >= d->instance_count || d->tri_geom[i].material >= d->material_count)
            return fail(WPT_ERR_INVALID_ARGUMENT, "triangle references an instance or material outside the arrays");
    }
    for (uint32_t i = 0; i < d->material_count; i++) {
        const wpt_material& m = d->materials[i];
        if (m.type > WPT_MAT_RGL)
            return fail(WPT_ERR_UNSUPPORTED, "material type is not known to the kernel");
        if (m.type == WPT_MAT_RGL) {
            if (m.tex[0] < 0 || uint32_t(m.tex[0]) >= d->rgl_count)
                return fail(WPT_ERR_INVALID_ARGUMENT, "material references a measured BRDF outside the array");
            if (m.normal_tex >= int32_t(d->texture_count))
                return fail(WPT_ERR_INVALID_ARGUMENT, "material references a normal map outside the array");
            continue;
        }
        if (m.type == WPT_MAT_TWOSIDED) {
            if (m.tex[0] < 0 || m.tex[1] < 0 || uint32_t(m.tex[0]) >= d->material_count || uint32_t(m.tex[1]) >= d->material_count)
                return fail(WPT_ERR_INVALID_ARGUMENT, "two-sided material references a material outside the array");
        } else {
            for (int k = 0; k < 5; k++)
                if (m.tex[k] >= int32_t(d->texture_count))
                    return fail(WPT_ERR_INVALID_ARGUMENT, "material references a texture outside the array");
        }
        if (m.normal_tex >= int32_t(d->texture_count))
            return fail(WPT_ERR_INVALID_ARGUMENT, "material references a normal map outside the array");
    }
    for (uint32_t i = 0; i < d->texture_count; i++) {
        const wpt_texture& t = d->textures[i];
        if (t.type > WPT_TEX_TRANSFORMER)
            return fail(WPT_ERR_UNSUPPORTED, "texture type is not known to the kernel");
        if (t.type == WPT_TEX_TRANSFORMER && (t.child < 0 || uint32_t(t.child) >= d->texture_count || uint32_t(t.child) >= i))
            return fail(WPT_ERR_INVALID_ARGUMENT, "texture transformer references a texture outside the array");
        if (t.type == WPT_TEX_IMAGE) {
            size_t cs = t.texel_type == WPT_TEXEL_U8 ? 1 : t.texel_type == WPT_TEXEL_U16 ? 2 : 4;
            if (t.width == 0 || t.height == 0 || t.comps < 1 || t.comps > 4 || t.texel_type > WPT_TEXEL_F32
                    || t.texel_offset + size_t(t.width) * t.height * t.comps * cs > d->texel_bytes)
                return fail(WPT_ERR_INVALID_ARGUMENT, "image texture lies outside the texel pool");
        }
    }
    if (d->rgl_count > 0 && (!d->rgl_brdfs || !d->rgl_data))
        return fail(WPT_ERR_INVALID_ARGUMENT, "measured BRDF arrays are NULL");
    for (uint32_t i = 0; i < d->rgl_count; i++) {
        /* every table of the model must lie inside the pool (the kernel indexes it with data-dependent offsets) */
        const wpt_rgl_brdf& b = d->rgl_brdfs[i];
        const wpt_rgl_warp* warps[5] = { &b.ndf, &b.sigma, &b.vndf, &b.luminance, &b.rgb };
        const uint32_t wantDims[5] = { 0, 0, 2, 2, 3 };
        for (int k = 0; k < 5; k++) {
            const wpt_rgl_warp& w = *warps[k];
            if (w.dims != wantDims[k] || w.size_x < 2 || w.size_y < 2)
                return fail(WPT_ERR_INVALID_ARGUMENT, "measured BRDF table has an unexpected shape");
            uint64_t slices = 1;
            for (uint32_t dim = 0; dim < w.dims; dim++) {
                if (w.param_size[dim] < 1 || uint64_t(w.param_values[dim]) + w.param_size[dim] > d->rgl_data_count)
                    return fail(WPT_ERR_INVALID_ARGUMENT, "measured BRDF parameter grid lies outside the pool");
                slices *= w.param_size[dim];
            }
            const uint64_t n = uint64_t(w.size_x) * w.size_y;
            const bool cdf = k == 2 || k == 3;
            if (uint64_t(w.data) + slices * n > d->rgl_data_count
                    || (cdf && (w.marginal_cdf == WPT_RGL_NONE || w.conditional_cdf == WPT_RGL_NONE
                            || uint64_t(w.marginal_cdf) + slices * w.size_y > d->rgl_data_count
                            || uint64_t(w.conditional_cdf) + slices * n > d->rgl_data_count)))
                return fail(WPT_ERR_INVALID_ARGUMENT, "measured BRDF table lies outside the pool");
        }
    }
    if (d->sphere_count > 0 && !d->spheres)
        return fail(WPT_ERR_INVALID_ARGUMENT, "sphere array is NULL");
    for (uint32_t i = 0; i < d->sphere_count; i++) {
        if (d->spheres[i].material >= d->material_count)
            return fail(WPT_ERR_INVALID_ARGUMENT, "sphere references a material outside the array");
        if (d->spheres[i].animation >= int32_t(d->animation_count))
            return fail(WPT_ERR_INVALID_ARGUMENT, "sphere refers to an animation outside the array");
    }
    for (uint32_t i = 0; i < d->hotspot_count; i++) {
        const wpt_hotspot& h = d->hotspots[i];
        if (h.kind > WPT_HOTSPOT_SPHERE)
            return fail(WPT_ERR_UNSUPPORTED, "hot spot kind is not known to the kernel");
        if (h.prim >= (h.kind == WPT_HOTSPOT_SPHERE ? d->sphere_count : d->tri_count))
            return fail(WPT_ERR_INVALID_ARGUMENT, "hot spot references a primitive outside the array");
    }
    if (d->animation_count > 0 && (!d->animations || (d->keyframe_count > 0 && !d->keyframes)))
        return fail(WPT_ERR_INVALID_ARGUMENT, "animation arrays are NULL");
    for (uint32_t i = 0; i < d->animation_count; i++) {
        const wpt_animation& a = d->animations[i];
        if (uint64_t(a.first_keyframe) + a.keyframe_count > d->keyframe_count)
            return fail(WPT_ERR_INVALID_ARGUMENT, "animation refers to key frames outside the array");
        for (uint32_t k = 1; k < a.keyframe_count; k++)
            if (!(d->keyframes[a.first_keyframe + k - 1].t < d->keyframes[a.first_keyframe + k].t))
                return fail(WPT_ERR_INVALID_ARGUMENT, "key frames must be sorted by ascending time");
    }
    for (uint32_t i = 0; i < d->instance_count; i++) {
        const wpt_instance& inst = d->instances[i];
        if (inst.animation >= int32_t(d->animation_count) || ((inst.flags & WPT_TRI_ANIMATE) && inst.animation < 0))
            return fail(WPT_ERR_INVALID_ARGUMENT, "mesh instance refers to an animation outside the array");
    }
    for (uint32_t i = 0; i < d->tri_count; i++) {
        const wpt_tri_geom& g = d->tri_geom[i];
        if ((g.flags & WPT_TRI_ANIMATE) && (g.instance >= d->instance_count || d->instances[g.instance].animation < 0))
            return fail(WPT_ERR_INVALID_ARGUMENT, "animated triangle without an animated instance");
    }
    for (uint32_t i = 0; i < d->hotspot_count; i++)
        if (d->hotspots[i].animation >= int32_t(d->animation_count))
            return fail(WPT_ERR_INVALID_ARGUMENT, "hot spot refers to an animation outside the array");
    if (d->envmap.type > WPT_ENV_CUBE)
        return fail(WPT_ERR_UNSUPPORTED, "environment map type is not known to the kernel");
    if (d->envmap.type == WPT_ENV_EQUIRECT && (d->envmap.tex < 0 || uint32_t(d->envmap.tex) >= d->texture_count))
        return fail(WPT_ERR_INVALID_ARGUMENT, "environment map references a texture outside the array");
    if (d->envmap.type == WPT_ENV_CUBE) {
        for (int k = 0; k < 6; k++)
            if (d->envmap.cube_tex[k] < 0 || uint32_t(d->envmap.cube_tex[k]) >= d->texture_count)
                return fail(WPT_ERR_INVALID_ARGUMENT, "environment cube map references a texture outside the array");
    }
    return WPT_OK;
}

} /* namespace */

extern "C" {

int wpt_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        g_error = std::string("hipGetDeviceCount: ") + hipGetErrorString(e);
        return 0;
    }
    return n;
}

wpt_status wpt_select_device(int device)
{
    HIP_TRY(hipSetDevice(device));
    return WPT_OK;
}

wpt_status wpt_current_device(int* device)
{
    if (!device)
        return fail(WPT_ERR_INVALID_ARGUMENT, "device is NULL");
    HIP_TRY(hipGetDevice(device));
    return WPT_OK;
}

wpt_status wpt_scene_upload(const wpt_scene_desc* desc, wpt_scene** out_scene)
{
    if (!out_scene)
        return fail(WPT_ERR_INVALID_ARGUMENT, "out_scene is NULL");
    *out_scene = nullptr;
    wpt_status st = validate(desc);
    if (st != WPT_OK)
        return st;
    if (wpt_device_count() <= 0)
        return fail(WPT_ERR_NO_DEVICE, "no HIP device is available; the path tracer has no CPU fallback");
    wpt_scene* s = new wpt_scene;
    HIP_TRY(hipGetDevice(&s->device));
    s->features = sceneFeatures(desc);
    s->nodeCount = desc->node_count;
    s->triCount = desc->tri_count;
    memset(&s->view, 0, sizeof(s->view));
    const float4* nodes = nullptr;
    const float4* geom = nullptr;
    const float4* attr = nullptr;
    std::vector<uint32_t> triNew; /* triangle index of the caller -> index on the device */
#define UP(call)                  \
    do {                          \
        st = (call);              \
        if (st != WPT_OK) {       \
            wpt_scene_free(s);    \
            return st;            \
        }                         \
    } while (0)
    {
        /* Device node form.  The reference pops a stack to find the next node after a subtree
         * (bvh.hpp:296,305); in depth-first order that node is the first one behind the subtree, so
         * its index is stored per node ("skip"), an inner node also carries the index of its first
         * child, and the kernel needs neither a stack nor any particular storage order.
         *
         * Storage order.  Every ray starts at the root, so the top of the tree is what all waves
         * of an XCD keep fetching; in depth-first order those nodes lie scattered over the whole
         * array (the right child of the root is half the array away), each dragging a 128-byte
         * line of rarely visited neighbours into the XCD's 4 MiB L2.  For trees larger than an L2
         * the nodes of the top levels are therefore stored first, level by level (WPT_TOP_NODES
         * nodes, 2 MiB by default, contiguous and dense), and the subtrees below them after that,
         * each depth-first as before (a walk that descends to a first child then reads the next
         * 32 bytes).  The visiting order is the tree's, not the array's: results do not change. */
        const uint32_t n = desc->node_count;
        if (n > NODE_INDEX_MASK)
            return fail(WPT_ERR_UNSUPPORTED, "more than 2^30 - 1 BVH nodes");
        /* Storage order of the triangles: that of their leaves in the tree's depth-first order, so that the leaves of a subtree --
         * which a ray tests one after the other, and neighbouring rays test too -- read neighbouring 48-byte records (a 128-byte
         * line holds the triangles of two or three sibling leaves) instead of wherever the meshes' own order put them.  Triangle
         * indices are identities only (leaf -> record, hot spot -> record, the candidate a light ray must end on): no value
         * depends on them.  wpt_set_walk(WPT_WALK_TRIANGLES_AS_GIVEN) keeps the caller's order (measurements). */
        triNew.assign(desc->tri_count, 0xffffffffu);
        {
            uint32_t next = 0;
            if (!(g_walk & WPT_WALK_TRIANGLES_AS_GIVEN))
                for (uint32_t i = 0; i < n; i++)
                    if (desc->nodes[i].kind == WPT_NODE_TRIANGLE && triNew[desc->nodes[i].link] == 0xffffffffu)
                        triNew[desc->nodes[i].link] = next++;
            for (uint32_t t = 0; t < desc->tri_count; t++) /* triangles no leaf refers to (or all, in the caller's order) */
                if (triNew[t] == 0xffffffffu)
                    triNew[t] = next++;
        }
        /* end[i] = first depth-first index behind the subtree of node i (validated above) */
        std::vector<uint32_t> end(n);
        for (uint32_t i = n; i-- > 0;)
            end[i] = desc->nodes[i].kind == WPT_NODE_INNER ? end[desc->nodes[i].link] : i + 1;
        const uint64_t slotCount = n;
        std::vector<uint32_t> place(size_t(n) + 1); /* depth-first index -> storage index; place[n] ends the walk */
        place[n] = n;
        uint32_t topNodes = g_topNodes;
        if (n <= topNodes) /* the whole tree is no larger than the part that would go in front: nothing to gain */
            topNodes = 0;
        {
            uint32_t cursor = 0;
            std::vector<uint32_t> level, next;
            if (topNodes > 0) {
                level.push_back(0);
                while (!level.empty() && cursor + level.size() <= topNodes) {
                    next.clear();
                    for (uint32_t i : level) {
                        place[i] = cursor;
                        cursor += 1;
                        if (desc->nodes[i].kind == WPT_NODE_INNER) {
                            next.push_back(i + 1);
                            next.push_back(desc->nodes[i].link);
                        }
                    }
                    level.swap(next);
                }
            } else {
                level.push_back(0);
            }
            /* the subtrees that did not make it into the top part, depth-first each, in depth-first order of their roots
             * (blocks of 2 - 6 levels stored level by level instead were measured: 52.3 - 51.8 against 52.7 Msamples/s on the
             * 10 M triangle scene, no difference on the Sponza-class one; locality of the nodes is not what that scene lacks) */
            std::sort(level.begin(), level.end());
            for (uint32_t root : level)
                for (uint32_t i = root; i < end[root]; i++) {
                    place[i] = cursor;
                    cursor += 1;
                }
            if (cursor != slotCount) {
                wpt_scene_free(s);
                return fail(WPT_ERR_INVALID_ARGUMENT, "BVH conversion: the links do not reach every node exactly once");
            }
        }
        /* one node of padding: kernels that fetch aligned pairs of nodes read the whole last pair */
        std::vector<float4> dev(size_t(slotCount) * 2 + 2, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
        s->view.nodeCount = n;
        for (uint32_t i = 0; i < n; i++) {
            const wpt_bvh_node& nd = desc->nodes[i];
            const uint32_t skip = place[end[i]];
            const uint32_t word = nd.kind == WPT_NODE_INNER ? (NODE_CHILD | place[i + 1]) : nd.kind == WPT_NODE_TRIANGLE ? triNew[nd.link]
                : nd.kind == WPT_NODE_SPHERE ? (PRIM_SPHERE | nd.link) : (NODE_CHILD | skip);
            float sk, wd;
            memcpy(&sk, &skip, 4);
            memcpy(&wd, &word, 4);
            dev[2 * size_t(place[i])] = make_float4(nd.lo[0], nd.hi[0], nd.lo[1], nd.lo[2]); /* nodeLo / nodeHi (wpt_device.h) */
            dev[2 * size_t(place[i]) + 1] = make_float4(nd.hi[1], nd.hi[2], sk, wd);
            for (int a = 0; a < 3; a++)
                if (nd.lo[a] != nd.lo[a] || nd.hi[a] != nd.hi[a])
                    s->view.boxesMayBeNan = 1u;
        }
        UP(uploadArray(s, dev.data(), dev.size(), &nodes));
        if (g_walk & WPT_WALK_WIDE) {
            /* The wide form (wpt_pathtrace.inc.h): the binary tree collapsed by one level.  Wide nodes are made for the root and
             * for every inner node that is an entry of a wide node, in depth-first order (a wide node's first inner entry follows
             * it).  The walk's argument needs finite boxes and every child's box within its parent's; its stack needs the tree's
             * worst case to fit.  A tree that fails any of the three has no wide form and is walked as it is. */
            bool ok = true;
            for (uint32_t i = 0; i < n && ok; i++) {
                const wpt_bvh_node& nd = desc->nodes[i];
                for (int a = 0; a < 3; a++)
                    ok = ok && std::isfinite(nd.lo[a]) && std::isfinite(nd.hi[a]);
                if (nd.kind == WPT_NODE_INNER) {
                    const uint32_t child[2] = { i + 1, nd.link };
                    for (int k = 0; k < 2; k++)
                        for (int a = 0; a < 3; a++)
                            ok = ok && desc->nodes[child[k]].lo[a] >= nd.lo[a] && desc->nodes[child[k]].hi[a] <= nd.hi[a];
                }
            }
            std::vector<float4> wide;
            std::vector<uint32_t> made;        /* binary node of each wide node, in order of creation */
            std::vector<uint32_t> entries;     /* 4 per wide node: binary nodes, 0xffffffff = none */
            if (ok) {
                std::vector<uint32_t> todo(1, 0u); /* depth first: a stack of binary nodes to make wide nodes for */
                while (!todo.empty()) {
                    const uint32_t x = todo.back();
                    todo.pop_back();
                    made.push_back(x);
                    uint32_t entry[4] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu };
                    int count = 0;
                    if (desc->nodes[x].kind == WPT_NODE_INNER) {
                        const uint32_t child[2] = { x + 1, desc->nodes[x].link };
                        for (int k = 0; k < 2; k++) {
                            if (desc->nodes[child[k]].kind == WPT_NODE_INNER) {
                                entry[count++] = child[k] + 1;
                                entry[count++] = desc->nodes[child[k]].link;
                            } else {
                                entry[count++] = child[k];
                            }
                        }
                    } else {
                        entry[count++] = x; /* a tree of one leaf */
                    }
                    for (int k = 0; k < 4; k++)
                        entries.push_back(entry[k]);
                    for (int k = count - 1; k >= 0; k--) /* the first inner entry is made next */
                        if (desc->nodes[entry[k]].kind == WPT_NODE_INNER && entry[k] != x)
                            todo.push_back(entry[k]);
                }
                ok = made.size() <= NODE_INDEX_MASK;
            }
            if (ok) {
                /* wide index of every binary node that has one; creation order is a pre-order, so a reverse pass sees children first */
                std::vector<uint32_t> wideOf(n, 0xffffffffu);
                for (size_t w = 0; w < made.size(); w++)
                    wideOf[made[w]] = uint32_t(w);
                std::vector<uint32_t> depth(made.size(), 0u); /* entries that can wait on the stack while the walk is below this wide node */
                for (size_t w = made.size(); w-- > 0;) {
                    int count = 0;
                    while (count < 4 && entries[4 * w + count] != 0xffffffffu)
                        count++;
                    uint32_t worst = 0;
                    for (int k = 0; k < count; k++) {
                        const uint32_t e = entries[4 * w + k];
                        const uint32_t below = (desc->nodes[e].kind == WPT_NODE_INNER && e != made[w]) ? depth[wideOf[e]] : 0u;
                        worst = std::max(worst, uint32_t(count - 1 - k) + below);
                    }
                    depth[w] = worst;
                }
                ok = depth[0] <= WIDE_STACK;
                wide.resize(made.size() * 8, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
                for (size_t w = 0; w < made.size() && ok; w++) {
                    float q[8][4];
                    uint32_t ref[4] = { WIDE_NONE, WIDE_NONE, WIDE_NONE, WIDE_NONE };
                    for (int r = 0; r < 8; r++)
                        for (int k = 0; k < 4; k++)
                            q[r][k] = 0.0f;
                    for (int k = 0; k < 4; k++) {
                        const uint32_t e = entries[4 * w + k];
                        if (e == 0xffffffffu)
                            continue;
                        const wpt_bvh_node& nd = desc->nodes[e];
                        for (int a = 0; a < 3; a++) {
                            q[a][k] = nd.lo[a];
                            q[3 + a][k] = nd.hi[a];
                        }
                        if (nd.kind == WPT_NODE_INNER)
                            ref[k] = NODE_CHILD | wideOf[e];
                        else if (nd.kind == WPT_NODE_TRIANGLE)
                            ref[k] = triNew[nd.link];
                        else if (nd.kind == WPT_NODE_SPHERE)
                            ref[k] = PRIM_SPHERE | nd.link;
                    }
                    memcpy(q[6], ref, 16);
                    for (int r = 0; r < 8; r++)
                        wide[8 * w + r] = make_float4(q[r][0], q[r][1], q[r][2], q[r][3]);
                }
            }
            if (ok)
                UP(uploadArray(s, wide.data(), wide.size(), &s->view.wideNodes));
        }
    }
    {
        std::vector<wpt_tri_geom> g(desc->tri_count);
        for (uint32_t t = 0; t < desc->tri_count; t++)
            g[triNew[t]] = desc->tri_geom[t];
        UP(uploadArray(s, reinterpret_cast<const float4*>(g.data()), size_t(desc->tri_count) * 3, &geom));
    }
    {
        std::vector<wpt_tri_attr> a(desc->tri_count);
        for (uint32_t t = 0; t < desc->tri_count; t++)
            a[triNew[t]] = desc->tri_attr[t];
        UP(uploadArray(s, reinterpret_cast<const float4*>(a.data()), size_t(desc->tri_count) * 6, &attr));
    }
    s->view.nodes = nodes;
    s->view.triGeom = geom;
    s->view.triAttr = attr;
    UP(uploadArray(s, desc->instances, desc->instance_count, &s->view.instances));
    UP(uploadArray(s, desc->materials, desc->material_count, &s->view.materials));
    s->view.materialCount = desc->material_count;
    {
        /* Image textures are decoded once, here, into one pool of RGBA float4 texels (16 bytes
         * per texel whatever the file format was: HBM is large, instructions per lookup are
         * not); the device copies of the texture records index that pool. */
        std::vector<wpt_texture> devTex(desc->textures, desc->textures + desc->texture_count);
        size_t texelCount = 0;
        for (wpt_texture& t : devTex) {
            if (t.type == WPT_TEX_IMAGE) {
                t.texel_offset = texelCount;
                texelCount += size_t(t.width) * t.height;
            }
        }
        UP(uploadArray(s, devTex.data(), devTex.size(), &s->view.textures));
        void* pool = nullptr;
        hipError_t e = hipMalloc(&pool, texelCount > 0 ? texelCount * sizeof(float4) : 16);
        if (e != hipSuccess) {
            wpt_scene_free(s);
            return fail(e == hipErrorOutOfMemory ? WPT_ERR_OUT_OF_MEMORY : WPT_ERR_HIP, std::string("hipMalloc for the decoded texel pool: ") + hipGetErrorString(e));
        }
        s->allocations.push_back(pool);
        s->view.texels4 = static_cast<const float4*>(pool);
        if (texelCount > 0) {
            uint8_t* raw = nullptr;
            e = hipMalloc(reinterpret_cast<void**>(&raw), desc->texel_bytes > 0 ? desc->texel_bytes : 16);
            if (e == hipSuccess)
                e = hipMemcpy(raw, desc->texels, desc->texel_bytes, hipMemcpyHostToDevice);
            for (uint32_t i = 0; e == hipSuccess && i < desc->texture_count; i++) {
                const wpt_texture& t = desc->textures[i];
                if (t.type != WPT_TEX_IMAGE)
                    continue;
                const size_t n = size_t(t.width) * t.height;
                hipLaunchKernelGGL(wpt_expand_texels_kernel, dim3(uint32_t((n + 255) / 256)), dim3(256), 0, 0,
                        raw, t, static_cast<float4*>(pool) + devTex[i].texel_offset);
                e = hipGetLastError();
            }
            if (e == hipSuccess)
                e = hipDeviceSynchronize();
            (void)hipFree(raw);
            if (e != hipSuccess) {
                wpt_scene_free(s);
                return fail(e == hipErrorOutOfMemory ? WPT_ERR_OUT_OF_MEMORY : WPT_ERR_HIP, std::string("texel decode: ") + hipGetErrorString(e));
            }
        }
    }
    {
        std::vector<wpt_hotspot> h(desc->hotspots, desc->hotspots + desc->hotspot_count);
        for (wpt_hotspot& hs : h)
            if (hs.kind != WPT_HOTSPOT_SPHERE)
                hs.prim = triNew[hs.prim];
        UP(uploadArray(s, h.data(), h.size(), &s->view.hotspots));
    }
    UP(uploadArray(s, desc->spheres, desc->sphere_count, &s->view.spheres));
    UP(uploadArray(s, desc->rgl_brdfs, desc->rgl_count, &s->view.rglBrdfs));
    {
        /* The measured BRDFs' pool, and behind it one interleaved table per BRDF whose colour and luminance warps share their
         * grids (wpt_rgl.h, rglColourInterleaved): red, green, blue and luminance of a grid point side by side, so that the up to
         * 128 look-ups an evaluation makes into those two warps come from 8 cache lines instead of 32.  The values are the
         * pool's own; which copy a look-up reads changes no bit. */
        std::vector<float> pool(desc->rgl_data, desc->rgl_data + desc->rgl_data_count);
        std::vector<uint32_t> rgbl(desc->rgl_count, WPT_RGL_NONE);
        for (uint32_t i = 0; i < desc->rgl_count; i++) {
            const wpt_rgl_brdf& b = desc->rgl_brdfs[i];
            if (!wptrgl::rglInterleavable(b))
                continue;
            const size_t size = size_t(b.rgb.size_x) * b.rgb.size_y;
            const size_t slices = size_t(b.luminance.param_size[0]) * b.luminance.param_size[1];
            const size_t at = (pool.size() + 3) & ~size_t(3); /* 16-byte records */
            if (at + slices * size * 4 > 0xfffffff0ull)
                continue;
            pool.resize(at + slices * size * 4);
            for (size_t sl = 0; sl < slices; sl++)
                for (size_t e = 0; e < size; e++) {
                    float* t = pool.data() + at + (sl * size + e) * 4;
                    for (size_t c = 0; c < 3; c++)
                        t[c] = desc->rgl_data[b.rgb.data + (sl * 3 + c) * size + e];
                    t[3] = desc->rgl_data[b.luminance.data + sl * size + e];
                }
            rgbl[i] = uint32_t(at);
        }
        UP(uploadArray(s, pool.data(), pool.size(), &s->view.rglData));
        UP(uploadArray(s, rgbl.data(), rgbl.size(), &s->view.rglRgbl));
    }
    UP(uploadArray(s, desc->animations, desc->animation_count, &s->view.animations));
    UP(uploadArray(s, desc->keyframes, desc->keyframe_count, &s->view.keyframes));
    s->animationCount = desc->animation_count;
    s->view.sphereCount = desc->sphere_count;
    for (int k = 0; k < 6; k++)
        s->view.envCube[k] = desc->envmap.cube_tex[k];
    {
        hipDeviceProp_t prop;
        s->cuCount = hipGetDeviceProperties(&prop, s->device) == hipSuccess ? prop.multiProcessorCount : 256;
    }
    s->view.triCount = desc->tri_count;
    s->view.hotspotCount = desc->hotspot_count;
    s->view.invHotspotCount = desc->hotspot_count ? 1.0f / float(desc->hotspot_count) : 0.0f;
    if (desc->hotspot_count > 0) {
        float4* face = nullptr;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&face), size_t(desc->hotspot_count) * sizeof(float4));
        if (e == hipSuccess) {
            s->allocations.push_back(face);
            hipLaunchKernelGGL(wpt_hotspot_face_kernel, dim3((desc->hotspot_count + 255) / 256), dim3(256), 0, 0, s->view, face);
            e = hipDeviceSynchronize();
        }
        if (e != hipSuccess) {
            wpt_scene_free(s);
            return fail(e == hipErrorOutOfMemory ? WPT_ERR_OUT_OF_MEMORY : WPT_ERR_HIP, std::string("hot spot faces: ") + hipGetErrorString(e));
        }
        s->view.hotspotFace = face;
    }
    s->view.envType = desc->envmap.type;
    s->view.envCompat = desc->envmap.compat;
    s->view.envTex = desc->envmap.tex;
    s->view.envN = 0;
    s->view.envLog2N = -1;
    if (desc->envmap.type != WPT_ENV_NONE && desc->envmap.N > 0) {
        const int N = desc->envmap.N;
        const size_t bins = size_t(N) * N;
        if (desc->envmap.M && desc->envmap.Ms && desc->envmap.Mcs) {
            s->envM.assign(desc->envmap.M, desc->envmap.M + bins);
            s->envMs.assign(desc->envmap.Ms, desc->envmap.Ms + bins);
            s->envMcs.assign(desc->envmap.Mcs, desc->envmap.Mcs + bins);
        } else {
            /* EnvironmentMap::initializeImportanceSampling (envmap.hpp:121-158): the per-bin
             * importance comes from the device's own L(); sum, sort and prefix sum run on the
             * host in the reference's sequential order */
            float* dImp = nullptr;
            hipError_t e = hipMalloc(reinterpret_cast<void**>(&dImp), bins * sizeof(float));
            if (e != hipSuccess) {
                wpt_scene_free(s);
                return fail(WPT_ERR_OUT_OF_MEMORY, "hipMalloc for the importance map failed");
            }
            hipLaunchKernelGGL(wpt_env_importance_kernel, dim3((bins + 255) / 256), dim3(256), 0, 0, s->view, N, dImp);
            s->envM.resize(bins);
            e = hipMemcpy(s->envM.data(), dImp, bins * sizeof(float), hipMemcpyDeviceToHost);
            (void)hipFree(dImp);
            if (e != hipSuccess) {
                wpt_scene_free(s);
                return fail(WPT_ERR_HIP, std::string("importance map: ") + hipGetErrorString(e));
            }
            float total = 0.0f;
            for (size_t i = 0; i < bins; i++)
                total += s->envM[i];
            for (size_t i = 0; i < bins; i++)
                s->envM[i] /= total;
            s->envMs.resize(bins);
            for (size_t i = 0; i < bins; i++)
                s->envMs[i] = int32_t(i);
            const std::vector<float>& M = s->envM;
            std::sort(s->envMs.begin(), s->envMs.end(), [&M](unsigned int i, unsigned int j) { return M[i] > M[j]; });
            s->envMcs.resize(bins);
            float sum = 0.0f;
            for (size_t i = 0; i < bins; i++) {
                sum += M[s->envMs[i]];
                s->envMcs[i] = sum;
            }
        }
        UP(uploadArray(s, s->envM.data(), bins, &s->view.envM));
        UP(uploadArray(s, s->envMs.data(), bins, &s->view.envMs));
        UP(uploadArray(s, s->envMcs.data(), bins, &s->view.envMcs));
        {
            /* start table for the sampling search (envD in wpt_device.h); only for a non-decreasing
             * cumulative table, which is what the construction gives -- a caller's own table that is
             * not gets the plain bisection */
            bool monotone = true;
            for (size_t i = 1; i < bins && monotone; i++)
                monotone = !(s->envMcs[i] < s->envMcs[i - 1]);
            if (monotone) {
                const uint32_t K = 65536;
                std::vector<int32_t> lut(K + 1);
                size_t i = 0;
                for (uint32_t k = 0; k < K; k++) {
                    const float t = float(k) / float(K);
                    while (i < bins && s->envMcs[i] < t)
                        i++;
                    lut[k] = int32_t(i < bins ? i : bins - 1);
                }
                lut[K] = int32_t(bins - 1);
                UP(uploadArray(s, lut.data(), lut.size(), &s->view.envLut));
                s->view.envLutSize = K;
            }
        }
        s->view.envN = N;
        s->view.envLog2N = -1;
        if (N > 0 && (N & (N - 1)) == 0)
            for (int b = 0; b < 31; b++)
                if ((1 << b) == N)
                    s->view.envLog2N = b;
    }
#undef UP
    *out_scene = s;
    return WPT_OK;
}

void wpt_scene_free(wpt_scene* scene)
{
    if (!scene)
        return;
    for (void* p : scene->allocations)
        (void)hipFree(p);
    delete scene;
}

/* copies the importance tables of an uploaded scene back (tests compare them with the oracle's) */
wpt_status wpt_scene_get_envmap_tables(const wpt_scene* scene, float* M, int32_t* Ms, float* Mcs)
{
    if (!scene || scene->view.envN <= 0)
        return fail(WPT_ERR_INVALID_ARGUMENT, "scene has no importance tables");
    size_t bins = scene->envM.size();
    memcpy(M, scene->envM.data(), bins * sizeof(float));
    memcpy(Ms, scene->envMs.data(), bins * sizeof(int32_t));
    memcpy(Mcs, scene->envMcs.data(), bins * sizeof(float));
    return WPT_OK;
}

/* one launch: a block of consecutive pixels (band_stride == 0) or interleaved bands of band_pixels pixels */
static wpt_status renderLaunch(wpt_scene* scene, const wpt_camera* camera, const wpt_params* params,
        uint32_t width, uint32_t height, uint32_t samples_sqrt, uint32_t block_start, uint32_t block_size,
        uint32_t band_pixels, uint32_t band_first, uint32_t band_stride,
        float* frame_device, wpt_counters* counters_device, void* hip_stream)
{
    if (!scene || !camera || !params || !frame_device)
        return fail(WPT_ERR_INVALID_ARGUMENT, "NULL argument");
    if (width == 0 || height == 0 || samples_sqrt == 0 || samples_sqrt > 65535 || width > 65535 || height > 65535)
        return fail(WPT_ERR_INVALID_ARGUMENT, "width, height and samples_sqrt must lie in 1 .. 65535");
    if (uint64_t(width) * height > 0xffffffffull || uint64_t(block_start) + (band_stride ? 0u : block_size) > uint64_t(width) * height)
        return fail(WPT_ERR_INVALID_ARGUMENT, "pixel block lies outside the frame");
    if (block_size == 0)
        return WPT_OK;
    KernelArgs args;
    args.bandPixels = band_pixels;
    args.bandFirst = band_first;
    args.bandStride = band_stride;
    args.sv = scene->view;
    args.cam = *camera;
    args.par = *params;
    args.width = width;
    args.height = height;
    args.samplesSqrt = samples_sqrt;
    args.invWidth = 1.0f / float(width);
    args.invHeight = 1.0f / float(height);
    args.invSamplesSqrt = 1.0f / float(samples_sqrt);
    args.invSamples = 1.0f / float(samples_sqrt * samples_sqrt);
    args.blockStart = block_start;
    args.blockSize = block_size;
    args.frame = frame_device;
    args.counters = counters_device;
    args.schedStats = g_schedStats;
    args.fuse = (g_variant & 0x20u) ? 0u : 1u; /* variant bit 0x20: separate SHADE / NEE-END / NEW rounds (the older scheduler) */

    /* a wave covers an 8x8 pixel tile when the block consists of whole groups of 8 rows */
    args.tiled = (width % 8 == 0 && block_start % width == 0 && block_size % (8 * width) == 0
            && (band_stride == 0 || band_pixels % (8 * width) == 0)) ? 1u : 0u;
    uint32_t need = scene->features | ((camera->lens_radius > 0.0f || camera->distortion_type != WPT_DISTORTION_NONE
                || camera->surround_mode != WPT_SURROUND_OFF || camera->stereoscopic_distance > 0.0f) ? FEAT_LENS : 0u);
    if (camera->surround_mode > WPT_SURROUND_360)
        return fail(WPT_ERR_UNSUPPORTED, "camera surround mode is not known to the kernel");
    if (camera->distortion_type > WPT_DISTORTION_OPENCV)
        return fail(WPT_ERR_UNSUPPORTED, "lens distortion model is not known to the kernel");
    if (camera->animation >= int32_t(scene->animationCount))
        return fail(WPT_ERR_INVALID_ARGUMENT, "camera refers to an animation outside the scene's array");
    /* an exposure interval changes every path (each camera ray draws its time), moving instances need the time too */
    if (params->t0 != params->t1)
        need |= FEAT_ANIM;
    const bool anim = (need & FEAT_ANIM) != 0;
    dim3 grid((block_size + WG - 1) / WG);
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    const bool count = counters_device != nullptr;
    /* The walk of a light ray towards the environment ends at its first accepted hit (wpt_pathtrace.inc.h: the answer it is
     * traced for is known there).  Counting launches walk on as the reference does, so that their counters are the
     * reference's; measurements (wpt_set_walk): WPT_WALK_COUNT_PRODUCT makes them count what the product kernel walks,
     * WPT_WALK_FULL_SHADOW switches the short cut off everywhere. */
    args.shadowWalksEnd = (g_walk & WPT_WALK_FULL_SHADOW) ? 0u : (count ? ((g_walk & WPT_WALK_COUNT_PRODUCT) ? 1u : 0u) : 1u);
    const size_t ldsBytes = size_t(scene->nodeCount) * 32 + size_t(scene->triCount) * 48;
    /* scheduler defaults from sweeps on the Cornell box (scene in LDS, short walks) and on the
     * Sponza-class scene (deep tree in HBM: traversal dominates, so long blocks may run with fewer
     * lanes and leaf tests earlier) */
    const bool smallScene = ldsBytes <= LDS_SCENE_MAX_BYTES;
    /* clamped: with more than 8 eighths the traversal block would leave before doing anything */
    args.leaveEighths = g_leaveEighths ? (g_leaveEighths > 8u ? 8u : g_leaveEighths) : (smallScene ? 1u : 3u);
    args.heavyMin = g_heavyMin ? g_heavyMin : (smallScene ? 16u : 8u);
    args.leafBias = g_leafBias ? g_leafBias : (smallScene ? 16u : 32u);
    /* variant bits 2-3: 0 = default, 1 = no kind of material ever stands back, 2 / 3 = fewer than 3 / 12 lanes */
    static const uint32_t waitBelowChoices[4] = { 6u, 0u, 3u, 12u };
    args.waitBelow = waitBelowChoices[(g_variant >> 2) & 0x3u];
    /* kernel choice.  Low nibble of the variant word: 1 = keep the scene in HBM, 2 = all features. */
    const uint32_t force = g_variant & 0x3u;
    const bool basic = (need & ~FEAT_BASIC) == 0 && force != 2;
    const bool lds = smallScene && force != 1;
    const bool rgl = (need & FEAT_RGL) != 0; /* measured BRDFs have their own instantiation */
    /* Pixel pool: frames with more pixels than the device has lanes at once are handed out pixel by pixel (the launchers
     * decide); variant bit 0x10: never.  The counter is allocated and freed in stream order, so launches in flight on any
     * number of streams never share one. */
    /* Wavefront form (wpt_wavefront.inc.h): trace and shade as two kernels that hand rays through HBM.  Not for counting
     * launches and moving scenes (those instantiations exist for the single kernel only). */
    const bool wfExists = !count && !anim;
    /* The library's own choice (measured): launches of 2^21 lanes and more whose scene has measured BRDFs --
     * long shading that pays for being sorted by kind of material, and enough lanes to fill the trace and the shade kernel one
     * after the other (tools/wf_threshold_probe.py, 16 spp, single kernel / wavefront: 115.5 / 100.7 Msamples/s at 2^20 lanes,
     * 118.0 / 116.4 at 1.97 M, 113.6 / 126.9 at 4.1 M, 114.0 / 135.3 at 8.3 M).  Scenes whose time is the walk stay with the
     * single kernel: the wavefront trace meets the same wall of the memory system (Sponza-class 120.6 against 158, 10 M
     * triangles 38 against 63 in round 3) and pays for the rays' way through HBM on top. */
    const bool wfAuto = rgl && block_size >= (1u << 21);
    if (wfExists && (g_wfMode == 1u || (g_wfMode == 0u && wfAuto))) {
        args.pool = nullptr;
        args.cuCount = uint32_t(scene->cuCount);
        args.materialsInLds = 0;
        args.rowStop = samples_sqrt;
        args.carry = nullptr;
        args.cost = nullptr;
        args.order = nullptr;
        args.orderCount = nullptr;
        const wptk::WfLaunchers& kernels = rgl ? wptk::wfFullRgl() : (basic ? wptk::wfBasic() : wptk::wfFull());
        uint32_t launches = 0;
        const hipError_t e = wptk::renderWavefront(args, kernels, g_wfConfig, stream, &launches);
        if (e == hipSuccess) {
            g_lastPasses.store(launches, std::memory_order_relaxed);
            g_kernelName.store("wf_trace + wf_shade", std::memory_order_relaxed);
            return WPT_OK;
        }
        /* The library's own choice must not fail where the single kernel would not: without the memory for the records
         * (256 B per lane) the frame is rendered by the single kernel below.  A forced wavefront render reports the error. */
        if (g_wfMode == 1u || e != hipErrorOutOfMemory)
            return fail(e == hipErrorOutOfMemory ? WPT_ERR_OUT_OF_MEMORY : WPT_ERR_HIP, std::string("wavefront render: ") + hipGetErrorString(e));
        (void)hipGetLastError();
    }
    uint32_t* pool = nullptr;
    const bool pooled = !count && !(g_variant & 0x10u) && block_size < 0x80000000u && grid.x > uint32_t(scene->cuCount);
    if (pooled && hipMallocAsync(reinterpret_cast<void**>(&pool), sizeof(uint32_t), stream) != hipSuccess) {
        (void)hipGetLastError();
        pool = nullptr;
    }
    args.pool = pool;
    args.cuCount = uint32_t(scene->cuCount);
    /* the material records join the scene in LDS where a quarter of a compute unit's 160 KiB holds a workgroup with them */
    args.materialsInLds = (!(g_variant & 0x80u) && COLD_BYTES + ldsBytes + size_t(scene->view.materialCount) * sizeof(wpt_material) <= LDS_BYTES_PER_WORKGROUP_AT_FOUR) ? 1u : 0u;
    args.rowStop = samples_sqrt;
    args.carry = nullptr;
    args.cost = nullptr;
    args.order = nullptr;
    args.orderCount = nullptr;
    /* the wide walk where the scene has that form (wpt_set_walk before the upload): product launches of the kernels that fetch
     * the scene from HBM; counting launches and moving scenes walk the binary tree */
    const bool wide = scene->view.wideNodes != nullptr && !count && !anim && !(basic && lds); /* (the kernel with the scene in LDS walks the binary tree) */
    g_kernelName.store(wide ? "wpt_pathtrace, wide walk" : nullptr, std::memory_order_relaxed);
    auto launch = [&](const wptk::KernelArgs& a) {
        if (anim) {
            /* its own instantiation, like the measured BRDFs */
            if (need & FEAT_RGL) {
                if (count)
                    launchFullRglAnimCount(a, grid, stream);
                else
                    launchFullRglAnim(a, grid, stream);
            } else if (count) {
                launchFullAnimCount(a, grid, stream);
            } else {
                launchFullAnim(a, grid, stream);
            }
        } else if (count) {
            if (basic)
                launchBasicCount(a, grid, stream);
            else if (rgl)
                launchFullRglCount(a, grid, stream);
            else
                launchFullCount(a, grid, stream);
        } else if (rgl) {
            if (wide)
                launchFullRglWide(a, grid, stream);
            else
                launchFullRgl(a, grid, stream);
        } else {
            if (basic && lds)
                launchBasicLds(a, grid, ldsBytes + (a.materialsInLds ? size_t(scene->view.materialCount) * sizeof(wpt_material) : 0), stream);
            else if (wide) /* also for the basic feature set: the wide walk exists in the all-features instantiations */
                launchFullWide(a, grid, stream);
            else if (basic)
                launchBasic(a, grid, stream);
            else
                launchFull(a, grid, stream);
        }
    };
    /* Two passes for the kernels that fetch the scene from HBM (variant bit 0x40: never): with 2 to 64 pixels per lane the
     * end of a launch, when lanes run out of pixels one by one, is a noticeable part of it.  The first pass renders one row
     * of strata of every pixel and times it, the second renders the rest, the 8x8 tiles that took longest first
     * (Sponza-class frame +2.9 %, 10 M triangles +1.8 %).  Not for the scene in LDS: there the launch is bound by how well
     * a wave's lanes keep in step, any order but the frame's own costs that more than the shorter end gives (Cornell
     * 938 against 954). */
    const uint64_t lanesAtOnce = uint64_t(scene->cuCount) * 4u * WG;
    const bool sceneInLds = basic && lds && !anim && !rgl;
    const bool twoPasses = pool != nullptr && !(g_variant & 0x40u) && !sceneInLds && samples_sqrt >= 8
            && uint64_t(block_size) >= 2u * lanesAtOnce && uint64_t(block_size) <= 64u * lanesAtOnce;
    float4* carry = nullptr;
    uint32_t *cost = nullptr, *order = nullptr, *work = nullptr;
    bool passesDone = false;
    if (twoPasses) {
        const size_t pixels = size_t(width) * height;
        if (hipMallocAsync(reinterpret_cast<void**>(&carry), pixels * 2 * sizeof(float4), stream) == hipSuccess
                && hipMallocAsync(reinterpret_cast<void**>(&cost), pixels * sizeof(uint32_t), stream) == hipSuccess
                && hipMallocAsync(reinterpret_cast<void**>(&order), size_t(block_size) * sizeof(uint32_t), stream) == hipSuccess
                && hipMallocAsync(reinterpret_cast<void**>(&work), (3 * wptk::ORDER_BUCKETS + 1) * sizeof(uint32_t), stream) == hipSuccess) {
            wptk::KernelArgs first = args;
            first.rowStop = 1;
            first.carry = carry;
            first.cost = cost;
            launch(first);
            wptk::launchOrderBuild(first, order, work, stream);
            wptk::KernelArgs second = args;
            second.carry = carry;
            second.order = order;
            second.orderCount = work + 3 * wptk::ORDER_BUCKETS;
            launch(second);
            passesDone = true;
        } else {
            (void)hipGetLastError();
        }
    }
    if (!passesDone)
        launch(args);
    g_lastPasses.store(passesDone ? 2u : 1u, std::memory_order_relaxed);
    const hipError_t launched = hipGetLastError();
    for (void* p : { static_cast<void*>(pool), static_cast<void*>(carry), static_cast<void*>(cost), static_cast<void*>(order), static_cast<void*>(work) })
        if (p)
            (void)hipFreeAsync(p, stream);
    HIP_TRY(launched);
    return WPT_OK;
}

wpt_status wpt_render_block_device(wpt_scene* scene, const wpt_camera* camera, const wpt_params* params,
        uint32_t width, uint32_t height, uint32_t samples_sqrt, uint32_t block_start, uint32_t block_size,
        float* frame_device, wpt_counters* counters_device, void* hip_stream)
{
    return renderLaunch(scene, camera, params, width, height, samples_sqrt, block_start, block_size, 0, 0, 0, frame_device, counters_device,
            hip_stream);
}

wpt_status wpt_render_bands_device(wpt_scene* scene, const wpt_camera* camera, const wpt_params* params,
        uint32_t width, uint32_t height, uint32_t samples_sqrt, uint32_t band_rows, uint32_t first_band, uint32_t band_stride,
        float* frame_device, wpt_counters* counters_device, void* hip_stream)
{
    if (band_rows == 0 || band_stride == 0 || first_band >= band_stride || width == 0 || height == 0)
        return fail(WPT_ERR_INVALID_ARGUMENT, "bands need band_rows > 0 and first_band < band_stride");
    const uint64_t bandPixels = uint64_t(band_rows) * width;
    const uint64_t bands = (uint64_t(height) + band_rows - 1) / band_rows;
    const uint64_t mine = first_band < bands ? (bands - first_band + band_stride - 1) / band_stride : 0;
    if (bandPixels > 0xffffffffull || mine * bandPixels > 0xffffffffull)
        return fail(WPT_ERR_INVALID_ARGUMENT, "bands too large");
    return renderLaunch(scene, camera, params, width, height, samples_sqrt, 0, uint32_t(mine * bandPixels), uint32_t(bandPixels), first_band,
            band_stride, frame_device, counters_device, hip_stream);
}

wpt_status wpt_render_bands(wpt_scene* scene, const wpt_camera* camera, const wpt_params* params,
        uint32_t width, uint32_t height, uint32_t samples_sqrt, uint32_t band_rows, uint32_t first_band, uint32_t band_stride,
        float* frame_host)
{
    if (!frame_host)
        return fail(WPT_ERR_INVALID_ARGUMENT, "frame_host is NULL");
    if (width == 0 || height == 0 || band_rows == 0 || band_stride == 0)
        return fail(WPT_ERR_INVALID_ARGUMENT, "bands need a frame, band_rows > 0 and band_stride > 0");
    float* dFrame = nullptr;
    const size_t rowBytes = size_t(width) * 3 * sizeof(float);
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dFrame), rowBytes * height));
    wpt_status st = wpt_render_bands_device(scene, camera, params, width, height, samples_sqrt, band_rows, first_band, band_stride, dFrame, nullptr,
            nullptr);
    if (st == WPT_OK)
        st = wpt_scene_check(scene);
    for (uint64_t band = first_band; st == WPT_OK && band * band_rows < height; band += band_stride) {
        const size_t row0 = size_t(band) * band_rows;
        const size_t rows = row0 + band_rows <= height ? band_rows : height - row0;
        hipError_t e = hipMemcpy(reinterpret_cast<char*>(frame_host) + row0 * rowBytes, reinterpret_cast<char*>(dFrame) + row0 * rowBytes, rows * rowBytes,
                hipMemcpyDeviceToHost);
        if (e != hipSuccess)
            st = fail(WPT_ERR_HIP, std::string("hipMemcpy: ") + hipGetErrorString(e));
    }
    (void)hipFree(dFrame);
    return st;
}

wpt_status wpt_render_block(wpt_scene* scene, const wpt_camera* camera, const wpt_params* params, uint32_t width,
        uint32_t height, uint32_t samples_sqrt, uint32_t block_start, uint32_t block_size, float* block_rgb)
{
    if (!block_rgb)
        return fail(WPT_ERR_INVALID_ARGUMENT, "block_rgb is NULL");
    if (block_size == 0)
        return WPT_OK;
    /* a frame-sized address space would waste memory for small blocks: allocate the block only
     * and bias the frame pointer so that pixel `block_start` lands at offset 0 */
    float* dBlock = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dBlock), size_t(block_size) * 3 * sizeof(float)));
    float* biased = dBlock - size_t(block_start) * 3;
    wpt_status st = wpt_render_block_device(scene, camera, params, width, height, samples_sqrt, block_start, block_size,
            biased, nullptr, nullptr);
    if (st == WPT_OK)
        st = wpt_scene_check(scene);
    if (st == WPT_OK) {
        hipError_t e = hipMemcpy(block_rgb, dBlock, size_t(block_size) * 3 * sizeof(float), hipMemcpyDeviceToHost);
        if (e != hipSuccess)
            st = fail(WPT_ERR_HIP, std::string("hipMemcpy: ") + hipGetErrorString(e));
    }
    (void)hipFree(dBlock);
    return st;
}

wpt_status wpt_ground_truth_device(wpt_scene* scene, const wpt_camera* camera, const wpt_camera* camera_prev,
        const wpt_camera* camera_next, const float times[3], const wpt_params* params, uint32_t width, uint32_t height,
        void* const arrays_device[WPT_GT_ARRAY_COUNT], void* hip_stream)
{
    if (!scene || !camera || !params || !arrays_device)
        return fail(WPT_ERR_INVALID_ARGUMENT, "NULL argument");
    if (width == 0 || height == 0 || uint64_t(width) * height > 0xffffffffull)
        return fail(WPT_ERR_INVALID_ARGUMENT, "width and height must be positive");
    if (camera->surround_mode > WPT_SURROUND_360)
        return fail(WPT_ERR_UNSUPPORTED, "camera surround mode is not known to the kernel");
    if (camera->distortion_type > WPT_DISTORTION_OPENCV)
        return fail(WPT_ERR_UNSUPPORTED, "lens distortion model is not known to the kernel");
    if ((arrays_device[WPT_GT_PIXEL_SPACE_OFFSET_TO_PREV] || arrays_device[WPT_GT_PIXEL_SPACE_OFFSET_TO_NEXT])
            && (camera->surround_mode != WPT_SURROUND_OFF || camera->stereoscopic_distance > 0.0f))
        return fail(WPT_ERR_UNSUPPORTED, "pixel space offsets exist for Surround_Off, non-stereoscopic cameras only (camera.hpp:207-208)");
    GroundTruthArgs args;
    args.scene = scene->view;
    args.cam = *camera;
    args.camPrev = camera_prev ? *camera_prev : *camera;
    args.camNext = camera_next ? *camera_next : *camera;
    args.par = *params;
    args.t0 = times ? times[0] : 0.0f;
    args.tPrev = times ? times[1] : 0.0f;
    args.tNext = times ? times[2] : 0.0f;
    args.par.t0 = args.par.t1 = args.t0; /* one moment, no exposure interval */
    args.width = width;
    args.height = height;
    bool any = false;
    for (int k = 0; k < WPT_GT_ARRAY_COUNT; k++) {
        args.array[k] = arrays_device[k];
        any = any || arrays_device[k];
    }
    if (!any)
        return WPT_OK;
    launchGroundTruth(args, static_cast<hipStream_t>(hip_stream));
    HIP_TRY(hipGetLastError());
    return WPT_OK;
}

wpt_status wpt_ground_truth(wpt_scene* scene, const wpt_camera* camera, const wpt_camera* camera_prev,
        const wpt_camera* camera_next, const float times[3], const wpt_params* params, uint32_t width, uint32_t height,
        void* const arrays_host[WPT_GT_ARRAY_COUNT])
{
    if (!arrays_host)
        return fail(WPT_ERR_INVALID_ARGUMENT, "NULL argument");
    const size_t pixels = size_t(width) * height;
    void* dev[WPT_GT_ARRAY_COUNT];
    for (int k = 0; k < WPT_GT_ARRAY_COUNT; k++)
        dev[k] = nullptr;
    wpt_status st = WPT_OK;
    for (int k = 0; k < WPT_GT_ARRAY_COUNT && st == WPT_OK; k++) {
        if (arrays_host[k] && pixels > 0) {
            hipError_t e = hipMalloc(&dev[k], pixels * wpt_gt_components[k] * 4);
            if (e != hipSuccess)
                st = fail(e == hipErrorOutOfMemory ? WPT_ERR_OUT_OF_MEMORY : WPT_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
        }
    }
    if (st == WPT_OK)
        st = wpt_ground_truth_device(scene, camera, camera_prev, camera_next, times, params, width, height, dev, nullptr);
    if (st == WPT_OK) {
        hipError_t e = hipDeviceSynchronize();
        for (int k = 0; k < WPT_GT_ARRAY_COUNT && e == hipSuccess; k++)
            if (dev[k])
                e = hipMemcpy(arrays_host[k], dev[k], pixels * wpt_gt_components[k] * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess)
            st = fail(WPT_ERR_HIP, std::string("ground truth: ") + hipGetErrorString(e));
    }
    for (int k = 0; k < WPT_GT_ARRAY_COUNT; k++)
        if (dev[k])
            (void)hipFree(dev[k]);
    return st;
}

wpt_status wpt_set_launch_config(uint32_t threads_per_group, uint32_t variant)
{
    if (threads_per_group != 0 && threads_per_group != WG)
        return fail(WPT_ERR_UNSUPPORTED, "this build uses 256 threads per workgroup");
    g_threadsPerGroup = WG;
    g_variant = variant & 0xffu;
    g_leaveEighths = 0;
    g_heavyMin = 0;
    g_leafBias = 0;
    if ((variant >> 8) & 0xffu)
        g_leaveEighths = ((variant >> 8) & 0xffu) - 1; /* byte 1: leave threshold in eighths, plus one */
    if ((variant >> 16) & 0xffu)
        g_heavyMin = ((variant >> 16) & 0xffu) - 1;     /* byte 2: lanes a long block needs, plus one */
    if ((variant >> 24) & 0xffu)
        g_leafBias = (variant >> 24) & 0xffu;           /* byte 3: leaf bias */
    return WPT_OK;
}

wpt_status wpt_set_wavefront(uint32_t mode, uint32_t groups, uint32_t chunk, uint32_t flags)
{
    if (mode > 2u)
        return fail(WPT_ERR_INVALID_ARGUMENT, "wavefront mode must be 0, 1 or 2");
    g_wfMode = mode;
    g_wfConfig.groups = groups & 0xffu;
    g_wfConfig.tracePerCu = (groups >> 8) & 0xffu; /* measurements: workgroups of the trace per compute unit */
    g_wfConfig.shadePerKind = (groups >> 16) & 1u; /* measurements: one shade launch per kind of material */
    g_wfConfig.chunk = chunk;
    g_wfConfig.buckets = (flags & 1u) ? 0u : 1u;
    g_wfConfig.refillIdle = (flags >> 8) & 0x3fu;
    g_wfConfig.leafBias = 0;
    /* bits 16-31: node steps a ray takes per launch of the trace before it is suspended (0 = default, 0xffff = no limit) */
    g_wfConfig.stepBudget = (flags >> 16) == 0xffffu ? 0xffffffffu : (flags >> 16);
    /* bits 1-7: nodes in front of the node array that the trace walks from LDS, in units of 128 (0 = default, 0x7f = none) */
    g_wfConfig.topNodes = ((flags >> 1) & 0x7fu) == 0x7fu ? 0xffffffffu : ((flags >> 1) & 0x7fu) * 128u;
    return WPT_OK;
}

wpt_status wpt_set_top_nodes(uint32_t nodes)
{
    g_topNodes = nodes & 0x7fffffffu;
    return WPT_OK;
}

wpt_status wpt_set_walk(uint32_t flags)
{
    if (flags & ~(WPT_WALK_WIDE | WPT_WALK_FULL_SHADOW | WPT_WALK_COUNT_PRODUCT | WPT_WALK_TRIANGLES_AS_GIVEN))
        return fail(WPT_ERR_INVALID_ARGUMENT, "unknown walk flag");
    g_walk = flags;
    return WPT_OK;
}

/* Waits for the device and reports an error of any launch since the last call (a fault inside a kernel surfaces
 * here).  The kernels themselves have no bounded waits that could run out: every loop ends with its work. */
wpt_status wpt_scene_check(wpt_scene* scene)
{
    if (!scene)
        return fail(WPT_ERR_INVALID_ARGUMENT, "scene is NULL");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipGetLastError());
    return WPT_OK;
}

/* profiling hook: device buffer of 11 uint64 that counted launches add their wave-scheduler
 * statistics to (rounds, loop iterations and lane counts per state); NULL switches it off */
namespace {
static wpt_status postprocLaunch(int op, const float* in, void* out, uint64_t pixels, float a, float b, uint32_t* maxBits, void* hip_stream)
{
    if (!in || pixels == 0 || pixels > 0x7fffffffull * 256ull)
        return fail(WPT_ERR_INVALID_ARGUMENT, "bad frame for post-processing");
    hipLaunchKernelGGL(wpt_postproc_kernel, dim3(uint32_t((pixels + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(hip_stream),
            op, in, out, pixels, a, b, maxBits);
    HIP_TRY(hipGetLastError());
    return WPT_OK;
}
}

wpt_status wpt_postproc_to_srgb(const float* rgb_device, uint8_t* srgb_device, uint64_t pixels, void* hip_stream)
{
    if (!srgb_device)
        return fail(WPT_ERR_INVALID_ARGUMENT, "NULL output");
    return postprocLaunch(0, rgb_device, srgb_device, pixels, 0.0f, 0.0f, nullptr, hip_stream);
}

wpt_status wpt_postproc_uniform_rational_quantization(const float* rgb_device, float* out_device, uint64_t pixels, float max_val,
        float brightness, void* hip_stream)
{
    if (!out_device)
        return fail(WPT_ERR_INVALID_ARGUMENT, "NULL output");
    return postprocLaunch(1, rgb_device, out_device, pixels, max_val, brightness, nullptr, hip_stream);
}

wpt_status wpt_postproc_scale_luminance(const float* rgb_device, float* out_device, uint64_t pixels, float factor, float clamp,
        void* hip_stream)
{
    if (!out_device)
        return fail(WPT_ERR_INVALID_ARGUMENT, "NULL output");
    return postprocLaunch(2, rgb_device, out_device, pixels, factor, clamp, nullptr, hip_stream);
}

wpt_status wpt_postproc_max_luminance(const float* rgb_device, uint64_t pixels, float* result_host, void* hip_stream)
{
    if (!result_host)
        return fail(WPT_ERR_INVALID_ARGUMENT, "NULL output");
    uint32_t* bits = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&bits), sizeof(uint32_t)));
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    hipError_t e = hipMemsetAsync(bits, 0, sizeof(uint32_t), stream);
    wpt_status st = e == hipSuccess ? postprocLaunch(3, rgb_device, nullptr, pixels, 0.0f, 0.0f, bits, hip_stream) : fail(WPT_ERR_HIP, hipGetErrorString(e));
    uint32_t hostBits = 0;
    if (st == WPT_OK) {
        e = hipMemcpyAsync(&hostBits, bits, sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess)
            e = hipStreamSynchronize(stream);
        if (e != hipSuccess)
            st = fail(WPT_ERR_HIP, std::string("max luminance: ") + hipGetErrorString(e));
    }
    (void)hipFree(bits);
    memcpy(result_host, &hostBits, sizeof(float));
    return st;
}

wpt_status wpt_postproc_host(int op, const float* rgb_host, void* out_host, uint64_t pixels, float a, float b)
{
    if (!rgb_host || !out_host || pixels == 0 || op < 0 || op > 3)
        return fail(WPT_ERR_INVALID_ARGUMENT, "bad post-processing request");
    if (wpt_device_count() <= 0)
        return fail(WPT_ERR_NO_DEVICE, "no HIP device is available; post-processing has no CPU fallback either");
    float* dIn = nullptr;
    void* dOut = nullptr;
    const size_t inBytes = size_t(pixels) * 3 * sizeof(float);
    const size_t outBytes = op == 0 ? size_t(pixels) * 3 : (op == 3 ? 0 : inBytes);
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dIn), inBytes));
    hipError_t e = hipMemcpy(dIn, rgb_host, inBytes, hipMemcpyHostToDevice);
    if (e == hipSuccess && outBytes > 0)
        e = hipMalloc(&dOut, outBytes);
    wpt_status st = e == hipSuccess ? WPT_OK : fail(WPT_ERR_HIP, std::string("post-processing buffers: ") + hipGetErrorString(e));
    if (st == WPT_OK) {
        if (op == 0)
            st = wpt_postproc_to_srgb(dIn, static_cast<uint8_t*>(dOut), pixels, nullptr);
        else if (op == 1)
            st = wpt_postproc_uniform_rational_quantization(dIn, static_cast<float*>(dOut), pixels, a, b, nullptr);
        else if (op == 2)
            st = wpt_postproc_scale_luminance(dIn, static_cast<float*>(dOut), pixels, a, b, nullptr);
        else
            st = wpt_postproc_max_luminance(dIn, pixels, static_cast<float*>(out_host), nullptr);
    }
    if (st == WPT_OK && outBytes > 0) {
        e = hipMemcpy(out_host, dOut, outBytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess)
            st = fail(WPT_ERR_HIP, std::string("post-processing download: ") + hipGetErrorString(e));
    }
    (void)hipFree(dIn);
    if (dOut)
        (void)hipFree(dOut);
    return st;
}

wpt_status wpt_set_scheduler_stats(unsigned long long* stats_device)
{
    g_schedStats = stats_device;
    return WPT_OK;
}

const char* wpt_kernel_name(void)
{
    /* the kernel family of the process's most recent render call */
    const char* name = g_kernelName.load(std::memory_order_relaxed);
    return name ? name : "wpt_pathtrace";
}

const char* wpt_device_name(int device)
{
    thread_local std::string name;
    name.clear();
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess)
        name = std::string(prop.name) + " (" + prop.gcnArchName + ", " + std::to_string(prop.multiProcessorCount) + " CUs)";
    return name.c_str();
}

uint32_t wpt_last_render_passes(void)
{
    return g_lastPasses.load(std::memory_order_relaxed);
}

const char* wpt_build_info(void)
{
#if defined(__clang_version__)
    return "hipcc / clang " __clang_version__ ", --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt "
           "-fno-gpu-flush-denormals-to-zero";
#else
    return "unknown compiler";
#endif
}

const char* wpt_last_error(void)
{
    return g_error.c_str();
}

/* test hook: evaluates one arithmetic primitive on the device for n inputs (device pointers) */
wpt_status wpt_selftest_aabb(int n, const float* boxes_device, const float* rays_device, int32_t* out_device)
{
    if (n <= 0 || !boxes_device || !rays_device || !out_device)
        return fail(WPT_ERR_INVALID_ARGUMENT, "bad self test arguments");
    hipLaunchKernelGGL(wpt_selftest_aabb_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, n, boxes_device, rays_device, out_device);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return WPT_OK;
}

wpt_status wpt_selftest_math(int op, int n, const float* a_device, const float* b_device, float* out_device)
{
    if (n <= 0)
        return WPT_OK;
    hipLaunchKernelGGL(wpt_selftest_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, op, n, a_device, b_device, out_device);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return WPT_OK;
}

} /* extern "C" */
