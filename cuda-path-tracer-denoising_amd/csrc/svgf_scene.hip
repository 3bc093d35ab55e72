// svgf_scene.hip — device-side producer driven by a list of scene primitives (on top of f1).
//
// The reference's scenes are lists of transformed unit cubes / unit spheres (and triangle meshes, which belong to the
// out-of-scope path tracer) read from a text format (src/scene.cpp); its first bounce fills the G-buffer from them
// (src/pathtrace.cu:317-323, intersection conventions of src/intersections.h:50,104: cube [-0.5,0.5]^3 and sphere
// r = 0.5 in object space, t measured in world space).  The host side (cuda-path-tracer-denoising_amd/scene.py) parses
// the text and builds the SvgfSceneGeom records; this kernel casts the primary rays against them and writes the
// denoiser's inputs with the same shading / noise stub as svgf_synth.hip.  Oracle: scene.render_scene (numpy), mirrored
// here operation for operation in fp32 with contraction off.
#include "svgf_kernels.h"
#include "../../include/svgf.h"

#include <hip/hip_runtime.h>

#include <cstring>

namespace {

struct SceneArgs {
    float right[3], up[3], view[3], o[3];
    float light[3];
    float plx, ply, cx, cy;
    float noise, fireflies, chroma_amp;
    int W, H, frame, seed, n_geoms;
    const SvgfSceneGeom *geoms;
    float *out_rgb;
    float *out_gbuf;
};

__device__ __forceinline__ float scene_hash(unsigned seed, unsigned frame, unsigned p, unsigned k)
{   // synth.py: hash_uniform
    unsigned x = p * 0x9E3779B1u + k * 0x85EBCA77u + frame * 0xC2B2AE3Du + seed * 0x27D4EB2Fu;
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
    return (float)(x >> 8) * (1.0f / 16777216.0f);
}

#pragma clang fp contract(off)
__device__ __forceinline__ void apply34(const float *m, const float v[3], float w, float out[3])
{   // 3x4 row-major times (v, w): ((m0*v0 + m1*v1) + m2*v2) + m3*w
#pragma clang fp contract(off)
    for (int r = 0; r < 3; r++) out[r] = ((m[4 * r] * v[0] + m[4 * r + 1] * v[1]) + m[4 * r + 2] * v[2]) + m[4 * r + 3] * w;
}

__device__ __forceinline__ void normalise3(float v[3])
{
#pragma clang fp contract(off)
    const float l = sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    v[0] = v[0] / l; v[1] = v[1] / l; v[2] = v[2] / l;
}

__global__ __launch_bounds__(256) void k_scene_frame(SceneArgs a)
{
#pragma clang fp contract(off)
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= a.W * a.H) return;
    const int x = p % a.W, y = p / a.W;
    const float inf = __builtin_huge_valf();

    const float sx = a.plx * ((float)x - a.cx), sy = a.ply * ((float)y - a.cy);
    float d[3];
    for (int c = 0; c < 3; c++) d[c] = (a.view[c] - a.right[c] * sx) - a.up[c] * sy;
    normalise3(d);
    const float o[3] = { a.o[0], a.o[1], a.o[2] };

    float t_best = inf;
    int gid = -1;
    float n[3] = { 0.0f, 0.0f, 0.0f }, ph[3] = { 0.0f, 0.0f, 0.0f };
    for (int k = 0; k < a.n_geoms; k++) {
        const SvgfSceneGeom &g = a.geoms[k];
        float qo[3], qd[3];
        apply34(g.inv, o, 1.0f, qo);
        apply34(g.inv, d, 0.0f, qd);
        normalise3(qd);
        bool hit;
        float tt, nw[3];
        if (g.type == 0) {      // unit cube: slab test, entering face (or leaving face when the origin is inside)
            float tmin = -1e38f, tmax = 1e38f;
            int amin = 0, amax = 0;
            for (int ax = 0; ax < 3; ax++) {
                const float t1 = (-0.5f - qo[ax]) / qd[ax], t2 = (0.5f - qo[ax]) / qd[ax];
                const float ta = fminf(t1, t2), tb = fmaxf(t1, t2);
                if (ta > 0.0f && ta > tmin) { tmin = ta; amin = ax; }
                if (tb < tmax) { tmax = tb; amax = ax; }
            }
            hit = (tmax >= tmin) && (tmax > 0.0f);
            const bool inside = tmin <= 0.0f;
            tt = inside ? tmax : tmin;
            const int axis = inside ? amax : amin;
            const float sgn = (qd[axis] < 0.0f) ? 1.0f : -1.0f;             // the face normal that looks at the ray
            const float no[3] = { axis == 0 ? sgn : 0.0f, axis == 1 ? sgn : 0.0f, axis == 2 ? sgn : 0.0f };
            apply34(g.xf, no, 0.0f, nw);
            normalise3(nw);
        } else {                // unit sphere, radius 0.5
            const float b = (qo[0] * qd[0] + qo[1] * qd[1]) + qo[2] * qd[2];
            const float rad = b * b - (((qo[0] * qo[0] + qo[1] * qo[1]) + qo[2] * qo[2]) - 0.25f);
            const float sq = sqrtf(fmaxf(rad, 0.0f));
            const float ta = -b + sq, tb = -b - sq;
            const bool both_pos = (ta > 0.0f) && (tb > 0.0f), both_neg = (ta < 0.0f) && (tb < 0.0f);
            tt = both_pos ? fminf(ta, tb) : fmaxf(ta, tb);
            hit = (rad >= 0.0f) && !both_neg;
            const float po[3] = { qo[0] + tt * qd[0], qo[1] + tt * qd[1], qo[2] + tt * qd[2] };
            for (int r = 0; r < 3; r++) nw[r] = (g.invT[3 * r] * po[0] + g.invT[3 * r + 1] * po[1]) + g.invT[3 * r + 2] * po[2];
            normalise3(nw);
        }
        const float po[3] = { qo[0] + tt * qd[0], qo[1] + tt * qd[1], qo[2] + tt * qd[2] };
        float pw[3];
        apply34(g.xf, po, 1.0f, pw);
        const float dv0 = o[0] - pw[0], dv1 = o[1] - pw[1], dv2 = o[2] - pw[2];
        const float tw = sqrtf((dv0 * dv0 + dv1 * dv1) + dv2 * dv2);        // t is measured in world space
        if (hit && (tw > 1e-4f) && (tw < t_best)) {
            t_best = tw; gid = k;
            n[0] = nw[0]; n[1] = nw[1]; n[2] = nw[2];
            ph[0] = pw[0]; ph[1] = pw[1]; ph[2] = pw[2];
        }
    }

    const bool miss = gid < 0;
    float pos[3];
    for (int c = 0; c < 3; c++) pos[c] = miss ? (o[c] + -1.0f * d[c]) : ph[c];      // t = -1 on a miss (src/pathtrace.cu:318)
    float alb[3] = { 0.0f, 0.0f, 0.0f };
    float emit = 0.0f;
    if (!miss) { alb[0] = a.geoms[gid].albedo[0]; alb[1] = a.geoms[gid].albedo[1]; alb[2] = a.geoms[gid].albedo[2]; emit = a.geoms[gid].emittance; }

    float tl[3];
    for (int c = 0; c < 3; c++) tl[c] = a.light[c] - pos[c];
    const float dist2 = (tl[0] * tl[0] + tl[1] * tl[1]) + tl[2] * tl[2];
    const float dl = sqrtf(dist2);
    const float lam = fmaxf(((tl[0] / dl) * n[0] + (tl[1] / dl) * n[1]) + (tl[2] / dl) * n[2], 0.0f);
    float shade = 0.15f + (30.0f * lam) / (4.0f + dist2);
    if (emit > 0.0f) shade = emit;

    const unsigned up = (unsigned)p;
    const float u = scene_hash(a.seed, a.frame, up, 0), v = scene_hash(a.seed, a.frame, up, 1);
    float mult = 1.0f + a.noise * (2.0f * u - 1.0f);
    if (v < a.fireflies) mult = mult * 6.0f;
    float col[3];
    for (int c = 0; c < 3; c++) {
        const float chroma = 1.0f + a.chroma_amp * (scene_hash(a.seed, a.frame, up, 2 + c) - 0.5f);
        col[c] = miss ? 0.0f : ((alb[c] * shade) * mult) * chroma;
    }
    float *o_rgb = a.out_rgb + 3 * (size_t)p;
    o_rgb[0] = col[0]; o_rgb[1] = col[1]; o_rgb[2] = col[2];
    float *g = a.out_gbuf + 13 * (size_t)p;
    g[0] = n[0]; g[1] = n[1]; g[2] = n[2];
    g[3] = pos[0]; g[4] = pos[1]; g[5] = pos[2];
    g[6] = alb[0]; g[7] = alb[1]; g[8] = alb[2];
    g[9] = 1.0f; g[10] = 1.0f; g[11] = 1.0f;
    reinterpret_cast<int *>(g)[12] = gid;
}

}  // namespace

extern "C" int svgf_scene_render(int device, void *out_rgb_dev, void *out_gbuffer_dev, int width, int height,
                                 const SvgfCamera *cam, const SvgfSynthParams *sp, const SvgfSceneGeom *geoms, int n_geoms,
                                 const float light[3], void *stream)
{
    if (!out_rgb_dev || !out_gbuffer_dev || !cam || !sp || !geoms || !light || width <= 0 || height <= 0) return SVGF_ERR_INVALID_ARG;
    if (n_geoms < 0 || n_geoms > SVGF_SCENE_MAX_GEOMS) return SVGF_ERR_INVALID_ARG;
    if ((long long)width * height >= (1LL << 31) / 16) return SVGF_ERR_UNSUPPORTED;
    SvgfDeviceGuard dev_guard(device);
    if (!dev_guard.ok) return SVGF_ERR_NO_DEVICE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    SvgfSceneGeom *d_geoms = nullptr;
    const size_t bytes = sizeof(SvgfSceneGeom) * (size_t)(n_geoms > 0 ? n_geoms : 1);
    if (hipMallocAsync(reinterpret_cast<void **>(&d_geoms), bytes, s) != hipSuccess) return SVGF_ERR_OOM;
    if (n_geoms > 0 && hipMemcpyAsync(d_geoms, geoms, sizeof(SvgfSceneGeom) * (size_t)n_geoms, hipMemcpyHostToDevice, s) != hipSuccess) {
        (void)hipFreeAsync(d_geoms, s);
        return SVGF_ERR_HIP;
    }
    SceneArgs a;
    for (int c = 0; c < 3; c++) { a.right[c] = cam->right[c]; a.up[c] = cam->up[c]; a.view[c] = cam->view[c]; a.o[c] = cam->position[c]; a.light[c] = light[c]; }
    a.plx = sp->pixel_length[0]; a.ply = sp->pixel_length[1];
    a.cx = (float)(width * 0.5 - 0.5); a.cy = (float)(height * 0.5 - 0.5);
    a.noise = sp->noise; a.fireflies = sp->fireflies; a.chroma_amp = 0.1f * sp->noise;
    a.W = width; a.H = height; a.frame = sp->frame; a.seed = sp->seed; a.n_geoms = n_geoms;
    a.geoms = d_geoms;
    a.out_rgb = static_cast<float *>(out_rgb_dev);
    a.out_gbuf = static_cast<float *>(out_gbuffer_dev);
    const int n = width * height;
    hipLaunchKernelGGL(k_scene_frame, dim3((n + 255) / 256), dim3(256), 0, s, a);
    const hipError_t e = hipGetLastError();
    (void)hipFreeAsync(d_geoms, s);
    return e == hipSuccess ? SVGF_OK : SVGF_ERR_HIP;
}
