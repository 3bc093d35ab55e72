// svgf_synth.hip — device-side producer of the synthetic 1-spp colour + G-buffer frames.
//
// In the reference the denoiser's inputs are produced on the device by the path tracer: primary rays
// (src/pathtrace.cu:187-208), first-hit G-buffer fill (src/pathtrace.cu:317-323: normal, position = origin + t*dir,
// albedo, ialbedo = 1, geomId, -1 and t = -1 on a miss) and the 1-spp radiance in dev_image.  The path tracer itself
// is out of scope; what the hot path needs from it is exactly that interface, filled with the analytic Cornell-like
// scene SURVEY.md §8(d) prescribes (5 walls, 2 spheres, 1 box, Lambert shading, multiplicative noise, fireflies).
// The scene and every arithmetic step are those of cuda-path-tracer-denoising_amd/synth.py (render_frame with
// noise_model="hash"), operation for operation in fp32 with contraction off, so the two agree bit for bit
// (tests/test_synth_producer.py); the numpy version is the oracle of this row.
//
// One thread per pixel, 64 B written per pixel (12 colour + 52 texel), no reads: HBM-write-bound.
#include "svgf_kernels.h"
#include "../../include/svgf.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>

namespace {

struct SynthArgs {
    float right[3], up[3], view[3], o[3];
    float plx, ply, cx, cy;
    float noise, fireflies, chroma_amp;
    int W, H, frame, seed;
    float *out_rgb;
    float *out_gbuf;           // AoS texels, or null: the planes below
    float *pl_nrm, *pl_pos, *pl_alb; int *pl_gid;
};

// scene of synth.py: room x in [-5,5], y in [0,10], z in [-5,5] (open towards +z), two spheres, one box, one light
__constant__ float kSphC[2][3] = { { 3.0f, 2.0f, 1.0f }, { -2.0f, 1.0f, 3.0f } };
__constant__ float kSphR[2] = { 1.5f, 1.0f };
__constant__ int kSphId[2] = { 6, 8 };
__constant__ float kBoxMin[3] = { -2.5f, 0.0f, -2.5f };
__constant__ float kBoxMax[3] = { 0.5f, 4.0f, 0.5f };
constexpr int kBoxId = 7;
__constant__ float kLight[3] = { 0.0f, 9.5f, 0.0f };
// walls: axis, coordinate, inward normal, geomId
__constant__ int kWallAxis[5] = { 1, 1, 2, 0, 0 };
__constant__ float kWallCoord[5] = { 0.0f, 10.0f, -5.0f, -5.0f, 5.0f };
__constant__ float kWallN[5][3] = { { 0, 1, 0 }, { 0, -1, 0 }, { 0, 0, 1 }, { 1, 0, 0 }, { -1, 0, 0 } };
__constant__ float kAlbedo[9][3] = { { 0.85f, 0.85f, 0.85f }, { 0.85f, 0.85f, 0.85f }, { 0.85f, 0.85f, 0.85f },
                                     { 0.85f, 0.35f, 0.35f }, { 0.35f, 0.85f, 0.35f }, { 0.0f, 0.0f, 0.0f },
                                     { 0.9f, 0.9f, 0.2f },    { 0.3f, 0.5f, 0.9f },    { 0.9f, 0.6f, 0.3f } };

__device__ __forceinline__ float synth_hash(unsigned seed, unsigned frame, unsigned p, unsigned k)
{   // synth.py: hash_uniform
    unsigned x = p * 0x9E3779B1u + k * 0x85EBCA77u + frame * 0xC2B2AE3Du + seed * 0x27D4EB2Fu;
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
    return (float)(x >> 8) * (1.0f / 16777216.0f);
}

#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void k_synth_frame(SynthArgs a)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= a.W * a.H) return;
    const int x = p % a.W, y = p / a.W;
    const float inf = __builtin_huge_valf();

    // primary ray (reference src/pathtrace.cu:199-202; synth.py: d)
    const float sx = a.plx * ((float)x - a.cx), sy = a.ply * ((float)y - a.cy);
    float d[3];
    for (int c = 0; c < 3; c++) d[c] = (a.view[c] - a.right[c] * sx) - a.up[c] * sy;
    const float len = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
    for (int c = 0; c < 3; c++) d[c] = d[c] / len;
    const float o[3] = { a.o[0], a.o[1], a.o[2] };

    float t_best = inf;
    int gid = -1;
    float n[3] = { 0.0f, 0.0f, 0.0f };
    auto consider = [&](float t, int g, float nx, float ny, float nz) {
        if ((t > 1e-4f) && (t < t_best)) { t_best = t; gid = g; n[0] = nx; n[1] = ny; n[2] = nz; }
    };
    for (int w = 0; w < 5; w++) {
        const int ax = kWallAxis[w];
        const float t = (kWallCoord[w] - o[ax]) / d[ax];
        const float ph0 = o[0] + t * d[0], ph1 = o[1] + t * d[1], ph2 = o[2] + t * d[2];
        const bool inside = (ph0 >= -5.001f) && (ph0 <= 5.001f) && (ph1 >= -0.001f) && (ph1 <= 10.001f) &&
                            (ph2 >= -5.001f) && (ph2 <= 5.001f);
        consider(inside ? t : inf, w, kWallN[w][0], kWallN[w][1], kWallN[w][2]);
    }
    for (int s = 0; s < 2; s++) {
        const float oc0 = o[0] - kSphC[s][0], oc1 = o[1] - kSphC[s][1], oc2 = o[2] - kSphC[s][2];
        const float r = kSphR[s];
        const float b = (d[0] * oc0 + d[1] * oc1) + d[2] * oc2;
        const float cc = ((oc0 * oc0 + oc1 * oc1) + oc2 * oc2) - r * r;
        const float disc = b * b - cc;
        const float t = (disc > 0.0f) ? (-b - sqrtf(fmaxf(disc, 0.0f))) : inf;
        const float ph0 = o[0] + t * d[0], ph1 = o[1] + t * d[1], ph2 = o[2] + t * d[2];
        consider(t, kSphId[s], (ph0 - kSphC[s][0]) / r, (ph1 - kSphC[s][1]) / r, (ph2 - kSphC[s][2]) / r);
    }
    {
        float tn[3], tf[3];
        for (int c = 0; c < 3; c++) {
            const float t0 = (kBoxMin[c] - o[c]) / d[c], t1 = (kBoxMax[c] - o[c]) / d[c];
            tn[c] = fminf(t0, t1);
            tf[c] = fmaxf(t0, t1);
        }
        const float tnear = fmaxf(fmaxf(tn[0], tn[1]), tn[2]);
        const float tfar = fminf(fminf(tf[0], tf[1]), tf[2]);
        const bool hitbox = (tnear <= tfar) && (tnear > 0.0f);
        int ax = 0;                                                  // first index of the maximum (numpy argmax)
        if (tn[1] > tn[ax]) ax = 1;
        if (tn[2] > tn[ax]) ax = 2;
        const float dv = d[ax];
        const float sgn = -((dv > 0.0f) ? 1.0f : ((dv < 0.0f) ? -1.0f : dv));
        consider(hitbox ? tnear : inf, kBoxId, ax == 0 ? sgn : 0.0f, ax == 1 ? sgn : 0.0f, ax == 2 ? sgn : 0.0f);
    }

    const bool miss = gid < 0;
    const float t_used = miss ? -1.0f : t_best;                     // reference src/pathtrace.cu:318: t = -1 on a miss
    float pos[3];
    for (int c = 0; c < 3; c++) pos[c] = o[c] + t_used * d[c];

    float alb[3] = { 0.0f, 0.0f, 0.0f };
    if (!miss) { alb[0] = kAlbedo[gid][0]; alb[1] = kAlbedo[gid][1]; alb[2] = kAlbedo[gid][2]; }
    if (gid == 0) {
        const long long s = (long long)(floorf(pos[0]) + floorf(pos[2]));
        const float k = 0.55f + 0.45f * (float)(s & 1);
        for (int c = 0; c < 3; c++) alb[c] = alb[c] * k;
    }

    float tl[3];
    for (int c = 0; c < 3; c++) tl[c] = kLight[c] - pos[c];
    const float dist2 = (tl[0] * tl[0] + tl[1] * tl[1]) + tl[2] * tl[2];
    const float dl = sqrtf(dist2);
    const float lam = fmaxf(((tl[0] / dl) * n[0] + (tl[1] / dl) * n[1]) + (tl[2] / dl) * n[2], 0.0f);
    const float shade = 0.15f + (30.0f * lam) / (4.0f + dist2);

    const unsigned up = (unsigned)p;
    const float u = synth_hash(a.seed, a.frame, up, 0), v = synth_hash(a.seed, a.frame, up, 1);
    float mult = 1.0f + a.noise * (2.0f * u - 1.0f);
    if (v < a.fireflies) mult = mult * 6.0f;
    float col[3];
    for (int c = 0; c < 3; c++) {
        const float chroma = 1.0f + a.chroma_amp * (synth_hash(a.seed, a.frame, up, 2 + c) - 0.5f);
        col[c] = miss ? 0.0f : ((alb[c] * shade) * mult) * chroma;
    }

    float *o_rgb = a.out_rgb + 3 * (size_t)p;
    o_rgb[0] = col[0]; o_rgb[1] = col[1]; o_rgb[2] = col[2];
    if (!a.out_gbuf) {        // planar: the denoiser's own current-frame planes (svgf_planar_gbuffer); albedo * ialbedo, ialbedo == 1
        a.pl_nrm[3 * (size_t)p] = n[0]; a.pl_nrm[3 * (size_t)p + 1] = n[1]; a.pl_nrm[3 * (size_t)p + 2] = n[2];
        a.pl_pos[3 * (size_t)p] = pos[0]; a.pl_pos[3 * (size_t)p + 1] = pos[1]; a.pl_pos[3 * (size_t)p + 2] = pos[2];
        if (a.pl_alb) { a.pl_alb[3 * (size_t)p] = alb[0]; a.pl_alb[3 * (size_t)p + 1] = alb[1]; a.pl_alb[3 * (size_t)p + 2] = alb[2]; }
        a.pl_gid[p] = gid;
        return;
    }
    float *g = a.out_gbuf + 13 * (size_t)p;                          // SvgfGBufferTexel, 52 B
    g[0] = n[0]; g[1] = n[1]; g[2] = n[2];
    g[3] = pos[0]; g[4] = pos[1]; g[5] = pos[2];
    g[6] = alb[0]; g[7] = alb[1]; g[8] = alb[2];
    g[9] = 1.0f; g[10] = 1.0f; g[11] = 1.0f;
    reinterpret_cast<int *>(g)[12] = gid;
}

}  // namespace

extern "C" {

int svgf_synth_camera(int frame, int moving, int width, int height, SvgfCamera *cam, float pixel_length[2])
{
    if (!cam || width <= 0 || height <= 0 || frame < 0) return SVGF_ERR_INVALID_ARG;
    const float PI = 3.14159265358979323846f;
    float lookat[3] = { 0.0f, 5.0f, 0.0f };
    float theta = PI * 0.5f, phi = 0.0f;
    if (moving) {   // reference src/main.cpp:156-169, speeds of SURVEY.md §8(d) config C3
        // the phases are accumulated in fp32, one addition per frame, advanced before use (src/main.cpp:158-162)
        volatile float tx = 0.0f, ty = 0.0f, tz = 0.0f, tt = 0.0f, tp = 0.0f;     // volatile: every sum rounded to fp32
        for (int k = 0; k <= frame; k++) { tx = tx + 0.02f; ty = ty + 0.01f; tz = tz + 0.01f; tt = tt + 0.01f; tp = tp + 0.02f; }
        lookat[0] = 2.0f * sinf(tx);
        lookat[1] = 5.0f + sinf(ty);
        lookat[2] = 1.5f * sinf(tz);
        theta = PI * 0.5f + PI / 18.0f * sinf(tt);
        phi = PI / 12.0f * sinf(tp);
    }
    const float z = 10.5f;
    const float off[3] = { z * sinf(phi) * sinf(theta), z * cosf(theta), z * cosf(phi) * sinf(theta) };
    const float l = sqrtf(off[0] * off[0] + off[1] * off[1] + off[2] * off[2]);
    for (int c = 0; c < 3; c++) cam->view[c] = -off[c] / l;
    // right = cross(view, +y) (not normalised), up = cross(right, view)   (reference src/main.cpp:180-184)
    cam->right[0] = -cam->view[2]; cam->right[1] = 0.0f; cam->right[2] = cam->view[0];
    cam->up[0] = cam->right[1] * cam->view[2] - cam->right[2] * cam->view[1];
    cam->up[1] = cam->right[2] * cam->view[0] - cam->right[0] * cam->view[2];
    cam->up[2] = cam->right[0] * cam->view[1] - cam->right[1] * cam->view[0];
    for (int c = 0; c < 3; c++) cam->position[c] = off[c] + lookat[c];
    if (pixel_length) {   // reference src/scene.cpp:159-166 with FOVY = 45 used as the half angle
        const float yscaled = tanf(45.0f * (PI / 180.0f));
        const float xscaled = yscaled * (float)width / (float)height;
        pixel_length[0] = 2.0f * xscaled / (float)width;
        pixel_length[1] = 2.0f * yscaled / (float)height;
    }
    return SVGF_OK;
}

static int synth_render_impl(int device, void *out_rgb_dev, void *out_gbuffer_dev, const SvgfPlanarGBuffer *planes, int width, int height,
                             const SvgfCamera *cam, const SvgfSynthParams *sp, void *stream)
{
    if (!out_rgb_dev || (!out_gbuffer_dev && !planes) || !cam || !sp || width <= 0 || height <= 0) return SVGF_ERR_INVALID_ARG;
    if (planes && (!planes->normal || !planes->position || !planes->geom_id)) return SVGF_ERR_INVALID_ARG;
    if ((long long)width * height >= (1LL << 31) / 16) return SVGF_ERR_UNSUPPORTED;
    SvgfDeviceGuard dev_guard(device);
    if (!dev_guard.ok) return SVGF_ERR_NO_DEVICE;
    SynthArgs a;
    for (int c = 0; c < 3; c++) { a.right[c] = cam->right[c]; a.up[c] = cam->up[c]; a.view[c] = cam->view[c]; a.o[c] = cam->position[c]; }
    a.plx = sp->pixel_length[0]; a.ply = sp->pixel_length[1];
    a.cx = (float)(width * 0.5 - 0.5); a.cy = (float)(height * 0.5 - 0.5);
    a.noise = sp->noise; a.fireflies = sp->fireflies; a.chroma_amp = 0.1f * sp->noise;
    a.W = width; a.H = height; a.frame = sp->frame; a.seed = sp->seed;
    a.out_rgb = static_cast<float *>(out_rgb_dev);
    a.out_gbuf = static_cast<float *>(out_gbuffer_dev);
    a.pl_nrm = planes ? planes->normal : nullptr; a.pl_pos = planes ? planes->position : nullptr;
    a.pl_alb = planes ? planes->albedo : nullptr; a.pl_gid = planes ? planes->geom_id : nullptr;
    const int n = width * height;
    hipLaunchKernelGGL(k_synth_frame, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? SVGF_OK : SVGF_ERR_HIP;
}

int svgf_synth_render(int device, void *out_rgb_dev, void *out_gbuffer_dev, int width, int height,
                      const SvgfCamera *cam, const SvgfSynthParams *sp, void *stream)
{
    if (!out_gbuffer_dev) return SVGF_ERR_INVALID_ARG;
    return synth_render_impl(device, out_rgb_dev, out_gbuffer_dev, nullptr, width, height, cam, sp, stream);
}

// SURVEY.md 8f row f1, second half: the repack fused into the producer (src/pathtrace.cu:317-323 is where the reference fills
// its AoS texel): the same pixels written straight into the denoiser's planes
int svgf_synth_render_planar(int device, void *out_rgb_dev, const SvgfPlanarGBuffer *out_planes, int width, int height,
                             const SvgfCamera *cam, const SvgfSynthParams *sp, void *stream)
{
    if (!out_planes) return SVGF_ERR_INVALID_ARG;
    return synth_render_impl(device, out_rgb_dev, nullptr, out_planes, width, height, cam, sp, stream);
}

}  // extern "C"
