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
    const int *geom_ids;       // object index written as geomId for primitive k (null: k itself)
    int n_tris;                // world-space triangles of the scene's meshes (Scene::loadMesh, src/scene.cpp:234-311)
    const float *tris;         // n_tris x 3 vertices x {pos3, normal3, uv2}
    const int *tri_ids;        // object index of each triangle's mesh
    const float *tri_albedo;   // n_tris x rgb (the mesh's material colour)
    const int *tri_tex;        // texture index per triangle, -1 = untextured (null: no textures)
    const int *tex_desc;       // per texture {byte offset into tex, width, height}, 8-bit RGB rows top to bottom
    const unsigned char *tex;
    float *out_rgb;
    float *out_gbuf;           // AoS texels, or null: the planes below (svgf_planar_gbuffer)
    float *pl_nrm, *pl_pos, *pl_alb;
    int *pl_gid;
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
    float alb[3] = { 0.0f, 0.0f, 0.0f };
    float emit = 0.0f;
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
            t_best = tw; gid = a.geom_ids ? a.geom_ids[k] : k;
            n[0] = nw[0]; n[1] = nw[1]; n[2] = nw[2];
            ph[0] = pw[0]; ph[1] = pw[1]; ph[2] = pw[2];
            alb[0] = g.albedo[0]; alb[1] = g.albedo[1]; alb[2] = g.albedo[2]; emit = g.emittance;
        }
    }
    // Triangle meshes: the nearest triangle of the whole scene (the reference walks one BVH over all of them and keeps the hit
    // if it belongs to the mesh being tested, src/pathtrace.cu:241-252), glm::intersectRayTriangle (gtx/intersect.inl:37-74),
    // normal = n0 * b.x + n1 * b.y + n2 * (1 - b.x - b.y) normalised (Triangle::Intersect, src/sceneStructs.h:168-172, sic)
    {
        int best = -1;
        float bt = t_best, bbx = 0.0f, bby = 0.0f;
        for (int i = 0; i < a.n_tris; i++) {
            const float *T = a.tris + 24 * (size_t)i;
            const float e1[3] = { T[8] - T[0], T[9] - T[1], T[10] - T[2] }, e2[3] = { T[16] - T[0], T[17] - T[1], T[18] - T[2] };
            const float pv[3] = { d[1] * e2[2] - e2[1] * d[2], d[2] * e2[0] - e2[2] * d[0], d[0] * e2[1] - e2[0] * d[1] };
            const float det = (e1[0] * pv[0] + e1[1] * pv[1]) + e1[2] * pv[2];
            if (det < 1.1920929e-7f) continue;
            const float f = 1.0f / det;
            const float sv[3] = { o[0] - T[0], o[1] - T[1], o[2] - T[2] };
            const float bx = f * ((sv[0] * pv[0] + sv[1] * pv[1]) + sv[2] * pv[2]);
            if (bx < 0.0f || bx > 1.0f) continue;
            const float q[3] = { sv[1] * e1[2] - e1[1] * sv[2], sv[2] * e1[0] - e1[2] * sv[0], sv[0] * e1[1] - e1[0] * sv[1] };
            const float by = f * ((d[0] * q[0] + d[1] * q[1]) + d[2] * q[2]);
            if (by < 0.0f || by + bx > 1.0f) continue;
            const float t = f * ((e2[0] * q[0] + e2[1] * q[1]) + e2[2] * q[2]);
            if (t > 0.0f && t < bt) { bt = t; best = i; bbx = bx; bby = by; }
        }
        if (best >= 0) {
            const float *T = a.tris + 24 * (size_t)best;
            const float w2 = 1.0f - bbx - bby;
            for (int c = 0; c < 3; c++) {
                n[c] = (T[3 + c] * bbx + T[11 + c] * bby) + T[19 + c] * w2;
                ph[c] = o[c] + bt * d[c];
                alb[c] = a.tri_albedo[3 * (size_t)best + c];
            }
            normalise3(n);
            const int tx = a.tri_tex ? a.tri_tex[best] : -1;
            if (tx >= 0) {      // Texture::getColor (src/sceneStructs.h:208-219) at the interpolated uv (Triangle::Intersect :162-164)
                const float u = (T[6] * w2 + T[14] * bbx) + T[22] * bby, v = (T[7] * w2 + T[15] * bbx) + T[23] * bby;
                const int tw = a.tex_desc[3 * tx + 1], th = a.tex_desc[3 * tx + 2];
                int X = (int)fminf(1.0f * (float)tw * u, 1.0f * (float)tw - 1.0f), Y = (int)fminf(1.0f * (float)th * (1.0f - v), 1.0f * (float)th - 1.0f);
                X = X < 0 ? 0 : X; Y = Y < 0 ? 0 : Y;                          // (the reference indexes out of bounds here)
                const unsigned char *px = a.tex + a.tex_desc[3 * tx] + 3 * ((size_t)Y * tw + X);
                for (int c = 0; c < 3; c++) alb[c] = 0.003921568627f * (float)px[c];
            }
            emit = 0.0f;
            gid = a.tri_ids[best];
            t_best = bt;
        }
    }

    const bool miss = gid < 0;
    float pos[3];
    for (int c = 0; c < 3; c++) pos[c] = miss ? (o[c] + -1.0f * d[c]) : ph[c];      // t = -1 on a miss (src/pathtrace.cu:318)

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
    if (!a.out_gbuf) {        // planar: the denoiser's own current-frame planes; albedo * ialbedo with ialbedo == 1
        a.pl_nrm[3 * (size_t)p] = n[0]; a.pl_nrm[3 * (size_t)p + 1] = n[1]; a.pl_nrm[3 * (size_t)p + 2] = n[2];
        a.pl_pos[3 * (size_t)p] = pos[0]; a.pl_pos[3 * (size_t)p + 1] = pos[1]; a.pl_pos[3 * (size_t)p + 2] = pos[2];
        if (a.pl_alb) { a.pl_alb[3 * (size_t)p] = alb[0]; a.pl_alb[3 * (size_t)p + 1] = alb[1]; a.pl_alb[3 * (size_t)p + 2] = alb[2]; }
        a.pl_gid[p] = gid;
        return;
    }
    float *g = a.out_gbuf + 13 * (size_t)p;
    g[0] = n[0]; g[1] = n[1]; g[2] = n[2];
    g[3] = pos[0]; g[4] = pos[1]; g[5] = pos[2];
    g[6] = alb[0]; g[7] = alb[1]; g[8] = alb[2];
    g[9] = 1.0f; g[10] = 1.0f; g[11] = 1.0f;
    reinterpret_cast<int *>(g)[12] = gid;
}

}  // namespace

static int scene_render_impl(int device, void *out_rgb_dev, void *out_gbuffer_dev, const SvgfPlanarGBuffer *planes, int width, int height,
                             const SvgfCamera *cam, const SvgfSynthParams *sp, const SvgfSceneGeom *geoms, int n_geoms,
                             const int *geom_ids, const float *tris, const int *tri_ids, const float *tri_albedo, int n_tris,
                             const int *tri_tex, const int *tex_desc, const unsigned char *tex_data, int n_tex,
                             const float light[3], void *stream)
{
    if (!out_rgb_dev || (!out_gbuffer_dev && !planes) || !cam || !sp || !light || width <= 0 || height <= 0) return SVGF_ERR_INVALID_ARG;
    if (!out_gbuffer_dev && (!planes->normal || !planes->position || !planes->geom_id)) return SVGF_ERR_INVALID_ARG;
    if (n_geoms < 0 || n_geoms > SVGF_SCENE_MAX_GEOMS || (n_geoms > 0 && !geoms)) return SVGF_ERR_INVALID_ARG;
    if (n_tris < 0 || n_tris > SVGF_SCENE_MAX_TRIS || (n_tris > 0 && (!tris || !tri_ids || !tri_albedo))) return SVGF_ERR_INVALID_ARG;
    if (n_tex < 0 || n_tex > 64 || (n_tex > 0 && (!tri_tex || !tex_desc || !tex_data || n_tris == 0))) return SVGF_ERR_INVALID_ARG;
    size_t n_texbytes = 0;              // bytes of tex_data the descriptors reach: that much is copied, not a byte more
    for (int k = 0; k < n_tex; k++) {
        if (tex_desc[3 * k] < 0 || tex_desc[3 * k + 1] <= 0 || tex_desc[3 * k + 2] <= 0) return SVGF_ERR_INVALID_ARG;
        const size_t end = (size_t)tex_desc[3 * k] + (size_t)3 * tex_desc[3 * k + 1] * tex_desc[3 * k + 2];
        if (end > n_texbytes) n_texbytes = end;
    }
    for (int i = 0; i < n_tris && n_tex > 0; i++)      // -1 = untextured; anything else must name one of the n_tex textures
        if (tri_tex[i] < -1 || tri_tex[i] >= n_tex) return SVGF_ERR_INVALID_ARG;
    const size_t b_tex = n_texbytes > 16 ? n_texbytes : 16;     // the allocation is padded, the copy is not
    if ((long long)width * height >= (1LL << 31) / 16) return SVGF_ERR_UNSUPPORTED;
    SvgfDeviceGuard dev_guard(device);
    if (!dev_guard.ok) return SVGF_ERR_NO_DEVICE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // one staging allocation: [geoms | geom_ids | tris | tri_ids | tri_albedo | tri_tex | tex_desc | tex]
    const size_t b_g = sizeof(SvgfSceneGeom) * (size_t)(n_geoms > 0 ? n_geoms : 1), b_gi = sizeof(int) * (size_t)(n_geoms > 0 ? n_geoms : 1);
    const size_t b_t = sizeof(float) * 24 * (size_t)(n_tris > 0 ? n_tris : 1), b_ti = sizeof(int) * (size_t)(n_tris > 0 ? n_tris : 1);
    const size_t b_ta = sizeof(float) * 3 * (size_t)(n_tris > 0 ? n_tris : 1);
    const size_t b_tt = b_ti, b_td = sizeof(int) * 3 * (size_t)(n_tex > 0 ? n_tex : 1);
    const size_t o_tt = b_g + b_gi + b_t + b_ti + b_ta, o_td = o_tt + b_tt, o_tex = (o_td + b_td + 15) & ~(size_t)15;
    // The uploads run on a library-owned stream `up`, not on the caller's: the host waits for THEM (the arrays are the caller's,
    // usually pageable, often temporaries of a binding, and must have been read when this function returns) without waiting for
    // whatever else is queued on `s` — the previous frame's whole denoise, typically — and without a host synchronisation on a
    // stream that may be under capture.  The kernel and the release of the staging memory stay asynchronous on `s`.
    // (one per device, created on first use under a lock and kept for the life of the process; a creation that fails is not
    // remembered — the next call tries again — and a device index outside the table is refused rather than folded onto slot 0,
    // whose stream belongs to another device)
    static hipStream_t up_streams[64];
    static std::mutex up_mutex;
    if (device < 0 || device >= 64) return SVGF_ERR_UNSUPPORTED;
    hipStream_t up = nullptr;
    {
        std::lock_guard<std::mutex> lock(up_mutex);
        if (!up_streams[device] && hipStreamCreateWithFlags(&up_streams[device], hipStreamNonBlocking) != hipSuccess) up_streams[device] = nullptr;
        up = up_streams[device];
    }
    if (!up) return SVGF_ERR_HIP;
    char *d = nullptr;
    if (hipMallocAsync(reinterpret_cast<void **>(&d), o_tex + b_tex, up) != hipSuccess) return SVGF_ERR_OOM;
    bool ok = true;
    if (n_geoms > 0) ok = ok && hipMemcpyAsync(d, geoms, sizeof(SvgfSceneGeom) * (size_t)n_geoms, hipMemcpyHostToDevice, up) == hipSuccess;
    if (n_geoms > 0 && geom_ids) ok = ok && hipMemcpyAsync(d + b_g, geom_ids, sizeof(int) * (size_t)n_geoms, hipMemcpyHostToDevice, up) == hipSuccess;
    if (n_tris > 0) {
        ok = ok && hipMemcpyAsync(d + b_g + b_gi, tris, sizeof(float) * 24 * (size_t)n_tris, hipMemcpyHostToDevice, up) == hipSuccess;
        ok = ok && hipMemcpyAsync(d + b_g + b_gi + b_t, tri_ids, sizeof(int) * (size_t)n_tris, hipMemcpyHostToDevice, up) == hipSuccess;
        ok = ok && hipMemcpyAsync(d + b_g + b_gi + b_t + b_ti, tri_albedo, sizeof(float) * 3 * (size_t)n_tris, hipMemcpyHostToDevice, up) == hipSuccess;
    }
    if (n_tex > 0) {
        ok = ok && hipMemcpyAsync(d + o_tt, tri_tex, sizeof(int) * (size_t)n_tris, hipMemcpyHostToDevice, up) == hipSuccess;
        ok = ok && hipMemcpyAsync(d + o_td, tex_desc, sizeof(int) * 3 * (size_t)n_tex, hipMemcpyHostToDevice, up) == hipSuccess;
        ok = ok && hipMemcpyAsync(d + o_tex, tex_data, n_texbytes, hipMemcpyHostToDevice, up) == hipSuccess;
    }
    ok = ok && hipStreamSynchronize(up) == hipSuccess;      // allocation and uploads complete: `s` may use the memory without an event
    if (!ok) { (void)hipFreeAsync(d, up); return SVGF_ERR_HIP; }
    SceneArgs a;
    for (int c = 0; c < 3; c++) { a.right[c] = cam->right[c]; a.up[c] = cam->up[c]; a.view[c] = cam->view[c]; a.o[c] = cam->position[c]; a.light[c] = light[c]; }
    a.plx = sp->pixel_length[0]; a.ply = sp->pixel_length[1];
    a.cx = (float)(width * 0.5 - 0.5); a.cy = (float)(height * 0.5 - 0.5);
    a.noise = sp->noise; a.fireflies = sp->fireflies; a.chroma_amp = 0.1f * sp->noise;
    a.W = width; a.H = height; a.frame = sp->frame; a.seed = sp->seed; a.n_geoms = n_geoms;
    a.geoms = reinterpret_cast<const SvgfSceneGeom *>(d);
    a.geom_ids = (n_geoms > 0 && geom_ids) ? reinterpret_cast<const int *>(d + b_g) : nullptr;
    a.n_tris = n_tris;
    a.tris = reinterpret_cast<const float *>(d + b_g + b_gi);
    a.tri_ids = reinterpret_cast<const int *>(d + b_g + b_gi + b_t);
    a.tri_albedo = reinterpret_cast<const float *>(d + b_g + b_gi + b_t + b_ti);
    a.tri_tex = n_tex > 0 ? reinterpret_cast<const int *>(d + o_tt) : nullptr;
    a.tex_desc = reinterpret_cast<const int *>(d + o_td);
    a.tex = reinterpret_cast<const unsigned char *>(d + o_tex);
    a.out_rgb = static_cast<float *>(out_rgb_dev);
    a.out_gbuf = static_cast<float *>(out_gbuffer_dev);
    a.pl_nrm = planes ? planes->normal : nullptr; a.pl_pos = planes ? planes->position : nullptr;
    a.pl_alb = planes ? planes->albedo : nullptr; a.pl_gid = planes ? planes->geom_id : nullptr;
    const int n = width * height;
    hipLaunchKernelGGL(k_scene_frame, dim3((n + 255) / 256), dim3(256), 0, s, a);
    const hipError_t e = hipGetLastError();
    (void)hipFreeAsync(d, s);
    return e == hipSuccess ? SVGF_OK : SVGF_ERR_HIP;
}

extern "C" int svgf_scene_render_mesh(int device, void *out_rgb_dev, void *out_gbuffer_dev, int width, int height,
                                      const SvgfCamera *cam, const SvgfSynthParams *sp, const SvgfSceneGeom *geoms, int n_geoms,
                                      const int *geom_ids, const float *tris, const int *tri_ids, const float *tri_albedo, int n_tris,
                                      const int *tri_tex, const int *tex_desc, const unsigned char *tex_data, int n_tex,
                                      const float light[3], void *stream)
{
    if (!out_gbuffer_dev) return SVGF_ERR_INVALID_ARG;
    return scene_render_impl(device, out_rgb_dev, out_gbuffer_dev, nullptr, width, height, cam, sp, geoms, n_geoms, geom_ids, tris, tri_ids,
                             tri_albedo, n_tris, tri_tex, tex_desc, tex_data, n_tex, light, stream);
}

// the same frame written into the planes of svgf_planar_gbuffer (the repack fused into the producer)
extern "C" int svgf_scene_render_mesh_planar(int device, void *out_rgb_dev, const SvgfPlanarGBuffer *out_planes, int width, int height,
                                             const SvgfCamera *cam, const SvgfSynthParams *sp, const SvgfSceneGeom *geoms, int n_geoms,
                                             const int *geom_ids, const float *tris, const int *tri_ids, const float *tri_albedo, int n_tris,
                                             const int *tri_tex, const int *tex_desc, const unsigned char *tex_data, int n_tex,
                                             const float light[3], void *stream)
{
    if (!out_planes) return SVGF_ERR_INVALID_ARG;
    return scene_render_impl(device, out_rgb_dev, nullptr, out_planes, width, height, cam, sp, geoms, n_geoms, geom_ids, tris, tri_ids,
                             tri_albedo, n_tris, tri_tex, tex_desc, tex_data, n_tex, light, stream);
}

extern "C" int svgf_scene_render(int device, void *out_rgb_dev, void *out_gbuffer_dev, int width, int height,
                                 const SvgfCamera *cam, const SvgfSynthParams *sp, const SvgfSceneGeom *geoms, int n_geoms,
                                 const float light[3], void *stream)
{
    if (!geoms) return SVGF_ERR_INVALID_ARG;
    return svgf_scene_render_mesh(device, out_rgb_dev, out_gbuffer_dev, width, height, cam, sp, geoms, n_geoms, nullptr, nullptr, nullptr,
                                  nullptr, 0, nullptr, nullptr, nullptr, 0, light, stream);
}
