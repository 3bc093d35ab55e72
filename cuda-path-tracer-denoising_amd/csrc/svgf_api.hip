// svgf_api.hip — host side of libsvgf_hip.so: context, history rotation, per-frame kernel sequence, C ABI.
//
// Mirrors the host sequence of reference denoise() (src/denoise.cu:349-402) with two structural changes that
// do not alter results:
//   * no device-to-device copies: the five per-frame cudaMemcpy's (src/denoise.cu:366,391,396-398) become
//     index rotation over ping-pong planes;
//   * variance is never updated in place: each a-trous level reads {colour,variance} plane A and writes plane B,
//     which is the "snapshot" semantics the parity contract fixes (SURVEY.md §7 hard parts, §8c).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>

#include "../../include/svgf.h"
#include "svgf_kernels.h"

#define SVGF_MAX_KERNELS_PER_FRAME (SVGF_MAX_LEVELS + 4)
// the prepare pass of the non-temporal mode fused into the first level (svgf_atrous_prepare_fused.hip, FUSED = 3): on by default
// where the first level runs the lane kernel anyway (measured: profiles/r04_exp_prepare_fused.log)
static const bool kPrepareFusedByDefault = true;
#ifdef SVGF_BUILD_EXPERIMENTS
// a row of the fused temporal + first-level kernel against a row of the plain lane kernel, as measured at 1920x1080 (DESIGN.md
// 5.8, profiles/r04_exp_fused_*.log: 176 us in its last version against 42.7 for the plain level, i.e. the fused kernel LOSES to
// temporal pass + level, 55 + 46 us): with this factor the automatic choice never fuses; kernel_variant 6 forces it
static const double kFusedRowFactor = 4.1;
// temporal frames: only the G-buffer split fused into the first level (FUSED = 4); a measured loss, off unless svgf_exp_set("split_fused", 1)
static const bool kSplitFusedByDefault = false;
#endif

struct svgf_ctx {
    int device, W, H;
    size_t n;
    float4 *cv[6];         // colour + variance planes: history, source, destination of a level (roles rotate).  [3..5]: the second
                           // set of a PIPELINED context (see `pipelined`), allocated when the first frame asks for it
    float *vp[6];          // zero-margined (W+2) x (H+2) copies of cv[k].w: what the step-16/32 levels take their 3x3 variance pre-blur from
    unsigned vp_valid;     // bit k: vp[k] holds the variance of cv[k]
    // Frame pipeline (SvgfParams::inputs_ready, round 5).  A frame's temporal pass needs of the previous frame only what exists
    // once the level feeding the colour history has run (level 1 with the reference's defaults); levels 2-5 of frame n and the
    // temporal pass + level 1 of frame n+1 are independent.  A pipelined context runs even frames on pipe[0] with planes 0-2 and
    // odd frames on pipe[1] with planes 3-5; ev_hist[q] (recorded behind the level that writes the history of a frame of parity q)
    // is all that the other stream's next temporal pass waits for, and the caller's stream waits for ev_done[q] at the end of the
    // call.  The kernels of two consecutive frames then share the chip: one frame's level tails, launch gaps and opening bursts
    // lie under the other's tap rows (two INDEPENDENT sequences on two streams: +9-10 % in aggregate,
    // profiles/r05_exp_two_sequences.log — the ceiling of this).
    int pipelined;         // the second plane set, the internal streams and the events exist (svgf_create_ex / svgf_enable_pipeline)
    int piped_mode;        // frames alternate between the two plane sets (every frame of the context, once on)
    int pipe_serial;       // the probe at enable time found the two internal streams on ONE hardware queue: the promise is refused
    int ever_captured;     // a frame of this context has been recorded into a graph: promised frames order themselves behind `stream`
    hipStream_t pipe[2];
    hipEvent_t ev_hist[2], ev_done[2], ev_in;
    hipEvent_t ev_tdone[2];   // planar frames: behind the temporal pass (the last reader of the PREVIOUS frame's G-buffer planes, which the next producer overwrites)
    int ev_tdone_valid[2];
    int last_modulated;       // the last frame's last level read the (single) albedo plane
    int ev_hist_valid[2];
    unsigned long long ev_hist_cap[2];      // the stream-capture id ev_hist[q] was last recorded under (0: eagerly)
    int ev_done_valid[2];
    unsigned long long ev_done_cap[2];
    long long pipe_frames;     // frames since the context became pipelined (parity = stream and plane set)
    int use_vplane;        // 0 only for A/B measurements (experiments build: svgf_exp_set("no_variance_plane", 1) before svgf_create)
    void *dump;            // 4 KB of scrap for the fused kernel (TemporalArgs::dump)
    char *arena;           // the one allocation cv[], nrm[], gid[], mom[], hlen[], pos[] and dump are carved from
    size_t arena_bytes;
    float4 *tp[2];         // cross-level reuse of the geometric terms: four terms per pixel, written by level L for level L+1 (lane kernels
                           // only, svgf_atrous_lane_reuse.hip; experiments build); allocated by svgf_create when use_reuse is set
    int use_split_fused;   // experiments build, svgf_exp_set("split_fused", 1): the G-buffer split of temporal frames in the first level's loaders
    int use_reuse;         // experiments build, svgf_exp_set("reuse", 1) before svgf_create: measured a loss, profiles/r04_ab_reuse_*.log
    int n_cu;              // compute units of the context's device (launch-geometry cost model of the kernel choice)
    signed char lane_cheaper[8];   // per log2(step): -1 not evaluated yet, 1 the lane kernel's estimate is the lower one
    signed char fuse_pays;         // -1 not evaluated yet, 1: the fused temporal + first-level kernel is the cheaper way through both
    float *nrm[2];
    int *gid[2];
    float *pos[2];
    float *albedo;         // planar path only (svgf_planar_gbuffer): packed float3 albedo * ialbedo, allocated on first use
    float2 *mom[2];
    int *hlen[2];
    int hist;      // cv index holding the colour history
    int acc;       // cv index the last temporal pass wrote
    int cur;       // mom/hlen index holding the history the next frame reads
    int gcur;      // nrm/gid/pos index holding the previous frame's planes
    float view_prev[16];   // column-major; identity until the first frame (reference src/denoise.cu:15)
    // state capture for tests
    int capture;
    float4 *cv_capture;
    // staging for svgf_denoise_host
    float *st_in, *st_out; void *st_g;
    // profiling
    int prof_frames;       // 0 = off
    int prof_stride;       // bracket every prof_stride-th frame
    long long frame_no;    // frames denoised since profile_enable
    long long prof_count;  // frames recorded since enable
    hipEvent_t *ev;        // prof_frames * SVGF_MAX_KERNELS_PER_FRAME * 2
    int *ev_kind;          // prof_frames * SVGF_MAX_KERNELS_PER_FRAME
    int *ev_n;             // prof_frames
    char err[512];
};

static char g_create_err[512] = "";

#define HIPC(ctx, call)                                                                              \
    do {                                                                                             \
        hipError_t e__ = (call);                                                                     \
        if (e__ != hipSuccess) {                                                                     \
            snprintf((ctx)->err, sizeof((ctx)->err), "%s failed: %s (%s:%d)", #call,                 \
                     hipGetErrorString(e__), __FILE__, __LINE__);                                    \
            return SVGF_ERR_HIP;                                                                     \
        }                                                                                            \
    } while (0)

// ---- host copy of the view-matrix construction (reference GetViewMatrix src/denoise.cu:342-347) -------------
// inverse of the column-major matrix [right|0, up|0, view|0, position|1] by 2x2-minor (cofactor) expansion in the
// operation order of glm 0.9.6.3 (external/include/glm/detail/type_mat4x4.inl:37-92), so fp32 rounding matches.
static void view_matrix_from_camera(const SvgfCamera *cam, float *out)
{
    float m[16];
    for (int r = 0; r < 3; r++) {
        m[0 + r] = cam->right[r]; m[4 + r] = cam->up[r]; m[8 + r] = cam->view[r]; m[12 + r] = cam->position[r];
    }
    m[3] = m[7] = m[11] = 0.0f; m[15] = 1.0f;
    auto M = [&](int c, int r) { return m[c * 4 + r]; };
    float fac[6][4];
    const int rows[6][2] = { {2, 3}, {1, 3}, {1, 2}, {0, 3}, {0, 2}, {0, 1} };
    for (int k = 0; k < 6; k++) {
        const int ra = rows[k][0], rb = rows[k][1];
        const float a = M(2, ra) * M(3, rb) - M(3, ra) * M(2, rb);
        const float b = M(1, ra) * M(3, rb) - M(3, ra) * M(1, rb);
        const float c = M(1, ra) * M(2, rb) - M(2, ra) * M(1, rb);
        fac[k][0] = a; fac[k][1] = a; fac[k][2] = b; fac[k][3] = c;
    }
    float vec[4][4];
    for (int r = 0; r < 4; r++) { vec[r][0] = M(1, r); vec[r][1] = vec[r][2] = vec[r][3] = M(0, r); }
    const int comb[4][6] = { {1, 0, 2, 1, 3, 2}, {0, 0, 2, 3, 3, 4}, {0, 1, 1, 3, 3, 5}, {0, 2, 1, 4, 2, 5} };
    float inv[16];
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++) {
            float t = vec[comb[c][0]][r] * fac[comb[c][1]][r] - vec[comb[c][2]][r] * fac[comb[c][3]][r];
            t = t + vec[comb[c][4]][r] * fac[comb[c][5]][r];
            inv[c * 4 + r] = ((c + r) & 1) ? t * -1.0f : t * 1.0f;
        }
    const float d0 = m[0] * inv[0], d1 = m[1] * inv[4], d2 = m[2] * inv[8], d3 = m[3] * inv[12];
    const float det = (d0 + d1) + (d2 + d3);
    const float rdet = 1.0f / det;
    for (int k = 0; k < 16; k++) out[k] = inv[k] * rdet;
}

#ifdef SVGF_BUILD_EXPERIMENTS
// ---- tuning table of the experiments build (libsvgf_hip_exp.so): name -> int, process-wide.  The product build has neither the
// table nor the entry point: SVGF_TUNE() is its default there and no environment variable is read anywhere in the library. ----
namespace {
struct ExpEntry { char name[32]; int value; };
ExpEntry g_exp[64];
int g_exp_n = 0;
std::mutex g_exp_mu;
}  // namespace
int svgf_exp_get(const char *name, int dflt)
{
    std::lock_guard<std::mutex> lk(g_exp_mu);
    for (int k = 0; k < g_exp_n; k++) if (!strcmp(g_exp[k].name, name)) return g_exp[k].value;
    return dflt;
}
extern "C" int svgf_exp_set(const char *name, int value)
{
    if (!name || strlen(name) >= sizeof(g_exp[0].name)) return SVGF_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(g_exp_mu);
    for (int k = 0; k < g_exp_n; k++) if (!strcmp(g_exp[k].name, name)) { g_exp[k].value = value; return SVGF_OK; }
    if (g_exp_n >= 64) return SVGF_ERR_OOM;
    strcpy(g_exp[g_exp_n].name, name);
    g_exp[g_exp_n++].value = value;
    return SVGF_OK;
}
extern "C" int svgf_exp_clear(void) { std::lock_guard<std::mutex> lk(g_exp_mu); g_exp_n = 0; return SVGF_OK; }
#endif

extern "C" int svgf_version(void) { return (SVGF_VERSION_MAJOR << 16) | SVGF_VERSION_MINOR; }
// 1: this library was built with -DSVGF_BUILD_EXPERIMENTS (parked kernel variants 5 / 6, tuning table); the product build says 0
extern "C" int svgf_build_has_experiments(void)
{
#ifdef SVGF_BUILD_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}
extern "C" int svgf_params_sizeof(void) { return (int)sizeof(SvgfParams); }

extern "C" int svgf_params_default(SvgfParams *p)
{
    if (!p) return SVGF_ERR_INVALID_ARG;
    memset(p, 0, sizeof(*p));
    p->temporal_enable = 0; p->spatial_enable = 0;            // reference src/main.cpp:50-52
    p->color_alpha = 0.2f; p->moment_alpha = 0.2f;            // :53-54
    p->blur_variance = 1;                                     // :55
    p->sigma_l = 0.45f; p->sigma_x = 0.35f; p->sigma_n = 0.2f; // :56-58
    p->atrous_nlevel = 5; p->history_level = 1;               // :59-60
    p->sepcolor = 0; p->addcolor = 0; p->right_view_option = 0;
    return SVGF_OK;
}

// The context's planes live in ONE allocation (the arena), each with kPlanePad bytes in front of and behind its W*H elements:
//   * the fused temporal + first-level kernel reads the 3x3 window around a reprojected position as three 3-element row
//     pieces, and a piece that starts one element left of an image row's first pixel (or ends one right of its last) must stay
//     inside memory the context owns at the plane's two ends;
//   * it addresses every plane as arena + 32-bit byte offset (svgf_atrous_lane_impl.h: LaneFused), which needs them within
//     4 GiB of one base — true by construction up to the sizes atrous_fused_supported() admits.
static const size_t kPlanePad = 128, kPlaneAlign = 4096;
static size_t plane_span(size_t bytes) { return (bytes + 2 * kPlanePad + kPlaneAlign - 1) / kPlaneAlign * kPlaneAlign; }
static void *plane_carve(svgf_ctx *c, size_t *cursor, size_t bytes)
{
    char *p = c->arena + *cursor + kPlanePad;
    *cursor += plane_span(bytes);
    return p;
}

static void free_all(svgf_ctx *c)
{
    if (c->arena) (void)hipFree(c->arena);          // cv[], nrm[], gid[], mom[], hlen[], pos[], dump
    for (int k = 0; k < 6; k++) if (c->vp[k]) (void)hipFree(c->vp[k]);
    for (int k = 3; k < 6; k++) if (c->cv[k]) (void)hipFree(reinterpret_cast<char *>(c->cv[k]) - kPlanePad);
    for (int q = 0; q < 2; q++) {
        if (c->pipe[q]) (void)hipStreamDestroy(c->pipe[q]);
        if (c->ev_hist[q]) (void)hipEventDestroy(c->ev_hist[q]);
        if (c->ev_done[q]) (void)hipEventDestroy(c->ev_done[q]);
        if (c->ev_tdone[q]) (void)hipEventDestroy(c->ev_tdone[q]);
    }
    if (c->ev_in) (void)hipEventDestroy(c->ev_in);
    for (int k = 0; k < 2; k++) if (c->tp[k]) (void)hipFree(c->tp[k]);
    if (c->albedo) (void)hipFree(c->albedo);
    if (c->cv_capture) (void)hipFree(c->cv_capture);
    if (c->st_in) (void)hipFree(c->st_in);
    if (c->st_out) (void)hipFree(c->st_out);
    if (c->st_g) (void)hipFree(c->st_g);
    if (c->ev) {
        for (long long k = 0; k < (long long)c->prof_frames * SVGF_MAX_KERNELS_PER_FRAME * 2; k++) (void)hipEventDestroy(c->ev[k]);
        free(c->ev); free(c->ev_kind); free(c->ev_n);
    }
}

static int zero_state(svgf_ctx *c)
{
    for (int k = 0; k < 6; k++) if (c->cv[k]) HIPC(c, hipMemset(c->cv[k], 0, c->n * sizeof(float4)));
    for (int k = 0; k < 6; k++) if (c->vp[k]) HIPC(c, hipMemset(c->vp[k], 0, (size_t)(c->W + 2) * (c->H + 2) * sizeof(float) + 64));
    c->vp_valid = 0;
    c->pipe_frames = 0; c->ev_hist_valid[0] = c->ev_hist_valid[1] = 0; c->ev_hist_cap[0] = c->ev_hist_cap[1] = 0;      // (a pipelined context stays pipelined)
    c->ev_done_valid[0] = c->ev_done_valid[1] = 0; c->ev_done_cap[0] = c->ev_done_cap[1] = 0;
    c->ev_tdone_valid[0] = c->ev_tdone_valid[1] = 0;
    for (int k = 0; k < 2; k++) {
        HIPC(c, hipMemset(c->nrm[k], 0, c->n * 3 * sizeof(float)));
        HIPC(c, hipMemset(c->gid[k], 0, c->n * sizeof(int)));
        HIPC(c, hipMemset(c->mom[k], 0, c->n * sizeof(float2)));
        HIPC(c, hipMemset(c->hlen[k], 0, c->n * sizeof(int)));
        HIPC(c, hipMemset(c->pos[k], 0, c->n * 3 * sizeof(float)));
    }
    // gcur is deliberately NOT reset: the pointers svgf_planar_gbuffer handed out (planes 1 - gcur) stay the ones the next
    // frame reads, so planar_gbuffer() -> reset() -> producer -> denoise_planar() works (both plane sets are zero again)
    c->hist = 0; c->acc = 0; c->cur = 0;
    // The clears above run on the legacy stream; a caller may enqueue the next svgf_denoise on a non-blocking stream that does not
    // order itself behind them: they are complete before svgf_create / svgf_reset return.
    HIPC(c, hipStreamSynchronize(nullptr));
    return SVGF_OK;
}

static int enable_pipeline(svgf_ctx *c);

extern "C" int svgf_create_ex(int device, int width, int height, unsigned flags, svgf_ctx **out)
{
    if (!out) return SVGF_ERR_INVALID_ARG;
    *out = nullptr;
    if (flags & ~(unsigned)SVGF_CREATE_PIPELINED) {
        snprintf(g_create_err, sizeof(g_create_err), "svgf_create_ex: unknown flags 0x%x", flags);
        return SVGF_ERR_INVALID_ARG;
    }
    if (width <= 0 || height <= 0 || (long long)width * height > (1LL << 30)) {
        snprintf(g_create_err, sizeof(g_create_err), "svgf_create: bad size %dx%d", width, height);
        return SVGF_ERR_INVALID_ARG;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        snprintf(g_create_err, sizeof(g_create_err), "svgf_create: no usable HIP device %d (count %d, %s)", device,
                 ndev, hipGetErrorString(e));
        return SVGF_ERR_NO_DEVICE;
    }
    SvgfDeviceGuard dev_guard(device);
    if (!dev_guard.ok) {
        snprintf(g_create_err, sizeof(g_create_err), "hipSetDevice(%d) failed", device);
        return SVGF_ERR_NO_DEVICE;
    }
    svgf_ctx *c = new (std::nothrow) svgf_ctx();
    if (!c) return SVGF_ERR_OOM;
    memset(c, 0, sizeof(*c));
    c->device = device; c->W = width; c->H = height; c->n = (size_t)width * height;
    for (int k = 0; k < 16; k++) c->view_prev[k] = (k % 5 == 0) ? 1.0f : 0.0f;
    c->use_vplane = SVGF_TUNE("no_variance_plane", 0) ? 0 : 1;
#ifdef SVGF_BUILD_EXPERIMENTS
    c->use_reuse = SVGF_TUNE("reuse", 0) ? 1 : 0;
    c->use_split_fused = SVGF_TUNE("split_fused", kSplitFusedByDefault ? 1 : 0) ? 1 : 0;
#endif
    if (hipDeviceGetAttribute(&c->n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || c->n_cu < 8) c->n_cu = 256;
    memset(c->lane_cheaper, -1, sizeof(c->lane_cheaper));
    c->fuse_pays = -1;
    bool ok = true;
    {
        const size_t n = c->n;
        c->arena_bytes = 3 * plane_span(n * sizeof(float4)) + 2 * (2 * plane_span(n * 3 * sizeof(float)) + 2 * plane_span(n * sizeof(int)) + plane_span(n * sizeof(float2)))
                         + plane_span(4096);
        ok = hipMalloc((void **)&c->arena, c->arena_bytes) == hipSuccess && hipMemset(c->arena, 0, c->arena_bytes) == hipSuccess;
        if (ok) {
            size_t cur = 0;
            for (int k = 0; k < 3; k++) c->cv[k] = (float4 *)plane_carve(c, &cur, n * sizeof(float4));
            for (int k = 0; k < 2; k++) {
                c->nrm[k] = (float *)plane_carve(c, &cur, n * 3 * sizeof(float));
                c->gid[k] = (int *)plane_carve(c, &cur, n * sizeof(int));
                c->mom[k] = (float2 *)plane_carve(c, &cur, n * sizeof(float2));
                c->hlen[k] = (int *)plane_carve(c, &cur, n * sizeof(int));
                c->pos[k] = (float *)plane_carve(c, &cur, n * 3 * sizeof(float));
            }
            c->dump = plane_carve(c, &cur, 4096);
            if (cur != c->arena_bytes) ok = false;         // (the size formula above and the carving below it disagree)
        }
    }
    // (+64 bytes: the step-16/32 lane kernel reads the variance plane in 16-byte pieces that may end 8 bytes behind the last margin)
    for (int k = 0; k < 3 && ok; k++) ok = hipMalloc((void **)&c->vp[k], (size_t)(width + 2) * (height + 2) * sizeof(float) + 64) == hipSuccess;
    // (the terms planes of the parked cross-level reuse: allocated here, never in the middle of a frame)
    for (int k = 0; k < 2 && ok && c->use_reuse; k++) ok = hipMalloc((void **)&c->tp[k], (size_t)(width + 64) * height * sizeof(float4)) == hipSuccess;
    if (!ok) {
        snprintf(g_create_err, sizeof(g_create_err), "svgf_create: hipMalloc failed for %dx%d", width, height);
        free_all(c); delete c;
        return SVGF_ERR_OOM;
    }
    if (zero_state(c) != SVGF_OK) {
        snprintf(g_create_err, sizeof(g_create_err), "%s", c->err);
        free_all(c); delete c;
        return SVGF_ERR_HIP;
    }
    if (flags & SVGF_CREATE_PIPELINED) {
        const int rc = enable_pipeline(c);
        if (rc != SVGF_OK) {
            snprintf(g_create_err, sizeof(g_create_err), "%s", c->err);
            free_all(c); delete c;
            return rc;
        }
    }
    *out = c;
    return SVGF_OK;
}

extern "C" int svgf_create(int device, int width, int height, svgf_ctx **out) { return svgf_create_ex(device, width, height, 0u, out); }

extern "C" int svgf_destroy(svgf_ctx *c)
{
    if (!c) return SVGF_OK;
    SvgfDeviceGuard dev_guard(c->device);
    (void)hipDeviceSynchronize();
    free_all(c);
    delete c;
    return SVGF_OK;
}

extern "C" int svgf_reset(svgf_ctx *c)
{
    if (!c) return SVGF_ERR_INVALID_ARG;
    SvgfDeviceGuard dev_guard(c->device);
    if (!dev_guard.ok) { snprintf(c->err, sizeof(c->err), "hipSetDevice(%d) failed", c->device); return SVGF_ERR_HIP; }
    HIPC(c, hipDeviceSynchronize());
    return zero_state(c);
}

extern "C" int svgf_is_pipelined(const svgf_ctx *c) { return (c && c->pipelined && c->piped_mode) ? 1 : 0; }
extern "C" int svgf_pipeline_status(const svgf_ctx *c) { return !c || !c->pipelined ? 0 : (c->pipe_serial ? 2 : 1); }

extern "C" int svgf_enable_pipeline(svgf_ctx *c)
{
    if (!c) return SVGF_ERR_INVALID_ARG;
    SvgfDeviceGuard dev_guard(c->device);
    if (!dev_guard.ok) { snprintf(c->err, sizeof(c->err), "hipSetDevice(%d) failed", c->device); return SVGF_ERR_HIP; }
    return enable_pipeline(c);
}

extern "C" int svgf_sync(svgf_ctx *c)
{
    if (!c) return SVGF_ERR_INVALID_ARG;
    SvgfDeviceGuard dev_guard(c->device);
    if (!dev_guard.ok) { snprintf(c->err, sizeof(c->err), "hipSetDevice(%d) failed", c->device); return SVGF_ERR_HIP; }
    HIPC(c, hipDeviceSynchronize());
    return SVGF_OK;
}

// Stream-scoped completion: wait for what has been enqueued on `stream` (this context's frames included) and nothing else.
extern "C" int svgf_sync_stream(svgf_ctx *c, void *stream)
{
    if (!c) return SVGF_ERR_INVALID_ARG;
    SvgfDeviceGuard dev_guard(c->device);
    if (!dev_guard.ok) { snprintf(c->err, sizeof(c->err), "hipSetDevice(%d) failed", c->device); return SVGF_ERR_HIP; }
    HIPC(c, hipStreamSynchronize((hipStream_t)stream));
    return SVGF_OK;
}

extern "C" const char *svgf_last_error(const svgf_ctx *c) { return c ? c->err : g_create_err; }
extern "C" int svgf_width(const svgf_ctx *c) { return c ? c->W : 0; }
extern "C" int svgf_height(const svgf_ctx *c) { return c ? c->H : 0; }

extern "C" int svgf_set_capture(svgf_ctx *c, int on)
{
    if (!c) return SVGF_ERR_INVALID_ARG;
    SvgfDeviceGuard dev_guard(c->device);
    if (!dev_guard.ok) { snprintf(c->err, sizeof(c->err), "hipSetDevice(%d) failed", c->device); return SVGF_ERR_HIP; }
    if (on && !c->cv_capture) {
        HIPC(c, hipMalloc((void **)&c->cv_capture, c->n * sizeof(float4)));
        HIPC(c, hipMemset(c->cv_capture, 0, c->n * sizeof(float4)));
    }
    c->capture = on ? 1 : 0;
    return SVGF_OK;
}

// ---- profiling ---------------------------------------------------------------------------------------------

extern "C" int svgf_profile_enable(svgf_ctx *c, int nframes)
{
    if (!c || nframes < 0) return SVGF_ERR_INVALID_ARG;
    SvgfDeviceGuard dev_guard(c->device);
    if (!dev_guard.ok) { snprintf(c->err, sizeof(c->err), "hipSetDevice(%d) failed", c->device); return SVGF_ERR_HIP; }
    if (c->ev && nframes == c->prof_frames) {
        // Same number of slots as before: only the counters restart, the events are reused.  Creating a few hundred events takes
        // 0.2-2 ms, and a benchmark that re-arms the profiler between its warm-up and its timed frames would let the GPU idle for
        // that long: 2 ms of idle time cost the following 20 frames 3 % (profiles/r04_clock_states.txt).
        c->prof_count = 0; c->frame_no = 0;
        memset(c->ev_n, 0, sizeof(int) * (size_t)nframes);
        return SVGF_OK;
    }
    if (c->ev) {
        for (long long k = 0; k < (long long)c->prof_frames * SVGF_MAX_KERNELS_PER_FRAME * 2; k++) (void)hipEventDestroy(c->ev[k]);
        free(c->ev); free(c->ev_kind); free(c->ev_n);
        c->ev = nullptr; c->ev_kind = nullptr; c->ev_n = nullptr;
    }
    c->prof_frames = 0; c->prof_count = 0; c->frame_no = 0;
    if (c->prof_stride < 1) c->prof_stride = 1;
    if (nframes == 0) return SVGF_OK;
    const long long ne = (long long)nframes * SVGF_MAX_KERNELS_PER_FRAME * 2;
    c->ev = (hipEvent_t *)calloc(ne, sizeof(hipEvent_t));
    c->ev_kind = (int *)calloc((size_t)nframes * SVGF_MAX_KERNELS_PER_FRAME, sizeof(int));
    c->ev_n = (int *)calloc(nframes, sizeof(int));
    if (!c->ev || !c->ev_kind || !c->ev_n) {
        free(c->ev); free(c->ev_kind); free(c->ev_n);
        c->ev = nullptr; c->ev_kind = nullptr; c->ev_n = nullptr;
        return SVGF_ERR_OOM;
    }
    for (long long k = 0; k < ne; k++) {
        hipError_t e = hipEventCreate(&c->ev[k]);
        if (e != hipSuccess) {          // give back what was created: profiling stays off
            for (long long j = 0; j < k; j++) (void)hipEventDestroy(c->ev[j]);
            free(c->ev); free(c->ev_kind); free(c->ev_n);
            c->ev = nullptr; c->ev_kind = nullptr; c->ev_n = nullptr;
            snprintf(c->err, sizeof(c->err), "svgf_profile_enable: hipEventCreate failed: %s", hipGetErrorString(e));
            return SVGF_ERR_HIP;
        }
    }
    c->prof_frames = nframes;
    return SVGF_OK;
}

extern "C" int svgf_profile_stride(svgf_ctx *c, int k)
{
    if (!c || k < 1) return SVGF_ERR_INVALID_ARG;
    c->prof_stride = k;
    return SVGF_OK;
}

extern "C" long long svgf_profile_frames(const svgf_ctx *c) { return c ? c->prof_count : 0; }

extern "C" int svgf_profile_read(svgf_ctx *c, int slot, int max_entries, int *kinds, float *ms, int *n_out)
{
    if (!c || !n_out || slot < 0 || slot >= c->prof_frames) return SVGF_ERR_INVALID_ARG;
    SvgfDeviceGuard dev_guard(c->device);
    if (!dev_guard.ok) { snprintf(c->err, sizeof(c->err), "hipSetDevice(%d) failed", c->device); return SVGF_ERR_HIP; }
    const int nk = c->ev_n[slot];
    int w = 0;
    for (int k = 0; k < nk && w < max_entries; k++, w++) {
        const long long base = ((long long)slot * SVGF_MAX_KERNELS_PER_FRAME + k) * 2;
        float t = 0.0f;
        HIPC(c, hipEventElapsedTime(&t, c->ev[base], c->ev[base + 1]));
        if (kinds) kinds[w] = c->ev_kind[slot * SVGF_MAX_KERNELS_PER_FRAME + k];
        if (ms) ms[w] = t;
    }
    *n_out = w;
    return SVGF_OK;
}

namespace {
// Times one launch when profiling is on: the event pair of the slot is armed for the next kernel launch of this thread and
// the launcher's SVGF_LAUNCH_KERNEL attaches it to the dispatch (svgf_kernels.h).  Nothing is recorded on the stream.
struct KernelTimer {
    svgf_ctx *c; hipStream_t s; int slot; int k; bool on;
    bool begin(int kind) {
        if (!on) return true;
        k = c->ev_n[slot];
        if (k >= SVGF_MAX_KERNELS_PER_FRAME) return true;
        const long long e = (long long)slot * SVGF_MAX_KERNELS_PER_FRAME + k;
        c->ev_kind[e] = kind;
        g_svgf_launch_events = SvgfLaunchEvents{ c->ev[e * 2], c->ev[e * 2 + 1] };
        return true;
    }
    bool end() {
        if (!on || k >= SVGF_MAX_KERNELS_PER_FRAME) return true;
        const bool consumed = (g_svgf_launch_events.start == nullptr);      // a launcher that did not go through the macro: no entry
        g_svgf_launch_events = SvgfLaunchEvents{ nullptr, nullptr };
        if (consumed) c->ev_n[slot] = k + 1;
        return true;
    }
};
}  // namespace

thread_local SvgfLaunchEvents g_svgf_launch_events = { nullptr, nullptr };

#define LAUNCH(kind, expr)                                                                           \
    do {                                                                                             \
        timer.begin(kind);                                                                           \
        const hipError_t le__ = (expr);                                                              \
        timer.end();          /* (also disarms the event pair when the launcher failed before launching) */ \
        if (le__ != hipSuccess) {                                                                    \
            snprintf(c->err, sizeof(c->err), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(le__), __FILE__, __LINE__); \
            return SVGF_ERR_HIP;                                                                     \
        }                                                                                            \
    } while (0)

// ---- the frame ------------------------------------------------------------------------------------------------

// ---- the frame pipeline's resources: second plane set, two internal streams, events; and the probe that decides whether the
// promise (inputs_ready = 1) can be honoured.  Called by svgf_create_ex(SVGF_CREATE_PIPELINED) or svgf_enable_pipeline, never by
// svgf_denoise: it allocates and synchronises the device.
__global__ void k_svgf_spin(unsigned long long ticks)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

// Do kernels on pipe[0] and pipe[1] overlap?  The HIP runtime spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues
// (default 4, read when the runtime starts); two streams that land on the same queue run their kernels one after the other, and
// pipelined frames are then 8-10 % SLOWER than ordered ones (the cross-stream events cost, nothing overlaps: profiles/r05_exp_pipeline.log).
// Two one-wave spin kernels of ~200 us, one per stream: together they take ~200 us on two queues and ~400 us on one.
static hipError_t probe_two_streams(int device, hipStream_t sa, hipStream_t sb, bool *overlap, double *ratio_out)
{
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) != hipSuccess || khz <= 0) khz = 100000;
    const double spin_us = 200.0;
    const unsigned long long ticks = (unsigned long long)(spin_us * 1e-6 * khz * 1e3);
    hipEvent_t e[4] = { nullptr, nullptr, nullptr, nullptr };
    hipError_t rc = hipSuccess;
    for (int k = 0; k < 4 && rc == hipSuccess; k++) rc = hipEventCreate(&e[k]);
    double best = 1e30;
#define SVGF_PROBE(call) do { if (rc == hipSuccess) rc = (call); } while (0)
    // (one probe at a time in this process: two contexts created from two host threads would time each other's spin kernels.  Work of
    // OTHER origin on the device can still delay one of the two kernels: up to eight trials, the best one counts, a clear answer ends it)
    static std::mutex probe_mu;
    std::lock_guard<std::mutex> probe_lock(probe_mu);
    for (int trial = 0; trial < 9 && rc == hipSuccess && best > 1.2; trial++) {       // trial 0 loads the code object
        SVGF_PROBE(hipDeviceSynchronize());
        SVGF_PROBE(hipEventRecord(e[0], sa));
        if (rc == hipSuccess) hipLaunchKernelGGL(k_svgf_spin, dim3(1), dim3(64), 0, sa, trial ? ticks : 1ull);
        SVGF_PROBE(hipEventRecord(e[1], sa));
        SVGF_PROBE(hipEventRecord(e[2], sb));
        if (rc == hipSuccess) hipLaunchKernelGGL(k_svgf_spin, dim3(1), dim3(64), 0, sb, trial ? ticks : 1ull);
        SVGF_PROBE(hipEventRecord(e[3], sb));
        SVGF_PROBE(hipGetLastError());
        SVGF_PROBE(hipDeviceSynchronize());
        if (!trial || rc != hipSuccess) continue;
        // span from the earlier start to the later end, on the device's own clock
        float a01 = 0, a23 = 0, a03 = 0, a21 = 0;
        SVGF_PROBE(hipEventElapsedTime(&a01, e[0], e[1]));
        SVGF_PROBE(hipEventElapsedTime(&a23, e[2], e[3]));
        SVGF_PROBE(hipEventElapsedTime(&a03, e[0], e[3]));
        SVGF_PROBE(hipEventElapsedTime(&a21, e[2], e[1]));
        double span = a03 > a21 ? a03 : a21;
        if (a01 > span) span = a01;
        if (a23 > span) span = a23;
        const double one = (a01 < a23 ? a01 : a23);
        if (one > 0 && span / one < best) best = span / one;
    }
#undef SVGF_PROBE
    for (int k = 0; k < 4; k++) if (e[k]) (void)hipEventDestroy(e[k]);
    *ratio_out = best;
    *overlap = best < 1.5;          // 1.0: side by side; 2.0: one after the other
    return rc;
}

static int probe_streams_overlap(svgf_ctx *c, bool *overlap, double *ratio_out)
{
    HIPC(c, probe_two_streams(c->device, c->pipe[0], c->pipe[1], overlap, ratio_out));
    return SVGF_OK;
}

// The same probe for a CALLER's two streams (inputs_ready = 2: the library runs such frames on the streams it is given and cannot
// know how the runtime mapped them to hardware queues).  Synchronises the device: call it once, at set-up.
extern "C" int svgf_streams_overlap(int device, void *stream_a, void *stream_b)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return SVGF_ERR_NO_DEVICE;
    if (stream_a == stream_b) return 0;
    SvgfDeviceGuard dev_guard(device);
    if (!dev_guard.ok) return SVGF_ERR_NO_DEVICE;
    bool overlap = false;
    double ratio = 0.0;
    if (probe_two_streams(device, (hipStream_t)stream_a, (hipStream_t)stream_b, &overlap, &ratio) != hipSuccess) return SVGF_ERR_HIP;
    return overlap ? 1 : 0;
}

static int enable_pipeline(svgf_ctx *c)
{
    if (c->pipelined) return SVGF_OK;
    for (int k = 3; k < 6; k++) {
        char *raw = nullptr;
        if (!c->cv[k]) {
            if (hipMalloc((void **)&raw, c->n * sizeof(float4) + 2 * kPlanePad) != hipSuccess) { snprintf(c->err, sizeof(c->err), "svgf_enable_pipeline: hipMalloc failed for the pipeline's planes"); return SVGF_ERR_OOM; }
            c->cv[k] = reinterpret_cast<float4 *>(raw + kPlanePad);
            HIPC(c, hipMemset(raw, 0, c->n * sizeof(float4) + 2 * kPlanePad));
        }
        if (!c->vp[k]) {
            const size_t vb = (size_t)(c->W + 2) * (c->H + 2) * sizeof(float) + 64;
            if (hipMalloc((void **)&c->vp[k], vb) != hipSuccess) { snprintf(c->err, sizeof(c->err), "svgf_enable_pipeline: hipMalloc failed for the pipeline's planes"); return SVGF_ERR_OOM; }
            HIPC(c, hipMemset(c->vp[k], 0, vb));
        }
    }
    for (int q = 0; q < 2; q++) {
        if (!c->pipe[q]) HIPC(c, hipStreamCreateWithFlags(&c->pipe[q], hipStreamNonBlocking));
        if (!c->ev_hist[q]) HIPC(c, hipEventCreateWithFlags(&c->ev_hist[q], hipEventDisableTiming));
        if (!c->ev_done[q]) HIPC(c, hipEventCreateWithFlags(&c->ev_done[q], hipEventDisableTiming));
        if (!c->ev_tdone[q]) HIPC(c, hipEventCreateWithFlags(&c->ev_tdone[q], hipEventDisableTiming));
    }
    if (!c->ev_in) HIPC(c, hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming));
    bool overlap = true;
    double ratio = 0.0;
    if (const int rc = probe_streams_overlap(c, &overlap, &ratio); rc != SVGF_OK) return rc;
    // The runtime hands out hardware queues in an order of its own, and two streams created one after the other can land on ONE queue
    // even when the process has several (seen in fresh C++ processes under the default GPU_MAX_HW_QUEUES = 4, examples/): replace
    // the second stream until the pair overlaps, a few times; with ONE queue in all, no replacement helps and the promise is refused.
    hipStream_t rejected[5];
    int n_rejected = 0;
    for (int k = 0; k < 5 && !overlap; k++) {
        hipStream_t cand = nullptr;
        if (hipStreamCreateWithFlags(&cand, hipStreamNonBlocking) != hipSuccess) break;
        rejected[n_rejected++] = c->pipe[1];
        c->pipe[1] = cand;
        if (const int rc = probe_streams_overlap(c, &overlap, &ratio); rc != SVGF_OK) break;
    }
    for (int k = 0; k < n_rejected; k++) (void)hipStreamDestroy(rejected[k]);
    HIPC(c, hipDeviceSynchronize());
    c->pipelined = 1;
    c->pipe_serial = overlap ? 0 : 1;
    c->piped_mode = overlap ? 1 : 0;      // (refused: frames stay plain ordered frames until one asks for inputs_ready = 2)
    c->pipe_frames = 0; c->ev_hist_valid[0] = c->ev_hist_valid[1] = 0;
    c->ev_done_valid[0] = c->ev_done_valid[1] = 0;
    if (!overlap)
        snprintf(c->err, sizeof(c->err), "frame pipeline: the context's internal streams share a hardware queue (six candidate pairs; two 200 us kernels took %.2fx one): "
                 "the inputs_ready = 1 promise is refused and such frames run ordered on the caller's stream; start the process with "
                 "GPU_MAX_HW_QUEUES=8 (read by the HIP runtime at initialisation)", ratio);
    return SVGF_OK;
}

// Auto selection between the two fast a-trous kernels for steps 2-32.  The lane-marching kernel works on 480-column strips (at
// steps 16 / 32: 120 / 60 lattice columns of 4 / 8 x-phases), the strip kernel on 256-column strips; both cut the image into
// (strip, y-phase, segment) workgroups that run in rounds of one per CU, and both know what their launch will cost:
// rounds x (segment rows + fixed rows) x the time of a row (1.86 us lane, 1.16 us strip: 42.7 against 48.8 us at 1920x1080).
// The cheaper one runs.  Measured against that model at nine sizes (profiles/r03_exp_widths*.log): within 5 %, same choice as
// the stopwatch everywhere — lane at 1920, 3840, 1600, 3440, 800 (steps 2-8), 2560 and 1280 (steps 2-8, 32); strip at 1024,
// 2048, and at steps 16 of 800 / 1280 / 2560.
static bool lane_pays(svgf_ctx *c, const AtrousArgs &a)
{
    int l = 0;
    while ((1 << l) < a.step && l < 7) l++;
    if (c->lane_cheaper[l] < 0)          // depends on the image size and the device only: evaluated once per context and step
        c->lane_cheaper[l] = atrous_lane_estimate_us(a, c->n_cu) <= atrous_strip_estimate_us(a, c->n_cu) ? 1 : 0;
    return c->lane_cheaper[l] != 0;
}

#ifdef SVGF_BUILD_EXPERIMENTS
// Temporal pass + first level as ONE kernel (svgf_atrous_fused.hip), or as two?  The fused kernel works on 240-column strips
// with both y-phases of a strip in one workgroup and pays the same rounds x (rows + 6) launch geometry as the lane kernel; what
// it saves is the temporal kernel (HBM-bound, 25.6 ns per kilopixel at 1080p) and 56 B/px of traffic between the two.  It runs
// when its estimate is below the estimate of the level it replaces plus the temporal pass.
static bool fuse_pays(svgf_ctx *c, const AtrousArgs &a)
{
    if (c->fuse_pays < 0) {
        const double lane = atrous_lane_supported(a) ? atrous_lane_estimate_us(a, c->n_cu) : 1e30;
        const double strip = atrous_strip_supported(a) ? atrous_strip_estimate_us(a, c->n_cu) : 1e30;
        const double level = lane < strip ? lane : strip;
        const double temporal_us = 0.0256e-3 * (double)c->W * (double)c->H;
        c->fuse_pays = (kFusedRowFactor * atrous_fused_estimate_us(a, c->n_cu) <= level + temporal_us && level < 1e29) ? 1 : 0;
    }
    return c->fuse_pays != 0;
}
#endif

// gbuffer_dev == nullptr: the planar path (svgf_denoise_planar) — the current-frame planes nrm/pos/gid[1 - gcur] (and `albedo`)
// were filled in place by the producer
static int denoise_frame(svgf_ctx *c, void *out_rgb_dev, const void *in_rgb_dev, const void *gbuffer_dev,
                         const SvgfCamera *cam, const SvgfParams *p, void *stream)
{
    if (!out_rgb_dev || !in_rgb_dev || !cam || !p) {
        snprintf(c->err, sizeof(c->err), "svgf_denoise: null argument");
        return SVGF_ERR_INVALID_ARG;
    }
    if (p->atrous_nlevel < 0 || p->atrous_nlevel > SVGF_MAX_LEVELS) {
        snprintf(c->err, sizeof(c->err), "svgf_denoise: atrous_nlevel %d outside 0..%d", p->atrous_nlevel, SVGF_MAX_LEVELS);
        return SVGF_ERR_INVALID_ARG;
    }
    if (p->kernel_variant < 0 || p->kernel_variant > 6 || p->kernel_variant == 3) {
        snprintf(c->err, sizeof(c->err), p->kernel_variant == 3
                 ? "svgf_denoise: kernel_variant %d (experimental shared-weight kernel) is no longer part of the library, see tools/experiments/"
                 : "svgf_denoise: kernel_variant %d unknown", p->kernel_variant);
        return SVGF_ERR_INVALID_ARG;
    }
#ifndef SVGF_BUILD_EXPERIMENTS
    if (p->kernel_variant == 5 || p->kernel_variant == 6) {
        snprintf(c->err, sizeof(c->err), "svgf_denoise: kernel_variant %d is a parked experiment (two-y-phase geometry / temporal pass fused into "
                 "the first level) and is not part of this build; build libsvgf_hip_exp.so (-DSVGF_BUILD_EXPERIMENTS)", p->kernel_variant);
        return SVGF_ERR_UNSUPPORTED;
    }
#endif
    // every level is validated before anything is enqueued or any context state changes: a failure half-way through the
    // cascade would leave the colour history of this frame next to the G-buffer / moments of the previous one
    if (p->kernel_variant == 2 && p->spatial_enable && p->right_view_option != 1 && p->right_view_option != 2) {
        for (int level = 1; level <= p->atrous_nlevel; level++) {
            AtrousArgs probe;
            memset(&probe, 0, sizeof(probe));
            probe.W = c->W; probe.H = c->H; probe.step = 1 << (p->paper_steps ? level - 1 : level);
            if (!atrous_strip_supported(probe)) {
                snprintf(c->err, sizeof(c->err), "svgf_denoise: strip kernel does not support %dx%d step %d", c->W, c->H, probe.step);
                return SVGF_ERR_UNSUPPORTED;
            }
        }
    }
    SvgfDeviceGuard dev_guard(c->device);
    if (!dev_guard.ok) { snprintf(c->err, sizeof(c->err), "hipSetDevice(%d) failed", c->device); return SVGF_ERR_HIP; }
    hipStream_t s_user = (hipStream_t)stream;
    // Pipelined frames (see svgf_ctx::pipelined).  The promise behind inputs_ready = 1: at call time the inputs are complete (and stay
    // untouched until the work of this call is done) — so the frame need not order itself behind the caller's stream, which has
    // waited for the PREVIOUS frame's end, and runs on an internal stream; only the kernel that writes `out` waits for the caller's
    // stream position (readers of that buffer enqueued behind earlier calls).
    // Without the promise a frame of a pipelined context runs on the caller's stream, like any frame of any context, behind the last
    // frame that used its plane set and the history of the frame before it; under stream capture nothing is promised (the frame is
    // recorded on the capturing stream).  The planar path and the experiments build's fused temporal pass never pipeline.
    // (The promise is a permission: it is used where there is something to overlap — a temporal pass and a cascade of two or more
    // levels.  A one-launch frame like BASELINE configs[0] loses more to the four cross-stream events of a pipelined frame than it
    // can gain: 0.0247 -> 0.0342 ms measured, profiles/r05_exp_pipeline.log.)
    // Nor where nothing CAN overlap: with the colour history taken from the LAST level the next temporal pass needs the whole frame.
    // inputs_ready == 2 asks for the pipeline WITHOUT the promise: the frame is ordered behind `stream` like any other work (its
    // inputs may be produced there, its output consumed there), and a caller that alternates TWO streams from frame to frame gets
    // the same overlap with nothing but stream semantics — a stream only ever waits for the frames that were given to it.
    // (planar frames too, ABI 0.9: the promise then covers the context's own current-frame planes — written, complete, and not
    // rewritten before the work of this call is done; a producer that refills them every frame takes the pointers through
    // svgf_planar_gbuffer_stream, which orders it behind the frames that still read them)
    const bool worth = p->temporal_enable && p->spatial_enable && p->atrous_nlevel >= 2 &&
                       p->right_view_option != 1 && p->right_view_option != 2 && p->history_level != p->atrous_nlevel;
    // Only a context whose pipeline resources exist (svgf_create_ex(SVGF_CREATE_PIPELINED) / svgf_enable_pipeline) takes either form up:
    // svgf_denoise never allocates.  The promise additionally needs the two internal streams on different hardware queues (probed when
    // the resources were created, svgf_pipeline_status() == 1): refused, a promised frame is a plain ordered frame.
    bool promise = (p->inputs_ready == 1) && worth && c->pipelined && !c->pipe_serial;
    bool want_pipeline = promise || ((p->inputs_ready == 2) && worth && c->pipelined);
    unsigned long long cap_id = 0;      // != 0: `stream` is being captured into a graph
    if (want_pipeline || c->piped_mode) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        unsigned long long id = 0;
        if (s_user && hipStreamGetCaptureInfo(s_user, &cs, &id) == hipSuccess && cs != hipStreamCaptureStatusNone) { cap_id = id ? id : 1; promise = false; }
        // a frame recorded into a graph did not run when it was recorded: the eager frames behind it order themselves behind the
        // caller's stream (where the graph is launched) until both parities' events are eager ones again
        if (!cap_id && (c->ev_hist_cap[0] || c->ev_hist_cap[1])) promise = false;
    }
#ifdef SVGF_BUILD_EXPERIMENTS
    if (p->kernel_variant == 6 || p->kernel_variant == 5 || c->use_reuse || c->use_split_fused) { promise = false; want_pipeline = false; }
#endif
    if (want_pipeline && !cap_id && !c->piped_mode) {      // (a state flip, nothing is allocated: the colour history lies in plane set 0)
        c->piped_mode = 1; c->pipe_frames = 0;
        c->ev_hist_valid[0] = c->ev_hist_valid[1] = 0; c->ev_done_valid[0] = c->ev_done_valid[1] = 0;
    }
    const bool piped = c->piped_mode != 0;
    if (piped && cap_id) c->ever_captured = 1;
    if (piped) c->ev_tdone_valid[c->pipe_frames & 1] = 0;      // (re-armed below by a planar frame's temporal pass)
    const int pq = piped ? (int)(c->pipe_frames & 1) : 0;
    const int pbase = piped ? 3 * pq : 0;          // this frame's plane set
    // A promised frame runs on the context's stream of its parity.  Every other frame of a pipelined context runs on the caller's own
    // stream, behind the previous frame of the same parity (same plane set), wherever that one ran: with one stream that is what an
    // ordered frame always was; with two streams used in turn (inputs_ready = 2) the caller's streams ARE the pipeline.
    hipStream_t s = (piped && promise) ? c->pipe[pq] : s_user;
    if (piped) {
        if (promise) {
            HIPC(c, hipEventRecord(c->ev_in, s_user));      // the caller's stream position at hand-over: the last level waits for it (`out`)
            // the first pipelined frame: all of it behind the caller's stream.  So is every promised frame of a context that has ever
            // been recorded into a graph: a replay of that graph on the caller's stream touches the same planes and is visible to this
            // frame only through the caller's stream position
            if (c->pipe_frames == 0 || c->ever_captured) HIPC(c, hipStreamWaitEvent(s, c->ev_in, 0));
        }
        if (c->ev_done_valid[pq] && c->ev_done_cap[pq] == cap_id) HIPC(c, hipStreamWaitEvent(s, c->ev_done[pq], 0));
    }
    float *out = (float *)out_rgb_dev;
    const float *in = (const float *)in_rgb_dev;
    const float *g = (const float *)gbuffer_dev;
    const int n = (int)c->n;

    KernelTimer timer{ c, s, 0, 0, false };
    if (c->prof_frames && (c->frame_no % (c->prof_stride > 0 ? c->prof_stride : 1)) == 0) {
        timer.on = true;
        timer.slot = (int)(c->prof_count % c->prof_frames);
        c->ev_n[timer.slot] = 0;
    }

    // 1) temporal accumulation, or constant variance (reference :360-371).  Writes a cv plane `acc` that does not hold the
    //    colour history, and the current-frame G-buffer planes.  When the frame runs the a-trous cascade on the fast path the
    //    temporal pass is not launched at all: the first level's loader waves accumulate the pixels they stage (fused
    //    kernel, svgf_atrous_fused.hip) and `acc` stays unwritten unless something other than that level needs it.
    //    (Rounds 1-3 ran this pass alone on a side stream beside the previous frame's trailing levels — it lost 3-8 % once the lane
    //    kernel ran all five levels; round 5's frame pipeline, above, runs whole frames of alternating parity on two streams.)
    const int old_hist = c->hist;
    const int acc = piped ? (pbase == old_hist ? pbase + 1 : pbase) : (old_hist + 1) % 3;
    // the other stream's frame: this frame's temporal pass (and everything behind it) starts when that frame's colour history,
    // moments, history lengths and G-buffer planes are final
    // (an event recorded under another capture, or eagerly while this frame is captured, or vice versa, is not waited for: such
    // frames are ordered through the caller's stream, see above)
    if (piped && c->ev_hist_valid[1 - pq] && c->ev_hist_cap[1 - pq] == cap_id) HIPC(c, hipStreamWaitEvent(s, c->ev_hist[1 - pq], 0));
    bool hist_event_recorded = false;
    const int gnew = 1 - c->gcur;
    const bool cascade = !(p->right_view_option == 1 || p->right_view_option == 2 || p->atrous_nlevel == 0 || !p->spatial_enable);
    TemporalArgs t;
    memset(&t, 0, sizeof(t));
    bool fused = false, split_fused = false;
    if (p->temporal_enable) {
        t.in_rgb = in; t.gbuf = g; t.cv_hist = c->cv[c->hist]; t.cv_acc = c->cv[acc];
        t.mom_hist = c->mom[c->cur]; t.mom_acc = c->mom[1 - c->cur];
        t.hlen = c->hlen[c->cur]; t.hlen_upd = c->hlen[1 - c->cur];
        t.nrm_prev = c->nrm[c->gcur]; t.gid_prev = c->gid[c->gcur];
        t.nrm_cur = c->nrm[gnew]; t.gid_cur = c->gid[gnew]; t.pos_cur = c->pos[gnew];
        memcpy(t.M, c->view_prev, sizeof(t.M));
        t.W = c->W; t.H = c->H; t.color_alpha_min = p->color_alpha; t.moment_alpha_min = p->moment_alpha;
        t.reproj_sx = p->reproj_scale[0]; t.reproj_sy = p->reproj_scale[1];
        t.pos_prev = c->pos[c->gcur]; t.pos_tol = p->reproj_position_tol;
        t.dump = c->dump; t.arena = c->arena; t.arena_bytes = c->arena_bytes;
#ifdef SVGF_BUILD_EXPERIMENTS      // parked: the temporal pass (or only its G-buffer split) in the first level's loaders, DESIGN.md 5.8
        if (cascade && (p->kernel_variant == 0 || p->kernel_variant == 6) && !p->paper_steps && p->spatial_variance_frames <= 0) {
            AtrousArgs probe;
            memset(&probe, 0, sizeof(probe));
            probe.W = c->W; probe.H = c->H; probe.step = 2;
            fused = atrous_fused_supported(probe, t) && (p->kernel_variant == 6 || fuse_pays(c, probe));
        }
        // The G-buffer split alone can ride in the first level's loaders (svgf_atrous_fused.hip, FUSED = 4): the temporal pass then
        // writes 28 B/px less.  Only where that level runs the lane kernel at step 2 and nothing reads the planes before it does.
        if (!fused && c->use_split_fused && g && cascade && p->kernel_variant == 0 && !p->paper_steps && p->spatial_variance_frames <= 0) {
            AtrousArgs probe;
            memset(&probe, 0, sizeof(probe));
            probe.W = c->W; probe.H = c->H; probe.step = 2; probe.src = c->cv[acc];
            split_fused = atrous_split_fused_supported(probe, t) && atrous_strip_supported(probe) && atrous_lane_supported(probe) && lane_pays(c, probe);
        }
#endif
        t.skip_split = split_fused ? 1 : 0;
        if (!fused) {
            LAUNCH(SVGF_KERNEL_TEMPORAL, launch_temporal(t, s));
            if (piped && !g && !cap_id) { HIPC(c, hipEventRecord(c->ev_tdone[pq], s)); c->ev_tdone_valid[pq] = 1; }
            if (p->spatial_variance_frames > 0)      // f4 extension: spatial variance estimate for short histories (its own profiling slot, same kind)
                LAUNCH(SVGF_KERNEL_TEMPORAL, launch_spatial_variance(c->cv[acc], c->mom[1 - c->cur], c->hlen[1 - c->cur], c->nrm[gnew], c->gid[gnew],
                                                                     c->W, c->H, p->spatial_variance_frames, s));
        }
    } else {
        // Non-temporal mode.  On the AoS boundary the prepare pass (variance = 10, colour copy, G-buffer split) is loads and
        // stores only and rides in the first level's loader waves when the cascade starts with the lane kernel at step 2
        // (svgf_atrous_fused.hip, FUSED = 3): no prepare launch, no colour plane written and read back.
        t.in_rgb = in; t.gbuf = g; t.cv_acc = c->cv[acc];
        t.nrm_cur = c->nrm[gnew]; t.gid_cur = c->gid[gnew]; t.pos_cur = c->pos[gnew];
        t.W = c->W; t.H = c->H;
        if (g && cascade && (p->kernel_variant == 0 || p->kernel_variant == 6) && !p->paper_steps) {
            AtrousArgs probe;
            memset(&probe, 0, sizeof(probe));
            probe.W = c->W; probe.H = c->H; probe.step = 2;
            fused = atrous_prepare_fused_supported(probe, t) && atrous_strip_supported(probe) && atrous_lane_supported(probe) &&
                    (p->kernel_variant == 6 || (kPrepareFusedByDefault && lane_pays(c, probe)));
        }
        if (!fused) LAUNCH(SVGF_KERNEL_PREPARE, launch_prepare(in, g, c->cv[acc], c->nrm[gnew], c->gid[gnew], c->pos[gnew], c->W, c->H, s));
    }
    c->vp_valid &= ~(1u << acc);     // the temporal / prepare pass writes no variance plane: the first level gathers cv.w
    c->acc = acc;
    c->hist = acc;                                   // color_history <- color_acc / input (:366,370)
    if (c->capture && !fused) { HIPC(c, hipMemcpyAsync(c->cv_capture, c->cv[acc], c->n * sizeof(float4), hipMemcpyDeviceToDevice, s)); }      // (before the early release below: the next frame's copy into the one capture buffer must not overtake this one)
    if (piped && cascade && p->temporal_enable && !(p->history_level >= 1 && p->history_level <= p->atrous_nlevel)) {
        // no level of this cascade writes the colour history: it IS the accumulated plane (which no level of this frame overwrites),
        // and everything else the next temporal pass reads was final before: the other stream may go on behind the temporal pass
        HIPC(c, hipEventRecord(c->ev_hist[pq], s));
        c->ev_hist_valid[pq] = 1; c->ev_hist_cap[pq] = cap_id; hist_event_recorded = true;
    }

    // 2) debug views, pass-through or the a-trous cascade (:373-394)
    if (p->right_view_option == 1) {
        LAUNCH(SVGF_KERNEL_DEBUGVIEW, launch_debug_hlen(c->hlen[c->cur], out, n, 100.0f, s));   // pre-update lengths (:374)
    } else if (p->right_view_option == 2) {
        LAUNCH(SVGF_KERNEL_DEBUGVIEW, launch_debug_var(c->cv[acc], out, n, 0.1f, s));            // (:377)
    } else if (p->atrous_nlevel == 0 || !p->spatial_enable) {
        LAUNCH(SVGF_KERNEL_COPYOUT, launch_copy_rgb(c->cv[c->hist], out, n, s));                 // (:382)
    } else {
        int src = c->hist;
        int prev_terms = -1, prev_step = 0;      // tp[] index the previous level stored its geometric terms in, and that level's step
        for (int level = 1; level <= p->atrous_nlevel; level++) {
            const bool last = (level == p->atrous_nlevel);
            const bool keep = (level == p->history_level);      // this level's output becomes the colour history (:391)
            const bool fuse_here = fused && level == 1;
            int dst = -1;
            if (!last || keep) {
                // (the fused level reads the OLD colour history while it writes: its destination is the third plane)
                for (int k = pbase; k < pbase + 3; k++) if (k != src && k != c->hist && !(fuse_here && k == old_hist)) { dst = k; break; }
            }
            AtrousArgs a;
            a.src = c->cv[src]; a.dst = dst >= 0 ? c->cv[dst] : nullptr; a.out_rgb = last ? out : nullptr;
            a.nrm = c->nrm[gnew]; a.pos = c->pos[gnew]; a.gbuf = g; a.albedo = c->albedo;
            a.W = c->W; a.H = c->H;
            a.step = 1 << (p->paper_steps ? level - 1 : level);   // reference: level starts at 1 => steps 2,4,8,16,32 (:98,386)
            a.sigma_c = p->sigma_l; a.sigma_n = p->sigma_n; a.sigma_x = p->sigma_x;
            a.blur_variance = p->blur_variance ? 1 : 0;
            a.modulate = (last && p->sepcolor && p->addcolor) ? 1 : 0;
            // pre-blur source of steps >= 16: the zero-margined 4-byte variance plane the producer level wrote next to its colour
            // plane (see below); the lane kernel at those steps REQUIRES it (its loaders blur the variance from it)
            a.var = (c->use_vplane && a.step >= 16 && ((c->vp_valid >> src) & 1u)) ? c->vp[src] : nullptr;
            a.var_dst = nullptr;
            a.tin = nullptr; a.tout = nullptr; a.t_m = 0; a.t_m_out = 0;
            bool strip = false, lattice = false;
            if (p->kernel_variant != 1) {
                strip = atrous_strip_supported(a);
                lattice = !strip && p->kernel_variant != 2 && atrous_lattice_supported(a);     // steps 64, 128, ...
            }
            enum { K_FUSED, K_LANE, K_LANE2Y, K_STRIP, K_LATTICE, K_GATHER } which;
            if (fuse_here) which = K_FUSED;
#ifdef SVGF_BUILD_EXPERIMENTS
            else if (strip && a.step == 2 && p->kernel_variant == 5 && atrous_lane_supported(a)) which = K_LANE2Y;
#endif
            else if (strip && atrous_lane_supported(a) && (p->kernel_variant >= 4 || (p->kernel_variant == 0 && lane_pays(c, a)))) which = K_LANE;
            else if (strip) which = K_STRIP;
            else if (lattice) which = K_LATTICE;
            else which = K_GATHER;
            // Pre-blur rows (y-1, y+1 of the full-resolution image) of the lane / strip kernels.  At steps >= 16 they are
            // read as 4-byte gathers out of 16-byte colour texels whose lines nobody else on the XCD touches (PMC: 75.6 /
            // 72.6 B/px fetched against 40 algorithmic); there the kernels read them from a zero-margined 4-byte variance plane
            // that the producer level writes next to its colour plane (51.3 / 48.8 B/px, +4 B/px written).  At steps 2-8
            // the neighbouring rows are the sibling y-phases' own rows, already in L2: a plane would ADD 8 B/px (measured,
            // profiles/r02_pmc_hbm.txt), so those levels keep reading colour.w and only the level feeding a step-16 level
            // writes the plane.
            if (which != K_LANE && which != K_STRIP) a.var = nullptr;
            if ((which == K_LANE || which == K_STRIP) && c->use_vplane) {
                if (dst >= 0 && !last && a.step >= 8) a.var_dst = c->vp[dst];
            }
            if (dst >= 0) { if (a.var_dst) c->vp_valid |= 1u << dst; else c->vp_valid &= ~(1u << dst); }
            int terms_out = -1;
#ifdef SVGF_BUILD_EXPERIMENTS
            // Cross-level reuse of the geometric terms (svgf_atrous_lane_reuse.hip): a lane-kernel level reads the four terms the
            // previous lane-kernel level stored for it (its step is twice that level's), and stores four for the next level if that
            // one will run the lane kernel at twice this step.
            if (which == K_LANE && c->use_reuse) {
                if (prev_terms >= 0 && a.step == 2 * prev_step) { a.tin = c->tp[prev_terms]; a.t_m = (c->W + a.step - 1) / a.step; }
                if (!last && 2 * a.step <= 32) {
                    AtrousArgs nx = a;
                    nx.step = 2 * a.step;
                    nx.var = (c->use_vplane && nx.step >= 16 && a.var_dst) ? a.var_dst : nullptr;
                    const bool next_lane = atrous_strip_supported(nx) && atrous_lane_supported(nx) &&
                                           (p->kernel_variant >= 4 || (p->kernel_variant == 0 && lane_pays(c, nx)));
                    if (next_lane) {
                        terms_out = (prev_terms == 0) ? 1 : 0;
                        a.tout = c->tp[terms_out]; a.t_m_out = (c->W + nx.step - 1) / nx.step;
                    }
                }
            }
#endif
            prev_terms = terms_out; prev_step = a.step;
            if (split_fused && level == 1 && which != K_LANE) {      // (the decision above and the kernel choice here use the same model)
                snprintf(c->err, sizeof(c->err), "svgf_denoise: internal error, the G-buffer split was left to a first level that does not run the lane kernel");
                return SVGF_ERR_HIP;
            }
            // The one kernel of a promised frame that writes the caller's `out`: behind what the caller's stream held when the frame
            // was handed over (a reader of the same buffer enqueued behind an earlier call, for instance).  The promise is about the
            // INPUTS only; the output buffer is protected by stream order like everywhere else.
            if (piped && promise && last) HIPC(c, hipStreamWaitEvent(s, c->ev_in, 0));
            switch (which) {
            case K_FUSED:
                // the accumulated plane itself is only written when something besides this level reads it: a later frame (the
                // history is not this level's output) or a test (svgf_set_capture)
                t.cv_acc = (!keep || c->capture) ? c->cv[acc] : nullptr;
#ifdef SVGF_BUILD_EXPERIMENTS
                if (p->temporal_enable) LAUNCH(SVGF_KERNEL_FUSED, launch_atrous_fused(a, t, s));
                else
#endif
                LAUNCH(SVGF_KERNEL_FUSED, launch_atrous_prepare_fused(a, t, s));
                if (c->capture) { HIPC(c, hipMemcpyAsync(c->cv_capture, c->cv[acc], c->n * sizeof(float4), hipMemcpyDeviceToDevice, s)); }
                break;
            case K_LANE:       // steps 2 .. 32: symmetric terms evaluated once
#ifdef SVGF_BUILD_EXPERIMENTS
                if (split_fused && level == 1) { LAUNCH(SVGF_KERNEL_ATROUS, launch_atrous_split_fused(a, t, s)); break; }
                if (a.tin || a.tout) { LAUNCH(SVGF_KERNEL_ATROUS, launch_atrous_lane_reuse(a, s)); break; }
#endif
                LAUNCH(SVGF_KERNEL_ATROUS, launch_atrous_lane(a, s));
                break;
#ifdef SVGF_BUILD_EXPERIMENTS
            case K_LANE2Y:  LAUNCH(SVGF_KERNEL_ATROUS, launch_atrous_lane_2y(a, s)); break;  // A/B partner of the fused kernel's geometry
#endif
            case K_STRIP:   LAUNCH(SVGF_KERNEL_ATROUS, launch_atrous_strip(a, s)); break;
            case K_LATTICE: LAUNCH(SVGF_KERNEL_ATROUS, launch_atrous_lattice(a, s)); break;
            default:        LAUNCH(SVGF_KERNEL_ATROUS, launch_atrous_gather(a, s)); break;
            }
            if (keep) c->hist = dst;
            if (piped && keep && !hist_event_recorded) {
                // everything the NEXT frame's temporal pass reads of this frame is final: the other stream may go on
                HIPC(c, hipEventRecord(c->ev_hist[pq], s));
                c->ev_hist_valid[pq] = 1; c->ev_hist_cap[pq] = cap_id; hist_event_recorded = true;
            }
            src = dst;
        }
    }

    if (piped) {
        // (frames without a cascade — debug views, pass-through, non-temporal frames — release the next frame at their end: they read
        // state the next temporal pass rewrites)
        if (!hist_event_recorded) { HIPC(c, hipEventRecord(c->ev_hist[pq], s)); c->ev_hist_valid[pq] = 1; c->ev_hist_cap[pq] = cap_id; }
        HIPC(c, hipEventRecord(c->ev_done[pq], s));
        c->ev_done_valid[pq] = 1; c->ev_done_cap[pq] = cap_id;
        if (s != s_user) HIPC(c, hipStreamWaitEvent(s_user, c->ev_done[pq], 0));      // what the caller enqueues behind this call sees `out`
        c->pipe_frames++;
    }
    c->last_modulated = (cascade && p->sepcolor && p->addcolor) ? 1 : 0;
    // 3) history rotation (:396-399): planes swap roles instead of being copied
    if (p->temporal_enable) c->cur = 1 - c->cur;
    c->gcur = gnew;
    view_matrix_from_camera(cam, c->view_prev);
    if (timer.on) c->prof_count++;
    c->frame_no++;
    return SVGF_OK;
}

extern "C" int svgf_denoise(svgf_ctx *c, void *out_rgb_dev, const void *in_rgb_dev, const void *gbuffer_dev,
                            const SvgfCamera *cam, const SvgfParams *p, void *stream)
{
    if (!c) return SVGF_ERR_INVALID_ARG;
    if (!gbuffer_dev) {
        snprintf(c->err, sizeof(c->err), "svgf_denoise: null argument");
        return SVGF_ERR_INVALID_ARG;
    }
    return denoise_frame(c, out_rgb_dev, in_rgb_dev, gbuffer_dev, cam, p, stream);
}

// ---- the planar path (the AoS -> plane repack fused into the producer) -----------------------
static int planar_gbuffer(svgf_ctx *c, SvgfPlanarGBuffer *out, bool have_stream, hipStream_t stream)
{
    if (!c || !out) return SVGF_ERR_INVALID_ARG;
    SvgfDeviceGuard dev_guard(c->device);
    if (!dev_guard.ok) { snprintf(c->err, sizeof(c->err), "hipSetDevice(%d) failed", c->device); return SVGF_ERR_HIP; }
    if (!c->albedo) {
        // (first call only.  The clear runs on the legacy stream; the producer that fills the plane may run on a non-blocking
        // stream that does not order itself behind it: the clear is complete before the pointer leaves this function.)
        HIPC(c, hipMalloc((void **)&c->albedo, c->n * 3 * sizeof(float)));
        HIPC(c, hipMemset(c->albedo, 0, c->n * 3 * sizeof(float)));
        HIPC(c, hipStreamSynchronize(nullptr));
    }
    // The planes handed out were the CURRENT planes of the frame before last (its levels read them) and are the PREVIOUS planes of
    // the last frame (its temporal pass reads them); the producer writes them on a stream of its own choosing.  On an ordered context
    // that stream has waited for every frame (stream order).  Frames of a PIPELINED context may still be running on another stream of
    // the caller's or on the context's own: svgf_planar_gbuffer waits for the device; svgf_planar_gbuffer_stream makes `stream` wait
    // for exactly the two things that still read the planes — the end of the frame before last and the temporal pass of the last
    // frame (and the last frame's end when its last level read the one albedo plane) — so that the producer of frame n+1 runs
    // beside the levels of frame n.
    if (c->piped_mode) {
        if (!have_stream) HIPC(c, hipDeviceSynchronize());
        else {
            const int pq = (int)(c->pipe_frames & 1);      // parity of the NEXT frame = parity of the frame before last
            if (c->ev_done_valid[pq] && !c->ev_done_cap[pq]) HIPC(c, hipStreamWaitEvent(stream, c->ev_done[pq], 0));
            if (c->ev_tdone_valid[1 - pq]) HIPC(c, hipStreamWaitEvent(stream, c->ev_tdone[1 - pq], 0));
            else if (c->ev_done_valid[1 - pq] && !c->ev_done_cap[1 - pq]) HIPC(c, hipStreamWaitEvent(stream, c->ev_done[1 - pq], 0));      // (the last frame was not a planar one)
            if (c->last_modulated && c->ev_done_valid[1 - pq] && !c->ev_done_cap[1 - pq]) HIPC(c, hipStreamWaitEvent(stream, c->ev_done[1 - pq], 0));
        }
    }
    const int gnew = 1 - c->gcur;        // the planes the next frame's temporal / prepare pass treats as "current"
    out->normal = c->nrm[gnew]; out->position = c->pos[gnew]; out->geom_id = c->gid[gnew]; out->albedo = c->albedo;
    return SVGF_OK;
}

extern "C" int svgf_planar_gbuffer(svgf_ctx *c, SvgfPlanarGBuffer *out) { return planar_gbuffer(c, out, false, nullptr); }
extern "C" int svgf_planar_gbuffer_stream(svgf_ctx *c, SvgfPlanarGBuffer *out, void *stream) { return planar_gbuffer(c, out, true, (hipStream_t)stream); }

extern "C" int svgf_denoise_planar(svgf_ctx *c, void *out_rgb_dev, const void *in_rgb_dev, const SvgfCamera *cam, const SvgfParams *p, void *stream)
{
    if (!c) return SVGF_ERR_INVALID_ARG;
    if (p && p->sepcolor && p->addcolor && !c->albedo) {
        snprintf(c->err, sizeof(c->err), "svgf_denoise_planar: sepcolor && addcolor needs the albedo plane of svgf_planar_gbuffer");
        return SVGF_ERR_INVALID_ARG;
    }
    return denoise_frame(c, out_rgb_dev, in_rgb_dev, nullptr, cam, p, stream);
}

extern "C" int svgf_denoise_host(svgf_ctx *c, float *out_rgb_host, const float *in_rgb_host,
                                 const SvgfGBufferTexel *gbuffer_host, const SvgfCamera *cam, const SvgfParams *p)
{
    if (!c) return SVGF_ERR_INVALID_ARG;
    if (!out_rgb_host || !in_rgb_host || !gbuffer_host) {
        snprintf(c->err, sizeof(c->err), "svgf_denoise_host: null argument");
        return SVGF_ERR_INVALID_ARG;
    }
    SvgfDeviceGuard dev_guard(c->device);
    if (!dev_guard.ok) { snprintf(c->err, sizeof(c->err), "hipSetDevice(%d) failed", c->device); return SVGF_ERR_HIP; }
    if (!c->st_in) HIPC(c, hipMalloc((void **)&c->st_in, c->n * 3 * sizeof(float)));
    if (!c->st_out) HIPC(c, hipMalloc((void **)&c->st_out, c->n * 3 * sizeof(float)));
    if (!c->st_g) HIPC(c, hipMalloc((void **)&c->st_g, c->n * sizeof(SvgfGBufferTexel)));
    HIPC(c, hipMemcpy(c->st_in, in_rgb_host, c->n * 3 * sizeof(float), hipMemcpyHostToDevice));
    HIPC(c, hipMemcpy(c->st_g, gbuffer_host, c->n * sizeof(SvgfGBufferTexel), hipMemcpyHostToDevice));
    int rc = svgf_denoise(c, c->st_out, c->st_in, c->st_g, cam, p, nullptr);
    if (rc != SVGF_OK) return rc;
    HIPC(c, hipDeviceSynchronize());
    HIPC(c, hipMemcpy(out_rgb_host, c->st_out, c->n * 3 * sizeof(float), hipMemcpyDeviceToHost));
    return SVGF_OK;
}

// ---- state inspection -------------------------------------------------------------------------------------------

extern "C" int svgf_read_state(svgf_ctx *c, int which, void *host_dst, unsigned long long host_bytes)
{
    if (!c || !host_dst) return SVGF_ERR_INVALID_ARG;
    SvgfDeviceGuard dev_guard(c->device);
    if (!dev_guard.ok) { snprintf(c->err, sizeof(c->err), "hipSetDevice(%d) failed", c->device); return SVGF_ERR_HIP; }
    HIPC(c, hipDeviceSynchronize());
    const size_t n = c->n;
    auto need = [&](size_t b) -> bool {
        if (host_bytes < b) { snprintf(c->err, sizeof(c->err), "svgf_read_state: buffer too small (%llu < %zu)", host_bytes, b); return false; }
        return true;
    };
    switch (which) {
    case SVGF_STATE_HISTORY_LENGTH:
        if (!need(n * sizeof(int))) return SVGF_ERR_INVALID_ARG;
        HIPC(c, hipMemcpy(host_dst, c->hlen[c->cur], n * sizeof(int), hipMemcpyDeviceToHost));
        return SVGF_OK;
    case SVGF_STATE_MOMENTS:
        if (!need(n * 2 * sizeof(float))) return SVGF_ERR_INVALID_ARG;
        HIPC(c, hipMemcpy(host_dst, c->mom[c->cur], n * 2 * sizeof(float), hipMemcpyDeviceToHost));
        return SVGF_OK;
    case SVGF_STATE_COLOR_HISTORY:
    case SVGF_STATE_COLOR_ACC:
    case SVGF_STATE_VARIANCE_TEMPORAL: {
        const bool is_hist = (which == SVGF_STATE_COLOR_HISTORY);
        if (!is_hist && !c->cv_capture) {
            snprintf(c->err, sizeof(c->err), "svgf_read_state: enable svgf_set_capture before the frame");
            return SVGF_ERR_INVALID_ARG;
        }
        const size_t out_bytes = (which == SVGF_STATE_VARIANCE_TEMPORAL) ? n * sizeof(float) : n * 3 * sizeof(float);
        if (!need(out_bytes)) return SVGF_ERR_INVALID_ARG;
        float4 *tmp = (float4 *)malloc(n * sizeof(float4));
        if (!tmp) return SVGF_ERR_OOM;
        hipError_t e = hipMemcpy(tmp, is_hist ? c->cv[c->hist] : c->cv_capture, n * sizeof(float4), hipMemcpyDeviceToHost);
        if (e != hipSuccess) { free(tmp); snprintf(c->err, sizeof(c->err), "hipMemcpy: %s", hipGetErrorString(e)); return SVGF_ERR_HIP; }
        float *d = (float *)host_dst;
        if (which == SVGF_STATE_VARIANCE_TEMPORAL) for (size_t k = 0; k < n; k++) d[k] = tmp[k].w;
        else for (size_t k = 0; k < n; k++) { d[3 * k] = tmp[k].x; d[3 * k + 1] = tmp[k].y; d[3 * k + 2] = tmp[k].z; }
        free(tmp);
        return SVGF_OK;
    }
    default:
        snprintf(c->err, sizeof(c->err), "svgf_read_state: unknown state %d", which);
        return SVGF_ERR_INVALID_ARG;
    }
}
