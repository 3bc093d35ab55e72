// denoise_compat.cpp — the reference's three entry points (reference src/denoise.h:6-8) on top of the C ABI of
// include/svgf.h.  Compile this file INSIDE the renderer's source tree in place of src/denoise.cu (it includes the
// renderer's own denoise.h / main.h for Scene, glm::vec3, GBufferTexel and the ui_* globals) and link libsvgf_hip.so.
// The renderer's call sites stay untouched:
//     runCuda():   denoiseFree(); denoiseInit(scene);              (reference src/main.cpp:194-201)
//     pathtrace(): denoise(dev_denoised_image, dev_image, dev_gbuffer);   (reference src/pathtrace.cu:436-438)
//
// Behaviour kept from the reference: one global denoiser, camera and parameters read at call time from
// scene->state.camera and the ui_* globals (src/denoise.cu:350-351,360-390), synchronous return
// (src/denoise.cu:401).  Behaviour added: HIP errors are reported on stderr instead of being ignored.
#include <cstdio>

#include <hip/hip_runtime.h>

#include "denoise.h"
#include "main.h"
#include "svgf.h"

static_assert(sizeof(GBufferTexel) == sizeof(SvgfGBufferTexel), "G-buffer texel layout must match (52 bytes)");
static_assert(sizeof(glm::vec3) == 12, "glm::vec3 must be 3 packed floats");

static Scene *g_scene = nullptr;
static svgf_ctx *g_ctx = nullptr;

void denoiseInit(Scene *scene)
{
    g_scene = scene;
    const Camera &cam = scene->state.camera;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (g_ctx) { svgf_destroy(g_ctx); g_ctx = nullptr; }
    // The parameter block has grown at its tail between ABI versions (0.2: 72 bytes, 0.3: 80) and svgf_denoise reads all of
    // the library's version of it: a shim compiled against another svgf.h than the libsvgf_hip.so it is linked with would
    // hand over a short struct whose tail fields (reproj_position_tol, spatial_variance_frames) are whatever follows it on
    // the stack.  Refuse loudly instead: no context is created and every denoise() call reports it.
    if (svgf_params_sizeof() != (int)sizeof(SvgfParams)) {
        fprintf(stderr, "denoiseInit: svgf.h / libsvgf_hip.so mismatch: sizeof(SvgfParams) is %d in this build of denoise_compat.cpp, "
                        "%d in the library (version %d.%d); rebuild the shim against the library's include/svgf.h\n",
                (int)sizeof(SvgfParams), svgf_params_sizeof(), svgf_version() >> 16, svgf_version() & 0xffff);
        return;
    }
    const int rc = svgf_create(dev, cam.resolution.x, cam.resolution.y, &g_ctx);
    if (rc != SVGF_OK) fprintf(stderr, "denoiseInit: svgf_create failed (%d): %s\n", rc, svgf_last_error(nullptr));
}

void denoiseFree()
{
    if (g_ctx) svgf_destroy(g_ctx);     // harmless when never initialised, like cudaFree(NULL) in the reference
    g_ctx = nullptr;
}

void denoise(glm::vec3 *output, glm::vec3 *input, GBufferTexel *gbuffer)
{
    if (!g_ctx || !g_scene) { fprintf(stderr, "denoise: denoiseInit has not been called\n"); return; }
    const Camera &cam = g_scene->state.camera;
    SvgfCamera c;
    for (int k = 0; k < 3; k++) {
        c.right[k] = cam.right[k]; c.up[k] = cam.up[k]; c.view[k] = cam.view[k]; c.position[k] = cam.position[k];
    }
    SvgfParams p;
    svgf_params_default(&p);
    p.temporal_enable = ui_temporal_enable; p.spatial_enable = ui_spatial_enable;
    p.color_alpha = ui_color_alpha; p.moment_alpha = ui_moment_alpha;
    p.blur_variance = ui_blurvariance;
    p.sigma_l = ui_sigmal; p.sigma_x = ui_sigmax; p.sigma_n = ui_sigman;
    p.atrous_nlevel = ui_atrous_nlevel; p.history_level = ui_history_level;
    p.sepcolor = ui_sepcolor; p.addcolor = ui_addcolor;
    p.right_view_option = ui_right_view_option;
    int rc = svgf_denoise(g_ctx, output, input, gbuffer, &c, &p, /*stream=*/nullptr);
#ifndef SVGF_COMPAT_NO_TRAILING_SYNC
    if (rc == SVGF_OK) rc = svgf_sync(g_ctx);                       // the reference returns after a device sync (src/denoise.cu:401)
#else
    // -DSVGF_COMPAT_NO_TRAILING_SYNC: return with the frame enqueued on the legacy default stream.  Everything pathtrace() does next
    // (sendTwoImagesToPBO on the same stream, src/pathtrace.cu:446; the blocking cudaMemcpy of the image, :449) is ordered behind it by
    // the stream, so results are unchanged; what goes away is one host round trip per frame (the PBO kernel is already queued when
    // the last level ends: 0.289 -> 0.278 ms per 1080p frame for denoise + pack kernel + the caller's own sync, `latency_caller_ms` of
    // bench.py's line, profiles/r06_bench_line_driver_cmd.json).
#endif
    if (rc != SVGF_OK) fprintf(stderr, "denoise: svgf error %d: %s\n", rc, svgf_last_error(g_ctx));
}
