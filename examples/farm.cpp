// examples/farm.cpp — N independent frame sequences on N GPUs from ONE C++ process, through the C ABI (include/svgf.h).
//
// What a renderer-side farm looks like (north_star: "independent frames/tiles are farmed across the 8 GPUs of one node as an
// embarrassingly-parallel batch (no RCCL collectives)", host code C++): one host thread, one svgf_ctx and two streams per GPU (every
// context runs its own sequence on the frame pipeline: frames in turn on its two streams, each with its own input / output buffers,
// SvgfParams::inputs_ready = 2 — examples/pipeline.cpp; `pipelined` = 0 on the command line orders every frame on one stream); the
// lifecycle per context is the reference's (src/main.cpp:192-201: denoiseFree + denoiseInit on reset, denoise per frame,
// denoiseFree at exit), handle-based instead of global.  Inputs come from the library's device-side producer (svgf_synth_render:
// the path tracer's role), so nothing crosses PCIe in the loop.  With fewer GPUs than contexts the contexts share devices
// (context k on device k mod n_devices) — which is how tests/test_dropin_gpu.py runs it on a one-GPU box.
//
//   hipcc --offload-arch=gfx950 -O2 -I include examples/farm.cpp -L cuda-path-tracer-denoising_amd -lsvgf_hip \
//         -Wl,-rpath,$PWD/cuda-path-tracer-denoising_amd -o examples/farm
//   examples/farm [contexts=8] [frames=64] [width=3840] [height=2160] [pipelined=1]
//
// Prints one line per context (device, frames, ms per frame, a checksum of the last output) and the aggregate Mpixels/s =
// pixels of all contexts / the time from the first context's first frame to the last context's last — what bench.py's barrier-bracketed
// MAX-time / SUM-pixels reduction measures across processes.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "svgf.h"

struct Result { int device = -1; int pipelined = 0; int rc = 0; double seconds = 0.0; double t_start = 0.0, t_end = 0.0; double checksum = 0.0; char err[256] = ""; };


// Two streams whose kernels really run side by side.  The HIP runtime spreads a process's streams over its hardware queues
// (GPU_MAX_HW_QUEUES, default 4) in an order of its own, and two streams created one after the other may land on ONE queue; frames in
// turn on such a pair gain nothing and pay a few per cent.  The library's probe (svgf_streams_overlap: two 200 us kernels, side by side
// or one after the other?) says; on a miss another stream is created and tried, the rejected ones are destroyed at the end.
// Returns 1 with st[0], st[1] set, or 0 with only st[0] (order the frames on it).
static int two_overlapping_streams(int device, hipStream_t st[2])
{
    st[0] = st[1] = nullptr;
    if (hipStreamCreateWithFlags(&st[0], hipStreamNonBlocking) != hipSuccess) return 0;
    hipStream_t rejected[6];
    int n_rejected = 0, found = 0;
    for (int k = 0; k < 6 && !found; k++) {
        hipStream_t cand = nullptr;
        if (hipStreamCreateWithFlags(&cand, hipStreamNonBlocking) != hipSuccess) break;
        if (svgf_streams_overlap(device, st[0], cand) == 1) { st[1] = cand; found = 1; }
        else rejected[n_rejected++] = cand;
    }
    for (int k = 0; k < n_rejected; k++) (void)hipStreamDestroy(rejected[k]);
    return found;
}

static void run_context(int k, int device, int W, int H, int frames, int pipelined, Result *r)
{
    r->device = device;
    const size_t n = (size_t)W * H;
    svgf_ctx *ctx = nullptr;
    float *rgb[2] = { nullptr, nullptr }, *out[2] = { nullptr, nullptr };
    void *gbuf[2] = { nullptr, nullptr };
    hipStream_t st[2] = { nullptr, nullptr };
    auto fail = [&](const char *what, int rc) {
        r->rc = rc ? rc : -1;
        snprintf(r->err, sizeof(r->err), "%s: %s", what, ctx ? svgf_last_error(ctx) : svgf_last_error(nullptr));
    };
    do {
        if (hipSetDevice(device) != hipSuccess) { fail("hipSetDevice", SVGF_ERR_NO_DEVICE); break; }
        bool ok = true;
        for (int q = 0; q < 2 && ok; q++)
            ok = hipMalloc((void **)&rgb[q], n * 12) == hipSuccess && hipMalloc((void **)&out[q], n * 12) == hipSuccess &&
                 hipMalloc(&gbuf[q], n * sizeof(SvgfGBufferTexel)) == hipSuccess;
        if (!ok) { fail("device buffers", SVGF_ERR_OOM); break; }
        // two streams in turn only pay when the runtime has put them on different hardware queues: pick such a pair FIRST (the library's
        // probe), and create a plain context whose frames are ordered on one stream when there is none (an ordered frame of a
        // pipelined context would still pay the pipeline's events: 2 %)
        const int overlap = two_overlapping_streams(device, st);
        if (!st[0]) { fail("hipStreamCreate", SVGF_ERR_HIP); break; }
        if (!overlap) { pipelined = 0; st[1] = nullptr; }
        r->pipelined = pipelined;
        if (int rc = svgf_create_ex(device, W, H, pipelined ? SVGF_CREATE_PIPELINED : 0u, &ctx)) { fail("svgf_create_ex", rc); break; }
        SvgfParams p;
        svgf_params_default(&p);
        p.temporal_enable = 1; p.spatial_enable = 1;          // full SVGF: temporal accumulation + 5 a-trous levels
        p.inputs_ready = pipelined ? 2 : 0;                   // the pipeline, every frame ordered behind the stream it is given to
        const auto t0 = std::chrono::steady_clock::now();
        for (int f = 0; f < frames && r->rc == 0; f++) {
            const int q = f & 1;
            hipStream_t s = pipelined ? st[q] : st[0];
            SvgfCamera cam;
            SvgfSynthParams sp = { f, 1000 + k, 0.6f, 0.02f, { 0.0f, 0.0f } };      // every context its own sequence (seed)
            if (int rc = svgf_synth_camera(f, /*moving=*/1, W, H, &cam, sp.pixel_length)) { fail("svgf_synth_camera", rc); break; }
            if (int rc = svgf_synth_render(device, rgb[q], gbuf[q], W, H, &cam, &sp, s)) { fail("svgf_synth_render", rc); break; }
            if (int rc = svgf_denoise(ctx, out[q], rgb[q], gbuf[q], &cam, &p, s)) { fail("svgf_denoise", rc); break; }   // asynchronous on s
        }
        if (r->rc) break;
        if (int rc = svgf_sync_stream(ctx, st[0])) { fail("svgf_sync_stream", rc); break; }      // this context's frames only
        if (st[1]) { if (int rc = svgf_sync_stream(ctx, st[1])) { fail("svgf_sync_stream", rc); break; } }
        r->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        r->t_start = std::chrono::duration<double>(t0.time_since_epoch()).count(); r->t_end = r->t_start + r->seconds;
        std::vector<float> h(3 * n);          // the last output; checksum = sum of every 61st value over the whole image
        if (hipMemcpy(h.data(), out[(frames - 1) & 1], h.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) { fail("hipMemcpy", SVGF_ERR_HIP); break; }
        for (size_t i = 0; i < h.size(); i += 61) r->checksum += h[i];
    } while (false);
    for (int q = 0; q < 2; q++) {
        if (st[q]) (void)hipStreamDestroy(st[q]);
        if (rgb[q]) (void)hipFree(rgb[q]);
        if (out[q]) (void)hipFree(out[q]);
        if (gbuf[q]) (void)hipFree(gbuf[q]);
    }
    svgf_destroy(ctx);          // NULL-safe, like denoiseFree
}

int main(int argc, char **argv)
{
    const int n_ctx = argc > 1 ? atoi(argv[1]) : 8, frames = argc > 2 ? atoi(argv[2]) : 64;
    const int W = argc > 3 ? atoi(argv[3]) : 3840, H = argc > 4 ? atoi(argv[4]) : 2160, pipelined = argc > 5 ? atoi(argv[5]) : 1;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) { fprintf(stderr, "farm: no HIP device (the library has no CPU path)\n"); return 2; }
    if (n_ctx <= 0 || frames <= 0 || W <= 0 || H <= 0) { fprintf(stderr, "usage: farm [contexts] [frames] [width] [height]\n"); return 2; }
    std::vector<Result> res(n_ctx);
    std::vector<std::thread> th;
    for (int k = 0; k < n_ctx; k++) th.emplace_back(run_context, k, k % n_dev, W, H, frames, pipelined, &res[k]);
    for (auto &t : th) t.join();
    double first_start = 1e300, last_end = 0.0;      // the contexts set themselves up (stream probe, allocation) at different speeds: wall time of the farm
    int bad = 0;
    for (int k = 0; k < n_ctx; k++) {
        if (res[k].rc) { bad++; printf("context %d device %d FAILED rc %d: %s\n", k, res[k].device, res[k].rc, res[k].err); continue; }
        printf("context %d device %d frames %d ms_per_frame %.4f checksum %.6f%s\n", k, res[k].device, frames, res[k].seconds / frames * 1e3, res[k].checksum,
               (pipelined && !res[k].pipelined) ? "  [no two streams on different hardware queues: frames ordered on one]" : "");
        if (res[k].t_start < first_start) first_start = res[k].t_start;
        if (res[k].t_end > last_end) last_end = res[k].t_end;
    }
    if (bad) return 1;
    printf("farm: %d contexts on %d device(s), %dx%d, %d frames each, %s: %.1f Mpixels/s aggregate (producer + denoiser)\n", n_ctx, n_dev, W, H, frames,
           pipelined ? "frames of a context in turn on two streams (pipeline)" : "frames of a context ordered on one stream",
           (double)n_ctx * frames * W * H / (last_end - first_start) / 1e6);
    return 0;
}
