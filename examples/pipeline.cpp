// examples/pipeline.cpp — a renderer's frame loop on the frame pipeline (SvgfParams::inputs_ready, svgf_create_ex; include/svgf.h ABI 0.9), in C++
// through the C ABI.
//
// The reference renders a frame and denoises it in turn, with a device synchronisation at the end of denoise() (src/pathtrace.cu:436-438,
// src/denoise.cu:401).  A renderer that double-buffers its frame can let three things overlap on one GPU instead:
//   * the PRODUCER of frame n+1 (the path tracer's role; here the library's device-side producer, svgf_synth_render),
//   * levels 2-5 of frame n,
//   * the temporal pass + level 1 of frame n+1 (which need of frame n only the level that feeds the colour history).
// All it takes is TWO streams used in turn — even frames on one, odd frames on the other, each with its own input and output
// buffers — and SvgfParams::inputs_ready = 2 ("pipeline, ordered behind the stream I pass"): a frame is rendered and denoised on
// its stream in stream order, what reads its output goes behind the call on the same stream, and the stream only ever waits for the
// frames that were given to it.  No events, no host synchronisation, no promise about buffers: plain stream semantics.  (With ONE
// stream the same overlap needs the promise inputs_ready = 1 — the inputs are complete at call time — which is what
// bench.py, whose inputs are resident, makes.)
//
//   hipcc --offload-arch=gfx950 -O2 -I include examples/pipeline.cpp -L cuda-path-tracer-denoising_amd -lsvgf_hip \
//         -Wl,-rpath,$PWD/cuda-path-tracer-denoising_amd -o examples/pipeline
//   examples/pipeline [frames=400] [width=1920] [height=1080]
//
// Runs the same sequence twice — in turn on one stream (the reference's order) and pipelined on two — prints ms per frame of both and
// checks that the last two outputs are bit-identical.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "svgf.h"

#define HIP_OK(x) do { if ((x) != hipSuccess) { fprintf(stderr, "%s failed\n", #x); return 1; } } while (0)
#define SVGF_OKAY(x) do { int rc__ = (x); if (rc__ != SVGF_OK) { fprintf(stderr, "%s failed (%d): %s\n", #x, rc__, svgf_last_error(ctx)); return 1; } } while (0)


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

static int run(bool pipelined, int frames, int W, int H, std::vector<float> (&last)[2], double *ms_per_frame)
{
    const size_t n = (size_t)W * H;
    svgf_ctx *ctx = nullptr;
    SVGF_OKAY(svgf_create_ex(0, W, H, pipelined ? SVGF_CREATE_PIPELINED : 0u, &ctx));      // the second plane set exists before the first frame
    float *rgb[2], *out[2];
    void *gbuf[2];
    hipStream_t st[2];
    for (int k = 0; k < 2; k++) {
        HIP_OK(hipMalloc((void **)&rgb[k], n * 12)); HIP_OK(hipMalloc((void **)&out[k], n * 12)); HIP_OK(hipMalloc(&gbuf[k], n * sizeof(SvgfGBufferTexel)));
    }
    const int overlap = two_overlapping_streams(0, st);
    if (!st[0]) { fprintf(stderr, "hipStreamCreate failed\n"); return 1; }
    if (!overlap) { if (pipelined) fprintf(stderr, "pipeline: no two streams on different hardware queues (GPU_MAX_HW_QUEUES?): frames in turn on one stream\n"); st[1] = st[0]; }
    SvgfParams p;
    svgf_params_default(&p);
    p.temporal_enable = 1; p.spatial_enable = 1;          // full SVGF, the reference's defaults otherwise (history_level 1)
    p.inputs_ready = (pipelined && overlap) ? 2 : 0;

    const auto t0 = std::chrono::steady_clock::now();
    for (int f = 0; f < frames; f++) {
        const int q = f & 1;
        hipStream_t s = pipelined ? st[q] : st[0];          // in turn: everything on one stream, the reference's order
        SvgfCamera cam;
        SvgfSynthParams sp = { f, 7, 0.6f, 0.02f, { 0.0f, 0.0f } };
        SVGF_OKAY(svgf_synth_camera(f, /*moving=*/1, W, H, &cam, sp.pixel_length));
        SVGF_OKAY(svgf_synth_render(0, rgb[q], gbuf[q], W, H, &cam, &sp, s));          // the path tracer's role: writes this frame's input set
        SVGF_OKAY(svgf_denoise(ctx, out[q], rgb[q], gbuf[q], &cam, &p, s));             // returns at once; `s` has waited for the frame's end
        /* ... a display pass reading out[q] would be enqueued on s here ... */
    }
    SVGF_OKAY(svgf_sync_stream(ctx, st[0]));
    SVGF_OKAY(svgf_sync_stream(ctx, st[1]));
    *ms_per_frame = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / frames * 1e3;
    for (int k = 0; k < 2; k++) {
        last[k].resize(3 * n);
        HIP_OK(hipMemcpy(last[k].data(), out[k], 3 * n * sizeof(float), hipMemcpyDeviceToHost));
    }
    printf("%-9s %dx%d, %d frames (producer + denoiser): %.4f ms per frame = %.0f Mpixels/s%s\n", pipelined ? "pipelined" : "in turn", W, H, frames,
           *ms_per_frame, (double)W * H / *ms_per_frame / 1e3, svgf_is_pipelined(ctx) ? "  [context pipelined]" : "");
    svgf_destroy(ctx);
    for (int k = 0; k < 2; k++) { (void)hipFree(rgb[k]); (void)hipFree(out[k]); (void)hipFree(gbuf[k]); }
    if (st[1] != st[0]) (void)hipStreamDestroy(st[1]);
    (void)hipStreamDestroy(st[0]);
    return 0;
}

int main(int argc, char **argv)
{
    const int frames = argc > 1 ? atoi(argv[1]) : 400, W = argc > 2 ? atoi(argv[2]) : 1920, H = argc > 3 ? atoi(argv[3]) : 1080;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) { fprintf(stderr, "pipeline: no HIP device (the library has no CPU path)\n"); return 2; }
    if (frames < 2 || W <= 0 || H <= 0) { fprintf(stderr, "usage: pipeline [frames >= 2] [width] [height]\n"); return 2; }
    HIP_OK(hipSetDevice(0));
    std::vector<float> a[2], b[2];
    double ms_a = 0.0, ms_b = 0.0;
    for (int warm = 0; warm < 2; warm++) {      // the first pass of each also warms the clocks up; the second is the one printed last
        if (run(false, frames, W, H, a, &ms_a)) return 1;
        if (run(true, frames, W, H, b, &ms_b)) return 1;
    }
    const bool same = a[0].size() == b[0].size() && memcmp(a[0].data(), b[0].data(), a[0].size() * sizeof(float)) == 0 &&
                      memcmp(a[1].data(), b[1].data(), a[1].size() * sizeof(float)) == 0;
    printf("last two outputs of the two runs bit-identical: %s; pipelined / in turn = %.3f\n", same ? "yes" : "NO", ms_b / ms_a);
    return same ? 0 : 1;
}
