// examples/cadence.cpp — a renderer's frame loop at a fixed rate (60 Hz, 144 Hz, ...) with the GPU idle between frames, in C++ through
// the C ABI, and WHY its frames run slower than frames enqueued back to back: probes of the shader clock and of the memory side.
//
// Per tick: producer (svgf_synth_render, the path tracer's role) -> svgf_denoise -> svgf_display_pack on one stream, then the stream is
// waited for (a renderer's swap) and the host sleeps until the next tick.  Around the denoise sit two HIP events (its GPU time) and two
// CLOCK PROBES: one-wave kernels that run a chain of 16 384 dependent v_fma_f32 — a fixed number of shader cycles — and time it with the
// constant 100 MHz counter (s_memrealtime): wall time per unit of shader work right before and right after the frame, i.e. the
// shader clock the waves really get.  (The clock sysfs / rocm-smi report reads 2.39-2.41 GHz in every state, DESIGN.md 6.2, and
// s_memtime itself ticks at a constant ~2.44 GHz on this part, whatever the shader clock is.)  A 128 MB device-to-device copy in front of
// every frame is the memory side's yardstick (HBM / fabric clocks).
//
//   examples/cadence [hz=60] [frames=120] [width=1920] [height=1080]
// prints the median ms per denoise and the probes' ns per dependent FMA at that rate, then the same for frames enqueued back to back.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "svgf.h"

#define HIP_OK(x) do { if ((x) != hipSuccess) { fprintf(stderr, "%s failed\n", #x); return 1; } } while (0)
#define SVGF_OKAY(x) do { int rc__ = (x); if (rc__ != SVGF_OK) { fprintf(stderr, "%s failed (%d): %s\n", #x, rc__, svgf_last_error(ctx)); return 1; } } while (0)

// A chain of `n` DEPENDENT v_fma_f32 in one wave takes n x (a fixed number of shader cycles): out[0] = s_memtime ticks, out[1] = ticks of
// the constant 100 MHz counter (s_memrealtime) the chain took.  out[1] is the shader clock's own yardstick: wall time per fixed
// amount of shader work, whatever any counter claims to run at.
__global__ void k_clock_probe(unsigned long long *out, int n, float seed)
{
    float x = seed + threadIdx.x * 1e-9f;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
#pragma unroll 16
    for (int i = 0; i < n; i++) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);
    asm volatile("" :: "v"(x));
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
    if (x == 12345.678f) out[2] = 0;      // (keeps the chain alive)
}

static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; }

int main(int argc, char **argv)
{
    const double hz = argc > 1 ? atof(argv[1]) : 60.0;
    const int frames = argc > 2 ? atoi(argv[2]) : 120, W = argc > 3 ? atoi(argv[3]) : 1920, H = argc > 4 ? atoi(argv[4]) : 1080;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) { fprintf(stderr, "cadence: no HIP device (the library has no CPU path)\n"); return 2; }
    if (hz <= 0 || frames < 8 || W <= 0 || H <= 0) { fprintf(stderr, "usage: cadence [hz > 0] [frames >= 8] [width] [height]\n"); return 2; }
    HIP_OK(hipSetDevice(0));
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0) != hipSuccess || khz <= 0) khz = 100000;
    const size_t n = (size_t)W * H;
    svgf_ctx *ctx = nullptr;
    SVGF_OKAY(svgf_create(0, W, H, &ctx));
    float *rgb, *out;
    void *gbuf, *pbo;
    unsigned long long *probe;
    hipStream_t s;
    HIP_OK(hipMalloc((void **)&rgb, n * 12)); HIP_OK(hipMalloc((void **)&out, n * 12)); HIP_OK(hipMalloc(&gbuf, n * sizeof(SvgfGBufferTexel)));
    HIP_OK(hipMalloc(&pbo, n * 8)); HIP_OK(hipMalloc((void **)&probe, (size_t)frames * 4 * sizeof(unsigned long long)));
    HIP_OK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    // a third probe runs BESIDE the frame: the same FMA chain, ten times as long (~0.45 ms), on a second stream, launched right before
    // the denoise — the shader clock while the chip is under the frame's load (one wave with a handful of registers finds a slot
    // beside the denoiser's workgroups)
    hipStream_t s2;
    HIP_OK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    unsigned long long *probe2;
    HIP_OK(hipMalloc((void **)&probe2, (size_t)frames * 4 * sizeof(unsigned long long)));
    // the memory side's yardstick: a 128 MB device-to-device copy in front of every frame, timed by two more events
    const size_t copy_bytes = 128u << 20;
    char *cp_a, *cp_b;
    HIP_OK(hipMalloc((void **)&cp_a, copy_bytes)); HIP_OK(hipMalloc((void **)&cp_b, copy_bytes));
    HIP_OK(hipMemset(cp_a, 1, copy_bytes));
    std::vector<hipEvent_t> evc((size_t)frames * 2);
    for (auto &e : evc) HIP_OK(hipEventCreate(&e));
    std::vector<hipEvent_t> ev((size_t)frames * 2);
    for (auto &e : ev) HIP_OK(hipEventCreate(&e));
    SvgfParams p;
    svgf_params_default(&p);
    p.temporal_enable = 1; p.spatial_enable = 1;
    const int chain = 16384;      // dependent FMAs per probe (~30 us)

    for (int mode = 0; mode < 2; mode++) {      // 0: one frame per tick, GPU idle in between; 1: the same frames back to back
        if (mode == 1) {                        // into the sustained state first
            for (int f = 0; f < 1500; f++) {
                SvgfCamera cam; SvgfSynthParams sp = { f, 7, 0.6f, 0.02f, { 0.0f, 0.0f } };
                SVGF_OKAY(svgf_synth_camera(0, 0, W, H, &cam, sp.pixel_length));
                SVGF_OKAY(svgf_denoise(ctx, out, rgb, gbuf, &cam, &p, s));
            }
        }
        auto next = std::chrono::steady_clock::now();
        for (int f = 0; f < frames; f++) {
            if (mode == 0) { std::this_thread::sleep_until(next); next += std::chrono::duration_cast<std::chrono::steady_clock::duration>(std::chrono::duration<double>(1.0 / hz)); }
            SvgfCamera cam;
            SvgfSynthParams sp = { f, 7, 0.6f, 0.02f, { 0.0f, 0.0f } };
            SVGF_OKAY(svgf_synth_camera(0, /*moving=*/0, W, H, &cam, sp.pixel_length));
            SVGF_OKAY(svgf_synth_render(0, rgb, gbuf, W, H, &cam, &sp, s));
            hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, s, probe + 4 * f, chain, 1.0f);
            HIP_OK(hipEventRecord(evc[2 * f], s));
            HIP_OK(hipMemcpyAsync(cp_b, cp_a, copy_bytes, hipMemcpyDeviceToDevice, s));
            HIP_OK(hipEventRecord(evc[2 * f + 1], s));
            HIP_OK(hipEventRecord(ev[2 * f], s));
            HIP_OK(hipStreamWaitEvent(s2, ev[2 * f], 0));
            hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, s2, probe2 + 4 * f, 10 * chain, 1.0f);
            SVGF_OKAY(svgf_denoise(ctx, out, rgb, gbuf, &cam, &p, s));
            HIP_OK(hipEventRecord(ev[2 * f + 1], s));
            hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, s, probe + 4 * f + 2, chain, 1.0f);
            SVGF_OKAY(svgf_display_pack(0, pbo, rgb, out, W, H, s));
            if (mode == 0) { SVGF_OKAY(svgf_sync_stream(ctx, s)); SVGF_OKAY(svgf_sync_stream(ctx, s2)); }      // the frame is on the screen before the next tick
        }
        SVGF_OKAY(svgf_sync_stream(ctx, s));
        SVGF_OKAY(svgf_sync_stream(ctx, s2));
        std::vector<unsigned long long> h((size_t)frames * 4), h2((size_t)frames * 4);
        HIP_OK(hipMemcpy(h.data(), probe, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(h2.data(), probe2, h2.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        std::vector<double> ms, ns_before, ns_after, cyc, tbs, ns_during;
        for (int f = 4; f < frames; f++) {
            float t = 0.0f;
            HIP_OK(hipEventElapsedTime(&t, ev[2 * f], ev[2 * f + 1]));
            ms.push_back(t);
            HIP_OK(hipEventElapsedTime(&t, evc[2 * f], evc[2 * f + 1]));
            tbs.push_back(2.0 * copy_bytes / (t * 1e-3) / 1e12);
            ns_before.push_back((double)h[4 * f + 1] / (khz * 1e3) * 1e9 / chain);          // ns per dependent FMA
            ns_after.push_back((double)h[4 * f + 3] / (khz * 1e3) * 1e9 / chain);
            cyc.push_back((double)h[4 * f] / chain);                                          // s_memtime ticks per dependent FMA
            ns_during.push_back((double)h2[4 * f + 1] / (khz * 1e3) * 1e9 / (10.0 * chain));
        }
        if (mode == 0) printf("%.0f Hz, GPU idle between frames:  ", hz); else printf("the same frames back to back:     ");
        printf("%dx%d, %d frames: %.4f ms per svgf_denoise (median, HIP events); a dependent v_fma_f32 takes %.3f ns before the frame, %.3f ns BESIDE it, %.3f ns after it "
               "(%.2f s_memtime ticks); a 128 MB device-to-device copy in front of the frame moves %.2f TB/s\n", W, H, frames - 4, median(ms), median(ns_before),
               median(ns_during), median(ns_after), median(cyc), median(tbs));
    }
    svgf_destroy(ctx);
    return 0;
}
