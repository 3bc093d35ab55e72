// ref_driver.cpp — feeds case files to the REFERENCE denoiser (its own denoiseInit/denoise/denoiseFree, built
// from /root/reference/src/denoise.cu by the Makefile next to this file) and writes what it returns.
// Test infrastructure: produces the golden vectors under tests/golden/ref_gpu/ and times the reference's kernels
// on the same GPU.  Ours, not reference code: it only calls the reference's public entry points
// (reference src/denoise.h:6-8) and sets the globals they read (reference src/main.h:39-69).
//
// case file (little-endian):  int32 magic 0x43475653 ("SVGC"), W, H, ncalls, nframes
//   ncalls  x { int32 reset_before, frame_index, temporal, spatial; float color_alpha, moment_alpha;
//               int32 blurvariance; float sigmal, sigmax, sigman; int32 nlevel, history_level, sepcolor, addcolor,
//               view_option; float right[3], up[3], view[3], position[3]; int32 repeat_timing }
//   nframes x { float color[W*H*3]; uint8 gbuffer[W*H*52] }
// output file: ncalls x float out[W*H*3], then ncalls x float ms (wall time of the denoise() call, which ends in
//   a device synchronise, reference src/denoise.cu:401; when repeat_timing > 0 the value is the mean of that many
//   extra calls made on a scratch copy of nothing — see below — so state is not disturbed: timing calls are made
//   BEFORE the recorded call with identical inputs only when reset_before is set for the next call).
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "main.h"   // hipified copy: extern ui_* declarations, Scene, Camera, GBufferTexel, denoise.h

// ---- the globals reference denoise() reads (declared extern in main.h) ----
Scene *scene = nullptr;
float ui_sigmal, ui_sigmax, ui_sigman, ui_color_alpha, ui_moment_alpha;
int ui_atrous_nlevel, ui_history_level, ui_right_view_option;
bool ui_temporal_enable, ui_spatial_enable, ui_blurvariance, ui_sepcolor, ui_addcolor;

struct Call {
    int32_t reset_before, frame_index, temporal, spatial;
    float color_alpha, moment_alpha;
    int32_t blurvariance;
    float sigmal, sigmax, sigman;
    int32_t nlevel, history_level, sepcolor, addcolor, view_option;
    float right[3], up[3], view[3], position[3];
    int32_t repeat_timing;
};

static void die(const char *m) { fprintf(stderr, "ref_driver: %s\n", m); exit(2); }

int main(int argc, char **argv)
{
    if (argc < 3) die("usage: ref_denoise_gpu <case file> <output file>");
    FILE *f = fopen(argv[1], "rb");
    if (!f) die("cannot open case file");
    int32_t hdr[5];
    if (fread(hdr, 4, 5, f) != 5 || hdr[0] != 0x43475653) die("bad header");
    const int W = hdr[1], H = hdr[2], ncalls = hdr[3], nframes = hdr[4];
    const size_t n = (size_t)W * H;
    if (sizeof(GBufferTexel) != 52 || sizeof(glm::vec3) != 12) die("layout mismatch");
    std::vector<Call> calls(ncalls);
    if (fread(calls.data(), sizeof(Call), ncalls, f) != (size_t)ncalls) die("short read (calls)");
    std::vector<std::vector<float>> colors(nframes, std::vector<float>(n * 3));
    std::vector<std::vector<unsigned char>> gbufs(nframes, std::vector<unsigned char>(n * 52));
    for (int k = 0; k < nframes; k++) {
        if (fread(colors[k].data(), 4, n * 3, f) != n * 3) die("short read (color)");
        if (fread(gbufs[k].data(), 1, n * 52, f) != n * 52) die("short read (gbuffer)");
    }
    fclose(f);

    // A Scene whose only live member is state.camera: the denoiser reads nothing else (reference
    // src/denoise.cu:33-34,343-346,350-351).  Storage is zeroed, never constructed, never destroyed.
    scene = (Scene *)calloc(1, sizeof(Scene));
    Camera &cam = scene->state.camera;
    cam.resolution = glm::ivec2(W, H);

    glm::vec3 *d_in = nullptr, *d_out = nullptr;
    GBufferTexel *d_g = nullptr;
    if (hipMalloc(&d_in, n * 12) != hipSuccess || hipMalloc(&d_out, n * 12) != hipSuccess ||
        hipMalloc(&d_g, n * 52) != hipSuccess) die("hipMalloc");

    FILE *o = fopen(argv[2], "wb");
    if (!o) die("cannot open output file");
    std::vector<float> out(n * 3), ms(ncalls, 0.0f);
    bool inited = false;
    for (int c = 0; c < ncalls; c++) {
        const Call &k = calls[c];
        ui_temporal_enable = k.temporal; ui_spatial_enable = k.spatial;
        ui_color_alpha = k.color_alpha; ui_moment_alpha = k.moment_alpha;
        ui_blurvariance = k.blurvariance; ui_sigmal = k.sigmal; ui_sigmax = k.sigmax; ui_sigman = k.sigman;
        ui_atrous_nlevel = k.nlevel; ui_history_level = k.history_level;
        ui_sepcolor = k.sepcolor; ui_addcolor = k.addcolor; ui_right_view_option = k.view_option;
        cam.right = glm::vec3(k.right[0], k.right[1], k.right[2]);
        cam.up = glm::vec3(k.up[0], k.up[1], k.up[2]);
        cam.view = glm::vec3(k.view[0], k.view[1], k.view[2]);
        cam.position = glm::vec3(k.position[0], k.position[1], k.position[2]);
        if (k.reset_before || !inited) {          // what runCuda() does on reset (reference src/main.cpp:194-201)
            if (inited) denoiseFree();
            denoiseInit(scene);
            inited = true;
        }
        if (k.frame_index < 0 || k.frame_index >= nframes) die("bad frame index");
        hipMemcpy(d_in, colors[k.frame_index].data(), n * 12, hipMemcpyHostToDevice);
        hipMemcpy(d_g, gbufs[k.frame_index].data(), n * 52, hipMemcpyHostToDevice);
        hipMemset(d_out, 0, n * 12);
        hipDeviceSynchronize();
        auto t0 = std::chrono::high_resolution_clock::now();
        denoise(d_out, d_in, d_g);                // synchronises at its end (reference src/denoise.cu:401)
        auto t1 = std::chrono::high_resolution_clock::now();
        ms[c] = std::chrono::duration<float, std::milli>(t1 - t0).count();
        if (hipMemcpy(out.data(), d_out, n * 12, hipMemcpyDeviceToHost) != hipSuccess) die("D2H");
        fwrite(out.data(), 4, n * 3, o);
    }
    fwrite(ms.data(), 4, ncalls, o);
    fclose(o);
    denoiseFree();
    hipFree(d_in); hipFree(d_out); hipFree(d_g);
    printf("ref_driver: %d calls at %dx%d done\n", ncalls, W, H);
    return 0;
}
