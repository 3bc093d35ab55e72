// ref_pt_capture.cpp — runs the REFERENCE's own path tracer (its pathtrace.cu / scene.cpp / bvhtree.cpp ..., built from
// /root/reference/src by the Makefile next to this file) on one of the reference's scenes and captures what it hands to
// its denoiser: the 1-spp colour and the G-buffer of every frame, plus the camera.  Test infrastructure: produces the
// real-scene fixtures under tests/golden/ref_scenes/ (SURVEY.md §8c fixtures 1-3).  Ours, not reference code:
//   * it calls the reference's public entry points only (src/pathtrace.h:6-8, class Scene src/scene.h:16-56);
//   * the three functions of src/denoise.h are defined HERE instead of linking src/denoise.cu: pathtrace() calls
//     denoise(out, in, gbuffer) (src/pathtrace.cu:436-438) and this implementation copies `in` and `gbuffer` to the host
//     and writes a deterministic test pattern to `out`, which the reference's own sendTwoImagesToPBO (src/pathtrace.cu:46-78)
//     then packs — that byte image is the golden for the display step;
//   * the camera handling restates resetCamera() / runCuda() (src/main.cpp:77-101,154-190) and the resolution-dependent
//     part of Scene::loadCamera (src/scene.cpp:159-166), because the fixtures use other resolutions than the scene files.
// The captured frames are written as a case file of ref_driver.cpp's format, so the reference's own denoiser
// (ref_denoise_gpu, built from src/denoise.cu) produces the expected outputs from exactly these inputs.
//
// usage: ref_pathtrace_capture <scene.txt> <W> <H> <nframes> <moving 0|1> <sepcolor 0|1> <out prefix>
//   (run with the scene's directory layout reachable as ../scenes/ — scene.cpp resolves models and textures that way)
// writes <prefix>.case (ref_driver format, one call per frame: temporal + spatial on, reference defaults),
//        <prefix>.pbo  (nframes x uchar4[2W x H]), <prefix>.pattern (nframes x float[W*H*3], what denoise() wrote to `out`)
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "main.h"   // hipified copy: extern ui_* declarations, Scene, Camera, GBufferTexel, pathtrace.h, denoise.h

// ---- globals declared extern in main.h that the path tracer reads (defaults of src/main.cpp:37-75) ----
Scene *scene = nullptr;
int frame = 0;
int ui_tracedepth = 4;
bool ui_shadowray = true, ui_reducevar = true, ui_denoise_enable = true, ui_sepcolor = false;
float ui_sintensity = 2.7f, ui_lightradius = 1.4f;

static std::vector<std::vector<float>> g_colors, g_patterns;
static std::vector<std::vector<unsigned char>> g_gbufs;
static int g_W = 0, g_H = 0;

void denoiseInit(Scene *) {}
void denoiseFree() {}
// The renderer's call site: src/pathtrace.cu:436-438.  Capture the inputs, hand back a pattern that exercises the display
// pack: in-range values, byte boundaries, negatives, > 1, NaN and +-inf.
void denoise(glm::vec3 *output, glm::vec3 *input, GBufferTexel *gbuffer)
{
    const size_t n = (size_t)g_W * g_H;
    std::vector<float> c(n * 3);
    std::vector<unsigned char> g(n * sizeof(GBufferTexel));
    if (hipMemcpy(c.data(), input, n * 12, hipMemcpyDeviceToHost) != hipSuccess) { fprintf(stderr, "capture: D2H colour\n"); exit(2); }
    if (hipMemcpy(g.data(), gbuffer, n * sizeof(GBufferTexel), hipMemcpyDeviceToHost) != hipSuccess) { fprintf(stderr, "capture: D2H gbuffer\n"); exit(2); }
    std::vector<float> pat(n * 3);
    const float specials[12] = { 0.0f, 1.0f, -0.25f, 1.5f, 0.5f, 1.0f / 255.0f, 254.999f / 255.0f, 255.0f / 255.0f, NAN, INFINITY, -INFINITY, 1e-30f };
    for (size_t k = 0; k < n * 3; k++) {
        const uint32_t h = (uint32_t)(k * 2654435761u) ^ (uint32_t)(g_colors.size() * 40503u);
        pat[k] = (k % 97 < 12) ? specials[k % 12] : (float)(h >> 8) / 16777216.0f * 1.2f - 0.1f;
    }
    if (hipMemcpy(output, pat.data(), n * 12, hipMemcpyHostToDevice) != hipSuccess) { fprintf(stderr, "capture: H2D pattern\n"); exit(2); }
    g_colors.push_back(std::move(c)); g_gbufs.push_back(std::move(g)); g_patterns.push_back(std::move(pat));
}

struct Call {            // ref_driver.cpp's call record
    int32_t reset_before, frame_index, temporal, spatial;
    float color_alpha, moment_alpha;
    int32_t blurvariance;
    float sigmal, sigmax, sigman;
    int32_t nlevel, history_level, sepcolor, addcolor, view_option;
    float right[3], up[3], view[3], position[3];
    int32_t repeat_timing;
};

int main(int argc, char **argv)
{
    if (argc < 8) { fprintf(stderr, "usage: %s <scene.txt> <W> <H> <nframes> <moving> <sepcolor> <out prefix>\n", argv[0]); return 2; }
    const int W = atoi(argv[2]), H = atoi(argv[3]), nframes = atoi(argv[4]);
    const bool moving = atoi(argv[5]) != 0;
    ui_sepcolor = atoi(argv[6]) != 0;
    const std::string prefix = argv[7];
    if (sizeof(GBufferTexel) != 52 || sizeof(glm::vec3) != 12) { fprintf(stderr, "layout mismatch\n"); return 2; }
    g_W = W; g_H = H;

    scene = new Scene(argv[1]);
    RenderState *renderState = &scene->state;
    Camera &cam = renderState->camera;
    // other resolution than the scene file's RES: redo the resolution-dependent part of Scene::loadCamera (src/scene.cpp:159-166)
    {
        const float fovy = cam.fov.y;
        cam.resolution = glm::ivec2(W, H);
        float yscaled = tan(fovy * (PI / 180));
        float xscaled = (yscaled * cam.resolution.x) / cam.resolution.y;
        float fovx = (atan(xscaled) * 180) / PI;
        cam.fov = glm::vec2(fovx, fovy);
        cam.pixelLength = glm::vec2(2 * xscaled / (float)cam.resolution.x, 2 * yscaled / (float)cam.resolution.y);
        renderState->image.resize((size_t)W * H);
        std::fill(renderState->image.begin(), renderState->image.end(), glm::vec3());
    }
    // resetCamera() (src/main.cpp:77-101)
    float zoom, theta, phi;
    glm::vec3 cameraPosition;
    {
        glm::vec3 view = cam.view;
        glm::vec3 viewXZ = glm::vec3(view.x, 0.0f, view.z);
        glm::vec3 viewZY = glm::vec3(0.0f, view.y, view.z);
        phi = glm::acos(glm::dot(glm::normalize(viewXZ), glm::vec3(0, 0, -1)));
        theta = glm::acos(glm::dot(glm::normalize(viewZY), glm::vec3(0, 1, 0)));
        zoom = glm::length(cam.position - cam.lookAt);
    }
    // camera automation speeds: the reference's defaults are 0 (static); the moving fixtures use the values SURVEY.md §8d
    // records for BASELINE configs[2]
    const float sx = moving ? 0.02f : 0.0f, sy = moving ? 0.01f : 0.0f, sz = moving ? 0.01f : 0.0f, st = moving ? 0.01f : 0.0f, sp = moving ? 0.02f : 0.0f;
    float tx = 0, ty = 0, tz = 0, tth = 0, tph = 0;

    uchar4 *pbo = nullptr;
    if (hipMalloc(&pbo, (size_t)2 * W * H * sizeof(uchar4)) != hipSuccess) { fprintf(stderr, "hipMalloc pbo\n"); return 2; }
    std::vector<std::vector<unsigned char>> pbos;
    std::vector<Call> calls;
    pathtraceInit(scene);
    for (int f = 0; f < nframes; f++) {
        if (moving) {                                   // runCuda(), src/main.cpp:156-169
            tx += sx; ty += sy; tz += sz; tth += st; tph += sp;
            cam.lookAt.x = 0.0f + 2.0f * sinf(tx);
            cam.lookAt.y = 5.0f + 1.0f * sinf(ty);
            cam.lookAt.z = 0.0f + 1.5f * sinf(tz);
            theta = PI * 0.5f + PI / 18 * sinf(tth);
            phi = PI * 0.0f + PI / 12 * sinf(tph);
        }
        if (moving || f == 0) {                         // camchanged branch, src/main.cpp:171-190
            cameraPosition.x = zoom * sin(phi) * sin(theta);
            cameraPosition.y = zoom * cos(theta);
            cameraPosition.z = zoom * cos(phi) * sin(theta);
            cam.view = -glm::normalize(cameraPosition);
            glm::vec3 v = cam.view;
            glm::vec3 u = glm::vec3(0, 1, 0);
            glm::vec3 r = glm::cross(v, u);
            cam.up = glm::cross(r, v);
            cam.right = r;
            cam.position = cameraPosition;
            cameraPosition += cam.lookAt;
            cam.position = cameraPosition;
        }
        pathtrace(pbo, frame++);                        // src/main.cpp:209
        std::vector<unsigned char> pb((size_t)2 * W * H * 4);
        if (hipMemcpy(pb.data(), pbo, pb.size(), hipMemcpyDeviceToHost) != hipSuccess) { fprintf(stderr, "D2H pbo\n"); return 2; }
        pbos.push_back(std::move(pb));
        Call k;
        memset(&k, 0, sizeof(k));
        k.reset_before = (f == 0); k.frame_index = f; k.temporal = 1; k.spatial = 1;
        k.color_alpha = 0.2f; k.moment_alpha = 0.2f; k.blurvariance = 1; k.sigmal = 0.45f; k.sigmax = 0.35f; k.sigman = 0.2f;
        k.nlevel = 5; k.history_level = 1; k.sepcolor = ui_sepcolor; k.addcolor = ui_sepcolor; k.view_option = 0;
        for (int j = 0; j < 3; j++) { k.right[j] = cam.right[j]; k.up[j] = cam.up[j]; k.view[j] = cam.view[j]; k.position[j] = cam.position[j]; }
        calls.push_back(k);
    }
    pathtraceFree();
    if ((int)g_colors.size() != nframes) { fprintf(stderr, "capture: denoise() was called %zu times for %d frames\n", g_colors.size(), nframes); return 2; }

    FILE *o = fopen((prefix + ".case").c_str(), "wb");
    if (!o) { fprintf(stderr, "cannot write %s.case\n", prefix.c_str()); return 2; }
    const int32_t hdr[5] = { 0x43475653, W, H, nframes, nframes };
    fwrite(hdr, 4, 5, o);
    fwrite(calls.data(), sizeof(Call), calls.size(), o);
    for (int f = 0; f < nframes; f++) { fwrite(g_colors[f].data(), 4, g_colors[f].size(), o); fwrite(g_gbufs[f].data(), 1, g_gbufs[f].size(), o); }
    fclose(o);
    o = fopen((prefix + ".pbo").c_str(), "wb");
    for (int f = 0; f < nframes; f++) fwrite(pbos[f].data(), 1, pbos[f].size(), o);
    fclose(o);
    o = fopen((prefix + ".pattern").c_str(), "wb");
    for (int f = 0; f < nframes; f++) fwrite(g_patterns[f].data(), 4, g_patterns[f].size(), o);
    fclose(o);
    printf("ref_pathtrace_capture: %s %dx%d, %d frames (%s camera, sepcolor %d) -> %s.{case,pbo,pattern}\n", argv[1], W, H, nframes,
           moving ? "moving" : "static", (int)ui_sepcolor, prefix.c_str());
    return 0;
}
