// ref_host_tools.cpp — host-side golden makers built from the REFERENCE's own scene loader and image writer
// (src/scene.cpp, src/utilities.cpp, src/bvhtree.cpp, src/boundingbox.cpp, src/image.cpp, src/stb.cpp, tinyobjloader,
// compiled from /root/reference by the Makefile next to this file).  Test infrastructure, runs without a GPU.  Ours,
// not reference code: it calls class Scene (src/scene.h:16-56) and class image (src/image.h:7-19) and prints / saves
// what they produce.
//   ref_host_tools scene <scene.txt>          -> JSON on stdout: camera, materials, geoms exactly as Scene parsed them
//   ref_host_tools mesh <scene.txt> <out.bin> -> what Scene::loadMesh / Texture::Load produced (src/scene.cpp:234-330): int32 ntri, ngeoms,
//                                                ntex; ngeoms x int32 {type, T_startidx, T_endidx}; ntri x 3 vertices x float {pos3, normal3,
//                                                uv2} (world space); ntex x {int32 w, h, components; bytes}
//   ref_host_tools png <out base name>        -> <base>.png written by image::savePNG (src/image.cpp:22-39) for a fixed
//                                                37x5 pattern (in-range, byte boundaries, < 0, > 1, NaN, +-inf), after
//                                                the x-mirror of saveImage() (src/main.cpp:131-152); <base>.f32 = pattern
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "scene.h"
#include "image.h"

static void pv(const char *name, const float *v, int n, bool comma = true)
{
    printf("\"%s\": [", name);
    for (int i = 0; i < n; i++) printf("%s%.9g", i ? ", " : "", v[i]);
    printf("]%s", comma ? ", " : "");
}

int main(int argc, char **argv)
{
    if (argc >= 3 && !strcmp(argv[1], "scene")) {
        Scene *s = new Scene(argv[2]);
        const Camera &c = s->state.camera;
        printf("{\"camera\": {\"resolution\": [%d, %d], ", c.resolution.x, c.resolution.y);
        pv("position", &c.position[0], 3); pv("lookAt", &c.lookAt[0], 3); pv("view", &c.view[0], 3); pv("up", &c.up[0], 3);
        pv("fov", &c.fov[0], 2); pv("pixelLength", &c.pixelLength[0], 2, false);
        printf("},\n \"materials\": [");
        for (size_t i = 0; i < s->materials.size(); i++) {
            const Material &m = s->materials[i];
            printf("%s\n  {", i ? "," : "");
            pv("color", &m.color[0], 3); pv("specular_color", &m.specular.color[0], 3);
            printf("\"specular_exponent\": %.9g, \"hasReflective\": %.9g, \"hasRefractive\": %.9g, \"indexOfRefraction\": %.9g, \"emittance\": %.9g, \"texid\": %d}",
                   m.specular.exponent, m.hasReflective, m.hasRefractive, m.indexOfRefraction, m.emittance, m.texid);
        }
        printf("],\n \"geoms\": [");
        for (size_t i = 0; i < s->geoms.size(); i++) {
            const Geom &g = s->geoms[i];
            printf("%s\n  {\"type\": \"%s\", \"materialid\": %d, ", i ? "," : "", g.type == SPHERE ? "sphere" : g.type == CUBE ? "cube" : "mesh", g.materialid);
            pv("translation", &g.translation[0], 3); pv("rotation", &g.rotation[0], 3); pv("scale", &g.scale[0], 3);
            pv("transform", &g.transform[0][0], 16); pv("inverseTransform", &g.inverseTransform[0][0], 16); pv("invTranspose", &g.invTranspose[0][0], 16);
            printf("\"T_startidx\": %d, \"T_endidx\": %d}", g.type == MESH ? g.T_startidx : -1, g.type == MESH ? g.T_endidx : -1);
        }
        printf("],\n \"n_triangles\": %zu, \"n_textures\": %zu, \"n_lights\": %zu}\n", s->triangles.size(), s->textures.size(), s->lights.size());
        return 0;
    }
    if (argc >= 4 && !strcmp(argv[1], "mesh")) {
        Scene *s = new Scene(argv[2]);
        FILE *o = fopen(argv[3], "wb");
        if (!o) return 2;
        const int32_t hdr[3] = { (int32_t)s->triangles.size(), (int32_t)s->geoms.size(), (int32_t)s->textures.size() };
        fwrite(hdr, 4, 3, o);
        for (const Geom &g : s->geoms) {
            const int32_t r[3] = { g.type == SPHERE ? 1 : g.type == CUBE ? 0 : 2, g.type == MESH ? g.T_startidx : -1, g.type == MESH ? g.T_endidx : -1 };
            fwrite(r, 4, 3, o);
        }
        for (const Triangle &t : s->triangles)
            for (int k = 0; k < 3; k++) {
                const float v[8] = { t.verts[k].pos.x, t.verts[k].pos.y, t.verts[k].pos.z, t.verts[k].normal.x, t.verts[k].normal.y, t.verts[k].normal.z,
                                     t.verts[k].uv.x, t.verts[k].uv.y };
                fwrite(v, 4, 8, o);
            }
        for (const Texture &t : s->textures) {
            const int32_t d[3] = { t.width, t.height, t.components };
            fwrite(d, 4, 3, o);
            fwrite(t.image, 1, (size_t)t.width * t.height * t.components, o);
        }
        fclose(o);
        return 0;
    }
    if (argc >= 3 && !strcmp(argv[1], "png")) {
        const int W = 37, H = 5;
        std::vector<float> pat((size_t)W * H * 3);
        const float specials[12] = { 0.0f, 1.0f, -0.25f, 1.5f, 0.5f, 1.0f / 255.0f, 254.999f / 255.0f, 255.0f / 255.0f, NAN, INFINITY, -INFINITY, 1e-30f };
        for (size_t k = 0; k < pat.size(); k++) {
            const uint32_t h = (uint32_t)(k * 2654435761u);
            pat[k] = (k % 7 < 3) ? specials[(k / 7) % 12] : (float)(h >> 8) / 16777216.0f * 1.2f - 0.1f;
        }
        image img(W, H);
        for (int x = 0; x < W; x++)                       // saveImage(), src/main.cpp:137-143: x mirror
            for (int y = 0; y < H; y++) {
                const int index = x + (y * W);
                img.setPixel(W - 1 - x, y, glm::vec3(pat[3 * index], pat[3 * index + 1], pat[3 * index + 2]));
            }
        img.savePNG(argv[2]);
        FILE *o = fopen((std::string(argv[2]) + ".f32").c_str(), "wb");
        fwrite(pat.data(), 4, pat.size(), o);
        fclose(o);
        return 0;
    }
    fprintf(stderr, "usage: %s scene <scene.txt> | png <out base>\n", argv[0]);
    return 2;
}
