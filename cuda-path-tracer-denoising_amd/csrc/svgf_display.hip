// svgf_display.hip — the step right after denoise(): pack for display, and save.
//
//   svgf_display_pack   reference sendTwoImagesToPBO (src/pathtrace.cu:45-77): two packed-rgb float images side by
//                       side into one (2W x H) RGBA8 buffer; channel = clamp((int)(v * 255.0), 0, 255) with the product
//                       in double, alpha = 0.  Float->int conversion saturates and maps NaN to 0 (what the CUDA
//                       original does); HBM-bound: 24 B read + 8 B written per pixel.
//   svgf_save_png       reference saveImage + image::savePNG (src/main.cpp:131-152, src/image.cpp:22-39): the image is
//                       mirrored in x (the renderer's rays run right-to-left), each channel is clamp(v,0,1) * 255.f
//                       truncated to a byte, written as an 8-bit RGB PNG.  The reference encodes with stb_image_write;
//                       this writer emits stored (uncompressed) deflate blocks — same pixels, larger file, no dependency.
#include "svgf_kernels.h"
#include "../../include/svgf.h"

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

namespace {

__device__ __forceinline__ int to_byte(float v)
{
    const int i = (int)((double)v * 255.0);        // v_cvt_i32_f64: saturating, NaN -> 0
    return min(max(i, 0), 255);
}

__global__ __launch_bounds__(256) void k_display_pack(uchar4 *pbo, const float *left, const float *right, int W, int H)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= W * H) return;
    const int x = p % W, y = p / W;
    const float *l = left + 3 * (size_t)p, *r = right + 3 * (size_t)p;
    uchar4 a, b;
    a.x = (unsigned char)to_byte(l[0]); a.y = (unsigned char)to_byte(l[1]); a.z = (unsigned char)to_byte(l[2]); a.w = 0;
    b.x = (unsigned char)to_byte(r[0]); b.y = (unsigned char)to_byte(r[1]); b.z = (unsigned char)to_byte(r[2]); b.w = 0;
    pbo[(size_t)y * 2 * W + x] = a;
    pbo[(size_t)y * 2 * W + W + x] = b;
}

// ---- minimal PNG encoder: IHDR + one IDAT of stored deflate blocks + IEND ----
uint32_t crc_table[256];
bool crc_ready = false;
uint32_t crc32_update(uint32_t c, const unsigned char *buf, size_t n)
{
    if (!crc_ready) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t k = i;
            for (int j = 0; j < 8; j++) k = (k & 1) ? 0xEDB88320u ^ (k >> 1) : k >> 1;
            crc_table[i] = k;
        }
        crc_ready = true;
    }
    for (size_t i = 0; i < n; i++) c = crc_table[(c ^ buf[i]) & 0xFF] ^ (c >> 8);
    return c;
}
void put_be32(std::vector<unsigned char> &v, uint32_t x)
{
    v.push_back((unsigned char)(x >> 24)); v.push_back((unsigned char)(x >> 16));
    v.push_back((unsigned char)(x >> 8)); v.push_back((unsigned char)x);
}
bool write_chunk(FILE *f, const char tag[4], const std::vector<unsigned char> &data)
{
    std::vector<unsigned char> head;
    put_be32(head, (uint32_t)data.size());
    head.insert(head.end(), tag, tag + 4);
    uint32_t c = crc32_update(0xFFFFFFFFu, reinterpret_cast<const unsigned char *>(tag), 4);
    c = crc32_update(c, data.data(), data.size()) ^ 0xFFFFFFFFu;
    std::vector<unsigned char> tail;
    put_be32(tail, c);
    return fwrite(head.data(), 1, head.size(), f) == head.size() &&
           (data.empty() || fwrite(data.data(), 1, data.size(), f) == data.size()) &&
           fwrite(tail.data(), 1, 4, f) == 4;
}

}  // namespace

extern "C" {

int svgf_display_pack(int device, void *pbo_rgba8_dev, const void *left_rgb_dev, const void *right_rgb_dev, int width,
                      int height, void *stream)
{
    if (!pbo_rgba8_dev || !left_rgb_dev || !right_rgb_dev || width <= 0 || height <= 0) return SVGF_ERR_INVALID_ARG;
    if ((long long)width * height >= (1LL << 31) / 16) return SVGF_ERR_UNSUPPORTED;
    SvgfDeviceGuard dev_guard(device);
    if (!dev_guard.ok) return SVGF_ERR_NO_DEVICE;
    const int n = width * height;
    hipLaunchKernelGGL(k_display_pack, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<uchar4 *>(pbo_rgba8_dev), static_cast<const float *>(left_rgb_dev),
                       static_cast<const float *>(right_rgb_dev), width, height);
    return hipGetLastError() == hipSuccess ? SVGF_OK : SVGF_ERR_HIP;
}

int svgf_save_png(const char *path, const float *rgb_host, int width, int height, int mirror_x)
{
    if (!path || !rgb_host || width <= 0 || height <= 0) return SVGF_ERR_INVALID_ARG;
    // scanlines: filter byte 0 + 3*W bytes
    const size_t row = 1 + 3 * (size_t)width;
    std::vector<unsigned char> raw(row * height);
    for (int y = 0; y < height; y++) {
        unsigned char *d = raw.data() + row * y;
        *d++ = 0;
        for (int x = 0; x < width; x++) {
            const int sx = mirror_x ? (width - 1 - x) : x;            // saveImage: img.setPixel(width - 1 - x, y, pix)
            const float *p = rgb_host + 3 * ((size_t)y * width + sx);
            for (int c = 0; c < 3; c++) {
                float v = p[c];
                v = (v < 0.0f) ? 0.0f : ((v > 1.0f) ? 1.0f : v);      // glm::clamp(pix, 0, 1); NaN fails both tests
                if (!(v == v)) v = 0.0f;                              // (unsigned char)NaN is undefined in C++: define 0
                *d++ = (unsigned char)(v * 255.0f);
            }
        }
    }
    // zlib stream of stored blocks
    std::vector<unsigned char> z;
    z.push_back(0x78); z.push_back(0x01);
    uint32_t s1 = 1, s2 = 0;
    for (size_t off = 0; off < raw.size();) {
        const size_t n = (raw.size() - off < 65535) ? raw.size() - off : 65535;
        z.push_back(off + n == raw.size() ? 1 : 0);
        z.push_back((unsigned char)(n & 0xFF)); z.push_back((unsigned char)(n >> 8));
        z.push_back((unsigned char)(~n & 0xFF)); z.push_back((unsigned char)((~n >> 8) & 0xFF));
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
        for (size_t i = 0; i < n; i++) { s1 = (s1 + raw[off + i]) % 65521u; s2 = (s2 + s1) % 65521u; }
        off += n;
    }
    put_be32(z, (s2 << 16) | s1);

    FILE *f = fopen(path, "wb");
    if (!f) return SVGF_ERR_INVALID_ARG;
    static const unsigned char sig[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    bool ok = fwrite(sig, 1, 8, f) == 8;
    std::vector<unsigned char> ihdr;
    put_be32(ihdr, (uint32_t)width); put_be32(ihdr, (uint32_t)height);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);   // 8-bit RGB
    ok = ok && write_chunk(f, "IHDR", ihdr) && write_chunk(f, "IDAT", z) && write_chunk(f, "IEND", {});
    ok = (fclose(f) == 0) && ok;
    return ok ? SVGF_OK : SVGF_ERR_INVALID_ARG;
}

}  // extern "C"
