// ubench8.hip — what a stream restricted to N compute units (hipExtStreamCreateWithCUMask) can move (gfx950).
// Question behind it: could the HBM-bound temporal pass of frame n+1 run on a FEW CUs beside the a-trous levels of
// frame n on the rest (CU partitioning instead of co-residency, which rounds 1-3 measured as a loss)?  It would have to move
// 156 B/px x 2.07 Mpx = 324 MB within four levels (~185 us), i.e. >= 1.75 TB/s, on 32-48 CUs.
// The kernel streams like the temporal pass: 7 x 16 B read and 3.5 x 16 B written per pixel, one pixel per thread.
// Patterns of the mask: "first" = CUs 0 .. N-1, "strided" = every (256 / N)-th CU (which spreads over the XCDs if the mask bits
// are numbered XCD-major, and does not if they are numbered round-robin).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench8.hip -o tools/ubench8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e__)); exit(1); } } while (0)

constexpr int NPX = 1920 * 1080;

__global__ __launch_bounds__(256) void k_stream(const float4 *__restrict__ in, float4 *__restrict__ out, int n)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    float4 a = in[p];
#pragma unroll
    for (int k = 1; k < 7; k++) {
        const float4 b = in[(size_t)k * n + p];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    out[p] = a;
    out[(size_t)n + p] = make_float4(a.y, a.z, a.w, a.x);
    out[(size_t)2 * n + p] = make_float4(a.z, a.w, a.x, a.y);
    reinterpret_cast<float2 *>(out + (size_t)3 * n)[p] = make_float2(a.x, a.w);
}

// a compute-bound neighbour: keeps the other CUs' VALU and some of their HBM path busy (2 x 16 B per pixel, ~400 FMAs)
__global__ __launch_bounds__(256) void k_busy(const float4 *__restrict__ in, float4 *__restrict__ out, int n, int iters)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    float4 a = in[p];
    for (int i = 0; i < iters; i++) { a.x = fmaf(a.x, 1.0001f, a.y); a.y = fmaf(a.y, 0.9999f, a.z); a.z = fmaf(a.z, 1.0002f, a.w); a.w = fmaf(a.w, 0.9998f, a.x); }
    out[p] = a;
}

static hipStream_t masked_stream(int ncu, bool strided, int total)
{
    std::vector<uint32_t> m((total + 31) / 32, 0u);
    for (int i = 0; i < ncu; i++) {
        const int cu = strided ? (int)((long long)i * total / ncu) : i;
        m[cu / 32] |= 1u << (cu % 32);
    }
    hipStream_t s;
    CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)m.size(), m.data()));
    return s;
}

static hipStream_t complement_stream(int ncu, bool strided, int total)
{
    std::vector<uint32_t> m((total + 31) / 32, 0u);
    for (int i = 0; i < total; i++) m[i / 32] |= 1u << (i % 32);
    for (int i = 0; i < ncu; i++) {
        const int cu = strided ? (int)((long long)i * total / ncu) : i;
        m[cu / 32] &= ~(1u << (cu % 32));
    }
    hipStream_t s;
    CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)m.size(), m.data()));
    return s;
}

int main()
{
    int total = 0;
    CK(hipDeviceGetAttribute(&total, hipDeviceAttributeMultiprocessorCount, 0));
    float4 *in, *out, *bin, *bout;
    CK(hipMalloc(&in, (size_t)7 * NPX * 16)); CK(hipMalloc(&out, (size_t)4 * NPX * 16));
    CK(hipMalloc(&bin, (size_t)NPX * 16)); CK(hipMalloc(&bout, (size_t)NPX * 16));
    CK(hipMemset(in, 0, (size_t)7 * NPX * 16)); CK(hipMemset(bin, 0, (size_t)NPX * 16));
    const double bytes = (double)NPX * (7 * 16 + 3 * 16 + 8);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("device CUs: %d; kernel moves %.1f MB per launch (168 B/px, 1920x1080)\n", total, bytes / 1e6);
    const int grid = (NPX + 255) / 256;
    for (int strided = 0; strided <= 1; strided++) {
        for (int ncu : {16, 32, 48, 64, 96, 128, 256}) {
            if (ncu > total) continue;
            hipStream_t s = masked_stream(ncu, strided != 0, total);
            for (int w = 0; w < 20; w++) hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, s, in, out, NPX);
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            const int reps = 50;
            for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, s, in, out, NPX);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / reps;
            // the same with the complement of the mask busy with a compute-bound kernel
            hipStream_t c = complement_stream(ncu == total ? 0 : ncu, strided != 0, total);
            double us_busy = 0.0;
            if (ncu < total) {
                for (int w = 0; w < 50; w++) hipLaunchKernelGGL(k_busy, dim3(grid), dim3(256), 0, c, bin, bout, NPX, 400);
                CK(hipEventRecord(e0, s));
                for (int r = 0; r < reps; r++) {
                    hipLaunchKernelGGL(k_busy, dim3(grid), dim3(256), 0, c, bin, bout, NPX, 400);
                    hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, s, in, out, NPX);
                }
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                CK(hipStreamSynchronize(c));
                CK(hipEventElapsedTime(&ms, e0, e1));
                us_busy = ms * 1e3 / reps;
            }
            printf("%-8s %3d CUs: %8.1f us per launch = %6.2f TB/s (%5.1f GB/s per CU)%s", strided ? "strided" : "first", ncu, us, bytes / us / 1e6, bytes / us / 1e3 / ncu,
                   ncu < total ? "" : "\n");
            if (ncu < total) printf("   | beside a compute-bound kernel on the other %d CUs: %8.1f us = %6.2f TB/s\n", total - ncu, us_busy, bytes / us_busy / 1e6);
            CK(hipStreamDestroy(s)); CK(hipStreamDestroy(c));
        }
    }
    return 0;
}
