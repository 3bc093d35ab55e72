// ubench9.hip — what a kernel boundary between two a-trous levels costs against a grid-wide barrier inside ONE persistent launch (gfx950).
// Question behind it: the five levels of a frame are five launches of 256 workgroups (one per CU,
// 768 threads, ~148 KB of LDS); a persistent kernel would keep the workgroups resident and separate the levels by a device-scope
// barrier (release: write back the XCD's L2; acquire: invalidate it), which is what the kernel boundary does implicitly.
// Each "level" here: every thread reads 8 x 16 B that ANOTHER workgroup wrote in the previous level (the opening burst: 25 MB over
// the chip), spins until T us have passed since the level began (s_memtime), and writes 8 x 16 B (25 MB over the chip).
//   A: five launches per frame (today)          B: one launch, five levels, four grid barriers          E: five empty launches
// Output: us per frame and the difference per level boundary.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench9.hip -o tools/ubench9      run: tools/ubench9 [T_us=40] [frames=400]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e__)); exit(1); } } while (0)

constexpr int NT = 768, NB = 256, PER = 8;
constexpr size_t kLds = 148 * 1024;

__device__ __forceinline__ void level_body(const float4 *__restrict__ src, float4 *__restrict__ dst, int lvl, long long ticks)
{
    extern __shared__ float4 sm[];
    const long long t0 = __builtin_amdgcn_s_memtime();
    // read what the workgroup (b + 37 * (lvl + 1)) % NB wrote: another XCD's L2 held it dirty when the previous level ended
    const int ob = (blockIdx.x + 37 * (lvl + 1)) % NB;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const float4 v = src[((size_t)ob * PER + k) * NT + threadIdx.x];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    while ((long long)__builtin_amdgcn_s_memtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    const float4 o = sm[(threadIdx.x + 1) % NT];
#pragma unroll
    for (int k = 0; k < PER; k++) dst[((size_t)blockIdx.x * PER + k) * NT + threadIdx.x] = make_float4(o.x + k, o.y, o.z, o.w + lvl);
}

__global__ __launch_bounds__(NT) void k_level(const float4 *src, float4 *dst, int lvl, long long ticks) { level_body(src, dst, lvl, ticks); }

__global__ __launch_bounds__(NT) void k_empty(float4 *dst) { if (dst == nullptr) __syncthreads(); }

// sense-reversing grid barrier: one arrival counter, one generation word
__device__ __forceinline__ void grid_barrier(unsigned *cnt, unsigned *gen, unsigned my_gen)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        // release: everything this workgroup wrote becomes visible device-wide before the arrival is counted
        const unsigned arrived = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1;
        if (arrived == (unsigned)NB * (my_gen + 1)) {
            __hip_atomic_store(gen, my_gen + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) <= my_gen) __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();      // (thread 0's acquire invalidated the CU's vector cache and the XCD's L2 for the whole workgroup; a fence in
                          // every thread — 196 608 L2 write-backs and invalidations per barrier — was measured first: 100 us per barrier)
}

__global__ __launch_bounds__(NT) void k_persist(float4 *p0, float4 *p1, long long ticks, unsigned *cnt, unsigned *gen, unsigned gen0)
{
    float4 *src = p0, *dst = p1;
    for (int lvl = 0; lvl < 5; lvl++) {
        level_body(src, dst, lvl, ticks);
        if (lvl < 4) grid_barrier(cnt, gen, gen0 + lvl);
        float4 *t = src; src = dst; dst = t;
    }
}

int main(int argc, char **argv)
{
    const double T_us = argc > 1 ? atof(argv[1]) : 40.0;
    const int frames = argc > 2 ? atoi(argv[2]) : 400;
    int dev = 0; CK(hipSetDevice(dev));
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, dev));
    if (pr.multiProcessorCount < NB) { fprintf(stderr, "needs %d CUs (one resident workgroup each) for the grid barrier, device has %d\n", NB, pr.multiProcessorCount); return 2; }
    // s_memtime counts at a constant 100 MHz on gfx9 (REFCLK): measure it instead of assuming
    CK(hipFuncSetAttribute((const void *)k_level, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds));
    CK(hipFuncSetAttribute((const void *)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds));
    CK(hipFuncSetAttribute((const void *)k_persist, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds));
    const size_t n = (size_t)NB * PER * NT;
    float4 *p0, *p1; unsigned *sync;
    CK(hipMalloc(&p0, n * sizeof(float4))); CK(hipMalloc(&p1, n * sizeof(float4))); CK(hipMalloc(&sync, 256));
    CK(hipMemset(p0, 0, n * sizeof(float4))); CK(hipMemset(p1, 0, n * sizeof(float4))); CK(hipMemset(sync, 0, 256));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // calibrate the s_memtime rate: a level asked to spin 1e5 ticks
    auto time_levels = [&](long long ticks, int nfr) {
        CK(hipEventRecord(e0, s));
        for (int f = 0; f < nfr; f++)
            for (int l = 0; l < 5; l++) hipLaunchKernelGGL(k_level, dim3(NB), dim3(NT), kLds, s, (l & 1) ? p1 : p0, (l & 1) ? p0 : p1, l, ticks);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return (double)ms * 1e3 / nfr;
    };
    (void)time_levels(1000, 50);
    const double t_a = time_levels(1000, 100), t_b = time_levels(101000, 100);
    const double ticks_per_us = 5.0 * 100000.0 / (t_b - t_a);
    const long long ticks = (long long)(T_us * ticks_per_us);
    printf("s_memtime: %.1f ticks per us; a level spins %lld ticks (%.1f us)\n", ticks_per_us, ticks, T_us);

    // warm up into the sustained state, then A / B / E alternating
    (void)time_levels(ticks, 1500);
    unsigned gen0 = 0;
    auto time_persist = [&](int nfr) {
        CK(hipEventRecord(e0, s));
        for (int f = 0; f < nfr; f++) { hipLaunchKernelGGL(k_persist, dim3(NB), dim3(NT), kLds, s, p0, p1, ticks, sync, sync + 32, gen0); gen0 += 4; }
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return (double)ms * 1e3 / nfr;
    };
    auto time_empty = [&](int nfr) {
        CK(hipEventRecord(e0, s));
        for (int f = 0; f < nfr; f++) for (int l = 0; l < 5; l++) hipLaunchKernelGGL(k_empty, dim3(NB), dim3(NT), kLds, s, p0);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return (double)ms * 1e3 / nfr;
    };
    for (int r = 0; r < 3; r++) {
        const double a = time_levels(ticks, frames);
        const double b = time_persist(frames);
        const double e = time_empty(frames);
        printf("run %d: A five launches %.2f us/frame (%.2f per level, %.2f over the spin) | B persistent %.2f us/frame (%.2f per level, %.2f over the spin) | "
               "E five empty launches %.2f us (%.2f each) | boundary - barrier = %.2f us per level\n",
               r, a, a / 5, a / 5 - T_us, b, b / 5, b / 5 - T_us, e, e / 5, (a - b) / 4);
    }
    // what the 2.6 us of an empty launch are made of: the same empty kernel in other shapes (256 workgroups each)
    struct Shape { int threads; size_t lds; const char *what; } shapes[] = {
        { 64, 0, "64 threads, no LDS" }, { 256, 0, "256 threads, no LDS" }, { 768, 0, "768 threads, no LDS" },
        { 768, 64 * 1024, "768 threads, 64 KB LDS" }, { 768, kLds, "768 threads, 148 KB LDS (the lane kernel's shape)" } };
    for (const Shape &sh : shapes) {
        for (int r = 0; r < 2; r++) {
            CK(hipEventRecord(e0, s));
            for (int f = 0; f < 2000; f++) hipLaunchKernelGGL(k_empty, dim3(NB), dim3(sh.threads), sh.lds, s, p0);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r == 1) printf("empty launch, 256 workgroups x %s: %.2f us each (2000 back to back)\n", sh.what, (double)ms * 1e3 / 2000);
        }
    }
    CK(hipDeviceSynchronize());
    return 0;
}
