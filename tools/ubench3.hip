// ubench3.hip — is a VALU-bound kernel power/clock limited on this part?  Runs a long packed-fp32 + transcendental loop on
// a varying number of workgroups and reports the shader clock seen from inside the kernel (s_memtime ticks per
// s_memrealtime tick; the latter is a constant 100 MHz) and the per-SIMD instruction rate.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench3.hip -o tools/ubench3
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k(float *out, unsigned long long *clk, int iters, float seed, int use_lds)
{
    __shared__ float lds[4096];
    v2f p[8];
    float a[4];
#pragma unroll
    for (int i = 0; i < 8; i++) p[i] = v2f{seed + threadIdx.x * 1e-3f + i, seed + 0.5f};
#pragma unroll
    for (int i = 0; i < 4; i++) a[i] = seed + i;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = seed;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) p[i] = __builtin_elementwise_fma(p[i], v2f{1.0001f, 0.9999f}, v2f{0.5f, 0.25f});
#pragma unroll
        for (int i = 0; i < 4; i++) a[i] = __builtin_amdgcn_exp2f(a[i] * 0.5f);
        if (use_lds) {
            const float4 v = *reinterpret_cast<const float4 *>(&lds[((threadIdx.x * 12 + it * 4) & 4092)]);
            p[0].x += v.x; p[1].x += v.y; p[2].x += v.z; p[3].x += v.w;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s + a[0] + a[1] + a[2] + a[3];
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

int main()
{
    float *d; hipMalloc(&d, 256 * 16 * 256 * sizeof(float));
    unsigned long long *c; hipMalloc(&c, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 400000;
    for (int use_lds = 0; use_lds <= 1; use_lds++)
        for (int blocks : { 256 * 8, 256 * 4, 256 * 2, 256, 128, 64, 16 }) {
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, c, 1000, 1.0f, use_lds);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, c, iters, 1.0f, use_lds);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[2]; hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
            const double mhz = (double)h[0] / (double)h[1] * 100.0;
            const double waves_per_simd = blocks * 4.0 / 1024.0;
            const double inst = (double)iters * 12;                      // 8 pk + 4 (mul + exp counted as one slot each => 12 VALU-ish)
            printf("lds=%d blocks %5d (%.2f waves/SIMD)  %8.2f ms   s_memtime/s_memrealtime => %7.1f MHz   %.2f ns per loop iteration per wave\n",
                   use_lds, blocks, waves_per_simd, ms, mhz, ms * 1e6 / iters / (waves_per_simd < 1 ? 1 : waves_per_simd));
            (void)inst;
        }
    return 0;
}
