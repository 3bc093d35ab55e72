// ubench.hip — VALU issue-rate microbenchmarks on gfx950: plain fp32, packed fp32, and the three transcendentals the
// a-trous tap uses (v_exp_f32, v_sqrt_f32, v_rcp_f32).  Prints instructions per clock per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o tools/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed)
{
    float a[8];
    v2f p[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = seed + threadIdx.x * 1e-3f + i; p[i] = v2f{a[i], a[i] + 0.5f}; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f);
            if (OP == 1) p[i] = __builtin_elementwise_fma(p[i], v2f{1.0001f, 0.9999f}, v2f{0.5f, 0.25f});
            if (OP == 2) a[i] = __builtin_amdgcn_exp2f(a[i]);
            if (OP == 3) a[i] = __builtin_amdgcn_sqrtf(a[i]);
            if (OP == 4) a[i] = __builtin_amdgcn_rcpf(a[i]);
            if (OP == 5) { a[i] = __builtin_amdgcn_exp2f(a[i]); p[i] = __builtin_elementwise_fma(p[i], v2f{1.0001f, 0.9999f}, v2f{0.5f, 0.25f}); }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
void run(const char *name, int insts_per_iter_per_lane)
{
    const int blocks = 256 * 8, iters = 4096;
    float *d; hipMalloc(&d, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 16, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winst = (double)blocks * 4 /*waves*/ * iters * insts_per_iter_per_lane;   // wave-instructions
    double per_s = winst / (ms * 1e-3);
    // 256 CUs x 4 SIMDs; report wave-instructions per ns chip-wide and cycles per wave-instruction per SIMD at 2.4 GHz
    printf("%-28s %8.3f ms  %9.2f Gwave-inst/s  => %.2f clk/wave-inst/SIMD @2.4GHz\n", name, ms, per_s / 1e9,
           2.4e9 * 1024.0 / per_s);
    hipFree(d);
}

int main()
{
    run<0>("v_fma_f32", 8);
    run<1>("v_pk_fma_f32", 8);
    run<2>("v_exp_f32", 8);
    run<3>("v_sqrt_f32", 8);
    run<4>("v_rcp_f32", 8);
    run<5>("v_exp_f32 + v_pk_fma_f32", 16);
    return 0;
}
