// ubench2.hip — does the transcendental pipe overlap with ordinary VALU issue on gfx950, and does instruction ORDER matter?
// Each kernel repeats a 17-instruction "tap-like" group (9 v_pk_fma_f32 + 5 v_fma_f32 + 2 v_sqrt_f32 + 1 v_exp_f32), all
// independent, in two orders: transcendentals clustered vs spread.  Also pure groups for reference.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench2.hip -o tools/ubench2
#include <hip/hip_runtime.h>
#include <cstdio>

#define PK(d) "v_pk_fma_f32 v[" #d ":" #d "+1], v[" #d ":" #d "+1], v[40:41], v[42:43]\n"
#define FM(d) "v_fma_f32 v" #d ", v" #d ", v40, v42\n"
#define SQ(d) "v_sqrt_f32 v" #d ", v" #d "\n"
#define EX(d) "v_exp_f32 v" #d ", v" #d "\n"

template <int V>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    float acc = threadIdx.x;
    for (int it = 0; it < iters; it++) {
        if (V == 0)   // clustered: 3 transcendentals back to back
            asm volatile(SQ(30) SQ(31) EX(32)
                         PK(0) PK(2) PK(4) PK(6) PK(8) PK(10) PK(12) PK(14) PK(16)
                         FM(20) FM(21) FM(22) FM(23) FM(24) ::: "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v20","v21","v22","v23","v24","v30","v31","v32","v40","v41","v42","v43");
        if (V == 1)   // spread
            asm volatile(SQ(30) PK(0) PK(2) PK(4) FM(20) FM(21) SQ(31) PK(6) PK(8) PK(10) FM(22) FM(23) EX(32) PK(12) PK(14) PK(16) FM(24)
                         ::: "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v20","v21","v22","v23","v24","v30","v31","v32","v40","v41","v42","v43");
        if (V == 2)   // no transcendentals
            asm volatile(PK(0) PK(2) PK(4) PK(6) PK(8) PK(10) PK(12) PK(14) PK(16) FM(20) FM(21) FM(22) FM(23) FM(24)
                         ::: "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v20","v21","v22","v23","v24","v40","v41","v42","v43");
        if (V == 3)   // transcendentals only
            asm volatile(SQ(30) SQ(31) EX(32) ::: "v30","v31","v32");
        if (V == 4)   // scalar-only version of the same flops: 18 + 5 v_fma
            asm volatile(FM(0) FM(1) FM(2) FM(3) FM(4) FM(5) FM(6) FM(7) FM(8) FM(9) FM(10) FM(11) FM(12) FM(13) FM(14) FM(15) FM(16) FM(17)
                         FM(20) FM(21) FM(22) FM(23) FM(24) SQ(30) SQ(31) EX(32)
                         ::: "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v20","v21","v22","v23","v24","v30","v31","v32","v40","v41","v42","v43");
        if (V == 5)   // scalar, spread
            asm volatile(SQ(30) FM(0) FM(1) FM(2) FM(3) FM(4) FM(5) FM(6) SQ(31) FM(7) FM(8) FM(9) FM(10) FM(11) FM(12) FM(13) EX(32) FM(14) FM(15) FM(16) FM(17)
                         FM(20) FM(21) FM(22) FM(23) FM(24)
                         ::: "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v20","v21","v22","v23","v24","v30","v31","v32","v40","v41","v42","v43");
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int V>
void run(const char *name, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd, iters = 20000;   // 256-thread blocks = 4 waves = 1 per SIMD
    float *d; hipMalloc(&d, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // time per group per wave, normalised per SIMD: total groups = blocks*4 waves*iters, spread over 1024 SIMDs
    double ns_per_group = ms * 1e6 / ((double)blocks * 4 * iters / 1024.0);
    printf("%-44s waves/SIMD %d  %8.3f ms  %7.2f ns per group per SIMD\n", name, waves_per_simd, ms, ns_per_group);
    hipFree(d);
}

int main()
{
    for (int w : {1, 2, 4}) {
        run<0>("tap mix, transcendentals clustered", w);
        run<1>("tap mix, transcendentals spread", w);
        run<2>("9 pk_fma + 5 fma only", w);
        run<3>("2 sqrt + 1 exp only", w);
        run<4>("scalar mix (23 fma + 3 trans), clustered", w);
        run<5>("scalar mix (23 fma + 3 trans), spread", w);
    }
    return 0;
}
