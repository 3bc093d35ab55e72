// ubench4.hip — issue cost of the instruction kinds a register-window a-trous tap is made of (gfx950): plain and
// DPP-fused VOP2, wave shifts, transcendentals, and the two composite streams ("geometry term", "tap") as a function
// of waves per SIMD.  Every body is one asm block on 16 accumulators so the compiler cannot reshape it.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench4.hip -o tools/ubench4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define DPPR " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define DPPL " wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define ROWR " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"

// %0..%7 accumulators a0..a7, %8..%15 b0..b7 (read-only-ish), %16 c, %17 d
#define OPERANDS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), \
                   "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : "v"(c), "v"(d)

template <int OP>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed)
{
    float a0 = seed + threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b0 = a0 * 0.5f, b1 = a1 * 0.5f, b2 = a2 * 0.5f, b3 = a3 * 0.5f, b4 = a4 * 0.5f, b5 = a5 * 0.5f, b6 = a6 * 0.5f, b7 = a7 * 0.5f;
    float c = 1.0001f + seed * 1e-6f, d = 0.25f + seed * 1e-6f;
    for (int it = 0; it < iters; it++) {
        if (OP == 0) asm volatile(
#define X(i) "v_fma_f32 %" #i ", %" #i ", %16, %17\n"
            R8(X) R8(X)
#undef X
            OPERANDS);
        if (OP == 1) asm volatile(
#define X(i) "v_fmac_f32 %" #i ", %16, %17\n"
            R8(X) R8(X)
#undef X
            OPERANDS);
        if (OP == 2) asm volatile(
#define X(i) "v_add_f32 %" #i ", %" #i ", %16\n"
            R8(X) R8(X)
#undef X
            OPERANDS);
        if (OP == 3) asm volatile(
#define X(i) "v_mul_f32 %" #i ", %" #i ", %16\n"
            R8(X) R8(X)
#undef X
            OPERANDS);
        if (OP == 4) asm volatile(      // dpp source is never written: no hazard
#define X(i) "v_sub_f32_dpp %" #i ", %16, %" #i DPPR
            R8(X) R8(X)
#undef X
            OPERANDS);
        if (OP == 5) asm volatile(
#define X(i) "v_fmac_f32_dpp %" #i ", %16, %17" DPPL
            R8(X) R8(X)
#undef X
            OPERANDS);
        if (OP == 6) asm volatile(
#define X(i) "v_mov_b32_dpp %" #i ", %16" DPPR
            R8(X) R8(X)
#undef X
            OPERANDS);
        if (OP == 7) asm volatile(
#define X(i) "v_fmac_f32_dpp %" #i ", %16, %17" ROWR
            R8(X) R8(X)
#undef X
            OPERANDS);
        if (OP == 8) asm volatile(
#define X(i) "v_exp_f32 %" #i ", %" #i "\n"
            R8(X) R8(X)
#undef X
            OPERANDS);
        if (OP == 9) asm volatile(
#define X(i) "v_sqrt_f32 %" #i ", %" #i "\n"
            R8(X) R8(X)
#undef X
            OPERANDS);
        if (OP == 10) asm volatile(     // 1 exp : 3 fma, interleaved
#define X(i) "v_exp_f32 %" #i ", %" #i "\n v_fma_f32 %1" #i "%%, %1" #i "%%, %16, %17\n"
#undef X
#define Y(i, j) "v_exp_f32 %" #i ", %" #i "\nv_fmac_f32 %" #j ", %16, %17\nv_fmac_f32 %" #j ", %17, %16\nv_fmac_f32 %" #j ", %16, %16\n"
            Y(0, 8) Y(1, 9) Y(2, 10) Y(3, 11) Y(4, 12) Y(5, 13) Y(6, 14) Y(7, 15)
#undef Y
            OPERANDS);
        if (OP == 11) asm volatile(     // 1 exp : 9 plain VALU (the tap's ratio), interleaved
#define Y(i, j) "v_exp_f32 %" #i ", %" #i "\nv_fmac_f32 %" #j ", %16, %17\nv_fmac_f32 %" #j ", %17, %16\nv_fmac_f32 %" #j ", %16, %16\n" \
                "v_add_f32 %" #j ", %" #j ", %16\nv_mul_f32 %" #j ", %" #j ", %17\nv_fmac_f32 %" #j ", %17, %16\nv_fmac_f32 %" #j ", %16, %16\n" \
                "v_add_f32 %" #j ", %" #j ", %16\nv_mul_f32 %" #j ", %" #j ", %17\n"
            Y(0, 8) Y(1, 9) Y(2, 10) Y(3, 11) Y(4, 12) Y(5, 13) Y(6, 14) Y(7, 15)
#undef Y
            OPERANDS);
        if (OP == 12) asm volatile(     // the 9 plain VALU of OP 11 alone
#define Y(i, j) "v_fmac_f32 %" #j ", %16, %17\nv_fmac_f32 %" #j ", %17, %16\nv_fmac_f32 %" #j ", %16, %16\n" \
                "v_add_f32 %" #j ", %" #j ", %16\nv_mul_f32 %" #j ", %" #j ", %17\nv_fmac_f32 %" #j ", %17, %16\nv_fmac_f32 %" #j ", %16, %16\n" \
                "v_add_f32 %" #j ", %" #j ", %16\nv_mul_f32 %" #j ", %" #j ", %17\n"
            Y(0, 8) Y(1, 9) Y(2, 10) Y(3, 11) Y(4, 12) Y(5, 13) Y(6, 14) Y(7, 15)
#undef Y
            OPERANDS);
        if (OP == 13) asm volatile(     // "tap" with DPP sources: sub_dpp, fma|.|, exp, mul, add, add, 3 fmac_dpp, fmac   (x8)
#define Y(i, j) "v_sub_f32_dpp %" #i ", %16, %17" DPPL "v_fma_f32 %" #i ", |%" #i "|, %17, %16\nv_exp_f32 %" #i ", %" #i "\n" \
                "v_mul_f32 %" #j ", %" #i ", %" #i "\nv_add_f32 %0, %0, %" #i "\nv_add_f32 %1, %1, %" #j "\n" \
                "v_fmac_f32_dpp %2, %16, %" #i DPPL "v_fmac_f32_dpp %3, %17, %" #i DPPL "v_fmac_f32_dpp %4, %16, %" #i DPPL \
                "v_fmac_f32_dpp %5, %17, %" #j DPPL
            Y(6, 8) Y(7, 9) Y(10, 11) Y(12, 13) Y(14, 15) Y(6, 8) Y(7, 9) Y(10, 11)
#undef Y
            OPERANDS);
        if (OP == 14) asm volatile(     // same tap without DPP
#define Y(i, j) "v_sub_f32 %" #i ", %16, %17\nv_fma_f32 %" #i ", |%" #i "|, %17, %16\nv_exp_f32 %" #i ", %" #i "\n" \
                "v_mul_f32 %" #j ", %" #i ", %" #i "\nv_add_f32 %0, %0, %" #i "\nv_add_f32 %1, %1, %" #j "\n" \
                "v_fmac_f32 %2, %16, %" #i "\nv_fmac_f32 %3, %17, %" #i "\nv_fmac_f32 %4, %16, %" #i "\n" \
                "v_fmac_f32 %5, %17, %" #j "\n"
            Y(6, 8) Y(7, 9) Y(10, 11) Y(12, 13) Y(14, 15) Y(6, 8) Y(7, 9) Y(10, 11)
#undef Y
            OPERANDS);
        if (OP == 15) asm volatile(     // "geometry term": 6 sub_dpp, 2 mul, 4 fmac, 2 sqrt, 2 fma   (x4)
#define Y(p, q, r, s, t, u) "v_sub_f32_dpp %" #p ", %16, %17" DPPL "v_sub_f32_dpp %" #q ", %17, %16" DPPL "v_sub_f32_dpp %" #r ", %16, %17" DPPL \
                "v_sub_f32_dpp %" #s ", %16, %17" DPPL "v_sub_f32_dpp %" #t ", %17, %16" DPPL "v_sub_f32_dpp %" #u ", %16, %17" DPPL \
                "v_mul_f32 %" #p ", %" #p ", %" #p "\nv_mul_f32 %" #s ", %" #s ", %" #s "\n" \
                "v_fmac_f32 %" #p ", %" #q ", %" #q "\nv_fmac_f32 %" #s ", %" #t ", %" #t "\n" \
                "v_fmac_f32 %" #p ", %" #r ", %" #r "\nv_fmac_f32 %" #s ", %" #u ", %" #u "\n" \
                "v_sqrt_f32 %" #p ", %" #p "\nv_sqrt_f32 %" #s ", %" #s "\n" \
                "v_fma_f32 %" #p ", %" #p ", %16, %17\nv_fma_f32 %" #p ", %" #s ", %17, %" #p "\n"
            Y(0, 1, 2, 3, 4, 5) Y(6, 7, 8, 9, 10, 11) Y(12, 13, 14, 15, 0, 1) Y(2, 3, 4, 5, 6, 7)
#undef Y
            OPERANDS);
        if (OP == 16) asm volatile(     // geometry term with packed math (no DPP): 3 pk_add, pk_mul, 2 pk_fma, 2 sqrt, 2 fma  (x4)
#define Y(p, q, r, s) "v_pk_add_f32 v[%" #p ":%" #q "], v[%" #p ":%" #q "], v[%" #r ":%" #s "]\n"
#undef Y
            "s_nop 0\n" OPERANDS);
        if (OP == 17) asm volatile(
#define X(i) "v_mov_b32 %" #i ", %16\n"
            R8(X) R8(X)
#undef X
            OPERANDS);
        if (OP == 18) asm volatile(     // max3/med3-like VOP3 2-src: v_max_f32
#define X(i) "v_max_f32 %" #i ", %" #i ", %16\n"
            R8(X) R8(X)
#undef X
            OPERANDS);
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7;
}

template <int OP>
void run(const char *name, int insts, int groups)
{
    static float *d = nullptr;
    if (!d) hipMalloc(&d, 4096 * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-44s", name);
    const int wps[5] = { 1, 2, 3, 4, 8 };
    for (int wi = 0; wi < 5; wi++) {
        const int blocks = 256 * wps[wi], iters = 4000;
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 64, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // time per instruction (or per group) per SIMD: each SIMD runs wps waves x iters x insts
        const double ns_per = ms * 1e6 / ((double)wps[wi] * iters * (groups ? groups : insts));
        printf("  w%d %7.3f", wps[wi], ns_per);
    }
    printf("   ns per %s per SIMD\n", groups ? "group" : "inst");
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main()
{
    run<0>("v_fma_f32 (3 vgpr src)", 16, 0);
    run<1>("v_fmac_f32", 16, 0);
    run<2>("v_add_f32", 16, 0);
    run<3>("v_mul_f32", 16, 0);
    run<17>("v_mov_b32", 16, 0);
    run<18>("v_max_f32", 16, 0);
    run<4>("v_sub_f32_dpp wave_shr:1", 16, 0);
    run<5>("v_fmac_f32_dpp wave_shl:1", 16, 0);
    run<6>("v_mov_b32_dpp wave_shr:1", 16, 0);
    run<7>("v_fmac_f32_dpp row_shr:1", 16, 0);
    run<8>("v_exp_f32", 16, 0);
    run<9>("v_sqrt_f32", 16, 0);
    run<10>("group: 1 exp + 3 fmac", 32, 8);
    run<11>("group: 1 exp + 9 plain", 80, 8);
    run<12>("group: 9 plain", 72, 8);
    run<13>("group: tap, DPP sources (9 + exp)", 80, 8);
    run<14>("group: tap, no DPP (9 + exp)", 80, 8);
    run<15>("group: geometry term, DPP (14 + 2 sqrt)", 64, 4);
    return 0;
}
