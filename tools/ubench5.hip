// ubench5.hip — issue cost of the register-window a-trous instruction stream on gfx950, with the transcendentals
// clustered the way the kernel issues them (5-tap batches).  Whole loops are written in asm on fixed registers so the
// stream is exactly what is listed.  Clocks are warmed up by a long dummy launch before every measurement.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench5.hip -o tools/ubench5
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define DL " wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define DR " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define NL "\n"

// registers: v10-v15 accumulators (sw, sw2, r, g, b, v); v16-v20 d/e/w of 5 taps; v21-v25 w2; v30-v39 "row data"
// (lq, r, g, b, v ...), v40-v44 t terms, v8 = lp, v9 = kl; geometry: v50-v79 differences, v80-v89 squared sums
#define TAP_PRE(k, SUF)  "v_sub_f32" SUF " v1" #k ", v3" #k ", v8" 
#define TAP5(SUBOP, FMACOP, MOD)                                                                          \
    SUBOP " v16, v30, v8" MOD SUBOP " v17, v31, v8" MOD SUBOP " v18, v32, v8" MOD SUBOP " v19, v33, v8" MOD SUBOP " v20, v34, v8" MOD \
    "v_fma_f32 v16, |v16|, v9, v40\nv_fma_f32 v17, |v17|, v9, v41\nv_fma_f32 v18, |v18|, v9, v42\nv_fma_f32 v19, |v19|, v9, v43\nv_fma_f32 v20, |v20|, v9, v44\n" \
    "v_exp_f32 v16, -v16\nv_exp_f32 v17, -v17\nv_exp_f32 v18, -v18\nv_exp_f32 v19, -v19\nv_exp_f32 v20, -v20\n" \
    ACC1(16, 21, FMACOP, MOD) ACC1(17, 22, FMACOP, MOD) ACC1(18, 23, FMACOP, MOD) ACC1(19, 24, FMACOP, MOD) ACC1(20, 25, FMACOP, MOD)
#define ACC1(w, w2, FMACOP, MOD)                                                                          \
    "v_mul_f32 v" #w2 ", v" #w ", v" #w "\nv_add_f32 v10, v10, v" #w "\nv_add_f32 v11, v11, v" #w2 "\n"   \
    FMACOP " v12, v35, v" #w MOD FMACOP " v13, v36, v" #w MOD FMACOP " v14, v37, v" #w MOD FMACOP " v15, v38, v" #w2 MOD
// 5 geometry terms: 30 differences (partner rows v30-v39 stand in for n/p of the partner), squares, 10 sqrt, 10 fma
#define G6(k, SUBOP, MOD)                                                                                 \
    SUBOP " v5" #k ", v30, v8" MOD SUBOP " v6" #k ", v31, v9" MOD SUBOP " v7" #k ", v32, v8" MOD          \
    SUBOP " v8" #k ", v33, v9" MOD SUBOP " v9" #k ", v34, v8" MOD SUBOP " v10" #k ", v35, v9" MOD
#define GSQ(k)                                                                                            \
    "v_mul_f32 v5" #k ", v5" #k ", v5" #k "\nv_mul_f32 v8" #k ", v8" #k ", v8" #k "\n"                    \
    "v_fmac_f32 v5" #k ", v6" #k ", v6" #k "\nv_fmac_f32 v8" #k ", v9" #k ", v9" #k "\n"                  \
    "v_fmac_f32 v5" #k ", v7" #k ", v7" #k "\nv_fmac_f32 v8" #k ", v10" #k ", v10" #k "\n"
#define GRT(k) "v_sqrt_f32 v5" #k ", v5" #k "\nv_sqrt_f32 v8" #k ", v8" #k "\n"
#define GT(k)  "v_fma_f32 v4" #k ", v5" #k ", v9, v8\nv_fma_f32 v4" #k ", v8" #k ", v8, v4" #k "\n"
#define GEO5(SUBOP, MOD)                                                                                  \
    G6(0, SUBOP, MOD) G6(1, SUBOP, MOD) G6(2, SUBOP, MOD) G6(3, SUBOP, MOD) G6(4, SUBOP, MOD)             \
    GSQ(0) GSQ(1) GSQ(2) GSQ(3) GSQ(4) GRT(0) GRT(1) GRT(2) GRT(3) GRT(4) GT(0) GT(1) GT(2) GT(3) GT(4)

#define CLOB "v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25", \
    "v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44",              \
    "v50","v51","v52","v53","v54","v60","v61","v62","v63","v64","v70","v71","v72","v73","v74",              \
    "v80","v81","v82","v83","v84","v90","v91","v92","v93","v94","v100","v101","v102","v103","v104","s20","scc","vcc"

#define INIT                                                                                              \
    "v_mov_b32 v8, %1\nv_mov_b32 v9, %2\n"                                                              \
    "v_mov_b32 v10, 0\nv_mov_b32 v11, 0\nv_mov_b32 v12, 0\nv_mov_b32 v13, 0\nv_mov_b32 v14, 0\nv_mov_b32 v15, 0\n" \
    "v_add_f32 v30, %1, %2\nv_add_f32 v31, v30, %2\nv_add_f32 v32, v31, %2\nv_add_f32 v33, v32, %2\nv_add_f32 v34, v33, %2\n" \
    "v_add_f32 v35, v34, %2\nv_add_f32 v36, v35, %2\nv_add_f32 v37, v36, %2\nv_add_f32 v38, v37, %2\nv_add_f32 v39, v38, %2\n" \
    "v_mov_b32 v40, %2\nv_mov_b32 v41, %2\nv_mov_b32 v42, %2\nv_mov_b32 v43, %2\nv_mov_b32 v44, %2\n"  \
    "s_mov_b32 s20, %3\n"
#define LOOP_HEAD "1:\n"
#define LOOP_TAIL "s_sub_u32 s20, s20, 1\ns_cmp_lg_u32 s20, 0\ns_cbranch_scc1 1b\n"                         \
    "v_add_f32 %0, v10, v11\nv_add_f32 %0, %0, v12\nv_add_f32 %0, %0, v13\nv_add_f32 %0, %0, v14\nv_add_f32 %0, %0, v15\n" \
    "v_add_f32 %0, %0, v40\nv_add_f32 %0, %0, v41\nv_add_f32 %0, %0, v42\nv_add_f32 %0, %0, v43\nv_add_f32 %0, %0, v44\n"

template <int OP>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed)
{
    float r, a = seed + threadIdx.x * 1e-3f, b = 0.37f + seed * 1e-3f;
    if (OP == 0) asm volatile(INIT LOOP_HEAD TAP5("v_sub_f32", "v_fmac_f32", NL) LOOP_TAIL : "=v"(r) : "v"(a), "v"(b), "s"(iters) : CLOB);
    if (OP == 1) asm volatile(INIT LOOP_HEAD TAP5("v_sub_f32_dpp", "v_fmac_f32_dpp", DL) LOOP_TAIL : "=v"(r) : "v"(a), "v"(b), "s"(iters) : CLOB);
    if (OP == 2) asm volatile(INIT LOOP_HEAD GEO5("v_sub_f32", NL) LOOP_TAIL : "=v"(r) : "v"(a), "v"(b), "s"(iters) : CLOB);
    if (OP == 3) asm volatile(INIT LOOP_HEAD GEO5("v_sub_f32_dpp", DR) LOOP_TAIL : "=v"(r) : "v"(a), "v"(b), "s"(iters) : CLOB);
    // one whole pixel-iteration of the register-window kernel: 12 geometry terms + 24 taps (+ nothing else)
    if (OP == 4) asm volatile(INIT LOOP_HEAD
                              TAP5("v_sub_f32_dpp", "v_fmac_f32_dpp", DL) TAP5("v_sub_f32_dpp", "v_fmac_f32_dpp", DR)
                              GEO5("v_sub_f32_dpp", DR) TAP5("v_sub_f32_dpp", "v_fmac_f32_dpp", DL)
                              GEO5("v_sub_f32_dpp", DL) TAP5("v_sub_f32_dpp", "v_fmac_f32_dpp", DL)
                              G6(0, "v_sub_f32_dpp", DL) G6(1, "v_sub_f32_dpp", DL) GSQ(0) GSQ(1) GRT(0) GRT(1) GT(0) GT(1)
                              TAP5("v_sub_f32", "v_fmac_f32", NL)
                              LOOP_TAIL : "=v"(r) : "v"(a), "v"(b), "s"(iters) : CLOB);
    // plain VALU only (tap without the exps), and exps only
    if (OP == 5) asm volatile(INIT LOOP_HEAD
        "v_exp_f32 v16, -v16\nv_exp_f32 v17, -v17\nv_exp_f32 v18, -v18\nv_exp_f32 v19, -v19\nv_exp_f32 v20, -v20\n"
        LOOP_TAIL : "=v"(r) : "v"(a), "v"(b), "s"(iters) : CLOB);
    if (OP == 6) asm volatile(INIT LOOP_HEAD
        "v_sub_f32 v16, v30, v8\nv_sub_f32 v17, v31, v8\nv_sub_f32 v18, v32, v8\nv_sub_f32 v19, v33, v8\nv_sub_f32 v20, v34, v8\n"
        "v_fma_f32 v16, |v16|, v9, v40\nv_fma_f32 v17, |v17|, v9, v41\nv_fma_f32 v18, |v18|, v9, v42\nv_fma_f32 v19, |v19|, v9, v43\nv_fma_f32 v20, |v20|, v9, v44\n"
        ACC1(16, 21, "v_fmac_f32", NL) ACC1(17, 22, "v_fmac_f32", NL) ACC1(18, 23, "v_fmac_f32", NL) ACC1(19, 24, "v_fmac_f32", NL) ACC1(20, 25, "v_fmac_f32", NL)
        LOOP_TAIL : "=v"(r) : "v"(a), "v"(b), "s"(iters) : CLOB);
    // accumulate part with the per-tap ops interleaved across taps (all muls, all adds, ...), plain
    if (OP == 7) asm volatile(INIT LOOP_HEAD
        "v_sub_f32 v16, v30, v8\nv_sub_f32 v17, v31, v8\nv_sub_f32 v18, v32, v8\nv_sub_f32 v19, v33, v8\nv_sub_f32 v20, v34, v8\n"
        "v_fma_f32 v16, |v16|, v9, v40\nv_fma_f32 v17, |v17|, v9, v41\nv_fma_f32 v18, |v18|, v9, v42\nv_fma_f32 v19, |v19|, v9, v43\nv_fma_f32 v20, |v20|, v9, v44\n"
        "v_exp_f32 v16, -v16\nv_exp_f32 v17, -v17\nv_exp_f32 v18, -v18\nv_exp_f32 v19, -v19\nv_exp_f32 v20, -v20\n"
        "v_mul_f32 v21, v16, v16\nv_mul_f32 v22, v17, v17\nv_mul_f32 v23, v18, v18\nv_mul_f32 v24, v19, v19\nv_mul_f32 v25, v20, v20\n"
        "v_fmac_f32 v12, v35, v16\nv_fmac_f32 v13, v36, v16\nv_fmac_f32 v14, v37, v16\nv_fmac_f32 v15, v38, v21\nv_add_f32 v10, v10, v16\nv_add_f32 v11, v11, v21\n"
        "v_fmac_f32 v12, v35, v17\nv_fmac_f32 v13, v36, v17\nv_fmac_f32 v14, v37, v17\nv_fmac_f32 v15, v38, v22\nv_add_f32 v10, v10, v17\nv_add_f32 v11, v11, v22\n"
        "v_fmac_f32 v12, v35, v18\nv_fmac_f32 v13, v36, v18\nv_fmac_f32 v14, v37, v18\nv_fmac_f32 v15, v38, v23\nv_add_f32 v10, v10, v18\nv_add_f32 v11, v11, v23\n"
        "v_fmac_f32 v12, v35, v19\nv_fmac_f32 v13, v36, v19\nv_fmac_f32 v14, v37, v19\nv_fmac_f32 v15, v38, v24\nv_add_f32 v10, v10, v19\nv_add_f32 v11, v11, v24\n"
        "v_fmac_f32 v12, v35, v20\nv_fmac_f32 v13, v36, v20\nv_fmac_f32 v14, v37, v20\nv_fmac_f32 v15, v38, v25\nv_add_f32 v10, v10, v20\nv_add_f32 v11, v11, v25\n"
        LOOP_TAIL : "=v"(r) : "v"(a), "v"(b), "s"(iters) : CLOB);
    // 20 independent fmas (rate reference) and 20 independent dpp fmacs
    if (OP == 8) asm volatile(INIT LOOP_HEAD
        "v_fma_f32 v50, v30, v8, v9\nv_fma_f32 v51, v31, v8, v9\nv_fma_f32 v52, v32, v8, v9\nv_fma_f32 v53, v33, v8, v9\nv_fma_f32 v54, v34, v8, v9\n"
        "v_fma_f32 v60, v30, v8, v9\nv_fma_f32 v61, v31, v8, v9\nv_fma_f32 v62, v32, v8, v9\nv_fma_f32 v63, v33, v8, v9\nv_fma_f32 v64, v34, v8, v9\n"
        "v_fma_f32 v70, v30, v8, v9\nv_fma_f32 v71, v31, v8, v9\nv_fma_f32 v72, v32, v8, v9\nv_fma_f32 v73, v33, v8, v9\nv_fma_f32 v74, v34, v8, v9\n"
        "v_fma_f32 v80, v30, v8, v9\nv_fma_f32 v81, v31, v8, v9\nv_fma_f32 v82, v32, v8, v9\nv_fma_f32 v83, v33, v8, v9\nv_fma_f32 v84, v34, v8, v9\n"
        LOOP_TAIL : "=v"(r) : "v"(a), "v"(b), "s"(iters) : CLOB);
    if (OP == 9) asm volatile(INIT LOOP_HEAD
        "v_sub_f32_dpp v50, v30, v8" DL "v_sub_f32_dpp v51, v31, v8" DL "v_sub_f32_dpp v52, v32, v8" DL "v_sub_f32_dpp v53, v33, v8" DL "v_sub_f32_dpp v54, v34, v8" DL
        "v_sub_f32_dpp v60, v30, v8" DL "v_sub_f32_dpp v61, v31, v8" DL "v_sub_f32_dpp v62, v32, v8" DL "v_sub_f32_dpp v63, v33, v8" DL "v_sub_f32_dpp v64, v34, v8" DL
        "v_sub_f32_dpp v70, v30, v8" DL "v_sub_f32_dpp v71, v31, v8" DL "v_sub_f32_dpp v72, v32, v8" DL "v_sub_f32_dpp v73, v33, v8" DL "v_sub_f32_dpp v74, v34, v8" DL
        "v_sub_f32_dpp v80, v30, v8" DL "v_sub_f32_dpp v81, v31, v8" DL "v_sub_f32_dpp v82, v32, v8" DL "v_sub_f32_dpp v83, v33, v8" DL "v_sub_f32_dpp v84, v34, v8" DL
        LOOP_TAIL : "=v"(r) : "v"(a), "v"(b), "s"(iters) : CLOB);
    if (OP == 10) asm volatile(INIT LOOP_HEAD
        "v_sub_f32 v50, v30, v8\nv_sub_f32 v51, v31, v8\nv_sub_f32 v52, v32, v8\nv_sub_f32 v53, v33, v8\nv_sub_f32 v54, v34, v8\n"
        "v_sub_f32 v60, v30, v8\nv_sub_f32 v61, v31, v8\nv_sub_f32 v62, v32, v8\nv_sub_f32 v63, v33, v8\nv_sub_f32 v64, v34, v8\n"
        "v_sub_f32 v70, v30, v8\nv_sub_f32 v71, v31, v8\nv_sub_f32 v72, v32, v8\nv_sub_f32 v73, v33, v8\nv_sub_f32 v74, v34, v8\n"
        "v_sub_f32 v80, v30, v8\nv_sub_f32 v81, v31, v8\nv_sub_f32 v82, v32, v8\nv_sub_f32 v83, v33, v8\nv_sub_f32 v84, v34, v8\n"
        LOOP_TAIL : "=v"(r) : "v"(a), "v"(b), "s"(iters) : CLOB);
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

__global__ void warm(float *out, int iters)
{
    float a = threadIdx.x;
    for (int i = 0; i < iters; i++) a = __builtin_fmaf(a, 1.0001f, 0.5f);
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

template <int OP>
void run(const char *name, int insts, double ns_target_iter)
{
    static float *d = nullptr;
    if (!d) (void)hipMalloc(&d, 8192 * 256 * sizeof(float));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("%-52s %3d inst:", name, insts);
    const int wps[5] = { 1, 2, 3, 4, 8 };
    for (int wi = 0; wi < 5; wi++) {
        const int blocks = 256 * wps[wi];
        const int iters = (int)(6e6 / (ns_target_iter * wps[wi]));     // ~6 ms per run
        hipLaunchKernelGGL(warm, dim3(4096), dim3(256), 0, 0, d, 300000);   // ~40 ms of full-chip VALU: clocks up
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 64, 1.0f);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double ns_per = ms * 1e6 / ((double)wps[wi] * iters);
        printf("  w%d %8.2f", wps[wi], ns_per);
    }
    printf("   ns per body per SIMD\n");
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
}

int main()
{
    run<8>("20 independent v_fma_f32", 20, 25);
    run<10>("20 independent v_sub_f32", 20, 25);
    run<9>("20 independent v_sub_f32_dpp wave_shl", 20, 40);
    run<5>("5 v_exp_f32", 5, 20);
    run<6>("5 taps without exps (45 plain)", 45, 50);
    run<0>("5 taps, own lane (45 plain + 5 exp)", 50, 75);
    run<7>("5 taps, own lane, accumulate interleaved", 50, 75);
    run<1>("5 taps, DPP sources", 50, 75);
    run<2>("5 geometry terms, plain (70 + 10 sqrt)", 80, 130);
    run<3>("5 geometry terms, DPP (70 + 10 sqrt)", 80, 130);
    run<4>("pixel-iteration: 12 geo + 25 taps, DPP (~442)", 442, 700);
    return 0;
}
