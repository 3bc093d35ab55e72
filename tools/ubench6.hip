// ubench6.hip — cycles per VALU wave-instruction on gfx950 measured INSIDE the kernel with s_memtime (shader clock), so
// that DVFS ramps, launch overhead and the wall clock do not enter.  One workgroup per CU, WPS waves per SIMD, every wave
// runs REPS x (one asm body of fixed registers) between two s_memtime reads; the host prints the median over all waves
// of cycles per instruction per SIMD (= wave cycles / instructions / ... x waves sharing the SIMD).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench6.hip -o tools/ubench6
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CLOB "v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31", \
             "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55", \
             "v56","v57","v58","v59","v60","v61","v62","v63"

// 16 independent accumulators v8..v23; sources v32.. (different banks: reg mod 4)
#define FMA16(A, B) \
    "v_fma_f32 v8, v" #A ", v" #B ", v8\n v_fma_f32 v9, v" #A ", v" #B ", v9\n v_fma_f32 v10, v" #A ", v" #B ", v10\n v_fma_f32 v11, v" #A ", v" #B ", v11\n" \
    "v_fma_f32 v12, v" #A ", v" #B ", v12\n v_fma_f32 v13, v" #A ", v" #B ", v13\n v_fma_f32 v14, v" #A ", v" #B ", v14\n v_fma_f32 v15, v" #A ", v" #B ", v15\n" \
    "v_fma_f32 v16, v" #A ", v" #B ", v16\n v_fma_f32 v17, v" #A ", v" #B ", v17\n v_fma_f32 v18, v" #A ", v" #B ", v18\n v_fma_f32 v19, v" #A ", v" #B ", v19\n" \
    "v_fma_f32 v20, v" #A ", v" #B ", v20\n v_fma_f32 v21, v" #A ", v" #B ", v21\n v_fma_f32 v22, v" #A ", v" #B ", v22\n v_fma_f32 v23, v" #A ", v" #B ", v23\n"
#define MUL16 \
    "v_mul_f32 v8, v32, v8\n v_mul_f32 v9, v33, v9\n v_mul_f32 v10, v34, v10\n v_mul_f32 v11, v35, v11\n v_mul_f32 v12, v32, v12\n v_mul_f32 v13, v33, v13\n v_mul_f32 v14, v34, v14\n v_mul_f32 v15, v35, v15\n" \
    "v_mul_f32 v16, v32, v16\n v_mul_f32 v17, v33, v17\n v_mul_f32 v18, v34, v18\n v_mul_f32 v19, v35, v19\n v_mul_f32 v20, v32, v20\n v_mul_f32 v21, v33, v21\n v_mul_f32 v22, v34, v22\n v_mul_f32 v23, v35, v23\n"
#define FMAC16 \
    "v_fmac_f32 v8, v32, v33\n v_fmac_f32 v9, v32, v33\n v_fmac_f32 v10, v32, v33\n v_fmac_f32 v11, v32, v33\n v_fmac_f32 v12, v32, v33\n v_fmac_f32 v13, v32, v33\n v_fmac_f32 v14, v32, v33\n v_fmac_f32 v15, v32, v33\n" \
    "v_fmac_f32 v16, v32, v33\n v_fmac_f32 v17, v32, v33\n v_fmac_f32 v18, v32, v33\n v_fmac_f32 v19, v32, v33\n v_fmac_f32 v20, v32, v33\n v_fmac_f32 v21, v32, v33\n v_fmac_f32 v22, v32, v33\n v_fmac_f32 v23, v32, v33\n"
#define PKFMA8 \
    "v_pk_fma_f32 v[8:9], v[32:33], v[34:35], v[8:9]\n v_pk_fma_f32 v[10:11], v[32:33], v[34:35], v[10:11]\n v_pk_fma_f32 v[12:13], v[32:33], v[34:35], v[12:13]\n v_pk_fma_f32 v[14:15], v[32:33], v[34:35], v[14:15]\n" \
    "v_pk_fma_f32 v[16:17], v[32:33], v[34:35], v[16:17]\n v_pk_fma_f32 v[18:19], v[32:33], v[34:35], v[18:19]\n v_pk_fma_f32 v[20:21], v[32:33], v[34:35], v[20:21]\n v_pk_fma_f32 v[22:23], v[32:33], v[34:35], v[22:23]\n"
#define PKADD8 \
    "v_pk_add_f32 v[8:9], v[32:33], v[8:9]\n v_pk_add_f32 v[10:11], v[32:33], v[10:11]\n v_pk_add_f32 v[12:13], v[32:33], v[12:13]\n v_pk_add_f32 v[14:15], v[32:33], v[14:15]\n" \
    "v_pk_add_f32 v[16:17], v[32:33], v[16:17]\n v_pk_add_f32 v[18:19], v[32:33], v[18:19]\n v_pk_add_f32 v[20:21], v[32:33], v[20:21]\n v_pk_add_f32 v[22:23], v[32:33], v[22:23]\n"
#define PKMUL8 \
    "v_pk_mul_f32 v[8:9], v[32:33], v[8:9]\n v_pk_mul_f32 v[10:11], v[32:33], v[10:11]\n v_pk_mul_f32 v[12:13], v[32:33], v[12:13]\n v_pk_mul_f32 v[14:15], v[32:33], v[14:15]\n" \
    "v_pk_mul_f32 v[16:17], v[32:33], v[16:17]\n v_pk_mul_f32 v[18:19], v[32:33], v[18:19]\n v_pk_mul_f32 v[20:21], v[32:33], v[20:21]\n v_pk_mul_f32 v[22:23], v[32:33], v[22:23]\n"
#define EXP8 \
    "v_exp_f32 v8, v40\n v_exp_f32 v9, v41\n v_exp_f32 v10, v42\n v_exp_f32 v11, v43\n v_exp_f32 v12, v40\n v_exp_f32 v13, v41\n v_exp_f32 v14, v42\n v_exp_f32 v15, v43\n"
#define SQRT8 \
    "v_sqrt_f32 v8, v40\n v_sqrt_f32 v9, v41\n v_sqrt_f32 v10, v42\n v_sqrt_f32 v11, v43\n v_sqrt_f32 v12, v40\n v_sqrt_f32 v13, v41\n v_sqrt_f32 v14, v42\n v_sqrt_f32 v15, v43\n"
#define RCP8 \
    "v_rcp_f32 v8, v40\n v_rcp_f32 v9, v41\n v_rcp_f32 v10, v42\n v_rcp_f32 v11, v43\n v_rcp_f32 v12, v40\n v_rcp_f32 v13, v41\n v_rcp_f32 v14, v42\n v_rcp_f32 v15, v43\n"
// exp results feeding FMAs a few instructions later: the tap's shape, 1 exp : 6 plain
#define TAPMIX \
    "v_exp_f32 v24, v40\n v_fma_f32 v8, v32, v33, v8\n v_fma_f32 v9, v32, v33, v9\n v_fma_f32 v10, v32, v33, v10\n v_fma_f32 v11, v32, v33, v11\n v_fma_f32 v12, v32, v33, v12\n v_fma_f32 v13, v32, v33, v13\n" \
    "v_exp_f32 v25, v41\n v_fma_f32 v14, v32, v33, v14\n v_fma_f32 v15, v32, v33, v15\n v_fma_f32 v16, v32, v33, v16\n v_fma_f32 v17, v32, v33, v17\n v_fma_f32 v18, v32, v33, v18\n v_fma_f32 v19, v32, v33, v19\n"
// exp clustered: 4 exp then 24 plain
#define TAPCLUSTER \
    "v_exp_f32 v24, v40\n v_exp_f32 v25, v41\n v_exp_f32 v26, v42\n v_exp_f32 v27, v43\n" FMA16(32, 33) \
    "v_fma_f32 v8, v32, v33, v8\n v_fma_f32 v9, v32, v33, v9\n v_fma_f32 v10, v32, v33, v10\n v_fma_f32 v11, v32, v33, v11\n v_fma_f32 v12, v32, v33, v12\n v_fma_f32 v13, v32, v33, v13\n v_fma_f32 v14, v32, v33, v14\n v_fma_f32 v15, v32, v33, v15\n"
// same-bank sources (v32, v36, v8+4k: all = 0 mod 4)
#define FMA_SAMEBANK \
    "v_fma_f32 v8, v32, v36, v8\n v_fma_f32 v12, v32, v36, v12\n v_fma_f32 v16, v32, v36, v16\n v_fma_f32 v20, v32, v36, v20\n v_fma_f32 v24, v32, v36, v24\n v_fma_f32 v28, v32, v36, v28\n v_fma_f32 v44, v32, v36, v44\n v_fma_f32 v48, v32, v36, v48\n" \
    "v_fma_f32 v8, v32, v36, v8\n v_fma_f32 v12, v32, v36, v12\n v_fma_f32 v16, v32, v36, v16\n v_fma_f32 v20, v32, v36, v20\n v_fma_f32 v24, v32, v36, v24\n v_fma_f32 v28, v32, v36, v28\n v_fma_f32 v44, v32, v36, v44\n v_fma_f32 v48, v32, v36, v48\n"
// two of the three sources in one bank (v32, v36 = bank 0), the accumulator in bank 1
#define FMA_2SAME \
    "v_fma_f32 v9, v32, v36, v9\n v_fma_f32 v13, v32, v36, v13\n v_fma_f32 v17, v32, v36, v17\n v_fma_f32 v21, v32, v36, v21\n v_fma_f32 v25, v32, v36, v25\n v_fma_f32 v29, v32, v36, v29\n v_fma_f32 v45, v32, v36, v45\n v_fma_f32 v49, v32, v36, v49\n" \
    "v_fma_f32 v9, v32, v36, v9\n v_fma_f32 v13, v32, v36, v13\n v_fma_f32 v17, v32, v36, v17\n v_fma_f32 v21, v32, v36, v21\n v_fma_f32 v25, v32, v36, v25\n v_fma_f32 v29, v32, v36, v29\n v_fma_f32 v45, v32, v36, v45\n v_fma_f32 v49, v32, v36, v49\n"
// one multiplicand and the accumulator in one bank (v32, v8+4k = bank 0), the other multiplicand in bank 1
#define FMA_2SAME_ACC \
    "v_fma_f32 v8, v32, v33, v8\n v_fma_f32 v12, v32, v33, v12\n v_fma_f32 v16, v32, v33, v16\n v_fma_f32 v20, v32, v33, v20\n v_fma_f32 v24, v32, v33, v24\n v_fma_f32 v28, v32, v33, v28\n v_fma_f32 v44, v32, v33, v44\n v_fma_f32 v48, v32, v33, v48\n" \
    "v_fma_f32 v8, v32, v33, v8\n v_fma_f32 v12, v32, v33, v12\n v_fma_f32 v16, v32, v33, v16\n v_fma_f32 v20, v32, v33, v20\n v_fma_f32 v24, v32, v33, v24\n v_fma_f32 v28, v32, v33, v28\n v_fma_f32 v44, v32, v33, v44\n v_fma_f32 v48, v32, v33, v48\n"
// packed: the three 64-bit sources spread as evenly as four banks allow ({0,1},{2,3},{0,1}) against all three on {0,1}
#define PKFMA_SPREAD \
    "v_pk_fma_f32 v[8:9], v[32:33], v[34:35], v[8:9]\n v_pk_fma_f32 v[12:13], v[32:33], v[34:35], v[12:13]\n v_pk_fma_f32 v[16:17], v[32:33], v[34:35], v[16:17]\n v_pk_fma_f32 v[20:21], v[32:33], v[34:35], v[20:21]\n" \
    "v_pk_fma_f32 v[24:25], v[32:33], v[34:35], v[24:25]\n v_pk_fma_f32 v[28:29], v[32:33], v[34:35], v[28:29]\n v_pk_fma_f32 v[44:45], v[32:33], v[34:35], v[44:45]\n v_pk_fma_f32 v[48:49], v[32:33], v[34:35], v[48:49]\n"
#define PKFMA_SAME \
    "v_pk_fma_f32 v[8:9], v[32:33], v[36:37], v[8:9]\n v_pk_fma_f32 v[12:13], v[32:33], v[36:37], v[12:13]\n v_pk_fma_f32 v[16:17], v[32:33], v[36:37], v[16:17]\n v_pk_fma_f32 v[20:21], v[32:33], v[36:37], v[20:21]\n" \
    "v_pk_fma_f32 v[24:25], v[32:33], v[36:37], v[24:25]\n v_pk_fma_f32 v[28:29], v[32:33], v[36:37], v[28:29]\n v_pk_fma_f32 v[44:45], v[32:33], v[36:37], v[44:45]\n v_pk_fma_f32 v[48:49], v[32:33], v[36:37], v[48:49]\n"
// 3 distinct banks, dst rotating
#define FMA_3BANK \
    "v_fma_f32 v8, v33, v34, v8\n v_fma_f32 v12, v33, v34, v12\n v_fma_f32 v16, v33, v34, v16\n v_fma_f32 v20, v33, v34, v20\n v_fma_f32 v24, v33, v34, v24\n v_fma_f32 v28, v33, v34, v28\n v_fma_f32 v44, v33, v34, v44\n v_fma_f32 v48, v33, v34, v48\n" \
    "v_fma_f32 v8, v33, v34, v8\n v_fma_f32 v12, v33, v34, v12\n v_fma_f32 v16, v33, v34, v16\n v_fma_f32 v20, v33, v34, v20\n v_fma_f32 v24, v33, v34, v24\n v_fma_f32 v28, v33, v34, v28\n v_fma_f32 v44, v33, v34, v44\n v_fma_f32 v48, v33, v34, v48\n"
// constant / SGPR operands: v_fma with one SGPR source, v_mul by literal-free inline constant
#define FMA_SGPR \
    "v_fma_f32 v8, s20, v33, v8\n v_fma_f32 v9, s20, v33, v9\n v_fma_f32 v10, s20, v33, v10\n v_fma_f32 v11, s20, v33, v11\n v_fma_f32 v12, s20, v33, v12\n v_fma_f32 v13, s20, v33, v13\n v_fma_f32 v14, s20, v33, v14\n v_fma_f32 v15, s20, v33, v15\n" \
    "v_fma_f32 v16, s20, v33, v16\n v_fma_f32 v17, s20, v33, v17\n v_fma_f32 v18, s20, v33, v18\n v_fma_f32 v19, s20, v33, v19\n v_fma_f32 v20, s20, v33, v20\n v_fma_f32 v21, s20, v33, v21\n v_fma_f32 v22, s20, v33, v22\n v_fma_f32 v23, s20, v33, v23\n"
#define ADD_E32 \
    "v_add_f32_e32 v8, v32, v8\n v_add_f32_e32 v9, v33, v9\n v_add_f32_e32 v10, v34, v10\n v_add_f32_e32 v11, v35, v11\n v_add_f32_e32 v12, v32, v12\n v_add_f32_e32 v13, v33, v13\n v_add_f32_e32 v14, v34, v14\n v_add_f32_e32 v15, v35, v15\n" \
    "v_add_f32_e32 v16, v32, v16\n v_add_f32_e32 v17, v33, v17\n v_add_f32_e32 v18, v34, v18\n v_add_f32_e32 v19, v35, v19\n v_add_f32_e32 v20, v32, v20\n v_add_f32_e32 v21, v33, v21\n v_add_f32_e32 v22, v34, v22\n v_add_f32_e32 v23, v35, v23\n"
// dependent chain on one register
#define FMA_DEP8 \
    "v_fma_f32 v8, v32, v33, v8\n v_fma_f32 v8, v32, v33, v8\n v_fma_f32 v8, v32, v33, v8\n v_fma_f32 v8, v32, v33, v8\n v_fma_f32 v8, v32, v33, v8\n v_fma_f32 v8, v32, v33, v8\n v_fma_f32 v8, v32, v33, v8\n v_fma_f32 v8, v32, v33, v8\n"
#define DPP8 \
    "v_mov_b32_dpp v8, v32 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v9, v33 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v10, v34 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v11, v35 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
    "v_mov_b32_dpp v12, v32 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v13, v33 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v14, v34 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v15, v35 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define DPPROW8 \
    "v_mov_b32_dpp v8, v32 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v9, v33 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v10, v34 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v11, v35 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
    "v_mov_b32_dpp v12, v32 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v13, v33 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v14, v34 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v15, v35 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
// LDS read stream beside VALU: 1 ds_read_b128 per 8 fma (addresses in v56, data to v60..63, never waited inside the body)
#define LDSMIX \
    "ds_read_b128 v[44:47], v56\n" "v_fma_f32 v8, v32, v33, v8\n v_fma_f32 v9, v32, v33, v9\n v_fma_f32 v10, v32, v33, v10\n v_fma_f32 v11, v32, v33, v11\n v_fma_f32 v12, v32, v33, v12\n v_fma_f32 v13, v32, v33, v13\n v_fma_f32 v14, v32, v33, v14\n v_fma_f32 v15, v32, v33, v15\n" \
    "ds_read_b128 v[48:51], v56 offset:48\n" "v_fma_f32 v16, v32, v33, v16\n v_fma_f32 v17, v32, v33, v17\n v_fma_f32 v18, v32, v33, v18\n v_fma_f32 v19, v32, v33, v19\n v_fma_f32 v20, v32, v33, v20\n v_fma_f32 v21, v32, v33, v21\n v_fma_f32 v22, v32, v33, v22\n v_fma_f32 v23, v32, v33, v23\n" \
    "s_waitcnt lgkmcnt(0)\n"

// what a scalar instruction costs the wave that issues it: 16 fma alone, then the same with an s_add / s_mul / s_cmp+s_cselect
// between every two of them
#define FMA2(a, b) "v_fma_f32 v" #a ", v32, v33, v" #a "\n v_fma_f32 v" #b ", v32, v33, v" #b "\n"
#define VS16_ADD FMA2(8, 9) "s_add_i32 s20, s20, 1\n" FMA2(10, 11) "s_add_i32 s21, s21, 1\n" FMA2(12, 13) "s_add_i32 s20, s20, 1\n" FMA2(14, 15) "s_add_i32 s21, s21, 1\n" \
                 FMA2(16, 17) "s_add_i32 s20, s20, 1\n" FMA2(18, 19) "s_add_i32 s21, s21, 1\n" FMA2(20, 21) "s_add_i32 s20, s20, 1\n" FMA2(22, 23) "s_add_i32 s21, s21, 1\n"
#define VS16_MUL FMA2(8, 9) "s_mulk_i32 s20, 3\n" FMA2(10, 11) "s_mulk_i32 s21, 3\n" FMA2(12, 13) "s_mulk_i32 s20, 3\n" FMA2(14, 15) "s_mulk_i32 s21, 3\n" \
                 FMA2(16, 17) "s_mulk_i32 s20, 3\n" FMA2(18, 19) "s_mulk_i32 s21, 3\n" FMA2(20, 21) "s_mulk_i32 s20, 3\n" FMA2(22, 23) "s_mulk_i32 s21, 3\n"
#define VS16_4 FMA2(8, 9) "s_add_i32 s20, s20, 1\n s_cmp_lt_i32 s20, 6\n s_cselect_b32 s21, 0, -6\n s_add_i32 s20, s20, s21\n" FMA2(10, 11) "s_add_i32 s20, s20, 1\n s_cmp_lt_i32 s20, 6\n s_cselect_b32 s21, 0, -6\n s_add_i32 s20, s20, s21\n" \
               FMA2(12, 13) "s_add_i32 s20, s20, 1\n s_cmp_lt_i32 s20, 6\n s_cselect_b32 s21, 0, -6\n s_add_i32 s20, s20, s21\n" FMA2(14, 15) "s_add_i32 s20, s20, 1\n s_cmp_lt_i32 s20, 6\n s_cselect_b32 s21, 0, -6\n s_add_i32 s20, s20, s21\n" \
               FMA2(16, 17) "s_add_i32 s20, s20, 1\n s_cmp_lt_i32 s20, 6\n s_cselect_b32 s21, 0, -6\n s_add_i32 s20, s20, s21\n" FMA2(18, 19) "s_add_i32 s20, s20, 1\n s_cmp_lt_i32 s20, 6\n s_cselect_b32 s21, 0, -6\n s_add_i32 s20, s20, s21\n" \
               FMA2(20, 21) "s_add_i32 s20, s20, 1\n s_cmp_lt_i32 s20, 6\n s_cselect_b32 s21, 0, -6\n s_add_i32 s20, s20, s21\n" FMA2(22, 23) "s_add_i32 s20, s20, 1\n s_cmp_lt_i32 s20, 6\n s_cselect_b32 s21, 0, -6\n s_add_i32 s20, s20, s21\n"
#define VS16_WAIT FMA2(8, 9) "s_waitcnt lgkmcnt(0)\n" FMA2(10, 11) "s_waitcnt lgkmcnt(0)\n" FMA2(12, 13) "s_waitcnt lgkmcnt(0)\n" FMA2(14, 15) "s_waitcnt lgkmcnt(0)\n" \
                  FMA2(16, 17) "s_waitcnt lgkmcnt(0)\n" FMA2(18, 19) "s_waitcnt lgkmcnt(0)\n" FMA2(20, 21) "s_waitcnt lgkmcnt(0)\n" FMA2(22, 23) "s_waitcnt lgkmcnt(0)\n"

template <int OP>
__global__ __launch_bounds__(1024) void k(unsigned long long *out, int reps, float seed)
{
    __shared__ float lds[64 * 12 * 4 + 64];
    for (int i = threadIdx.x; i < 64 * 12 * 4 + 64; i += blockDim.x) lds[i] = seed;
    __syncthreads();
    const unsigned lds_addr = (threadIdx.x & 63) * 48u;
    // initialise the registers the bodies read
    asm volatile("v_mov_b32 v32, %0\n v_mov_b32 v33, %1\n v_mov_b32 v34, %0\n v_mov_b32 v35, %1\n v_mov_b32 v36, %0\n v_mov_b32 v37, %1\n"
                 "v_mov_b32 v40, %2\n v_mov_b32 v41, %2\n v_mov_b32 v42, %2\n v_mov_b32 v43, %2\n v_mov_b32 v56, %3\n"
                 "v_mov_b32 v8, 0\n v_mov_b32 v9, 0\n v_mov_b32 v10, 0\n v_mov_b32 v11, 0\n v_mov_b32 v12, 0\n v_mov_b32 v13, 0\n v_mov_b32 v14, 0\n v_mov_b32 v15, 0\n"
                 "v_mov_b32 v16, 0\n v_mov_b32 v17, 0\n v_mov_b32 v18, 0\n v_mov_b32 v19, 0\n v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n"
                 "v_mov_b32 v24, 0\n v_mov_b32 v28, 0\n v_mov_b32 v44, 0\n v_mov_b32 v48, 0\n s_mov_b32 s20, 0x3f7ff000\n"
                 :: "v"(0.999f + seed * 1e-9f), "v"(1e-6f * seed), "v"(-0.5f - seed * 1e-9f), "v"(lds_addr) : CLOB, "s20");
    unsigned long long t0 = 0, t1 = 0;
    for (int pass = 0; pass < 2; pass++) {            // pass 0 warms the instruction cache
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
        for (int r = 0; r < reps; r++) {
            if (OP == 0) asm volatile(FMA16(32, 33) FMA16(34, 35) ::: CLOB);
            if (OP == 1) asm volatile(MUL16 MUL16 ::: CLOB);
            if (OP == 2) asm volatile(FMAC16 FMAC16 ::: CLOB);
            if (OP == 3) asm volatile(PKFMA8 PKFMA8 PKFMA8 PKFMA8 ::: CLOB);
            if (OP == 4) asm volatile(PKADD8 PKADD8 PKADD8 PKADD8 ::: CLOB);
            if (OP == 5) asm volatile(PKMUL8 PKMUL8 PKMUL8 PKMUL8 ::: CLOB);
            if (OP == 6) asm volatile(EXP8 EXP8 EXP8 EXP8 ::: CLOB);
            if (OP == 7) asm volatile(SQRT8 SQRT8 SQRT8 SQRT8 ::: CLOB);
            if (OP == 8) asm volatile(RCP8 RCP8 RCP8 RCP8 ::: CLOB);
            if (OP == 9) asm volatile(TAPMIX TAPMIX ::: CLOB);                  // 28 inst: 4 exp + 24 fma
            if (OP == 10) asm volatile(TAPCLUSTER ::: CLOB);                    // 28 inst: 4 exp + 24 fma
            if (OP == 11) asm volatile(FMA_SAMEBANK FMA_SAMEBANK ::: CLOB);
            if (OP == 12) asm volatile(FMA_3BANK FMA_3BANK ::: CLOB);
            if (OP == 13) asm volatile(FMA_SGPR FMA_SGPR ::: CLOB, "s20");
            if (OP == 14) asm volatile(ADD_E32 ADD_E32 ::: CLOB);
            if (OP == 15) asm volatile(FMA_DEP8 FMA_DEP8 FMA_DEP8 FMA_DEP8 ::: CLOB);
            if (OP == 16) asm volatile(DPP8 DPP8 DPP8 DPP8 ::: CLOB);
            if (OP == 17) asm volatile(DPPROW8 DPPROW8 DPPROW8 DPPROW8 ::: CLOB);
            if (OP == 18) asm volatile(LDSMIX LDSMIX ::: CLOB, "memory");       // 32 fma + 4 ds_read_b128
            if (OP == 19) asm volatile(VS16_ADD VS16_ADD ::: CLOB, "s20", "s21", "scc");
            if (OP == 20) asm volatile(VS16_MUL VS16_MUL ::: CLOB, "s20", "s21", "scc");
            if (OP == 21) asm volatile(VS16_4 VS16_4 ::: CLOB, "s20", "s21", "scc");
            if (OP == 22) asm volatile(VS16_WAIT VS16_WAIT ::: CLOB, "memory");
            if (OP == 23) asm volatile(FMA_2SAME FMA_2SAME ::: CLOB);
            if (OP == 24) asm volatile(FMA_2SAME_ACC FMA_2SAME_ACC ::: CLOB);
            if (OP == 25) asm volatile(PKFMA_SPREAD PKFMA_SPREAD PKFMA_SPREAD PKFMA_SPREAD ::: CLOB);
            if (OP == 26) asm volatile(PKFMA_SAME PKFMA_SAME PKFMA_SAME PKFMA_SAME ::: CLOB);
        }
        t1 = __builtin_amdgcn_s_memtime();
    }
    float sink;
    asm volatile("v_add_f32 %0, v8, v9\n v_add_f32 %0, %0, v24" : "=v"(sink) :: CLOB);
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2] = t1 - t0;
        out[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2 + 1] = (unsigned long long)sink;
    }
}

static unsigned long long *g_out = nullptr;

template <int OP>
void run(const char *name, int insts)
{
    unsigned long long *d = g_out;
    printf("%-52s", name);
    const int wps[4] = { 1, 2, 3, 4 };
    const int reps = 400;
    for (int wi = 0; wi < 4; wi++) {
        const int waves = 4 * wps[wi];
        for (int warm = 0; warm < 3; warm++) hipLaunchKernelGGL(k<OP>, dim3(256), dim3(64 * waves), 0, 0, d, reps, 1.0f);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * waves * 2);
        (void)hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::vector<double> c;
        for (int i = 0; i < 256 * waves; i++) c.push_back((double)h[2 * i] / ((double)reps * insts));
        std::sort(c.begin(), c.end());
        // per wave cycles/inst; per SIMD issue interval = that / waves per SIMD
        printf("  w%d %6.2f (/SIMD %5.2f)", wps[wi], c[c.size() / 2], c[c.size() / 2] / wps[wi]);
    }
    printf("   cycles per inst\n");
}

int main()
{
    // clocks up
    (void)hipMalloc(&g_out, 256 * 16 * 2 * sizeof(unsigned long long));
    for (int i = 0; i < 50; i++) hipLaunchKernelGGL(k<0>, dim3(256), dim3(1024), 0, 0, g_out, 2000, 1.0f);
    (void)hipDeviceSynchronize();
    run<0>("v_fma_f32 x32 independent (16 acc)", 32);
    run<1>("v_mul_f32 x32", 32);
    run<2>("v_fmac_f32 x32", 32);
    run<14>("v_add_f32_e32 x32", 32);
    run<11>("v_fma_f32 all operands same bank", 32);
    run<12>("v_fma_f32 three banks", 32);
    run<23>("v_fma_f32 both multiplicands in one bank", 32);
    run<24>("v_fma_f32 multiplicand + accumulator in one bank", 32);
    run<13>("v_fma_f32 sgpr source", 32);
    run<15>("v_fma_f32 dependent chain", 32);
    run<3>("v_pk_fma_f32 x32", 32);
    run<25>("v_pk_fma_f32 sources on banks {0,1},{2,3},{0,1}", 32);
    run<26>("v_pk_fma_f32 all sources on banks {0,1}", 32);
    run<4>("v_pk_add_f32 x32", 32);
    run<5>("v_pk_mul_f32 x32", 32);
    run<6>("v_exp_f32 x32", 32);
    run<7>("v_sqrt_f32 x32", 32);
    run<8>("v_rcp_f32 x32", 32);
    run<9>("tap mix: 1 exp + 6 fma interleaved (x4 = 28)", 28);
    run<10>("tap cluster: 4 exp then 24 fma (28)", 28);
    run<16>("v_mov_b32_dpp wave_shr/shl x32", 32);
    run<17>("v_mov_b32_dpp row_shr/shl x32", 32);
    run<18>("32 fma + 4 ds_read_b128 (+wait) (count 38)", 38);
    run<19>("32 fma + 16 s_add_i32 (per fma: count 32)", 32);
    run<20>("32 fma + 16 s_mulk_i32 (per fma)", 32);
    run<21>("32 fma + 16 x {add,cmp,cselect,add} (per fma)", 32);
    run<22>("32 fma + 16 s_waitcnt (satisfied) (per fma)", 32);
    return 0;
}
