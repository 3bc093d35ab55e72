/*
 * svgf_oracle.h — CPU restatement of the reference SVGF denoiser (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle: a plain-C restatement of what reference `src/denoise.cu` computes.
 * It is NOT part of the product.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it, and there only as the checker / the timed CPU baseline.
 * The product (libsvgf_hip.so) never links, loads or falls back to anything in oracle/.
 *
 * Pinning status: see the header of svgf_oracle.c.
 */
#ifndef SVGF_ORACLE_H_
#define SVGF_ORACLE_H_

#include "../include/svgf.h"   /* boundary structs only: SvgfGBufferTexel, SvgfCamera, SvgfParams */

#ifdef __cplusplus
extern "C" {
#endif

/* How reads of `variance` inside one a-trous launch see concurrent writes (reference src/denoise.cu:111,117,153,161
 * reads and writes the same buffer in place => a data race on a GPU). */
#define ORACLE_VARIANCE_SNAPSHOT 0   /* every read sees the pre-launch value (Jacobi).  THE PARITY CONTRACT. */
#define ORACLE_VARIANCE_INPLACE  1   /* sequential execution, 8x8 blocks row-major, threads row-major (Gauss-Seidel) */

typedef struct oracle_ctx oracle_ctx;

oracle_ctx *svgf_oracle_create(int width, int height);                 /* denoiseInit, src/denoise.cu:31-61 */
void        svgf_oracle_destroy(oracle_ctx *c);                        /* denoiseFree, src/denoise.cu:63-74 */
void        svgf_oracle_reset(oracle_ctx *c);                          /* free+init,   src/main.cpp:192-201 */
void        svgf_oracle_set_threads(oracle_ctx *c, int nthreads);      /* OpenMP rows (snapshot mode only); 1 = scalar */
void        svgf_oracle_set_variance_mode(oracle_ctx *c, int mode);

/* denoise(), src/denoise.cu:349-402.  Host pointers.  Returns 0. */
int svgf_oracle_denoise(oracle_ctx *c, float *out_rgb, const float *in_rgb,
                        const SvgfGBufferTexel *gbuffer, const SvgfCamera *cam, const SvgfParams *p);

/* same `which` codes as svgf_read_state (include/svgf.h) */
int svgf_oracle_read_state(oracle_ctx *c, int which, void *dst, unsigned long long bytes);

/* ---- kernel-level entry points (unit goldens) ---- */

/* ATrousFilter, src/denoise.cu:77-170.  variance_in is read (snapshot) and variance_out written;
 * pass the same pointer for both together with inplace=1 to get the literal in-place behaviour. */
void svgf_oracle_atrous(const float *colorin, float *colorout, const float *variance_in, float *variance_out,
                        const SvgfGBufferTexel *gbuffer, int W, int H, int level, int is_last,
                        float sigma_c, float sigma_n, float sigma_x, int blur_variance, int addcolor,
                        int inplace, int nthreads);

/* BackProjection, src/denoise.cu:185-317 (+ isReprjValid :172-182). */
void svgf_oracle_backproject(float *variance_out, const int *history_length, int *history_length_update,
                             const float *moment_history, const float *color_history,
                             float *moment_acc, float *color_acc,
                             const float *current_color, const SvgfGBufferTexel *current_gbuffer,
                             const SvgfGBufferTexel *prev_gbuffer, const float prev_viewmat[16],
                             int W, int H, float color_alpha_min, float moment_alpha_min, int nthreads);
/* the same with SvgfParams::reproj_scale and ::reproj_position_tol (extensions); (0, 0, 0) ==
 * svgf_oracle_backproject */
void svgf_oracle_backproject_ex(float *variance_out, const int *history_length, int *history_length_update,
                             const float *moment_history, const float *color_history,
                             float *moment_acc, float *color_acc,
                             const float *current_color, const SvgfGBufferTexel *current_gbuffer,
                             const SvgfGBufferTexel *prev_gbuffer, const float prev_viewmat[16],
                             int W, int H, float color_alpha_min, float moment_alpha_min, int nthreads,
                                float reproj_sx, float reproj_sy, float pos_tol);
/* SvgfParams::spatial_variance_frames (extension, f4): 7x7 spatial variance estimate for histories shorter than K */
void svgf_oracle_spatial_variance(float *variance, const float *moment_acc, const int *history_length_update,
                                  const SvgfGBufferTexel *g, int W, int H, int K, int nthreads);

/* GetViewMatrix, src/denoise.cu:342-347: inverse of the column-major matrix [right|up|view|position]. */
void svgf_oracle_view_matrix(const SvgfCamera *cam, float out_colmajor[16]);

#ifdef __cplusplus
}
#endif
#endif
