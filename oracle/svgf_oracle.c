/*
 * svgf_oracle.c — CPU restatement of the reference SVGF denoiser.  TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * What it restates: reference `src/denoise.cu` (ZheyuanXie/CUDA-Path-Tracer-Denoising), function by function:
 *   svgf_oracle_atrous       <- ATrousFilter      src/denoise.cu:77-170
 *   reproj_valid             <- isReprjValid      src/denoise.cu:172-182
 *   svgf_oracle_backproject  <- BackProjection    src/denoise.cu:185-317
 *   (variance fill)          <- EstimateVariance  src/denoise.cu:320-329
 *   (debug view)             <- DebugView<T>      src/denoise.cu:331-340
 *   svgf_oracle_view_matrix  <- GetViewMatrix     src/denoise.cu:342-347  (glm::inverse, glm 0.9.6.3
 *                                                 external/include/glm/detail/type_mat4x4.inl:37-92)
 *   svgf_oracle_denoise      <- denoise           src/denoise.cu:349-402
 *   svgf_oracle_create/reset <- denoiseInit       src/denoise.cu:31-61
 * It is written from the behaviour, not transliterated: plain C arrays, explicit float/double
 * conversions where the reference's bare literals promote to double (src/denoise.cu:121,138,143-145,159,196,252),
 * glm's evaluation order where it decides rounding (dot(vec3) = (x+y)+z, func_geometric.inl:64-72;
 * mat4*vec4 = (m0 v0 + m1 v1) + (m2 v2 + m3 v3), type_mat4x4.inl:617-628).
 *
 * Deliberate definitions where the reference is undefined (last paragraph):
 *   - variance race: ORACLE_VARIANCE_SNAPSHOT is the contract (reads see pre-launch values);
 *     ORACLE_VARIANCE_INPLACE reproduces a sequential in-place execution for information.
 *   - a NaN reprojected coordinate is treated as "no valid history" (the reference would index texel (int)NaN).
 *   - buffers the reference leaves uninitialised (history_length_update, color_history, color_acc,
 *     gbuffer_prev; src/denoise.cu:43,52,53,56) are zero-initialised.
 *   - min/max follow the GPU's fminf/fmaxf semantics (a NaN operand loses), as CUDA/HIP `min`/`max` do.
 *
 * PINNING STATUS: the reference ships no tests, goldens or fixtures for this path.  The oracle is
 * pinned against outputs of the reference's own src/denoise.cu, built for gfx950 by oracle/ref/Makefile
 * (hipify-perl from the image + two mechanical fix-ups of non-path headers, see oracle/ref/README.md) and RUN on an
 * MI355X; those outputs are committed under tests/golden/ref_gpu/ with the generating script
 * (tests/golden/make_ref_gpu_goldens.py).  tests/test_oracle_vs_reference.py checks this file against them.
 * Race-free configurations (uniform variance: temporal off, or first frame; and the temporal pass itself) are
 * compared exactly to tolerance; raced configurations are compared statistically.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC svgf_oracle.c -o libsvgf_oracle.so -lm
 */
#include "svgf_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------------ */
/* constants (src/denoise.cu:82-91)                                             */
/* ------------------------------------------------------------------------------------------------ */

static const double kBinomial5[5] = { 1.0, 4.0, 6.0, 4.0, 1.0 };   /* h = outer([1 4 6 4 1]/16) */
static const double kBinomial3[3] = { 1.0, 2.0, 1.0 };             /* gaussian = outer([1 2 1]/4) */

static float atrous_h(int i, int j)   /* i,j in [-2,2]; all 25 values are exact in fp32 */
{
    return (float)((kBinomial5[i + 2] * kBinomial5[j + 2]) / 256.0);
}
static float gauss3(int i, int j)     /* i,j in [-1,1] */
{
    return (float)((kBinomial3[i + 1] * kBinomial3[j + 1]) / 16.0);
}

/* luminance with the reference's double promotion (src/denoise.cu:121,138,196) */
static float luminance(const float *c)
{
    double l = 0.2126 * (double)c[0] + 0.7152 * (double)c[1];
    l = l + 0.0722 * (double)c[2];
    return (float)l;
}

/* glm::distance(vec3,vec3) = sqrt(dot(d,d)), dot = (x+y)+z */
static float dist3(const float *a, const float *b)
{
    float dx = b[0] - a[0], dy = b[1] - a[1], dz = b[2] - a[2];
    float s = dx * dx + dy * dy;
    s = s + dz * dz;
    return sqrtf(s);
}

/* ------------------------------------------------------------------------------------------------ */
/* ATrousFilter                                                                                       */
/* ------------------------------------------------------------------------------------------------ */

static void atrous_pixel(int x, int y, const float *colorin, float *colorout, const float *var_rd, float *var_wr,
                         const SvgfGBufferTexel *g, int W, int H, int step, int is_last,
                         float sigma_c, float sigma_n, float sigma_x, int blur_variance, int addcolor)
{
    const int p = x + y * W;

    /* centre variance: optional 3x3 gaussian with out-of-image taps dropped and renormalised (:102-118) */
    float var;
    if (blur_variance) {
        float sum = 0.0f, sumw = 0.0f;
        for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++) {
                int lx = x + dx, ly = y + dy;
                if (lx >= 0 && ly >= 0 && lx < W && ly < H) {
                    float gw = gauss3(dx, dy);
                    sum += gw * var_rd[lx + ly * W];
                    sumw += gw;
                }
            }
        var = fmaxf(sum / sumw, 0.0f);
    } else {
        var = fmaxf(var_rd[p], 0.0f);
    }

    const float lp = luminance(&colorin[3 * p]);
    const float *pp = g[p].position;
    const float *np = g[p].normal;

    /* the three denominators (:143-145); `+ 1e-6` is a double literal */
    const double den_l = (double)(sqrtf(var) * sigma_c) + 1e-6;
    const double den_n = (double)sigma_n + 1e-6;
    const double den_x = (double)sigma_x + 1e-6;

    float csum[3] = { 0.0f, 0.0f, 0.0f };
    float vsum = 0.0f, wsum = 0.0f, w2sum = 0.0f;

    for (int i = -2; i <= 2; i++) {          /* i: x offset (outer), j: y offset (inner)  (:130-133) */
        for (int j = -2; j <= 2; j++) {
            int xq = x + step * i, yq = y + step * j;
            if (xq < 0 || xq >= W || yq < 0 || yq >= H) continue;
            int q = xq + yq * W;
            float lq = luminance(&colorin[3 * q]);
            float dl = fabsf(lq - lp);                                   /* glm::distance(float,float) */
            float wl = expf((float)(-(double)dl / den_l));
            float wn = fminf(1.0f, expf((float)(-(double)dist3(np, g[q].normal) / den_n)));
            float wx = fminf(1.0f, expf((float)(-(double)dist3(pp, g[q].position) / den_x)));
            float w = atrous_h(i, j) * wl;
            w = w * wn;
            w = w * wx;
            wsum += w;
            w2sum += w * w;
            csum[0] += colorin[3 * q + 0] * w;
            csum[1] += colorin[3 * q + 1] * w;
            csum[2] += colorin[3 * q + 2] * w;
            vsum += (var_rd[q] * w) * w;
        }
    }

    float o[3];
    if ((double)wsum > 10e-6) {               /* NaN compares false -> falls through (:159-164) */
        o[0] = csum[0] / wsum; o[1] = csum[1] / wsum; o[2] = csum[2] / wsum;
        var_wr[p] = vsum / w2sum;
    } else {
        o[0] = colorin[3 * p]; o[1] = colorin[3 * p + 1]; o[2] = colorin[3 * p + 2];
        if (var_wr != var_rd) var_wr[p] = var_rd[p];   /* reference leaves variance[p] untouched */
    }
    if (is_last && addcolor) {                 /* re-modulate by albedo * ialbedo (:166-168) */
        for (int k = 0; k < 3; k++) o[k] *= g[p].albedo[k] * g[p].ialbedo[k];
    }
    colorout[3 * p] = o[0]; colorout[3 * p + 1] = o[1]; colorout[3 * p + 2] = o[2];
}

void svgf_oracle_atrous(const float *colorin, float *colorout, const float *variance_in, float *variance_out,
                        const SvgfGBufferTexel *gbuffer, int W, int H, int level, int is_last,
                        float sigma_c, float sigma_n, float sigma_x, int blur_variance, int addcolor,
                        int inplace, int nthreads)
{
    const int step = 1 << level;               /* level starts at 1 => steps 2,4,8,16,32 (:98,386) */
    if (inplace) {
        /* sequential launch order: 8x8 blocks row-major, threads row-major inside a block (:354-357) */
        float *v = variance_out;
        if (variance_in != variance_out) memcpy(variance_out, variance_in, sizeof(float) * (size_t)W * H);
        for (int by = 0; by < H; by += 8)
            for (int bx = 0; bx < W; bx += 8)
                for (int ty = 0; ty < 8; ty++)
                    for (int tx = 0; tx < 8; tx++) {
                        int x = bx + tx, y = by + ty;
                        if (x < W && y < H)
                            atrous_pixel(x, y, colorin, colorout, v, v, gbuffer, W, H, step, is_last,
                                         sigma_c, sigma_n, sigma_x, blur_variance, addcolor);
                    }
        return;
    }
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
            atrous_pixel(x, y, colorin, colorout, variance_in, variance_out, gbuffer, W, H, step, is_last,
                         sigma_c, sigma_n, sigma_x, blur_variance, addcolor);
}

/* ------------------------------------------------------------------------------------------------ */
/* BackProjection                                                                                     */
/* ------------------------------------------------------------------------------------------------ */

/* isReprjValid (:172-182) for integer current pixel p and a float previous coordinate (qx,qy) that is
 * integral-valued when finite. Returns the previous pixel index or -1. */
/* pos_tol: SvgfParams::reproj_position_tol (extension; 0 = the reference's test) */
static float g_pos_tol = 0.0f;
#ifdef _OPENMP
#pragma omp threadprivate(g_pos_tol)
#endif
static int reproj_valid(int W, int H, int p, float qx, float qy,
                        const SvgfGBufferTexel *cur, const SvgfGBufferTexel *prev)
{
    if (!(qx == qx) || !(qy == qy)) return -1;                      /* NaN: defined as invalid */
    if (qx < 0.0f || qx >= (float)W || qy < 0.0f || qy >= (float)H) return -1;
    int q = (int)(qx + qy * (float)W);
    if (prev[q].geomId == -1 || prev[q].geomId != cur[p].geomId) return -1;
    if (dist3(prev[q].normal, cur[p].normal) > 1e-1f) return -1;
    if (g_pos_tol > 0.0f && !(dist3(prev[q].position, cur[p].position) <= g_pos_tol)) return -1;
    return q;
}

static void backproject_pixel(int x, int y, float *variance_out, const int *hl, int *hl_upd,
                              const float *mom_hist, const float *col_hist, float *mom_acc, float *col_acc,
                              const float *cur_col, const SvgfGBufferTexel *cur_g, const SvgfGBufferTexel *prev_g,
                              const float *M, int W, int H, float ca_min, float ma_min, float rsx, float rsy, float pos_tol)
{
    g_pos_tol = pos_tol;
    const int p = x + y * W;
    const int N = hl[p];                                   /* at the CURRENT pixel (:194) */
    const float *s = &cur_col[3 * p];
    const float lum = luminance(s);

    if (N > 0 && cur_g[p].geomId != -1) {
        /* previous-frame view space, glm mat4*vec4 order; M is column-major M[c*4+r] (:201) */
        const float *P = cur_g[p].position;
        float vs[3];
        for (int r = 0; r < 3; r++) {
            float a0 = M[0 * 4 + r] * P[0] + M[1 * 4 + r] * P[1];
            float a1 = M[2 * 4 + r] * P[2] + M[3 * 4 + r] * 1.0f;
            vs[r] = a0 + a1;
        }
        /* no tan(fov), no aspect (:202-207) */
        float clipx = vs[0] / vs[2], clipy = vs[1] / vs[2];
        /* SvgfParams::reproj_scale: not in the reference; 0 keeps its mapping */
        if (rsx > 0.0f) clipx = clipx / rsx;
        if (rsy > 0.0f) clipy = clipy / rsy;
        float ndcx = -clipx * 0.5f + 0.5f, ndcy = -clipy * 0.5f + 0.5f;
        float prevx = ndcx * (float)W - 0.5f, prevy = ndcy * (float)H - 0.5f;
        float fx = floorf(prevx), fy = floorf(prevy);
        float fracx = prevx - fx, fracy = prevy - fy;

        int valid = (fx >= 0.0f && fy >= 0.0f && fx < (float)W && fy < (float)H);
        int q4[4];
        static const int ox[4] = { 0, 1, 0, 1 }, oy[4] = { 0, 0, 1, 1 };
        for (int k = 0; k < 4; k++) {
            q4[k] = reproj_valid(W, H, p, fx + (float)ox[k], fy + (float)oy[k], cur_g, prev_g);
            valid = valid && (q4[k] >= 0);
        }

        float pc[3] = { 0.0f, 0.0f, 0.0f }, pm[2] = { 0.0f, 0.0f }, plen = 0.0f;

        if (valid) {                                       /* bilinear (:234-259) */
            float sumw = 0.0f;
            float w[4] = { (1 - fracx) * (1 - fracy), fracx * (1 - fracy), (1 - fracx) * fracy, fracx * fracy };
            for (int k = 0; k < 4; k++) {
                int q = q4[k];
                pc[0] += w[k] * col_hist[3 * q]; pc[1] += w[k] * col_hist[3 * q + 1]; pc[2] += w[k] * col_hist[3 * q + 2];
                pm[0] += w[k] * mom_hist[2 * q]; pm[1] += w[k] * mom_hist[2 * q + 1];
                plen += w[k] * (float)hl[q];
                sumw += w[k];
            }
            if ((double)sumw >= 0.01) {
                pc[0] /= sumw; pc[1] /= sumw; pc[2] /= sumw;
                pm[0] /= sumw; pm[1] /= sumw;
                plen /= sumw;
            }
        } else {                                           /* 3x3 box fallback around floor (:262-286) */
            float cnt = 0.0f;
            for (int yy = -1; yy <= 1; yy++)
                for (int xx = -1; xx <= 1; xx++) {
                    int q = reproj_valid(W, H, p, fx + (float)xx, fy + (float)yy, cur_g, prev_g);
                    if (q >= 0) {
                        pc[0] += col_hist[3 * q]; pc[1] += col_hist[3 * q + 1]; pc[2] += col_hist[3 * q + 2];
                        pm[0] += mom_hist[2 * q]; pm[1] += mom_hist[2 * q + 1];
                        plen += (float)hl[q];
                        cnt += 1.0f;
                    }
                }
            if (cnt > 0.0f) {
                pc[0] /= cnt; pc[1] /= cnt; pc[2] /= cnt;
                pm[0] /= cnt; pm[1] /= cnt;
                plen /= cnt;
                valid = 1;
            }
        }

        if (valid) {
            float ca = fmaxf(1.0f / (float)(N + 1), ca_min);      /* alpha on the CURRENT side for colour (:297) */
            float ma = fmaxf(1.0f / (float)(N + 1), ma_min);      /* alpha on the HISTORY side for moments (:300-301) */
            hl_upd[p] = (int)plen + 1;                            /* truncation, uncapped (:294) */
            for (int k = 0; k < 3; k++) col_acc[3 * p + k] = s[k] * ca + pc[k] * (1.0f - ca);
            float m1 = ma * pm[0] + (1.0f - ma) * lum;
            float m2 = ma * pm[1] + ((1.0f - ma) * lum) * lum;
            mom_acc[2 * p] = m1; mom_acc[2 * p + 1] = m2;
            float v = m2 - m1 * m1;
            variance_out[p] = v > 0.0f ? v : 0.0f;
            return;
        }
    }
    /* no usable history (:311-315) */
    hl_upd[p] = 1;
    col_acc[3 * p] = s[0]; col_acc[3 * p + 1] = s[1]; col_acc[3 * p + 2] = s[2];
    mom_acc[2 * p] = lum; mom_acc[2 * p + 1] = lum * lum;
    variance_out[p] = 100.0f;
}

void svgf_oracle_backproject_ex(float *variance_out, const int *history_length, int *history_length_update,
                                const float *moment_history, const float *color_history,
                                float *moment_acc, float *color_acc,
                                const float *current_color, const SvgfGBufferTexel *current_gbuffer,
                                const SvgfGBufferTexel *prev_gbuffer, const float prev_viewmat[16],
                                int W, int H, float color_alpha_min, float moment_alpha_min, int nthreads,
                                float reproj_sx, float reproj_sy, float pos_tol)
{
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
            backproject_pixel(x, y, variance_out, history_length, history_length_update, moment_history,
                              color_history, moment_acc, color_acc, current_color, current_gbuffer,
                              prev_gbuffer, prev_viewmat, W, H, color_alpha_min, moment_alpha_min, reproj_sx, reproj_sy, pos_tol);
}

/* SvgfParams::spatial_variance_frames (extension; the reference's EstimateVariance is a TODO stub,
 * src/denoise.cu:320-329).  Pixels whose updated history is shorter than K frames: variance from the luminance moments of
 * the 7x7 neighbourhood, taps accepted by the reference's own consistency predicate (geomId, normal distance <= 0.1;
 * src/denoise.cu:176-180), centre always; sums in raster order in fp32; boosted by max(1, 4 / history length)
 * (Schied et al. 2017, section 4.2). */
void svgf_oracle_spatial_variance(float *variance, const float *moment_acc, const int *history_length_update,
                                  const SvgfGBufferTexel *g, int W, int H, int K, int nthreads)
{
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const int p = x + y * W;
            const int hl = history_length_update[p];
            if (hl >= K) continue;
            float s1 = 0.0f, s2 = 0.0f, cnt = 0.0f;
            for (int yy = -3; yy <= 3; yy++)
                for (int xx = -3; xx <= 3; xx++) {
                    const int qx = x + xx, qy = y + yy;
                    if (qx < 0 || qx >= W || qy < 0 || qy >= H) continue;
                    const int q = qx + qy * W;
                    if (q != p) {
                        if (g[q].geomId != g[p].geomId) continue;
                        if (!(dist3(g[q].normal, g[p].normal) <= 1e-1f)) continue;
                    }
                    s1 += moment_acc[2 * q]; s2 += moment_acc[2 * q + 1]; cnt += 1.0f;
                }
            const float m1 = s1 / cnt, m2 = s2 / cnt;
            float v = m2 - m1 * m1;
            v = v > 0.0f ? v : 0.0f;
            const float boost = 4.0f / (float)(hl > 0 ? hl : 1);
            variance[p] = v * (boost > 1.0f ? boost : 1.0f);
        }
}

void svgf_oracle_backproject(float *variance_out, const int *history_length, int *history_length_update,
                             const float *moment_history, const float *color_history,
                             float *moment_acc, float *color_acc,
                             const float *current_color, const SvgfGBufferTexel *current_gbuffer,
                             const SvgfGBufferTexel *prev_gbuffer, const float prev_viewmat[16],
                             int W, int H, float color_alpha_min, float moment_alpha_min, int nthreads)
{   /* the reference's BackProjection */
    svgf_oracle_backproject_ex(variance_out, history_length, history_length_update, moment_history, color_history,
                               moment_acc, color_acc, current_color, current_gbuffer, prev_gbuffer, prev_viewmat, W, H,
                               color_alpha_min, moment_alpha_min, nthreads, 0.0f, 0.0f, 0.0f);
}

/* ------------------------------------------------------------------------------------------------ */
/* GetViewMatrix: 4x4 inverse by 2x2 sub-determinant (cofactor) expansion, column-major m[c*4+r].     */
/* Operation order follows glm 0.9.6.3 compute_inverse so that fp32 rounding agrees with the          */
/* reference host code (type_mat4x4.inl:37-92).                                                       */
/* ------------------------------------------------------------------------------------------------ */

static float det2(const float *m, int c0, int r0, int c1, int r1, int c2, int r2, int c3, int r3)
{   /* m[c0][r0]*m[c1][r1] - m[c2][r2]*m[c3][r3] */
    return m[c0 * 4 + r0] * m[c1 * 4 + r1] - m[c2 * 4 + r2] * m[c3 * 4 + r3];
}

static void invert4x4(const float *m, float *out)
{
    /* six rows of 2x2 minors; each "fac" is the 4-vector (a, a, b, c) of three minors */
    float fac[6][4];
    static const int rows[6][2] = { {2,3}, {1,3}, {1,2}, {0,3}, {0,2}, {0,1} };
    for (int k = 0; k < 6; k++) {
        int ra = rows[k][0], rb = rows[k][1];
        float a = det2(m, 2, ra, 3, rb, 3, ra, 2, rb);
        float b = det2(m, 1, ra, 3, rb, 3, ra, 1, rb);
        float c = det2(m, 1, ra, 2, rb, 2, ra, 1, rb);
        fac[k][0] = a; fac[k][1] = a; fac[k][2] = b; fac[k][3] = c;
    }
    float vec[4][4];                                   /* vec[r] = (m[1][r], m[0][r], m[0][r], m[0][r]) */
    for (int r = 0; r < 4; r++) {
        vec[r][0] = m[1 * 4 + r]; vec[r][1] = m[0 * 4 + r]; vec[r][2] = m[0 * 4 + r]; vec[r][3] = m[0 * 4 + r];
    }
    /* inv column c = va*fac[fa] - vb*fac[fb] + vc*fac[fc], then alternating signs */
    static const int comb[4][6] = {
        /* va fa  vb fb  vc fc */
        { 1, 0,  2, 1,  3, 2 },
        { 0, 0,  2, 3,  3, 4 },
        { 0, 1,  1, 3,  3, 5 },
        { 0, 2,  1, 4,  2, 5 } };
    float inv[16];
    for (int c = 0; c < 4; c++) {
        for (int r = 0; r < 4; r++) {
            float t = vec[comb[c][0]][r] * fac[comb[c][1]][r] - vec[comb[c][2]][r] * fac[comb[c][3]][r];
            t = t + vec[comb[c][4]][r] * fac[comb[c][5]][r];
            float sign = ((c + r) & 1) ? -1.0f : 1.0f;
            inv[c * 4 + r] = t * sign;
        }
    }
    float d0 = m[0] * inv[0 * 4 + 0], d1 = m[1] * inv[1 * 4 + 0], d2 = m[2] * inv[2 * 4 + 0], d3 = m[3] * inv[3 * 4 + 0];
    float det = (d0 + d1) + (d2 + d3);
    float rdet = 1.0f / det;
    for (int k = 0; k < 16; k++) out[k] = inv[k] * rdet;
}

void svgf_oracle_view_matrix(const SvgfCamera *cam, float out[16])
{
    float m[16];
    for (int r = 0; r < 3; r++) {
        m[0 * 4 + r] = cam->right[r]; m[1 * 4 + r] = cam->up[r]; m[2 * 4 + r] = cam->view[r]; m[3 * 4 + r] = cam->position[r];
    }
    m[3] = 0.0f; m[7] = 0.0f; m[11] = 0.0f; m[15] = 1.0f;
    invert4x4(m, out);
}

/* ------------------------------------------------------------------------------------------------ */
/* context + denoise()                                                                                */
/* ------------------------------------------------------------------------------------------------ */

struct oracle_ctx {
    int W, H, nthreads, variance_mode;
    long frames;                        /* svgf_oracle_denoise calls since create: the planes hold state once it is > 0 */
    float view_prev[16];                /* static glm::mat4, identity by default ctor, never reset (:15) */
    float *temp[2];                     /* vec3 ping-pong (:27) */
    int   *history_length, *history_length_update;
    float *moment_history, *moment_acc; /* vec2 */
    float *color_history, *color_acc;   /* vec3 */
    SvgfGBufferTexel *gbuffer_prev;
    float *variance, *variance_tmp;     /* variance_tmp: snapshot for ORACLE_VARIANCE_SNAPSHOT */
    float *variance_temporal;           /* copy of variance right after the temporal pass (state inspection) */
    float *in_local;                    /* the call's inputs, copied in row blocks by the threads that read them */
    SvgfGBufferTexel *g_local;          /* (placement only: on a two-socket host the caller's arrays sit on one node) */
};

/* rows zeroed by the thread that will later process them (static blocks): first touch places the pages next to it */
static void par_zero(void *dst, size_t bytes_per_row, int rows, int nthreads)
{
    int y;
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
    for (y = 0; y < rows; y++) memset((char *)dst + (size_t)y * bytes_per_row, 0, bytes_per_row);
}

static void zero_history(oracle_ctx *c)
{
    const size_t W = (size_t)c->W;
    const int H = c->H, nt = c->nthreads;
    par_zero(c->history_length, W * sizeof(int), H, nt);
    par_zero(c->history_length_update, W * sizeof(int), H, nt);
    par_zero(c->moment_history, W * 2 * sizeof(float), H, nt);
    par_zero(c->moment_acc, W * 2 * sizeof(float), H, nt);
    par_zero(c->variance, W * sizeof(float), H, nt);
    par_zero(c->color_history, W * 3 * sizeof(float), H, nt);
    par_zero(c->color_acc, W * 3 * sizeof(float), H, nt);
    par_zero(c->gbuffer_prev, W * sizeof(SvgfGBufferTexel), H, nt);
    par_zero(c->variance_temporal, W * sizeof(float), H, nt);
}

static void free_planes(oracle_ctx *c)
{
    free(c->temp[0]); free(c->temp[1]); free(c->history_length); free(c->history_length_update);
    free(c->moment_history); free(c->moment_acc); free(c->color_history); free(c->color_acc);
    free(c->gbuffer_prev); free(c->variance); free(c->variance_tmp); free(c->variance_temporal);
    free(c->in_local); free(c->g_local);
    c->temp[0] = c->temp[1] = NULL; c->history_length = c->history_length_update = NULL;
    c->moment_history = c->moment_acc = NULL; c->color_history = c->color_acc = NULL;
    c->gbuffer_prev = NULL; c->variance = c->variance_tmp = c->variance_temporal = NULL;
    c->in_local = NULL; c->g_local = NULL;
}

/* malloc, then zeroed in parallel by c->nthreads threads: the state keeps the reference's all-zero start (denoiseInit's
 * cudaMemset, :37-59) and every page is first touched by the thread that owns its rows */
static int alloc_planes(oracle_ctx *c)      /* 0, or -1 when a plane could not be allocated (everything is freed again) */
{
    const size_t W = (size_t)c->W, n = W * c->H;
    const int H = c->H, nt = c->nthreads;
    c->temp[0] = (float *)malloc(n * 3 * sizeof(float));
    c->temp[1] = (float *)malloc(n * 3 * sizeof(float));
    c->history_length = (int *)malloc(n * sizeof(int));
    c->history_length_update = (int *)malloc(n * sizeof(int));
    c->moment_history = (float *)malloc(n * 2 * sizeof(float));
    c->moment_acc = (float *)malloc(n * 2 * sizeof(float));
    c->color_history = (float *)malloc(n * 3 * sizeof(float));
    c->color_acc = (float *)malloc(n * 3 * sizeof(float));
    c->gbuffer_prev = (SvgfGBufferTexel *)malloc(n * sizeof(SvgfGBufferTexel));
    c->variance = (float *)malloc(n * sizeof(float));
    c->variance_tmp = (float *)malloc(n * sizeof(float));
    c->variance_temporal = (float *)malloc(n * sizeof(float));
    c->in_local = (float *)malloc(n * 3 * sizeof(float));
    c->g_local = (SvgfGBufferTexel *)malloc(n * sizeof(SvgfGBufferTexel));
    if (!c->temp[0] || !c->temp[1] || !c->history_length || !c->history_length_update || !c->moment_history || !c->moment_acc ||
        !c->color_history || !c->color_acc || !c->gbuffer_prev || !c->variance || !c->variance_tmp || !c->variance_temporal ||
        !c->in_local || !c->g_local) {
        free_planes(c);
        return -1;
    }
    par_zero(c->temp[0], W * 3 * sizeof(float), H, nt);
    par_zero(c->temp[1], W * 3 * sizeof(float), H, nt);
    par_zero(c->variance_tmp, W * sizeof(float), H, nt);
    par_zero(c->in_local, W * 3 * sizeof(float), H, nt);
    par_zero(c->g_local, W * sizeof(SvgfGBufferTexel), H, nt);
    zero_history(c);
    return 0;
}

oracle_ctx *svgf_oracle_create(int W, int H)
{
    if (W <= 0 || H <= 0) return NULL;
    oracle_ctx *c = (oracle_ctx *)calloc(1, sizeof(*c));
    if (!c) return NULL;
    c->W = W; c->H = H; c->nthreads = 1; c->variance_mode = ORACLE_VARIANCE_SNAPSHOT;
    for (int k = 0; k < 16; k++) c->view_prev[k] = (k % 5 == 0) ? 1.0f : 0.0f;
    if (alloc_planes(c) != 0) { free(c); return NULL; }
    return c;
}

void svgf_oracle_destroy(oracle_ctx *c)
{
    if (!c) return;
    free_planes(c);
    free(c);
}

void svgf_oracle_reset(oracle_ctx *c) { if (c) zero_history(c); }
/* Changing the thread count re-creates the planes (contents preserved would need a copy: the only caller sets it right after
 * svgf_oracle_create, before the first frame, so the state is the all-zero start either way) so that they are first touched
 * by the threads that will work on them. */
void svgf_oracle_set_threads(oracle_ctx *c, int n)
{
    if (!c) return;
    n = n > 0 ? n : 1;
    if (n == c->nthreads) return;
    c->nthreads = n;
    /* The planes are first-touched by the threads that will work on them (alloc_planes): re-placing them for the new thread
     * count is only done while they hold no state, i.e. before the first frame; later the count changes and the pages stay. */
    if (c->frames == 0) {
        free_planes(c);
        if (alloc_planes(c) != 0) { c->nthreads = 1; (void)alloc_planes(c); }
    }
}
void svgf_oracle_set_variance_mode(oracle_ctx *c, int m) { if (c) c->variance_mode = m; }

/* The reference's device-to-device copies (:366,391,396-398).  Row blocks dealt statically to the same threads that
 * process those rows in the kernels above: no serial section, and on a multi-socket host every page is first touched by
 * (hence placed next to) the thread that will read it. */
static void par_copy(void *dst, const void *src, size_t bytes_per_row, int rows, int nthreads)
{
    int y;
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
    for (y = 0; y < rows; y++)
        memcpy((char *)dst + (size_t)y * bytes_per_row, (const char *)src + (size_t)y * bytes_per_row, bytes_per_row);
}

int svgf_oracle_denoise(oracle_ctx *c, float *out, const float *in, const SvgfGBufferTexel *g,
                        const SvgfCamera *cam, const SvgfParams *p)
{
    const int W = c->W, H = c->H;
    const size_t n = (size_t)W * H;
    const int nt = c->nthreads;
    if (!c->in_local) return -1;        /* the planes could not be allocated */
    c->frames++;
    /* inputs next to the threads that read them (taps reach at most 64 rows beyond a thread's own block) */
    par_copy(c->in_local, in, (size_t)W * 3 * sizeof(float), H, nt);
    par_copy(c->g_local, g, (size_t)W * sizeof(SvgfGBufferTexel), H, nt);
    in = c->in_local;
    g = c->g_local;

    /* 1) temporal accumulation or constant variance (:360-371) */
    if (p->temporal_enable) {
        svgf_oracle_backproject_ex(c->variance, c->history_length, c->history_length_update, c->moment_history,
                                c->color_history, c->moment_acc, c->color_acc, in, g, c->gbuffer_prev,
                                c->view_prev, W, H, p->color_alpha, p->moment_alpha, c->nthreads,
                                p->reproj_scale[0], p->reproj_scale[1], p->reproj_position_tol);
        if (p->spatial_variance_frames > 0)
            svgf_oracle_spatial_variance(c->variance, c->moment_acc, c->history_length_update, g, W, H,
                                         p->spatial_variance_frames, c->nthreads);
        par_copy(c->color_history, c->color_acc, (size_t)W * 3 * sizeof(float), H, nt);
    } else {
        for (size_t k = 0; k < n; k++) c->variance[k] = 10.0f;
        par_copy(c->color_history, in, (size_t)W * 3 * sizeof(float), H, nt);
    }
    par_copy(c->variance_temporal, c->variance, (size_t)W * sizeof(float), H, nt);

    /* 2) debug views, pass-through, or the a-trous cascade (:373-394) */
    if (p->right_view_option == 1) {
        for (size_t k = 0; k < n; k++) { float v = (float)c->history_length[k] / 100.0f; out[3*k] = out[3*k+1] = out[3*k+2] = v; }
    } else if (p->right_view_option == 2) {
        for (size_t k = 0; k < n; k++) { float v = c->variance[k] / 0.1f; out[3*k] = out[3*k+1] = out[3*k+2] = v; }
    } else if (p->atrous_nlevel == 0 || !p->spatial_enable) {
        memcpy(out, c->color_history, n * 3 * sizeof(float));
    } else {
        const int addcolor = (p->sepcolor && p->addcolor);
        for (int level = 1; level <= p->atrous_nlevel; level++) {
            const float *src = (level == 1) ? c->color_history : c->temp[level % 2];
            float *dst = (level == p->atrous_nlevel) ? out : c->temp[(level + 1) % 2];
            /* dilation exponent: the reference's 2^level, or 2^(level-1) with SvgfParams::paper_steps (extension, f4) */
            const int lexp = p->paper_steps ? level - 1 : level;
            if (c->variance_mode == ORACLE_VARIANCE_INPLACE) {
                svgf_oracle_atrous(src, dst, c->variance, c->variance, g, W, H, lexp, level == p->atrous_nlevel,
                                   p->sigma_l, p->sigma_n, p->sigma_x, p->blur_variance, addcolor, 1, 1);
            } else {
                par_copy(c->variance_tmp, c->variance, (size_t)W * sizeof(float), H, nt);
                svgf_oracle_atrous(src, dst, c->variance_tmp, c->variance, g, W, H, lexp, level == p->atrous_nlevel,
                                   p->sigma_l, p->sigma_n, p->sigma_x, p->blur_variance, addcolor, 0, c->nthreads);
            }
            if (level == p->history_level) par_copy(c->color_history, dst, (size_t)W * 3 * sizeof(float), H, nt);
        }
    }

    /* 3) history rotation — executed in every mode (:396-399) */
    par_copy(c->gbuffer_prev, g, (size_t)W * sizeof(SvgfGBufferTexel), H, nt);
    par_copy(c->moment_history, c->moment_acc, (size_t)W * 2 * sizeof(float), H, nt);
    par_copy(c->history_length, c->history_length_update, (size_t)W * sizeof(int), H, nt);
    svgf_oracle_view_matrix(cam, c->view_prev);
    return 0;
}

int svgf_oracle_read_state(oracle_ctx *c, int which, void *dst, unsigned long long bytes)
{
    size_t n = (size_t)c->W * c->H;
    const void *src; size_t need;
    switch (which) {
    case SVGF_STATE_HISTORY_LENGTH:    src = c->history_length;    need = n * sizeof(int); break;
    case SVGF_STATE_MOMENTS:           src = c->moment_history;    need = n * 2 * sizeof(float); break;
    case SVGF_STATE_COLOR_HISTORY:     src = c->color_history;     need = n * 3 * sizeof(float); break;
    case SVGF_STATE_VARIANCE_TEMPORAL: src = c->variance_temporal; need = n * sizeof(float); break;
    case SVGF_STATE_COLOR_ACC:         src = c->color_acc;         need = n * 3 * sizeof(float); break;
    default: return -1;
    }
    if (bytes < need) return -1;
    memcpy(dst, src, need);
    return 0;
}
