// svgf_kernels.hip — temporal accumulation, G-buffer split, strict a-trous gather, debug/copy kernels (gfx950).
//
// The temporal kernel and the strict gather kernel keep the reference's arithmetic order and double promotions
// (reference src/denoise.cu:121,138,143-145,159,196,252) with FMA contraction off, so that they agree with the CPU
// oracle to the last ulp except inside expf.  They are the correctness anchors; the bandwidth-shaped a-trous
// kernel lives in svgf_atrous_strip.hip.
#include "svgf_kernels.h"
#include "svgf_temporal.h"

#define SVGF_BLOCK 256

static inline int div_up(long long a, int b) { return (int)((a + b - 1) / b); }

// ----------------------------------------------------------------------------------------------------
// helpers
// ----------------------------------------------------------------------------------------------------

// luminance / distance in the reference's operation order: svgf_temporal.h (shared with the fused kernel)
__device__ __forceinline__ float lum_strict(float r, float g, float b) { return svgf_lum_strict(r, g, b); }
__device__ __forceinline__ float dist3_strict(float ax, float ay, float az, float bx, float by, float bz) { return svgf_dist3_strict(ax, ay, az, bx, by, bz); }

// ----------------------------------------------------------------------------------------------------
// temporal accumulation  (reference BackProjection src/denoise.cu:185-317, isReprjValid :172-182)
// One thread per pixel; the history taps are a data-dependent gather around the reprojected position, served
// by L1/L2 (neighbouring pixels reproject to neighbouring taps).  Also splits the 52-B texel into planes.
// ----------------------------------------------------------------------------------------------------

__device__ __forceinline__ int reproj_valid(const TemporalArgs &a, float qx, float qy, int gid, float nx, float ny, float nz)
{
    const int q = svgf_tap_index(a, qx, qy);                          // bounds (:173-176); NaN coordinate: defined as invalid
    if (q < 0) return -1;
    const int gq = a.gid_prev[q];
    if (gq == -1 || gq != gid) return -1;                             // (before the normal is fetched: most rejected taps end here)
    const float *n = a.nrm_prev + 3 * (size_t)q;
    return svgf_normals_close(n[0], n[1], n[2], nx, ny, nz) ? q : -1;
}

// SvgfParams::reproj_position_tol (f4 extension): the tap's previous-frame world position must lie within tol of the
// current pixel's
__device__ __forceinline__ int reproj_valid_pos(const TemporalArgs &a, int q, float px, float py, float pz)
{
    if (q < 0 || !(a.pos_tol > 0.0f)) return q;
    const float *pp = a.pos_prev + 3 * (size_t)q;
    return (dist3_strict(pp[0], pp[1], pp[2], px, py, pz) <= a.pos_tol) ? q : -1;
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_temporal(TemporalArgs a)
{
#pragma clang fp contract(off)
    const int n = a.W * a.H;
    const int p = blockIdx.x * BLOCK + threadIdx.x;
    if (p >= n) return;

    float nx, ny, nz, px, py, pz;
    int gid;
    if (a.gbuf) {             // the boundary's AoS texel (52 B): read once, split into the planes every later kernel reads
        const float *t = a.gbuf + 13 * (size_t)p;
        nx = t[0]; ny = t[1]; nz = t[2];
        px = t[3]; py = t[4]; pz = t[5];
        gid = __float_as_int(t[12]);
        if (!a.skip_split) {
            a.nrm_cur[3 * (size_t)p] = nx; a.nrm_cur[3 * (size_t)p + 1] = ny; a.nrm_cur[3 * (size_t)p + 2] = nz;
            a.pos_cur[3 * (size_t)p] = px; a.pos_cur[3 * (size_t)p + 1] = py; a.pos_cur[3 * (size_t)p + 2] = pz;
            a.gid_cur[p] = gid;
        }
    } else {                  // planar path: the producer wrote the planes in place (svgf_planar_gbuffer), 28 B read, nothing split
        nx = a.nrm_cur[3 * (size_t)p]; ny = a.nrm_cur[3 * (size_t)p + 1]; nz = a.nrm_cur[3 * (size_t)p + 2];
        px = a.pos_cur[3 * (size_t)p]; py = a.pos_cur[3 * (size_t)p + 1]; pz = a.pos_cur[3 * (size_t)p + 2];
        gid = a.gid_cur[p];
    }

    const float cr = a.in_rgb[3 * (size_t)p], cg = a.in_rgb[3 * (size_t)p + 1], cb = a.in_rgb[3 * (size_t)p + 2];
    const float lum = lum_strict(cr, cg, cb);
    const int N = a.hlen[p];

    bool valid = false;
    SvgfHistSum hs = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
    if (N > 0 && gid != -1) {
        const SvgfReproj rp = svgf_reproject(a, px, py, pz);          // previous-frame pixel coordinate (:198-209)
        const float fx = rp.fx, fy = rp.fy;

        valid = (fx >= 0.0f && fy >= 0.0f && fx < (float)a.W && fy < (float)a.H);
        int q4[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            q4[k] = reproj_valid_pos(a, reproj_valid(a, fx + (float)(k & 1), fy + (float)(k >> 1), gid, nx, ny, nz), px, py, pz);
            valid = valid && (q4[k] >= 0);
        }

        if (valid) {                                                  // bilinear (:234-259)
            float w[4];
            svgf_bilinear_weights(rp.fracx, rp.fracy, w);
            float sumw = 0.0f;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int q = q4[k];
                const float4 ch = a.cv_hist[q];
                const float2 mh = a.mom_hist[q];
                svgf_hist_add_weighted(hs, w[k], ch.x, ch.y, ch.z, mh.x, mh.y, a.hlen[q]);
                sumw += w[k];
            }
            if ((double)sumw >= 0.01) svgf_hist_div(hs, sumw);
        } else {                                                      // 3x3 box around floor (:262-286)
            float cnt = 0.0f;
            for (int yy = -1; yy <= 1; yy++)
                for (int xx = -1; xx <= 1; xx++) {
                    // the four taps with xx, yy in {0, 1} are the bilinear taps tested above: same arguments, same answer
                    const int q = (xx >= 0 && yy >= 0) ? q4[xx + 2 * yy]
                                                       : reproj_valid_pos(a, reproj_valid(a, fx + (float)xx, fy + (float)yy, gid, nx, ny, nz), px, py, pz);
                    if (q >= 0) {
                        const float4 ch = a.cv_hist[q];
                        const float2 mh = a.mom_hist[q];
                        svgf_hist_add(hs, ch.x, ch.y, ch.z, mh.x, mh.y, a.hlen[q]);
                        cnt += 1.0f;
                    }
                }
            if (cnt > 0.0f) {
                svgf_hist_div(hs, cnt);
                valid = true;
            }
        }
    }
    const SvgfTemporalOut o = svgf_temporal_blend(a, cr, cg, cb, lum, N, valid, hs);
    a.hlen_upd[p] = o.hlen;
    a.mom_acc[p] = o.mom;
    a.cv_acc[p] = o.cv;
}

hipError_t launch_temporal(const TemporalArgs &a, hipStream_t s)
{
    const long long n = (long long)a.W * a.H;
    SVGF_LAUNCH_KERNEL(k_temporal<SVGF_BLOCK>, dim3(div_up(n, SVGF_BLOCK)), dim3(SVGF_BLOCK), 0, s, a);
    return hipGetLastError();
}

// ----------------------------------------------------------------------------------------------------
// SvgfParams::spatial_variance_frames (f4 extension; the reference's EstimateVariance is a stub, src/denoise.cu:320-329):
// pixels whose updated history is shorter than K frames take their variance from the luminance moments of their 7x7
// neighbourhood (taps with the same geomId and |n_q - n_p| <= 0.1, the reference's own consistency predicate; sums in
// raster order), boosted by max(1, 4 / history length) (Schied et al. 2017, section 4.2).  Writes cv_acc.w only.
// ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(SVGF_BLOCK) void k_spatial_variance(float4 *__restrict__ cv_acc, const float2 *__restrict__ mom_acc,
                                                                const int *__restrict__ hlen_upd, const float *__restrict__ nrm,
                                                                const int *__restrict__ gid, int W, int H, int K)
{
#pragma clang fp contract(off)
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * (SVGF_BLOCK / 64) + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const int p = x + y * W;
    const int hl = hlen_upd[p];
    if (hl >= K) return;
    const int g0 = gid[p];
    const float nx = nrm[3 * (size_t)p], ny = nrm[3 * (size_t)p + 1], nz = nrm[3 * (size_t)p + 2];
    float s1 = 0.0f, s2 = 0.0f, cnt = 0.0f;
    for (int yy = -3; yy <= 3; yy++)
        for (int xx = -3; xx <= 3; xx++) {
            const int qx = x + xx, qy = y + yy;
            if (qx < 0 || qx >= W || qy < 0 || qy >= H) continue;
            const int q = qx + qy * W;
            if (q != p) {
                if (gid[q] != g0) continue;
                const float *n = nrm + 3 * (size_t)q;
                if (!(dist3_strict(n[0], n[1], n[2], nx, ny, nz) <= 1e-1f)) continue;
            }
            const float2 m = mom_acc[q];
            s1 += m.x; s2 += m.y; cnt += 1.0f;
        }
    const float m1 = s1 / cnt, m2 = s2 / cnt;
    float v = m2 - m1 * m1;
    v = v > 0.0f ? v : 0.0f;
    const float boost = 4.0f / (float)(hl > 0 ? hl : 1);
    cv_acc[p].w = v * (boost > 1.0f ? boost : 1.0f);
}

hipError_t launch_spatial_variance(float4 *cv_acc, const float2 *mom_acc, const int *hlen_upd, const float *nrm, const int *gid,
                                   int W, int H, int K, hipStream_t s)
{
    SVGF_LAUNCH_KERNEL(k_spatial_variance, dim3((W + 63) / 64, (H + SVGF_BLOCK / 64 - 1) / (SVGF_BLOCK / 64)), dim3(SVGF_BLOCK), 0, s,
                       cv_acc, mom_acc, hlen_upd, nrm, gid, W, H, K);
    return hipGetLastError();
}

// ----------------------------------------------------------------------------------------------------
// non-temporal prepare: variance = 10 (reference EstimateVariance :320-329), colour = input (:370), split texel
// ----------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(SVGF_BLOCK) void k_prepare(const float *__restrict__ in_rgb, const float *__restrict__ gbuf,
                                                       float4 *__restrict__ cv, float *__restrict__ nrm,
                                                       int *__restrict__ gid, float *__restrict__ pos, int n)
{
    const int p = blockIdx.x * SVGF_BLOCK + threadIdx.x;
    if (p >= n) return;
    if (gbuf) {               // null on the planar path: the producer filled the planes itself
        const float *t = gbuf + 13 * (size_t)p;
        nrm[3 * (size_t)p] = t[0]; nrm[3 * (size_t)p + 1] = t[1]; nrm[3 * (size_t)p + 2] = t[2];
        pos[3 * (size_t)p] = t[3]; pos[3 * (size_t)p + 1] = t[4]; pos[3 * (size_t)p + 2] = t[5];
        gid[p] = __float_as_int(t[12]);
    }
    cv[p] = make_float4(in_rgb[3 * (size_t)p], in_rgb[3 * (size_t)p + 1], in_rgb[3 * (size_t)p + 2], 10.0f);
}

hipError_t launch_prepare(const float *in_rgb, const float *gbuf, float4 *cv, float *nrm, int *gid, float *pos,
                          int W, int H, hipStream_t s)
{
    const long long n = (long long)W * H;
    SVGF_LAUNCH_KERNEL(k_prepare, dim3(div_up(n, SVGF_BLOCK)), dim3(SVGF_BLOCK), 0, s, in_rgb, gbuf, cv, nrm, gid, pos, (int)n);
    return hipGetLastError();
}

// ----------------------------------------------------------------------------------------------------
// strict a-trous gather  (reference ATrousFilter src/denoise.cu:77-170), snapshot variance semantics:
// variance is read from src.w and written to dst.w, never in place.
// ----------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(SVGF_BLOCK) void k_atrous_gather(AtrousArgs a)
{
#pragma clang fp contract(off)
    const int n = a.W * a.H;
    const int p = blockIdx.x * SVGF_BLOCK + threadIdx.x;
    if (p >= n) return;
    const int x = p % a.W, y = p / a.W;

    const float4 cp = a.src[p];
    float var;
    if (a.blur_variance) {                                            // 3x3 gaussian, borders renormalised (:102-118)
        float sum = 0.0f, sumw = 0.0f;
        for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++) {
                const int lx = x + dx, ly = y + dy;
                if (lx >= 0 && ly >= 0 && lx < a.W && ly < a.H) {
                    const float gw = (float)((2 - (dx & 1)) * (2 - (dy & 1))) * 0.0625f;   // [1 2 1]x[1 2 1]/16
                    sum += gw * a.src[lx + ly * a.W].w;
                    sumw += gw;
                }
            }
        var = fmaxf(sum / sumw, 0.0f);
    } else {
        var = fmaxf(cp.w, 0.0f);
    }

    const float lp = lum_strict(cp.x, cp.y, cp.z);
    const float npx = a.nrm[3 * (size_t)p], npy = a.nrm[3 * (size_t)p + 1], npz = a.nrm[3 * (size_t)p + 2];
    const float ppx = a.pos[3 * (size_t)p], ppy = a.pos[3 * (size_t)p + 1], ppz = a.pos[3 * (size_t)p + 2];

    const double den_l = (double)(sqrtf(var) * a.sigma_c) + 1e-6;
    const double den_n = (double)a.sigma_n + 1e-6;
    const double den_x = (double)a.sigma_x + 1e-6;

    float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f, vsum = 0.0f, wsum = 0.0f, w2sum = 0.0f;
    for (int i = -2; i <= 2; i++) {
        for (int j = -2; j <= 2; j++) {
            const int xq = x + a.step * i, yq = y + a.step * j;
            if (xq < 0 || xq >= a.W || yq < 0 || yq >= a.H) continue;
            const int q = xq + yq * a.W;
            const float4 cq = a.src[q];
            const float lq = lum_strict(cq.x, cq.y, cq.z);
            const float *nq = a.nrm + 3 * (size_t)q;
            const float *pq = a.pos + 3 * (size_t)q;
            const float wl = expf((float)(-(double)fabsf(lq - lp) / den_l));
            const float wn = fminf(1.0f, expf((float)(-(double)dist3_strict(npx, npy, npz, nq[0], nq[1], nq[2]) / den_n)));
            const float wx = fminf(1.0f, expf((float)(-(double)dist3_strict(ppx, ppy, ppz, pq[0], pq[1], pq[2]) / den_x)));
            const float hi = (i == 0) ? 6.0f : ((i == 1 || i == -1) ? 4.0f : 1.0f);
            const float hj = (j == 0) ? 6.0f : ((j == 1 || j == -1) ? 4.0f : 1.0f);
            float w = (hi * hj * (1.0f / 256.0f)) * wl;                 // h[k] exact in fp32
            w = w * wn;
            w = w * wx;
            wsum += w;
            w2sum += w * w;
            c0 += cq.x * w; c1 += cq.y * w; c2 += cq.z * w;
            vsum += (cq.w * w) * w;
        }
    }

    float o0, o1, o2, ov;
    if ((double)wsum > 10e-6) {                                       // NaN -> false -> pass-through (:159-164)
        o0 = c0 / wsum; o1 = c1 / wsum; o2 = c2 / wsum; ov = vsum / w2sum;
    } else {
        o0 = cp.x; o1 = cp.y; o2 = cp.z; ov = cp.w;
    }
    if (a.modulate) svgf_modulate(a, (unsigned)p, o0, o1, o2);        // last level: * albedo * ialbedo (:166-168)
    if (a.dst) a.dst[p] = make_float4(o0, o1, o2, ov);
    if (a.out_rgb) { a.out_rgb[3 * (size_t)p] = o0; a.out_rgb[3 * (size_t)p + 1] = o1; a.out_rgb[3 * (size_t)p + 2] = o2; }
}

hipError_t launch_atrous_gather(const AtrousArgs &a, hipStream_t s)
{
    const long long n = (long long)a.W * a.H;
    SVGF_LAUNCH_KERNEL(k_atrous_gather, dim3(div_up(n, SVGF_BLOCK)), dim3(SVGF_BLOCK), 0, s, a);
    return hipGetLastError();
}

// ----------------------------------------------------------------------------------------------------
// debug views (reference DebugView :331-340) and pass-through copy (:382)
// ----------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(SVGF_BLOCK) void k_debug_hlen(const int *__restrict__ hlen, float *__restrict__ out, int n, float scale)
{
    const int p = blockIdx.x * SVGF_BLOCK + threadIdx.x;
    if (p >= n) return;
    const float v = (float)hlen[p] / scale;
    out[3 * (size_t)p] = v; out[3 * (size_t)p + 1] = v; out[3 * (size_t)p + 2] = v;
}
__global__ __launch_bounds__(SVGF_BLOCK) void k_debug_var(const float4 *__restrict__ cv, float *__restrict__ out, int n, float scale)
{
    const int p = blockIdx.x * SVGF_BLOCK + threadIdx.x;
    if (p >= n) return;
    const float v = cv[p].w / scale;
    out[3 * (size_t)p] = v; out[3 * (size_t)p + 1] = v; out[3 * (size_t)p + 2] = v;
}
__global__ __launch_bounds__(SVGF_BLOCK) void k_copy_rgb(const float4 *__restrict__ cv, float *__restrict__ out, int n)
{
    const int p = blockIdx.x * SVGF_BLOCK + threadIdx.x;
    if (p >= n) return;
    const float4 c = cv[p];
    out[3 * (size_t)p] = c.x; out[3 * (size_t)p + 1] = c.y; out[3 * (size_t)p + 2] = c.z;
}

hipError_t launch_debug_hlen(const int *hlen, float *out_rgb, int n, float scale, hipStream_t s)
{
    SVGF_LAUNCH_KERNEL(k_debug_hlen, dim3(div_up(n, SVGF_BLOCK)), dim3(SVGF_BLOCK), 0, s, hlen, out_rgb, n, scale);
    return hipGetLastError();
}
hipError_t launch_debug_var(const float4 *cv, float *out_rgb, int n, float scale, hipStream_t s)
{
    SVGF_LAUNCH_KERNEL(k_debug_var, dim3(div_up(n, SVGF_BLOCK)), dim3(SVGF_BLOCK), 0, s, cv, out_rgb, n, scale);
    return hipGetLastError();
}
hipError_t launch_copy_rgb(const float4 *cv, float *out_rgb, int n, hipStream_t s)
{
    SVGF_LAUNCH_KERNEL(k_copy_rgb, dim3(div_up(n, SVGF_BLOCK)), dim3(SVGF_BLOCK), 0, s, cv, out_rgb, n);
    return hipGetLastError();
}
