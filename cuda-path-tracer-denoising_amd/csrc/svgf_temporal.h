// svgf_temporal.h — the per-pixel arithmetic of the temporal pass (reference BackProjection src/denoise.cu:185-317 and
// isReprjValid :172-182), factored into device functions so that it lives in ONE place: k_temporal (svgf_kernels.hip, one
// thread per pixel) and the loader waves of the fused temporal + first-level kernel (svgf_atrous_fused.hip) both call them.
//
// Everything here keeps the reference's operation order with FMA contraction off (the temporal goldens are bit-exact):
// glm mat4 * vec4 association (m0 v0 + m1 v1) + (m2 v2 + m3 v3), luminance in double, alpha on the current side for colour
// (:297) and on the history side for the moments (:300-301), (int) truncation of the interpolated history length (:294).
#pragma once
#include "svgf_kernels.h"

// luminance with the reference's double promotion (src/denoise.cu:121,138,196)
__device__ __forceinline__ float svgf_lum_strict(float r, float g, float b)
{
#pragma clang fp contract(off)
    double l = 0.2126 * (double)r + 0.7152 * (double)g;
    l = l + 0.0722 * (double)b;
    return (float)l;
}

// glm::distance(vec3,vec3): sqrt((dx*dx + dy*dy) + dz*dz)
__device__ __forceinline__ float svgf_dist3_strict(float ax, float ay, float az, float bx, float by, float bz)
{
#pragma clang fp contract(off)
    float dx = bx - ax, dy = by - ay, dz = bz - az;
    float s = dx * dx + dy * dy;
    s = s + dz * dz;
    return sqrtf(s);
}

// previous-frame pixel coordinate of world position (px,py,pz) (:198-209): floor and fraction of the reprojected position
struct SvgfReproj { float fx, fy, fracx, fracy; };
__device__ __forceinline__ SvgfReproj svgf_reproject(const TemporalArgs &a, float px, float py, float pz)
{
#pragma clang fp contract(off)
    float vs[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        float a0 = a.M[0 * 4 + r] * px + a.M[1 * 4 + r] * py;
        float a1 = a.M[2 * 4 + r] * pz + a.M[3 * 4 + r] * 1.0f;
        vs[r] = a0 + a1;
    }
    float clipx = vs[0] / vs[2], clipy = vs[1] / vs[2];          // no tan(fov), no aspect (:202-203)
    if (a.reproj_sx > 0.0f) clipx = clipx / a.reproj_sx;          // f4 extension: exact for any fov / aspect
    if (a.reproj_sy > 0.0f) clipy = clipy / a.reproj_sy;
    float ndcx = -clipx * 0.5f + 0.5f, ndcy = -clipy * 0.5f + 0.5f;
    float prevx = ndcx * (float)a.W - 0.5f, prevy = ndcy * (float)a.H - 0.5f;
    SvgfReproj r;
    r.fx = floorf(prevx); r.fy = floorf(prevy);
    r.fracx = prevx - r.fx; r.fracy = prevy - r.fy;
    return r;
}

// bounds part of isReprjValid (:173-176): texel index of the tap at float coordinate (qx, qy), -1 when it is outside the
// screen (a NaN coordinate is DEFINED as outside; the reference would index texel (int)NaN)
__device__ __forceinline__ int svgf_tap_index(const TemporalArgs &a, float qx, float qy)
{
    if (!(qx == qx) || !(qy == qy)) return -1;
    if (qx < 0.0f || qx >= (float)a.W || qy < 0.0f || qy >= (float)a.H) return -1;
    return (int)qx + (int)qy * a.W;
}

// consistency part of isReprjValid (:177-180) on the tap's previous-frame geomId / normal.  `distance(n_prev, n_cur) > 1e-1f`
// is evaluated on the SQUARED distance: sqrtf is correctly rounded and monotonic, so sqrtf(s) > 0.1f  <=>  s > 0x3c23d70b (the
// largest float whose root still rounds to <= 0.1f; found by stepping through the floats around 0.01, tools/experiments/
// temporal_branch_free_validity_32bit_offsets.patch carried the same constant in round 1).  NaN compares false either way, i.e.
// a NaN distance PASSES, as in the reference.  Branch-free: the fused kernel evaluates it for nine taps per pixel.
__device__ __forceinline__ bool svgf_normals_close(float nqx, float nqy, float nqz, float nx, float ny, float nz)
{
#pragma clang fp contract(off)
    const float dx = nx - nqx, dy = ny - nqy, dz = nz - nqz;          // glm::distance(a, b) = length(b - a): b is the current normal
    float s = dx * dx + dy * dy;
    s = s + dz * dz;
    return !(s > __uint_as_float(0x3c23d70bu));
}
__device__ __forceinline__ bool svgf_tap_consistent(int gq, float nqx, float nqy, float nqz, int gid, float nx, float ny, float nz)
{
    return (gq != -1) & (gq == gid) & svgf_normals_close(nqx, nqy, nqz, nx, ny, nz);
}

// bilinear weights of the four taps (0,0) (1,0) (0,1) (1,1) (:237-240)
__device__ __forceinline__ void svgf_bilinear_weights(float fracx, float fracy, float (&w)[4])
{
#pragma clang fp contract(off)
    w[0] = (1 - fracx) * (1 - fracy); w[1] = fracx * (1 - fracy); w[2] = (1 - fracx) * fracy; w[3] = fracx * fracy;
}

// history value being gathered: colour, moments, (float) length
struct SvgfHistSum { float pc0, pc1, pc2, pm0, pm1, plen; };
// bilinear tap (:242-249): sum += w * tap
__device__ __forceinline__ void svgf_hist_add_weighted(SvgfHistSum &h, float w, float c0, float c1, float c2, float m0, float m1, int len)
{
#pragma clang fp contract(off)
    h.pc0 += w * c0; h.pc1 += w * c1; h.pc2 += w * c2;
    h.pm0 += w * m0; h.pm1 += w * m1;
    h.plen += w * (float)len;
}
// 3x3 fallback tap (:272-279): sum += tap
__device__ __forceinline__ void svgf_hist_add(SvgfHistSum &h, float c0, float c1, float c2, float m0, float m1, int len)
{
#pragma clang fp contract(off)
    h.pc0 += c0; h.pc1 += c1; h.pc2 += c2;
    h.pm0 += m0; h.pm1 += m1;
    h.plen += (float)len;
}
__device__ __forceinline__ void svgf_hist_div(SvgfHistSum &h, float d)
{
#pragma clang fp contract(off)
    h.pc0 /= d; h.pc1 /= d; h.pc2 /= d; h.pm0 /= d; h.pm1 /= d; h.plen /= d;
}

// The accumulated pixel.  `valid`: a usable history value (pc*, pm*, plen: interpolated colour, moments, length) was found.
struct SvgfTemporalOut { float4 cv; float2 mom; int hlen; };
__device__ __forceinline__ SvgfTemporalOut svgf_temporal_blend(const TemporalArgs &a, float cr, float cg, float cb, float lum, int N,
                                                              bool valid, const SvgfHistSum &hs)
{
#pragma clang fp contract(off)
    const float pc0 = hs.pc0, pc1 = hs.pc1, pc2 = hs.pc2, pm0 = hs.pm0, pm1 = hs.pm1, plen = hs.plen;
    SvgfTemporalOut o;
    if (valid) {
        const float ca = fmaxf(1.0f / (float)(N + 1), a.color_alpha_min);   // alpha on the current side (:297)
        const float ma = fmaxf(1.0f / (float)(N + 1), a.moment_alpha_min);  // alpha on the history side (:300-301)
        o.hlen = (int)plen + 1;
        const float m1 = ma * pm0 + (1.0f - ma) * lum;
        const float m2 = ma * pm1 + ((1.0f - ma) * lum) * lum;
        o.mom = make_float2(m1, m2);
        const float v = m2 - m1 * m1;
        o.cv = make_float4(cr * ca + pc0 * (1.0f - ca), cg * ca + pc1 * (1.0f - ca), cb * ca + pc2 * (1.0f - ca), v > 0.0f ? v : 0.0f);
    } else {                                                               // no usable history (:311-315)
        o.hlen = 1;
        o.mom = make_float2(lum, lum * lum);
        o.cv = make_float4(cr, cg, cb, 100.0f);
    }
    return o;
}
