// svgf_atrous_lattice.hip — a-trous level for large dilations (S = 64, 128, ...; levels 6-7 of the reference's 0..7
// slider, src/preview.cpp:327) as dense 5x5 stencils over LATTICE SUB-IMAGES, for gfx950.
//
// Same result as the other a-trous kernels: one level of reference ATrousFilter (src/denoise.cu:77-170), snapshot
// variance.  What changes is the decomposition.  With dilation S the image splits into S*S independent sub-images
// {(xph + S*m, yph + S*r)}: a tap of pixel (m, r) of a sub-image is pixel (m + i, r + j) of the SAME sub-image.  The
// strip kernels keep x contiguous and pay 4*S halo columns per strip, which stops fitting LDS beyond S = 32
// (SURVEY.md §7 hard parts).  For S >= 64 a sub-image is small (1080p: 30 x 17 pixels at S = 64, 15 x 9 at S = 128;
// 4K: 60 x 34), so a workgroup takes K adjacent x-phases of one y-phase — K whole sub-images, or a band of their rows
// when they are tall — into LDS with a 2-pixel border, and every output pixel reads its 25 taps from there: each input
// pixel is fetched once per band instead of 25 times (the strict gather kernel, which was the fallback for these steps:
// 205-215 us per 1080p level), and the border is the image border, so there is no halo in x at all.
//
// K (2 .. 8) is what keeps the global accesses from being fully scattered: the K phases are K contiguous pixels, so a
// group of K lanes reads K*16 contiguous bytes of the colour plane (K*12 of the normal / position planes), writes its
// outputs the same way, and finds most of the 3x3 variance pre-blur neighbours in lines it touches anyway.
// LDS layout (48-byte records as in svgf_atrous_strip.hip: A = {n.x,p.x,n.y,p.y}, B = {n.z,p.z,lum,-}, C = {r,g,b,var})
// is [row][phase][lattice column], so a tap is +-1 record in x and +-K*tw records in y, and consecutive lanes of the
// compute loop (which run over the lattice columns of one phase) read consecutive records: conflict-free b128 reads.
// Out-of-image encoding (luminance = +inf => weight 0), NaN handling (`careful`) and the tap arithmetic are the strip
// kernel's.
#include "svgf_kernels.h"

#include <type_traits>

namespace {

constexpr float kLog2e = 1.44269504088896340736f;
constexpr int PXB = 48;                   // bytes per staged pixel
#ifndef SVGF_LATTICE_NT
#define SVGF_LATTICE_NT 1024
#define SVGF_LATTICE_LDS_KB 150
#endif
constexpr int NT = SVGF_LATTICE_NT;       // 1024 threads / 150 KB: one workgroup per CU, 16 waves
constexpr int kLdsBudget = SVGF_LATTICE_LDS_KB * 1024;

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

struct LatticeGeom {
    int log2s, log2k;
    int tw;          // staged lattice columns per phase row: ceil(W / S) + 4
    int pstride;     // records per phase row in LDS (>= tw, padded: see geometry())
    int band_rows;   // output lattice rows per workgroup
    int n_bands;     // bands per sub-image
    float kn, kx;    // log2(e) / (sigma_n + 1e-6), log2(e) / (sigma_x + 1e-6)
};

__device__ __forceinline__ float lum_f64(float r, float g, float b)
{   // reference luminance: double products, rounded once to float (src/denoise.cu:121,138)
    double l = 0.2126 * (double)r + 0.7152 * (double)g;
    l = l + 0.0722 * (double)b;
    return (float)l;
}

__device__ __forceinline__ constexpr float neg_log2_binom(int i)
{   // -log2 of the 5-tap binomial [1 4 6 4 1]/16
    return (i == 0) ? 1.4150374992788437f : ((i == 1 || i == -1) ? 2.0f : 4.0f);
}

struct Acc { v2f rg, bv, ww; };     // (sum w r, sum w g), (sum w b, sum w^2 var), (sum w, sum w^2)

template <bool HASVAR>
__global__ __launch_bounds__(NT) void k_atrous_lattice(AtrousArgs a, LatticeGeom gm)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int nan_seen;

    const int S = 1 << gm.log2s, K = 1 << gm.log2k;
    const int W = a.W, H = a.H;
    // work item: (group of K x-phases, y-phase, band); the x-groups of one (y-phase, band) are consecutive workgroups
    int bid = blockIdx.x;
    const int xg = bid & ((S >> gm.log2k) - 1); bid >>= (gm.log2s - gm.log2k);
    const int yph = bid & (S - 1); bid >>= gm.log2s;
    const int band = bid;
    const int xph0 = xg << gm.log2k;
    if (xph0 >= W || yph >= H) return;
    const int mw = (W - xph0 + S - 1) >> gm.log2s;          // lattice columns of the group's first (widest) phase
    const int mh = (H - yph + S - 1) >> gm.log2s;           // lattice rows
    const int r0 = band * gm.band_rows;
    if (r0 >= mh) return;
    const int bh = min(gm.band_rows, mh - r0);
    const int tw = gm.tw, th = bh + 4, P = gm.pstride;
    const int tid = threadIdx.x;
    if (tid == 0) nan_seen = 0;
    __syncthreads();

    // ---- stage rows r0-2 .. r0+bh+1, lattice columns -2 .. tw-3 of the K phases (branch-free: clamped coordinates).
    //      Lane order: phase fastest, so K lanes read K contiguous pixels. ----
    const float inf = __builtin_huge_valf();
    for (int idx = tid; idx < (tw * th) << gm.log2k; idx += NT) {
        const int k = idx & (K - 1), rest = idx >> gm.log2k;
        const int ty = rest / tw, tx = rest - ty * tw;
        const int m = tx - 2, r = r0 + ty - 2;
        const int xs = xph0 + k + (m << gm.log2s), ys = yph + (r << gm.log2s);
        const bool ok = (m >= 0) && (r >= 0) && (xs < W) && (ys < H);
        const unsigned q = (unsigned)min(max(ys, 0), H - 1) * (unsigned)W + (unsigned)min(max(xs, 0), W - 1);
        const float4 cv = a.src[q];
        const float *n = a.nrm + 3u * (size_t)q;
        const float *p = a.pos + 3u * (size_t)q;
        const float nx = n[0], ny = n[1], nz = n[2], px = p[0], py = p[1], pz = p[2];
        const float mag = fabsf(nx) + fabsf(ny) + fabsf(nz) + fabsf(px) + fabsf(py) + fabsf(pz);
        if (!(mag < inf)) nan_seen = 1;                     // NaN or inf in a normal / position (rare)
        char *d = smem + (size_t)(((ty << gm.log2k) + k) * P + tx) * PXB;
        *reinterpret_cast<float4 *>(d) = make_float4(nx, px, ny, py);
        *reinterpret_cast<float4 *>(d + 16) = make_float4(nz, pz, ok ? lum_f64(cv.x, cv.y, cv.z) : inf, 0.0f);
        *reinterpret_cast<float4 *>(d + 32) = ok ? cv : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const bool careful = (nan_seen != 0);
    const float kn = gm.kn, kx = gm.kx;
    const int rowb = (P << gm.log2k) * PXB;                  // bytes between lattice rows in LDS

    // ---- outputs: one pixel per thread and pass, same lane order ----
    for (int o = tid; o < (mw * bh) << gm.log2k; o += NT) {
        const int k = o & (K - 1), rest = o >> gm.log2k;
        const int oy = rest / mw, ox = rest - oy * mw;
        const int x = xph0 + k + (ox << gm.log2s), y = yph + ((r0 + oy) << gm.log2s);
        if (x >= W) continue;                                // the narrower phases of the group
        const unsigned p = (unsigned)y * (unsigned)W + (unsigned)x;
        const char *win = smem + (size_t)(((oy << gm.log2k) + k) * P + ox) * PXB;      // record of tap (-2, -2)
        const char *rowc = win + 2 * rowb + 2 * PXB;                                    // the centre
        const v4f A = *reinterpret_cast<const v4f *>(rowc);
        const v4f B = *reinterpret_cast<const v4f *>(rowc + 16);
        const v4f C = *reinterpret_cast<const v4f *>(rowc + 32);

        // centre variance: 3x3 gaussian over the FULL-resolution neighbours, out-of-image taps dropped and renormalised
        // (:102-118); read from global (the neighbours belong to other sub-images, mostly of this workgroup's lines)
        float var = C.w;
        if (a.blur_variance) {
            float sum = 0.0f, sumw = 0.0f;
#pragma unroll
            for (int dy = -1; dy <= 1; dy++)
#pragma unroll
                for (int dx = -1; dx <= 1; dx++) {
                    const int lx = x + dx, ly = y + dy;
                    const bool in = lx >= 0 && ly >= 0 && lx < W && ly < H;
                    const float gw = in ? (float)((2 - (dx & 1)) * (2 - (dy & 1))) * 0.0625f : 0.0f;   // [1 2 1]x[1 2 1]/16
                    const float v = (dx == 0 && dy == 0) ? C.w : a.src[(unsigned)min(max(ly, 0), H - 1) * (unsigned)W + (unsigned)min(max(lx, 0), W - 1)].w;
                    sum = fmaf(gw, in ? v : 0.0f, sum);
                    sumw += gw;
                }
            var = sum * __builtin_amdgcn_rcpf(sumw);
        }
        var = fmaxf(var, 0.0f);
        const float lp = B.z;
        const float kl = kLog2e * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(var) * a.sigma_c + 1e-6f);
        const v2f c0 = v2f{A.x, A.y}, c1 = v2f{A.z, A.w}, c2 = v2f{B.x, B.y};

        Acc acc;
        auto taps = [&](auto careful_tag) {
            constexpr bool CAREFUL = decltype(careful_tag)::value;
            if (!CAREFUL) {     // the centre tap has weight exactly h = 9/64
                constexpr float w0 = 0.140625f;
                acc.ww = v2f{w0, w0 * w0};
                acc.rg = v2f{w0 * C.x, w0 * C.y};
                acc.bv = v2f{w0 * C.z, (w0 * w0) * C.w};
            } else {
                acc.rg = v2f{0.0f, 0.0f}; acc.bv = v2f{0.0f, 0.0f}; acc.ww = v2f{0.0f, 0.0f};
            }
#pragma unroll
            for (int j = -2; j <= 2; j++) {
                const char *rowp = win + (j + 2) * rowb;
                // the five taps of a row in stages (as in the strip kernel): their dependency chains interleave
                v4f Aq[5], Bq[5], Cq[5];
#pragma unroll
                for (int i = 0; i < 5; i++) {
                    Aq[i] = *reinterpret_cast<const v4f *>(rowp + i * PXB);
                    Bq[i] = *reinterpret_cast<const v4f *>(rowp + i * PXB + 16);
                    Cq[i] = *reinterpret_cast<const v4f *>(rowp + i * PXB + 32);
                }
                float e[5];
#pragma unroll
                for (int i = 0; i < 5; i++) {
                    if (i == 2 && j == 0 && !CAREFUL) continue;
                    const v2f d0 = Aq[i].xy - c0, d1 = Aq[i].zw - c1, d2 = Bq[i].xy - c2;
                    v2f s2 = d0 * d0;
                    s2 = __builtin_elementwise_fma(d1, d1, s2);
                    s2 = __builtin_elementwise_fma(d2, d2, s2);                   // (|dn|^2, |dp|^2)
                    float dn = __builtin_amdgcn_sqrtf(s2.x), dx = __builtin_amdgcn_sqrtf(s2.y);
                    if (CAREFUL) {      // min(1, exp(-NaN)) == 1 in the reference: a NaN distance contributes nothing
                        dn = fmaxf(dn, 0.0f);
                        dx = fmaxf(dx, 0.0f);
                    }
                    float t = fmaf(fabsf(Bq[i].z - lp), kl, neg_log2_binom(i - 2) + neg_log2_binom(j));
                    t = fmaf(dn, kn, t);
                    e[i] = fmaf(dx, kx, t);
                }
#pragma unroll
                for (int i = 0; i < 5; i++) {
                    if (i == 2 && j == 0 && !CAREFUL) continue;
                    const float w = __builtin_amdgcn_exp2f(-e[i]);
                    if (HASVAR) {
                        v2f wv;
                        wv.x = w;
                        wv.y = w * w;
                        acc.ww += wv;
                        acc.rg = __builtin_elementwise_fma(Cq[i].xy, v2f{w, w}, acc.rg);
                        acc.bv = __builtin_elementwise_fma(Cq[i].zw, wv, acc.bv);
                    } else {
                        acc.ww.x += w;
                        acc.rg = __builtin_elementwise_fma(Cq[i].xy, v2f{w, w}, acc.rg);
                        acc.bv.x = fmaf(Cq[i].z, w, acc.bv.x);
                    }
                }
            }
        };
        if (careful) taps(std::true_type{}); else taps(std::false_type{});

        float o0, o1, o2, ov;
        if (acc.ww.x > 1e-5f) {                                     // NaN -> false -> pass-through (:159-164)
            const float rw = __builtin_amdgcn_rcpf(acc.ww.x);
            o0 = acc.rg.x * rw; o1 = acc.rg.y * rw; o2 = acc.bv.x * rw;
            ov = HASVAR ? acc.bv.y * __builtin_amdgcn_rcpf(acc.ww.y) : 0.0f;
        } else {
            o0 = C.x; o1 = C.y; o2 = C.z; ov = C.w;
        }
        if (a.modulate) svgf_modulate(a, (unsigned)p, o0, o1, o2);  // last level: * albedo * ialbedo (:166-168)
        if (a.dst) a.dst[p] = make_float4(o0, o1, o2, ov);
        if (a.out_rgb) { float *q = a.out_rgb + 3u * (size_t)p; q[0] = o0; q[1] = o1; q[2] = o2; }
    }
}

// Tile geometry.  K: the largest of 8, 4, 2, 1 phases per workgroup whose tile holds whole sub-images or at least
// 8-row bands within the LDS budget.  pstride: the row of one phase is padded so that the 16 lanes a b128 LDS access
// serves per cycle (K phases x 16/K consecutive columns, 12 dwords apart) fall on 16 distinct 4-bank groups:
// 12 * pstride mod 64 must be 32 (K = 2), 48 (K = 4) or 24 (K = 8).
bool geometry(const AtrousArgs &a, LatticeGeom &gm)
{
    if (a.step < 64 || (a.step & (a.step - 1))) return false;
    if ((long long)a.W * a.H * 16 >= (1LL << 32)) return false;     // 32-bit element offsets in the kernel
    int log2s = 0;
    while ((1 << log2s) < a.step) log2s++;
    if (log2s > 12) return false;
    const int S = a.step;
    gm.log2s = log2s;
    gm.tw = (a.W + S - 1) / S + 4;
    const int mh_max = (a.H + S - 1) / S;
    int chosen = -1, rows_fit = 0;
    for (int log2k = 3; log2k >= 0 && chosen < 0; log2k--) {
        const int K = 1 << log2k;
        const int want = (K == 1) ? -1 : (K == 2 ? 32 : (K == 4 ? 48 : 24));
        int P = gm.tw;
        while (want >= 0 && (12 * P) % 64 != want) P++;
        const int fit = kLdsBudget / (K * P * PXB) - 4;
        if (fit >= (mh_max < 8 ? mh_max : 8) || (K == 1 && fit >= 1)) { chosen = log2k; rows_fit = fit; gm.pstride = P; }
    }
    if (chosen < 0) return false;                                   // a single lattice row does not fit: gather kernel
    gm.log2k = chosen;
    if (rows_fit > mh_max) rows_fit = mh_max;
    gm.n_bands = (mh_max + rows_fit - 1) / rows_fit;
    gm.band_rows = (mh_max + gm.n_bands - 1) / gm.n_bands;          // equal bands
    if (((long long)S * S >> chosen) * gm.n_bands > (1LL << 30)) return false;
    gm.kn = (float)(1.4426950408889634 / ((double)a.sigma_n + 1e-6));
    gm.kx = (float)(1.4426950408889634 / ((double)a.sigma_x + 1e-6));
    return true;
}

template <bool HASVAR>
hipError_t launch_cfg(const AtrousArgs &a, const LatticeGeom &gm, hipStream_t s)
{
    static SvgfLaunchCache cache;
    int dev_id = 0;
    if (hipError_t e = cache.init(reinterpret_cast<const void *>(&k_atrous_lattice<HASVAR>), kLdsBudget, &dev_id); e != hipSuccess) return e;
    const size_t lds = (size_t)(gm.pstride << gm.log2k) * (gm.band_rows + 4) * PXB;
    const unsigned nblocks = (((unsigned)a.step * (unsigned)a.step) >> gm.log2k) * (unsigned)gm.n_bands;
    SVGF_LAUNCH_KERNEL(k_atrous_lattice<HASVAR>, dim3(nblocks), dim3(NT), lds, s, a, gm);
    return hipGetLastError();
}

}  // namespace

bool atrous_lattice_supported(const AtrousArgs &a)
{
    LatticeGeom gm;
    return geometry(a, gm);
}

hipError_t launch_atrous_lattice(const AtrousArgs &a, hipStream_t s)
{
    LatticeGeom gm;
    if (!geometry(a, gm)) return hipErrorInvalidValue;
    return a.dst ? launch_cfg<true>(a, gm, s) : launch_cfg<false>(a, gm, s);
}
