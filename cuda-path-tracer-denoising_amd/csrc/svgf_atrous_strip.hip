// svgf_atrous_strip.hip — edge-avoiding a-trous level as an LDS strip-marching stencil for gfx950 (CDNA4).
//
// What it computes: one level of reference ATrousFilter (src/denoise.cu:77-170) with snapshot variance semantics.
//
// Why this shape.  A level with dilation S reads 25 taps at p + S*(i,j).  Pixels with the same y mod S form
// independent "lattice rows"; a tap row j of output row y is simply lattice row b+j (y = phase + S*b).  A workgroup
// therefore owns a strip of TX contiguous columns and ONE y-phase, and marches down that phase's lattice rows keeping
// the last 4+ROWS of them in an LDS ring: every input row is fetched from HBM/L2 once per strip (coalesced, 16 B
// and 12 B per lane), the 5x vertical reuse comes from the ring, the 5x horizontal reuse from reading the ring at
// x + S*i.  The only halo is 2*S columns left and right of the strip (and 2 lattice rows at the ends of a segment), so
// the same kernel serves S = 2 .. 32 where a square tile + halo could not (a 64x64 tile at S = 32 needs a 192x192
// halo tile = 1.7 MB of LDS; SURVEY.md §7 hard parts).
//
// LDS layout: array-of-pixels, 48 B per pixel = three 16-B slots
//     A = {n.x, p.x, n.y, p.y}   B = {n.z, p.z, luminance, variance}   C = {r, g, b, -}
// read with ds_read_b128.  The 48-B pixel stride is conflict-free for b128: a 16-lane group covers byte offsets
// 48*l, i.e. 12*l dwords mod 64 = 16 distinct 4-bank slots.
// Out-of-image pixels are staged with luminance = +inf, which makes their weight exp2(-inf) = 0 with no per-tap
// bounds test (the reference skips those taps, :134).
//
// Arithmetic (fast path): the reference's three edge-stopping factors exp(-a)*min(1,exp(-b))*min(1,exp(-c)) with
// a,b,c >= 0 equal exp(-(a+b+c)); it is evaluated as one v_exp_f32 of a base-2 exponent.  Non-finite
// normals/positions make the reference's min(1, exp(NaN)) return 1 (fminf semantics); a workgroup that stages such a
// texel switches to the `CAREFUL` tap routine that reproduces this, so ordinary frames pay nothing for it.
#include "svgf_kernels.h"

#include <cstdlib>

namespace {

constexpr float kLog2e = 1.44269504088896340736f;

struct StripGeom {
    int n_strips;   // strips of TX columns
    int n_segs;     // lattice-row segments per phase
    int seg_rows;   // lattice rows per segment
    int n_groups;   // S * n_segs
    float kn, kx;   // log2(e) / (sigma_n + 1e-6), log2(e) / (sigma_x + 1e-6)
};

struct Px {   // one staged pixel in registers
    float4 cv;
    float nx, ny, nz, px, py, pz;
    bool valid;
};

__device__ __forceinline__ float lum_f64(float r, float g, float b)
{   // reference luminance: double products, rounded once to float (src/denoise.cu:121,138)
    double l = 0.2126 * (double)r + 0.7152 * (double)g;
    l = l + 0.0722 * (double)b;
    return (float)l;
}

__device__ __forceinline__ bool finite3(float a, float b, float c)
{
    const float inf = __builtin_huge_valf();
    return (fabsf(a) < inf) && (fabsf(b) < inf) && (fabsf(c) < inf);
}

template <bool CAREFUL>
__device__ __forceinline__ void tap(const char *lds_px, float h, float lp, float kl, float kn, float kx,
                                    float npx, float npy, float npz, float ppx, float ppy, float ppz,
                                    float &c0, float &c1, float &c2, float &vsum, float &wsum, float &w2sum)
{
    const float4 A = *reinterpret_cast<const float4 *>(lds_px);
    const float4 B = *reinterpret_cast<const float4 *>(lds_px + 16);
    const float4 C = *reinterpret_cast<const float4 *>(lds_px + 32);
    const float dnx = A.x - npx, dny = A.z - npy, dnz = B.x - npz;
    const float dpx = A.y - ppx, dpy = A.w - ppy, dpz = B.y - ppz;
    float dn = __builtin_amdgcn_sqrtf(dnx * dnx + dny * dny + dnz * dnz);
    float dx = __builtin_amdgcn_sqrtf(dpx * dpx + dpy * dpy + dpz * dpz);
    if (CAREFUL) {          // min(1, exp(-NaN)) == 1 in the reference: a NaN distance contributes nothing
        dn = fmaxf(dn, 0.0f);
        dx = fmaxf(dx, 0.0f);
    }
    float e = fabsf(B.z - lp) * kl;
    e = fmaf(dn, kn, e);
    e = fmaf(dx, kx, e);
    const float w = h * __builtin_amdgcn_exp2f(-e);
    const float w2 = w * w;
    wsum += w;
    w2sum += w2;
    c0 = fmaf(C.x, w, c0);
    c1 = fmaf(C.y, w, c1);
    c2 = fmaf(C.z, w, c2);
    vsum = fmaf(B.w, w2, vsum);
}

template <int LOG2S, int TX, int ROWS>
__global__ __launch_bounds__(TX * ROWS) void k_atrous_strip(AtrousArgs a, StripGeom gm)
{
    constexpr int S = 1 << LOG2S;
    constexpr int RW = TX + 4 * S;          // staged pixels per lattice row
    constexpr int R = 4 + 2 * ROWS;         // ring slots: 4 + ROWS live, ROWS incoming
    constexpr int PPT = (RW + TX - 1) / TX; // staged pixels per thread per row
    constexpr int PXB = 48;                 // bytes per staged pixel

    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *nan_seen = reinterpret_cast<int *>(smem + (size_t)R * RW * PXB);

    // ---- work item: (strip, y-phase, segment); all strips of one (phase, segment) share an XCD's L2 ----
    const int bid = blockIdx.x;
    const int xcd = bid & 7, k = bid >> 3;
    const int g = xcd + 8 * (k / gm.n_strips);
    const int strip = k % gm.n_strips;
    if (g >= gm.n_groups) return;
    const int phase = g / gm.n_segs, seg = g % gm.n_segs;
    const int W = a.W, H = a.H;
    if (phase >= H) return;
    const int nb = (H - phase + S - 1) >> LOG2S;        // lattice rows in this phase
    const int b0 = seg * gm.seg_rows;
    const int b1 = min(b0 + gm.seg_rows, nb);
    if (b0 >= b1) return;
    const int x0 = strip * TX;

    const int tid = threadIdx.x;
    const int r = tid / TX;                 // which of the ROWS rows of an iteration this thread serves
    const int tx = tid - r * TX;

    if (tid == 0) *nan_seen = 0;

    auto slot_of = [&](int br) { return (br - (b0 - 2)) % R; };   // br >= b0-2 always

    // global -> registers for lattice row br, pixels tx + k*TX
    auto stage_load = [&](int br, Px(&px)[PPT]) {
        const int y = phase + (br << LOG2S);
        const bool row_ok = (br >= 0) && (y < H);
#pragma unroll
        for (int kk = 0; kk < PPT; kk++) {
            const int xi = tx + kk * TX;
            const int xs = x0 - 2 * S + xi;
            px[kk].valid = row_ok && (xi < RW) && (xs >= 0) && (xs < W);
            if (px[kk].valid) {
                const size_t q = (size_t)y * W + xs;
                px[kk].cv = a.src[q];
                const float *n = a.nrm + 3 * q;
                const float *p = a.pos + 3 * q;
                px[kk].nx = n[0]; px[kk].ny = n[1]; px[kk].nz = n[2];
                px[kk].px = p[0]; px[kk].py = p[1]; px[kk].pz = p[2];
            }
        }
    };
    // registers -> LDS ring slot
    auto stage_store = [&](int br, const Px(&px)[PPT]) {
        char *rowbase = smem + (size_t)slot_of(br) * RW * PXB;
#pragma unroll
        for (int kk = 0; kk < PPT; kk++) {
            const int xi = tx + kk * TX;
            if (xi < RW) {
                float4 A, B, C;
                if (px[kk].valid) {
                    A = make_float4(px[kk].nx, px[kk].px, px[kk].ny, px[kk].py);
                    B = make_float4(px[kk].nz, px[kk].pz, lum_f64(px[kk].cv.x, px[kk].cv.y, px[kk].cv.z), px[kk].cv.w);
                    C = make_float4(px[kk].cv.x, px[kk].cv.y, px[kk].cv.z, 0.0f);
                    if (!finite3(px[kk].nx, px[kk].ny, px[kk].nz) || !finite3(px[kk].px, px[kk].py, px[kk].pz)) *nan_seen = 1;
                } else {
                    A = make_float4(0.f, 0.f, 0.f, 0.f);
                    B = make_float4(0.f, 0.f, __builtin_huge_valf(), 0.f);   // luminance = +inf => weight 0
                    C = make_float4(0.f, 0.f, 0.f, 0.f);
                }
                char *d = rowbase + (size_t)xi * PXB;
                *reinterpret_cast<float4 *>(d) = A;
                *reinterpret_cast<float4 *>(d + 16) = B;
                *reinterpret_cast<float4 *>(d + 32) = C;
            }
        }
    };

    // variance of the two full-resolution neighbour rows (y-1, y+1) of output row bo, columns x-1..x+1, for the
    // 3x3 pre-blur (:102-118).  Own column per lane; the wave's edge lanes fetch their outer neighbour too.
    struct BlurPre { float vm, vp, em, ep; };
    const int lane = tid & 63;
    auto blur_load = [&](int bo, BlurPre &bp) {
        bp.vm = bp.vp = bp.em = bp.ep = 0.0f;
        const int x = x0 + tx;
        const int y = phase + (bo << LOG2S);
        if (!a.blur_variance || bo >= b1 || x >= W) return;
        const int xe = (lane == 0) ? x - 1 : x + 1;                       // edge lanes' extra column
        const bool edge = (lane == 0 || lane == 63) && xe >= 0 && xe < W;
        if (y - 1 >= 0) {
            bp.vm = a.src[(size_t)(y - 1) * W + x].w;
            if (edge) bp.em = a.src[(size_t)(y - 1) * W + xe].w;
        }
        if (y + 1 < H) {
            bp.vp = a.src[(size_t)(y + 1) * W + x].w;
            if (edge) bp.ep = a.src[(size_t)(y + 1) * W + xe].w;
        }
    };

    // ---- prologue: rows b0-2 .. b0+ROWS+1 into the ring ----
    {
        Px px[PPT];
        for (int rr = 0; rr < 4 + ROWS; rr += ROWS) {
            const int br = b0 - 2 + rr + r;
            if (rr + r < 4 + ROWS) {
                stage_load(br, px);
                stage_store(br, px);
            }
        }
    }
    BlurPre bp;
    blur_load(b0 + r, bp);
    __syncthreads();

    const float kn = gm.kn, kx = gm.kx;
    const int x = x0 + tx;

    for (int bc = b0; bc < b1; bc += ROWS) {
        // prefetch the next ROWS lattice rows (global -> registers) and the next blur rows
        Px nxt[PPT];
        const int bin = bc + ROWS + 2 + r;
        const bool more = (bc + ROWS < b1);
        if (more) stage_load(bin, nxt);
        BlurPre bpn;
        if (more) blur_load(bc + ROWS + r, bpn);

        const int bo = bc + r;
        if (bo < b1 && x < W) {
            const int y = phase + (bo << LOG2S);
            const char *rowc = smem + (size_t)slot_of(bo) * RW * PXB + (size_t)(tx + 2 * S) * PXB;
            const float4 A = *reinterpret_cast<const float4 *>(rowc);
            const float4 B = *reinterpret_cast<const float4 *>(rowc + 16);
            const float4 C = *reinterpret_cast<const float4 *>(rowc + 32);

            // centre variance
            float var = B.w;
            if (a.blur_variance) {
                // row y from the ring (columns x-1, x, x+1), rows y-1 / y+1 from the prefetch + lane shuffles
                const float v0l = *reinterpret_cast<const float *>(rowc - PXB + 28);
                const float v0r = *reinterpret_cast<const float *>(rowc + PXB + 28);
                const bool up = (y - 1 >= 0), dn = (y + 1 < H);
                const float wr_m = up ? 0.25f : 0.0f, wr_p = dn ? 0.25f : 0.0f;
                const float sv = wr_m * bp.vm + wr_p * bp.vp;             // vertical partial of own column (rows y-1,y+1)
                float svl = __shfl_up(sv, 1), svr = __shfl_down(sv, 1);
                const float se = wr_m * bp.em + wr_p * bp.ep;
                if (lane == 0) svl = se;
                if (lane == 63) svr = se;
                const bool lf = (x - 1 >= 0), rt = (x + 1 < W);
                const float wc_l = lf ? 0.25f : 0.0f, wc_r = rt ? 0.25f : 0.0f;
                // full column sums (row y weight 0.5)
                const float col_c = sv + 0.5f * B.w;
                const float col_l = svl + 0.5f * v0l;
                const float col_r = svr + 0.5f * v0r;
                const float sum = wc_l * col_l + 0.5f * col_c + wc_r * col_r;
                const float sumw = (wr_m + 0.5f + wr_p) * (wc_l + 0.5f + wc_r);
                var = sum / sumw;
            }
            var = fmaxf(var, 0.0f);
            const float kl = kLog2e / (__builtin_amdgcn_sqrtf(var) * a.sigma_c + 1e-6f);
            const float lp = B.z;

            float c0 = 0.f, c1 = 0.f, c2 = 0.f, vsum = 0.f, wsum = 0.f, w2sum = 0.f;
            const bool careful = (*nan_seen != 0);
            const char *colbase = smem + (size_t)tx * PXB;
            if (!careful) {
#pragma unroll
                for (int j = -2; j <= 2; j++) {
                    const char *rowp = colbase + (size_t)slot_of(bo + j) * RW * PXB;
                    const float hj = (j == 0) ? 6.0f : ((j == 1 || j == -1) ? 4.0f : 1.0f);
#pragma unroll
                    for (int i = -2; i <= 2; i++) {
                        const float hi = (i == 0) ? 6.0f : ((i == 1 || i == -1) ? 4.0f : 1.0f);
                        tap<false>(rowp + (i + 2) * S * PXB, hi * hj * (1.0f / 256.0f), lp, kl, kn, kx,
                                   A.x, A.z, B.x, A.y, A.w, B.y, c0, c1, c2, vsum, wsum, w2sum);
                    }
                }
            } else {
                for (int j = -2; j <= 2; j++) {
                    const char *rowp = colbase + (size_t)slot_of(bo + j) * RW * PXB;
                    const float hj = (j == 0) ? 6.0f : ((j == 1 || j == -1) ? 4.0f : 1.0f);
#pragma unroll
                    for (int i = -2; i <= 2; i++) {
                        const float hi = (i == 0) ? 6.0f : ((i == 1 || i == -1) ? 4.0f : 1.0f);
                        tap<true>(rowp + (i + 2) * S * PXB, hi * hj * (1.0f / 256.0f), lp, kl, kn, kx,
                                  A.x, A.z, B.x, A.y, A.w, B.y, c0, c1, c2, vsum, wsum, w2sum);
                    }
                }
            }

            float o0, o1, o2, ov;
            if (wsum > 1e-5f) {                                     // NaN -> false -> pass-through (:159-164)
                const float rw = __builtin_amdgcn_rcpf(wsum);
                o0 = c0 * rw; o1 = c1 * rw; o2 = c2 * rw;
                ov = vsum * __builtin_amdgcn_rcpf(w2sum);
            } else {
                o0 = C.x; o1 = C.y; o2 = C.z; ov = B.w;
            }
            const size_t p = (size_t)y * W + x;
            if (a.modulate) {                                      // last level: * albedo * ialbedo (:166-168)
                const float *t = a.gbuf + 13 * p;
                o0 *= t[6] * t[9]; o1 *= t[7] * t[10]; o2 *= t[8] * t[11];
            }
            if (a.dst) a.dst[p] = make_float4(o0, o1, o2, ov);
            if (a.out_rgb) { float *o = a.out_rgb + 3 * p; o[0] = o0; o[1] = o1; o[2] = o2; }
        }

        // the incoming rows go to the slots that held rows bc-2 .. bc-2+ROWS-1's predecessors: slot_of(bin) is not
        // read by this iteration (it reads rows bc-2 .. bc+ROWS+1), so no barrier is needed before the store.
        if (more) stage_store(bin, nxt);
        bp = bpn;
        __syncthreads();
    }
}

struct StripCfg { int log2s, tx, rows; };

template <int LOG2S, int TX, int ROWS>
hipError_t launch_cfg(const AtrousArgs &a, hipStream_t s)
{
    constexpr int S = 1 << LOG2S, RW = TX + 4 * S, R = 4 + 2 * ROWS;
    const size_t lds = (size_t)R * RW * 48 + 16;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_atrous_strip<LOG2S, TX, ROWS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    StripGeom gm;
    gm.n_strips = (a.W + TX - 1) / TX;
    const int nb_max = (a.H + S - 1) / S;
    // enough (strip, phase, segment) items to fill 256 CUs a few times over, but segments of at least 16 rows
    int segs = 1;
    const int target = 1024;
    while (gm.n_strips * S * segs < target && (nb_max + segs) / (segs + 1) >= 16) segs++;
    if (const char *e = getenv("SVGF_STRIP_SEGS")) { int v = atoi(e); if (v > 0) segs = v; }
    gm.n_segs = segs;
    gm.seg_rows = (nb_max + segs - 1) / segs;
    gm.seg_rows = ((gm.seg_rows + ROWS - 1) / ROWS) * ROWS;
    gm.n_groups = S * gm.n_segs;
    gm.kn = (float)(1.4426950408889634 / ((double)a.sigma_n + 1e-6));
    gm.kx = (float)(1.4426950408889634 / ((double)a.sigma_x + 1e-6));
    const int groups_pad = (gm.n_groups + 7) / 8 * 8;
    const int nblocks = groups_pad * gm.n_strips;
    hipLaunchKernelGGL((k_atrous_strip<LOG2S, TX, ROWS>), dim3(nblocks), dim3(TX * ROWS), lds, s, a, gm);
    return hipGetLastError();
}

// default configuration per dilation; SVGF_STRIP_TX / SVGF_STRIP_ROWS override for tuning runs
void pick(int log2s, int &tx, int &rows)
{
    static const int def_tx[6] = { 0, 128, 128, 256, 256, 256 };
    static const int def_rows[6] = { 0, 2, 2, 2, 2, 1 };
    tx = def_tx[log2s]; rows = def_rows[log2s];
    if (const char *e = getenv("SVGF_STRIP_TX")) { int v = atoi(e); if (v == 64 || v == 128 || v == 256) tx = v; }
    if (const char *e = getenv("SVGF_STRIP_ROWS")) { int v = atoi(e); if (v == 1 || v == 2) rows = v; }
    // LDS budget: (4 + 2*rows) * (tx + 4S) * 48 + 16 <= 160 KiB
    const int S = 1 << log2s;
    while ((size_t)(4 + 2 * rows) * (tx + 4 * S) * 48 + 16 > 160 * 1024 && rows > 1) rows--;
}

}  // namespace

bool atrous_strip_supported(const AtrousArgs &a)
{
    if (a.step < 2 || a.step > 32 || (a.step & (a.step - 1))) return false;
    if ((long long)a.W * a.H >= (1LL << 31)) return false;
    return true;
}

#define STRIP_CASE(L, T, Rr) if (log2s == L && tx == T && rows == Rr) return launch_cfg<L, T, Rr>(a, s);

hipError_t launch_atrous_strip(const AtrousArgs &a, hipStream_t s)
{
    int log2s = 0;
    while ((1 << log2s) < a.step) log2s++;
    int tx, rows;
    pick(log2s, tx, rows);
    STRIP_CASE(1, 64, 1) STRIP_CASE(1, 64, 2) STRIP_CASE(1, 128, 1) STRIP_CASE(1, 128, 2) STRIP_CASE(1, 256, 1) STRIP_CASE(1, 256, 2)
    STRIP_CASE(2, 64, 1) STRIP_CASE(2, 64, 2) STRIP_CASE(2, 128, 1) STRIP_CASE(2, 128, 2) STRIP_CASE(2, 256, 1) STRIP_CASE(2, 256, 2)
    STRIP_CASE(3, 64, 1) STRIP_CASE(3, 64, 2) STRIP_CASE(3, 128, 1) STRIP_CASE(3, 128, 2) STRIP_CASE(3, 256, 1) STRIP_CASE(3, 256, 2)
    STRIP_CASE(4, 64, 1) STRIP_CASE(4, 64, 2) STRIP_CASE(4, 128, 1) STRIP_CASE(4, 128, 2) STRIP_CASE(4, 256, 1) STRIP_CASE(4, 256, 2)
    STRIP_CASE(5, 64, 1) STRIP_CASE(5, 64, 2) STRIP_CASE(5, 128, 1) STRIP_CASE(5, 128, 2) STRIP_CASE(5, 256, 1) STRIP_CASE(5, 256, 2)
    return hipErrorInvalidValue;
}
