// svgf_atrous_share.hip — a-trous level with SHARED geometric weights, for dilations S = 2, 4, 8 (gfx950).
//
// Same result as svgf_atrous_strip.hip (one level of reference ATrousFilter, src/denoise.cu:77-170, snapshot variance),
// same strip-marching decomposition, same loader-wave staging.  What changes is the arithmetic, which bounds the
// kernel (fp32 VALU): the geometric part of the edge-stopping exponent,
//     t(p,q) = log2(1/h) + kn*|n_p - n_q| + kx*|x_p - x_q|,
// is symmetric in (p,q), and every pixel pair (p,q) of a level is a tap of p AND a tap of q.  The strip kernel
// evaluates it twice (6 packed ops + 2 v_sqrt each time).  Here each pair is evaluated once:
//     pass G  every pixel computes t for its 12 "forward" partners (rows below, and right neighbours in its own row)
//             and publishes them in an LDS ring T;                                    12 x (6 pk + 2 sqrt + 2 fma)
//     pass C  every output pixel walks its 24 taps with t taken from its own registers (forward taps) or from the
//             partner's T entry (backward taps) and does only the colour part:        24 x (sub, fma, v_exp, mul, 3 pk)
// i.e. ~0.69x the VALU work of the strip kernel per output pixel.
//
// LDS (TX = 256 columns, RW = TX + 4S staged columns, 2 output rows per iteration):
//     GA[6][RW] float4 {n.x,p.x,n.y,p.y}, GB[6][RW] float2 {n.z,p.z}   geometry of lattice rows bc .. bc+3 (+2 incoming)
//     KC[8][RW] float4 {r,g,b,var},       KL[8][RW] float  luminance    colour of lattice rows bc-2 .. bc+3 (+2 incoming)
//     T [4][12][RW] float                                               forward terms of rows bc-2 .. bc+1
//     blur rows (as in the strip kernel)
// = 151 KB at S = 8; S = 16, 32 do not fit and stay on the strip kernel.
//
// Work-group = 512 compute threads + 2 loader groups of 128 (as in the strip kernel).  T entries of the 2S halo columns
// on either side (needed by the outermost output columns) are computed by the loader group about to issue its loads in
// that iteration; the forward terms of the two halo rows above a segment are computed once in the prologue.
// Two barriers per iteration: T published -> pass C;  pass C done / new rows committed -> next pass G.
//
// STATUS (round 1): EXPERIMENTAL, opt-in (SvgfParams::kernel_variant = 3).  Results are correct (tests/test_parity_gpu.py
// runs it against the reference goldens).  With two loader groups (12 waves, 168 VGPRs: no spilling) and the six-stage
// software pipeline below it runs 66-71 us per 1080p level against 55 us for the strip kernel: the second barrier per
// iteration cuts the work into short phases in which the waves of a SIMD stall together.
#include "svgf_kernels.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace {

constexpr float kLog2e = 1.44269504088896340736f;
constexpr int TX = 256, ROWS = 2;
constexpr int kLoaderGroups = 2, kLoaderGroup = 128, kLoaderThreads = kLoaderGroups * kLoaderGroup;
constexpr int NC = TX * ROWS, NT = NC + kLoaderThreads;
constexpr int RG = 6, RK = 8, RT = 4, NF = 12;

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

struct ShareGeom {
    int n_strips, n_segs, seg_rows, n_groups;
    float kn, kx;
    unsigned long long *dbg;   // tuning only (SVGF_SHARE_DBG): per-phase s_memtime stamps of one workgroup
    int dbg_block;
};

struct Px {
    float4 cv;
    float nx, ny, nz, px, py, pz;
    int row_col;   // (ring row index br - (b0-2)) << 16 | xi ; bit 31 set = out-of-image pixel
};

__device__ __forceinline__ float lum_f64(float r, float g, float b)
{   // reference luminance: double products, rounded once to float (src/denoise.cu:121,138)
    double l = 0.2126 * (double)r + 0.7152 * (double)g;
    l = l + 0.0722 * (double)b;
    return (float)l;
}

__device__ __forceinline__ constexpr float neg_log2_binom(int i)
{
    return (i == 0) ? 1.4150374992788437f : ((i == 1 || i == -1) ? 2.0f : 4.0f);
}
// forward partner f: f = 0..4 -> (i = f-2, j = 1); 5..9 -> (i = f-7, j = 2); 10 -> (1,0); 11 -> (2,0)
__device__ __forceinline__ constexpr int fwd_i(int f) { return f < 5 ? f - 2 : (f < 10 ? f - 7 : f - 9); }
__device__ __forceinline__ constexpr int fwd_j(int f) { return f < 5 ? 1 : (f < 10 ? 2 : 0); }

template <int LOG2S>
struct Lds {
    static constexpr int S = 1 << LOG2S;
    static constexpr int RW = TX + 4 * S;
    static constexpr int BW = TX + 2;
    static constexpr int GA = 0;
    static constexpr int GB = GA + RG * RW * 16;
    static constexpr int KC = GB + RG * RW * 8;
    static constexpr int KL = KC + RK * RW * 16;
    static constexpr int T = KL + RK * RW * 4;
    static constexpr int BLUR = T + RT * RW * NF * 4;
    static constexpr int NAN_SEEN = BLUR + 2 * ROWS * 2 * BW * 4;
    static constexpr int BYTES = NAN_SEEN + 16;
};

template <int LOG2S>
__global__ __launch_bounds__(NT) void k_atrous_share(AtrousArgs a, ShareGeom gm)
{
    using L = Lds<LOG2S>;
    constexpr int S = L::S, RW = L::RW, BW = L::BW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *blur = reinterpret_cast<float *>(smem + L::BLUR);
    int *nan_seen = reinterpret_cast<int *>(smem + L::NAN_SEEN);

    // ---- work item (same mapping as the strip kernel) ----
    const int bid = blockIdx.x;
    const int xcd = bid & 7, kk = bid >> 3;
    const int g = xcd + 8 * (kk / gm.n_strips);
    const int strip = kk % gm.n_strips;
    if (g >= gm.n_groups) return;
    const int phase = g / gm.n_segs, seg = g % gm.n_segs;
    const int W = a.W, H = a.H;
    if (phase >= H) return;
    const int nb = (H - phase + S - 1) >> LOG2S;
    const int b0 = seg * gm.seg_rows;
    const int b1 = min(b0 + gm.seg_rows, nb);
    if (b0 >= b1) return;
    const int x0 = strip * TX;
    const int tid = threadIdx.x;
    if (tid == 0) *nan_seen = 0;
    int dbg_it = 0;
    auto stamp = [&](int phase_id) {
        if (gm.dbg && bid == gm.dbg_block && (tid & 63) == 0 && dbg_it < 16)
            gm.dbg[((tid >> 6) * 16 + dbg_it) * 8 + phase_id] = __builtin_amdgcn_s_memtime();
    };

    // ring slots of lattice row br, given rel = br - (b0 - 2) >= 0
    auto slotG = [&](int rel) { return rel % RG; };
    auto slotK = [&](int rel) { return rel % RK; };
    auto slotT = [&](int rel) { return rel % RT; };

    // ---------------- staging (global -> registers -> LDS), branch-free, coordinates clamped ----------------
    auto rows_load = [&](auto &px, int br_first, int nrows, int wi, int nw) {
        constexpr int M = sizeof(px) / sizeof(px[0]);
        const int total = nrows * RW;
#pragma unroll
        for (int m = 0; m < M; m++) {
            const int idx = min(wi + m * nw, total - 1);
            const int rr = idx / RW, xi = idx - rr * RW;
            const int br = br_first + rr;
            const int y = phase + (br << LOG2S);
            const int xs = x0 - 2 * S + xi;
            const bool ok = (br >= 0) && (y < H) && (xs >= 0) && (xs < W);
            px[m].row_col = (((br - (b0 - 2)) << 16) | xi) | (ok ? 0 : (int)0x80000000);
            const unsigned q = (unsigned)min(max(y, 0), H - 1) * (unsigned)W + (unsigned)min(max(xs, 0), W - 1);
            px[m].cv = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(a.src) + q * 16u);
            const float *n = reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.nrm) + q * 12u);
            const float *p = reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.pos) + q * 12u);
            px[m].nx = n[0]; px[m].ny = n[1]; px[m].nz = n[2];
            px[m].px = p[0]; px[m].py = p[1]; px[m].pz = p[2];
        }
    };
    auto rows_store = [&](const auto &px) {
        constexpr int M = sizeof(px) / sizeof(px[0]);
        const float inf = __builtin_huge_valf();
#pragma unroll
        for (int m = 0; m < M; m++) {
            const bool ok = px[m].row_col >= 0;
            const int rel = (px[m].row_col & 0x7fffffff) >> 16, xi = px[m].row_col & 0xffff;
            const float lum = lum_f64(px[m].cv.x, px[m].cv.y, px[m].cv.z);
            const float mag = fabsf(px[m].nx) + fabsf(px[m].ny) + fabsf(px[m].nz) + fabsf(px[m].px) + fabsf(px[m].py) + fabsf(px[m].pz);
            if (!(mag < inf)) *nan_seen = 1;
            const int og = slotG(rel) * RW + xi, ok_ = slotK(rel) * RW + xi;
            *reinterpret_cast<float4 *>(smem + L::GA + og * 16) = make_float4(px[m].nx, px[m].px, px[m].ny, px[m].py);
            *reinterpret_cast<float2 *>(smem + L::GB + og * 8) = make_float2(px[m].nz, px[m].pz);
            *reinterpret_cast<float4 *>(smem + L::KC + ok_ * 16) = ok ? px[m].cv : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float *>(smem + L::KL + ok_ * 4) = ok ? lum : inf;     // +inf => weight 0
        }
    };
    auto blur_load = [&](auto &v, int bo_first, int wi, int nw) {
        constexpr int M = sizeof(v) / sizeof(v[0]);
#pragma unroll
        for (int m = 0; m < M; m++) {
            const int idx = wi + m * nw;
            v[m] = 0.0f;
            if (a.blur_variance && idx < ROWS * 2 * BW) {
                const int rr = idx / (2 * BW), rem = idx - rr * (2 * BW);
                const int d = rem / BW, xi = rem - d * BW;
                const int y = phase + ((bo_first + rr) << LOG2S) + (d ? 1 : -1);
                const int xs = x0 - 1 + xi;
                if (y >= 0 && y < H && xs >= 0 && xs < W && bo_first + rr < b1) v[m] = a.src[(unsigned)y * (unsigned)W + (unsigned)xs].w;
            }
        }
    };
    auto blur_store = [&](const auto &v, int parity, int wi, int nw) {
        constexpr int M = sizeof(v) / sizeof(v[0]);
#pragma unroll
        for (int m = 0; m < M; m++) {
            const int idx = wi + m * nw;
            if (idx < ROWS * 2 * BW) blur[parity * (ROWS * 2 * BW) + idx] = v[m];
        }
    };

    // ---------------- pass G: forward terms of pixel (xi, lattice row rel) -> tF[] and (optionally) the T ring -------
    const float kn = gm.kn, kx = gm.kx;
    // Addresses are (per-row base register) + (compile-time tap offset): the three row bases are computed once per call.
    // HALO = true (loader threads, halo columns): partner columns are clamped into the staged row (their entries
    // are never read), which costs per-tap address arithmetic; compute threads never need it.
    auto wrap = [](int v, int n) { return v >= n ? v - n : (v < 0 ? v + n : v); };   // v in (-n, 2n)
    auto pass_g = [&](auto careful_tag, auto halo_tag, int rel, int xi, float (&tF)[NF]) {
        constexpr bool CAREFUL = decltype(careful_tag)::value;
        constexpr bool HALO = decltype(halo_tag)::value;
        int rowA[3], rowB[3];
        const int sg = slotG(rel);
#pragma unroll
        for (int jj = 0; jj < 3; jj++) {
            const int o = wrap(sg + jj, RG) * RW + (HALO ? 0 : xi);
            rowA[jj] = L::GA + o * 16;
            rowB[jj] = L::GB + o * 8;
        }
        const v4f Ac = *reinterpret_cast<const v4f *>(smem + rowA[0] + (HALO ? xi * 16 : 0));
        const v2f Bc = *reinterpret_cast<const v2f *>(smem + rowB[0] + (HALO ? xi * 8 : 0));
        const v2f c0 = Ac.xy, c1 = Ac.zw;
        // two batches of six partners: bounds the live registers
#pragma unroll
        for (int half = 0; half < 2; half++) {
            v2f s2[6];
#pragma unroll
            for (int k = 0; k < 6; k++) {
                const int f = half * 6 + k;
                const int dxq = HALO ? min(max(xi + fwd_i(f) * S, 0), RW - 1) : fwd_i(f) * S;
                const v4f Aq = *reinterpret_cast<const v4f *>(smem + rowA[fwd_j(f)] + dxq * 16);
                const v2f Bq = *reinterpret_cast<const v2f *>(smem + rowB[fwd_j(f)] + dxq * 8);
                const v2f d0 = Aq.xy - c0, d1 = Aq.zw - c1, d2 = Bq - Bc;
                v2f t = d0 * d0;
                t = __builtin_elementwise_fma(d1, d1, t);
                s2[k] = __builtin_elementwise_fma(d2, d2, t);
            }
            __builtin_amdgcn_sched_barrier(0x100);
            float dn[6], dx[6];
#pragma unroll
            for (int k = 0; k < 6; k++) {
                dn[k] = __builtin_amdgcn_sqrtf(s2[k].x);
                dx[k] = __builtin_amdgcn_sqrtf(s2[k].y);
            }
            __builtin_amdgcn_sched_barrier(0x100);
#pragma unroll
            for (int k = 0; k < 6; k++) {
                const int f = half * 6 + k;
                float n_ = dn[k], x_ = dx[k];
                if (CAREFUL) { n_ = fmaxf(n_, 0.0f); x_ = fmaxf(x_, 0.0f); }     // min(1, exp(-NaN)) == 1 in the reference
                float t = fmaf(n_, kn, neg_log2_binom(fwd_i(f)) + neg_log2_binom(fwd_j(f)));
                tF[f] = fmaf(x_, kx, t);
            }
        }
        // T[slot][f][column]: a backward reader takes one f at consecutive columns -> conflict-free ds_read_b32
        char *trow = smem + L::T + (slotT(rel) * NF * RW + xi) * 4;
#pragma unroll
        for (int f = 0; f < NF; f++) *reinterpret_cast<float *>(trow + f * RW * 4) = tF[f];
    };
    auto pass_g_dispatch = [&](auto halo_tag, int rel, int xi, float (&tF)[NF]) {
        if (*nan_seen != 0) pass_g(std::true_type{}, halo_tag, rel, xi, tF);
        else pass_g(std::false_type{}, halo_tag, rel, xi, tF);
    };

    // ---------------- loader bookkeeping (identical scheme to the strip kernel) ----------------
    constexpr int ML = (ROWS * RW + kLoaderGroup - 1) / kLoaderGroup;
    constexpr int MBL = (ROWS * 2 * BW + kLoaderGroup - 1) / kLoaderGroup;
    const bool is_loader = (tid >= NC);
    const int lgroup = is_loader ? (tid - NC) / kLoaderGroup : -1;
    const int llane = is_loader ? (tid - NC) % kLoaderGroup : 0;
    Px lpx[ML];
    float lbv[MBL];
    auto loader_issue = [&](int j) {
        const int bcj = b0 + j * ROWS;
        if (bcj < b1) {
            rows_load(lpx, bcj + 2, ROWS, llane, kLoaderGroup);     // new rows of iteration j: bcj+2, bcj+3
            blur_load(lbv, bcj, llane, kLoaderGroup);
        }
    };
    auto loader_commit = [&](int j) {
        const int bcj = b0 + j * ROWS;
        if (bcj < b1) {
            rows_store(lpx);
            if (a.blur_variance) blur_store(lbv, j & 1, llane, kLoaderGroup);
        }
    };

    // ---------------- prologue: rows b0-2 .. b0+3, blur rows of iteration 0, forward terms of halo rows b0-2, b0-1 -----
    {
        constexpr int M = (6 * RW + NT - 1) / NT;
        Px px[M];
        rows_load(px, b0 - 2, 6, tid, NT);
        constexpr int MB = (ROWS * 2 * BW + NT - 1) / NT;
        float bv[MB];
        blur_load(bv, b0, tid, NT);
        rows_store(px);
        if (a.blur_variance) blur_store(bv, 0, tid, NT);
    }
    if (is_loader && lgroup >= 1) loader_issue(lgroup);
    __syncthreads();
    {
        float tdrop[NF];
        if (tid < 2 * RW) pass_g_dispatch(std::true_type{}, tid / RW, tid % RW, tdrop);   // rel 0,1 = rows b0-2, b0-1, every column
    }
    __syncthreads();

    if (is_loader) {
        // ================================ loader waves ================================
        __builtin_amdgcn_s_setprio(3);
        int it = 0;
        for (int bc = b0; bc < b1; bc += ROWS, it++, dbg_it++) {
            stamp(0);
            if ((it + 1) % kLoaderGroups == lgroup) loader_commit(it + 1);
            else if (it % kLoaderGroups == lgroup) {
                // the group whose turn it is to issue first computes the forward terms of the 2S halo columns left and
                // right (rows bc, bc+1) — its registers are free at this point — and then puts its loads in flight
                if (llane < 8 * S) {
                    const int rr = llane / (4 * S), hc = llane % (4 * S);
                    const int xi = hc < 2 * S ? hc : (TX + hc);             // [0,2S) and [TX+2S, RW)
                    float tdrop[NF];
                    pass_g_dispatch(std::true_type{}, bc - (b0 - 2) + rr, xi, tdrop);
                }
                loader_issue(it + kLoaderGroups);
            }
            stamp(1);
#ifndef SVGF_SHARE_HACK_NO_A
            __syncthreads();     // A: T published
#endif
            stamp(2);
            __syncthreads();     // B: pass C done, new rows committed
            stamp(3);
        }
        return;
    }

    // ================================ compute waves ================================
    const int r = tid / TX;
    const int tx = tid - r * TX;
    const int x = x0 + tx;
    const int xi = tx + 2 * S;
    // One iteration of a compute thread is a software pipeline of six stages whose LDS reads are issued one stage ahead
    // of their arithmetic (sched_barrier(0) pins the stage boundaries; registers: two batches of 6 partners in flight):
    //   G0 G1   forward terms of partners 0-5 / 6-11 (geometry rows rel .. rel+2)      -> tF[] and the T ring
    //   F0 F1   colour part of the 12 FORWARD taps, t from tF[] (no other thread's data)
    //   ---- barrier A: every T entry of rows bc-2 .. bc+1 is published ----
    //   B0 B1   colour part of the 12 BACKWARD taps, t = the partner's T entry
    struct GBatch { v4f A[6]; v2f B[6]; };
    struct CBatch { v4f C[6]; float l[6]; };
    auto body = [&](auto careful_tag, int bc, int it) {
        constexpr bool CAREFUL = decltype(careful_tag)::value;
        const int bo = bc + r;
        const int rel = bo - (b0 - 2);
        const bool live = (bo < b1 && x < W);
        const int y = phase + (bo << LOG2S);
        stamp(0);

        // row bases (one modulo each); every tap address is base + compile-time offset
        int ga[3], gb[3], kcb[5], klb[5], tb[3];
        const int sg = slotG(rel), sk = slotK(rel), st = slotT(rel);
#pragma unroll
        for (int jj = 0; jj < 3; jj++) {
            const int o = wrap(sg + jj, RG) * RW + xi;
            ga[jj] = L::GA + o * 16;
            gb[jj] = L::GB + o * 8;
            tb[jj] = L::T + (wrap(st - jj, RT) * NF * RW + xi) * 4;
        }
#pragma unroll
        for (int jj = 0; jj < 5; jj++) {
            const int o = wrap(sk + jj - 2, RK) * RW + xi;
            kcb[jj] = L::KC + o * 16;
            klb[jj] = L::KL + o * 4;
        }
        auto g_load = [&](auto half_tag, GBatch &b) {
            constexpr int HALF = decltype(half_tag)::value;
#pragma unroll
            for (int k = 0; k < 6; k++) {
                const int f = HALF * 6 + k;
                b.A[k] = *reinterpret_cast<const v4f *>(smem + ga[fwd_j(f)] + fwd_i(f) * S * 16);
                b.B[k] = *reinterpret_cast<const v2f *>(smem + gb[fwd_j(f)] + fwd_i(f) * S * 8);
            }
        };
        auto c_load = [&](auto back_tag, auto half_tag, CBatch &b) {
            constexpr bool BACK = decltype(back_tag)::value;
            constexpr int HALF = decltype(half_tag)::value;
#pragma unroll
            for (int k = 0; k < 6; k++) {
                const int f = HALF * 6 + k;
                const int di = BACK ? -fwd_i(f) : fwd_i(f), dj = BACK ? -fwd_j(f) : fwd_j(f);
                b.C[k] = *reinterpret_cast<const v4f *>(smem + kcb[dj + 2] + di * S * 16);
                b.l[k] = *reinterpret_cast<const float *>(smem + klb[dj + 2] + di * S * 4);
            }
        };

        // ---- issue: centre geometry + both G batches ----
        GBatch g0, g1;
        const v4f Ac = *reinterpret_cast<const v4f *>(smem + ga[0]);
        const v2f Bc = *reinterpret_cast<const v2f *>(smem + gb[0]);
        g_load(std::integral_constant<int, 0>{}, g0);
        const float4 C = *reinterpret_cast<const float4 *>(smem + kcb[2]);
        const float lp = *reinterpret_cast<const float *>(smem + klb[2]);

        // centre variance: 3x3 gaussian with out-of-image taps dropped and renormalised (:102-118)
        float var = C.w;
        if (a.blur_variance) {
            const float *bl = blur + (it & 1) * (ROWS * 2 * BW) + r * (2 * BW) + tx;
            const float m0 = bl[0], m1 = bl[1], m2 = bl[2];
            const float p0 = bl[BW], p1 = bl[BW + 1], p2 = bl[BW + 2];
            const float c0v = *reinterpret_cast<const float *>(smem + kcb[2] - 16 + 12);
            const float c2v = *reinterpret_cast<const float *>(smem + kcb[2] + 16 + 12);
            const float wr_m = (y - 1 >= 0) ? 0.25f : 0.0f, wr_p = (y + 1 < H) ? 0.25f : 0.0f;
            const float wc_l = (x - 1 >= 0) ? 0.25f : 0.0f, wc_r = (x + 1 < W) ? 0.25f : 0.0f;
            const float col_l = wr_m * m0 + 0.5f * c0v + wr_p * p0;
            const float col_c = wr_m * m1 + 0.5f * C.w + wr_p * p1;
            const float col_r = wr_m * m2 + 0.5f * c2v + wr_p * p2;
            const float sum = wc_l * col_l + 0.5f * col_c + wc_r * col_r;
            const float sumw = (wr_m + 0.5f + wr_p) * (wc_l + 0.5f + wc_r);
            var = sum * __builtin_amdgcn_rcpf(sumw);
        }
        var = fmaxf(var, 0.0f);
        const float kl = kLog2e * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(var) * a.sigma_c + 1e-6f);
        // centre tap: weight exactly h = 9/64
        constexpr float w0 = 0.140625f;
        v2f acc_ww = v2f{w0, w0 * w0};
        v2f acc_rg = v2f{w0 * C.x, w0 * C.y};
        v2f acc_bv = v2f{w0 * C.z, (w0 * w0) * C.w};

        float tF[NF];
        const v2f c0 = Ac.xy, c1 = Ac.zw;
        auto g_math = [&](auto half_tag, const GBatch &b) {
            constexpr int HALF = decltype(half_tag)::value;
            v2f s2[6];
#pragma unroll
            for (int k = 0; k < 6; k++) {
                const v2f d0 = b.A[k].xy - c0, d1 = b.A[k].zw - c1, d2 = b.B[k] - Bc;
                v2f t = d0 * d0;
                t = __builtin_elementwise_fma(d1, d1, t);
                s2[k] = __builtin_elementwise_fma(d2, d2, t);
            }
            __builtin_amdgcn_sched_barrier(0x100);
            float dn[6], dx[6];
#pragma unroll
            for (int k = 0; k < 6; k++) {
                dn[k] = __builtin_amdgcn_sqrtf(s2[k].x);
                dx[k] = __builtin_amdgcn_sqrtf(s2[k].y);
            }
            __builtin_amdgcn_sched_barrier(0x100);
#pragma unroll
            for (int k = 0; k < 6; k++) {
                const int f = HALF * 6 + k;
                float n_ = dn[k], x_ = dx[k];
                if (CAREFUL) { n_ = fmaxf(n_, 0.0f); x_ = fmaxf(x_, 0.0f); }     // min(1, exp(-NaN)) == 1 in the reference
                const float t = fmaf(n_, kn, neg_log2_binom(fwd_i(f)) + neg_log2_binom(fwd_j(f)));
                tF[f] = fmaf(x_, kx, t);
            }
        };
        auto c_math = [&](const CBatch &b, const float (&tt)[6]) {
            float w[6];
#pragma unroll
            for (int k = 0; k < 6; k++) w[k] = fmaf(fabsf(b.l[k] - lp), kl, tt[k]);
            __builtin_amdgcn_sched_barrier(0x100);
#pragma unroll
            for (int k = 0; k < 6; k++) w[k] = __builtin_amdgcn_exp2f(-w[k]);
            __builtin_amdgcn_sched_barrier(0x100);
#pragma unroll
            for (int k = 0; k < 6; k++) {
                v2f wv;
                wv.x = w[k];
                wv.y = w[k] * w[k];
                acc_ww += wv;
                acc_rg = __builtin_elementwise_fma(b.C[k].xy, v2f{w[k], w[k]}, acc_rg);
                acc_bv = __builtin_elementwise_fma(b.C[k].zw, wv, acc_bv);
            }
        };

        CBatch f0, f1, k0, k1;
        __builtin_amdgcn_sched_barrier(0);
        g_load(std::integral_constant<int, 1>{}, g1);
        g_math(std::integral_constant<int, 0>{}, g0);
        __builtin_amdgcn_sched_barrier(0);
        c_load(std::false_type{}, std::integral_constant<int, 0>{}, f0);
        g_math(std::integral_constant<int, 1>{}, g1);
        __builtin_amdgcn_sched_barrier(0);
        c_load(std::false_type{}, std::integral_constant<int, 1>{}, f1);
        {   // T[slot][f][column]: a backward reader takes one f at consecutive columns -> conflict-free ds_read_b32
            char *trow = smem + tb[0];
#pragma unroll
            for (int f = 0; f < NF; f++) *reinterpret_cast<float *>(trow + f * RW * 4) = tF[f];
        }
        stamp(1);
        {
            const float tt[6] = {tF[0], tF[1], tF[2], tF[3], tF[4], tF[5]};
            c_math(f0, tt);
        }
        __builtin_amdgcn_sched_barrier(0);
        c_load(std::true_type{}, std::integral_constant<int, 0>{}, k0);
        {
            const float tt[6] = {tF[6], tF[7], tF[8], tF[9], tF[10], tF[11]};
            c_math(f1, tt);
        }
        __builtin_amdgcn_sched_barrier(0);
        c_load(std::true_type{}, std::integral_constant<int, 1>{}, k1);
        stamp(2);
#ifndef SVGF_SHARE_HACK_NO_A
        __syncthreads();     // A: every T entry of rows bc-2 .. bc+1 is published
#endif
        stamp(3);

        // backward: partner q = p - (i*S, j rows) published the term as ITS forward entry f
        float tb0[6], tb1[6];
#pragma unroll
        for (int k = 0; k < 6; k++) {
            tb0[k] = *reinterpret_cast<const float *>(smem + tb[fwd_j(k)] + (k * RW - fwd_i(k) * S) * 4);
            tb1[k] = *reinterpret_cast<const float *>(smem + tb[fwd_j(6 + k)] + ((6 + k) * RW - fwd_i(6 + k) * S) * 4);
        }
        c_math(k0, tb0);
        c_math(k1, tb1);

        if (live) {
            const float o_r = acc_rg.x, o_g = acc_rg.y, o_b = acc_bv.x, vsum = acc_bv.y, wsum = acc_ww.x, w2sum = acc_ww.y;
            float o0, o1, o2, ov;
            if (wsum > 1e-5f) {                                     // NaN -> false -> pass-through (:159-164)
                const float rw = __builtin_amdgcn_rcpf(wsum);
                o0 = o_r * rw; o1 = o_g * rw; o2 = o_b * rw;
                ov = vsum * __builtin_amdgcn_rcpf(w2sum);
            } else {
                o0 = C.x; o1 = C.y; o2 = C.z; ov = C.w;
            }
            const unsigned p = (unsigned)y * (unsigned)W + (unsigned)x;
            if (a.modulate) {                                      // last level: * albedo * ialbedo (:166-168)
                const float *t = a.gbuf + 13u * (size_t)p;
                o0 *= t[6] * t[9]; o1 *= t[7] * t[10]; o2 *= t[8] * t[11];
            }
            if (a.dst) a.dst[p] = make_float4(o0, o1, o2, ov);
            if (a.out_rgb) { float *o = a.out_rgb + 3u * p; o[0] = o0; o[1] = o1; o[2] = o2; }
        }
        stamp(4);
        __syncthreads();     // B: T rows and ring slots may be overwritten
        stamp(5);
    };

    int it = 0;
    for (int bc = b0; bc < b1; bc += ROWS, it++, dbg_it++) {
        if (*nan_seen != 0) body(std::true_type{}, bc, it);       // wave-uniform: every thread reads the same LDS word
        else body(std::false_type{}, bc, it);
    }
}

template <int LOG2S>
hipError_t launch_share_cfg(const AtrousArgs &a, hipStream_t s)
{
    using L = Lds<LOG2S>;
    constexpr int S = L::S;
    const size_t lds = L::BYTES;
    // per device: one process may own contexts on several GPUs (include/svgf.h: handle-based), and the opt-in to more
    // than 64 KB of dynamic LDS is a per-device function attribute
    static bool attr_done[64] = {};
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    if (dev_id < 0 || dev_id >= 64) dev_id = 0;
    if (!attr_done[dev_id]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_atrous_share<LOG2S>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done[dev_id] = true;
    }
    ShareGeom gm;
    gm.n_strips = (a.W + TX - 1) / TX;
    const int nb_max = (a.H + S - 1) / S;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
    }
    // segment length minimising rounds * (L + fixed cost); one work-group per CU (LDS-bound), see the strip kernel
    int best_L = ((nb_max + ROWS - 1) / ROWS) * ROWS;
    long best_cost = -1;
    for (int Lr = ROWS * 4; Lr <= nb_max + ROWS; Lr += ROWS) {
        const int segs_l = (nb_max + Lr - 1) / Lr;
        const long blocks = (long)gm.n_strips * S * segs_l;
        const long rounds = (blocks + n_cu - 1) / n_cu;
        const long cost = rounds * (Lr + 10);
        if (best_cost < 0 || cost <= best_cost) { best_cost = cost; best_L = Lr; }
    }
    gm.seg_rows = best_L;
    gm.n_segs = (nb_max + best_L - 1) / best_L;
    gm.n_groups = S * gm.n_segs;
    gm.kn = (float)(1.4426950408889634 / ((double)a.sigma_n + 1e-6));
    gm.kx = (float)(1.4426950408889634 / ((double)a.sigma_x + 1e-6));
    const int groups_pad = (gm.n_groups + 7) / 8 * 8;
    const int nblocks = groups_pad * gm.n_strips;
    gm.dbg = nullptr; gm.dbg_block = 0;
    static unsigned long long *dbg_buf = nullptr;
    const char *dbg_env = getenv("SVGF_SHARE_DBG");
    if (dbg_env) {
        if (!dbg_buf) (void)hipMalloc((void **)&dbg_buf, 16 * 16 * 8 * sizeof(unsigned long long));
        (void)hipMemsetAsync(dbg_buf, 0, 16 * 16 * 8 * sizeof(unsigned long long), s);
        gm.dbg = dbg_buf; gm.dbg_block = atoi(dbg_env);
    }
    hipLaunchKernelGGL((k_atrous_share<LOG2S>), dim3(nblocks), dim3(NT), lds, s, a, gm);
    if (dbg_env) {
        static int prints = 0;
        (void)hipStreamSynchronize(s);
        unsigned long long h[16 * 16 * 8];
        (void)hipMemcpy(h, dbg_buf, sizeof(h), hipMemcpyDeviceToHost);
        if (prints++ < 6) {
            const int nw = NT / 64, nlw = kLoaderThreads / 64;
            fprintf(stderr, "[share dbg] S=%d blocks=%d segs=%d seg_rows=%d lds=%zu waves=%d (last %d = loaders)\n", S, nblocks, gm.n_segs,
                    gm.seg_rows, lds, nw, nlw);
            const int show[4] = { 0, nw - nlw - 1, nw - nlw, nw - 1 };
            for (int si = 0; si < 4; si++) {
                const int w = show[si];
                for (int it = 0; it < 16 && h[(w * 16 + it) * 8]; it++) {
                    unsigned long long *t = &h[(w * 16 + it) * 8];
                    if (w >= nw - nlw)
                        fprintf(stderr, "  loader %2d it %2d: t0=%6llu work %6llu barA %5llu barB %5llu\n", w, it, t[0] - h[0], t[1] - t[0], t[2] - t[1], t[3] - t[2]);
                    else
                        fprintf(stderr, "  wave %2d it %2d: t0=%6llu G %5llu fwd %5llu barA %5llu bwd+out %5llu barB %5llu\n", w, it, t[0] - h[0],
                                t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4]);
                }
            }
        }
    }
    return hipGetLastError();
}

}  // namespace

bool atrous_share_supported(const AtrousArgs &a)
{
    if (a.step != 2 && a.step != 4 && a.step != 8) return false;
    if ((long long)a.W * a.H * 16 >= (1LL << 32)) return false;
    return true;
}

hipError_t launch_atrous_share(const AtrousArgs &a, hipStream_t s)
{
    switch (a.step) {
    case 2: return launch_share_cfg<1>(a, s);
    case 4: return launch_share_cfg<2>(a, s);
    case 8: return launch_share_cfg<3>(a, s);
    default: return hipErrorInvalidValue;
    }
}
