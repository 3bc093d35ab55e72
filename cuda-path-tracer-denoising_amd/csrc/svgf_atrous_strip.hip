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
// Wave specialisation.  Measured on MI355X (profiles/r01_strip_phase_timeline.log): with every wave doing
// load -> math -> store -> barrier in lockstep, the 25-tap math ran VALU-saturated but only ~45 % of the time; the
// rest was vector-memory issue, LDS staging and barrier skew that nothing overlapped.  The workgroup is therefore
// TX*ROWS compute threads + two loader groups (2 waves each) that take turns: a loader fetches the next ROWS lattice
// rows (and the two full-resolution neighbour rows the 3x3 variance pre-blur needs) while the compute waves evaluate taps, converts
// them to the LDS layout and publishes them at the single barrier that ends the iteration.  Compute waves issue no
// vector-memory loads at all.
//
// LDS layout: array-of-pixels, 48 B per pixel = three 16-B slots
//     A = {n.x, p.x, n.y, p.y}   B = {n.z, p.z, luminance, -}   C = {r, g, b, variance}
// read with ds_read_b128; the (normal, position) component pairs and the (r,g) / (b,variance) pairs sit in adjacent
// registers so the per-tap arithmetic is written directly as packed fp32 (v_pk_add/mul/fma_f32) with no shuffling.
// The 48-B pixel stride is conflict-free for b128: a 16-lane group covers byte offsets 48*l, i.e. 12*l dwords mod 64
// = 16 distinct 4-bank slots.  Out-of-image pixels are staged with luminance = +inf, which makes their weight
// exp2(-inf) = 0 with no per-tap bounds test (the reference skips those taps, :134).
//
// Arithmetic (fast path): the reference's three edge-stopping factors exp(-a)*min(1,exp(-b))*min(1,exp(-c)) with
// a,b,c >= 0 equal exp(-(a+b+c)); it is evaluated as one v_exp_f32 of a base-2 exponent with the filter weight h
// folded in.  Non-finite normals/positions make the reference's min(1, exp(NaN)) return 1 (fminf
// semantics); a workgroup that stages such a texel switches to the `CAREFUL` tap routine that reproduces this, so
// ordinary frames pay nothing for it.
#include "svgf_kernels.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace {

constexpr float kLog2e = 1.44269504088896340736f;
#ifndef SVGF_LOADER_GROUPS
#define SVGF_LOADER_GROUPS 2
#endif
#ifndef SVGF_LOADER_DIV
#define SVGF_LOADER_DIV 2
#endif
// Wave priorities (s_setprio).  The SIMD arbiter serves the oldest ready wave first, so of the two compute waves that
// share a SIMD the older one used to finish its iteration ~2000 cycles early and idle at the barrier while the younger
// ran alone, unable to hide its own latencies (profiles/r01_strip_phase_timeline_v3.log: taps 4300 vs 5650 cycles).
// With SVGF_PROGRESS_PRIO a compute wave starts every iteration at priority 3 and lowers itself as it completes tap
// rows: whichever wave is behind outranks the one that is ahead, both finish together, and the iteration shrinks from
// ~7250 to ~6300 cycles (profiles/r01_exp_progress_prio.log).  Loader waves sit at SVGF_LOADER_PRIO.
// Round 6 (A/B harness, 2048x1152, profiles/r06_ab_lane_prio.txt): 3 held through the first TWO tap rows, then 2, then 1 (was 3, 2, 1, 0),
// loaders at 1 (was 2): 51.0 -> 50.2 us per level, as the lane kernel's schedule of the same round.
#ifndef SVGF_PROGRESS_PRIO
#define SVGF_PROGRESS_PRIO 1
#endif
#ifndef SVGF_LOADER_PRIO
#define SVGF_LOADER_PRIO 1
#endif
// ROWS <= 2: SVGF_LOADER_GROUPS groups of TX / SVGF_LOADER_DIV threads take turns (issue / in flight / commit).
// ROWS == 3: 12 compute waves leave room for 4 loader waves (1024 threads): one group of TX threads that commits and
//            re-issues every iteration.
__host__ __device__ constexpr int loader_groups(int rows) { return rows >= 3 ? 1 : SVGF_LOADER_GROUPS; }
__host__ __device__ constexpr int loader_group(int tx, int rows) { return rows >= 3 ? tx : tx / SVGF_LOADER_DIV; }
__host__ __device__ constexpr int loader_threads(int tx, int rows) { return loader_groups(rows) * loader_group(tx, rows); }

struct StripGeom {
    int n_strips;   // strips of TX columns
    int n_segs;     // lattice-row segments per phase
    int seg_rows;   // lattice rows per segment
    int n_groups;   // S * n_segs
    float kn, kx;   // log2(e) / (sigma_n + 1e-6), log2(e) / (sigma_x + 1e-6)
    unsigned long long *dbg;   // tuning only (experiments build, svgf_exp_set("strip_dbg", <block>)): per-phase s_memtime stamps of one workgroup
    int dbg_block;
};

struct Px {   // one staged pixel in registers
    float4 cv;
    float nx, ny, nz, px, py, pz;
    int lds_off;   // byte offset of its slot in the LDS ring; bit 31 or bit 30 set = out-of-image pixel
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

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

struct Centre {          // per output pixel, loop invariant over the 25 taps
    v2f nx_px, ny_py, nz_pz;
    float lp, kl, kn, kx;
};
struct Acc {
    v2f rg;              // sum w*r, sum w*g
    v2f bv;              // sum w*b, sum w^2*variance
    v2f ww;              // sum w,   sum w^2
};

// One tap: 3 x ds_read_b128, then 9 packed + 5 scalar VALU, 2 v_sqrt + 1 v_exp.
//   neg_log2_h = -log2(h[k]) is folded into the exponent: h*2^-e == 2^-(e - log2 h).
template <bool CAREFUL>
__device__ __forceinline__ void tap(const char *lds_px, float neg_log2_h, const Centre &c, Acc &acc)
{
    const v4f A = *reinterpret_cast<const v4f *>(lds_px);
    const v4f B = *reinterpret_cast<const v4f *>(lds_px + 16);
    const v4f C = *reinterpret_cast<const v4f *>(lds_px + 32);
    const v2f d0 = A.xy - c.nx_px, d1 = A.zw - c.ny_py, d2 = B.xy - c.nz_pz;
    v2f s = d0 * d0;
    s = __builtin_elementwise_fma(d1, d1, s);
    s = __builtin_elementwise_fma(d2, d2, s);                  // (|dn|^2, |dp|^2)
    float dn = __builtin_amdgcn_sqrtf(s.x);
    float dx = __builtin_amdgcn_sqrtf(s.y);
    if (CAREFUL) {          // min(1, exp(-NaN)) == 1 in the reference: a NaN distance contributes nothing
        dn = fmaxf(dn, 0.0f);
        dx = fmaxf(dx, 0.0f);
    }
    float e = fmaf(fabsf(B.z - c.lp), c.kl, neg_log2_h);
    e = fmaf(dn, c.kn, e);
    e = fmaf(dx, c.kx, e);
    const float w = __builtin_amdgcn_exp2f(-e);
    v2f wv;
    wv.x = w;
    wv.y = w * w;
    acc.ww += wv;
    acc.rg = __builtin_elementwise_fma(C.xy, v2f{w, w}, acc.rg);
    acc.bv = __builtin_elementwise_fma(C.zw, wv, acc.bv);
}

// -log2 of the 5-tap binomial [1 4 6 4 1]/16 (log2(6/16) is the only inexact one)
__device__ __forceinline__ constexpr float neg_log2_binom(int i)
{
    return (i == 0) ? 1.4150374992788437f : ((i == 1 || i == -1) ? 2.0f : 4.0f);
}

// HASVAR = false: the level's filtered variance is not needed (last level, no colour-history copy): the two variance
// accumulators (sum w^2, sum w^2 var) and the w*w product drop out of every tap.
template <int LOG2S, int TX, int ROWS, bool HASVAR>
__global__ __launch_bounds__(TX * ROWS + loader_threads(TX, ROWS)) void k_atrous_strip(AtrousArgs a, StripGeom gm)
{
    constexpr int S = 1 << LOG2S;
    constexpr int RW = TX + 4 * S;          // staged pixels per lattice row
    constexpr int R = 4 + 2 * ROWS;         // ring slots: 4 + ROWS live, ROWS incoming
    constexpr int PXB = 48;                 // bytes per staged pixel
    constexpr int NC = TX * ROWS;           // compute threads
    constexpr int kLoaderGroups = loader_groups(ROWS);
    constexpr int kLoaderGroup = loader_group(TX, ROWS);
    constexpr int kLoaderThreads = loader_threads(TX, ROWS);
    constexpr int NT = NC + kLoaderThreads; // + the loader waves
    constexpr int BW = TX + 2;              // blur row: columns x0-1 .. x0+TX
    constexpr int RING_BYTES = R * RW * PXB;
    constexpr int BLUR_FLOATS = 2 * ROWS * 2 * BW;   // [iteration parity][row of iteration][y-1 | y+1][BW]

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *blur = reinterpret_cast<float *>(smem + RING_BYTES);
    int *nan_seen = reinterpret_cast<int *>(smem + RING_BYTES + BLUR_FLOATS * 4);

    // every kernel argument in one batch of s_loads at entry (see svgf_atrous_lane.hip: three dependent scalar-memory round
    // trips in front of the prologue's first global load otherwise)
    asm volatile("" :: "s"(a.src), "s"(a.dst), "s"(a.out_rgb), "s"(a.nrm), "s"(a.pos), "s"(a.gbuf), "s"(a.var), "s"(a.var_dst), "s"(a.W), "s"(a.H),
                 "s"(a.sigma_c), "s"(a.blur_variance), "s"(a.modulate), "s"(gm.n_strips), "s"(gm.n_segs), "s"(gm.seg_rows),
                 "s"(gm.n_groups), "s"(gm.kn), "s"(gm.kx));

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
    // where the pre-blur rows come from: the zero-margined 4-byte variance plane of the source ((W+2) x (H+2), written by
    // the producer next to its colour plane) when there is one — contiguous dwords — else the .w of the 16-byte colour
    // texels (4 useful bytes per 16 fetched, the 1.55-1.65x over-fetch of steps 16/32).  vbase points at pixel (0, 0).
    const bool vplane = (a.var != nullptr);
    const char *vbase = vplane ? reinterpret_cast<const char *>(a.var) + ((size_t)W + 3) * 4 : reinterpret_cast<const char *>(a.src) + 12;
    const unsigned vxs = vplane ? 4u : 16u, vys = vplane ? (unsigned)(W + 2) * 4u : (unsigned)W * 16u;

    if (tid == 0) *nan_seen = 0;
    __syncthreads();           // the flag is initialised before any wave's prologue can raise it
    // the one kernel argument only the compute waves use: fetch it now rather than behind the prologue barrier
    // (an s_load from the kernarg segment that misses the scalar cache costs several hundred cycles)
    float sigma_c = a.sigma_c;
    asm volatile("" : "+s"(sigma_c));

    // ring slot of lattice row br (>= b0-2).  R is 8 for ROWS = 2 (a mask); for other R the iteration loops keep the
    // slot of row bc-2 in `ring_base` (wave-uniform, advanced by ROWS per iteration) and wrap small offsets from it.
    int ring_base = 0, ring_bc = b0;
    auto slot_of = [&](int br) {
        if constexpr ((R & (R - 1)) == 0) return (br - (b0 - 2)) & (R - 1);
        else {
            int s = ring_base + (br - (ring_bc - 2));      // offset in [0, 3*ROWS+2) < 2R
            s -= (s >= R) ? R : 0;
            s -= (s >= R) ? R : 0;
            return s;
        }
    };
    auto slot_mod = [&](int br) { return (br - (b0 - 2)) % R; };
    auto ring_advance = [&]() { ring_bc += ROWS; ring_base += ROWS; ring_base -= (ring_base >= R) ? R : 0; };

    // ---- staging, split in two halves so that a batch of loads can stay in flight across a barrier ----
    // rows_load : global -> registers for lattice rows br_first .. br_first+nrows-1, pixels wi, wi+nw, ...
    // rows_store: registers -> LDS ring (layout conversion, luminance, non-finite detection)
    // Both are branch-free: coordinates are clamped into the image so every lane always loads valid memory, and an
    // out-of-image pixel is then encoded as {luminance = +inf, colour/variance = 0} (weight exp2(-inf) = 0; the zeroed
    // colour keeps 0 * x finite).  Its normal/position slots keep the clamped pixel's finite values.
    auto rows_load = [&](auto &px, int br_first, int nrows, int wi, int nw) {
        constexpr int M = sizeof(px) / sizeof(px[0]);
        const int total = nrows * RW;
#pragma unroll
        for (int m = 0; m < M; m++) {
            const int idx = min(wi + m * nw, total - 1);       // surplus lanes repeat the last pixel (same value, same slot)
            const int rr = idx / RW, xi = idx - rr * RW;
            const int br = br_first + rr;
            const int y = phase + (br << LOG2S);
            const int xs = x0 - 2 * S + xi;
            const bool ok = (br >= 0) && (y < H) && (xs >= 0) && (xs < W);
            px[m].lds_off = ((slot_mod(br) * RW + xi) * PXB) | (ok ? 0 : (int)0x80000000);
            const unsigned q = (unsigned)min(max(y, 0), H - 1) * (unsigned)W + (unsigned)min(max(xs, 0), W - 1);   // < 2^28
            // 32-bit byte offsets from the (wave-uniform) plane bases: global_load with SGPR base + VGPR offset
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
            const bool ok = (unsigned)px[m].lds_off < 0x40000000u;     // bits 31 / 30: row / column outside the image
            const float lum = lum_f64(px[m].cv.x, px[m].cv.y, px[m].cv.z);
            const float mag = fabsf(px[m].nx) + fabsf(px[m].ny) + fabsf(px[m].nz) + fabsf(px[m].px) + fabsf(px[m].py) + fabsf(px[m].pz);
            if (!(mag < inf)) *nan_seen = 1;                    // NaN or inf in a normal / position (rare)
            char *d = smem + (px[m].lds_off & 0x3fffffff);
            *reinterpret_cast<float4 *>(d) = make_float4(px[m].nx, px[m].px, px[m].ny, px[m].py);
            *reinterpret_cast<float4 *>(d + 16) = make_float4(px[m].nz, px[m].pz, ok ? lum : inf, 0.0f);
            *reinterpret_cast<float4 *>(d + 32) = ok ? px[m].cv : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    // variance of the full-resolution rows y-1 and y+1 of output rows bo_first .. +ROWS-1 (3x3 pre-blur, :102-118)
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
                if (y >= 0 && y < H && xs >= 0 && xs < W && bo_first + rr < b1) v[m] = *reinterpret_cast<const float *>(vbase + (unsigned)y * vys + (unsigned)xs * vxs);
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

    // Loader groups: kLoaderGroups groups of kLoaderGroup threads; group q owns the rows that iterations j with
    // j % kLoaderGroups == q newly need.  A group issues its loads at the start of iteration j-kLoaderGroups, keeps
    // them in flight across kLoaderGroups-1 barriers, converts and stores them during iteration j-1, and the barrier
    // ending j-1 publishes them: every global load has kLoaderGroups-1 whole iterations (~3.5 us each) to land.
    // Two groups (12 waves per workgroup = 3 per SIMD) beat three (14 waves): the level itself is as fast or faster,
    // and each SIMD keeps 128 VGPRs free, which is what lets the next frame's temporal pass (48 VGPRs, side stream)
    // become co-resident in useful numbers (profiles/r01_exp_loader_groups.log: +4..6 % whole-frame at 1080p and 4K).
    constexpr int ML = (ROWS * RW + kLoaderGroup - 1) / kLoaderGroup;
    constexpr int NBT = TX / kLoaderGroup;              // blur columns 0 .. TX-1 per thread and (row, side) combination
    constexpr int NBC = ROWS * 2;                       // combinations c = 2 * (row of the iteration) + (0: y-1 | 1: y+1)
    static_assert(TX % kLoaderGroup == 0 && NBC * 2 <= kLoaderGroup, "loader mapping");
    const bool is_loader = (tid >= NC);
    const int lgroup = is_loader ? (tid - NC) / kLoaderGroup : -1;
    const int llane = is_loader ? (tid - NC) % kLoaderGroup : 0;
    Px lpx[ML];
    float lbv[NBC * NBT + 1];
    // Everything about a loader thread's pixels that does not change from iteration to iteration is computed once:
    // per pixel an issue then costs two selects, two adds and two address scalings (the lattice row, its clamped
    // source row and its ring slot are wave-uniform and live in SGPRs); per blur value nothing at all (SGPR row base +
    // invariant VGPR offset).  The loader waves share the SIMDs with the VALU-bound compute waves, so their
    // instruction count matters as much as their latency.
    int l_xq[ML];       // clamped source column
    int l_lds[ML];      // xi * PXB, bit 30 set if the column lies outside the image
    int l_rr[ML];       // which of the ROWS new rows the pixel belongs to
    int b_voff[NBT];    // byte offset of {clamped column}.w inside a source row
    bool b_ok[NBT];
    // (filled in inside the prologue, between the issue of its global loads and their use)
    auto loader_invariants = [&]() {
        if (is_loader) {
#pragma unroll
            for (int m = 0; m < ML; m++) {
                const int idx = min(llane + m * kLoaderGroup, ROWS * RW - 1);   // surplus lanes repeat the last pixel
                const int rr = idx / RW, xi = idx - rr * RW;
                const int xs = x0 - 2 * S + xi;
                l_xq[m] = min(max(xs, 0), W - 1);
                l_lds[m] = (xi * PXB) | ((xs >= 0 && xs < W) ? 0 : 0x40000000);
                l_rr[m] = rr;
            }
#pragma unroll
            for (int t = 0; t < NBT; t++) {
                const int xs = x0 - 1 + llane + t * kLoaderGroup;
                b_ok[t] = (xs >= 0 && xs < W);
                b_voff[t] = (int)((unsigned)min(max(xs, 0), W - 1) * vxs);
            }
        }
    };
    // blur value number NBC*NBT: columns TX, TX+1 of every combination, one per lane (lanes 0 .. 2*NBC-1)
    auto blur_extra_coords = [&](int bcj, int &c, int &xi, bool &ok, unsigned &q) {
        c = llane >> 1;
        xi = TX + (llane & 1);
        const int rr = c >> 1;
        const int y = phase + ((bcj + rr) << LOG2S) + ((c & 1) ? 1 : -1);
        const int xs = x0 - 1 + xi;
        ok = (llane < 2 * NBC) && y >= 0 && y < H && xs >= 0 && xs < W && (bcj + rr < b1);
        q = (unsigned)min(max(y, 0), H - 1) * vys + (unsigned)min(max(xs, 0), W - 1) * vxs;      // byte offset from vbase
    };
    // new rows of iteration j: lattice rows b0 + j*ROWS + 2 .. +ROWS-1 (+ROWS); its output rows start at b0 + j*ROWS
    auto loader_issue = [&](int j) {
        const int bcj = b0 + j * ROWS;
        if (bcj < b1) {
            int rowq[ROWS], ldsrow[ROWS];
#pragma unroll
            for (int rr = 0; rr < ROWS; rr++) {     // wave-uniform
                const int br = bcj + 2 + rr;
                const int y = phase + (br << LOG2S);
                rowq[rr] = min(y, H - 1) * W;
                ldsrow[rr] = (slot_of(br) * RW * PXB) | (y < H ? 0 : (int)0x80000000);
            }
#pragma unroll
            for (int m = 0; m < ML; m++) {
                int rq = rowq[0], lr = ldsrow[0];
#pragma unroll
                for (int rr = 1; rr < ROWS; rr++) { rq = (l_rr[m] == rr) ? rowq[rr] : rq; lr = (l_rr[m] == rr) ? ldsrow[rr] : lr; }
                const unsigned q = (unsigned)(rq + l_xq[m]);                         // < 2^28
                lpx[m].lds_off = lr + l_lds[m];                                      // bit 31: row, bit 30: column outside
                lpx[m].cv = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(a.src) + q * 16u);
                const float *n = reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.nrm) + q * 12u);
                const float *p = reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.pos) + q * 12u);
                lpx[m].nx = n[0]; lpx[m].ny = n[1]; lpx[m].nz = n[2];
                lpx[m].px = p[0]; lpx[m].py = p[1]; lpx[m].pz = p[2];
            }
            if (a.blur_variance) {
#pragma unroll
                for (int c = 0; c < NBC; c++) {
                    const int y = phase + ((bcj + (c >> 1)) << LOG2S) + ((c & 1) ? 1 : -1);              // wave-uniform
                    const char *rowp = vbase + (size_t)min(max(y, 0), H - 1) * vys;
#pragma unroll
                    for (int t = 0; t < NBT; t++) lbv[c * NBT + t] = *reinterpret_cast<const float *>(rowp + b_voff[t]);
                }
                int c, xi; bool ok; unsigned q;
                blur_extra_coords(bcj, c, xi, ok, q);
                lbv[NBC * NBT] = *reinterpret_cast<const float *>(vbase + q);
            }
        }
    };
    auto loader_commit = [&](int j) {
        const int bcj = b0 + j * ROWS;
        if (bcj < b1) {
            rows_store(lpx);
            if (a.blur_variance) {
                float *bp = blur + (j & 1) * (ROWS * 2 * BW) + llane;
#pragma unroll
                for (int c = 0; c < NBC; c++) {
                    const int y = phase + ((bcj + (c >> 1)) << LOG2S) + ((c & 1) ? 1 : -1);
                    const bool row_ok = y >= 0 && y < H && (bcj + (c >> 1) < b1);                      // wave-uniform
#pragma unroll
                    for (int t = 0; t < NBT; t++)
                        bp[c * BW + t * kLoaderGroup] = (row_ok && b_ok[t]) ? lbv[c * NBT + t] : 0.0f;
                }
                int c, xi; bool ok; unsigned q;
                blur_extra_coords(bcj, c, xi, ok, q);
                if (llane < 2 * NBC) blur[(j & 1) * (ROWS * 2 * BW) + c * BW + xi] = ok ? lbv[NBC * NBT] : 0.0f;
            }
        }
    };

    // in-kernel timeline (tools, profiles/r01_strip_phase_timeline*.log): compiled in only with -DSVGF_STRIP_TIMELINE,
    // because each stamp is a branch that splits the iteration into basic blocks the scheduler cannot move loads across
    int dbg_it = 0;
    auto stamp = [&](int phase_id) {
#ifdef SVGF_STRIP_TIMELINE
        if (gm.dbg && bid == gm.dbg_block && (tid & 63) == 0 && dbg_it < 16)
            gm.dbg[((tid >> 6) * 16 + dbg_it) * 8 + phase_id] = __builtin_amdgcn_s_memtime();
#else
        (void)phase_id;
#endif
    };

    // ---- prologue: every thread helps stage rows b0-2 .. b0+ROWS+1 and the blur rows of iteration 0;
    //      loader group 1 also puts the loads of iteration 1 in flight ----
    {
        constexpr int M = ((4 + ROWS) * RW + NT - 1) / NT;
        Px px[M];
        stamp(7);
        rows_load(px, b0 - 2, 4 + ROWS, tid, NT);
        constexpr int MB = (ROWS * 2 * BW + NT - 1) / NT;
        float bv[MB];
        blur_load(bv, b0, tid, NT);
        stamp(1);
        loader_invariants();            // independent of the loads in flight
        rows_store(px);
        if (a.blur_variance) blur_store(bv, 0, tid, NT);
        stamp(4);
    }
    __syncthreads();

    if (is_loader) {
        // ================================ loader waves ================================
        // The loads of iterations 1 .. kLoaderGroups-1 are issued at the top of iteration 0, not in front of the
        // prologue barrier: carrying loaded registers into the loop made the compiler copy them behind an
        // s_waitcnt vmcnt(0) that held the whole workgroup at the first barrier for a global-load latency.  Group 1
        // therefore issues and commits iteration 1's rows within iteration 0 (the loads land well inside its ~3 us).
        __builtin_amdgcn_s_setprio(SVGF_LOADER_PRIO);
        int it = 0;
        for (int bc = b0; bc < b1; bc += ROWS, it++, dbg_it++) {
            stamp(0);
            if (kLoaderGroups == 1) {                                   // single group: commit, then re-issue, every iteration
                if (it == 0) loader_issue(1);
                loader_commit(it + 1);
                loader_issue(it + 2);
            } else {
                if (it == 0 && lgroup >= 1) loader_issue(lgroup);
                if ((it + 1) % kLoaderGroups == lgroup) loader_commit(it + 1);      // issued kLoaderGroups-1 iterations ago
                else if (it % kLoaderGroups == lgroup) loader_issue(it + kLoaderGroups);
            }
            stamp(5);
            __syncthreads();
            stamp(6);
            ring_advance();
        }
        return;
    }

    // ================================ compute waves ================================
    const int r = tid / TX;                 // which of the ROWS rows of an iteration this thread serves
    const int tx = tid - r * TX;
    const int x = x0 + tx;
    // An SGPR or literal source makes a VOP3 occupy the VALU as long as a packed instruction does (tools/ubench6.hip): the two
    // slopes and the five distinct values of -log2 h of the 24 taps live in VGPRs (profiles/r03_ab_lane_operands.log)
    float kn = gm.kn, kx = gm.kx;
    asm volatile("" : "+v"(kn), "+v"(kx));
    float hc_a = 1.4150374992788437f + 2.0f, hc_b = 4.0f, hc_c = 1.4150374992788437f + 4.0f, hc_d = 6.0f, hc_e = 8.0f;
    asm volatile("" : "+v"(hc_a), "+v"(hc_b), "+v"(hc_c), "+v"(hc_d), "+v"(hc_e));
    auto nlh = [&](int io, int j) -> float {      // -log2 h of tap (io, j), never the centre
        const int ai = io < 0 ? -io : io, aj = j < 0 ? -j : j;
        const int lo = ai < aj ? ai : aj, hi = ai < aj ? aj : ai;
        return (lo == 0 && hi == 1) ? hc_a : (lo == 1 && hi == 1) ? hc_b : (lo == 0 && hi == 2) ? hc_c : (lo == 1 && hi == 2) ? hc_d : hc_e;
    };
    const char *colbase = smem + (size_t)tx * PXB;

    int it = 0;
    for (int bc = b0; bc < b1; bc += ROWS, it++, dbg_it++) {
        stamp(0);
#if SVGF_PROGRESS_PRIO
        __builtin_amdgcn_s_setprio(3);
#endif
        const int bo = bc + r;
        if (bo < b1 && x < W) {
            const int y = phase + (bo << LOG2S);
            const char *rowc = colbase + (size_t)slot_of(bo) * RW * PXB + (size_t)(2 * S) * PXB;
            // the centre pixel and its 3x3 variance neighbourhood are read in one basic block (one LDS round trip)
            const float4 A = *reinterpret_cast<const float4 *>(rowc);
            const float4 B = *reinterpret_cast<const float4 *>(rowc + 16);
            const float4 C = *reinterpret_cast<const float4 *>(rowc + 32);
            const bool careful = (*nan_seen != 0);
            const float *bl = blur + (it & 1) * (ROWS * 2 * BW) + r * (2 * BW) + tx;    // column x-1 of row y-1
            const float m0 = bl[0], m1 = bl[1], m2 = bl[2];
            const float p0 = bl[BW], p1 = bl[BW + 1], p2 = bl[BW + 2];
            const float c0v = *reinterpret_cast<const float *>(rowc - PXB + 44);
            const float c2v = *reinterpret_cast<const float *>(rowc + PXB + 44);

            // centre variance: 3x3 gaussian with out-of-image taps dropped and renormalised (:102-118).  Evaluated
            // unconditionally and selected (with blur_variance off the blur rows hold stale LDS, never used).
            float var;
            {
                const float wr_m = (y - 1 >= 0) ? 0.25f : 0.0f, wr_p = (y + 1 < H) ? 0.25f : 0.0f;
                const float wc_l = (x - 1 >= 0) ? 0.25f : 0.0f, wc_r = (x + 1 < W) ? 0.25f : 0.0f;
                const float col_l = wr_m * m0 + 0.5f * c0v + wr_p * p0;
                const float col_c = wr_m * m1 + 0.5f * C.w + wr_p * p1;
                const float col_r = wr_m * m2 + 0.5f * c2v + wr_p * p2;
                const float sum = wc_l * col_l + 0.5f * col_c + wc_r * col_r;
                const float sumw = (wr_m + 0.5f + wr_p) * (wc_l + 0.5f + wc_r);
                const float blurred = sum * __builtin_amdgcn_rcpf(sumw);
                var = a.blur_variance ? blurred : C.w;
            }
            var = fmaxf(var, 0.0f);

            Centre c;
            c.nx_px = v2f{A.x, A.y}; c.ny_py = v2f{A.z, A.w}; c.nz_pz = v2f{B.x, B.y};
            c.lp = B.z;
            c.kl = kLog2e * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(var) * sigma_c + 1e-6f);
            c.kn = kn; c.kx = kx;
            stamp(2);

            Acc acc;
            if (!careful) {
                // The centre tap has weight exactly h = 9/64 (all three distances are 0): no arithmetic needed.
                constexpr float w0 = 0.140625f;
                acc.ww = v2f{w0, w0 * w0};
                acc.rg = v2f{w0 * C.x, w0 * C.y};
                acc.bv = v2f{w0 * C.z, (w0 * w0) * C.w};
                // A tap row (5 taps) is evaluated in stages so that the five taps' dependency chains interleave and the
                // transcendentals issue back to back (measured: clustered v_sqrt/v_exp overlap with packed VALU, spread
                // ones do not — profiles/r01_ubench2_trans_overlap.log).  The geometry slots {A,B} of row j+1 are
                // prefetched while row j is evaluated; the colour slot C of row j is fetched behind its geometry math.
                v4f Ac[5], Bc[5], An[5], Bn[5];
                {
                    const char *rowp = colbase + (size_t)slot_of(bo - 2) * RW * PXB;
#pragma unroll
                    for (int i = 0; i < 5; i++) {
                        Ac[i] = *reinterpret_cast<const v4f *>(rowp + i * S * PXB);
                        Bc[i] = *reinterpret_cast<const v4f *>(rowp + i * S * PXB + 16);
                    }
                }
#pragma unroll
                for (int j = -2; j <= 2; j++) {
                    const char *rowp = colbase + (size_t)slot_of(bo + j) * RW * PXB;
                    if (j < 2) {
                        const char *rown = colbase + (size_t)slot_of(bo + j + 1) * RW * PXB;
#pragma unroll
                        for (int i = 0; i < 5; i++) {
                            An[i] = *reinterpret_cast<const v4f *>(rown + i * S * PXB);
                            Bn[i] = *reinterpret_cast<const v4f *>(rown + i * S * PXB + 16);
                        }
                    }
                    v2f s2[5];
                    float dl[5];
#pragma unroll
                    for (int i = 0; i < 5; i++) {
                        if (i == 2 && j == 0) continue;
                        const v2f d0 = Ac[i].xy - c.nx_px, d1 = Ac[i].zw - c.ny_py, d2 = Bc[i].xy - c.nz_pz;
                        v2f t = d0 * d0;
                        t = __builtin_elementwise_fma(d1, d1, t);
                        s2[i] = __builtin_elementwise_fma(d2, d2, t);
                        dl[i] = Bc[i].z - c.lp;
                    }
                    v4f Cc[5];
#pragma unroll
                    for (int i = 0; i < 5; i++)
                        if (!(i == 2 && j == 0)) Cc[i] = *reinterpret_cast<const v4f *>(rowp + i * S * PXB + 32);
                    __builtin_amdgcn_sched_barrier(0x100);      // only LDS reads may move across
                    float dn[5], dx[5];
#pragma unroll
                    for (int i = 0; i < 5; i++) {
                        if (i == 2 && j == 0) continue;
                        dn[i] = __builtin_amdgcn_sqrtf(s2[i].x);
                        dx[i] = __builtin_amdgcn_sqrtf(s2[i].y);
                    }
                    __builtin_amdgcn_sched_barrier(0x100);
                    float e[5];
#pragma unroll
                    for (int i = 0; i < 5; i++) {
                        if (i == 2 && j == 0) continue;
                        float t = fmaf(fabsf(dl[i]), c.kl, nlh(i - 2, j));
                        t = fmaf(dn[i], c.kn, t);
                        e[i] = fmaf(dx[i], c.kx, t);
                    }
                    __builtin_amdgcn_sched_barrier(0x100);
                    float w[5];
#pragma unroll
                    for (int i = 0; i < 5; i++)
                        if (!(i == 2 && j == 0)) w[i] = __builtin_amdgcn_exp2f(-e[i]);
                    __builtin_amdgcn_sched_barrier(0x100);
#pragma unroll
                    for (int i = 0; i < 5; i++) {
                        if (i == 2 && j == 0) continue;
                        if (HASVAR) {
                            v2f wv;
                            wv.x = w[i];
                            wv.y = w[i] * w[i];
                            acc.ww += wv;
                            acc.rg = __builtin_elementwise_fma(Cc[i].xy, v2f{w[i], w[i]}, acc.rg);
                            acc.bv = __builtin_elementwise_fma(Cc[i].zw, wv, acc.bv);
                        } else {
                            acc.ww.x += w[i];
                            acc.rg = __builtin_elementwise_fma(Cc[i].xy, v2f{w[i], w[i]}, acc.rg);
                            acc.bv.x = fmaf(Cc[i].z, w[i], acc.bv.x);
                        }
                    }
                    if (j < 2) {
#pragma unroll
                        for (int i = 0; i < 5; i++) { Ac[i] = An[i]; Bc[i] = Bn[i]; }
                    }
                    // the wave that is ahead lowers its own priority, so the two compute waves of a SIMD finish together
#if SVGF_PROGRESS_PRIO
                    if (j == -1) __builtin_amdgcn_s_setprio(3);
                    if (j == 0) __builtin_amdgcn_s_setprio(2);
                    if (j == 1) __builtin_amdgcn_s_setprio(1);
#endif
                }
            } else {
                acc.rg = v2f{0.0f, 0.0f}; acc.bv = v2f{0.0f, 0.0f}; acc.ww = v2f{0.0f, 0.0f};
                for (int j = -2; j <= 2; j++) {
                    const char *rowp = colbase + (size_t)slot_of(bo + j) * RW * PXB;
#pragma unroll
                    for (int i = -2; i <= 2; i++)
                        tap<true>(rowp + (i + 2) * S * PXB, neg_log2_binom(i) + neg_log2_binom(j), c, acc);
                }
            }
            stamp(3);
            const float c0 = acc.rg.x, c1 = acc.rg.y, c2 = acc.bv.x, vsum = acc.bv.y, wsum = acc.ww.x, w2sum = acc.ww.y;

            float o0, o1, o2, ov;
            if (wsum > 1e-5f) {                                     // NaN -> false -> pass-through (:159-164)
                const float rw = __builtin_amdgcn_rcpf(wsum);
                o0 = c0 * rw; o1 = c1 * rw; o2 = c2 * rw;
                ov = HASVAR ? vsum * __builtin_amdgcn_rcpf(w2sum) : 0.0f;
            } else {
                o0 = C.x; o1 = C.y; o2 = C.z; ov = C.w;
            }
            const unsigned p = (unsigned)y * (unsigned)W + (unsigned)x;
            if (a.modulate) svgf_modulate(a, p, o0, o1, o2);       // last level: * albedo * ialbedo (:166-168)
            if (a.dst) a.dst[p] = make_float4(o0, o1, o2, ov);
            if (a.var_dst) a.var_dst[(unsigned)(y + 1) * (unsigned)(W + 2) + (unsigned)(x + 1)] = ov;
            if (a.out_rgb) { float *o = a.out_rgb + 3u * p; o[0] = o0; o[1] = o1; o[2] = o2; }
        }
        stamp(5);
        __syncthreads();
        stamp(6);
        ring_advance();
    }
}

// Segment length: every (strip, phase, segment) is one workgroup and `capacity` of them run at a time, so the grid runs in
// rounds of equal-length workgroups.  Returns the minimum of rounds * (L + fixed cost) — in lattice rows; the fixed cost being
// the 4 halo rows + the exposed prologue latency — and the segment length L that reaches it.
long strip_segment_search(int n_strips, int S, int nb_max, int ROWS, int capacity, int *best_L_out)
{
    int best_L = nb_max;
    long best_cost = -1;
    static const int fixed_rows = SVGF_TUNE("strip_fixed_rows", 8);   // tuning only
    for (int L = ROWS * 4; L <= nb_max + ROWS; L++) {        // L need not be a multiple of ROWS: the last iteration idles rows
        const int segs_l = (nb_max + L - 1) / L;
        // (phase, segment) groups are dealt round-robin to the 8 XCDs (blockIdx % 8), all strips of a group to the same
        // XCD: the busiest XCD, with ceil(groups / 8) groups, sets the number of rounds
        const long blocks_xcd = (long)n_strips * ((S * segs_l + 7) / 8);
        const long cap_xcd = capacity / 8 > 0 ? capacity / 8 : 1;        // (a device with fewer than 8 CUs: one workgroup per "XCD" at a time)
        const long rounds = (blocks_xcd + cap_xcd - 1) / cap_xcd;
        const long cost = rounds * ((L + ROWS - 1) / ROWS * ROWS + fixed_rows);
        if (best_cost < 0 || cost <= best_cost) { best_cost = cost; best_L = L; }   // ties: fewer, longer workgroups
    }
    *best_L_out = best_L;
    return best_cost;
}

template <int LOG2S, int TX, int ROWS, bool HASVAR>
hipError_t launch_cfg(const AtrousArgs &a, hipStream_t s)
{
    constexpr int S = 1 << LOG2S, RW = TX + 4 * S, R = 4 + 2 * ROWS, BW = TX + 2;
    const size_t lds = (size_t)R * RW * 48 + (size_t)2 * ROWS * 2 * BW * 4 + 16;
    static SvgfLaunchCache cache;
    int dev_id = 0;
    if (hipError_t e = cache.init(reinterpret_cast<const void *>(&k_atrous_strip<LOG2S, TX, ROWS, HASVAR>), (int)lds, &dev_id); e != hipSuccess) return e;
    const int n_cu = cache.n_cu[dev_id];
    StripGeom gm;
    gm.n_strips = (a.W + TX - 1) / TX;
    const int nb_max = (a.H + S - 1) / S;
    // Segment length: every (strip, phase, segment) is one workgroup, and LDS admits `bpc` workgroups per CU, so the
    // grid runs in ceil(blocks / (CUs * bpc)) rounds of equal-length workgroups.  Pick the segment length L that
    // minimises rounds * (L + fixed cost), the fixed cost being the 4 halo rows + the exposed prologue latency.
    int bpc = (int)((160 * 1024) / lds);
    constexpr int kLoaderThreads = loader_threads(TX, ROWS);
    if (bpc > 2048 / (TX * ROWS + kLoaderThreads)) bpc = 2048 / (TX * ROWS + kLoaderThreads);
    if (bpc < 1) bpc = 1;
    const int capacity = n_cu * bpc;
    int best_L = nb_max;
    (void)strip_segment_search(gm.n_strips, S, nb_max, ROWS, capacity, &best_L);
    if (const int v = SVGF_TUNE("strip_segrows", 0); v > 0) best_L = v;
    gm.seg_rows = best_L;
    gm.n_segs = (nb_max + best_L - 1) / best_L;
    gm.n_groups = S * gm.n_segs;
    gm.dbg = nullptr; gm.dbg_block = 0;
    static unsigned long long *dbg_buf = nullptr;
    const int dbg_block = SVGF_TUNE("strip_dbg", -1);
    const bool dbg_env = dbg_block >= 0;
    if (dbg_env) {
        if (!dbg_buf) (void)hipMalloc((void **)&dbg_buf, 16 * 16 * 8 * sizeof(unsigned long long));
        (void)hipMemsetAsync(dbg_buf, 0, 16 * 16 * 8 * sizeof(unsigned long long), s);
        gm.dbg = dbg_buf; gm.dbg_block = dbg_block;
    }
    gm.kn = (float)(1.4426950408889634 / ((double)a.sigma_n + 1e-6));
    gm.kx = (float)(1.4426950408889634 / ((double)a.sigma_x + 1e-6));
    const int groups_pad = (gm.n_groups + 7) / 8 * 8;
    const int nblocks = groups_pad * gm.n_strips;
    SVGF_LAUNCH_KERNEL((k_atrous_strip<LOG2S, TX, ROWS, HASVAR>), dim3(nblocks), dim3(TX * ROWS + kLoaderThreads), lds, s, a, gm);
    if (dbg_env) {
        static int prints = 0;
        (void)hipStreamSynchronize(s);
        unsigned long long h[16 * 16 * 8];
        (void)hipMemcpy(h, dbg_buf, sizeof(h), hipMemcpyDeviceToHost);
        static int skip = SVGF_TUNE("strip_dbg_skip", 0);    // warm launches only
        if (skip > 0) skip--;
        else if (prints++ < 10) {
            const int nw = (TX * ROWS + kLoaderThreads) / 64;
            fprintf(stderr, "[strip dbg] S=%d TX=%d ROWS=%d blocks=%d segs=%d seg_rows=%d lds=%zu waves=%d (last 6 = loaders)\n", S, TX,
                    ROWS, nblocks, gm.n_segs, gm.seg_rows, lds, nw);
            const int nlw = kLoaderThreads / 64;
            const int show[4] = { 0, nw - nlw - 1, nw - nlw, nw - 1 };
            for (int si = 0; si < 4; si++) {
                const int w = show[si];
                if (h[(w * 16) * 8 + 7])
                    fprintf(stderr, "  wave %2d prologue: entry..loads issued %5llu, ..stored %6llu, ..first iteration %6llu\n", w,
                            h[(w * 16) * 8 + 1] - h[(w * 16) * 8 + 7], h[(w * 16) * 8 + 4] - h[(w * 16) * 8 + 7],
                            h[(w * 16) * 8 + 0] - h[(w * 16) * 8 + 7]);
                for (int it = 0; it < 16 && h[(w * 16 + it) * 8]; it++) {
                    unsigned long long *t = &h[(w * 16 + it) * 8];
                    if (w >= nw - nlw)
                        fprintf(stderr, "  loader  it %2d: t0=%6llu stage %6llu barrier %5llu\n", it, t[0] - h[0], t[5] - t[0], t[6] - t[5]);
                    else
                        fprintf(stderr, "  wave %2d it %2d: t0=%6llu centre %5llu taps %6llu out %5llu barrier %5llu\n", w, it, t[0] - h[0],
                                t[2] - t[0], t[3] - t[2], t[5] - t[3], t[6] - t[5]);
                }
            }
        }
    }
    return hipGetLastError();
}

// default configuration per dilation; svgf_exp_set("strip_tx" / "strip_rows") override it in the experiments build
void pick(int log2s, int W, int &tx, int &rows)
{
    // 256 columns x 2 rows per workgroup everywhere (profiles/r01_exp_tx_rows.log):
    //  * 128-column strips (two workgroups per CU) run a lone S <= 8 level 3-4 % faster at 1080p (equal at 3840), but
    //    the whole frame gets slower (6.46 vs 6.94 Gpix/s) once the next frame's temporal pass shares the GPU with
    //    levels 2-3; at S >= 16 the 4S halo columns make narrow strips lose outright;
    //  * ROWS = 3 (12 compute waves) is correct but not faster: two compute waves already saturate a SIMD's VALU.
    (void)W;
    tx = 256; rows = 2;
    if (const int v = SVGF_TUNE("strip_tx", 0); v == 128 || v == 256) tx = v;
    if (const int v = SVGF_TUNE("strip_rows", 0); v >= 1 && v <= 3) rows = v;
    if (rows == 3 && tx != 256) rows = 2;
    // LDS budget: ring + blur rows <= 160 KiB
    const int S = 1 << log2s;
    while ((size_t)(4 + 2 * rows) * (tx + 4 * S) * 48 + (size_t)2 * rows * 2 * (tx + 2) * 4 + 16 > 160 * 1024 && rows > 1) rows--;
}

}  // namespace

bool atrous_strip_supported(const AtrousArgs &a)
{
    if (a.step < 1 || a.step > 32 || (a.step & (a.step - 1))) return false;      // step 1: SvgfParams::paper_steps
    if ((long long)a.W * a.H * 16 >= (1LL << 32)) return false;   // 32-bit element offsets in the kernel
    return true;
}

// Estimated duration of a level on this kernel: the launch geometry's cost in lattice rows x 1.16 us (1920x1080: one round of
// 34 + 8 rows = 48.8 us; profiles/r03_exp_widths*.log: within 5 % at eight other sizes).  Used by the automatic kernel choice.
double atrous_strip_estimate_us(const AtrousArgs &a, int n_cu)
{
    int log2s = 0;
    while ((1 << log2s) < a.step) log2s++;
    int tx, rows;
    pick(log2s, a.W, tx, rows);
    const int S = 1 << log2s;
    const size_t lds = (size_t)(4 + 2 * rows) * (tx + 4 * S) * 48 + (size_t)2 * rows * 2 * (tx + 2) * 4 + 16;
    int bpc = (int)((160 * 1024) / lds);
    const int threads = tx * rows + loader_threads(tx, rows);
    if (bpc > 2048 / threads) bpc = 2048 / threads;
    if (bpc < 1) bpc = 1;
    int L = 0;
    return 1.162 * (double)strip_segment_search((a.W + tx - 1) / tx, S, (a.H + S - 1) / S, rows, n_cu * bpc, &L);
}

#define STRIP_CASE(L, T, Rr) if (log2s == L && tx == T && rows == Rr) return a.dst ? launch_cfg<L, T, Rr, true>(a, s) : launch_cfg<L, T, Rr, false>(a, s);

hipError_t launch_atrous_strip(const AtrousArgs &a, hipStream_t s)
{
    int log2s = 0;
    while ((1 << log2s) < a.step) log2s++;
    int tx, rows;
    pick(log2s, a.W, tx, rows);
    // product build: 256 columns x 2 rows per workgroup at every step (what pick() returns without tuning; 155.7 KB of LDS at step 32)
    STRIP_CASE(0, 256, 2) STRIP_CASE(1, 256, 2) STRIP_CASE(2, 256, 2) STRIP_CASE(3, 256, 2) STRIP_CASE(4, 256, 2) STRIP_CASE(5, 256, 2)
#ifdef SVGF_BUILD_EXPERIMENTS
    STRIP_CASE(1, 256, 3) STRIP_CASE(2, 256, 3) STRIP_CASE(3, 256, 3)
    STRIP_CASE(0, 128, 1) STRIP_CASE(0, 128, 2) STRIP_CASE(0, 256, 1) STRIP_CASE(0, 256, 3)
    STRIP_CASE(1, 128, 1) STRIP_CASE(1, 128, 2) STRIP_CASE(1, 256, 1)
    STRIP_CASE(2, 128, 1) STRIP_CASE(2, 128, 2) STRIP_CASE(2, 256, 1)
    STRIP_CASE(3, 128, 1) STRIP_CASE(3, 128, 2) STRIP_CASE(3, 256, 1)
    STRIP_CASE(4, 128, 1) STRIP_CASE(4, 128, 2) STRIP_CASE(4, 256, 1)
    STRIP_CASE(5, 128, 1) STRIP_CASE(5, 128, 2) STRIP_CASE(5, 256, 1)
#endif
    return hipErrorInvalidValue;
}
