// svgf_atrous_lane_impl.h — a-trous level with the symmetric part of every tap evaluated ONCE, for S = 2 .. 32 (gfx950).
// The kernel template; instantiated by svgf_atrous_lane.hip (plain levels) and svgf_atrous_fused.hip (temporal pass fused
// into the first level's loader waves).
//
// Same result as svgf_atrous_strip.hip (one level of reference ATrousFilter, src/denoise.cu:77-170, snapshot variance)
// and the same staging machinery (LDS ring filled by loader waves, one barrier per iteration).  What changes is who
// evaluates what.  The geometric part of the edge-stopping exponent,
//     t(p,q) = -log2 h + kn |n_p - n_q| + kx |x_p - x_q|,
// is symmetric, and every pair (p,q) is a tap of p and a tap of q; the strip kernel pays 6 packed + 2 scalar VALU and
// 2 v_sqrt for it twice.  Here a LANE owns one lattice column and marches down its rows, one row per iteration:
//     * at row b it evaluates t for its 12 FORWARD partners only (rows b+1, b+2 and the two right-hand neighbours in
//       row b) and uses them for its own forward taps;
//     * it keeps the terms of rows b+1 / b+2 in registers for one / two iterations, and its 12 BACKWARD taps take
//       their t from the lane that owns the partner column: partner (x + i, b - j) published it j iterations ago as
//       its forward term (-i, +j), so it arrives by a DPP wave shift of |i| lanes (v_mov_b32_dpp wave_shr/wave_shl:1,
//       which the compiler folds into the consuming VALU op where it can).
// 12 geometry evaluations per output pixel instead of 24, no extra LDS traffic, no extra barrier.
// The same ownership gives the backward and left-hand taps their LUMINANCE without LDS: it is the centre luminance the owning
// lane held one or two iterations ago (or holds now), four lane shifts away.
//
// For the partner of column x +- i to be lane +- i, the 64 lanes of a wave hold 64 consecutive columns of ONE x-phase
// (pixel columns x, x+S, x+2S, ...): the LDS ring stores every row phase-major ([phase][lattice column], 48-byte
// records, so tap addresses are base + i*48 and stay bank-conflict-free), the loaders scatter into that layout, and
// the outermost two lanes on either side of a wave are halo lanes (their shifted-in terms are garbage, they store
// nothing): a wave produces 60 columns, a workgroup 8 waves x 60 x ... = 480 contiguous pixel columns.
//     S = 2: 4 waves per x-phase     S = 4: 2 waves per x-phase     S = 8: 1 wave per x-phase
// S = 16, 32: a workgroup holds 8 ADJACENT x-phases only ("chunks", template parameter LOG2P), one wave each;
// see the comment in front of the kernel.
//
// LDS: ring 6 rows (b-2 .. b+2 live, b+3 incoming) x (480 + 4S) pixels x 48 B = 140.5 .. 147.5 KB, + pre-blur rows.
#pragma once
#include "svgf_kernels.h"
#include "svgf_temporal.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>

#ifndef SVGF_LANE_SPLIT_PROLOGUE
#define SVGF_LANE_SPLIT_PROLOGUE 1
#endif
#ifndef SVGF_LANE_PRIO
// progress-based wave priorities: stage order A (start, after 1/4, after 2/3 of the row), stage order B (start, 1/3, 2/3, 3/4).
// Order B keeps the higher priority for longer: its waves are the ones that reach the barrier last (profiles/r03_ab_lane_prio.log:
// 3,2,1 / 3,2,1,0 -> 3,2,1 / 3,3,2,1 is -1.5 % per level; no priorities at all +6 %).  Round 6 (profiles/r06_ab_lane_prio.txt, 22
// schedules on the A/B harness): order B holds 3 until 3/4 of its row and ends at 2, order A ends at 2 instead of 1, the loader
// waves drop from 2 to 1: -1.5 % per 1080p level (40.2 -> 39.6 us), -0.5 % at 4K and 1280x720; lowering order A's start or holding its
// 3 for longer LOSES 2-5 %
#define SVGF_LANE_PRIO 3, 2, 2, 3, 3, 3, 2
#endif
#ifndef SVGF_LANE_LOADER_PRIO
#define SVGF_LANE_LOADER_PRIO 1
#endif

namespace {

constexpr float kLog2e = 1.44269504088896340736f;
constexpr int NWC = 8;                       // compute waves
constexpr int LOUT = 60;                     // output lanes per wave (2 halo lanes either side)
constexpr int TXO = NWC * LOUT;              // 480 output pixels per workgroup and iteration, for every S (one row of 480, or two rows of 240)
constexpr int kLoaderGroups = 2, kLoaderGroup = 128, kLoaderThreads = kLoaderGroups * kLoaderGroup;
constexpr int NC = NWC * 64, NT = NC + kLoaderThreads;
constexpr int PXB = 48;
constexpr int R = 6;                         // ring slots: rows b-2 .. b+2 live, b+3 incoming
constexpr int BW = TXO + 2;                  // pre-blur row: pixel columns x0-1 .. x0+TXO (one y-phase per workgroup only)

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

struct LaneGeom {
    int n_strips, n_segs, seg_rows, n_groups;
    float kn, kx;
    unsigned long long *dbg;   // tuning only (-DSVGF_LANE_TIMELINE + svgf_exp_set("lane_dbg", <block>)): s_memtime stamps of one workgroup
    int dbg_block;
};

struct Px {
    float4 cv;
    float nx, ny, nz, px, py, pz;
    int lds_off;   // byte offset of the record in the ring; bit 31 or bit 30 set = out-of-image pixel; bit 29 (FUSED == 3 only): the
                   // workgroup owns the pixel (it writes the pixel's split G-buffer planes)
    int gid;       // FUSED == 3 only: the texel's geomId, and the pixel index the planes are written at
    unsigned q;
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

// value of v in lane (self + K), K in -2 .. 2 (wave-wide; lanes shifted in from outside the wave read 0)
template <int K>
__device__ __forceinline__ float lane_from(float v)
{
    int x = __builtin_bit_cast(int, v);
    if constexpr (K == 1 || K == 2) x = __builtin_amdgcn_update_dpp(0, x, 0x130, 0xf, 0xf, true);      // wave_shl:1
    if constexpr (K == 2) x = __builtin_amdgcn_update_dpp(0, x, 0x130, 0xf, 0xf, true);
    if constexpr (K == -1 || K == -2) x = __builtin_amdgcn_update_dpp(0, x, 0x138, 0xf, 0xf, true);    // wave_shr:1
    if constexpr (K == -2) x = __builtin_amdgcn_update_dpp(0, x, 0x138, 0xf, 0xf, true);
    return __builtin_bit_cast(float, x);
}

// LOG2P < LOG2S (steps 16, 32): a workgroup's 8 waves hold only P = 8 of the S x-phases — "chunks" of 8 adjacent pixels (one
// 128-byte line of colour) every S pixels — and 60 lattice columns of each: at 1920 pixels a phase has 120 (60) lattice columns,
// so two (one) lattice strips x two (four) phase groups = four workgroups tile a row exactly.  (P = 4 with two waves per phase
// also tiles 1920 at step 16 and was the first version: same speed at 1080p, 6-9 % slower at 3000-3840 columns, where its
// 64-byte pieces of colour rows cost more than they carry; profiles/r03_exp_s16_p8.log.)  Everything downstream of the staging is unchanged (a wave is still 64 consecutive
// lattice columns of one x-phase); what changes is (a) the pixel a staged column stands for, (b) the 3x3 variance pre-blur:
// a centre's x-1 / x+1 neighbours lie in phases the workgroup may not stage, so for P < S the LOADER threads compute the
// blurred variance of every pixel of the incoming output row from the producer's 4-byte variance plane (three rows x
// {one dwordx4 + two dwords} per four pixels) and the compute lanes read one float instead of eight.
//
// LOG2Y = 1 (round 4; S = 2 only): a workgroup holds BOTH y-phases of a 240-column strip instead of one y-phase of a 480-column
// strip — waves 0-3 march down the even rows, waves 4-7 (their SIMD partners) down the odd rows, the ring holds 6 lattice rows
// of each.  Same lanes, same LDS, same halo rows; what it buys is that rows y-1 and y+1 of every centre are ring rows, so the
// 3x3 variance pre-blur reads its eight neighbours from the ring and no pre-blur rows are loaded at all — which is what lets
// the temporal pass be fused into this level (FUSED): the loader waves PRODUCE the rows they stage (svgf_temporal.h, the
// arithmetic of k_temporal) instead of loading them, and the accumulated variance of rows y +- 1 exists nowhere but in the ring.
struct LaneNoTemporal {};
// FUSED: the temporal pass's arguments plus, for every plane of the context the fused stages address PER LANE (history taps, the
// stores of owned / not-owned pixels), its byte offset from the context's one allocation (TemporalArgs::arena): a per-lane choice
// between two planes is then a 32-bit select on offsets over ONE scalar base, not a 64-bit select between two pointers.  The
// read offsets carry the "- 2 elements" of t_window() already.
struct LaneFused : TemporalArgs {
    unsigned o_cv_hist, o_mom_hist, o_hlen, o_gid_prev, o_nrm_prev;                              // history taps (element - 2)
    unsigned o_hlen_upd, o_mom_acc, o_cv_acc, o_nrm_cur, o_pos_cur, o_gid_cur, o_dump;          // stores (cv_acc = dump when unwanted)
};
//
// REUSE (round 4): the geometric part of a pair's exponent, g(p, q) = kn |n_p - n_q| + kx |x_p - x_q|, does not depend on the
// level, and the pairs at lattice offsets (+2, 0), (-2, +2), (0, +2), (+2, +2) of step S ARE the pairs at offsets (+1, 0), (-1, +1),
// (0, +1), (+1, +1) of step 2S: four of the twelve forward terms of every level after the first were evaluated by the level
// before.  Bit 1 (OUT): a lane stores those four terms of its centre (minus the level's -log2 h constants) as one float4 in
// a.tout[p]; bit 0 (IN): it reads them from a.tin[p] one iteration ahead and skips four of its twelve geometry evaluations
// (24 packed + 8 plain VALU, 8 v_sqrt and 4 ds_read_b128 per pixel) for one 16-byte load and four additions.
// MEASURED (profiles/r04_ab_reuse_*.log): parity-green and a LOSS — 265 -> 275 us per 1080p frame at best (coalesced plane),
// 292 with the pixel-major plane, 956 -> 1154 us at 4K.  The compute waves of this kernel issue no vector-memory LOAD and one
// 16-byte store per lane and row; one more store costs a level 2.5 us (the eight waves store in lock step: the output stage
// grows from ~300 to ~600 of 5 300 ticks), one load 1-4 us (its first use waits, in order, for the previous row's stores),
// against ~3 us the four evaluations are worth, and at 4K the two extra planes (266 MB per level) meet an HBM that is no longer
// half idle.  Experiments build only (svgf_exp_set("reuse", 1) before svgf_create); kept because it is correct and small.
// FUSED: 0 no, 1 fused temporal pass reading the AoS G-buffer, 2 fused temporal pass reading the producer's planes,
//        3 fused PREPARE pass (non-temporal mode, reference EstimateVariance :320-329 + :370 and the G-buffer split): the loaders
//          stage colour from the 1-spp input, variance = 10 and normal / position from the AoS texels, and write the split planes
//          of the pixels the workgroup owns; standard geometry (one y-phase per workgroup), the pre-blur rows are the constant 10
//        4 fused G-BUFFER SPLIT only (temporal frames): colour + variance come from the plane the temporal pass wrote, normal /
//          position / geomId from the AoS texels; the loaders write the split planes the temporal pass then does not write
template <int LOG2S, bool HASVAR, int LOG2P = LOG2S, int LOG2Y = 0, int FUSED = 0, int REUSE = 0>
__global__ __launch_bounds__(NT) void k_atrous_lane(AtrousArgs a, LaneGeom gm, std::conditional_t<FUSED != 0, LaneFused, LaneNoTemporal> ta)
{
    constexpr bool REUSE_IN = (REUSE & 1) != 0, REUSE_OUT = (REUSE & 2) != 0;
    constexpr int S = 1 << LOG2S;
    constexpr int P = 1 << LOG2P;                // x-phases held by one workgroup
    constexpr int YP = 1 << LOG2Y;               // y-phases held by one workgroup
    constexpr bool CHUNKED = (P < S);
    static_assert(YP == 1 || (YP == 2 && S == 2 && P == S), "two y-phases per workgroup: step 2 only");
    constexpr bool TFUSED = (FUSED == 1 || FUSED == 2), PFUSED = (FUSED == 3 || FUSED == 4);      // PFUSED: the loaders read texels and write the split planes
    static_assert(!TFUSED || YP == 2, "the fused temporal pass needs the pre-blur rows in the ring");
#ifndef SVGF_BUILD_EXPERIMENTS
    static_assert((FUSED == 0 || FUSED == 3) && YP == 1 && REUSE == 0,
                  "product build: the plain level and the fused prepare pass only; FUSED = 1 / 2 / 4, LOG2Y = 1 and REUSE are parked "
                  "experiments (-DSVGF_BUILD_EXPERIMENTS, libsvgf_hip_exp.so)");
#endif
    static_assert(!PFUSED || (YP == 1 && P == S), "the fused prepare pass uses the plain geometry of steps <= 8");
    constexpr int WPP = NWC / (P * YP);          // waves per x-phase (and y-phase)
    constexpr int TXW = TXO / YP;                // output pixel columns per workgroup
    constexpr int M = LOUT * WPP + 4;            // lattice columns per phase in the ring (2 halo either side)
    // phase stride in records, padded so that (a) consecutive phases do not start on the same bank for the readers and (b) the
    // loaders' 16-byte stores, which are served eight consecutive lanes = pixels at a time with banks counted mod 32, do not
    // collide: with S = 4 those eight lanes are phases 0..3 of two lattice columns, and 124 * 12 = 16 (mod 32) put phases 0 / 2
    // and 1 / 3 on the same banks (a third of the kernel's remaining conflict cycles); 126 * 12 = 8 (mod 32) spreads all eight
    constexpr int MP = (P == 4) ? M + 2 : ((M * 12 % 64 == 0) ? M + 1 : M);
    constexpr int RW = P * M;                    // staged pixel columns (= TXO + 4S when P == S)
    constexpr int ROWB = P * MP * PXB;           // bytes per ring row (one lattice row of one y-phase)
    constexpr int BM = (BW + S - 1) / S;         // pre-blur lattice columns per phase
    constexpr int RING_BYTES = R * YP * ROWB;    // ring row of (lattice row slot s, y-phase yp): s * YP + yp
    constexpr int BLUR_ROW = S * BM;             // floats per pre-blur row (phase-major)
    constexpr int BLUR_BUF = (YP > 1) ? 0 : (CHUNKED ? P * M : 2 * BLUR_ROW);      // [y-1 | y+1], or the blurred variance [phase][column]; none with both y-phases in the ring
    constexpr int TXL = LOUT * WPP;              // lattice columns a workgroup outputs per phase
    static_assert(CHUNKED || RW == TXW + 4 * S, "layout");
    // chunked x-phases: each loader thread of a group blurs ONE unit of four staged columns per row (vblur_load / vblur_store)
    static_assert(!CHUNKED || RW / 4 <= kLoaderGroup, "the loader-side variance blur covers a row with one unit per loader thread");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *blur = reinterpret_cast<float *>(smem + RING_BYTES);           // [iteration parity][y-1 | y+1][phase][BM]
    int *nan_seen = reinterpret_cast<int *>(smem + RING_BYTES + 2 * BLUR_BUF * 4);

    // Every kernel argument is fetched by ONE batch of s_loads at entry.  Left to itself the compiler fetches them where
    // control flow first needs them — LaneGeom for the work-item decode, W/H behind the first early return, the plane
    // pointers behind the second — three dependent scalar-memory round trips (several hundred cycles each, cold) in
    // front of the first global load of the prologue.
    asm volatile("" :: "s"(a.src), "s"(a.dst), "s"(a.out_rgb), "s"(a.nrm), "s"(a.pos), "s"(a.gbuf), "s"(a.var), "s"(a.var_dst), "s"(a.W), "s"(a.H),
                 "s"(a.sigma_c), "s"(a.blur_variance), "s"(a.modulate), "s"(gm.n_strips), "s"(gm.n_segs), "s"(gm.seg_rows),
                 "s"(gm.n_groups), "s"(gm.kn), "s"(gm.kx));

    // ---- work item: (strip, y-phase, segment) as in the strip kernel ----
    const int bid = blockIdx.x;
    const int xcd = bid & 7, kk = bid >> 3;
    const int g = xcd + 8 * (kk / gm.n_strips);
    const int strip = kk % gm.n_strips;
    if (g >= gm.n_groups) return;
    // `phase`: first y-phase of the workgroup (its y-phases are phase .. phase + YP - 1; image row of lattice row br of
    // y-phase yp: phase + yp + br * S)
    const int phase = (g / gm.n_segs) * YP, seg = g % gm.n_segs;
    const int W = a.W, H = a.H;
    if (phase >= H) return;
    const int nb = (H - phase + S - 1) >> LOG2S;
    const int b0 = seg * gm.seg_rows;
    const int b1 = min(b0 + gm.seg_rows, nb);
    if (b0 >= b1) return;
    const int x0 = strip * TXW;                  // P == S: first output column of the strip
    // staged column xi = (lattice column c = xi / P, phase ph = xi % P) stands for image column (lbase + c) * S + pbase + ph
    const int lbase = (CHUNKED ? (strip / (S / P)) : strip) * TXL - 2;
    const int pbase = CHUNKED ? (strip % (S / P)) * P : 0;
    auto xs_of = [&](int xi) {
        if constexpr (!CHUNKED) return x0 - 2 * S + xi;
        else return ((xi >> LOG2P) + lbase) * S + pbase + (xi & (P - 1));
    };
    const int tid = threadIdx.x;
    // where the pre-blur rows come from: the zero-margined 4-byte variance plane of the source ((W+2) x (H+2), written by
    // the producer next to its colour plane) when there is one — contiguous dwords — else the .w of the 16-byte colour
    // texels (4 useful bytes per 16 fetched).  vbase points at pixel (0, 0).
    const bool vplane = (a.var != nullptr);
    const char *vbase = vplane ? reinterpret_cast<const char *>(a.var) + ((size_t)W + 3) * 4 : reinterpret_cast<const char *>(a.src) + 12;
    const unsigned vxs = vplane ? 4u : 16u, vys = vplane ? (unsigned)(W + 2) * 4u : (unsigned)W * 16u;
    if (tid == 0) *nan_seen = 0;
    __syncthreads();           // the flag is initialised before any wave's prologue can raise it
    float sigma_c = a.sigma_c;
    asm volatile("" : "+s"(sigma_c));
    int dbg_it = 0;
    auto stamp = [&](int id) {
#ifdef SVGF_LANE_TIMELINE
        if (gm.dbg && bid == gm.dbg_block && (tid & 63) == 0 && dbg_it < 16)
            gm.dbg[((tid >> 6) * 16 + dbg_it) * 8 + id] = __builtin_amdgcn_s_memtime();
#else
        (void)id; (void)dbg_it;
#endif
    };

    // prologue marks of the timeline: slot 7 of iteration 0 = kernel entry, of iteration 1 = prologue barrier passed
    auto stamp_at = [&](int it_slot) {
#ifdef SVGF_LANE_TIMELINE
        if (gm.dbg && bid == gm.dbg_block && (tid & 63) == 0)
            gm.dbg[((tid >> 6) * 16 + it_slot) * 8 + 7] = __builtin_amdgcn_s_memtime();
#else
        (void)it_slot;
#endif
    };
    stamp_at(0);

    // ring slot of lattice row br: `ring_base` is the slot of row ring_b - 2 (wave-uniform, advanced once per iteration)
    int ring_base = 0, ring_b = b0;
    auto slot_of = [&](int br) {
        int s = ring_base + (br - (ring_b - 2));      // in [0, 2R)
        s -= (s >= R) ? R : 0;
        return s;
    };
    auto slot_mod = [&](int br) { return (br - (b0 - 2)) % R; };
    auto ring_advance = [&]() { ring_b += 1; ring_base += 1; ring_base -= (ring_base >= R) ? R : 0; };
    // record of staged pixel column xi (0 .. RW-1) inside a ring row: phase-major
    auto rec_of = [&](int xi) { return ((xi & (P - 1)) * MP + (xi >> LOG2P)) * PXB; };
    // element of pre-blur pixel column xb (0 .. BW-1) inside a pre-blur row
    auto bel_of = [&](int xb) { return (xb & (S - 1)) * BM + (xb >> LOG2S); };

    // ---------------- staging (global -> registers -> LDS ring), branch-free, coordinates clamped ----------------
    // one pixel's loads.  PFUSED: the level's input does not exist as planes yet — colour comes from the 1-spp image, the variance
    // is the constant of the non-temporal mode (:327), normal / position / geomId from the boundary's 52-byte texel
    auto stage_load = [&](Px &lp, unsigned q) {
        if constexpr (PFUSED) {
            const float *g = reinterpret_cast<const float *>(reinterpret_cast<const char *>(ta.gbuf) + q * 52u);
            if constexpr (FUSED == 3) {
                const float *c3 = reinterpret_cast<const float *>(reinterpret_cast<const char *>(ta.in_rgb) + q * 12u);
                lp.cv = make_float4(c3[0], c3[1], c3[2], 10.0f);
            } else {
                lp.cv = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(a.src) + q * 16u);
            }
            lp.nx = g[0]; lp.ny = g[1]; lp.nz = g[2];
            lp.px = g[3]; lp.py = g[4]; lp.pz = g[5];
            lp.gid = __float_as_int(g[12]);
            lp.q = q;
        } else {
            lp.cv = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(a.src) + q * 16u);
            const float *n = reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.nrm) + q * 12u);
            const float *p = reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.pos) + q * 12u);
            lp.nx = n[0]; lp.ny = n[1]; lp.nz = n[2];
            lp.px = p[0]; lp.py = p[1]; lp.pz = p[2];
        }
    };
    auto rows_load = [&](auto &px, int br_first, int nrows, int wi, int nw) {
        constexpr int N = sizeof(px) / sizeof(px[0]);
        const int total = nrows * YP * RW;
#pragma unroll
        for (int m = 0; m < N; m++) {
            const int idx = min(wi + m * nw, total - 1);
            const int rr = idx / RW, xi = idx - rr * RW;
            const int br = br_first + rr / YP, yp = rr % YP;
            const int y = phase + yp + (br << LOG2S);
            const int xs = xs_of(xi);
            const bool ok = (br >= 0) && (y < H) && (xs >= 0) && (xs < W);
            px[m].lds_off = ((slot_mod(br) * YP + yp) * ROWB + rec_of(xi)) | (ok ? 0 : (int)0x80000000);
            if constexpr (PFUSED) {      // owned: one of this workgroup's output pixels (the thread that stages index `idx` exists once)
                const bool owned = ok && (wi + m * nw < total) && (br >= b0) && (br < b1) && (xi >= 2 * S) && (xi < 2 * S + TXO);
                px[m].lds_off |= owned ? 0x20000000 : 0;
            }
            const unsigned q = (unsigned)min(max(y, 0), H - 1) * (unsigned)W + (unsigned)min(max(xs, 0), W - 1);
            stage_load(px[m], q);
        }
    };
    auto rows_store = [&](const auto &px) {
        constexpr int N = sizeof(px) / sizeof(px[0]);
        const float inf = __builtin_huge_valf();
#pragma unroll
        for (int m = 0; m < N; m++) {
            const bool ok = PFUSED ? ((px[m].lds_off & (int)0xC0000000) == 0) : ((unsigned)px[m].lds_off < 0x40000000u);
            const float lum = lum_f64(px[m].cv.x, px[m].cv.y, px[m].cv.z);
            const float mag = fabsf(px[m].nx) + fabsf(px[m].ny) + fabsf(px[m].nz) + fabsf(px[m].px) + fabsf(px[m].py) + fabsf(px[m].pz);
            if (!(mag < inf)) *nan_seen = 1;
            char *d = smem + (px[m].lds_off & (PFUSED ? 0x1fffffff : 0x3fffffff));
            *reinterpret_cast<float4 *>(d) = make_float4(px[m].nx, px[m].px, px[m].ny, px[m].py);
            *reinterpret_cast<float4 *>(d + 16) = make_float4(px[m].nz, px[m].pz, ok ? lum : inf, 0.0f);
            *reinterpret_cast<float4 *>(d + 32) = ok ? px[m].cv : make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (PFUSED) {
                // the planes the rest of the frame (levels >= 2) and the next frame (previous normals / geomIds) read: what the
                // prepare kernel would have written (launch_prepare), by the one workgroup that owns the pixel
                if (px[m].lds_off & 0x20000000) {
                    const unsigned q = px[m].q;
                    float *n = reinterpret_cast<float *>(reinterpret_cast<char *>(ta.nrm_cur) + q * 12u);
                    float *p = reinterpret_cast<float *>(reinterpret_cast<char *>(ta.pos_cur) + q * 12u);
                    n[0] = px[m].nx; n[1] = px[m].ny; n[2] = px[m].nz;
                    p[0] = px[m].px; p[1] = px[m].py; p[2] = px[m].pz;
                    ta.gid_cur[q] = px[m].gid;
                    if constexpr (FUSED == 3) { if (ta.cv_acc) ta.cv_acc[q] = px[m].cv; }       // only when something besides this level reads the plane
                }
            }
        }
    };
    // pre-blur rows y-1, y+1 of output row bo (3x3 variance blur, :102-118); element e = d * BW + xb
    auto blur_load = [&](auto &v, int bo, int wi, int nw) {
        constexpr int N = sizeof(v) / sizeof(v[0]);
#pragma unroll
        for (int m = 0; m < N; m++) {
            const int e = wi + m * nw;
            v[m] = 0.0f;
            if (a.blur_variance && e < 2 * BW) {
                const int d = e / BW, xb = e - d * BW;
                const int y = phase + (bo << LOG2S) + (d ? 1 : -1);
                const int xs = x0 - 1 + xb;
                if (y >= 0 && y < H && xs >= 0 && xs < W && bo < b1) {
                    if constexpr (FUSED == 3) v[m] = 10.0f;     // the variance of every pixel in the non-temporal mode (:327)
                    else v[m] = *reinterpret_cast<const float *>(vbase + (unsigned)y * vys + (unsigned)xs * vxs);
                }
            }
        }
    };
    auto blur_store = [&](const auto &v, int parity, int wi, int nw) {
        constexpr int N = sizeof(v) / sizeof(v[0]);
#pragma unroll
        for (int m = 0; m < N; m++) {
            const int e = wi + m * nw;
            if (e < 2 * BW) {
                const int d = e / BW, xb = e - d * BW;
                blur[parity * BLUR_BUF + d * BLUR_ROW + bel_of(xb)] = v[m];
            }
        }
    };

    // CHUNKED: the blurred variance (3x3 gaussian, out-of-image taps dropped and renormalised, :102-118) of the pixels of output
    // row bo, four consecutive staged columns (one lattice column, phases ph0 .. ph0+3) per thread: from the zero-margined
    // variance plane rows y-1, y, y+1 it needs pixels x .. x+3 (one dwordx4) and x-1, x+4 (two dwords).  Same expression as
    // the compute lanes evaluate for P == S.
    struct VB { float4 r[3]; float l[3], e[3]; };
    auto vblur_load = [&](VB &v, int bo, int unit) {
        const int xi0 = min(unit, RW / 4 - 1) * 4;
        const int xq = min(max(xs_of(xi0), 0), W - 1);
        const int y = phase + (bo << LOG2S);
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const char *rowp = vbase + (long)min(max(y + d - 1, -1), H) * (long)vys + (long)xq * 4;
            v.r[d] = *reinterpret_cast<const float4 *>(rowp);
            v.l[d] = *reinterpret_cast<const float *>(rowp - 4);
            v.e[d] = *reinterpret_cast<const float *>(rowp + 16);
        }
    };
    auto vblur_store = [&](const VB &v, int bo, int parity, int unit) {
        if (unit >= RW / 4) return;
        const int xi0 = unit * 4;
        const int c = xi0 >> LOG2P, ph0 = xi0 & (P - 1);
        const int x = xs_of(xi0), y = phase + (bo << LOG2S);
        const float wr_m = (y - 1 >= 0) ? 0.25f : 0.0f, wr_p = (y + 1 < H) ? 0.25f : 0.0f;
        const float rm[6] = { v.l[0], v.r[0].x, v.r[0].y, v.r[0].z, v.r[0].w, v.e[0] };
        const float rc[6] = { v.l[1], v.r[1].x, v.r[1].y, v.r[1].z, v.r[1].w, v.e[1] };
        const float rp[6] = { v.l[2], v.r[2].x, v.r[2].y, v.r[2].z, v.r[2].w, v.e[2] };
        float *bb = blur + parity * BLUR_BUF + ph0 * M + c;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int xk = x + k;
            const float wc_l = (xk - 1 >= 0) ? 0.25f : 0.0f, wc_r = (xk + 1 < W) ? 0.25f : 0.0f;
            const float col_l = wr_m * rm[k] + 0.5f * rc[k] + wr_p * rp[k];
            const float col_c = wr_m * rm[k + 1] + 0.5f * rc[k + 1] + wr_p * rp[k + 1];
            const float col_r = wr_m * rm[k + 2] + 0.5f * rc[k + 2] + wr_p * rp[k + 2];
            const float sum = wc_l * col_l + 0.5f * col_c + wc_r * col_r;
            const float sumw = (wr_m + 0.5f + wr_p) * (wc_l + 0.5f + wc_r);
            bb[k * M] = sum * __builtin_amdgcn_rcpf(sumw);
        }
    };

    // ---------------- loader threads: per-thread invariants ----------------
    constexpr int ML = (RW + kLoaderGroup - 1) / kLoaderGroup;
    constexpr int MBL = (2 * BW + kLoaderGroup - 1) / kLoaderGroup;
    const bool is_loader = (tid >= NC);
    const int lgroup = is_loader ? (tid - NC) / kLoaderGroup : -1;
    const int llane = is_loader ? (tid - NC) % kLoaderGroup : 0;
    Px lpx[ML * YP];
    float lbv[MBL];
    VB lvb;
    int l_xq[ML], l_lds[ML];
    bool l_own[ML];                   // PFUSED: the staged column is one of the workgroup's output columns, inside the image, and this thread's own
    int b_voff[MBL], b_lds[MBL];      // b_voff < 0: element does not exist or its column is outside the image
    bool b_d[MBL];
    // (filled in inside the prologue, between the issue of its global loads and their use)
    auto loader_invariants = [&]() {
        if (is_loader) {
#pragma unroll
            for (int m = 0; m < ML; m++) {
                const int xi = min(llane + m * kLoaderGroup, RW - 1);
                const int xs = xs_of(xi);
                l_xq[m] = min(max(xs, 0), W - 1);
                l_lds[m] = rec_of(xi) | ((xs >= 0 && xs < W) ? 0 : 0x40000000);
                l_own[m] = (llane + m * kLoaderGroup < RW) && xs >= 0 && xs < W && xi >= 2 * S && xi < 2 * S + TXO;
            }
            if constexpr (!CHUNKED && YP == 1) {
#pragma unroll
                for (int m = 0; m < MBL; m++) {
                    const int e = llane + m * kLoaderGroup;
                    const int d = e / BW, xb = e - d * BW;
                    const int xs = x0 - 1 + xb;
                    b_d[m] = (d != 0);
                    b_lds[m] = (e < 2 * BW) ? (d * BLUR_ROW + bel_of(xb)) : -1;
                    b_voff[m] = (e < 2 * BW && xs >= 0 && xs < W) ? (int)((unsigned)xs * vxs) : -1;
                }
            }
        }
    };
    // iteration j outputs row b0 + j and newly needs lattice row b0 + j + 2
    auto loader_issue = [&](int j) {
        const int bo = b0 + j;
        if (bo < b1) {
            const int br = bo + 2;
#pragma unroll
            for (int yp = 0; yp < YP; yp++) {
                const int y = phase + yp + (br << LOG2S);
                const int rowq = min(y, H - 1) * W;
                const int ldsrow = ((slot_of(br) * YP + yp) * ROWB) | (y < H ? 0 : (int)0x80000000);
#pragma unroll
                for (int m = 0; m < ML; m++) {
                    const unsigned q = (unsigned)(rowq + l_xq[m]);
                    Px &lp = lpx[yp * ML + m];
                    lp.lds_off = ldsrow + l_lds[m];
                    if constexpr (PFUSED) lp.lds_off |= (br < b1 && y < H && l_own[m]) ? 0x20000000 : 0;
                    stage_load(lp, q);
                }
            }
            if constexpr (YP > 1) {
                // both y-phases are ring rows: the pre-blur reads its neighbours from the ring, nothing else to stage
            } else if constexpr (CHUNKED) {
                if (a.blur_variance) vblur_load(lvb, bo, llane);
            } else if (a.blur_variance) {
                const int ym = phase + (bo << LOG2S) - 1, yp = ym + 2;
                const unsigned rm = (unsigned)min(max(ym, 0), H - 1) * vys, rp = (unsigned)min(max(yp, 0), H - 1) * vys;
#pragma unroll
                for (int m = 0; m < MBL; m++)
                    if constexpr (FUSED == 3) lbv[m] = 10.0f;
                    else lbv[m] = *reinterpret_cast<const float *>(vbase + (b_d[m] ? rp : rm) + (unsigned)max(b_voff[m], 0));
            }
        }
    };
    auto loader_commit = [&](int j) {
        const int bo = b0 + j;
        if (bo < b1) {
            rows_store(lpx);
            if constexpr (YP > 1) {
            } else if constexpr (CHUNKED) {
                if (a.blur_variance) vblur_store(lvb, bo, j & 1, llane);
            } else if (a.blur_variance) {
                const int ym = phase + (bo << LOG2S) - 1, yp = ym + 2;
                const bool okm = ym >= 0 && ym < H, okp = yp >= 0 && yp < H;
                float *bb = blur + (j & 1) * BLUR_BUF;
#pragma unroll
                for (int m = 0; m < MBL; m++)
                    if (b_lds[m] >= 0) bb[b_lds[m]] = ((b_d[m] ? okp : okm) && b_voff[m] >= 0) ? lbv[m] : 0.0f;
            }
        }
    };

#ifdef SVGF_BUILD_EXPERIMENTS
#include "svgf_atrous_lane_tfused.inc.h"      // FUSED = 1 / 2: prologue + loader loop of the fused temporal pass (parked)
#endif
    if constexpr (!TFUSED) {
#if SVGF_LANE_SPLIT_PROLOGUE
    // ---------------- prologue in two steps.  All 256 workgroups start at once and each wants five ring rows, a burst that
    // takes 3-4 us to arrive.  The first warm-up row only needs rows b0-2 .. b0: the compute waves stage those (and the
    // pre-blur rows of iteration 0) and pass barrier A as soon as THEIR loads have landed; the loader waves fetch rows
    // b0+1, b0+2 at the same time, compute their invariants while the loads fly, pass barrier A without waiting for them
    // and publish the two rows at barrier B, which the compute waves reach after the first warm-up row.  The two paths
    // are separate straight-line code (a join in between makes the compiler park loaded registers behind s_waitcnt 0). --
    if (is_loader) {
        constexpr int N2 = (2 * YP * RW + kLoaderThreads - 1) / kLoaderThreads;
        Px px2[N2];
        rows_load(px2, b0 + 1, 2, tid - NC, kLoaderThreads);
        stamp_at(2);
        loader_invariants();
        __syncthreads();                // A
        stamp_at(1);
        rows_store(px2);
        __syncthreads();                // B
    } else {
        constexpr int N = (3 * YP * RW + NC - 1) / NC;
        Px px[N];
        rows_load(px, b0 - 2, 3, tid, NC);
        constexpr int NB = (2 * BW + NC - 1) / NC;
        float bv[NB];
        VB vb0;
        if constexpr (YP > 1) { }
        else if constexpr (CHUNKED) { if (a.blur_variance) vblur_load(vb0, b0, tid); }
        else blur_load(bv, b0, tid, NC);
        stamp_at(2);
        rows_store(px);
        if constexpr (YP > 1) { }
        else if constexpr (CHUNKED) { if (a.blur_variance) vblur_store(vb0, b0, 0, tid); }
        else if (a.blur_variance) blur_store(bv, 0, tid, NC);
        __syncthreads();                // A
        stamp_at(1);
    }
#else
    // ---------------- prologue: every thread helps stage rows b0-2 .. b0+2 and the pre-blur rows of iteration 0 ----------
    {
        constexpr int N = (5 * YP * RW + NT - 1) / NT;
        Px px[N];
        rows_load(px, b0 - 2, 5, tid, NT);
        constexpr int NB = (2 * BW + NT - 1) / NT;
        float bv[NB];
        VB vb0;
        if constexpr (YP > 1) { }
        else if constexpr (CHUNKED) { if (a.blur_variance) vblur_load(vb0, b0, tid); }
        else blur_load(bv, b0, tid, NT);
        loader_invariants();
        rows_store(px);
        if constexpr (YP > 1) { }
        else if constexpr (CHUNKED) { if (a.blur_variance) vblur_store(vb0, b0, 0, tid); }
        else if (a.blur_variance) blur_store(bv, 0, tid, NT);
    }
    __syncthreads();
    stamp_at(1);
#endif

    if (is_loader) {
        // ================================ loader waves (as in the strip kernel) ================================
        __builtin_amdgcn_s_setprio(SVGF_LANE_LOADER_PRIO);
        int it = 0;
        for (int bo = b0; bo < b1; bo++, it++, dbg_it++) {
            stamp(0);
            if (it == 0 && lgroup >= 1) loader_issue(lgroup);
            if ((it + 1) % kLoaderGroups == lgroup) loader_commit(it + 1);
            else if (it % kLoaderGroups == lgroup) loader_issue(it + kLoaderGroups);
            stamp(5);
            __syncthreads();
            stamp(6);
            ring_advance();
        }
        return;
    }
    }   // !TFUSED

    // ================================ compute waves ================================
    const int lane = tid & 63, wv = tid >> 6;
    const bool flip = (wv >= NWC / 2);              // stage order of this wave, see body()
    const int ypw = wv / (NWC / YP);                // y-phase (within the workgroup) of this wave
    const int yphase = phase + ypw;                 // image row of this wave's lattice row br: yphase + br * S
    const int xph = (wv % (NWC / YP)) / WPP;        // x-phase of this wave
    const int mcol = (wv % WPP) * LOUT + lane;      // lattice column inside the phase, 0 .. M-1
    const int xi = xph + P * mcol;                  // staged pixel column
    const int x = xs_of(xi);                        // image column
    const bool out_lane = (lane >= 2) && (lane < 2 + LOUT) && (x < W);
    // An SGPR or literal source makes a VOP3 occupy the VALU as long as a packed instruction does (tools/ubench6.hip: 3.0
    // cycles per SIMD against 1.7 with three waves, 4.5 against 2.5 with two): the two slopes and the five distinct values of
    // -log2 h of the forward / own-row taps live in VGPRs (profiles/r03_ab_lane_operands.log: -1 % per level).
    float kn = gm.kn, kx = gm.kx;
    asm volatile("" : "+v"(kn), "+v"(kx));
#ifndef SVGF_LANE_NO_VCONST
    float hc_a = 1.4150374992788437f + 2.0f, hc_b = 4.0f, hc_c = 1.4150374992788437f + 4.0f, hc_d = 6.0f, hc_e = 8.0f;
    asm volatile("" : "+v"(hc_a), "+v"(hc_b), "+v"(hc_c), "+v"(hc_d), "+v"(hc_e));
    // -log2 h of tap (io, j) for j = 0, 1, 2 and |io| <= 2 (never the centre)
    auto nlh = [&](int io, int j) -> float {
        const int ai = io < 0 ? -io : io;
        const int code = (ai == 0 ? 0 : (ai == 1 ? 1 : 2)) + (j == 0 ? 0 : (j == 1 ? 1 : 2));      // 1.415 -> 0, 2 -> 1, 4 -> 2; sums 1 .. 4
        return (ai == 0 && j == 1) || (ai == 1 && j == 0) ? hc_a : (code == 2 && ai != 0 && j != 0) ? hc_b
             : ((ai == 0 && j == 2) || (ai == 2 && j == 0)) ? hc_c : (code == 3) ? hc_d : hc_e;
    };
#else
    auto nlh = [&](int io, int j) -> float { return neg_log2_binom(io) + neg_log2_binom(j); };
#endif
    // base of the lane's tap window: record of column mcol-2, so that tap i = 0..4 (offset i-2) sits at +i*PXB and every
    // address is base + non-negative immediate (the ds_read offset field is unsigned)
    const char *colbase = smem + ypw * ROWB + (xph * MP + mcol - 2) * PXB;     // + ring_row(br): slot_of(br) * YP * ROWB
    constexpr int RSTR = YP * ROWB;                 // distance between consecutive lattice-row slots of one y-phase
    // neighbours x-1, x+1 of the centre in its own ring row (other x-phases), and the 3x3 pre-blur elements
    const int off_l = rec_of(max(xi - 1, 0)) + 44, off_r = rec_of(min(xi + 1, RW - 1)) + 44;
    const int xb = min(max(xi - 2 * S + 1, 1), BW - 2);
    const int be_l = bel_of(xb - 1), be_c = bel_of(xb), be_r = bel_of(xb + 1);

    // geometry evaluation of one partner: (|dn|^2, |dx|^2) -> t
    // (the centre is passed NEGATED: q + (-c) keeps the three differences on v_pk_add_f32; with q - c hipcc splits them)
    auto geo = [&](const v4f &Aq, const v4f &Bq, const v2f &nc0, const v2f &nc1, const v2f &nc2) {
        const v2f d0 = Aq.xy + nc0, d1 = Aq.zw + nc1, d2 = Bq.xy + nc2;
        v2f t = d0 * d0;
        t = __builtin_elementwise_fma(d1, d1, t);
        return __builtin_elementwise_fma(d2, d2, t);
    };

    // forward terms kept across iterations: index 0..4 <-> partner column offset -2..+2
    // The queue holds the terms ALREADY MOVED to the lane that will consume them, indexed by the consumer's tap
    // (k = 0..4 <-> tap column offset k-2): the consumer's partner x+(k-2) published the pair as its forward offset
    // -(k-2), i.e. as its element 4-k.  The shifts are issued where the terms are produced (the throughput-bound forward
    // rows) so that the backward rows, which open the next iterations, do not wait on two dependent DPP moves.
    float pF1[5], pF2[5], ppF2[5];     // for the consumer's row -1 (next iteration), row -2 (in two iterations), row -2 (next)
#pragma unroll
    for (int i = 0; i < 5; i++) { pF1[i] = 0.0f; pF2[i] = 0.0f; ppF2[i] = 0.0f; }
    auto publish = [&](const float (&F1)[5], const float (&F2)[5]) {
#pragma unroll
        for (int i = 0; i < 5; i++) ppF2[i] = pF2[i];
        pF1[0] = lane_from<-2>(F1[4]); pF1[1] = lane_from<-1>(F1[3]); pF1[2] = F1[2]; pF1[3] = lane_from<1>(F1[1]); pF1[4] = lane_from<2>(F1[0]);
        pF2[0] = lane_from<-2>(F2[4]); pF2[1] = lane_from<-1>(F2[3]); pF2[2] = F2[2]; pF2[3] = lane_from<1>(F2[1]); pF2[4] = lane_from<2>(F2[0]);
    };

    // forward rows (j = +1, +2) of the centre in lattice row br: 10 evaluations -> F1, F2 (+ optional taps)
    struct Acc { v2f rg, bv, ww; };
    auto accumulate = [&](Acc &acc, const v4f &Cq, float w) {
        if (HASVAR) {
            v2f wv2;
            wv2.x = w;
            wv2.y = w * w;
            acc.ww += wv2;
            acc.rg = __builtin_elementwise_fma(Cq.xy, v2f{w, w}, acc.rg);
            acc.bv = __builtin_elementwise_fma(Cq.zw, wv2, acc.bv);
        } else {
            acc.ww.x += w;
            acc.rg = __builtin_elementwise_fma(Cq.xy, v2f{w, w}, acc.rg);
            acc.bv.x = fmaf(Cq.z, w, acc.bv.x);
        }
    };

    struct ColRow { v4f C[5]; };                          // backward row: colour slots (luminance comes from the owning lanes, see do_col)
    struct GeoRow { v4f A[5], B[5]; };                    // forward row: geometry slots (the colour slot is read mid-row)
    struct OwnRow { v4f A[2], B[2], C[2], Cb[2]; };       // own row: +1, +2 full records; -1, -2 colour
    auto load_col = [&](ColRow &r, int br) {
        const char *rowp = colbase + slot_of(br) * RSTR;
#pragma unroll
        for (int i = 0; i < 5; i++) {
            r.C[i] = *reinterpret_cast<const v4f *>(rowp + i * PXB + 32);
        }
    };
    // (first_row: the forward row +1, whose three inner partners take their geometric term from the previous level with REUSE_IN:
    // only their B slot — the luminance — is read)
    auto load_geo = [&](GeoRow &r, int br, bool first_row = false) {
        const char *rowp = colbase + slot_of(br) * RSTR;
#pragma unroll
        for (int i = 0; i < 5; i++) {
            if (!(REUSE_IN && first_row && i >= 1 && i <= 3)) r.A[i] = *reinterpret_cast<const v4f *>(rowp + i * PXB);
            r.B[i] = *reinterpret_cast<const v4f *>(rowp + i * PXB + 16);
        }
    };
    auto load_own = [&](OwnRow &r, int br) {
        const char *rowp = colbase + slot_of(br) * RSTR;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            if (!(REUSE_IN && k == 0)) r.A[k] = *reinterpret_cast<const v4f *>(rowp + (k + 3) * PXB);
            r.B[k] = *reinterpret_cast<const v4f *>(rowp + (k + 3) * PXB + 16);
            r.C[k] = *reinterpret_cast<const v4f *>(rowp + (k + 3) * PXB + 32);
            r.Cb[k] = *reinterpret_cast<const v4f *>(rowp + (1 - k) * PXB + 32);
        }
    };
    // Backward row.  The taps' geometry terms come from the queue, and their LUMINANCE from registers too: lpH is the luminance
    // this lane had as the CENTRE of that row, one or two iterations ago; tap column x + io is lane + io, whose centre luminance
    // of that row is exactly the tap's (out-of-image pixels carry +inf in both places).  Four lane shifts instead of five 16-byte
    // LDS reads, in the rows whose VALU work is too short to hide LDS reads (profiles/r03_exp_lane_knockout.log: eight reads of
    // the backward rows cost 4.8 % of a level, the same eight in the forward rows 1.6 %), and 20 VGPRs fewer
    // (profiles/r03_ab_lane_lum_dpp.log: -1.1 us per level, -1.9 on the last).
    auto do_col = [&](Acc &acc, const ColRow &r, const float (&tt)[5], float lp, float kl, float lpH) {
        float e[5], w[5];
        const float l0 = lane_from<-2>(lpH), l1 = lane_from<-1>(lpH), l3 = lane_from<1>(lpH), l4 = lane_from<2>(lpH);
        e[0] = fmaf(fabsf(l0 - lp), kl, tt[0]); e[1] = fmaf(fabsf(l1 - lp), kl, tt[1]); e[2] = fmaf(fabsf(lpH - lp), kl, tt[2]);
        e[3] = fmaf(fabsf(l3 - lp), kl, tt[3]); e[4] = fmaf(fabsf(l4 - lp), kl, tt[4]);
        __builtin_amdgcn_sched_barrier(0x100);
#pragma unroll
        for (int i = 0; i < 5; i++) w[i] = __builtin_amdgcn_exp2f(-e[i]);
        __builtin_amdgcn_sched_barrier(0x100);
#pragma unroll
        for (int i = 0; i < 5; i++) accumulate(acc, r.C[i], w[i]);
    };
    // (tin: REUSE_IN — the previous level's geometric terms {g(+1,0), g(-1,+1), g(0,+1), g(+1,+1)} of this centre: the forward row
    // +1 takes partners -1, 0, +1 from it)
    auto do_geo = [&](Acc &acc, const GeoRow &r, int br, auto jtag, float (&F)[5], const v2f &c0, const v2f &c1,
                      const v2f &c2, float lp, float kl, GeoRow *next, int br_next, const v4f &tin) {
        constexpr int j = decltype(jtag)::value;
        constexpr bool TAKE = REUSE_IN && j == 1;
        const char *rowp = colbase + slot_of(br) * RSTR;
        v2f s2[5];
        v4f Cq[5];
        float lq[5];
#pragma unroll
        for (int i = 0; i < 5; i++) {
            if (!(TAKE && i >= 1 && i <= 3)) s2[i] = geo(r.A[i], r.B[i], c0, c1, c2);
            lq[i] = r.B[i].z;
        }
#pragma unroll
        for (int i = 0; i < 5; i++) Cq[i] = *reinterpret_cast<const v4f *>(rowp + i * PXB + 32);
        if (next) load_geo(*next, br_next);        // the next forward row's geometry, once this row's is consumed
        __builtin_amdgcn_sched_barrier(0x100);
        float dn[5], dx[5];
#pragma unroll
        for (int i = 0; i < 5; i++) {
            if (TAKE && i >= 1 && i <= 3) continue;
            dn[i] = __builtin_amdgcn_sqrtf(s2[i].x);
            dx[i] = __builtin_amdgcn_sqrtf(s2[i].y);
        }
        __builtin_amdgcn_sched_barrier(0x100);
        float e[5], w[5];
#pragma unroll
        for (int i = 0; i < 5; i++) {
            float t;
            if (TAKE && i >= 1 && i <= 3) {
                t = (i == 1 ? tin.y : (i == 2 ? tin.z : tin.w)) + nlh(i - 2, j);
            } else {
                t = fmaf(dn[i], kn, nlh(i - 2, j));
                t = fmaf(dx[i], kx, t);
            }
            F[i] = t;
            e[i] = fmaf(fabsf(lq[i] - lp), kl, t);
        }
        __builtin_amdgcn_sched_barrier(0x100);
#pragma unroll
        for (int i = 0; i < 5; i++) w[i] = __builtin_amdgcn_exp2f(-e[i]);
        __builtin_amdgcn_sched_barrier(0x100);
#pragma unroll
        for (int i = 0; i < 5; i++) accumulate(acc, Cq[i], w[i]);
    };

    // end of a tap row: nothing of this row may sink below, no LDS read of a later row may rise above (keeps the live
    // ranges of a row's 5 x 48 bytes of tap data from piling up: without it hipcc hoists every load of the iteration)
    auto row_fence = [&](Acc &acc) {
        float a0 = acc.rg.x, a1 = acc.rg.y, a2 = acc.bv.x, a3 = acc.bv.y, a4 = acc.ww.x, a5 = acc.ww.y;
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : : "memory");
        acc.rg = v2f{a0, a1}; acc.bv = v2f{a2, a3}; acc.ww = v2f{a4, a5};
    };

    // ---- warm-up: rows b0-2 and b0-1 publish their forward terms (no output, no new ring rows needed) ----
    float lp1 = 0.0f, lp2 = 0.0f;          // this lane's centre luminance one / two rows back
#pragma unroll 1
    for (int bw = b0 - 2; bw < b0; bw++) {
#if SVGF_LANE_SPLIT_PROLOGUE
        if constexpr (!TFUSED) { if (bw == b0 - 1) __syncthreads(); }     // rows b0+1, b0+2 are published by the loader threads (see the prologue)
#endif
        const char *rowc = colbase + slot_of(bw) * RSTR + 2 * PXB;
        const v4f A = *reinterpret_cast<const v4f *>(rowc);
        const v4f B = *reinterpret_cast<const v4f *>(rowc + 16);
        lp2 = lp1; lp1 = B.z;
        float F1[5] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f }, F2[5];
#pragma unroll
        for (int j = 1; j <= 2; j++) {
            // the terms row b0-2 shares with row b0-1 are never consumed (b0-1 is not an output row)
            if (j == 1 && bw == b0 - 2) continue;
            const char *rowp = colbase + slot_of(bw + j) * RSTR;
#pragma unroll
            for (int i = 0; i < 5; i++) {
                const v4f Aq = *reinterpret_cast<const v4f *>(rowp + i * PXB);
                const v4f Bq = *reinterpret_cast<const v4f *>(rowp + i * PXB + 16);
                const v2f s2 = geo(Aq, Bq, -A.xy, -A.zw, -B.xy);
                const float dn = fmaxf(__builtin_amdgcn_sqrtf(s2.x), 0.0f), dx = fmaxf(__builtin_amdgcn_sqrtf(s2.y), 0.0f);
                float t = fmaf(dn, kn, neg_log2_binom(i - 2) + neg_log2_binom(j));
                t = fmaf(dx, kx, t);
                if (j == 1) F1[i] = t; else F2[i] = t;
            }
        }
        publish(F1, F2);
    }

    // One iteration = one output row.  Once a non-finite normal / position has been staged (rare; the flag only ever
    // goes from 0 to 1, and is set before the row that needs it becomes anybody's partner) the workgroup switches to a
    // plain 24-tap loop that keeps the reference's min(1, exp(-NaN)) == 1 and shares nothing.
    // The centre's geometry and the variance of its two row neighbours are in the ring long before the row becomes the output
    // row: the variant without variance accumulators (150 / 138 VGPRs, room for 10 more) reads them at the END of the previous
    // iteration, in front of its output stage, so that the LDS round trip is over when the barrier opens — after the barrier
    // all eight waves ask at once, with nothing else to issue (-0.7 .. -1.3 us on that level).  At 164 VGPRs the carried
    // registers cost the other variants what the prefetch brings (profiles/r03_ab_lane_centre_prefetch.log): they read at the top.
#ifdef SVGF_LANE_PREFETCH_ALL
    constexpr bool PREFETCH_CENTRE = true;
#else
    constexpr bool PREFETCH_CENTRE = !HASVAR;
#endif
    struct Centre { v4f A, B; float c0v, c2v; float up[3], dn[3]; };
    const int off_c = (xph * MP + mcol) * PXB + 44;      // this lane's own column, like off_l / off_r
    auto prefetch_centre = [&](Centre &cn, int bo) {
        const char *rowc = colbase + slot_of(bo) * RSTR + 2 * PXB;
        cn.A = *reinterpret_cast<const v4f *>(rowc);
        cn.B = *reinterpret_cast<const v4f *>(rowc + 16);
        if constexpr (!CHUNKED) {
            // variance of the row neighbours x-1, x+1 (other x-phases): read as their whole C slot — a b128 is conflict-free at
            // the 48-byte lane stride, a 4-byte read is 4-way conflicted (lanes 8 apart share a bank)
            const char *ringrow = smem + ypw * ROWB + slot_of(bo) * RSTR;
            cn.c0v = reinterpret_cast<const v4f *>(ringrow + off_l - 12)->w;
            cn.c2v = reinterpret_cast<const v4f *>(ringrow + off_r - 12)->w;
            if constexpr (YP > 1) {
                // rows y-1 / y+1 are the OTHER y-phase's ring rows: for the even rows (ypw 0) lattice rows bo-1 / bo of phase 1,
                // for the odd rows lattice rows bo / bo+1 of phase 0
                const char *uprow = smem + (ypw ? slot_of(bo) * RSTR : slot_of(bo - 1) * RSTR + ROWB);
                const char *dnrow = smem + (ypw ? slot_of(bo + 1) * RSTR : slot_of(bo) * RSTR + ROWB);
                cn.up[0] = reinterpret_cast<const v4f *>(uprow + off_l - 12)->w;
                cn.up[1] = reinterpret_cast<const v4f *>(uprow + off_c - 12)->w;
                cn.up[2] = reinterpret_cast<const v4f *>(uprow + off_r - 12)->w;
                cn.dn[0] = reinterpret_cast<const v4f *>(dnrow + off_l - 12)->w;
                cn.dn[1] = reinterpret_cast<const v4f *>(dnrow + off_c - 12)->w;
                cn.dn[2] = reinterpret_cast<const v4f *>(dnrow + off_r - 12)->w;
            }
        } else { cn.c0v = 0.0f; cn.c2v = 0.0f; }
    };
    // the previous level's geometric terms of this lane's centre in lattice row br (halo lanes and lanes right of / below the
    // image too: their forward terms are their neighbours' backward terms; a clamped address keeps what they read finite, and
    // an out-of-image pixel has weight 0 whatever its terms are)
    // The terms plane is laid out for the CONSUMER's lanes: [row][x mod S'][x / S'] with S' the consumer's step (M' = a.t_m lattice
    // columns per phase): the 64 lanes of a consumer wave — consecutive lattice columns of one x-phase — read 1 KB of consecutive
    // bytes, and a producer wave (step S'/2: its even lanes belong to consumer phase ph, its odd lanes to ph + S'/2) writes two
    // runs of 512 consecutive bytes.  (Pixel-major, the first version, had every lane of both on its own cache line from step 8
    // on: the levels got 5-20 % SLOWER, profiles/r04_ab_reuse_pixel_major.log.)
    auto fetch_terms = [&](int br) -> v4f {
        const int yc = min(yphase + (br << LOG2S), H - 1), xc = min(max(x, 0), W - 1);
        const unsigned idx = ((unsigned)(yc << LOG2S) + (unsigned)(xc & (S - 1))) * (unsigned)a.t_m + (unsigned)(xc >> LOG2S);
        return *reinterpret_cast<const v4f *>(reinterpret_cast<const char *>(a.tin) + idx * 16u);
    };
    // tcur: REUSE_IN — the previous level's four geometric terms of this iteration's centre (fetched one iteration ahead)
    auto body = [&](int bo, int it, bool careful, Centre &cen, v4f &tcur) {
        stamp(0);
        constexpr int PR[7] = { SVGF_LANE_PRIO };
        if constexpr (PR[0] == PR[3]) __builtin_amdgcn_s_setprio(PR[0]);
        else { if (flip) __builtin_amdgcn_s_setprio(PR[3]); else __builtin_amdgcn_s_setprio(PR[0]); }
        const int y = yphase + (bo << LOG2S);
        if constexpr (!PREFETCH_CENTRE) prefetch_centre(cen, bo);
        const v4f A = cen.A, B = cen.B;
        const v4f C = *reinterpret_cast<const v4f *>(colbase + slot_of(bo) * RSTR + 2 * PXB + 32);
        ColRow r0;
        GeoRow g1;
        float var;
        if constexpr (CHUNKED) {
            // the loader threads computed the blurred variance of this row from the variance plane (see vblur_store)
            const float bvar = blur[(it & 1) * BLUR_BUF + xph * M + mcol];
            if (flip) load_geo(g1, bo + 1, true); else load_col(r0, bo - 2);
            var = a.blur_variance ? bvar : C.w;
        } else {
        float m0, m1, m2, p0, p1, p2;
        if constexpr (YP > 1) {
            m0 = cen.up[0]; m1 = cen.up[1]; m2 = cen.up[2]; p0 = cen.dn[0]; p1 = cen.dn[1]; p2 = cen.dn[2];
        } else {
            const float *bl = blur + (it & 1) * BLUR_BUF;
            m0 = bl[be_l]; m1 = bl[be_c]; m2 = bl[be_r];
            p0 = bl[BLUR_ROW + be_l]; p1 = bl[BLUR_ROW + be_c]; p2 = bl[BLUR_ROW + be_r];
        }
        const float c0v = cen.c0v, c2v = cen.c2v;
        if (flip) load_geo(g1, bo + 1, true); else load_col(r0, bo - 2);       // the first tap row of this wave's stage order
        {   // centre variance: 3x3 gaussian with out-of-image taps dropped and renormalised (:102-118)
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
        }
        var = fmaxf(var, 0.0f);
        const float lp = B.z;
        const float kl = kLog2e * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(var) * sigma_c + 1e-6f);
        const v2f c0 = v2f{-A.x, -A.y}, c1 = v2f{-A.z, -A.w}, c2 = v2f{-B.x, -B.y};      // negated centre, see geo()

        stamp(1);
        // centre tap: weight exactly h = 9/64
        constexpr float w0 = 0.140625f;
        Acc acc;
        acc.ww = v2f{w0, w0 * w0};
        acc.rg = v2f{w0 * C.x, w0 * C.y};
        acc.bv = v2f{w0 * C.z, (w0 * w0) * C.w};

        v4f tout_v = v4f{0.0f, 0.0f, 0.0f, 0.0f};             // REUSE_OUT: g(+2,0), g(-2,+2), g(0,+2), g(+2,+2) of this centre
        if (careful) {
            acc.rg = v2f{0.0f, 0.0f}; acc.bv = v2f{0.0f, 0.0f}; acc.ww = v2f{0.0f, 0.0f};
#pragma unroll 1
            for (int j = -2; j <= 2; j++) {
                const char *rowp = colbase + slot_of(bo + j) * RSTR;
#pragma unroll 1
                for (int i = -2; i <= 2; i++) {
                    const v4f Aq = *reinterpret_cast<const v4f *>(rowp + (i + 2) * PXB);
                    const v4f Bq = *reinterpret_cast<const v4f *>(rowp + (i + 2) * PXB + 16);
                    const v4f Cq = *reinterpret_cast<const v4f *>(rowp + (i + 2) * PXB + 32);
                    const v2f s2 = geo(Aq, Bq, c0, c1, c2);
                    const float dn = fmaxf(__builtin_amdgcn_sqrtf(s2.x), 0.0f), dx = fmaxf(__builtin_amdgcn_sqrtf(s2.y), 0.0f);
                    const int ai = i < 0 ? -i : i, aj = j < 0 ? -j : j;
                    const float nl = (ai == 0 ? 1.4150374992788437f : (ai == 1 ? 2.0f : 4.0f)) + (aj == 0 ? 1.4150374992788437f : (aj == 1 ? 2.0f : 4.0f));
                    float e = fmaf(fabsf(Bq.z - lp), kl, nl);
                    e = fmaf(dn, kn, e);
                    e = fmaf(dx, kx, e);
                    accumulate(acc, Cq, __builtin_amdgcn_exp2f(-e));
                    if constexpr (REUSE_OUT) {
                        // the next level's workgroups that do not stage the non-finite texel read these terms like any others
                        const float g = fmaf(dx, kx, dn * kn);
                        if (j == 0 && i == 2) tout_v.x = g;
                        if (j == 2 && i == -2) tout_v.y = g;
                        if (j == 2 && i == 0) tout_v.z = g;
                        if (j == 2 && i == 2) tout_v.w = g;
                    }
                }
            }
        } else {
        // Each row's first LDS reads are issued one row ahead (in front of the previous row's fence).
        // own row: the two right-hand neighbours are evaluated, the two left-hand ones arrive from lanes x-1, x-2
        float tf[2];                                          // own row's forward terms (+1, 0), (+2, 0)
        auto do_own = [&](const OwnRow &r2) {
            float e[4];
#pragma unroll
            for (int k = 0; k < 2; k++) {
                if (REUSE_IN && k == 0) { tf[0] = tcur.x + nlh(1, 0); continue; }
                const v2f s2 = geo(r2.A[k], r2.B[k], c0, c1, c2);
                const float dn = __builtin_amdgcn_sqrtf(s2.x), dx = __builtin_amdgcn_sqrtf(s2.y);
                const float t = fmaf(dn, kn, nlh(k + 1, 0));
                tf[k] = fmaf(dx, kx, t);
            }
            const float tb1 = lane_from<-1>(tf[0]), tb2 = lane_from<-2>(tf[1]);
            const float ll2 = lane_from<-2>(lp), ll1 = lane_from<-1>(lp);      // the left-hand neighbours' luminance: their centre's
            e[0] = fmaf(fabsf(ll2 - lp), kl, tb2);          // io = -2
            e[1] = fmaf(fabsf(ll1 - lp), kl, tb1);          // io = -1
            e[2] = fmaf(fabsf(r2.B[0].z - lp), kl, tf[0]);      // io = +1
            e[3] = fmaf(fabsf(r2.B[1].z - lp), kl, tf[1]);      // io = +2
            __builtin_amdgcn_sched_barrier(0x100);
            float w[4];
#pragma unroll
            for (int k = 0; k < 4; k++) w[k] = __builtin_amdgcn_exp2f(-e[k]);
            __builtin_amdgcn_sched_barrier(0x100);
            accumulate(acc, r2.Cb[1], w[0]);
            accumulate(acc, r2.Cb[0], w[1]);
            accumulate(acc, r2.C[0], w[2]);
            accumulate(acc, r2.C[1], w[3]);
        };
        float F1[5], F2[5];
        if (!flip) {
            // ---- stage order A (waves 0-3): backward rows, own row, forward rows ----
            // backward rows: colour part only; t comes from the queue (already moved to this lane by publish())
            ColRow r1;
            load_col(r1, bo - 1);
            do_col(acc, r0, ppF2, lp, kl, lp2);
            OwnRow r2;
            load_own(r2, bo);
            row_fence(acc);
            do_col(acc, r1, pF1, lp, kl, lp1);
            __builtin_amdgcn_s_setprio(PR[1]);              // ~1/4 of the row's work done
            load_geo(g1, bo + 1, true);
            row_fence(acc);
            stamp(2);
            do_own(r2);
            stamp(3);
            row_fence(acc);
            // forward rows: evaluate, use, and keep for the partners
            do_geo(acc, g1, bo + 1, std::integral_constant<int, 1>{}, F1, c0, c1, c2, lp, kl, nullptr, 0, tcur);
            __builtin_amdgcn_s_setprio(PR[2]);              // ~2/3
            row_fence(acc);
            GeoRow g2;
            load_geo(g2, bo + 2);
            do_geo(acc, g2, bo + 2, std::integral_constant<int, 2>{}, F2, c0, c1, c2, lp, kl, nullptr, 0, tcur);
            row_fence(acc);
        } else {
            // ---- stage order B (waves 4-7, which share their SIMDs with waves 0-3): forward rows, own row, backward
            //      rows.  The forward rows are VALU-heavy, the backward rows LDS-heavy: with the two waves of a SIMD in
            //      opposite orders the two kinds of work overlap instead of queueing up behind the same pipe. ----
            do_geo(acc, g1, bo + 1, std::integral_constant<int, 1>{}, F1, c0, c1, c2, lp, kl, nullptr, 0, tcur);
            __builtin_amdgcn_s_setprio(PR[4]);              // ~1/3
            row_fence(acc);
            GeoRow g2;
            load_geo(g2, bo + 2);
            do_geo(acc, g2, bo + 2, std::integral_constant<int, 2>{}, F2, c0, c1, c2, lp, kl, nullptr, 0, tcur);
            __builtin_amdgcn_s_setprio(PR[5]);              // ~2/3
            OwnRow r2;
            load_own(r2, bo);
            row_fence(acc);
            stamp(2);
            load_col(r0, bo - 2);
            do_own(r2);
            __builtin_amdgcn_s_setprio(PR[6]);              // ~3/4
            stamp(3);
            row_fence(acc);
            ColRow r1;
            load_col(r1, bo - 1);
            do_col(acc, r0, ppF2, lp, kl, lp2);
            row_fence(acc);
            do_col(acc, r1, pF1, lp, kl, lp1);
            row_fence(acc);
        }
        publish(F1, F2);
        if constexpr (REUSE_OUT) tout_v = v4f{tf[1] - nlh(2, 0), F2[0] - nlh(-2, 2), F2[2] - nlh(0, 2), F2[4] - nlh(2, 2)};
        }

        stamp(4);
        Centre nxt;
        if constexpr (PREFETCH_CENTRE) prefetch_centre(nxt, bo + 1);          // row bo+1 has been in the ring since iteration it-2
        if (out_lane && (YP == 1 || y < H)) {
            const float r0 = acc.rg.x, r1 = acc.rg.y, r2 = acc.bv.x, vsum = acc.bv.y, wsum = acc.ww.x, w2sum = acc.ww.y;
            float o0, o1, o2, ov;
            if (wsum > 1e-5f) {                                     // NaN -> false -> pass-through (:159-164)
                const float rw = __builtin_amdgcn_rcpf(wsum);
                o0 = r0 * rw; o1 = r1 * rw; o2 = r2 * rw;
                ov = HASVAR ? vsum * __builtin_amdgcn_rcpf(w2sum) : 0.0f;
            } else {
                o0 = C.x; o1 = C.y; o2 = C.z; ov = C.w;
            }
            const unsigned p = (unsigned)y * (unsigned)W + (unsigned)x;
            if (a.modulate) svgf_modulate(a, p, o0, o1, o2);       // last level: * albedo * ialbedo (:166-168)
            if (a.dst) a.dst[p] = make_float4(o0, o1, o2, ov);
            if (a.var_dst) a.var_dst[(unsigned)(y + 1) * (unsigned)(W + 2) + (unsigned)(x + 1)] = ov;
            if (a.out_rgb) { float *o = a.out_rgb + 3u * p; o[0] = o0; o[1] = o1; o[2] = o2; }
            if constexpr (REUSE_OUT) {       // consumer's layout: step 2S, a.t_m_out lattice columns per phase
                const unsigned idx = ((unsigned)(y << (LOG2S + 1)) + (unsigned)(x & (2 * S - 1))) * (unsigned)a.t_m_out + (unsigned)(x >> (LOG2S + 1));
                *reinterpret_cast<v4f *>(reinterpret_cast<char *>(a.tout) + idx * 16u) = tout_v;
            }
        }
        // (requesting them at the TOP of the iteration instead, in front of the output stores whose acknowledgements an in-order
        // vmcnt otherwise waits for, was measured: worse, profiles/r04_ab_reuse_early_load.log)
        if constexpr (REUSE_IN) tcur = fetch_terms(bo + 1);
        if constexpr (PREFETCH_CENTRE) cen = nxt;
        lp2 = lp1; lp1 = lp;
    };

    int it = 0;
    Centre cen;
    if constexpr (PREFETCH_CENTRE) prefetch_centre(cen, b0);
    v4f tcur = v4f{0.0f, 0.0f, 0.0f, 0.0f};
    if constexpr (REUSE_IN) tcur = fetch_terms(b0);
    for (int bo = b0; bo < b1; bo++, it++) {
        body(bo, it, *nan_seen != 0, cen, tcur);
        stamp(5);
        __syncthreads();
        stamp(6);
        ring_advance();
        dbg_it++;
    }
}

// strips: TXO contiguous pixel columns (one x-phase per wave group), or (lattice-column strip, group of P phases) pairs (chunked)
inline int lane_strip_count(int W, int S, int YP = 1)
{
    if (S <= 8) return (W + TXO / YP - 1) / (TXO / YP);
    const int P = 8, txl = LOUT * (NWC / P);      // steps 16, 32: chunks of 8 x-phases, one wave of 60 lattice columns each
    return (((W + S - 1) / S + txl - 1) / txl) * (S / P);
}

// segment length: one workgroup per CU (LDS-bound); the busiest XCD sets the number of rounds (see the strip kernel).  Returns
// the minimum of rounds * (L + 6) in lattice rows and the segment length that reaches it.
// (S = number of y-phase groups a strip is cut into: the step, or step / 2 with both y-phases in one workgroup)
inline long lane_segment_search(int n_strips, int S, int nb_max, int n_cu, int *best_L_out)
{
    int best_L = nb_max;
    long best_cost = -1;
    const long cu_xcd = n_cu / 8 > 0 ? n_cu / 8 : 1;
    for (int L = 4; L <= nb_max + 1; L++) {
        const int segs_l = (nb_max + L - 1) / L;
        const long blocks_xcd = (long)n_strips * ((S * segs_l + 7) / 8);
        const long rounds = (blocks_xcd + cu_xcd - 1) / cu_xcd;
        const long cost = rounds * (L + 6);
        if (best_cost < 0 || cost <= best_cost) { best_cost = cost; best_L = L; }
    }
    *best_L_out = best_L;
    return best_cost;
}

template <int LOG2S, bool HASVAR, int LOG2P = LOG2S, int LOG2Y = 0, int FUSED = 0, int REUSE = 0>
hipError_t launch_lane_cfg(const AtrousArgs &a, hipStream_t s, const LaneFused *ta = nullptr)
{
    constexpr int S = 1 << LOG2S, P = 1 << LOG2P, YP = 1 << LOG2Y, M = LOUT * (NWC / (P * YP)) + 4, MP = (P == 4) ? M + 2 : ((M * 12 % 64 == 0) ? M + 1 : M), BM = (BW + S - 1) / S;
    constexpr size_t kLds = (size_t)R * YP * P * MP * PXB + (size_t)2 * (YP > 1 ? 0 : (P < S ? P * M : 2 * S * BM)) * 4 + 16 + ((FUSED == 1 || FUSED == 2) ? 80 : 0);
    const size_t lds = kLds;
    static_assert(kLds <= 160 * 1024, "LDS budget");
    static SvgfLaunchCache cache;
    int dev_id = 0;
    if (hipError_t e = cache.init(reinterpret_cast<const void *>(&k_atrous_lane<LOG2S, HASVAR, LOG2P, LOG2Y, FUSED, REUSE>), (int)lds, &dev_id); e != hipSuccess) return e;
    const int n_cu = cache.n_cu[dev_id];
    LaneGeom gm;
    gm.n_strips = lane_strip_count(a.W, S, YP);
    static_assert(P == S || ((S == 16 || S == 32) && P == 8), "lane_strip_count knows these chunk sizes");
    const int nb_max = (a.H + S - 1) / S;
    int best_L = nb_max;
    (void)lane_segment_search(gm.n_strips, S / YP, nb_max, n_cu, &best_L);
    if (const int v = SVGF_TUNE("lane_segrows", 0); v > 0) best_L = v;      // experiments build only (tools/experiments/exp_small_frames.sh)
    gm.seg_rows = best_L;
    gm.n_segs = (nb_max + best_L - 1) / best_L;
    gm.n_groups = (S / YP) * gm.n_segs;
    gm.kn = (float)(1.4426950408889634 / ((double)a.sigma_n + 1e-6));
    gm.kx = (float)(1.4426950408889634 / ((double)a.sigma_x + 1e-6));
    const int groups_pad = (gm.n_groups + 7) / 8 * 8;
    const int nblocks = groups_pad * gm.n_strips;
    gm.dbg = nullptr; gm.dbg_block = 0;
#ifdef SVGF_LANE_TIMELINE
    static unsigned long long *dbg_buf = nullptr;
    const int dbg_block = SVGF_TUNE("lane_dbg", -1);
    const bool dbg_env = dbg_block >= 0;
    if (dbg_env) {
        if (!dbg_buf) (void)hipMalloc((void **)&dbg_buf, 16 * 16 * 8 * sizeof(unsigned long long));
        (void)hipMemsetAsync(dbg_buf, 0, 16 * 16 * 8 * sizeof(unsigned long long), s);
        gm.dbg = dbg_buf; gm.dbg_block = dbg_block;
    }
#endif
    if constexpr (FUSED != 0) SVGF_LAUNCH_KERNEL((k_atrous_lane<LOG2S, HASVAR, LOG2P, LOG2Y, FUSED, REUSE>), dim3(nblocks), dim3(NT), lds, s, a, gm, *ta);
    else SVGF_LAUNCH_KERNEL((k_atrous_lane<LOG2S, HASVAR, LOG2P, LOG2Y, 0, REUSE>), dim3(nblocks), dim3(NT), lds, s, a, gm, LaneNoTemporal{});
#ifdef SVGF_LANE_TIMELINE
    if (dbg_env) {
        static int skip = SVGF_TUNE("lane_dbg_skip", 0), prints = 0;
        (void)hipStreamSynchronize(s);
        unsigned long long h[16 * 16 * 8];
        (void)hipMemcpy(h, dbg_buf, sizeof(h), hipMemcpyDeviceToHost);
        if (skip > 0) skip--;
        else if (prints++ < 4) {
            fprintf(stderr, "[lane dbg] S=%d blocks=%d segs=%d seg_rows=%d lds=%zu\n", S, nblocks, gm.n_segs, gm.seg_rows, lds);
            const int show[4] = { 0, NWC - 1, NWC, NWC + 3 };
            for (int si = 0; si < 4; si++) {
                const int w = show[si];
                if (h[(w * 16) * 8 + 7])
                    fprintf(stderr, "  wave %2d prologue: entry .. loads issued %6llu, .. barrier passed %6llu, .. first iteration %6llu ticks\n", w,
                            h[(w * 16 + 2) * 8 + 7] - h[(w * 16) * 8 + 7], h[(w * 16 + 1) * 8 + 7] - h[(w * 16) * 8 + 7],
                            h[(w * 16) * 8 + 0] - h[(w * 16) * 8 + 7]);
                for (int it = 0; it < 8 && h[(w * 16 + it) * 8]; it++) {
                    unsigned long long *t = &h[(w * 16 + it) * 8];
                    if (w >= NWC && (FUSED == 1 || FUSED == 2)) fprintf(stderr, "  loader %2d it %2d: t0=%6llu first pixel: C2 (blend, commit) %5llu C1 (consistency, history request) %5llu B %5llu A %5llu | second pixel %6llu | barrier %5llu\n", w, it,
                                                   t[0] - h[0], t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[6] - t[5]);
                    else if (w >= NWC) fprintf(stderr, "  loader %2d it %2d: t0=%6llu work %6llu barrier %5llu\n", w, it, t[0] - h[0], t[5] - t[0], t[6] - t[5]);
                    else fprintf(stderr, "  wave %2d it %2d: t0=%6llu centre %5llu back rows %5llu own %5llu fwd rows %5llu out %5llu barrier %5llu\n", w, it,
                                 t[0] - h[0], t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[6] - t[5]);
                }
            }
        }
    }
#endif
    return hipGetLastError();
}

}  // namespace
