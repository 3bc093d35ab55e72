// svgf_atrous_pw.hip — a-trous level, lane-marching with a PRIVATE LDS ring per wave (gfx950), steps 1 .. 32.
//
// Same result as the other a-trous kernels (one level of reference ATrousFilter, src/denoise.cu:77-170, snapshot
// variance) and the same tap arithmetic as svgf_atrous_lane.hip: a lane owns one lattice column and marches down its
// rows, the symmetric geometric part of every pair is evaluated once and handed to the partner lane by a DPP wave
// shift.  What changes is the staging.  In the lane kernel four loader waves fill one workgroup-wide ring (48-byte
// records, 6 rows) and every iteration ends in an s_barrier; that ring is 140-150 KB, so a CU holds 8 compute waves,
// two per SIMD, and two waves do not keep a SIMD's VALU busy (PMC: 63 % VALU-active).  Here
//   * the 64 lanes of a wave hold 64 consecutive lattice columns of ONE x-phase (60 outputs + 2 halo columns either
//     side), and every tap of those outputs lies inside the same 64 columns: a wave needs nobody else's data;
//   * so every wave keeps its own ring and fills it itself — each lane loads the next row of its own column straight
//     from the planes, one iteration ahead, converts it and ds_writes it at the END of the iteration into the slots
//     of the rows that just left the window — and never meets a barrier (LDS operations of one wave are ordered).
//     Colour {r,g,b,var} + luminance live 5 rows (b-2 .. b+2), geometry {n,p} 3 rows (b .. b+2): 172 B per column,
//     11.7 KB per wave, TWELVE waves per CU = three per SIMD;
//   * the 3x3 variance pre-blur (src/denoise.cu:102-118) reads its neighbourhood from a zero-margined 4-byte variance
//     plane (three dwordx3 loads per output pixel, one iteration ahead) instead of LDS-staged rows;
//   * the waves that share a SIMD publish their row counters in LDS and whoever is behind takes the higher priority
//     (the SIMD arbiter alone serves the oldest wave first and lets it run away).
// The lane's global accesses are strided by S pixels; the x-phases of a strip are waves of the same workgroup and
// touch the same cache lines at about the same time, which the vector L1 / L2 absorb.
//
// Work decomposition: workgroup = 12 waves = 3 consecutive row segments x 4 waves; the 4 waves of a segment are
// min(S,4) consecutive x-phases x 4/min(S,4) blocks of 60 lattice columns of one y-phase (S <= 4: 240 contiguous pixel
// columns; S >= 8: four phases of a 60-column block).  Waves w, w+4, w+8 share a SIMD.
#include "svgf_kernels.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <vector>

namespace {

constexpr float kLog2e = 1.44269504088896340736f;
constexpr int GW = 4;                        // waves per segment group
constexpr int NSG = 3;                       // segments per workgroup
constexpr int NW = GW * NSG;                 // 12 waves per workgroup, all of them compute
constexpr int LOUT = 60;                     // output lanes per wave (2 halo lanes either side)
constexpr int RECS = 64 + 4;                 // records per ring row: the wave's 64 columns + 2 never-written pads either side
constexpr int CSLOTS = 5, GSLOTS = 3;        // colour rows b-2 .. b+2, geometry rows b .. b+2
// per-wave LDS layout
constexpr int OFF_C = 0;                                   // float4 {r,g,b,var}      [CSLOTS][RECS]
constexpr int OFF_A = OFF_C + CSLOTS * RECS * 16;          // float4 {n.x,p.x,n.y,p.y} [GSLOTS][RECS]
constexpr int OFF_Z = OFF_A + GSLOTS * RECS * 16;          // float2 {n.z,p.z}        [GSLOTS][RECS]
constexpr int OFF_L = OFF_Z + GSLOTS * RECS * 8;           // float  luminance        [CSLOTS][RECS]
constexpr int RINGB = OFF_L + CSLOTS * RECS * 4;           // 11696 B per wave
static_assert(RINGB % 16 == 0, "ring alignment");

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v3f __attribute__((ext_vector_type(3)));
typedef float v4f __attribute__((ext_vector_type(4)));

struct PwGeom {
    int n_strips, n_segs, seg_rows, n_groups;   // n_segs: row segments per y-phase; n_groups = S * ceil(n_segs / NSG)
    int n_pg;                  // x-phase groups per column-block group (S / min(S, 4))
    float kn, kx;
    unsigned long long *dbg;   // tuning only (-DSVGF_PW_TIMELINE + SVGF_PW_DBG=1): s_memtime stamps of every wave
};

struct Row {                   // one staged pixel: the next row of the lane's own column, exactly as loaded
    v4f cv;
    v3f n, p;
};

// a wave-uniform pointer pinned in SGPRs, so that pointer + 32-bit lane offset selects the saddr form of global_load
typedef const __attribute__((address_space(1))) char *gptr;
__device__ __forceinline__ gptr uniform_ptr(const char *p)
{
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (gptr)(((unsigned long long)hi << 32) | lo);
}
template <typename T>
__device__ __forceinline__ T gload(gptr row, unsigned off)
{
    return *(const __attribute__((address_space(1))) T *)(row + off);
}

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

template <int LOG2S, bool HASVAR>
__global__ __launch_bounds__(NW * 64) void k_atrous_pw(AtrousArgs a, PwGeom gm)
{
    constexpr int S = 1 << LOG2S;
    constexpr int PH = S < GW ? S : GW;          // x-phases per segment group
    constexpr int CB = GW / PH;                  // 60-column blocks per phase per segment group

    extern __shared__ __attribute__((aligned(16))) char smem[];

    asm volatile("" :: "s"(a.src), "s"(a.dst), "s"(a.out_rgb), "s"(a.nrm), "s"(a.pos), "s"(a.gbuf), "s"(a.var), "s"(a.var_dst), "s"(a.W), "s"(a.H),
                 "s"(a.sigma_c), "s"(a.blur_variance), "s"(a.modulate), "s"(gm.n_strips), "s"(gm.n_segs), "s"(gm.seg_rows),
                 "s"(gm.n_groups), "s"(gm.n_pg), "s"(gm.kn), "s"(gm.kx));

    // ---- work item: (strip, y-phase, segment triple); blockIdx % 8 (= XCD) selects the (y-phase, triple) group so that
    //      the strips and x-phase groups that share cache lines share one L2 ----
    const int bid = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto stamp = [&](int id) {
#ifdef SVGF_PW_TIMELINE
        if (gm.dbg && lane == 0) {
            gm.dbg[((size_t)bid * NW + wv) * 4 + id] = __builtin_amdgcn_s_memtime();
            if (id == 0) gm.dbg[((size_t)bid * NW + wv) * 4 + 3] = (__builtin_amdgcn_s_memrealtime() << 16) | (__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (15 << 11)) & 0xffff);
        }
#else
        (void)id;
#endif
    };
    stamp(0);
    const int xcd = bid & 7, kk = bid >> 3;
    const int g = xcd + 8 * (kk / gm.n_strips);
    const int strip = kk % gm.n_strips;
    if (g >= gm.n_groups) return;
    const int n_trip = (gm.n_segs + NSG - 1) / NSG;
    const int yph = g / n_trip, trip = g % n_trip;
    const int W = a.W, H = a.H;
    if (yph >= H) return;
    // ---- progress words (see body()); written before any wave can leave ----
    __shared__ int prog[NW];
    auto prog_set = [&](int v) { if (lane == 0) __hip_atomic_store(&prog[wv], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto prog_get = [&](int w) { return __hip_atomic_load(&prog[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    prog_set(0);
    const int seg = trip * NSG + wv / GW;
    const int nb = (H - yph + S - 1) >> LOG2S;
    const int b0 = seg * gm.seg_rows;
    const int b1 = min(b0 + gm.seg_rows, nb);
    const int wq = wv % GW;
    // the wave's x-phase and first lattice column; lane l holds lattice column kcol, pixel column x
    const int pg = strip % gm.n_pg, cbg = strip / gm.n_pg;
    const int xph = pg * PH + wq / CB;
    const int kcol = (cbg * CB + wq % CB) * LOUT - 2 + lane;
    const int x = xph + (kcol << LOG2S);
    const bool col_ok = (kcol >= 0) && (x < W);
    const bool out_lane = (lane >= 2) && (lane < 2 + LOUT) && col_ok;
    const int xc = min(max(x, 0), W - 1);
    // nothing to do for this wave: mark it finished for its SIMD partners and leave
    if (b0 >= b1 || xph >= W || (cbg * CB + wq % CB) * LOUT * S + xph >= W) {
        prog_set(0x7fffffff);
        return;
    }
    float sigma_c = a.sigma_c;
    asm volatile("" : "+s"(sigma_c));
    const float kn = gm.kn, kx = gm.kx;

    // ---- the wave's private ring.  Row slots rotate: the iteration with centre b keeps cs0 = colour slot of row b-2,
    //      gs0 = geometry slot of row b ----
    const unsigned wave_off = (unsigned)wv * RINGB;
    const unsigned c16 = wave_off + OFF_C + lane * 16, a16 = wave_off + OFF_A + lane * 16;
    const unsigned z8 = wave_off + OFF_Z + lane * 8, l4 = wave_off + OFF_L + lane * 4;
    int cs0 = 0, gs0 = 0;                            // wave-uniform
    auto cslot = [&](int d) { int s = cs0 + d; s -= (s >= CSLOTS) ? CSLOTS : 0; return s; };     // d = br - (b-2) in 0..5
    auto gslot = [&](int d) { int s = gs0 + d; s -= (s >= GSLOTS) ? GSLOTS : 0; return s; };     // d = br - b     in 0..3
    auto ring_advance = [&]() { cs0 = cslot(1); gs0 = gslot(1); };
    // pointers to tap 0 (column lane-2) of a row; tap i at + i * elem
    auto Cp = [&](int dc) { return smem + c16 + cslot(dc) * (RECS * 16); };
    auto Lp = [&](int dc) { return smem + l4 + cslot(dc) * (RECS * 4); };
    auto Ap = [&](int dg) { return smem + a16 + gslot(dg) * (RECS * 16); };
    auto Zp = [&](int dg) { return smem + z8 + gslot(dg) * (RECS * 8); };

    // ---- staging: global -> registers (one iteration ahead) -> ring ----
    // every address is a wave-uniform row pointer (SGPRs) + a per-lane constant 32-bit byte offset
    bool careful = false;                            // a non-finite normal / position has entered the wave's window
    const unsigned xo16 = (unsigned)xc * 16u, xo12 = (unsigned)xc * 12u;
    // Loads are UNCONDITIONAL (a conditional load becomes a phi, and the phi's copies sit right behind the load together
    // with an s_waitcnt for it); a load that is not needed gets lane mask 0, i.e. all lanes read the row's first pixel.
    auto row_load = [&](Row &r, int br, unsigned mask) {
        const int y = yph + (br << LOG2S);
        const size_t rowq = (size_t)min(max(y, 0), H - 1) * (size_t)W;
        const gptr rc = uniform_ptr(reinterpret_cast<const char *>(a.src) + rowq * 16u);
        const gptr rn = uniform_ptr(reinterpret_cast<const char *>(a.nrm) + rowq * 12u);
        const gptr rp = uniform_ptr(reinterpret_cast<const char *>(a.pos) + rowq * 12u);
        r.cv = gload<v4f>(rc, xo16 & mask);
        r.n = gload<v3f>(rn, xo12 & mask);
        r.p = gload<v3f>(rp, xo12 & mask);
    };
    // store row br; dc / dg: its slot distances in the rings as seen from the current ring position
    auto row_store = [&](Row &r, int br, int dc, int dg) {
        asm volatile("" : "+v"(r.cv), "+v"(r.n), "+v"(r.p));          // nothing of the conversion rises above this point
        const int y = yph + (br << LOG2S);
        const bool ok = col_ok && (br >= 0) && (y < H);
        const float inf = __builtin_huge_valf();
        const float lum = lum_f64(r.cv.x, r.cv.y, r.cv.z);
        const float mag = fabsf(r.n.x) + fabsf(r.n.y) + fabsf(r.n.z) + fabsf(r.p.x) + fabsf(r.p.y) + fabsf(r.p.z);
        if (__builtin_amdgcn_ballot_w64(!(mag < inf)) != 0) careful = true;
        *reinterpret_cast<v4f *>(Ap(dg) + 2 * 16) = v4f{r.n.x, r.p.x, r.n.y, r.p.y};
        *reinterpret_cast<v2f *>(Zp(dg) + 2 * 8) = v2f{r.n.z, r.p.z};
        *reinterpret_cast<v4f *>(Cp(dc) + 2 * 16) = ok ? r.cv : v4f{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<float *>(Lp(dc) + 2 * 4) = ok ? lum : inf;
    };
    // 3x3 pre-blur neighbourhood of the centre (x, y) of output row bo (:102-118): rows y-1, y, y+1, columns x-1 .. x+1 of
    // the variance plane (W+2 x H+2, zero margins): three dwordx3 loads at (y + dy + 1, x) of the plane.
    struct Blur { v3f m, c, p; };
    const unsigned vpitch = (unsigned)W + 2u;
    const unsigned vo = (unsigned)xc * 4u;
    auto blur_load = [&](Blur &b, int bo, unsigned mask) {
        const int y = min(yph + (bo << LOG2S), H - 1);
        const gptr rm = uniform_ptr(reinterpret_cast<const char *>(a.var) + (size_t)y * vpitch * 4u);      // plane row y = image row y-1
        const gptr r0 = uniform_ptr(reinterpret_cast<const char *>(a.var) + (size_t)(y + 1) * vpitch * 4u);
        const gptr rp = uniform_ptr(reinterpret_cast<const char *>(a.var) + (size_t)(y + 2) * vpitch * 4u);
        b.m = gload<v3f>(rm, vo & mask);
        b.c = gload<v3f>(r0, vo & mask);
        b.p = gload<v3f>(rp, vo & mask);
    };

    // geometry evaluation of one partner: (|dn|^2, |dx|^2); A = {n.x,p.x,n.y,p.y}, Z = {n.z,p.z}
    // (the centre is passed NEGATED: q + (-c) keeps the three differences on v_pk_add_f32; with q - c hipcc splits them)
    auto geo = [&](const v4f &Aq, const v2f &Zq, const v2f &nc0, const v2f &nc1, const v2f &nc2) {
        const v2f d0 = Aq.xy + nc0, d1 = Aq.zw + nc1, d2 = Zq + nc2;
        v2f t = d0 * d0;
        t = __builtin_elementwise_fma(d1, d1, t);
        return __builtin_elementwise_fma(d2, d2, t);
    };

    // forward terms kept across iterations, already moved to the lane that consumes them (see svgf_atrous_lane.hip)
    float pF1[5], pF2[5], ppF2[5];
#pragma unroll
    for (int i = 0; i < 5; i++) { pF1[i] = 0.0f; pF2[i] = 0.0f; ppF2[i] = 0.0f; }
    auto publish = [&](const float (&F1)[5], const float (&F2)[5]) {
#pragma unroll
        for (int i = 0; i < 5; i++) ppF2[i] = pF2[i];
        pF1[0] = lane_from<-2>(F1[4]); pF1[1] = lane_from<-1>(F1[3]); pF1[2] = F1[2]; pF1[3] = lane_from<1>(F1[1]); pF1[4] = lane_from<2>(F1[0]);
        pF2[0] = lane_from<-2>(F2[4]); pF2[1] = lane_from<-1>(F2[3]); pF2[2] = F2[2]; pF2[3] = lane_from<1>(F2[1]); pF2[4] = lane_from<2>(F2[0]);
    };

    // ---- prologue: all five rows of the first window are requested at once, but only b0-2 .. b0 are waited for: the
    //      two warm-up rows run while b0+1 and b0+2 are still in flight ----
    Blur bl;
    Row pr[5];
#pragma unroll
    for (int k = 0; k < 5; k++) row_load(pr[k], b0 - 2 + k, ~0u);
    blur_load(bl, b0, ~0u);
    // ring position for the first warm-up row (centre b0-2): colour slot of row b0-4+d is d, geometry slot of row b0-2+d is d
#pragma unroll
    for (int k = 0; k < 3; k++) row_store(pr[k], b0 - 2 + k, 2 + k, k);

    // ---- warm-up: rows b0-2 and b0-1 publish their forward terms (no output) ----
    auto warm = [&](int bw) {
        const v4f A = *reinterpret_cast<const v4f *>(Ap(0) + 2 * 16);
        const v2f Z = *reinterpret_cast<const v2f *>(Zp(0) + 2 * 8);
        float F1[5] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f }, F2[5];
#pragma unroll
        for (int j = 1; j <= 2; j++) {
            if (j == 1 && bw == b0 - 2) continue;          // (b0-2, b0-1) pairs are never consumed
            const char *ap = Ap(j), *zp = Zp(j);
#pragma unroll
            for (int i = 0; i < 5; i++) {
                const v4f Aq = *reinterpret_cast<const v4f *>(ap + i * 16);
                const v2f Zq = *reinterpret_cast<const v2f *>(zp + i * 8);
                const v2f s2 = geo(Aq, Zq, -A.xy, -A.zw, -Z);
                const float dn = fmaxf(__builtin_amdgcn_sqrtf(s2.x), 0.0f), dx = fmaxf(__builtin_amdgcn_sqrtf(s2.y), 0.0f);
                float t = fmaf(dn, kn, neg_log2_binom(i - 2) + neg_log2_binom(j));
                t = fmaf(dx, kx, t);
                if (j == 1) F1[i] = t; else F2[i] = t;
            }
        }
        publish(F1, F2);
    };
    warm(b0 - 2);
    row_store(pr[3], b0 + 1, 5, 3);                  // takes the slots of rows b0-4 (unused) and b0-2 (geometry done)
    ring_advance();
    warm(b0 - 1);
    row_store(pr[4], b0 + 2, 5, 3);
    ring_advance();
    stamp(1);
    // rows are requested TWO iterations before they enter the window (one iteration of slack proved too little when
    // HBM is busy: waves stalled at the end of an iteration and drifted apart by 15 %)
    Row ra, rb;                                          // two register sets, alternating (a copy would wait for the load)
    row_load(ra, b0 + 3, (b0 + 3 <= b1 + 1) ? ~0u : 0u);

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

    struct ColRow { v4f C[5]; float l[5]; };              // colour-only row: colour + luminance of the 5 taps
    struct GeoRow { v4f A[5]; v2f Z[5]; };                // forward row: geometry (colour + luminance are read mid-row)
    struct OwnRow { v4f A[2]; v2f Z[2]; v4f C[2], Cb[2]; float l[2], lf[2]; };   // own row: +1, +2 full; -1, -2 colour + luminance
    auto load_col = [&](ColRow &r, int dc) {
        const char *cp = Cp(dc), *lp = Lp(dc);
#pragma unroll
        for (int i = 0; i < 5; i++) {
            r.C[i] = *reinterpret_cast<const v4f *>(cp + i * 16);
            r.l[i] = *reinterpret_cast<const float *>(lp + i * 4);
        }
    };
    auto load_geo = [&](GeoRow &r, int dg) {
        const char *ap = Ap(dg), *zp = Zp(dg);
#pragma unroll
        for (int i = 0; i < 5; i++) {
            r.A[i] = *reinterpret_cast<const v4f *>(ap + i * 16);
            r.Z[i] = *reinterpret_cast<const v2f *>(zp + i * 8);
        }
    };
    auto load_own = [&](OwnRow &r) {
        const char *ap = Ap(0), *zp = Zp(0), *cp = Cp(2), *lp = Lp(2);
#pragma unroll
        for (int k = 0; k < 2; k++) {
            r.A[k] = *reinterpret_cast<const v4f *>(ap + (k + 3) * 16);
            r.Z[k] = *reinterpret_cast<const v2f *>(zp + (k + 3) * 8);
            r.C[k] = *reinterpret_cast<const v4f *>(cp + (k + 3) * 16);
            r.lf[k] = *reinterpret_cast<const float *>(lp + (k + 3) * 4);
            r.Cb[k] = *reinterpret_cast<const v4f *>(cp + (1 - k) * 16);
            r.l[k] = *reinterpret_cast<const float *>(lp + (1 - k) * 4);
        }
    };
    auto do_col = [&](Acc &acc, const ColRow &r, const float (&tt)[5], float lp, float kl) {
        float e[5], w[5];
#pragma unroll
        for (int i = 0; i < 5; i++) e[i] = fmaf(fabsf(r.l[i] - lp), kl, tt[i]);
        __builtin_amdgcn_sched_barrier(0x100);
#pragma unroll
        for (int i = 0; i < 5; i++) w[i] = __builtin_amdgcn_exp2f(-e[i]);
        __builtin_amdgcn_sched_barrier(0x100);
#pragma unroll
        for (int i = 0; i < 5; i++) accumulate(acc, r.C[i], w[i]);
    };
    auto do_geo = [&](Acc &acc, const GeoRow &r, int dc, auto jtag, float (&F)[5], const v2f &c0, const v2f &c1,
                      const v2f &c2, float lp, float kl) {
        constexpr int j = decltype(jtag)::value;
        const char *cp = Cp(dc), *lpp = Lp(dc);
        v2f s2[5];
        v4f Cq[5];
        float lq[5];
#pragma unroll
        for (int i = 0; i < 5; i++) s2[i] = geo(r.A[i], r.Z[i], c0, c1, c2);
#pragma unroll
        for (int i = 0; i < 5; i++) { Cq[i] = *reinterpret_cast<const v4f *>(cp + i * 16); lq[i] = *reinterpret_cast<const float *>(lpp + i * 4); }
        __builtin_amdgcn_sched_barrier(0x100);
        float dn[5], dx[5];
#pragma unroll
        for (int i = 0; i < 5; i++) {
            dn[i] = __builtin_amdgcn_sqrtf(s2[i].x);
            dx[i] = __builtin_amdgcn_sqrtf(s2[i].y);
        }
        __builtin_amdgcn_sched_barrier(0x100);
        float e[5], w[5];
#pragma unroll
        for (int i = 0; i < 5; i++) {
            float t = fmaf(dn[i], kn, neg_log2_binom(i - 2) + neg_log2_binom(j));
            t = fmaf(dx[i], kx, t);
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

    // end of a tap row: nothing of this row may sink below, no LDS read of a later row may rise above
    auto row_fence = [&](Acc &acc) {
        float a0 = acc.rg.x, a1 = acc.rg.y, a2 = acc.bv.x, a3 = acc.bv.y, a4 = acc.ww.x, a5 = acc.ww.y;
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : : "memory");
        acc.rg = v2f{a0, a1}; acc.bv = v2f{a2, a3}; acc.ww = v2f{a4, a5};
    };

    // nr: row bo+3, requested during the previous iteration, stored at the end of this one; nr2: receives row bo+4
    auto body = [&](int bo, Row &nr, Row &nr2) {
        const int y = yph + (bo << LOG2S);
        // The SIMD arbiter serves the oldest ready wave first: left alone, one wave of a SIMD runs ahead, finishes early and
        // leaves the others to run the rest of their segments with fewer partners.  Each wave publishes its row count and
        // takes the higher priority while it is not ahead of its SIMD partners (w +- 4, w +- 8).
        const int mine = bo - b0 + 1;
        prog_set(mine);
        const int p1 = prog_get(wv >= 2 * GW ? wv - 2 * GW : wv + GW), p2 = prog_get(wv >= GW ? wv - GW : wv + 2 * GW);
        const int theirs = __builtin_amdgcn_readfirstlane(min(p1 == 0 ? mine : p1, p2 == 0 ? mine : p2));
        if (theirs >= mine) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(0);

        // next row of the lane's own column and the pre-blur values of the next output row: in flight during this iteration
        const bool more_rows = (bo + 3 <= b1 + 1);               // the last output row b1-1 needs rows up to b1+1
        Blur nbl;                                                // (issued first: it is waited for first, vmcnt is in order)
        blur_load(nbl, bo + 1, (bo + 1 < b1) ? ~0u : 0u);
        row_load(nr2, bo + 4, (bo + 4 <= b1 + 1) ? ~0u : 0u);

        const v4f A = *reinterpret_cast<const v4f *>(Ap(0) + 2 * 16);
        const v2f Z = *reinterpret_cast<const v2f *>(Zp(0) + 2 * 8);
        const v4f C = *reinterpret_cast<const v4f *>(Cp(2) + 2 * 16);
        const float lp = *reinterpret_cast<const float *>(Lp(2) + 2 * 4);
        ColRow r0;
        load_col(r0, 0);                                         // the first tap row
        float var;
        {   // centre variance: 3x3 gaussian with out-of-image taps dropped and renormalised (:102-118)
            const float wr_m = (y - 1 >= 0) ? 0.25f : 0.0f, wr_p = (y + 1 < H) ? 0.25f : 0.0f;
            const float wc_l = (x - 1 >= 0) ? 0.25f : 0.0f, wc_r = (x + 1 < W) ? 0.25f : 0.0f;
            const float col_l = wr_m * bl.m.x + 0.5f * bl.c.x + wr_p * bl.p.x;
            const float col_c = wr_m * bl.m.y + 0.5f * C.w + wr_p * bl.p.y;
            const float col_r = wr_m * bl.m.z + 0.5f * bl.c.z + wr_p * bl.p.z;
            const float sum = wc_l * col_l + 0.5f * col_c + wc_r * col_r;
            const float sumw = (wr_m + 0.5f + wr_p) * (wc_l + 0.5f + wc_r);
            const float blurred = sum * __builtin_amdgcn_rcpf(sumw);
            var = a.blur_variance ? blurred : C.w;
        }
        var = fmaxf(var, 0.0f);
        const float kl = kLog2e * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(var) * sigma_c + 1e-6f);
        const v2f c0 = v2f{-A.x, -A.y}, c1 = v2f{-A.z, -A.w}, c2 = -Z;      // negated centre, see geo()

        // centre tap: weight exactly h = 9/64
        constexpr float w0 = 0.140625f;
        Acc acc;
        acc.ww = v2f{w0, w0 * w0};
        acc.rg = v2f{w0 * C.x, w0 * C.y};
        acc.bv = v2f{w0 * C.z, (w0 * w0) * C.w};

        if (careful) {
            // a non-finite normal / position is in the window: plain 24-tap loop keeping the reference's
            // min(1, exp(-NaN)) == 1 (fmaxf drops the NaN distance), nothing shared; the queue is not maintained
            // because the flag never clears inside a segment.  The ring keeps geometry for rows b .. b+2 only, so this
            // (rare, slow) path reads the partners' normals and positions from the planes.
            acc.rg = v2f{0.0f, 0.0f}; acc.bv = v2f{0.0f, 0.0f}; acc.ww = v2f{0.0f, 0.0f};
#pragma unroll 1
            for (int j = -2; j <= 2; j++) {
                const char *cp = Cp(j + 2), *lpp = Lp(j + 2);
#pragma unroll 1
                for (int i = -2; i <= 2; i++) {
                    const int xq = x + i * S, yq = y + j * S;
                    const unsigned q = (unsigned)min(max(yq, 0), H - 1) * (unsigned)W + (unsigned)min(max(xq, 0), W - 1);
                    const float *nq = a.nrm + 3 * (size_t)q, *pq = a.pos + 3 * (size_t)q;
                    const v4f Aq = v4f{nq[0], pq[0], nq[1], pq[1]};
                    const v2f Zq = v2f{nq[2], pq[2]};
                    const v4f Cq = *reinterpret_cast<const v4f *>(cp + (i + 2) * 16);
                    const float lq = *reinterpret_cast<const float *>(lpp + (i + 2) * 4);
                    const v2f s2 = geo(Aq, Zq, c0, c1, c2);
                    const float dn = fmaxf(__builtin_amdgcn_sqrtf(s2.x), 0.0f), dx = fmaxf(__builtin_amdgcn_sqrtf(s2.y), 0.0f);
                    const int ai = i < 0 ? -i : i, aj = j < 0 ? -j : j;
                    const float nl = (ai == 0 ? 1.4150374992788437f : (ai == 1 ? 2.0f : 4.0f)) + (aj == 0 ? 1.4150374992788437f : (aj == 1 ? 2.0f : 4.0f));
                    float e = fmaf(fabsf(lq - lp), kl, nl);
                    e = fmaf(dn, kn, e);
                    e = fmaf(dx, kx, e);
                    accumulate(acc, Cq, __builtin_amdgcn_exp2f(-e));
                }
            }
        } else {
        auto do_own = [&](const OwnRow &r2) {
            float e[4], tf[2];
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const v2f s2 = geo(r2.A[k], r2.Z[k], c0, c1, c2);
                const float dn = __builtin_amdgcn_sqrtf(s2.x), dx = __builtin_amdgcn_sqrtf(s2.y);
                const float t = fmaf(dn, kn, neg_log2_binom(k + 1) + neg_log2_binom(0));
                tf[k] = fmaf(dx, kx, t);
            }
            const float tb1 = lane_from<-1>(tf[0]), tb2 = lane_from<-2>(tf[1]);
            e[0] = fmaf(fabsf(r2.l[1] - lp), kl, tb2);          // io = -2
            e[1] = fmaf(fabsf(r2.l[0] - lp), kl, tb1);          // io = -1
            e[2] = fmaf(fabsf(r2.lf[0] - lp), kl, tf[0]);       // io = +1
            e[3] = fmaf(fabsf(r2.lf[1] - lp), kl, tf[1]);       // io = +2
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
        // backward rows (colour only, terms from the queue), own row, forward rows (evaluate, use, keep for the partners)
        ColRow r1;
        load_col(r1, 1);
        do_col(acc, r0, ppF2, lp, kl);
        OwnRow r2;
        load_own(r2);
        row_fence(acc);
        do_col(acc, r1, pF1, lp, kl);
        GeoRow g1;
        load_geo(g1, 1);
        row_fence(acc);
        do_own(r2);
        row_fence(acc);
        do_geo(acc, g1, 3, std::integral_constant<int, 1>{}, F1, c0, c1, c2, lp, kl);
        row_fence(acc);
        GeoRow g2;
        load_geo(g2, 2);
        do_geo(acc, g2, 4, std::integral_constant<int, 2>{}, F2, c0, c1, c2, lp, kl);
        row_fence(acc);
        publish(F1, F2);
        }

        // The row loaded at the top becomes ring row bo+3: its colour takes the slot of row bo-2, its geometry the slot of
        // row bo (both have left the window; the wave's LDS operations are ordered).  This sits in front of the output
        // stores on purpose: vmcnt counts stores too, and a wait for these loads placed behind the stores (or carried
        // over the loop edge) would wait for the stores' acknowledgements as well.
        if (more_rows) row_store(nr, bo + 3, 5, 3);
        asm volatile("" : "+v"(nbl.m), "+v"(nbl.c), "+v"(nbl.p));
        bl = nbl;

        if (out_lane) {
            const float r0v = acc.rg.x, r1v = acc.rg.y, r2v = acc.bv.x, vsum = acc.bv.y, wsum = acc.ww.x, w2sum = acc.ww.y;
            float o0, o1, o2, ov;
            if (wsum > 1e-5f) {                                     // NaN -> false -> pass-through (:159-164)
                const float rw = __builtin_amdgcn_rcpf(wsum);
                o0 = r0v * rw; o1 = r1v * rw; o2 = r2v * rw;
                ov = HASVAR ? vsum * __builtin_amdgcn_rcpf(w2sum) : 0.0f;
            } else {
                o0 = C.x; o1 = C.y; o2 = C.z; ov = C.w;
            }
            const unsigned p = (unsigned)y * (unsigned)W + (unsigned)x;
            if (a.modulate) {                                      // last level: * albedo * ialbedo (:166-168)
                const float *t = a.gbuf + 13u * (size_t)p;
                o0 *= t[6] * t[9]; o1 *= t[7] * t[10]; o2 *= t[8] * t[11];
            }
            if (a.dst) a.dst[p] = make_float4(o0, o1, o2, ov);
            if (a.var_dst) a.var_dst[(unsigned)(y + 1) * vpitch + (unsigned)(x + 1)] = ov;
            if (a.out_rgb) { float *o = a.out_rgb + 3u * p; o[0] = o0; o[1] = o1; o[2] = o2; }
        }
    };

    for (int bo = b0; bo < b1; bo += 2) {
        body(bo, ra, rb);
        ring_advance();
        if (bo + 1 < b1) {
            body(bo + 1, rb, ra);
            ring_advance();
        }
    }
    prog_set(0x7fffffff);
    stamp(2);
}

struct PwPlan { int n_strips, n_pg, seg_rows, n_segs, n_groups, nblocks; };

template <int S>
PwPlan pw_plan(int W, int H, int n_cu)
{
    constexpr int PH = S < GW ? S : GW, CB = GW / PH;
    PwPlan p;
    const int mx = (W + S - 1) / S;                                     // lattice columns of the widest x-phase
    const int n_cbg = (mx + LOUT * CB - 1) / (LOUT * CB);
    p.n_pg = S / PH;
    p.n_strips = p.n_pg * n_cbg;
    const int nb_max = (H + S - 1) / S;
    // segment length: one 12-wave workgroup per CU; the busiest XCD sets the number of rounds.  A segment costs its rows
    // plus ~3 rows' worth of prologue and warm-up.
    int best_L = nb_max;
    long best_cost = -1;
    for (int L = 3; L <= nb_max + 1; L++) {
        const int segs_l = (nb_max + L - 1) / L;
        const int trips = (segs_l + NSG - 1) / NSG;
        const long blocks_xcd = (long)p.n_strips * ((S * trips + 7) / 8);
        const long rounds = (blocks_xcd + n_cu / 8 - 1) / (n_cu / 8);
        const long cost = rounds * (L + 3);
        if (best_cost < 0 || cost <= best_cost) { best_cost = cost; best_L = L; }
    }
    p.seg_rows = best_L;
    p.n_segs = (nb_max + best_L - 1) / best_L;
    p.n_groups = S * ((p.n_segs + NSG - 1) / NSG);
    p.nblocks = (p.n_groups + 7) / 8 * 8 * p.n_strips;
    return p;
}

template <int LOG2S, bool HASVAR>
hipError_t launch_pw_cfg(const AtrousArgs &a, hipStream_t s)
{
    constexpr int S = 1 << LOG2S;
    const size_t lds = (size_t)NW * RINGB;
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    if (dev_id < 0 || dev_id >= 64) dev_id = 0;
    static std::once_flag attr_once[64];
    static hipError_t attr_err[64];
    static int n_cu_dev[64];
    std::call_once(attr_once[dev_id], [&]() {
        attr_err[dev_id] = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_atrous_pw<LOG2S, HASVAR>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev_id) != hipSuccess || n <= 0) n = 256;
        n_cu_dev[dev_id] = n;
    });
    if (attr_err[dev_id] != hipSuccess) return attr_err[dev_id];
    const PwPlan pl = pw_plan<S>(a.W, a.H, n_cu_dev[dev_id]);
    PwGeom gm;
    gm.n_pg = pl.n_pg; gm.n_strips = pl.n_strips; gm.seg_rows = pl.seg_rows; gm.n_segs = pl.n_segs; gm.n_groups = pl.n_groups;
    gm.kn = (float)(1.4426950408889634 / ((double)a.sigma_n + 1e-6));
    gm.kx = (float)(1.4426950408889634 / ((double)a.sigma_x + 1e-6));
    const int nblocks = pl.nblocks;
    gm.dbg = nullptr;
#ifdef SVGF_PW_TIMELINE
    static unsigned long long *dbg_buf = nullptr;
    const bool dbg_on = getenv("SVGF_PW_DBG") != nullptr;
    const size_t dbg_n = (size_t)nblocks * NW * 4;
    if (dbg_on) {
        if (!dbg_buf) (void)hipMalloc((void **)&dbg_buf, 8192 * NW * 4 * sizeof(unsigned long long));
        (void)hipMemsetAsync(dbg_buf, 0, dbg_n * sizeof(unsigned long long), s);
        gm.dbg = dbg_buf;
    }
#endif
    hipLaunchKernelGGL((k_atrous_pw<LOG2S, HASVAR>), dim3(nblocks), dim3(NW * 64), lds, s, a, gm);
#ifdef SVGF_PW_TIMELINE
    static int prints = 0;
    if (dbg_on && prints < 12) {
        (void)hipStreamSynchronize(s);
        std::vector<unsigned long long> h(dbg_n);
        (void)hipMemcpy(h.data(), dbg_buf, dbg_n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::vector<double> st, pe, en, du;
        unsigned long long t0 = ~0ull;                                   // s_memtime bases differ across the chip: align on s_memrealtime
        for (size_t k = 0; k < dbg_n; k += 4) if (h[k] && (h[k + 3] >> 16) < t0) t0 = h[k + 3] >> 16;
        for (size_t k = 0; k < dbg_n; k += 4) if (h[k] && h[k + 2]) {
            const double s0 = (double)((h[k + 3] >> 16) - t0) * 24.0;    // 10 ns ticks -> 2.4 GHz cycles
            st.push_back(s0); pe.push_back(s0 + (double)(h[k + 1] - h[k])); en.push_back(s0 + (double)(h[k + 2] - h[k])); du.push_back((double)(h[k + 2] - h[k]));
        }
        auto q = [](std::vector<double> v, double f) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[(size_t)(f * (v.size() - 1))]; };
        if (prints++ >= 6) {
            fprintf(stderr, "[pw dbg] S=%d blocks=%d seg_rows=%d n_segs=%d waves=%zu | start min/med/max %.0f %.0f %.0f | prologue done %.0f %.0f %.0f | end %.0f %.0f %.0f | life %.0f %.0f %.0f ticks\n",
                    S, nblocks, gm.seg_rows, gm.n_segs, st.size(), q(st, 0), q(st, .5), q(st, 1), q(pe, 0), q(pe, .5), q(pe, 1), q(en, 0), q(en, .5), q(en, 1), q(du, 0), q(du, .5), q(du, 1));
            if (prints == 8)
                for (int b = 0; b < nblocks; b += nblocks / 8 > 0 ? nblocks / 8 : 1) {
                    const unsigned long long *w0 = &h[(size_t)b * NW * 4];
                    if (!w0[0]) continue;
                    fprintf(stderr, "  wg %4d life:", b);
                    for (int w = 0; w < NW; w++) fprintf(stderr, " %6llu", w0[w * 4 + 2] - w0[w * 4]);
                    fprintf(stderr, "  simd:");
                    for (int w = 0; w < NW; w++) fprintf(stderr, " %llu", (w0[w * 4 + 3] >> 4) & 3);
                    fprintf(stderr, "  prologue:");
                    for (int w = 0; w < NW; w++) fprintf(stderr, " %5llu", w0[w * 4 + 1] - w0[w * 4]);
                    fprintf(stderr, "\n");
                }
        }
    }
#endif
    return hipGetLastError();
}

}  // namespace

bool atrous_pw_supported(const AtrousArgs &a)
{
    if (a.step != 1 && a.step != 2 && a.step != 4 && a.step != 8 && a.step != 16 && a.step != 32) return false;
    if ((long long)a.W * a.H * 16 >= (1LL << 32)) return false;
    return true;
}

hipError_t launch_atrous_pw(const AtrousArgs &a, hipStream_t s)
{
    switch (a.step) {
    case 1: return a.dst ? launch_pw_cfg<0, true>(a, s) : launch_pw_cfg<0, false>(a, s);
    case 2: return a.dst ? launch_pw_cfg<1, true>(a, s) : launch_pw_cfg<1, false>(a, s);
    case 4: return a.dst ? launch_pw_cfg<2, true>(a, s) : launch_pw_cfg<2, false>(a, s);
    case 8: return a.dst ? launch_pw_cfg<3, true>(a, s) : launch_pw_cfg<3, false>(a, s);
    case 16: return a.dst ? launch_pw_cfg<4, true>(a, s) : launch_pw_cfg<4, false>(a, s);
    case 32: return a.dst ? launch_pw_cfg<5, true>(a, s) : launch_pw_cfg<5, false>(a, s);
    default: return hipErrorInvalidValue;
    }
}
