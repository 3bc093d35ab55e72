// svgf_atrous_dma.hip — a-trous level with TWELVE computing waves per CU fed by LDS-DMA (gfx950), steps 2 .. 16.
//
// Same result as svgf_atrous_strip.hip (one level of reference ATrousFilter, src/denoise.cu:77-170, snapshot variance) and
// the same decomposition: a workgroup owns TX contiguous columns of ONE y-phase and marches down that phase's lattice rows,
// ROWS rows per iteration, with the last 4+ROWS of them (and the ROWS incoming ones) in an LDS ring.  What changes is who
// fills the ring and how many waves compute.
//
// Why.  On gfx950 a wave issues at most one instruction every ~5 cycles whatever its kind, while a plain VALU instruction
// occupies the SIMD for 2 (packed: ~3-4.5, transcendental: ~6-8): two computing waves per SIMD leave a third of the VALU
// idle, three fill it (tools/ubench6.hip, profiles/r03_ubench6_cycles_per_inst.log: per-wave cost of an instruction is the
// same with 1, 2 or 3 waves on the SIMD and grows only with the fourth).  The strip / lane kernels run 2 compute waves + 1
// mostly parked loader wave per SIMD (the loaders need VGPRs for staging, so a fourth wave does not fit).  Here staging needs
// NO registers and almost no instructions: every wave issues a dozen `global_load_lds` per iteration — the data goes from
// L2/HBM straight into the ring — so all 12 waves of the workgroup (3 per SIMD, <= 168 VGPRs) evaluate taps.
//
// What DMA can deliver.  One `global_load_lds_dword[x4]` moves 4 (16) bytes per lane from a PER-LANE global address to
// LDS[M0 + 4 (16) * lane]: linear in LDS, arbitrary in memory.  The ring keeps the strip kernel's three 16-byte records
//     A = {n.x, p.x, n.y, p.y}   B = {n.z, p.z, luminance, luminance}   C = {r, g, b, variance}
// as three arrays per row (structure of arrays, 16-byte lane stride: every ds_read_b128 is conflict-free):
//     C: dwordx4, one pixel per lane, a straight copy of 64 pixels of the colour plane;
//     A, B: dword gathers, four lanes per pixel, interleaving the packed-float3 normal / position planes (and, for B, the
//           4-byte luminance plane the producer of the colour plane wrote next to it) on the fly: 16 pixels per instruction.
// Nobody inspects the staged values, so what the loaders of the other kernels did in registers moves elsewhere:
//     * luminance (double products, rounded once: src/denoise.cu:121,138) is computed ONCE per pixel by the kernel that
//       produces the colour plane (temporal / prepare pass, or this kernel's output stage) and stored in a 4-byte plane;
//     * non-finite normals / positions are detected by the temporal / prepare pass (which reads every texel anyway) and
//       raise a per-frame flag: with the flag up every workgroup takes the CAREFUL tap routine (min(1, exp(-NaN)) == 1);
//     * out-of-image pixels (the reference skips those taps, :134) must be staged as {luminance = +inf, colour = 0}:
//       lanes whose pixel lies outside the image fetch from a constant page instead (a per-lane select on the 32-bit
//       offset; the pages and all planes live in ONE allocation so that 32-bit offsets reach them).
// Requirements checked by atrous_dma_supported(): a luminance plane for the source, W % 4 == 0, the arena below 4 GiB.
#include "svgf_kernels.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace {

constexpr float kLog2e = 1.44269504088896340736f;

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

struct DmaGeom {
    int n_strips, n_segs, seg_rows, n_groups;
    float kn, kx;
    unsigned pos_minus_nrm;     // byte distance between the position and the normal plane (same arena)
    unsigned lum_minus_nrm;     // byte distance between the source's luminance plane and the normal plane
};

__device__ __forceinline__ constexpr float neg_log2_binom(int i)
{   // -log2 of the 5-tap binomial [1 4 6 4 1]/16
    return (i == 0) ? 1.4150374992788437f : ((i == 1 || i == -1) ? 2.0f : 4.0f);
}

// One LDS-DMA instruction: lane l copies 16 (4) bytes from sbase + voff[l] + IMM to LDS[lds + 16 (4) * l].
// Hidden from the compiler on purpose: its waitcnt pass would put a vmcnt(0) in front of the next ds_read (it cannot tell
// which LDS bytes a DMA writes); the kernel waits once per iteration itself (dma_wait()).
template <int LDS_OFF, int IMM>
__device__ __forceinline__ void dma_x4(unsigned lds, const void *sbase, unsigned voff)
{
    static_assert(IMM >= -4096 && IMM < 4096, "global instruction offset field");
    asm volatile("s_add_u32 m0, %0, %4\n\tglobal_load_lds_dwordx4 %1, %2 offset:%3" ::"s"(lds), "v"(voff), "s"(sbase), "n"(IMM), "n"(LDS_OFF) : "memory", "scc");
}
template <int LDS_OFF, int IMM>
__device__ __forceinline__ void dma_x1(unsigned lds, const void *sbase, unsigned voff)
{
    static_assert(IMM >= -4096 && IMM < 4096, "global instruction offset field");
    asm volatile("s_add_u32 m0, %0, %4\n\tglobal_load_lds_dword %1, %2 offset:%3" ::"s"(lds), "v"(voff), "s"(sbase), "n"(IMM), "n"(LDS_OFF) : "memory", "scc");
}
// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <typename T>
__device__ __forceinline__ T uniform(T v)
{   // tell the compiler a value is wave-uniform (kept in an SGPR)
    static_assert(sizeof(T) == 4, "32-bit values");
    return __builtin_bit_cast(T, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

struct Centre { v2f nx_px, ny_py, nz_pz; float lp, kl, kn, kx; };
struct Acc { v2f rg, bv, ww; };

template <bool HASVAR>
__device__ __forceinline__ void accumulate(Acc &acc, const v4f &C, float w)
{
    if (HASVAR) {
        v2f wv;
        wv.x = w;
        wv.y = w * w;
        acc.ww += wv;
        acc.rg = __builtin_elementwise_fma(C.xy, v2f{w, w}, acc.rg);
        acc.bv = __builtin_elementwise_fma(C.zw, wv, acc.bv);
    } else {
        acc.ww.x += w;
        acc.rg = __builtin_elementwise_fma(C.xy, v2f{w, w}, acc.rg);
        acc.bv.x = fmaf(C.z, w, acc.bv.x);
    }
}

template <int LOG2S, int TX, int ROWS, bool HASVAR>
__global__ __launch_bounds__(TX * ROWS) void k_atrous_dma(AtrousArgs a, DmaGeom gm)
{
    constexpr int S = 1 << LOG2S;
    constexpr int RW = TX + 4 * S;             // staged pixels per lattice row
    constexpr int R = 4 + 2 * ROWS;            // ring slots: 4 + ROWS live, ROWS incoming
    constexpr int NT = TX * ROWS;
    constexpr int NW = NT / 64;                // 12 waves
    constexpr int WPR = NW / ROWS;             // waves per row group
    constexpr int BW = TX + 2;                 // pre-blur row: columns x0-1 .. x0+TX
    constexpr int ARR = RW * 16;               // bytes of one record array of a ring row
    constexpr int SLOT = 3 * ARR;              // A | B | C
    constexpr int RING_BYTES = R * SLOT;
    constexpr int BLUR_ROW = BW;               // floats per (output row, side)
    constexpr int BLUR_BUF = ROWS * 2 * BLUR_ROW;      // floats per iteration parity
    static_assert(NW * 64 == NT && WPR * ROWS == NW && TX % 64 == 0, "workgroup shape");
    constexpr int NPA = (RW + 15) / 16;        // A / B gather pieces per row (16 pixels each, the last one possibly partial)

    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS byte addresses (the DMA destination is an address, not a pointer)
    const unsigned lds_ring = (unsigned)(size_t)smem;           // = 0 for the only dynamic array of the kernel
    const unsigned lds_blur = lds_ring + RING_BYTES;
    const float *blur = reinterpret_cast<const float *>(smem + RING_BYTES);

    // every kernel argument in one batch of s_loads at entry (svgf_atrous_lane.hip has the story)
    asm volatile("" :: "s"(a.src), "s"(a.dst), "s"(a.out_rgb), "s"(a.nrm), "s"(a.gbuf), "s"(a.W), "s"(a.H), "s"(a.sigma_c), "s"(a.blur_variance),
                 "s"(a.modulate), "s"(a.lum), "s"(a.lum_dst), "s"(a.nan_flag), "s"(a.zero_page), "s"(a.inf_page), "s"(gm.n_strips), "s"(gm.n_segs),
                 "s"(gm.seg_rows), "s"(gm.n_groups), "s"(gm.kn), "s"(gm.kx), "s"(gm.pos_minus_nrm), "s"(gm.lum_minus_nrm));

    // ---- work item: (strip, y-phase, segment), as in the strip kernel ----
    const int bid = blockIdx.x;
    const int xcd = bid & 7, kb = bid >> 3;
    const int g = xcd + 8 * (kb / gm.n_strips);
    const int strip = kb % gm.n_strips;
    if (g >= gm.n_groups) return;
    const int phase = g / gm.n_segs, seg = g % gm.n_segs;
    const int W = a.W, H = a.H;
    if (phase >= H) return;
    const int nb = (H - phase + S - 1) >> LOG2S;
    const int b0 = seg * gm.seg_rows;
    const int b1 = min(b0 + gm.seg_rows, nb);
    if (b0 >= b1) return;
    const int x0 = strip * TX;
    const int xf = x0 - 2 * S;                 // image column of staged pixel 0

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = uniform(tid >> 6);
    const int wrow = wv / WPR;                 // row group of this wave: computes row bc + wrow, stages incoming row wrow
    const int q = wv % WPR;                    // which share of a row's DMA pieces this wave issues
    const bool careful = (*a.nan_flag != 0);   // raised by the temporal / prepare pass of this frame

    // per-lane source offsets of the four piece kinds (invariant)
    const unsigned voff_c = (unsigned)lane * 16u;                                                         // C / pre-blur: one pixel (its .w) per lane
    const unsigned voff_w = voff_c + 12u;
    const unsigned voff_a = (unsigned)(lane >> 2) * 12u + ((lane & 2) ? 4u : 0u) + ((lane & 1) ? gm.pos_minus_nrm : 0u);
    const unsigned voff_b0 = (unsigned)(lane >> 2) * 12u + 8u + ((lane & 1) ? gm.pos_minus_nrm : 0u);  // lanes 4m, 4m+1: n.z, p.z
    const bool lum_lane = (lane & 2) != 0;                                                                // lanes 4m+2, 4m+3: luminance

    // ring slot of lattice row br (>= b0-2): `ring_base` is the slot of row ring_bc - 2, advanced by ROWS per iteration
    int ring_base = 0, ring_bc = b0;
    auto slot_of = [&](int br) {
        int s = ring_base + (br - (ring_bc - 2));      // offset in [0, 3*ROWS+2) < 2R
        s -= (s >= R) ? R : 0;
        s -= (s >= R) ? R : 0;
        return s;
    };
    auto ring_advance = [&]() { ring_bc += ROWS; ring_base += ROWS; ring_base -= (ring_base >= R) ? R : 0; };

    // ---------------- staging: this wave's share (q of WPR) of the DMA pieces of one ring row / one output row's pre-blur ----
    // Ring row br -> slot.  Out-of-image ROWS and COLUMNS are redirected per lane: colour -> zero page, luminance -> +inf
    // page (weight exp2(-inf) = 0; 0 * 0 stays finite); normals / positions of such pixels are never looked at again
    // (their weight is 0 in both tap routines), so those lanes fetch whatever the clamped row holds there.
    auto stage_ring_row = [&](int br, auto qtag) {
        constexpr int Q = decltype(qtag)::value;
        const int y = phase + (br << LOG2S);
        const bool row_ok = (br >= 0) && (y < H);
        const int yc = min(max(y, 0), H - 1);
        const long long p0 = (long long)yc * W + xf;                         // pixel index of staged pixel 0 (may be < 0 in row 0)
        const char *sb_c = reinterpret_cast<const char *>(a.src) + p0 * 16;
        const char *sb_n = reinterpret_cast<const char *>(a.nrm) + p0 * 12;
        const unsigned slot = lds_ring + (unsigned)slot_of(br) * SLOT;
        const unsigned wlim = row_ok ? (unsigned)W : 0u;                   // column test bound: 0 sends every lane to the page
        // luminance lanes of the B gather read the 4-byte plane: offset from the NORMAL row base = (lum - nrm) + p0*4 - p0*12
        const unsigned lum_row = gm.lum_minus_nrm - (unsigned)(p0 * 8);     // mod 2^32: the planes share one < 4 GiB arena
        const unsigned inf_off = (unsigned)(reinterpret_cast<const char *>(a.inf_page) - sb_n);
        const unsigned zero_off = (unsigned)(reinterpret_cast<const char *>(a.zero_page) - sb_c);
        // A: 16 pixels per piece; pieces k = Q, Q + WPR, ... of the row are this wave's
#ifndef SVGF_DMA_EXP_NOGATHER
        static_for<0, (NPA + WPR - 1) / WPR>([&](auto it_) {
            constexpr int k = Q + WPR * decltype(it_)::value;
            if constexpr (k < NPA) {
                if (k * 16 + 16 <= RW || lane < (RW - k * 16) * 4) dma_x1<k * 256, k * 192>(slot, sb_n, voff_a);
            }
        });
        // B: n.z, p.z from the planes, luminance (twice) from its plane or +inf
        static_for<0, (NPA + WPR - 1) / WPR>([&](auto it_) {
            constexpr int k = (Q + 1) % WPR + WPR * decltype(it_)::value;          // shifted so that the shares of A and B do not stack
            if constexpr (k < NPA) {
                const int xs = xf + k * 16 + (lane >> 2);
                const unsigned vl = ((unsigned)xs < wlim) ? lum_row + (unsigned)(k * 16 + (lane >> 2)) * 4u : inf_off + (unsigned)(lane & 1) * 4u;
                if (k * 16 + 16 <= RW || lane < (RW - k * 16) * 4) dma_x1<ARR + k * 256, k * 192>(slot, sb_n, lum_lane ? vl - (unsigned)(k * 192) : voff_b0);
            }
        });
#endif
        // C: 64 pixels per piece, the last one partial (offsets -2048 .. 2048 around sb_c + 2048)
        static_for<0, ((RW + 63) / 64 + WPR - 1) / WPR>([&](auto it_) {
            constexpr int k = (Q + 2) % WPR + WPR * decltype(it_)::value;
            if constexpr (k < (RW + 63) / 64) {
                const int xs = xf + k * 64 + lane;
                const unsigned v = ((unsigned)xs < wlim) ? voff_c : zero_off + (unsigned)(lane & 3) * 16u - (unsigned)(k * 1024);
                if (k * 64 + 64 <= RW || lane < RW - k * 64) dma_x4<2 * ARR + k * 1024, k * 1024 - 2048>(slot, sb_c + 2048, v);
            }
        });
    };
    // pre-blur rows y-1 / y+1 of output row bo (3x3 variance blur, :102-118): the .w of 64 colour texels per piece, or 0
    auto stage_blur_row = [&](int bo, int rr, int parity, auto qtag) {
        constexpr int Q = decltype(qtag)::value;
        constexpr int NP = (BW + 63) / 64;
        const int yo = phase + (bo << LOG2S);
        const unsigned dst0 = lds_blur + (unsigned)((parity * BLUR_BUF + rr * 2 * BLUR_ROW) * 4);
        static_for<0, (2 * NP + WPR - 1) / WPR>([&](auto it_) {
            constexpr int k2 = (Q + 3) % WPR + WPR * decltype(it_)::value;
            if constexpr (k2 < 2 * NP) {
                constexpr int d = k2 / NP, k = k2 % NP;                     // side (0: y-1, 1: y+1), piece
                const int y = yo + (d ? 1 : -1);
                const bool row_ok = (y >= 0) && (y < H) && (bo < b1);
                const int yc = min(max(y, 0), H - 1);
                const long long p0 = (long long)yc * W + (x0 - 1);
                const char *sb = reinterpret_cast<const char *>(a.src) + p0 * 16;
                const unsigned zero_off = (unsigned)(reinterpret_cast<const char *>(a.zero_page) - sb);
                const int xs = x0 - 1 + k * 64 + lane;
                const unsigned v = (row_ok && (unsigned)xs < (unsigned)W) ? voff_w : zero_off + (unsigned)(lane & 3) * 4u - (unsigned)(k * 1024);
                if (k * 64 + 64 <= BW || lane < BW - k * 64) dma_x1<(d * BLUR_ROW + k * 64) * 4, k * 1024 - 2048>(dst0, sb + 2048, v);
            }
        });
    };
    auto with_q = [&](auto &&f) {       // run f with this wave's share index as a compile-time constant
        if constexpr (WPR == 4) {
            if (q == 0) f(std::integral_constant<int, 0>{}); else if (q == 1) f(std::integral_constant<int, 1>{});
            else if (q == 2) f(std::integral_constant<int, 2>{}); else f(std::integral_constant<int, 3>{});
        } else {
            static_assert(WPR == 2, "waves per row group");
            if (q == 0) f(std::integral_constant<int, 0>{}); else f(std::integral_constant<int, 1>{});
        }
    };

    // ---------------- prologue: rows b0-2 .. b0+ROWS+1 and the pre-blur rows of iteration 0 ----------------
    with_q([&](auto qt) {
        for (int pr = wrow; pr < 4 + ROWS; pr += ROWS) stage_ring_row(b0 - 2 + pr, qt);
        if (a.blur_variance) stage_blur_row(b0 + wrow, wrow, 0, qt);
    });
    float sigma_c = a.sigma_c;
    asm volatile("" : "+s"(sigma_c));
    dma_wait();
    __syncthreads();

    // ================================ all waves compute ================================
    const int tx = tid - wrow * TX;
    const int x = x0 + tx;
    const float kn = gm.kn, kx = gm.kx;
    const char *colbase = smem + (size_t)tx * 16;

    int it = 0;
    for (int bc = b0; bc < b1; bc += ROWS, it++) {
        // this wave's share of the rows the NEXT iteration newly needs: in flight during the whole iteration
#ifndef SVGF_DMA_EXP_NOSTAGE        // experiments only (tools/experiments/exp_dma_split.sh): wrong results, timing of the rest
        with_q([&](auto qt) {
            stage_ring_row(bc + ROWS + 2 + wrow, qt);
            if (a.blur_variance) stage_blur_row(bc + ROWS + wrow, wrow, (it + 1) & 1, qt);
        });
#endif
        __builtin_amdgcn_s_setprio(3);
        const int bo = bc + wrow;
        const bool active = (bo < b1) && (x < W);
        float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f, ov = 0.0f;
        const int y = phase + (bo << LOG2S);
        if (active) {
            const char *rowc = colbase + (size_t)slot_of(bo) * SLOT + (size_t)(2 * S) * 16;
            const v4f A = *reinterpret_cast<const v4f *>(rowc);
            const v4f B = *reinterpret_cast<const v4f *>(rowc + ARR);
            const v4f C = *reinterpret_cast<const v4f *>(rowc + 2 * ARR);
            const float *bl = blur + (it & 1) * BLUR_BUF + wrow * (2 * BLUR_ROW) + tx;     // column x-1 of row y-1
            const float m0 = bl[0], m1 = bl[1], m2 = bl[2];
            const float p0 = bl[BLUR_ROW], p1 = bl[BLUR_ROW + 1], p2 = bl[BLUR_ROW + 2];
            const float c0v = *reinterpret_cast<const float *>(rowc + 2 * ARR - 16 + 12);
            const float c2v = *reinterpret_cast<const float *>(rowc + 2 * ARR + 16 + 12);
            float var;
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
            var = fmaxf(var, 0.0f);
            Centre c;
            c.nx_px = v2f{A.x, A.y}; c.ny_py = v2f{A.z, A.w}; c.nz_pz = v2f{B.x, B.y};
            c.lp = B.z;
            c.kl = kLog2e * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(var) * sigma_c + 1e-6f);
            c.kn = kn; c.kx = kx;

            Acc acc;
            if (!careful) {
                constexpr float w0 = 0.140625f;                  // centre tap: weight exactly h = 9/64
                acc.ww = v2f{w0, w0 * w0};
                acc.rg = v2f{w0 * C.x, w0 * C.y};
                acc.bv = v2f{w0 * C.z, (w0 * w0) * C.w};
                // a tap row (5 taps) in stages, so that the five dependency chains interleave and the transcendentals issue
                // back to back; the geometry records of row j+1 are read while row j is evaluated (svgf_atrous_strip.hip)
                v4f Ac[5], Bc[5], An[5], Bn[5];
                {
                    const char *rowp = colbase + (size_t)slot_of(bo - 2) * SLOT;
#pragma unroll
                    for (int i = 0; i < 5; i++) {
                        Ac[i] = *reinterpret_cast<const v4f *>(rowp + i * S * 16);
                        Bc[i] = *reinterpret_cast<const v4f *>(rowp + ARR + i * S * 16);
                    }
                }
#pragma unroll
                for (int j = -2; j <= 2; j++) {
                    const char *rowp = colbase + (size_t)slot_of(bo + j) * SLOT;
                    if (j < 2) {
                        const char *rown = colbase + (size_t)slot_of(bo + j + 1) * SLOT;
#pragma unroll
                        for (int i = 0; i < 5; i++) {
                            An[i] = *reinterpret_cast<const v4f *>(rown + i * S * 16);
                            Bn[i] = *reinterpret_cast<const v4f *>(rown + ARR + i * S * 16);
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
                        if (!(i == 2 && j == 0)) Cc[i] = *reinterpret_cast<const v4f *>(rowp + 2 * ARR + i * S * 16);
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
                        float t = fmaf(fabsf(dl[i]), c.kl, neg_log2_binom(i - 2) + neg_log2_binom(j));
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
                    for (int i = 0; i < 5; i++)
                        if (!(i == 2 && j == 0)) accumulate<HASVAR>(acc, Cc[i], w[i]);
                    if (j < 2) {
#pragma unroll
                        for (int i = 0; i < 5; i++) { Ac[i] = An[i]; Bc[i] = Bn[i]; }
                    }
                    // whichever wave of a SIMD is behind outranks the ones ahead (svgf_atrous_strip.hip)
                    if (j == -1) __builtin_amdgcn_s_setprio(2);
                    if (j == 0) __builtin_amdgcn_s_setprio(1);
                    if (j == 1) __builtin_amdgcn_s_setprio(0);
                }
            } else {
                // min(1, exp(-NaN)) == 1 in the reference: a NaN distance contributes nothing to the exponent
                acc.rg = v2f{0.0f, 0.0f}; acc.bv = v2f{0.0f, 0.0f}; acc.ww = v2f{0.0f, 0.0f};
#pragma unroll 1
                for (int j = -2; j <= 2; j++) {
                    const char *rowp = colbase + (size_t)slot_of(bo + j) * SLOT;
#pragma unroll 1
                    for (int i = -2; i <= 2; i++) {
                        const v4f Aq = *reinterpret_cast<const v4f *>(rowp + (i + 2) * S * 16);
                        const v4f Bq = *reinterpret_cast<const v4f *>(rowp + ARR + (i + 2) * S * 16);
                        const v4f Cq = *reinterpret_cast<const v4f *>(rowp + 2 * ARR + (i + 2) * S * 16);
                        const v2f d0 = Aq.xy - c.nx_px, d1 = Aq.zw - c.ny_py, d2 = Bq.xy - c.nz_pz;
                        v2f s = d0 * d0;
                        s = __builtin_elementwise_fma(d1, d1, s);
                        s = __builtin_elementwise_fma(d2, d2, s);
                        const float dn = fmaxf(__builtin_amdgcn_sqrtf(s.x), 0.0f), dx = fmaxf(__builtin_amdgcn_sqrtf(s.y), 0.0f);
                        const int ai = i < 0 ? -i : i, aj = j < 0 ? -j : j;
                        const float nl = (ai == 0 ? 1.4150374992788437f : (ai == 1 ? 2.0f : 4.0f)) + (aj == 0 ? 1.4150374992788437f : (aj == 1 ? 2.0f : 4.0f));
                        float e = fmaf(fabsf(Bq.z - c.lp), c.kl, nl);
                        e = fmaf(dn, c.kn, e);
                        e = fmaf(dx, c.kx, e);
                        accumulate<HASVAR>(acc, Cq, __builtin_amdgcn_exp2f(-e));
                    }
                }
            }
            const float r0 = acc.rg.x, r1 = acc.rg.y, r2 = acc.bv.x, vsum = acc.bv.y, wsum = acc.ww.x, w2sum = acc.ww.y;
            if (wsum > 1e-5f) {                                     // NaN -> false -> pass-through (:159-164)
                const float rw = __builtin_amdgcn_rcpf(wsum);
                o0 = r0 * rw; o1 = r1 * rw; o2 = r2 * rw;
                ov = HASVAR ? vsum * __builtin_amdgcn_rcpf(w2sum) : 0.0f;
            } else {
                o0 = C.x; o1 = C.y; o2 = C.z; ov = C.w;
            }
        }
        // the rows requested at the top of this iteration have landed long ago; waiting for them HERE, in front of the
        // output stores, keeps the wait from also covering the stores' round trip (vmcnt counts both)
        dma_wait();
        if (active) {
            const unsigned p = (unsigned)y * (unsigned)W + (unsigned)x;
            if (a.lum_dst) a.lum_dst[p] = svgf_lum_strict(o0, o1, o2);     // what the next level stages as this pixel's luminance
            if (a.modulate) {                                      // last level: * albedo * ialbedo (:166-168)
                const float *t = a.gbuf + 13u * (size_t)p;
                o0 *= t[6] * t[9]; o1 *= t[7] * t[10]; o2 *= t[8] * t[11];
            }
            if (a.dst) a.dst[p] = make_float4(o0, o1, o2, ov);
            if (a.var_dst) a.var_dst[(unsigned)(y + 1) * (unsigned)(W + 2) + (unsigned)(x + 1)] = ov;
            if (a.out_rgb) { float *o = a.out_rgb + 3u * p; o[0] = o0; o[1] = o1; o[2] = o2; }
        }
        __builtin_amdgcn_s_barrier();         // LDS only: the DMA writes are complete (dma_wait), nobody needs the global stores yet
        ring_advance();
    }
}

template <int LOG2S, int TX, int ROWS, bool HASVAR>
hipError_t launch_dma_cfg(const AtrousArgs &a, hipStream_t s)
{
    constexpr int S = 1 << LOG2S, RW = TX + 4 * S, R = 4 + 2 * ROWS, BW = TX + 2;
    const size_t lds = (size_t)R * RW * 48 + (size_t)2 * ROWS * 2 * BW * 4;
    static_assert((size_t)R * RW * 48 + (size_t)2 * ROWS * 2 * BW * 4 <= 160 * 1024, "LDS budget");
    static SvgfLaunchCache cache;
    int dev_id = 0;
    if (hipError_t e = cache.init(reinterpret_cast<const void *>(&k_atrous_dma<LOG2S, TX, ROWS, HASVAR>), (int)lds, &dev_id); e != hipSuccess) return e;
    const int n_cu = cache.n_cu[dev_id];
    DmaGeom gm;
    gm.n_strips = (a.W + TX - 1) / TX;
    const int nb_max = (a.H + S - 1) / S;
    // segment length: one workgroup per CU (LDS), rounds x (rows + fixed cost) minimised as in the strip kernel
    int best_L = nb_max;
    long best_cost = -1;
    for (int L = ROWS * 2; L <= nb_max + ROWS; L++) {
        const int segs_l = (nb_max + L - 1) / L;
        const long blocks_xcd = (long)gm.n_strips * ((S * segs_l + 7) / 8);
        const long rounds = (blocks_xcd + n_cu / 8 - 1) / (n_cu / 8);
        const long cost = rounds * ((L + ROWS - 1) / ROWS * ROWS + 8);
        if (best_cost < 0 || cost <= best_cost) { best_cost = cost; best_L = L; }
    }
    if (const char *e = getenv("SVGF_DMA_SEGROWS")) { int v = atoi(e); if (v > 0) best_L = v; }     // tuning only
    gm.seg_rows = best_L;
    gm.n_segs = (nb_max + best_L - 1) / best_L;
    gm.n_groups = S * gm.n_segs;
    gm.kn = (float)(1.4426950408889634 / ((double)a.sigma_n + 1e-6));
    gm.kx = (float)(1.4426950408889634 / ((double)a.sigma_x + 1e-6));
    gm.pos_minus_nrm = (unsigned)(reinterpret_cast<const char *>(a.pos) - reinterpret_cast<const char *>(a.nrm));
    gm.lum_minus_nrm = (unsigned)(reinterpret_cast<const char *>(a.lum) - reinterpret_cast<const char *>(a.nrm));
    const int groups_pad = (gm.n_groups + 7) / 8 * 8;
    const int nblocks = groups_pad * gm.n_strips;
    hipLaunchKernelGGL((k_atrous_dma<LOG2S, TX, ROWS, HASVAR>), dim3(nblocks), dim3(TX * ROWS), lds, s, a, gm);
    return hipGetLastError();
}

}  // namespace

bool atrous_dma_supported(const AtrousArgs &a)
{
    if (a.step != 2 && a.step != 4 && a.step != 8 && a.step != 16) return false;
    if (!a.lum || !a.nan_flag || !a.zero_page || !a.inf_page) return false;
    if (a.W % 4 != 0 || a.W < 16) return false;                // 16-byte luminance chunks never straddle the image edge
    if (a.arena_bytes == 0 || a.arena_bytes >= (1ULL << 32)) return false;    // 32-bit offsets between the planes
    if (a.var != nullptr) return false;                       // pre-blur rows come from colour.w here
    return true;
}

hipError_t launch_atrous_dma(const AtrousArgs &a, hipStream_t s)
{
    switch (a.step) {
    case 2: return a.dst ? launch_dma_cfg<1, 256, 3, true>(a, s) : launch_dma_cfg<1, 256, 3, false>(a, s);
    case 4: return a.dst ? launch_dma_cfg<2, 256, 3, true>(a, s) : launch_dma_cfg<2, 256, 3, false>(a, s);
    case 8: return a.dst ? launch_dma_cfg<3, 256, 3, true>(a, s) : launch_dma_cfg<3, 256, 3, false>(a, s);
    case 16: return a.dst ? launch_dma_cfg<4, 128, 6, true>(a, s) : launch_dma_cfg<4, 128, 6, false>(a, s);
    default: return hipErrorInvalidValue;
    }
}
