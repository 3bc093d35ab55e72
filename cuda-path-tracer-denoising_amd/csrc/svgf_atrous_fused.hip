// svgf_atrous_fused.hip — EXPERIMENTS build only (libsvgf_hip_exp.so): the temporal pass fused into the first a-trous level
// (round 4; DESIGN.md 5.8: parity-green, 176 us against 97 us for the two kernels it replaces — parked), the G-buffer split alone in the
// first level's loaders (FUSED = 4, parked) and the two-y-phase geometry without fusion (kernel_variant 5).
//
// The temporal pass (reference BackProjection, src/denoise.cu:185-317, launched :362-367) is HBM- and latency-bound with the
// VALU mostly idle; the a-trous level that follows it (src/denoise.cu:77-170, first iteration of the loop :386-392) is
// instruction-issue-bound with HBM half idle, and the level reads back what the pass has just written (16 B/px each way) plus
// the planes it split the G-buffer into (24 B/px).  Here the two share ONE sweep over the frame: the loader waves of the
// lane-marching kernel (svgf_atrous_lane_impl.h, FUSED) accumulate the pixels they stage instead of loading them — the
// arithmetic is k_temporal's, function for function (svgf_temporal.h) — so the accumulated colour + variance exists only in the
// LDS ring, and only the workgroup that OWNS a pixel writes its moments, history length and split G-buffer planes.
// The 3x3 variance pre-blur of the level needs the accumulated variance of rows y-1 / y+1: a workgroup therefore holds BOTH
// y-phases of a 240-column strip (LOG2Y = 1), which makes those rows ring rows.  Halo rows and columns are accumulated by
// more than one workgroup (1.27x at 1080p); nothing is communicated between workgroups.
//
// Also here: the same two-y-phase geometry WITHOUT the fusion (kernel_variant 5), the A/B partner that separates what the
// geometry costs from what the fusion brings.
#include <cstring>

#ifndef SVGF_BUILD_EXPERIMENTS
#error "svgf_atrous_fused.hip holds parked experiments: compile it with -DSVGF_BUILD_EXPERIMENTS (build.py: build_hip(experiments=True))"
#endif
#include "svgf_atrous_lane_impl.h"

// byte offset of a context plane from the context's allocation; false when it does not fit 32 bits
static bool arena_offset(const TemporalArgs &t, const void *plane, long long bias, size_t extent, unsigned *out)
{
    if (!t.arena || !plane) return false;
    const long long off = (const char *)plane - (const char *)t.arena + bias;
    if (off < 0 || (unsigned long long)off + extent > (unsigned long long)t.arena_bytes || (unsigned long long)off + extent >= (1ULL << 32)) return false;
    *out = (unsigned)off;
    return true;
}

static bool fused_offsets(const TemporalArgs &t, LaneFused *f)
{
    static_cast<TemporalArgs &>(*f) = t;
    const size_t n = (size_t)t.W * t.H;
    bool ok = true;
    // taps: element index + 2 (t_window), up to two elements beyond either end of the plane (the planes' padding)
    ok = ok && arena_offset(t, t.cv_hist, -2 * 16, (n + 4) * 16, &f->o_cv_hist);
    ok = ok && arena_offset(t, t.mom_hist, -2 * 8, (n + 4) * 8, &f->o_mom_hist);
    ok = ok && arena_offset(t, t.hlen, -2 * 4, (n + 4) * 4, &f->o_hlen);
    ok = ok && arena_offset(t, t.gid_prev, -2 * 4, (n + 4) * 4, &f->o_gid_prev);
    ok = ok && arena_offset(t, t.nrm_prev, -2 * 12, (n + 4) * 12, &f->o_nrm_prev);
    ok = ok && arena_offset(t, t.hlen_upd, 0, n * 4, &f->o_hlen_upd);
    ok = ok && arena_offset(t, t.mom_acc, 0, n * 8, &f->o_mom_acc);
    ok = ok && arena_offset(t, t.dump, 0, 4096, &f->o_dump);
    if (t.cv_acc) ok = ok && arena_offset(t, t.cv_acc, 0, n * 16, &f->o_cv_acc);
    else f->o_cv_acc = f->o_dump;
    if (t.gbuf) {                  // the AoS path splits the G-buffer into the context's planes
        ok = ok && arena_offset(t, t.nrm_cur, 0, n * 12, &f->o_nrm_cur);
        ok = ok && arena_offset(t, t.pos_cur, 0, n * 12, &f->o_pos_cur);
        ok = ok && arena_offset(t, t.gid_cur, 0, n * 4, &f->o_gid_cur);
    } else f->o_nrm_cur = f->o_pos_cur = f->o_gid_cur = f->o_dump;
    return ok;
}

bool atrous_fused_supported(const AtrousArgs &a, const TemporalArgs &t)
{
    if (a.step != 2) return false;                                   // the reference's first level (src/denoise.cu:98,386)
    if ((long long)a.W * a.H + 8 >= (1LL << 24)) return false;       // 24-bit element indices (v_mad_u32_u24), 32-bit byte offsets
    if (t.pos_tol > 0.0f) return false;                              // f4 extension, k_temporal only
    LaneFused f;
    return fused_offsets(t, &f);                                     // every plane inside the context's one allocation
}

double atrous_fused_estimate_us(const AtrousArgs &a, int n_cu)
{
    int L = 0;
    return 1.857 * (double)lane_segment_search(lane_strip_count(a.W, 2, 2), 1, (a.H + 1) / 2, n_cu, &L);
}

hipError_t launch_atrous_fused(const AtrousArgs &a, const TemporalArgs &t, hipStream_t s)
{
    if (a.step != 2) return hipErrorInvalidValue;
    LaneFused f;
    if (!t.dump || !fused_offsets(t, &f)) return hipErrorInvalidValue;
    if (t.gbuf) return a.dst ? launch_lane_cfg<1, true, 1, 1, 1>(a, s, &f) : launch_lane_cfg<1, false, 1, 1, 1>(a, s, &f);
    return a.dst ? launch_lane_cfg<1, true, 1, 1, 2>(a, s, &f) : launch_lane_cfg<1, false, 1, 1, 2>(a, s, &f);
}

// Temporal frames on the AoS boundary: only the G-buffer SPLIT moves into the first level's loaders (FUSED = 4).  The temporal
// pass keeps its arithmetic and its accumulated plane but no longer writes NRM / POS / GID (28 of its 56 B/px of stores); the
// loaders read the texel instead of the two planes and write the planes for the pixels their workgroup owns.
bool atrous_split_fused_supported(const AtrousArgs &a, const TemporalArgs &t)
{
    if (a.step != 2 || !t.gbuf || !a.src) return false;
    if ((long long)a.W * a.H * 52 >= (1LL << 32)) return false;
    return t.nrm_cur && t.pos_cur && t.gid_cur;
}

hipError_t launch_atrous_split_fused(const AtrousArgs &a, const TemporalArgs &t, hipStream_t s)
{
    if (!atrous_split_fused_supported(a, t)) return hipErrorInvalidValue;
    LaneFused f;
    memset(&f, 0, sizeof(f));
    static_cast<TemporalArgs &>(f) = t;
    return a.dst ? launch_lane_cfg<1, true, 1, 0, 4>(a, s, &f) : launch_lane_cfg<1, false, 1, 0, 4>(a, s, &f);
}

// step 2 with both y-phases in one workgroup, not fused (reads the accumulated plane like every other level)
hipError_t launch_atrous_lane_2y(const AtrousArgs &a, hipStream_t s)
{
    if (a.step != 2) return hipErrorInvalidValue;
    return a.dst ? launch_lane_cfg<1, true, 1, 1, 0>(a, s) : launch_lane_cfg<1, false, 1, 1, 0>(a, s);
}
