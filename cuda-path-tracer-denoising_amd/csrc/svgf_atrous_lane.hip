// svgf_atrous_lane.hip — the plain a-trous levels on the lane-marching kernel (svgf_atrous_lane_impl.h): steps 1 .. 32.
#include "svgf_atrous_lane_impl.h"

// Estimated duration of a level on this kernel: the launch geometry's cost in lattice rows x 1.86 us (1920x1080: one round of
// 17 + 6 rows = 42.7 us; profiles/r03_exp_widths*.log: within 5 % at eight other sizes).  Used by the automatic kernel choice.
double atrous_lane_estimate_us(const AtrousArgs &a, int n_cu)
{
    const int S = a.step;
    int L = 0;
    return 1.857 * (double)lane_segment_search(lane_strip_count(a.W, S), S, (a.H + S - 1) / S, n_cu, &L);
}

bool atrous_lane_supported(const AtrousArgs &a)
{
    if (a.step == 16 || a.step == 32) {            // chunked x-phases: the loaders blur the variance from the 4-byte plane
        if ((long long)a.W * a.H * 16 >= (1LL << 32)) return false;
        return a.var != nullptr || !a.blur_variance;
    }
    if (a.step != 1 && a.step != 2 && a.step != 4 && a.step != 8) return false;   // step 1: SvgfParams::paper_steps
    if ((long long)a.W * a.H * 16 >= (1LL << 32)) return false;
    return true;
}

hipError_t launch_atrous_lane(const AtrousArgs &a, hipStream_t s)
{
    if (a.tin || a.tout) return hipErrorInvalidValue;                 // cross-level reuse of the geometric terms: svgf_atrous_lane_reuse.hip (experiments build)
    switch (a.step) {
    case 1: return a.dst ? launch_lane_cfg<0, true>(a, s) : launch_lane_cfg<0, false>(a, s);
    case 2: return a.dst ? launch_lane_cfg<1, true>(a, s) : launch_lane_cfg<1, false>(a, s);
    case 4: return a.dst ? launch_lane_cfg<2, true>(a, s) : launch_lane_cfg<2, false>(a, s);
    case 8: return a.dst ? launch_lane_cfg<3, true>(a, s) : launch_lane_cfg<3, false>(a, s);
    case 16: return a.dst ? launch_lane_cfg<4, true, 3>(a, s) : launch_lane_cfg<4, false, 3>(a, s);
    case 32: return a.dst ? launch_lane_cfg<5, true, 3>(a, s) : launch_lane_cfg<5, false, 3>(a, s);
    default: return hipErrorInvalidValue;
    }
}
