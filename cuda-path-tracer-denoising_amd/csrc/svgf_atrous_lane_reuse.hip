// svgf_atrous_lane_reuse.hip — the lane-marching a-trous kernel with cross-level reuse of the geometric terms (round 4).
//
// Reference ATrousFilter (src/denoise.cu:130-156) evaluates |n_p - n_q| and |x_p - x_q| for all 25 taps of all five levels.
// The lane kernel (svgf_atrous_lane_impl.h) already evaluates each PAIR of a level once instead of twice; this file adds the
// instantiations that also share pairs BETWEEN levels: the pairs at lattice offsets (+2,0), (-2,+2), (0,+2), (+2,+2) of step S are
// the pairs at offsets (+1,0), (-1,+1), (0,+1), (+1,+1) of step 2S, and their geometric term kn |dn| + kx |dx| depends on the
// G-buffer only.  A level that has a lane-kernel successor stores those four terms per pixel (16 B/px written), the successor
// reads them (16 B/px read) and skips four of its twelve evaluations: 8 of 24 v_sqrt, 24 of 72 packed and 8 of 24 plain VALU
// instructions and 4 of ~48 ds_read_b128 per pixel, on a kernel that is bound by instruction issue and leaves HBM half idle.
#ifndef SVGF_BUILD_EXPERIMENTS
#error "svgf_atrous_lane_reuse.hip is a parked experiment: compile it with -DSVGF_BUILD_EXPERIMENTS (build.py: build_hip(experiments=True))"
#endif
#include "svgf_atrous_lane_impl.h"

namespace {
template <int LOG2S, int LOG2P = LOG2S>
hipError_t launch_reuse_step(const AtrousArgs &a, hipStream_t s)
{
    const int reuse = (a.tin ? 1 : 0) | (a.tout ? 2 : 0);
    if (!a.dst) {               // last level: no variance accumulators, nothing to hand on
        if (reuse != 1) return hipErrorInvalidValue;
        return launch_lane_cfg<LOG2S, false, LOG2P, 0, 0, 1>(a, s);
    }
    switch (reuse) {
    case 1: return launch_lane_cfg<LOG2S, true, LOG2P, 0, 0, 1>(a, s);
    case 2: return launch_lane_cfg<LOG2S, true, LOG2P, 0, 0, 2>(a, s);
    case 3: return launch_lane_cfg<LOG2S, true, LOG2P, 0, 0, 3>(a, s);
    default: return hipErrorInvalidValue;
    }
}
}  // namespace

// a.tin and / or a.tout set; same support conditions as launch_atrous_lane
hipError_t launch_atrous_lane_reuse(const AtrousArgs &a, hipStream_t s)
{
    switch (a.step) {
    case 1: return launch_reuse_step<0>(a, s);
    case 2: return launch_reuse_step<1>(a, s);
    case 4: return launch_reuse_step<2>(a, s);
    case 8: return launch_reuse_step<3>(a, s);
    case 16: return launch_reuse_step<4, 3>(a, s);
    case 32: return launch_reuse_step<5, 3>(a, s);
    default: return hipErrorInvalidValue;
    }
}
