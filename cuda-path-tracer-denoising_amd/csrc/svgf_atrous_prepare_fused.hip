// svgf_atrous_prepare_fused.hip — non-temporal mode: the prepare pass fused into the first a-trous level (round 4, default).
//
// Reference EstimateVariance (src/denoise.cu:320-329: variance = 10) + the colour copy behind it (:370) + this library's G-buffer
// split are loads and stores only; here they ride in the loader waves of the lane-marching kernel's first level
// (svgf_atrous_lane_impl.h, FUSED = 3): BASELINE configs[0] 0.0344 -> 0.0248 ms per frame, output bit-identical to the prepare
// kernel + level (tests/test_prepare_fused_gpu.py).
#include <cstring>

#include "svgf_atrous_lane_impl.h"

// Non-temporal mode (reference EstimateVariance :320-329 + colour copy :370, and the G-buffer split of this library's prepare
// kernel) fused into the first level: FUSED = 3 of the lane kernel.  Unlike the temporal pass this is loads and stores only — the
// loaders fetch the 1-spp colour and the texel instead of three planes and write the split planes of the pixels their workgroup
// owns — so it fits the loader waves: one launch and 108 B/px of traffic less per frame.
bool atrous_prepare_fused_supported(const AtrousArgs &a, const TemporalArgs &t)
{
    if (a.step != 2 || !t.gbuf || !t.in_rgb) return false;          // the AoS boundary, the reference's first level
    if ((long long)a.W * a.H * 52 >= (1LL << 32)) return false;      // 32-bit byte offsets into the texel array
    return t.nrm_cur && t.pos_cur && t.gid_cur;
}

hipError_t launch_atrous_prepare_fused(const AtrousArgs &a, const TemporalArgs &t, hipStream_t s)
{
    if (!atrous_prepare_fused_supported(a, t)) return hipErrorInvalidValue;
    LaneFused f;
    memset(&f, 0, sizeof(f));
    static_cast<TemporalArgs &>(f) = t;
    return a.dst ? launch_lane_cfg<1, true, 1, 0, 3>(a, s, &f) : launch_lane_cfg<1, false, 1, 0, 3>(a, s, &f);
}

