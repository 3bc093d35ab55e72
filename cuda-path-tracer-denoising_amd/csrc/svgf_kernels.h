// svgf_kernels.h — launch wrappers shared between the host orchestration (svgf_api.hip) and the
// kernel translation units.  Internal; the public boundary is include/svgf.h.
//
// Device data layout (all planes W*H elements, index p = x + y*W, resident in HBM for the context's life):
//   CV[k]   float4 {r, g, b, variance}   colour+variance planes; ping-pong / colour history (16 B/px each)
//   NRM[k]  packed float3 normal         current / previous frame (12 B/px each)
//   GID[k]  int geomId                   current / previous frame (4 B/px each)
//   POS     packed float3 position       current frame (12 B/px)
//   MOM[k]  float2 {m1, m2}              moment history / accumulation (8 B/px each)
//   HLEN[k] int                          history length / update (4 B/px each)
// The 52-byte AoS G-buffer texel of the boundary (reference src/sceneStructs.h:113-119) is read exactly once per
// frame, by the temporal (or prepare) kernel, which splits it into the NRM/POS/GID planes the a-trous levels read.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <mutex>

// Tuning knobs and parked kernel variants exist in the EXPERIMENTS build only (-DSVGF_BUILD_EXPERIMENTS -> libsvgf_hip_exp.so, used
// by tools/experiments/ and the tests marked `experiments`): a name -> int table filled through svgf_exp_set().  The product build
// compiles every SVGF_TUNE() to its default, reads no environment variable and instantiates none of the parked kernels.
#ifdef SVGF_BUILD_EXPERIMENTS
int svgf_exp_get(const char *name, int dflt);       // svgf_api.hip
#define SVGF_TUNE(name, dflt) svgf_exp_get(name, dflt)
#else
#define SVGF_TUNE(name, dflt) (dflt)
#endif

// Profiling hand-off.  svgf_api.hip's KernelTimer arms an event pair for the NEXT kernel launch of this host thread; the launch
// macro hands it to hipExtLaunchKernelGGL, which attaches the events to the DISPATCH itself: start / stop are the kernel's own
// begin / end timestamps (what rocprofv3 reports), and nothing is added to the stream.  hipEventRecord pairs around a launch —
// rounds 1-3 — put a barrier packet on either side of every kernel: the intervals read 2-3 us long and an instrumented frame
// ran 23-46 us longer (profiles/r04_clock_states.txt).
struct SvgfLaunchEvents { hipEvent_t start, stop; };
extern thread_local SvgfLaunchEvents g_svgf_launch_events;      // defined in svgf_api.hip
#define SVGF_LAUNCH_KERNEL(kernel, grid, block, lds, stream, ...)                                                        \
    do {                                                                                                                 \
        if (g_svgf_launch_events.start) {                                                                                \
            const SvgfLaunchEvents ev__ = g_svgf_launch_events;                                                          \
            g_svgf_launch_events = SvgfLaunchEvents{ nullptr, nullptr };                                                 \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, ev__.start, ev__.stop, 0, __VA_ARGS__);              \
        } else {                                                                                                         \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                           \
        }                                                                                                                \
    } while (0)

// Per-device launch state of ONE kernel instantiation (a function-local static of its launcher).  The opt-in to more than
// 64 KB of dynamic LDS is a per-device function attribute and the CU count a device property; the ABI allows one context
// per GPU in one process, driven from several host threads, so both are set up once per device under std::call_once.
struct SvgfLaunchCache {
    std::once_flag once[64];
    hipError_t err[64];
    int n_cu[64];
    // returns the current device's slot (0..63) through *dev; the attribute is only requested when lds_bytes > 0
    hipError_t init(const void *kernel, int lds_bytes, int *dev)
    {
        int d = 0;
        (void)hipGetDevice(&d);
        if (d < 0 || d >= 64) d = 0;
        std::call_once(once[d], [&]() {
            err[d] = lds_bytes > 0 ? hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) : hipSuccess;
            int n = 0;
            if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n <= 0) n = 256;
            n_cu[d] = n;
        });
        *dev = d;
        return err[d];
    }
};

// Entry points switch to their context's device and put the caller's device back on return (a host that drives several
// GPUs from one thread — torch, the renderer through denoise_compat — must not find its current device changed).
struct SvgfDeviceGuard {
    int prev;
    bool ok;
    explicit SvgfDeviceGuard(int device) : prev(-1), ok(false)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = (prev == device) || hipSetDevice(device) == hipSuccess;
    }
    ~SvgfDeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    SvgfDeviceGuard(const SvgfDeviceGuard &) = delete;
    SvgfDeviceGuard &operator=(const SvgfDeviceGuard &) = delete;
};

struct AtrousArgs {
    const float4 *src;        // CV plane read (colour + variance snapshot)
    float4 *dst;              // CV plane written, may be null on the last level
    float *out_rgb;           // packed rgb output (user buffer), null except on the last level
    const float *nrm;         // packed float3
    const float *pos;         // packed float3
    const float *gbuf;        // raw 52-B texels, only read when modulate != 0 (albedo*ialbedo, reference :166-168); null on the planar path
    const float *albedo;      // planar path (svgf_denoise_planar): packed float3 albedo*ialbedo per pixel, read instead of gbuf
    int W, H, step;
    float sigma_c, sigma_n, sigma_x;
    int blur_variance;
    int modulate;
    const float *var;         // optional 4-B/px variance plane holding src[].w (pre-blur neighbourhood reads), may be null
    float *var_dst;           // optional 4-B/px variance plane written next to dst, may be null
    // cross-level reuse of the geometric terms (lane kernel only, svgf_atrous_lane_impl.h REUSE): per pixel {g(+1,0), g(-1,+1),
    // g(0,+1), g(+1,+1)} in units of THIS level's step, g = kn |dn| + kx |dx|, written by the previous level (whose step is half
    // this one's and whose sigma_n / sigma_x are the same); tout: the same for the next level.  Either may be null.
    // Layout: element ((y * S' + x mod S') * M' + x / S') with S' the CONSUMING level's step and M' = ceil(W / S') (t_m for tin,
    // t_m_out for tout): consecutive lanes of a consumer wave read consecutive elements.  Plane size (W + 64) * H elements.
    const float4 *tin;
    float4 *tout;
    int t_m, t_m_out;
};

struct TemporalArgs {
    const float *in_rgb;      // packed rgb, current 1-spp colour
    const float *gbuf;        // raw 52-B texels; null on the planar path: nrm_cur / pos_cur / gid_cur then hold this frame's
                              // G-buffer already (written by the producer) and are read instead of being written
    const float4 *cv_hist;    // colour history (rgb used)
    float4 *cv_acc;           // out: {colour_acc, variance}
    const float2 *mom_hist; float2 *mom_acc;
    const int *hlen; int *hlen_upd;
    const float *nrm_prev; const int *gid_prev;
    float *nrm_cur; int *gid_cur; float *pos_cur;
    float M[16];              // previous view matrix, column-major
    int W, H;
    float color_alpha_min, moment_alpha_min;
    float reproj_sx, reproj_sy;   // SvgfParams::reproj_scale; 0 = reference mapping
    const float *pos_prev;    // previous frame's positions (packed float3), read only when pos_tol > 0
    float pos_tol;            // SvgfParams::reproj_position_tol; 0 = the reference's consistency test
    void *dump;               // fused kernel only: >= 4 KB of scrap the stores of pixels a workgroup does not own go to
    const void *arena;        // fused kernel only: the ONE allocation all the context's planes (and dump) live in, and its size:
    size_t arena_bytes;       // the kernel addresses them as arena + 32-bit offset (svgf_atrous_lane_impl.h: LaneFused)
    int skip_split;           // k_temporal on the AoS boundary: do not write nrm_cur / pos_cur / gid_cur (the first a-trous level's
                              // loaders will: svgf_atrous_fused.hip, FUSED = 4)
};

hipError_t launch_temporal(const TemporalArgs &a, hipStream_t s);
// SvgfParams::spatial_variance_frames (f4): variance of short-history pixels from the 7x7 neighbourhood's moments
hipError_t launch_spatial_variance(float4 *cv_acc, const float2 *mom_acc, const int *hlen_upd, const float *nrm, const int *gid,
                                   int W, int H, int K, hipStream_t s);
// non-temporal mode: variance = 10, colour = input, split G-buffer (reference EstimateVariance :320-329 + :370)
// (gbuf null: the planar path, the planes are already filled and only the colour plane is written)
hipError_t launch_prepare(const float *in_rgb, const float *gbuf, float4 *cv, float *nrm, int *gid, float *pos,
                          int W, int H, hipStream_t s);
hipError_t launch_atrous_gather(const AtrousArgs &a, hipStream_t s);   // strict one-thread-per-pixel gather kernel
hipError_t launch_atrous_strip(const AtrousArgs &a, hipStream_t s);    // LDS strip-marching kernel (fast path)
bool       atrous_strip_supported(const AtrousArgs &a);
double     atrous_strip_estimate_us(const AtrousArgs &a, int n_cu);   // launch-geometry cost model (automatic kernel choice)
hipError_t launch_atrous_lane(const AtrousArgs &a, hipStream_t s);     // lane-marching kernel, symmetric terms shared by DPP (steps 1 .. 32)
bool       atrous_lane_supported(const AtrousArgs &a);
double     atrous_lane_estimate_us(const AtrousArgs &a, int n_cu);
// non-temporal mode: the prepare pass (variance fill + G-buffer split) fused into the first level (step 2, AoS boundary;
// svgf_atrous_prepare_fused.hip)
hipError_t launch_atrous_prepare_fused(const AtrousArgs &a, const TemporalArgs &t, hipStream_t s);
bool       atrous_prepare_fused_supported(const AtrousArgs &a, const TemporalArgs &t);
#ifdef SVGF_BUILD_EXPERIMENTS
// ---- parked variants (measured losses; experiments build only) ----
hipError_t launch_atrous_lane_reuse(const AtrousArgs &a, hipStream_t s);   // lane kernel with a.tin / a.tout (cross-level reuse of geometric terms)
// temporal pass fused into the first level (step 2): the lane kernel's loader waves accumulate the pixels they stage
// (svgf_atrous_fused.hip).  t.cv_acc may be null: the accumulated colour then exists only in LDS.
hipError_t launch_atrous_fused(const AtrousArgs &a, const TemporalArgs &t, hipStream_t s);
bool       atrous_fused_supported(const AtrousArgs &a, const TemporalArgs &t);
double     atrous_fused_estimate_us(const AtrousArgs &a, int n_cu);
// temporal frames: the G-buffer split alone in the first level's loaders (the temporal pass then runs with skip_split)
hipError_t launch_atrous_split_fused(const AtrousArgs &a, const TemporalArgs &t, hipStream_t s);
bool       atrous_split_fused_supported(const AtrousArgs &a, const TemporalArgs &t);
hipError_t launch_atrous_lane_2y(const AtrousArgs &a, hipStream_t s);  // step 2, both y-phases per workgroup, not fused (A/B)
#endif
hipError_t launch_atrous_lattice(const AtrousArgs &a, hipStream_t s);  // lattice sub-images in LDS (steps >= 64)
bool       atrous_lattice_supported(const AtrousArgs &a);
// albedo * ialbedo of the last level's re-modulation (:166-168), from the AoS texel or from the planar path's plane
__device__ __forceinline__ void svgf_modulate(const AtrousArgs &a, unsigned p, float &o0, float &o1, float &o2)
{
    if (a.gbuf) {
        const float *t = a.gbuf + 13u * (size_t)p;
        o0 *= t[6] * t[9]; o1 *= t[7] * t[10]; o2 *= t[8] * t[11];
    } else {
        const float *t = a.albedo + 3u * (size_t)p;
        o0 *= t[0]; o1 *= t[1]; o2 *= t[2];
    }
}

// out = float(value)/scale broadcast to rgb (reference DebugView :331-340)
hipError_t launch_debug_hlen(const int *hlen, float *out_rgb, int n, float scale, hipStream_t s);
hipError_t launch_debug_var(const float4 *cv, float *out_rgb, int n, float scale, hipStream_t s);
hipError_t launch_copy_rgb(const float4 *cv, float *out_rgb, int n, hipStream_t s);   // reference :382
