/*
 * svgf.h — C ABI of the MI355X-native SVGF denoiser (libsvgf_hip.so).
 *
 * This is the drop-in boundary for the reference's denoiser entry points
 *     void denoiseInit(Scene *scene);                         (reference src/denoise.h:6,  src/denoise.cu:31-61)
 *     void denoiseFree();                                     (reference src/denoise.h:7,  src/denoise.cu:63-74)
 *     void denoise(glm::vec3 *out, glm::vec3 *in,
 *                  GBufferTexel *gbuffer);                    (reference src/denoise.h:8,  src/denoise.cu:349-402)
 * The reference keeps its state in file-static globals and reads 13 `ui_*`
 * globals + `scene->state.camera` at call time (src/main.h:39-69, src/denoise.cu:350-399).
 * Here the same information crosses the boundary explicitly: a context handle
 * (so one context per GPU can coexist), a camera block and a parameter block.
 * `cuda-path-tracer-denoising_amd/csrc/denoise_compat.cpp` rebuilds the three legacy functions (declared by the
 * renderer's own denoise.h) on top of these entry points; INTEGRATION.md shows the binding.
 *
 * Every entry point switches to its context's device and restores the caller's current device before it returns.
 * Contexts are independent: one per GPU (or several per GPU) may be driven from different host threads; one context
 * must not be used from two threads at once.
 *
 * Plain C: pointers and sizes only, no C++/torch types.  All image pointers are
 * DEVICE pointers unless the function name ends in `_host`.
 *
 * Wire formats (kept bit-identical to the reference):
 *   colour image : W*H packed float[3]  (glm::vec3, 12 B), index p = x + y*W   (src/pathtrace.cu:193)
 *   G-buffer     : W*H SvgfGBufferTexel (52 B, align 4)                        (src/sceneStructs.h:113-119)
 */
#ifndef SVGF_H_
#define SVGF_H_

#ifdef __cplusplus
extern "C" {
#endif

#define SVGF_VERSION_MAJOR 0
#define SVGF_VERSION_MINOR 9

/* ---- error codes (every entry point returns one of these; the library never exits) ---- */
#define SVGF_OK                 0
#define SVGF_ERR_INVALID_ARG   -1
#define SVGF_ERR_NO_DEVICE     -2   /* no usable HIP device / HIP runtime error at create */
#define SVGF_ERR_OOM           -3
#define SVGF_ERR_HIP           -4   /* a HIP call or kernel launch failed; see svgf_last_error */
#define SVGF_ERR_UNSUPPORTED   -5

/* G-buffer texel, layout of reference `struct GBufferTexel` (src/sceneStructs.h:113-119):
 * normal@0 position@12 albedo@24 ialbedo@36 geomId@48, sizeof == 52, alignof == 4.
 * geomId == -1 marks a ray miss (src/pathtrace.cu:317-323). */
typedef struct SvgfGBufferTexel {
    float normal[3];
    float position[3];
    float albedo[3];
    float ialbedo[3];
    int   geomId;
} SvgfGBufferTexel;

/* The camera fields `denoise()` reads from `scene->state.camera`
 * (src/sceneStructs.h:74-83; used at src/denoise.cu:33-34,343-346,350-351).
 * right/up are NOT normalised in the reference app (src/main.cpp:180-184); pass them as they are. */
typedef struct SvgfCamera {
    float right[3];
    float up[3];
    float view[3];
    float position[3];
} SvgfCamera;

/* The `ui_*` globals `denoise()` reads (src/main.h:39-69, defaults src/main.cpp:49-62). */
typedef struct SvgfParams {
    int   temporal_enable;    /* ui_temporal_enable  (default 0) */
    int   spatial_enable;     /* ui_spatial_enable   (default 0) */
    float color_alpha;        /* ui_color_alpha      (0.2)  */
    float moment_alpha;       /* ui_moment_alpha     (0.2)  */
    int   blur_variance;      /* ui_blurvariance     (1)    */
    float sigma_l;            /* ui_sigmal           (0.45) luminance edge-stop */
    float sigma_x;            /* ui_sigmax           (0.35) position  edge-stop */
    float sigma_n;            /* ui_sigman           (0.2)  normal    edge-stop */
    int   atrous_nlevel;      /* ui_atrous_nlevel    (5), 0..SVGF_MAX_LEVELS */
    int   history_level;      /* ui_history_level    (1)    */
    int   sepcolor;           /* ui_sepcolor         (0)    */
    int   addcolor;           /* ui_addcolor         (0)    */
    int   right_view_option;  /* ui_right_view_option (0): 0 image, 1 history length, 2 variance */
    /* --- extensions; 0 == reference behaviour --- */
    int   kernel_variant;     /* 0 auto (steps 2-32: the cheaper of the lane-marching and the LDS strip kernel by their
                                 launch-geometry cost model — lane at 1920, 3840, 1600, 3440 columns ..., strip at 1024,
                                 2048 ...; the lane kernel at steps 16-32 also needs the variance plane the previous level
                                 leaves; lattice sub-image kernel for steps >= 64; gather where none applies; on NON-temporal
                                 frames the prepare pass (variance fill + G-buffer split) rides in the first level's loader
                                 waves whenever that level runs the lane kernel at step 2 on the AoS boundary: one launch
                                 less, +39 % on BASELINE configs[0]),
                                 1 strict gather kernel on every level,
                                 2 LDS strip kernel for every step 2-32 (error if a step is unsupported, raised before
                                 anything is enqueued),
                                 4 lane-marching kernel wherever it is supported (steps 2-32) whatever the image width,
                                 strip / lattice for the rest; the temporal (or prepare) pass is always its own kernel.
                                 3 is retired (SVGF_ERR_INVALID_ARG).  5 and 6 name parked experiments (two-y-phase geometry;
                                 temporal pass fused into the first level — measured losses) that exist only in
                                 the experiments build of these sources (libsvgf_hip_exp.so, -DSVGF_BUILD_EXPERIMENTS): this
                                 library answers SVGF_ERR_UNSUPPORTED, before anything is enqueued. */
    int   inputs_ready;       /* The FRAME PIPELINE (ABI 0.8; explicit resources since 0.9).  Only a context whose pipeline resources
                                 exist — svgf_create_ex(.., SVGF_CREATE_PIPELINED, ..) or svgf_enable_pipeline(ctx): a second set of colour
                                 planes (+60 B/px), two internal streams, events — looks at this field; any other context orders every
                                 frame on `stream`, and svgf_denoise never allocates or synchronises.
                                 1 is a promise about THIS call: the inputs (colour, G-buffer) are COMPLETE at call time — not merely
                                 enqueued on `stream` — and stay untouched until the work of this call is done.  The library then does
                                 not order the frame behind `stream`: it runs even and odd frames on its two internal streams with
                                 two sets of colour planes, starts a frame's temporal pass as soon as the level that feeds the
                                 previous frame's colour history has run (history_level 1: the previous frame's levels 2-5 and this
                                 frame's temporal pass + level 1 share the GPU), lets only the kernel that writes `output` wait for
                                 what `stream` held at call time (readers of the same buffer behind earlier calls), and makes
                                 `stream` wait for the frame's end — so what the caller enqueues behind the call sees `output`,
                                 exactly as without the promise.  Results are bit-identical to ordered frames
                                 (tests/test_pipeline_gpu.py).  What it is worth depends on the box: +5-10 % at 1080p where the socket
                                 has power headroom, nothing where the ordered frames already run at the power limit.
                                 The promise NEEDS the two internal streams on different hardware queues of the HIP runtime
                                 (GPU_MAX_HW_QUEUES, default 4, read by the runtime when the process starts; a process that also holds
                                 an RCCL communicator or many streams of its own should export 8): on one queue pipelined frames are
                                 8-10 % SLOWER than ordered ones.  The library probes this when the resources are created (two 200 us
                                 kernels, one per stream: side by side or one after the other?), replaces its second stream up to five
                                 times when the pair serialises, and REFUSES the promise when they still
                                 serialise — svgf_pipeline_status() == 2, svgf_last_error() says why, promised frames run as
                                 ordinary ordered frames.
                                 0 = everything ordered on `stream` (the reference's behaviour).
                                 2 = the pipeline WITHOUT the promise: the frame is ordered behind `stream` like any other work
                                 (inputs produced there, output consumed there), and a caller that alternates TWO streams from
                                 frame to frame — each with its own input / output buffers — gets the same overlap from plain
                                 stream semantics, because a stream only waits for the frames that were given to it
                                 (examples/pipeline.cpp: producer + denoiser of consecutive frames overlapping, no events).
                                 Planar frames (svgf_denoise_planar) take both forms up too (ABI 0.9): the promise then covers the
                                 context's own current-frame planes — filled, complete, and not refilled before the work of the
                                 call is done; a producer that refills them every frame takes the pointers through
                                 svgf_planar_gbuffer_stream.  Ignored — the frame is ordered on `stream` — while `stream` is being
                                 captured into a graph.  Once a frame of the context HAS been captured, promised frames order
                                 themselves behind `stream` entirely (a replay of the graph on `stream` uses the same planes): mix
                                 graph replay and the promise only if that is acceptable.  One-launch frames (temporal off, one level)
                                 and frames whose colour history is the LAST level's output have nothing to overlap and never take
                                 the pipeline up. */
    float reproj_scale[2];    /* "next" row f4, paper-faithful reprojection: if > 0, the previous-frame clip
                                 coordinate is divided by it before the ndc mapping: (tan(FOVY) * W / H, tan(FOVY)) =
                                 (pixelLength.x * W / 2, pixelLength.y * H / 2) makes the reprojection exact for any field
                                 of view and aspect ratio.  0 = the reference's mapping (src/denoise.cu:202-203: "no
                                 tan(fov), no aspect", exact only for tan(FOVY) = 1 and W = H; at 16:9 a static camera
                                 keeps its history on ~16 % of the pixels) */
    int   paper_steps;        /* "next" row f4: 1 = a-trous level k (1-based) uses dilation 2^(k-1) = 1, 2, 4, ... as in the
                                 SVGF paper; 0 = the reference's 2^k = 2, 4, 8, ... (its level counter starts at 1,
                                 src/denoise.cu:98,386).  Appended in ABI 0.2 (sizeof(SvgfParams) 68 -> 72). */
    float reproj_position_tol;/* "next" row f4, the consistency test the reference's README (:39) names but isReprjValid
                                 (src/denoise.cu:172-182) does not have: if > 0, a history tap is also rejected when the world
                                 position stored for it in the previous frame lies further than this from the current
                                 pixel's position (Schied et al. 2017 test depth; world position is what the G-buffer of this
                                 renderer carries).  0 = the reference's test (bounds, geomId, normal).  ABI 0.3. */
    int   spatial_variance_frames; /* "next" row f4, the spatial variance estimate the reference leaves as a TODO
                                 (src/denoise.cu:326, EstimateVariance is a constant): if K > 0, a pixel whose updated history
                                 is shorter than K frames takes its variance from the luminance moments of its 7x7
                                 neighbourhood instead of from its own (Schied et al. 2017, section 4.2): taps that pass the
                                 reference's own consistency predicate (same geomId, |n_q - n_p| <= 0.1) count with weight 1,
                                 variance = max(0, mean(m2) - mean(m1)^2) * max(1, 4 / history length).  0 = reference
                                 (temporal variance, 100 without history).  ABI 0.3 (sizeof(SvgfParams) 72 -> 80). */
} SvgfParams;

#define SVGF_MAX_LEVELS 10

typedef struct svgf_ctx svgf_ctx;

/* Library / ABI version: (major << 16) | minor. */
int svgf_version(void);

/* Fill *p with the reference defaults of src/main.cpp:49-62. */
int svgf_params_default(SvgfParams *p);

/* denoiseInit equivalent: allocate per-pixel history state for a width x height image on HIP
 * device `device` and zero the history (src/denoise.cu:31-61).  *out receives the handle. */
int svgf_create(int device, int width, int height, svgf_ctx **out);

/* svgf_create with flags (ABI 0.9).  SVGF_CREATE_PIPELINED: also create the frame pipeline's resources now (see
 * SvgfParams::inputs_ready) — second colour-plane set, two internal streams, events, the hardware-queue probe — so that no frame of
 * the sequence ever allocates or synchronises.  Unknown flag bits are SVGF_ERR_INVALID_ARG. */
#define SVGF_CREATE_PIPELINED 1u
int svgf_create_ex(int device, int width, int height, unsigned flags, svgf_ctx **out);

/* The same resources for an existing context, between frames (allocates, synchronises the device; not under stream capture).
 * Idempotent.  Returns SVGF_OK also when the probe refuses the promise: ask svgf_pipeline_status. */
int svgf_enable_pipeline(svgf_ctx *ctx);

/* 0: the context has no pipeline resources (inputs_ready is ignored); 1: it has, and promised frames run on the internal streams;
 * 2: it has, but the two internal streams share a hardware queue of the HIP runtime — the promise (inputs_ready = 1) is refused and
 *    such frames run ordered on the caller's stream (inputs_ready = 2, the caller's own two streams, still works);
 *    svgf_last_error(ctx) holds the explanation right after svgf_create_ex / svgf_enable_pipeline. */
int svgf_pipeline_status(const svgf_ctx *ctx);

/* The library's queue probe for a CALLER's two streams (what inputs_ready = 2 runs on): 1 = kernels enqueued on `stream_a` and `stream_b`
 * run side by side, 0 = the HIP runtime has put the two streams on one hardware queue and they run one after the other (frames in
 * turn on them then gain nothing and pay a few per cent: use one stream and inputs_ready = 0 instead, as examples/farm.cpp and
 * examples/pipeline.cpp do), < 0 = error.  Two 200 us one-wave kernels and two device synchronisations: call it once, at set-up. */
int svgf_streams_overlap(int device, void *stream_a, void *stream_b);

/* denoiseFree equivalent (src/denoise.cu:63-74).  NULL is accepted. */
int svgf_destroy(svgf_ctx *ctx);

/* denoiseFree + denoiseInit on the same size (what runCuda() does on reset, src/main.cpp:192-201):
 * history length, moments and variance go back to zero; nothing is reallocated. */
int svgf_reset(svgf_ctx *ctx);

/* denoise() equivalent (src/denoise.cu:349-402).  Enqueues the whole frame on `stream`
 * (a hipStream_t, NULL = default stream) and returns WITHOUT synchronising; the legacy
 * shim adds the hipDeviceSynchronize() the reference ends with (src/denoise.cu:401).
 * `cam` is the camera of THIS frame; it is retained as the previous view for the next call
 * (src/denoise.cu:399). */
int svgf_denoise(svgf_ctx *ctx, void *out_rgb_dev, const void *in_rgb_dev,
                 const void *gbuffer_dev, const SvgfCamera *cam, const SvgfParams *params,
                 void *stream);

/* Convenience for tests/tools: same call with HOST pointers (uploads, runs, downloads, syncs). */
int svgf_denoise_host(svgf_ctx *ctx, float *out_rgb_host, const float *in_rgb_host,
                      const SvgfGBufferTexel *gbuffer_host, const SvgfCamera *cam,
                      const SvgfParams *params);

/* Block until everything enqueued on the context's DEVICE has finished (hipDeviceSynchronize: what the reference's denoise()
 * ends with, src/denoise.cu:401, and what the legacy shim calls). */
int svgf_sync(svgf_ctx *ctx);

/* Stream-scoped completion (ABI 0.7): block until what has been enqueued on `stream` — the stream the frames were given to
 * svgf_denoise with — has finished, and nothing else: a renderer with several streams does not stall the others.
 * svgf_denoise itself never synchronises and never allocates (the frame pipeline's resources are created by svgf_create_ex /
 * svgf_enable_pipeline, ABI 0.9); frames without the inputs_ready promise touch no stream but `stream`, so such a frame may be captured
 * into a hipGraph (hipStreamBeginCapture .. svgf_denoise .. hipStreamEndCapture) and replayed: the launches are recorded with the plane
 * roles of the captured call, so replay a graph of TWO consecutive frames (or any even number) to keep the context's rotation
 * consistent (tests/test_stream_gpu.py).  Promised frames (inputs_ready = 1 on a pipelined context) run on the context's internal
 * streams and `stream` waits for their end. */
int svgf_sync_stream(svgf_ctx *ctx, void *stream);

/* 1 while the context's frames alternate between its two plane sets (the frame pipeline is in use: resources created and either the
 * probe accepted the promise or a frame has asked for inputs_ready = 2), else 0. */
int svgf_is_pipelined(const svgf_ctx *ctx);

/* 0 for this (product) build; 1 for the experiments build of the same sources, which additionally accepts kernel_variant 5 / 6
 * and exports a tuning entry point that is deliberately not declared here. */
int svgf_build_has_experiments(void);

/* Message of the last error on this context (or of the last failed svgf_create if ctx == NULL). */
const char *svgf_last_error(const svgf_ctx *ctx);

int svgf_width(const svgf_ctx *ctx);
int svgf_height(const svgf_ctx *ctx);

/* ---- state inspection (tests, sequence goldens).  All copy W*H elements to HOST memory, synchronously. ---- */
#define SVGF_STATE_HISTORY_LENGTH   0   /* int32[W*H]   history length that the NEXT frame will read (src/denoise.cu:398) */
#define SVGF_STATE_MOMENTS          1   /* float[2*W*H] moment history of the NEXT frame (src/denoise.cu:397) */
#define SVGF_STATE_COLOR_HISTORY    2   /* float[3*W*H] colour history of the NEXT frame (src/denoise.cu:366,370,391) */
#define SVGF_STATE_VARIANCE_TEMPORAL 3  /* float[W*H]   variance after the temporal pass of the LAST frame (src/denoise.cu:306,315,327) */
#define SVGF_STATE_COLOR_ACC        4   /* float[3*W*H] colour after the temporal pass of the LAST frame (src/denoise.cu:297,313) */
int svgf_read_state(svgf_ctx *ctx, int which, void *host_dst, unsigned long long host_bytes);

/* Keep a copy of the temporal pass output of every following frame so that SVGF_STATE_VARIANCE_TEMPORAL and
 * SVGF_STATE_COLOR_ACC can be read back (costs one extra 16 B/px device copy per frame; tests only). */
int svgf_set_capture(svgf_ctx *ctx, int on);

/* ---- per-kernel timing with HIP events on the launch stream (bench / roofline) ----
 * svgf_profile_enable(ctx, nframes): allocate event pairs for `nframes` frames (0 = off) and restart the frame
 * counter; every following svgf_denoise brackets each kernel it launches with an event pair in slot
 * (frame_counter % nframes).  No synchronisation happens inside svgf_denoise.
 * svgf_profile_read(ctx, slot, ...): after a sync, returns for that slot the kernels launched, in launch order:
 * kind code (SVGF_KERNEL_*) and elapsed milliseconds. */
#define SVGF_KERNEL_TEMPORAL   1
#define SVGF_KERNEL_PREPARE    2   /* non-temporal variance fill + G-buffer split */
#define SVGF_KERNEL_ATROUS     3
#define SVGF_KERNEL_DEBUGVIEW  4
#define SVGF_KERNEL_COPYOUT    5
#define SVGF_KERNEL_FUSED      6   /* prepare pass of a NON-temporal frame + first a-trous level in one launch (ABI 0.6): such a frame reports
                                      one FUSED entry in place of PREPARE + the first ATROUS; temporal frames never do in this build */
/* Timing source: an event pair attached to each kernel dispatch (hipExtLaunchKernelGGL): the kernel's own begin / end
 * timestamps, nothing recorded on the stream.
 * svgf_profile_stride(ctx, k): time only every k-th frame (k >= 1, default 1).
 * svgf_profile_enable(ctx, n) with the n already armed restarts the counters without re-creating events. */
int svgf_profile_enable(svgf_ctx *ctx, int nframes);
int svgf_profile_stride(svgf_ctx *ctx, int every_kth_frame);
long long svgf_profile_frames(const svgf_ctx *ctx);
int svgf_profile_read(svgf_ctx *ctx, int slot, int max_entries, int *kinds, float *ms, int *n_out);

/* ---- "next" row f1: device-side producer of the denoiser's inputs ---------------------------------
 * In the reference the 1-spp colour and the G-buffer are produced on the device by the path tracer (primary rays
 * src/pathtrace.cu:187-208, first-hit G-buffer fill src/pathtrace.cu:317-323) and handed to denoise() as device
 * pointers (src/pathtrace.cu:436-438).  These two entry points stand in for that producer with the analytic
 * Cornell-like scene of SURVEY.md 8(d): svgf_synth_camera replaces runCuda()'s camera update (src/main.cpp:154-190,
 * pixelLength of src/scene.cpp:159-166), svgf_synth_render replaces generateRayFromCamera + the first-bounce G-buffer
 * write + the 1-spp radiance.  Stateless; outputs are packed rgb (12 B/px) and SvgfGBufferTexel (52 B/px) in device
 * memory, ready for svgf_denoise on the same stream. */
typedef struct SvgfSynthParams {
    int   frame;             /* frame index: noise stream, and camera phase when the camera moves */
    int   seed;              /* sequence seed */
    float noise;             /* multiplicative noise amplitude (0.6) */
    float fireflies;         /* fraction of pixels 6x brighter (0.02) */
    float pixel_length[2];   /* Camera::pixelLength, as svgf_synth_camera returns it */
} SvgfSynthParams;
int svgf_synth_camera(int frame, int moving, int width, int height, SvgfCamera *cam, float pixel_length[2]);
int svgf_synth_render(int device, void *out_rgb_dev, void *out_gbuffer_dev, int width, int height,
                      const SvgfCamera *cam, const SvgfSynthParams *sp, void *stream);

/* ---- "next" row f1, second half: the AoS -> plane repack fused into the producer ---------
 * svgf_denoise reads the reference's 52-byte AoS texel (src/sceneStructs.h:113-119) once per frame and splits it into the
 * planes every later kernel works on (packed float3 normals, packed float3 positions, int geomId): 52 B/px fetched, 28 B/px
 * written, only because the producer and the consumer do not share a layout.  A producer that can write planes gets the
 * context's OWN current-frame planes from svgf_planar_gbuffer and fills them in place (src/pathtrace.cu:317-323 is where
 * the reference fills its texel); svgf_denoise_planar then runs the frame without touching an AoS G-buffer at all.
 *   normal, position: packed float3 per pixel; geom_id: int per pixel (-1 = miss); albedo: packed float3 per pixel holding
 *   albedo * ialbedo (only read on the last level when sepcolor && addcolor; may be left unwritten otherwise).
 * The pointers are valid for exactly ONE svgf_denoise_planar call on this context (the planes rotate with the history: ask
 * again for the next frame); producer and svgf_denoise_planar must be ordered on the same stream.  svgf_reset between
 * svgf_planar_gbuffer and svgf_denoise_planar is allowed: it clears both plane sets (so fill the planes AFTER the reset) but
 * does not change which set the pointers name.  Results are bit-identical
 * to svgf_denoise on the same texels (tests/test_planar_inputs.py).  AoS and planar frames may alternate on one context. */
typedef struct SvgfPlanarGBuffer {
    float *normal;
    float *position;
    int   *geom_id;
    float *albedo;
} SvgfPlanarGBuffer;
int svgf_planar_gbuffer(svgf_ctx *ctx, SvgfPlanarGBuffer *out);
/* The same for a PIPELINED context whose producer runs on `stream` (ABI 0.9): instead of waiting for the device (what
 * svgf_planar_gbuffer does on a pipelined context) it makes `stream` wait for exactly what still reads the planes it hands out — the
 * end of the frame before last, whose levels read them as the current G-buffer, and the temporal pass of the last frame, which reads
 * them as the previous one — so that the producer of frame n+1 runs beside the levels of frame n.  Use with inputs_ready = 2 and
 * two streams in turn (producer, svgf_denoise_planar and the consumer of `output` of one frame on one stream).  Not under capture. */
int svgf_planar_gbuffer_stream(svgf_ctx *ctx, SvgfPlanarGBuffer *out, void *stream);
int svgf_denoise_planar(svgf_ctx *ctx, void *out_rgb_dev, const void *in_rgb_dev, const SvgfCamera *cam, const SvgfParams *p, void *stream);
/* svgf_synth_render writing those planes instead of the AoS texel (same pixels, same values) */
int svgf_synth_render_planar(int device, void *out_rgb_dev, const SvgfPlanarGBuffer *out_planes, int width, int height,
                             const SvgfCamera *cam, const SvgfSynthParams *sp, void *stream);
/* sizeof(SvgfParams) of THIS build of the library: the struct has grown at its tail (ABI 0.2: 72 bytes, 0.3: 80) and
 * svgf_denoise reads all of it, so a binding compiled against an older header must not be mixed with a newer library:
 * compare this with the size of your own struct at load time (the Python binding and denoise_compat.cpp do). */
int svgf_params_sizeof(void);

/* ---- "next" row f3: scene-driven producer --------------------------------------------------------
 * The reference's scenes are transformed unit cubes / spheres (src/scene.cpp:47-117, src/intersections.h:50,104) and OBJ
 * triangle meshes (svgf_scene_render_mesh below).  The host side (scene.py: the MATERIAL / OBJECT / CAMERA text format)
 * builds these records; svgf_scene_render casts the primary rays against them and writes colour + G-buffer like
 * svgf_synth_render does for its built-in scene.  geomId = index into `geoms`.  `geoms` and `light` are host memory:
 * the library has read them completely when the call returns (it waits for its own uploads; the kernel behind them stays
 * asynchronous on `stream`). */
#define SVGF_SCENE_MAX_GEOMS 64
typedef struct SvgfSceneGeom {
    int   type;              /* 0 cube [-0.5,0.5]^3, 1 sphere r = 0.5 (object space) */
    int   material;          /* material id of the scene file (informational) */
    float albedo[3];         /* material RGB */
    float emittance;         /* > 0: emitter */
    float xf[12];            /* object -> world, 3x4 row-major: T * Rx * Ry * Rz * S (src/utilities.cpp:65-72) */
    float inv[12];           /* world -> object */
    float invT[9];           /* transpose of inv's 3x3 block (normals of curved primitives) */
} SvgfSceneGeom;
int svgf_scene_render(int device, void *out_rgb_dev, void *out_gbuffer_dev, int width, int height,
                      const SvgfCamera *cam, const SvgfSynthParams *sp, const SvgfSceneGeom *geoms, int n_geoms,
                      const float light[3], void *stream);
/* The same with the scene's triangle meshes (Scene::loadMesh's world-space triangles, src/scene.cpp:234-311; host arrays, read
 * completely before the call returns like `geoms`; every triangle is tested for every pixel — no BVH, the reference's
 * scenes have 38 .. 4 968 triangles — so n_tris x width x height is what a call costs): `tris` = n_tris x 3 vertices x {pos[3], normal[3], uv[2]}, `tri_ids` = the object index written as
 * geomId for each triangle, `tri_albedo` = n_tris x rgb (the mesh's material colour), `geom_ids` = the object index written
 * as geomId for each primitive (NULL: its position in `geoms`).  Textures (n_tex may be 0): `tex_data` = 8-bit RGB images, rows
 * top to bottom, `tex_desc` = n_tex x {byte offset into tex_data, width, height}, `tri_tex` = texture index per triangle or -1
 * (anything outside [-1, n_tex) is SVGF_ERR_INVALID_ARG);
 * a textured triangle's albedo is Texture::getColor at the interpolated uv (src/sceneStructs.h:208-219, :162-164).  First hit as the
 * reference's computeIntersection (src/pathtrace.cu:211-276): nearest of primitives and the nearest triangle of the scene
 * (glm::intersectRayTriangle), mesh normal interpolated with Triangle::Intersect's corner weights (src/sceneStructs.h:168-172). */
#define SVGF_SCENE_MAX_TRIS (1 << 20)
int svgf_scene_render_mesh(int device, void *out_rgb_dev, void *out_gbuffer_dev, int width, int height,
                           const SvgfCamera *cam, const SvgfSynthParams *sp, const SvgfSceneGeom *geoms, int n_geoms,
                           const int *geom_ids, const float *tris, const int *tri_ids, const float *tri_albedo, int n_tris,
                           const int *tri_tex, const int *tex_desc, const unsigned char *tex_data, int n_tex,
                           const float light[3], void *stream);
/* The same frame written into the planes of svgf_planar_gbuffer instead of AoS texels (same pixels, same values; n_geoms and
 * n_tris may each be 0): the producer side of svgf_denoise_planar for scenes in the reference's format. */
int svgf_scene_render_mesh_planar(int device, void *out_rgb_dev, const SvgfPlanarGBuffer *out_planes, int width, int height,
                                  const SvgfCamera *cam, const SvgfSynthParams *sp, const SvgfSceneGeom *geoms, int n_geoms,
                                  const int *geom_ids, const float *tris, const int *tri_ids, const float *tri_albedo, int n_tris,
                                  const int *tri_tex, const int *tex_desc, const unsigned char *tex_data, int n_tex,
                                  const float light[3], void *stream);

/* ---- "next" row f2: the step right after denoise() ------------------------------------------------
 * svgf_display_pack: reference sendTwoImagesToPBO (src/pathtrace.cu:45-77, launched at :446): `left` (the 1-spp
 *   image) and `right` (the denoised image), both packed rgb floats in device memory, side by side into a
 *   (2*width) x height RGBA8 buffer in device memory; channel = clamp((int)(v * 255.0), 0, 255), alpha 0.
 * svgf_save_png: reference saveImage + image::savePNG (src/main.cpp:131-152, src/image.cpp:22-39) for a HOST image:
 *   clamp(v, 0, 1) * 255.f truncated to a byte, 8-bit RGB PNG; mirror_x != 0 flips the image in x as saveImage does. */
int svgf_display_pack(int device, void *pbo_rgba8_dev, const void *left_rgb_dev, const void *right_rgb_dev, int width,
                      int height, void *stream);
int svgf_save_png(const char *path, const float *rgb_host, int width, int height, int mirror_x);

#ifdef __cplusplus
}
#endif
#endif /* SVGF_H_ */
