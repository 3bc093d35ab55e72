"""GPU parity tests (run with -m gpu on an MI355X).  Everything goes through the C ABI of libsvgf_hip.so.

Tolerances (north_star: 1e-4 relative per channel):
  * temporal pass: BIT-EXACT against the oracle and against the reference goldens (-ffp-contract=off build);
  * a-trous, strict gather kernel (kernel_variant=1): <= 2e-6;
  * a-trous, LDS strip kernel (kernel_variant=2 / auto): <= 1e-5 (one fused exp2 instead of three expf, fp32
    denominators, reciprocal instead of division — each ~1e-7).
relerr() uses max(|ref|, 1e-2) as the denominator.
"""
import numpy as np
import pytest

from conftest import denoiser_for, golden_cases, load_golden, relerr, replay

pytestmark = pytest.mark.gpu

CASES = golden_cases()
TOL_GATHER, TOL_STRIP = 2e-6, 1e-5


class Engine:
    """Adapter: Denoiser with the replay() interface (numpy in / numpy out through svgf_denoise_host)."""

    def __init__(self, pkg, W, H, variant=0):
        self.d = denoiser_for(pkg, W, H, variant)       # (variants 5 / 6: the experiments build of the same sources)
        self.variant = variant

    def reset(self):
        self.d.reset()

    def denoise(self, color, gbuffer, cam, p):
        p.kernel_variant = self.variant
        return self.d.denoise_host(color, gbuffer, cam, p)

    def free(self):
        self.d.free()


@pytest.fixture(scope="module", autouse=True)
def native_loaded(pkg):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    lib = pkg.load_library()          # raises when libsvgf_hip.so is absent: no fallback exists
    assert lib.svgf_version() > 0


@pytest.mark.parametrize("variant", [0, 1, 2, 4, pytest.param(5, marks=pytest.mark.experiments), pytest.param(6, marks=pytest.mark.experiments)])
@pytest.mark.parametrize("name", CASES)
def test_hip_matches_reference_goldens(pkg, name, variant):
    """variant 0: the library's choice (prepare pass of non-temporal frames in the first level's loaders), 1 gather, 2 strip,
    4 lane kernels with the temporal / prepare pass on its own; in the experiments build (libsvgf_hip_exp.so) also 5, the fused
    kernel's two-y-phase geometry without the fusion, and 6, the temporal pass fused into the first level on every frame that
    can be."""
    z, runs = load_golden(name)
    W, H = int(z["W"]), int(z["H"])
    for tag in runs:
        nl = int(z[f"call_params_{tag}"][0][8])
        if variant == 2 and nl > 5:
            continue                                   # steps 64,128 are served by the gather kernel
        if variant >= 4 and name.startswith("temporal_"):
            continue                                   # no a-trous level runs: same as variant 0
        e = Engine(pkg, W, H, variant)
        got = replay(pkg, e, z, tag)
        e.free()
        ref = z[f"ref_nofma_out_{tag}"]
        if name.startswith("temporal_"):
            assert np.array_equal(got, ref, equal_nan=True), f"{name}:{tag} temporal pass not bit-exact"
        else:
            err = relerr(got, ref)
            assert err.max() <= (TOL_GATHER if variant == 1 else TOL_STRIP), f"{name}:{tag} v{variant} max rel {err.max():.3e}"


SEQ = [  # (W, H, frames, moving, params)
    (320, 180, 5, True, dict(temporal_enable=1, spatial_enable=1)),
    (257, 131, 4, True, dict(temporal_enable=1, spatial_enable=1, history_level=3, blur_variance=0)),
    (200, 200, 6, False, dict(temporal_enable=1, spatial_enable=1, atrous_nlevel=7, history_level=7, sepcolor=1, addcolor=1)),
    (64, 300, 3, True, dict(temporal_enable=1, spatial_enable=1, atrous_nlevel=2, history_level=0, sigma_l=0.7)),
    (513, 65, 3, False, dict(temporal_enable=0, spatial_enable=1, atrous_nlevel=5)),
    (33, 7, 3, False, dict(temporal_enable=1, spatial_enable=1, atrous_nlevel=4)),
]


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("cfg", SEQ, ids=[f"{c[0]}x{c[1]}" for c in SEQ])
def test_sequences_match_oracle_including_state(pkg, orc, cfg, variant):
    W, H, n, moving, kw = cfg
    p = pkg.reference_defaults().set(**kw)
    p.kernel_variant = variant
    tol = TOL_GATHER if variant == 1 else TOL_STRIP
    d = pkg.Denoiser(W, H, 0)
    d.set_capture(True)
    o = orc.Oracle(pkg, W, H, threads=8)
    for f in range(n):
        c, g, cam = pkg.synth.render_frame(W, H, f, seed=17, moving=moving)
        got = d.denoise_host(c, g, cam, p)
        ref = o.denoise(c, g, cam, p)
        assert relerr(got, ref).max() <= tol, f"frame {f}: {relerr(got, ref).max():.3e}"     # no growth allowance along the sequence
        # temporal state is bit-exact as long as the colour history fed back is (history_level 0) or on frame 0
        assert np.array_equal(d.read_state(0), o.read_state(0)), f"history length differs at frame {f}"
        assert relerr(d.read_state(3), o.read_state(3)).max() <= 2e-4, "variance after temporal pass"
        assert relerr(d.read_state(1), o.read_state(1)).max() <= 1e-5, "moments"
        assert relerr(d.read_state(2), o.read_state(2)).max() <= tol, "colour history"
        if p.temporal_enable:            # colour_acc is only defined by the temporal pass
            assert relerr(d.read_state(4), o.read_state(4)).max() <= tol, "colour_acc"
    d.free(); o.free()


def test_randomised_parameter_sweep_matches_oracle(pkg, orc):
    """24 seeded random configurations (size, every SvgfParams field, camera motion, mode switches between frames)."""
    rng = np.random.default_rng(20260928)
    for case in range(24):
        W, H = int(rng.integers(1, 400)), int(rng.integers(1, 260))
        nframes = int(rng.integers(1, 5))
        moving = bool(rng.integers(0, 2))
        base = dict(color_alpha=float(rng.uniform(0.02, 1.0)), moment_alpha=float(rng.uniform(0.02, 1.0)),
                    blur_variance=int(rng.integers(0, 2)), sigma_l=float(rng.uniform(0.05, 4.0)),
                    sigma_x=float(rng.uniform(0.05, 2.0)), sigma_n=float(rng.uniform(0.02, 1.0)),
                    atrous_nlevel=int(rng.integers(0, 8)), sepcolor=int(rng.integers(0, 2)), addcolor=int(rng.integers(0, 2)))
        base["history_level"] = int(rng.integers(0, base["atrous_nlevel"] + 2))
        d = pkg.Denoiser(W, H, 0)
        o = orc.Oracle(pkg, W, H, threads=8)
        for f in range(nframes):
            p = pkg.reference_defaults().set(temporal_enable=int(rng.integers(0, 2)), spatial_enable=int(rng.integers(0, 4) > 0),
                                             right_view_option=int(rng.choice([0, 0, 0, 1, 2])), **base)
            c, g, cam = pkg.synth.render_frame(W, H, f, seed=1000 + case, moving=moving)
            got = d.denoise_host(c, g, cam, p)
            ref = o.denoise(c, g, cam, p)
            e = relerr(got, ref)
            assert e.max() <= TOL_STRIP, f"case {case} ({W}x{H}) frame {f}: {e.max():.3e} params {base}"
            assert np.array_equal(d.read_state(0), o.read_state(0)), f"case {case} frame {f}: history length"
        d.free(); o.free()


def test_1080p_full_svgf_matches_oracle(pkg, orc):
    """BASELINE config 2 size.  The oracle takes a few seconds per frame on 16 threads."""
    W, H = 1920, 1080
    p = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1)
    d = pkg.Denoiser(W, H, 0)
    o = orc.Oracle(pkg, W, H, threads=16)
    worst = 0.0
    for f in range(2):
        c, g, cam = pkg.synth.render_frame(W, H, f, seed=23, moving=False)
        got = d.denoise_host(c, g, cam, p)
        ref = o.denoise(c, g, cam, p)
        e = relerr(got, ref)
        assert e.max() <= 1e-4 and np.quantile(e, 0.9999) <= 1e-5, f"frame {f}: max {e.max():.3e}"
        worst = max(worst, float(e.max()))
    d.free(); o.free()
    print(f"BASELINE configs[1] (1920x1080, temporal + 5 levels, static camera) vs oracle: worst max-rel {worst:.2e}")


@pytest.mark.parametrize("size", [(1920, 1080), (3840, 640), (333, 517), (64, 64), (40, 200), (131, 3)])
def test_levels_6_and_7_use_lattice_kernel_and_match_oracle(pkg, orc, size):
    """Steps 64 and 128 (levels 6-7 of the reference's 0..7 slider, src/preview.cpp:327) run on the lattice sub-image
    kernel (svgf_atrous_lattice.hip): whole sub-images (1080p), row bands (3840 wide: 9-row bands), sub-images of one
    pixel, images narrower than the step.  Checked against the CPU oracle and against the strict gather kernel."""
    W, H = size
    d = pkg.Denoiser(W, H, 0)
    o = orc.Oracle(pkg, W, H, threads=16)
    for f, kw in enumerate([dict(temporal_enable=1, history_level=7, blur_variance=1),
                            dict(temporal_enable=1, history_level=6, blur_variance=0, sepcolor=1, addcolor=1)]):
        p = pkg.reference_defaults().set(spatial_enable=1, atrous_nlevel=7, **kw)
        c, g, cam = pkg.synth.render_frame(W, H, f, seed=41, moving=True)
        ref = o.denoise(c, g, cam, p)
        p.kernel_variant = 0
        got = d.denoise_host(c, g, cam, p)
        e = relerr(got, ref)
        assert e.max() <= TOL_STRIP * 4, f"{W}x{H} frame {f}: {e.max():.3e}"      # flat: no growth allowance along the sequence
        assert relerr(d.read_state(2), o.read_state(2)).max() <= TOL_STRIP * 4, "colour history (level 6/7 output)"
    d.free(); o.free()
    # the same frame through the lattice kernel and through the strict gather kernel
    c, g, cam = pkg.synth.render_frame(W, H, 0, seed=43, moving=False)
    outs = {}
    for variant in (0, 1):
        d = pkg.Denoiser(W, H, 0)
        p = pkg.reference_defaults().set(spatial_enable=1, atrous_nlevel=7, kernel_variant=variant)
        outs[variant] = d.denoise_host(c, g, cam, p)
        d.free()
    assert relerr(outs[0], outs[1]).max() <= 2e-5


def test_lattice_kernel_keeps_the_nan_semantics(pkg, orc):
    """A NaN position / normal texel at steps 64-128: min(1, exp(NaN)) == 1 (the `careful` path of the kernel)."""
    W, H = 300, 200
    c, g, cam = pkg.synth.render_frame(W, H, 0, seed=47, moving=False)
    g = g.copy()
    g["position"][70, 150] = np.nan
    g["normal"][130, 20, 1] = np.inf
    p = pkg.reference_defaults().set(spatial_enable=1, atrous_nlevel=7)
    o = orc.Oracle(pkg, W, H, threads=8)
    ref = o.denoise(c, g, cam, p)
    o.free()
    d = pkg.Denoiser(W, H, 0)
    got = d.denoise_host(c, g, cam, p)
    d.free()
    both_nan = np.isnan(got) & np.isnan(ref)
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    assert relerr(np.where(both_nan, 0, got), np.where(both_nan, 0, ref)).max() <= TOL_STRIP * 4


@pytest.mark.parametrize("size,variant", [((1920, 70), 0), ((300, 200), 4), ((123, 77), 4), ((3840, 48), 0)])
def test_lane_kernel_at_steps_16_and_32_nan_texels_and_options(pkg, orc, size, variant):
    """Steps 16 / 32 on the lane-marching kernel (chunked x-phases, variance pre-blur computed by the loader threads): widths
    where the automatic choice takes it (1920, 3840) and small images with kernel_variant 4; non-finite normal / position texels
    (the `careful` path), variance blur off, re-modulation on the last level, the history fed by level 4 / 5, paper step sizes
    (step 16 as the LAST level: the variant without variance accumulators)."""
    W, H = size
    o = orc.Oracle(pkg, W, H, threads=16)
    d = pkg.Denoiser(W, H, 0)
    cfgs = [dict(temporal_enable=1, history_level=1, blur_variance=1),
            dict(temporal_enable=1, history_level=5, blur_variance=0, sepcolor=1, addcolor=1),
            dict(temporal_enable=1, history_level=4, blur_variance=1, paper_steps=1),
            dict(temporal_enable=0, history_level=0, blur_variance=1)]
    for f, kw in enumerate(cfgs):
        p = pkg.reference_defaults().set(spatial_enable=1, atrous_nlevel=5, kernel_variant=variant, **kw)
        c, g, cam = pkg.synth.render_frame(W, H, f, seed=53, moving=True)
        g = g.copy()
        if f >= 1:
            g["position"][H // 3, W // 2] = np.nan
            g["normal"][H // 2, min(20, W - 1), 1] = np.inf
        ref = o.denoise(c, g, cam, p)
        got = d.denoise_host(c, g, cam, p)
        assert np.array_equal(np.isnan(got), np.isnan(ref)), f"{W}x{H} frame {f}: NaN pattern"
        both = np.isnan(got) & np.isnan(ref)
        e = relerr(np.where(both, 0, got), np.where(both, 0, ref))
        assert e.max() <= TOL_STRIP * 2, f"{W}x{H} variant {variant} frame {f} {kw}: {e.max():.3e}"
    d.free(); o.free()


def test_4k_size_independent_properties(pkg):
    """BASELINE config 4 size (3840x2160): properties that need no CPU oracle run."""
    import torch
    W, H = 3840, 2160
    c, g, cam = pkg.synth.render_frame(W, H, 0, seed=29, moving=False)
    p = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1)
    outs = {}
    for variant in (1, 2):
        d = pkg.Denoiser(W, H, 0)
        p.kernel_variant = variant
        outs[variant] = [d.denoise_host(c, g, cam, p) for _ in range(2)]
        hl = d.read_state(0)
        d.free()
        assert hl.max() == 2 and hl.min() >= 1
    for f in range(2):
        e = relerr(outs[2][f], outs[1][f])                         # two independent kernels agree at full size
        assert e.max() <= 2e-5, f"4K frame {f}: strip vs gather {e.max():.3e}"
    # constant colour is a fixed point of the whole pipeline
    cc = np.full_like(c, 0.5)
    d = pkg.Denoiser(W, H, 0)
    p.kernel_variant = 0
    out = d.denoise_host(cc, g, cam, p)
    d.free()
    assert relerr(out, cc).max() <= 1e-6
    del torch


def test_4k_full_svgf_matches_oracle_static_and_moving(pkg, orc):
    """BASELINE configs[3] size, 3840x2160, full SVGF (temporal + 5 levels) with the library's default kernel selection —
    the size where segment lengths, strip counts and the lane / strip choice differ from every golden — against the CPU oracle
    on EVERY frame of a static and of a moving six-frame sequence (the history passes the 1 / alpha = 5 plateau): <= 1e-4
    relative per channel (north_star's bar).  The frames come from the device producer (bit-exact with the numpy generator,
    tests/test_synth_producer.py) and are brought to the host for the oracle."""
    import torch
    W, H = 3840, 2160
    p = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, atrous_nlevel=5, history_level=1)
    d_in = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    d_g = torch.empty((H * W * 52,), dtype=torch.uint8, device="cuda")
    out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    worst = 0.0
    for moving in (False, True):
        d = pkg.Denoiser(W, H, 0)
        o = orc.Oracle(pkg, W, H, threads=min(64, __import__("os").cpu_count() or 1))
        for f in range(6):
            cam = pkg.synth.camera_for_frame(f, moving)
            pkg.binding.synth_render(d_in, d_g, W, H, cam, f, seed=41, device=0)
            d.denoise(out, d_in, d_g, pkg.SvgfCamera.from_dict(cam), p)
            torch.cuda.synchronize()
            c = d_in.cpu().numpy()
            g = d_g.cpu().numpy().view(pkg.synth.GBUFFER_DTYPE).reshape(H, W)
            got = out.cpu().numpy()
            ref = o.denoise(c, g, cam, p)
            e = relerr(got, ref)
            worst = max(worst, float(e.max()))
            assert e.max() <= 1e-4, f"4K {'moving' if moving else 'static'} frame {f}: max rel {e.max():.3e}"
            assert np.array_equal(d.read_state(0), o.read_state(0)), "history length"
        d.free(); o.free()
    print(f"4K worst relative error over 12 frames: {worst:.3e}")


def test_device_pointers_streams_determinism_and_reset(pkg):
    import torch
    W, H = 384, 216
    p = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1)
    frames = [pkg.synth.render_frame(W, H, f, seed=31, moving=True) for f in range(3)]
    tin = [torch.from_numpy(f[0]).cuda() for f in frames]
    tg = [torch.from_numpy(f[1].view(np.uint8).reshape(-1)).cuda() for f in frames]
    d = pkg.Denoiser(W, H, 0)
    host = [d.denoise_host(f[0], f[1], f[2], p) for f in frames]
    d.reset()                                                       # denoiseFree + denoiseInit
    side = torch.cuda.Stream()
    outs = []
    with torch.cuda.stream(side):
        for k in range(3):
            out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
            d.denoise(out, tin[k], tg[k], frames[k][2], p, stream=side)
            outs.append(out)
    side.synchronize()
    for k in range(3):
        assert np.array_equal(outs[k].cpu().numpy(), host[k]), f"frame {k}: device-pointer/stream path differs or reset is incomplete"
    d.free()


def test_two_contexts_interleaved_on_two_streams(pkg):
    """Contexts are independent (handle-based ABI): two denoisers of different sizes driven alternately on two streams,
    give exactly what each gives alone.  Catches state that leaked into statics (LDS attribute caches, segment heuristics,
    debug buffers) and stream mix-ups."""
    import torch
    sizes = [(1920, 1080), (640, 360)]
    N = 6
    p = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1)
    data = []
    for W, H in sizes:
        frames = [pkg.synth.render_frame(W, H, f, seed=53, moving=True) for f in range(3)]
        data.append(([torch.from_numpy(f[0]).cuda() for f in frames],
                     [torch.from_numpy(f[1].view(np.uint8).reshape(-1)).cuda() for f in frames], [f[2] for f in frames]))
    def run(which, streams):
        ds = {i: pkg.Denoiser(sizes[i][0], sizes[i][1], 0) for i in which}
        outs = {i: [torch.empty((sizes[i][1], sizes[i][0], 3), dtype=torch.float32, device="cuda") for _ in range(N)] for i in which}
        torch.cuda.synchronize()
        for f in range(N):
            for i in which:
                tin, tg, cams = data[i]
                with torch.cuda.stream(streams[i]):
                    ds[i].denoise(outs[i][f], tin[f % 3], tg[f % 3], cams[f % 3], p, stream=streams[i])
        torch.cuda.synchronize()
        res = {i: [o.cpu().numpy() for o in outs[i]] for i in which}
        for d in ds.values():
            d.free()
        return res
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    alone = {**run([0], {0: s0}), **run([1], {1: s1})}
    both = run([0, 1], {0: s0, 1: s1})
    for i in (0, 1):
        for f in range(N):
            assert np.array_equal(both[i][f], alone[i][f]), f"context {i} frame {f} differs when interleaved with the other context"


@pytest.mark.parametrize("history_level", [0, 1, 3, 5])
def test_back_to_back_asynchronous_frames_equal_synchronised_frames(pkg, history_level):
    """Eight frames enqueued back to back on one stream with no host synchronisation in between must give exactly what the
    same frames give when the host waits after every call, for every position of the history level: the plane rotation
    (three colour planes; the fused first level reads the OLD colour history while it writes the new one) must not let a
    frame overwrite what a kernel still in flight reads.  Also: the same frames promised (SvgfParams::inputs_ready = 1) on a
    context created pipelined: bit-identical."""
    import torch
    W, H, N = 1920, 1080, 8
    frames = [pkg.synth.render_frame(W, H, f, seed=37, moving=True) for f in range(4)]
    tin = [torch.from_numpy(f[0]).cuda() for f in frames]
    tg = [torch.from_numpy(f[1].view(np.uint8).reshape(-1)).cuda() for f in frames]
    res = {}
    for mode in ("sync", "async", "async+inputs_ready"):
        p = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, history_level=history_level,
                                         inputs_ready=1 if mode.endswith("ready") else 0)
        d = pkg.Denoiser(W, H, 0, pipelined=mode.endswith("ready"))
        outs = [torch.empty((H, W, 3), dtype=torch.float32, device="cuda") for _ in range(N)]
        stream = torch.cuda.current_stream()
        torch.cuda.synchronize()
        for k in range(N):
            d.denoise(outs[k], tin[k % 4], tg[k % 4], frames[k % 4][2], p, stream=stream)
            if mode == "sync":
                d.sync()
        d.sync()
        res[mode] = ([o.cpu().numpy() for o in outs], d.read_state(0), d.read_state(1), d.read_state(2))
        d.free()
    for mode in ("async", "async+inputs_ready"):
        for k in range(N):
            assert np.array_equal(res["sync"][0][k], res[mode][0][k]), f"frame {k} differs ({mode}, history_level {history_level})"
        for a, b in zip(res["sync"][1:], res[mode][1:]):
            assert np.array_equal(a, b)


def test_error_codes(pkg):
    import ctypes
    d = pkg.Denoiser(64, 64, 0)
    c, g, cam = pkg.synth.render_frame(64, 64, 0)
    with pytest.raises(pkg.SvgfError, match="atrous_nlevel"):
        d.denoise_host(c, g, cam, pkg.reference_defaults().set(spatial_enable=1, atrous_nlevel=11))
    with pytest.raises(pkg.SvgfError, match="strip kernel does not support"):
        d.denoise_host(c, g, cam, pkg.reference_defaults().set(spatial_enable=1, atrous_nlevel=6, kernel_variant=2))
    with pytest.raises(pkg.SvgfError, match="no longer part of the library"):
        d.denoise_host(c, g, cam, pkg.reference_defaults().set(spatial_enable=1, kernel_variant=3))
    for v in (5, 6):       # parked experiments: not in the product build, refused before anything is enqueued
        with pytest.raises(pkg.SvgfError, match="-5.*parked experiment"):
            d.denoise_host(c, g, cam, pkg.reference_defaults().set(spatial_enable=1, kernel_variant=v))
    with pytest.raises(pkg.SvgfError, match="null argument"):
        d.denoise(None, None, None, cam, pkg.reference_defaults())
    out = d.denoise_host(c, g, cam, pkg.reference_defaults())      # the context stays usable after errors
    assert np.array_equal(out, c)
    d.free()
    lib = pkg.load_library()
    h = ctypes.c_void_p()
    assert lib.svgf_create(99, 8, 8, ctypes.byref(h)) == -2
    assert lib.svgf_build_has_experiments() == 0 and not hasattr(lib, "svgf_exp_set")


@pytest.mark.experiments
@pytest.mark.parametrize("knobs", [{"strip_rows": 3}, {"strip_tx": 128}, {"strip_rows": 1}, {"strip_tx": 128, "strip_rows": 1}])
def test_strip_kernel_tuning_configurations_stay_correct(pkg, knobs, experiments_lib):
    """The strip kernel's alternative shapes (12 compute waves, 128-column strips, one row per iteration) exist in the experiments
    build only, behind svgf_exp_set; they must keep giving the reference's result."""
    for k, v in knobs.items():
        experiments_lib.exp_set(k, v)
    for name in ("atrous_rand37x23_n5", "atrous_synth128x72_n5", "atrous_nanpos40x32_n2", "full_static96x54"):
        if name not in CASES:
            continue
        z, runs = load_golden(name)
        W, H = int(z["W"]), int(z["H"])
        for tag in runs:
            if int(z[f"call_params_{tag}"][0][8]) > 5:
                continue
            e = Engine(pkg, W, H, 2)
            got = replay(pkg, e, z, tag)
            e.free()
            err = relerr(got, z[f"ref_nofma_out_{tag}"])
            assert err.max() <= TOL_STRIP, f"{knobs} {name}:{tag} max rel {err.max():.3e}"


def test_1080p_moving_64_frames_full_svgf_every_frame(pkg, orc):
    """BASELINE configs[2] as worded: 1920x1080, 64-frame moving-camera sequence, full SVGF (temporal + 5 levels), the
    library's default kernel selection, cross-frame state carried on both sides.  HIP vs the CPU oracle on EVERY frame:
    <= 1e-4 relative per channel (north_star's bar), and the error does not grow along the sequence (no `tol * (f + 1)`
    allowance): the worst frame of the second half is no worse than twice the worst frame of the first half."""
    import torch
    W, H, N = 1920, 1080, 64
    p = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, atrous_nlevel=5, history_level=1)
    d = pkg.Denoiser(W, H, 0)
    o = orc.Oracle(pkg, W, H, threads=min(64, __import__("os").cpu_count() or 1))
    rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    gb = torch.empty((H * W * 52,), dtype=torch.uint8, device="cuda")
    out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    worst, p9999 = [], []
    for f in range(N):
        cam = pkg.synth.camera_for_frame(f, True)
        pkg.binding.synth_render(rgb, gb, W, H, cam, f, seed=61)          # the device-side producer (bit-identical to synth.py)
        d.denoise(out, rgb, gb, cam, p)
        torch.cuda.synchronize()
        c = rgb.cpu().numpy()
        g = gb.cpu().numpy().view(pkg.synth.GBUFFER_DTYPE).reshape(H, W)
        ref = o.denoise(c, g, cam, p)
        e = relerr(out.cpu().numpy(), ref)
        worst.append(float(e.max())); p9999.append(float(np.quantile(e, 0.9999)))
        assert e.max() <= 1e-4, f"frame {f}: max rel {e.max():.3e}"
    assert np.array_equal(d.read_state(0), o.read_state(0)), "history length after 64 frames"
    d.free(); o.free()
    assert max(worst[32:]) <= 2.0 * max(worst[:32]) + 1e-6, f"error grows along the sequence: {max(worst[:32]):.2e} -> {max(worst[32:]):.2e}"
    print(f"64-frame 1080p moving: worst frame max rel {max(worst):.2e}, worst p99.99 {max(p9999):.2e}")


def test_two_host_threads_two_contexts(pkg):
    """Two host threads, each with its own context and stream, denoise different sequences at the same time (the ABI is
    handle-based: one context per thread; the per-device launch caches are std::call_once-guarded).  Each thread must get
    exactly what it gets alone, and the calling thread's current device is left as it was."""
    import threading
    import torch
    sizes = [(960, 540), (480, 270)]
    N = 5
    p = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1)
    frames = [[pkg.synth.render_frame(W, H, f, seed=71 + i, moving=True) for f in range(N)] for i, (W, H) in enumerate(sizes)]

    def work(i, outs, errs):
        try:
            W, H = sizes[i]
            d = pkg.Denoiser(W, H, 0)
            st = torch.cuda.Stream()
            res = []
            for f in range(N):
                c, g, cam = frames[i][f]
                o = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
                d.denoise(o, torch.from_numpy(c).cuda(), torch.from_numpy(g.view(np.uint8).reshape(-1)).cuda(), cam, p, stream=st)
                res.append(o)
            st.synchronize()
            outs[i] = [r.cpu().numpy() for r in res]
            d.free()
        except Exception as e:      # noqa: BLE001
            errs.append(e)

    alone, errs = {}, []
    for i in range(2):
        work(i, alone, errs)
    assert not errs, errs
    both = {}
    ts = [threading.Thread(target=work, args=(i, both, errs)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for i in range(2):
        for f in range(N):
            assert np.array_equal(both[i][f], alone[i][f]), f"thread {i} frame {f}"
    assert torch.cuda.current_device() == 0


def test_eight_host_threads_eight_contexts(pkg):
    """The shape of the 8-GPU farm inside ONE process, as far as a one-GPU box can take it: eight host threads, each with its own
    context, stream and sequence (four sizes, so different launch geometries, kernel instantiations and per-device launch caches are
    hit concurrently for the first time), with per-kernel profiling armed on every context (the profiler hands its event pair to the
    launcher through a thread-local).  Each thread must get bit for bit what it gets alone; every context reports its own kernels."""
    import threading
    import torch
    sizes = [(960, 540), (480, 270), (1920, 136), (640, 360), (800, 800), (333, 111), (1280, 90), (96, 54)]
    N = 4
    p = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1)
    frames = [[pkg.synth.render_frame(W, H, f, seed=171 + i, moving=True) for f in range(N)] for i, (W, H) in enumerate(sizes)]
    dev_frames = [[(torch.from_numpy(c).cuda(), torch.from_numpy(g.view(np.uint8).reshape(-1)).cuda(), cam) for c, g, cam in fr] for fr in frames]
    start = threading.Barrier(len(sizes))

    def work(i, outs, kinds, errs, together):
        try:
            W, H = sizes[i]
            d = pkg.Denoiser(W, H, 0)
            d.profile_enable(N)
            st = torch.cuda.Stream()
            if together:
                start.wait()
            res = []
            for f in range(N):
                c, g, cam = dev_frames[i][f]
                o = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
                d.denoise(o, c, g, cam, p, stream=st)
                res.append(o)
            d.sync_stream(st)
            outs[i] = [r.cpu().numpy() for r in res]
            kinds[i] = [[k for k, ms in d.profile_read(s)] for s in range(N)]
            assert all(ms > 0 for s in range(N) for k, ms in d.profile_read(s))
            d.free()
        except Exception as e:      # noqa: BLE001
            errs.append((i, e))

    alone, kinds_alone, errs = {}, {}, []
    for i in range(len(sizes)):
        work(i, alone, kinds_alone, errs, False)
    assert not errs, errs
    both, kinds_both = {}, {}
    ts = [threading.Thread(target=work, args=(i, both, kinds_both, errs, True)) for i in range(len(sizes))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for i in range(len(sizes)):
        for f in range(N):
            assert np.array_equal(both[i][f], alone[i][f]), f"thread {i} ({sizes[i]}) frame {f}"
        assert kinds_both[i] == kinds_alone[i] and len(kinds_both[i][0]) == 6, f"thread {i}: kernels recorded {kinds_both[i]}"
    assert torch.cuda.current_device() == 0
