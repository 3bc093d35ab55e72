"""The temporal pass fused into the first a-trous level (svgf_atrous_fused.hip, kernel_variant 6 / the default where the cost
model says so) against the same frames with the temporal pass as its own kernel (kernel_variant 4) and against the CPU oracle.

What must hold:
  * the TEMPORAL arithmetic is the same function calls (svgf_temporal.h): history length, moments, accumulated colour and
    variance are BIT-IDENTICAL between the two paths as long as the colour history they read is (frame 0 always; every frame
    with history_level 0, where the history is the accumulated plane itself — which the fused kernel then has to write);
  * the level's arithmetic is the lane kernel's: outputs agree to summation-order noise (a pixel may sit in a wave of the
    other stage order in the 240-column geometry), <= 2e-6, and with the oracle to the usual 1e-5;
  * halo rows / columns, image borders, odd heights (the second y-phase has one row fewer), images narrower than a strip,
    non-finite G-buffer texels, the planar input path, mode switches between fused and unfused frames on one context.
"""
import numpy as np
import pytest

from conftest import relerr

pytestmark = [pytest.mark.gpu, pytest.mark.experiments]


@pytest.fixture(autouse=True)
def _experiments_build(experiments_lib):
    """kernel_variant 6 (the parked fused temporal + first-level kernel) exists in libsvgf_hip_exp.so only."""
    yield

SIZES = [(320, 180), (257, 131), (1920, 38), (500, 37), (33, 7), (241, 64), (239, 5), (1, 1), (5, 3), (960, 90)]


def _run(pkg, W, H, frames, variant, capture=True, **kw):
    d = pkg.Denoiser(W, H, 0)
    d.set_capture(capture)
    p = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, kernel_variant=variant, **kw)
    res = []
    for c, g, cam in frames:
        out = d.denoise_host(c, g, cam, p)
        st = [d.read_state(k) for k in range(5)] if capture else [d.read_state(k) for k in range(3)]
        res.append((out, st))
    d.free()
    return res


@pytest.mark.parametrize("moving", [False, True])
@pytest.mark.parametrize("size", SIZES, ids=[f"{w}x{h}" for w, h in SIZES])
def test_fused_temporal_state_is_bit_identical_with_history_level_0(pkg, size, moving):
    """history_level 0: the colour history is the accumulated plane, so the temporal state of the two paths can be compared bit
    for bit on every frame (and the fused kernel has to write the accumulated plane)."""
    W, H = size
    frames = [pkg.synth.render_frame(W, H, f, seed=31, moving=moving) for f in range(4)]
    a = _run(pkg, W, H, frames, 4, history_level=0)
    b = _run(pkg, W, H, frames, 6, history_level=0)
    for f in range(len(frames)):
        for k, name in enumerate(("history length", "moments", "colour history", "variance (temporal)", "colour_acc")):
            assert np.array_equal(a[f][1][k], b[f][1][k], equal_nan=True), f"{W}x{H} frame {f}: {name} differs between fused and unfused"
        assert relerr(b[f][0], a[f][0]).max() <= 2e-6, f"{W}x{H} frame {f}: output {relerr(b[f][0], a[f][0]).max():.3e}"


@pytest.mark.parametrize("kw", [dict(), dict(history_level=3, blur_variance=0), dict(atrous_nlevel=1), dict(atrous_nlevel=1, history_level=0),
                                dict(atrous_nlevel=2, history_level=2, sepcolor=1, addcolor=1), dict(color_alpha=0.05, moment_alpha=0.6, sigma_l=1.5),
                                dict(reproj_scale=True)],
                         ids=["defaults", "hist3-noblur", "one-level", "one-level-hist0", "two-levels-modulated", "alphas", "reproj-scale"])
@pytest.mark.parametrize("size", [(320, 180), (257, 131), (200, 200)], ids=["320x180", "257x131", "200x200"])
def test_fused_sequences_match_oracle(pkg, orc, size, kw):
    W, H = size
    kw = dict(kw)
    if kw.pop("reproj_scale", False):
        kw["reproj_scale"] = (float(np.tan(np.radians(45.0)) * W / H), float(np.tan(np.radians(45.0))))
    p = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, kernel_variant=6, **kw)
    d = pkg.Denoiser(W, H, 0)
    o = orc.Oracle(pkg, W, H, threads=8)
    worst = 0.0
    for f in range(6):
        c, g, cam = pkg.synth.render_frame(W, H, f, seed=5, moving=(f >= 2))
        got = d.denoise_host(c, g, cam, p)
        ref = o.denoise(c, g, cam, p)
        e = relerr(got, ref).max()
        worst = max(worst, e)
        assert e <= 1e-5, f"{W}x{H} {kw} frame {f}: {e:.3e}"
        assert np.array_equal(d.read_state(0), o.read_state(0)), f"frame {f}: history length"
        assert relerr(d.read_state(1), o.read_state(1)).max() <= 1e-5, f"frame {f}: moments"
    d.free(); o.free()


def test_fused_with_non_finite_texels_and_all_miss(pkg, orc):
    """NaN / inf normals and positions (the workgroup's careful path), NaN colour (passes through), a frame of misses."""
    W, H = 300, 97
    rng = np.random.default_rng(3)
    p = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, kernel_variant=6)
    d = pkg.Denoiser(W, H, 0)
    o = orc.Oracle(pkg, W, H, threads=8)
    for f in range(4):
        c, g, cam = pkg.synth.render_frame(W, H, f, seed=77, moving=True)
        c = c.copy(); g = g.copy()
        if f == 1:
            for _ in range(12):
                y, x = int(rng.integers(0, H)), int(rng.integers(0, W))
                g["position"][y, x] = np.nan if rng.integers(0, 2) else np.inf
            for _ in range(6):
                y, x = int(rng.integers(0, H)), int(rng.integers(0, W))
                g["normal"][y, x, int(rng.integers(0, 3))] = np.nan
        if f == 2:
            c[int(rng.integers(0, H)), int(rng.integers(0, W)), 1] = np.nan
        if f == 3:
            g["geomId"][:] = -1
        got = d.denoise_host(c, g, cam, p)
        ref = o.denoise(c, g, cam, p)
        assert np.array_equal(np.isnan(got), np.isnan(ref)), f"frame {f}: NaN pattern"
        e = relerr(got, ref)
        assert e.max() <= 2e-5, f"frame {f}: {e.max():.3e}"
        assert np.array_equal(d.read_state(0), o.read_state(0)), f"frame {f}: history length"
    d.free(); o.free()


def test_fused_and_unfused_frames_alternate_on_one_context(pkg, orc):
    """Mode switches: fused frame, temporal-only frame (debug view), non-temporal frame, fused again, unfused lane, fused."""
    W, H = 333, 121
    d = pkg.Denoiser(W, H, 0)
    o = orc.Oracle(pkg, W, H, threads=8)
    seq = [dict(kernel_variant=6), dict(kernel_variant=6, right_view_option=2), dict(kernel_variant=6, temporal_enable=0),
           dict(kernel_variant=6), dict(kernel_variant=4), dict(kernel_variant=0), dict(kernel_variant=6, spatial_enable=0), dict(kernel_variant=6)]
    for f, kw in enumerate(seq):
        base = dict(temporal_enable=1, spatial_enable=1)
        base.update(kw)
        p = pkg.reference_defaults().set(**base)
        c, g, cam = pkg.synth.render_frame(W, H, f, seed=9, moving=True)
        got = d.denoise_host(c, g, cam, p)
        ref = o.denoise(c, g, cam, p)
        assert relerr(got, ref).max() <= 1e-5, f"frame {f} {kw}: {relerr(got, ref).max():.3e}"
        assert np.array_equal(d.read_state(0), o.read_state(0)), f"frame {f}: history length"
    d.free(); o.free()


def test_fused_at_1080p_matches_the_unfused_path_and_the_default_choice_follows_the_cost_model(pkg):
    """1920x1080 (BASELINE configs[1]): kernel_variant 6 runs the fused kernel (one launch fewer per frame) and agrees with
    kernel_variant 4 (temporal pass + lane kernel) to summation-order noise on every frame of a moving sequence.  The default
    (kernel_variant 0) follows the launch-geometry cost model, which at the fused kernel's measured speed
    keeps the temporal pass as its own kernel."""
    import torch
    W, H = 1920, 1080
    outs = {}
    kinds = {}
    for v in (0, 4, 6):
        d = pkg.Denoiser(W, H, 0)
        d.profile_enable(1)
        p = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, kernel_variant=v)
        res = []
        for f in range(4):
            c, g, cam = pkg.synth.render_frame(W, H, f, seed=2, moving=True)
            res.append(d.denoise_host(c, g, cam, p))
        torch.cuda.synchronize()
        kinds[v] = [k for k, _ in d.profile_read(0)]
        outs[v] = res
        d.free()
    assert pkg.binding.KERNEL_FUSED in kinds[6] and pkg.binding.KERNEL_TEMPORAL not in kinds[6], kinds[6]
    assert pkg.binding.KERNEL_TEMPORAL in kinds[4] and pkg.binding.KERNEL_FUSED not in kinds[4], kinds[4]
    assert pkg.binding.KERNEL_TEMPORAL in kinds[0] and pkg.binding.KERNEL_FUSED not in kinds[0], kinds[0]
    for f in range(4):
        e = relerr(outs[6][f], outs[4][f])
        assert e.max() <= 1e-5, f"frame {f}: {e.max():.3e}"
        assert np.array_equal(outs[0][f], outs[4][f]), f"frame {f}: the default choice is the unfused lane path at this size"
