"""Non-temporal mode: the prepare pass (variance = 10, colour copy, G-buffer split; reference EstimateVariance :320-329 + :370)
fused into the first a-trous level (svgf_atrous_fused.hip, FUSED = 3; the default where the first level runs the lane kernel,
forced by kernel_variant 6) against the same frames with the prepare kernel launched on its own (kernel_variant 4) and the oracle.

What must hold: the ring records the loaders build are the ones the unfused loaders would have read back from the planes, so the
OUTPUT IS BIT-IDENTICAL to kernel_variant 4; the split planes are complete (a temporal frame that follows reads the previous
normals / geomIds the fused level wrote, every pixel exactly once: strips, y-phases, segments, halo rows and columns); the colour
plane is written when — and only when — something besides the first level reads it (history_level != 1, capture)."""
import numpy as np
import pytest

from conftest import denoiser_for, relerr

pytestmark = pytest.mark.gpu

SIZES = [(800, 800), (320, 180), (257, 131), (1920, 38), (500, 37), (33, 7), (481, 64), (479, 5), (1, 1), (5, 3), (961, 90)]


def _seq(pkg, W, H, n, moving=True, seed=17):
    return [pkg.synth.render_frame(W, H, f, seed=seed, moving=moving) for f in range(n)]


def _run(pkg, W, H, frames, modes, variant, capture=False, **kw):
    """modes[f] = temporal_enable of frame f.  (kernel_variant 6 — force the fusion whatever the cost model says — exists in the
    experiments build only: the same FUSED = 3 instantiation of the same sources; the product library's own copy runs wherever
    variant 0 chooses the lane kernel for the first level: 800x800, 1920x38, 961x90, 481x64 ... below.)"""
    d = denoiser_for(pkg, W, H, variant)
    d.set_capture(capture)
    res = []
    for (c, g, cam), t in zip(frames, modes):
        p = pkg.reference_defaults().set(temporal_enable=t, spatial_enable=1, kernel_variant=variant, **kw)
        out = d.denoise_host(c, g, cam, p)
        res.append((out, [d.read_state(k) for k in range(5 if capture else 3)]))
    d.free()
    return res


@pytest.mark.parametrize("size", SIZES, ids=[f"{w}x{h}" for w, h in SIZES])
def test_fused_prepare_is_bit_identical_to_the_prepare_kernel_and_its_planes_feed_a_temporal_frame(pkg, size):
    W, H = size
    frames = _seq(pkg, W, H, 4)
    modes = [0, 0, 1, 1]            # two non-temporal frames, then temporal frames that reproject into what they left behind
    a = _run(pkg, W, H, frames, modes, 4)
    for v in (6, 0):
        b = _run(pkg, W, H, frames, modes, v)
        for f in range(4):
            if v == 6 and modes[f]:
                # kernel_variant 6 also fuses the TEMPORAL pass into the first level of the temporal frames; that kernel agrees with
                # the unfused path to summation-order noise (tests/test_fused_gpu.py), its temporal state bit for bit
                assert relerr(b[f][0], a[f][0]).max() <= 5e-6, f"{W}x{H} variant 6 temporal frame {f}: {relerr(b[f][0], a[f][0]).max():.3e}"
                assert np.array_equal(a[f][1][0], b[f][1][0]) and np.array_equal(a[f][1][1], b[f][1][1], equal_nan=True), f"{W}x{H} frame {f}: history length / moments"
                continue
            if v == 0:
                # the automatic choice may run the strip kernel on some levels at this size (kernel_variant 4 forces the lane kernel on
                # all): the same arithmetic in another summation order
                assert relerr(b[f][0], a[f][0]).max() <= 2e-6, f"{W}x{H} variant 0 frame {f}: {relerr(b[f][0], a[f][0]).max():.3e}"
                assert np.array_equal(a[f][1][0], b[f][1][0]) and np.array_equal(a[f][1][1], b[f][1][1], equal_nan=True), f"{W}x{H} frame {f}: history length / moments"
                continue
            assert np.array_equal(a[f][0], b[f][0], equal_nan=True), f"{W}x{H} variant {v} frame {f}: output differs (max rel {relerr(b[f][0], a[f][0]).max():.3e})"
            for k in range(3):
                assert np.array_equal(a[f][1][k], b[f][1][k], equal_nan=True), f"{W}x{H} variant {v} frame {f}: state {k} differs"


@pytest.mark.parametrize("kw", [dict(atrous_nlevel=1), dict(atrous_nlevel=5, history_level=0), dict(atrous_nlevel=3, history_level=2, sepcolor=1, addcolor=1),
                                dict(atrous_nlevel=5, history_level=7, blur_variance=0), dict(atrous_nlevel=2, sigma_l=1.5, sigma_n=0.05)],
                         ids=["one-level", "hist0", "hist2-modulated", "hist-beyond-noblur", "sigmas"])
@pytest.mark.parametrize("size", [(320, 180), (257, 131)], ids=["320x180", "257x131"])
def test_fused_prepare_against_the_oracle_and_the_colour_plane_when_it_is_needed(pkg, orc, size, kw):
    W, H = size
    frames = _seq(pkg, W, H, 4, seed=23)
    modes = [0, 0, 0, 0]      # (kernel_variant 6 would also fuse the temporal pass of temporal frames: tests/test_fused_gpu.py)
    got = _run(pkg, W, H, frames, modes, 6, capture=True, **kw)
    ref = _run(pkg, W, H, frames, modes, 4, capture=True, **kw)
    o = orc.Oracle(pkg, W, H, threads=8)
    for f, ((c, g, cam), t) in enumerate(zip(frames, modes)):
        p = pkg.reference_defaults().set(temporal_enable=t, spatial_enable=1, **kw)
        want = o.denoise(c, g, cam, p)
        assert relerr(got[f][0], want).max() <= 1e-5, f"{kw} frame {f}: vs oracle {relerr(got[f][0], want).max():.3e}"
        assert np.array_equal(got[f][0], ref[f][0], equal_nan=True), f"{kw} frame {f}: output differs from the unfused path"
        for k in range(5):          # history length, moments, colour history, variance, captured colour_acc
            assert np.array_equal(got[f][1][k], ref[f][1][k], equal_nan=True), f"{kw} frame {f}: state {k} differs from the unfused path"
    o.free()


def test_fused_prepare_with_non_finite_texels_and_misses(pkg, orc):
    W, H = 300, 97
    rng = np.random.default_rng(5)
    d = denoiser_for(pkg, W, H, 6)
    o = orc.Oracle(pkg, W, H, threads=8)
    for f in range(3):
        c, g, cam = pkg.synth.render_frame(W, H, f, seed=7, moving=True)
        c = c.copy(); g = g.copy()
        if f == 0:
            for _ in range(10):
                y, x = int(rng.integers(0, H)), int(rng.integers(0, W))
                g["position"][y, x] = np.nan if rng.integers(0, 2) else np.inf
            for _ in range(5):
                y, x = int(rng.integers(0, H)), int(rng.integers(0, W))
                g["normal"][y, x, int(rng.integers(0, 3))] = np.nan
        if f == 1:
            c[int(rng.integers(0, H)), int(rng.integers(0, W)), 2] = np.nan
        if f == 2:
            g["geomId"][:] = -1
        p = pkg.reference_defaults().set(temporal_enable=0, spatial_enable=1, kernel_variant=6)
        got = d.denoise_host(c, g, cam, p)
        ref = o.denoise(c, g, cam, p)
        assert np.array_equal(np.isnan(got), np.isnan(ref)), f"frame {f}: NaN pattern"
        assert relerr(got, ref).max() <= 2e-5, f"frame {f}: {relerr(got, ref).max():.3e}"
    d.free(); o.free()


def test_config1_runs_one_launch_per_frame_and_matches_the_oracle(pkg, orc):
    """BASELINE configs[0]: 800x800, temporal off, one level.  The default choice fuses (no prepare launch), kernel_variant 4 does not."""
    import torch
    W, H = 800, 800
    c, g, cam = pkg.synth.render_frame(W, H, 0, seed=1, moving=False)
    o = orc.Oracle(pkg, W, H, threads=8)
    kinds = {}
    for v in (0, 4):
        d = pkg.Denoiser(W, H, 0)
        d.profile_enable(1)
        p = pkg.reference_defaults().set(temporal_enable=0, spatial_enable=1, atrous_nlevel=1, kernel_variant=v)
        out = d.denoise_host(c, g, cam, p)
        torch.cuda.synchronize()
        kinds[v] = [k for k, _ in d.profile_read(0)]
        if v == 0:
            ref = o.denoise(c, g, cam, p)
            assert relerr(out, ref).max() <= 1e-5, f"{relerr(out, ref).max():.3e}"
            print(f"BASELINE configs[0] (800x800, temporal off, one level) vs oracle: worst max-rel {relerr(out, ref).max():.2e}")
        d.free()
    o.free()
    assert kinds[0] == [pkg.binding.KERNEL_FUSED], kinds[0]
    assert kinds[4] == [pkg.binding.KERNEL_PREPARE, pkg.binding.KERNEL_ATROUS], kinds[4]
