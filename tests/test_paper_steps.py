"""`SvgfParams::paper_steps` ("next" row f4): a-trous level k uses dilation 2^(k-1) = 1, 2, 4, ... as in the
SVGF paper instead of the reference's 2, 4, 8, ... (src/denoise.cu:98,386).  0 keeps the reference behaviour, so every
other test doubles as the regression test of the default."""
import numpy as np
import pytest

from conftest import relerr


def test_oracle_paper_steps_shifts_the_dilation_by_one_level(pkg, orc):
    """CPU: with paper_steps the oracle's level k is the reference's level k-1 kernel; checked through the one-level case
    (step 1 touches the 5x5 neighbourhood only) and through 'n levels with paper steps = step-1 level + (n-1) reference
    levels' on a static frame without temporal feedback."""
    W, H = 48, 36
    c, g, cam = pkg.synth.render_frame(W, H, 0, seed=3, moving=False)
    o = orc.Oracle(pkg, W, H, threads=4)
    p1 = pkg.reference_defaults().set(spatial_enable=1, atrous_nlevel=1, paper_steps=1)
    out1 = o.denoise(c, g, cam, p1)
    o.reset()
    # a pixel whose 5x5 neighbourhood is changed must change, one 3 pixels away must not (step 1 => reach 2)
    c2 = c.copy(); c2[20, 20] += 0.5
    out1b = o.denoise(c2, g, cam, p1)
    o.free()
    changed = np.argwhere(np.abs(out1b - out1).max(axis=2) > 0)
    assert len(changed) > 0
    assert np.abs(changed - np.array([20, 20])).max() <= 2, "step-1 level reached further than 2 pixels"
    # reference stepping reaches 4 pixels with one level
    o = orc.Oracle(pkg, W, H, threads=4)
    p0 = pkg.reference_defaults().set(spatial_enable=1, atrous_nlevel=1)
    a = o.denoise(c, g, cam, p0); o.reset(); b = o.denoise(c2, g, cam, p0); o.free()
    far = np.argwhere(np.abs(b - a).max(axis=2) > 0)
    assert np.abs(far - np.array([20, 20])).max() == 4


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("size", [(320, 180), (257, 131), (1920, 1080)])
def test_hip_paper_steps_match_oracle(pkg, orc, size, variant):
    W, H = size
    frames = 2 if W < 1000 else 1
    p = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, atrous_nlevel=5, paper_steps=1, kernel_variant=variant)
    d = pkg.Denoiser(W, H, 0)
    o = orc.Oracle(pkg, W, H, threads=16)
    for f in range(frames):
        c, g, cam = pkg.synth.render_frame(W, H, f, seed=59, moving=True)
        got = d.denoise_host(c, g, cam, p)
        ref = o.denoise(c, g, cam, p)
        e = relerr(got, ref)
        assert e.max() <= 1e-5 * 4, f"{W}x{H} variant {variant} frame {f}: {e.max():.3e}"      # flat: no growth allowance
        assert np.array_equal(d.read_state(0), o.read_state(0))
    d.free(); o.free()


@pytest.mark.gpu
def test_hip_paper_steps_every_level_count(pkg, orc):
    """1 .. 8 levels with paper steps: dilations 1 .. 128 (lane, strip and lattice kernels), history level = last."""
    W, H = 300, 170
    c, g, cam = pkg.synth.render_frame(W, H, 0, seed=61, moving=False)
    for n in range(1, 9):
        p = pkg.reference_defaults().set(spatial_enable=1, atrous_nlevel=n, history_level=n, paper_steps=1)
        d = pkg.Denoiser(W, H, 0)
        o = orc.Oracle(pkg, W, H, threads=8)
        got = d.denoise_host(c, g, cam, p)
        ref = o.denoise(c, g, cam, p)
        assert relerr(got, ref).max() <= 4e-5, f"{n} levels: {relerr(got, ref).max():.3e}"
        assert relerr(d.read_state(2), o.read_state(2)).max() <= 4e-5
        d.free(); o.free()
