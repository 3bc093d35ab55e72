"""CPU tests of the oracle's own behaviour (the reference quirks SURVEY.md §8a lists) and of host-side logic."""
import numpy as np
import pytest

from conftest import relerr


def _seq(pkg, W, H, n, moving=False, seed=3):
    return [pkg.synth.render_frame(W, H, f, seed=seed, moving=moving) for f in range(n)]


def test_view_matrix_is_the_inverse(pkg, orc):
    for f in (0, 7):
        cam = pkg.synth.camera_for_frame(f, moving=True)
        m = orc.view_matrix(pkg, cam).reshape(4, 4).T.astype(np.float64)     # column-major -> math layout
        M = np.eye(4)
        M[:3, 0], M[:3, 1], M[:3, 2], M[:3, 3] = cam["right"], cam["up"], cam["view"], cam["position"]
        assert np.abs(m @ M - np.eye(4)).max() < 2e-5


def test_square_static_camera_accumulates_history_and_16x9_mostly_does_not(pkg, orc):
    """Reference quirk: reprojection omits tan(fov)/aspect (src/denoise.cu:202-207) -> exact only for square FOVY=45."""
    p = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=0)
    for (W, H, lo, hi) in ((80, 80, 0.95, 1.0), (128, 72, 0.05, 0.45)):
        o = orc.Oracle(pkg, W, H, threads=4)
        for (c, g, cam) in _seq(pkg, W, H, 4):
            o.denoise(c, g, cam, p)
        hl = o.read_state(0)
        hit = g["geomId"] >= 0
        frac = (hl[hit] > 1).mean()
        o.free()
        assert lo <= frac <= hi, (W, H, frac)


def test_history_length_counts_frames_uncapped(pkg, orc):
    W = H = 48
    p = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=0)
    o = orc.Oracle(pkg, W, H, threads=2)
    fr = _seq(pkg, W, H, 1)[0]
    for _ in range(7):
        o.denoise(*fr, p)
    hl = o.read_state(0)
    o.free()
    assert hl.max() == 7 and hl.min() >= 1


def test_atrous_steps_start_at_two_and_constant_image_is_a_fixed_point(pkg, orc):
    W, H = 40, 30
    c, g = pkg.synth.random_frame(W, H, seed=1)
    c[...] = 0.25
    cam = pkg.synth.camera_for_frame(0, False)
    p = pkg.reference_defaults().set(spatial_enable=1, atrous_nlevel=5)
    o = orc.Oracle(pkg, W, H)
    out = o.denoise(c, g, cam, p)
    o.free()
    assert relerr(out, c).max() < 1e-6
    # level 1 uses step 2: an impulse at (10,10) spreads to (12,10) but never to (11,10)
    c2 = np.zeros((H, W, 3), np.float32)
    c2[10, 10] = 1.0
    g2 = g.copy()
    g2["normal"] = (0, 0, 1); g2["position"] = 0.0
    p1 = pkg.reference_defaults().set(spatial_enable=1, atrous_nlevel=1, sigma_l=1e6)
    o = orc.Oracle(pkg, W, H)
    out = o.denoise(c2, g2, cam, p1)
    o.free()
    assert out[10, 12, 0] > 0 and out[10, 11, 0] == 0 and out[12, 10, 0] > 0 and out[11, 10, 0] == 0


def test_debug_views_and_passthrough(pkg, orc):
    W, H = 32, 24
    c, g, cam = pkg.synth.render_frame(W, H, 0, seed=2)
    o = orc.Oracle(pkg, W, H)
    out = o.denoise(c, g, cam, pkg.reference_defaults().set(temporal_enable=1, right_view_option=2))
    assert np.allclose(out, 100.0 / 0.1)                         # no history -> variance 100, shown / 0.1
    out = o.denoise(c, g, cam, pkg.reference_defaults().set(temporal_enable=1, right_view_option=1))
    assert np.allclose(out, 1.0 / 100.0)                         # PRE-update lengths of this frame = 1
    out = o.denoise(c, g, cam, pkg.reference_defaults().set(temporal_enable=0, spatial_enable=0))
    assert np.array_equal(out, c)                                # pass-through copies the input
    o.free()


def test_history_level_selects_the_level_fed_back(pkg, orc):
    W = H = 48
    c, g, cam = pkg.synth.render_frame(W, H, 0, seed=4)
    outs = {}
    for hl in (0, 1, 3):
        o = orc.Oracle(pkg, W, H, threads=2)
        o.denoise(c, g, cam, pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, atrous_nlevel=3, history_level=hl))
        outs[hl] = (o.read_state(2), o.read_state(4))            # colour history, colour_acc
        o.free()
    assert np.array_equal(outs[0][0], outs[0][1])                # level 0: history = temporal output
    o = orc.Oracle(pkg, W, H, threads=2)
    final = o.denoise(c, g, cam, pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, atrous_nlevel=3, history_level=3))
    o.free()
    assert np.array_equal(outs[3][0], final)                     # level == nlevel: history = final output
    assert not np.array_equal(outs[1][0], outs[0][0])


def test_shard_covers_every_sequence_once(pkg):
    for n in (0, 1, 7, 8, 64):
        for ws in (1, 2, 4, 8):
            got = sorted(i for r in range(ws) for i in pkg.farm.shard(n, ws, r))
            assert got == list(range(n))
    with pytest.raises(ValueError):
        pkg.farm.shard(4, 2, 2)
