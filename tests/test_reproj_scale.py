"""SURVEY.md §8(f) row f4, first item: fov/aspect-correct reprojection (SvgfParams::reproj_scale).

Not in the reference (its mapping has "no tan(fov), no aspect", src/denoise.cu:202-203, exact only for tan(FOVY)=1 and
W=H).  The oracle carries the same extension, so the parity bar is the hot path's: integer state bit-exact, fp32 state
bit-exact for the temporal pass, <= 1e-4 relative after the a-trous levels.  scale (0,0) must be the reference path."""
import numpy as np
import pytest

from conftest import relerr


def _scales(pkg, W, H):
    plx, ply = pkg.synth._pixel_length(W, H, 45.0)
    return float(plx) * W / 2.0, float(ply) * H / 2.0        # (tan(FOVY) * W / H, tan(FOVY))


def _params(pkg, sx, sy, nlevel=5):
    p = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, atrous_nlevel=nlevel, history_level=1)
    p.reproj_scale[0], p.reproj_scale[1] = sx, sy
    return p


def test_exact_reprojection_keeps_history_at_16_9(pkg, orc):
    """Static camera, 16:9: the reference mapping loses the history of most pixels every frame;
    with reproj_scale every visible surface pixel re-finds itself."""
    W, H, nf = 320, 180, 5
    sx, sy = _scales(pkg, W, H)
    assert sx == pytest.approx(16.0 / 9.0, rel=1e-6) and sy == pytest.approx(1.0, rel=1e-6)
    frames = [pkg.synth.render_frame(W, H, f, seed=4, noise_model="hash") for f in range(nf)]
    hit = frames[0][1]["geomId"] >= 0
    frac = {}
    for name, (ax, ay) in {"reference": (0.0, 0.0), "exact": (sx, sy)}.items():
        o = orc.Oracle(pkg, W, H, threads=4)
        for c, g, cam in frames:
            o.denoise(c, g, cam, _params(pkg, ax, ay, nlevel=1))
        hl = o.read_state(0)
        frac[name] = float(np.count_nonzero(hl[hit] == nf)) / float(np.count_nonzero(hit))
        o.free()
    assert frac["reference"] < 0.35, frac
    assert frac["exact"] > 0.97, frac


def test_zero_scale_is_the_reference_path(pkg, orc):
    W, H = 96, 64
    frames = [pkg.synth.render_frame(W, H, f, seed=2, moving=True, noise_model="hash") for f in range(3)]
    outs = []
    for pr in (pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1), _params(pkg, 0.0, 0.0)):
        o = orc.Oracle(pkg, W, H)
        outs.append([o.denoise(c, g, cam, pr) for c, g, cam in frames])
        o.free()
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,moving", [(320, 180, False), (320, 180, True), (200, 200, True), (257, 131, True)])
def test_hip_matches_oracle_with_reproj_scale(pkg, orc, W, H, moving):
    nf = 4
    sx, sy = _scales(pkg, W, H)
    params = _params(pkg, sx, sy)
    frames = [pkg.synth.render_frame(W, H, f, seed=6, moving=moving, noise_model="hash") for f in range(nf)]
    den = pkg.Denoiser(W, H)
    den.set_capture(True)
    o = orc.Oracle(pkg, W, H, threads=8)
    for f, (c, g, cam) in enumerate(frames):
        got = den.denoise_host(c, g, cam, params)
        ref = o.denoise(c, g, cam, params)
        assert np.array_equal(den.read_state(pkg.binding.STATE_HISTORY_LENGTH), o.read_state(0)), f"history length, frame {f}"
        assert np.array_equal(den.read_state(pkg.binding.STATE_MOMENTS), o.read_state(1)), f"moments, frame {f}"
        assert np.array_equal(den.read_state(pkg.binding.STATE_VARIANCE_TEMPORAL), o.read_state(3)), f"variance, frame {f}"
        e = relerr(got, ref)
        assert float(np.quantile(e, 0.999)) <= 1e-4, f"frame {f}: p99.9 {float(np.quantile(e, 0.999)):.2e}"
    if not moving:   # and it does what it is for
        hit = frames[0][1]["geomId"] >= 0
        hl = den.read_state(pkg.binding.STATE_HISTORY_LENGTH)
        assert np.count_nonzero(hl[hit] == nf) > 0.97 * np.count_nonzero(hit)
    den.free()
    o.free()
