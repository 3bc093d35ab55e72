"""Cross-level reuse of the geometric terms (svgf_atrous_lane_reuse.hip; off by default because it measured slower, on with
svgf_exp_set("reuse", 1) before svgf_create, experiments build only): a lane-kernel level stores four of its pair terms per pixel and the
next level reads them instead of evaluating them.  Same results as without it to rounding (g + c instead of fma(dx, kx,
fma(dn, kn, c))), checked against the CPU oracle at sizes that take the lane kernel on every level, with non-finite texels (the
careful path has to hand on usable terms too) and with the paper's dilation sequence."""
import numpy as np
import pytest

from conftest import relerr

pytestmark = [pytest.mark.gpu, pytest.mark.experiments]


@pytest.mark.parametrize("size,kw", [((1920, 70), dict()), ((300, 200), dict(kernel_variant=4)), ((123, 77), dict(kernel_variant=4, blur_variance=0)),
                                     ((641, 97), dict(kernel_variant=4, paper_steps=1, atrous_nlevel=6)), ((480, 33), dict(kernel_variant=4, atrous_nlevel=3, history_level=2))],
                         ids=["1920x70-auto", "300x200", "123x77-noblur", "641x97-paper-steps", "480x33-three-levels"])
def test_reuse_of_geometric_terms_matches_oracle(pkg, orc, size, kw, experiments_lib):
    experiments_lib.exp_set("reuse", 1)
    W, H = size
    p = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, **kw)
    d = pkg.Denoiser(W, H, 0)
    o = orc.Oracle(pkg, W, H, threads=8)
    rng = np.random.default_rng(11)
    for f in range(4):
        c, g, cam = pkg.synth.render_frame(W, H, f, seed=21, moving=True)
        g = g.copy()
        if f == 2:                                    # non-finite normals / positions: the workgroups that stage them run the careful path
            for _ in range(5):
                g["position"][int(rng.integers(0, H)), int(rng.integers(0, W))] = np.nan
                g["normal"][int(rng.integers(0, H)), int(rng.integers(0, W)), 0] = np.inf
        got = d.denoise_host(c, g, cam, p)
        ref = o.denoise(c, g, cam, p)
        assert np.array_equal(np.isnan(got), np.isnan(ref)), f"{W}x{H} frame {f}: NaN pattern"
        e = relerr(got, ref)
        assert e.max() <= 2e-5, f"{W}x{H} {kw} frame {f}: {e.max():.3e}"
    d.free(); o.free()
