"""Full-size parity against the REFERENCE ITSELF (GPU box): oracle/_ref/ref_denoise_gpu_nofma is the reference's own
src/denoise.cu built for gfx950 (oracle/ref/Makefile).  It is run here on the BASELINE sizes (1920x1080, and 800x800 =
BASELINE config 1) on the same synthetic inputs as the HIP library, and the two outputs are compared directly.

  * config 1 (temporal off, 1 a-trous level) and every temporal-off run are race-free in the reference
    (uniform variance): tolerance 1e-5 relative;
  * the temporal pass (spatial off) is race-free: bit-exact;
  * full SVGF at 1080p: the reference updates `variance` in place (src/denoise.cu:111,117,153,161), so at this size its
    result depends on workgroup scheduling and it does not even reproduce itself: measured on MI355X, two runs of the
    reference binary agree within 1e-4 on 98.0 % of the values of frame 2 (max 8.6e-2).  The HIP result (snapshot
    semantics) sits in the same cloud: 96.6 % within 1e-4 (max 9.4e-2).  The test bounds the disagreement by a small
    multiple of the reference's own run-to-run disagreement.
"""
import os
import struct
import subprocess
import tempfile

import numpy as np
import pytest

from conftest import ROOT, relerr, PARAM_KEYS

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "oracle", "_ref", "ref_denoise_gpu_nofma")
needs_ref = pytest.mark.skipif(not os.path.exists(BIN), reason="reference binary is built only where /root/reference exists")


def run_reference(W, H, frames, cams, calls):
    """calls: list of (reset, frame_index, params dict).  Returns outputs [ncalls, H, W, 3]."""
    with tempfile.TemporaryDirectory() as td:
        case, outp = os.path.join(td, "case.bin"), os.path.join(td, "out.bin")
        with open(case, "wb") as f:
            f.write(struct.pack("<5i", 0x43475653, W, H, len(calls), len(frames)))
            for (reset, fi, p) in calls:
                cam = np.concatenate([np.asarray(cams[fi][k], dtype=np.float32) for k in ("right", "up", "view", "position")])
                f.write(struct.pack("<4i2fi3f5i12fi", int(reset), fi, int(p["temporal_enable"]), int(p["spatial_enable"]),
                                    p["color_alpha"], p["moment_alpha"], int(p["blur_variance"]), p["sigma_l"], p["sigma_x"],
                                    p["sigma_n"], int(p["atrous_nlevel"]), int(p["history_level"]), int(p["sepcolor"]),
                                    int(p["addcolor"]), int(p["right_view_option"]), *[float(v) for v in cam], 0))
            for (c, g) in frames:
                f.write(np.ascontiguousarray(c, dtype="<f4").tobytes())
                f.write(np.ascontiguousarray(g).tobytes())
        r = subprocess.run([BIN, case, outp], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert r.returncode == 0, r.stdout
        raw = np.fromfile(outp, dtype="<f4")
    return raw[: len(calls) * H * W * 3].reshape(len(calls), H, W, 3)


def defaults(**kw):
    p = dict(temporal_enable=0, spatial_enable=0, color_alpha=0.2, moment_alpha=0.2, blur_variance=1, sigma_l=0.45,
             sigma_x=0.35, sigma_n=0.2, atrous_nlevel=5, history_level=1, sepcolor=0, addcolor=0, right_view_option=0)
    p.update(kw)
    return p


def run_hip(pkg, W, H, frames, cams, calls):
    d = pkg.Denoiser(W, H, 0)
    outs = []
    for (reset, fi, p) in calls:
        if reset:
            d.reset()
        pp = pkg.SvgfParams()
        pp.set(**{k: (p[k] if isinstance(p[k], float) else int(p[k])) for k in PARAM_KEYS})
        outs.append(d.denoise_host(frames[fi][0], frames[fi][1], cams[fi], pp))
    d.free()
    return np.stack(outs)


def synth(pkg, W, H, n, moving=False, seed=71):
    fr, cams = [], []
    for f in range(n):
        c, g, cam = pkg.synth.render_frame(W, H, f, seed=seed, moving=moving)
        fr.append((c, g)); cams.append(cam)
    return fr, cams


@needs_ref
def test_baseline_config1_800x800_one_level_no_temporal(pkg):
    fr, cams = synth(pkg, 800, 800, 1)
    calls = [(1, 0, defaults(spatial_enable=1, atrous_nlevel=1))]
    ref = run_reference(800, 800, fr, cams, calls)
    got = run_hip(pkg, 800, 800, fr, cams, calls)
    assert relerr(got, ref).max() <= 1e-5


@needs_ref
def test_1080p_five_levels_no_temporal(pkg):
    fr, cams = synth(pkg, 1920, 1080, 1)
    calls = [(1, 0, defaults(spatial_enable=1, atrous_nlevel=5))]
    ref = run_reference(1920, 1080, fr, cams, calls)
    got = run_hip(pkg, 1920, 1080, fr, cams, calls)
    assert relerr(got, ref).max() <= 1e-5


@needs_ref
def test_1080p_temporal_pass_bit_exact(pkg):
    fr, cams = synth(pkg, 1920, 1080, 3, moving=True)
    calls = [(1 if f == 0 else 0, f, defaults(temporal_enable=1, spatial_enable=0)) for f in range(3)]
    ref = run_reference(1920, 1080, fr, cams, calls)
    got = run_hip(pkg, 1920, 1080, fr, cams, calls)
    assert np.array_equal(got, ref)


@needs_ref
def test_1080p_full_svgf_statistical(pkg):
    fr, cams = synth(pkg, 1920, 1080, 3, moving=False)
    calls = [(1 if f == 0 else 0, f, defaults(temporal_enable=1, spatial_enable=1)) for f in range(3)]
    ref = run_reference(1920, 1080, fr, cams, calls)
    ref2 = run_reference(1920, 1080, fr, cams, calls)
    got = run_hip(pkg, 1920, 1080, fr, cams, calls)
    e0 = relerr(got[0], ref[0])
    assert e0.max() <= 1e-5, "first frame is race-free (uniform variance)"
    e = relerr(got[2], ref[2])
    spread = relerr(ref2[2], ref[2])
    print(f"frame 2: HIP vs reference: frac<=1e-4 {(e <= 1e-4).mean():.5f} max {e.max():.3e}; "
          f"reference run-to-run: frac<=1e-4 {(spread <= 1e-4).mean():.5f} max {spread.max():.3e}")
    bad, bad_ref = float((e > 1e-4).mean()), float((spread > 1e-4).mean())
    assert bad <= max(3.0 * bad_ref, 0.01) + 0.02, (bad, bad_ref)
    assert np.median(e) <= 1e-6 and e.max() <= 0.5
