"""the reference's scene text format (host side, scene.py) and the primitive-driven producer
(csrc/svgf_scene.hip) against its numpy oracle (scene.render_scene).  Bar: geomId exact, fp32 fields bit-exact up to
1e-5 of the pixels (the kernel mirrors the numpy arithmetic operation for operation)."""
import os

import numpy as np
import pytest

from conftest import ROOT

SCENE = os.path.join(ROOT, "tests", "golden", "scenes", "box_room.txt")


def _scene(pkg):
    return pkg.scene.parse_scene(open(SCENE).read())


def test_parse_scene_text_format(pkg):
    sc = _scene(pkg)
    assert sorted(sc.materials) == [0, 1, 2, 3, 4]
    assert sc.materials[0]["emittance"] == 4.0 and sc.materials[2]["rgb"] == (0.85, 0.45, 0.20)
    assert sc.camera["res"] == (640, 360) and sc.camera["fovy"] == 45.0 and sc.camera["eye"] == (0.0, 4.5, 11.0)
    assert [o["type"] for o in sc.objects] == ["cube"] * 6 + ["sphere", "cube", "sphere", "mesh"]
    assert sc.objects[7]["rotat"] == (0.0, 27.5, 0.0) and sc.objects[9]["file"] == "somewhere.obj"
    assert sc.skipped == [9]
    with pytest.raises(ValueError):
        pkg.scene.parse_scene("OBJECT 3\ncube\nmaterial 0\nTRANS 0 0 0\nROTAT 0 0 0\nSCALE 1 1 1\n")


def test_transforms_are_consistent(pkg):
    g = pkg.scene.geom_array(_scene(pkg))
    assert len(g) == 9 and g.dtype.itemsize == 156
    for rec in g:
        xf = np.vstack([rec["xf"].reshape(3, 4), [0, 0, 0, 1]]).astype(np.float64)
        inv = np.vstack([rec["inv"].reshape(3, 4), [0, 0, 0, 1]]).astype(np.float64)
        assert np.allclose(xf @ inv, np.eye(4), atol=2e-6)
        assert np.allclose(rec["invT"].reshape(3, 3), inv[:3, :3].T, atol=0)
    # T * Rx * Ry * Rz * S against a float64 composition
    o = _scene(pkg).objects[8]
    ax, ay, az = np.radians(o["rotat"])
    Rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    Ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
    Rz = np.array([[np.cos(az), -np.sin(az), 0], [np.sin(az), np.cos(az), 0], [0, 0, 1]])
    M = Rx @ Ry @ Rz @ np.diag(o["scale"])
    assert np.allclose(g[8]["xf"].reshape(3, 4)[:, :3], M, atol=1e-6)
    assert np.allclose(g[8]["xf"].reshape(3, 4)[:, 3], o["trans"], atol=0)
    assert np.allclose(pkg.scene.light_position(g), [0, 8.9, 0])


def test_oracle_render_is_sane(pkg):
    sc = _scene(pkg)
    g = pkg.scene.geom_array(sc)
    cam = pkg.scene.camera_for_frame(sc, 0, False)
    assert np.allclose(cam["position"], sc.camera["eye"], atol=1e-5)
    col, gb = pkg.scene.render_scene(160, 90, 0, g, cam)
    ids = set(np.unique(gb["geomId"]).tolist())
    assert {1, 3, 4, 5, 6, 7, 8} <= ids and ids <= set(range(-1, 9))
    hit = gb["geomId"] >= 0
    assert np.allclose(np.linalg.norm(gb["normal"][hit], axis=-1), 1.0, atol=1e-5)
    ball = gb["geomId"] == 6                       # points of the ball lie on its surface
    assert np.allclose(np.linalg.norm(gb["position"][ball] - np.array([-2, 1.6, 1], np.float32), axis=-1), 1.6, atol=1e-4)
    assert np.all(col[~hit] == 0) and np.isfinite(col).all()


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,frame,moving", [(160, 90, 0, False), (333, 177, 7, True), (640, 360, 31, True)])
def test_device_scene_producer_matches_numpy(pkg, W, H, frame, moving):
    import torch
    sc = _scene(pkg)
    g = pkg.scene.geom_array(sc)
    cam = pkg.scene.camera_for_frame(sc, frame, moving)
    rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    gbt = torch.empty((H * W * 52,), dtype=torch.uint8, device="cuda")
    pkg.binding.scene_render(rgb, gbt, W, H, cam, g, frame, seed=5)
    torch.cuda.synchronize()
    got_rgb = rgb.cpu().numpy()
    got = gbt.cpu().numpy().view(pkg.synth.GBUFFER_DTYPE).reshape(H, W)
    ref_rgb, ref = pkg.scene.render_scene(W, H, frame, g, cam, seed=5)
    n = W * H
    lim = max(2, n // 20000)
    assert int(np.count_nonzero(got["geomId"] != ref["geomId"])) <= lim
    same = got["geomId"] == ref["geomId"]
    for f in ("normal", "position", "albedo", "ialbedo"):
        bad = (got[f] != ref[f]).any(axis=-1) & same
        assert int(np.count_nonzero(bad)) <= lim, f"{f}: {int(np.count_nonzero(bad))} pixels differ"
        assert np.allclose(got[f][same], ref[f][same], rtol=1e-5, atol=1e-5)
    bad = (got_rgb != ref_rgb).any(axis=-1) & same
    assert int(np.count_nonzero(bad)) <= lim
    assert np.allclose(got_rgb[same], ref_rgb[same], rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_scene_sequence_through_the_denoiser(pkg, orc):
    import torch
    sc = _scene(pkg)
    g = pkg.scene.geom_array(sc)
    W, H = 192, 108
    den = pkg.Denoiser(W, H)
    params = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, atrous_nlevel=5, history_level=1)
    engine = orc.Oracle(pkg, W, H, threads=8)
    rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    gbt = torch.empty((H * W * 52,), dtype=torch.uint8, device="cuda")
    out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    s = torch.cuda.current_stream()
    for f in range(3):
        cam = pkg.scene.camera_for_frame(sc, f, True)
        pkg.binding.scene_render(rgb, gbt, W, H, cam, g, f, seed=9, stream=s)
        den.denoise(out, rgb, gbt, cam, params, stream=s)
        torch.cuda.synchronize()
        c = rgb.cpu().numpy()
        gb = gbt.cpu().numpy().view(pkg.synth.GBUFFER_DTYPE).reshape(H, W)
        ref = engine.denoise(c, gb, cam, params)          # oracle fed with the very frames the device produced
        err = np.abs(out.cpu().numpy() - ref) / np.maximum(np.abs(ref), 1e-2)
        assert float(np.quantile(err, 0.999)) <= 1e-4, f"frame {f}"
    den.free()
