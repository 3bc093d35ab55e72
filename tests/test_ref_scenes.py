"""Real-scene fixtures (SURVEY.md §8c fixtures 1-3): the reference's OWN path tracer rendered its own scenes
(scenes/cornell.txt, scenes/room.txt — textured OBJ meshes, mesh normals, stale normal/albedo on misses,
src/pathtrace.cu:267-272,316-323) on an MI355X and handed 1-spp colour + G-buffer to denoise(); the reference's own
denoiser (src/denoise.cu) produced the expected outputs.  Made by tests/golden/make_ref_scene_goldens.py from the
binaries of oracle/ref/Makefile (hipify-perl builds: a translated build of the reference on another runtime, see
DESIGN.md §3).  The same files carry the goldens of the rows next to the path: what the reference's
sendTwoImagesToPBO packs (f2), what its image::savePNG writes (f2), what its Scene loader parses (f3).
"""
import json
import os
import struct
import zlib

import numpy as np
import pytest

from conftest import ROOT, relerr, replay

DIR = os.path.join(ROOT, "tests", "golden", "ref_scenes")
CASES = sorted(f[:-4] for f in os.listdir(DIR) if f.endswith(".npz") and not f.endswith("_producer_inputs.npz"))


def test_fixtures_present():
    assert CASES == ["bunny128x72_static", "cornell128x72_moving", "cornell96_static", "room128x72_static_sepcolor"]
    z = np.load(os.path.join(DIR, "cornell96_static.npz"))
    g = z["gbuffer"]
    assert g.dtype.itemsize == 52 and set(np.unique(g["geomId"])) >= {-1, 0, 3}      # misses, the light, the mesh
    assert np.all(g["ialbedo"] == 1.0)                                                   # src/pathtrace.cu:321


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_on_real_scenes(pkg, orc, name):
    """Against the -ffp-contract=off build of the reference: every frame of the 4-frame full-SVGF sequence <= 2e-6
    (measured 4e-7 .. 8e-7).  Against the default-flags build the reference differs from ITSELF by up to 0.85 relative
    on the room scene (contraction flips reprojection decisions, 10.6 % of the values move by > 1e-4): that build is
    only required to agree on frame 0, which has no history."""
    z = np.load(os.path.join(DIR, name + ".npz"))
    o = orc.Oracle(pkg, int(z["W"]), int(z["H"]), threads=4)
    got = replay(pkg, o, z, "out")
    o.free()
    for f in range(got.shape[0]):
        e = relerr(got[f], z["ref_nofma_out_out"][f])
        assert e.max() <= 2e-6, f"{name} frame {f}: max rel {e.max():.3e}"
    assert relerr(got[0], z["ref_out_out"][0]).max() <= 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 1, 2, 4])
@pytest.mark.parametrize("name", CASES)
def test_hip_matches_reference_on_real_scenes(pkg, name, variant):
    from test_parity_gpu import Engine, TOL_GATHER, TOL_STRIP
    z = np.load(os.path.join(DIR, name + ".npz"))
    e = Engine(pkg, int(z["W"]), int(z["H"]), variant)
    got = replay(pkg, e, z, "out")
    e.free()
    tol = TOL_GATHER if variant == 1 else TOL_STRIP
    for f in range(got.shape[0]):
        err = relerr(got[f], z["ref_nofma_out_out"][f])
        assert err.max() <= tol, f"{name} v{variant} frame {f}: max rel {err.max():.3e}"


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_display_pack_matches_reference_pbo(pkg, name):
    """svgf_display_pack against the bytes the reference's sendTwoImagesToPBO (src/pathtrace.cu:46-78) wrote for the
    same two float images: left = the path-traced frame, right = a pattern with NaN, +-inf, negatives, values > 1 and
    values on byte boundaries."""
    import torch
    z = np.load(os.path.join(DIR, name + ".npz"))
    W, H = int(z["W"]), int(z["H"])
    for f in range(z["color"].shape[0]):
        pbo = torch.zeros((H, 2 * W, 4), dtype=torch.uint8, device="cuda")
        pkg.binding.display_pack(pbo, torch.from_numpy(z["color"][f].copy()).cuda(), torch.from_numpy(z["pattern"][f].copy()).cuda(), W, H)
        torch.cuda.synchronize()
        assert np.array_equal(pbo.cpu().numpy(), z["pbo"][f]), f"{name} frame {f}"


def _png_pixels(path):
    """8-bit RGB PNG reader with all five filter types (the reference writes through stb_image_write)."""
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    off, idat, hdr = 8, b"", None
    while off < len(b):
        n, tag = struct.unpack(">I4s", b[off:off + 8])
        data = b[off + 8:off + 8 + n]
        assert struct.unpack(">I", b[off + 8 + n:off + 12 + n])[0] == (zlib.crc32(tag + data) & 0xFFFFFFFF)
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", data)
        elif tag == b"IDAT":
            idat += data
        off += 12 + n
    w, h, depth, ctype = hdr[:4]
    assert (depth, ctype) == (8, 2)
    raw = zlib.decompress(idat)
    stride = 3 * w
    out = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    for y in range(h):
        ft = raw[y * (stride + 1)]
        line = np.frombuffer(raw, np.uint8, stride, y * (stride + 1) + 1).astype(np.int32)
        cur = np.zeros(stride, np.int32)
        for i in range(stride):
            a = cur[i - 3] if i >= 3 else 0
            bb = prev[i]
            c = prev[i - 3] if i >= 3 else 0
            if ft == 0: p = 0
            elif ft == 1: p = a
            elif ft == 2: p = bb
            elif ft == 3: p = (a + bb) // 2
            else:
                pa, pb, pc = abs(bb - c), abs(a - c), abs(a + bb - 2 * c)
                p = a if (pa <= pb and pa <= pc) else (bb if pb <= pc else c)
            cur[i] = (line[i] + p) & 255
        out[y] = cur
        prev = cur
    return out.reshape(h, w, 3)


def test_save_png_matches_reference_png(pkg, tmp_path):
    """svgf_save_png against the file the reference's saveImage() + image::savePNG (src/main.cpp:131-152,
    src/image.cpp:22-39) wrote for the same 37x5 float pattern (x mirror, clamp, truncation; NaN, +-inf, byte boundaries)."""
    W, H = 37, 5
    pat = np.fromfile(os.path.join(DIR, "ref_savepng_37x5.f32"), "<f4").reshape(H, W, 3)
    want = _png_pixels(os.path.join(DIR, "ref_savepng_37x5.png"))
    path = str(tmp_path / "ours.png")
    pkg.binding.save_png(path, pat, mirror_x=True)
    got = _png_pixels(path)
    assert want.shape == (H, W, 3)
    # NaN: the reference's glm::clamp(NaN, 0, 1) is min(max(NaN, 0), 1) with `a < b ? b : a` selects -> NaN survives the
    # clamp and (unsigned char)(NaN * 255.f) is undefined behaviour in C++; every other value must agree exactly
    finite = ~np.isnan(pat[:, ::-1])
    assert np.array_equal(got[finite], want[finite])


REF_SCENES_DIR = "/root/reference/scenes"


@pytest.mark.skipif(not os.path.isdir(REF_SCENES_DIR), reason="the reference's scene files exist in the build container only")
@pytest.mark.parametrize("scene", ["cornell", "room", "bunny", "diamond"])
def test_scene_parser_matches_reference_loader(pkg, scene):
    """scene.parse_scene / geom_array against what the reference's own Scene loader (src/scene.cpp) parsed from its four
    scene files (tests/golden/ref_scenes/scene_records.json, dumped by oracle/_ref/ref_host_tools)."""
    rec = json.load(open(os.path.join(DIR, "scene_records.json")))[scene]
    sc = pkg.scene.parse_scene(open(os.path.join(REF_SCENES_DIR, scene + ".txt")).read())
    assert tuple(rec["camera"]["resolution"]) == sc.camera["res"]
    assert np.allclose(rec["camera"]["position"], sc.camera["eye"], atol=0) and np.allclose(rec["camera"]["lookAt"], sc.camera["lookat"], atol=0)
    assert np.allclose(rec["camera"]["up"], sc.camera["up"], atol=0) and np.float32(rec["camera"]["fov"][1]) == np.float32(sc.camera["fovy"])
    assert len(rec["materials"]) == len(sc.materials) and len(rec["geoms"]) == len(sc.objects)
    for i, m in enumerate(rec["materials"]):
        assert np.allclose(m["color"], np.float32(sc.materials[i]["rgb"]), atol=0) and np.float32(m["emittance"]) == np.float32(sc.materials[i]["emittance"])
    for g, o in zip(rec["geoms"], sc.objects):
        assert g["type"] == o["type"] and g["materialid"] == o["material"]
        assert np.allclose(g["translation"], np.float32(o["trans"]), atol=0) and np.allclose(g["rotation"], np.float32(o["rotat"]), atol=0)
        assert np.allclose(g["scale"], np.float32(o["scale"]), atol=0)
    # primitive transforms: glm's T * Rx * Ry * Rz * S (column-major 4x4) against geom_array's 3x4 row-major records
    prim = [g for g in rec["geoms"] if g["type"] in ("cube", "sphere")]
    arr = pkg.scene.geom_array(sc)
    assert len(arr) == len(prim)
    for g, r in zip(prim, arr):
        M = np.array(g["transform"], np.float64).reshape(4, 4).T          # glm stores columns
        Mi = np.array(g["inverseTransform"], np.float64).reshape(4, 4).T
        assert np.allclose(r["xf"].reshape(3, 4), M[:3], rtol=2e-6, atol=2e-6)
        assert np.allclose(r["inv"].reshape(3, 4), Mi[:3], rtol=2e-5, atol=2e-6)
        assert r["material"] == g["materialid"]


@pytest.mark.skipif(not os.path.isdir(REF_SCENES_DIR), reason="the reference's scene / model / texture files exist in the build container only")
@pytest.mark.parametrize("scene", ["cornell", "room", "bunny", "diamond"])
def test_obj_loader_matches_reference_triangles(pkg, scene, tmp_path):
    """mesh.scene_triangles (our OBJ reader + Scene::loadMesh's transforms) against the triangle list the reference's own
    loader built (ref_host_tools mesh, i.e. tinyobjloader + src/scene.cpp:234-311): the same triangles, corner for corner,
    as a set — the reference's loader emits shapes in another order, which only changes triangle ids."""
    import subprocess
    tool = os.path.join(ROOT, "oracle", "_ref", "ref_host_tools")
    if not os.path.exists(tool):
        pytest.skip("oracle/_ref/ref_host_tools not built")
    from importlib import import_module
    mesh = import_module(pkg.__name__ + ".mesh")
    out = str(tmp_path / "m.bin")
    subprocess.run([tool, "mesh", scene + ".txt", out], cwd=os.path.join(ROOT, "oracle", "_ref", "scenes"), check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    raw = open(out, "rb").read()
    ntri, ng, _ = struct.unpack_from("<3i", raw, 0)
    ref = np.frombuffer(raw, "<f4", ntri * 24, 12 + ng * 12).reshape(ntri, 24)
    sc = pkg.scene.parse_scene(open(os.path.join(REF_SCENES_DIR, scene + ".txt")).read())
    tp, tn, tu, tobj = mesh.scene_triangles(sc, os.path.join(REF_SCENES_DIR, "Models"))
    assert len(tp) == ntri
    ours = np.concatenate([tp, tn, tu], axis=-1).reshape(ntri, 24)
    key = lambda a: a[np.lexsort(np.round(a[:, ::-1].astype(np.float64), 4).T)]          # noqa: E731
    assert np.allclose(key(ours), key(ref), rtol=2e-6, atol=2e-6)


@pytest.mark.skipif(not os.path.isdir(REF_SCENES_DIR), reason="the reference's scene / model / texture files exist in the build container only")
@pytest.mark.parametrize("name,scene", [("cornell96_static", "cornell"), ("cornell128x72_moving", "cornell"), ("room128x72_static_sepcolor", "room")])
def test_first_hit_with_meshes_matches_reference_gbuffer(pkg, name, scene):
    """mesh.first_hit_gbuffer — primitives + triangle meshes + textures — against the G-buffer the reference's own path
    tracer wrote for the same camera (src/pathtrace.cu:316-323): geomId per pixel, world position, interpolated mesh
    normal (with the reference's corner weights), material / texture albedo (JPEG decoders differ by <= 2/255)."""
    from importlib import import_module
    mesh = import_module(pkg.__name__ + ".mesh")
    z = np.load(os.path.join(DIR, name + ".npz"))
    W, H = int(z["W"]), int(z["H"])
    sc = pkg.scene.parse_scene(open(os.path.join(REF_SCENES_DIR, scene + ".txt")).read())
    tris = mesh.scene_triangles(sc, os.path.join(REF_SCENES_DIR, "Models"))
    tex = mesh.load_textures(sc, os.path.join(REF_SCENES_DIR, "Textures"))
    for f in (0, z["cams"].shape[0] - 1):
        c = z["cams"][f]
        cam = dict(right=c[0:3].astype(np.float32), up=c[3:6].astype(np.float32), view=c[6:9].astype(np.float32),
                   position=c[9:12].astype(np.float32), fovy_deg=sc.camera["fovy"])
        with np.errstate(all="ignore"):
            gb = mesh.first_hit_gbuffer(W, H, sc, cam, tris, tex)
        ref = z["gbuffer"][f]
        same = gb["geomId"] == ref["geomId"]
        assert same.mean() >= 0.995, f"{name} frame {f}: geomId agrees on {same.mean():.4f}"
        hit = same & (ref["geomId"] >= 0)
        assert np.abs(gb["position"][hit] - ref["position"][hit]).max() <= 1e-3
        assert np.abs(gb["normal"][hit] - ref["normal"][hit]).max() <= 1e-3
        assert np.abs(gb["albedo"][hit] - ref["albedo"][hit]).max() <= 2.01 / 255.0
        miss = same & (ref["geomId"] < 0)
        if miss.any():
            assert np.abs(gb["position"][miss] - ref["position"][miss]).max() <= 1e-5          # origin - direction (:317)


@pytest.mark.gpu
@pytest.mark.parametrize("name,scene", [("cornell96_static", "cornell"), ("cornell128x72_moving", "cornell"), ("room128x72_static_sepcolor", "room")])
def test_device_producer_with_meshes_matches_reference_gbuffer(pkg, name, scene):
    """svgf_scene_render_mesh — the device-side producer with the scene's primitives AND triangle meshes
    (tests/golden/ref_scenes/<scene>_producer_inputs.npz: scene.geom_array + mesh.scene_triangles of the reference's files) —
    against the G-buffer the reference's own path tracer wrote for the same camera: geomId, world position, normal (mesh
    normals with the reference's corner weights), albedo of untextured objects; then the produced frame goes through the
    denoiser like any other."""
    import torch
    z = np.load(os.path.join(DIR, name + ".npz"))
    pi = np.load(os.path.join(DIR, scene + "_producer_inputs.npz"))
    W, H = int(z["W"]), int(z["H"])
    rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    gbt = torch.empty((H * W * 52,), dtype=torch.uint8, device="cuda")
    den = pkg.Denoiser(W, H, 0)
    p = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1)
    for f in range(z["cams"].shape[0]):
        c = z["cams"][f]
        cam = dict(right=c[0:3].astype(np.float32), up=c[3:6].astype(np.float32), view=c[6:9].astype(np.float32),
                   position=c[9:12].astype(np.float32), fovy_deg=float(pi["fovy"]))
        textures = [pi[k] for k in sorted(pi.files) if k.startswith("texture") and k != "textured_objects"] if "tri_tex" in pi.files else None
        pkg.binding.scene_render_mesh(rgb, gbt, W, H, cam, pi["geoms"], pi["geom_ids"], pi["tris"], pi["tri_ids"], pi["tri_albedo"], frame=f,
                                      tri_tex=pi["tri_tex"] if textures else None, textures=textures)
        gb = gbt.cpu().numpy().view(pkg.synth.GBUFFER_DTYPE).reshape(H, W)
        ref = z["gbuffer"][f]
        same = gb["geomId"] == ref["geomId"]
        assert same.mean() >= 0.995, f"{name} frame {f}: geomId agrees on {same.mean():.4f}"
        hit = same & (ref["geomId"] >= 0)
        assert np.abs(gb["position"][hit] - ref["position"][hit]).max() <= 1e-3
        assert np.abs(gb["normal"][hit] - ref["normal"][hit]).max() <= 1e-3
        plain = hit & ~np.isin(ref["geomId"], pi["textured_objects"])
        assert np.abs(gb["albedo"][plain] - ref["albedo"][plain]).max() <= 1e-6
        if textures:            # textured mesh: Texture::getColor at the interpolated uv; PIL and stb decode the JPEG within 2 levels
            tex_px = hit & np.isin(ref["geomId"], pi["textured_objects"])
            assert tex_px.any() and np.abs(gb["albedo"][tex_px] - ref["albedo"][tex_px]).max() <= 2.01 / 255.0
        miss = same & (ref["geomId"] < 0)
        if miss.any():
            assert np.abs(gb["position"][miss] - ref["position"][miss]).max() <= 1e-5
        col = rgb.cpu().numpy()
        assert np.isfinite(col).all() and (col[gb["geomId"] < 0] == 0).all()
        out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
        den.denoise(out, rgb, gbt, cam, p)
        torch.cuda.synchronize()
        assert np.isfinite(out.cpu().numpy()).all()
    den.free()


def test_camera_automation_reproduces_the_reference_cameras(pkg):
    """f3, second half: `scene.camera_for_frame(moving=True)` is runCuda's camera automation (src/main.cpp:156-190) — phases
    ACCUMULATED in fp32, one addition per frame, advanced before use, orbit radius |EYE - LOOKAT|.  The cameras the reference's
    own path tracer used for the moving fixture were captured with the frames (`cams` = right | up | view | position, written
    by oracle/ref/ref_pt_capture.cpp from scene->state.camera): reproduced to fp32 rounding."""
    rec = json.load(open(os.path.join(DIR, "scene_records.json")))["cornell"]["camera"]
    sc = pkg.scene.Scene(materials={}, objects=[], camera=dict(eye=rec["position"], lookat=rec["lookAt"], fovy=rec["fov"][1]), skipped=[])
    z = np.load(os.path.join(DIR, "cornell128x72_moving.npz"))
    for f in range(z["cams"].shape[0]):
        cam = pkg.scene.camera_for_frame(sc, f, True)
        got = np.concatenate([cam["right"], cam["up"], cam["view"], cam["position"]]).astype(np.float32)
        assert np.abs(got - z["cams"][f]).max() <= 4e-7, f"frame {f}: {np.abs(got - z['cams'][f]).max():.3e}"
    zs = np.load(os.path.join(DIR, "cornell96_static.npz"))
    cam = pkg.scene.camera_for_frame(sc, 0, False)
    got = np.concatenate([cam["right"], cam["up"], cam["view"], cam["position"]]).astype(np.float32)
    # static: the reference rebuilds EYE from (zoom, theta, phi) through acos / sin / cos: one ulp of a coordinate of 5 .. 10.5
    assert np.abs(got - zs["cams"][0]).max() <= 1e-6


@pytest.mark.gpu
def test_room_1080p_moving_64_frames_device_producer_to_denoiser_vs_oracle(pkg, orc):
    """BASELINE configs[2] as worded — room.txt (textured OBJ meshes) at 1920x1080, moving camera, 64-frame sequence, full SVGF:
    the scene's primitives and 2 810 triangles (tests/golden/ref_scenes/room_producer_inputs.npz, made from the reference's files
    by tests/golden/make_producer_inputs.py) are ray-cast by the device producer (svgf_scene_render_mesh, no BVH: every triangle
    per pixel) with the reference's camera automation, the frames go straight into svgf_denoise on the device, and the CPU
    oracle gets the same 64 frames: <= 1e-4 relative per channel on EVERY frame (north_star's bar), no growth along the sequence."""
    import torch
    W, H, N = 1920, 1080, 64
    pi = np.load(os.path.join(DIR, "room_producer_inputs.npz"))
    rec = json.load(open(os.path.join(DIR, "scene_records.json")))["room"]["camera"]
    sc = pkg.scene.Scene(materials={}, objects=[], camera=dict(eye=rec["position"], lookat=rec["lookAt"], fovy=rec["fov"][1]), skipped=[])
    p = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, atrous_nlevel=5, history_level=1)
    den = pkg.Denoiser(W, H, 0)
    o = orc.Oracle(pkg, W, H, threads=min(64, os.cpu_count() or 1))
    rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    gbt = torch.empty((H * W * 52,), dtype=torch.uint8, device="cuda")
    out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    worst = []
    for f in range(N):
        cam = pkg.scene.camera_for_frame(sc, f, True)
        pkg.binding.scene_render_mesh(rgb, gbt, W, H, cam, pi["geoms"], pi["geom_ids"], pi["tris"], pi["tri_ids"], pi["tri_albedo"], frame=f)
        den.denoise(out, rgb, gbt, cam, p)
        torch.cuda.synchronize()
        col = rgb.cpu().numpy()
        gb = gbt.cpu().numpy().view(pkg.synth.GBUFFER_DTYPE).reshape(H, W)
        if f == 0:
            assert (gb["geomId"] >= 0).mean() > 0.5, "the room should fill most of the frame"
        ref = o.denoise(col, gb, cam, p)
        got = out.cpu().numpy()
        err = relerr(got, ref)
        worst.append(float(err.max()))
        assert err.max() <= 1e-4, f"frame {f}: max rel {err.max():.3e}"
    print(f"BASELINE configs[2] (room.txt 1920x1080, 64 moving frames, full SVGF) vs oracle: worst max-rel {max(worst):.2e} (first 8: {max(worst[:8]):.2e}, last 8: {max(worst[-8:]):.2e})")
    den.free(); o.free()


@pytest.mark.gpu
def test_bunny_4k_static_device_producer_to_denoiser_vs_oracle(pkg, orc):
    """BASELINE configs[3] as worded — bunny.txt at 3840x2160, full SVGF: the scene's six primitives and the bunny's 4 968
    triangles (tests/golden/ref_scenes/bunny_producer_inputs.npz, made from the reference's scene + OBJ files by
    tests/golden/make_producer_inputs.py) are ray-cast at 4K by the device producer, two frames of the static camera go through
    svgf_denoise on the device, and the CPU oracle gets the same frames: <= 1e-4 relative per channel."""
    import torch
    W, H, N = 3840, 2160, 2
    pi = np.load(os.path.join(DIR, "bunny_producer_inputs.npz"))
    rec = json.load(open(os.path.join(DIR, "scene_records.json")))["bunny"]["camera"]
    sc = pkg.scene.Scene(materials={}, objects=[], camera=dict(eye=rec["position"], lookat=rec["lookAt"], fovy=rec["fov"][1]), skipped=[])
    p = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, atrous_nlevel=5, history_level=1)
    den = pkg.Denoiser(W, H, 0)
    o = orc.Oracle(pkg, W, H, threads=min(64, os.cpu_count() or 1))
    rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    gbt = torch.empty((H * W * 52,), dtype=torch.uint8, device="cuda")
    out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    worst = 0.0
    for f in range(N):
        cam = pkg.scene.camera_for_frame(sc, f, False)
        pkg.binding.scene_render_mesh(rgb, gbt, W, H, cam, pi["geoms"], pi["geom_ids"], pi["tris"], pi["tri_ids"], pi["tri_albedo"], frame=f)
        den.denoise(out, rgb, gbt, cam, p)
        torch.cuda.synchronize()
        col = rgb.cpu().numpy()
        gb = gbt.cpu().numpy().view(pkg.synth.GBUFFER_DTYPE).reshape(H, W)
        if f == 0:
            hit_bunny = np.isin(gb["geomId"], np.unique(pi["tri_ids"]))
            assert hit_bunny.mean() > 0.02, f"the bunny should be in the picture ({hit_bunny.mean():.4f} of the pixels)"
        ref = o.denoise(col, gb, cam, p)
        got = out.cpu().numpy()
        err = relerr(got, ref)
        assert err.max() <= 1e-4, f"frame {f}: max rel {err.max():.3e}"
        worst = max(worst, float(err.max()))
    den.free(); o.free()
    print(f"BASELINE configs[3] (bunny.txt 3840x2160, full SVGF, 2 frames) vs oracle: worst max-rel {worst:.2e}")


@pytest.mark.gpu
def test_room_4k_static_device_producer_to_denoiser_vs_oracle(pkg, orc):
    """The single-GPU leg of BASELINE configs[4] — room.txt (primitives + 2 810 triangles of its OBJ meshes) at 3840x2160, full SVGF,
    what every rank of `bench.py --gpus 8 --config 4k-room` runs: two frames of the static camera from the device producer through
    svgf_denoise, against the CPU oracle on the same frames: <= 1e-4 relative per channel (conftest.relerr)."""
    import torch
    W, H, N = 3840, 2160, 2
    pi = np.load(os.path.join(DIR, "room_producer_inputs.npz"))
    rec = json.load(open(os.path.join(DIR, "scene_records.json")))["room"]["camera"]
    sc = pkg.scene.Scene(materials={}, objects=[], camera=dict(eye=rec["position"], lookat=rec["lookAt"], fovy=rec["fov"][1]), skipped=[])
    p = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, atrous_nlevel=5, history_level=1)
    den = pkg.Denoiser(W, H, 0)
    o = orc.Oracle(pkg, W, H, threads=min(64, os.cpu_count() or 1))
    rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    gbt = torch.empty((H * W * 52,), dtype=torch.uint8, device="cuda")
    out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    worst = 0.0
    for f in range(N):
        cam = pkg.scene.camera_for_frame(sc, f, False)
        pkg.binding.scene_render_mesh(rgb, gbt, W, H, cam, pi["geoms"], pi["geom_ids"], pi["tris"], pi["tri_ids"], pi["tri_albedo"], frame=f)
        den.denoise(out, rgb, gbt, cam, p)
        torch.cuda.synchronize()
        col = rgb.cpu().numpy()
        gb = gbt.cpu().numpy().view(pkg.synth.GBUFFER_DTYPE).reshape(H, W)
        if f == 0:
            assert (gb["geomId"] >= 0).mean() > 0.5, "the room should fill most of the frame"
            assert np.isin(gb["geomId"], np.unique(pi["tri_ids"])).mean() > 0.01, "the room's meshes should be in the picture"
        ref = o.denoise(col, gb, cam, p)
        err = relerr(out.cpu().numpy(), ref)
        worst = max(worst, float(err.max()))
        assert err.max() <= 1e-4, f"frame {f}: max rel {err.max():.3e}"
    print(f"BASELINE configs[4], one GPU's share (room.txt 3840x2160, full SVGF, 2 frames) vs oracle: worst max-rel {worst:.2e}")
    den.free(); o.free()
