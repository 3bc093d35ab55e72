"""SURVEY.md §8f row f1, second half: the AoS -> plane repack fused into the producer (svgf_planar_gbuffer /
svgf_denoise_planar / svgf_synth_render_planar).  The planar path must give exactly what svgf_denoise gives on the same
texels: goldens of the reference replayed through both, and the device producer writing planes against the same producer
writing AoS texels."""
import ctypes

import numpy as np
import pytest

from conftest import denoiser_for, golden_cases, load_golden, replay

pytestmark = pytest.mark.gpu


def _hip():
    """The HIP runtime already loaded into this process (torch's), for plain host-to-device copies into raw pointers."""
    for ln in open("/proc/self/maps"):
        if "libamdhip64" in ln:
            return ctypes.CDLL(ln.split()[-1])
    raise RuntimeError("no HIP runtime loaded")


class AosEngine:
    def __init__(self, pkg, W, H):
        self.d = pkg.Denoiser(W, H, device=0)

    def reset(self):
        self.d.reset()

    def denoise(self, color, gbuffer, cam, p):
        return self.d.denoise_host(color, gbuffer, cam, p)

    def free(self):
        self.d.free()


class PlanarEngine(AosEngine):
    """Splits the texels on the host, copies the fields into the context's current-frame planes (what a plane-writing
    producer does on the device) and runs svgf_denoise_planar."""

    def denoise(self, color, gbuffer, cam, p):
        import torch
        hip = _hip()
        hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        g = self.d.planar_gbuffer()
        gb = np.ascontiguousarray(gbuffer).reshape(-1)
        for dst, arr in ((g.normal, gb["normal"].astype(np.float32)), (g.position, gb["position"].astype(np.float32)),
                         (g.geom_id, gb["geomId"].astype(np.int32)), (g.albedo, (gb["albedo"] * gb["ialbedo"]).astype(np.float32))):
            arr = np.ascontiguousarray(arr)
            assert hip.hipMemcpy(dst, arr.ctypes.data, arr.nbytes, 1) == 0
        H, W = self.d.height, self.d.width
        tin = torch.from_numpy(np.ascontiguousarray(color, dtype=np.float32)).cuda()
        out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
        self.d.denoise_planar(out, tin, cam, p)
        torch.cuda.synchronize()
        return out.cpu().numpy().reshape(np.asarray(color).shape)


@pytest.mark.parametrize("name", [c for c in golden_cases() if c.startswith(("full_", "temporal_", "atrous_synth", "atrous_nan", "atrous_rand64"))])
def test_planar_path_is_bit_identical_to_the_aos_path_on_the_goldens(pkg, name):
    z, runs = load_golden(name)
    W, H = int(z["W"]), int(z["H"])
    for tag in runs:
        if int(z[f"call_params_{tag}"][0][8]) > 7:
            continue
        a, b = AosEngine(pkg, W, H), PlanarEngine(pkg, W, H)
        ra, rb = replay(pkg, a, z, tag), replay(pkg, b, z, tag)
        a.free(); b.free()
        assert np.array_equal(ra, rb, equal_nan=True), f"{name}:{tag}: the planar path differs from the AoS path"


def test_device_producer_writing_planes_equals_the_one_writing_texels(pkg):
    """svgf_synth_render_planar -> svgf_denoise_planar against svgf_synth_render -> svgf_denoise, 6 moving frames at 640x360,
    sepcolor + addcolor on so that the albedo plane is exercised: bit-identical outputs and history."""
    import torch
    W, H, N = 640, 360, 6
    p = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, sepcolor=1, addcolor=1)
    rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    gbt = torch.empty((H * W * 52,), dtype=torch.uint8, device="cuda")
    res = {}
    for planar in (False, True):
        d = pkg.Denoiser(W, H, 0)
        outs = []
        out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
        for f in range(N):
            cam = pkg.synth.camera_for_frame(f, True)
            if planar:
                pkg.binding.synth_render_planar(rgb, d.planar_gbuffer(), W, H, cam, f, seed=77)
                d.denoise_planar(out, rgb, cam, p)
            else:
                pkg.binding.synth_render(rgb, gbt, W, H, cam, f, seed=77)
                d.denoise(out, rgb, gbt, cam, p)
            torch.cuda.synchronize()
            outs.append(out.cpu().numpy().copy())
        res[planar] = (outs, d.read_state(0), d.read_state(1), d.read_state(2))
        d.free()
    for f in range(N):
        assert np.array_equal(res[False][0][f], res[True][0][f]), f"frame {f}"
    for a, b in zip(res[False][1:], res[True][1:]):
        assert np.array_equal(a, b)


def test_scene_producer_writing_planes_equals_the_one_writing_texels(pkg):
    """svgf_scene_render_mesh_planar -> svgf_denoise_planar against svgf_scene_render_mesh -> svgf_denoise on room.txt's
    primitives + 2 810 mesh triangles (the reference's scene, tests/golden/ref_scenes/room_producer_inputs.npz) with the
    reference's camera automation, 4 moving frames at 480x270, sepcolor + addcolor on so that the albedo plane is read:
    bit-identical outputs and history."""
    import json
    import os
    import torch
    from conftest import ROOT
    DIR = os.path.join(ROOT, "tests", "golden", "ref_scenes")
    W, H, N = 480, 270, 4
    pi = np.load(os.path.join(DIR, "room_producer_inputs.npz"))
    rec = json.load(open(os.path.join(DIR, "scene_records.json")))["room"]["camera"]
    sc = pkg.scene.Scene(materials={}, objects=[], camera=dict(eye=rec["position"], lookat=rec["lookAt"], fovy=rec["fov"][1]), skipped=[])
    p = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, sepcolor=1, addcolor=1)
    rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    gbt = torch.empty((H * W * 52,), dtype=torch.uint8, device="cuda")
    res = {}
    for planar in (False, True):
        d = pkg.Denoiser(W, H, 0)
        outs = []
        out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
        for f in range(N):
            cam = pkg.scene.camera_for_frame(sc, f, True)
            target = d.planar_gbuffer() if planar else gbt
            pkg.binding.scene_render_mesh(rgb, target, W, H, cam, pi["geoms"], pi["geom_ids"], pi["tris"], pi["tri_ids"], pi["tri_albedo"], frame=f)
            if planar:
                d.denoise_planar(out, rgb, cam, p)
            else:
                d.denoise(out, rgb, gbt, cam, p)
            torch.cuda.synchronize()
            outs.append(out.cpu().numpy().copy())
        res[planar] = (outs, d.read_state(0), d.read_state(1), d.read_state(2))
        d.free()
    assert np.isfinite(res[False][0][-1]).all() and res[False][0][-1].max() > 0
    for f in range(N):
        assert np.array_equal(res[False][0][f], res[True][0][f]), f"frame {f}"
    for a, b in zip(res[False][1:], res[True][1:]):
        assert np.array_equal(a, b)


def test_planar_and_aos_frames_alternate_on_one_context(pkg):
    """The planes rotate with the history whichever entry point fed them: alternating frames equal an all-AoS run."""
    import torch
    W, H, N = 320, 200, 6
    p = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1)
    rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    gbt = torch.empty((H * W * 52,), dtype=torch.uint8, device="cuda")
    out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    res = {}
    for mixed in (False, True):
        d = pkg.Denoiser(W, H, 0)
        outs = []
        for f in range(N):
            cam = pkg.synth.camera_for_frame(f, True)
            if mixed and f % 2 == 1:
                pkg.binding.synth_render_planar(rgb, d.planar_gbuffer(), W, H, cam, f, seed=5)
                d.denoise_planar(out, rgb, cam, p)
            else:
                pkg.binding.synth_render(rgb, gbt, W, H, cam, f, seed=5)
                d.denoise(out, rgb, gbt, cam, p)
            torch.cuda.synchronize()
            outs.append(out.cpu().numpy().copy())
        res[mixed] = outs
        d.free()
    for f in range(N):
        assert np.array_equal(res[False][f], res[True][f]), f"frame {f}"


def test_planar_pointers_survive_a_reset_and_the_fused_first_level_reads_planes(pkg):
    """(a) svgf_planar_gbuffer() -> svgf_reset() -> producer -> svgf_denoise_planar(): the reset keeps the plane set the
    pointers name (it used to switch back to set 0, so the producer's planes were not the ones the next frame read), with the
    reset falling on an odd and on an even frame.  (b) the planar path through the fused temporal + first-level kernel
    (kernel_variant 6: its loader threads then read normal / position / geomId planes instead of texels) equals the AoS
    path through the same kernel bit for bit."""
    import torch
    W, H = 352, 198
    rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    gbt = torch.empty((H * W * 52,), dtype=torch.uint8, device="cuda")
    out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    for variant in (0, 6):
        p = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, sepcolor=1, addcolor=1, kernel_variant=variant)
        for reset_at in (1, 2):
            res = {}
            for planar in (False, True):
                d = denoiser_for(pkg, W, H, variant)      # (variant 6: the experiments build)
                outs = []
                for f in range(5):
                    cam = pkg.synth.camera_for_frame(f, True)
                    if planar:
                        planes = d.planar_gbuffer()
                        if f == reset_at:
                            d.reset()                     # between asking for the planes and filling them
                        pkg.binding.synth_render_planar(rgb, planes, W, H, cam, f, seed=5)
                        d.denoise_planar(out, rgb, cam, p)
                    else:
                        if f == reset_at:
                            d.reset()
                        pkg.binding.synth_render(rgb, gbt, W, H, cam, f, seed=5)
                        d.denoise(out, rgb, gbt, cam, p)
                    torch.cuda.synchronize()
                    outs.append(out.cpu().numpy().copy())
                res[planar] = (outs, d.read_state(0), d.read_state(1))
                d.free()
            for f in range(5):
                assert np.array_equal(res[False][0][f], res[True][0][f]), f"variant {variant}, reset at {reset_at}: frame {f}"
            assert np.array_equal(res[False][1], res[True][1]) and np.array_equal(res[False][2], res[True][2])


@pytest.mark.gpu
def test_planar_frames_on_the_frame_pipeline(pkg):
    """ABI 0.9: planar frames take the frame pipeline up.  (a) the promise (inputs_ready = 1) on a static scene whose planes of both
    parities were filled once; (b) inputs_ready = 2 with two streams in turn and a producer that REFILLS the planes every frame
    through svgf_planar_gbuffer_stream (its stream waits for the frames that still read them: the temporal pass of the last frame,
    the levels of the frame before last) — both bit-identical to ordered planar frames."""
    import torch
    W, H, N = 640, 360, 12
    base = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, atrous_nlevel=5, history_level=1)
    cams = [pkg.synth.camera_for_frame(f, True) for f in range(N)]
    res = {}
    for mode in ("ordered", "two-streams"):
        d = pkg.Denoiser(W, H, 0, pipelined=(mode != "ordered"))
        assert mode == "ordered" or d.pipeline_status() == 1, d.last_error()
        p = pkg.SvgfParams.from_buffer_copy(base).set(inputs_ready=0 if mode == "ordered" else 2)
        st = [torch.cuda.Stream(), torch.cuda.Stream()]
        rgb = [torch.empty((H, W, 3), dtype=torch.float32, device="cuda") for _ in range(2)]
        out = [torch.empty((H, W, 3), dtype=torch.float32, device="cuda") for _ in range(2)]
        keep = [torch.empty((H, W, 3), dtype=torch.float32, device="cuda") for _ in range(N)]
        torch.cuda.synchronize()
        for f in range(N):
            q = (f & 1) if mode != "ordered" else 0
            with torch.cuda.stream(st[q]):
                planes = d.planar_gbuffer(stream=st[q]) if mode != "ordered" else d.planar_gbuffer()
                pkg.binding.synth_render_planar(rgb[q], planes, W, H, cams[f], f, seed=19, stream=st[q])
                d.denoise_planar(out[q], rgb[q], cams[f], p, stream=st[q])
                keep[f].copy_(out[q], non_blocking=True)
        torch.cuda.synchronize()
        assert mode == "ordered" or d.is_pipelined()
        res[mode] = ([k.cpu().numpy() for k in keep], d.read_state(0), d.read_state(1), d.read_state(2))
        d.free()
    for f in range(N):
        assert np.array_equal(res["ordered"][0][f], res["two-streams"][0][f]), f"frame {f}"
    for a, b in zip(res["ordered"][1:], res["two-streams"][1:]):
        assert np.array_equal(a, b)
    # (a) static scene, planes of both parities filled once, every frame promised
    cam = pkg.synth.camera_for_frame(0, False)
    got = {}
    for promised in (False, True):
        d = pkg.Denoiser(W, H, 0, pipelined=promised)
        p = pkg.SvgfParams.from_buffer_copy(base).set(inputs_ready=1 if promised else 0)
        rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
        outs = [torch.empty((H, W, 3), dtype=torch.float32, device="cuda") for _ in range(10)]
        for _ in range(2):
            pkg.binding.synth_render_planar(rgb, d.planar_gbuffer(), W, H, cam, 0, seed=23)
            d.denoise_planar(outs[0], rgb, cam, p)
        torch.cuda.synchronize()
        for k in range(10):
            d.denoise_planar(outs[k], rgb, cam, p)
        d.sync()
        got[promised] = [o.cpu().numpy() for o in outs]
        assert d.is_pipelined() == promised
        d.free()
    for k in range(10):
        assert np.array_equal(got[False][k], got[True][k]), f"promised planar frame {k}"
