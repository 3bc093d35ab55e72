#!/usr/bin/env python3
"""Visual QA: produce -> denoise -> display-pack N frames on the device, save frame N-1 side by side
(1-spp | denoised, as the reference's viewer shows them) as a PNG with svgf_save_png.
usage: snapshot.py [--size 640x360] [--frames 16] [--moving] [--exact-reprojection] [--out gpurun_out/side_by_side.png]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="640x360")
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--moving", action="store_true")
    ap.add_argument("--scene", default=None, help="scene text file (reference format): primitives rendered by svgf_scene_render")
    ap.add_argument("--exact-reprojection", action="store_true", help="SvgfParams::reproj_scale = (tan(fovy) W/H, tan(fovy))")
    ap.add_argument("--paper-steps", action="store_true", help="SvgfParams::paper_steps: dilations 1,2,4,8,16 instead of 2..32")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "side_by_side.png"))
    a = ap.parse_args()
    import torch
    pkg = ge.load_package()
    W, H = map(int, a.size.split("x"))
    den = pkg.Denoiser(W, H)
    params = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, atrous_nlevel=5, history_level=1)
    if a.paper_steps:
        params.paper_steps = 1
    if a.exact_reprojection:
        plx, ply = pkg.synth._pixel_length(W, H, 45.0)
        params.reproj_scale[0], params.reproj_scale[1] = float(plx) * W / 2.0, float(ply) * H / 2.0
    rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    gb = torch.empty((H * W * 52,), dtype=torch.uint8, device="cuda")
    out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    s = torch.cuda.current_stream()
    scene = geoms = None
    if a.scene:
        scene = pkg.scene.parse_scene(open(a.scene).read())
        geoms = pkg.scene.geom_array(scene)
    for f in range(a.frames):
        if scene is not None:
            cam = pkg.scene.camera_for_frame(scene, f, a.moving)
            pkg.binding.scene_render(rgb, gb, W, H, cam, geoms, f, seed=1, stream=s)
        else:
            cam = pkg.synth.camera_for_frame(f, a.moving)
            pkg.binding.synth_render(rgb, gb, W, H, cam, f, seed=1, stream=s)
        den.denoise(out, rgb, gb, cam, params, stream=s)
    torch.cuda.synchronize()
    side = np.concatenate([rgb.cpu().numpy(), out.cpu().numpy()], axis=1)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    pkg.binding.save_png(a.out, side, mirror_x=False)
    n, c = rgb.cpu().numpy(), out.cpu().numpy()
    print(f"saved {a.out}: {2 * W}x{H}; mean |d/dx| 1-spp {np.abs(np.diff(n, axis=1)).mean():.4f} -> denoised "
          f"{np.abs(np.diff(c, axis=1)).mean():.4f}")
    den.free()


if __name__ == "__main__":
    main()
