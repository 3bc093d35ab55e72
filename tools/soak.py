#!/usr/bin/env python3
"""Soak test of the cross-frame overlap: two contexts fed the same device-produced moving-camera sequence, one with
SvgfParams::inputs_ready = 1 (temporal pass of frame f+1 beside levels 2-5 of frame f), one fully ordered on the
caller's stream.  Every output must be bit-identical.  usage: soak.py [--size 1920x1080] [--frames 2000]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="1920x1080")
    ap.add_argument("--frames", type=int, default=2000)
    ap.add_argument("--random", type=int, default=0, metavar="K",
                    help="every K frames draw new parameters (levels 0..8, history level, paper steps, pre-blur, debug views, "
                         "temporal on/off) for both contexts")
    a = ap.parse_args()
    import torch
    pkg = ge.load_package()
    W, H = map(int, a.size.split("x"))
    da, db = pkg.Denoiser(W, H), pkg.Denoiser(W, H)
    pa = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, atrous_nlevel=5, history_level=1, inputs_ready=1)
    pb = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, atrous_nlevel=5, history_level=1, inputs_ready=0)
    nbuf = 64    # a 64-frame moving-camera sequence produced up front and replayed, so calls go back to back
    rgb = [torch.empty((H, W, 3), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
    gb = [torch.empty((H * W * 52,), dtype=torch.uint8, device="cuda") for _ in range(nbuf)]
    cams = [pkg.synth.camera_for_frame(k, True) for k in range(nbuf)]
    s = torch.cuda.current_stream()
    for k in range(nbuf):
        pkg.binding.synth_render(rgb[k], gb[k], W, H, cams[k], k, seed=77, stream=s)
    torch.cuda.synchronize()              # inputs_ready promises the inputs are complete at call time
    oa = [torch.empty((H, W, 3), dtype=torch.float32, device="cuda") for _ in range(2)]
    ob = [torch.empty((H, W, 3), dtype=torch.float32, device="cuda") for _ in range(2)]
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    bad = 0
    import random
    rng = random.Random(20260928)
    for f in range(a.frames):
        k = f % nbuf
        if a.random and f % a.random == 0 and f > 0:
            nl = rng.randint(0, 8)
            kw = dict(atrous_nlevel=nl, history_level=rng.randint(0, nl + 1), paper_steps=rng.randint(0, 1),
                      blur_variance=rng.randint(0, 1), right_view_option=rng.choice([0, 0, 0, 0, 1, 2]),
                      temporal_enable=rng.choice([1, 1, 1, 0]), spatial_enable=rng.choice([1, 1, 1, 0]),
                      sepcolor=rng.randint(0, 1), addcolor=rng.randint(0, 1))
            pa.set(**kw); pb.set(**kw)
        da.denoise(oa[f & 1], rgb[k], gb[k], cams[k], pa, stream=sa)      # no synchronisation between calls
        db.denoise(ob[f & 1], rgb[k], gb[k], cams[k], pb, stream=sb)
        if f % 97 == 96 or f == a.frames - 1:
            torch.cuda.synchronize()
            same = bool(torch.equal(oa[f & 1], ob[f & 1]))
            fin = bool(torch.isfinite(oa[f & 1]).all())
            if not (same and fin):
                bad += 1
                print(f"frame {f}: identical={same} finite={fin}")
    torch.cuda.synchronize()
    print(f"soak {W}x{H}, {a.frames} frames, overlap vs ordered: {'OK, bit-identical at every check' if bad == 0 else str(bad) + ' MISMATCHES'}")
    da.free(); db.free()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
