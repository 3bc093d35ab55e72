#!/usr/bin/env python3
"""Soak test: two contexts fed the same device-produced moving-camera sequence on two streams, thousands of frames without
a synchronisation between calls, parameters redrawn every K frames (--random K).
Context A is always a PIPELINED context (svgf_create_ex(SVGF_CREATE_PIPELINED)) whose frames carry the inputs_ready = 1 promise.
  --pair same (the default): context B runs the same frames ordered on its stream (inputs_ready = 0): bit-identical at every check.
  --pair default-gather: B runs the strict gather kernel on every level (kernel_variant 1): <= 1e-5 relative (1e-4 while levels
        6+ are on: steps >= 64), history lengths bit for bit.
  --pair default-fused: B forces the parked fused temporal + first-level kernel of the experiments build (kernel_variant 6; it
        falls back to the unfused path by itself for parameter draws it does not support): <= 1e-5.
usage: soak.py [--size 1920x1080] [--frames 2000] [--random 25] [--pair same]"""
import argparse
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # the pipelined context's two internal streams on hardware queues of their own (include/svgf.h)
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
    ap.add_argument("--pair", default="same", choices=["default-fused", "default-gather", "same"])
    a = ap.parse_args()
    import torch
    pkg = ge.load_package()
    W, H = map(int, a.size.split("x"))
    vb = {"default-fused": 6, "default-gather": 1, "same": 0}[a.pair]
    da, db = pkg.Denoiser(W, H, pipelined=True), pkg.Denoiser(W, H, experiments=(vb == 6))      # (the parked fused kernel lives in libsvgf_hip_exp.so)
    print(f"context A: pipeline status {da.pipeline_status()} {da.last_error()}")
    pa = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, atrous_nlevel=5, history_level=1, inputs_ready=1)
    pb = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, atrous_nlevel=5, history_level=1, inputs_ready=0, kernel_variant=vb)
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
    bad, worst = 0, 0.0
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
            if a.pair == "same":
                same = bool(torch.equal(oa[f & 1], ob[f & 1]))
            else:
                den = torch.clamp(torch.abs(oa[f & 1]), min=1e-3)
                err = float((torch.abs(oa[f & 1] - ob[f & 1]) / den).max())
                worst = max(worst, err)
                # steps >= 64 (levels 6+) run the lattice kernel, which sums in another order than the lane / gather kernels: the suite's 1e-4
                tol = 1e-4 if pa.atrous_nlevel >= 6 else 1e-5
                same = err <= tol and bool((da.read_state(0) == db.read_state(0)).all())
            fin = bool(torch.isfinite(oa[f & 1]).all())
            if not (same and fin):
                bad += 1
                print(f"frame {f}: agree={same} (max rel {err if a.pair != 'same' else 0.0:.2e}) finite={fin} params nlevel={pa.atrous_nlevel} hist={pa.history_level} paper={pa.paper_steps} view={pa.right_view_option} t={pa.temporal_enable} s={pa.spatial_enable}")
    torch.cuda.synchronize()
    print(f"soak {W}x{H}, {a.frames} frames, {a.pair}: " + (f"OK at every check (worst relative difference {worst:.2e}, history lengths identical)" if bad == 0 and a.pair != "same"
          else "OK, bit-identical at every check" if bad == 0 else str(bad) + " MISMATCHES"))
    da.free(); db.free()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
