#!/usr/bin/env python3
"""Developer probe (GPU box): per-kernel timings of the HIP path at a given size, strip vs gather kernel,
plus the strip kernel's deviation from the strict gather kernel on the same frames.

  python tools/probe.py --size 1920x1080 --frames 12 [--variants 1,2] [--env SVGF_STRIP_TX=256,SVGF_STRIP_ROWS=1]
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import __graft_entry__ as ge  # noqa: E402
import telemetry  # noqa: E402

KIND = {1: "temporal", 2: "prepare", 3: "atrous", 4: "debug", 5: "copyout", 6: "fused"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="1920x1080")
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--variants", default="1,2")
    ap.add_argument("--nlevel", type=int, default=5)
    ap.add_argument("--check", action="store_true", help="compare outputs of the variants frame by frame")
    ap.add_argument("--overlap", type=int, default=0, help="SvgfParams.inputs_ready")
    ap.add_argument("--blur", type=int, default=1, help="SvgfParams.blur_variance")
    ap.add_argument("--reps", type=int, default=20, help="frames of the back-to-back wall-time loop (no per-kernel events)")
    ap.add_argument("--planar", action="store_true", help="G-buffer handed over as planes (svgf_denoise_planar), static scene")
    ap.add_argument("--sustain", type=float, default=0.0, help="seconds of back-to-back frames in front of every measurement (the sustained clock / "
                    "power state bench.py measures in, DESIGN.md 6.2); 0 = measure from wherever the GPU is (cold after start-up)")
    ap.add_argument("--telemetry-json", default=None, help="append one JSON line per variant: {variant, frame_us, telemetry summary} (the A/B scripts' clock check)")
    a = ap.parse_args()
    import torch
    pkg = ge.load_package()
    W, H = map(int, a.size.split("x"))
    n = W * H
    nsrc = 2
    frames = [pkg.synth.render_frame(W, H, f, seed=5, moving=False) for f in range(nsrc)]
    d_in = [torch.from_numpy(f[0]).cuda() for f in frames]
    d_g = [torch.from_numpy(f[1].view(np.uint8).reshape(-1)).cuda() for f in frames]
    cam = [pkg.SvgfCamera.from_dict(f[2]) for f in frames]
    out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    results = {}
    for v in [int(x) for x in a.variants.split(",")]:
        d = pkg.Denoiser(W, H, 0, experiments=v in (5, 6), pipelined=a.overlap != 0)      # parked variants live in libsvgf_hip_exp.so
        p = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, atrous_nlevel=a.nlevel, kernel_variant=v, inputs_ready=a.overlap, blur_variance=a.blur)
        if a.planar:      # both plane sets (they alternate with the history) get the static scene's G-buffer; d.denoise then means denoise_planar
            cam_dict = pkg.synth.camera_for_frame(0, False)
            for _ in range(2):
                pkg.binding.synth_render_planar(d_in[0], d.planar_gbuffer(), W, H, cam_dict, 0, seed=5, device=0)
                d.denoise_planar(out, d_in[0], cam[0], p)
            torch.cuda.synchronize()
            d.denoise = lambda o, i, g, c, pp, _d=d: _d.denoise_planar(o, i, c, pp)
        def sustain():
            if a.sustain <= 0:
                return
            import time
            t, k = time.perf_counter(), 0
            while time.perf_counter() - t < a.sustain:
                d.denoise(out, d_in[k % nsrc], d_g[k % nsrc], cam[k % nsrc], p)
                k += 1
                if k % 32 == 0:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
        d.profile_enable(a.frames)
        sustain()
        d.profile_enable(a.frames)        # (same slot count: only the counters restart, no idle time)
        outs = []
        for f in range(a.frames):
            d.denoise(out, d_in[f % nsrc], d_g[f % nsrc], cam[f % nsrc], p)
            if a.check:
                torch.cuda.synchronize()
                outs.append(out.cpu().numpy().copy())
        d.sync()
        # whole-frame wall time without per-kernel events
        d.profile_enable(0)
        torch.cuda.synchronize()
        sustain()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        reps = a.reps
        with telemetry.Sampler(0) as tm:
            e0.record()
            for f in range(reps):
                d.denoise(out, d_in[f % nsrc], d_g[f % nsrc], cam[f % nsrc], p)
            e1.record(); torch.cuda.synchronize()
        frame_ms = e0.elapsed_time(e1) / reps
        tms = tm.summary()
        d.profile_enable(a.frames)
        sustain()
        d.profile_enable(a.frames)
        for f in range(a.frames):
            d.denoise(out, d_in[f % nsrc], d_g[f % nsrc], cam[f % nsrc], p)
        d.sync()
        rows = [d.profile_read(s) for s in range(a.frames)]
        steady = rows[2:]
        nk = len(steady[0])
        med = [float(np.median([r[k][1] for r in steady])) for k in range(nk)]
        kinds = [steady[0][k][0] for k in range(nk)]
        print(f"variant {v}: frame wall {frame_ms*1e3:.1f} us = {n/frame_ms/1e3:.1f} Mpix/s ({reps} frames back to back); sum of kernels {sum(med)*1e3:.1f} us")
        if a.telemetry_json:
            import json
            with open(a.telemetry_json, "a") as fh:
                fh.write(json.dumps({"variant": v, "frame_us": frame_ms * 1e3, "telemetry": tms}) + "\n")
        print(f"   telemetry during the wall-time loop: sclk {tms['sclk_mhz']} MHz, power {tms['power_w']} W, temp {tms['temp_c']} C ({tms['n']} samples, {tms['source']})")
        lvl = 0
        for k in range(nk):
            name = KIND[kinds[k]]
            extra = ""
            if kinds[k] == 6:
                lvl += 1
                extra = f" step {1 << lvl:3d}  (temporal pass + level {lvl})"
            if kinds[k] == 3:
                lvl += 1
                gbs = 56.0 * n / (med[k] * 1e-3) / 1e9
                extra = f" step {1 << lvl:3d}  {gbs:7.0f} GB/s algorithmic ({gbs/8000*100:.1f}% of 8 TB/s)"
            print(f"   {name:9s} {med[k]*1e3:8.1f} us{extra}")
        results[v] = outs
        d.free()
    if a.check and len(results) == 2:
        (va, oa), (vb, ob) = results.items()
        for f in range(len(oa)):
            e = np.abs(oa[f] - ob[f]) / np.maximum(np.abs(oa[f]), 1e-2)
            print(f"frame {f}: variant {vb} vs {va}: max rel {e.max():.2e} p99.9 {np.quantile(e, 0.999):.2e} frac>1e-4 {(e > 1e-4).mean():.2e}")


if __name__ == "__main__":
    main()
