#!/usr/bin/env python3
"""Round 5: what N INDEPENDENT sequences on N streams of ONE GPU give in aggregate, against one sequence on one stream.

Two contexts share nothing, so whatever the hardware can overlap between the kernels of two frames — one frame's level tails and
launch gaps under the other's tap rows, the HBM-bound temporal pass beside an issue-bound level — it is free to overlap here: the
aggregate figure is the CEILING of what a cross-frame overlap inside one sequence (round 1-3's `inputs_ready`, removed in round 4)
could reach, and it is what a render farm gets from running several sequences per GPU (BASELINE configs[4] is such a farm).
usage: exp_r05_two_sequences.py [--size 1920x1080] [--frames 400] [--contexts 1,2,3,4]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="1920x1080")
    ap.add_argument("--frames", type=int, default=400)
    ap.add_argument("--contexts", default="1,2,3,4")
    ap.add_argument("--batch", type=int, default=1, help="frames enqueued per context before switching to the next one")
    a = ap.parse_args()
    W, H = (int(v) for v in a.size.split("x"))
    import torch
    pkg = ge.load_package()
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import telemetry
    p = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, atrous_nlevel=5, history_level=1)
    nmax = max(int(v) for v in a.contexts.split(","))
    dens, ins, gbs, cams, outs, streams = [], [], [], [], [], []
    for c in range(nmax):
        dens.append(pkg.Denoiser(W, H, 0))
        cam = [pkg.synth.camera_for_frame(f, False) for f in range(4)]
        di = [torch.empty((H, W, 3), dtype=torch.float32, device="cuda") for _ in range(4)]
        dg = [torch.empty((H * W * 52,), dtype=torch.uint8, device="cuda") for _ in range(4)]
        for f in range(4):
            pkg.binding.synth_render(di[f], dg[f], W, H, cam[f], frame=f, seed=1000 + c)
        ins.append(di); gbs.append(dg); cams.append(cam)
        outs.append(torch.empty((H, W, 3), dtype=torch.float32, device="cuda"))
        streams.append(torch.cuda.Stream())
    torch.cuda.synchronize()

    def run(n, frames):
        for f in range(0, frames, a.batch):
            for c in range(n):
                for k in range(a.batch):
                    dens[c].denoise(outs[c], ins[c][(f + k) % 4], gbs[c][(f + k) % 4], cams[c][(f + k) % 4], p, stream=streams[c])

    for n in (int(v) for v in a.contexts.split(",")):
        for rep in range(2):
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.6:      # sustained state
                run(n, 40)
                torch.cuda.synchronize()
            tm = telemetry.Sampler(period_s=0.002)
            tm.start()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(n, a.frames)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            tm.stop()
            s = tm.summary(t0, t1)
            us = (t1 - t0) / (a.frames * n) * 1e6
            print(f"{n} context(s) on {n} stream(s), {a.frames} frames each, {a.batch} per turn: {us:.1f} us per frame in aggregate = "
                  f"{W * H / us:.0f} Mpix/s   (power {s.get('power_w', {}).get('median')} W, sclk {s.get('sclk_mhz', {}).get('median')} MHz)", flush=True)
    for d in dens:
        d.free()


if __name__ == "__main__":
    main()
