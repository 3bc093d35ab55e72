#!/usr/bin/env python3
"""Why one kernel has four durations: the GPU's clock state at the moment of measurement.

The same 1080p a-trous level has been quoted at 41, 43, 46, 50 and 57 us in this repository.  The kernels were the same; what
differed was what the GPU had been doing in the milliseconds BEFORE the measurement and what sat between the kernels while it
ran.  This tool measures exactly those two things on one box, in one process, with the shader clock / socket power sampled
every millisecond beside each measurement:

  sustained      20 frames enqueued back to back right behind 0.6 s of back-to-back frames (bench.py's timed region)
  idle g ms      the same 20 frames after the GPU sat idle for g milliseconds (g = 0.2 ... 1000): DVFS leaves the sustained
                 state within a millisecond or two and takes several frames to come back; `first 4` / `last 4` show the ramp
  events         HIP event pairs around every kernel of every frame (what svgf_profile_enable does: each pair widens the gap
                 between two kernels and is where "in-bench events: +2-3 us per launch" comes from)
  sync per call  svgf_denoise + svgf_sync per frame (the GPU idles while the host returns and re-enqueues)
  cold           the first frames of a process, before any warm-up (what tools/probe.py --frames 12 measures)

    python tools/clock_states.py [--size 1920x1080] [--json out.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as ge  # noqa: E402
import telemetry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="1920x1080")
    ap.add_argument("--json", default=None)
    ap.add_argument("--frames", type=int, default=20)
    a = ap.parse_args()
    import torch
    W, H = (int(v) for v in a.size.split("x"))
    pkg = ge.load_package()
    dev = torch.device("cuda", 0)
    params = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, atrous_nlevel=5, history_level=1)
    nsrc = 4
    cam_dicts = [pkg.synth.camera_for_frame(f, False) for f in range(nsrc)]
    d_in = [torch.empty((H, W, 3), dtype=torch.float32, device=dev) for _ in range(nsrc)]
    d_g = [torch.empty((H * W * 52,), dtype=torch.uint8, device=dev) for _ in range(nsrc)]
    for f in range(nsrc):
        pkg.binding.synth_render(d_in[f], d_g[f], W, H, cam_dicts[f], f, seed=1000, device=0)
    torch.cuda.synchronize(dev)
    cams = [pkg.SvgfCamera.from_dict(c) for c in cam_dicts]
    out = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
    den = pkg.Denoiser(W, H, device=0)
    stream = torch.cuda.current_stream(dev)
    tm = telemetry.Sampler(0, period_s=0.001)
    tm.start()
    K = a.frames

    def frame(i):
        den.denoise(out, d_in[i % nsrc], d_g[i % nsrc], cams[i % nsrc], params, stream=stream)

    def sustain(seconds=0.6):
        t = time.perf_counter()
        n = 0
        while time.perf_counter() - t < seconds:
            frame(n); n += 1
            if n % 32 == 0:
                torch.cuda.synchronize(dev)
        torch.cuda.synchronize(dev)

    def batch(n, events=False, sync_each=False):
        """n frames; returns (ms per frame overall, ms per frame of the first 4, of the last 4, level mean us or None, telemetry)."""
        den.profile_stride(1)
        den.profile_enable(n if events else 0)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        t0 = time.perf_counter()
        evs[0].record(stream)
        for i in range(n):
            frame(i)
            if sync_each:
                den.sync()
            evs[i + 1].record(stream)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        per = [evs[i].elapsed_time(evs[i + 1]) for i in range(n)]
        lvl = None
        if events:
            ms = [m for s in range(min(n, den.profile_frames())) for k, m in den.profile_read(s) if k == pkg.binding.KERNEL_ATROUS]
            lvl = float(np.mean(ms)) * 1e3 if ms else None
            den.profile_enable(0)
        wall = (t1 - t0) / n * 1e3
        return {"ms_per_frame": round(wall if sync_each else float(np.sum(per)) / n, 4), "first4_ms": round(float(np.mean(per[:4])), 4),
                "last4_ms": round(float(np.mean(per[-4:])), 4), "level_mean_us_events": None if lvl is None else round(lvl, 2),
                "telemetry": tm.summary(t0, t1)}

    rows = []

    def add(name, idle_ms, r):
        r = dict(r); r["scenario"] = name; r["idle_before_ms"] = idle_ms
        rows.append(r)
        t = r["telemetry"]
        clk = t.get("sclk_mhz") or {}
        pw = t.get("power_w") or {}
        print(f"{name:34s} idle {idle_ms:7.1f} ms | {r['ms_per_frame']:.4f} ms/frame (first 4: {r['first4_ms']:.4f}, last 4: {r['last4_ms']:.4f})"
              f" | level (events) {r['level_mean_us_events']} us | sclk {clk.get('min')}-{clk.get('median')}-{clk.get('max')} MHz, {pw.get('median')} W, {t.get('n')} samples", flush=True)

    # cold: the process has done nothing but allocate and render its inputs
    time.sleep(1.0)
    add("cold (first frames of a process)", 1000.0, batch(12))
    time.sleep(1.0)
    add("cold, events on every kernel", 1000.0, batch(12, events=True))
    for rep in range(2):
        sustain()
        add("sustained", 0.0, batch(K))
        sustain()
        add("sustained, events on every kernel", 0.0, batch(K, events=True))
        sustain()
        add("sustained, sync per call", 0.0, batch(K, sync_each=True))
        for g in (0.2, 0.5, 1.0, 2.0, 5.0, 20.0, 100.0, 1000.0):
            sustain()
            time.sleep(g * 1e-3)
            add("after idle", g, batch(K))
    tm.stop()
    if a.json:
        with open(a.json, "w") as fh:
            json.dump({"size": a.size, "frames": K, "rows": rows}, fh, indent=1)


if __name__ == "__main__":
    main()
