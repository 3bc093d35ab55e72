#!/usr/bin/env python3
"""Time series of ms/frame (HIP events on the caller's stream every 20 frames) across mode switches: ordered -> pipelined -> ordered."""
import os, sys, time, json
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import __graft_entry__ as ge
import telemetry
pkg = ge.load_package()
W, H = 1920, 1080
dev = torch.device("cuda", 0)
cam = [pkg.synth.camera_for_frame(f, False) for f in range(4)]
d_in = [torch.empty((H, W, 3), dtype=torch.float32, device=dev) for _ in range(4)]
d_g = [torch.empty((H * W * 52,), dtype=torch.uint8, device=dev) for _ in range(4)]
for f in range(4):
    pkg.binding.synth_render(d_in[f], d_g[f], W, H, cam[f], f, seed=1000)
cams = [pkg.SvgfCamera.from_dict(c) for c in cam]
outs = [torch.empty((H, W, 3), dtype=torch.float32, device=dev) for _ in range(2)]
base = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, atrous_nlevel=5, history_level=1)
dp = pkg.Denoiser(W, H, 0, pipelined=True); pp = pkg.SvgfParams.from_buffer_copy(base).set(inputs_ready=1)
do = pkg.Denoiser(W, H, 0); po = base
s = torch.cuda.current_stream(dev)
G = 20
plan = [("ordered", 1600), ("pipelined", 3200), ("ordered", 1600), ("pipelined", 1600)]
tm = telemetry.Sampler(0, period_s=0.005).start()
evs, marks = [], []
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e0.record(s); evs.append(e0)
i = 0
for mode, n in plan:
    d, p = (dp, pp) if mode == "pipelined" else (do, po)
    marks.append((mode, len(evs)))
    for k in range(n):
        d.denoise(outs[i & 1], d_in[i % 4], d_g[i % 4], cams[i % 4], p, stream=s)
        i += 1
        if (k + 1) % G == 0:
            e = torch.cuda.Event(enable_timing=True); e.record(s); evs.append(e)
torch.cuda.synchronize()
tm.stop()
ms = [evs[j].elapsed_time(evs[j + 1]) / G for j in range(len(evs) - 1)]
t = np.cumsum([0] + [m * G for m in ms])
for (mode, start), nxt in zip(marks, [m[1] for m in marks[1:]] + [len(evs)]):
    seg = ms[start - 1:nxt - 1]
    print(f"{mode}: {len(seg)} groups of {G} frames starting at t={t[start-1]:.0f} ms")
    # print at log-spaced positions
    for a, b in ((0, 1), (1, 2), (2, 4), (4, 8), (8, 16), (16, 32), (32, 64), (64, 128), (128, 256)):
        if a < len(seg):
            print(f"   groups {a:3d}-{min(b, len(seg)):3d} ({a*G*0.26:6.0f} ms..): {np.mean(seg[a:b]):.4f} ms/frame")
pw = [(r[0], r[1], r[2]) for r in tm.samples]
print("power samples (every 40th):", [(round(x[1] or 0), round(x[2] or 0)) for x in pw[::40]])
