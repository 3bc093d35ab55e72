#!/usr/bin/env python3
"""Regions of K frames between synchronisations: what precedes a region, and how long it is."""
import os, sys, time, json
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import __graft_entry__ as ge
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
prof = "--prof" in sys.argv
if prof:
    for d in (dp, do):
        d.profile_stride(10); d.profile_enable(20)
ctr = [0]
def frames(mode, n):
    d, p = (dp, pp) if mode == "P" else (do, po)
    for k in range(n):
        i = ctr[0]; ctr[0] += 1
        d.denoise(outs[i & 1], d_in[i % 4], d_g[i % 4], cams[i % 4], p, stream=s)
def region(mode, n):
    if prof: (dp if mode == "P" else do).profile_enable(20)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    frames(mode, n)
    th = time.perf_counter() - t0
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return dt / n * 1e3, th / n * 1e3
def sustain(mode, sec):
    t = time.perf_counter(); k = 0
    while time.perf_counter() - t < sec:
        frames(mode, 32); k += 32
        if k % 128 == 0: torch.cuda.synchronize()
    torch.cuda.synchronize()
def show(tag, rs):
    print(f"{tag:60s} " + " ".join(f"{r[0]:.4f}({r[1]:.3f})" for r in rs), flush=True)
sustain("P", 0.6)
show("after 0.6 s pipelined: 6 consecutive P regions of 20", [region("P", 20) for _ in range(6)])
sustain("O", 0.4)
show("after 0.4 s ordered: 6 consecutive O regions of 20", [region("O", 20) for _ in range(6)])
rs = []
for _ in range(4):
    rs.append(region("O", 20)); rs.append(region("P", 20))
show("alternating O P O P ... regions of 20", rs)
show("then 6 consecutive P regions of 20", [region("P", 20) for _ in range(6)])
sustain("P", 0.4)
show("after 0.4 s pipelined: P regions of 20, 40, 100, 400, 20, 20", [region("P", n) for n in (20, 40, 100, 400, 20, 20)])
sustain("O", 0.4)
show("after 0.4 s ordered: O regions of 20, 40, 100, 400, 20, 20", [region("O", n) for n in (20, 40, 100, 400, 20, 20)])
show("P regions of 20 right after ordered sustain", [region("P", 20) for _ in range(8)])
