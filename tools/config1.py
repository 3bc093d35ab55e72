#!/usr/bin/env python3
"""BASELINE config 1: 800x800, temporal off, ONE a-trous level — single-threaded CPU (oracle port) beside the HIP path."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch
pkg = ge.load_package(); orc = ge.load_oracle()
W = H = 800
c, g, cam = pkg.synth.render_frame(W, H, 0, seed=3)
p = pkg.reference_defaults().set(temporal_enable=0, spatial_enable=1, atrous_nlevel=1)
for threads in (1, 64):
    o = orc.Oracle(pkg, W, H, threads=threads)
    o.denoise(c, g, cam, p)
    t0 = time.perf_counter(); n = 3
    for _ in range(n): ref = o.denoise(c, g, cam, p)
    dt = (time.perf_counter() - t0) / n
    print(f"CPU oracle, {threads:2d} thread(s): {dt*1e3:8.2f} ms/call = {W*H/dt/1e6:8.2f} Mpix/s")
    o.free()
d = pkg.Denoiser(W, H, 0)
tin = torch.from_numpy(c).cuda(); tg = torch.from_numpy(g.view(np.uint8).reshape(-1)).cuda()
out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
for _ in range(10): d.denoise(out, tin, tg, cam, p)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): d.denoise(out, tin, tg, cam, p)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 200
got = out.cpu().numpy()
err = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-2)
print(f"HIP (MI355X): {ms*1e3:8.1f} us/call = {W*H/ms/1e3:8.1f} Mpix/s ; max rel err vs oracle {err.max():.2e}")
