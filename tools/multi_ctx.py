#!/usr/bin/env python3
"""Aggregate ms per 1080p frame of K independent sequences on one GPU (each its own context and caller stream), ordered or pipelined:
the ceiling of deeper frame pipelines.  usage: multi_ctx.py [name=value ...]   (svgf_exp_set knobs -> experiments library)"""
import os, sys, time, json
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import __graft_entry__ as ge
import telemetry
pkg = ge.load_package()
knobs = [a for a in sys.argv[1:] if "=" in a]
if knobs:
    pkg.binding.use_experiments_library(True)
    for kv in knobs:
        k, v = kv.split("="); pkg.binding.exp_set(k, int(v))
W, H = 1920, 1080
dev = torch.device("cuda", 0)
cam = [pkg.synth.camera_for_frame(f, False) for f in range(4)]
d_in = [torch.empty((H, W, 3), dtype=torch.float32, device=dev) for _ in range(4)]
d_g = [torch.empty((H * W * 52,), dtype=torch.uint8, device=dev) for _ in range(4)]
for f in range(4):
    pkg.binding.synth_render(d_in[f], d_g[f], W, H, cam[f], f, seed=1000)
cams = [pkg.SvgfCamera.from_dict(c) for c in cam]
base = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, atrous_nlevel=5, history_level=1)
tm = telemetry.Sampler(0, period_s=0.004).start()
res = {"knobs": knobs}
for K, piped in ((1, False), (1, True), (2, False), (2, True), (3, False), (3, True), (4, False)):
    ctxs = [pkg.Denoiser(W, H, 0, pipelined=piped) for _ in range(K)]
    p = pkg.SvgfParams.from_buffer_copy(base).set(inputs_ready=1 if piped else 0)
    st = [torch.cuda.Stream() for _ in range(K)]
    outs = [[torch.empty((H, W, 3), dtype=torch.float32, device=dev) for _ in range(2)] for _ in range(K)]
    ctr = [0]
    def frames(n):
        for _ in range(n):
            i = ctr[0]; ctr[0] += 1
            c = i % K; j = i // K
            ctxs[c].denoise(outs[c][j & 1], d_in[j % 4], d_g[j % 4], cams[j % 4], p, stream=st[c])
    def region(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); frames(n); torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    t = time.perf_counter()
    while time.perf_counter() - t < 0.5:
        frames(64); torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = [region(120) for _ in range(5)]
    t1 = time.perf_counter()
    tms = tm.summary(t0, t1)
    key = f"{K}x{'P' if piped else 'O'}"
    res[key] = round(float(np.median(r)), 5)
    res[key + "_w"] = tms["power_w"]["median"] if tms["power_w"] else None
    for c in ctxs: c.free()
tm.stop()
print("MC " + json.dumps(res), flush=True)
