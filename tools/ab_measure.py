#!/usr/bin/env python3
"""One process, one library (whatever libsvgf_hip.so is in place): sustained ms per 1080p frame ordered and pipelined, socket power
of both, and the kernels' own durations.  Prints one JSON line.  Used by tools/ab.sh to A/B prebuilt libraries on one box."""
import os, sys, time, json
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import __graft_entry__ as ge
import telemetry
pkg = ge.load_package()
size = sys.argv[1] if len(sys.argv) > 1 else "1920x1080"
knobs = [a for a in sys.argv[2:] if "=" in a]      # name=value: svgf_exp_set knobs -> the experiments build of the library is used
if knobs:
    pkg.binding.use_experiments_library(True)
    for kv in knobs:
        k, v = kv.split("="); pkg.binding.exp_set(k, int(v))
W, H = map(int, size.split("x"))
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
ctr = [0]
def frames(mode, n):
    d, p = (dp, pp) if mode == "P" else (do, po)
    for k in range(n):
        i = ctr[0]; ctr[0] += 1
        d.denoise(outs[i & 1], d_in[i % 4], d_g[i % 4], cams[i % 4], p, stream=s)
def region(mode, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    frames(mode, n)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def sustain(mode, sec):
    t = time.perf_counter(); k = 0
    while time.perf_counter() - t < sec:
        frames(mode, 32); k += 32
        if k % 128 == 0: torch.cuda.synchronize()
    torch.cuda.synchronize()
tm = telemetry.Sampler(0, period_s=0.004).start()
res = {"knobs": knobs, "lib_sha": __import__("hashlib").sha256(open(pkg.binding.LIB_PATH, "rb").read()).hexdigest()[:10], "size": size}
for mode in ("O", "P"):
    sustain(mode, 0.5)
    t0 = time.perf_counter()
    r = [region(mode, 100) for _ in range(5)]
    t1 = time.perf_counter()
    tms = tm.summary(t0, t1)
    res[mode] = round(float(np.median(r)), 5)
    res[mode + "_w"] = tms["power_w"]["median"] if tms["power_w"] else None
    res[mode + "_mhz"] = tms["sclk_mhz"]["median"] if tms["sclk_mhz"] else None
    res[mode + "_mJ"] = round(res[mode] * res[mode + "_w"], 2) if res[mode + "_w"] else None
tm.stop()
do.profile_stride(1); do.profile_enable(16)
sustain("O", 0.3)
do.profile_enable(16)
frames("O", 16); torch.cuda.synchronize()
rows = [do.profile_read(k) for k in range(16)]
res["temporal_us"] = round(float(np.mean([r[0][1] for r in rows])) * 1e3, 2)
res["levels_us"] = [round(float(np.mean([r[1 + l][1] for r in rows])) * 1e3, 2) for l in range(5)]
res["level_mean_us"] = round(float(np.mean(res["levels_us"])), 2)
print("AB " + json.dumps(res), flush=True)
