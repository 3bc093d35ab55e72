#!/usr/bin/env python3
"""Generate golden vectors by RUNNING THE REFERENCE'S OWN denoiser on an MI355X.

The reference (ZheyuanXie/CUDA-Path-Tracer-Denoising) ships no tests or fixtures for its denoiser, so the oracle is
pinned against outputs of the reference itself: `oracle/_ref/ref_denoise_gpu` is the reference's src/denoise.cu
built for gfx950 by oracle/ref/Makefile (hipify-perl + two header fix-ups, see that Makefile), driven by
oracle/ref/ref_driver.cpp.  This script
  1. builds the seeded inputs (cuda-path-tracer-denoising_amd/synth.py) for every case below,
  2. writes a case file, runs the reference binary on the GPU, reads back what denoise() returned,
  3. stores inputs + parameters + cameras + reference outputs as tests/golden/ref_gpu/<case>.npz
     (np.savez_compressed; raw little-endian arrays inside).
Run on a GPU box:   python tests/golden/make_ref_gpu_goldens.py --out gpurun_out/ref_gpu
then copy gpurun_out/ref_gpu/*.npz to tests/golden/ref_gpu/ and commit.  `--compare` additionally runs the CPU
oracle and the HIP library on every case and prints the three-way differences.

Cases marked race_free=True do not depend on how the reference's in-place variance update races
(reference src/denoise.cu:111,117,153,161): the variance plane is uniform there (temporal off -> 10.0, first
temporal frame -> 100.0; a uniform plane is a fixed point of the update) or the a-trous pass does not run.
"""
from __future__ import annotations

import argparse
import json
import os
import struct
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

REF_BIN = os.path.join(ROOT, "oracle", "_ref", "ref_denoise_gpu")              # default hipcc flags (FMA contraction on)
REF_BIN_NOFMA = os.path.join(ROOT, "oracle", "_ref", "ref_denoise_gpu_nofma")  # -ffp-contract=off
MAGIC = 0x43475653

KEEP_DEFAULT_BUILD = {"temporal_moving64", "temporal_static96x54", "full_static96x54"}

PARAM_KEYS = ["temporal_enable", "spatial_enable", "color_alpha", "moment_alpha", "blur_variance", "sigma_l",
              "sigma_x", "sigma_n", "atrous_nlevel", "history_level", "sepcolor", "addcolor", "right_view_option"]


def default_params(**kw):
    p = dict(temporal_enable=0, spatial_enable=0, color_alpha=0.2, moment_alpha=0.2, blur_variance=1, sigma_l=0.45,
             sigma_x=0.35, sigma_n=0.2, atrous_nlevel=5, history_level=1, sepcolor=0, addcolor=0, right_view_option=0)
    p.update(kw)
    return p


def pack_call(reset, frame_index, p, cam, repeat=0):
    return struct.pack("<4i2fi3f5i12fi", int(reset), int(frame_index), int(p["temporal_enable"]),
                       int(p["spatial_enable"]), p["color_alpha"], p["moment_alpha"], int(p["blur_variance"]),
                       p["sigma_l"], p["sigma_x"], p["sigma_n"], int(p["atrous_nlevel"]), int(p["history_level"]),
                       int(p["sepcolor"]), int(p["addcolor"]), int(p["right_view_option"]),
                       *[float(v) for k in ("right", "up", "view", "position") for v in cam[k]], int(repeat))


def run_reference(W, H, calls, frames, workdir, binary=None):
    """calls: list of (reset, frame_index, params, cam).  frames: list of (color, gbuffer).  Returns (outs, ms)."""
    os.makedirs(workdir, exist_ok=True)
    case = os.path.join(workdir, "case.bin")
    outp = os.path.join(workdir, "out.bin")
    with open(case, "wb") as f:
        f.write(struct.pack("<5i", MAGIC, W, H, len(calls), len(frames)))
        for (reset, fi, p, cam) in calls:
            f.write(pack_call(reset, fi, p, cam))
        for (c, g) in frames:
            f.write(np.ascontiguousarray(c, dtype="<f4").tobytes())
            f.write(np.ascontiguousarray(g).tobytes())
    r = subprocess.run([binary or REF_BIN, case, outp], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    if r.returncode != 0:
        raise RuntimeError("reference binary failed: " + r.stdout)
    raw = np.fromfile(outp, dtype="<f4")
    n = W * H * 3
    outs = raw[: n * len(calls)].reshape(len(calls), H, W, 3).copy()
    ms = raw[n * len(calls):].copy()
    os.remove(case)
    os.remove(outp)
    return outs, ms


# ------------------------------------------------------------------------------------------------------------------
# case list
# ------------------------------------------------------------------------------------------------------------------

def synth_frames(pkg, W, H, nframes, seed, moving):
    fr, cams = [], []
    for f in range(nframes):
        c, g, cam = pkg.synth.render_frame(W, H, f, seed=seed, moving=moving)
        fr.append((c, g))
        cams.append({k: np.asarray(cam[k], dtype=np.float32) for k in ("right", "up", "view", "position")})
    return fr, cams


def build_cases(pkg):
    """Returns list of dict(name, W, H, frames, cams, calls=[(reset, frame_index, params)], race_free, note)."""
    cases = []
    S = pkg.synth

    # A/B. a-trous alone, temporal off (variance == 10 everywhere): every step size, borders, blur on/off, other sigmas,
    #      albedo re-modulation; and the first temporal frame + a-trous (variance == 100 everywhere).  One shared input.
    fr, cams = synth_frames(pkg, 96, 96, 1, seed=11, moving=False)
    runs = {}
    for nl in (1, 2, 3, 5, 7):
        runs[f"n{nl}"] = [(1, 0, default_params(spatial_enable=1, atrous_nlevel=nl))]
    runs["n3_noblur"] = [(1, 0, default_params(spatial_enable=1, atrous_nlevel=3, blur_variance=0))]
    runs["n2_sigmas"] = [(1, 0, default_params(spatial_enable=1, atrous_nlevel=2, sigma_l=0.7, sigma_n=0.05, sigma_x=1.5))]
    runs["n2_addcolor"] = [(1, 0, default_params(spatial_enable=1, atrous_nlevel=2, sepcolor=1, addcolor=1))]
    runs["n0"] = [(1, 0, default_params(spatial_enable=1, atrous_nlevel=0))]
    runs["frame0_full"] = [(1, 0, default_params(temporal_enable=1, spatial_enable=1))]
    cases.append(dict(name="atrous_synth96", W=96, H=96, frames=fr, cams=cams, race_free=True, runs=runs,
                      note="single frame; nK = temporal off + K a-trous levels; frame0_full = temporal on, first frame"))
    fr169, cams169 = synth_frames(pkg, 128, 72, 1, seed=12, moving=False)
    cases.append(dict(name="atrous_synth128x72_n5", W=128, H=72, frames=fr169, cams=cams169, race_free=True,
                      calls=[(1, 0, default_params(spatial_enable=1, atrous_nlevel=5))], note="16:9"))
    # 470x11: the library's auto selection takes the lane-marching kernel (480-column strips fit); 1280x9: it takes the
    # strip kernel for every step (svgf_api.hip lane_pays)
    for (W, H, seed) in ((37, 23, 5), (5, 3, 6), (1, 1, 7), (64, 9, 8), (470, 11, 15), (1280, 9, 16)):
        c, g = S.random_frame(W, H, seed=seed)
        cam0 = S.camera_for_frame(0, False)
        cam0 = {k: np.asarray(cam0[k], dtype=np.float32) for k in ("right", "up", "view", "position")}
        cases.append(dict(name=f"atrous_rand{W}x{H}_n5", W=W, H=H, frames=[(c, g)], cams=[cam0], race_free=True,
                          calls=[(1, 0, default_params(spatial_enable=1, atrous_nlevel=5))], note="random texels, odd size"))
    # all-miss frame and zero-variance-like constant colour
    c, g = S.random_frame(48, 40, seed=9)
    g["geomId"] = -1
    cases.append(dict(name="atrous_allmiss48x40_n3", W=48, H=40, frames=[(c, g)], cams=[cam0], race_free=True,
                      calls=[(1, 0, default_params(spatial_enable=1, atrous_nlevel=3))], note="geomId=-1 everywhere"))
    c, g = S.random_frame(48, 40, seed=10)
    c[...] = np.float32(0.37)
    cases.append(dict(name="atrous_constcolor48x40_n3", W=48, H=40, frames=[(c, g)], cams=[cam0], race_free=True,
                      calls=[(1, 0, default_params(spatial_enable=1, atrous_nlevel=3))], note="constant colour"))
    # NaN position texel (min(1, expf(NaN)) -> 1 on the GPU)
    c, g = S.random_frame(40, 32, seed=13)
    g["position"][10, 17] = np.nan
    cases.append(dict(name="atrous_nanpos40x32_n2", W=40, H=32, frames=[(c, g)], cams=[cam0], race_free=True,
                      calls=[(1, 0, default_params(spatial_enable=1, atrous_nlevel=2))], note="one NaN position texel"))
    c, g = S.random_frame(40, 32, seed=14)
    c[5, 6, 1] = np.nan
    cases.append(dict(name="atrous_nancolor40x32_n2", W=40, H=32, frames=[(c, g)], cams=[cam0], race_free=True,
                      calls=[(1, 0, default_params(spatial_enable=1, atrous_nlevel=2))], note="one NaN colour texel"))

    # C. temporal pass alone over sequences: out = colour_acc (spatial off), variance (view 2), history length (view 1)
    for (nm, W, H, moving) in (("static64", 64, 64, False), ("moving64", 64, 64, True), ("static96x54", 96, 54, False),
                               ("moving96x54", 96, 54, True)):
        frs, cs = synth_frames(pkg, W, H, 5, seed=21, moving=moving)
        runs = {}
        for (tag, kw) in (("acc", dict()), ("var", dict(right_view_option=2)), ("hlen", dict(right_view_option=1))):
            runs[tag] = [(1 if f == 0 else 0, f, default_params(temporal_enable=1, spatial_enable=0, **kw)) for f in range(5)]
        cases.append(dict(name=f"temporal_{nm}", W=W, H=H, frames=frs, cams=cs, race_free=True, runs=runs,
                          note="temporal on, spatial off: a-trous never runs; acc = colour_acc, var = variance/0.1, hlen = (pre-update) history length/100"))
    frs, cs = synth_frames(pkg, 64, 64, 4, seed=22, moving=True)
    calls = [(1 if f == 0 else 0, f, default_params(temporal_enable=1, spatial_enable=0, color_alpha=0.05, moment_alpha=0.5)) for f in range(4)]
    cases.append(dict(name="temporal_moving64_alphas_acc", W=64, H=64, frames=frs, cams=cs, race_free=True, calls=calls, note="other alphas"))

    # D. full SVGF sequences (temporal + spatial): raced in the reference after frame 0
    for (nm, W, H, moving) in (("static64", 64, 64, False), ("moving80", 80, 80, True), ("static96x54", 96, 54, False)):
        frs, cs = synth_frames(pkg, W, H, 4, seed=31, moving=moving)
        runs = {}
        for hl in ((1, 0, 5) if nm == "moving80" else (1,)):
            runs[f"h{hl}"] = [(1 if f == 0 else 0, f, default_params(temporal_enable=1, spatial_enable=1, history_level=hl)) for f in range(4)]
        cases.append(dict(name=f"full_{nm}", W=W, H=H, frames=frs, cams=cs, race_free=False, runs=runs,
                          note="temporal + 5 a-trous levels; hK = history_level K; frames >= 1 depend on the reference's variance race"))
    # mode switches mid-sequence: off, on, on, off, on
    frs, cs = synth_frames(pkg, 64, 64, 5, seed=41, moving=False)
    modes = [0, 1, 1, 0, 1]
    calls = [(1 if f == 0 else 0, f, default_params(temporal_enable=modes[f], spatial_enable=0)) for f in range(5)]
    cases.append(dict(name="temporal_switch64_acc", W=64, H=64, frames=frs, cams=cs, race_free=True, calls=calls,
                      note="temporal toggled between frames, spatial off"))
    return cases


def relerr(a, b):
    return np.abs(a - b) / np.maximum(np.abs(b), 1e-2)


def summarize(a, b):
    both_nan = np.isnan(a) & np.isnan(b)
    e = relerr(np.where(both_nan, 0, a), np.where(both_nan, 0, b))
    e = np.where(np.isnan(e), np.inf, e)
    return dict(max=float(e.max()), p999=float(np.quantile(e, 0.999)), mean=float(e[np.isfinite(e)].mean() if np.isfinite(e).any() else np.inf),
                frac_gt_1e4=float((e > 1e-4).mean()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "ref_gpu"))
    ap.add_argument("--compare", action="store_true", help="also run the CPU oracle and the HIP library")
    ap.add_argument("--time-1080p", action="store_true", help="time the reference's denoise() at 1920x1080")
    ap.add_argument("--only", default="", help="comma-separated case names: generate just these")
    a = ap.parse_args()
    pkg = ge.load_package()
    if not os.path.exists(REF_BIN):
        raise SystemExit(f"{REF_BIN} missing (built only where /root/reference exists; it travels with the snapshot)")
    os.makedirs(a.out, exist_ok=True)
    work = os.path.join("/tmp", "svgf_ref_work")
    report = {}
    cases = build_cases(pkg)
    if a.only:
        cases = [c for c in cases if c["name"] in a.only.split(",")]
    orc = ge.load_oracle() if a.compare else None
    for cs in cases:
        W, H = cs["W"], cs["H"]
        runs = cs.get("runs") or {"out": cs["calls"]}
        arrays = dict(
            W=np.int32(W), H=np.int32(H), race_free=np.int32(1 if cs["race_free"] else 0),
            color=np.stack([f[0] for f in cs["frames"]]).astype("<f4"),
            gbuffer=np.stack([f[1] for f in cs["frames"]]),
            cams=np.stack([np.concatenate([c[k] for k in ("right", "up", "view", "position")]) for c in cs["cams"]]).astype("<f4"),
            runs=np.bytes_(",".join(runs.keys()).encode()), note=np.bytes_(cs["note"].encode()))
        for tag, rcalls in runs.items():
            calls = [(r, fi, p, cs["cams"][fi]) for (r, fi, p) in rcalls]
            outs, ms = run_reference(W, H, calls, cs["frames"], work)
            outs2, _ = run_reference(W, H, calls, cs["frames"], work)   # is the reference deterministic here?
            rerun = summarize(outs2, outs)
            outs_nf, _ = run_reference(W, H, calls, cs["frames"], work, binary=REF_BIN_NOFMA)
            arrays[f"call_reset_{tag}"] = np.array([c[0] for c in rcalls], dtype=np.int32)
            arrays[f"call_frame_{tag}"] = np.array([c[1] for c in rcalls], dtype=np.int32)
            arrays[f"call_params_{tag}"] = np.array([[float(c[2][k]) for k in PARAM_KEYS] for c in rcalls], dtype="<f8")
            # primary golden: the -ffp-contract=off build (arithmetic exactly as the reference source states it).
            # The default-flags build (FMA contraction on) is kept for a subset, to bound what contraction changes.
            arrays[f"ref_nofma_out_{tag}"] = outs_nf.astype("<f4")
            if cs["name"] in KEEP_DEFAULT_BUILD:
                arrays[f"ref_out_{tag}"] = outs.astype("<f4")
            entry = dict(W=W, H=H, ncalls=len(calls), race_free=cs["race_free"], ref_rerun=rerun)
            if a.compare:
                P = pkg.SvgfParams
                o = orc.Oracle(pkg, W, H, threads=8)
                d = pkg.Denoiser(W, H, device=0)
                o_out, d_out = [], []
                for (r, fi, p, cam) in calls:
                    if r:
                        o.reset(); d.reset()
                    pp = P(); pp.set(**{k: (p[k] if isinstance(p[k], float) else int(p[k])) for k in PARAM_KEYS})
                    o_out.append(o.denoise(cs["frames"][fi][0], cs["frames"][fi][1], cam, pp))
                    d_out.append(d.denoise_host(cs["frames"][fi][0], cs["frames"][fi][1], cam, pp))
                o.free(); d.free()
                o_out = np.stack(o_out); d_out = np.stack(d_out)
                entry["oracle_vs_ref"] = summarize(o_out, outs)
                entry["oracle_vs_ref_nofma"] = summarize(o_out, outs_nf)
                entry["hip_vs_ref_nofma"] = summarize(d_out, outs_nf)
                entry["hip_vs_ref"] = summarize(d_out, outs)
                entry["hip_vs_oracle"] = summarize(d_out, o_out)
                entry["per_call_oracle_vs_ref_fracgt1e4"] = [summarize(o_out[i], outs[i])["frac_gt_1e4"] for i in range(len(calls))]
            report[cs["name"] + ":" + tag] = entry
            line = f"{cs['name'] + ':' + tag:40s} rerun max {rerun['max']:.1e}"
            if a.compare:
                line += (f" | oracle-refNOFMA max {entry['oracle_vs_ref_nofma']['max']:.1e} >1e-4 {entry['oracle_vs_ref_nofma']['frac_gt_1e4']:.4f}"
                         f" | oracle-ref max {entry['oracle_vs_ref']['max']:.1e} >1e-4 {entry['oracle_vs_ref']['frac_gt_1e4']:.4f}"
                         f" | hip-ref max {entry['hip_vs_ref']['max']:.1e} >1e-4 {entry['hip_vs_ref']['frac_gt_1e4']:.4f}"
                         f" | hip-oracle max {entry['hip_vs_oracle']['max']:.1e}")
            print(line, flush=True)
        np.savez_compressed(os.path.join(a.out, cs["name"] + ".npz"), **arrays)

    if a.time_1080p:
        W, H = 1920, 1080
        frs, cs_ = synth_frames(pkg, W, H, 2, seed=51, moving=False)
        full = default_params(temporal_enable=1, spatial_enable=1)
        calls = [(1 if i == 0 else 0, i % 2, full, cs_[i % 2]) for i in range(8)]
        t0 = time.time()
        _, ms = run_reference(W, H, calls, frs, work)
        c1 = default_params(temporal_enable=0, spatial_enable=1, atrous_nlevel=1)
        _, ms1 = run_reference(W, H, [(1, 0, c1, cs_[0])] * 1 + [(0, 0, c1, cs_[0])] * 5, frs, work)
        report["_timing_reference_on_mi355x"] = dict(
            W=W, H=H, full_svgf_ms_per_call=[float(x) for x in ms], atrous1_ms_per_call=[float(x) for x in ms1],
            note="wall time of the reference's denoise() (its own kernels + 5 D2D copies + device sync), hipified, on this GPU",
            wall_s=time.time() - t0)
        print("reference denoise() on this GPU, 1080p full SVGF ms:", [round(float(x), 3) for x in ms])
        print("reference denoise() on this GPU, 1080p 1 level  ms:", [round(float(x), 3) for x in ms1])
    with open(os.path.join(a.out, "report.json"), "w") as f:
        json.dump(report, f, indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
