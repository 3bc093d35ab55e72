#!/usr/bin/env python3
"""Real-scene fixtures (SURVEY.md §8c fixtures 1-3) made by RUNNING THE REFERENCE: its own path tracer renders its own
scenes (cornell.txt, room.txt) and hands 1-spp colour + G-buffer to denoise(); its own denoiser turns them into the
expected outputs; its own sendTwoImagesToPBO / image::savePNG / Scene loader give the goldens of the rows next to the
path (display pack, PNG writer, scene text format).

Binaries (oracle/ref/Makefile, built only where /root/reference exists; they travel to the GPU box with the snapshot):
  oracle/_ref/ref_pathtrace_capture   reference pathtrace.cu + scene loader + our capturing denoise()   (needs a GPU)
  oracle/_ref/ref_denoise_gpu[_nofma] reference denoise.cu                                              (needs a GPU)
  oracle/_ref/ref_host_tools          reference Scene loader / image::savePNG                           (host only)

  python tests/golden/make_ref_scene_goldens.py --gpu  --out gpurun_out/ref_scenes     # on the GPU box
  python tests/golden/make_ref_scene_goldens.py --host --out tests/golden/ref_scenes   # in the build container
then copy gpurun_out/ref_scenes/*.npz to tests/golden/ref_scenes/ and commit.  The .npz files use the schema of
tests/golden/ref_gpu/ (make_ref_gpu_goldens.py) plus `pbo` / `pattern` (display-pack golden).
"""
from __future__ import annotations

import argparse
import json
import os
import struct
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
PARAM_KEYS = ["temporal_enable", "spatial_enable", "color_alpha", "moment_alpha", "blur_variance", "sigma_l",
              "sigma_x", "sigma_n", "atrous_nlevel", "history_level", "sepcolor", "addcolor", "right_view_option"]
# name, scene, W, H, frames, moving camera, sepcolor+addcolor
CASES = [
    ("cornell96_static", "cornell.txt", 96, 96, 4, 0, 0),
    ("cornell128x72_moving", "cornell.txt", 128, 72, 4, 1, 0),
    ("room128x72_static_sepcolor", "room.txt", 128, 72, 4, 0, 1),
    ("bunny128x72_static", "bunny.txt", 128, 72, 4, 0, 0),          # BASELINE configs[3]'s scene (4 968 triangles)
]
CALL_FMT = "<4i2fi3f5i12fi"


def run(cmd, cwd=None):
    r = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    if r.returncode != 0:
        raise RuntimeError(" ".join(cmd) + "\n" + r.stdout[-3000:])
    return r.stdout


def gpu_part(out):
    pkg = ge.load_package()
    os.makedirs(out, exist_ok=True)
    work = "/tmp/svgf_ref_scene_work"
    os.makedirs(work, exist_ok=True)
    for (name, scene, W, H, n, moving, sep) in CASES:
        prefix = os.path.join(work, name)
        # scene.cpp resolves models / textures as ../scenes/...: run inside the copied scenes directory
        print(run([os.path.join(REF, "ref_pathtrace_capture"), scene, str(W), str(H), str(n), str(moving), str(sep), prefix],
                  cwd=os.path.join(REF, "scenes")).strip().splitlines()[-1])
        raw = open(prefix + ".case", "rb").read()
        hdr = struct.unpack_from("<5i", raw, 0)
        assert hdr[1:] == (W, H, n, n)
        csz = struct.calcsize(CALL_FMT)
        calls = [struct.unpack_from(CALL_FMT, raw, 20 + k * csz) for k in range(n)]
        off = 20 + n * csz
        npx = W * H
        color = np.zeros((n, H, W, 3), "<f4")
        gbuf = np.zeros((n, H, W), pkg.synth.GBUFFER_DTYPE)
        for f in range(n):
            color[f] = np.frombuffer(raw, "<f4", npx * 3, off).reshape(H, W, 3); off += npx * 12
            gbuf[f] = np.frombuffer(raw, pkg.synth.GBUFFER_DTYPE, npx, off).reshape(H, W); off += npx * 52
        outs = {}
        for tag, binary in (("ref_nofma_out", "ref_denoise_gpu_nofma"), ("ref_out", "ref_denoise_gpu")):
            run([os.path.join(REF, binary), prefix + ".case", prefix + ".out"])
            o = np.fromfile(prefix + ".out", "<f4")
            outs[tag] = o[: n * npx * 3].reshape(n, H, W, 3).copy()
        params = np.array([[c[2], c[3], c[4], c[5], c[6], c[7], c[8], c[9], c[10], c[11], c[12], c[13], c[14]] for c in calls], "<f8")
        cams = np.array([c[15:27] for c in calls], "<f4")
        pbo = np.fromfile(prefix + ".pbo", np.uint8).reshape(n, H, 2 * W, 4)
        pattern = np.fromfile(prefix + ".pattern", "<f4").reshape(n, H, W, 3)
        np.savez_compressed(
            os.path.join(out, name + ".npz"), W=np.int32(W), H=np.int32(H), race_free=np.int32(0), color=color, gbuffer=gbuf,
            cams=cams, runs=np.bytes_(b"out"), note=np.bytes_((f"{scene} rendered by the reference's own path tracer, {n} frames, "
                                                               f"{'moving' if moving else 'static'} camera, sepcolor=addcolor={sep}").encode()),
            call_reset_out=np.array([c[0] for c in calls], np.int32), call_frame_out=np.array([c[1] for c in calls], np.int32),
            call_params_out=params, ref_nofma_out_out=outs["ref_nofma_out"], ref_out_out=outs["ref_out"], pbo=pbo, pattern=pattern)
        hit = float((gbuf["geomId"] >= 0).mean())
        print(f"  {name}: hit fraction {hit:.3f}, colour mean {float(np.nanmean(color)):.4f}, "
              f"ref nofma-vs-default max rel {float(np.nanmax(np.abs(outs['ref_out'] - outs['ref_nofma_out']) / np.maximum(np.abs(outs['ref_nofma_out']), 1e-2))):.2e}")
    print("wrote", out)


def host_part(out):
    os.makedirs(out, exist_ok=True)
    rec = {}
    for s in ("cornell", "room", "bunny", "diamond"):
        txt = run([os.path.join(REF, "ref_host_tools"), "scene", s + ".txt"], cwd=os.path.join(REF, "scenes"))
        rec[s] = json.loads(txt[txt.index('{"camera"'):])
    with open(os.path.join(out, "scene_records.json"), "w") as f:
        json.dump(rec, f, indent=0, separators=(",", ":"))
    base = os.path.join(out, "ref_savepng_37x5")
    run([os.path.join(REF, "ref_host_tools"), "png", base])
    print("wrote", out, {k: (len(v["materials"]), len(v["geoms"])) for k, v in rec.items()})


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpu", action="store_true")
    ap.add_argument("--host", action="store_true")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    if a.gpu:
        gpu_part(a.out)
    if a.host:
        host_part(a.out)
