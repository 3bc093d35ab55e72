"""Build recipe of the product library (no cmake, no JIT cache): everything lands in-tree so it travels to the GPU box.
The checkers (CPU restatement, the reference's own denoise.cu) have their own recipe outside this package."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libsvgf_hip.so")
# The experiments build (-DSVGF_BUILD_EXPERIMENTS): the product's sources + the parked kernel variants and the tuning table
# (svgf_exp_set).  Test / tools infrastructure: nothing on the product path loads it (binding.load_library(experiments=True) does).
LIB_EXP = os.path.join(_HERE, "libsvgf_hip_exp.so")

HIP_SOURCES = ["svgf_api.hip", "svgf_kernels.hip", "svgf_atrous_strip.hip", "svgf_atrous_lane.hip", "svgf_atrous_prepare_fused.hip", "svgf_atrous_lattice.hip", "svgf_synth.hip", "svgf_scene.hip", "svgf_display.hip"]
HIP_SOURCES_EXPERIMENTS = ["svgf_atrous_lane_reuse.hip", "svgf_atrous_fused.hip"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function"]
# Per-file flags.  svgf_atrous_fused.hip: SimplifyCFG's common-code sinking merges the last store of the "bilinear history" branch
# with the last store of the "fallback consistency data" branch of the temporal stage into ONE store through a phi of two
# pointers, which keeps those request registers in scratch memory (every load followed by s_waitcnt vmcnt(0) + scratch_store).
HIPCC_FILE_FLAGS = {"svgf_atrous_fused.hip": ["-mllvm", "-simplifycfg-sink-common=false"]}


def _newer(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def _run(cmd: list[str], cwd: str | None = None):
    r = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: " + " ".join(cmd) + "\n" + r.stdout)
    return r.stdout


def hipcc_path() -> str:
    p = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(p):
        raise RuntimeError("hipcc not found")
    return p


def build_hip(force: bool = False, experiments: bool = False) -> str:
    """Compile the HIP kernels + C ABI for gfx950 into cuda-path-tracer-denoising_amd/libsvgf_hip.so (the product), or, with
    experiments=True, into libsvgf_hip_exp.so (-DSVGF_BUILD_EXPERIMENTS: parked kernel variants 5 / 6, cross-level term reuse, the
    tuning table behind svgf_exp_set; what tools/experiments/ and the tests marked `experiments` load)."""
    LIB = LIB_EXP if experiments else globals()["LIB"]
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES + (HIP_SOURCES_EXPERIMENTS if experiments else [])]
    deps = srcs + [os.path.join(CSRC, h) for h in ("svgf_kernels.h", "svgf_temporal.h", "svgf_atrous_lane_impl.h", "svgf_atrous_lane_tfused.inc.h")] + [os.path.join(ROOT, "include", "svgf.h")]
    if not force and _newer(LIB, deps):
        return LIB
    tmp = LIB + f".tmp{os.getpid()}"
    extra = os.environ.get("SVGF_EXTRA_HIPCC_FLAGS", "").split() if experiments else []      # -D switches of tools/experiments/ (experiments build only)
    if experiments:
        extra = ["-DSVGF_BUILD_EXPERIMENTS"] + extra
    # one hipcc -c per translation unit, in parallel (the a-trous kernels are template-heavy: ~70 s serial, ~30 s so),
    # objects in a scratch directory, then one link
    import concurrent.futures
    import tempfile
    cflags = [f for f in HIPCC_FLAGS if f != "-shared"]
    with tempfile.TemporaryDirectory(prefix="svgf_build_") as scratch:
        objs = [os.path.join(scratch, os.path.basename(x) + ".o") for x in srcs]
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 1)) as pool:
            list(pool.map(lambda so: _run([hipcc_path()] + cflags + HIPCC_FILE_FLAGS.get(os.path.basename(so[0]), []) + extra + ["-c", so[0], "-o", so[1]]),
                          zip(srcs, objs)))
        _run([hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", tmp])
    os.replace(tmp, LIB)              # atomic: concurrent ranks never see a half-written library
    return LIB


def build_examples(force: bool = False) -> str:
    """examples/farm.cpp (N contexts on N GPUs from one C++ process), examples/pipeline.cpp (a renderer's frame loop on the frame
    pipeline) and examples/cadence.cpp (a fixed-rate loop + the effective shader clock), through the C ABI -> examples/farm,
    examples/pipeline, examples/cadence, linked against the product library in-tree."""
    out = ""
    for name in ("pipeline", "farm", "cadence"):
        src = os.path.join(ROOT, "examples", name + ".cpp")
        out = os.path.join(ROOT, "examples", name)
        if not force and _newer(out, [src, os.path.join(ROOT, "include", "svgf.h"), LIB]):
            continue
        tmp = out + f".tmp{os.getpid()}"
        _run([hipcc_path(), "--offload-arch=gfx950", "-O2", "-I", os.path.join(ROOT, "include"), src, "-L", _HERE, "-lsvgf_hip",
              "-Wl,-rpath," + _HERE, "-Wl,-rpath,$ORIGIN/../cuda-path-tracer-denoising_amd", "-o", tmp])
        os.replace(tmp, out)
    return out
