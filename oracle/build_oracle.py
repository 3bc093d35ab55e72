"""Build recipes of the CHECKERS (test infrastructure, never a product dependency): the CPU oracle oracle/svgf_oracle.c ->
oracle/libsvgf_oracle.so, and — only where /root/reference exists — the reference's own denoise.cu for gfx950 -> oracle/_ref/.
Called by __graft_entry__.build() and the tests; nothing under cuda-path-tracer-denoising_amd/ imports this file."""
from __future__ import annotations

import os
import subprocess

ORACLE_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(ORACLE_DIR)
ORACLE_LIB = os.path.join(ORACLE_DIR, "libsvgf_oracle.so")


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


def build_oracle(force: bool = False) -> str:
    """Compile the CPU oracle (test infrastructure) into oracle/libsvgf_oracle.so."""
    src = os.path.join(ORACLE_DIR, "svgf_oracle.c")
    deps = [src, os.path.join(ORACLE_DIR, "svgf_oracle.h"), os.path.join(ROOT, "include", "svgf.h")]
    if not force and _newer(ORACLE_LIB, deps):
        return ORACLE_LIB
    tmp = ORACLE_LIB + f".tmp{os.getpid()}"
    _run(["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-shared", "-fPIC", "-std=c11", "-Wall", src, "-o", tmp, "-lm"])
    os.replace(tmp, ORACLE_LIB)
    return ORACLE_LIB


def build_reference(force: bool = False) -> str | None:
    """Build the reference's own denoise.cu for gfx950 into oracle/_ref/ (only where /root/reference exists)."""
    if not os.path.isdir("/root/reference/src"):
        return None
    out = os.path.join(ORACLE_DIR, "_ref", "ref_denoise_gpu")
    mk = os.path.join(ORACLE_DIR, "ref", "Makefile")
    if not os.path.exists(mk):
        return None
    if force and os.path.exists(out):
        os.remove(out)
    _run(["make", "-s", "-C", os.path.join(ORACLE_DIR, "ref")])       # raises with make's output when the recipe fails
    scenes = os.path.join(ORACLE_DIR, "_ref", "scenes")
    if not os.path.exists(out) or not os.path.isdir(scenes):
        raise RuntimeError(f"reference build: make succeeded but {out} or {scenes} is missing")
    return out
