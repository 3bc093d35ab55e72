"""The C-ABI library builds for gfx950, loads on a CPU-only host and exports every symbol include/svgf.h declares.
No compute is issued here (there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def header_functions():
    src = open(os.path.join(ROOT, "include", "svgf.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(svgf_[a-z_]+)\s*\(", src)))


def test_header_symbols_exported(pkg):
    lib = pkg.load_library()
    names = header_functions()
    assert len(names) >= 16
    for n in names:
        assert hasattr(lib, n), f"libsvgf_hip.so does not export {n}"
    assert sorted(pkg.binding.EXPORTS) == names


def test_struct_layouts_match_reference(pkg):
    # GBufferTexel: 52 B, align 4, offsets 0/12/24/36/48 (reference src/sceneStructs.h:113-119)
    dt = pkg.synth.GBUFFER_DTYPE
    assert dt.itemsize == 52
    assert [dt.fields[k][1] for k in ("normal", "position", "albedo", "ialbedo", "geomId")] == [0, 12, 24, 36, 48]
    assert ctypes.sizeof(pkg.SvgfCamera) == 48
    assert ctypes.sizeof(pkg.SvgfParams) == 20 * 4


def test_params_default_matches_reference_ui_defaults(pkg):
    lib = pkg.load_library()
    p = pkg.SvgfParams()
    assert lib.svgf_params_default(ctypes.byref(p)) == 0
    q = pkg.reference_defaults()        # reference src/main.cpp:49-62
    for name, _ in pkg.SvgfParams._fields_[:15]:
        assert getattr(p, name) == pytest.approx(getattr(q, name)), name
    assert lib.svgf_version() == (0 << 16) | 9


def test_params_size_is_exported_and_checked(pkg):
    """SvgfParams grows at its tail between ABI versions (0.2: 72 bytes, 0.3: 80) and svgf_denoise reads all of it: the
    library says how large ITS struct is so that a binding can refuse a mismatch at load time (binding.load_library does)."""
    lib = pkg.load_library()
    assert lib.svgf_params_sizeof() == ctypes.sizeof(pkg.SvgfParams) == 80
    assert lib.svgf_planar_gbuffer(None, None) == -1 and lib.svgf_denoise_planar(None, None, None, None, None, None) == -1


def test_error_paths_without_gpu(pkg):
    import torch
    lib = pkg.load_library()
    h = ctypes.c_void_p()
    assert lib.svgf_create(0, 0, 10, ctypes.byref(h)) == -1            # SVGF_ERR_INVALID_ARG
    assert lib.svgf_create(0, 16, 16, None) == -1
    assert lib.svgf_create_ex(0, 16, 16, 0x80, ctypes.byref(h)) == -1  # unknown flag bits
    assert lib.svgf_enable_pipeline(None) == -1 and lib.svgf_pipeline_status(None) == 0
    assert lib.svgf_destroy(None) == 0                                 # denoiseFree on NULL is harmless
    assert lib.svgf_reset(None) == -1
    if not torch.cuda.is_available():
        rc = lib.svgf_create(0, 16, 16, ctypes.byref(h))
        assert rc == -2, "without a GPU the library must fail loudly (SVGF_ERR_NO_DEVICE), not fall back"
        assert b"no usable HIP device" in lib.svgf_last_error(None)
        with pytest.raises(pkg.SvgfError):
            pkg.Denoiser(16, 16)


def test_product_never_touches_the_oracle():
    """The shipped path must not import/link/execute anything under oracle/ — nor build it (oracle/build_oracle.py does)."""
    pk = os.path.join(ROOT, "cuda-path-tracer-denoising_amd")
    for dirpath, _, files in os.walk(pk):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "svgf_oracle" not in text and "oracle_py" not in text and "build_oracle" not in text, os.path.join(dirpath, f)
    import subprocess
    out = subprocess.run(["ldd", os.path.join(pk, "libsvgf_hip.so")], stdout=subprocess.PIPE, text=True).stdout
    assert "oracle" not in out
    # nor do the tools: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
    for dirpath, _, files in os.walk(os.path.join(ROOT, "tools")):
        for f in files:
            if f.endswith((".py", ".sh")):
                text = open(os.path.join(dirpath, f)).read()
                assert "load_oracle" not in text and "svgf_oracle" not in text and "oracle_py" not in text, os.path.join(dirpath, f)
    bench = open(os.path.join(ROOT, "bench.py")).read()
    assert bench.count("load_oracle") == 1 and "def cpu_baseline" in bench      # one use, inside the cpu_baseline leg


def test_product_build_has_no_experiments_and_reads_no_environment(pkg):
    """The product library: no getenv anywhere in csrc/ (tuning knobs are svgf_exp_set of the experiments build), no tuning entry
    point, none of the parked kernels (FUSED = 1 / 2 / 4, two-y-phase geometry, cross-level term reuse), under 1.5 MB."""
    import subprocess
    pk = os.path.join(ROOT, "cuda-path-tracer-denoising_amd")
    for f in os.listdir(os.path.join(pk, "csrc")):
        assert "getenv" not in open(os.path.join(pk, "csrc", f)).read(), f
    lib = os.path.join(pk, "libsvgf_hip.so")
    syms = subprocess.run(["nm", "-D", "--defined-only", lib], stdout=subprocess.PIPE, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in syms.splitlines() if ln.strip()}
    assert "svgf_exp_set" not in exported and "svgf_exp_clear" not in exported
    assert "getenv" not in subprocess.run(["nm", "-D", "--undefined-only", lib], stdout=subprocess.PIPE, text=True, check=True).stdout
    assert os.path.getsize(lib) < 1.5 * 1024 * 1024, os.path.getsize(lib)
    assert pkg.load_library().svgf_build_has_experiments() == 0
    # the parked kernels: template arguments <LOG2S, HASVAR, LOG2P, LOG2Y, FUSED, REUSE> of k_atrous_lane in the device code objects
    names = subprocess.run(["strings", "-n", "20", lib], stdout=subprocess.PIPE, text=True, check=True).stdout
    lane = sorted({ln.strip() for ln in names.splitlines() if "k_atrous_laneILi" in ln and ln.strip().startswith("_ZN")})
    assert lane, "the lane kernel's symbols are in the library"
    import re
    for sym in lane:
        m = re.search(r"k_atrous_laneILi(\d+)ELb([01])ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", sym)
        assert m, sym
        log2y, fused, reuse = int(m.group(4)), int(m.group(5)), int(m.group(6))
        assert log2y == 0 and fused in (0, 3) and reuse == 0, sym
    exp = os.path.join(pk, "libsvgf_hip_exp.so")
    if os.path.exists(exp):      # the experiments build of the same sources carries what the product build leaves out
        e = pkg.load_library(experiments=True)
        assert e.svgf_build_has_experiments() == 1 and hasattr(e, "svgf_exp_set")
