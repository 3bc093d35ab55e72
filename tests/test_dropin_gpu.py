"""Drop-in proof at link level (GPU box): oracle/_ref/ref_driver_svgf is the SAME driver program that feeds the reference
(oracle/ref/ref_driver.cpp calls denoiseInit / denoise / denoiseFree and sets the ui_* globals), linked against
cuda-path-tracer-denoising_amd/csrc/denoise_compat.cpp + libsvgf_hip.so instead of the reference's denoise.cu.
Its outputs on the golden case files must match what the reference binary produced."""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from conftest import ROOT, load_golden, relerr, PARAM_KEYS

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "oracle", "_ref", "ref_driver_svgf")


def run_driver(z, tag):
    W, H = int(z["W"]), int(z["H"])
    nf = len(z["color"])
    calls = len(z[f"call_frame_{tag}"])
    with tempfile.TemporaryDirectory() as td:
        case, outp = os.path.join(td, "case.bin"), os.path.join(td, "out.bin")
        with open(case, "wb") as f:
            f.write(struct.pack("<5i", 0x43475653, W, H, calls, nf))
            for i in range(calls):
                p = dict(zip(PARAM_KEYS, z[f"call_params_{tag}"][i]))
                fi = int(z[f"call_frame_{tag}"][i])
                cam = z["cams"][fi]
                f.write(struct.pack("<4i2fi3f5i12fi", int(z[f"call_reset_{tag}"][i]), fi, int(p["temporal_enable"]),
                                    int(p["spatial_enable"]), p["color_alpha"], p["moment_alpha"], int(p["blur_variance"]),
                                    p["sigma_l"], p["sigma_x"], p["sigma_n"], int(p["atrous_nlevel"]), int(p["history_level"]),
                                    int(p["sepcolor"]), int(p["addcolor"]), int(p["right_view_option"]), *[float(v) for v in cam], 0))
            for k in range(nf):
                f.write(np.ascontiguousarray(z["color"][k], dtype="<f4").tobytes())
                f.write(np.ascontiguousarray(z["gbuffer"][k]).tobytes())
        r = subprocess.run([BIN, case, outp], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
        assert r.returncode == 0, r.stdout
        raw = np.fromfile(outp, dtype="<f4")
    return raw[: calls * H * W * 3].reshape(calls, H, W, 3)


@pytest.mark.skipif(not os.path.exists(BIN), reason="oracle/_ref/ref_driver_svgf is built only where /root/reference exists")
@pytest.mark.parametrize("name,tag", [("atrous_synth96", "n5"), ("atrous_synth96", "n2_addcolor"), ("temporal_moving64", "acc"),
                                      ("temporal_moving64", "hlen"), ("full_moving80", "h1"), ("full_static96x54", "h1"),
                                      ("temporal_switch64_acc", "out"), ("atrous_rand5x3_n5", "out")])
def test_reference_driver_linked_against_svgf(name, tag):
    z, _ = load_golden(name)
    got = run_driver(z, tag)
    ref = z[f"ref_nofma_out_{tag}"]
    if name.startswith("temporal_"):
        assert np.array_equal(got, ref, equal_nan=True)
    else:
        assert relerr(got, ref).max() <= 1e-5
