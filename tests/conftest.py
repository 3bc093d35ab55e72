import os
import sys

import numpy as np
import pytest

# The frame-pipeline tests want the context's two internal streams on hardware queues of their own (the library probes this and
# refuses the promise otherwise, include/svgf.h): the HIP runtime reads the limit once, when it starts — before torch is imported.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden", "ref_gpu")
PARAM_KEYS = ["temporal_enable", "spatial_enable", "color_alpha", "moment_alpha", "blur_variance", "sigma_l",
              "sigma_x", "sigma_n", "atrous_nlevel", "history_level", "sepcolor", "addcolor", "right_view_option"]
FLOAT_KEYS = {"color_alpha", "moment_alpha", "sigma_l", "sigma_x", "sigma_n"}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "experiments: runs a parked kernel variant / tuning knob of the EXPERIMENTS build "
                            "(libsvgf_hip_exp.so, -DSVGF_BUILD_EXPERIMENTS); the product library has none of them")


@pytest.fixture(scope="session")
def pkg():
    ge.build()          # hipcc cross-compiles for gfx950 without a GPU; gcc builds the oracle
    return ge.load_package()


@pytest.fixture(scope="session")
def orc(pkg):
    return ge.load_oracle()


EXPERIMENT_VARIANTS = (5, 6)     # kernel_variant values that exist in the experiments build only


def denoiser_for(pkg, W, H, variant=0, device=0):
    """A context in the product library, or — for the parked variants 5 / 6 — in the experiments build of the same sources."""
    return pkg.Denoiser(W, H, device, experiments=variant in EXPERIMENT_VARIANTS)


@pytest.fixture
def experiments_lib(pkg):
    """Tests of parked experiments: every context / producer call of the test goes to libsvgf_hip_exp.so, with an empty tuning table
    before and after."""
    if not os.path.exists(pkg.binding.LIB_EXP_PATH):
        pytest.skip("libsvgf_hip_exp.so not built (python -c 'import __graft_entry__ as g; g.build()')")
    pkg.binding.use_experiments_library(True)
    pkg.binding.exp_clear()
    yield pkg.binding
    pkg.binding.exp_clear()
    pkg.binding.use_experiments_library(False)


def relerr(a, b):
    """per-channel relative error with an absolute floor of 1e-2 in the denominator (images are O(0.01..2))."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    both_nan = np.isnan(a) & np.isnan(b)
    e = np.abs(np.where(both_nan, 0, a) - np.where(both_nan, 0, b)) / np.maximum(np.abs(np.where(both_nan, 0, b)), 1e-2)
    return np.where(np.isnan(e), np.inf, e)


def params_from_row(pkg, row):
    p = pkg.SvgfParams()
    for k, v in zip(PARAM_KEYS, row):
        setattr(p, k, float(v) if k in FLOAT_KEYS else int(round(float(v))))
    return p


def cam_from_row(pkg, row):
    c = pkg.SvgfCamera()
    for i, k in enumerate(("right", "up", "view", "position")):
        for j in range(3):
            getattr(c, k)[j] = float(row[3 * i + j])
    return c


def golden_cases():
    import glob
    return sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    runs = bytes(z["runs"]).decode().split(",")
    return z, runs


def replay(pkg, engine, z, tag):
    """Feed one golden run (sequence of calls) to `engine` (Oracle or a Denoiser adapter); returns stacked outputs."""
    outs = []
    for i in range(len(z[f"call_frame_{tag}"])):
        if int(z[f"call_reset_{tag}"][i]):
            engine.reset()
        fi = int(z[f"call_frame_{tag}"][i])
        p = params_from_row(pkg, z[f"call_params_{tag}"][i])
        cam = cam_from_row(pkg, z["cams"][fi])
        outs.append(engine.denoise(z["color"][fi], z["gbuffer"][fi], cam, p))
    return np.stack(outs)
