"""MI355X-native SVGF denoiser behind the reference's denoise.h entry points.

Directory layout:
  csrc/                HIP kernels (gfx950) + the C ABI of include/svgf.h  -> libsvgf_hip.so (built in-tree)
  binding.py           ctypes binding + `Denoiser`, the host-side mirror of denoiseInit/denoise/denoiseFree
  farm.py              sequence sharding + timing reduction for the 1/2/4/8-GPU runs (no collectives on the data path)
  synth.py             seeded synthetic 1-spp colour + G-buffer frames (test / bench inputs)
  build.py             hipcc / gcc recipes used by __graft_entry__.build()
The directory name contains '-', so it is loaded through `__graft_entry__.load_package()` under the module
name `cuda_path_tracer_denoising_amd`.
"""
from . import binding, build, farm, scene, synth  # noqa: F401
from .binding import Denoiser, SvgfCamera, SvgfParams, SvgfError, load_library, reference_defaults  # noqa: F401
