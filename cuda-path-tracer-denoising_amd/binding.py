"""ctypes binding of libsvgf_hip.so (include/svgf.h) and a host-side mirror of the reference interface.

The reference's interface for this path is three free functions over global state
(reference src/denoise.h:6-8): denoiseInit(scene) / denoise(out, in, gbuffer) / denoiseFree(), with the tunables
in `ui_*` globals (src/main.h:39-69).  `Denoiser` keeps those names and meanings:
    d = Denoiser(width, height, device=0)     # denoiseInit
    d.ui.temporal_enable = 1 ...              # the ui_* block
    d.denoise(out, inp, gbuffer, camera)      # denoise (device pointers / torch tensors)
    d.free()                                  # denoiseFree
There is NO CPU fallback: if the HIP library is missing or no GPU is present, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsvgf_hip.so")
LIB_EXP_PATH = os.path.join(_HERE, "libsvgf_hip_exp.so")     # -DSVGF_BUILD_EXPERIMENTS: parked variants + tuning table (tests / tools only)

SVGF_OK = 0
STATE_HISTORY_LENGTH, STATE_MOMENTS, STATE_COLOR_HISTORY, STATE_VARIANCE_TEMPORAL, STATE_COLOR_ACC = range(5)
KERNEL_TEMPORAL, KERNEL_PREPARE, KERNEL_ATROUS, KERNEL_DEBUGVIEW, KERNEL_COPYOUT, KERNEL_FUSED = 1, 2, 3, 4, 5, 6
MAX_LEVELS = 10

# every symbol include/svgf.h declares
EXPORTS = ["svgf_version", "svgf_params_default", "svgf_create", "svgf_destroy", "svgf_reset", "svgf_denoise",
           "svgf_denoise_host", "svgf_sync", "svgf_last_error", "svgf_width", "svgf_height", "svgf_read_state",
           "svgf_set_capture", "svgf_profile_enable", "svgf_profile_stride", "svgf_profile_frames", "svgf_profile_read",
           "svgf_synth_camera", "svgf_synth_render", "svgf_scene_render", "svgf_scene_render_mesh", "svgf_display_pack", "svgf_save_png",
           "svgf_planar_gbuffer", "svgf_denoise_planar", "svgf_synth_render_planar", "svgf_params_sizeof", "svgf_scene_render_mesh_planar",
           "svgf_sync_stream", "svgf_build_has_experiments", "svgf_is_pipelined", "svgf_create_ex", "svgf_enable_pipeline",
           "svgf_pipeline_status", "svgf_planar_gbuffer_stream", "svgf_streams_overlap"]
CREATE_PIPELINED = 1


class SvgfCamera(C.Structure):
    _fields_ = [("right", C.c_float * 3), ("up", C.c_float * 3), ("view", C.c_float * 3), ("position", C.c_float * 3)]

    @classmethod
    def from_dict(cls, d):
        c = cls()
        for k in ("right", "up", "view", "position"):
            for i in range(3):
                getattr(c, k)[i] = float(d[k][i])
        return c


class SvgfParams(C.Structure):
    _fields_ = [("temporal_enable", C.c_int), ("spatial_enable", C.c_int), ("color_alpha", C.c_float),
                ("moment_alpha", C.c_float), ("blur_variance", C.c_int), ("sigma_l", C.c_float),
                ("sigma_x", C.c_float), ("sigma_n", C.c_float), ("atrous_nlevel", C.c_int),
                ("history_level", C.c_int), ("sepcolor", C.c_int), ("addcolor", C.c_int),
                ("right_view_option", C.c_int), ("kernel_variant", C.c_int), ("inputs_ready", C.c_int),
                ("reproj_scale", C.c_float * 2), ("paper_steps", C.c_int), ("reproj_position_tol", C.c_float),
                ("spatial_variance_frames", C.c_int)]

    def set(self, **kw):
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError(k)
            setattr(self, k, v)
        return self


class SvgfPlanarGBuffer(C.Structure):
    """Device pointers of the context's current-frame planes (svgf_planar_gbuffer): packed float3 normal / position / albedo,
    int geomId."""
    _fields_ = [("normal", C.c_void_p), ("position", C.c_void_p), ("geom_id", C.c_void_p), ("albedo", C.c_void_p)]


class SvgfSynthParams(C.Structure):
    _fields_ = [("frame", C.c_int), ("seed", C.c_int), ("noise", C.c_float), ("fireflies", C.c_float),
                ("pixel_length", C.c_float * 2)]


def reference_defaults() -> SvgfParams:
    """ui_* defaults of reference src/main.cpp:49-62 (pure Python copy; the library's svgf_params_default is
    checked against this in tests)."""
    p = SvgfParams()
    p.set(temporal_enable=0, spatial_enable=0, color_alpha=0.2, moment_alpha=0.2, blur_variance=1, sigma_l=0.45,
          sigma_x=0.35, sigma_n=0.2, atrous_nlevel=5, history_level=1, sepcolor=0, addcolor=0, right_view_option=0)
    return p


class SvgfError(RuntimeError):
    pass


_lib = None
_lib_exp = None
_default_experiments = False


def use_experiments_library(on: bool = True):
    """Tests / tools of parked experiments: make libsvgf_hip_exp.so the library every following Denoiser / producer call of this
    process uses (an explicit call — the binding reads no environment variable; the product path never calls this)."""
    global _default_experiments
    _default_experiments = bool(on)


def exp_set(name: str, value: int):
    """svgf_exp_set of the experiments build (process-wide; read by svgf_create / the launchers of libsvgf_hip_exp.so)."""
    lib = load_library(experiments=True)
    if lib.svgf_exp_set(name.encode(), int(value)) != SVGF_OK:
        raise SvgfError(f"svgf_exp_set({name!r}, {value}) failed")


def exp_clear():
    load_library(experiments=True).svgf_exp_clear()


def load_library(path: str | None = None, experiments: bool = False):
    """Load libsvgf_hip.so (or, experiments=True / after use_experiments_library(), libsvgf_hip_exp.so) and declare prototypes.
    Fails loudly when the library is absent.  No environment variable is read."""
    global _lib, _lib_exp
    if path is None and not experiments and _default_experiments:
        experiments = True
    if path is None and experiments:
        if _lib_exp is None:
            _lib_exp = load_library(LIB_EXP_PATH)
            _lib_exp.svgf_exp_set.argtypes = [C.c_char_p, C.c_int]
        return _lib_exp
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    # PyTorch-ROCm wheels bundle their own libamdhip64.so.  If this library were loaded first it would bind the
    # system copy, and a process in which two HIP runtimes initialise can end up with one of them seeing no device.
    # Importing torch first (when it is installed) makes both share the copy torch already mapped.  torch is not
    # otherwise needed by the library.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    if not os.path.exists(path):
        raise SvgfError(f"{path} not found: build it first (python -c 'import __graft_entry__ as g; g.build()'). "
                        "There is no CPU fallback.")
    lib = C.CDLL(path)
    vp, ip = C.c_void_p, C.c_int
    lib.svgf_version.restype = ip
    lib.svgf_params_default.argtypes = [C.POINTER(SvgfParams)]
    lib.svgf_create.argtypes = [ip, ip, ip, C.POINTER(vp)]
    lib.svgf_create_ex.argtypes = [ip, ip, ip, C.c_uint, C.POINTER(vp)]
    lib.svgf_enable_pipeline.argtypes = [vp]
    lib.svgf_pipeline_status.argtypes = [vp]
    lib.svgf_pipeline_status.restype = ip
    lib.svgf_streams_overlap.argtypes = [ip, vp, vp]
    lib.svgf_streams_overlap.restype = ip
    lib.svgf_destroy.argtypes = [vp]
    lib.svgf_reset.argtypes = [vp]
    lib.svgf_denoise.argtypes = [vp, vp, vp, vp, C.POINTER(SvgfCamera), C.POINTER(SvgfParams), vp]
    lib.svgf_denoise_host.argtypes = [vp, vp, vp, vp, C.POINTER(SvgfCamera), C.POINTER(SvgfParams)]
    lib.svgf_sync.argtypes = [vp]
    lib.svgf_sync_stream.argtypes = [vp, vp]
    lib.svgf_is_pipelined.argtypes = [vp]
    lib.svgf_is_pipelined.restype = ip
    lib.svgf_last_error.argtypes = [vp]
    lib.svgf_last_error.restype = C.c_char_p
    lib.svgf_width.argtypes = [vp]
    lib.svgf_height.argtypes = [vp]
    lib.svgf_read_state.argtypes = [vp, ip, vp, C.c_ulonglong]
    lib.svgf_set_capture.argtypes = [vp, ip]
    lib.svgf_profile_enable.argtypes = [vp, ip]
    lib.svgf_profile_stride.argtypes = [vp, ip]
    lib.svgf_profile_frames.argtypes = [vp]
    lib.svgf_profile_frames.restype = C.c_longlong
    lib.svgf_profile_read.argtypes = [vp, ip, ip, C.POINTER(ip), C.POINTER(C.c_float), C.POINTER(ip)]
    lib.svgf_synth_camera.argtypes = [ip, ip, ip, ip, C.POINTER(SvgfCamera), C.POINTER(C.c_float)]
    lib.svgf_synth_render.argtypes = [ip, vp, vp, ip, ip, C.POINTER(SvgfCamera), C.POINTER(SvgfSynthParams), vp]
    lib.svgf_scene_render.argtypes = [ip, vp, vp, ip, ip, C.POINTER(SvgfCamera), C.POINTER(SvgfSynthParams), vp, ip,
                                      C.POINTER(C.c_float), vp]
    lib.svgf_scene_render_mesh.argtypes = [ip, vp, vp, ip, ip, C.POINTER(SvgfCamera), C.POINTER(SvgfSynthParams), vp, ip,
                                           vp, vp, vp, vp, ip, vp, vp, vp, ip, C.POINTER(C.c_float), vp]
    lib.svgf_scene_render_mesh_planar.argtypes = [ip, vp, C.POINTER(SvgfPlanarGBuffer), ip, ip, C.POINTER(SvgfCamera), C.POINTER(SvgfSynthParams), vp, ip,
                                                  vp, vp, vp, vp, ip, vp, vp, vp, ip, C.POINTER(C.c_float), vp]
    lib.svgf_planar_gbuffer.argtypes = [vp, C.POINTER(SvgfPlanarGBuffer)]
    lib.svgf_planar_gbuffer_stream.argtypes = [vp, C.POINTER(SvgfPlanarGBuffer), vp]
    lib.svgf_denoise_planar.argtypes = [vp, vp, vp, C.POINTER(SvgfCamera), C.POINTER(SvgfParams), vp]
    lib.svgf_synth_render_planar.argtypes = [ip, vp, C.POINTER(SvgfPlanarGBuffer), ip, ip, C.POINTER(SvgfCamera), C.POINTER(SvgfSynthParams), vp]
    lib.svgf_params_sizeof.restype = ip
    if lib.svgf_params_sizeof() != C.sizeof(SvgfParams):     # the struct grows at its tail between ABI versions
        raise SvgfError(f"libsvgf_hip.so was built with sizeof(SvgfParams) = {lib.svgf_params_sizeof()}, this binding has {C.sizeof(SvgfParams)}")
    lib.svgf_display_pack.argtypes = [ip, vp, vp, vp, ip, ip, vp]
    lib.svgf_save_png.argtypes = [C.c_char_p, vp, ip, ip, ip]
    if path == LIB_PATH:
        _lib = lib
    return lib


def _ptr(x):
    """device pointer of a torch tensor / int / None"""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if hasattr(x, "data_ptr"):
        if not x.is_contiguous():
            raise SvgfError("tensor must be contiguous")
        return x.data_ptr()
    raise TypeError(type(x))


class Denoiser:
    """One SVGF context on one GPU (denoiseInit .. denoiseFree of the reference, handle-based)."""

    def __init__(self, width: int, height: int, device: int = 0, experiments: bool = False, pipelined: bool = False):
        """pipelined=True: svgf_create_ex(SVGF_CREATE_PIPELINED) — the frame pipeline's resources exist from the start
        (SvgfParams.inputs_ready is ignored by any other context).
        experiments=True: the context lives in libsvgf_hip_exp.so (kernel_variant 5 / 6, exp_set knobs) — tests and tools only."""
        self.lib = load_library(experiments=experiments)
        self.width, self.height = int(width), int(height)
        self.ui = SvgfParams()
        self.lib.svgf_params_default(C.byref(self.ui))
        h = C.c_void_p()
        rc = self.lib.svgf_create_ex(int(device), self.width, self.height, CREATE_PIPELINED if pipelined else 0, C.byref(h))
        if rc != SVGF_OK:
            raise SvgfError(f"svgf_create_ex({device},{width},{height}) -> {rc}: {self.lib.svgf_last_error(None).decode()}")
        self.h = h

    def _check(self, rc, what):
        if rc != SVGF_OK:
            raise SvgfError(f"{what} -> {rc}: {self.lib.svgf_last_error(self.h).decode()}")

    # --- reference-named lifecycle ---
    def free(self):
        if getattr(self, "h", None):
            self.lib.svgf_destroy(self.h)
            self.h = None

    denoiseFree = free

    def reset(self):
        self._check(self.lib.svgf_reset(self.h), "svgf_reset")

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def denoise(self, out, inp, gbuffer, camera, params: SvgfParams | None = None, stream=None):
        """Device pointers (ints) or contiguous CUDA torch tensors.  Asynchronous on `stream`."""
        cam = camera if isinstance(camera, SvgfCamera) else SvgfCamera.from_dict(camera)
        p = params if params is not None else self.ui
        s = None if stream is None else (stream if isinstance(stream, int) else stream.cuda_stream)
        self._check(self.lib.svgf_denoise(self.h, _ptr(out), _ptr(inp), _ptr(gbuffer), C.byref(cam), C.byref(p), s),
                    "svgf_denoise")

    def planar_gbuffer(self, stream=None) -> SvgfPlanarGBuffer:
        """The planes the NEXT denoise_planar() call consumes: a producer fills them in place.
        stream given: svgf_planar_gbuffer_stream — on a pipelined context the producer's stream waits for exactly the frames that
        still read those planes instead of the host waiting for the device."""
        g = SvgfPlanarGBuffer()
        if stream is None:
            self._check(self.lib.svgf_planar_gbuffer(self.h, C.byref(g)), "svgf_planar_gbuffer")
        else:
            s = stream if isinstance(stream, int) else stream.cuda_stream
            self._check(self.lib.svgf_planar_gbuffer_stream(self.h, C.byref(g), s), "svgf_planar_gbuffer_stream")
        return g

    def denoise_planar(self, out, inp, camera, params: SvgfParams | None = None, stream=None):
        """One frame whose G-buffer was written into planar_gbuffer()'s planes.  Asynchronous on `stream`."""
        cam = camera if isinstance(camera, SvgfCamera) else SvgfCamera.from_dict(camera)
        p = params if params is not None else self.ui
        s = None if stream is None else (stream if isinstance(stream, int) else stream.cuda_stream)
        self._check(self.lib.svgf_denoise_planar(self.h, _ptr(out), _ptr(inp), C.byref(cam), C.byref(p), s), "svgf_denoise_planar")

    def denoise_host(self, color: np.ndarray, gbuffer: np.ndarray, camera, params: SvgfParams | None = None) -> np.ndarray:
        """numpy in, numpy out (uploads, runs, downloads, synchronises)."""
        color = np.ascontiguousarray(color, dtype=np.float32)
        gbuffer = np.ascontiguousarray(gbuffer)
        assert color.size == 3 * self.width * self.height and gbuffer.nbytes == 52 * self.width * self.height
        out = np.empty_like(color)
        cam = camera if isinstance(camera, SvgfCamera) else SvgfCamera.from_dict(camera)
        p = params if params is not None else self.ui
        self._check(self.lib.svgf_denoise_host(self.h, out.ctypes.data, color.ctypes.data, gbuffer.ctypes.data,
                                               C.byref(cam), C.byref(p)), "svgf_denoise_host")
        return out

    def sync(self):
        self._check(self.lib.svgf_sync(self.h), "svgf_sync")

    def is_pipelined(self) -> bool:
        """True while the context's frames alternate between its two plane sets (include/svgf.h: svgf_is_pipelined)."""
        return bool(self.lib.svgf_is_pipelined(self.h))

    def enable_pipeline(self):
        """svgf_enable_pipeline: create the frame pipeline's resources now (allocates, synchronises the device, probes the queues)."""
        self._check(self.lib.svgf_enable_pipeline(self.h), "svgf_enable_pipeline")

    def pipeline_status(self) -> int:
        """0 no resources, 1 promise honoured, 2 promise refused (internal streams share a hardware queue)."""
        return int(self.lib.svgf_pipeline_status(self.h))

    def last_error(self) -> str:
        return self.lib.svgf_last_error(self.h).decode()

    def sync_stream(self, stream=None):
        """Wait for what has been enqueued on `stream` only (svgf_sync waits for the whole device, as the reference does)."""
        s = None if stream is None else (stream if isinstance(stream, int) else stream.cuda_stream)
        self._check(self.lib.svgf_sync_stream(self.h, s), "svgf_sync_stream")

    def set_capture(self, on: bool = True):
        self._check(self.lib.svgf_set_capture(self.h, 1 if on else 0), "svgf_set_capture")

    def read_state(self, which: int) -> np.ndarray:
        n = self.width * self.height
        shape, dt = {STATE_HISTORY_LENGTH: ((self.height, self.width), np.int32),
                     STATE_MOMENTS: ((self.height, self.width, 2), np.float32),
                     STATE_COLOR_HISTORY: ((self.height, self.width, 3), np.float32),
                     STATE_VARIANCE_TEMPORAL: ((self.height, self.width), np.float32),
                     STATE_COLOR_ACC: ((self.height, self.width, 3), np.float32)}[which]
        a = np.empty(shape, dtype=dt)
        assert a.size >= n
        self._check(self.lib.svgf_read_state(self.h, which, a.ctypes.data, a.nbytes), "svgf_read_state")
        return a

    # --- profiling ---
    def profile_enable(self, nframes: int):
        self._check(self.lib.svgf_profile_enable(self.h, int(nframes)), "svgf_profile_enable")

    def profile_stride(self, k: int):
        self._check(self.lib.svgf_profile_stride(self.h, int(k)), "svgf_profile_stride")

    def profile_frames(self) -> int:
        return int(self.lib.svgf_profile_frames(self.h))

    def profile_read(self, slot: int):
        kinds = (C.c_int * 32)()
        ms = (C.c_float * 32)()
        n = C.c_int()
        self._check(self.lib.svgf_profile_read(self.h, slot, 32, kinds, ms, C.byref(n)), "svgf_profile_read")
        return [(kinds[i], ms[i]) for i in range(n.value)]


# --- the denoiser's inputs produced on the device -------------------------------------------
def streams_overlap(stream_a, stream_b, device: int = 0) -> bool:
    """svgf_streams_overlap: do kernels on the caller's two streams run side by side (what inputs_ready = 2 needs)?"""
    lib = load_library()
    sa = stream_a if isinstance(stream_a, int) else stream_a.cuda_stream
    sb = stream_b if isinstance(stream_b, int) else stream_b.cuda_stream
    rc = lib.svgf_streams_overlap(int(device), sa, sb)
    if rc < 0:
        raise SvgfError(f"svgf_streams_overlap failed ({rc})")
    return bool(rc)


def synth_camera(frame: int, moving: bool, width: int, height: int):
    """svgf_synth_camera: (SvgfCamera, (plx, ply)) computed by the library (C float math)."""
    lib = load_library()
    cam = SvgfCamera()
    pl = (C.c_float * 2)()
    rc = lib.svgf_synth_camera(int(frame), int(bool(moving)), int(width), int(height), C.byref(cam), pl)
    if rc != SVGF_OK:
        raise SvgfError(f"svgf_synth_camera failed ({rc})")
    return cam, (pl[0], pl[1])


def synth_render(out_rgb, out_gbuffer, width: int, height: int, camera, frame: int, seed: int = 1,
                 noise: float = 0.6, fireflies: float = 0.02, pixel_length=None, device: int = 0, stream=None):
    """svgf_synth_render: writes packed rgb (H*W*3 float32) and 52-byte texels (H*W*52 bytes) into device memory
    (torch CUDA tensors or raw pointers).  `camera` is an SvgfCamera or a synth.camera_for_frame() dict;
    `pixel_length` defaults to synth._pixel_length(width, height, 45) so that host and device producers agree."""
    lib = load_library()
    cam = camera if isinstance(camera, SvgfCamera) else SvgfCamera.from_dict(camera)
    if pixel_length is None:
        from . import synth as _synth
        pixel_length = _synth._pixel_length(width, height, 45.0)
    sp = SvgfSynthParams(int(frame), int(seed), float(noise), float(fireflies))
    sp.pixel_length[0] = float(pixel_length[0])
    sp.pixel_length[1] = float(pixel_length[1])
    s = None if stream is None else (stream if isinstance(stream, int) else stream.cuda_stream)
    rc = lib.svgf_synth_render(int(device), _ptr(out_rgb), _ptr(out_gbuffer), int(width), int(height), C.byref(cam),
                               C.byref(sp), s)
    if rc != SVGF_OK:
        raise SvgfError(f"svgf_synth_render failed ({rc})")


def synth_render_planar(out_rgb, planes: SvgfPlanarGBuffer, width: int, height: int, camera, frame: int, seed: int = 1,
                        noise: float = 0.6, fireflies: float = 0.02, pixel_length=None, device: int = 0, stream=None):
    """svgf_synth_render_planar: the frame of synth_render written into a Denoiser's planar_gbuffer() planes."""
    lib = load_library()
    cam = camera if isinstance(camera, SvgfCamera) else SvgfCamera.from_dict(camera)
    if pixel_length is None:
        from . import synth as _synth
        pixel_length = _synth._pixel_length(width, height, 45.0)
    sp = SvgfSynthParams(int(frame), int(seed), float(noise), float(fireflies))
    sp.pixel_length[0] = float(pixel_length[0])
    sp.pixel_length[1] = float(pixel_length[1])
    s = None if stream is None else (stream if isinstance(stream, int) else stream.cuda_stream)
    rc = lib.svgf_synth_render_planar(int(device), _ptr(out_rgb), C.byref(planes), int(width), int(height), C.byref(cam), C.byref(sp), s)
    if rc != SVGF_OK:
        raise SvgfError(f"svgf_synth_render_planar failed ({rc})")


# --- the step after denoise() ----------------------------------------------------------------
def display_pack(pbo, left, right, width: int, height: int, device: int = 0, stream=None):
    """svgf_display_pack: `left` | `right` (packed rgb float, device) -> (height, 2*width, 4) uint8 in device memory."""
    lib = load_library()
    s = None if stream is None else (stream if isinstance(stream, int) else stream.cuda_stream)
    rc = lib.svgf_display_pack(int(device), _ptr(pbo), _ptr(left), _ptr(right), int(width), int(height), s)
    if rc != SVGF_OK:
        raise SvgfError(f"svgf_display_pack failed ({rc})")


def save_png(path: str, rgb: np.ndarray, mirror_x: bool = True):
    """svgf_save_png: host float32 (H, W, 3) -> 8-bit RGB PNG, with the reference's x mirror by default."""
    lib = load_library()
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    h, w = rgb.shape[0], rgb.shape[1]
    rc = lib.svgf_save_png(os.fsencode(path), rgb.ctypes.data, int(w), int(h), int(bool(mirror_x)))
    if rc != SVGF_OK:
        raise SvgfError(f"svgf_save_png failed ({rc})")


# --- scene-driven producer -------------------------------------------------------------------
def scene_render(out_rgb, out_gbuffer, width: int, height: int, camera, geoms: np.ndarray, frame: int, seed: int = 1,
                 noise: float = 0.6, fireflies: float = 0.02, pixel_length=None, light=None, device: int = 0, stream=None):
    """svgf_scene_render: `geoms` is a scene.SCENE_GEOM_DTYPE array (scene.geom_array(scene.parse_scene(text)))."""
    from . import scene as _scene
    from . import synth as _synth
    lib = load_library()
    cam = camera if isinstance(camera, SvgfCamera) else SvgfCamera.from_dict(camera)
    if pixel_length is None:
        fovy = camera.get("fovy_deg", 45.0) if isinstance(camera, dict) else 45.0
        pixel_length = _synth._pixel_length(width, height, fovy)
    sp = SvgfSynthParams(int(frame), int(seed), float(noise), float(fireflies))
    sp.pixel_length[0] = float(pixel_length[0])
    sp.pixel_length[1] = float(pixel_length[1])
    geoms = np.ascontiguousarray(geoms, dtype=_scene.SCENE_GEOM_DTYPE)
    lp = _scene.light_position(geoms) if light is None else np.asarray(light, dtype=np.float32)
    larr = (C.c_float * 3)(float(lp[0]), float(lp[1]), float(lp[2]))
    s = None if stream is None else (stream if isinstance(stream, int) else stream.cuda_stream)
    rc = lib.svgf_scene_render(int(device), _ptr(out_rgb), _ptr(out_gbuffer), int(width), int(height), C.byref(cam), C.byref(sp),
                               geoms.ctypes.data, int(len(geoms)), larr, s)
    if rc != SVGF_OK:
        raise SvgfError(f"svgf_scene_render failed ({rc})")


def scene_render_mesh(out_rgb, out_gbuffer, width: int, height: int, camera, geoms: np.ndarray, geom_ids, tris, tri_ids, tri_albedo,
                      frame: int, tri_tex=None, textures=None, seed: int = 1, noise: float = 0.6, fireflies: float = 0.02, pixel_length=None, light=None,
                      device: int = 0, stream=None):
    """svgf_scene_render_mesh: primitives (`geoms`, geomId = geom_ids[k]) + world-space triangles (`tris` float32[n,3,8] =
    pos, normal, uv per corner; geomId = tri_ids[i]; albedo = tri_albedo[i]).  The library has read the host arrays completely
    when it returns (it waits for its uploads); the kernel itself stays asynchronous on `stream`.
    `out_gbuffer` may be a SvgfPlanarGBuffer (Denoiser.planar_gbuffer()): svgf_scene_render_mesh_planar writes the planes."""
    from . import scene as _scene
    from . import synth as _synth
    lib = load_library()
    cam = camera if isinstance(camera, SvgfCamera) else SvgfCamera.from_dict(camera)
    if pixel_length is None:
        fovy = camera.get("fovy_deg", 45.0) if isinstance(camera, dict) else 45.0
        pixel_length = _synth._pixel_length(width, height, fovy)
    sp = SvgfSynthParams(int(frame), int(seed), float(noise), float(fireflies))
    sp.pixel_length[0] = float(pixel_length[0])
    sp.pixel_length[1] = float(pixel_length[1])
    geoms = np.ascontiguousarray(geoms, dtype=_scene.SCENE_GEOM_DTYPE)
    gi = np.ascontiguousarray(geom_ids, dtype=np.int32)
    tr = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1, 24)
    ti = np.ascontiguousarray(tri_ids, dtype=np.int32)
    ta = np.ascontiguousarray(tri_albedo, dtype=np.float32).reshape(-1, 3)
    # textures: list of uint8[h, w, 3]; tri_tex: index into that list per triangle, -1 = none
    n_tex, tt, td, tx = 0, None, None, None
    if textures is not None and len(textures) and tri_tex is not None:
        n_tex = len(textures)
        tt = np.ascontiguousarray(tri_tex, dtype=np.int32)
        offs, blobs, o = [], [], 0
        for t in textures:
            t = np.ascontiguousarray(t, dtype=np.uint8)
            offs += [o, t.shape[1], t.shape[0]]
            blobs.append(t.reshape(-1))
            o += t.size
        td = np.array(offs, dtype=np.int32)
        tx = np.concatenate(blobs)
    lp = _scene.light_position(geoms) if light is None else np.asarray(light, dtype=np.float32)
    larr = (C.c_float * 3)(float(lp[0]), float(lp[1]), float(lp[2]))
    s = None if stream is None else (stream if isinstance(stream, int) else stream.cuda_stream)
    planar = isinstance(out_gbuffer, SvgfPlanarGBuffer)
    fn = lib.svgf_scene_render_mesh_planar if planar else lib.svgf_scene_render_mesh
    rc = fn(int(device), _ptr(out_rgb), C.byref(out_gbuffer) if planar else _ptr(out_gbuffer), int(width), int(height), C.byref(cam), C.byref(sp),
            geoms.ctypes.data if len(geoms) else None, int(len(geoms)), gi.ctypes.data if len(gi) else None,
            tr.ctypes.data if len(tr) else None, ti.ctypes.data if len(ti) else None, ta.ctypes.data if len(ta) else None, int(len(tr)),
            tt.ctypes.data if n_tex else None, td.ctypes.data if n_tex else None, tx.ctypes.data if n_tex else None, int(n_tex), larr, s)
    if rc != SVGF_OK:
        raise SvgfError(f"svgf_scene_render_mesh failed ({rc})")
    if stream is None:
        import torch
        torch.cuda.synchronize(int(device))
