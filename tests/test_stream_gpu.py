"""Stream-scoped completion and graph capture through the C ABI (include/svgf.h: svgf_sync_stream, ABI 0.7).

The reference's denoise() ends with cudaDeviceSynchronize (src/denoise.cu:401) and svgf_sync / the legacy shim keep that; a renderer
with several streams wants to wait for ITS frames only, and one that replays a fixed frame loop wants to capture it once:
  * svgf_sync_stream(ctx, stream) returns when the frames enqueued on `stream` are done, while another stream is still busy;
  * svgf_denoise neither synchronises, allocates nor touches another stream, so a pair of frames captured into a hipGraph
    (hipStreamBeginCapture .. 2 x svgf_denoise .. hipStreamEndCapture) replays bit-identically to the same frames run eagerly.
    Two frames per graph: the context's moment / history-length and G-buffer planes alternate with the frame parity, and a captured
    launch carries the plane roles of the call it was captured from.  The previous view matrix is a kernel argument, so a captured
    loop is valid for the camera sequence it was captured with (here: a static camera).
"""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _hip():
    """The HIP runtime already loaded into this process (torch's)."""
    for ln in open("/proc/self/maps"):
        if "libamdhip64" in ln:
            return ctypes.CDLL(ln.split()[-1])
    raise RuntimeError("no HIP runtime loaded")


def _frames(pkg, W, H, n, seed):
    import torch
    fr = [pkg.synth.render_frame(W, H, f, seed=seed, moving=False) for f in range(n)]
    return ([torch.from_numpy(f[0]).cuda() for f in fr], [torch.from_numpy(f[1].view(np.uint8).reshape(-1)).cuda() for f in fr],
            pkg.SvgfCamera.from_dict(fr[0][2]))


def test_sync_stream_waits_for_its_stream_only(pkg):
    import torch
    W, H, N = 640, 360, 6
    cols, gbs, cam = _frames(pkg, W, H, N, seed=3)
    p = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1)
    # reference result: default stream, device-wide sync (the reference's behaviour)
    d = pkg.Denoiser(W, H, 0)
    want = []
    for f in range(N):
        o = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
        d.denoise(o, cols[f], gbs[f], cam, p)
        d.sync()
        want.append(o.cpu().numpy())
    d.free()
    # the same frames on a side stream while ANOTHER stream is kept busy for much longer than the frames take
    d = pkg.Denoiser(W, H, 0)
    s, busy = torch.cuda.Stream(), torch.cuda.Stream()
    big = torch.empty((1 << 28,), dtype=torch.float32, device="cuda")       # 1 GiB
    done = torch.cuda.Event()
    with torch.cuda.stream(busy):
        for _ in range(200):
            big.add_(1.0)            # ~0.4 ms each at HBM speed: tens of milliseconds in total
        done.record()
    outs = [torch.empty((H, W, 3), dtype=torch.float32, device="cuda") for _ in range(N)]
    for f in range(N):
        d.denoise(outs[f], cols[f], gbs[f], cam, p, stream=s)
    d.sync_stream(s)
    still_busy = not done.query()
    for f in range(N):       # (the outputs are complete: a device-to-host copy on the legacy stream would wait for `busy` too,
        got = torch.empty_like(outs[f], device="cpu").pin_memory()      # so copy on `s`)
        with torch.cuda.stream(s):
            got.copy_(outs[f], non_blocking=True)
        d.sync_stream(s)
        assert np.array_equal(got.numpy(), want[f]), f"frame {f}"
    assert still_busy, "svgf_sync_stream returned only after the other stream had drained: it must not be a device-wide wait"
    torch.cuda.synchronize()
    d.free()
    assert pkg.load_library().svgf_sync_stream(None, None) == -1        # SVGF_ERR_INVALID_ARG


def test_a_captured_pair_of_frames_replays_bit_identically(pkg):
    import torch
    hip = _hip()
    W, H, ROUNDS = 800, 450, 5
    cols, gbs, cam = _frames(pkg, W, H, 2 * ROUNDS, seed=9)
    p = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1)
    # eager
    d = pkg.Denoiser(W, H, 0)
    want = []
    for f in range(2 * ROUNDS):
        o = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
        d.denoise(o, cols[f], gbs[f], cam, p)
        want.append(o.cpu().numpy())
    d.free()
    # captured: fixed input / output buffers (their addresses are baked into the graph), refreshed before every replay
    d = pkg.Denoiser(W, H, 0)
    s = torch.cuda.Stream()
    cin = [torch.empty_like(cols[0]) for _ in range(2)]
    gin = [torch.empty_like(gbs[0]) for _ in range(2)]
    out = [torch.empty((H, W, 3), dtype=torch.float32, device="cuda") for _ in range(2)]
    got = []
    # round 0 runs eagerly on the stream (first launches set per-kernel attributes, which a capture must not contain)
    with torch.cuda.stream(s):
        for k in range(2):
            cin[k].copy_(cols[k]); gin[k].copy_(gbs[k])
    for k in range(2):
        d.denoise(out[k], cin[k], gin[k], cam, p, stream=s)
    d.sync_stream(s)
    got += [out[0].cpu().numpy(), out[1].cpu().numpy()]
    graph, gexec = ctypes.c_void_p(), ctypes.c_void_p()
    hip.hipStreamBeginCapture.argtypes = [ctypes.c_void_p, ctypes.c_int]
    hip.hipStreamEndCapture.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
    hip.hipGraphInstantiate.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    hip.hipGraphLaunch.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    hip.hipGraphExecDestroy.argtypes = [ctypes.c_void_p]
    hip.hipGraphDestroy.argtypes = [ctypes.c_void_p]
    assert hip.hipStreamBeginCapture(s.cuda_stream, 2) == 0           # hipStreamCaptureModeRelaxed
    for k in range(2):
        d.denoise(out[k], cin[k], gin[k], cam, p, stream=s)           # recorded, not executed
    assert hip.hipStreamEndCapture(s.cuda_stream, ctypes.byref(graph)) == 0 and graph.value
    assert hip.hipGraphInstantiate(ctypes.byref(gexec), graph, None, None, 0) == 0
    # the two captured calls advanced the context's HOST state by one full parity cycle: replays continue where round 0 ended
    for r in range(1, ROUNDS):
        with torch.cuda.stream(s):
            for k in range(2):
                cin[k].copy_(cols[2 * r + k]); gin[k].copy_(gbs[2 * r + k])
        assert hip.hipGraphLaunch(gexec, s.cuda_stream) == 0
        d.sync_stream(s)
        got += [out[0].cpu().numpy(), out[1].cpu().numpy()]
    hip.hipGraphExecDestroy(gexec); hip.hipGraphDestroy(graph)
    d.free()
    for f in range(2 * ROUNDS):
        assert np.array_equal(got[f], want[f]), f"frame {f}: graph replay differs from the eager run"


def test_cpp_farm_example_eight_contexts_from_one_process(pkg):
    """examples/farm.cpp: eight contexts (threads, streams) driven from one C++ process through the C ABI, sharing this box's one
    GPU (context k on device k mod n_devices; on an 8-GPU node: one per GPU).  Every context succeeds, the sequences are
    independent (eight different checksums), and a context's result does not depend on how many others run beside it."""
    import os
    import re
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "examples", "farm")
    assert os.path.exists(exe), "examples/farm is built by __graft_entry__.build()"

    def run(n):
        r = subprocess.run([exe, str(n), "5", "640", "360"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
        assert r.returncode == 0, r.stdout
        cs = {int(m.group(1)): m.group(2) for m in re.finditer(r"context (\d+) device \d+ frames 5 ms_per_frame [0-9.]+ checksum ([-0-9.]+)", r.stdout)}
        assert len(cs) == n and "Mpixels/s aggregate" in r.stdout, r.stdout
        return cs

    eight, one = run(8), run(1)
    assert len(set(eight.values())) == 8, eight
    assert eight[0] == one[0]
    # every context pipelines its own sequence (two streams in turn); with the last argument 0 its frames are ordered on one stream: same result
    r = subprocess.run([exe, "2", "5", "640", "360", "0"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "ordered on one stream" in r.stdout, r.stdout
    m = re.search(r"context 0 device \d+ frames 5 ms_per_frame [0-9.]+ checksum ([-0-9.]+)", r.stdout)
    assert m and m.group(1) == one[0], r.stdout


def test_streams_overlap_probe_for_the_callers_streams(pkg):
    """svgf_streams_overlap: the library's hardware-queue probe for a CALLER's two streams (what inputs_ready = 2 runs on).  With
    GPU_MAX_HW_QUEUES=1 (a subprocess: the runtime reads it at start-up) any two streams serialise -> 0; the same stream twice -> 0; and
    examples/farm then orders each context's frames on one stream and says so."""
    import os
    import subprocess
    import sys
    import torch
    from conftest import ROOT
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    assert pkg.binding.streams_overlap(s0, s0) is False
    assert pkg.binding.streams_overlap(s0, s1) in (True, False)          # whatever the runtime's mapping is in this process: a clean answer
    code = ("import sys; sys.path.insert(0, %r); import torch, __graft_entry__ as ge; pkg = ge.load_package(); "
            "print('OVERLAP', int(pkg.binding.streams_overlap(torch.cuda.Stream(), torch.cuda.Stream())))" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=dict(os.environ, GPU_MAX_HW_QUEUES="1"))
    assert r.returncode == 0 and "OVERLAP 0" in r.stdout, r.stdout + r.stderr[-1000:]
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=dict(os.environ, GPU_MAX_HW_QUEUES="8"))
    assert r.returncode == 0 and "OVERLAP 1" in r.stdout, r.stdout + r.stderr[-1000:]
    r = subprocess.run([os.path.join(ROOT, "examples", "farm"), "1", "6", "640", "360", "1"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300,
                       env=dict(os.environ, GPU_MAX_HW_QUEUES="1"))
    assert r.returncode == 0 and "no two streams on different hardware queues" in r.stdout, r.stdout
