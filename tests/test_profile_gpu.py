"""svgf_profile_*: per-kernel durations from event pairs attached to the dispatches (hipExtLaunchKernelGGL).

What must hold: every launch of a profiled frame has an entry of the right kind; the durations are the kernels' own (their
sum fits inside the frame's wall time, a 1080p a-trous level is tens of microseconds, not milliseconds and not zero); re-arming
with the same slot count restarts the counters without losing the ability to time; the stride times every k-th frame only; and
an un-profiled context is untouched by another thread's armed events (the hand-off is thread-local)."""
import threading

import numpy as np
import pytest
from conftest import relerr

pytestmark = pytest.mark.gpu


def _frames(pkg, W, H, n):
    return [pkg.synth.render_frame(W, H, f, seed=3, moving=False) for f in range(n)]


def test_every_launch_is_timed_and_the_durations_are_the_kernels_own(pkg):
    import torch
    W, H = 1920, 1080
    d = pkg.Denoiser(W, H, 0)
    p = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, atrous_nlevel=5, history_level=1)
    c, g, cam = _frames(pkg, W, H, 1)[0]
    dc = torch.from_numpy(c).cuda(); dg = torch.from_numpy(g.view(np.uint8).reshape(-1)).cuda()
    out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    cam_s = pkg.SvgfCamera.from_dict(cam)
    for _ in range(8):
        d.denoise(out, dc, dg, cam_s, p)
    d.profile_stride(1)
    d.profile_enable(6)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(6):
        d.denoise(out, dc, dg, cam_s, p)
    e1.record(); torch.cuda.synchronize()
    wall_ms = e0.elapsed_time(e1)
    assert d.profile_frames() == 6
    total = 0.0
    for s in range(6):
        rows = d.profile_read(s)
        assert [k for k, _ in rows] == [pkg.binding.KERNEL_TEMPORAL] + [pkg.binding.KERNEL_ATROUS] * 5, rows
        for k, ms in rows:
            assert 0.005 < ms < 0.5, f"slot {s}: kind {k} lasted {ms} ms"      # a 1080p pass: tens of microseconds
        total += sum(ms for _, ms in rows)
    assert total <= wall_ms * 1.02, f"kernel durations {total:.3f} ms exceed the wall time of their frames {wall_ms:.3f} ms"
    assert total >= wall_ms * 0.5, f"kernel durations {total:.3f} ms are far below the wall time {wall_ms:.3f} ms: not the kernels' durations"
    # re-arm with the same slot count: counters restart, timing still works
    d.profile_enable(6)
    assert d.profile_frames() == 0
    d.denoise(out, dc, dg, cam_s, p)
    torch.cuda.synchronize()
    assert d.profile_frames() == 1 and len(d.profile_read(0)) == 6
    # stride: every 3rd frame only
    d.profile_stride(3)
    d.profile_enable(4)
    for _ in range(7):
        d.denoise(out, dc, dg, cam_s, p)
    torch.cuda.synchronize()
    assert d.profile_frames() == 3          # frames 0, 3, 6
    d.profile_stride(1)
    d.profile_enable(0)
    d.denoise(out, dc, dg, cam_s, p)
    torch.cuda.synchronize()
    d.free()


def test_two_threads_one_profiled_one_not(pkg, orc):
    """The armed event pair is thread-local: a context driven from another host thread without profiling keeps its results and
    the profiled context keeps one entry per launch."""
    W, H = 320, 180
    frames = _frames(pkg, W, H, 4)
    p = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1)
    ref = []
    o = orc.Oracle(pkg, W, H, threads=4)
    for c, g, cam in frames:
        ref.append(o.denoise(c, g, cam, p))
    o.free()
    res = {}

    def run(tag, profiled):
        d = pkg.Denoiser(W, H, 0)
        if profiled:
            d.profile_enable(len(frames))
        outs = [d.denoise_host(c, g, cam, p) for c, g, cam in frames]
        n = [len(d.profile_read(s)) for s in range(d.profile_frames())] if profiled else []
        d.free()
        res[tag] = (outs, n)
    ts = [threading.Thread(target=run, args=("a", True)), threading.Thread(target=run, args=("b", False))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for tag in ("a", "b"):
        for f in range(len(frames)):
            assert relerr(res[tag][0][f], ref[f]).max() <= 1e-5, f"thread {tag} frame {f}"
    assert res["a"][1] == [6] * len(frames), res["a"][1]
