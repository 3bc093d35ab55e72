"""The frame pipeline (SvgfParams::inputs_ready, ABI 0.8; explicit resources since 0.9): consecutive frames of ONE sequence on two internal streams.

A frame's temporal pass needs of the previous frame only what exists once the level that feeds the colour history has run (level 1 with
the reference's defaults, src/denoise.cu:391); levels 2-5 of frame n and the temporal pass + level 1 of frame n+1 are independent.  A
caller that promises `inputs_ready` (inputs complete at call time) lets the library run even and odd frames on two
internal streams with two plane sets; the kernels are the same, the data are the same, so every result must be BIT-IDENTICAL to the
same frames ordered on one stream — for every position of the history level, across mode switches, debug views, non-temporal frames,
resets, on odd sizes, under a stream capture, on two caller streams used in turn without any promise (inputs_ready = 2), and with
readers of a reused output buffer enqueued behind earlier calls.
(tests/test_parity_gpu.py::test_back_to_back_asynchronous_frames_equal_synchronised_frames covers history levels 0 / 1 / 3 / 5 at
1920x1080 with the state read-back.)"""
import ctypes
import threading

import numpy as np
import pytest
from conftest import relerr

pytestmark = pytest.mark.gpu


def _inputs(pkg, W, H, n, seed, moving=True):
    import torch
    fr = [pkg.synth.render_frame(W, H, f, seed=seed, moving=moving) for f in range(n)]
    return fr, [torch.from_numpy(f[0]).cuda() for f in fr], [torch.from_numpy(f[1].view(np.uint8).reshape(-1)).cuda() for f in fr]


_CREATE_LOCK = threading.Lock()


def _run(pkg, W, H, fr, tin, tg, plist, stream=None, reset_at=()):
    """One context, one frame per entry of plist (SvgfParams), every frame its own output buffer."""
    import torch
    with _CREATE_LOCK:      # (the queue probe of svgf_create_ex times two kernels: not beside another thread's frames — test_two_pipelined_contexts...)
        torch.cuda.synchronize()
        d = pkg.Denoiser(W, H, 0, pipelined=any(p.inputs_ready for p in plist))      # (only a context created pipelined looks at inputs_ready)
    assert not any(p.inputs_ready for p in plist) or d.pipeline_status() == 1, d.last_error()
    outs = [torch.empty((H, W, 3), dtype=torch.float32, device="cuda") for _ in plist]
    s = stream or torch.cuda.current_stream()
    for k, p in enumerate(plist):
        if k in reset_at:
            d.reset()
        d.denoise(outs[k], tin[k % len(tin)], tg[k % len(tg)], fr[k % len(fr)][2], p, stream=s)
    d.sync()
    res = [o.cpu().numpy() for o in outs], d.read_state(0), d.read_state(1), d.read_state(2)
    d.free()
    return res


@pytest.mark.parametrize("size", [(640, 360), (257, 131), (33, 7), (1, 1), (961, 90)])
def test_pipelined_frames_are_bit_identical_to_ordered_frames(pkg, size):
    W, H = size
    fr, tin, tg = _inputs(pkg, W, H, 5, seed=11)
    base = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, atrous_nlevel=5, history_level=1)
    want = _run(pkg, W, H, fr, tin, tg, [base] * 10)
    got = _run(pkg, W, H, fr, tin, tg, [pkg.SvgfParams.from_buffer_copy(base).set(inputs_ready=1)] * 10)
    for k in range(10):
        assert np.array_equal(want[0][k], got[0][k]), f"{W}x{H} frame {k}"
    for a, b in zip(want[1:], got[1:]):
        assert np.array_equal(a, b)


def test_mode_switches_debug_views_non_temporal_frames_and_a_reset(pkg):
    """The promise comes and goes from frame to frame; some frames are debug views (which read the history lengths the next temporal
    pass rewrites), some have no cascade, some no temporal pass, the history level moves, and the context is reset in the middle."""
    W, H = 480, 270
    fr, tin, tg = _inputs(pkg, W, H, 6, seed=5)
    ref = pkg.reference_defaults()
    mk = lambda **kw: pkg.SvgfParams.from_buffer_copy(ref).set(**kw)      # noqa: E731
    seq = [dict(temporal_enable=1, spatial_enable=1, history_level=1), dict(temporal_enable=1, spatial_enable=1, history_level=1),
           dict(temporal_enable=1, spatial_enable=1, history_level=1, right_view_option=1),      # history-length view
           dict(temporal_enable=1, spatial_enable=1, history_level=3), dict(temporal_enable=1, spatial_enable=1, history_level=0),
           dict(temporal_enable=1, spatial_enable=0), dict(temporal_enable=0, spatial_enable=1, atrous_nlevel=2),
           dict(temporal_enable=1, spatial_enable=1, history_level=5), dict(temporal_enable=1, spatial_enable=1, history_level=1, right_view_option=2),
           dict(temporal_enable=1, spatial_enable=1, history_level=2, atrous_nlevel=3), dict(temporal_enable=1, spatial_enable=1, history_level=1),
           dict(temporal_enable=1, spatial_enable=1, history_level=1), dict(temporal_enable=1, spatial_enable=1, history_level=6),
           dict(temporal_enable=1, spatial_enable=1, history_level=1)]
    promise = [1, 1, 1, 0, 1, 1, 1, 0, 0, 1, 1, 1, 1, 1]
    want = _run(pkg, W, H, fr, tin, tg, [mk(**kw) for kw in seq], reset_at=(9,))
    got = _run(pkg, W, H, fr, tin, tg, [mk(inputs_ready=pr, **kw) for kw, pr in zip(seq, promise)], reset_at=(9,))
    for k in range(len(seq)):
        assert np.array_equal(want[0][k], got[0][k]), f"frame {k} ({seq[k]}, promise {promise[k]})"
    for a, b in zip(want[1:], got[1:]):
        assert np.array_equal(a, b)


def test_pipelined_1080p_sequence_matches_the_oracle(pkg, orc):
    W, H = 1920, 1080
    p = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, atrous_nlevel=5, history_level=1, inputs_ready=1)
    fr, tin, tg = _inputs(pkg, W, H, 4, seed=71)
    got = _run(pkg, W, H, fr, tin, tg, [p] * 4)[0]
    o = orc.Oracle(pkg, W, H, threads=16)
    worst = 0.0
    for k in range(4):
        ref = o.denoise(fr[k][0], fr[k][1], fr[k][2], p)
        e = relerr(got[k], ref)
        assert e.max() <= 1e-4, f"frame {k}: {e.max():.3e}"
        worst = max(worst, float(e.max()))
    o.free()
    print(f"pipelined 1920x1080, 4 moving frames vs oracle: worst max-rel {worst:.2e}")


def test_a_pipelined_context_joins_a_stream_capture(pkg):
    """Under hipStreamBeginCapture nothing can be promised: the frames of a pipelined context are recorded on the capturing stream
    itself, so the whole frame is part of the graph.  Two frames per graph (plane
    parities), replayed three times, against the same eight frames run eagerly.  Then PROMISED eager frames again, enqueued right
    behind the last replay without any host synchronisation: a context that has been captured orders them behind the caller's stream
    (the replay uses the same planes and is visible only there), so the sequence goes on bit-identically."""
    import torch
    for ln in open("/proc/self/maps"):
        if "libamdhip64" in ln:
            hip = ctypes.CDLL(ln.split()[-1])
            break
    W, H = 640, 360
    fr, tin, tg = _inputs(pkg, W, H, 2, seed=9, moving=False)
    p = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, atrous_nlevel=5, history_level=1, inputs_ready=1)
    want = _run(pkg, W, H, fr, tin, tg, [p] * 12)[0]
    d = pkg.Denoiser(W, H, 0, pipelined=True)
    s = torch.cuda.Stream()
    outs = [torch.empty((H, W, 3), dtype=torch.float32, device="cuda") for _ in range(2)]
    tail = [torch.empty((H, W, 3), dtype=torch.float32, device="cuda") for _ in range(4)]
    got = []
    for k in range(2):          # eager, promised: on the context's internal streams
        d.denoise(outs[k], tin[k], tg[k], fr[k][2], p, stream=s)
    s.synchronize()
    got += [o.cpu().numpy() for o in outs]
    graph, gexec = ctypes.c_void_p(), ctypes.c_void_p()
    assert hip.hipStreamBeginCapture(ctypes.c_void_p(s.cuda_stream), 0) == 0
    for k in range(2):
        d.denoise(outs[k], tin[k], tg[k], fr[k][2], p, stream=s)
    assert hip.hipStreamEndCapture(ctypes.c_void_p(s.cuda_stream), ctypes.byref(graph)) == 0
    assert hip.hipGraphInstantiate(ctypes.byref(gexec), graph, None, None, 0) == 0
    for rep in range(3):
        assert hip.hipGraphLaunch(gexec, ctypes.c_void_p(s.cuda_stream)) == 0
        if rep < 2:
            s.synchronize()
            got += [o.cpu().numpy() for o in outs]
    # behind the LAST replay, not waited for: copies of its two outputs, then four promised eager frames
    with torch.cuda.stream(s):
        last = [o.clone() for o in outs]
    for k in range(4):
        d.denoise(tail[k], tin[k & 1], tg[k & 1], fr[k & 1][2], p, stream=s)
    s.synchronize()
    got += [o.cpu().numpy() for o in last] + [o.cpu().numpy() for o in tail]
    hip.hipGraphExecDestroy(gexec); hip.hipGraphDestroy(graph)
    d.free()
    for k in range(12):
        assert np.array_equal(want[k], got[k]), f"frame {k}"


def test_two_pipelined_contexts_from_two_host_threads(pkg):
    W, H, N = 800, 450, 12
    fr, tin, tg = _inputs(pkg, W, H, 4, seed=21)
    p0 = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, atrous_nlevel=5, history_level=1)
    want = _run(pkg, W, H, fr, tin, tg, [p0] * N)[0]
    res = {}

    def work(tag):
        import torch
        p = pkg.SvgfParams.from_buffer_copy(p0).set(inputs_ready=1)
        res[tag] = _run(pkg, W, H, fr, tin, tg, [p] * N, stream=torch.cuda.Stream())[0]
    ts = [threading.Thread(target=work, args=(t,)) for t in ("a", "b")]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for tag in ("a", "b"):
        for k in range(N):
            assert np.array_equal(want[k], res[tag][k]), f"thread {tag} frame {k}"


def test_profile_entries_of_pipelined_frames(pkg):
    W, H = 1920, 1080
    fr, tin, tg = _inputs(pkg, W, H, 2, seed=3, moving=False)
    import torch
    p = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, atrous_nlevel=5, history_level=1, inputs_ready=1)
    d = pkg.Denoiser(W, H, 0, pipelined=True)
    d.profile_enable(6)
    outs = [torch.empty((H, W, 3), dtype=torch.float32, device="cuda") for _ in range(2)]
    for k in range(6):
        d.denoise(outs[k & 1], tin[k & 1], tg[k & 1], fr[k & 1][2], p)
    d.sync()
    for s in range(6):
        ent = d.profile_read(s)
        assert [k for k, _ in ent] == [pkg.binding.KERNEL_TEMPORAL] + [pkg.binding.KERNEL_ATROUS] * 5, ent
        assert all(0.005 < ms < 1.0 for _, ms in ent), ent
    d.free()


def test_cpp_pipeline_example_is_bit_identical_to_frames_in_turn(pkg):
    """examples/pipeline.cpp: a renderer's frame loop in C++ through the C ABI — the producer of frame n+1 on its own stream beside
    frame n's denoising, the inputs_ready promise kept by double buffering and events — against the reference's order (render,
    denoise, render, denoise on one stream).  The program itself compares the last two outputs bit for bit."""
    import os
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "examples", "pipeline")
    assert os.path.exists(exe), "examples/pipeline is built by __graft_entry__.build()"
    for args in (["24", "640", "360"], ["9", "257", "131"]):
        r = subprocess.run([exe] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
        assert r.returncode == 0 and "bit-identical: yes" in r.stdout and "[context pipelined]" in r.stdout, r.stdout


def test_cpp_cadence_example_runs(pkg):
    """examples/cadence.cpp: a fixed-rate frame loop (producer + denoise + pack per tick, GPU idle in between) with the shader-clock and
    memory probes of DESIGN.md 6.2, through the C ABI."""
    import os
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "examples", "cadence")
    assert os.path.exists(exe), "examples/cadence is built by __graft_entry__.build()"
    r = subprocess.run([exe, "240", "24", "640", "360"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.count("ms per svgf_denoise") == 2 and "back to back" in r.stdout, r.stdout


def test_two_alternating_streams_without_the_promise(pkg):
    """inputs_ready = 2: the pipeline, every frame ordered behind the stream it was given to.  Even frames on one stream, odd frames on
    another, the inputs of each frame COPIED into its buffers on that stream right before the call (so they are not complete at call
    time) and the output read back by a copy enqueued behind the call on the same stream: bit-identical to ordered frames."""
    import torch
    W, H, N = 640, 360, 10
    fr, tin, tg = _inputs(pkg, W, H, 5, seed=31)
    p0 = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, atrous_nlevel=5, history_level=1)
    want = _run(pkg, W, H, fr, tin, tg, [p0] * N)[0]
    p = pkg.SvgfParams.from_buffer_copy(p0).set(inputs_ready=2)
    d = pkg.Denoiser(W, H, 0, pipelined=True)
    st = [torch.cuda.Stream(), torch.cuda.Stream()]
    cin = [torch.empty_like(tin[0]) for _ in range(2)]
    cg = [torch.empty_like(tg[0]) for _ in range(2)]
    out = [torch.empty((H, W, 3), dtype=torch.float32, device="cuda") for _ in range(2)]
    keep = [torch.empty((H, W, 3), dtype=torch.float32, device="cuda") for _ in range(N)]
    torch.cuda.synchronize()
    for k in range(N):
        q = k & 1
        with torch.cuda.stream(st[q]):
            cin[q].copy_(tin[k % 5], non_blocking=True)
            cg[q].copy_(tg[k % 5], non_blocking=True)
            d.denoise(out[q], cin[q], cg[q], fr[k % 5][2], p, stream=st[q])
            keep[k].copy_(out[q], non_blocking=True)
    torch.cuda.synchronize()
    assert d.is_pipelined()
    got = [o.cpu().numpy() for o in keep]
    d.free()
    for k in range(N):
        assert np.array_equal(want[k], got[k]), f"frame {k}"


def test_readers_of_the_output_enqueued_behind_earlier_calls_are_safe(pkg):
    """The promise is about the inputs.  ONE output buffer for every frame, and behind every call a copy of it enqueued on the caller's
    stream (never waited for by the host): the kernel of frame n+1 that writes the buffer must wait for that copy — it does, through
    the caller's stream position recorded at hand-over — or the copies would hold torn frames.  (The first version of the promise
    covered the output buffer as well; a soak loop that reused four buffers with enqueued readers broke it on 8 % of its frames.)"""
    import torch
    W, H, N = 640, 360, 40
    fr, tin, tg = _inputs(pkg, W, H, 5, seed=13)
    p0 = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, atrous_nlevel=5, history_level=1)
    want = _run(pkg, W, H, fr, tin, tg, [p0] * N)[0]
    p = pkg.SvgfParams.from_buffer_copy(p0).set(inputs_ready=1)
    d = pkg.Denoiser(W, H, 0, pipelined=True)
    s = torch.cuda.Stream()
    out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    keep = [torch.empty((H, W, 3), dtype=torch.float32, device="cuda") for _ in range(N)]
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        for k in range(N):
            d.denoise(out, tin[k % 5], tg[k % 5], fr[k % 5][2], p, stream=s)
            keep[k].copy_(out, non_blocking=True)
            for _ in range(3):          # keep the caller's stream busy for a while behind the copy
                keep[k].mul_(1.0)
    torch.cuda.synchronize()
    assert d.is_pipelined()
    got = [o.cpu().numpy() for o in keep]
    d.free()
    for k in range(N):
        assert np.array_equal(want[k], got[k]), f"frame {k}"


def test_promise_is_refused_when_the_internal_streams_share_a_hardware_queue(pkg):
    """GPU_MAX_HW_QUEUES=1 (read by the HIP runtime at start-up, hence a subprocess): the probe of svgf_create_ex finds the two internal
    streams serialised, svgf_pipeline_status says 2, svgf_last_error explains, and promised frames then run as ordinary ordered frames
    — so they are not slower than ordered frames (on one queue the pipelined form used to lose 8-10 %), and bit-identical."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    code = r"""
import sys, time, json
sys.path.insert(0, %r)
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
W, H = 1920, 1080
fr = [pkg.synth.render_frame(W, H, f, seed=3, moving=False) for f in range(2)]
tin = [torch.from_numpy(f[0]).cuda() for f in fr]; tg = [torch.from_numpy(f[1].view(np.uint8).reshape(-1)).cuda() for f in fr]
outs = [torch.empty((H, W, 3), dtype=torch.float32, device='cuda') for _ in range(2)]
base = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, atrous_nlevel=5, history_level=1)
res = {}
for name, piped in (('ordered', False), ('promised', True)):
    d = pkg.Denoiser(W, H, 0, pipelined=piped)
    p = pkg.SvgfParams.from_buffer_copy(base).set(inputs_ready=1 if piped else 0)
    res[name + '_status'] = d.pipeline_status(); res[name + '_err'] = d.last_error()
    def burst(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(n): d.denoise(outs[k & 1], tin[k & 1], tg[k & 1], fr[k & 1][2], p)
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    burst(600)
    res[name] = float(np.median([burst(40) for _ in range(5)]))
    res[name + '_sum'] = float(outs[1].double().sum().item()); res[name + '_piped'] = d.is_pipelined()
    d.free()
print('RESULT ' + json.dumps(res))
""" % ROOT
    env = dict(os.environ, GPU_MAX_HW_QUEUES="1")
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    import json
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    print(res)
    assert res["ordered_status"] == 0 and res["promised_status"] == 2 and "hardware queue" in res["promised_err"] and not res["promised_piped"]
    assert res["promised"] <= 1.03 * res["ordered"], res      # (both are ordered frames now: 0.2585 against 0.2579 measured)
    assert res["promised_sum"] == res["ordered_sum"]


def test_the_library_finds_an_overlapping_stream_pair_under_the_default_queue_setting(pkg):
    """No GPU_MAX_HW_QUEUES in the environment (the HIP runtime's default: 4 hardware queues): two streams created one after the other
    can land on one queue, so svgf_create_ex replaces its second internal stream until the pair overlaps.  Eight contexts in a fresh
    process, some created behind other streams: every one ends with the promise honoured."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    code = ("import sys; sys.path.insert(0, %r); import torch, __graft_entry__ as ge; pkg = ge.load_package(); st = []\n"
            "out = []\n"
            "for k in range(8):\n"
            "    d = pkg.Denoiser(320, 180, 0, pipelined=True); out.append(d.pipeline_status()); st.append(torch.cuda.Stream()); d.free()\n"
            "print('STATUS', out)" % ROOT)
    env = dict(os.environ)
    env.pop("GPU_MAX_HW_QUEUES", None)
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "STATUS [1, 1, 1, 1, 1, 1, 1, 1]" in r.stdout, r.stdout + r.stderr[-1500:]
