#!/usr/bin/env python3
"""bench.py — SVGF Mpixels/s (full pipeline) at 1080p on N GPUs of one node, plus the a-trous kernel's HBM roofline.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one svgf_denoise() call = one full SVGF pass (temporal accumulation + 5 edge-avoiding a-trous levels +
history rotation) over one 1920x1080 frame of the synthetic Cornell-like sequence (BASELINE.json configs[1]); inputs
are resident in HBM before the timed region starts.  N > 1: one process and one SVGF context per GPU, every rank
denoises its own independent sequence (weak scaling, no data-path collective); value = pixels of all ranks / max time.

Rank 0 prints ONE JSON line.  `value` is the pipelined throughput (K calls enqueued back to back, one synchronisation at
the end); `latency_ms_sync` is the metric as SURVEY.md §8(d)(i) words it — the median wall time of one svgf_denoise +
svgf_sync pair, what the reference's synchronous denoise() (src/denoise.cu:401) gives its caller.  `roofline` is for the
dominant kernel (the a-trous level): algorithmic bytes per launch (56 B/pixel) / mean launch duration from
HIP events attached to the kernel dispatches of the launch stream inside the timed region (hipExtLaunchKernelGGL: the kernel's
own begin / end timestamps; they agree with rocprofv3's durations of the same command to 2 %, profiles/).  `telemetry` holds shader clock / power / temperature
sampled while each of the three measurements ran (tools/telemetry.py).  `cpu_baseline` (N == 1 only) times the CPU oracle —
a port, not the product — on the host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as ge  # noqa: E402
import telemetry  # noqa: E402

W, H = 1920, 1080
NLEVEL = 5
ATROUS_BYTES_PER_PIXEL = 56.0      # R: colour 12 + normal 12 + position 12 + variance 4; W: colour 12 + variance 4
FRAME_BYTES_PER_PIXEL = 404.0      # temporal 124 + 5 x 56
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: 8 TB/s spec


def kernel_sources_sha16():
    """Fingerprint of the a-trous KERNEL sources the PMC traffic record belongs to (the host side, svgf_api.hip, holds no kernel code:
    which kernel ran on which level is in the record itself, per level)."""
    import hashlib
    h = hashlib.sha256()
    for name in ("svgf_atrous_lane_impl.h", "svgf_atrous_lane.hip", "svgf_atrous_prepare_fused.hip"):      # (what the recorded workloads, 1920 and 3840 columns, launch)
        with open(os.path.join(ROOT, "cuda-path-tracer-denoising_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def physical_cores():
    """(physical cores, CPU model string) of this host, from /proc/cpuinfo; SMT siblings are counted once."""
    try:
        seen, model, phys, core = set(), "", None, None
        for ln in open("/proc/cpuinfo"):
            k, _, v = ln.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name" and not model:
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None and core is not None:
                seen.add((phys, core)); phys = core = None
        if phys is not None and core is not None:
            seen.add((phys, core))
        if seen:
            return len(seen), model
    except OSError:
        pass
    return os.cpu_count() or 1, ""


def cgroup_cpu_quota():
    """CPUs' worth of time this container may use (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return max(1, int(int(q) / int(per)))
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return max(1, q // per)
    except (OSError, ValueError):
        pass
    return None


def cpu_baseline(pkg, frames, params, budget_s=25.0, threads=None, what="full-SVGF", all_cores=True, min_frames=5, one_thread=True):
    """CPU oracle (oracle/svgf_oracle.c, OpenMP) on the same workload, bounded sample.  Threads are bound one per physical
    core (OMP_PLACES=cores, OMP_PROC_BIND=spread, set in main() before libgomp starts) and rows are dealt in static blocks, so
    that the all-cores figure is not the SMT + unbound one of round 2 (slower than 64 threads)."""
    orc = ge.load_oracle()
    cores, cpu_model = physical_cores()
    quota = cgroup_cpu_quota()
    # The GPU boxes of this pool are containers with a CFS quota (cpu.max = 16 CPUs on a 2 x 64-core host): threads beyond the
    # quota are throttled, which is what made round 2's "all cores" figure slower than its 64-thread one
    # (tests/cpu_oracle_thread_scaling.py: 16 threads 8.7, 24: 10.4, 64: 5.2-6.5, 128: 3.5-4.2 Mpix/s).  The leg uses as many
    # threads as the container can actually run.
    usable = min(cores, quota) if quota else cores
    threads = min(usable, 64) if threads is None else threads
    o = orc.Oracle(pkg, W, H, threads=threads)
    ts = []
    t_start = time.perf_counter()
    for f in range(1 + max(5, min_frames)):
        c, g, cam = frames[f % len(frames)]
        t0 = time.perf_counter()
        o.denoise(c, g, cam, params)
        dt = time.perf_counter() - t0
        if f >= 1:                       # frame 0 has no history (cheaper temporal pass): not representative
            ts.append(dt)
        if time.perf_counter() - t_start > budget_s and len(ts) >= 1:
            break
    o.free()
    n = len(ts)
    mp = lambda t: round(W * H / t / 1e6, 3)      # noqa: E731
    res = {"value": mp(float(np.median(ts))), "unit": "Mpixels/s", "cores": threads, "kind": "port",
           "statistic": "median over the sampled frames", "frames": n, "min": mp(max(ts)), "max": mp(min(ts)),
           "kind_note": "oracle/svgf_oracle.c, the CPU restatement of src/denoise.cu pinned to the reference's own outputs (SURVEY.md 8(d) "
                        "allows it as the CPU baseline); NOT the reference's source rebuilt for CPU, which north_star words",
           "sample": f"{n} steady-state frames of the same {W}x{H} {what} workload, oracle/svgf_oracle.c "
                     f"(gcc -O2, OpenMP static row blocks, threads bound to cores; {threads} threads; host: {cores} physical cores, "
                     f"{os.cpu_count()} hardware threads, {cpu_model}; container CPU quota: {quota if quota else 'none'})",
           "host": {"physical_cores": cores, "hardware_threads": os.cpu_count(), "cpu_model": cpu_model, "cgroup_cpu_quota": quota}}
    if one_thread and threads > 1:           # the scalar figure beside it (BASELINE configs[0] words its CPU leg "single-threaded")
        o1 = orc.Oracle(pkg, W, H, threads=1)
        t1 = []
        for f in range(3):
            c, g, cam = frames[f % len(frames)]
            t0 = time.perf_counter()
            o1.denoise(c, g, cam, params)
            if f >= 1:
                t1.append(time.perf_counter() - t0)
        o1.free()
        res["one_thread"] = {"value": mp(float(np.median(t1))), "unit": "Mpixels/s", "cores": 1, "frames": len(t1), "min": mp(max(t1)), "max": mp(min(t1))}
    if all_cores and usable > threads:       # north_star: "the same box's host cores" — every core the container may use
        oa = orc.Oracle(pkg, W, H, threads=usable)
        ts = []
        for f in range(3):
            c, g, cam = frames[f % len(frames)]
            t0 = time.perf_counter()
            oa.denoise(c, g, cam, params)
            if f >= 1:
                ts.append(time.perf_counter() - t0)
        oa.free()
        res["all_cores"] = {"value": round(W * H / (sum(ts) / len(ts)) / 1e6, 3), "unit": "Mpixels/s", "cores": usable,
                            "sample": f"{len(ts)} steady-state frames, one bound thread on each of the {usable} usable physical cores"}
    return res


def pkg_cpu_slice(rank, world):
    """N > 1: each rank's host thread (and its telemetry sampler) is pinned to its own slice of the CPUs this process may use, so
    that eight launch loops on a 16-CPU cgroup do not migrate over each other.  Returns the slice (None: not pinned)."""
    if world <= 1:
        return None
    try:
        cpus = sorted(os.sched_getaffinity(0))
        per = max(1, len(cpus) // world)
        mine = cpus[(rank * per) % len(cpus):(rank * per) % len(cpus) + per] or cpus
        os.sched_setaffinity(0, mine)
        return mine
    except (AttributeError, OSError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--config", default="1080p-static", choices=["1080p-static", "1080p-moving", "4k-static", "4k-moving", "4k-room", "config1"],
                    help="1080p-static = BASELINE configs[1] (the default, the headline metric); 1080p-moving = configs[2] "
                         "(64-frame moving-camera sequence); 4k-static = configs[3]; 4k-moving = the same at 3840x2160; 4k-room = "
                         "configs[4] as worded: room.txt (its primitives + 2 810 triangles, ray-cast by the device producer) at "
                         "3840x2160, one independent frame sequence per GPU — the command for the 8-GPU node is `bench.py --gpus 8 "
                         "--config 4k-room`; config1 = configs[0]: 800x800, temporal off, one a-trous level, CPU leg single-threaded")
    ap.add_argument("--kernel-variant", type=int, default=0,
                    help="SvgfParams::kernel_variant (0 = the library's default choice; 2 strip, 4 lane-marching kernel ...)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="order every frame on the one stream (SvgfParams::inputs_ready = 0).  Default: the frame pipeline (ABI 0.8) — "
                         "the inputs of this benchmark are resident and two output buffers alternate, which is what inputs_ready "
                         "promises; consecutive frames then run on two internal streams (frame n's levels 2-5 beside frame n+1's "
                         "temporal pass + level 1), bit-identical results")
    ap.add_argument("--overlap", action="store_true", help="(old name of the default; kept so that old command lines still run)")
    ap.add_argument("--no-overlap", action="store_true", help="same as --no-pipeline")
    ap.add_argument("--latency-calls", type=int, default=60,
                    help="svgf_denoise + svgf_sync pairs of the latency measurement (SURVEY.md 8(d)(i): median of >= 50 after 10 warm-ups)")
    ap.add_argument("--host-inputs", action="store_true",
                    help="render the synthetic frames with numpy and upload them (default: the device-side producer, "
                         "svgf_synth_render, SURVEY.md 8f row f1; both give the same frames bit for bit)")
    ap.add_argument("--planar-inputs", action="store_true",
                    help="SURVEY.md 8f row f1: the producer writes the G-buffer straight into the denoiser's planes (svgf_planar_gbuffer / "
                         "svgf_denoise_planar) instead of handing over 52-byte AoS texels; static-camera configs only (the planes of "
                         "both history parities are filled once, before the timed region, like the AoS inputs)")
    ap.add_argument("--cadence-hz", type=float, default=60.0,
                    help="also measure the state an interactive renderer lives in (N = 1 only): producer + svgf_denoise + display pack once per "
                         "1/Hz seconds with the GPU idle in between, timed by HIP events around the denoise; 0 = skip the leg")
    ap.add_argument("--cadence-frames", type=int, default=40)
    ap.add_argument("--trial-reps", type=int, default=3, help="pipelined / ordered regions of --steps frames each, alternating, that decide which way the timed steps run")
    ap.add_argument("--min-warmup-seconds", type=float, default=0.6,
                    help="untimed steps continue after --warmup until this much wall time has passed (clock ramp, history "
                         "fill): a --steps 20 run then measures what a --steps 200 run measures")
    a = ap.parse_args()

    # the CPU-baseline leg: one OpenMP thread per physical core, pinned (read by libgomp when the oracle library is loaded)
    # The frame pipeline wants its two internal streams on hardware queues of their own.  The HIP runtime spreads a process's streams
    # over GPU_MAX_HW_QUEUES queues (default 4): enough for a plain process, not once RCCL has created its streams (every multi-GPU
    # rank) — the pipelined frames then share a queue and gain nothing (measured: 0.259 against 0.237 ms per frame with 8 queues,
    # profiles/r05_exp_pipeline.log).  Read by the runtime at initialisation, so it is set before torch is imported; a value from
    # the environment wins.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    cpus_of_rank = None
    if a.gpus > 1 and "WORLD_SIZE" in os.environ:
        # N > 1: this rank's slice of the CPUs, taken BEFORE any OpenMP runtime starts (torch's binds the main thread to one core when
        # OMP_PROC_BIND is set, and an affinity mask read after that is that one core: every rank would pin itself to the same one)
        cpus_of_rank = pkg_cpu_slice(int(os.environ.get("RANK", "0")), int(os.environ["WORLD_SIZE"]))
    elif a.gpus == 1:
        os.environ.setdefault("OMP_PLACES", "cores")
        os.environ.setdefault("OMP_PROC_BIND", "spread")     # the CPU-baseline leg (N = 1 only): one thread per core, spread over the CCDs: the oracle lives in L3 (close: 5.2, spread: 8.7 Mpix/s at 16 threads)

    import torch
    # SVGF_BENCH_SHARE_DEVICE=1 (tests only): every rank uses device 0 and the ranks rendezvous over gloo — N processes with N
    # contexts on ONE GPU.  It exercises everything of an N-GPU run that does not need N GPUs (the launcher below, one context per
    # process, barrier + MAX-time / SUM-pixels reduction, the single JSON line); its throughput figure means nothing.
    share_device = bool(os.environ.get("SVGF_BENCH_SHARE_DEVICE"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher — one process per GPU, same rendezvous torchrun would set up
        if not torch.cuda.is_available() or (torch.cuda.device_count() < a.gpus and not share_device):
            raise SystemExit(f"bench.py --gpus {a.gpus}: only {torch.cuda.device_count() if torch.cuda.is_available() else 0} GPU(s) visible")
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        procs = []
        for r in range(a.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
        rcs = [p.wait() for p in procs]
        raise SystemExit(max(rcs))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if share_device else int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    dist = None
    if world > 1 or os.environ.get("SVGF_BENCH_FORCE_DIST"):   # the env var exercises the RCCL path with one rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:      # SVGF_BENCH_FORCE_DIST without a launcher: any free port
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        torch.cuda.set_device(local_rank)
        if share_device:      # RCCL wants one GPU per rank; two ranks on one device rendezvous over gloo (CPU tensors)
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    # (created here, long before it is used: new_group is itself a collective, and a rank waiting in it right before the timed region
    # would idle its GPU)
    busy_group = dist.new_group(backend="gloo") if (dist is not None and world > 1) else None
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    global W, H
    if a.config.startswith("4k"):
        W, H = 3840, 2160
    if a.config == "config1":
        W, H = 800, 800
    moving = a.config.endswith("moving")
    # one rank per node builds (normally a no-op: the libraries are prebuilt in-tree); the others wait
    if local_rank == 0:
        ge.build()
    if dist is not None:
        dist.barrier()
    pkg = ge.load_package()
    params = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, atrous_nlevel=NLEVEL, history_level=1)
    params.set(kernel_variant=a.kernel_variant)
    if a.config == "config1":   # BASELINE configs[0]: the reference's own CPU-runnable case
        params.set(temporal_enable=0, atrous_nlevel=1)
    pipeline = not (a.no_pipeline or a.no_overlap)      # (--planar-inputs: the planes of both parities are filled once, before the timed region: the promise holds)
    params.set(inputs_ready=1 if pipeline else 0)

    # every rank owns one independent sequence (its own seed); 4 distinct noisy frames, static camera, resident in HBM
    seq = pkg.farm.shard(world, world, rank)[0]
    nsrc = 64 if moving else 4
    if a.config == "4k-room":
        # BASELINE configs[4]: room.txt at 3840x2160.  The scene's primitives and triangles are data made from the reference's files
        # by tests/golden/make_producer_inputs.py (the reference itself does not exist on the GPU box); the device producer
        # ray-casts them (svgf_scene_render_mesh) with the reference's camera, static, noise stream seeded per rank.
        gd = os.path.join(ROOT, "tests", "golden", "ref_scenes")
        pi = np.load(os.path.join(gd, "room_producer_inputs.npz"))
        rec = json.load(open(os.path.join(gd, "scene_records.json")))["room"]["camera"]
        sc = pkg.scene.Scene(materials={}, objects=[], camera=dict(eye=rec["position"], lookat=rec["lookAt"], fovy=rec["fov"][1]), skipped=[])
        cam_dicts = [pkg.scene.camera_for_frame(sc, 0, False) for f in range(nsrc)]
        d_in = [torch.empty((H, W, 3), dtype=torch.float32, device=dev) for _ in range(nsrc)]
        d_g = [torch.empty((H * W * 52,), dtype=torch.uint8, device=dev) for _ in range(nsrc)]
        for f in range(nsrc):
            pkg.binding.scene_render_mesh(d_in[f], d_g[f], W, H, cam_dicts[f], pi["geoms"], pi["geom_ids"], pi["tris"], pi["tri_ids"], pi["tri_albedo"],
                                          frame=f, seed=1000 + seq, device=local_rank)
        torch.cuda.synchronize(dev)
        frames = None
    elif a.host_inputs:
        frames = [pkg.synth.render_frame(W, H, f, seed=1000 + seq, moving=moving, noise_model="hash") for f in range(nsrc)]
        d_in = [torch.from_numpy(f[0]).to(dev) for f in frames]
        d_g = [torch.from_numpy(f[1].view(np.uint8).reshape(-1)).to(dev) for f in frames]
        cam_dicts = [f[2] for f in frames]
    else:   # produced where the reference produces them: on the device (the path tracer's role)
        cam_dicts = [pkg.synth.camera_for_frame(f, moving) for f in range(nsrc)]
        d_in = [torch.empty((H, W, 3), dtype=torch.float32, device=dev) for _ in range(nsrc)]
        d_g = [torch.empty((H * W * 52,), dtype=torch.uint8, device=dev) for _ in range(nsrc)]
        for f in range(nsrc):
            pkg.binding.synth_render(d_in[f], d_g[f], W, H, cam_dicts[f], f, seed=1000 + seq, device=local_rank)
        torch.cuda.synchronize(dev)
        frames = None
    cams = [pkg.SvgfCamera.from_dict(c) for c in cam_dicts]
    outs = [torch.empty((H, W, 3), dtype=torch.float32, device=dev) for _ in range(2)]      # alternate: frame n+1 may write while frame n still does
    out = outs[0]
    # the pipelined context is CREATED pipelined (svgf_create_ex: second plane set, internal streams, hardware-queue probe): no frame
    # of the sequence allocates or synchronises.  The library refuses the promise when its two streams share a hardware queue.
    t_cr = time.perf_counter()
    den = pkg.Denoiser(W, H, device=local_rank, pipelined=pipeline)
    create_ms = (time.perf_counter() - t_cr) * 1e3
    pipeline_status = den.pipeline_status()
    pipeline_refused = den.last_error() if (pipeline and pipeline_status != 1) else None
    pipeline = pipeline and pipeline_status == 1
    if not pipeline:
        params.set(inputs_ready=0)
    stream = torch.cuda.current_stream(dev)
    # the ordered twin: a second context whose frames are never promised (latency leg, ordered leg, and the fallback below)
    den_o, op = den, params
    if pipeline:
        den_o = pkg.Denoiser(W, H, device=local_rank)
        op = pkg.SvgfParams.from_buffer_copy(params).set(inputs_ready=0)
    PROFILE_STRIDE = 10     # an event pair attached to every kernel dispatch of every 10th timed step (a timed frame runs ~25 us longer)
    den.profile_stride(PROFILE_STRIDE)
    den.profile_enable(a.steps)

    if a.planar_inputs:
        if moving or a.host_inputs:
            raise SystemExit("--planar-inputs: static-camera configs with the device producer only")
        for dd, pr in {id(den): (den, params), id(den_o): (den_o, op)}.values():
            for _ in range(2):      # both plane sets (they alternate with the history) get the static scene's G-buffer
                pkg.binding.synth_render_planar(d_in[0], dd.planar_gbuffer(), W, H, cam_dicts[0], 0, seed=1000 + seq, device=local_rank)
                dd.denoise_planar(out, d_in[0], cams[0], pr, stream=stream)
            torch.cuda.synchronize(dev)

    cur = {"den": den, "params": params}      # what the timed steps run on (the pipeline trial below may swap in the ordered twin)
    calls = {"n": 0}                          # every frame this process has enqueued (warm-up, trial, settling, timed, other legs)

    def step(i):
        k = i % nsrc
        calls["n"] += 1
        if a.planar_inputs:
            cur["den"].denoise_planar(outs[i & 1], d_in[k], cams[k], cur["params"], stream=stream)
        else:
            cur["den"].denoise(outs[i & 1], d_in[k], d_g[k], cams[k], cur["params"], stream=stream)
        return W * H

    def step_ordered(i):
        k = i % nsrc
        calls["n"] += 1
        if a.planar_inputs:
            den_o.denoise_planar(outs[i & 1], d_in[k], cams[k], op, stream=stream)
        else:
            den_o.denoise(outs[i & 1], d_in[k], d_g[k], cams[k], op, stream=stream)

    # Clock / power / temperature sampler (tools/telemetry.py): created and started BEFORE the warm-up.  Finding its hwmon node
    # (a sysfs walk, a device-property query) takes milliseconds; between the warm-up and the timed region that was enough idle
    # time for the GPU to leave its sustained power state, and the first timed frames paid the ramp back up: 0.266 -> 0.313 ms per
    # step on the driver's 20-step command with identical kernels (profiles/r04_ab_bench_r03_vs_r04_tree.log, r04_bisect*.log).
    # (SVGF_BENCH_NO_TELEMETRY=1: no sampler thread at all.)
    tm_all = telemetry.Sampler(local_rank, period_s=float(os.environ.get("SVGF_BENCH_TELEMETRY_PERIOD", "0.005")))
    if not os.environ.get("SVGF_BENCH_NO_TELEMETRY"):
        tm_all.start()
    # warm-up is run with profiling slots too, then the frame counter restarts so slots hold the timed steps only
    # the first 12 frames of the process (cold: clocks and socket power not ramped yet): timed, part of the warm-up
    torch.cuda.synchronize(dev)
    t_f0 = time.perf_counter()
    step(0)               # the very first frame: kernel code load (the pipeline's planes and streams exist since svgf_create_ex)
    torch.cuda.synchronize(dev)
    first_frame_ms = (time.perf_counter() - t_f0) * 1e3
    t_c0 = time.perf_counter()
    for i in range(1, 13):
        step(i)
    torch.cuda.synchronize(dev)
    cold_ms = (time.perf_counter() - t_c0) / 12 * 1e3
    t_w = time.perf_counter()
    n_w = 13
    while n_w < a.warmup or time.perf_counter() - t_w < a.min_warmup_seconds:
        step(n_w)
        n_w += 1
        if n_w % 32 == 0:
            torch.cuda.synchronize(dev)
    torch.cuda.synchronize(dev)
    # Which way do the timed steps run?  A trial with the timed region's OWN estimator: regions of exactly --steps frames between two
    # synchronisations, same profiling stride — --trial-reps consecutive regions of each way, each way behind 0.25 s of its own frames
    # back to back; the pipeline runs only if its MEDIAN region beats the ordered median by >= 3 %.  Then the chosen way runs for
    # another 0.3 s before the timed region.  Both settling stretches matter: the socket's power controller reacts to a CHANGE of load
    # within tens of milliseconds — 20 timed steps right behind a trial that alternated 5 ms of ordered and 5 ms of pipelined frames ran
    # at 0.268-0.273 ms while the regions before them ran at 0.242 and the regions 30 ms later at 0.243 (profiles/r06_bench_settle.log);
    # round 5's trial (minimum of two 48-frame bursts, 1 % margin) said 0.249 against 0.267 on the driver's box and the timed steps
    # then ran at 0.270 against 0.268 ordered.
    def region(fn, n):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(n):
            fn(i)
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / n * 1e3

    def step_pipe(i):
        k = i % nsrc
        calls["n"] += 1
        if a.planar_inputs:
            den.denoise_planar(outs[i & 1], d_in[k], cams[k], params, stream=stream)
        else:
            den.denoise(outs[i & 1], d_in[k], d_g[k], cams[k], params, stream=stream)

    def settle(fn, seconds):
        t, k = time.perf_counter(), 0
        while time.perf_counter() - t < seconds:
            fn(k)
            k += 1
            if k % 32 == 0:
                torch.cuda.synchronize(dev)
        torch.cuda.synchronize(dev)

    pipeline_trial = None
    if pipeline:
        den_o.profile_stride(PROFILE_STRIDE)
        den_o.profile_enable(a.steps)
        tr_o, tr_p = [], []
        settle(step_ordered, 0.25)
        for _ in range(max(1, a.trial_reps)):
            den_o.profile_enable(a.steps); tr_o.append(region(step_ordered, a.steps))
        settle(step_pipe, 0.25)
        for _ in range(max(1, a.trial_reps)):
            den.profile_enable(a.steps); tr_p.append(region(step_pipe, a.steps))
        m_o, m_p = float(np.median(tr_o)), float(np.median(tr_p))
        pipeline_trial = {"pipelined_ms": [round(t, 5) for t in tr_p], "ordered_ms": [round(t, 5) for t in tr_o],
                          "pipelined_median_ms": round(m_p, 5), "ordered_median_ms": round(m_o, 5), "frames_each": a.steps,
                          "rule": "pipelined iff its median region <= 0.97 x the ordered median (consecutive regions of --steps frames, each way behind 0.25 s of its own frames)"}
        if m_p > 0.97 * m_o:      # not worth it on this box: the timed steps run ordered (and the line says so)
            pipeline = False
            cur["den"], cur["params"] = den_o, op
        settle(step, 0.3)         # the chosen way, back to back, right up to the timed region
    den_first, den = den, cur["den"]
    t_warm_end = time.perf_counter()
    untimed_before = calls["n"]
    den.profile_enable(a.steps)
    t_region0 = time.perf_counter()
    # N > 1: the ranks' warm-ups and trials end at different times; they meet on a host-side (gloo) group while their GPUs keep running
    # untimed frames (farm.timed_region), then the bracketing barrier + synchronisation, then the K timed steps
    dt, pixels = pkg.farm.timed_region(step, a.steps, 0, lambda: torch.cuda.synchronize(dev), dist=dist, device=None if share_device else dev,
                                       busy_group=busy_group)
    t_region1 = time.perf_counter()
    if os.environ.get("SVGF_BENCH_DEBUG"):
        dbg = [region(step, a.steps) for _ in range(4)]
        dbg2 = [pkg.farm.timed_region(step, a.steps, 0, lambda: torch.cuda.synchronize(dev), dist=dist, device=None if share_device else dev)[0] / a.steps * 1e3 for _ in range(3)]
        print(f"[debug] timed {dt / a.steps * 1e3:.4f}; 4 more regions: {dbg}; 3 more timed_region: {dbg2}", file=sys.stderr)
    # every rank's own clock around its own K steps (the reduction above keeps only the maximum), its device and its CPU slice
    mine = {"rank": rank, "device": local_rank, "device_name": torch.cuda.get_device_name(dev), "ms_per_step": round(pkg.farm.last_local_seconds() / a.steps * 1e3, 5),
            "cpus": cpus_of_rank}
    per_rank = [mine]
    if dist is not None and world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    # per-kernel durations of the timed steps (HIP events attached to the dispatches on the launch stream)
    atrous_ms, temporal_ms, fused_ms = [], [], []
    for s in range(min(a.steps, den.profile_frames())):
        for kind, ms in den.profile_read(s):
            if kind == pkg.binding.KERNEL_ATROUS:
                atrous_ms.append(ms)
            elif kind == pkg.binding.KERNEL_TEMPORAL:
                temporal_ms.append(ms)
            elif kind == pkg.binding.KERNEL_FUSED:
                fused_ms.append(ms)

    # The metric as SURVEY.md 8(d)(i) defines it: wall time of ONE svgf_denoise call, device-synchronised (what the reference's
    # synchronous denoise() delivers, src/denoise.cu:401), median of >= 50 calls after 10 warm-ups.  No per-kernel events; it
    # follows the timed region directly, so the clocks are the sustained ones (the telemetry of the two is reported side by side).
    # A caller that waits after every call promises nothing (the reference's shim does not): the latency is measured on a context
    # of its own whose frames are ordered on the stream — in a pipelined context the four cross-stream events of a frame cost such a
    # caller 16 us per call (0.271 -> 0.287 ms, profiles/r05_exp_pipeline.log) and buy it nothing.
    den.profile_enable(0)
    for i in range(10):
        step_ordered(i); den_o.sync()
    lat = []
    t_lat0 = time.perf_counter()
    for i in range(max(50, a.latency_calls)):
        t0 = time.perf_counter()
        step_ordered(i)
        den_o.sync()
        lat.append(time.perf_counter() - t0)
    t_lat1 = time.perf_counter()
    lat_ms = np.asarray(lat) * 1e3
    # What the reference's caller does next (src/pathtrace.cu:446-449): the side-by-side pack kernel, then a blocking copy.  With the shim's
    # trailing device sync (src/denoise.cu:401) the pack kernel is launched after a host round trip; without it
    # (-DSVGF_COMPAT_NO_TRAILING_SYNC in denoise_compat.cpp) it is already queued when the last level ends.  Same results either way.
    caller_ms = None
    if world == 1 and not a.planar_inputs:
        pbo_l = torch.empty((H, 2 * W, 4), dtype=torch.uint8, device=dev)
        legs = {"denoise_sync_pack_sync": [], "denoise_pack_sync": []}
        for i in range(2 * max(50, a.latency_calls) + 20):
            which = "denoise_sync_pack_sync" if i & 1 else "denoise_pack_sync"
            t0 = time.perf_counter()
            step_ordered(i)
            if i & 1:
                den_o.sync()
            pkg.binding.display_pack(pbo_l, d_in[i % nsrc], outs[i & 1], W, H, device=local_rank, stream=stream)
            den_o.sync()
            if i >= 20:
                legs[which].append((time.perf_counter() - t0) * 1e3)
        caller_ms = {k: round(float(np.median(v)), 5) for k, v in legs.items()}
        caller_ms["what"] = ("median wall time of svgf_denoise [+ svgf_sync] + svgf_display_pack + sync per frame, the two orders alternating: the reference's "
                             "(device sync inside denoise(), src/denoise.cu:401) and the shim built with -DSVGF_COMPAT_NO_TRAILING_SYNC")
    if not (np.isfinite(outs[0].sum().item()) and np.isfinite(outs[1].sum().item())):
        raise SystemExit("bench: non-finite output")

    # The OTHER leg, beside the headline: the same --steps frames the other way (ordered if the timed steps ran pipelined, pipelined if
    # they ran ordered), an identical region behind its own stretch of back-to-back frames (the latency loop above idles the GPU).
    other_leg = None
    if pipeline_trial is not None:
        other_is_ordered = pipeline
        fn_other = step_ordered if other_is_ordered else step_pipe
        t_o = time.perf_counter()
        i = 0
        while time.perf_counter() - t_o < 0.4:
            fn_other(i)
            i += 1
            if i % 32 == 0:
                torch.cuda.synchronize(dev)
        (den_o if other_is_ordered else den_first).profile_stride(PROFILE_STRIDE)
        (den_o if other_is_ordered else den_first).profile_enable(a.steps)
        t_oth0 = time.perf_counter()
        other_ms = region(fn_other, a.steps)
        t_oth1 = time.perf_counter()
        other_leg = {"ms_per_step": round(other_ms, 5), "value": round(W * H / other_ms / 1e3, 2), "unit": "Mpixels/s",
                     "telemetry": tm_all.summary(t_oth0, t_oth1),
                     "what": ("the same steps with every frame ORDERED on the one stream (a second context, inputs_ready = 0)" if other_is_ordered else
                              "the same steps PIPELINED (inputs_ready = 1 on the context created with SVGF_CREATE_PIPELINED)")
                             + ": one region of --steps frames between two synchronisations, after 0.4 s of the same frames; rank 0"}

    # the same kernels with EVERY launch of 16 consecutive frames timed (outside the timed region, sustained clock state)
    iso_params = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, atrous_nlevel=NLEVEL, history_level=1)
    iso_params.set(kernel_variant=a.kernel_variant)
    if a.config == "config1":
        iso_params.set(temporal_enable=0, atrous_nlevel=1)
    def iso_step(i):
        if a.planar_inputs:
            den_o.denoise_planar(out, d_in[i % nsrc], cams[i % nsrc], iso_params, stream=stream)
        else:
            den_o.denoise(out, d_in[i % nsrc], d_g[i % nsrc], cams[i % nsrc], iso_params, stream=stream)

    den_o.profile_stride(1)
    den_o.profile_enable(16)
    for i in range(256):      # back into the sustained clock state (the latency loop above idles the GPU between calls: DESIGN.md 6.2)
        iso_step(i)
    torch.cuda.synchronize(dev)
    den_o.profile_enable(16)    # same slot count: counters restart, no event is re-created, no idle time
    t_iso0 = time.perf_counter()
    for i in range(16):
        iso_step(i)
    torch.cuda.synchronize(dev)
    t_iso1 = time.perf_counter()
    iso_atrous_ms = [ms for s in range(den_o.profile_frames()) for kind, ms in den_o.profile_read(s)
                     if kind == pkg.binding.KERNEL_ATROUS]
    iso_temporal_ms = [ms for s in range(den_o.profile_frames()) for kind, ms in den_o.profile_read(s)
                       if kind == pkg.binding.KERNEL_TEMPORAL]
    iso_fused_ms = [ms for s in range(den_o.profile_frames()) for kind, ms in den_o.profile_read(s) if kind == pkg.binding.KERNEL_FUSED]
    den_o.profile_enable(0)

    # The state an interactive renderer lives in: one frame per 1/Hz seconds — producer (svgf_synth_render), svgf_denoise, display pack
    # (svgf_display_pack) enqueued together, the GPU idle until the next tick.  After >= 5 ms of idle time the socket has left its
    # sustained state and the same kernels run 7-18 % longer.  Timed by HIP events around the denoise and around
    # the whole frame, on the launch stream.
    cadence = None
    if a.cadence_hz > 0 and world == 1 and not a.planar_inputs and a.config != "4k-room":
        pbo = torch.empty((H, 2 * W, 4), dtype=torch.uint8, device=dev)
        c_in = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
        c_g = torch.empty((H * W * 52,), dtype=torch.uint8, device=dev)
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(a.cadence_frames)]
        period = 1.0 / a.cadence_hz
        n_prof = 12      # further ticks behind the timed ones, with an event pair on every dispatch: the kernels' own durations in this state
        ev += [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(n_prof)]
        den_o.profile_enable(0)
        torch.cuda.synchronize(dev)
        t_cad0 = time.perf_counter()
        nxt = t_cad0
        host_ms = []
        for f in range(a.cadence_frames + n_prof):
            if f == a.cadence_frames:
                den_o.profile_stride(1)
                den_o.profile_enable(n_prof)
            while time.perf_counter() < nxt:
                time.sleep(0.0002)
            nxt += period
            k = f % nsrc
            th0 = time.perf_counter()
            ev[f][0].record(stream)
            pkg.binding.synth_render(c_in, c_g, W, H, cam_dicts[k], f, seed=1000 + seq, device=local_rank, stream=stream)
            ev[f][1].record(stream)
            den_o.denoise(out, c_in, c_g, cams[k], op, stream=stream)
            ev[f][2].record(stream)
            pkg.binding.display_pack(pbo, c_in, out, W, H, device=local_rank, stream=stream)
            ev[f][3].record(stream)
            den_o.sync_stream(stream)          # the frame is on the screen before the next one starts (a renderer's swap)
            host_ms.append((time.perf_counter() - th0) * 1e3)
        t_cad1 = time.perf_counter()
        cad_rows = [den_o.profile_read(f) for f in range(min(n_prof, den_o.profile_frames()))]
        cad_t = [ms for r in cad_rows for kind, ms in r if kind == pkg.binding.KERNEL_TEMPORAL]
        cad_l = [ms for r in cad_rows for kind, ms in r if kind in (pkg.binding.KERNEL_ATROUS, pkg.binding.KERNEL_FUSED)]
        den_o.profile_enable(0)
        den_ms = [ev[f][1].elapsed_time(ev[f][2]) for f in range(4, a.cadence_frames)]
        frm_ms = [ev[f][0].elapsed_time(ev[f][3]) for f in range(4, a.cadence_frames)]
        cadence = {"hz": a.cadence_hz, "frames": a.cadence_frames - 4, "ms_per_step": round(float(np.median(den_ms)), 5),
                   "ms_per_step_p10_p90": [round(float(np.quantile(den_ms, 0.1)), 5), round(float(np.quantile(den_ms, 0.9)), 5)],
                   "frame_ms_producer_denoise_pack": round(float(np.median(frm_ms)), 5),
                   "host_ms_enqueue_to_done": round(float(np.median(host_ms[4:a.cadence_frames])), 5),
                   "kernels_us": {"temporal": round(float(np.mean(cad_t)) * 1e3, 2) if cad_t else None,
                                  "atrous_level_mean": round(float(np.mean(cad_l)) * 1e3, 2) if cad_l else None,
                                  "vs_sustained": {"temporal": round(float(np.mean(cad_t)) / float(np.mean(iso_temporal_ms)), 3) if (cad_t and iso_temporal_ms) else None,
                                                   "atrous_level_mean": round(float(np.mean(cad_l)) / float(np.mean(iso_atrous_ms or iso_fused_ms)), 3) if (cad_l and (iso_atrous_ms or iso_fused_ms)) else None}},
                   "telemetry": tm_all.summary(t_cad0, t_cad1),
                   "what": "one frame per 1/hz s: svgf_synth_render + svgf_denoise (ordered, inputs_ready = 0) + svgf_display_pack on one stream, then the "
                           "stream is waited for and the GPU idles until the next tick; ms_per_step = HIP events around the svgf_denoise of each "
                           "frame (median; the first 4 frames dropped; no per-kernel events): the same kernels as the sustained figure, in the state "
                           "a chip that idled a moment ago is in (DESIGN.md 6.2); kernels_us = the dispatches' own durations on 12 further ticks"}
    tm_all.stop()
    # config1 (non-temporal, ONE level): that level carries the prepare pass in its loader waves (one launch per frame, kind FUSED):
    # it is the a-trous launch of this configuration, timed with the prepare work inside it
    level_is_fused = not atrous_ms and bool(fused_ms)
    if level_is_fused:
        atrous_ms = list(fused_ms)
        iso_atrous_ms = list(iso_fused_ms)

    if rank == 0:
        traffic, traffic_note, sq_rec = None, "no PMC record", None
        try:   # HBM bytes per launch and SIMD activity from the committed PMC passes (profiles/pmc_traffic.json).  Counter passes
            # cannot run inside the timed region, so the figures are only reported while the kernels they were measured on are unchanged.
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                rec = json.load(f)
            if rec.get("kernel_sources_sha16") == kernel_sources_sha16():
                r_hbm = rec.get("by_resolution", {}).get(f"{W}x{H}")
                if r_hbm and a.config in ("1080p-static", "4k-static") and a.kernel_variant == 0 and not a.planar_inputs:
                    traffic = r_hbm["mean_bytes_per_launch"]
                    traffic_note = f"measured on these kernel sources (sha16 {rec['kernel_sources_sha16']}) at {W}x{H}, {r_hbm.get('source_file', '')}"
                else:
                    traffic_note = f"no PMC pass for this workload ({a.config}, {W}x{H})"
                sq_rec = rec.get("sq_activity", {}).get(f"{W}x{H}")
            else:
                traffic_note = "dropped: the a-trous kernel sources changed since the PMC passes (profiles/pmc_traffic.json)"
        except Exception:
            traffic = None
        value = pixels / dt / 1e6
        a_ms = float(np.mean(atrous_ms))
        # algorithmic bytes of the timed launch: 56 B/px for a plain level; the fused prepare + level launch of config1 reads the
        # 1-spp colour (12) and the G-buffer fields it needs (normal, position, geomId: 28) and writes the output (12) and the split
        # planes (28) — and, when something besides this level reads them, the colour + variance plane (16) and the level's own (16)
        bytes_px = ATROUS_BYTES_PER_PIXEL
        if level_is_fused:
            bytes_px = 12.0 + 28.0 + 12.0 + 28.0 + (16.0 if params.history_level != 1 else 0.0) + (16.0 if params.atrous_nlevel > 1 or params.history_level == 1 else 0.0)
        achieved = bytes_px * W * H / (a_ms * 1e-3) / 1e9
        iso_us = float(np.mean(iso_atrous_ms)) * 1e3
        iso_gbs = bytes_px * W * H / (iso_us * 1e-6) / 1e9
        line = {
            "metric": ("SVGF Mpixels/s (one non-temporal level, BASELINE configs[0]) at 800x800; \u00e0-trous HBM GB/s vs roofline" if a.config == "config1"
                       else "SVGF Mpixels/s (full pipeline) at 4K; \u00e0-trous HBM GB/s vs roofline" if a.config.startswith("4k")
                       else "SVGF Mpixels/s (full pipeline) at 1080p; \u00e0-trous HBM GB/s vs roofline"),
            "value": round(value, 2), "unit": "Mpixels/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "warmup_steps_run": n_w,
            # every frame this process ran before the K timed ones: the warm-up (>= --warmup steps and >= --min-warmup-seconds), the trial's
            # regions and the settling stretches (0.25 s each way + 0.3 s of the chosen way)
            "untimed_steps_before_timed_region": untimed_before,
            "ms_per_step": round(dt / a.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak",
            # which state of the GPU the K timed steps ran in: "sustained" = directly behind >= min_warmup_seconds of
            # back-to-back frames; `cold_ms_per_step` = the first 12 frames of this process (rank 0), for comparison
            "state": "sustained", "sustained_after_s": round(t_warm_end - t_c0, 3), "cold_ms_per_step": round(cold_ms, 5),
            "cold_is": "frames 2-13 of the process", "first_frame_ms": round(first_frame_ms, 3),
            # the frame pipeline (SvgfParams::inputs_ready, include/svgf.h): consecutive frames of the sequence on two internal streams.
            # `ordered`: the same K steps of this process with every frame ordered on the one stream (round 4's way; rank 0)
            "frame_pipeline": pipeline, "frame_pipeline_trial": pipeline_trial,
            "frame_pipeline_status": pipeline_status, "frame_pipeline_refused": pipeline_refused, "context_create_ms": round(create_ms, 3),
            ("ordered" if (pipeline or pipeline_trial is None) else "pipelined"): other_leg,
            "per_rank": per_rank,
            # one svgf_denoise + svgf_sync, median of the calls below (host wall clock around the pair)
            "latency_ms_sync": round(float(np.median(lat_ms)), 5),
            "latency": {"calls": len(lat), "warmup_calls": 10, "median_ms": round(float(np.median(lat_ms)), 5),
                        "p10_ms": round(float(np.quantile(lat_ms, 0.1)), 5), "p90_ms": round(float(np.quantile(lat_ms, 0.9)), 5),
                        "mpixels_per_s": round(W * H / (float(np.median(lat_ms)) * 1e-3) / 1e6, 1),
                        "what": "wall time of svgf_denoise + svgf_sync per call (rank 0), what the reference's synchronous denoise() gives its caller"
                                + ("; measured on a context of its own with inputs_ready = 0: a caller that waits after every call promises nothing" if pipeline else "")},
            "latency_caller_ms": caller_ms,
            "idle_before_timed_region_ms": round((t_region0 - t_warm_end) * 1e3, 3),      # GPU idle between warm-up and timed steps
            "telemetry": {"timed_region": tm_all.summary(t_region0, t_region1), "latency_calls": tm_all.summary(t_lat0, t_lat1),
                          "isolated_16_frames": tm_all.summary(t_iso0, t_iso1)},
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"cornell-like {W}x{H}, variance fill + ONE a-trous level (temporal off), " if a.config == "config1" else
                                    f"room.txt (primitives + 2 810 triangles, device ray-cast) {W}x{H}, full SVGF (temporal + 5 a-trous levels, history_level 1), " if a.config == "4k-room" else
                                    f"cornell-like {W}x{H}, full SVGF (temporal + 5 a-trous levels, history_level 1), ")
                                   + ("moving camera, 64-frame sequence replayed" if moving else "static camera, steady-state history")
                                   + ("; G-buffer handed over as planes written in place by the producer (svgf_denoise_planar)" if a.planar_inputs else "")
                                   + "; one independent sequence per GPU", "name": a.config,
                       "width": W, "height": H, "atrous_levels": 1 if a.config == "config1" else NLEVEL,
                       "parallelism": f"replicas{world}" + ("-sharing-one-device(test-only)" if share_device else "")},
            # the dominant kernel's OWN figure: every a-trous launch of 16 consecutive ORDERED frames (one kernel at a time on the GPU),
            # sustained state, the dispatches' own begin / end timestamps — what `rocprofv3 --kernel-trace --stats` of an ordered run
            # reproduces (profiles/).  The same launches as they ran among the timed steps (with the frame pipeline: sharing the GPU with
            # the other frame's kernels) are under `timed_region`.
            "roofline": {"bound": "hbm", "achieved": round(iso_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(iso_gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_measured_live": False, "traffic_provenance": traffic_note,
                         "traffic_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, profiles/pmc_traffic.json (FETCH x2 per MI355X_MICROARCH.md)",
                         "kernel": "one a-trous level: k_atrous_lane (steps 2-32; k_atrous_strip where the library's cost model prefers it), mean over the level launches of 16 ordered frames; in config1 the one launch per frame that carries the prepare pass in its loader waves", "bytes_per_launch": bytes_px * W * H, "bytes_per_pixel": bytes_px,
                         "mean_launch_us": round(iso_us, 2), "launches_timed": len(iso_atrous_ms),
                         "launch_includes_fused_prepare_pass": level_is_fused,
                         "note": "durations are the dispatches' own begin / end timestamps (HIP events attached by hipExtLaunchKernelGGL on the launch stream); frac x peak x mean_launch_us = bytes_per_launch",
                         "timed_region": {"mean_launch_us": round(a_ms * 1e3, 2), "achieved": round(achieved, 1), "frac": round(achieved / HBM_PEAK_GBS, 4),
                                          "launches_timed": len(atrous_ms), "every_kth_frame": PROFILE_STRIDE,
                                          # with the frame pipeline the kernels of two consecutive frames share the GPU: a launch of the timed
                                          # region lasts longer than the same launch alone, while more than one is in flight
                                          "kernels_in_flight_mean": round(((0.0 if level_is_fused else sum(atrous_ms)) + sum(temporal_ms) + sum(fused_ms)) / max(1, len(temporal_ms) + len(fused_ms)) / (dt / a.steps * 1e3), 3)
                                                                    if (temporal_ms or fused_ms) else None,
                                          "note": ("frame pipeline on: the kernels of consecutive frames overlap" if pipeline else "everything ordered on one stream")},
                         # the secondary limiter.  24 taps x (2 v_sqrt + 1 v_exp) + 5 (centre, normalisation) per pixel-level
                         "transcendental_gops_isolated": round(77 * W * H / (iso_us * 1e-6) / 1e9, 1),
                         "transcendental_gops_isolated_is": "a formula, not a counter: 77 transcendental operations per pixel-level x pixels / the isolated launch duration",
                         # what actually bounds the kernel: SIMD instruction issue, 3/4 of it fp32 VALU.
                         # ~950 flop per pixel-level (24 taps x 37 + centre/normalisation) against the 157.3 TFLOP/s
                         # packed-fp32 vector peak; the PMC figure (SQ_ACTIVE_INST_ANY / SIMD time) is in profiles/.
                         "valu_view": {"fp32_tflops_isolated": round(950 * W * H / (iso_us * 1e-6) / 1e12, 1),
                                       "fp32_tflops_isolated_is": "a formula, not a counter: 950 flop per pixel-level x pixels / the isolated launch duration",
                                       "fp32_vector_peak_tflops": 157.3,
                                       # SQ_ACTIVE_INST_ANY x 4 / (launch duration x sclk x 1024 SIMDs) of the committed SQ pass, or null
                                       "simd_instruction_active_pmc": sq_rec["simd_instruction_active_mean"] if sq_rec else None,
                                       "simd_instruction_active_pmc_source": (f"profiles/pmc_traffic.json sq_activity[{W}x{H}] ({', '.join(sq_rec['source_files'])}; sclk {sq_rec['sclk_mhz']:.0f} MHz)"
                                                                              if sq_rec else "no SQ pass recorded for these kernel sources at this size")}},
            # the kernels' own durations (16 ordered frames, as `roofline`); `kernels_us_timed_region`: as they ran among the timed steps
            "kernels_us": {"temporal": round(float(np.mean(iso_temporal_ms)) * 1e3, 2) if iso_temporal_ms else None,
                           "fused_prepare_plus_level1": round(float(np.mean(iso_fused_ms)) * 1e3, 2) if iso_fused_ms else None,
                           "atrous_level_mean": round(iso_us, 2)},
            "kernels_us_timed_region": {"temporal": round(float(np.mean(temporal_ms)) * 1e3, 2) if temporal_ms else None,
                           "fused_prepare_plus_level1": round(float(np.mean(fused_ms)) * 1e3, 2) if fused_ms else None,
                           "atrous_level_mean": round(a_ms * 1e3, 2), "atrous_launches_per_frame": round(len(atrous_ms) / max(1, len(fused_ms) + len(temporal_ms)), 2) if (fused_ms or temporal_ms) else None},
            "cadence": cadence,
            "frame_algorithmic_gbs": round((bytes_px if a.config == "config1" else FRAME_BYTES_PER_PIXEL)
                                           * W * H / (dt / a.steps) / 1e9 / world * 1.0, 1),
        }
        if world == 1 and not a.no_cpu_baseline and a.config in ("1080p-static", "config1"):
            if frames is None:   # bring the device-produced frames to the host for the CPU leg
                frames = [(d_in[f].cpu().numpy(), d_g[f].cpu().numpy().view(pkg.synth.GBUFFER_DTYPE).reshape(H, W), cam_dicts[f])
                          for f in range(min(nsrc, 4))]
            host_frames = [(f[0], f[1], f[2]) for f in frames]
            if a.config == "config1":   # "single-threaded" is how BASELINE.json words this configuration
                line["cpu_baseline"] = cpu_baseline(pkg, host_frames, params, threads=1, what="one-level non-temporal", all_cores=False)
            else:
                line["cpu_baseline"] = cpu_baseline(pkg, host_frames, params)
        print(json.dumps(line), flush=True)
    for dd in {id(x): x for x in (den, den_o, den_first)}.values():
        dd.free()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
