"""Thread scaling of the CPU oracle (test infrastructure; what bench.py's cpu_baseline leg times) on this host: Mpixels/s at 1920x1080 full SVGF.
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); orc = ge.load_oracle()
W, H = 1920, 1080
frames = [pkg.synth.render_frame(W, H, f, seed=3, moving=False, noise_model="hash") for f in range(2)]
p = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1)
print("OMP_PLACES", os.environ.get("OMP_PLACES"), "OMP_PROC_BIND", os.environ.get("OMP_PROC_BIND"), "affinity", len(os.sched_getaffinity(0)))
for t in [int(x) for x in sys.argv[1:]]:
    o = orc.Oracle(pkg, W, H, threads=t)
    ts = []
    for f in range(3):
        t0 = time.perf_counter(); o.denoise(*frames[f % 2], p); ts.append(time.perf_counter() - t0)
    o.free()
    print(f"threads {t:4d}: {W*H/min(ts[1:])/1e6:7.3f} Mpix/s  ({min(ts[1:])*1e3:.0f} ms per frame)", flush=True)
