"""Does replaying a captured frame (hipGraph) shorten the kernel-boundary cost?  Timing only: a captured frame bakes the
plane rotation in, so replaying it is not a valid denoising sequence."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
W, H = 1920, 1080
c, g, cam = pkg.synth.render_frame(W, H, 0, seed=5, moving=False)
tin = torch.from_numpy(c).cuda(); tg = torch.from_numpy(g.view(np.uint8).reshape(-1)).cuda()
out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
for temporal in (1, 0):
    p = pkg.reference_defaults().set(temporal_enable=temporal, spatial_enable=1, atrous_nlevel=5, inputs_ready=0)
    d = pkg.Denoiser(W, H, 0)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(20):
            d.denoise(out, tin, tg, cam, p, stream=s)
    s.synchronize()
    def timed(fn, n=300):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s):
            e0.record(s)
            for _ in range(n): fn()
            e1.record(s)
        e1.synchronize()
        return e0.elapsed_time(e1) * 1000 / n
    direct = timed(lambda: d.denoise(out, tin, tg, cam, p, stream=s))
    try:
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            d.denoise(out, tin, tg, cam, p, stream=s)
        graph = timed(lambda: gr.replay())
        # ten frames per graph
        gr10 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr10, stream=s):
            for _ in range(10): d.denoise(out, tin, tg, cam, p, stream=s)
        graph10 = timed(lambda: gr10.replay(), 30) / 10
        print(f"temporal={temporal}: direct {direct:.1f} us/frame, graph replay {graph:.1f} us/frame, 10-frame graph {graph10:.1f} us/frame")
    except Exception as e:
        print(f"temporal={temporal}: direct {direct:.1f} us/frame; capture failed: {e!r}")
    d.free()
