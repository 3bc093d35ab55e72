import sys, time
sys.path.insert(0, '/root/repo')
import __graft_entry__ as ge
import torch
pkg = ge.load_package()
for (W, H) in ((1920, 1080), (3840, 2160), (800, 800), (1280, 720)):
    cam = pkg.synth.camera_for_frame(0, False)
    d_in = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    d_g = torch.empty((H * W * 52,), dtype=torch.uint8, device="cuda")
    pkg.binding.synth_render(d_in, d_g, W, H, cam, 0, seed=3, device=0)
    out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    cs = pkg.SvgfCamera.from_dict(cam)
    for nl in (1, 5):
        res = {}
        for v in (0, 4):
            d = pkg.Denoiser(W, H, 0)
            p = pkg.reference_defaults().set(temporal_enable=0, spatial_enable=1, atrous_nlevel=nl, kernel_variant=v)
            t = time.perf_counter()
            while time.perf_counter() - t < 0.5:
                for _ in range(32): d.denoise(out, d_in, d_g, cs, p)
                torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200): d.denoise(out, d_in, d_g, cs, p)
            e1.record(); torch.cuda.synchronize()
            res[v] = e0.elapsed_time(e1) / 200 * 1e3
            d.free()
        print(f"{W}x{H} non-temporal, {nl} level(s): default (prepare fused) {res[0]:.1f} us, kernel_variant 4 (prepare kernel) {res[4]:.1f} us", flush=True)
