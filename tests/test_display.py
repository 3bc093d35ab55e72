"""display pack (reference sendTwoImagesToPBO, src/pathtrace.cu:45-77) and PNG save
(reference saveImage + image::savePNG, src/main.cpp:131-152, src/image.cpp:22-39).  Byte-exact against numpy
restatements of those few lines."""
import struct
import zlib

import numpy as np
import pytest


def read_png_rgb8(path):
    """Tiny PNG reader for what svgf_save_png writes (8-bit RGB, filter 0, one IDAT); checks every CRC."""
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    off, chunks = 8, []
    while off < len(b):
        n, tag = struct.unpack(">I4s", b[off:off + 8])
        data = b[off + 8:off + 8 + n]
        (crc,) = struct.unpack(">I", b[off + 8 + n:off + 12 + n])
        assert crc == (zlib.crc32(tag + data) & 0xFFFFFFFF), tag
        chunks.append((tag, data))
        off += 12 + n
    assert [t for t, _ in chunks] == [b"IHDR", b"IDAT", b"IEND"]
    w, h, depth, ctype, comp, flt, inter = struct.unpack(">IIBBBBB", chunks[0][1])
    assert (depth, ctype, comp, flt, inter) == (8, 2, 0, 0, 0)
    raw = np.frombuffer(zlib.decompress(chunks[1][1]), dtype=np.uint8).reshape(h, 1 + 3 * w)   # also checks Adler-32
    assert np.all(raw[:, 0] == 0)
    return raw[:, 1:].reshape(h, w, 3)


def save_png_oracle(rgb, mirror_x):
    v = np.clip(rgb, np.float32(0), np.float32(1))
    v = np.where(np.isnan(v), np.float32(0), v).astype(np.float32)
    out = (v * np.float32(255.0)).astype(np.uint8)            # truncation, as (unsigned char) pix.x
    return out[:, ::-1] if mirror_x else out


@pytest.mark.parametrize("w,h", [(1, 1), (5, 3), (257, 131), (640, 360)])
def test_save_png_matches_reference_conversion(pkg, tmp_path, w, h):
    rng = np.random.default_rng(w * 1000 + h)
    img = (rng.random((h, w, 3), dtype=np.float32) * 1.6 - 0.3).astype(np.float32)     # values below 0 and above 1 too
    img.flat[0] = np.nan
    if img.size > 4:
        img.flat[3] = np.inf
        img.flat[4] = -np.inf
    for mirror in (True, False):
        path = str(tmp_path / f"t_{w}x{h}_{int(mirror)}.png")
        pkg.binding.save_png(path, img, mirror_x=mirror)
        got = read_png_rgb8(path)
        assert np.array_equal(got, save_png_oracle(img, mirror))
    lib = pkg.load_library()
    assert lib.svgf_save_png(None, None, 4, 4, 0) == -1
    assert lib.svgf_display_pack(0, None, None, None, 4, 4, None) == -1


def pack_oracle(left, right):
    def conv(a):
        with np.errstate(invalid="ignore", over="ignore"):
            d = a.astype(np.float64) * 255.0
            i = np.where(np.isnan(d), 0.0, np.clip(d, -1e9, 1e9)).astype(np.int64)     # saturating cast, NaN -> 0
        return np.clip(i, 0, 255).astype(np.uint8)
    h, w = left.shape[:2]
    out = np.zeros((h, 2 * w, 4), dtype=np.uint8)
    out[:, :w, :3] = conv(left)
    out[:, w:, :3] = conv(right)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("w,h", [(5, 3), (257, 131), (1920, 1080)])
def test_display_pack_matches_reference_conversion(pkg, w, h):
    import torch
    rng = np.random.default_rng(w + h)
    left = (rng.random((h, w, 3), dtype=np.float32) * 1.6 - 0.3).astype(np.float32)
    right = (rng.random((h, w, 3), dtype=np.float32) * 3.0).astype(np.float32)
    left.flat[0], left.flat[1], left.flat[2] = np.nan, np.inf, -np.inf
    right.flat[5] = np.float32(1.0 / 255.0 * 37.0)          # values sitting on byte boundaries
    right.flat[6] = np.float32(1.0)
    pbo = torch.zeros((h, 2 * w, 4), dtype=torch.uint8, device="cuda")
    pkg.binding.display_pack(pbo, torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda(), w, h)
    torch.cuda.synchronize()
    assert np.array_equal(pbo.cpu().numpy(), pack_oracle(left, right))


@pytest.mark.gpu
def test_producer_denoise_display_save_chain(pkg, tmp_path):
    """The whole on-device chain of the reference's frame: produce -> denoise -> pack for display; then save."""
    import torch
    W, H = 320, 180
    den = pkg.Denoiser(W, H)
    params = pkg.reference_defaults().set(temporal_enable=1, spatial_enable=1, atrous_nlevel=5, history_level=1)
    rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    gb = torch.empty((H * W * 52,), dtype=torch.uint8, device="cuda")
    out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    pbo = torch.empty((H, 2 * W, 4), dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream()
    for f in range(6):
        cam = pkg.synth.camera_for_frame(f, False)
        pkg.binding.synth_render(rgb, gb, W, H, cam, f, seed=3, stream=s)
        den.denoise(out, rgb, gb, cam, params, stream=s)
        pkg.binding.display_pack(pbo, rgb, out, W, H, stream=s)
    torch.cuda.synchronize()
    side = pbo.cpu().numpy()
    assert np.array_equal(side, pack_oracle(rgb.cpu().numpy(), out.cpu().numpy()))
    noisy, clean = side[:, :W, :3].astype(np.float32), side[:, W:, :3].astype(np.float32)
    # the denoised half is smoother than the 1-spp half (mean absolute horizontal gradient)
    assert np.abs(np.diff(clean, axis=1)).mean() < 0.6 * np.abs(np.diff(noisy, axis=1)).mean()
    path = str(tmp_path / "frame.png")
    pkg.binding.save_png(path, out.cpu().numpy())
    assert read_png_rgb8(path).shape == (H, W, 3)
    den.free()
