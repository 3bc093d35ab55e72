"""Seeded synthetic 1-spp colour + G-buffer frames for tests and bench (numpy, host side).

The reference's input producer is its path tracer (reference src/pathtrace.cu:300-401, G-buffer fill :317-323),
which is out of scope.  On the GPU box the reference scenes do not exist, so inputs come from
this analytic Cornell-like ray-cast: 5 walls + 2 spheres + 1 box, `geomId` per object, a miss gives
geomId = -1 with position = eye - dir (t = -1, reference src/pathtrace.cu:318), albedo per object (checker on the
floor), ialbedo = 1, colour = Lambert shading x albedo x multiplicative noise with a few fireflies.

Camera model is the reference's:
  basis      src/main.cpp:171-190   (view = -normalize(offset), right = cross(view, +y) NOT normalised, up = cross(right, view))
  automation src/main.cpp:156-169   (lookAt / theta / phi sinusoids, each phase advanced by its speed per frame)
  rays       src/pathtrace.cu:199-202 with pixelLength of src/scene.cpp:159-166 (FOVY used as the HALF angle)
"""
from __future__ import annotations

import numpy as np

GBUFFER_DTYPE = np.dtype([("normal", "<f4", 3), ("position", "<f4", 3), ("albedo", "<f4", 3),
                          ("ialbedo", "<f4", 3), ("geomId", "<i4")])
assert GBUFFER_DTYPE.itemsize == 52

F = np.float32
PI = F(3.14159265358979323846)

# camera speeds used for "moving camera" sequences (the reference defaults are 0)
MOVING_SPEEDS = dict(x=0.02, y=0.01, z=0.01, theta=0.01, phi=0.02)


def camera_for_frame(frame: int, moving: bool, zoom: float = 10.5, fovy_deg: float = 45.0):
    """Returns dict(right, up, view, position) float32[3] for `frame` (0-based) — what runCuda() leaves in
    scene->state.camera before pathtrace() is called (src/main.cpp:154-190)."""
    lookat = np.array([0.0, 5.0, 0.0], dtype=F)
    theta = F(PI * F(0.5))
    phi = F(0.0)
    if moving:
        # the phases are ACCUMULATED in fp32, one addition per frame, and advanced BEFORE use (src/main.cpp:158-162:
        # `camera_tx += ui_camera_speed_x` ...), not computed as speed * (frame + 1)
        tx = ty = tz = tt = tp = F(0.0)
        for _ in range(frame + 1):
            tx = F(tx + F(MOVING_SPEEDS["x"])); ty = F(ty + F(MOVING_SPEEDS["y"])); tz = F(tz + F(MOVING_SPEEDS["z"]))
            tt = F(tt + F(MOVING_SPEEDS["theta"])); tp = F(tp + F(MOVING_SPEEDS["phi"]))
        lookat = np.array([F(2.0) * np.sin(tx), F(5.0) + np.sin(ty), F(1.5) * np.sin(tz)], dtype=F)
        theta = F(PI * F(0.5) + PI / F(18) * np.sin(tt))
        phi = F(PI / F(12) * np.sin(tp))
    z = F(zoom)
    offset = np.array([z * np.sin(phi) * np.sin(theta), z * np.cos(theta), z * np.cos(phi) * np.sin(theta)], dtype=F)
    view = (-offset / np.sqrt(np.sum(offset * offset, dtype=F))).astype(F)
    right = np.cross(view, np.array([0, 1, 0], dtype=F)).astype(F)
    up = np.cross(right, view).astype(F)
    position = (offset + lookat).astype(F)
    return dict(right=right, up=up, view=view, position=position, fovy_deg=fovy_deg)


def _pixel_length(W: int, H: int, fovy_deg: float):
    yscaled = F(np.tan(F(fovy_deg) * (PI / F(180))))
    xscaled = F(yscaled * F(W) / F(H))
    return F(2) * xscaled / F(W), F(2) * yscaled / F(H)


# scene: axis-aligned room x in [-5,5], y in [0,10], z in [-5,5] (open towards +z), two spheres, one box
_SPHERES = [(np.array([3.0, 2.0, 1.0], F), F(1.5), 6), (np.array([-2.0, 1.0, 3.0], F), F(1.0), 8)]
_BOX = (np.array([-2.5, 0.0, -2.5], F), np.array([0.5, 4.0, 0.5], F), 7)
_ALBEDO = {0: (0.85, 0.85, 0.85), 1: (0.85, 0.85, 0.85), 2: (0.85, 0.85, 0.85), 3: (0.85, 0.35, 0.35),
           4: (0.35, 0.85, 0.35), 6: (0.9, 0.9, 0.2), 7: (0.3, 0.5, 0.9), 8: (0.9, 0.6, 0.3)}
_LIGHT = np.array([0.0, 9.5, 0.0], F)


def hash_uniform(seed: int, frame: int, n: int, k: int):
    """Counter-based uniform float32 in [0,1) per pixel index 0..n-1 and stream k: a 32-bit integer mix of
    (seed, frame, pixel, k), top 24 bits scaled by 2^-24.  Plain uint32 arithmetic, so the device-side producer
    (csrc/svgf_synth.hip: synth_hash) reproduces it bit for bit."""
    with np.errstate(over="ignore"):
        x = (np.arange(n, dtype=np.uint32) * np.uint32(0x9E3779B1) + np.uint32((k * 0x85EBCA77) & 0xFFFFFFFF)
             + np.uint32((frame * 0xC2B2AE3D) & 0xFFFFFFFF) + np.uint32((seed * 0x27D4EB2F) & 0xFFFFFFFF))
        x ^= x >> np.uint32(15)
        x *= np.uint32(0x2C1B3C6D)
        x ^= x >> np.uint32(12)
        x *= np.uint32(0x297A2D39)
        x ^= x >> np.uint32(15)
    return ((x >> np.uint32(8)).astype(np.float32) * F(1.0 / 16777216.0)).astype(F)


def scene_constants():
    """The analytic scene as flat float32 arrays, in the layout csrc/svgf_synth.hip takes (SvgfSynthScene)."""
    sph = []
    for c, r, g in _SPHERES:
        sph.append(dict(c=c, r=r, gid=g))
    return dict(spheres=sph, box_min=_BOX[0], box_max=_BOX[1], box_gid=_BOX[2], light=_LIGHT, albedo=_ALBEDO)


def render_frame(W: int, H: int, frame: int, seed: int = 1, moving: bool = False, noise: float = 0.6,
                 fireflies: float = 0.02, cam: dict | None = None, noise_model: str = "pcg64"):
    """Returns (color float32[H,W,3], gbuffer GBUFFER_DTYPE[H,W], camera dict).
    noise_model "pcg64": numpy's default generator (the model the committed goldens were made with);
                "hash" : hash_uniform(), the model the device-side producer implements (tests/test_synth_device_gpu.py)."""
    if cam is None:
        cam = camera_for_frame(frame, moving)
    plx, ply = _pixel_length(W, H, cam["fovy_deg"])
    xs = (np.arange(W, dtype=F) - F(W * 0.5 - 0.5))[None, :]
    ys = (np.arange(H, dtype=F) - F(H * 0.5 - 0.5))[:, None]
    d = (cam["view"][None, None, :] - cam["right"][None, None, :] * (plx * xs)[..., None]
         - cam["up"][None, None, :] * (ply * ys)[..., None]).astype(F)
    d /= np.sqrt(np.sum(d * d, axis=-1, keepdims=True, dtype=F))
    o = cam["position"].astype(F)

    t_best = np.full((H, W), np.inf, dtype=F)
    gid = np.full((H, W), -1, dtype=np.int32)
    nrm = np.zeros((H, W, 3), dtype=F)

    def consider(t, g, n):
        nonlocal t_best, gid, nrm
        hit = (t > F(1e-4)) & (t < t_best)
        t_best = np.where(hit, t, t_best)
        gid = np.where(hit, np.int32(g), gid)
        nrm = np.where(hit[..., None], n, nrm)

    with np.errstate(divide="ignore", invalid="ignore"):
        # walls: (axis, coordinate, inward normal, geomId, bounds on the other two axes)
        walls = [(1, 0.0, (0, 1, 0), 0), (1, 10.0, (0, -1, 0), 1), (2, -5.0, (0, 0, 1), 2),
                 (0, -5.0, (1, 0, 0), 3), (0, 5.0, (-1, 0, 0), 4)]
        for axis, coord, n, g in walls:
            t = (F(coord) - o[axis]) / d[..., axis]
            ph = o[None, None, :] + t[..., None] * d
            inside = ((ph[..., 0] >= -5.001) & (ph[..., 0] <= 5.001) & (ph[..., 1] >= -0.001) & (ph[..., 1] <= 10.001)
                      & (ph[..., 2] >= -5.001) & (ph[..., 2] <= 5.001))
            consider(np.where(inside, t, np.inf).astype(F), g, np.broadcast_to(np.array(n, F), (H, W, 3)))
        for c, r, g in _SPHERES:
            oc = o - c
            b = np.sum(d * oc[None, None, :], axis=-1, dtype=F)
            cc = F(F(oc[0] * oc[0] + oc[1] * oc[1]) + oc[2] * oc[2]) - r * r      # fixed order (the device producer mirrors it)
            disc = b * b - cc
            t = np.where(disc > 0, -b - np.sqrt(np.maximum(disc, 0)), np.inf).astype(F)
            ph = o[None, None, :] + t[..., None] * d
            consider(t, g, ((ph - c[None, None, :]) / r).astype(F))
        bmin, bmax, g = _BOX
        t0 = (bmin[None, None, :] - o[None, None, :]) / d
        t1 = (bmax[None, None, :] - o[None, None, :]) / d
        tn = np.minimum(t0, t1)
        tf = np.maximum(t0, t1)
        tnear = np.max(tn, axis=-1)
        tfar = np.min(tf, axis=-1)
        hitbox = (tnear <= tfar) & (tnear > 0)
        ax = np.argmax(tn, axis=-1)
        nb = np.zeros((H, W, 3), dtype=F)
        sgn = -np.sign(np.take_along_axis(d, ax[..., None], axis=-1))[..., 0]
        np.put_along_axis(nb, ax[..., None], sgn[..., None].astype(F), axis=-1)
        consider(np.where(hitbox, tnear, np.inf).astype(F), g, nb)

    miss = gid < 0
    t_used = np.where(miss, F(-1.0), t_best).astype(F)
    pos = (o[None, None, :] + t_used[..., None] * d).astype(F)

    alb = np.zeros((H, W, 3), dtype=F)
    for g, a in _ALBEDO.items():
        alb = np.where((gid == g)[..., None], np.array(a, F), alb)
    checker = ((np.floor(pos[..., 0]) + np.floor(pos[..., 2])).astype(np.int64) & 1).astype(F)
    alb = np.where((gid == 0)[..., None], alb * (F(0.55) + F(0.45) * checker)[..., None], alb).astype(F)

    tl = _LIGHT[None, None, :] - pos
    dist2 = np.sum(tl * tl, axis=-1, dtype=F)
    ldir = tl / np.sqrt(dist2)[..., None]
    lam = np.maximum(np.sum(ldir * nrm, axis=-1, dtype=F), F(0))
    shade = (F(0.15) + F(30.0) * lam / (F(4.0) + dist2)).astype(F)

    if noise_model == "hash":
        u = hash_uniform(seed, frame, W * H, 0).reshape(H, W)
        v = hash_uniform(seed, frame, W * H, 1).reshape(H, W)
        rc = np.stack([hash_uniform(seed, frame, W * H, 2 + c).reshape(H, W) for c in range(3)], axis=-1)
    elif noise_model == "pcg64":
        rng = np.random.default_rng([int(seed), int(frame), W, H])
        u = rng.random((H, W), dtype=np.float32)
        v = rng.random((H, W), dtype=np.float32)
        rc = rng.random((H, W, 3), dtype=np.float32)
    else:
        raise ValueError(noise_model)
    mult = (F(1.0) + F(noise) * (F(2.0) * u - F(1.0))).astype(F)
    mult = np.where(v < F(fireflies), mult * F(6.0), mult).astype(F)
    chroma = (F(1.0) + F(F(0.1) * F(noise)) * (rc - F(0.5))).astype(F)      # amplitude as an fp32 product (device mirrors it)
    color = (alb * shade[..., None] * mult[..., None] * chroma).astype(F)
    color = np.where(miss[..., None], F(0), color).astype(F)

    gb = np.zeros((H, W), dtype=GBUFFER_DTYPE)
    gb["normal"] = nrm
    gb["position"] = pos
    gb["albedo"] = alb
    gb["ialbedo"] = F(1.0)
    gb["geomId"] = gid
    return np.ascontiguousarray(color), gb, cam


def random_frame(W: int, H: int, seed: int = 0):
    """Unstructured random colour/G-buffer (piecewise-constant geomId patches) for kernel stress tests."""
    rng = np.random.default_rng(seed)
    color = (rng.random((H, W, 3), dtype=np.float32) * F(2.0)).astype(F)
    gb = np.zeros((H, W), dtype=GBUFFER_DTYPE)
    patch = max(1, min(W, H) // 6)
    ids = rng.integers(-1, 6, size=((H + patch - 1) // patch, (W + patch - 1) // patch), dtype=np.int32)
    gid = np.kron(ids, np.ones((patch, patch), dtype=np.int32))[:H, :W]
    n = rng.normal(size=(6 + 1, 3)).astype(F)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    gb["normal"] = n[gid + 1] + (rng.normal(size=(H, W, 3)).astype(F) * F(0.02))
    yy, xx = np.mgrid[0:H, 0:W].astype(F)
    gb["position"] = np.stack([xx * F(0.05), yy * F(0.05), gid.astype(F) * F(0.7)], axis=-1) + \
        rng.normal(size=(H, W, 3)).astype(F) * F(0.01)
    gb["albedo"] = rng.random((H, W, 3), dtype=np.float32)
    gb["ialbedo"] = F(1.0)
    gb["geomId"] = gid
    return np.ascontiguousarray(color), gb
