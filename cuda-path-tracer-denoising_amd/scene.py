"""Scene text format of the reference and a primitive-driven input producer.

The reference describes its scenes in a small text format (reference src/scene.cpp:9-46 dispatch, :47-117 OBJECT,
:119-177 CAMERA, materials after that; e.g. scenes/cornell.txt):

    MATERIAL <id>          RGB r g b / SPECEX / SPECRGB / REFL / REFR / REFRIOR / EMITTANCE [/ TEXTURE file]
    CAMERA                 RES w h / FOVY deg / FILE name            then EYE / LOOKAT / UP until a blank line
    OBJECT <id>            cube | sphere | mesh / material <id> / TRANS / ROTAT / SCALE [/ mesh file]

`parse_scene` reads that format; `geom_array` turns the cube / sphere objects into the flat primitive records the device
producer takes (`SvgfSceneGeom`, include/svgf.h): transform = T * Rx * Ry * Rz * S as in
utilityCore::buildTransformationMatrix (src/utilities.cpp:65-72), its inverse and inverse-transpose.  Triangle meshes
belong to the path tracer's BVH machinery, which is out of scope: mesh objects are listed in `Scene.skipped`.

`render_scene` is the numpy ORACLE of csrc/svgf_scene.hip: a primary-ray cast against the primitives (unit cube
[-0.5,0.5]^3 / unit sphere r = 0.5 in object space, the reference's conventions, src/intersections.h:50,104) that fills
the denoiser's inputs exactly as the reference's first bounce does (src/pathtrace.cu:317-323), with the same shading /
noise stub as synth.py.  The device kernel mirrors it operation for operation in fp32.
"""
from __future__ import annotations

import dataclasses

import numpy as np

from . import synth

F = np.float32

SCENE_GEOM_DTYPE = np.dtype([("type", "<i4"), ("material", "<i4"), ("albedo", "<f4", 3), ("emittance", "<f4"),
                             ("xf", "<f4", 12), ("inv", "<f4", 12), ("invT", "<f4", 9)])
assert SCENE_GEOM_DTYPE.itemsize == 156
CUBE, SPHERE = 0, 1
MAX_GEOMS = 64


@dataclasses.dataclass
class Scene:
    materials: dict
    objects: list
    camera: dict
    skipped: list


def parse_scene(text: str) -> Scene:
    """Tokenises like the reference: whitespace-separated tokens per line, sections end at a blank line, anything that
    is not a section keyword at the start of a line (comments, stray text) is ignored."""
    lines = [ln.strip() for ln in text.replace("\r\n", "\n").replace("\r", "\n").split("\n")]
    materials, objects, camera, skipped = {}, [], {}, []
    i = 0

    def section(start):
        j = start
        out = []
        while j < len(lines) and lines[j] != "":
            out.append(lines[j].split())
            j += 1
        return out, j

    while i < len(lines):
        tok = lines[i].split()
        if not tok:
            i += 1
            continue
        if tok[0] == "MATERIAL":
            body, i = section(i + 1)
            m = {"rgb": (0.0, 0.0, 0.0), "emittance": 0.0}
            for t in body:
                if t[0] == "RGB":
                    m["rgb"] = tuple(float(v) for v in t[1:4])
                elif t[0] == "EMITTANCE":
                    m["emittance"] = float(t[1])
                elif t[0] in ("SPECEX", "REFL", "REFR", "REFRIOR"):
                    m[t[0].lower()] = float(t[1])
                elif t[0] == "SPECRGB":
                    m["specrgb"] = tuple(float(v) for v in t[1:4])
                elif t[0] == "TEXTURE":
                    m["texture"] = t[1]
            materials[int(tok[1])] = m
        elif tok[0] == "OBJECT":
            body, i = section(i + 1)
            o = {"id": int(tok[1]), "type": body[0][0], "material": int(body[1][1]),
                 "trans": (0.0, 0.0, 0.0), "rotat": (0.0, 0.0, 0.0), "scale": (1.0, 1.0, 1.0)}
            for t in body[2:]:
                if t[0] in ("TRANS", "ROTAT", "SCALE"):
                    o[t[0].lower()] = tuple(float(v) for v in t[1:4])
                elif o["type"] == "mesh":
                    o["file"] = t[0]
            if o["id"] != len(objects):
                raise ValueError(f"OBJECT {o['id']} does not match the expected number of objects ({len(objects)})")
            objects.append(o)
        elif tok[0] == "CAMERA":
            body, i = section(i + 1)
            for t in body:
                if t[0] == "RES":
                    camera["res"] = (int(t[1]), int(t[2]))
                elif t[0] == "FOVY":
                    camera["fovy"] = float(t[1])
                elif t[0] == "FILE":
                    camera["file"] = t[1]
                elif t[0] in ("EYE", "LOOKAT", "UP"):
                    camera[t[0].lower()] = tuple(float(v) for v in t[1:4])
        else:
            i += 1
    for o in objects:
        if o["type"] not in ("cube", "sphere"):
            skipped.append(o["id"])
    return Scene(materials, objects, camera, skipped)


def _transform(trans, rotat, scale):
    """T * Rx * Ry * Rz * S in fp32 with a fixed operation order (3x4, row-major), its inverse
    S^-1 * Rz^T * Ry^T * Rx^T * T^-1, and the transpose of the inverse's 3x3 block."""
    PI = F(3.14159265358979323846)
    ax, ay, az = (F(v) * PI / F(180) for v in rotat)
    cx, sx, cy, sy, cz, sz = np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay), np.cos(az), np.sin(az)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]], dtype=F)
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], dtype=F)
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]], dtype=F)

    def mm(a, b):       # 3x3 product, each element ((a0*b0 + a1*b1) + a2*b2) in fp32
        out = np.zeros((3, 3), dtype=F)
        for r in range(3):
            for c in range(3):
                out[r, c] = F(F(a[r, 0] * b[0, c] + a[r, 1] * b[1, c]) + a[r, 2] * b[2, c])
        return out

    Rm = mm(mm(Rx, Ry), Rz)
    s = np.array(scale, dtype=F)
    t = np.array(trans, dtype=F)
    M = (Rm * s[None, :]).astype(F)                                 # R * diag(s)
    xf = np.concatenate([M, t[:, None]], axis=1).astype(F)         # 3x4
    Mi = (Rm.T / s[:, None]).astype(F)                              # diag(1/s) * R^T
    ti = np.zeros(3, dtype=F)
    for r in range(3):
        ti[r] = -F(F(Mi[r, 0] * t[0] + Mi[r, 1] * t[1]) + Mi[r, 2] * t[2])
    inv = np.concatenate([Mi, ti[:, None]], axis=1).astype(F)
    invT = Mi.T.copy()
    return xf, inv, invT


def geom_array(scene: Scene) -> np.ndarray:
    rec = []
    for o in scene.objects:
        if o["type"] not in ("cube", "sphere"):
            continue
        m = scene.materials[o["material"]]
        xf, inv, invT = _transform(o["trans"], o["rotat"], o["scale"])
        g = np.zeros((), dtype=SCENE_GEOM_DTYPE)
        g["type"] = CUBE if o["type"] == "cube" else SPHERE
        g["material"] = o["material"]
        g["albedo"] = np.array(m["rgb"], dtype=F)
        g["emittance"] = F(m["emittance"])
        g["xf"], g["inv"], g["invT"] = xf.reshape(-1), inv.reshape(-1), invT.reshape(-1)
        rec.append(g)
    if len(rec) > MAX_GEOMS:
        raise ValueError(f"{len(rec)} primitives, at most {MAX_GEOMS}")
    return np.array(rec, dtype=SCENE_GEOM_DTYPE)


def camera_for_frame(scene: Scene, frame: int, moving: bool):
    """The reference's per-frame camera (src/main.cpp:154-190).  Static: the scene's EYE / LOOKAT with the basis runCuda
    builds (view = -normalize(EYE - LOOKAT), right = cross(view, +y) not normalised, up = cross(right, view)).
    Moving (ui_automate_camera): zoom = |EYE - LOOKAT| and the automation's own orbit angles and look-at point, which
    replace the scene's LOOKAT (src/main.cpp:163-167)."""
    eye, look = np.array(scene.camera["eye"], dtype=F), np.array(scene.camera["lookat"], dtype=F)
    fovy = scene.camera.get("fovy", 45.0)
    offset = (eye - look).astype(F)
    zoom = np.sqrt(np.sum(offset * offset, dtype=F))
    if moving:
        return synth.camera_for_frame(frame, True, zoom=float(zoom), fovy_deg=fovy)
    view = (-offset / zoom).astype(F)
    right = np.cross(view, np.array([0, 1, 0], dtype=F)).astype(F)
    up = np.cross(right, view).astype(F)
    return dict(right=right, up=up, view=view, position=(offset + look).astype(F), fovy_deg=fovy)


def light_position(geoms: np.ndarray):
    """The shading stub's point light: centre of the first emissive primitive, else the synthetic scene's light."""
    for g in geoms:
        if g["emittance"] > 0:
            return np.array([g["xf"][3], g["xf"][7], g["xf"][11]], dtype=F)
    return np.array([0.0, 9.5, 0.0], dtype=F)


def render_scene(W: int, H: int, frame: int, geoms: np.ndarray, cam: dict, seed: int = 1, noise: float = 0.6,
                 fireflies: float = 0.02):
    """numpy oracle of svgf_scene_render: (color float32[H,W,3], gbuffer GBUFFER_DTYPE[H,W])."""
    plx, ply = synth._pixel_length(W, H, cam.get("fovy_deg", 45.0))
    xs = (np.arange(W, dtype=F) - F(W * 0.5 - 0.5))[None, :]
    ys = (np.arange(H, dtype=F) - F(H * 0.5 - 0.5))[:, None]
    d = (cam["view"][None, None, :] - cam["right"][None, None, :] * (plx * xs)[..., None]
         - cam["up"][None, None, :] * (ply * ys)[..., None]).astype(F)
    d /= np.sqrt(np.sum(d * d, axis=-1, keepdims=True, dtype=F))
    o = cam["position"].astype(F)
    inf = F(np.inf)

    t_best = np.full((H, W), inf, dtype=F)
    gid = np.full((H, W), -1, dtype=np.int32)
    nrm = np.zeros((H, W, 3), dtype=F)
    pos_hit = np.zeros((H, W, 3), dtype=F)

    def apply(m12, v, w):       # 3x4 row-major times (v, w): ((m0*v0 + m1*v1) + m2*v2) + m3*w
        m = m12.reshape(3, 4)
        return np.stack([((m[r, 0] * v[..., 0] + m[r, 1] * v[..., 1]).astype(F) + m[r, 2] * v[..., 2]).astype(F) + m[r, 3] * F(w)
                         for r in range(3)], axis=-1).astype(F)

    def normalise(v):
        return (v / np.sqrt(np.sum(v * v, axis=-1, keepdims=True, dtype=F))).astype(F)

    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        for k, g in enumerate(geoms):
            qo = apply(g["inv"], np.broadcast_to(o, (H, W, 3)), 1.0)
            qd = normalise(apply(g["inv"], d, 0.0))
            if g["type"] == CUBE:
                t1 = ((F(-0.5) - qo) / qd).astype(F)
                t2 = ((F(0.5) - qo) / qd).astype(F)
                ta, tb = np.minimum(t1, t2), np.maximum(t1, t2)
                tmin = np.full((H, W), F(-1e38), dtype=F)
                tmax = np.full((H, W), F(1e38), dtype=F)
                amin = np.zeros((H, W), dtype=np.int32)
                amax = np.zeros((H, W), dtype=np.int32)
                for ax in range(3):
                    up = (ta[..., ax] > 0) & (ta[..., ax] > tmin)
                    tmin = np.where(up, ta[..., ax], tmin)
                    amin = np.where(up, ax, amin)
                    dn = tb[..., ax] < tmax
                    tmax = np.where(dn, tb[..., ax], tmax)
                    amax = np.where(dn, ax, amax)
                hit = (tmax >= tmin) & (tmax > 0)
                inside = tmin <= 0
                tt = np.where(inside, tmax, tmin).astype(F)
                axis = np.where(inside, amax, amin)
                sgn = np.where(np.take_along_axis(qd, axis[..., None], axis=-1)[..., 0] < 0, F(1), F(-1)).astype(F)   # faces the ray
                n_obj = np.zeros((H, W, 3), dtype=F)
                np.put_along_axis(n_obj, axis[..., None], sgn[..., None], axis=-1)
                p_obj = (qo + tt[..., None] * qd).astype(F)
                n_w = normalise(apply(g["xf"], n_obj, 0.0))
            else:
                b = np.sum(qo * qd, axis=-1, dtype=F)
                rad = (b * b - (np.sum(qo * qo, axis=-1, dtype=F) - F(0.25))).astype(F)
                sq = np.sqrt(np.maximum(rad, F(0)))
                ta_, tb_ = (-b + sq).astype(F), (-b - sq).astype(F)
                both_pos = (ta_ > 0) & (tb_ > 0)
                both_neg = (ta_ < 0) & (tb_ < 0)
                tt = np.where(both_pos, np.minimum(ta_, tb_), np.maximum(ta_, tb_)).astype(F)
                hit = (rad >= 0) & ~both_neg
                p_obj = (qo + tt[..., None] * qd).astype(F)
                m = g["invT"].reshape(3, 3)
                n_w = normalise(np.stack([((m[r, 0] * p_obj[..., 0] + m[r, 1] * p_obj[..., 1]).astype(F)
                                           + m[r, 2] * p_obj[..., 2]).astype(F) for r in range(3)], axis=-1))
            p_w = apply(g["xf"], p_obj, 1.0)
            dv = (o[None, None, :] - p_w).astype(F)
            t_w = np.sqrt(np.sum(dv * dv, axis=-1, dtype=F))
            take = hit & (t_w > F(1e-4)) & (t_w < t_best)
            t_best = np.where(take, t_w, t_best)
            gid = np.where(take, np.int32(k), gid)
            nrm = np.where(take[..., None], n_w, nrm)
            pos_hit = np.where(take[..., None], p_w, pos_hit)

    miss = gid < 0
    pos = np.where(miss[..., None], (o[None, None, :] + F(-1.0) * d).astype(F), pos_hit).astype(F)
    alb = np.zeros((H, W, 3), dtype=F)
    emit = np.zeros((H, W), dtype=F)
    for k, g in enumerate(geoms):
        alb = np.where((gid == k)[..., None], g["albedo"], alb)
        emit = np.where(gid == k, g["emittance"], emit)
    light = light_position(geoms)
    tl = light[None, None, :] - pos
    dist2 = np.sum(tl * tl, axis=-1, dtype=F)
    with np.errstate(divide="ignore", invalid="ignore"):
        ldir = tl / np.sqrt(dist2)[..., None]
    lam = np.maximum(np.sum(ldir * nrm, axis=-1, dtype=F), F(0))
    shade = (F(0.15) + F(30.0) * lam / (F(4.0) + dist2)).astype(F)
    shade = np.where(emit > 0, emit, shade).astype(F)               # emitters show their emittance

    u = synth.hash_uniform(seed, frame, W * H, 0).reshape(H, W)
    v = synth.hash_uniform(seed, frame, W * H, 1).reshape(H, W)
    rc = np.stack([synth.hash_uniform(seed, frame, W * H, 2 + c).reshape(H, W) for c in range(3)], axis=-1)
    mult = (F(1.0) + F(noise) * (F(2.0) * u - F(1.0))).astype(F)
    mult = np.where(v < F(fireflies), mult * F(6.0), mult).astype(F)
    chroma = (F(1.0) + F(F(0.1) * F(noise)) * (rc - F(0.5))).astype(F)
    color = (alb * shade[..., None] * mult[..., None] * chroma).astype(F)
    color = np.where(miss[..., None], F(0), color).astype(F)

    gb = np.zeros((H, W), dtype=synth.GBUFFER_DTYPE)
    gb["normal"] = nrm
    gb["position"] = pos
    gb["albedo"] = alb
    gb["ialbedo"] = F(1.0)
    gb["geomId"] = gid
    return np.ascontiguousarray(color), gb
