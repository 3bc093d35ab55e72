"""Triangle meshes of the reference's scene format (the part scene.py skipped): OBJ loading the
way the reference's loader sees it (src/scene.cpp:234-311 through tinyobjloader: per-corner position / normal / texcoord,
polygons fanned into triangles, shapes in file order), the world-space triangle list Scene::loadMesh builds (positions
through `transform`, normals through `invTranspose`, NOT renormalised), and a numpy first-hit pass that adds those
triangles to scene.render_scene's primitives exactly as the reference's first bounce fills its G-buffer
(src/pathtrace.cu:211-276,316-323; glm::intersectRayTriangle, external/include/glm/gtx/intersect.inl:37-74;
Triangle::Intersect, src/sceneStructs.h:157-180 — note its normal weights (b.x, b.y, 1-b.x-b.y) against the uv weights
(1-b.x-b.y, b.x, b.y); Texture::getColor, src/sceneStructs.h:208-219).

Host-side Python only: it is the oracle / fixture checker of this row (tests/test_ref_scenes.py compares it with the
G-buffers the reference's own path tracer wrote); the device producer (csrc/svgf_scene.hip) still casts primitives only.
"""
from __future__ import annotations

import os

import numpy as np

from . import scene as scene_mod
from . import synth

F = np.float32


def load_obj(path: str):
    """Returns (pos[N,3,3], nrm[N,3,3] or None, uv[N,3,2] or None) per triangle, corners in face order."""
    v, vn, vt = [], [], []
    tp, tn, tu = [], [], []
    has_n = has_t = True
    for line in open(path, "r", errors="replace"):
        tok = line.split()
        if not tok:
            continue
        if tok[0] == "v":
            v.append([float(x) for x in tok[1:4]])
        elif tok[0] == "vn":
            vn.append([float(x) for x in tok[1:4]])
        elif tok[0] == "vt":
            vt.append([float(tok[1]), float(tok[2]) if len(tok) > 2 else 0.0])
        elif tok[0] == "f":
            corners = []
            for c in tok[1:]:
                parts = c.split("/")
                iv = int(parts[0])
                it = int(parts[1]) if len(parts) > 1 and parts[1] else 0
                inn = int(parts[2]) if len(parts) > 2 and parts[2] else 0
                corners.append((iv - 1 if iv > 0 else len(v) + iv, it - 1 if it > 0 else (len(vt) + it if it < 0 else -1),
                                inn - 1 if inn > 0 else (len(vn) + inn if inn < 0 else -1)))
            for k in range(2, len(corners)):                       # tinyobjloader: triangle fan (0, k-1, k)
                tri = (corners[0], corners[k - 1], corners[k])
                tp.append([v[c[0]] for c in tri])
                if all(c[2] >= 0 for c in tri):
                    tn.append([vn[c[2]] for c in tri])
                else:
                    has_n = False
                    tn.append([[0, 0, 0]] * 3)
                if all(c[1] >= 0 for c in tri):
                    tu.append([vt[c[1]] for c in tri])
                else:
                    has_t = False
                    tu.append([[0, 0]] * 3)
    pos = np.array(tp, dtype=F).reshape(-1, 3, 3)
    return pos, (np.array(tn, dtype=F).reshape(-1, 3, 3) if has_n and len(vn) else None), \
        (np.array(tu, dtype=F).reshape(-1, 3, 2) if has_t and len(vt) else None)


def scene_triangles(sc: scene_mod.Scene, models_dir: str):
    """World-space triangles of every mesh object, as Scene::loadMesh stores them; tri_obj = index of the OBJECT."""
    P, N, U, O = [], [], [], []
    for o in sc.objects:
        if o["type"] != "mesh":
            continue
        pos, nrm, uv = load_obj(os.path.join(models_dir, o["file"]))
        xf, inv, invT = scene_mod._transform(o["trans"], o["rotat"], o["scale"])
        M = xf.reshape(3, 4)
        wp = (pos @ M[:, :3].T + M[:, 3]).astype(F)
        wn = (nrm @ invT.reshape(3, 3).T).astype(F) if nrm is not None else np.zeros_like(pos)
        P.append(wp); N.append(wn); U.append(uv if uv is not None else np.zeros(pos.shape[:2] + (2,), F))
        O.append(np.full(len(pos), o["id"], np.int32))
    if not P:
        return np.zeros((0, 3, 3), F), np.zeros((0, 3, 3), F), np.zeros((0, 3, 2), F), np.zeros((0,), np.int32)
    return np.concatenate(P), np.concatenate(N), np.concatenate(U), np.concatenate(O)


def load_textures(sc: scene_mod.Scene, textures_dir: str):
    """{material id: uint8[h, w, 3]} for materials with a TEXTURE line (stb_image decodes JPEG like PIL does, +-1 level)."""
    from PIL import Image
    out = {}
    for mid, m in sc.materials.items():
        if "texture" in m:
            out[mid] = np.asarray(Image.open(os.path.join(textures_dir, m["texture"])).convert("RGB"), dtype=np.uint8)
    return out


def first_hit_gbuffer(W: int, H: int, sc: scene_mod.Scene, cam: dict, tris, textures=None):
    """G-buffer of the reference's first bounce (geomId = object index, -1 on a miss) for a scene with primitives AND
    meshes.  Normal / albedo of a MISS are left 0 here (the reference leaves stale values there, src/pathtrace.cu:267-272)."""
    prim_objs = [o["id"] for o in sc.objects if o["type"] in ("cube", "sphere")]
    geoms = scene_mod.geom_array(sc)
    _, gb = scene_mod.render_scene(W, H, 0, geoms, cam)
    o = cam["position"].astype(F)
    plx, ply = synth._pixel_length(W, H, cam.get("fovy_deg", 45.0))
    xs = (np.arange(W, dtype=F) - F(W * 0.5 - 0.5))[None, :]
    ys = (np.arange(H, dtype=F) - F(H * 0.5 - 0.5))[:, None]
    d = (cam["view"][None, None, :] - cam["right"][None, None, :] * (plx * xs)[..., None]
         - cam["up"][None, None, :] * (ply * ys)[..., None]).astype(F)
    d /= np.sqrt(np.sum(d * d, axis=-1, keepdims=True, dtype=F))
    gid = np.where(gb["geomId"] >= 0, np.array(prim_objs + [0], np.int32)[np.maximum(gb["geomId"], 0)], -1).astype(np.int32)
    hitp = gb["geomId"] >= 0
    dv = gb["position"] - o
    t_best = np.where(hitp, np.sqrt(np.sum(dv * dv, axis=-1, dtype=F)), F(np.inf)).astype(F)
    nrm, pos, alb = gb["normal"].copy(), gb["position"].copy(), gb["albedo"].copy()
    tp, tn, tu, tobj = tris
    if len(tp):
        dd = d.reshape(-1, 3)
        best_t = np.full(len(dd), np.inf, F)
        best_i = np.full(len(dd), -1, np.int64)
        best_b = np.zeros((len(dd), 2), F)
        eps = np.finfo(F).eps
        for i in range(len(tp)):                                   # glm::intersectRayTriangle, every triangle of the scene
            v0, v1, v2 = tp[i]
            e1, e2 = (v1 - v0).astype(F), (v2 - v0).astype(F)
            p = np.cross(dd, e2).astype(F)
            a = (p @ e1).astype(F)
            with np.errstate(divide="ignore", invalid="ignore"):
                f = (F(1) / a).astype(F)
                s = (o - v0).astype(F)
                bx = (f * (p @ s)).astype(F)
                q = np.cross(s, e1).astype(F)
                by = (f * (dd @ q)).astype(F)
                t = (f * F(e2 @ q)).astype(F)
            ok = (a >= eps) & (bx >= 0) & (bx <= 1) & (by >= 0) & (bx + by <= 1) & (t >= 0) & (t > 0) & (t < best_t)
            best_t = np.where(ok, t, best_t); best_i = np.where(ok, i, best_i)
            best_b = np.where(ok[:, None], np.stack([bx, by], -1), best_b)
        best_t, best_i, best_b = best_t.reshape(H, W), best_i.reshape(H, W), best_b.reshape(H, W, 2)
        take = (best_i >= 0) & (best_t < t_best)
        ti = np.maximum(best_i, 0)
        bx, by = best_b[..., 0:1], best_b[..., 1:2]
        n = (tn[ti][..., 0, :] * bx + tn[ti][..., 1, :] * by + tn[ti][..., 2, :] * (F(1) - bx - by)).astype(F)   # (sic) :168-170
        with np.errstate(divide="ignore", invalid="ignore"):
            n = (n / np.sqrt(np.sum(n * n, axis=-1, keepdims=True, dtype=F))).astype(F)
        uv = (tu[ti][..., 0, :] * (F(1) - bx - by) + tu[ti][..., 1, :] * bx + tu[ti][..., 2, :] * by).astype(F)
        gid = np.where(take, tobj[ti], gid)
        nrm = np.where(take[..., None], n, nrm)
        pos = np.where(take[..., None], (o + best_t[..., None] * d).astype(F), pos)
        mat = np.array([sc.objects[k]["material"] for k in range(len(sc.objects))], np.int32)
        a_mesh = np.zeros((H, W, 3), F)
        for k, ob in enumerate(sc.objects):
            if ob["type"] != "mesh":
                continue
            sel = take & (tobj[ti] == ob["id"])
            m = sc.materials[ob["material"]]
            if textures and ob["material"] in textures:
                tex = textures[ob["material"]]
                th, tw = tex.shape[:2]
                X = np.minimum(F(1) * tw * uv[..., 0], F(1) * tw - F(1)).astype(np.int64)
                Y = np.minimum(F(1) * th * (F(1) - uv[..., 1]), F(1) * th - F(1)).astype(np.int64)
                col = (F(0.003921568627) * tex[np.clip(Y, 0, th - 1), np.clip(X, 0, tw - 1)].astype(F)).astype(F)
            else:
                col = np.broadcast_to(np.array(m["rgb"], F), (H, W, 3))
            a_mesh = np.where(sel[..., None], col, a_mesh)
        alb = np.where(take[..., None], a_mesh, alb)
        del mat
    out = np.zeros((H, W), dtype=synth.GBUFFER_DTYPE)
    miss = gid < 0
    out["normal"] = np.where(miss[..., None], F(0), nrm)
    out["position"] = np.where(miss[..., None], (o + F(-1.0) * d).astype(F), pos)
    out["albedo"] = np.where(miss[..., None], F(0), alb)
    out["ialbedo"] = F(1.0)
    out["geomId"] = gid
    return out
