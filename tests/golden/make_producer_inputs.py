#!/usr/bin/env python3
"""Make tests/golden/ref_scenes/<scene>_producer_inputs.npz: what svgf_scene_render_mesh needs to ray-cast one of the
reference's scenes on the GPU box, where /root/reference does not exist — the parsed primitives (scene.geom_array), the
world-space triangles of its OBJ meshes (mesh.scene_triangles: position, normal, uv per corner), the object id and albedo
of every triangle, which objects are textured, the camera's FOVY.  DATA derived from the reference's scene files by this
repository's own parser (pinned to the reference's loader by tests/test_ref_scenes.py), not source text.

Run in the build container only:
    python tests/golden/make_producer_inputs.py bunny            # -> tests/golden/ref_scenes/bunny_producer_inputs.npz
    python tests/golden/make_producer_inputs.py room --textures  # also stores the decoded textures (large)
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

REF_SCENES = "/root/reference/scenes"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scene")
    ap.add_argument("--textures", action="store_true", help="store the decoded textures and the per-triangle texture index")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "ref_scenes"))
    a = ap.parse_args()
    pkg = ge.load_package()
    import importlib
    mesh = importlib.import_module(pkg.__name__ + ".mesh")
    sc = pkg.scene.parse_scene(open(os.path.join(REF_SCENES, a.scene + ".txt")).read())
    geoms = pkg.scene.geom_array(sc)
    geom_ids = np.array([o["id"] for o in sc.objects if o["type"] in ("cube", "sphere")], dtype=np.int32)
    pos, nrm, uv, tri_obj = mesh.scene_triangles(sc, os.path.join(REF_SCENES, "Models"))
    tris = np.concatenate([pos, nrm, uv], axis=2).astype(np.float32)                # [n, 3, 8]
    obj_by_id = {o["id"]: o for o in sc.objects}
    tri_albedo = np.array([sc.materials[obj_by_id[int(i)]["material"]]["rgb"] for i in tri_obj], dtype=np.float32).reshape(-1, 3)
    textured = sorted({int(i) for i in tri_obj if "texture" in sc.materials[obj_by_id[int(i)]["material"]]})
    out = dict(geoms=geoms, geom_ids=geom_ids, tris=tris, tri_ids=tri_obj.astype(np.int32), tri_albedo=tri_albedo,
               textured_objects=np.array(textured, dtype=np.int32), fovy=np.float32(sc.camera["fovy"]))
    if a.textures and textured:
        tex = mesh.load_textures(sc, os.path.join(REF_SCENES, "Textures"))
        mids = sorted(tex)
        out["tri_tex"] = np.array([mids.index(obj_by_id[int(i)]["material"]) if obj_by_id[int(i)]["material"] in tex else -1 for i in tri_obj], dtype=np.int32)
        for k, m in enumerate(mids):
            out[f"texture{k}"] = tex[m]
    path = os.path.join(a.out, a.scene + "_producer_inputs.npz")
    np.savez_compressed(path, **out)
    print(f"{path}: {len(geoms)} primitives, {len(tris)} triangles, textured objects {textured}, {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
