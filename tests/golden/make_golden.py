"""Generates tests/golden/golden_v2.npz from the CPU oracle: for each named scene a fixed ray set with the
oracle's hit records, and a tiny radiance-sum image.  The reference holds no vectors for this level
(no ray/hit fixture, no image test), so these pin the oracle against regressions and give the GPU
tests committed numbers to hit; they are produced by this repo's restatement, not by the reference.

Run from the repo root:  python tests/golden/make_golden.py
"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import fixed_ray_set  # noqa: E402

pkg = importlib.import_module("rust-raytracer_amd")
orc = pkg.bindings.Backend(C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so")), "orc_", "oracle", sum_dtype="float64")

SCENES = ["random_spheres", "final", "cornell_box", "cornell_box_smoke", "three_spheres", "perlin_spheres", "lighted_cone_frustum", "scad_shapes"]
KW = dict(width=32, spp=4, depth=8)
out = {}
for name in SCENES:
    sc = pkg.scenes.build(orc, name, **KW)
    w, h = sc.image_size()
    ys, xs = np.mgrid[0:h, 0:w]
    xyss = np.stack([xs.ravel(), ys.ravel(), np.zeros(w * h), np.zeros(w * h)], axis=1).astype(np.uint32)
    cam = sc.camera_rays(xyss, seed=3)
    o, d, t = fixed_ray_set(sc, 3000, seed=77)
    o = np.vstack([cam[:, 0:3], o])
    d = np.vstack([cam[:, 3:6], d])
    t = np.concatenate([cam[:, 6], t])
    hits = sc.trace_rays(o, d, t, seed=9)
    img, _, st = sc.render_frame(seed=21, sum_dtype=np.float64)
    out[name + "/o"] = o
    out[name + "/d"] = d
    out[name + "/time"] = t
    out[name + "/hit"] = hits["hit"].astype(np.int8)
    out[name + "/leaf_id"] = hits["leaf_id"]
    out[name + "/t"] = hits["t"]
    out[name + "/normal"] = hits["normal"]
    out[name + "/uv"] = np.stack([hits["u"], hits["v"]], axis=1)
    out[name + "/image_sum"] = img
    out[name + "/segments"] = np.array([st["segments"]], dtype=np.int64)
    kinds, ids, boxes = sc.dump_tree()
    out[name + "/tree_kinds"] = kinds.astype(np.int8)
    out[name + "/tree_ids"] = ids
    out[name + "/tree_boxes"] = boxes
    print(name, len(t), "rays", int(hits["hit"].sum()), "hits", "segments", st["segments"])
path = os.path.join(ROOT, "tests", "golden", "golden_v2.npz")
np.savez_compressed(path, **out)
print(path, os.path.getsize(path))
