#!/usr/bin/env python3
"""Prints the rays of the exact-tie parity test on which the GPU and the oracle disagree."""
import sys, os, importlib, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("rust-raytracer_amd")
import test_gpu_parity as T
gpu = pkg.load_gpu()
orc = pkg.bindings.Backend(C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so")), "orc_", "oracle", sum_dtype="float64")
for variant in (0, 1, 2):
    sg, so = T._tie_scene(gpu, variant), T._tie_scene(orc, variant)
    o, d = T._tie_rays()
    hg = sg.trace_rays(o, d, None, seed=1)
    ho = so.trace_rays(o, d, None, seed=1)
    bad = np.nonzero((hg["hit"] != ho["hit"]) | (hg["leaf_id"] != ho["leaf_id"]))[0]
    kinds, ids, boxes = so.dump_tree()
    pos = {}
    for k, i in enumerate(ids):
        pos.setdefault(int(i), k)
    print("variant", variant, "bad", len(bad), "gpu-later", sum(pos[int(hg["leaf_id"][i])] > pos[int(ho["leaf_id"][i])] for i in bad if hg["hit"][i] and ho["hit"][i]))
    for i in bad[:10]:
        print(" ray", i, "o", o[i], "d", d[i], "| gpu hit", hg["hit"][i], hg["leaf_id"][i], hg["t"][i], "| oracle", ho["hit"][i], ho["leaf_id"][i], ho["t"][i], "| dfs pos gpu/oracle", pos.get(int(hg["leaf_id"][i])), pos.get(int(ho["leaf_id"][i])), "mat", hg["material_kind"][i], ho["material_kind"][i])
