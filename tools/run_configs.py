#!/usr/bin/env python3
"""Renders the other BASELINE configurations once on one GPU (reduced spp, stated) and prints throughput + traversal stats.
usage: run_configs.py [final] [cornell] [lattice[:n]] [rtiow] [scad]"""
import importlib, json, os, sys, time
import torch  # noqa: F401  (shares the HIP runtime, as bench.py does)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("rust-raytracer_amd")
gpu = pkg.load_gpu()

def run(name, build, spp_note, repeats=2):
    t0 = time.time()
    sc = build()
    t_build = time.time() - t0
    w, h = sc.image_size()
    best = None
    for _ in range(repeats):
        t0 = time.perf_counter()
        img, _, st = sc.render_frame(seed=1, flags=pkg.bindings.RENDER_TIME_KERNELS)
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, st)
    dt, st = best
    import numpy as np
    out = {
        "config": name, "image": f"{w}x{h}", "spp": sc.sqrt_spp() ** 2, "note": spp_note, "build_s": round(t_build, 2),
        "frame_s": round(dt, 3), "path_samples_per_s": st["paths"] / dt, "segments_per_path": st["segments"] / st["paths"],
        "bvh_visits_per_segment": st["bvh_visits"] / st["segments"], "prim_tests_per_segment": st["prim_tests"] / st["segments"],
        "traversal_ms": st["ms_trace"], "bin_shade_ms": st["ms_shade"], "device_ms": st["ms_total"], "batches": st["batches"],
        "mean_radiance": float(np.nanmean(img) / sc.sqrt_spp() ** 2), "nan_pixels": int(np.isnan(img).any(axis=-1).sum()),
    }
    print(json.dumps(out), flush=True)

args = sys.argv[1:] or ["final", "cornell", "rtiow"]
for a in args:
    if a == "rtiow":
        run("S2 RTIOW final 1920x1080", lambda: pkg.scenes.build(gpu, "random_spheres", width=1920, spp=64, depth=50), "64 of 1024 spp")
    elif a == "final":
        run("S3 Next-Week final 800x800", lambda: pkg.scenes.build(gpu, "final", width=800, spp=400, depth=40), "400 of 10000 spp")
    elif a == "cornell":
        run("S4 Cornell box 600x600", lambda: pkg.scenes.build(gpu, "cornell_box", width=600, spp=729, depth=50), "729 of 4096 spp")
    elif a.startswith("lattice"):
        n = int(a.split(":")[1]) if ":" in a else 100
        def b():
            sc = gpu.new_scene()
            pkg.scenes.scad_lattice(sc, n=n, width=1920, spp=64, depth=50)
            sc.commit()
            return sc
        run(f"S5 scad-shaped lattice {n}^3 1920x1080", b, "64 of 256 spp")
    elif a == "scad":
        run("scad shapes (spheres, cubes, cylinders, a light) 1280x720", lambda: pkg.scenes.build(gpu, "scad_shapes", width=1280, spp=256, depth=50), "256 spp")
