"""Renders the benchmark scene at 64 spp with and without the defocus disk (camera.rs:389-394): what its rejection sampling costs."""
import importlib, sys, os, time
import torch
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("rust-raytracer_amd")
gpu = pkg.load_gpu()
for defocus in (0.6, 0.0):
    sc = gpu.new_scene()
    pkg.scenes.random_spheres(sc, width=1920, spp=64, depth=50)
    sc.set_camera(image_width=1920, samples_per_pixel=64, max_depth=50, aspect_ratio=16.0 / 9.0, vertical_fov=20.0, look_from=(13.0, 2.0, 3.0),
                  look_at=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), defocus_angle=defocus, focus_distance=10.0, background=(0.7, 0.8, 1.0))
    sc.commit()
    best = None
    for _ in range(3):
        img, _, st = sc.render_frame(seed=1, flags=pkg.bindings.RENDER_TIME_KERNELS)
        if best is None or st["ms_total"] < best["ms_total"]:
            best = st
    print("defocus", defocus, "total %.1f trav %.1f shade %.1f seg/path %.3f" % (best["ms_total"], best["ms_trace"], best["ms_shade"], best["segments"] / best["paths"]))
