"""Times a small render of the sphere scene with and without the one-kernel tail (CHIP_TAIL_PATHS).  usage: tail_time.py [width] [spp]"""
import importlib, os, sys, time
import torch  # noqa: F401
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("rust-raytracer_amd")
gpu = pkg.load_gpu()
width = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1
sc = pkg.scenes.build(gpu, "random_spheres", width=width, spp=spp, depth=50)
for setting in ("0", "4194304", "0", "4194304"):
    os.environ["CHIP_TAIL_PATHS"] = setting
    sc.render_frame(seed=1)
    t0 = time.perf_counter()
    for _ in range(5):
        img, _, st = sc.render_frame(seed=1, flags=pkg.bindings.RENDER_TIME_KERNELS)
    dt = (time.perf_counter() - t0) / 5
    print("CHIP_TAIL_PATHS=%s: %.2f ms per frame, %d paths, %d segments, trace %.2f ms shade %.2f ms, launches %d" % (setting, 1e3 * dt, st["paths"], st["segments"], st["ms_trace"], st["ms_shade"], st["trace_launches"]), flush=True)
