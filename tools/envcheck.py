"""Renders scenes with and without an environment setting (read per render call) and compares the frames bit for bit.
usage: envcheck.py [--scenes a,b,c] VAR=value [VAR=value ...]      (scene names of scenes.py, or lattice)"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("rust-raytracer_amd")
gpu = pkg.load_gpu()
args = sys.argv[1:]
scenes = ["final", "lattice"]
if args and args[0] == "--scenes":
    scenes = args[1].split(",")
    args = args[2:]
settings = dict(a.split("=", 1) for a in args)
def build(name):
    if name == "lattice":
        sc = gpu.new_scene(); pkg.scenes.scad_lattice(sc, n=20, width=480, spp=16, depth=20); sc.commit(); return sc
    return pkg.scenes.build(gpu, name, width=256, spp=16, depth=20)
for name in scenes:
    sc = build(name)
    for flags in (0, pkg.bindings.RENDER_EXACT_SHADING):
        for k in settings: os.environ.pop(k, None)
        a, _, sa = sc.render_frame(seed=3, flags=flags)
        os.environ.update(settings)
        b, _, sb = sc.render_frame(seed=3, flags=flags)
        for k in settings: os.environ.pop(k, None)
        same = np.array_equal(a.view(np.uint32), b.view(np.uint32))
        print(name, flags, settings, "identical" if same else "DIFFERENT", sa["segments"], sb["segments"], sa["bvh_visits"], sb["bvh_visits"], flush=True)
        assert same and sa["segments"] == sb["segments"]  # (visit counts may differ where descents overflow their stacks: those rays are traced twice)
