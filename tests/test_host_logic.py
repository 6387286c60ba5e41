"""CPU-side checks of libcaustic_hip.so: the library loads and exports every symbol the header declares,
and the host logic (scene graph, bounding boxes, BVH layout, derived camera, argument checking) agrees
with the oracle.  No kernel is launched here."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ALL_SCENES = ["random_spheres", "final", "cornell_box", "cornell_box_smoke", "three_spheres", "checkered_spheres", "earth", "perlin_spheres", "quads", "lighted_sphere",
              "lighted_cone_frustum", "scad_shapes", "scad_shapes_cone_light"]


def test_library_exports_every_declared_symbol(gpu_lib):
    text = open(os.path.join(ROOT, "include", "caustic_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(chip_[a-z0-9_]+)\s*\(", text)))
    assert len(names) >= 35
    for n in names:
        assert hasattr(gpu_lib.lib, n), f"{n} declared in caustic_hip.h but not exported"


@pytest.mark.parametrize("name", ALL_SCENES)
def test_bvh_layout_and_boxes_equal_oracle(pkg, gpu_lib, oracle, name):
    """bounding_volume_hierarchy.rs:16-46: same topology, same leaf order, bit-identical boxes."""
    sg = gpu_lib.new_scene()
    so = oracle.new_scene()
    pkg.scenes.SCENES[name](sg, width=64, spp=4, depth=4)
    pkg.scenes.SCENES[name](so, width=64, spp=4, depth=4)
    kg, ig, bg = sg.dump_tree()
    ko, io, bo = so.dump_tree()
    assert np.array_equal(kg, ko)
    assert np.array_equal(ig, io)
    assert np.array_equal(bg.view(np.uint64), bo.view(np.uint64))
    assert np.array_equal(sg.camera_derived().view(np.uint64), so.camera_derived().view(np.uint64))
    assert sg.image_size() == so.image_size() and sg.sqrt_spp() == so.sqrt_spp()


@pytest.mark.parametrize("name", ALL_SCENES)
def test_culling_trees_reach_every_primitive_once(pkg, gpu_lib, name):
    """The 4-wide culling trees (SAH split of each hierarchy's members, Groups opened) reach exactly what the
    reference's hierarchies, Groups and instances reach, each primitive once; nested boxes contain one another."""
    sg = gpu_lib.new_scene()
    pkg.scenes.SCENES[name](sg, width=64, spp=4, depth=4)
    n = sg.check_culling_trees()
    kinds, _, _ = sg.dump_tree()
    # every sphere / quad / disc / cone wall of the host tree is a primitive; the check also counts media and
    # sphere-only Translates, and a single member held as both children of a hierarchy shows twice in the dump
    prims = int(np.isin(kinds, [1, 2, 8, 10]).sum())
    assert 0 < n <= prims + int((kinds == 6).sum()) + int((kinds == 4).sum())


def test_culling_trees_of_unusual_graphs(gpu_lib):
    def camera(sc):
        sc.set_camera(image_width=16, aspect_ratio=1.0, samples_per_pixel=1, max_depth=2, vertical_fov=40, look_from=(0, 0, 9), look_at=(0, 0, 0),
                      background=(0.5, 0.5, 0.5))

    # world is a Group of Groups, instances around Groups, an empty Group, an empty hierarchy, a hierarchy of one
    sc = gpu_lib.new_scene()
    m = sc.lambertian_color(0.5, 0.5, 0.5)
    inner = sc.group([sc.sphere((0, 0, 0), 0.5, m), sc.quad((0, 0, 0), (1, 0, 0), (0, 1, 0), m), sc.group([])])
    boxed = sc.translate(sc.rotate(sc.box((0, 0, 0), (1, 1, 1), m), (0, 1, 0), 30.0), (2, 0, 0))
    one = sc.bvh([sc.sphere((3, 0, 0), 0.25, m)])
    lonely = sc.translate(sc.group([sc.sphere((0, 2, 0), 0.25, m)]), (0, 1, 0))
    many = sc.bvh([sc.box((k, -3, 0), (k + 0.5, -2.5, 0.5), m) for k in range(9)])
    sc.set_world(sc.group([inner, boxed, one, sc.bvh([]), lonely, sc.scale(sc.group([many, inner]), 2.0, 1.0, 1.0)]))
    sc.set_lights(None)
    camera(sc)
    # inner (sphere + quad) twice, the rotated box's 6 quads, the single sphere, the sphere-only Translate, 9 boxes
    assert sc.check_culling_trees() == 2 * 2 + 6 + 1 + 1 + 9 * 6
    # 30 nested Groups: the culling tree holds their 31 spheres without a level per Group
    sc = gpu_lib.new_scene()
    m = sc.lambertian_color(0.5, 0.5, 0.5)
    node = sc.sphere((0, 0, 0), 0.5, m)
    for k in range(30):
        node = sc.group([node, sc.sphere((k + 1.0, 0, 0), 0.4, m)])
    sc.set_world(sc.bvh([node, sc.sphere((0, -100, 0), 99, m)]))
    sc.set_lights(None)
    camera(sc)
    assert sc.check_culling_trees() == 32


def test_culling_trees_of_the_instance_lattice(pkg, gpu_lib):
    """20^3 Translate(Group[Sphere]) instances built through chip_scene_from_flat: each is one unit of the culling tree."""
    sc = gpu_lib.new_scene()
    pkg.scenes.scad_lattice(sc, n=20, width=64, spp=4, depth=4, bulk=True)
    n = sc.check_culling_trees()
    assert 8000 <= n <= 8002  # the instances plus the ground


def test_bvh_small_counts(gpu_lib, oracle):
    """len 0 -> two empty groups, 1 -> (n0, n0), 2 -> (n0, n1) unsorted, 3 -> sorted split at len/2."""
    def tree(be, centers):
        sc = be.new_scene()
        m = sc.lambertian_color(0.5, 0.5, 0.5)
        ids = [sc.sphere(c, 0.5, m) for c in centers]
        sc.set_world(sc.bvh(ids))
        return sc.dump_tree()

    for centers in ([], [(0, 0, 0)], [(5, 0, 0), (-5, 0, 0)], [(5, 0, 0), (-5, 0, 0), (0, 0, 0)], [(0, 0, 3), (0, 0, 1), (0, 0, 2), (0, 0, 0), (0, 0, 1)]):
        kg, ig, bg = tree(gpu_lib, centers)
        ko, io, bo = tree(oracle, centers)
        assert np.array_equal(kg, ko) and np.array_equal(ig, io)
        assert np.array_equal(bg.view(np.uint64), bo.view(np.uint64))
    k, i, b = tree(gpu_lib, [])
    assert k.tolist() == [0, 3, 3]  # BVH over two empty Groups
    k, i, b = tree(gpu_lib, [(0, 0, 0)])
    assert k.tolist() == [0, 1, 1] and i.tolist() == [1, 0, 0]  # the single node is both children
    k, i, b = tree(gpu_lib, [(5, 0, 0), (-5, 0, 0)])
    assert i.tolist() == [2, 0, 1]  # two nodes keep their given order
    k, i, b = tree(gpu_lib, [(5, 0, 0), (-5, 0, 0), (0, 0, 0)])
    # sorted by x-min: ids 1,2,0 ; mid = 1 -> left BVH{1} (1 twice), right BVH{2,0}
    assert k.tolist() == [0, 0, 1, 1, 0, 1, 1] and i.tolist() == [3, -1, 1, 1, -1, 2, 0]
    # stable sort keeps equal keys in input order: z-min keys 2.5,.5,1.5,-.5,.5 -> ids 3,1,4,2,0
    k, i, b = tree(gpu_lib, [(0, 0, 3), (0, 0, 1), (0, 0, 2), (0, 0, 0), (0, 0, 1)])
    leaves = [x for x, kk in zip(i.tolist(), k.tolist()) if kk == 1]
    assert leaves == [3, 1, 4, 4, 2, 0]  # {3,1} | {4} (stored twice) + {2,0}


def test_moving_sphere_and_rotate_boxes(gpu_lib, oracle):
    def tree(be):
        sc = be.new_scene()
        m = sc.lambertian_color(0.5, 0.5, 0.5)
        s = sc.sphere((1, 2, 3), 0.5, m, direction=(0.0, 0.4, -2.0))
        bx = sc.box((0, 0, 0), (1, 2, 3), m)
        r = sc.rotate(bx, (1.0, 2.0, -0.5), 33.0)
        t = sc.translate(r, (10.0, -3.0, 0.25))
        q = sc.quad((0, 0, 0), (1, 0, 0), (0, 1, 0), m)  # flat in z: padded by 1e-4
        sc.set_world(sc.bvh([s, t, q]))
        return sc.dump_tree()

    kg, ig, bg = tree(gpu_lib)
    ko, io, bo = tree(oracle)
    assert np.array_equal(kg, ko) and np.array_equal(ig, io)
    assert np.array_equal(bg.view(np.uint64), bo.view(np.uint64))
    sphere_box = bg[list(ig).index(0)]
    np.testing.assert_allclose(sphere_box, [0.5, 1.5, 1.5, 2.9, 0.5, 3.5])
    quad_box = bg[list(ig).index(4)]
    assert quad_box[4] == -0.00005 and quad_box[5] == 0.00005


def test_host_argument_errors(pkg, gpu_lib):
    sc = gpu_lib.new_scene()
    with pytest.raises(pkg.CausticError):
        sc.lambertian_tex(3)
    with pytest.raises(pkg.CausticError):
        sc.sphere((0, 0, 0), 1.0, 0)
    m = sc.lambertian_color(0.1, 0.2, 0.3)
    s = sc.sphere((0, 0, 0), 1.0, m)
    q = sc.quad((0, 0, 0), (1, 0, 0), (0, 1, 0), m)
    with pytest.raises(pkg.CausticError):
        sc.sphere_set_direction(q, (0, 1, 0))  # not a sphere
    g = sc.group([s, q])
    with pytest.raises(pkg.CausticError):
        sc.sphere_set_direction(s, (0, 1, 0))  # already referenced (Rust: mutated before Arc::new)
    with pytest.raises(pkg.CausticError):
        sc.group([s, 99])
    with pytest.raises(pkg.CausticError):
        sc.set_world(99)
    with pytest.raises(pkg.CausticError):
        sc.tex_checker(1.0, 0, 77)
    with pytest.raises(pkg.CausticError):
        sc.set_camera(image_width=0)
    with pytest.raises(pkg.CausticError):
        sc.render_frame()  # not committed
    sc.set_world(g)
    assert "scene" in gpu_lib.last_error() or gpu_lib.last_error()


def test_image_height_rule(gpu_lib, oracle):
    """camera.rs:127-128: height = (width / aspect) as u32, at least 1."""
    for be in (gpu_lib, oracle):
        for width, aspect, want in [(1920, 16.0 / 9.0, 1080), (400, 16.0 / 9.0, 225), (300, 16.0 / 9.0, 168), (10, 100.0, 1), (800, 1.0, 800)]:
            sc = be.new_scene()
            sc.set_camera(image_width=width, aspect_ratio=aspect)
            assert sc.image_size() == (width, want)


def test_abi_layouts_agree(pkg):
    """The structs that cross the C ABI, three ways: include/caustic_hip.h compiled as plain C with tests/abi/abi_check.c's
    static assertions, the const assertions of bindings/rust/caustic-hip-sys/src/lib.rs (read as text: there is no Rust
    toolchain here), and the ctypes structures the tests drive the library with."""
    import ctypes as C
    import re
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "tests", "abi", "abi_check.c")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only", "-I", os.path.join(root, "include"), src])
    text = open(src).read()
    c_sizes = {m.group(1): int(m.group(2)) for m in re.finditer(r"SIZE\((\w+), (\d+)\);", text)}
    c_offs = {(m.group(1), m.group(2)): int(m.group(3)) for m in re.finditer(r"OFF\((\w+), (\w+), (\d+)\);", text)}
    assert len(c_sizes) == 6 and len(c_offs) > 50
    rust = open(os.path.join(root, "bindings", "rust", "caustic-hip-sys", "src", "lib.rs")).read()
    r_sizes, r_offs = {}, {}
    for m in re.finditer(r"abi_layout!\((\w+), (\d+), ([^;]*)\);", rust):
        r_sizes[m.group(1)] = int(m.group(2))
        for f in re.finditer(r"(\w+) = (\d+)", m.group(3)):
            r_offs[(m.group(1), f.group(1))] = int(f.group(2))
    assert r_sizes == c_sizes and r_offs == c_offs
    B = pkg.bindings
    ct = {"chip_camera_desc": B.CameraDesc, "chip_render_opts": B.RenderOpts, "chip_stats": B.Stats, "chip_flat_scene": B.FlatScene}
    for name, cls in ct.items():
        assert C.sizeof(cls) == c_sizes[name]
        for fname, _ in cls._fields_:
            assert getattr(cls, fname).offset == c_offs[(name, fname)], (name, fname)
    assert B.HIT_DTYPE.itemsize == c_sizes["chip_hit"] and B.FLAT_NODE_DTYPE.itemsize == c_sizes["chip_flat_node"]
    for dt, name in ((B.HIT_DTYPE, "chip_hit"), (B.FLAT_NODE_DTYPE, "chip_flat_node")):
        for fname in dt.names:
            assert dt.fields[fname][1] == c_offs[(name, fname)], (name, fname)
    # every function the Rust crate declares exists in the header, and the other way round
    header = open(os.path.join(root, "include", "caustic_hip.h")).read()
    h_fns = set(re.findall(r"\b(chip_\w+)\(", header))
    r_fns = set(re.findall(r"pub fn (chip_\w+)\(", rust))
    assert h_fns == r_fns, (sorted(h_fns - r_fns), sorted(r_fns - h_fns))


def test_png_writer_round_trips(pkg, gpu_lib, tmp_path):
    """chip_write_png (crates/cli/src/main.rs:188's img.save): a reader gets the same pixels back.  Host-side only."""
    from PIL import Image

    rng = np.random.default_rng(3)
    for w, h in ((1, 1), (7, 5), (300, 257)):  # the last one needs more than one 64 KB stored block
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        path = str(tmp_path / f"t{w}x{h}.png")
        pkg.write_png(path, img)
        back = np.asarray(Image.open(path).convert("RGB"))
        assert back.shape == img.shape and np.array_equal(back, img)
    with pytest.raises(pkg.CausticError, match="cannot open"):
        pkg.write_png(str(tmp_path / "no_such_dir" / "x.png"), img)
