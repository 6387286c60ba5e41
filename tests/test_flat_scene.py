"""`chip_scene_from_flat`: many constructor calls in one library call.  The scene must be the one a loop over the
per-node constructors builds (same ids, same tree, same hits, same image); the loop runs on the CPU oracle."""
import numpy as np
import pytest

from helpers import compare_hits


def mixed_records(B, sc):
    """Records of every constructor kind, with references to earlier records and to nodes made before the call."""
    grey = sc.lambertian_color(0.6, 0.6, 0.6)
    steel = sc.metal(0.8, 0.8, 0.9, 0.1)
    glass = sc.dielectric(1.5)
    checker = sc.tex_checker(0.5, sc.tex_solid(0.1, 0.1, 0.1), sc.tex_solid(0.9, 0.9, 0.9))
    ground = sc.sphere((0.0, -1000.0, 0.0), 1000.0, sc.lambertian_tex(checker))  # exists before the bulk call
    rec = np.zeros(17, dtype=B.FLAT_NODE_DTYPE)

    def put(i, kind, mat=0, first=0, count=0, flags=0, v=()):
        rec[i]["kind"], rec[i]["mat"], rec[i]["first"], rec[i]["count"], rec[i]["flags"] = kind, mat, first, count, flags
        rec[i]["v"][: len(v)] = v

    put(0, B.NODE_SPHERE, grey, v=(0.0, 1.0, 0.0, 1.0))
    put(1, B.NODE_SPHERE, steel, flags=B.FLAT_MOVING, v=(2.5, 0.6, 0.0, 0.6, 0.0, 0.4, 0.0))
    put(2, B.NODE_QUAD, steel, v=(-4.0, 0.0, -3.0, 0.0, 3.0, 0.0, 3.0, 0.0, 0.0))
    put(3, B.NODE_BOX, grey, v=(-0.5, -0.5, -0.5, 0.5, 0.5, 0.5))
    put(4, B.NODE_ROTATE, first=3, v=(0.0, 1.0, 0.0, 30.0))
    put(5, B.NODE_SCALE, first=4, v=(1.5, 1.0, 0.75))
    put(6, B.NODE_TRANSLATE, first=5, v=(-2.5, 0.5, 1.0))
    put(7, B.NODE_DISC, steel, v=(0.0, 0.0, 0.0, 0.8, 0.0, 1.0, 0.2))
    put(8, B.NODE_TRANSLATE, first=7, v=(1.0, 2.5, -1.0))
    put(9, B.NODE_CONE, grey, v=(0.0, 0.0, 0.0, 1.2, 0.3, 0.7))
    put(10, B.NODE_TRANSLATE, first=9, v=(4.0, 0.0, 1.5))
    put(11, B.NODE_SPHERE, glass, v=(0.0, 0.7, 2.5, 0.7))
    put(12, B.NODE_MEDIUM, first=11, v=(0.8, 0.2, 0.4, 0.9))
    put(13, B.NODE_SPHERE, glass, v=(-1.5, 0.5, 3.0, 0.5))
    put(14, B.NODE_MEDIUM, checker, first=13, flags=B.FLAT_MEDIUM_TEX, v=(2.0,))
    put(15, B.NODE_GROUP, first=0, count=3)
    put(16, B.NODE_BVH, first=3, count=8)
    members = np.array([0, 1, 2, ~ground, 15, 6, 8, 10, 11, 12, 14], dtype=np.int32)
    return rec, members


def finish(sc, world, width=96):
    sc.set_world(int(world))
    sc.set_lights(None)
    sc.set_camera(image_width=width, aspect_ratio=1.5, samples_per_pixel=16, max_depth=12, vertical_fov=35, look_from=(3.0, 3.0, 11.0),
                  look_at=(0.0, 0.8, 0.0), background=(0.7, 0.8, 1.0))
    sc.commit()
    return sc


def test_flat_records_build_the_same_graph_as_the_constructor_loop(pkg, gpu_lib, oracle):
    """Host side only (no device needed): ids of the bulk call equal the ids of the loop."""
    B = pkg.bindings
    sg, so = gpu_lib.new_scene(), oracle.new_scene()
    rg, mg = mixed_records(B, sg)
    ro, mo = mixed_records(B, so)
    assert np.array_equal(mg, mo)  # (material / texture handles are each library's own; node ids are creation order in both)
    ids_g = sg.from_flat(rg, mg)
    ids_o = so.replay_flat(ro, mo)
    assert np.array_equal(ids_g, ids_o)
    assert len(set(ids_g.tolist())) == len(ids_g)


def test_flat_errors_name_the_record(pkg, gpu_lib):
    B = pkg.bindings
    sc = gpu_lib.new_scene()
    m = sc.lambertian_color(0.5, 0.5, 0.5)
    rec = np.zeros(2, dtype=B.FLAT_NODE_DTYPE)
    rec[0]["kind"], rec[0]["mat"] = B.NODE_SPHERE, m
    rec[0]["v"][3] = 1.0
    rec[1]["kind"], rec[1]["first"] = B.NODE_TRANSLATE, 1  # itself
    with pytest.raises(pkg.CausticError, match="flat record 1"):
        sc.from_flat(rec)
    rec[1]["first"] = ~12345  # no such node in the scene
    with pytest.raises(pkg.CausticError, match="flat record 1"):
        sc.from_flat(rec)
    rec[1]["kind"], rec[1]["first"], rec[1]["count"] = B.NODE_GROUP, 0, 2
    with pytest.raises(pkg.CausticError, match="member range"):
        sc.from_flat(rec, np.array([0], dtype=np.int32))
    rec[1]["kind"] = B.NODE_CONE_WALL  # only ConeFrustum::new makes one
    with pytest.raises(pkg.CausticError, match="no constructor"):
        sc.from_flat(rec)
    rec[0]["mat"] = 99
    with pytest.raises(pkg.CausticError, match="flat record 0"):
        sc.from_flat(rec[:1])
    assert sc.from_flat(rec[:0]).size == 0


@pytest.mark.gpu
def test_flat_scene_renders_like_the_loop_built_one(pkg, gpu, oracle):
    B = pkg.bindings
    sg, so, sl = gpu.new_scene(), oracle.new_scene(), gpu.new_scene()
    rg, mg = mixed_records(B, sg)
    ro, mo = mixed_records(B, so)
    rl, ml = mixed_records(B, sl)
    finish(sg, sg.from_flat(rg, mg)[-1])
    finish(so, so.replay_flat(ro, mo)[-1])
    finish(sl, sl.replay_flat(rl, ml)[-1])
    tg, to = sg.dump_tree(), so.dump_tree()
    for a, b in zip(tg, to):
        assert np.array_equal(a, b)
    rng = np.random.default_rng(5)
    n = 20000
    o = np.tile(np.array([3.0, 3.0, 11.0]), (n, 1)) + rng.normal(0, 0.3, (n, 3))
    d = np.array([0.0, 0.8, 0.0]) + rng.uniform(-5, 5, (n, 3)) - o
    t = rng.uniform(0, 1, n)
    info = compare_hits(sg.trace_rays(o, d, t, seed=3), so.trace_rays(o, d, t, seed=3))
    assert info["n_hit"] > n // 4
    a, _, sa = sg.render_frame(seed=2)
    b, _, sb = sl.render_frame(seed=2)
    assert sa["segments"] == sb["segments"]
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    # every node family at once (media beside discs, a cone and a Scale; a tilted disc that reaches outside its own
    # bounding box, disc.rs:25-35, which only the reference's visiting order finds): same-seed render against the oracle
    e, _, se = sg.render_frame(seed=2, flags=B.RENDER_EXACT_SHADING)
    o_img, _, so_st = so.render_frame(seed=2, sum_dtype=np.float64)
    assert se["segments"] == so_st["segments"]
    diff = np.abs(e.astype(np.float64) - o_img)
    assert (diff > 1e-3 * (1.0 + np.abs(o_img))).mean() < 2e-3


@pytest.mark.gpu
def test_lattice_bulk_equals_loop(pkg, gpu, oracle):
    """The OpenSCAD-shaped lattice (top-level BVH over n^3 Translate(Group[Sphere])): built in one call on the GPU
    library, node by node on the oracle."""
    def build(be, bulk):
        sc = be.new_scene()
        pkg.scenes.scad_lattice(sc, n=7, width=128, spp=4, depth=12, bulk=bulk)
        sc.commit()
        return sc
    sg, sl, so = build(gpu, True), build(gpu, False), build(oracle, False)
    for a, b, c in zip(sg.dump_tree(), sl.dump_tree(), so.dump_tree()):
        assert np.array_equal(a, b) and np.array_equal(a, c)
    rng = np.random.default_rng(1)
    n = 20000
    o = np.tile(np.array([11.2, 6.3, 11.2]), (n, 1))
    d = rng.uniform(-4, 4, (n, 3)) + np.array([0.0, 3.0, 0.0]) - o
    info = compare_hits(sg.trace_rays(o, d, None, seed=1), so.trace_rays(o, d, None, seed=1))
    assert info["n_hit"] > n // 4
    a, _, sa = sg.render_frame(seed=4)
    b, _, sb = sl.render_frame(seed=4)
    assert sa["segments"] == sb["segments"]
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.gpu
def test_media_with_general_boundaries_beside_extended_nodes(pkg, gpu, oracle):
    """Media whose boundaries need the general sub-traversal (an ellipsoid = Scale(Sphere), a rotated box, a cone
    frustum) in a scene that also holds discs: hits and a same-seed render against the oracle."""
    B = pkg.bindings

    def build(be):
        sc = be.new_scene()
        grey = sc.lambertian_color(0.6, 0.6, 0.6)
        steel = sc.metal(0.8, 0.8, 0.9, 0.05)
        glass = sc.dielectric(1.5)
        ground = sc.sphere((0.0, -1000.0, 0.0), 1000.0, grey)
        ellipsoid = sc.translate(sc.scale(sc.sphere((0.0, 0.0, 0.0), 1.0, glass), 1.6, 0.8, 1.0), (-2.5, 0.9, 0.0))
        box = sc.translate(sc.rotate_y(sc.box((-0.7, 0.0, -0.7), (0.7, 1.4, 0.7), grey), 25.0), (0.0, 0.0, 1.0))
        cone = sc.translate(sc.cone_frustum((0.0, 0.0, 0.0), 1.5, 0.2, 0.9, grey), (2.6, 0.0, -0.5))
        world = [
            ground,
            sc.constant_medium_color(ellipsoid, 0.9, 0.9, 0.5, 0.2),
            sc.constant_medium_color(box, 1.5, 0.2, 0.4, 0.9),
            sc.constant_medium_color(cone, 2.5, 0.9, 0.9, 0.9),
            sc.translate(sc.disc((0.0, 0.0, 0.0), 0.9, (0.0, 1.0, 0.0), steel), (0.5, 2.4, -1.5)),
            sc.sphere((-0.8, 0.5, 2.6), 0.5, steel),
        ]
        sc.set_world(sc.bvh(world))
        sc.set_lights(None)
        sc.set_camera(image_width=96, aspect_ratio=1.5, samples_per_pixel=16, max_depth=12, vertical_fov=35, look_from=(2.0, 3.0, 10.0),
                      look_at=(0.0, 0.8, 0.0), background=(0.7, 0.8, 1.0))
        sc.commit()
        return sc

    sg, so = build(gpu), build(oracle)
    for a, b in zip(sg.dump_tree(), so.dump_tree()):
        assert np.array_equal(a, b)
    rng = np.random.default_rng(8)
    n = 20000
    o = np.tile(np.array([2.0, 3.0, 10.0]), (n, 1)) + rng.normal(0, 0.3, (n, 3))
    d = np.array([0.0, 0.8, 0.0]) + rng.uniform(-4, 4, (n, 3)) - o
    hg, ho = sg.trace_rays(o, d, None, seed=3), so.trace_rays(o, d, None, seed=3)
    info = compare_hits(hg, ho)
    assert info["n_hit"] > n // 4
    assert (hg["material_kind"] == 4).sum() > 200  # Isotropic: rays that scattered inside a medium
    e, _, se = sg.render_frame(seed=2, flags=B.RENDER_EXACT_SHADING)
    o_img, _, so_st = so.render_frame(seed=2, sum_dtype=np.float64)
    assert se["segments"] == so_st["segments"]
    diff = np.abs(e.astype(np.float64) - o_img)
    assert (diff > 1e-3 * (1.0 + np.abs(o_img))).mean() < 2e-3
    a, _, sa = sg.render_frame(seed=2)
    b, _, sb = sg.render_frame(seed=2, flags=B.RENDER_REFERENCE_ORDER)
    assert sa["segments"] == sb["segments"]
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def nested_groups(sc, depth, width=64):
    """Group[Group[...[Sphere, Sphere]..., Sphere], Sphere]: every level leaves a continuation on the traversal stack."""
    mat = sc.lambertian_color(0.7, 0.4, 0.3)
    node = sc.sphere((0.0, 0.5, 0.0), 0.5, mat)
    for i in range(depth):
        s = sc.sphere((0.45 * (i + 1) * (-1) ** i, 0.4, -0.3 * i), 0.4, sc.metal(0.8, 0.8, 0.8, 0.0) if i % 3 == 0 else mat)
        node = sc.group([node, s])
    ground = sc.sphere((0.0, -1000.0, 0.0), 1000.0, sc.lambertian_color(0.5, 0.5, 0.5))
    sc.set_world(sc.bvh([ground, node]))
    sc.set_lights(None)
    sc.set_camera(image_width=width, aspect_ratio=1.5, samples_per_pixel=16, max_depth=8, vertical_fov=50, look_from=(0.0, 3.0, 12.0),
                  look_at=(0.0, 0.5, -2.0), background=(0.7, 0.8, 1.0))
    sc.commit()
    return sc


@pytest.mark.gpu
def test_deep_nesting_spills_the_traversal_stack_and_still_matches(pkg, gpu, oracle):
    """20 nested groups need 40 stack entries: more than the per-lane LDS part of the stack in every kernel variant,
    so the spill slab in device memory is exercised; results must not change."""
    B = pkg.bindings
    sg, so = nested_groups(gpu.new_scene(), 20), nested_groups(oracle.new_scene(), 20)
    rng = np.random.default_rng(2)
    n = 20000
    o = np.tile(np.array([0.0, 3.0, 12.0]), (n, 1))
    d = np.array([0.0, 0.5, -2.0]) + rng.uniform(-6, 6, (n, 3)) * np.array([1.5, 0.5, 1.0]) - o
    info = compare_hits(sg.trace_rays(o, d, None, seed=1), so.trace_rays(o, d, None, seed=1))
    assert info["n_hit"] > n // 4
    e, _, se = sg.render_frame(seed=2, flags=B.RENDER_EXACT_SHADING)
    o_img, _, so_st = so.render_frame(seed=2, sum_dtype=np.float64)
    assert se["segments"] == so_st["segments"]
    diff = np.abs(e.astype(np.float64) - o_img)
    assert (diff > 1e-3 * (1.0 + np.abs(o_img))).mean() < 2e-3
    a, _, sa = sg.render_frame(seed=2)
    b, _, sb = sg.render_frame(seed=2, flags=B.RENDER_REFERENCE_ORDER)
    assert sa["segments"] == sb["segments"]
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.gpu
def test_nesting_deeper_than_the_device_stack_is_an_error(pkg, gpu):
    """40 nested groups need 80 entries when they are walked in the reference's order, more than any variant's stack:
    the calls that walk them so (rays with a zero direction component, reference-order renders) fail with a message
    instead of returning an image with missing objects.  The ordered traversal finds a Group's children in a culling
    tree and needs no stack entry per level: it renders the scene."""
    B = pkg.bindings
    sg = nested_groups(gpu.new_scene(), 40)
    o = np.array([[0.0, 3.0, 12.0]] * 64)
    d = np.array([[0.0, -2.5, -14.0]] * 64)
    with pytest.raises(pkg.CausticError, match="stack overflow"):
        sg.trace_rays(o, d, None, seed=1)
    with pytest.raises(pkg.CausticError, match="stack overflow"):
        sg.render_frame(seed=1, flags=B.RENDER_REFERENCE_ORDER)
    img, _, st = sg.render_frame(seed=1)
    assert np.isfinite(img).all() and st["paths"] > 0
    # the scene handle stays usable
    ok = nested_groups(gpu.new_scene(), 3)
    img, _, st = ok.render_frame(seed=1)
    assert np.isfinite(img).all() and st["paths"] > 0
