"""A ConstantMedium takes any node as its boundary (constant_medium.rs:19-29), so the boundary may hold media of its
own.  Each test of the enclosing medium then asks the boundary twice (constant_medium.rs:42-45) and every such query
makes the media inside draw a fresh free-flight sample (constant_medium.rs:64).  The oracle's keyed form and the device
give a medium inside a boundary a run of sample keys per query of the enclosing medium and run the enclosing medium's
tests one after the other; the oracle's literal form (one draw from the stream per test, reference visiting order) is
the yardstick both are compared with as images.
Parity unpinned above the math level: the reference has no test for a medium inside a boundary; what is expected
follows from the cited code."""
import numpy as np
import pytest

ORC_MEDIA_LITERAL = 1 << 16  # one draw from the stream per test of a medium (constant_medium.rs:64), reference visiting order

SHAPES = ["medium_is_boundary", "group_boundary", "twice_tested", "box_inside_instances"]


def _nested_scene(be, shape, width=24, spp=16, depth=6):
    sc = be.new_scene()
    lam = sc.lambertian_color(0.5, 0.5, 0.5)
    far = sc.sphere((0.0, 50.0, 0.0), 1.0, lam)
    if shape == "medium_is_boundary":
        # the boundary is itself a medium: its scatter points are the "surface" the enclosing medium starts and ends at
        inner = sc.constant_medium_color(sc.sphere((0.0, 0.0, 0.0), 1.5, lam), 1.5, 1.0, 1.0, 1.0)
        outer = sc.constant_medium_color(inner, 0.8, 0.2, 0.3, 0.9)
        world = sc.bvh([outer, far])
    elif shape == "group_boundary":
        inner = sc.constant_medium_color(sc.sphere((0.3, 0.0, 0.0), 0.7, lam), 2.0, 1.0, 1.0, 1.0)
        outer = sc.constant_medium_color(sc.group([sc.sphere((0.0, 0.0, 0.0), 1.5, lam), inner]), 0.6, 0.2, 0.3, 0.9)
        world = sc.bvh([outer, far])
    elif shape == "twice_tested":
        # the enclosing medium alone in a hierarchy (tested twice per world.hit, bvh.rs:22-28), the medium inside alone in a
        # hierarchy of the boundary (tested twice per boundary query): 2 x 2 x 2 draws of the inner medium per world.hit
        inner = sc.constant_medium_color(sc.sphere((0.0, 0.2, 0.0), 0.8, lam), 1.0, 1.0, 1.0, 1.0)
        outer = sc.constant_medium_color(sc.group([sc.sphere((0.0, 0.0, 0.0), 1.5, lam), sc.bvh([inner])]), 0.5, 0.2, 0.3, 0.9)
        world = sc.group([sc.bvh([outer]), far])
    else:
        # the medium inside is bounded by a box under a Rotate under a Translate (queried in place, six quad tests)
        box = sc.translate(sc.rotate_y(sc.box((-0.5, -0.5, -0.5), (0.5, 0.5, 0.5), lam), 20.0), (0.2, 0.1, 0.0))
        inner = sc.constant_medium_color(box, 2.5, 1.0, 1.0, 1.0)
        outer = sc.translate(sc.constant_medium_color(sc.group([sc.sphere((0.0, 0.0, 0.0), 1.5, lam), inner]), 0.7, 0.2, 0.3, 0.9), (0.1, 0.0, 0.0))
        world = sc.bvh([outer, far, sc.sphere((3.0, 0.0, 0.0), 0.5, lam)])
    sc.set_world(world)
    sc.set_camera(image_width=width, samples_per_pixel=spp, max_depth=depth, vertical_fov=30, look_from=(0.5, 0.3, 9), look_at=(0, 0, 0),
                  background=(1, 1, 1))
    sc.commit()
    return sc


def _rays(n, seed=5):
    rng = np.random.default_rng(seed)
    o = np.tile(np.array([[0.0, 0.0, -6.0]]), (n, 1)) + rng.uniform(-0.05, 0.05, (n, 3))
    target = rng.uniform(-1.6, 1.6, (n, 3)) * np.array([1.0, 1.0, 0.0])
    return o, target - o


def _blocks(be, shape, seed, flags=0, width=36, spp=64, depth=5):
    sc = _nested_scene(be, shape, width=width, spp=spp, depth=depth)
    img, _, _ = sc.render_frame(seed=seed, flags=flags)
    img = np.asarray(img, dtype=np.float64) / float(sc.sqrt_spp() ** 2)
    h, w, _ = img.shape
    b = 4
    return img[: h - h % b, : w - w % b].reshape(b, (h - h % b) // b, b, (w - w % b) // b, 3).mean(axis=(1, 3)), img.mean(axis=(0, 1))


@pytest.mark.parametrize("shape", SHAPES)
def test_oracle_keyed_nested_media_render_like_literal_stream_draws(oracle, shape):
    """The oracle's two forms against each other, independently seeded, block by block: equal up to Monte Carlo noise."""
    keyed_a, mean_a = _blocks(oracle, shape, 11)
    keyed_b, mean_b = _blocks(oracle, shape, 12)
    literal, mean_l = _blocks(oracle, shape, 13, ORC_MEDIA_LITERAL)
    noise = np.sqrt(np.mean((keyed_a - keyed_b) ** 2))
    d_lit = np.sqrt(np.mean((0.5 * (keyed_a + keyed_b) - literal) ** 2))
    assert noise > 0 and d_lit < 1.5 * noise, (d_lit, noise)
    assert np.all(np.abs(mean_l - 0.5 * (mean_a + mean_b)) < 0.02 * np.abs(mean_l) + 2e-3)


def test_oracle_inner_medium_draws_per_query(oracle):
    """A ray through the "medium_is_boundary" shape: the enclosing medium only exists between two scatter points of the
    medium inside, drawn independently by the two boundary queries.  With one shared draw the second query (which starts
    0.0001 behind the first answer) would find the same exponential sample again far less often than a fresh one does."""
    n = 40_000
    o, d = _rays(n)
    hits = _nested_scene(oracle, "medium_is_boundary").trace_rays(o, d, seed=3)
    frac = hits["hit"].mean()
    assert 0.02 < frac < 0.9  # some rays scatter in the enclosing medium, some pass


def test_oracle_rejects_two_levels_of_nesting(oracle):
    sc = oracle.new_scene()
    lam = sc.lambertian_color(0.5, 0.5, 0.5)
    a = sc.constant_medium_color(sc.sphere((0.0, 0.0, 0.0), 0.5, lam), 1.0, 1.0, 1.0, 1.0)
    b = sc.constant_medium_color(sc.group([sc.sphere((0.0, 0.0, 0.0), 1.0, lam), a]), 1.0, 1.0, 1.0, 1.0)
    c = sc.constant_medium_color(sc.group([sc.sphere((0.0, 0.0, 0.0), 2.0, lam), b]), 1.0, 1.0, 1.0, 1.0)
    with pytest.raises(Exception, match="one level"):
        sc.set_world(sc.bvh([c]))
        sc.set_camera(image_width=8, samples_per_pixel=1, max_depth=2, look_from=(0, 0, 20), look_at=(0, 0, 0))
        sc.commit()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", SHAPES)
def test_gpu_nested_media_trace_and_render_like_the_oracle(pkg, oracle, gpu, shape):
    """Same seed, same keys: the device's hits are the oracle's (t through log(): last bits), and its exact-shading render
    has the oracle's segments."""
    n = 40_000
    o, d = _rays(n)
    sg, so = _nested_scene(gpu, shape), _nested_scene(oracle, shape)
    want, got = so.trace_rays(o, d, seed=3), sg.trace_rays(o, d, seed=3)
    assert np.array_equal(got["hit"], want["hit"])
    m = want["hit"] == 1
    assert m.sum() > 500 and (~m).sum() > 500
    assert np.array_equal(got["leaf_id"][m], want["leaf_id"][m])
    assert np.abs(got["t"][m] - want["t"][m]).max() <= 1e-11 * np.abs(want["t"][m]).max()
    for flags in (pkg.bindings.RENDER_EXACT_SHADING, pkg.bindings.RENDER_EXACT_SHADING | pkg.bindings.RENDER_REFERENCE_ORDER):
        e, _, se = sg.render_frame(seed=2, flags=flags)
        o_img, _, so_st = so.render_frame(seed=2, sum_dtype=np.float64)
        assert se["segments"] == so_st["segments"]
        assert (np.abs(e.astype(np.float64) - o_img) > 1e-3 * (1.0 + np.abs(o_img))).mean() < 2e-3
    f, _, _ = sg.render_frame(seed=2)  # the default shading mode: statistically the same picture
    assert abs(float(f.mean()) - float(o_img.mean())) < 0.03 * float(o_img.mean())


@pytest.mark.gpu
@pytest.mark.parametrize("shape", SHAPES[:2])
def test_gpu_nested_media_render_like_literal_stream_draws(oracle, gpu, shape):
    """The device's frames against the oracle's literal form, independently seeded, block by block."""
    dev_a, mean_a = _blocks(gpu, shape, 21)
    dev_b, mean_b = _blocks(gpu, shape, 22)
    literal, mean_l = _blocks(oracle, shape, 13, ORC_MEDIA_LITERAL)
    noise = np.sqrt(np.mean((dev_a - dev_b) ** 2))
    d_lit = np.sqrt(np.mean((0.5 * (dev_a + dev_b) - literal) ** 2))
    assert noise > 0 and d_lit < 1.5 * noise, (d_lit, noise)
    assert np.all(np.abs(mean_l - 0.5 * (mean_a + mean_b)) < 0.02 * np.abs(mean_l) + 2e-3)


@pytest.mark.gpu
def test_gpu_rejects_what_the_nested_form_does_not_cover(gpu):
    """Two levels of nesting, and a medium inside a boundary whose own boundary is neither a sphere nor a box."""
    def commit(build):
        sc = gpu.new_scene()
        lam = sc.lambertian_color(0.5, 0.5, 0.5)
        sc.set_world(sc.bvh([build(sc, lam)]))
        sc.set_camera(image_width=8, samples_per_pixel=1, max_depth=2, look_from=(0, 0, 20), look_at=(0, 0, 0))
        sc.commit()

    def two_levels(sc, lam):
        a = sc.constant_medium_color(sc.sphere((0.0, 0.0, 0.0), 0.5, lam), 1.0, 1.0, 1.0, 1.0)
        b = sc.constant_medium_color(sc.group([sc.sphere((0.0, 0.0, 0.0), 1.0, lam), a]), 1.0, 1.0, 1.0, 1.0)
        return sc.constant_medium_color(sc.group([sc.sphere((0.0, 0.0, 0.0), 2.0, lam), b]), 1.0, 1.0, 1.0, 1.0)

    def loose_inner_boundary(sc, lam):
        two = sc.group([sc.sphere((0.0, 0.0, 0.0), 0.5, lam), sc.sphere((0.4, 0.0, 0.0), 0.5, lam)])
        a = sc.constant_medium_color(two, 1.0, 1.0, 1.0, 1.0)
        return sc.constant_medium_color(sc.group([sc.sphere((0.0, 0.0, 0.0), 2.0, lam), a]), 1.0, 1.0, 1.0, 1.0)

    with pytest.raises(Exception, match="one level"):
        commit(two_levels)
    with pytest.raises(Exception, match="sphere or a box"):
        commit(loose_inner_boundary)
