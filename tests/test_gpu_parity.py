"""GPU parity tests: every call goes through the C ABI of libcaustic_hip.so; the CPU oracle is the checker.

Bars (from the hot path's correctness statement):
  * deterministic pieces identical: camera rays, hit/miss and closest primitive on a fixed ray set;
    t, normal, point, uv within 1e-5 relative;
  * Monte Carlo image: with the same Philox seed the GPU and the oracle consume the same uniforms, so
    radiance sums agree to rounding except for the rare path an ulp sends another way; against an
    independent high-spp oracle render PSNR >= 40 dB, with consistent per-pixel variance.
"""
import numpy as np
import pytest

from helpers import build_pair, compare_hits, fixed_ray_set, gamma_image, psnr

pytestmark = pytest.mark.gpu

ALL_SCENES = [
    "random_spheres",
    "final",
    "cornell_box",
    "cornell_box_smoke",
    "three_spheres",
    "checkered_spheres",
    "earth",
    "perlin_spheres",
    "quads",
    "lighted_sphere",
    "lighted_cone_frustum",
    "scad_shapes",
    "scad_shapes_cone_light",
]


@pytest.mark.parametrize("name", ["random_spheres", "three_spheres", "cornell_box"])
def test_camera_rays_bit_identical(pkg, gpu, oracle, name):
    sg, so = build_pair(pkg, gpu, oracle, name, width=64, spp=16, depth=4)
    w, h = sg.image_size()
    assert (w, h) == so.image_size()
    q = sg.sqrt_spp()
    ys, xs = np.mgrid[0:h, 0:w]
    rng = np.random.default_rng(5)
    xyss = np.stack([xs.ravel(), ys.ravel(), rng.integers(0, q, w * h), rng.integers(0, q, w * h)], axis=1).astype(np.uint32)
    rg = sg.camera_rays(xyss, seed=11)
    ro = so.camera_rays(xyss, seed=11)
    assert np.array_equal(rg.view(np.uint64), ro.view(np.uint64))
    assert np.array_equal(sg.camera_derived().view(np.uint64), so.camera_derived().view(np.uint64))


@pytest.mark.parametrize("name", ALL_SCENES)
def test_fixed_ray_set_hits(pkg, gpu, oracle, name):
    """Hit/miss + closest primitive identical, t / normal / pt / uv within 1e-5 relative."""
    sg, so = build_pair(pkg, gpu, oracle, name, width=96, spp=1, depth=4)
    w, h = sg.image_size()
    ys, xs = np.mgrid[0:h, 0:w]
    xyss = np.stack([xs.ravel(), ys.ravel(), np.zeros(w * h), np.zeros(w * h)], axis=1).astype(np.uint32)
    cam = so.camera_rays(xyss, seed=3)
    o, d, t = fixed_ray_set(so, 200_000)
    o = np.vstack([cam[:, 0:3], o])
    d = np.vstack([cam[:, 3:6], d])
    t = np.concatenate([cam[:, 6], t])
    hg = sg.trace_rays(o, d, t, seed=9)
    ho = so.trace_rays(o, d, t, seed=9)
    info = compare_hits(hg, ho)
    print(name, info)
    assert info["n_hit"] > 0.05 * info["n"]


@pytest.mark.parametrize("name", ALL_SCENES)
def test_same_seed_render_matches_oracle(pkg, gpu, oracle, name):
    """Exact-shading mode consumes the Philox stream exactly as the oracle does: identical paths."""
    kw = dict(width=80, spp=16, depth=12)
    sg, so = build_pair(pkg, gpu, oracle, name, **kw)
    img_g, rgb8_g, st_g = sg.render_frame(seed=21, want_rgb8=True, flags=pkg.bindings.RENDER_EXACT_SHADING)
    img_o, rgb8_o, st_o = so.render_frame(seed=21, want_rgb8=True, sum_dtype=np.float64)
    assert st_g["paths"] == st_o["paths"]
    diff = np.abs(img_g.astype(np.float64) - img_o)
    bad = diff > 1e-3 * (1.0 + np.abs(img_o))
    print(name, "segments gpu/oracle", st_g["segments"], st_o["segments"], "bad fraction", bad.mean(), "max diff", diff.max())
    # identical uniforms => identical paths, except where one ulp flips a decision
    assert bad.mean() < 2e-3
    assert abs(st_g["segments"] - st_o["segments"]) <= 2e-4 * st_o["segments"] + 4
    assert (np.abs(rgb8_g.astype(int) - rgb8_o.astype(int)) > 1).mean() < 2e-3


@pytest.mark.parametrize("name", ALL_SCENES)
def test_same_seed_default_shading_close_to_oracle(pkg, gpu, oracle, name):
    """Default (fast-arithmetic) shading: same uniforms, sample directions equal to ~1e-7, so nearly every
    path is still the oracle's path; the few that an ulp sends elsewhere bound the differing fraction."""
    kw = dict(width=80, spp=16, depth=12)
    sg, so = build_pair(pkg, gpu, oracle, name, **kw)
    img_g, _, st_g = sg.render_frame(seed=21)
    img_o, _, st_o = so.render_frame(seed=21, sum_dtype=np.float64)
    diff = np.abs(img_g.astype(np.float64) - img_o)
    bad = diff > 2e-3 * (1.0 + np.abs(img_o))
    print(name, "segments gpu/oracle", st_g["segments"], st_o["segments"], "bad fraction", bad.mean(), "max diff", diff.max())
    assert bad.mean() < 1e-2
    assert abs(st_g["segments"] - st_o["segments"]) <= 2e-3 * st_o["segments"] + 8
    # means agree far inside the Monte Carlo noise
    assert abs(img_g.mean() - img_o.mean()) <= 2e-3 * abs(img_o.mean()) + 1e-6


def test_image_psnr_and_variance_vs_independent_oracle(pkg, gpu, oracle):
    """PSNR >= 40 dB of the GPU image against an independently seeded, higher-spp oracle render of the
    same scene, and per-pixel variance ratio within [0.8, 1.25] (mean over a pixel subsample)."""
    kw = dict(width=96, depth=50)
    sg = pkg.scenes.build(gpu, "random_spheres", spp=16384, **kw)
    so = pkg.scenes.build(oracle, "random_spheres", spp=4096, **kw)
    img_g, _, _ = sg.render_frame(seed=1)
    img_o, _, _ = so.render_frame(seed=2, sum_dtype=np.float64)
    p = psnr(gamma_image(img_g, sg.sqrt_spp()), gamma_image(img_o, so.sqrt_spp()))
    print("psnr", p)
    assert p >= 40.0
    # variance: K independently seeded 16-spp renders per side
    K = 24
    sg2 = pkg.scenes.build(gpu, "random_spheres", spp=16, **kw)
    so2 = pkg.scenes.build(oracle, "random_spheres", spp=16, **kw)
    G = np.stack([sg2.render_frame(seed=100 + k)[0].astype(np.float64) for k in range(K)])
    O = np.stack([so2.render_frame(seed=500 + k, sum_dtype=np.float64)[0] for k in range(K)])
    vg = G.var(axis=0, ddof=1).mean(axis=-1).ravel()
    vo = O.var(axis=0, ddof=1).mean(axis=-1).ravel()
    sel = vo > np.percentile(vo, 25)
    ratio = vg[sel].mean() / vo[sel].mean()
    print("variance ratio", ratio)
    assert 0.8 <= ratio <= 1.25


def test_cornell_psnr_vs_independent_oracle(pkg, gpu, oracle):
    """Light / mixture-PDF importance sampling path (clamp and pdf cut included): PSNR >= 40 dB."""
    kw = dict(width=64, depth=50)
    sg = pkg.scenes.build(gpu, "cornell_box", spp=16384, **kw)
    so = pkg.scenes.build(oracle, "cornell_box", spp=4096, **kw)
    img_g, _, _ = sg.render_frame(seed=1)
    img_o, _, _ = so.render_frame(seed=2, sum_dtype=np.float64)
    p = psnr(gamma_image(img_g, sg.sqrt_spp()), gamma_image(img_o, so.sqrt_spp()))
    print("psnr", p)
    assert p >= 40.0


def test_determinism_and_partition_invariance(pkg, gpu):
    """Size-independent properties at a larger size: same seed twice is bit-identical; rendering the
    strata in two interleaved halves, in small batches, or as a rectangle gives the same sums."""
    sg = pkg.scenes.build(gpu, "random_spheres", width=480, spp=16, depth=50)
    a, _, st = sg.render_frame(seed=5)
    b, _, _ = sg.render_frame(seed=5)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    c, _, _ = sg.render_frame(seed=6)
    assert not np.array_equal(a, c)
    h0, _, _ = sg.render_frame(seed=5, stratum_offset=0, stratum_stride=2)
    h1, _, _ = sg.render_frame(seed=5, stratum_offset=1, stratum_stride=2)
    np.testing.assert_allclose(h0 + h1, a, rtol=2e-6, atol=1e-6)
    # a pixel's running sum takes its samples one by one in stratum order, wherever the batches are cut
    d, _, _ = sg.render_frame(seed=5, strata_per_batch=3)
    assert np.array_equal(d.view(np.uint32), a.view(np.uint32))
    w, h = sg.image_size()
    rect = (37, 201, 11, 97)
    r, _, _ = sg.render_frame(seed=5, rect=rect)
    assert np.array_equal(r, a[rect[2] : rect[3], rect[0] : rect[1]])
    assert np.isfinite(a).all() and a.min() >= 0.0
    assert st["paths"] == w * h * 16


def test_spp_not_a_square_uses_floor_sqrt(pkg, gpu, oracle):
    """camera.rs:131: spp=10 renders floor(sqrt(10))^2 = 9 samples."""
    sg, so = build_pair(pkg, gpu, oracle, "random_spheres", width=48, spp=10, depth=6)
    assert sg.sqrt_spp() == so.sqrt_spp() == 3
    _, _, st = sg.render_frame(seed=1)
    w, h = sg.image_size()
    assert st["paths"] == w * h * 9


def test_error_paths(pkg, gpu):
    be = gpu
    sc = be.new_scene()
    with pytest.raises(pkg.CausticError):
        sc.commit()  # no world
    m = sc.lambertian_color(0.5, 0.5, 0.5)
    s = sc.sphere((0, 0, -1), 0.5, m)
    with pytest.raises(pkg.CausticError):
        sc.sphere((0, 0, 0), 1.0, 99)  # bad material id
    sc.set_world(s)
    with pytest.raises(pkg.CausticError):
        sc.commit()  # no camera
    sc.set_camera(image_width=16, samples_per_pixel=4, max_depth=3, background=(1, 1, 1))
    sc.commit()
    with pytest.raises(pkg.CausticError):
        sc.render_frame(rect=(0, 17, 0, 16))
    with pytest.raises(pkg.CausticError):
        sc.render_frame(stratum_offset=2, stratum_stride=2)
    img, _, _ = sc.render_frame(seed=1)
    assert img.shape == (16, 16, 3) and np.isfinite(img).all()


def test_depth_zero_and_empty_world(pkg, gpu, oracle):
    """max_depth = 0 returns black (camera.rs:259-261); an empty BVH (two empty Groups, bvh.rs:23-26) shows the background."""
    for be in (gpu, oracle):
        sc = be.new_scene()
        sc.set_world(sc.bvh([]))
        sc.set_camera(image_width=8, samples_per_pixel=4, max_depth=0, background=(0.25, 0.5, 1.0))
        sc.commit()
        img, _, _ = sc.render_frame(seed=1, sum_dtype=np.float32 if be is gpu else np.float64)
        assert np.all(img == 0.0)
        sc.set_camera(image_width=8, samples_per_pixel=4, max_depth=2, background=(0.25, 0.5, 1.0))
        sc.commit()
        img, _, _ = sc.render_frame(seed=1, sum_dtype=np.float32 if be is gpu else np.float64)
        np.testing.assert_allclose(img / 4.0, np.broadcast_to(np.array([0.25, 0.5, 1.0]), img.shape), rtol=1e-6)


def _tie_scene(be, variant):
    """Primitives that produce bitwise-equal hit distances: duplicated spheres (one of each pair behind an
    exactly representable Translate), coplanar overlapping quads, and quads tangent to spheres.  The
    winner of such a tie depends on the order the reference visits the primitives in (group.rs:47-52,
    bvh.rs:66-70, inclusive `contains` in quad.rs:152 against exclusive `surrounds` in sphere.rs:110)."""
    rng = np.random.default_rng(7 + variant)
    sc = be.new_scene()
    lam = sc.lambertian_color(0.5, 0.5, 0.5)
    met = sc.metal(0.8, 0.8, 0.8, 0.0)
    die = sc.dielectric(1.5)
    lit = sc.diffuse_light_color(1, 1, 1)
    nodes = []
    for k in range(24):
        c = rng.integers(-24, 25, 3) / 4.0
        r = float(rng.integers(2, 7)) / 4.0
        a = sc.sphere(tuple(c), r, lam)
        off = rng.integers(-3, 4, 3).astype(float)
        b = sc.translate(sc.sphere(tuple(c - off), r, met), tuple(off))
        nodes += [a, b] if k % 2 else [b, a]
        if k % 3 == 0:  # a quad tangent to the sphere's top, and one through its centre plane
            nodes.append(sc.quad((c[0] - 2, c[1] + r, c[2] - 2), (4, 0, 0), (0, 0, 4), die))
    for k in range(12):
        z = float(rng.integers(-20, 21)) / 4.0
        x = float(rng.integers(-16, 9)) / 4.0
        y = float(rng.integers(-16, 9)) / 4.0
        nodes.append(sc.quad((x, y, z), (2, 0, 0), (0, 2, 0), lam))
        nodes.append(sc.quad((x + 1, y + 0.5, z), (2, 0, 0), (0, 2, 0), lit))
        nodes.append(sc.quad((x - 1, y - 0.5, z), (4, 0, 0), (0, 4, 0), met))
    order = rng.permutation(len(nodes))
    nodes = [nodes[i] for i in order]
    if variant == 0:
        world = sc.bvh(nodes)
    elif variant == 1:
        h = len(nodes) // 2
        world = sc.bvh([sc.group(nodes[:h]), sc.bvh(nodes[h:])])
    else:
        world = sc.group([sc.bvh(nodes[i::3]) for i in range(3)])
    sc.set_world(world)
    sc.set_camera(image_width=32, samples_per_pixel=1, max_depth=2, look_from=(0, 0, 20), look_at=(0, 0, 0), background=(1, 1, 1))
    sc.commit()
    return sc


def _tie_rays(n=400_000):
    """Generic rays (no special alignment with the scene): a coplanar pair of quads or a duplicated sphere
    still gives bitwise-equal distances, because both primitives evaluate the same expression."""
    rng = np.random.default_rng(99)
    o = rng.uniform(-10.0, 10.0, (n, 3))
    target = rng.uniform(-6.0, 6.0, (n, 3))
    d = (target - o) * rng.uniform(0.2, 3.0, (n, 1))
    return o, d


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_exact_ties_follow_reference_order(gpu, oracle, variant):
    """A scene full of coincident surfaces: where two primitives report the same distance, the GPU's
    ordered traversal must name the primitive the reference's left-to-right recursion names.  (Rays that
    touch a bounding box exactly at the distance of an earlier hit are excluded by using generic rays: the
    reference then rejects the box with its `max <= min` test, aabb.rs:207-225, which only an exact f64
    replay of its visiting order reproduces.)"""
    sg, so = _tie_scene(gpu, variant), _tie_scene(oracle, variant)
    o, d = _tie_rays()
    hg = sg.trace_rays(o, d, None, seed=1)
    ho = so.trace_rays(o, d, None, seed=1)
    info = compare_hits(hg, ho)
    print("ties", variant, info)
    assert info["n_hit"] > 0.1 * len(o) and info["t_bit_identical"]


@pytest.mark.parametrize("name", ["random_spheres", "cornell_box", "final", "cornell_box_smoke", "scad_shapes"])
def test_ordered_traversal_equals_reference_order_render(pkg, gpu, name):
    """The default ordered traversal and the literal replay of the reference's visiting order find the
    same (t, primitive) for every segment, so whole renders are bit-identical."""
    sg = pkg.scenes.build(gpu, name, width=160, spp=16, depth=20)
    a, _, sa = sg.render_frame(seed=3)
    b, _, sb = sg.render_frame(seed=3, flags=pkg.bindings.RENDER_REFERENCE_ORDER)
    assert sa["segments"] == sb["segments"]
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_overlapped_batches_equal_single_stream(pkg, gpu, monkeypatch):
    """Many small batches run two at a time on two streams (the production path of a large frame) add up
    to exactly the image of the same batches run one after the other."""
    sg = pkg.scenes.build(gpu, "random_spheres", width=200, spp=64, depth=30)
    monkeypatch.setenv("CHIP_STREAMS", "1")
    a, _, sa = sg.render_frame(seed=9, strata_per_batch=5)
    monkeypatch.delenv("CHIP_STREAMS")
    monkeypatch.setenv("CHIP_OVERLAP_MIN_PATHS", "1000")
    b, _, sb = sg.render_frame(seed=9, strata_per_batch=5)
    assert min(sa["batches"], sb["batches"]) >= 12  # (a smaller pool cuts the overlapped run into more batches)
    assert sa["segments"] == sb["segments"]
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_zero_component_rays(pkg, gpu, oracle):
    """Rays with an exactly zero direction component go through the literal traversal (their slab arithmetic
    forms NaNs in the reference, aabb.rs:202-228); a mirror scene rendered with the same seed must also match.
    The literal *queue* of the render path is exercised by the reference-order render test above."""
    def build(be):
        sc = be.new_scene()
        mirror = sc.metal(0.9, 0.9, 0.9, 0.0)
        red = sc.lambertian_color(0.8, 0.2, 0.2)
        nodes = [sc.sphere((0, 0, -3), 1.0, red), sc.sphere((0, 0, -1005), 1000.0, mirror), sc.sphere((2.5, 0, -3), 1.0, mirror)]
        sc.set_world(sc.bvh(nodes))
        sc.set_camera(image_width=48, aspect_ratio=1.0, samples_per_pixel=16, max_depth=8, vertical_fov=40, look_from=(0, 0, 4), look_at=(0, 0, 0),
                      background=(0.6, 0.7, 1.0))
        sc.commit()
        return sc
    sg, so = build(gpu), build(oracle)
    o = np.array([[0.0, 0.0, 4.0], [2.5, 0.0, 4.0], [0.0, 1.0, 4.0], [-1.0, 0.0, 4.0], [0.0, 0.0, -3.0], [0.0, 0.0, 4.0]])
    d = np.array([[0.0, 0.0, -1.0], [0.0, 0.0, -2.0], [0.0, -0.25, -1.0], [0.25, 0.0, -1.0], [0.0, 1.0, 0.0], [0.0, 1.0, 0.0]])
    info = compare_hits(sg.trace_rays(o, d, None, seed=1), so.trace_rays(o, d, None, seed=1))
    assert info["n_hit"] >= 5
    img_g, _, st_g = sg.render_frame(seed=4, flags=pkg.bindings.RENDER_EXACT_SHADING)
    img_o, _, st_o = so.render_frame(seed=4, sum_dtype=np.float64)
    assert st_g["segments"] == st_o["segments"]
    diff = np.abs(img_g.astype(np.float64) - img_o)
    assert (diff > 1e-3 * (1.0 + np.abs(img_o))).mean() < 2e-3


def test_rgb8_quantisation_variants(pkg, gpu, oracle):
    """8-bit output: x 255.999 (crates/cli/src/main.rs:193-198) by default, x 255.0 (crates/wasm/src/lib.rs:269-276)
    with the flag; both must agree with the oracle's bytes."""
    sg, so = build_pair(pkg, gpu, oracle, "three_spheres", width=64, spp=16, depth=8)
    for fl in (0, pkg.bindings.RENDER_QUANTISE_255):
        _, g8, _ = sg.render_frame(seed=2, want_rgb8=True, flags=fl | pkg.bindings.RENDER_EXACT_SHADING)
        _, o8, _ = so.render_frame(seed=2, want_rgb8=True, sum_dtype=np.float64, flags=fl)
        assert (np.abs(g8.astype(int) - o8.astype(int)) > 1).mean() < 2e-3
    _, a8, _ = sg.render_frame(seed=2, want_rgb8=True)
    _, b8, _ = sg.render_frame(seed=2, want_rgb8=True, flags=pkg.bindings.RENDER_QUANTISE_255)
    assert (a8.astype(int) - b8.astype(int)).min() >= 0 and (a8 != b8).any()


def test_full_size_frame_properties(pkg, gpu, monkeypatch):
    """The benchmark's frame size (1920 x 1080, depth 50; 16 of its 1024 strata so that it runs in a blink), checked
    through properties that do not need the oracle: batches overlapped on two streams give the bits of the same batches
    run one after the other, a rectangle is the crop of the frame, and two ranks' strata add up to the frame."""
    sg = pkg.scenes.build(gpu, "random_spheres", width=1920, spp=16, depth=50)
    assert sg.image_size() == (1920, 1080)
    monkeypatch.setenv("CHIP_STREAMS", "1")
    a, _, sa = sg.render_frame(seed=3, strata_per_batch=4)
    monkeypatch.delenv("CHIP_STREAMS")
    b, _, sb = sg.render_frame(seed=3, strata_per_batch=4)
    assert min(sa["batches"], sb["batches"]) >= 4
    assert sa["paths"] == sb["paths"] == 1920 * 1080 * 16
    assert sa["segments"] == sb["segments"]
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.isfinite(a).all() and a.min() >= 0.0
    rect = (700, 1500, 300, 900)  # xmin, xmax, ymin, ymax
    c, _, _ = sg.render_frame(seed=3, rect=rect, strata_per_batch=4)
    assert np.array_equal(c.view(np.uint32), a[rect[2]:rect[3], rect[0]:rect[1]].view(np.uint32))
    r0, _, s0 = sg.render_frame(seed=3, stratum_offset=0, stratum_stride=2)
    r1, _, s1 = sg.render_frame(seed=3, stratum_offset=1, stratum_stride=2)
    assert s0["segments"] + s1["segments"] == sa["segments"]
    assert np.allclose(r0 + r1, a, rtol=1e-5, atol=1e-5)


def test_headline_frame_against_oracle_and_rank_shares(pkg, gpu, oracle):
    """One frame of the benchmark's configuration in full (RTIOW final scene, 1920 x 1080, 1024 strata, depth 50: 2.1 G
    paths).  (i) Four rectangles of it (centre, a corner, the horizon with the small spheres, the big glass sphere)
    against the oracle's render of the same rectangles with the same seed: both follow the same Philox stream, so a
    value differs only by f32 summation and by the rare path whose f32 shading arithmetic took another branch than the
    oracle's f64 (one path is a thousandth of a pixel).  Measured: median 3e-6, 99th percentile 3e-5, largest 1.3e-4
    relative, rectangle means 4e-7; bars: 99 % within 2e-4, all within 5e-3, means within 2e-5.  (ii) The frame is
    the sum of the eight rank shares the 8-GPU job renders (strata r, r + 8, ...) to f32 addition order (measured
    2e-5), segment for segment.  (iii) The same rectangles rendered with the reference's f64 shading order have the
    oracle's segments to the last one (10 M in all) and its values to 1e-5 (f32 sums of 1024 terms)."""
    sg = pkg.scenes.build(gpu, "random_spheres", width=1920, spp=1024, depth=50)
    so = pkg.scenes.build(oracle, "random_spheres", width=1920, spp=1024, depth=50)
    full, _, st = sg.render_frame(seed=4)
    assert st["paths"] == 1920 * 1080 * 1024
    assert np.isfinite(full).all() and full.min() >= 0.0
    for rect in ((944, 976, 528, 552), (0, 32, 0, 24), (1200, 1232, 640, 664), (930, 962, 700, 724)):  # xmin, xmax, ymin, ymax
        want, _, sw = so.render_frame(seed=4, rect=rect, sum_dtype=np.float64)
        exact, _, se = sg.render_frame(seed=4, rect=rect, flags=pkg.bindings.RENDER_EXACT_SHADING)
        rel = np.abs(exact.astype(np.float64) - want) / (1.0 + np.abs(want))
        assert se["segments"] == sw["segments"] and rel.max() <= 5e-5, (rect, se["segments"], sw["segments"], float(rel.max()))
        got = full[rect[2]:rect[3], rect[0]:rect[1]].astype(np.float64)
        rel = np.abs(got - want) / (1.0 + np.abs(want))  # (the values are sums over the 1024 strata)
        assert np.quantile(rel, 0.99) <= 2e-4 and rel.max() <= 5e-3, (rect, float(np.quantile(rel, 0.99)), float(rel.max()))
        assert abs(got.mean() - want.mean()) <= 2e-5 * want.mean(), (rect, got.mean(), want.mean())
    acc = np.zeros(full.shape, dtype=np.float64)
    segments = 0
    for r in range(8):
        part, _, sr = sg.render_frame(seed=4, stratum_offset=r, stratum_stride=8)
        acc += part
        segments += sr["segments"]
    assert segments == st["segments"]
    assert np.allclose(acc, full, rtol=1e-4, atol=1e-3)


def _rect_stats(full, want, rect):
    got = full[rect[2]:rect[3], rect[0]:rect[1]].astype(np.float64)
    rel = np.abs(got - want) / (1.0 + np.abs(want))  # (the values are sums over the strata)
    return float(np.median(rel)), float(np.quantile(rel, 0.99)), float(rel.max()), float(abs(got.mean() - want.mean()) / want.mean())


def test_reference_cli_configuration_against_oracle(pkg, gpu, oracle):
    """The reference's own CPU-runnable case in full: the RTIOW final scene at 400 x 225, `samples_per_pixel` 10 (which
    camera.rs:168-170 turns into 3 x 3 strata), depth 50.  Every value of the frame against the oracle's same-seed
    frame, in the default mode (f32 shading arithmetic) and with the reference's f64 shading order."""
    sg = pkg.scenes.build(gpu, "random_spheres", width=400, spp=10, depth=50)
    so = pkg.scenes.build(oracle, "random_spheres", width=400, spp=10, depth=50)
    assert sg.image_size() == (400, 225) and sg.sqrt_spp() == 3
    want, _, sw = so.render_frame(seed=9, sum_dtype=np.float64)
    for flags in (0, pkg.bindings.RENDER_EXACT_SHADING):
        got, _, st = sg.render_frame(seed=9, flags=flags)
        assert st["paths"] == sw["paths"] == 400 * 225 * 9
        med, p99, worst, mean = _rect_stats(got, want, (0, 400, 0, 225))
        if flags:  # measured: the oracle's segments to the last one, values to 1.8e-7 relative (f32 sums)
            assert st["segments"] == sw["segments"] and worst <= 2e-6, (st["segments"], sw["segments"], worst)
        else:      # measured: 8 of 2 069 429 segments apart, 99th percentile 1e-7, one value 1.5e-2, frame mean 1e-8
            assert abs(st["segments"] - sw["segments"]) <= 1e-4 * sw["segments"]
            assert p99 <= 1e-5 and worst <= 0.2 and mean <= 1e-6, (med, p99, worst, mean)


def test_cornell_full_configuration_against_oracle(pkg, gpu, oracle):
    """The Cornell-box configuration in full (600 x 600, 4096 strata, depth 50: 1.5 G paths with light and mixture-PDF
    sampling): three rectangles (the glass sphere, the light's edge, a wall corner) against the oracle's same-seed
    render of them.  Both run the reference's f64 shading arithmetic here, so values differ by f32 summation only
    (measured: 99th percentile 1.5e-6, largest 1.7e-6 relative, rectangle means 1.6e-7); the bars leave room for a path
    that takes another branch (one path is 2.4e-4 of a pixel): 99 % within 2e-5, all within 1e-3, means within 5e-6."""
    sg = pkg.scenes.build(gpu, "cornell_box", width=600, spp=4096, depth=50)
    so = pkg.scenes.build(oracle, "cornell_box", width=600, spp=4096, depth=50)
    full, _, st = sg.render_frame(seed=6)
    assert st["paths"] == 600 * 600 * 4096
    assert np.isfinite(full).all() and full.min() >= 0.0
    for rect in ((180, 196, 420, 432), (200, 216, 40, 52), (2, 18, 580, 592)):
        want, _, sw = so.render_frame(seed=6, rect=rect, sum_dtype=np.float64)
        _, _, sr = sg.render_frame(seed=6, rect=rect, flags=pkg.bindings.RENDER_EXACT_SHADING)
        assert sr["segments"] == sw["segments"], (rect, sr["segments"], sw["segments"])  # the oracle's segments to the last one
        med, p99, worst, mean = _rect_stats(full, want, rect)
        assert p99 <= 2e-5 and worst <= 1e-3 and mean <= 5e-6, (rect, med, p99, worst, mean)


def test_next_week_rank_share_of_full_configuration_against_oracle(pkg, gpu, oracle):
    """The Next-Week final scene's configuration (800 x 800, 10 000 strata over 8 GPUs, depth 40): the share of rank 3
    (strata 3, 11, 19, ...: 1250 per pixel, 0.8 G paths through media, instanced boxes, image and noise textures) and
    three rectangles of it against the oracle's same-seed render of the same share (measured: 99th percentile 1e-6,
    largest 1.8e-6 relative, rectangle means 4e-8; bars as for the Cornell box)."""
    sg = pkg.scenes.build(gpu, "final", width=800, spp=10000, depth=40)
    so = pkg.scenes.build(oracle, "final", width=800, spp=10000, depth=40)
    share, _, st = sg.render_frame(seed=8, stratum_offset=3, stratum_stride=8)
    assert st["paths"] == 800 * 800 * 1250
    assert np.isfinite(share).all() and share.min() >= 0.0
    for rect in ((392, 404, 380, 390), (120, 132, 600, 610), (560, 572, 300, 310)):
        want, _, sw = so.render_frame(seed=8, rect=rect, stratum_offset=3, stratum_stride=8, sum_dtype=np.float64)
        _, _, sr = sg.render_frame(seed=8, rect=rect, stratum_offset=3, stratum_stride=8, flags=pkg.bindings.RENDER_EXACT_SHADING)
        assert sr["segments"] == sw["segments"], (rect, sr["segments"], sw["segments"])  # the oracle's segments to the last one
        med, p99, worst, mean = _rect_stats(share, want, rect)
        assert p99 <= 2e-5 and worst <= 1e-3 and mean <= 5e-6, (rect, med, p99, worst, mean)


@pytest.mark.parametrize("variant", ["bare_quad", "bare_sphere", "empty_group", "group_with_unsampled_members", "disc", "frustum", "nested_groups",
                                     "nested_empty_and_frustum"])
def test_light_list_shapes(pkg, gpu, oracle, variant):
    """SceneData.lights in every shape camera.rs:278-305 can meet: a node that is not a Group (its own pdf_value /
    random), an empty Group (group.rs:60-78: pdf 0, direction (0,1,0)), a Group with members that keep the Node
    defaults (object/mod.rs:59-65: pdf 0, direction (1,0,0)), a Disc, a ConeFrustum, Groups inside the Group (each level
    takes 1 / len of its parent's share of the pdf and one uniform pick of a sample).  Same-seed render against the
    oracle with the reference's f64 operation order."""
    def build(be):
        sc = be.new_scene()
        white = sc.lambertian_color(0.73, 0.73, 0.73)
        red = sc.lambertian_color(0.65, 0.05, 0.05)
        light = sc.diffuse_light_color(15.0, 15.0, 15.0)
        glass = sc.dielectric(1.5)
        lq = sc.quad((3.0, 5.0, -1.0), (-2.0, 0.0, 0.0), (0.0, 0.0, -2.0), light)
        ld = sc.disc((-2.5, 4.0, -2.0), 0.8, (0.0, -1.0, 0.0), light)
        gs = sc.sphere((0.0, 1.0, -2.0), 1.0, glass)
        box = sc.box((-3.5, 0.0, -4.0), (-2.0, 2.0, -2.5), white)
        ts = sc.translate(sc.sphere((0.0, 0.0, 0.0), 0.6, red), (2.5, 0.6, -1.0))
        cone = sc.cone_frustum((2.5, 3.0, -3.5), 1.0, 0.3, 0.7, light)
        world = [sc.quad((-6.0, 0.0, -6.0), (12.0, 0.0, 0.0), (0.0, 0.0, 8.0), white), sc.quad((-6.0, 0.0, -6.0), (12.0, 0.0, 0.0), (0.0, 8.0, 0.0), red),
                 lq, ld, gs, box, ts, cone]
        sc.set_world(sc.bvh(world))
        lights = {
            "bare_quad": lambda: lq,
            "bare_sphere": lambda: gs,
            "empty_group": lambda: sc.group([]),
            "group_with_unsampled_members": lambda: sc.group([lq, box, ts]),
            "disc": lambda: sc.group([ld, lq]),
            "frustum": lambda: cone,
            "nested_groups": lambda: sc.group([lq, sc.group([ld, sc.group([gs, lq])]), ts]),
            "nested_empty_and_frustum": lambda: sc.group([sc.group([]), sc.group([cone, ld]), lq]),
        }[variant]()
        sc.set_lights(lights)
        sc.set_camera(image_width=64, aspect_ratio=1.5, samples_per_pixel=36, max_depth=8, vertical_fov=45, look_from=(0.0, 3.0, 9.0),
                      look_at=(0.0, 1.5, -2.0), background=(0.02, 0.02, 0.03))
        sc.commit()
        return sc

    sg, so = build(gpu), build(oracle)
    img_g, _, st_g = sg.render_frame(seed=6, flags=pkg.bindings.RENDER_EXACT_SHADING)
    img_o, _, st_o = so.render_frame(seed=6, sum_dtype=np.float64)
    assert st_g["segments"] == st_o["segments"]
    diff = np.abs(img_g.astype(np.float64) - img_o)
    assert (diff > 1e-3 * (1.0 + np.abs(img_o))).mean() < 2e-3
    assert img_o.mean() > 0.0
    fast, _, _ = sg.render_frame(seed=6)
    assert (np.abs(fast.astype(np.float64) - img_o) > 2e-3 * (1.0 + np.abs(img_o))).mean() < 1e-2


def test_small_pool_row_bands_and_streams_give_the_same_bits(pkg, gpu, monkeypatch):
    """A path pool far smaller than one stratum of the frame: the frame is cut into bands of rows and the strata into
    many batches, run on one, two or three streams.  The sums are bit-identical to the single-batch render."""
    kw = dict(width=480, spp=16, depth=50)
    monkeypatch.delenv("CHIP_POOL_PATHS", raising=False)
    a, _, sa = pkg.scenes.build(gpu, "random_spheres", **kw).render_frame(seed=5)
    assert sa["batches"] == 1
    monkeypatch.setenv("CHIP_POOL_PATHS", "65536")
    monkeypatch.setenv("CHIP_OVERLAP_MIN_PATHS", "1000")
    sc = pkg.scenes.build(gpu, "random_spheres", **kw)  # a new scene: pools are sized on first use
    for streams in ("1", "2", "3"):
        monkeypatch.setenv("CHIP_STREAMS", streams)
        b, _, sb = sc.render_frame(seed=5)
        assert sb["batches"] >= 32 and sb["paths"] == sa["paths"] and sb["segments"] == sa["segments"]
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), streams
    rect = (37, 201, 11, 197)
    r, _, _ = sc.render_frame(seed=5, rect=rect)
    assert np.array_equal(r.view(np.uint32), a[rect[2] : rect[3], rect[0] : rect[1]].view(np.uint32))


def test_ray_times_outside_the_shutter_interval(pkg, gpu, oracle):
    """Moving spheres leave their bounding boxes for ray times outside [0, 1] (the boxes cover the motion over that
    interval only, sphere.rs:36-47); `trace_rays` accepts any time and must still answer as the reference would."""
    sg, so = build_pair(pkg, gpu, oracle, "random_spheres", width=64, spp=4, depth=4)
    rng = np.random.default_rng(3)
    n = 100_000
    o = np.array([13.0, 2.0, 3.0]) + rng.normal(0.0, 0.2, (n, 3))
    target = np.stack([rng.uniform(-11, 11, n), rng.uniform(-1.5, 2.5, n), rng.uniform(-11, 11, n)], axis=1)
    d = target - o
    t = rng.uniform(-3.0, 4.0, n)
    hg, ho = sg.trace_rays(o, d, t, seed=1), so.trace_rays(o, d, t, seed=1)
    info = compare_hits(hg, ho)
    assert info["n_hit"] > n // 4
    # some of these hits are on spheres that have left their boxes: their centres lie above or below the motion segment
    moved = (ho["hit"] == 1) & ((t < 0.0) | (t > 1.0)) & (ho["t"] < 100.0) & (np.abs(ho["pt"][:, 1] - 0.2) > 0.75) & (ho["pt"][:, 1] < 2.9)
    assert moved.sum() > 100


@pytest.mark.parametrize("cam", [
    dict(image_width=1, aspect_ratio=16.0 / 9.0, samples_per_pixel=1, max_depth=5),      # (1 / 1.78) as u32 = 0 rows -> 1 (camera.rs:127-129)
    dict(image_width=3, aspect_ratio=0.05, samples_per_pixel=4, max_depth=5),             # 3 x 60: a tall sliver
    dict(image_width=40, aspect_ratio=2.0, samples_per_pixel=9, max_depth=50, defocus_angle=25.0, focus_distance=0.5, vertical_fov=150.0),
    dict(image_width=33, aspect_ratio=1.3, samples_per_pixel=2, max_depth=1, vertical_fov=1.0),  # spp 2 -> 1 stratum; one segment per path
])
def test_camera_edge_cases(pkg, gpu, oracle, cam):
    def build(be):
        sc = be.new_scene()
        pkg.scenes.random_spheres(sc, width=16, spp=1, depth=1)
        sc.set_camera(look_from=(13.0, 2.0, 3.0), look_at=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), background=(0.7, 0.8, 1.0), **cam)
        sc.commit()
        return sc
    sg, so = build(gpu), build(oracle)
    assert sg.image_size() == so.image_size() and sg.sqrt_spp() == so.sqrt_spp()
    w, h = sg.image_size()
    q = sg.sqrt_spp()
    xyss = np.array([(x, y, sx, sy) for y in range(h) for x in range(w) for sy in range(q) for sx in range(q)], dtype=np.uint32)
    rg, ro = sg.camera_rays(xyss, seed=4), so.camera_rays(xyss, seed=4)
    assert np.array_equal(rg.view(np.uint64), ro.view(np.uint64))
    img_g, _, st_g = sg.render_frame(seed=4, flags=pkg.bindings.RENDER_EXACT_SHADING)
    img_o, _, st_o = so.render_frame(seed=4, sum_dtype=np.float64)
    assert img_g.shape == img_o.shape == (h, w, 3)
    assert st_g["paths"] == st_o["paths"] == w * h * q * q and st_g["segments"] == st_o["segments"]
    diff = np.abs(img_g.astype(np.float64) - img_o)
    assert (diff > 1e-3 * (1.0 + np.abs(img_o))).mean() < 5e-3


def test_degenerate_camera_all_nan_rays(pkg, gpu, oracle):
    """`up` parallel to the view direction: camera.rs:146-148 normalises a zero vector and every ray is NaN.  NaN rays
    fail every comparison of the intersection code, so the reference shows the background; so must the GPU path."""
    out = []
    for be in (gpu, oracle):
        sc = be.new_scene()
        pkg.scenes.random_spheres(sc, width=16, spp=1, depth=1)
        sc.set_camera(image_width=24, aspect_ratio=1.5, samples_per_pixel=4, max_depth=6, look_from=(0.0, 5.0, 0.0), look_at=(0.0, 0.0, 0.0),
                      up=(0.0, 1.0, 0.0), background=(0.7, 0.8, 1.0))
        sc.commit()
        img, rgb8, st = sc.render_frame(seed=1, want_rgb8=True, **({} if be is gpu else {"sum_dtype": np.float64}))
        assert st["segments"] == st["paths"] == 24 * 16 * 4
        out.append((img, rgb8))
    np.testing.assert_allclose(out[0][0], out[1][0], rtol=1e-6)
    assert np.array_equal(out[0][1], out[1][1])
    assert np.all(out[0][1].reshape(-1, 3) == np.array([214, 228, 255]))


def test_negative_radius_sphere(pkg, gpu, oracle):
    """The hollow-glass idiom: a sphere of negative radius inside a glass sphere.  Sphere::new keeps the sign
    (sphere.rs:21-29), so the outward normal (p - c) / r points inward (sphere.rs:124)."""
    def build(be):
        sc = be.new_scene()
        glass = sc.dielectric(1.5)
        nodes = [
            sc.sphere((0.0, -100.5, -1.0), 100.0, sc.lambertian_color(0.8, 0.8, 0.0)),
            sc.sphere((0.0, 0.0, -1.0), 0.5, sc.lambertian_color(0.1, 0.2, 0.5)),
            sc.sphere((-1.0, 0.0, -1.0), 0.5, glass),
            sc.sphere((-1.0, 0.0, -1.0), -0.4, glass),
            sc.sphere((1.0, 0.0, -1.0), -0.5, sc.metal(0.8, 0.6, 0.2, 0.3), direction=(0.0, 0.3, 0.0)),
        ]
        sc.set_world(sc.bvh(nodes))
        sc.set_lights(None)
        sc.set_camera(image_width=96, aspect_ratio=16.0 / 9.0, samples_per_pixel=16, max_depth=20, vertical_fov=40, look_from=(-2.0, 2.0, 1.0),
                      look_at=(0.0, 0.0, -1.0), background=(0.7, 0.8, 1.0))
        sc.commit()
        return sc
    sg, so = build(gpu), build(oracle)
    for a, b in zip(sg.dump_tree(), so.dump_tree()):
        assert np.array_equal(a, b)
    rng = np.random.default_rng(4)
    n = 50_000
    o = np.array([-2.0, 2.0, 1.0]) + rng.normal(0.0, 0.3, (n, 3))
    d = np.stack([rng.uniform(-2, 2, n), rng.uniform(-0.6, 0.8, n), rng.uniform(-2, 0, n)], axis=1) - o
    t = rng.random(n)
    hg, ho = sg.trace_rays(o, d, t, seed=1), so.trace_rays(o, d, t, seed=1)
    info = compare_hits(hg, ho)
    assert info["n_hit"] > n // 2
    img_g, _, st_g = sg.render_frame(seed=3, flags=pkg.bindings.RENDER_EXACT_SHADING)
    img_o, _, st_o = so.render_frame(seed=3, sum_dtype=np.float64)
    assert st_g["segments"] == st_o["segments"]
    diff = np.abs(img_g.astype(np.float64) - img_o)
    assert (diff > 1e-3 * (1.0 + np.abs(img_o))).mean() < 2e-3
    fast, _, _ = sg.render_frame(seed=3)
    assert (np.abs(fast.astype(np.float64) - img_o) > 2e-3 * (1.0 + np.abs(img_o))).mean() < 1e-2


@pytest.mark.parametrize("root", ["group", "translated_bvh", "rotated_group_of_bvhs", "medium"])
def test_world_root_shapes(pkg, gpu, oracle, root):
    """SceneData.world is any Node (lib.rs:38-43), not necessarily a BoundingVolumeHierarchy: a plain Group (no
    hierarchy in the scene at all), an instance around a hierarchy, a rotated Group of hierarchies, a ConstantMedium."""
    def build(be):
        sc = be.new_scene()
        rng = np.random.default_rng(11)
        mats = [sc.lambertian_color(0.7, 0.3, 0.3), sc.metal(0.8, 0.8, 0.8, 0.1), sc.dielectric(1.5)]
        spheres = [sc.sphere(tuple(rng.uniform(-3, 3, 3)), float(rng.uniform(0.2, 0.6)), mats[i % 3]) for i in range(40)]
        if root == "group":
            world = sc.group(spheres)
        elif root == "translated_bvh":
            world = sc.translate(sc.bvh(spheres), (0.5, -0.25, 1.0))
        elif root == "rotated_group_of_bvhs":
            world = sc.rotate(sc.group([sc.bvh(spheres[:20]), sc.bvh(spheres[20:])]), (1.0, 1.0, 0.0), 33.0)
        else:
            world = sc.constant_medium_color(sc.sphere((0.0, 0.0, 0.0), 3.0, mats[2]), 0.4, 0.9, 0.9, 0.9)
        sc.set_world(world)
        sc.set_lights(None)
        sc.set_camera(image_width=64, aspect_ratio=1.0, samples_per_pixel=16, max_depth=10, vertical_fov=50, look_from=(0.0, 1.0, 10.0),
                      look_at=(0.0, 0.0, 0.0), background=(0.7, 0.8, 1.0))
        sc.commit()
        return sc
    sg, so = build(gpu), build(oracle)
    rng = np.random.default_rng(6)
    n = 20000
    o = np.array([0.0, 1.0, 10.0]) + rng.normal(0.0, 0.5, (n, 3))
    d = rng.uniform(-4, 4, (n, 3)) - o
    info = compare_hits(sg.trace_rays(o, d, None, seed=2), so.trace_rays(o, d, None, seed=2))
    assert info["n_hit"] > n // 10
    img_g, _, st_g = sg.render_frame(seed=3, flags=pkg.bindings.RENDER_EXACT_SHADING)
    img_o, _, st_o = so.render_frame(seed=3, sum_dtype=np.float64)
    assert st_g["segments"] == st_o["segments"]
    diff = np.abs(img_g.astype(np.float64) - img_o)
    assert (diff > 1e-3 * (1.0 + np.abs(img_o))).mean() < 2e-3
    a, _, sa = sg.render_frame(seed=3)
    b, _, sb = sg.render_frame(seed=3, flags=pkg.bindings.RENDER_REFERENCE_ORDER)
    assert sa["segments"] == sb["segments"] and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_shared_subgraphs_instanced_several_times(pkg, gpu, oracle):
    """The scene graph is a DAG in the reference (Arc<dyn Node> clones): one hierarchy of spheres and one box placed
    three times each through different instances, and one sphere that is a member of two hierarchies."""
    def build(be):
        sc = be.new_scene()
        rng = np.random.default_rng(21)
        mats = [sc.lambertian_color(0.2, 0.6, 0.3), sc.metal(0.9, 0.7, 0.4, 0.05), sc.dielectric(1.5), sc.lambertian_tex(sc.tex_checker(0.4, sc.tex_solid(0.1, 0.1, 0.1), sc.tex_solid(0.9, 0.9, 0.9)))]
        cluster = sc.bvh([sc.sphere(tuple(rng.uniform(-0.8, 0.8, 3)), float(rng.uniform(0.1, 0.3)), mats[i % 4]) for i in range(24)])
        box = sc.box((-0.4, -0.4, -0.4), (0.4, 0.4, 0.4), mats[3])
        shared = sc.sphere((0.0, 0.0, 0.0), 0.5, mats[1])
        world = [
            sc.sphere((0.0, -1001.0, 0.0), 1000.0, mats[3]),
            sc.translate(cluster, (-2.5, 0.0, 0.0)),
            sc.translate(sc.rotate(cluster, (0.0, 0.0, 1.0), 90.0), (0.0, 0.0, 0.0)),
            sc.translate(sc.scale(cluster, 1.5, 0.5, 1.0), (2.5, 0.0, 0.0)),
            sc.translate(box, (-2.5, 1.6, 0.0)),
            sc.translate(sc.rotate_y(box, 45.0), (0.0, 1.6, 0.0)),
            sc.rotate(sc.translate(box, (2.5, 1.6, 0.0)), (1.0, 0.0, 0.0), 10.0),
            sc.bvh([shared, sc.translate(shared, (0.0, 0.0, 2.0))]),
            sc.translate(sc.bvh([shared]), (0.0, 0.0, -2.5)),
        ]
        sc.set_world(sc.bvh(world))
        sc.set_lights(None)
        sc.set_camera(image_width=96, aspect_ratio=1.5, samples_per_pixel=16, max_depth=12, vertical_fov=40, look_from=(0.0, 2.5, 9.0),
                      look_at=(0.0, 0.5, 0.0), background=(0.7, 0.8, 1.0))
        sc.commit()
        return sc
    sg, so = build(gpu), build(oracle)
    for a, b in zip(sg.dump_tree(), so.dump_tree()):
        assert np.array_equal(a, b)
    rng = np.random.default_rng(6)
    n = 40000
    o = np.array([0.0, 2.5, 9.0]) + rng.normal(0.0, 0.3, (n, 3))
    d = np.stack([rng.uniform(-4, 4, n), rng.uniform(-1, 2.5, n), rng.uniform(-3, 3, n)], axis=1) - o
    info = compare_hits(sg.trace_rays(o, d, None, seed=2), so.trace_rays(o, d, None, seed=2))
    assert info["n_hit"] > n // 2
    img_g, _, st_g = sg.render_frame(seed=3, flags=pkg.bindings.RENDER_EXACT_SHADING)
    img_o, _, st_o = so.render_frame(seed=3, sum_dtype=np.float64)
    assert st_g["segments"] == st_o["segments"]
    diff = np.abs(img_g.astype(np.float64) - img_o)
    assert (diff > 1e-3 * (1.0 + np.abs(img_o))).mean() < 2e-3


@pytest.mark.parametrize("spp", [10000, 65536 + 600])
def test_many_strata_few_pixels(pkg, gpu, oracle, spp):
    """100 x 100 and 257 x 257 strata on a handful of pixels (the Next-Week final configuration's 10 000 spp): the stratum
    index arithmetic (sample -> (s_x, s_y), camera.rs:335-341) at sizes the other tests do not reach."""
    def build(be):
        sc = be.new_scene()
        pkg.scenes.three_spheres(sc, width=16, spp=1, depth=1)
        sc.set_camera(image_width=3, aspect_ratio=1.5, samples_per_pixel=spp, max_depth=6, vertical_fov=40, look_from=(-2.0, 2.0, 1.0),
                      look_at=(0.0, 0.0, -1.0), background=(0.7, 0.8, 1.0), defocus_angle=4.0, focus_distance=3.4)
        sc.commit()
        return sc
    sg, so = build(gpu), build(oracle)
    q = sg.sqrt_spp()
    assert q == so.sqrt_spp() == int(np.floor(np.sqrt(spp)))
    w, h = sg.image_size()
    rng = np.random.default_rng(1)
    k = 20000
    xyss = np.stack([rng.integers(0, w, k), rng.integers(0, h, k), rng.integers(0, q, k), rng.integers(0, q, k)], axis=1).astype(np.uint32)
    xyss[:4] = [(0, 0, 0, 0), (w - 1, h - 1, q - 1, q - 1), (0, h - 1, q - 1, 0), (w - 1, 0, 0, q - 1)]
    assert np.array_equal(sg.camera_rays(xyss, seed=9).view(np.uint64), so.camera_rays(xyss, seed=9).view(np.uint64))
    img_g, _, st_g = sg.render_frame(seed=9, flags=pkg.bindings.RENDER_EXACT_SHADING)
    img_o, _, st_o = so.render_frame(seed=9, sum_dtype=np.float64)
    assert st_g["paths"] == st_o["paths"] == w * h * q * q and st_g["segments"] == st_o["segments"]
    # a pixel is the sum of tens of thousands of samples here; the few paths per pixel that part ways with the oracle's
    # (last-bit differences between the device's and glibc's libm, amplified by the bounces) move it by parts in 10^4
    np.testing.assert_allclose(img_g, img_o, rtol=2e-3)


def test_textures_on_every_primitive_under_instances(pkg, gpu, oracle):
    """Noise, turbulence, nested checker and image textures (texture/*.rs) read (u, v, p) of hit records that come back
    through Translate / Rotate / Scale: on a quad, a box, a disc, a cone frustum and an ellipsoid; also as the emission
    of a light and as the albedo of a medium."""
    S = pkg.scenes

    def build(be):
        sc = be.new_scene()
        rng = S.SceneRandom(5)
        noise = sc.tex_perlin_noise(3.0, S.perlin_tables(rng))
        turb = sc.tex_perlin_turbulence(1.5, 5, S.perlin_tables(rng))
        img = sc.tex_image(S.earth_texels()[::8, ::8].copy())
        inner = sc.tex_checker(0.25, noise, sc.tex_solid(0.9, 0.2, 0.2))
        nested = sc.tex_checker(1.0, inner, img)
        m_noise, m_turb, m_img, m_nested = (sc.lambertian_tex(t) for t in (noise, turb, img, nested))
        glow = sc.diffuse_light_tex(turb)
        world = [
            sc.quad((-6.0, -1.0, -6.0), (12.0, 0.0, 0.0), (0.0, 0.0, 12.0), m_nested),
            sc.translate(sc.rotate_y(sc.box((-0.6, -0.6, -0.6), (0.6, 0.6, 0.6), m_img), 30.0), (-2.5, 0.0, 0.0)),
            sc.translate(sc.scale(sc.sphere((0.0, 0.0, 0.0), 1.0, m_turb), 1.4, 0.7, 1.0), (0.0, 0.0, 0.0)),
            sc.rotate(sc.translate(sc.disc((0.0, 0.0, 0.0), 1.0, (0.0, 1.0, 0.0), m_noise), (2.5, 0.2, 0.0)), (0.0, 0.0, 1.0), 12.0),
            sc.translate(sc.cone_frustum((0.0, -1.0, 0.0), 1.5, 0.2, 0.7, m_img), (0.0, 0.0, -2.5)),
            sc.quad((-1.0, 3.0, -1.0), (2.0, 0.0, 0.0), (0.0, 0.0, 2.0), glow),
            sc.constant_medium_tex(sc.sphere((2.5, 1.5, -2.0), 0.9, m_img), 1.5, nested),
            sc.sphere((-2.5, 1.5, -2.0), 0.8, m_nested, direction=(0.0, 0.4, 0.0)),
        ]
        sc.set_world(sc.bvh(world))
        sc.set_lights(sc.group([world[5]]))
        sc.set_camera(image_width=96, aspect_ratio=1.5, samples_per_pixel=16, max_depth=10, vertical_fov=45, look_from=(0.0, 3.0, 9.0),
                      look_at=(0.0, 0.3, 0.0), background=(0.3, 0.35, 0.45))
        sc.commit()
        return sc
    sg, so = build(gpu), build(oracle)
    img_g, _, st_g = sg.render_frame(seed=3, flags=pkg.bindings.RENDER_EXACT_SHADING)
    img_o, _, st_o = so.render_frame(seed=3, sum_dtype=np.float64)
    assert st_g["segments"] == st_o["segments"]
    diff = np.abs(img_g.astype(np.float64) - img_o)
    assert (diff > 1e-3 * (1.0 + np.abs(img_o))).mean() < 3e-3
    fast, _, _ = sg.render_frame(seed=3)
    assert (np.abs(fast.astype(np.float64) - img_o) > 2e-3 * (1.0 + np.abs(img_o))).mean() < 1.5e-2


@pytest.mark.gpu
def test_progressive_blocks_end_in_the_frame(pkg, gpu, tmp_path):
    """chip_render_frame_blocks: the tiles of crates/cli/src/main.rs:83-106 (10 x 10, clipped at the edges, last pushed
    first), refined over passes; the last pass's tiles assemble chip_render_frame's bytes, and the PNG of that frame
    reads back the same."""
    from PIL import Image

    sc = pkg.scenes.build(gpu, "cornell_box", width=75, spp=16, depth=8)
    w, h = sc.image_size()
    _, want, _ = sc.render_frame(seed=6, want_rgb8=True)
    seen, frames = [], {}

    def on_block(xmin, xmax, ymin, ymax, rgb8, done, total):
        seen.append((xmin, xmax, ymin, ymax, done, total))
        frames.setdefault(done, np.zeros((h, w, 3), dtype=np.uint8))[ymin:ymax, xmin:xmax] = rgb8

    st = sc.render_frame_blocks(on_block, block_size=10, passes=4, seed=6)
    assert st["paths"] == w * h * 16
    assert sorted(frames) == [4, 8, 12, 16] and all(t == 16 for *_, t in seen)
    per_pass = [b for b in seen if b[4] == 4]
    assert len(per_pass) == 8 * 8 and per_pass[0][:4] == (70, 75, 70, 75) and per_pass[-1][:4] == (0, 10, 0, 10)
    assert np.array_equal(frames[16], want)
    # earlier passes show the same picture with fewer samples
    err = [np.abs(frames[d].astype(int) - want.astype(int)).mean() for d in (4, 8, 12, 16)]
    assert err[0] > err[1] > err[2] > err[3] == 0.0 and err[0] < 40
    path = str(tmp_path / "out.png")
    pkg.write_png(path, frames[16])
    assert np.array_equal(np.asarray(Image.open(path).convert("RGB")), want)
    # a rectangle with the wasm caller's quantisation
    rect = (12, 40, 30, 55)
    _, want_r, _ = sc.render_frame(seed=6, rect=rect, want_rgb8=True, flags=pkg.bindings.RENDER_QUANTISE_255)
    got_r = np.zeros_like(want_r)

    def on_rect(xmin, xmax, ymin, ymax, rgb8, done, total):
        if done == total:
            got_r[ymin - rect[2]:ymax - rect[2], xmin - rect[0]:xmax - rect[0]] = rgb8

    sc.render_frame_blocks(on_rect, block_size=16, passes=2, seed=6, rect=rect, flags=pkg.bindings.RENDER_QUANTISE_255)
    assert np.array_equal(got_r, want_r)


@pytest.mark.gpu
def test_compact_sphere_form_equals_general_form(pkg, gpu, monkeypatch):
    """Small scenes of bare spheres run a form of the traversal kernel with 11-bit node references and 4-byte stack
    words (k_extend_small); a ray whose descent needs more stack than LDS holds is handed to the literal pass.  The
    image must not depend on which form ran or on how many rays took the hand-over: the closest hit of a ray is the
    same however it is found (ties by the reference's order)."""
    monkeypatch.delenv("CHIP_NO_SMALL_FORM", raising=False)
    monkeypatch.delenv("CHIP_SMALL_STACK", raising=False)
    sc = pkg.scenes.build(gpu, "random_spheres", width=160, spp=16, depth=12)
    want, _, st0 = sc.render_frame(seed=5)
    monkeypatch.setenv("CHIP_NO_SMALL_FORM", "1")
    general, _, st1 = sc.render_frame(seed=5)
    monkeypatch.delenv("CHIP_NO_SMALL_FORM")
    assert np.array_equal(want, general)
    assert st0["segments"] == st1["segments"] and st0["bvh_visits"] != 0
    # four stack words per lane: most descents overflow and go through the literal pass
    monkeypatch.setenv("CHIP_SMALL_STACK", "4")
    tight, _, st2 = sc.render_frame(seed=5)
    assert np.array_equal(want, tight)
    assert st2["segments"] == st0["segments"]


@pytest.mark.gpu
def test_first_bounce_in_one_kernel_equals_two_kernel_form(pkg, gpu, oracle, monkeypatch):
    """Small sphere scenes with solid colours trace and shade bounce 0 in one kernel (k_first_small: camera ray,
    closest hit and material code per wave, bounce-1 entries handed out in ranges); CHIP_NO_FUSED_FIRST=1 selects the
    traversal kernel followed by the bounce-0 shade kernel.  Both forms run the same arithmetic per path, so the images
    are the same bits in both shading modes, also when a batch leaves empty entries in bounce 1 (a frame large enough for
    ranged hand-out) and with four stack words per lane, when most camera rays go through the literal queue and are
    shaded by the shade kernel's pass over that queue.  The one-kernel form is also compared with the oracle."""
    monkeypatch.delenv("CHIP_NO_FUSED_FIRST", raising=False)
    X = pkg.bindings.RENDER_EXACT_SHADING
    for name, kw, seed in (("random_spheres", dict(width=160, spp=16, depth=12), 5), ("three_spheres", dict(width=120, spp=9, depth=1), 2),
                           ("three_spheres", dict(width=120, spp=9, depth=2), 2), ("random_spheres", dict(width=1920, spp=4, depth=6), 3)):
        sc = pkg.scenes.build(gpu, name, **kw)
        for flags in (0, X):
            one, _, st1 = sc.render_frame(seed=seed, flags=flags)
            monkeypatch.setenv("CHIP_NO_FUSED_FIRST", "1")
            two, _, st2 = sc.render_frame(seed=seed, flags=flags)
            monkeypatch.delenv("CHIP_NO_FUSED_FIRST")
            assert st1["segments"] == st2["segments"] and st1["paths"] == st2["paths"], (name, kw, flags)
            assert np.array_equal(one.view(np.uint32), two.view(np.uint32)), (name, kw, flags)
    sc = pkg.scenes.build(gpu, "random_spheres", width=160, spp=16, depth=12)
    want, _, st0 = sc.render_frame(seed=5)
    monkeypatch.setenv("CHIP_SMALL_STACK", "4")
    tight, _, st1 = sc.render_frame(seed=5)
    monkeypatch.delenv("CHIP_SMALL_STACK")
    assert st0["segments"] == st1["segments"] and np.array_equal(want.view(np.uint32), tight.view(np.uint32))
    so = pkg.scenes.build(oracle, "random_spheres", width=96, spp=4, depth=12)
    sg = pkg.scenes.build(gpu, "random_spheres", width=96, spp=4, depth=12)
    img_o, _, st_o = so.render_frame(seed=7, sum_dtype=np.float64)
    img_g, _, st_g = sg.render_frame(seed=7, flags=X)
    assert st_g["segments"] == st_o["segments"]
    diff = np.abs(img_g.astype(np.float64) - img_o)
    assert (diff > 1e-3 * (1.0 + np.abs(img_o))).mean() < 2e-3


@pytest.mark.gpu
def test_flat_form_equals_the_hierarchy_walk(pkg, gpu, oracle, monkeypatch):
    """A world of at most 32 spheres and quads under Translates and Rotates (the Cornell box, the quads scene, a sphere
    or two with a light) is traced by trying every primitive for every ray, all lanes in step (k_extend_flat);
    CHIP_NO_FLAT_FORM=1 keeps the walk through the culling trees.  The closest hit and its tie rule are the same, so
    images and segment counts are the same bits in both shading modes; the flat form is also compared with the oracle."""
    monkeypatch.delenv("CHIP_NO_FLAT_FORM", raising=False)
    X = pkg.bindings.RENDER_EXACT_SHADING
    for name, kw, seed in (("cornell_box", dict(width=120, spp=16, depth=12), 3), ("quads", dict(width=120, spp=9, depth=8), 2),
                           ("lighted_sphere", dict(width=120, spp=9, depth=8), 4), ("earth", dict(width=96, spp=4, depth=4), 1),
                           ("perlin_spheres", dict(width=96, spp=4, depth=6), 5), ("cornell_box", dict(width=600, spp=4, depth=50), 1)):
        sc = pkg.scenes.build(gpu, name, **kw)
        for flags in (0, X):
            flat, _, st1 = sc.render_frame(seed=seed, flags=flags)
            monkeypatch.setenv("CHIP_NO_FLAT_FORM", "1")
            walk, _, st2 = sc.render_frame(seed=seed, flags=flags)
            monkeypatch.delenv("CHIP_NO_FLAT_FORM")
            assert st1["segments"] == st2["segments"] and st1["paths"] == st2["paths"], (name, kw, flags)
            # (the flat form is the one that ran; the earth scene is one bare sphere, with no hierarchy to visit either way)
            assert st1["bvh_visits"] == 0 and (st2["bvh_visits"] > 0 or name == "earth"), (name, kw, flags)
            assert np.array_equal(flat.view(np.uint32), walk.view(np.uint32)), (name, kw, flags)
    so = pkg.scenes.build(oracle, "cornell_box", width=64, spp=4, depth=10)
    sg = pkg.scenes.build(gpu, "cornell_box", width=64, spp=4, depth=10)
    img_o, _, st_o = so.render_frame(seed=7, sum_dtype=np.float64)
    img_g, _, st_g = sg.render_frame(seed=7, flags=X)
    assert st_g["segments"] == st_o["segments"] and st_g["bvh_visits"] == 0
    diff = np.abs(img_g.astype(np.float64) - img_o)
    assert (diff > 1e-3 * (1.0 + np.abs(img_o))).mean() < 2e-3


@pytest.mark.gpu
def test_shading_from_lds_tables_equals_shading_from_memory(pkg, gpu, oracle, monkeypatch):
    """A scene of a few primitives, materials and lights with plain textures (the Cornell boxes, the quads scene, spheres
    under a light, checkered spheres, OpenSCAD-shaped scenes with discs, cone frusta and Scale) is shaded from copies of its record tables in LDS; CHIP_NO_SHADE_TABLES=1 leaves
    them in memory.  Same records, same arithmetic: the frames are the same bits in both shading modes, at bounce 0 and
    later; a scene whose tables do not fit (the Next-Week scene) takes the other kernel either way."""
    monkeypatch.delenv("CHIP_NO_SHADE_TABLES", raising=False)
    X = pkg.bindings.RENDER_EXACT_SHADING
    for name, kw, seed in (("cornell_box", dict(width=120, spp=16, depth=12), 3), ("cornell_box_smoke", dict(width=120, spp=16, depth=12), 5),
                           ("quads", dict(width=120, spp=9, depth=8), 2), ("lighted_sphere", dict(width=120, spp=9, depth=8), 4),
                           ("checkered_spheres", dict(width=120, spp=9, depth=8), 6), ("final", dict(width=96, spp=4, depth=8), 1),
                           ("scad_shapes", dict(width=120, spp=9, depth=8), 7), ("lighted_cone_frustum", dict(width=120, spp=9, depth=8), 8),
                           ("scad_shapes_cone_light", dict(width=120, spp=9, depth=8), 9)):
        sc = pkg.scenes.build(gpu, name, **kw)
        for flags in (0, X, X | pkg.bindings.RENDER_REFERENCE_ORDER):
            lds, _, st1 = sc.render_frame(seed=seed, flags=flags)
            monkeypatch.setenv("CHIP_NO_SHADE_TABLES", "1")
            mem, _, st2 = sc.render_frame(seed=seed, flags=flags)
            monkeypatch.delenv("CHIP_NO_SHADE_TABLES")
            assert st1["segments"] == st2["segments"] and st1["paths"] == st2["paths"], (name, kw, flags)
            assert np.array_equal(lds.view(np.uint32), mem.view(np.uint32)), (name, kw, flags)
    for name in ("cornell_box", "cornell_box_smoke"):
        so = pkg.scenes.build(oracle, name, width=64, spp=4, depth=10)
        sg = pkg.scenes.build(gpu, name, width=64, spp=4, depth=10)
        img_o, _, st_o = so.render_frame(seed=7, sum_dtype=np.float64)
        img_g, _, st_g = sg.render_frame(seed=7, flags=X)
        assert st_g["segments"] == st_o["segments"]
        diff = np.abs(img_g.astype(np.float64) - img_o)
        assert (diff > 1e-3 * (1.0 + np.abs(img_o))).mean() < 2e-3


@pytest.mark.gpu
def test_tail_of_a_batch_in_one_kernel_equals_bounce_by_bounce(pkg, gpu, oracle, monkeypatch):
    """Once a bounce of a small sphere scene holds few enough entries (CHIP_TAIL_PATHS, 4 M by default), one kernel follows
    those paths to their ends: it traces, shades in place and keeps the scattered ray in registers (k_tail_small).  Every
    sample must be the bit pattern the bounce-by-bounce kernels produce (CHIP_TAIL_PATHS=0), in both shading modes, with
    most descents overflowing the compact form's stack (the literal form inside the kernel then traces them) and at depths
    that end paths inside the kernel by camera.rs:259-261; the frame is also compared with the oracle's."""
    X = pkg.bindings.RENDER_EXACT_SHADING
    for name, kw, seed in (("random_spheres", dict(width=400, spp=9, depth=50), 1), ("random_spheres", dict(width=160, spp=16, depth=7), 2),
                           ("three_spheres", dict(width=200, spp=9, depth=50), 3), ("random_spheres", dict(width=1920, spp=4, depth=50), 4)):
        sc = pkg.scenes.build(gpu, name, **kw)
        for stack in (None, "4"):
            if stack:
                monkeypatch.setenv("CHIP_SMALL_STACK", stack)
            for flags in (0, X):
                monkeypatch.delenv("CHIP_TAIL_PATHS", raising=False)
                tail, _, st1 = sc.render_frame(seed=seed, flags=flags)
                monkeypatch.setenv("CHIP_TAIL_PATHS", "0")
                plain, _, st2 = sc.render_frame(seed=seed, flags=flags)
                monkeypatch.setenv("CHIP_TAIL_PATHS", "3000")  # (the kernel takes over late, with a handful of paths)
                late, _, st3 = sc.render_frame(seed=seed, flags=flags)
                monkeypatch.delenv("CHIP_TAIL_PATHS")
                assert st1["segments"] == st2["segments"] == st3["segments"] and st1["paths"] == st2["paths"], (name, kw, stack, flags)
                assert np.array_equal(tail.view(np.uint32), plain.view(np.uint32)), (name, kw, stack, flags)
                assert np.array_equal(late.view(np.uint32), plain.view(np.uint32)), (name, kw, stack, flags)
            monkeypatch.delenv("CHIP_SMALL_STACK", raising=False)
    so = pkg.scenes.build(oracle, "random_spheres", width=96, spp=4, depth=50)
    sg = pkg.scenes.build(gpu, "random_spheres", width=96, spp=4, depth=50)
    img_o, _, st_o = so.render_frame(seed=7, sum_dtype=np.float64)
    img_g, _, st_g = sg.render_frame(seed=7, flags=X)
    assert st_g["segments"] == st_o["segments"]
    diff = np.abs(img_g.astype(np.float64) - img_o)
    assert (diff > 1e-3 * (1.0 + np.abs(img_o))).mean() < 2e-3


@pytest.mark.gpu
def test_literal_hand_over_of_every_fifth_ray(pkg, gpu):
    """The ordered traversal hands a ray it cannot take (a zero direction component, a descent deeper than the compact
    form's stack) to the literal traversal, which runs as a pass of its own after it: a ray or two per frame in the
    book scenes.  The test build of the library (csrc/Makefile: libcaustic_hip_handover.so) hands every fifth ray
    over, in both forms of the kernel (the LDS-resident one of the early bounces and the plain one of the short late
    queues) and at bounce 0 as well: the image and the segment count must be those of the product library, since the
    closest hit of a ray does not depend on the order in which the tree was searched (ties by the reference's
    order)."""
    import ctypes as C
    import os

    path = os.path.join(os.path.dirname(pkg.bindings.GPU_LIB_PATH), "libcaustic_hip_handover.so")
    assert os.path.exists(path), "build it with make -C rust-raytracer_amd/csrc"
    forced = pkg.bindings.Backend(C.CDLL(path), "chip_", "libcaustic_hip_handover")
    for name, kw in (("random_spheres", dict(width=160, spp=16, depth=12)), ("random_spheres", dict(width=48, spp=4, depth=8)),
                     ("three_spheres", dict(width=200, spp=9, depth=10)), ("checkered_spheres", dict(width=64, spp=4, depth=10))):
        a = pkg.scenes.build(gpu, name, **kw)
        b = pkg.scenes.build(forced, name, **kw)
        want, _, st_a = a.render_frame(seed=11)
        got, _, st_b = b.render_frame(seed=11)
        assert np.array_equal(want, got), (name, kw)
        assert st_a["segments"] == st_b["segments"]

