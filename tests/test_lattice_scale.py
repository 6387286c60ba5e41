"""The OpenSCAD-shaped scene at its full size: a top-level BoundingVolumeHierarchy over 100^3 = one million
`Translate(Group[Sphere])` instances plus a ground sphere, which is what crates/openscad's interpreter emits for
nested `for` loops of `translate() sphere()` (crates/openscad/src/interpreter/modules.rs:139-163,274; mod.rs:190).
At this size the culling nodes, instances and spheres (64 MB each) come from HBM, node indices need 20 of their 28
bits and the scene arrives through one chip_scene_from_flat call; the oracle builds the same graph from the same
records (orc_scene_from_flat) and answers the same rays on the CPU."""
import numpy as np
import pytest

from helpers import compare_hits

N = 100
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]


@pytest.fixture(scope="module")
def lattice(pkg, gpu, oracle):
    def build(be):
        sc = be.new_scene()
        pkg.scenes.scad_lattice(sc, n=N, width=1920, spp=4, depth=50)
        sc.commit()
        return sc
    return build(gpu), build(oracle)


def _rays(sc, n_each, seed=11):
    """Camera rays of a pixel subsample, rays from random points inside and around the lattice (most start between
    spheres, a few inside one), rays that leave upwards (misses), and axis-parallel rays."""
    rng = np.random.default_rng(seed)
    w, h = sc.image_size()
    xy = np.stack([rng.integers(0, w, n_each), rng.integers(0, h, n_each), rng.integers(0, 2, n_each), rng.integers(0, 2, n_each)], axis=1)
    cam = sc.camera_rays(xy.astype(np.uint32), seed=5)
    lo, hi = np.array([-N / 2.0, 0.0, -N / 2.0]), np.array([N / 2.0, float(N), N / 2.0])
    o1 = lo - 2.0 + rng.random((n_each, 3)) * (hi - lo + 4.0)
    d1 = rng.normal(size=(n_each, 3)) * rng.uniform(0.25, 4.0, (n_each, 1))
    o2 = lo + rng.random((n_each, 3)) * (hi - lo)
    d2 = np.abs(rng.normal(size=(n_each, 3))) * np.array([0.2, 1.0, 0.2]) + 1e-3
    k = n_each // 8
    o3 = lo + rng.random((k, 3)) * (hi - lo)
    d3 = np.zeros((k, 3))
    d3[np.arange(k), rng.integers(0, 3, k)] = rng.choice([-1.0, 1.0], k)
    o = np.vstack([cam[:, 0:3], o1, o2, o3])
    d = np.vstack([cam[:, 3:6], d1, d2, d3])
    return o, d


def test_million_instance_tree_equals_oracle(lattice):
    sg, so = lattice
    (kg, ig, bg), (ko, io, bo) = sg.dump_tree(), so.dump_tree()
    assert kg.shape[0] > 4 * N ** 3  # hierarchy nodes + Translate + Group + Sphere records
    assert np.array_equal(kg, ko) and np.array_equal(ig, io)
    assert np.array_equal(bg.view(np.uint64), bo.view(np.uint64))


def test_million_instance_hits_equal_oracle(lattice):
    sg, so = lattice
    o, d = _rays(sg, 80_000)
    assert o.shape[0] >= 200_000
    hg, ho = sg.trace_rays(o, d, None, seed=1), so.trace_rays(o, d, None, seed=1)
    info = compare_hits(hg, ho)
    assert info["t_bit_identical"]
    m = ho["hit"] == 1
    assert np.array_equal(hg["pt"][m].view(np.uint64), ho["pt"][m].view(np.uint64))
    assert np.array_equal(hg["normal"][m].view(np.uint64), ho["normal"][m].view(np.uint64))
    assert info["n_hit"] > 100_000 and (~m).sum() > 1_000
    assert np.unique(ho["leaf_id"][m]).size > 20_000  # the rays reach all over the lattice


def test_million_instance_render_ordered_equals_reference_order(pkg, lattice):
    """1920 x 1080, 4 strata, depth 50: the ordered traversal (f32 culling nodes from HBM) and the literal replay of the
    reference's visiting order give the same image bit for bit."""
    sg, _ = lattice
    assert sg.image_size() == (1920, 1080)
    a, _, sa = sg.render_frame(seed=3)
    b, _, sb = sg.render_frame(seed=3, flags=pkg.bindings.RENDER_REFERENCE_ORDER)
    assert sa["paths"] == sb["paths"] == 1920 * 1080 * 4
    assert sa["segments"] == sb["segments"]
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.isfinite(a).all() and a.min() >= 0.0 and a.mean() > 0.05


def test_million_instance_rect_and_rank_split(lattice):
    sg, _ = lattice
    a, _, sa = sg.render_frame(seed=3)
    rect = (700, 1500, 300, 900)
    c, _, _ = sg.render_frame(seed=3, rect=rect)
    assert np.array_equal(c.view(np.uint32), a[rect[2]:rect[3], rect[0]:rect[1]].view(np.uint32))
    r0, _, s0 = sg.render_frame(seed=3, stratum_offset=0, stratum_stride=2)
    r1, _, s1 = sg.render_frame(seed=3, stratum_offset=1, stratum_stride=2)
    assert s0["segments"] + s1["segments"] == sa["segments"]
    assert np.allclose(r0 + r1, a, rtol=1e-5, atol=1e-5)  # the same samples, added in another order in f32


def _same_seed_window(pkg, sg, so):
    rect = (900, 996, 500, 564)
    e, _, se = sg.render_frame(seed=2, rect=rect, flags=pkg.bindings.RENDER_EXACT_SHADING)
    o_img, _, so_st = so.render_frame(seed=2, rect=rect, sum_dtype=np.float64)
    diff = np.abs(e.astype(np.float64) - o_img)
    bad = (diff > 1e-3 * (1.0 + np.abs(o_img))).any(axis=-1).mean()
    return se["segments"], so_st["segments"], float(bad), float(e.mean()), float(o_img.mean())


def test_million_instance_same_seed_render_equals_oracle(pkg, lattice):
    """A window of the full-size frame rendered with the reference's f64 shading order against the oracle, same seed.
    Inside the lattice a path keeps bouncing between spheres (a Sinai billiard), and every diffuse bounce takes its
    direction through sin / cos, whose last bit differs between the device's and the host's math library: a difference
    of 1e-16 doubles or more at every bounce off a small sphere, so at depth 50 a small share of the paths end up
    somewhere else on the two sides.  The test therefore allows that share here and asks for sample-by-sample
    equality at a depth where the amplification has no room (next test)."""
    sg, so = lattice
    seg_g, seg_o, bad, mean_g, mean_o = _same_seed_window(pkg, sg, so)
    assert abs(seg_g - seg_o) <= 5e-3 * seg_o
    assert bad < 0.05
    assert abs(mean_g - mean_o) <= 0.02 * mean_o


def _set_camera(scenes, spp, depth):
    d = 1.6 * N
    for sc in scenes:
        sc.set_camera(image_width=1920, samples_per_pixel=spp, max_depth=depth, aspect_ratio=16.0 / 9.0, vertical_fov=40.0, look_from=(d, 0.9 * N, d),
                      look_at=(0.0, 0.45 * N, 0.0), up=(0.0, 1.0, 0.0), defocus_angle=0.0, background=(0.7, 0.8, 1.0))
        sc.commit()


def test_million_instance_full_configuration(pkg, lattice):
    """The configuration in full: 1920 x 1080, 256 strata, depth 50 (0.53 G paths).  (From here on the tests of this
    module replace the camera of the shared scenes.)  A window of the frame against the oracle's same-seed render of
    it: the paths that end up elsewhere on the two sides (see above) are each 1/256 of a pixel now.  And the frame is
    the sum of the eight rank shares of the 8-GPU job, segment for segment."""
    sg, so = lattice
    _set_camera((sg, so), 256, 50)
    full, _, st = sg.render_frame(seed=2)
    assert st["paths"] == 1920 * 1080 * 256
    assert np.isfinite(full).all() and full.min() >= 0.0
    rect = (900, 996, 500, 564)
    want, _, _ = so.render_frame(seed=2, rect=rect, sum_dtype=np.float64)
    got = full[rect[2]:rect[3], rect[0]:rect[1]].astype(np.float64)
    rel = np.abs(got - want) / (1.0 + np.abs(want))
    # measured: median 1.3e-5, 90th percentile 3.8e-4, 99th 3.4e-3, largest 5.6e-2 relative; window mean 2.6e-6
    assert np.median(rel) <= 2e-4 and np.quantile(rel, 0.99) <= 2e-2, (float(np.median(rel)), float(np.quantile(rel, 0.99)), float(rel.max()))
    assert abs(got.mean() - want.mean()) <= 1e-4 * want.mean()
    acc = np.zeros(full.shape, dtype=np.float64)
    segments = 0
    for r in range(8):
        part, _, sr = sg.render_frame(seed=2, stratum_offset=r, stratum_stride=8)
        acc += part
        segments += sr["segments"]
    assert segments == st["segments"]
    assert np.allclose(acc, full, rtol=1e-4, atol=1e-3)


def test_million_instance_same_seed_render_equals_oracle_at_depth_6(pkg, lattice):
    sg, so = lattice
    _set_camera((sg, so), 4, 6)
    seg_g, seg_o, bad, _, _ = _same_seed_window(pkg, sg, so)
    assert seg_g == seg_o
    assert bad < 2e-3
