"""A ConstantMedium draws a fresh free-flight sample every time it is tested (constant_medium.rs:41-80), and a
BoundingVolumeHierarchy built over a single object holds that object as both children and visits both
(bvh.rs:22-26, 58-76).  A medium that is alone in a hierarchy is therefore tested twice per world.hit and
scatters as if it were twice as dense; the oracle and the device both have to reproduce that.
Parity unpinned above the math level: the reference has no test for this; the expected transmission follows
from the two cited pieces of code."""
import numpy as np
import pytest


def _scene(be, alone: bool, density=0.5):
    sc = be.new_scene()
    boundary = sc.sphere((0.0, 0.0, 0.0), 1.0, sc.lambertian_color(0.5, 0.5, 0.5))
    medium = sc.constant_medium_color(boundary, density, 1.0, 1.0, 1.0)
    members = [medium] if alone else [medium, sc.sphere((0.0, 50.0, 0.0), 1.0, sc.lambertian_color(0.5, 0.5, 0.5))]
    sc.set_world(sc.bvh(members))
    sc.set_camera(image_width=8, samples_per_pixel=1, max_depth=2, look_from=(0, 0, 20), look_at=(0, 0, 0), background=(1, 1, 1))
    sc.commit()
    return sc


def _rays(n):
    # through the centre of the unit sphere along +z: 2 units inside the boundary
    o = np.tile(np.array([[0.0, 0.0, -5.0]]), (n, 1))
    d = np.tile(np.array([[0.0, 0.0, 1.0]]), (n, 1))
    # generic directions (no zero component) that still pass close to the centre
    rng = np.random.default_rng(5)
    d[:, :2] = rng.uniform(1e-3, 2e-3, (n, 2))
    return o, d


@pytest.mark.parametrize("alone,expected", [(True, np.exp(-2.0)), (False, np.exp(-1.0))])
def test_oracle_transmission_through_medium(oracle, alone, expected):
    n = 40_000
    o, d = _rays(n)
    hits = _scene(oracle, alone).trace_rays(o, d, seed=3)
    transmitted = 1.0 - hits["hit"].mean()
    assert abs(transmitted - expected) < 4.0 * np.sqrt(expected * (1.0 - expected) / n) + 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("alone", [True, False])
def test_gpu_medium_hits_equal_oracle(oracle, gpu, alone):
    n = 40_000
    o, d = _rays(n)
    want = _scene(oracle, alone).trace_rays(o, d, seed=3)
    got = _scene(gpu, alone).trace_rays(o, d, seed=3)
    assert np.array_equal(got["hit"], want["hit"])
    m = want["hit"] == 1
    assert m.any() and (~m).any()
    # the free-flight distance goes through log(): the device's and the host's differ in the last bits
    assert np.abs(got["t"][m] - want["t"][m]).max() <= 1e-12 * np.abs(want["t"][m]).max()


@pytest.mark.gpu
def test_gpu_single_member_and_empty_hierarchies(oracle, gpu):
    """Hierarchies over one object and over none (bvh.rs:22-28) nested in a larger world."""
    def build(be):
        sc = be.new_scene()
        lam = sc.lambertian_color(0.5, 0.5, 0.5)
        single = sc.bvh([sc.sphere((0.0, 0.0, 0.0), 1.0, lam)])
        empty = sc.bvh([])
        others = [sc.sphere((3.0 * i - 6.0, 0.5 * i, -2.0), 0.8, lam) for i in range(5)]
        sc.set_world(sc.group([empty, single, sc.bvh(others), sc.bvh([])]))
        sc.set_camera(image_width=8, samples_per_pixel=1, max_depth=2, look_from=(0, 0, 20), look_at=(0, 0, 0), background=(1, 1, 1))
        sc.commit()
        return sc
    rng = np.random.default_rng(11)
    n = 20_000
    o = rng.uniform(-8.0, 8.0, (n, 3)) + np.array([0.0, 0.0, 12.0])
    d = rng.uniform(-4.0, 4.0, (n, 3)) - o
    want = build(oracle).trace_rays(o, d, seed=1)
    got = build(gpu).trace_rays(o, d, seed=1)
    assert want["hit"].sum() > 100
    assert np.array_equal(got["hit"], want["hit"])
    assert np.array_equal(got["leaf_id"], want["leaf_id"])
    m = want["hit"] == 1
    assert np.array_equal(got["t"][m], want["t"][m])


def _shared_scene(be, density=0.5, how="translates"):
    """One ConstantMedium node placed twice in the world (the reference shares Arc<dyn Node>s the same way).  Each
    copy draws its own free-flight sample whenever a ray reaches it (constant_medium.rs:64), so a ray through both
    sees exp(-density * (A + B))."""
    sc = be.new_scene()
    lam = sc.lambertian_color(0.5, 0.5, 0.5)
    medium = sc.constant_medium_color(sc.sphere((0.0, 0.0, 0.0), 1.0, lam), density, 1.0, 1.0, 1.0)
    far = sc.sphere((0.0, 50.0, 0.0), 1.0, lam)
    if how == "translates":
        # four members: the hierarchy splits them two and two, so neither copy sits alone in a sub-hierarchy (which
        # would test it twice, bvh.rs:27-28)
        members = [sc.translate(medium, (0.0, 0.0, -2.0)), sc.translate(medium, (0.0, 0.0, 2.0)), far, sc.sphere((0.0, 60.0, 0.0), 1.0, lam)]
        world = sc.bvh(members)
    elif how == "group_twice":  # the same node twice in one Group: tested twice per world.hit, like two media in one place
        world = sc.group([medium, medium, far])
    else:  # nested single-member hierarchies: 2 x 2 tests of the one medium
        world = sc.bvh([sc.bvh([sc.bvh([medium])]), far])
    sc.set_world(world)
    sc.set_camera(image_width=24, samples_per_pixel=16, max_depth=6, vertical_fov=30, look_from=(0.5, 0.3, 14), look_at=(0, 0, 0), background=(1, 1, 1))
    sc.commit()
    return sc


@pytest.mark.parametrize("how,inside", [("translates", 4.0), ("group_twice", 4.0), ("nested_single", 8.0)])
def test_oracle_shared_medium_draws_independently(oracle, how, inside):
    n = 60_000
    o, d = _rays(n)
    o[:, 2] = -8.0
    hits = _shared_scene(oracle, how=how).trace_rays(o, d, seed=3)
    expected = np.exp(-0.5 * inside)
    transmitted = 1.0 - hits["hit"].mean()
    assert abs(transmitted - expected) < 4.0 * np.sqrt(expected * (1.0 - expected) / n) + 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("how,inside", [("translates", 4.0), ("group_twice", 4.0), ("nested_single", 8.0)])
def test_gpu_shared_medium_equals_oracle(pkg, oracle, gpu, how, inside):
    n = 60_000
    o, d = _rays(n)
    o[:, 2] = -8.0
    sg, so = _shared_scene(gpu, how=how), _shared_scene(oracle, how=how)
    want, got = so.trace_rays(o, d, seed=3), sg.trace_rays(o, d, seed=3)
    assert np.array_equal(got["hit"], want["hit"])
    m = want["hit"] == 1
    assert np.array_equal(got["leaf_id"][m], want["leaf_id"][m])
    assert np.abs(got["t"][m] - want["t"][m]).max() <= 1e-12 * np.abs(want["t"][m]).max()
    expected = np.exp(-0.5 * inside)
    assert abs((1.0 - got["hit"].mean()) - expected) < 4.0 * np.sqrt(expected * (1.0 - expected) / n) + 1e-3
    e, _, se = sg.render_frame(seed=2, flags=pkg.bindings.RENDER_EXACT_SHADING)
    o_img, _, so_st = so.render_frame(seed=2, sum_dtype=np.float64)
    assert se["segments"] == so_st["segments"]
    assert (np.abs(e.astype(np.float64) - o_img) > 1e-3 * (1.0 + np.abs(o_img))).mean() < 2e-3


# ---- the keyed form against the literal one, as images of the scene whose media sit alone in their hierarchies
ORC_MEDIA_LITERAL = 1 << 16   # one draw from the stream per test of a medium (constant_medium.rs:64), reference visiting order
ORC_MEDIA_SINGLE_KEY = 2 << 16  # one keyed sample per medium however often it is tested: wrong on purpose


def _final_blocks(oracle, pkg, seed, flags, width=60, spp=144, depth=12):
    sc = pkg.scenes.build(oracle, "final", width=width, spp=spp, depth=depth)
    img, _, _ = sc.render_frame(seed=seed, flags=flags)
    img = img / float(sc.sqrt_spp() ** 2)
    h, w, _ = img.shape
    b = 6  # 6 x 6 blocks of pixels
    blocks = img[: h - h % b, : w - w % b].reshape(b, (h - h % b) // b, b, (w - w % b) // b, 3).mean(axis=(1, 3))
    return blocks, img.mean(axis=(0, 1))


def test_keyed_medium_samples_render_like_literal_stream_draws(oracle, pkg):
    """Next-Week final scene (crates/cli/src/scene/final_scene.rs): both of its media are the only member of a
    BoundingVolumeHierarchy span, so the reference tests each twice per world.hit (bvh.rs:22-28), drawing a fresh
    stream sample every time (constant_medium.rs:64).  The oracle's order-independent form (the nearest of the
    medium's keyed samples, the one the device shares) must give the same image as that literal form up to Monte
    Carlo noise; a form that draws one sample per medium must not (it was this test's first failure mode: 13 % off
    in the fog).  Parity unpinned above the math level: the reference holds no image of this scene."""
    keyed_a, mean_a = _final_blocks(oracle, pkg, 11, 0)
    keyed_b, mean_b = _final_blocks(oracle, pkg, 12, 0)
    literal, mean_l = _final_blocks(oracle, pkg, 13, ORC_MEDIA_LITERAL)
    single, mean_s = _final_blocks(oracle, pkg, 14, ORC_MEDIA_SINGLE_KEY)
    noise = np.sqrt(np.mean((keyed_a - keyed_b) ** 2))   # two independent renders of the same form
    d_lit = np.sqrt(np.mean((0.5 * (keyed_a + keyed_b) - literal) ** 2))
    d_single = np.sqrt(np.mean((0.5 * (keyed_a + keyed_b) - single) ** 2))
    assert noise > 0
    # difference of an independent render from the mean of two has sqrt(1.5 / 2) of the pair's own difference
    assert d_lit < 1.5 * noise, (d_lit, noise)
    assert np.all(np.abs(mean_l - 0.5 * (mean_a + mean_b)) < 0.02 * np.abs(mean_l) + 1e-3)
    # the negative control: the test can tell a wrong sampling form from noise
    assert d_single > 2.0 * noise, (d_single, noise)


@pytest.mark.gpu
def test_gpu_final_scene_renders_like_literal_stream_draws(oracle, gpu, pkg):
    """The device's render of the Next-Week final scene (keyed medium samples, ordered traversal) against the oracle's
    literal form (reference visiting order, one stream draw per medium test), independently seeded: equal up to
    Monte Carlo noise, block by block."""
    dev_a, mean_a = _final_blocks(gpu, pkg, 21, 0)
    dev_b, mean_b = _final_blocks(gpu, pkg, 22, 0)
    literal, mean_l = _final_blocks(oracle, pkg, 13, ORC_MEDIA_LITERAL)
    noise = np.sqrt(np.mean((dev_a - dev_b) ** 2))
    d_lit = np.sqrt(np.mean((0.5 * (dev_a + dev_b) - literal) ** 2))
    assert noise > 0 and d_lit < 1.5 * noise, (d_lit, noise)
    assert np.all(np.abs(mean_l - 0.5 * (mean_a + mean_b)) < 0.02 * np.abs(mean_l) + 1e-3)
