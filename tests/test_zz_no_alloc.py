"""chip_render_accumulate_noalloc: the render call that allocates nothing (its path pool, counters and spill slab
live in a static device array).  Checked against chip_render_accumulate_device, whose kernels it shares: the radiance
sums must be the same bits, on two calls in a row, the second with a smaller input."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(sc, shape, **kw):
    h, w = shape
    a = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda:0")
    b = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    sa = sc.render_accumulate_noalloc(a.data_ptr(), **kw)
    sb = sc.render_accumulate_device(b.data_ptr(), **kw)
    torch.cuda.synchronize()
    return a.cpu().numpy(), b.cpu().numpy(), sa, sb


def test_noalloc_render_equals_allocating_render(pkg, gpu):
    torch.cuda.set_device(0)
    sc = pkg.scenes.build(gpu, "random_spheres", width=208, spp=16, depth=12)
    w, h = sc.image_size()
    assert w * h <= gpu.fn("noalloc_max_pixels")()
    a, b, sa, sb = _both(sc, (h, w), seed=9)
    assert sa["paths"] == sb["paths"] == w * h * 16 and sa["segments"] == sb["segments"]
    assert sa["batches"] > sb["batches"]  # the static pool is smaller than the allocated one
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and a.max() > 0
    # second call, smaller input: a rectangle and half of the strata (the scratch keeps the first call's contents)
    rect = (30, 94, 20, 60)
    a2, b2, sa2, sb2 = _both(sc, (40, 64), seed=9, rect=rect, stratum_offset=1, stratum_stride=2)
    assert sa2["paths"] == sb2["paths"] == 64 * 40 * 8 and sa2["segments"] == sb2["segments"]
    assert np.array_equal(a2.view(np.uint32), b2.view(np.uint32)) and a2.max() > 0
    # a scene with lights, quads, instances and a medium through the same entry point
    sm = pkg.scenes.build(gpu, "cornell_box_smoke", width=96, spp=9, depth=10)
    w2, h2 = sm.image_size()
    a3, b3, sa3, sb3 = _both(sm, (h2, w2), seed=4)
    assert sa3["segments"] == sb3["segments"]
    assert np.array_equal(a3.view(np.uint32), b3.view(np.uint32)) and a3.max() > 0


def test_noalloc_full_pool_with_spills(pkg, gpu, monkeypatch):
    """A batch that fills the static pool on the traversal variants with 768-thread blocks (rotated quads, media): the
    grid is the pool rounded up to whole blocks, so it has more threads than the pool has slots, and every thread owns a
    spill slab in the static array.  Two LDS stack entries make most descents spill."""
    torch.cuda.set_device(0)
    monkeypatch.setenv("CHIP_LDS_STACK", "2")
    for name in ("cornell_box", "final"):
        sc = pkg.scenes.build(gpu, name, width=128, spp=4, depth=8)
        w, h = sc.image_size()
        assert (w, h) == (128, 128)
        a, b, sa, sb = _both(sc, (h, w), seed=6)
        assert sa["paths"] == sb["paths"] == 128 * 128 * 4 and sa["segments"] == sb["segments"]
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and a.max() > 0


def test_noalloc_deep_nesting_mirrors_and_the_oracle(pkg, gpu, oracle):
    """The static spill slab and the literal queue inside the static pool: 20 nested Groups (40 stack entries in the
    reference's visiting order) around a scene with axis-parallel mirror rays, rendered in the reference's order (every
    ray through the literal queue) and in the default order, against the allocating entry point bit for bit and against
    the oracle's same-seed render."""
    torch.cuda.set_device(0)

    def build(be):
        sc = be.new_scene()
        mirror = sc.metal(0.9, 0.9, 0.9, 0.0)
        red = sc.lambertian_color(0.8, 0.2, 0.2)
        node = sc.group([sc.sphere((0, 0, -3), 1.0, red), sc.sphere((0, 0, -1005), 1000.0, mirror), sc.sphere((2.5, 0, -3), 1.0, mirror),
                         sc.quad((-3.0, -1.0, -6.0), (6.0, 0.0, 0.0), (0.0, 4.0, 0.0), mirror)])
        for _ in range(20):
            node = sc.group([node])
        sc.set_world(node)
        sc.set_camera(image_width=64, aspect_ratio=1.0, samples_per_pixel=16, max_depth=8, vertical_fov=40, look_from=(0, 0, 4), look_at=(0, 0, 0),
                      background=(0.6, 0.7, 1.0))
        sc.commit()
        return sc

    sg, so = build(gpu), build(oracle)
    img_o, _, st_o = so.render_frame(seed=4, sum_dtype=np.float64)
    for flags in (pkg.bindings.RENDER_EXACT_SHADING | pkg.bindings.RENDER_REFERENCE_ORDER, pkg.bindings.RENDER_EXACT_SHADING):
        a, b, sa, sb = _both(sg, (64, 64), seed=4, flags=flags)
        assert sa["segments"] == sb["segments"] == st_o["segments"]
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and a.max() > 0
        diff = np.abs(a.reshape(img_o.shape).astype(np.float64) - img_o)
        assert (diff > 1e-3 * (1.0 + np.abs(img_o))).mean() < 2e-3


def test_noalloc_rejects_what_exceeds_its_scratch(pkg, gpu):
    torch.cuda.set_device(0)
    sc = pkg.scenes.build(gpu, "random_spheres", width=400, spp=4, depth=8)  # 400 x 225 = 90000 pixels
    w, h = sc.image_size()
    buf = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda:0")
    with pytest.raises(pkg.CausticError, match="more pixels"):
        sc.render_accumulate_noalloc(buf.data_ptr(), seed=1)
    st = sc.render_accumulate_noalloc(buf.data_ptr(), seed=1, rect=(0, 256, 0, 128))  # exactly the maximum
    assert st["paths"] == 256 * 128 * 4
    deep = pkg.scenes.build(gpu, "three_spheres", width=32, spp=1, depth=300)
    with pytest.raises(pkg.CausticError, match="max_depth"):
        deep.render_accumulate_noalloc(buf.data_ptr(), seed=1)
