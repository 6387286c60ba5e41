"""chip_commit_devices / chip_render_frame_multi: one frame over several GPUs from one process (the shape of the
reference's caller, crates/cli/src/main.rs:119-186: deal the frame's work to workers, gather their results).
On a one-GPU machine the device list names that GPU more than once; every entry still gets its own replica, host
thread, streams and path pools, so the whole path runs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,kw", [("random_spheres", dict(width=160, spp=16, depth=12)), ("cornell_box", dict(width=64, spp=16, depth=10))])
def test_multi_equals_sum_of_rank_renders(pkg, gpu, name, kw):
    sc = pkg.scenes.build(gpu, name, **kw)
    n_dev = pkg.gpu_device_count()
    devices = [g % n_dev for g in range(3)]
    sc.commit_devices(devices)
    full, rgb_full, s_full = sc.render_frame(seed=5, want_rgb8=True)
    multi, rgb_multi, s_multi = sc.render_frame(seed=5, want_rgb8=True, multi=True)
    assert s_multi["paths"] == s_full["paths"] and s_multi["segments"] == s_full["segments"]
    # replica g renders strata g, g+3, ...; the sums are added in replica order
    parts = [sc.render_frame(seed=5, stratum_offset=g, stratum_stride=3)[0] for g in range(3)]
    want = parts[0].copy()
    for p in parts[1:]:
        want += p
    assert np.array_equal(multi.view(np.uint32), want.view(np.uint32))
    assert np.allclose(multi, full, rtol=1e-5, atol=1e-5)
    assert (np.abs(rgb_multi.astype(int) - rgb_full.astype(int)) <= 1).all()
    # a rectangle and a strata subset through the same path
    rect = (10, 50, 4, 30)
    a, _, _ = sc.render_frame(seed=5, rect=rect, stratum_offset=1, stratum_stride=2, multi=True)
    b, _, _ = sc.render_frame(seed=5, rect=rect, stratum_offset=1, stratum_stride=2)
    assert np.allclose(a, b, rtol=1e-5, atol=1e-5)
    # one replica is the plain render
    sc.commit_devices([0])
    one, _, _ = sc.render_frame(seed=5, multi=True)
    assert np.array_equal(one.view(np.uint32), full.view(np.uint32))


def test_multi_needs_commit_devices_and_follows_recommit(pkg, gpu):
    sc = pkg.scenes.build(gpu, "three_spheres", width=48, spp=4, depth=6)
    with pytest.raises(pkg.CausticError, match="chip_commit_devices"):
        sc.render_frame(seed=1, multi=True)
    with pytest.raises(pkg.CausticError, match="no such device"):
        sc.commit_devices([0, 4096])
    sc.commit_devices([0, 0])
    a, _, _ = sc.render_frame(seed=1, multi=True)
    sc.set_camera(image_width=48, samples_per_pixel=4, max_depth=6, vertical_fov=30, look_from=(0, 1, 6), look_at=(0, 0, -1), background=(0.7, 0.8, 1.0))
    sc.commit()  # replicas describe the previous commit: gone
    with pytest.raises(pkg.CausticError, match="chip_commit_devices"):
        sc.render_frame(seed=1, multi=True)
    sc.commit_devices([0, 0])
    b, _, _ = sc.render_frame(seed=1, multi=True)
    c, _, _ = sc.render_frame(seed=1)
    assert np.allclose(b, c, rtol=1e-5, atol=1e-5) and a.max() > 0


def test_eight_replicas_full_size_frame_and_row_bands(pkg, gpu):
    """Eight replicas (the 8-GPU node's shape; on a one-GPU machine the device list names that GPU eight times) at the
    benchmark's frame size.  With 64 strata every replica takes eight of them and the sums are added in replica order:
    the bits of the same eight renders added on the host.  With 4 strata there are fewer strata than replicas, so the
    call is dealt as bands of rows instead (crates/cli/src/main.rs:83-106 deals blocks of pixels): every pixel's samples
    are then added on one replica, in stratum order, and the frame is the single-device frame bit for bit."""
    n_dev = pkg.gpu_device_count()
    devices = [g % n_dev for g in range(8)]
    sc = pkg.scenes.build(gpu, "random_spheres", width=1920, spp=64, depth=50)
    assert sc.image_size() == (1920, 1080)
    sc.commit_devices(devices)
    multi, _, sm = sc.render_frame(seed=11, multi=True)
    assert sm["paths"] == 1920 * 1080 * 64
    want = None
    segs = 0
    for g in range(8):
        part, _, st = sc.render_frame(seed=11, stratum_offset=g, stratum_stride=8)
        segs += st["segments"]
        want = part.copy() if want is None else want + part
    assert sm["segments"] == segs
    assert np.array_equal(multi.view(np.uint32), want.view(np.uint32))
    # fewer strata than replicas: a call that covers strata 3, 19, 35, 51 of the 64
    few, _, sf = sc.render_frame(seed=11, stratum_offset=3, stratum_stride=16, multi=True)
    one, _, so = sc.render_frame(seed=11, stratum_offset=3, stratum_stride=16)
    assert sf["paths"] == so["paths"] == 1920 * 1080 * 4 and sf["segments"] == so["segments"]
    assert sf["batches"] >= 8  # one band per replica
    assert np.array_equal(few.view(np.uint32), one.view(np.uint32))
    # ... also on a rectangle whose height is not a multiple of anything convenient
    rect = (100, 740, 37, 300)
    a, _, _ = sc.render_frame(seed=11, rect=rect, stratum_offset=3, stratum_stride=16, multi=True)
    assert np.array_equal(a.view(np.uint32), one[rect[2]:rect[3], rect[0]:rect[1]].view(np.uint32))
