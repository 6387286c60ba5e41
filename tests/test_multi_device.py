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
