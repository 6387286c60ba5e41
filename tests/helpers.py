"""Shared helpers for the parity tests."""
import numpy as np


def build_pair(pkg, gpu, oracle, name, **kw):
    return pkg.scenes.build(gpu, name, **kw), pkg.scenes.build(oracle, name, **kw)


def scene_bounds(sc):
    """World bounding box from the tree dump (record 0 is the root)."""
    _, _, boxes = sc.dump_tree()
    b = boxes[0]
    lo = np.array([b[0], b[2], b[4]])
    hi = np.array([b[1], b[3], b[5]])
    return lo, hi


def fixed_ray_set(sc, n_random, seed=1234, clip=2000.0):
    """Deterministic ray set: every pixel's stratum-(0,0) camera ray is added by the caller; this
    returns n_random rays with origins in (a clipped version of) the world box and isotropic
    directions, plus axis-parallel rays (zero direction components exercise the inf / NaN slab
    arithmetic of axis_aligned_bounding_box.rs:202-205)."""
    rng = np.random.default_rng(seed)
    lo, hi = scene_bounds(sc)
    lo = np.maximum(lo, -clip)
    hi = np.minimum(hi, clip)
    ext = hi - lo
    o = lo - 0.1 * ext + rng.random((n_random, 3)) * (1.2 * ext)
    d = rng.normal(size=(n_random, 3))
    d *= rng.uniform(0.25, 4.0, size=(n_random, 1))  # directions are not unit length in the reference either
    t = rng.random(n_random)
    # axis-parallel rays
    k = max(6, n_random // 50)
    oa = lo + rng.random((k, 3)) * ext
    da = np.zeros((k, 3))
    da[np.arange(k), rng.integers(0, 3, k)] = rng.choice([-1.0, 1.0], k)
    ta = rng.random(k)
    return np.vstack([o, oa]), np.vstack([d, da]), np.concatenate([t, ta])


def compare_hits(hg, ho, rel_tol=1e-5):
    """Asserts the north-star bar: identical hit/miss, t and normal within 1e-5 relative.
    Returns a dict of how close the two actually are."""
    assert hg.shape == ho.shape
    assert np.array_equal(hg["hit"], ho["hit"]), f"hit/miss differs on {(hg['hit'] != ho['hit']).sum()} rays"
    m = ho["hit"] == 1
    assert np.array_equal(hg["leaf_id"][m], ho["leaf_id"][m]), "closest primitive differs"
    assert np.array_equal(hg["front_face"][m], ho["front_face"][m])
    assert np.array_equal(hg["material_kind"][m], ho["material_kind"][m])
    tg, to = hg["t"][m], ho["t"][m]
    rel_t = np.abs(tg - to) / np.maximum(np.abs(to), 1e-300)
    assert rel_t.max(initial=0.0) <= rel_tol, f"t differs by {rel_t.max():.3e} relative"
    dn = np.abs(hg["normal"][m] - ho["normal"][m]).max(initial=0.0)
    assert dn <= rel_tol, f"normal differs by {dn:.3e}"
    dp = np.abs(hg["pt"][m] - ho["pt"][m]) / (1.0 + np.abs(ho["pt"][m]))
    assert dp.max(initial=0.0) <= rel_tol
    duv = max(np.abs(hg["u"][m] - ho["u"][m]).max(initial=0.0), np.abs(hg["v"][m] - ho["v"][m]).max(initial=0.0))
    assert duv <= rel_tol, f"uv differs by {duv:.3e}"
    return {
        "n": int(hg.size),
        "n_hit": int(m.sum()),
        "t_bit_identical": bool(np.array_equal(tg.view(np.uint64), to.view(np.uint64))),
        "max_rel_t": float(rel_t.max(initial=0.0)),
        "max_dn": float(dn),
        "max_duv": float(duv),
    }


def gamma_image(sum_rgb, sqrt_spp):
    """camera.rs:343-344 on a radiance-sum image."""
    s = np.nan_to_num(np.asarray(sum_rgb, dtype=np.float64), nan=0.0, posinf=np.inf, neginf=-np.inf)
    v = s / float(sqrt_spp * sqrt_spp)
    g = np.sqrt(np.maximum(v, 0.0))
    return np.clip(g, 0.0, 0.999)


def psnr(a, b, peak=1.0):
    mse = float(np.mean((np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)) ** 2))
    if mse == 0:
        return float("inf")
    return 10.0 * np.log10(peak * peak / mse)
