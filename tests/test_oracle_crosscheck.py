"""Second, independently written check of the oracle on small cases: a numpy brute-force closest hit
(every ray against every sphere / quad, no BVH, no recursion) must name the same primitive and the same
t as the oracle's recursive BVH traversal."""
import numpy as np

from helpers import fixed_ray_set


def brute_force_spheres(o, d, tm, centers, motion, radii, tmin=0.001):
    c = centers[None, :, :] + tm[:, None, None] * motion[None, :, :]
    oc = c - o[:, None, :]
    a = (d * d).sum(-1)[:, None]
    h = (d[:, None, :] * oc).sum(-1)
    cc = (oc * oc).sum(-1) - radii[None, :] ** 2
    disc = h * h - a * cc
    ok = disc >= 0
    sq = np.sqrt(np.where(ok, disc, 0.0))
    t0 = (h - sq) / a
    t1 = (h + sq) / a
    t = np.where(t0 > tmin, t0, np.where(t1 > tmin, t1, np.inf))
    t = np.where(ok, t, np.inf)
    best = t.argmin(axis=1)
    return best, t[np.arange(len(o)), best]


def test_random_spheres_bvh_equals_brute_force(pkg, oracle):
    # rebuild the scene's sphere list by replaying the builder with a recording shim
    class Rec:
        def __init__(self):
            self.spheres = []

        def __getattr__(self, name):
            def f(*a, **k):
                return 0
            return f

        def sphere(self, center, radius, mat, direction=None):
            self.spheres.append((center, radius, direction or (0.0, 0.0, 0.0)))
            return len(self.spheres) - 1

    rec = Rec()
    pkg.scenes.random_spheres(rec, width=64, spp=1, depth=1)
    centers = np.array([s[0] for s in rec.spheres], dtype=np.float64)
    radii = np.array([s[1] for s in rec.spheres], dtype=np.float64)
    motion = np.array([s[2] for s in rec.spheres], dtype=np.float64)
    sc = pkg.scenes.build(oracle, "random_spheres", width=64, spp=1, depth=1)
    o, d, tm = fixed_ray_set(sc, 4000, seed=5, clip=30.0)
    hits = sc.trace_rays(o, d, tm, seed=1)
    best, t = brute_force_spheres(o, d, tm, centers, motion, radii)
    hit_bf = np.isfinite(t)
    assert np.array_equal(hits["hit"] == 1, hit_bf)
    m = hit_bf
    assert np.array_equal(hits["leaf_id"][m], best[m])  # node ids are creation order = list order
    assert np.abs(hits["t"][m] - t[m]).max() <= 1e-9 * np.abs(t[m]).max()


def test_quads_scene_equals_brute_force(pkg, oracle):
    sc = pkg.scenes.build(oracle, "quads", width=64, spp=1, depth=1)
    quads = [
        ((-3.0, -2.0, 5.0), (0.0, 0.0, -4.0), (0.0, 4.0, 0.0)),
        ((-2.0, -2.0, 0.0), (4.0, 0.0, 0.0), (0.0, 4.0, 0.0)),
        ((3.0, -2.0, 1.0), (0.0, 0.0, 4.0), (0.0, 4.0, 0.0)),
        ((-2.0, 3.0, 1.0), (4.0, 0.0, 0.0), (0.0, 0.0, 4.0)),
        ((-2.0, -3.0, 5.0), (4.0, 0.0, 0.0), (0.0, 0.0, -4.0)),
    ]
    o, d, tm = fixed_ray_set(sc, 3000, seed=6)
    hits = sc.trace_rays(o, d, tm, seed=1)
    best_t = np.full(len(o), np.inf)
    best = np.full(len(o), -1)
    for i, (q, u, v) in enumerate(quads):
        q, u, v = map(np.array, (q, u, v))
        n = np.cross(u, v)
        w = n / n.dot(n)
        nu = n / np.linalg.norm(n)
        denom = d @ nu
        with np.errstate(divide="ignore", invalid="ignore"):
            t = (nu.dot(q) - o @ nu) / denom
        p = o + t[:, None] * d - q
        al = np.cross(p, v) @ w
        be = np.cross(u, p) @ w
        ok = (np.abs(denom) >= 1e-8) & (t >= 0.001) & (al >= 0) & (al <= 1) & (be >= 0) & (be <= 1) & (t <= best_t)
        best_t = np.where(ok, t, best_t)
        best = np.where(ok, i, best)
    assert np.array_equal(hits["hit"] == 1, best >= 0)
    m = best >= 0
    assert np.array_equal(hits["leaf_id"][m], best[m])
    assert np.abs(hits["t"][m] - best_t[m]).max() <= 1e-9 * np.abs(best_t[m]).max()
    assert np.abs(hits["u"][m]).max() <= 1.0 and hits["u"][m].min() >= 0.0
