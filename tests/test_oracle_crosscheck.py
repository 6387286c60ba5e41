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


def test_disc_cone_scale_equal_numpy(pkg, oracle):
    """Disc, cone-frustum wall and Scale of the oracle against formulas written directly in numpy: an ellipsoid
    (Scale of a unit sphere, then Translate), a frustum (two discs + wall) and a free disc in one Group."""
    sx, sy, sz = 1.5, 0.75, 2.0
    ell_c = np.array([0.5, 1.0, -1.0])
    base, h, r_top, r_bot = np.array([-3.0, 0.0, 0.5]), 2.0, 0.5, 1.25
    dc, dr, dn = np.array([3.0, 1.0, 0.0]), 1.0, np.array([0.0, 0.0, 1.0])
    sc = oracle.new_scene()
    m = sc.lambertian_color(0.5, 0.5, 0.5)
    nodes = [
        sc.translate(sc.scale(sc.sphere((0, 0, 0), 1.0, m), sx, sy, sz), tuple(ell_c)),
        sc.cone_frustum(tuple(base), h, r_top, r_bot, m),
        sc.disc(tuple(dc), dr, tuple(dn), m),
    ]
    sc.set_world(sc.group(nodes))
    sc.set_camera(image_width=8, samples_per_pixel=1, max_depth=1)
    sc.commit()
    rng = np.random.default_rng(3)
    n = 20000
    o = rng.uniform(-6, 6, (n, 3))
    d = rng.uniform(-4, 4, (n, 3)) - o * 0.5
    hits = sc.trace_rays(o, d, None, seed=1)
    tmin = 0.001
    INF = np.inf

    def first_in_range(t0, t1):
        return np.where(t0 > tmin, t0, np.where(t1 > tmin, t1, INF))

    # ellipsoid: unit sphere in the space x / s
    s = np.array([sx, sy, sz])
    oo, dd = (o - ell_c) / s, d / s
    a = (dd * dd).sum(-1); hb = -(oo * dd).sum(-1); c = (oo * oo).sum(-1) - 1.0
    disc = hb * hb - a * c
    sq = np.sqrt(np.maximum(disc, 0))
    t_ell = np.where(disc >= 0, first_in_range((hb - sq) / a, (hb + sq) / a), INF)
    # discs (plane hit inside the radius, t >= tmin)
    def disc_t(center, radius, normal):
        den = d @ normal
        t = ((center - o) @ normal) / np.where(np.abs(den) < 1e-8, np.nan, den)
        p = o + t[:, None] * d
        ok = (t >= tmin) & (((p - center) ** 2).sum(-1) <= radius * radius)
        return np.where(ok, t, INF)
    t_disc = disc_t(dc, dr, dn)
    t_top = disc_t(base + np.array([0, h, 0]), r_top, np.array([0.0, 1.0, 0.0]))
    t_bot = disc_t(base, r_bot, np.array([0.0, -1.0, 0.0]))
    # wall: radius r(y) = r_bot + k y; candidates in the reference's order, cone.rs:204-243
    lo = o - base
    k = (r_top - r_bot) / h
    A = d[:, 0] ** 2 + d[:, 2] ** 2 - k * k * d[:, 1] ** 2
    Bq = 2.0 * (lo[:, 0] * d[:, 0] + lo[:, 2] * d[:, 2] - k * k * lo[:, 1] * d[:, 1] - k * r_bot * d[:, 1])
    Cq = lo[:, 0] ** 2 + lo[:, 2] ** 2 - (r_bot + k * lo[:, 1]) ** 2
    D = Bq * Bq - 4 * A * Cq
    sqD = np.sqrt(np.maximum(D, 0))
    ta, tb = (-Bq - sqD) / (2 * A), (-Bq + sqD) / (2 * A)
    def y_ok(t):
        y = lo[:, 1] + t * d[:, 1]
        return (y >= 0) & (y <= h)
    first = np.where(ta >= tmin, ta, np.where(tb >= tmin, tb, INF))
    t_wall = np.where(np.isfinite(first) & y_ok(first), first, np.where((first != tb) & (tb >= tmin) & y_ok(tb), tb, INF))
    t_wall = np.where((D >= 0) & (np.abs(A) >= 1e-8), t_wall, INF)
    t_all = np.stack([t_ell, t_top, t_bot, t_wall, t_disc], axis=1)
    best = t_all.argmin(axis=1)
    t = t_all[np.arange(n), best]
    hit_np = np.isfinite(t)
    # rays within rounding of a rim or a tangent are ambiguous between the two computations
    sorted_t = np.sort(t_all, axis=1)
    clear = (np.abs(D) > 1e-9) & (np.abs(disc) > 1e-9)
    assert (clear.mean() > 0.99) and hit_np.mean() > 0.1
    assert np.array_equal((hits["hit"] == 1)[clear], hit_np[clear])
    mk = clear & hit_np
    assert np.abs(hits["t"][mk] - t[mk]).max() <= 1e-9 * (1 + np.abs(t[mk]).max())
    # surface invariants of the reported points and normals
    p = hits["pt"]; nrm = hits["normal"]
    e = mk & (best == 0)
    q = (p[e] - ell_c) / s
    assert np.abs((q * q).sum(-1) - 1.0).max() < 1e-9
    g = q / s
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    assert np.abs(np.abs((g * nrm[e]).sum(-1)) - 1.0).max() < 1e-9      # normal along the ellipsoid's gradient
    w = mk & (best == 3)
    pl = p[w] - base
    assert np.abs(np.hypot(pl[:, 0], pl[:, 2]) - (r_bot + k * pl[:, 1])).max() < 1e-9
    assert np.abs(np.linalg.norm(nrm[mk], axis=1) - 1.0).max() < 1e-12
    assert ((nrm[mk] * d[mk]).sum(-1) <= 0).all()                        # set_face_normal: against the ray
