"""The oracle against every executable assertion the reference holds for the hot path: the doctests of
crates/core's math value types, the sphere UV table and the RNG contracts.
Above this level the reference has no vectors, so the oracle is 'parity unpinned' there."""
import ctypes as C
import math

import numpy as np
import pytest


def _d3(*v):
    return (C.c_double * 3)(*v)


def _out(n):
    return (C.c_double * n)()


@pytest.fixture(scope="module")
def lib(oracle):
    L = oracle.lib
    L.orc_kat_ray_at.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
    L.orc_kat_aabb_hit.argtypes = [C.c_void_p] * 4 + [C.c_double, C.c_double]
    L.orc_kat_interval.argtypes = [C.c_double] * 4 + [C.c_void_p]
    L.orc_kat_interval_union.argtypes = [C.c_double] * 4 + [C.c_void_p]
    L.orc_kat_refract.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
    L.orc_kat_philox_stream.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_size_t, C.c_void_p]
    L.orc_kat_mock_random.argtypes = [C.c_size_t, C.c_void_p]
    L.orc_kat_perlin_tables.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_kat_texture_value.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
    return L


def test_ray_at(lib):
    o = _out(3)
    lib.orc_kat_ray_at(_d3(0, 0, 0), _d3(1, 0, 0), 2.0, o)  # ray.rs:19-21
    assert list(o) == [2.0, 0.0, 0.0]
    lib.orc_kat_ray_at(_d3(1, 0, 0), _d3(0, 1, 0), 3.0, o)  # ray.rs:105-111
    assert list(o) == [1.0, 3.0, 0.0]


def test_aabb(lib):
    # axis_aligned_bounding_box.rs:187-193
    assert lib.orc_kat_aabb_hit(_d3(0, 0, 0), _d3(1, 1, 1), _d3(-1, 0.5, 0.5), _d3(1, 0, 0), 0.0, math.inf) == 1
    assert lib.orc_kat_aabb_hit(_d3(0, 0, 0), _d3(1, 1, 1), _d3(-1, 1.5, 0.5), _d3(1, 0, 0), 0.0, math.inf) == 0
    # a ray starting past the box misses (t interval empty)
    assert lib.orc_kat_aabb_hit(_d3(0, 0, 0), _d3(1, 1, 1), _d3(2, 0.5, 0.5), _d3(1, 0, 0), 0.0, math.inf) == 0
    # axis_aligned_bounding_box.rs:240-244
    assert lib.orc_kat_aabb_longest_axis(_d3(0, 0, 0), _d3(5, 2, 3)) == 0
    assert lib.orc_kat_aabb_longest_axis(_d3(0, 0, 0), _d3(2, 5, 3)) == 1
    assert lib.orc_kat_aabb_longest_axis(_d3(0, 0, 0), _d3(2, 3, 5)) == 2
    assert lib.orc_kat_aabb_longest_axis(_d3(0, 0, 0), _d3(4, 4, 4)) == 2  # ties go to the later axis (strict >)
    assert lib.orc_kat_aabb_longest_axis(_d3(0, 0, 0), _d3(4, 4, 1)) == 1
    # corners in any order, flat boxes padded to 1e-4 (aabb.rs:89-107,273-285)
    b = _out(6)
    lib.orc_kat_aabb_from_points(_d3(2, 0, 5), _d3(0, 2, 5), b)
    assert list(b)[:4] == [0.0, 2.0, 0.0, 2.0]
    assert b[4] == 5.0 - 0.00005 and b[5] == 5.0 + 0.00005


def test_interval(lib):
    o = _out(5)
    f = lib.orc_kat_interval(0.0, 10.0, 5.0, 4.0, o)  # interval.rs:15-17,97-101,120-123
    assert f & 1 and f & 2 and not f & 4
    assert lib.orc_kat_interval(0.0, 10.0, 15.0, 0.0, o) & 1 == 0
    assert lib.orc_kat_interval(0.0, 10.0, 0.0, 0.0, o) & 3 == 1  # boundary: contained, not surrounded
    assert lib.orc_kat_interval(0.0, 10.0, 10.0, 0.0, o) & 3 == 1
    assert lib.orc_kat_interval(0.0, 10.0, -1.0, 0.0, o) & 3 == 0
    lib.orc_kat_interval(5.0, 15.0, 0.0, 4.0, o)  # interval.rs:143-146,160-161
    assert (o[0], o[1], o[2]) == (3.0, 17.0, 10.0)
    lib.orc_kat_interval(0.0, 10.0, 0.0, 5.0, o)  # interval.rs:206-209 (Add<f64>)
    assert (o[3], o[4]) == (5.0, 15.0)
    assert lib.orc_kat_interval(5.0, 3.0, 0.0, 0.0, o) & 4  # interval.rs:178-190
    assert lib.orc_kat_interval(5.0, 5.0, 0.0, 0.0, o) & 4
    assert not lib.orc_kat_interval(3.0, 5.0, 0.0, 0.0, o) & 4
    assert lib.orc_kat_interval(math.inf, -math.inf, 0.0, 0.0, o) & 4  # EMPTY
    u = _out(2)
    lib.orc_kat_interval_union(0.0, 5.0, 3.0, 8.0, u)  # interval.rs:71-75
    assert (u[0], u[1]) == (0.0, 8.0)


def test_vector(lib):
    o = _out(11)
    lib.orc_kat_vec(_d3(3, 4, 0), _d3(0, 0, 0), o)  # vector.rs:167-168,185-186
    assert o[0] == 5.0 and o[1] == 25.0
    lib.orc_kat_vec(_d3(1, 2, 3), _d3(4, 5, 6), o)  # vector.rs:207-209
    assert o[2] == 32.0
    lib.orc_kat_vec(_d3(1, 0, 0), _d3(0, 1, 0), o)  # vector.rs:229-232
    assert (o[3], o[4], o[5]) == (0.0, 0.0, 1.0)
    lib.orc_kat_vec(_d3(1, -1, 0), _d3(0, 1, 0), o)  # vector.rs:289-292
    assert (o[6], o[7], o[8]) == (1.0, 1.0, 0.0)
    lib.orc_kat_vec(_d3(1e-9, 1e-9, 1e-9), _d3(0, 0, 0), o)  # vector.rs:266-270
    assert o[9] == 1.0
    lib.orc_kat_vec(_d3(0.1, 0, 0), _d3(0, 0, 0), o)
    assert o[9] == 0.0
    lib.orc_kat_vec(_d3(3, 4, 0), _d3(0, 0, 0), o)  # vector.rs:249-251
    assert abs(o[10] - 1.0) < 1e-10
    # refract: Snell's law for a 45 degree ray entering glass (vector.rs:304-309)
    r = _out(3)
    s = math.sqrt(0.5)
    lib.orc_kat_refract(_d3(s, -s, 0), _d3(0, 1, 0), 1.0 / 1.5, r)
    assert abs(math.hypot(r[0], r[1]) - 1.0) < 1e-12
    assert abs(r[0] - s / 1.5) < 1e-12 and r[1] < 0


def test_color(lib):
    o = _out(9)
    lib.orc_kat_color(_d3(1.0, math.nan, 0.5), o)  # color.rs:148-152
    assert (o[0], o[1], o[2]) == (1.0, 0.0, 0.5)
    assert math.isnan(o[7])  # f64::clamp keeps NaN (color.rs:162-168)
    lib.orc_kat_color(_d3(0.25, 0.5, 1.0), o)  # color.rs:124-126
    assert o[3] == 0.5 and abs(o[4] - 0.7071067811865476) < 1e-15 and o[5] == 0.999
    lib.orc_kat_color(_d3(-1.0, 0.0, 20.0), o)  # negative -> 0 (color.rs:175-177); clamp(0,10)
    assert (o[3], o[4]) == (0.0, 0.0)
    assert (o[6], o[7], o[8]) == (0.0, 0.0, 10.0)


def test_sphere_uv_table(lib):
    # object/sphere.rs:62-69
    table = [((1, 0, 0), (0.5, 0.5)), ((-1, 0, 0), (0.0, 0.5)), ((0, 1, 0), (0.5, 1.0)), ((0, -1, 0), (0.5, 0.0)), ((0, 0, 1), (0.25, 0.5)), ((0, 0, -1), (0.75, 0.5))]
    o = _out(2)
    for p, (u, v) in table:
        lib.orc_kat_sphere_uv(_d3(*p), o)
        # (-1,0,0): atan2(-0, -1) = -pi -> u = 0 (or 1 when z is +0: both name the same meridian)
        assert abs(o[0] - u) < 1e-12 or (p == (-1, 0, 0) and abs(o[0] - 1.0) < 1e-12)
        assert abs(o[1] - v) < 1e-12


def test_matrix(lib):
    o = _out(3)
    m = (C.c_double * 9)(2, 0, 0, 0, 2, 0, 0, 0, 2)
    lib.orc_kat_matvec(m, _d3(1, 2, 3), o)  # matrix.rs:80-86
    assert list(o) == [2.0, 4.0, 6.0]
    m = (C.c_double * 9)(0, -1, 0, 1, 0, 0, 0, 0, 1)  # matrix.rs:40-44 rotation about z
    lib.orc_kat_matvec(m, _d3(1, 0, 0), o)
    assert list(o) == [0.0, 1.0, 0.0]


def test_mock_random_formula(lib):
    o = _out(8)
    lib.orc_kat_mock_random(8, o)  # random.rs:152-158
    for i in range(8):
        assert o[i] == ((i * 1103515245 + 12345) % 2147483648) / 2147483648.0


def test_philox_known_answers(lib):
    """Philox4x32-10 vectors published with Random123 (kat_vectors)."""
    cases = [
        ((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
        ((0xFFFFFFFF,) * 4, (0xFFFFFFFF, 0xFFFFFFFF), (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
        ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0), (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
    ]
    for ctr, key, want in cases:
        out = (C.c_uint32 * 4)()
        lib.orc_kat_philox((C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), out)
        assert tuple(out) == want


def test_render_stream_range_and_uniformity(lib):
    """random.rs:54-79 contracts on the render stream: rand in [0,1)."""
    n = 20000
    buf = (C.c_double * n)()
    lib.orc_kat_philox_stream(12345, 7, 3, 2, n, buf)
    a = np.frombuffer(buf, dtype=np.float64)
    assert a.min() >= 0.0 and a.max() < 1.0
    assert abs(a.mean() - 0.5) < 0.01 and abs(a.var() - 1 / 12) < 0.005
    # streams of different (pixel, stratum, bounce) are distinct
    buf2 = (C.c_double * n)()
    lib.orc_kat_philox_stream(12345, 7, 3, 3, n, buf2)
    assert not np.array_equal(a, np.frombuffer(buf2, dtype=np.float64))


def test_perlin_tables_and_ranges(lib, pkg, oracle):
    """The Python table builder equals the oracle's Perlin::new on the same uniform stream
    (utils/perlin.rs:60-76,209-230); noise in [-1,1], turbulence >= 0 (perlin.rs:17-24)."""
    rng = pkg.scenes.SceneRandom(9)
    probe = pkg.scenes.SceneRandom(9)
    vals = np.array([probe.rand() for _ in range(6000)])
    t = pkg.scenes.perlin_tables(rng)
    rv = np.zeros((256, 3))
    px = np.zeros(256, dtype=np.int32)
    py = np.zeros(256, dtype=np.int32)
    pz = np.zeros(256, dtype=np.int32)
    lib.orc_kat_perlin_tables(vals.ctypes.data, vals.size, rv.ctypes.data, px.ctypes.data, py.ctypes.data, pz.ctypes.data)
    assert np.array_equal(rv, t["rand_vec"])
    assert np.array_equal(px, t["perm_x"]) and np.array_equal(py, t["perm_y"]) and np.array_equal(pz, t["perm_z"])
    assert sorted(px.tolist()) == list(range(256)) and px[255] == 255  # index 255 is never permuted
    sc = oracle.new_scene()
    noise = sc.tex_perlin_noise(1.0, t)
    turb = sc.tex_perlin_turbulence(0.0, 7, t)
    o = _out(3)
    pts = np.random.default_rng(0).uniform(-20, 20, size=(500, 3))
    for p in pts:
        lib.orc_kat_texture_value(sc.h, noise, 0.0, 0.0, _d3(*p), o)
        n = 2.0 * o[0] - 1.0
        assert -1.0 <= n <= 1.0
        lib.orc_kat_texture_value(sc.h, turb, 0.0, 0.0, _d3(*p), o)
        assert 0.0 <= o[0] <= 1.0


def test_scene_random_array_matches_scalar(pkg):
    a = pkg.scenes.SceneRandom(3)
    b = pkg.scenes.SceneRandom(3)
    v = a.rand_array(100)
    assert [b.rand() for _ in range(100)] == v.tolist()
    assert a.rand() == b.rand()
