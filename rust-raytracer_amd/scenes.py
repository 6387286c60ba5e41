"""Scene definitions restated from the reference's crates/cli/src/scene/*.rs.

Each builder drives a `bindings.Scene` (GPU library or, in the tests, the CPU
oracle) through the same constructor calls the Rust code makes, in the same
order, so node / material / texture ids are identical across backends.

The reference draws scene randomness (sphere placement, box heights, Perlin
tables) from an unseeded `rand::random()` (crates/core/src/random.rs:28-40), so
its scenes differ on every run.  Here the draws come from `SceneRandom`, a
seeded splitmix64 stream consumed in the reference's draw order; "the scene"
is therefore the instance produced by a given seed (default 1).

Resolution, samples per pixel and depth are parameters (None = the value
hard-coded in the reference's scene file).
"""
from __future__ import annotations

import math
import os

import numpy as np

_MASK = (1 << 64) - 1


class SceneRandom:
    """Seeded uniform stream with the helper forms of random.rs:169-176."""

    def __init__(self, seed: int = 1):
        self.state = (seed * 0x9E3779B97F4A7C15 + 0x632BE59BD9B4E019) & _MASK

    def next_u64(self) -> int:
        self.state = (self.state + 0x9E3779B97F4A7C15) & _MASK
        z = self.state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK
        return z ^ (z >> 31)

    def rand(self) -> float:
        return (self.next_u64() >> 11) * (1.0 / 9007199254740992.0)

    def rand_interval(self, lo: float, hi: float) -> float:
        return lo + (hi - lo) * self.rand()

    def rand_int_interval(self, lo: int, hi: int) -> int:
        return lo + int(self.rand() * (hi - lo))

    def rand_array(self, n: int) -> np.ndarray:
        """n uniforms, identical to n successive rand() calls (vectorised)."""
        idx = np.arange(1, n + 1, dtype=np.uint64)
        with np.errstate(over="ignore"):
            z = np.uint64(self.state) + idx * np.uint64(0x9E3779B97F4A7C15)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
        self.state = (self.state + n * 0x9E3779B97F4A7C15) & _MASK
        return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def random_unit(rng: SceneRandom):
    """Vector3::random_unit, crates/core/src/vector.rs:79-87."""
    while True:
        x = rng.rand_interval(-1.0, 1.0)
        y = rng.rand_interval(-1.0, 1.0)
        z = rng.rand_interval(-1.0, 1.0)
        l2 = x * x + y * y + z * z
        if 1e-160 < l2 <= 1.0:
            s = math.sqrt(l2)
            return (x / s, y / s, z / s)


def perlin_tables(rng: SceneRandom):
    """Perlin::new, crates/core/src/utils/perlin.rs:60-76,209-230: 256 random unit vectors, then
    three permutations shuffled for i = 254..1 with target in [0, i) (index 255 is never touched)."""
    rand_vec = np.array([random_unit(rng) for _ in range(256)], dtype=np.float64)

    def perm():
        p = list(range(256))
        for i in range(254, 0, -1):
            t = rng.rand_int_interval(0, i)
            p[i], p[t] = p[t], p[i]
        return np.array(p, dtype=np.int32)

    return {"rand_vec": rand_vec, "perm_x": perm(), "perm_y": perm(), "perm_z": perm()}


_EARTH = None


def earth_texels() -> np.ndarray:
    """Decoded texels (512 x 1024 x 3 u8) of the reference's crates/cli/assets/earth-map.jpg,
    stored as a fixture so no JPEG decoder is involved at run time."""
    global _EARTH
    if _EARTH is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "earth_map_texels.npz")
        _EARTH = np.load(path)["rgb"]
    return _EARTH


def _camera(sc, width, spp, depth, defaults, **kw):
    sc.set_camera(
        image_width=defaults[0] if width is None else width,
        samples_per_pixel=defaults[1] if spp is None else spp,
        max_depth=defaults[2] if depth is None else depth,
        **kw,
    )


def random_spheres(sc, width=None, spp=None, depth=None, seed=1):
    """RTIOW final scene with moving diffuse spheres: crates/cli/src/scene/random_spheres.rs:11-100."""
    rng = SceneRandom(seed)
    world = []
    ground = sc.lambertian_color(0.5, 0.5, 0.5)
    world.append(sc.sphere((0.0, -1000.0, 0.0), 1000.0, ground))
    for a in range(-11, 11):
        for b in range(-11, 11):
            choose_mat = rng.rand()
            cx = a + 0.9 * rng.rand()
            cz = b + 0.9 * rng.rand()
            center = (cx, 0.2, cz)
            dx, dy, dz = cx - 4.0, 0.2 - 0.2, cz - 0.0
            if math.sqrt(dx * dx + dy * dy + dz * dz) > 0.9:
                if choose_mat < 0.8:
                    c1 = (rng.rand(), rng.rand(), rng.rand())
                    c2 = (rng.rand(), rng.rand(), rng.rand())
                    m = sc.lambertian_color(c1[0] * c2[0], c1[1] * c2[1], c1[2] * c2[2])
                    world.append(sc.sphere(center, 0.2, m, direction=(0.0, rng.rand_interval(0.0, 0.5), 0.0)))
                elif choose_mat < 0.95:
                    albedo = (rng.rand_interval(0.5, 1.0), rng.rand_interval(0.5, 1.0), rng.rand_interval(0.5, 1.0))
                    fuzz = rng.rand_interval(0.0, 0.5)
                    world.append(sc.sphere(center, 0.2, sc.metal(*albedo, fuzz)))
                else:
                    world.append(sc.sphere(center, 0.2, sc.dielectric(1.5)))
    world.append(sc.sphere((0.0, 1.0, 0.0), 1.0, sc.dielectric(1.5)))
    world.append(sc.sphere((-4.0, 1.0, 0.0), 1.0, sc.lambertian_color(0.4, 0.2, 0.1)))
    world.append(sc.sphere((4.0, 1.0, 0.0), 1.0, sc.metal(0.7, 0.6, 0.5, 0.0)))
    sc.set_world(sc.bvh(world))
    sc.set_lights(None)
    _camera(
        sc, width, spp, depth, (300, 10, 50),
        aspect_ratio=16.0 / 9.0, vertical_fov=20.0, look_from=(13.0, 2.0, 3.0), look_at=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0),
        defocus_angle=0.6, focus_distance=10.0, background=(0.7, 0.8, 1.0),
    )
    return sc


def final_scene(sc, width=None, spp=None, depth=None, seed=1):
    """The Next Week final scene: crates/cli/src/scene/final_scene.rs:16-172."""
    rng = SceneRandom(seed)
    world = []
    ground = sc.lambertian_color(0.48, 0.83, 0.53)
    for i in range(20):
        for j in range(20):
            w = 100.0
            x0 = -1000.0 + i * w
            z0 = -1000.0 + j * w
            y0 = 0.0
            x1 = x0 + w
            y1 = rng.rand_interval(1.0, 101.0)
            z1 = z0 + w
            world.append(sc.box((x0, y0, z0), (x1, y1, z1), ground))
    light = sc.diffuse_light_color(1.0, 1.0, 1.0)
    world.append(sc.quad((123.0, 554.0, 147.0), (300.0, 0.0, 0.0), (0.0, 0.0, 265.0), light))
    # moving sphere: center2 - center1 = (30, 0, 0)
    center1 = (400.0, 400.0, 200.0)
    center2 = (center1[0] + 30.0, center1[1] + 0.0, center1[2] + 0.0)
    world.append(
        sc.sphere(center1, 50.0, sc.lambertian_color(0.7, 0.3, 0.1), direction=(center2[0] - center1[0], center2[1] - center1[1], center2[2] - center1[2]))
    )
    world.append(sc.sphere((260.0, 150.0, 45.0), 50.0, sc.dielectric(1.5)))
    world.append(sc.sphere((0.0, 150.0, 145.0), 50.0, sc.metal(0.8, 0.8, 0.9, 1.0)))
    boundary = sc.sphere((360.0, 150.0, 145.0), 70.0, sc.dielectric(1.5))
    world.append(boundary)
    world.append(sc.constant_medium_color(boundary, 0.2, 0.2, 0.4, 0.9))
    boundary = sc.sphere((0.0, 0.0, 0.0), 5000.0, sc.dielectric(1.5))
    world.append(sc.constant_medium_color(boundary, 0.0001, 1.0, 1.0, 1.0))
    earth = sc.lambertian_tex(sc.tex_image(earth_texels()))
    world.append(sc.sphere((400.0, 200.0, 400.0), 100.0, earth))
    pertext = sc.tex_perlin_noise(0.2, perlin_tables(rng))
    world.append(sc.sphere((220.0, 280.0, 300.0), 80.0, sc.lambertian_tex(pertext)))
    white = sc.lambertian_color(0.73, 0.73, 0.73)
    boxes2 = []
    for _ in range(1000):
        c = (rng.rand_interval(0.0, 165.0), rng.rand_interval(0.0, 165.0), rng.rand_interval(0.0, 165.0))
        boxes2.append(sc.sphere(c, 10.0, white))
    world.append(sc.translate(sc.rotate_y(sc.bvh(boxes2), 15.0), (-100.0, 270.0, 395.0)))
    sc.set_world(sc.bvh(world))
    light1 = sc.quad((123.0, 554.0, 147.0), (300.0, 0.0, 0.0), (0.0, 0.0, 265.0), sc.empty_material())
    sc.set_lights(sc.group([light1]))
    _camera(
        sc, width, spp, depth, (400, 400, 10),
        aspect_ratio=1.0, vertical_fov=40.0, look_from=(478.0, 278.0, -600.0), look_at=(278.0, 278.0, 0.0), up=(0.0, 1.0, 0.0),
        defocus_angle=0.0,
    )
    return sc


def _cornell_walls(sc, light_q, light_u, light_v, emit):
    red = sc.lambertian_color(0.65, 0.05, 0.05)
    white = sc.lambertian_color(0.73, 0.73, 0.73)
    green = sc.lambertian_color(0.12, 0.45, 0.15)
    light = sc.diffuse_light_color(emit, emit, emit)
    world = [
        sc.quad((555.0, 0.0, 0.0), (0.0, 555.0, 0.0), (0.0, 0.0, 555.0), green),
        sc.quad((0.0, 0.0, 0.0), (0.0, 555.0, 0.0), (0.0, 0.0, 555.0), red),
        sc.quad(light_q, light_u, light_v, light),
        sc.quad((0.0, 0.0, 0.0), (555.0, 0.0, 0.0), (0.0, 0.0, 555.0), white),
        sc.quad((555.0, 555.0, 555.0), (-555.0, 0.0, 0.0), (0.0, 0.0, -555.0), white),
        sc.quad((0.0, 0.0, 555.0), (555.0, 0.0, 0.0), (0.0, 555.0, 0.0), white),
    ]
    return world, white


def _cornell_camera(sc, width, spp, depth, defaults):
    _camera(
        sc, width, spp, depth, defaults,
        aspect_ratio=1.0, vertical_fov=40.0, look_from=(278.0, 278.0, -800.0), look_at=(278.0, 278.0, 0.0), up=(0.0, 1.0, 0.0),
        background=(0.0, 0.0, 0.0), defocus_angle=0.0,
    )


def cornell_box(sc, width=None, spp=None, depth=None, seed=1):
    """Cornell box with a glass sphere and quad + sphere light sampling: crates/cli/src/scene/cornell_box.rs:11-119."""
    world, white = _cornell_walls(sc, (343.0, 554.0, 332.0), (-130.0, 0.0, 0.0), (0.0, 0.0, -105.0), 15.0)
    box1 = sc.box((0.0, 0.0, 0.0), (165.0, 330.0, 165.0), white)
    box1 = sc.rotate_y(box1, 15.0)
    box1 = sc.translate(box1, (265.0, 0.0, 295.0))
    world.append(box1)
    world.append(sc.sphere((190.0, 90.0, 190.0), 90.0, sc.dielectric(1.5)))
    sc.set_world(sc.bvh(world))
    light1 = sc.quad((343.0, 554.0, 332.0), (-130.0, 0.0, 0.0), (0.0, 0.0, -105.0), sc.empty_material())
    light2 = sc.sphere((190.0, 90.0, 190.0), 90.0, sc.empty_material())
    sc.set_lights(sc.group([light1, light2]))
    _cornell_camera(sc, width, spp, depth, (600, 100, 50))
    return sc


def cornell_box_smoke(sc, width=None, spp=None, depth=None, seed=1):
    """Cornell box with two smoke boxes: crates/cli/src/scene/cornell_box_smoke.rs:13-120."""
    world, white = _cornell_walls(sc, (113.0, 554.0, 127.0), (330.0, 0.0, 0.0), (0.0, 0.0, 305.0), 5.0)
    box1 = sc.box((0.0, 0.0, 0.0), (165.0, 330.0, 165.0), white)
    box1 = sc.rotate_y(box1, 15.0)
    box1 = sc.translate(box1, (265.0, 0.0, 295.0))
    world.append(sc.constant_medium_color(box1, 0.01, 0.0, 0.0, 0.0))
    box2 = sc.box((0.0, 0.0, 0.0), (165.0, 165.0, 165.0), white)
    box2 = sc.rotate_y(box2, -18.0)
    box2 = sc.translate(box2, (130.0, 0.0, 65.0))
    world.append(sc.constant_medium_color(box2, 0.01, 0.7, 0.7, 0.7))
    sc.set_world(sc.bvh(world))
    light1 = sc.quad((343.0, 554.0, 332.0), (-130.0, 0.0, 0.0), (0.0, 0.0, -105.0), sc.empty_material())
    sc.set_lights(sc.group([light1]))
    _cornell_camera(sc, width, spp, depth, (600, 200, 50))
    return sc


def three_spheres(sc, width=None, spp=None, depth=None, seed=1):
    """crates/cli/src/scene/three_spheres.rs:12-70 (camera left at CameraBuilder defaults except as set there)."""
    ground = sc.lambertian_tex(sc.tex_checker(0.32, sc.tex_solid(0.2, 0.3, 0.1), sc.tex_solid(0.9, 0.9, 0.9)))
    center = sc.lambertian_color(0.1, 0.2, 0.5)
    left = sc.dielectric(1.5)
    bubble = sc.dielectric(1.0 / 1.5)
    right = sc.metal(0.8, 0.6, 0.2, 0.2)
    world = [
        sc.sphere((0.0, -100.5, -1.0), 100.0, ground),
        sc.sphere((0.0, 0.0, -1.2), 0.5, center),
        sc.sphere((-1.0, 0.0, -1.0), 0.5, left),
        sc.sphere((-1.0, 0.0, -1.0), 0.4, bubble),
        sc.sphere((1.0, 0.0, -1.0), 0.5, right),
    ]
    sc.set_world(sc.bvh(world))
    sc.set_lights(None)
    _camera(sc, width, spp, depth, (600, 10, 50), aspect_ratio=16.0 / 9.0, defocus_angle=0.6, focus_distance=1.0, background=(0.7, 0.8, 1.0))
    return sc


def checkered_spheres(sc, width=None, spp=None, depth=None, seed=1):
    """crates/cli/src/scene/checkered_spheres.rs:12-54."""
    checker = sc.lambertian_tex(sc.tex_checker(0.32, sc.tex_solid(0.2, 0.3, 0.1), sc.tex_solid(0.9, 0.9, 0.9)))
    world = [sc.sphere((0.0, -10.0, 0.0), 10.0, checker), sc.sphere((0.0, 10.0, 0.0), 10.0, checker)]
    sc.set_world(sc.bvh(world))
    sc.set_lights(None)
    _camera(
        sc, width, spp, depth, (300, 10, 50),
        aspect_ratio=16.0 / 9.0, vertical_fov=20.0, look_from=(13.0, 2.0, 3.0), look_at=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0),
        defocus_angle=0.0, background=(0.7, 0.8, 1.0),
    )
    return sc


def earth(sc, width=None, spp=None, depth=None, seed=1):
    """crates/cli/src/scene/earth.rs:10-35: the world is a bare Sphere, no BVH."""
    surface = sc.lambertian_tex(sc.tex_image(earth_texels()))
    globe = sc.sphere((0.0, 0.0, 0.0), 2.0, surface)
    sc.set_world(globe)
    sc.set_lights(None)
    _camera(
        sc, width, spp, depth, (300, 10, 50),
        aspect_ratio=16.0 / 9.0, vertical_fov=20.0, look_from=(0.0, 0.0, 12.0), look_at=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0),
        defocus_angle=0.0, background=(0.7, 0.8, 1.0),
    )
    return sc


def perlin_spheres(sc, width=None, spp=None, depth=None, seed=1):
    """crates/cli/src/scene/perlin_spheres.rs:12-59."""
    rng = SceneRandom(seed)
    noise = sc.lambertian_tex(sc.tex_perlin_noise(4.0, perlin_tables(rng)))
    turb = sc.lambertian_tex(sc.tex_perlin_turbulence(4.0, 7, perlin_tables(rng)))
    world = [
        sc.sphere((0.0, -1000.0, 0.0), 1000.0, noise),
        sc.sphere((0.0, 2.0, -2.0), 2.0, noise),
        sc.sphere((0.0, 2.0, 2.0), 2.0, turb),
    ]
    sc.set_world(sc.bvh(world))
    sc.set_lights(None)
    _camera(
        sc, width, spp, depth, (400, 10, 50),
        aspect_ratio=16.0 / 9.0, vertical_fov=20.0, look_from=(15.0, 2.0, 3.0), look_at=(0.0, 1.5, 0.0), up=(0.0, 1.0, 0.0),
        defocus_angle=0.0, background=(0.7, 0.8, 1.0),
    )
    return sc


def quads(sc, width=None, spp=None, depth=None, seed=1):
    """crates/cli/src/scene/quads.rs:11-74."""
    world = [
        sc.quad((-3.0, -2.0, 5.0), (0.0, 0.0, -4.0), (0.0, 4.0, 0.0), sc.lambertian_color(1.0, 0.2, 0.2)),
        sc.quad((-2.0, -2.0, 0.0), (4.0, 0.0, 0.0), (0.0, 4.0, 0.0), sc.lambertian_color(0.2, 1.0, 0.2)),
        sc.quad((3.0, -2.0, 1.0), (0.0, 0.0, 4.0), (0.0, 4.0, 0.0), sc.lambertian_color(0.2, 0.2, 1.0)),
        sc.quad((-2.0, 3.0, 1.0), (4.0, 0.0, 0.0), (0.0, 0.0, 4.0), sc.lambertian_color(1.0, 0.5, 0.0)),
        sc.quad((-2.0, -3.0, 5.0), (4.0, 0.0, 0.0), (0.0, 0.0, -4.0), sc.lambertian_color(0.2, 0.8, 0.8)),
    ]
    sc.set_world(sc.bvh(world))
    sc.set_lights(None)
    _camera(
        sc, width, spp, depth, (400, 10, 50),
        aspect_ratio=1.0, vertical_fov=80.0, look_from=(0.0, 0.0, 9.0), look_at=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0),
        defocus_angle=0.0, background=(0.7, 0.8, 1.0),
    )
    return sc


def lighted_sphere(sc, width=None, spp=None, depth=None, seed=1):
    """crates/cli/src/scene/lighted_sphere.rs:12-70."""
    rng = SceneRandom(seed)
    per = sc.lambertian_tex(sc.tex_perlin_turbulence(4.0, 7, perlin_tables(rng)))
    white = sc.diffuse_light_color(4.0, 4.0, 4.0)
    blue = sc.diffuse_light_color(0.0, 0.0, 2.0)
    world = [
        sc.sphere((0.0, -1000.0, 0.0), 1000.0, per),
        sc.sphere((0.0, 2.0, 0.0), 2.0, per),
        sc.quad((3.0, 1.0, -2.0), (2.0, 0.0, 0.0), (0.0, 2.0, 0.0), white),
        sc.sphere((0.0, 7.0, 0.0), 2.0, blue),
    ]
    sc.set_world(sc.bvh(world))
    sc.set_lights(None)
    _camera(
        sc, width, spp, depth, (400, 50, 50),
        aspect_ratio=16.0 / 9.0, vertical_fov=20.0, look_from=(26.0, 3.0, 6.0), look_at=(0.0, 2.0, 0.0), up=(0.0, 1.0, 0.0),
        defocus_angle=0.0, background=(0.0, 0.0, 0.0),
    )
    return sc


def lighted_cone_frustum(sc, width=None, spp=None, depth=None, seed=1):
    """crates/cli/src/scene/lighted_cone_frustum.rs:12-80."""
    rng = SceneRandom(seed)
    per = sc.lambertian_tex(sc.tex_perlin_turbulence(4.0, 7, perlin_tables(rng)))
    white = sc.diffuse_light_color(4.0, 4.0, 4.0)
    blue = sc.diffuse_light_color(0.0, 0.0, 2.0)
    world = [
        sc.sphere((0.0, -1000.0, 0.0), 1000.0, per),
        sc.cone_frustum((0.0, 1.5, 0.0), 2.0, 1.0, 2.0, per),
        sc.quad((3.0, 1.0, -2.0), (2.0, 0.0, 0.0), (0.0, 2.0, 0.0), white),
        sc.sphere((0.0, 7.0, 0.0), 2.0, blue),
    ]
    sc.set_world(sc.bvh(world))
    sc.set_lights(None)
    _camera(
        sc, width, spp, depth, (400, 50, 50),
        aspect_ratio=16.0 / 9.0, vertical_fov=20.0, look_from=(26.0, 6.0, 6.0), look_at=(0.0, 2.0, 0.0), up=(0.0, 1.0, 0.0),
        defocus_angle=0.0, background=(0.0, 0.0, 0.0),
    )
    return sc


def scad_shapes(sc, width=None, spp=None, depth=None, seed=1, light="disc"):
    """The node vocabulary crates/openscad's interpreter adds to the book scenes (modules.rs:76-105 circle -> Disc,
    165-232 cylinder -> ConeFrustum, 328-346 scale -> Scale), in the nesting it produces: scale / rotate / translate
    chains around cylinders, cones, discs, spheres and boxes, lit by a disc (or by a cone frustum, whose side wall
    has the reference's quirky pdf_value / random, cone.rs:296-347)."""
    grey = sc.lambertian_color(0.73, 0.73, 0.73)
    red = sc.lambertian_color(0.65, 0.05, 0.05)
    green = sc.lambertian_color(0.12, 0.45, 0.15)
    steel = sc.metal(0.8, 0.85, 0.88, 0.1)
    glass = sc.dielectric(1.5)
    lamp = sc.diffuse_light_color(12.0, 12.0, 12.0)
    world = [
        sc.quad((-8.0, 0.0, -8.0), (16.0, 0.0, 0.0), (0.0, 0.0, 16.0), grey),             # floor
        sc.quad((-8.0, 0.0, -8.0), (16.0, 0.0, 0.0), (0.0, 10.0, 0.0), green),            # back wall
        sc.cone_frustum((-3.0, 0.0, 0.0), 3.0, 1.0, 1.0, red),                              # cylinder
        sc.cone_frustum((3.0, 0.0, 1.0), 2.5, 0.0, 1.5, steel),                             # cone (no top disc)
        sc.translate(sc.scale(sc.sphere((0.0, 0.0, 0.0), 1.0, glass), 1.5, 0.75, 1.0), (0.0, 0.75, 2.5)),   # ellipsoid
        sc.translate(sc.rotate(sc.scale(sc.box((-0.5, -0.5, -0.5), (0.5, 0.5, 0.5), grey), 2.0, 1.0, 0.5), (0.0, 1.0, 0.0), 30.0),
                     (0.0, 0.5, -2.5)),                                                     # scaled, rotated box
        sc.scale(sc.translate(sc.cone_frustum((0.0, 0.0, 0.0), 1.0, 0.5, 1.0, green), (2.0, 0.0, -1.0)), 1.0, 2.0, -1.0),  # mirrored frustum
        sc.translate(sc.rotate(sc.disc((0.0, 0.0, 0.0), 1.25, (0.0, 1.0, 0.0), steel), (1.0, 0.0, 0.0), 70.0), (-5.0, 2.0, -3.0)),  # tilted mirror disc
    ]
    if light == "disc":
        light_node = sc.disc((0.0, 7.5, 0.0), 1.5, (0.0, -1.0, 0.0), lamp)
    else:
        light_node = sc.cone_frustum((0.0, 6.5, 0.0), 1.0, 0.75, 1.5, lamp)
    world.append(light_node)
    sc.set_world(sc.bvh(world))
    sc.set_lights(light_node)
    _camera(
        sc, width, spp, depth, (400, 64, 20),
        aspect_ratio=4.0 / 3.0, vertical_fov=35.0, look_from=(0.0, 5.0, 16.0), look_at=(0.0, 2.0, 0.0), up=(0.0, 1.0, 0.0),
        defocus_angle=0.0, background=(0.02, 0.02, 0.03),
    )
    return sc


def scad_shapes_cone_light(sc, **kw):
    return scad_shapes(sc, light="cone", **kw)


def scad_lattice(sc, n=100, width=None, spp=None, depth=None, seed=1, bulk=None):
    """Synthetic scene with the shape crates/openscad's interpreter emits for nested `for` loops of
    `translate() sphere()` (crates/openscad/src/interpreter/modules.rs:139-163,274; mod.rs:190-194):
    one top-level BVH over n^3 `Translate(Group[Sphere at origin])` chains plus a ground sphere, no lights.
    Radii U(.2,.45); 80/15/5 % Lambertian / Metal / Dielectric."""
    rng = SceneRandom(seed)
    total = n * n * n
    u = rng.rand_array(total * 6).reshape(total, 6)
    world = [sc.sphere((0.0, -1000.5, 0.0), 1000.0, sc.lambertian_color(0.5, 0.5, 0.5))]
    idx = 0
    half = n / 2.0
    if bulk is None:
        bulk = hasattr(sc.be.lib, sc.be.prefix + "scene_from_flat")
    if bulk:
        # the same nodes in the same creation order, handed over in one call
        from . import bindings as B

        mats = np.empty(total, dtype=np.int32)
        for e in range(total):
            r = u[e]
            if r[1] < 0.8:
                mats[e] = sc.lambertian_color(r[2] * r[3], r[3] * r[4], r[4] * r[5])
            elif r[1] < 0.95:
                mats[e] = sc.metal(0.5 + 0.5 * r[2], 0.5 + 0.5 * r[3], 0.5 + 0.5 * r[4], 0.5 * r[5])
            else:
                mats[e] = sc.dielectric(1.5)
        rec = np.zeros(3 * total + 1, dtype=B.FLAT_NODE_DTYPE)
        e = np.arange(total)
        sph, grp, tr = rec[0:-1:3], rec[1:-1:3], rec[2:-1:3]
        sph["kind"] = B.NODE_SPHERE
        sph["mat"] = mats
        sph["v"][:, 3] = 0.2 + 0.25 * u[:, 0]
        grp["kind"] = B.NODE_GROUP
        grp["first"] = e
        grp["count"] = 1
        tr["kind"] = B.NODE_TRANSLATE
        tr["first"] = 3 * e + 1
        i3, j3, k3 = e // (n * n), (e // n) % n, e % n
        tr["v"][:, 0] = i3 - half + 0.5
        tr["v"][:, 1] = j3 + 0.5
        tr["v"][:, 2] = k3 - half + 0.5
        members = np.concatenate([3 * e, [~np.int32(world[0])], 3 * e + 2]).astype(np.int32)
        rec[-1]["kind"] = B.NODE_BVH
        rec[-1]["first"] = total
        rec[-1]["count"] = total + 1
        ids = sc.from_flat(rec, members)
        sc.set_world(int(ids[-1]))
        n_iter = 0
    else:
        n_iter = n
    for i in range(n_iter):
        for j in range(n):
            for k in range(n):
                r = u[idx]
                radius = 0.2 + 0.25 * r[0]
                if r[1] < 0.8:
                    m = sc.lambertian_color(r[2] * r[3], r[3] * r[4], r[4] * r[5])
                elif r[1] < 0.95:
                    m = sc.metal(0.5 + 0.5 * r[2], 0.5 + 0.5 * r[3], 0.5 + 0.5 * r[4], 0.5 * r[5])
                else:
                    m = sc.dielectric(1.5)
                s = sc.sphere((0.0, 0.0, 0.0), radius, m)
                g = sc.group([s])
                world.append(sc.translate(g, (i - half + 0.5, j + 0.5, k - half + 0.5)))
                idx += 1
    if not bulk:
        sc.set_world(sc.bvh(world))
    sc.set_lights(None)
    d = 1.6 * n
    _camera(
        sc, width, spp, depth, (1920, 256, 50),
        aspect_ratio=16.0 / 9.0, vertical_fov=40.0, look_from=(d, 0.9 * n, d), look_at=(0.0, 0.45 * n, 0.0), up=(0.0, 1.0, 0.0),
        defocus_angle=0.0, background=(0.7, 0.8, 1.0),
    )
    return sc


SCENES = {
    "random_spheres": random_spheres,
    "final": final_scene,
    "cornell_box": cornell_box,
    "cornell_box_smoke": cornell_box_smoke,
    "three_spheres": three_spheres,
    "checkered_spheres": checkered_spheres,
    "earth": earth,
    "perlin_spheres": perlin_spheres,
    "quads": quads,
    "lighted_sphere": lighted_sphere,
    "lighted_cone_frustum": lighted_cone_frustum,
    "scad_shapes": scad_shapes,
    "scad_shapes_cone_light": scad_shapes_cone_light,
}


def build(backend, name, **kw):
    """Creates, fills and commits a scene on `backend`.  `name` is one of SCENES or the path of a `.scad` file,
    the two forms the reference's command line takes (crates/cli/src/main.rs:30-62)."""
    if name.endswith(".scad"):
        from . import scad  # (scad imports this module)
        import os
        folder = os.path.dirname(os.path.abspath(name))
        loader = kw.pop("image_loader", None)
        with open(name) as fh:
            sc, messages = scad.run_openscad(fh.read(), backend, scene_seed=kw.pop("seed", 1), image_loader=loader)
        errors = [text for level, text in messages if level == "error"]
        if errors:
            raise scad.ScadError(f"{os.path.basename(name)} in {folder}: " + "; ".join(errors))
        return sc
    sc = backend.new_scene()
    SCENES[name](sc, **kw)
    sc.commit()
    return sc
