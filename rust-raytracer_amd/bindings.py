"""ctypes bindings over the C ABI of include/caustic_hip.h.

`Backend` wraps one shared library that exports the scene vocabulary of the
reference's crates/core (sphere, quad, box, group, bvh, translate, rotate,
constant medium, materials, textures, camera) under a function-name prefix.
The product library (`libcaustic_hip.so`, prefix ``chip_``) is loaded with
`load_gpu()`; it has no CPU fallback and raises if the library or a HIP device
is missing.  The same wrapper class is reused by the test suite for the CPU
oracle, which exports the same vocabulary under another prefix.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CAUSTIC_HIP_LIB selects another build of the same library (used to A/B kernel variants).
GPU_LIB_PATH = os.environ.get("CAUSTIC_HIP_LIB") or os.path.join(_HERE, "libcaustic_hip.so")


class CameraDesc(C.Structure):
    """CameraBuilder's fields, crates/core/src/camera.rs:34-89 (defaults :106-120)."""

    _fields_ = [
        ("vertical_fov", C.c_double),
        ("aspect_ratio", C.c_double),
        ("image_width", C.c_uint32),
        ("reserved0", C.c_uint32),
        ("look_from", C.c_double * 3),
        ("look_at", C.c_double * 3),
        ("up", C.c_double * 3),
        ("defocus_angle", C.c_double),
        ("focus_distance", C.c_double),
        ("samples_per_pixel", C.c_uint32),
        ("max_depth", C.c_uint32),
        ("background", C.c_double * 3),
    ]


class RenderOpts(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64),
        ("stratum_offset", C.c_uint32),
        ("stratum_stride", C.c_uint32),
        ("xmin", C.c_uint32),
        ("xmax", C.c_uint32),
        ("ymin", C.c_uint32),
        ("ymax", C.c_uint32),
        ("strata_per_batch", C.c_uint32),
        ("flags", C.c_uint32),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("paths", C.c_uint64),
        ("segments", C.c_uint64),
        ("bvh_visits", C.c_uint64),
        ("prim_tests", C.c_uint64),
        ("ms_total", C.c_double),
        ("ms_trace", C.c_double),
        ("ms_shade", C.c_double),
        ("ms_other", C.c_double),
        ("trace_launches", C.c_uint64),
        ("batches", C.c_uint64),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


HIT_DTYPE = np.dtype(
    [
        ("t", "<f8"),
        ("pt", "<f8", (3,)),
        ("normal", "<f8", (3,)),
        ("u", "<f8"),
        ("v", "<f8"),
        ("hit", "<i4"),
        ("leaf_id", "<i4"),
        ("front_face", "<i4"),
        ("material_kind", "<i4"),
    ]
)
assert HIT_DTYPE.itemsize == 88

RENDER_TIME_KERNELS = 1
RENDER_EXACT_SHADING = 2  # shade kernels keep the reference's f64 operation order (see caustic_hip.h)
RENDER_REFERENCE_ORDER = 4  # literal replay of the reference's traversal order instead of the ordered traversal
RENDER_QUANTISE_255 = 8  # 8-bit output scaled by 255.0 (wasm caller) instead of 255.999 (CLI)

NODE_KIND_NAMES = {0: "bvh", 1: "sphere", 2: "quad", 3: "group", 4: "translate", 5: "rotate", 6: "medium", 7: "box", 8: "disc", 9: "cone",
                   10: "cone_wall", 11: "scale"}
(NODE_BVH, NODE_SPHERE, NODE_QUAD, NODE_GROUP, NODE_TRANSLATE, NODE_ROTATE, NODE_MEDIUM, NODE_BOX, NODE_DISC, NODE_CONE, NODE_CONE_WALL,
 NODE_SCALE) = range(12)  # CHIP_NODE_* of include/caustic_hip.h


# chip_block_fn: (user, xmin, xmax, ymin, ymax, rgb8, strata_done, strata_total)
BLOCK_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint8), C.c_uint32, C.c_uint32)


class CausticError(RuntimeError):
    pass


class FlatScene(C.Structure):
    _fields_ = [("nodes", C.c_void_p), ("n_nodes", C.c_size_t), ("members", C.c_void_p), ("n_members", C.c_size_t)]


# chip_flat_node: one constructor call per record (kind, material, child / member range, flags, scalar arguments)
FLAT_NODE_DTYPE = np.dtype(
    [("kind", "<i4"), ("mat", "<i4"), ("first", "<i4"), ("count", "<i4"), ("flags", "<u4"), ("reserved", "<u4"), ("v", "<f8", (9,))]
)
assert FLAT_NODE_DTYPE.itemsize == 96
SCENE_EXACT_SHADING_FORCED = 1
SCENE_REFERENCE_ORDER_FORCED = 2
SCENE_FUSED_FIRST_BOUNCE = 4
FLAT_MOVING = 1
FLAT_MEDIUM_TEX = 2


def _vec3(v):
    return (C.c_double * 3)(float(v[0]), float(v[1]), float(v[2]))


def _ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


class Backend:
    """One loaded library exporting `<prefix>scene_create`, `<prefix>sphere`, ..."""

    def __init__(self, lib: C.CDLL, prefix: str, name: str, sum_dtype=np.float32):
        self.lib = lib
        self.prefix = prefix
        self.name = name
        self.sum_dtype = np.dtype(sum_dtype)  # what render_frame writes: the GPU library float, the oracle double
        self._declare()

    def fn(self, name):
        return getattr(self.lib, self.prefix + name)

    def _declare(self):
        d = C.c_double
        i = C.c_int
        p = C.c_void_p
        pd = C.POINTER(C.c_double)
        pi32 = C.POINTER(C.c_int32)
        sigs = {
            "last_error": (C.c_char_p, []),
            "scene_create": (p, []),
            "scene_destroy": (None, [p]),
            "tex_solid": (i, [p, d, d, d]),
            "tex_checker": (i, [p, d, i, i]),
            "tex_image": (i, [p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint8)]),
            "tex_perlin_noise": (i, [p, d, pd, pi32, pi32, pi32]),
            "tex_perlin_turbulence": (i, [p, d, C.c_uint32, pd, pi32, pi32, pi32]),
            "lambertian_color": (i, [p, d, d, d]),
            "lambertian_tex": (i, [p, i]),
            "metal": (i, [p, d, d, d, d]),
            "dielectric": (i, [p, d]),
            "diffuse_light_color": (i, [p, d, d, d]),
            "diffuse_light_tex": (i, [p, i]),
            "empty_material": (i, [p]),
            "sphere": (i, [p, pd, d, i]),
            "sphere_set_direction": (i, [p, i, pd]),
            "quad": (i, [p, pd, pd, pd, i]),
            "box": (i, [p, pd, pd, i]),
            "group": (i, [p, pi32, i]),
            "bvh": (i, [p, pi32, i]),
            "translate": (i, [p, i, pd]),
            "rotate": (i, [p, i, pd, d]),
            "scale": (i, [p, i, d, d, d]),
            "disc": (i, [p, pd, d, pd, i]),
            "cone_frustum": (i, [p, pd, d, d, d, i]),
            "constant_medium_color": (i, [p, i, d, d, d, d]),
            "constant_medium_tex": (i, [p, i, d, i]),
            "scene_set_world": (i, [p, i]),
            "scene_set_lights": (i, [p, i]),
            "scene_set_camera": (i, [p, C.POINTER(CameraDesc)]),
            "commit": (i, [p]),
            "image_size": (i, [p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
            "sqrt_spp": (i, [p]),
            "camera_derived": (i, [p, pd]),
            "trace_rays": (i, [p, C.c_uint64, C.c_size_t, pd, pd, pd, p]),
            "camera_rays": (i, [p, C.c_uint64, C.c_size_t, C.POINTER(C.c_uint32), pd]),
            "dump_tree": (C.c_int64, [p, pi32, pi32, pd, C.c_size_t]),
            "render_frame": (i, [p, C.POINTER(RenderOpts), p, C.POINTER(C.c_uint8), C.POINTER(Stats)]),
        }
        for name, (res, args) in sigs.items():
            f = self.fn(name)
            f.restype = res
            f.argtypes = args
        # entry points only the GPU library has
        for name, (res, args) in {
            "device_count": (i, []),
            "render_accumulate_device": (i, [p, C.POINTER(RenderOpts), p, p, C.POINTER(Stats)]),
            "resolve_device": (i, [p, p, C.c_size_t, p, p, p]),
            "scene_from_flat": (C.c_int64, [p, C.POINTER(FlatScene), pi32]),
            "commit_devices": (i, [p, pi32, i]),
            "check_culling_trees": (C.c_int64, [p]),
            "render_frame_blocks": (i, [p, C.POINTER(RenderOpts), C.c_uint32, C.c_uint32, BLOCK_FN, p, C.POINTER(Stats)]),
            "write_png": (i, [C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint8)]),
            "render_frame_multi": (i, [p, C.POINTER(RenderOpts), p, C.POINTER(C.c_uint8), C.POINTER(Stats)]),
            "render_accumulate_noalloc": (i, [p, C.POINTER(RenderOpts), p, p, C.POINTER(Stats)]),
            "noalloc_max_pixels": (C.c_uint32, []),
            "scene_traits": (i, [p]),
        }.items():
            if hasattr(self.lib, self.prefix + name):
                f = self.fn(name)
                f.restype = res
                f.argtypes = args

    def last_error(self) -> str:
        m = self.fn("last_error")()
        return m.decode() if m else ""

    def check(self, rc):
        if rc < 0:
            raise CausticError(f"{self.name}: {self.last_error()}")
        return rc

    def new_scene(self) -> "Scene":
        return Scene(self)


@dataclass
class Scene:
    """Handle-based builder mirroring crates/core's constructors; ids are creation-order ints."""

    be: Backend
    h: int = field(init=False)

    def __post_init__(self):
        self.h = self.be.fn("scene_create")()
        if not self.h:
            raise CausticError("scene_create failed")
        self._keep = []

    def close(self):
        if self.h:
            self.be.fn("scene_destroy")(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _call(self, name, *args):
        return self.be.check(self.be.fn(name)(self.h, *args))

    # textures
    def tex_solid(self, r, g, b):
        return self._call("tex_solid", r, g, b)

    def tex_checker(self, scale, even, odd):
        return self._call("tex_checker", scale, even, odd)

    def tex_image(self, rgb: np.ndarray):
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        h, w, c = rgb.shape
        assert c == 3
        return self._call("tex_image", w, h, _ptr(rgb, C.c_uint8))

    def _perlin_args(self, tables):
        rv = np.ascontiguousarray(tables["rand_vec"], dtype=np.float64)
        px = np.ascontiguousarray(tables["perm_x"], dtype=np.int32)
        py = np.ascontiguousarray(tables["perm_y"], dtype=np.int32)
        pz = np.ascontiguousarray(tables["perm_z"], dtype=np.int32)
        assert rv.shape == (256, 3) and px.shape == py.shape == pz.shape == (256,)
        return rv, px, py, pz

    def tex_perlin_noise(self, scale, tables):
        rv, px, py, pz = self._perlin_args(tables)
        return self._call("tex_perlin_noise", scale, _ptr(rv, C.c_double), _ptr(px, C.c_int32), _ptr(py, C.c_int32), _ptr(pz, C.c_int32))

    def tex_perlin_turbulence(self, scale, depth, tables):
        rv, px, py, pz = self._perlin_args(tables)
        return self._call(
            "tex_perlin_turbulence", scale, depth, _ptr(rv, C.c_double), _ptr(px, C.c_int32), _ptr(py, C.c_int32), _ptr(pz, C.c_int32)
        )

    # materials
    def lambertian_color(self, r, g, b):
        return self._call("lambertian_color", r, g, b)

    def lambertian_tex(self, tex):
        return self._call("lambertian_tex", tex)

    def metal(self, r, g, b, fuzz):
        return self._call("metal", r, g, b, fuzz)

    def dielectric(self, ri):
        return self._call("dielectric", ri)

    def diffuse_light_color(self, r, g, b):
        return self._call("diffuse_light_color", r, g, b)

    def diffuse_light_tex(self, tex):
        return self._call("diffuse_light_tex", tex)

    def empty_material(self):
        return self._call("empty_material")

    # nodes
    def sphere(self, center, radius, mat, direction=None):
        n = self._call("sphere", _vec3(center), radius, mat)
        if direction is not None:
            self._call("sphere_set_direction", n, _vec3(direction))
        return n

    def sphere_set_direction(self, node, direction):
        return self._call("sphere_set_direction", node, _vec3(direction))

    def quad(self, q, u, v, mat):
        return self._call("quad", _vec3(q), _vec3(u), _vec3(v), mat)

    def box(self, a, b, mat):
        return self._call("box", _vec3(a), _vec3(b), mat)

    def _ids(self, nodes):
        a = np.ascontiguousarray(nodes, dtype=np.int32)
        return _ptr(a, C.c_int32), int(a.size), a

    def group(self, nodes):
        p, n, keep = self._ids(nodes)
        return self._call("group", p, n)

    def bvh(self, nodes):
        p, n, keep = self._ids(nodes)
        return self._call("bvh", p, n)

    def translate(self, node, offset):
        return self._call("translate", node, _vec3(offset))

    def rotate(self, node, axis, degrees):
        return self._call("rotate", node, _vec3(axis), degrees)

    def scale(self, node, sx, sy, sz):
        return self._call("scale", node, sx, sy, sz)

    def disc(self, center, radius, normal, mat):
        return self._call("disc", _vec3(center), radius, _vec3(normal), mat)

    def cone_frustum(self, base, height, top_radius, bottom_radius, mat):
        return self._call("cone_frustum", _vec3(base), height, top_radius, bottom_radius, mat)

    def rotate_y(self, node, degrees):  # rotate.rs:87-89
        return self.rotate(node, (0.0, 1.0, 0.0), degrees)

    def constant_medium_color(self, boundary, density, r, g, b):
        return self._call("constant_medium_color", boundary, density, r, g, b)

    def constant_medium_tex(self, boundary, density, tex):
        return self._call("constant_medium_tex", boundary, density, tex)

    def from_flat(self, records: np.ndarray, members=None) -> np.ndarray:
        """Runs one constructor per record of a FLAT_NODE_DTYPE array in a single library call
        (`scene_from_flat`); returns the node id of every record."""
        records = np.ascontiguousarray(records, dtype=FLAT_NODE_DTYPE)
        members = np.ascontiguousarray(members if members is not None else [], dtype=np.int32)
        ids = np.empty(records.size, dtype=np.int32)
        fs = FlatScene(records.ctypes.data, records.size, members.ctypes.data, members.size)
        self._call("scene_from_flat", C.byref(fs), _ptr(ids, C.c_int32))
        return ids

    def replay_flat(self, records: np.ndarray, members=None) -> np.ndarray:
        """The loop over the per-node constructors that `from_flat` is defined to equal."""
        members = np.asarray(members if members is not None else [], dtype=np.int32)
        ids = np.full(records.size, -1, dtype=np.int32)

        def ref(c, i):
            if c < 0:
                return int(~c)
            if c >= i:
                raise CausticError(f"flat record {i} refers to a record that is not before it")
            return int(ids[c])

        for i, r in enumerate(records):
            k, v, mat = int(r["kind"]), [float(x) for x in r["v"]], int(r["mat"])
            if k == NODE_SPHERE:
                ids[i] = self.sphere(v[0:3], v[3], mat, v[4:7] if int(r["flags"]) & FLAT_MOVING else None)
            elif k == NODE_QUAD:
                ids[i] = self.quad(v[0:3], v[3:6], v[6:9], mat)
            elif k == NODE_BOX:
                ids[i] = self.box(v[0:3], v[3:6], mat)
            elif k == NODE_DISC:
                ids[i] = self.disc(v[0:3], v[3], v[4:7], mat)
            elif k == NODE_CONE:
                ids[i] = self.cone_frustum(v[0:3], v[3], v[4], v[5], mat)
            elif k in (NODE_GROUP, NODE_BVH):
                m = [ref(int(c), i) for c in members[int(r["first"]) : int(r["first"]) + int(r["count"])]]
                ids[i] = self.group(m) if k == NODE_GROUP else self.bvh(m)
            elif k == NODE_TRANSLATE:
                ids[i] = self.translate(ref(int(r["first"]), i), v[0:3])
            elif k == NODE_ROTATE:
                ids[i] = self.rotate(ref(int(r["first"]), i), v[0:3], v[3])
            elif k == NODE_SCALE:
                ids[i] = self.scale(ref(int(r["first"]), i), v[0], v[1], v[2])
            elif k == NODE_MEDIUM:
                if int(r["flags"]) & FLAT_MEDIUM_TEX:
                    ids[i] = self.constant_medium_tex(ref(int(r["first"]), i), v[0], mat)
                else:
                    ids[i] = self.constant_medium_color(ref(int(r["first"]), i), v[0], v[1], v[2], v[3])
            else:
                raise CausticError(f"flat record {i}: kind {k} has no constructor")
        return ids

    # scene data
    def set_world(self, node):
        return self._call("scene_set_world", node)

    def set_lights(self, node):
        return self._call("scene_set_lights", -1 if node is None else node)

    def set_camera(self, **kw):
        d = CameraDesc()
        d.vertical_fov = kw.get("vertical_fov", 90.0)
        d.aspect_ratio = kw.get("aspect_ratio", 1.0)
        d.image_width = kw.get("image_width", 100)
        d.look_from = _vec3(kw.get("look_from", (0, 0, 0)))
        d.look_at = _vec3(kw.get("look_at", (0, 0, -1)))
        d.up = _vec3(kw.get("up", (0, 1, 0)))
        d.defocus_angle = kw.get("defocus_angle", 0.0)
        d.focus_distance = kw.get("focus_distance", 10.0)
        d.samples_per_pixel = kw.get("samples_per_pixel", 10)
        d.max_depth = kw.get("max_depth", 10)
        d.background = _vec3(kw.get("background", (0, 0, 0)))
        return self._call("scene_set_camera", C.byref(d))

    def commit(self):
        return self._call("commit")

    def image_size(self):
        w, h = C.c_uint32(), C.c_uint32()
        self._call("image_size", C.byref(w), C.byref(h))
        return w.value, h.value

    def sqrt_spp(self):
        return self._call("sqrt_spp")

    def traits(self):
        """SCENE_* bits: what takes the committed scene off the library's default render path."""
        return self._call("scene_traits")

    def camera_derived(self):
        out = np.zeros(24, dtype=np.float64)
        self._call("camera_derived", _ptr(out, C.c_double))
        return out

    # rendering
    def _opts(self, seed=1, stratum_offset=0, stratum_stride=1, rect=None, strata_per_batch=0, flags=0):
        o = RenderOpts()
        o.seed = seed
        o.stratum_offset = stratum_offset
        o.stratum_stride = stratum_stride
        if rect is not None:
            o.xmin, o.xmax, o.ymin, o.ymax = rect
        o.strata_per_batch = strata_per_batch
        o.flags = flags
        return o

    def _rect_shape(self, rect):
        w, h = self.image_size()
        if rect is None:
            return h, w
        return rect[3] - rect[2], rect[1] - rect[0]

    def render_frame(self, seed=1, stratum_offset=0, stratum_stride=1, rect=None, strata_per_batch=0, flags=0, want_rgb8=False, sum_dtype=None,
                     multi=False):
        """Returns (sum_rgb[h,w,3], rgb8 or None, stats dict).  sum_rgb is the un-normalised radiance sum, in the
        backend's own type (passing another `sum_dtype` is an error: the library writes its type into the buffer).
        `multi` renders through `render_frame_multi` (after `commit_devices`)."""
        if sum_dtype is not None and np.dtype(sum_dtype) != self.be.sum_dtype:
            raise CausticError(f"{self.be.name} writes {self.be.sum_dtype} radiance sums, not {np.dtype(sum_dtype)}")
        o = self._opts(seed, stratum_offset, stratum_stride, rect, strata_per_batch, flags)
        rh, rw = self._rect_shape(rect)
        out = np.zeros((rh, rw, 3), dtype=self.be.sum_dtype)
        rgb8 = np.zeros((rh, rw, 3), dtype=np.uint8) if want_rgb8 else None
        st = Stats()
        self._call(
            "render_frame_multi" if multi else "render_frame",
            C.byref(o),
            out.ctypes.data_as(C.c_void_p),
            _ptr(rgb8, C.c_uint8) if want_rgb8 else None,
            C.byref(st),
        )
        return out, rgb8, st.as_dict()

    def render_accumulate_device(self, d_sum_ptr: int, seed=1, stratum_offset=0, stratum_stride=1, rect=None, strata_per_batch=0, flags=0, stream=0):
        """Adds this call's radiance sums into a device float buffer (rect_h*rect_w*3) owned by the caller."""
        o = self._opts(seed, stratum_offset, stratum_stride, rect, strata_per_batch, flags)
        st = Stats()
        self._call("render_accumulate_device", C.byref(o), C.c_void_p(d_sum_ptr), C.c_void_p(stream), C.byref(st))
        return st.as_dict()

    def render_accumulate_noalloc(self, d_sum_ptr: int, seed=1, stratum_offset=0, stratum_stride=1, rect=None, strata_per_batch=0, flags=0, stream=0):
        """`render_accumulate_device` through the entry point that allocates nothing (small rectangles only)."""
        o = self._opts(seed, stratum_offset, stratum_stride, rect, strata_per_batch, flags)
        st = Stats()
        self._call("render_accumulate_noalloc", C.byref(o), C.c_void_p(d_sum_ptr), C.c_void_p(stream), C.byref(st))
        return st.as_dict()

    def render_frame_blocks(self, on_block, block_size=10, passes=1, seed=1, stratum_offset=0, stratum_stride=1, rect=None, flags=0):
        """Renders progressively and calls on_block(xmin, xmax, ymin, ymax, rgb8[h, w, 3], strata_done, strata_total) per tile
        and pass (crates/cli's block loop / the web worker pool's per-result repaint)."""
        o = self._opts(seed, stratum_offset, stratum_stride, rect, 0, flags)
        st = Stats()

        def thunk(_user, xmin, xmax, ymin, ymax, data, done, total):
            n = (xmax - xmin) * (ymax - ymin) * 3
            on_block(xmin, xmax, ymin, ymax, np.ctypeslib.as_array(data, shape=(n,)).reshape(ymax - ymin, xmax - xmin, 3).copy(), done, total)

        self._call("render_frame_blocks", C.byref(o), block_size, passes, BLOCK_FN(thunk), None, C.byref(st))
        return st.as_dict()

    def commit_devices(self, devices):
        a = np.ascontiguousarray(devices, dtype=np.int32)
        return self._call("commit_devices", _ptr(a, C.c_int32), int(a.size))

    def resolve_device(self, d_sum_ptr: int, npixels: int, d_gamma_ptr: int = 0, d_rgb8_ptr: int = 0, stream=0):
        return self._call(
            "resolve_device", C.c_void_p(d_sum_ptr), npixels, C.c_void_p(d_gamma_ptr or None), C.c_void_p(d_rgb8_ptr or None), C.c_void_p(stream)
        )

    def trace_rays(self, origin, direction, time=None, seed=1):
        o = np.ascontiguousarray(origin, dtype=np.float64)
        d = np.ascontiguousarray(direction, dtype=np.float64)
        n = o.shape[0]
        assert o.shape == (n, 3) and d.shape == (n, 3)
        t = None
        if time is not None:
            t = np.ascontiguousarray(time, dtype=np.float64)
            assert t.shape == (n,)
        out = np.zeros(n, dtype=HIT_DTYPE)
        self._call(
            "trace_rays", seed, n, _ptr(o, C.c_double), _ptr(d, C.c_double), _ptr(t, C.c_double) if t is not None else None, out.ctypes.data_as(C.c_void_p)
        )
        return out

    def camera_rays(self, xyss, seed=1):
        a = np.ascontiguousarray(xyss, dtype=np.uint32)
        n = a.shape[0]
        assert a.shape == (n, 4)
        out = np.zeros((n, 7), dtype=np.float64)
        self._call("camera_rays", seed, n, _ptr(a, C.c_uint32), _ptr(out, C.c_double))
        return out

    def check_culling_trees(self):
        """Host-side consistency check of the GPU library's culling trees; returns the number of primitives they reach."""
        return self.be.check(self.be.fn("check_culling_trees")(self.h))

    def dump_tree(self):
        """Preorder (kind, creation id, bbox) records of the world tree."""
        n = self.be.check(self.be.fn("dump_tree")(self.h, None, None, None, 0))
        kinds = np.zeros(n, dtype=np.int32)
        ids = np.zeros(n, dtype=np.int32)
        boxes = np.zeros((n, 6), dtype=np.float64)
        m = self.be.check(self.be.fn("dump_tree")(self.h, _ptr(kinds, C.c_int32), _ptr(ids, C.c_int32), _ptr(boxes, C.c_double), n))
        assert m == n
        return kinds, ids, boxes


_gpu_backend = None


def load_gpu() -> Backend:
    """Loads libcaustic_hip.so.  Raises if the library is missing: there is no CPU fallback."""
    global _gpu_backend
    if _gpu_backend is None:
        if not os.path.exists(GPU_LIB_PATH):
            raise CausticError(f"{GPU_LIB_PATH} is missing: build it with `make -C {os.path.join(_HERE, 'csrc')}` (hipcc, gfx950)")
        _gpu_backend = Backend(C.CDLL(GPU_LIB_PATH), "chip_", "libcaustic_hip")
    return _gpu_backend


def write_png(path: str, rgb8: np.ndarray):
    """crates/cli/src/main.rs:188: the frame as an 8-bit RGB PNG file (host-side, no GPU involved)."""
    be = load_gpu()
    a = np.ascontiguousarray(rgb8, dtype=np.uint8)
    h, w, c = a.shape
    assert c == 3
    be.check(be.fn("write_png")(os.fsencode(path), w, h, _ptr(a, C.c_uint8)))


def gpu_device_count() -> int:
    be = load_gpu()
    return be.check(be.fn("device_count")())
