//! Raw FFI for `libcaustic_hip` (see `include/caustic_hip.h` of the caustic-hip repository, whose comments cite the
//! `caustic_core` constructor or method each function mirrors).  Every function returns `< 0` on error with the
//! message in `chip_last_error()`; ids returned by the scene-building calls are plain `c_int`s that belong to the
//! scene they were created in.  Written without a Rust toolchain at hand: treat as a starting point.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct chip_scene {
    _private: [u8; 0],
}

/// The fields of `caustic_core::CameraBuilder` (camera.rs:34-89).
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct chip_camera_desc {
    pub vertical_fov: f64,
    pub aspect_ratio: f64,
    pub image_width: u32,
    pub reserved0: u32,
    pub look_from: [f64; 3],
    pub look_at: [f64; 3],
    pub up: [f64; 3],
    pub defocus_angle: f64,
    pub focus_distance: f64,
    pub samples_per_pixel: u32,
    pub max_depth: u32,
    pub background: [f64; 3],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct chip_render_opts {
    pub seed: u64,
    /// This caller renders strata `offset, offset + stride, ...` of the `sqrt_spp^2` strata of every pixel.
    pub stratum_offset: u32,
    pub stratum_stride: u32,
    /// Pixel rectangle `[xmin, xmax) x [ymin, ymax)`; all zero = the whole image.
    pub xmin: u32,
    pub xmax: u32,
    pub ymin: u32,
    pub ymax: u32,
    pub strata_per_batch: u32,
    pub flags: u32,
}

pub const CHIP_RENDER_TIME_KERNELS: u32 = 1;
pub const CHIP_RENDER_EXACT_SHADING: u32 = 2;
pub const CHIP_RENDER_REFERENCE_ORDER: u32 = 4;
pub const CHIP_RENDER_QUANTISE_255: u32 = 8;
pub const CHIP_SCENE_EXACT_SHADING_FORCED: c_int = 1;
pub const CHIP_SCENE_REFERENCE_ORDER_FORCED: c_int = 2;
pub const CHIP_SCENE_FUSED_FIRST_BOUNCE: c_int = 4;

/// `HitRecord` of `object/mod.rs:31-39`, filled by `chip_trace_rays`.
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct chip_hit {
    pub t: f64,
    pub pt: [f64; 3],
    pub normal: [f64; 3],
    pub u: f64,
    pub v: f64,
    pub hit: i32,
    pub leaf_id: i32,
    pub front_face: i32,
    pub material_kind: i32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct chip_stats {
    pub paths: u64,
    pub segments: u64,
    pub bvh_visits: u64,
    pub prim_tests: u64,
    pub ms_total: f64,
    pub ms_trace: f64,
    pub ms_shade: f64,
    pub ms_other: f64,
    pub trace_launches: u64,
    pub batches: u64,
}

pub const CHIP_FLAT_MOVING: u32 = 1;
pub const CHIP_FLAT_MEDIUM_TEX: u32 = 2;

/// One constructor call of `chip_scene_from_flat` (see caustic_hip.h for the meaning of `v` per kind).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct chip_flat_node {
    pub kind: i32,
    pub mat: i32,
    pub first: i32,
    pub count: i32,
    pub flags: u32,
    pub reserved: u32,
    pub v: [f64; 9],
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct chip_flat_scene {
    pub nodes: *const chip_flat_node,
    pub n_nodes: usize,
    pub members: *const i32,
    pub n_members: usize,
}

/// Tile callback of `chip_render_frame_blocks`: (user, xmin, xmax, ymin, ymax, rgb8, strata_done, strata_total).
pub type chip_block_fn = unsafe extern "C" fn(*mut c_void, u32, u32, u32, u32, *const u8, u32, u32);

// Layout of everything that crosses the ABI: the numbers of tests/abi/abi_check.c (which asserts them against the C
// header) and of the ctypes structures in rust-raytracer_amd/bindings.py.
macro_rules! abi_layout {
    ($t:ty, $size:expr, $($f:ident = $off:expr),*) => {
        const _: () = assert!(core::mem::size_of::<$t>() == $size);
        $(const _: () = assert!(core::mem::offset_of!($t, $f) == $off);)*
    };
}
abi_layout!(chip_camera_desc, 144, vertical_fov = 0, aspect_ratio = 8, image_width = 16, reserved0 = 20, look_from = 24, look_at = 48, up = 72,
            defocus_angle = 96, focus_distance = 104, samples_per_pixel = 112, max_depth = 116, background = 120);
abi_layout!(chip_render_opts, 40, seed = 0, stratum_offset = 8, stratum_stride = 12, xmin = 16, xmax = 20, ymin = 24, ymax = 28, strata_per_batch = 32, flags = 36);
abi_layout!(chip_hit, 88, t = 0, pt = 8, normal = 32, u = 56, v = 64, hit = 72, leaf_id = 76, front_face = 80, material_kind = 84);
abi_layout!(chip_stats, 80, paths = 0, segments = 8, bvh_visits = 16, prim_tests = 24, ms_total = 32, ms_trace = 40, ms_shade = 48, ms_other = 56,
            trace_launches = 64, batches = 72);
abi_layout!(chip_flat_node, 96, kind = 0, mat = 4, first = 8, count = 12, flags = 16, reserved = 20, v = 24);
abi_layout!(chip_flat_scene, 32, nodes = 0, n_nodes = 8, members = 16, n_members = 24);

extern "C" {
    pub fn chip_last_error() -> *const c_char;
    pub fn chip_device_count() -> c_int;
    pub fn chip_scene_create() -> *mut chip_scene;
    pub fn chip_scene_destroy(s: *mut chip_scene);
    pub fn chip_tex_solid(s: *mut chip_scene, r: f64, g: f64, b: f64) -> c_int;
    pub fn chip_tex_checker(s: *mut chip_scene, scale: f64, even: c_int, odd: c_int) -> c_int;
    pub fn chip_tex_image(s: *mut chip_scene, w: u32, h: u32, rgb: *const u8) -> c_int;
    pub fn chip_tex_perlin_noise(s: *mut chip_scene, scale: f64, rand_vec: *const f64, perm_x: *const i32, perm_y: *const i32, perm_z: *const i32) -> c_int;
    pub fn chip_tex_perlin_turbulence(s: *mut chip_scene, scale: f64, depth: u32, rand_vec: *const f64, perm_x: *const i32, perm_y: *const i32, perm_z: *const i32) -> c_int;
    pub fn chip_lambertian_color(s: *mut chip_scene, r: f64, g: f64, b: f64) -> c_int;
    pub fn chip_lambertian_tex(s: *mut chip_scene, tex: c_int) -> c_int;
    pub fn chip_metal(s: *mut chip_scene, r: f64, g: f64, b: f64, fuzz: f64) -> c_int;
    pub fn chip_dielectric(s: *mut chip_scene, refraction_index: f64) -> c_int;
    pub fn chip_diffuse_light_color(s: *mut chip_scene, r: f64, g: f64, b: f64) -> c_int;
    pub fn chip_diffuse_light_tex(s: *mut chip_scene, tex: c_int) -> c_int;
    pub fn chip_empty_material(s: *mut chip_scene) -> c_int;
    pub fn chip_sphere(s: *mut chip_scene, center: *const f64, radius: f64, mat: c_int) -> c_int;
    pub fn chip_sphere_set_direction(s: *mut chip_scene, node: c_int, direction: *const f64) -> c_int;
    pub fn chip_quad(s: *mut chip_scene, q: *const f64, u: *const f64, v: *const f64, mat: c_int) -> c_int;
    pub fn chip_box(s: *mut chip_scene, a: *const f64, b: *const f64, mat: c_int) -> c_int;
    pub fn chip_group(s: *mut chip_scene, nodes: *const i32, n: c_int) -> c_int;
    pub fn chip_bvh(s: *mut chip_scene, nodes: *const i32, n: c_int) -> c_int;
    pub fn chip_translate(s: *mut chip_scene, node: c_int, offset: *const f64) -> c_int;
    pub fn chip_rotate(s: *mut chip_scene, node: c_int, axis: *const f64, angle_degrees: f64) -> c_int;
    pub fn chip_scale(s: *mut chip_scene, node: c_int, sx: f64, sy: f64, sz: f64) -> c_int;
    pub fn chip_disc(s: *mut chip_scene, center: *const f64, radius: f64, normal: *const f64, mat: c_int) -> c_int;
    pub fn chip_cone_frustum(s: *mut chip_scene, base: *const f64, height: f64, top_radius: f64, bottom_radius: f64, mat: c_int) -> c_int;
    pub fn chip_constant_medium_color(s: *mut chip_scene, boundary: c_int, density: f64, r: f64, g: f64, b: f64) -> c_int;
    pub fn chip_constant_medium_tex(s: *mut chip_scene, boundary: c_int, density: f64, tex: c_int) -> c_int;
    pub fn chip_scene_from_flat(s: *mut chip_scene, flat: *const chip_flat_scene, node_ids: *mut i32) -> i64;
    pub fn chip_scene_set_world(s: *mut chip_scene, node: c_int) -> c_int;
    pub fn chip_scene_set_lights(s: *mut chip_scene, node: c_int) -> c_int;
    pub fn chip_scene_set_camera(s: *mut chip_scene, desc: *const chip_camera_desc) -> c_int;
    pub fn chip_commit(s: *mut chip_scene) -> c_int;
    pub fn chip_image_size(s: *mut chip_scene, width: *mut u32, height: *mut u32) -> c_int;
    pub fn chip_sqrt_spp(s: *mut chip_scene) -> c_int;
    /// CHIP_SCENE_* bits of the committed scene: what takes it off the default render path (forced exact shading, forced reference order).
    pub fn chip_scene_traits(s: *mut chip_scene) -> c_int;
    pub fn chip_camera_derived(s: *mut chip_scene, out24: *mut f64) -> c_int;
    pub fn chip_render_frame(s: *mut chip_scene, opts: *const chip_render_opts, out_sum_rgb: *mut f32, out_rgb8: *mut u8, stats: *mut chip_stats) -> c_int;
    pub fn chip_render_accumulate_device(s: *mut chip_scene, opts: *const chip_render_opts, d_sum_rgb: *mut f32, stream: *mut c_void, stats: *mut chip_stats) -> c_int;
    pub fn chip_resolve_device(s: *mut chip_scene, d_sum_rgb: *const f32, npixels: usize, d_out_gamma: *mut f32, d_out_rgb8: *mut u8, stream: *mut c_void) -> c_int;
    pub fn chip_resolve_device_flags(s: *mut chip_scene, d_sum_rgb: *const f32, npixels: usize, d_out_gamma: *mut f32, d_out_rgb8: *mut u8, flags: u32, stream: *mut c_void) -> c_int;
    /// The frame tile by tile, progressively (see caustic_hip.h); `on_block` runs on the calling thread.
    pub fn chip_render_frame_blocks(s: *mut chip_scene, opts: *const chip_render_opts, block_size: u32, passes: u32, on_block: chip_block_fn, user: *mut c_void, stats: *mut chip_stats) -> c_int;
    pub fn chip_write_png(path: *const c_char, width: u32, height: u32, rgb8: *const u8) -> c_int;
    /// `chip_render_accumulate_device` without any allocation (rectangle <= `chip_noalloc_max_pixels()` pixels, max_depth <= 254).
    pub fn chip_render_accumulate_noalloc(s: *mut chip_scene, opts: *const chip_render_opts, d_sum_rgb: *mut f32, stream: *mut c_void, stats: *mut chip_stats) -> c_int;
    pub fn chip_noalloc_max_pixels() -> u32;
    /// One process, several GPUs: replicate the committed scene, then render frames whose strata are dealt over the replicas.
    pub fn chip_commit_devices(s: *mut chip_scene, devices: *const c_int, n: c_int) -> c_int;
    pub fn chip_render_frame_multi(s: *mut chip_scene, opts: *const chip_render_opts, out_sum_rgb: *mut f32, out_rgb8: *mut u8, stats: *mut chip_stats) -> c_int;
    pub fn chip_trace_rays(s: *mut chip_scene, seed: u64, n: usize, origin: *const f64, direction: *const f64, time: *const f64, out: *mut chip_hit) -> c_int;
    pub fn chip_camera_rays(s: *mut chip_scene, seed: u64, n: usize, xyss: *const u32, rays: *mut f64) -> c_int;
    pub fn chip_dump_tree(s: *mut chip_scene, kinds: *mut i32, ids: *mut i32, boxes: *mut f64, cap: usize) -> i64;
    pub fn chip_check_culling_trees(s: *mut chip_scene) -> i64;
}

/// The last error message of this thread.
pub fn last_error() -> String {
    unsafe { std::ffi::CStr::from_ptr(chip_last_error()).to_string_lossy().into_owned() }
}
