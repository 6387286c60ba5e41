//! The two pieces a maintainer of the reference wires in (written without a Rust toolchain at hand: a starting point).
//!
//! 1. [`SceneExporter`]: the visitor `caustic_core`'s `Node` / `Material` / `Texture` impls call from an
//!    `export(&self, b: &mut dyn SceneExporter) -> NodeId` method added next to `hit` (`object/mod.rs:54-68`).  Most
//!    fields of the core types are private (`sphere.rs:13-18`, `quad.rs:14-33`, `camera.rs:202-231`), so the walk has to
//!    live inside the crate; every impl forwards the arguments its own constructor took, in the same order.
//! 2. [`HipScene`]: owns the `chip_scene`, implements `SceneExporter` by calling the C ABI, and offers the frame calls
//!    that replace `crates/cli/src/main.rs:119-186` and `crates/wasm/src/lib.rs:215-242`.
use caustic_hip_sys as sys;
use std::collections::HashMap;
use std::ffi::{c_void, CStr, CString};

pub type NodeId = i32;
pub type MaterialId = i32;
pub type TextureId = i32;

#[derive(Debug)]
pub struct HipError(pub String);
pub type Result<T> = std::result::Result<T, HipError>;

/// What `caustic_core` calls while it walks its own scene graph.  One method per constructor of the core crate.
pub trait SceneExporter {
    fn tex_solid(&mut self, rgb: [f64; 3]) -> Result<TextureId>; // solid_color.rs:14-17
    fn tex_checker(&mut self, scale: f64, even: TextureId, odd: TextureId) -> Result<TextureId>; // checker_texture.rs:13-35
    fn tex_image(&mut self, width: u32, height: u32, rgb8: &[u8]) -> Result<TextureId>; // image_texture.rs:16-29
    /// `Perlin`'s tables (utils/perlin.rs:27-36): 256 unit vectors and three permutations.
    fn tex_perlin_noise(&mut self, scale: f64, rand_vec: &[[f64; 3]; 256], perm: [&[i32; 256]; 3]) -> Result<TextureId>;
    fn tex_perlin_turbulence(&mut self, scale: f64, depth: u32, rand_vec: &[[f64; 3]; 256], perm: [&[i32; 256]; 3]) -> Result<TextureId>;
    fn lambertian(&mut self, tex: TextureId) -> Result<MaterialId>; // lambertian.rs:17-19
    fn metal(&mut self, albedo: [f64; 3], fuzz: f64) -> Result<MaterialId>; // metal.rs:14-16
    fn dielectric(&mut self, refraction_index: f64) -> Result<MaterialId>; // dielectric.rs:17-19
    fn diffuse_light(&mut self, tex: TextureId) -> Result<MaterialId>; // diffuse_light.rs:16-18
    fn empty_material(&mut self) -> Result<MaterialId>; // empty.rs:7-9
    fn sphere(&mut self, center: [f64; 3], radius: f64, direction: Option<[f64; 3]>, mat: MaterialId) -> Result<NodeId>; // sphere.rs:21-34
    fn quad(&mut self, q: [f64; 3], u: [f64; 3], v: [f64; 3], mat: MaterialId) -> Result<NodeId>; // quad.rs:59-76
    fn box_primitive(&mut self, a: [f64; 3], b: [f64; 3], mat: MaterialId) -> Result<NodeId>; // box_node.rs:15-74
    fn disc(&mut self, center: [f64; 3], radius: f64, normal: [f64; 3], mat: MaterialId) -> Result<NodeId>; // disc.rs:24-54
    fn cone_frustum(&mut self, base: [f64; 3], height: f64, top_radius: f64, bottom_radius: f64, mat: MaterialId) -> Result<NodeId>; // cone.rs:19-75
    fn group(&mut self, children: &[NodeId]) -> Result<NodeId>; // group.rs:22-34
    /// The node list `BoundingVolumeHierarchy::new` was given (bvh.rs:16-46), in its order: the library repeats the build.
    fn bvh(&mut self, members: &[NodeId]) -> Result<NodeId>;
    fn translate(&mut self, child: NodeId, offset: [f64; 3]) -> Result<NodeId>; // translate.rs:15-22
    fn rotate(&mut self, child: NodeId, axis: [f64; 3], degrees: f64) -> Result<NodeId>; // rotate.rs:19-79
    fn scale(&mut self, child: NodeId, s: [f64; 3]) -> Result<NodeId>; // scale.rs:18-60
    fn constant_medium(&mut self, boundary: NodeId, density: f64, tex: TextureId) -> Result<NodeId>; // constant_medium.rs:19-29
    /// Shared nodes (`Arc<dyn Node>` reachable along several ways) are exported once: an impl asks here first, with the
    /// address of its `Arc`'s payload as key, and records the id it got from its own constructor call.
    fn seen(&self, key: usize) -> Option<NodeId>;
    fn remember(&mut self, key: usize, id: NodeId);
}

pub struct HipScene {
    raw: *mut sys::chip_scene,
    memo: HashMap<usize, NodeId>,
}

fn last_error() -> HipError {
    // SAFETY: chip_last_error returns a NUL-terminated string owned by the library (thread-local)
    HipError(unsafe { CStr::from_ptr(sys::chip_last_error()) }.to_string_lossy().into_owned())
}
fn check(rc: i32) -> Result<i32> {
    if rc < 0 { Err(last_error()) } else { Ok(rc) }
}

/// The fields of `CameraBuilder` (camera.rs:34-89), which are public.
pub use sys::chip_camera_desc as CameraDesc;
pub use sys::chip_stats as Stats;

impl HipScene {
    pub fn new() -> Result<Self> {
        let raw = unsafe { sys::chip_scene_create() };
        if raw.is_null() { return Err(HipError("chip_scene_create failed".into())); }
        Ok(HipScene { raw, memo: HashMap::new() })
    }
    pub fn raw(&self) -> *mut sys::chip_scene { self.raw }
    /// `SceneData` (lib.rs:38-43): world, lights, camera; then flatten and upload to the current HIP device.
    pub fn finish(&mut self, world: NodeId, lights: Option<NodeId>, camera: &CameraDesc) -> Result<()> {
        unsafe {
            check(sys::chip_scene_set_world(self.raw, world))?;
            check(sys::chip_scene_set_lights(self.raw, lights.unwrap_or(-1)))?;
            check(sys::chip_scene_set_camera(self.raw, camera))?;
            check(sys::chip_commit(self.raw))?;
        }
        Ok(())
    }
    pub fn image_size(&self) -> Result<(u32, u32)> {
        let (mut w, mut h) = (0u32, 0u32);
        check(unsafe { sys::chip_image_size(self.raw, &mut w, &mut h) })?;
        Ok((w, h))
    }
    fn opts(seed: u64, rect: Option<(u32, u32, u32, u32)>) -> sys::chip_render_opts {
        let (xmin, xmax, ymin, ymax) = rect.unwrap_or((0, 0, 0, 0));
        sys::chip_render_opts { seed, stratum_offset: 0, stratum_stride: 1, xmin, xmax, ymin, ymax, strata_per_batch: 0, flags: 0 }
    }
    /// Replaces the thread pool over 10 x 10 blocks of `crates/cli/src/main.rs:119-186`: the whole frame as the bytes
    /// `color_to_image_rgb` would have produced (main.rs:193-198).
    pub fn render_frame_rgb8(&mut self, seed: u64) -> Result<(Vec<u8>, Stats)> {
        let (w, h) = self.image_size()?;
        let mut rgb8 = vec![0u8; (w as usize) * (h as usize) * 3];
        let mut stats = Stats::default();
        check(unsafe { sys::chip_render_frame(self.raw, &Self::opts(seed, None), std::ptr::null_mut(), rgb8.as_mut_ptr(), &mut stats) })?;
        Ok((rgb8, stats))
    }
    /// `crates/wasm/src/lib.rs:215-242`'s `render(xmin, xmax, ymin, ymax)`: the rectangle's pixels, quantised with
    /// 255.0 as that caller does (lib.rs:269-276).
    pub fn render_rect_rgb8(&mut self, seed: u64, xmin: u32, xmax: u32, ymin: u32, ymax: u32) -> Result<Vec<u8>> {
        let mut o = Self::opts(seed, Some((xmin, xmax, ymin, ymax)));
        o.flags = sys::CHIP_RENDER_QUANTISE_255;
        let mut rgb8 = vec![0u8; ((xmax - xmin) as usize) * ((ymax - ymin) as usize) * 3];
        check(unsafe { sys::chip_render_frame(self.raw, &o, std::ptr::null_mut(), rgb8.as_mut_ptr(), std::ptr::null_mut()) })?;
        Ok(rgb8)
    }
    /// The frame tile by tile, refined over `passes` groups of samples: what the progress bar of crates/cli
    /// (main.rs:110-117,179) and the web worker pool's per-result repaint consume.
    pub fn render_frame_blocks<F: FnMut(u32, u32, u32, u32, &[u8], u32, u32)>(&mut self, seed: u64, block_size: u32, passes: u32, mut on_block: F) -> Result<Stats> {
        unsafe extern "C" fn thunk<F: FnMut(u32, u32, u32, u32, &[u8], u32, u32)>(user: *mut c_void, xmin: u32, xmax: u32, ymin: u32, ymax: u32, rgb8: *const u8, done: u32, total: u32) {
            let f = &mut *(user as *mut F);
            let n = ((xmax - xmin) as usize) * ((ymax - ymin) as usize) * 3;
            f(xmin, xmax, ymin, ymax, std::slice::from_raw_parts(rgb8, n), done, total);
        }
        let mut stats = Stats::default();
        check(unsafe {
            sys::chip_render_frame_blocks(self.raw, &Self::opts(seed, None), block_size, passes, thunk::<F>, &mut on_block as *mut F as *mut c_void, &mut stats)
        })?;
        Ok(stats)
    }
    /// All GPUs of the machine from this one process (after `finish`): strata dealt over the devices, sums gathered on the first.
    pub fn use_devices(&mut self, devices: &[i32]) -> Result<()> {
        check(unsafe { sys::chip_commit_devices(self.raw, devices.as_ptr(), devices.len() as i32) }).map(|_| ())
    }
    pub fn render_frame_multi_rgb8(&mut self, seed: u64) -> Result<(Vec<u8>, Stats)> {
        let (w, h) = self.image_size()?;
        let mut rgb8 = vec![0u8; (w as usize) * (h as usize) * 3];
        let mut stats = Stats::default();
        check(unsafe { sys::chip_render_frame_multi(self.raw, &Self::opts(seed, None), std::ptr::null_mut(), rgb8.as_mut_ptr(), &mut stats) })?;
        Ok((rgb8, stats))
    }
    /// `img.save("../../target/out.png")` (main.rs:188).
    pub fn save_png(path: &str, width: u32, height: u32, rgb8: &[u8]) -> Result<()> {
        assert_eq!(rgb8.len(), (width as usize) * (height as usize) * 3);
        let c = CString::new(path).map_err(|e| HipError(e.to_string()))?;
        check(unsafe { sys::chip_write_png(c.as_ptr(), width, height, rgb8.as_ptr()) }).map(|_| ())
    }
}

impl Drop for HipScene {
    fn drop(&mut self) {
        unsafe { sys::chip_scene_destroy(self.raw) }
    }
}

impl SceneExporter for HipScene {
    fn tex_solid(&mut self, c: [f64; 3]) -> Result<TextureId> { check(unsafe { sys::chip_tex_solid(self.raw, c[0], c[1], c[2]) }) }
    fn tex_checker(&mut self, scale: f64, even: TextureId, odd: TextureId) -> Result<TextureId> { check(unsafe { sys::chip_tex_checker(self.raw, scale, even, odd) }) }
    fn tex_image(&mut self, width: u32, height: u32, rgb8: &[u8]) -> Result<TextureId> {
        assert_eq!(rgb8.len(), (width as usize) * (height as usize) * 3);
        check(unsafe { sys::chip_tex_image(self.raw, width, height, rgb8.as_ptr()) })
    }
    fn tex_perlin_noise(&mut self, scale: f64, rand_vec: &[[f64; 3]; 256], perm: [&[i32; 256]; 3]) -> Result<TextureId> {
        check(unsafe { sys::chip_tex_perlin_noise(self.raw, scale, rand_vec.as_ptr() as *const f64, perm[0].as_ptr(), perm[1].as_ptr(), perm[2].as_ptr()) })
    }
    fn tex_perlin_turbulence(&mut self, scale: f64, depth: u32, rand_vec: &[[f64; 3]; 256], perm: [&[i32; 256]; 3]) -> Result<TextureId> {
        check(unsafe { sys::chip_tex_perlin_turbulence(self.raw, scale, depth, rand_vec.as_ptr() as *const f64, perm[0].as_ptr(), perm[1].as_ptr(), perm[2].as_ptr()) })
    }
    fn lambertian(&mut self, tex: TextureId) -> Result<MaterialId> { check(unsafe { sys::chip_lambertian_tex(self.raw, tex) }) }
    fn metal(&mut self, a: [f64; 3], fuzz: f64) -> Result<MaterialId> { check(unsafe { sys::chip_metal(self.raw, a[0], a[1], a[2], fuzz) }) }
    fn dielectric(&mut self, ri: f64) -> Result<MaterialId> { check(unsafe { sys::chip_dielectric(self.raw, ri) }) }
    fn diffuse_light(&mut self, tex: TextureId) -> Result<MaterialId> { check(unsafe { sys::chip_diffuse_light_tex(self.raw, tex) }) }
    fn empty_material(&mut self) -> Result<MaterialId> { check(unsafe { sys::chip_empty_material(self.raw) }) }
    fn sphere(&mut self, center: [f64; 3], radius: f64, direction: Option<[f64; 3]>, mat: MaterialId) -> Result<NodeId> {
        let id = check(unsafe { sys::chip_sphere(self.raw, center.as_ptr(), radius, mat) })?;
        if let Some(d) = direction {
            check(unsafe { sys::chip_sphere_set_direction(self.raw, id, d.as_ptr()) })?; // before the sphere is referenced, as sphere.rs:31-34 mutates before Arc::new
        }
        Ok(id)
    }
    fn quad(&mut self, q: [f64; 3], u: [f64; 3], v: [f64; 3], mat: MaterialId) -> Result<NodeId> { check(unsafe { sys::chip_quad(self.raw, q.as_ptr(), u.as_ptr(), v.as_ptr(), mat) }) }
    fn box_primitive(&mut self, a: [f64; 3], b: [f64; 3], mat: MaterialId) -> Result<NodeId> { check(unsafe { sys::chip_box(self.raw, a.as_ptr(), b.as_ptr(), mat) }) }
    fn disc(&mut self, c: [f64; 3], radius: f64, n: [f64; 3], mat: MaterialId) -> Result<NodeId> { check(unsafe { sys::chip_disc(self.raw, c.as_ptr(), radius, n.as_ptr(), mat) }) }
    fn cone_frustum(&mut self, base: [f64; 3], height: f64, top: f64, bottom: f64, mat: MaterialId) -> Result<NodeId> {
        check(unsafe { sys::chip_cone_frustum(self.raw, base.as_ptr(), height, top, bottom, mat) })
    }
    fn group(&mut self, children: &[NodeId]) -> Result<NodeId> { check(unsafe { sys::chip_group(self.raw, children.as_ptr(), children.len() as i32) }) }
    fn bvh(&mut self, members: &[NodeId]) -> Result<NodeId> { check(unsafe { sys::chip_bvh(self.raw, members.as_ptr(), members.len() as i32) }) }
    fn translate(&mut self, child: NodeId, off: [f64; 3]) -> Result<NodeId> { check(unsafe { sys::chip_translate(self.raw, child, off.as_ptr()) }) }
    fn rotate(&mut self, child: NodeId, axis: [f64; 3], degrees: f64) -> Result<NodeId> { check(unsafe { sys::chip_rotate(self.raw, child, axis.as_ptr(), degrees) }) }
    fn scale(&mut self, child: NodeId, s: [f64; 3]) -> Result<NodeId> { check(unsafe { sys::chip_scale(self.raw, child, s[0], s[1], s[2]) }) }
    fn constant_medium(&mut self, boundary: NodeId, density: f64, tex: TextureId) -> Result<NodeId> { check(unsafe { sys::chip_constant_medium_tex(self.raw, boundary, density, tex) }) }
    fn seen(&self, key: usize) -> Option<NodeId> { self.memo.get(&key).copied() }
    fn remember(&mut self, key: usize, id: NodeId) { self.memo.insert(key, id); }
}

// What an `export` impl inside caustic_core looks like (sphere.rs), for reference:
//
//     fn export(self: &Arc<Self>, b: &mut dyn SceneExporter) -> Result<NodeId> {
//         let key = Arc::as_ptr(self) as *const () as usize;
//         if let Some(id) = b.seen(key) { return Ok(id); }
//         let mat = self.material.export(b)?;
//         let moving = self.center.direction != Vector3::ZERO;
//         let id = b.sphere(self.center.origin.into(), self.radius, moving.then(|| self.center.direction.into()), mat)?;
//         b.remember(key, id);
//         Ok(id)
//     }
//
// and the patched frame loop of crates/cli/src/main.rs:119-186:
//
//     let mut scene = HipScene::new()?;
//     let world = scene_data.world.export(&mut scene)?;
//     let lights = scene_data.lights.as_ref().map(|l| l.export(&mut scene)).transpose()?;
//     scene.finish(world, lights, &scene_data.camera.builder_fields())?;
//     let pb = ProgressBar::new(passes as u64);
//     let (w, h) = scene.image_size()?;
//     let mut frame = vec![0u8; (w * h * 3) as usize];
//     scene.render_frame_blocks(seed, BLOCK_SIZE, passes, |xmin, xmax, ymin, ymax, rgb8, done, total| {
//         for (row, y) in (ymin..ymax).enumerate() {
//             let n = ((xmax - xmin) * 3) as usize;
//             frame[((y * w + xmin) * 3) as usize..][..n].copy_from_slice(&rgb8[row * n..][..n]);
//         }
//         if xmin == 0 && ymin == 0 { pb.set_position((done * passes / total) as u64); }
//     })?;
//     HipScene::save_png("../../target/out.png", w, h, &frame)?;
