/* caustic_hip.h — C ABI of libcaustic_hip.so, the MI355X (gfx950) render path for
 * the scene vocabulary of the reference's crates/core.
 *
 * The reference has no FFI layer: its seam is the Rust call
 *   Camera::render(&self, ctx, x, y, world: &dyn Node, lights) -> Color   (crates/core/src/camera.rs:324-331)
 * over SceneData{camera, world, lights} (crates/core/src/lib.rs:38-43), driven per pixel by
 * crates/cli/src/main.rs:136-146 and per rectangle by crates/wasm/src/lib.rs:215-242.
 * A GPU backend replaces the block/frame loop, so this ABI is "build the same scene
 * graph through the same constructors, then render a rectangle / frame".
 *
 * Conventions: every input is copied at call time (the caller keeps ownership);
 * outputs go to caller-allocated buffers; geometry crosses the ABI as double
 * because the reference is f64 end to end.  Functions returning int give
 * >= 0 (an id, or 0 for success) or < 0 on error, with the message available from
 * chip_last_error() on the calling thread.  Ids are creation-order integers per
 * category (textures, materials, nodes).  A scene may be used from one host
 * thread at a time; distinct scenes are independent.  The random stream is an
 * internal counter-based Philox4x32-10 keyed on (seed, pixel, stratum, bounce),
 * replacing RenderContext.random (crates/core/src/lib.rs:34-36).
 */
#ifndef CAUSTIC_HIP_H
#define CAUSTIC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct chip_scene chip_scene;

/* CameraBuilder's public fields, crates/core/src/camera.rs:34-89 (defaults :106-120). */
typedef struct chip_camera_desc {
  double vertical_fov;
  double aspect_ratio;
  uint32_t image_width;
  uint32_t reserved0;
  double look_from[3];
  double look_at[3];
  double up[3];
  double defocus_angle;
  double focus_distance;
  uint32_t samples_per_pixel;
  uint32_t max_depth;
  double background[3];
} chip_camera_desc;

/* What one render call covers.  The strata of camera.rs:335-341 are numbered
 * s = s_y * sqrt_spp + s_x; this call renders those with s % stratum_stride ==
 * stratum_offset (offset 0 / stride 1 = all of them), which is how samples are
 * split over GPUs.  The rectangle is [xmin,xmax) x [ymin,ymax) as in
 * crates/wasm/src/lib.rs:215; xmax == ymax == 0 means the whole image. */
typedef struct chip_render_opts {
  uint64_t seed;
  uint32_t stratum_offset;
  uint32_t stratum_stride;
  uint32_t xmin, xmax, ymin, ymax;
  uint32_t strata_per_batch; /* 0 = choose from the path-pool capacity */
  uint32_t flags;            /* CHIP_RENDER_* bits */
} chip_render_opts;

/* HitRecord of crates/core/src/object/mod.rs:31-39 for chip_trace_rays. */
typedef struct chip_hit {
  double t;
  double pt[3];
  double normal[3];
  double u, v;
  int32_t hit;           /* 0 = miss */
  int32_t leaf_id;       /* node id of the primitive / box / medium that produced the record */
  int32_t front_face;
  int32_t material_kind; /* CHIP_MAT_* */
} chip_hit;

typedef struct chip_stats {
  uint64_t paths;      /* camera samples traced */
  uint64_t segments;   /* world.hit calls (ray segments) */
  uint64_t bvh_visits; /* BVH nodes whose box was tested */
  uint64_t prim_tests; /* sphere / quad intersection tests */
  double ms_total;     /* device time of the whole call (HIP events) */
  double ms_trace;     /* sum of the traversal kernel's launch durations (HIP events; only with CHIP_RENDER_TIME_KERNELS) */
  double ms_shade;     /* same for the shade kernels */
  double ms_other;
  uint64_t trace_launches; /* traversal kernel launches in this call */
  uint64_t batches;        /* path-pool batches in this call */
} chip_stats;

enum {
  CHIP_RENDER_TIME_KERNELS = 1, /* bracket traversal / shade launches with HIP events and report their durations */
  /* Shade kernels keep the reference's f64 operation order (divide, sqrt, sin, cos) for scatter directions and
   * PDFs, so a render consumes the Philox stream exactly as the f64 CPU restatement does.  The default shade
   * kernels use f32-seeded Newton reciprocals / f32 trigonometry there (same estimator, last-digit differences
   * in sampled directions).  Traversal and hit records are f64 / reference order in both modes. */
  CHIP_RENDER_EXACT_SHADING = 2,
  /* Trace every ray by replaying the reference's recursion literally (bounding_volume_hierarchy.rs:58-76: own
   * box, left, then right over the shrunken interval, f64 slab tests) instead of the default ordered traversal
   * (nearer child first, conservative f32 culling, exact ties settled by the reference's visiting order).  The two
   * agree except where a ray meets a bounding box exactly at the distance of a hit found earlier, which the
   * reference's `max <= min` slab rejection (axis_aligned_bounding_box.rs:207-225) turns into a miss.  Rays with a
   * zero direction component always take the literal path (their slab arithmetic forms NaNs). */
  CHIP_RENDER_REFERENCE_ORDER = 4,
  /* 8-bit output as the wasm caller quantises it, (c * 255.0) as u8 (crates/wasm/src/lib.rs:269-276), instead of
   * the CLI's c * 255.999 (crates/cli/src/main.rs:193-198). */
  CHIP_RENDER_QUANTISE_255 = 8
};
enum { CHIP_MAT_LAMBERTIAN = 0, CHIP_MAT_METAL = 1, CHIP_MAT_DIELECTRIC = 2, CHIP_MAT_DIFFUSE_LIGHT = 3, CHIP_MAT_ISOTROPIC = 4, CHIP_MAT_EMPTY = 5 };
/* node kinds reported by chip_dump_tree */
enum { CHIP_NODE_BVH = 0, CHIP_NODE_SPHERE = 1, CHIP_NODE_QUAD = 2, CHIP_NODE_GROUP = 3, CHIP_NODE_TRANSLATE = 4, CHIP_NODE_ROTATE = 5, CHIP_NODE_MEDIUM = 6, CHIP_NODE_BOX = 7, CHIP_NODE_DISC = 8,
       CHIP_NODE_CONE = 9, CHIP_NODE_CONE_WALL = 10, CHIP_NODE_SCALE = 11 };

const char* chip_last_error(void);
/* Number of HIP devices visible; < 0 when the HIP runtime cannot be initialised. */
int chip_device_count(void);

chip_scene* chip_scene_create(void);
void chip_scene_destroy(chip_scene* s);

/* Textures: crates/core/src/texture/ (all files) */
int chip_tex_solid(chip_scene* s, double r, double g, double b);             /* solid_color.rs:14-17 */
int chip_tex_checker(chip_scene* s, double scale, int even, int odd);         /* checker_texture.rs:13-35 */
int chip_tex_image(chip_scene* s, uint32_t w, uint32_t h, const uint8_t* rgb); /* image_texture.rs:16-29, image.rs:58-66 */
/* Perlin tables (utils/perlin.rs:27-36) are passed in, so the caller's RNG decides them as in
 * Perlin::new(random): rand_vec = 256 x 3 doubles, perm_* = 256 int32 each. */
int chip_tex_perlin_noise(chip_scene* s, double scale, const double* rand_vec, const int32_t* perm_x, const int32_t* perm_y,
                          const int32_t* perm_z); /* perlin_noise.rs:18-22 */
int chip_tex_perlin_turbulence(chip_scene* s, double scale, uint32_t depth, const double* rand_vec, const int32_t* perm_x,
                               const int32_t* perm_y, const int32_t* perm_z); /* perlin_turbulence.rs:20-27 */

/* Materials: crates/core/src/material/ (all files) */
int chip_lambertian_color(chip_scene* s, double r, double g, double b); /* lambertian.rs:21-25 */
int chip_lambertian_tex(chip_scene* s, int tex);                         /* lambertian.rs:17-19 */
int chip_metal(chip_scene* s, double r, double g, double b, double fuzz); /* metal.rs:14-16 */
int chip_dielectric(chip_scene* s, double refraction_index);             /* dielectric.rs:17-19 */
int chip_diffuse_light_color(chip_scene* s, double r, double g, double b); /* diffuse_light.rs:20-24 */
int chip_diffuse_light_tex(chip_scene* s, int tex);                        /* diffuse_light.rs:16-18 */
int chip_empty_material(chip_scene* s);                                    /* empty.rs:7-9 */

/* Nodes: crates/core/src/object/ (all files) */
int chip_sphere(chip_scene* s, const double center[3], double radius, int mat);            /* sphere.rs:21-29 */
/* Only before the sphere is referenced by another node (Rust mutates before Arc::new). */
int chip_sphere_set_direction(chip_scene* s, int node, const double direction[3]);         /* sphere.rs:31-34 */
int chip_quad(chip_scene* s, const double q[3], const double u[3], const double v[3], int mat); /* quad.rs:59-76 */
int chip_box(chip_scene* s, const double a[3], const double b[3], int mat);                /* box_node.rs:15-74 */
int chip_group(chip_scene* s, const int32_t* nodes, int n);                                /* group.rs:22-34 */
int chip_bvh(chip_scene* s, const int32_t* nodes, int n);                                  /* bounding_volume_hierarchy.rs:16-46 */
int chip_translate(chip_scene* s, int node, const double offset[3]);                       /* translate.rs:15-22 */
int chip_rotate(chip_scene* s, int node, const double axis[3], double angle_degrees);      /* rotate.rs:19-79 */
int chip_scale(chip_scene* s, int node, double sx, double sy, double sz);                   /* scale.rs:18-60 */
int chip_disc(chip_scene* s, const double center[3], double radius, const double normal[3], int mat); /* disc.rs:24-54 */
/* ConeFrustum::new: top disc and bottom disc (each only when its radius > 1e-4) and the side wall, as a Group. */
int chip_cone_frustum(chip_scene* s, const double base[3], double height, double top_radius, double bottom_radius, int mat); /* cone.rs:19-75 */
/* The boundary may be any node (constant_medium.rs:19-29), media included one level deep: a medium inside a boundary
 * must itself be bounded by a sphere or a box and must not hold media in its own boundary (chip_commit reports the
 * shapes beyond that as errors). */
int chip_constant_medium_color(chip_scene* s, int boundary, double density, double r, double g, double b); /* constant_medium.rs:31-37 */
int chip_constant_medium_tex(chip_scene* s, int boundary, double density, int tex);        /* constant_medium.rs:19-29 */

/* Many nodes in one call, for scenes of 10^5..10^6 objects such as the ones the OpenSCAD front end builds in a loop
 * (crates/openscad/src/interpreter/modules.rs:76-105 per child; one million Translate(Group[Sphere]) for a 100^3 `for`
 * nest): record i runs the constructor named by `kind` with the same arguments, in the same order, as the per-node
 * functions above, so the scene is the one a loop over those functions would have built.
 *   v[] per kind: SPHERE center[3], radius, (direction[3] with CHIP_FLAT_MOVING) | QUAD q[3], u[3], v[3] |
 *   BOX a[3], b[3] | TRANSLATE offset[3] | ROTATE axis[3], degrees | SCALE sx, sy, sz | DISC center[3], radius, normal[3] |
 *   CONE base[3], height, top_radius, bottom_radius | MEDIUM density, (r, g, b unless CHIP_FLAT_MEDIUM_TEX: then `mat`
 *   is a texture id).
 * `first` is the child of an instance or medium, or the first entry in `members` of a GROUP / BVH with `count` entries.
 * A child or member value c >= 0 is record c of this call (c < i); c < 0 is the existing scene node ~c.
 * node_ids (nullable) receives the node id of every record.  Returns the number of records, < 0 on error
 * (chip_last_error names the record; nodes made before it stay in the scene). */
enum { CHIP_FLAT_MOVING = 1, CHIP_FLAT_MEDIUM_TEX = 2 };
typedef struct chip_flat_node {
  int32_t kind; /* CHIP_NODE_SPHERE / QUAD / BOX / GROUP / BVH / TRANSLATE / ROTATE / SCALE / DISC / CONE / MEDIUM */
  int32_t mat;
  int32_t first;
  int32_t count;
  uint32_t flags; /* CHIP_FLAT_* */
  uint32_t reserved;
  double v[9];
} chip_flat_node;
typedef struct chip_flat_scene {
  const chip_flat_node* nodes;
  size_t n_nodes;
  const int32_t* members;
  size_t n_members;
} chip_flat_scene;
int64_t chip_scene_from_flat(chip_scene* s, const chip_flat_scene* flat, int32_t* node_ids);

/* SceneData, crates/core/src/lib.rs:38-43; lights < 0 = None. */
int chip_scene_set_world(chip_scene* s, int node);
int chip_scene_set_lights(chip_scene* s, int node);
int chip_scene_set_camera(chip_scene* s, const chip_camera_desc* desc); /* CameraBuilder::build, camera.rs:126-182 */

/* Flatten the scene graph into device buffers on the current HIP device. */
int chip_commit(chip_scene* s);
int chip_image_size(chip_scene* s, uint32_t* width, uint32_t* height); /* camera.rs:397-404 */
int chip_sqrt_spp(chip_scene* s);                                       /* camera.rs:131 */
/* How the committed scene will be rendered, as CHIP_SCENE_* bits (no counterpart in the reference, which has one way to
 * render): some scene contents take the library off its default path, at a cost in speed that a caller may want to know
 * about instead of finding it in the frame time.  < 0 when the scene is not committed. */
enum {
  /* Every render shades with the reference's f64 operation order, as if CHIP_RENDER_EXACT_SHADING were set: the scene
   * holds a Disc whose normal is not of unit length (disc.rs:24-54 keeps the normal as given, and lambertian.rs then
   * weighs a bounce by its length, which the fast arithmetic's unit-normal shortcuts do not reproduce). */
  CHIP_SCENE_EXACT_SHADING_FORCED = 1,
  /* Every ray is traced in the reference's own visiting order, as if CHIP_RENDER_REFERENCE_ORDER were set: the scene holds
   * a primitive that reaches outside its own bounding box (a tilted Disc, disc.rs:25-35), which only that order finds. */
  CHIP_SCENE_REFERENCE_ORDER_FORCED = 2,
  /* Bounce 0 runs in one kernel (traversal and shading fused): small scenes of bare spheres with solid colours. */
  CHIP_SCENE_FUSED_FIRST_BOUNCE = 4
};
int chip_scene_traits(chip_scene* s);
/* The derived Camera fields of camera.rs:166-181 as 24 doubles: center, pixel00_loc, pixel_delta_u,
 * pixel_delta_v, defocus_disk_u, defocus_disk_v, then pixel_samples_scale, reciprocal_sqrt_spp,
 * defocus_angle, sqrt_spp, image_width, image_height. */
int chip_camera_derived(chip_scene* s, double* out24);

/* Replaces the caller loop of crates/cli/src/main.rs:119-186 (and one call of crates/wasm/src/lib.rs:215-242
 * per rectangle).  out_sum_rgb (host, rect_w*rect_h*3 floats, nullable) receives the un-normalised
 * linear radiance sum over the strata rendered by this call (NaN preserved, camera.rs:339);
 * out_rgb8 (host, nullable) receives camera.rs:343-344 + main.rs:193-198 applied to that sum with
 * the full pixel_samples_scale, i.e. it is the final image only when all strata were rendered. */
int chip_render_frame(chip_scene* s, const chip_render_opts* opts, float* out_sum_rgb, uint8_t* out_rgb8, chip_stats* stats);
/* Same, but the sum is ADDED into a device buffer the caller owns (float, rect_w*rect_h*3, on the
 * current device) and nothing is copied to the host: the form used when an external collective
 * reduces per-GPU sums.  Work is queued on `stream` (a hipStream_t, NULL = default stream) and
 * the call returns after that work completed. */
int chip_render_accumulate_device(chip_scene* s, const chip_render_opts* opts, float* d_sum_rgb, void* stream, chip_stats* stats);
/* camera.rs:343-344 (+ main.rs:193-198 when out_rgb8 is set) on a device sum buffer; out_gamma /
 * out_rgb8 are device pointers, either nullable. */
int chip_resolve_device(chip_scene* s, const float* d_sum_rgb, size_t npixels, float* d_out_gamma, uint8_t* d_out_rgb8, void* stream);
/* The same with the 8-bit scale chosen by `flags` (CHIP_RENDER_QUANTISE_255 or 0). */
int chip_resolve_device_flags(chip_scene* s, const float* d_sum_rgb, size_t npixels, float* d_out_gamma, uint8_t* d_out_rgb8, uint32_t flags,
                              void* stream);

/* The frame delivered block by block, progressively: what the reference's front ends consume.  crates/cli cuts the
 * frame into BLOCK_SIZE x BLOCK_SIZE tiles, renders them on worker threads and paints each finished tile
 * (crates/cli/src/main.rs:27,83-106,165-182); the web front end's worker pool does the same with rectangles and repaints
 * per result (webapp/frontend/src/RenderWorkerPool.ts:68-77 over crates/wasm/src/lib.rs:215-242).  A GPU renders all
 * pixels at once, so the progression here is over samples: the strata this call covers are rendered in `passes`
 * groups, and after each group every block_size x block_size tile of the rectangle (clipped at its edges) is handed to
 * `on_block` with camera.rs:343-344 applied to the samples taken so far, quantised like chip_render_frame's out_rgb8
 * (rgb8 = (xmax - xmin) * (ymax - ymin) * 3 bytes, valid during the call only).  Tiles arrive in the order the
 * reference's workers pop them, last pushed first (main.rs:132).  After the last pass of a call that covers every
 * stratum the tiles hold chip_render_frame's bytes.  on_block runs on the calling thread, between GPU passes. */
typedef void (*chip_block_fn)(void* user, uint32_t xmin, uint32_t xmax, uint32_t ymin, uint32_t ymax, const uint8_t* rgb8,
                              uint32_t strata_done, uint32_t strata_total);
int chip_render_frame_blocks(chip_scene* s, const chip_render_opts* opts, uint32_t block_size, uint32_t passes, chip_block_fn on_block, void* user,
                             chip_stats* stats);
/* 8-bit RGB PNG file of a frame (crates/cli/src/main.rs:188 `img.save`); the zlib stream uses stored blocks. */
int chip_write_png(const char* path, uint32_t width, uint32_t height, const uint8_t* rgb8);

/* chip_render_accumulate_device for callers that cannot allocate (tight serving loops, harnesses that forbid
 * hipMalloc, code that must not disturb the allocator between graph launches).  This entry point has no counterpart in
 * the reference.  It allocates nothing, but it is not itself capturable into a HIP graph: it reads live counts back to
 * size later launches and synchronises `stream` before it returns.  It calls no hipMalloc,
 * hipFree, stream / event creation or other allocating routine: its path pool (32768 path slots with their counters,
 * hit records, literal queue and traversal spill slabs, about 20 MB) is a static device array that every GPU the
 * library runs on holds until the process ends, and the streams, events and pinned words it needs are made by
 * chip_commit.  Limits, rejected with an error: the rectangle may hold at most chip_noalloc_max_pixels() = 32768 pixels
 * (any number of strata; they are rendered batch after batch on `stream`), max_depth at most 254.
 * CHIP_RENDER_TIME_KERNELS is ignored.  Same sums, bit for bit, as chip_render_accumulate_device.  All calls on one
 * GPU share that GPU's array, from any scene: calls that overlap on one GPU race. */
int chip_render_accumulate_noalloc(chip_scene* s, const chip_render_opts* opts, float* d_sum_rgb, void* stream, chip_stats* stats);
uint32_t chip_noalloc_max_pixels(void);

/* One frame over several GPUs of the machine, from one process: the shape of the reference's caller, which deals the
 * work of a frame to `num_cpus` threads and gathers their results (crates/cli/src/main.rs:119-186).
 * chip_commit_devices copies the committed scene to every device in `devices` (n entries, 1..64; an id may occur more
 * than once, each entry gets a replica of its own, which is how a one-GPU machine runs this path).  It must be called
 * again after every chip_commit.  chip_render_frame_multi then renders like chip_render_frame: replica g takes every
 * n-th of the strata the call covers (stratum_offset + g * stride, stride * n) on a host thread and stream of its own;
 * as soon as a replica has finished, its f32 radiance sums travel to a staging slot of its own on the first device
 * (peer copy over the GPU interconnect, while slower replicas still render), and one kernel adds the slots in replica
 * order, so the image does not depend on timing; the first device resolves it.  The sums equal the one-GPU frame's up
 * to the order of f32 additions.  A call that covers fewer strata than there are replicas is dealt as bands of rows of
 * the rectangle instead (the reference deals blocks of pixels, main.rs:83-106): replica g renders every stratum of
 * band g and the bands are copied side by side, which gives the one-GPU frame bit for bit.
 * stats: counts are summed over the replicas, ms_total is the slowest. */
int chip_commit_devices(chip_scene* s, const int* devices, int n);
int chip_render_frame_multi(chip_scene* s, const chip_render_opts* opts, float* out_sum_rgb, uint8_t* out_rgb8, chip_stats* stats);

/* Test hooks.  chip_trace_rays: closest hit of world over [0.001, inf) (camera.rs:264) for n host
 * rays (origin, direction: n x 3 doubles; time: n doubles or NULL = 0; any time is accepted, also outside
 * the [0, 1) the camera draws, ray.rs / camera.rs:371).  Medium free-flight draws use the Philox stream keyed
 * (seed, ray index, 0, bounce 0). */
int chip_trace_rays(chip_scene* s, uint64_t seed, size_t n, const double* origin, const double* direction, const double* time,
                    chip_hit* out);
/* Camera rays of camera.rs:359-374 for n (x, y, s_x, s_y) tuples; rays = n x 7 doubles. */
int chip_camera_rays(chip_scene* s, uint64_t seed, size_t n, const uint32_t* xyss, double* rays);
/* Preorder dump of the host scene tree (BVH: left then right): node kind, creation id (-1 for
 * nodes built internally), bbox as xmin,xmax,ymin,ymax,zmin,zmax.  Returns the record count,
 * which may exceed cap (then only cap records were written). */
int64_t chip_dump_tree(chip_scene* s, int32_t* kinds, int32_t* ids, double* boxes, size_t cap);
/* Consistency check of the library's own acceleration structure (no counterpart in the reference, needs no GPU):
 * flattens the scene and verifies that the culling trees the ordered traversal walks reach exactly the primitives,
 * media and instances that the reference's structure (bounding_volume_hierarchy.rs:16-46 hierarchies, Groups,
 * instances) reaches, each once, that nested culling boxes contain one another, and that unused child slots cannot be
 * hit.  Returns the number of primitives reached, < 0 with chip_last_error() set when a check fails. */
int64_t chip_check_culling_trees(chip_scene* s);

#ifdef __cplusplus
}
#endif
#endif /* CAUSTIC_HIP_H */
