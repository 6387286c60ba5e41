// TEST INFRASTRUCTURE ONLY — C entry points of the CPU oracle (see caustic_oracle.hpp).
// Loaded through ctypes by tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg.  The scene vocabulary mirrors the constructors of the
// reference's crates/core so the same scene-definition code can drive the oracle
// and the GPU library.  Handles are creation-order integers per category.
#include <atomic>
#include <chrono>
#include <mutex>
#include <string>
#include <thread>

#include "caustic_oracle.hpp"

using namespace orc;

struct OrcScene {
  std::vector<std::shared_ptr<Texture>> textures;
  std::vector<std::shared_ptr<Material>> materials;
  std::vector<NodeP> nodes;
  NodeP world, lights;
  std::unique_ptr<Camera> camera;
  CameraDesc cam_desc;
};

struct OrcCameraDesc {  // same field order as chip_camera_desc
  double vertical_fov, aspect_ratio;
  uint32_t image_width, pad0;
  double look_from[3], look_at[3], up[3];
  double defocus_angle, focus_distance;
  uint32_t samples_per_pixel, max_depth;
  double background[3];
};

struct OrcRenderOpts {  // same field order as chip_render_opts
  uint64_t seed;
  uint32_t stratum_offset, stratum_stride;
  uint32_t xmin, xmax, ymin, ymax;
  uint32_t strata_per_batch;  // ignored by the oracle
  uint32_t flags;             // bit0: use the forward-only (iterative) estimator; low 8 bits of (flags>>8): threads (0 = hw);
                              // bits 16-17: RenderContext::media_mode (0 keyed samples, 1 one stream draw per test, 2 negative control)
};

struct OrcHit {  // same layout as chip_hit
  double t, pt[3], normal[3], u, v;
  int32_t hit, leaf_id, front_face, material_kind;
};

struct OrcStats {  // same layout as chip_stats
  uint64_t paths, segments, bvh_visits, prim_tests;
  double ms_total, ms_trace, ms_shade, ms_other;
  uint64_t trace_launches, batches;
};

static thread_local std::string g_err;
static int fail(const char* msg) {
  g_err = msg;
  return -1;
}

static Vec3 v3(const double* p) { return Vec3(p[0], p[1], p[2]); }

extern "C" {

const char* orc_last_error(void) { return g_err.c_str(); }

OrcScene* orc_scene_create(void) { return new OrcScene(); }
void orc_scene_destroy(OrcScene* s) { delete s; }

#define CHECK_TEX(id) if ((id) < 0 || (size_t)(id) >= s->textures.size()) return fail("bad texture id")
#define CHECK_MAT(id) if ((id) < 0 || (size_t)(id) >= s->materials.size()) return fail("bad material id")
#define CHECK_NODE(id) if ((id) < 0 || (size_t)(id) >= s->nodes.size()) return fail("bad node id")

static int add_tex(OrcScene* s, std::shared_ptr<Texture> t) {
  s->textures.push_back(std::move(t));
  return (int)s->textures.size() - 1;
}
static int add_mat(OrcScene* s, std::shared_ptr<Material> m) {
  s->materials.push_back(std::move(m));
  return (int)s->materials.size() - 1;
}
static int add_node(OrcScene* s, NodeP n) {
  n->id = (int)s->nodes.size();
  s->nodes.push_back(std::move(n));
  return (int)s->nodes.size() - 1;
}

int orc_tex_solid(OrcScene* s, double r, double g, double b) { return add_tex(s, std::make_shared<SolidColor>(Color(r, g, b))); }
int orc_tex_checker(OrcScene* s, double scale, int even, int odd) {
  CHECK_TEX(even);
  CHECK_TEX(odd);
  return add_tex(s, std::make_shared<CheckerTexture>(scale, s->textures[even], s->textures[odd]));
}
int orc_tex_image(OrcScene* s, uint32_t w, uint32_t h, const uint8_t* rgb) { return add_tex(s, std::make_shared<ImageTexture>(w, h, rgb)); }
static void fill_perlin(Perlin& p, const double* rand_vec, const int32_t* px, const int32_t* py, const int32_t* pz) {
  for (int i = 0; i < 256; i++) {
    p.rand_vec[i] = Vec3(rand_vec[3 * i], rand_vec[3 * i + 1], rand_vec[3 * i + 2]);
    p.perm_x[i] = px[i];
    p.perm_y[i] = py[i];
    p.perm_z[i] = pz[i];
  }
}
int orc_tex_perlin_noise(OrcScene* s, double scale, const double* rand_vec, const int32_t* px, const int32_t* py, const int32_t* pz) {
  auto t = std::make_shared<PerlinNoiseTexture>();
  fill_perlin(t->noise, rand_vec, px, py, pz);
  t->scale = scale;
  return add_tex(s, t);
}
int orc_tex_perlin_turbulence(OrcScene* s, double scale, uint32_t depth, const double* rand_vec, const int32_t* px, const int32_t* py,
                              const int32_t* pz) {
  auto t = std::make_shared<PerlinTurbulenceTexture>();
  fill_perlin(t->noise, rand_vec, px, py, pz);
  t->scale = scale;
  t->depth = depth;
  return add_tex(s, t);
}

int orc_lambertian_color(OrcScene* s, double r, double g, double b) {
  return add_mat(s, std::make_shared<Lambertian>(std::make_shared<SolidColor>(Color(r, g, b))));
}
int orc_lambertian_tex(OrcScene* s, int tex) {
  CHECK_TEX(tex);
  return add_mat(s, std::make_shared<Lambertian>(s->textures[tex]));
}
int orc_metal(OrcScene* s, double r, double g, double b, double fuzz) { return add_mat(s, std::make_shared<Metal>(Color(r, g, b), fuzz)); }
int orc_dielectric(OrcScene* s, double ior) { return add_mat(s, std::make_shared<Dielectric>(ior)); }
int orc_diffuse_light_color(OrcScene* s, double r, double g, double b) {
  return add_mat(s, std::make_shared<DiffuseLight>(std::make_shared<SolidColor>(Color(r, g, b))));
}
int orc_diffuse_light_tex(OrcScene* s, int tex) {
  CHECK_TEX(tex);
  return add_mat(s, std::make_shared<DiffuseLight>(s->textures[tex]));
}
int orc_empty_material(OrcScene* s) { return add_mat(s, std::make_shared<EmptyMaterial>()); }

int orc_sphere(OrcScene* s, const double* c, double r, int mat) {
  CHECK_MAT(mat);
  return add_node(s, std::make_shared<Sphere>(v3(c), r, s->materials[mat]));
}
int orc_sphere_set_direction(OrcScene* s, int node, const double* d) {
  CHECK_NODE(node);
  auto* sp = dynamic_cast<Sphere*>(s->nodes[node].get());
  if (!sp) return fail("node is not a sphere");
  if (s->nodes[node].use_count() > 1) return fail("sphere already referenced");
  sp->set_direction(v3(d));
  return 0;
}
int orc_quad(OrcScene* s, const double* q, const double* u, const double* v, int mat) {
  CHECK_MAT(mat);
  return add_node(s, std::make_shared<Quad>(v3(q), v3(u), v3(v), s->materials[mat]));
}
int orc_box(OrcScene* s, const double* a, const double* b, int mat) {
  CHECK_MAT(mat);
  return add_node(s, std::make_shared<BoxPrimitive>(v3(a), v3(b), s->materials[mat]));
}
int orc_group(OrcScene* s, const int32_t* ids, int n) {
  auto g = std::make_shared<Group>();
  for (int i = 0; i < n; i++) {
    CHECK_NODE(ids[i]);
    g->push(s->nodes[ids[i]]);
  }
  return add_node(s, g);
}
int orc_bvh(OrcScene* s, const int32_t* ids, int n) {
  std::vector<NodeP> v;
  for (int i = 0; i < n; i++) {
    CHECK_NODE(ids[i]);
    v.push_back(s->nodes[ids[i]]);
  }
  return add_node(s, std::make_shared<BVH>(v));
}
int orc_translate(OrcScene* s, int node, const double* off) {
  CHECK_NODE(node);
  return add_node(s, std::make_shared<Translate>(s->nodes[node], v3(off)));
}
int orc_rotate(OrcScene* s, int node, const double* axis, double deg) {
  CHECK_NODE(node);
  return add_node(s, std::make_shared<Rotate>(s->nodes[node], v3(axis), deg));
}
int orc_scale(OrcScene* s, int node, double sx, double sy, double sz) {
  CHECK_NODE(node);
  return add_node(s, std::make_shared<Scale>(s->nodes[node], sx, sy, sz));
}
int orc_disc(OrcScene* s, const double* center, double radius, const double* normal, int mat) {
  CHECK_MAT(mat);
  return add_node(s, std::make_shared<Disc>(v3(center), radius, v3(normal), s->materials[mat]));
}
int orc_cone_frustum(OrcScene* s, const double* base, double height, double top_radius, double bottom_radius, int mat) {
  CHECK_MAT(mat);
  return add_node(s, std::make_shared<ConeFrustum>(v3(base), height, top_radius, bottom_radius, s->materials[mat]));
}
int orc_constant_medium_color(OrcScene* s, int boundary, double density, double r, double g, double b) {
  CHECK_NODE(boundary);
  return add_node(s, std::make_shared<ConstantMedium>(s->nodes[boundary], density, std::make_shared<SolidColor>(Color(r, g, b))));
}
int orc_constant_medium_tex(OrcScene* s, int boundary, double density, int tex) {
  CHECK_NODE(boundary);
  CHECK_TEX(tex);
  return add_node(s, std::make_shared<ConstantMedium>(s->nodes[boundary], density, s->textures[tex]));
}

// Many constructor calls in one (same record layout and meaning as chip_scene_from_flat of include/caustic_hip.h):
// record i runs the constructor named by `kind` through the functions above, in order, so the graph is the one a
// loop over them builds.  Lets the tests build 10^6-node scenes on the oracle without 10^6 calls from Python.
struct OrcFlatNode {
  int32_t kind, mat, first, count;
  uint32_t flags, reserved;
  double v[9];
};
struct OrcFlatScene {
  const OrcFlatNode* nodes;
  size_t n_nodes;
  const int32_t* members;
  size_t n_members;
};
int64_t orc_scene_from_flat(OrcScene* s, const OrcFlatScene* flat, int32_t* node_ids) {
  if (!flat || (flat->n_nodes && !flat->nodes) || (flat->n_members && !flat->members)) return fail("null argument");
  std::vector<int32_t> made(flat->n_nodes, -1), group;
  for (size_t i = 0; i < flat->n_nodes; i++) {
    const OrcFlatNode& R = flat->nodes[i];
    bool bad_ref = false;
    auto resolve = [&](int32_t c) -> int32_t {
      if (c < 0) return ~c;
      if ((size_t)c >= i) {
        bad_ref = true;
        return -1;
      }
      return made[(size_t)c];
    };
    int id = -1;
    switch (R.kind) {
      case K_SPHERE:
        id = orc_sphere(s, R.v, R.v[3], R.mat);
        if (id >= 0 && (R.flags & 1u) && orc_sphere_set_direction(s, id, R.v + 4) < 0) id = -1;
        break;
      case K_QUAD: id = orc_quad(s, R.v, R.v + 3, R.v + 6, R.mat); break;
      case K_BOX: id = orc_box(s, R.v, R.v + 3, R.mat); break;
      case K_DISC: id = orc_disc(s, R.v, R.v[3], R.v + 4, R.mat); break;
      case K_CONE: id = orc_cone_frustum(s, R.v, R.v[3], R.v[4], R.v[5], R.mat); break;
      case K_GROUP:
      case K_BVH: {
        if (R.first < 0 || R.count < 0 || (size_t)R.first + (size_t)R.count > flat->n_members) return fail("flat record: member range outside the member list");
        group.resize((size_t)R.count);
        for (int32_t j = 0; j < R.count; j++) group[(size_t)j] = resolve(flat->members[(size_t)R.first + (size_t)j]);
        if (!bad_ref) id = R.kind == K_GROUP ? orc_group(s, group.data(), R.count) : orc_bvh(s, group.data(), R.count);
        break;
      }
      case K_TRANSLATE:
      case K_ROTATE:
      case K_SCALE:
      case K_MEDIUM: {
        const int32_t child = resolve(R.first);
        if (bad_ref) break;
        if (R.kind == K_TRANSLATE) id = orc_translate(s, child, R.v);
        else if (R.kind == K_ROTATE) id = orc_rotate(s, child, R.v, R.v[3]);
        else if (R.kind == K_SCALE) id = orc_scale(s, child, R.v[0], R.v[1], R.v[2]);
        else if (R.flags & 2u) id = orc_constant_medium_tex(s, child, R.v[0], R.mat);
        else id = orc_constant_medium_color(s, child, R.v[0], R.v[1], R.v[2], R.v[3]);
        break;
      }
      default: return fail("flat record: kind has no constructor");
    }
    if (bad_ref) return fail("flat record refers to a record that is not before it");
    if (id < 0) return -1;
    made[i] = id;
  }
  if (node_ids) std::copy(made.begin(), made.end(), node_ids);
  return (int64_t)flat->n_nodes;
}

int orc_scene_set_world(OrcScene* s, int node) {
  CHECK_NODE(node);
  // every place a medium occupies in the world becomes a node of its own with its own sample keys (Node::instantiate)
  s->world = s->nodes[node]->instantiate(s->nodes[node]);
  s->world->count_tests(1);
  uint32_t next_sample = 0;
  try {
    s->world->number_media(next_sample);
  } catch (const std::exception& e) {
    s->world.reset();
    return fail(e.what());
  }
  return 0;
}
int orc_scene_set_lights(OrcScene* s, int node) {
  if (node < 0) {
    s->lights.reset();
    return 0;
  }
  CHECK_NODE(node);
  s->lights = s->nodes[node];
  return 0;
}
int orc_scene_set_camera(OrcScene* s, const OrcCameraDesc* d) {
  CameraDesc c;
  c.vertical_fov = d->vertical_fov;
  c.aspect_ratio = d->aspect_ratio;
  c.image_width = d->image_width;
  for (int i = 0; i < 3; i++) {
    c.look_from[i] = d->look_from[i];
    c.look_at[i] = d->look_at[i];
    c.up[i] = d->up[i];
    c.background[i] = d->background[i];
  }
  c.defocus_angle = d->defocus_angle;
  c.focus_distance = d->focus_distance;
  c.samples_per_pixel = d->samples_per_pixel;
  c.max_depth = d->max_depth;
  s->cam_desc = c;
  s->camera.reset(new Camera(c));
  return 0;
}
int orc_commit(OrcScene* s) {
  if (!s->world) return fail("no world");
  if (!s->camera) return fail("no camera");
  return 0;
}
int orc_image_size(OrcScene* s, uint32_t* w, uint32_t* h) {
  if (!s->camera) return fail("no camera");
  *w = s->camera->image_width;
  *h = s->camera->image_height;
  return 0;
}
int orc_sqrt_spp(OrcScene* s) { return s->camera ? (int)s->camera->sqrt_spp : -1; }

// The 14 derived camera fields of camera.rs:166-181 as 24 doubles:
// center, pixel00, du, dv, disk_u, disk_v (3 each), then scale, recip_sqrt_spp, defocus_angle, sqrt_spp, w, h.
int orc_camera_derived(OrcScene* s, double* out) {
  if (!s->camera) return fail("no camera");
  const Camera& c = *s->camera;
  const Vec3* vs[6] = {&c.center, &c.pixel00_loc, &c.pixel_delta_u, &c.pixel_delta_v, &c.defocus_disk_u, &c.defocus_disk_v};
  for (int i = 0; i < 6; i++) {
    out[3 * i] = vs[i]->x;
    out[3 * i + 1] = vs[i]->y;
    out[3 * i + 2] = vs[i]->z;
  }
  out[18] = c.pixel_samples_scale;
  out[19] = c.reciprocal_sqrt_spp;
  out[20] = c.defocus_angle;
  out[21] = c.sqrt_spp;
  out[22] = c.image_width;
  out[23] = c.image_height;
  return 0;
}

// Camera rays for (x, y, s_x, s_y) tuples using the Philox stream of the render
// (camera.rs:359-374).  rays: n x 7 doubles (origin, direction, time).
int orc_camera_rays(OrcScene* s, uint64_t seed, size_t n, const uint32_t* xyss, double* rays) {
  if (!s->camera) return fail("no camera");
  PhiloxRandom rng;
  rng.seed(seed);
  RenderContext ctx;
  ctx.random = &rng;
  const Camera& c = *s->camera;
  for (size_t i = 0; i < n; i++) {
    uint32_t x = xyss[4 * i], y = xyss[4 * i + 1], sx = xyss[4 * i + 2], sy = xyss[4 * i + 3];
    rng.set_path(y * c.image_width + x, sy * c.sqrt_spp + sx);
    Ray r = c.get_ray(ctx, x, y, sx, sy);
    double* o = rays + 7 * i;
    o[0] = r.origin.x; o[1] = r.origin.y; o[2] = r.origin.z;
    o[3] = r.direction.x; o[4] = r.direction.y; o[5] = r.direction.z;
    o[6] = r.time;
  }
  return 0;
}

// Closest hit of world for a fixed ray set, interval [0.001, inf) as in camera.rs:264.
// Medium free-flight draws come from the Philox stream keyed (ray index, 0, bounce 0).
int orc_trace_rays(OrcScene* s, uint64_t seed, size_t n, const double* o, const double* d, const double* tm, OrcHit* out) {
  if (!s->world) return fail("no world");
  unsigned nthreads = std::max(1u, std::thread::hardware_concurrency());
  std::vector<std::thread> th;
  std::atomic<size_t> next{0};
  for (unsigned t = 0; t < nthreads; t++)
    th.emplace_back([&]() {
      PhiloxRandom rng;
      rng.seed(seed);
      RenderContext ctx;
      ctx.random = &rng;
      for (;;) {
        size_t b = next.fetch_add(4096);
        if (b >= n) break;
        size_t e = std::min(n, b + 4096);
        for (size_t i = b; i < e; i++) {
          rng.set_path((uint32_t)i, 0);
          rng.set_bounce(0);
          Ray r(v3(o + 3 * i), v3(d + 3 * i), tm ? tm[i] : 0.0);
          HitRecord h;
          OrcHit& oh = out[i];
          std::memset(&oh, 0, sizeof(oh));
          if (s->world->hit(ctx, r, Interval(0.001, kInf), h)) {
            oh.hit = 1;
            oh.t = h.t;
            oh.pt[0] = h.pt.x; oh.pt[1] = h.pt.y; oh.pt[2] = h.pt.z;
            oh.normal[0] = h.normal.x; oh.normal[1] = h.normal.y; oh.normal[2] = h.normal.z;
            oh.u = h.u;
            oh.v = h.v;
            oh.leaf_id = h.leaf_id;
            oh.front_face = h.front_face ? 1 : 0;
            oh.material_kind = h.material->kind();
          } else {
            oh.leaf_id = -1;
            oh.material_kind = -1;
          }
        }
      }
    });
  for (auto& t : th) t.join();
  return 0;
}

// Preorder dump of the world tree: kind, creation id (-1 for interior nodes), bbox
// as xmin,xmax,ymin,ymax,zmin,zmax.  Returns the number of records (which may exceed cap).
static void dump_rec(const Node* n, int32_t* kinds, int32_t* ids, double* boxes, size_t cap, size_t& count) {
  if (count < cap) {
    kinds[count] = n->kind();
    ids[count] = n->id;
    const AABB& b = n->bounding_box();
    double* o = boxes + 6 * count;
    o[0] = b.x.min; o[1] = b.x.max; o[2] = b.y.min; o[3] = b.y.max; o[4] = b.z.min; o[5] = b.z.max;
  }
  count++;
  switch (n->kind()) {
    case K_BVH: {
      auto* b = static_cast<const BVH*>(n);
      dump_rec(b->left.get(), kinds, ids, boxes, cap, count);
      dump_rec(b->right.get(), kinds, ids, boxes, cap, count);
      break;
    }
    case K_GROUP:
      for (auto& c : static_cast<const Group*>(n)->nodes) dump_rec(c.get(), kinds, ids, boxes, cap, count);
      break;
    case K_BOX:
      for (auto& c : static_cast<const BoxPrimitive*>(n)->group.nodes) dump_rec(c.get(), kinds, ids, boxes, cap, count);
      break;
    case K_CONE:
      for (auto& c : static_cast<const ConeFrustum*>(n)->group.nodes) dump_rec(c.get(), kinds, ids, boxes, cap, count);
      break;
    case K_SCALE: dump_rec(static_cast<const Scale*>(n)->object.get(), kinds, ids, boxes, cap, count); break;
    case K_TRANSLATE: dump_rec(static_cast<const Translate*>(n)->object.get(), kinds, ids, boxes, cap, count); break;
    case K_ROTATE: dump_rec(static_cast<const Rotate*>(n)->object.get(), kinds, ids, boxes, cap, count); break;
    case K_MEDIUM: dump_rec(static_cast<const ConstantMedium*>(n)->boundary.get(), kinds, ids, boxes, cap, count); break;
    default: break;
  }
}
int64_t orc_dump_tree(OrcScene* s, int32_t* kinds, int32_t* ids, double* boxes, size_t cap) {
  if (!s->world) return fail("no world");
  size_t count = 0;
  dump_rec(s->world.get(), kinds, ids, boxes, cap, count);
  return (int64_t)count;
}

// The caller loop of crates/cli/src/main.rs:83-182: 10x10 pixel blocks, LIFO shared
// queue, one thread per hardware thread; output is the un-normalised linear sum per
// pixel (double, row-major over the rect, 3 channels).  out_rgb8 (nullable) gets
// camera.rs:343-344 + main.rs:193-198 applied to the sum.
int orc_render_frame(OrcScene* s, const OrcRenderOpts* opts, double* out_sum, uint8_t* out_rgb8, OrcStats* stats) {
  if (!s->world || !s->camera) return fail("scene not complete");
  const Camera& cam = *s->camera;
  uint32_t xmin = opts->xmin, xmax = opts->xmax, ymin = opts->ymin, ymax = opts->ymax;
  if (xmax == 0 && ymax == 0) {
    xmin = 0; ymin = 0; xmax = cam.image_width; ymax = cam.image_height;
  }
  if (xmax > cam.image_width || ymax > cam.image_height || xmin > xmax || ymin > ymax) return fail("bad rect");
  uint32_t stride = opts->stratum_stride ? opts->stratum_stride : 1;
  if (opts->stratum_offset >= stride) return fail("bad stratum offset");
  uint32_t rw = xmax - xmin;
  struct Work { uint32_t x0, x1, y0, y1; };
  std::vector<Work> work;
  const uint32_t B = 10;  // main.rs:27
  for (uint32_t y = ymin; y < ymax; y += B)
    for (uint32_t x = xmin; x < xmax; x += B) work.push_back({x, std::min(x + B, xmax), y, std::min(y + B, ymax)});
  std::mutex mu;
  unsigned nthreads = (opts->flags >> 8) & 0xFF;
  if (nthreads == 0) nthreads = std::max(1u, std::thread::hardware_concurrency());
  bool iterative = opts->flags & 1;
  std::atomic<uint64_t> seg{0}, bvh{0}, prim{0};
  auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nthreads; t++)
    th.emplace_back([&]() {
      PhiloxRandom rng;
      rng.seed(opts->seed);
      RenderContext ctx;
      ctx.random = &rng;
      ctx.media_mode = (int)((opts->flags >> 16) & 3u);
      for (;;) {
        Work w;
        {
          std::lock_guard<std::mutex> lk(mu);
          if (work.empty()) break;
          w = work.back();  // Vec::pop, main.rs:132
          work.pop_back();
        }
        for (uint32_t y = w.y0; y < w.y1; y++)
          for (uint32_t x = w.x0; x < w.x1; x++) {
            Color sum(0, 0, 0);
            if (!iterative) {
              sum = cam.render_sum(ctx, x, y, *s->world, s->lights.get(), opts->stratum_offset, stride);
            } else {
              for (uint32_t sy = 0; sy < cam.sqrt_spp; sy++)
                for (uint32_t sx = 0; sx < cam.sqrt_spp; sx++) {
                  uint32_t st = sy * cam.sqrt_spp + sx;
                  if (st % stride != opts->stratum_offset) continue;
                  rng.set_path(y * cam.image_width + x, st);
                  Ray r = cam.get_ray(ctx, x, y, sx, sy);
                  sum = sum + ray_color_iterative(cam, ctx, r, *s->world, s->lights.get());
                }
            }
            size_t idx = ((size_t)(y - ymin) * rw + (x - xmin)) * 3;
            out_sum[idx] = sum.r;
            out_sum[idx + 1] = sum.g;
            out_sum[idx + 2] = sum.b;
            if (out_rgb8) {
              Color g = cam.resolve(sum);
              const double q = (opts->flags & 8u) ? 255.0 : 255.999;  // wasm lib.rs:269-276 : cli main.rs:193-198
              out_rgb8[idx] = (uint8_t)(g.r * q);
              out_rgb8[idx + 1] = (uint8_t)(g.g * q);
              out_rgb8[idx + 2] = (uint8_t)(g.b * q);
            }
          }
      }
      seg += ctx.n_segments;
      bvh += ctx.n_bvh_visits;
      prim += ctx.n_prim_tests;
    });
  for (auto& t : th) t.join();
  auto t1 = std::chrono::steady_clock::now();
  if (stats) {
    uint32_t nstrata = 0;
    for (uint32_t st = 0; st < cam.sqrt_spp * cam.sqrt_spp; st++)
      if (st % stride == opts->stratum_offset) nstrata++;
    stats->paths = (uint64_t)rw * (ymax - ymin) * nstrata;
    stats->segments = seg;
    stats->bvh_visits = bvh;
    stats->prim_tests = prim;
    stats->ms_total = std::chrono::duration<double, std::milli>(t1 - t0).count();
    stats->ms_trace = stats->ms_shade = stats->ms_other = 0;
    stats->trace_launches = stats->batches = 0;
  }
  return 0;
}

// ------------------------------------------------------------------ known-answer probes
// Thin accessors so the reference's doctest values can be asserted from Python.
void orc_kat_ray_at(const double* o, const double* d, double t, double* out) {
  Vec3 p = Ray(v3(o), v3(d)).at(t);
  out[0] = p.x; out[1] = p.y; out[2] = p.z;
}
int orc_kat_aabb_hit(const double* a, const double* b, const double* o, const double* d, double tmin, double tmax) {
  return AABB::from_points(v3(a), v3(b)).hit(Ray(v3(o), v3(d)), Interval(tmin, tmax)) ? 1 : 0;
}
int orc_kat_aabb_longest_axis(const double* a, const double* b) { return AABB::from_points(v3(a), v3(b)).longest_axis(); }
void orc_kat_aabb_from_points(const double* a, const double* b, double* out6) {
  AABB bb = AABB::from_points(v3(a), v3(b));
  out6[0] = bb.x.min; out6[1] = bb.x.max; out6[2] = bb.y.min; out6[3] = bb.y.max; out6[4] = bb.z.min; out6[5] = bb.z.max;
}
int orc_kat_interval(double mn, double mx, double x, double delta, double* out) {
  Interval i(mn, mx);
  Interval e = i.expand(delta);
  out[0] = e.min; out[1] = e.max; out[2] = i.size();
  Interval sh = i.shifted(delta);
  out[3] = sh.min; out[4] = sh.max;
  return (i.contains(x) ? 1 : 0) | (i.surrounds(x) ? 2 : 0) | (i.is_empty() ? 4 : 0);
}
void orc_kat_interval_union(double a0, double a1, double b0, double b1, double* out) {
  Interval r = Interval::from_intervals(Interval(a0, a1), Interval(b0, b1));
  out[0] = r.min; out[1] = r.max;
}
void orc_kat_vec(const double* a, const double* b, double* out) {
  Vec3 x = v3(a), y = v3(b);
  out[0] = x.length();
  out[1] = x.length_squared();
  out[2] = x.dot(y);
  Vec3 c = x.cross(y);
  out[3] = c.x; out[4] = c.y; out[5] = c.z;
  Vec3 r = x.reflect(y);
  out[6] = r.x; out[7] = r.y; out[8] = r.z;
  out[9] = x.is_near_zero() ? 1.0 : 0.0;
  Vec3 un = x.unit();
  out[10] = un.length();
}
void orc_kat_refract(const double* uv, const double* n, double eta, double* out) {
  Vec3 r = v3(uv).refract(v3(n), eta);
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_kat_color(const double* c, double* out) {
  Color x(c[0], c[1], c[2]);
  Color z = nan_to_zero(x);
  out[0] = z.r; out[1] = z.g; out[2] = z.b;
  Color g = linear_to_gamma(x);
  out[3] = g.r; out[4] = g.g; out[5] = g.b;
  Color cl = color_clamp(x, 0.0, 10.0);
  out[6] = cl.r; out[7] = cl.g; out[8] = cl.b;
}
void orc_kat_sphere_uv(const double* p, double* out) { Sphere::get_uv(v3(p), out[0], out[1]); }
void orc_kat_matvec(const double* m9, const double* v, double* out) {
  Mat3 m;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) m.m[i][j] = m9[3 * i + j];
  Vec3 r = m.mul(v3(v));
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_kat_mock_random(size_t n, double* out) {
  ListRandom r = ListRandom::with_length(n);
  for (size_t i = 0; i < n; i++) out[i] = r.rand();
}
void orc_kat_philox(const uint32_t* ctr, const uint32_t* key, uint32_t* out) { philox4x32_10(ctr, key, out); }
// n uniforms of the render stream for (pixel, stratum, bounce)
void orc_kat_philox_stream(uint64_t seed, uint32_t pixel, uint32_t stratum, uint32_t bounce, size_t n, double* out) {
  PhiloxRandom r;
  r.seed(seed);
  r.set_path(pixel, stratum);
  r.set_bounce(bounce);
  for (size_t i = 0; i < n; i++) out[i] = r.rand();
}
// Perlin tables built by perlin.rs:60-76 from a value-list stream (for checking the
// Python table builder), and noise / turbulence probes.
void orc_kat_perlin_tables(const double* values, size_t nvalues, double* rand_vec, int32_t* px, int32_t* py, int32_t* pz) {
  ListRandom r;
  r.values.assign(values, values + nvalues);
  Perlin p = Perlin::create(r);
  for (int i = 0; i < 256; i++) {
    rand_vec[3 * i] = p.rand_vec[i].x; rand_vec[3 * i + 1] = p.rand_vec[i].y; rand_vec[3 * i + 2] = p.rand_vec[i].z;
    px[i] = p.perm_x[i]; py[i] = p.perm_y[i]; pz[i] = p.perm_z[i];
  }
}
int orc_kat_texture_value(OrcScene* s, int tex, double u, double v, const double* pt, double* out) {
  CHECK_TEX(tex);
  Color c = s->textures[tex]->value(u, v, v3(pt));
  out[0] = c.r; out[1] = c.g; out[2] = c.b;
  return 0;
}

}  // extern "C"
