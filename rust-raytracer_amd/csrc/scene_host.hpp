// Host side of libcaustic_hip: the scene graph built through the C ABI, its
// bounding boxes and BVH (laid out exactly as the reference builds them, so the
// traversal order and the hit/miss decisions are the reference's), the derived
// camera, and the flattening of the graph into the SoA tables of device_scene.h.
// Citations are to the reference, paths relative to crates/core/src/.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/caustic_hip.h"
#include "device_scene.h"

namespace chip {

constexpr double kInf = std::numeric_limits<double>::infinity();
constexpr double kPi = 3.14159265358979323846;

// Box as xmin,xmax,ymin,ymax,zmin,zmax.
struct Box6 {
  double v[6];
};

// axis_aligned_bounding_box.rs:273-285 with interval.rs:148-151,163-165
inline Box6 box_pad(Box6 b) {
  const double delta = 0.0001;
  for (int a = 0; a < 3; a++) {
    if (b.v[2 * a + 1] - b.v[2 * a] < delta) {
      double padding = delta / 2.0;
      b.v[2 * a] = b.v[2 * a] - padding;
      b.v[2 * a + 1] = b.v[2 * a + 1] + padding;
    }
  }
  return b;
}
inline Box6 box_empty() { return box_pad(Box6{{kInf, -kInf, kInf, -kInf, kInf, -kInf}}); }  // aabb.rs:41-47
inline Box6 box_from_points(const double* p, const double* q) {                           // aabb.rs:89-107
  Box6 b;
  for (int a = 0; a < 3; a++) {
    if (p[a] <= q[a]) { b.v[2 * a] = p[a]; b.v[2 * a + 1] = q[a]; }
    else { b.v[2 * a] = q[a]; b.v[2 * a + 1] = p[a]; }
  }
  return box_pad(b);
}
inline Box6 box_union(const Box6& x, const Box6& y) {  // aabb.rs:134-140, interval.rs:77-82
  Box6 b;
  for (int a = 0; a < 3; a++) {
    b.v[2 * a] = std::fmin(x.v[2 * a], y.v[2 * a]);
    b.v[2 * a + 1] = std::fmax(x.v[2 * a + 1], y.v[2 * a + 1]);
  }
  return box_pad(b);
}
inline int box_longest_axis(const Box6& b) {  // aabb.rs:246-258
  double sx = b.v[1] - b.v[0], sy = b.v[3] - b.v[2], sz = b.v[5] - b.v[4];
  if (sx > sy) return sx > sz ? 0 : 2;
  return sy > sz ? 1 : 2;
}
// f64::total_cmp as a strict-weak "less"
inline bool total_less(double a, double b) {
  int64_t ia, ib;
  std::memcpy(&ia, &a, 8);
  std::memcpy(&ib, &b, 8);
  ia ^= (int64_t)((uint64_t)(ia >> 63) >> 1);
  ib ^= (int64_t)((uint64_t)(ib >> 63) >> 1);
  return ia < ib;
}

struct HTexture {
  DTexture d;
};
struct HNode {
  int kind = -1;  // CHIP_NODE_*
  int id = -1;    // creation id, -1 for internally built nodes
  Box6 box;
  double a[17] = {0};  // sphere: c,dc,r | quad: q,u,v,n,w,d,area | translate: off | rotate: m[9] | medium: -1/density
  int mat = -1;
  int c0 = -1, c1 = -1;   // bvh: left,right; translate/rotate/medium: child in c0
  std::vector<int> kids;  // group / box children
  bool referenced = false;
};

struct Flat {
  std::vector<DBvhNode> bvh;
  std::vector<DWNode> wnodes;          // 4-wide culling nodes of the ordered traversal, in slots of bvh[] (same indices for hierarchy roots)
  std::vector<Box6> bvh_child_box[2];  // boxes of each BVH node's left / right child (flattening scratch)
  std::vector<uint8_t> bvh_is_root;    // node was the top of a BoundingVolumeHierarchy::new call (flattening scratch)
  std::vector<DSphere> spheres;
  std::vector<DQuad> quads;
  std::vector<DGroup> groups;
  std::vector<nref_t> gchild;
  std::vector<float> gbox;  // 6 per gchild entry
  std::vector<DTranslate> translates;
  std::vector<DRotate> rotates;
  std::vector<DScale> scales;
  std::vector<DDisc> discs;
  std::vector<DCone> cones;
  std::vector<DMedium> media;
  std::vector<nref_t> xchain;
  std::vector<DLight> lights;
  std::vector<DLightNode> ltree;
  nref_t root = NREF_NONE;
  uint32_t has_lights = 0, lights_is_group = 0;
  uint32_t loose_geometry = 0;  // some primitive of the world is not bounded by its own box
  uint32_t nonunit_normals = 0;  // some Disc keeps a normal that is not of unit length (Disc::new stores it as given, disc.rs:24-41)
  std::vector<DFlatPrim> flat_list;  // the world as a list of primitives, when it is small enough (build_flat_list)
  uint32_t n_wide = 0;          // culling nodes in use: wnodes[0 .. n_wide)
  nref_t ord_root = NREF_NONE;  // where the ordered traversal starts: root, or the culling tree of a Group root's children
};

struct HostScene {
  std::vector<DTexture> textures;
  std::vector<DPerlin> perlins;
  std::vector<uint8_t> texels;
  std::vector<DMaterial> materials;
  std::vector<HNode> nodes;       // all host nodes, user-created and internal
  std::vector<int> handle;        // creation id -> index into nodes
  int world = -1, lights = -1;    // creation ids
  bool has_camera = false;
  chip_camera_desc cam_desc;
  DCamera cam;

  int add_node(HNode&& n, bool user) {
    if (user) {
      n.id = (int)handle.size();
      handle.push_back((int)nodes.size());
    }
    nodes.push_back(std::move(n));
    return (int)nodes.size() - 1;
  }

  // ---- constructors (each returns the host index)
  int make_sphere(const double* c, double r, int mat, bool user) {  // sphere.rs:21-29
    HNode n;
    n.kind = CHIP_NODE_SPHERE;
    for (int i = 0; i < 3; i++) { n.a[i] = c[i]; n.a[3 + i] = 0.0; }
    n.a[6] = r;
    n.mat = mat;
    double lo[3] = {c[0] - r, c[1] - r, c[2] - r}, hi[3] = {c[0] + r, c[1] + r, c[2] + r};
    n.box = box_from_points(lo, hi);
    return add_node(std::move(n), user);
  }
  void sphere_set_direction(int h, const double* d) {  // sphere.rs:31-47
    HNode& n = nodes[h];
    for (int i = 0; i < 3; i++) n.a[3 + i] = d[i];
    double r = n.a[6];
    double lo[3], hi[3];
    // center.at(0.0) = origin + 0.0 * direction, center.at(1.0) = origin + 1.0 * direction (ray.rs:113-115)
    for (int i = 0; i < 3; i++) { double c0 = n.a[i] + d[i] * 0.0; lo[i] = c0 - r; hi[i] = c0 + r; }
    Box6 b1 = box_from_points(lo, hi);
    for (int i = 0; i < 3; i++) { double c1 = n.a[i] + d[i] * 1.0; lo[i] = c1 - r; hi[i] = c1 + r; }
    Box6 b2 = box_from_points(lo, hi);
    n.box = box_union(b1, b2);
  }
  int make_quad(const double* q, const double* u, const double* v, int mat, bool user) {  // quad.rs:59-96
    HNode n;
    n.kind = CHIP_NODE_QUAD;
    double nx = u[1] * v[2] - u[2] * v[1], ny = u[2] * v[0] - u[0] * v[2], nz = u[0] * v[1] - u[1] * v[0];
    double nn = nx * nx + ny * ny + nz * nz;  // length_squared order: xs + ys + zs
    double nl = std::sqrt(nn);
    double normal[3] = {nx / nl, ny / nl, nz / nl};
    double d = normal[0] * q[0] + normal[1] * q[1] + normal[2] * q[2];
    double ndn = nx * nx + ny * ny + nz * nz;  // n.dot(&n)
    double w[3] = {nx / ndn, ny / ndn, nz / ndn};
    for (int i = 0; i < 3; i++) {
      n.a[i] = q[i]; n.a[3 + i] = u[i]; n.a[6 + i] = v[i]; n.a[9 + i] = normal[i]; n.a[12 + i] = w[i];
    }
    n.a[15] = d;
    n.a[16] = nl;  // area = n.length()
    n.mat = mat;
    double quv[3], qu[3], qv[3];
    for (int i = 0; i < 3; i++) { quv[i] = q[i] + u[i] + v[i]; qu[i] = q[i] + u[i]; qv[i] = q[i] + v[i]; }
    n.box = box_union(box_from_points(q, quv), box_from_points(qu, qv));
    return add_node(std::move(n), user);
  }
  int make_group(const std::vector<int>& kids, int kind, bool user) {  // group.rs:15-34
    HNode n;
    n.kind = kind;
    n.box = box_empty();
    for (int k : kids) {
      n.box = box_union(n.box, nodes[k].box);
      nodes[k].referenced = true;
    }
    n.kids = kids;
    return add_node(std::move(n), user);
  }
  int make_box(const double* a, const double* b, int mat) {  // box_node.rs:15-74
    double mn[3], mx[3];
    for (int i = 0; i < 3; i++) { mn[i] = std::fmin(a[i], b[i]); mx[i] = std::fmax(a[i], b[i]); }
    double dx[3] = {mx[0] - mn[0], 0.0, 0.0}, dy[3] = {0.0, mx[1] - mn[1], 0.0}, dz[3] = {0.0, 0.0, mx[2] - mn[2]};
    double ndx[3] = {-dx[0], -dx[1], -dx[2]}, ndz[3] = {-dz[0], -dz[1], -dz[2]};
    std::vector<int> faces;
    double q0[3] = {mn[0], mn[1], mx[2]};
    faces.push_back(make_quad(q0, dx, dy, mat, false));  // front
    double q1[3] = {mx[0], mn[1], mx[2]};
    faces.push_back(make_quad(q1, ndz, dy, mat, false));  // right
    double q2[3] = {mx[0], mn[1], mn[2]};
    faces.push_back(make_quad(q2, ndx, dy, mat, false));  // back
    double q3[3] = {mn[0], mn[1], mn[2]};
    faces.push_back(make_quad(q3, dz, dy, mat, false));  // left
    double q4[3] = {mn[0], mx[1], mx[2]};
    faces.push_back(make_quad(q4, dx, ndz, mat, false));  // top
    double q5[3] = {mn[0], mn[1], mn[2]};
    faces.push_back(make_quad(q5, dx, dz, mat, false));  // bottom
    return make_group(faces, CHIP_NODE_BOX, true);
  }
  // bounding_volume_hierarchy.rs:16-46; items are host indices
  int make_bvh(const std::vector<int>& items, bool user) {
    Box6 span = box_empty();
    for (int k : items) span = box_union(span, nodes[k].box);
    int left, right;
    if (items.empty()) {
      left = make_group({}, CHIP_NODE_GROUP, false);
      right = make_group({}, CHIP_NODE_GROUP, false);
    } else if (items.size() == 1) {
      left = right = items[0];
    } else if (items.size() == 2) {
      left = items[0];
      right = items[1];
    } else {
      int axis = box_longest_axis(span);
      std::vector<int> sorted = items;
      // slice::sort_by is a stable sort; key = bbox.axis_interval(axis).min under total_cmp (bvh.rs:87-91)
      std::stable_sort(sorted.begin(), sorted.end(),
                       [&](int x, int y) { return total_less(nodes[x].box.v[2 * axis], nodes[y].box.v[2 * axis]); });
      size_t mid = sorted.size() / 2;
      left = make_bvh(std::vector<int>(sorted.begin(), sorted.begin() + mid), false);
      right = make_bvh(std::vector<int>(sorted.begin() + mid, sorted.end()), false);
    }
    HNode n;
    n.kind = CHIP_NODE_BVH;
    n.c0 = left;
    n.c1 = right;
    n.box = box_union(nodes[left].box, nodes[right].box);
    nodes[left].referenced = true;
    nodes[right].referenced = true;
    return add_node(std::move(n), user);
  }
  // disc.rs:24-54
  int make_disc(const double* c, double r, const double* nrm, int mat, bool user) {
    HNode n;
    n.kind = CHIP_NODE_DISC;
    for (int i = 0; i < 3; i++) { n.a[i] = c[i]; n.a[3 + i] = nrm[i]; }
    n.a[6] = r;
    n.mat = mat;
    const double delta = 1e-4;
    double lo[3], hi[3];
    for (int i = 0; i < 3; i++) {
      double radius_i = std::fabs(nrm[i]) > 0.9 ? 0.0 : r;
      double ext = radius_i + std::fabs(nrm[i]) * delta;
      lo[i] = c[i] - ext;
      hi[i] = c[i] + ext;
    }
    n.box = box_from_points(lo, hi);
    return add_node(std::move(n), user);
  }
  // ConeFrustumWall::new, cone.rs:116-147
  int make_cone_wall(const double* base, double h, double r1, double r0, int mat) {
    HNode n;
    n.kind = CHIP_NODE_CONE_WALL;
    for (int i = 0; i < 3; i++) n.a[i] = base[i];
    n.a[3] = h;
    n.a[4] = r0;
    n.a[5] = r1;
    n.mat = mat;
    double max_radius = std::fmax(r0, r1);
    double lo[3] = {base[0] - max_radius, base[1], base[2] - max_radius};
    double hi[3] = {base[0] + max_radius, base[1] + h, base[2] + max_radius};
    n.box = box_from_points(lo, hi);
    return add_node(std::move(n), false);
  }
  // ConeFrustum::new, cone.rs:19-75
  int make_cone_frustum(const double* base, double h, double top_radius, double bottom_radius, int mat) {
    std::vector<int> parts;
    if (top_radius > 1e-4) {
      double c[3] = {base[0], base[1] + h, base[2]}, nrm[3] = {0.0, 1.0, 0.0};
      parts.push_back(make_disc(c, top_radius, nrm, mat, false));
    }
    if (bottom_radius > 1e-4) {
      double c[3] = {base[0], base[1], base[2]}, nrm[3] = {0.0, -1.0, 0.0};
      parts.push_back(make_disc(c, bottom_radius, nrm, mat, false));
    }
    parts.push_back(make_cone_wall(base, h, top_radius, bottom_radius, mat));
    return make_group(parts, CHIP_NODE_CONE, true);
  }
  // scale.rs:18-91
  int make_scale(int child, double sx, double sy, double sz) {
    HNode n;
    n.kind = CHIP_NODE_SCALE;
    n.c0 = child;
    const double sc[3] = {sx, sy, sz};
    for (int i = 0; i < 3; i++) {
      n.a[i] = sc[i];
      n.a[3 + i] = std::fabs(sc[i]) > 1e-9 ? 1.0 / sc[i] : kInf;
    }
    const Box6& ob = nodes[child].box;
    Box6 b;
    for (int i = 0; i < 3; i++) {
      b.v[2 * i] = std::fmin(ob.v[2 * i] * sc[i], ob.v[2 * i + 1] * sc[i]);
      b.v[2 * i + 1] = std::fmax(ob.v[2 * i] * sc[i], ob.v[2 * i + 1] * sc[i]);
    }
    n.box = box_pad(b);
    nodes[child].referenced = true;
    return add_node(std::move(n), true);
  }
  int make_translate(int child, const double* off) {  // translate.rs:15-22, aabb.rs:307-313 (no re-padding)
    HNode n;
    n.kind = CHIP_NODE_TRANSLATE;
    n.c0 = child;
    for (int i = 0; i < 3; i++) n.a[i] = off[i];
    const Box6& cb = nodes[child].box;
    for (int a = 0; a < 3; a++) { n.box.v[2 * a] = cb.v[2 * a] + off[a]; n.box.v[2 * a + 1] = cb.v[2 * a + 1] + off[a]; }
    nodes[child].referenced = true;
    return add_node(std::move(n), true);
  }
  int make_rotate(int child, const double* axis_in, double deg) {  // rotate.rs:19-79, 96-131
    HNode n;
    n.kind = CHIP_NODE_ROTATE;
    n.c0 = child;
    double radians = deg * (kPi / 180.0);
    // two libm calls as rotate.rs:21-22 has them: a compiler that sees sin and cos of one value may call sincos
    // instead, whose cosine is not always cos()'s to the last place (one ulp apart at 56.301153430308204 degrees)
    volatile double radians_again = radians;
    double s = std::sin(radians), c = std::cos(radians_again);
    double al = std::sqrt(axis_in[0] * axis_in[0] + axis_in[1] * axis_in[1] + axis_in[2] * axis_in[2]);
    double x = axis_in[0] / al, y = axis_in[1] / al, z = axis_in[2] / al;
    double omc = 1.0 - c;
    double* m = n.a;
    m[0] = c + x * x * omc;     m[1] = x * y * omc - z * s; m[2] = x * z * omc + y * s;
    m[3] = y * x * omc + z * s; m[4] = c + y * y * omc;     m[5] = y * z * omc - x * s;
    m[6] = z * x * omc - y * s; m[7] = z * y * omc + x * s; m[8] = c + z * z * omc;
    const Box6& ob = nodes[child].box;
    double mn[3] = {kInf, kInf, kInf}, mx[3] = {-kInf, -kInf, -kInf};
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 2; j++)
        for (int k = 0; k < 2; k++) {
          double fi = (double)i, fj = (double)j, fk = (double)k;
          double px = fi * ob.v[1] + (1.0 - fi) * ob.v[0];
          double py = fj * ob.v[3] + (1.0 - fj) * ob.v[2];
          double pz = fk * ob.v[5] + (1.0 - fk) * ob.v[4];
          double rx = m[0] * px + m[1] * py + m[2] * pz;
          double ry = m[3] * px + m[4] * py + m[5] * pz;
          double rz = m[6] * px + m[7] * py + m[8] * pz;
          mn[0] = std::fmin(mn[0], rx); mn[1] = std::fmin(mn[1], ry); mn[2] = std::fmin(mn[2], rz);
          mx[0] = std::fmax(mx[0], rx); mx[1] = std::fmax(mx[1], ry); mx[2] = std::fmax(mx[2], rz);
        }
    n.box = box_from_points(mn, mx);
    nodes[child].referenced = true;
    return add_node(std::move(n), true);
  }
  int make_medium(int boundary, double density, int isotropic_mat) {  // constant_medium.rs:19-37, 82-84
    HNode n;
    n.kind = CHIP_NODE_MEDIUM;
    n.c0 = boundary;
    n.a[0] = -1.0 / density;
    n.mat = isotropic_mat;
    n.box = nodes[boundary].box;
    nodes[boundary].referenced = true;
    return add_node(std::move(n), true);
  }

  // CameraBuilder::build, camera.rs:126-182
  void build_camera(const chip_camera_desc& b) {
    cam_desc = b;
    DCamera& c = cam;
    uint32_t ih = (uint32_t)((double)b.image_width / b.aspect_ratio);
    c.image_height = ih < 1 ? 1 : ih;
    c.image_width = b.image_width;
    c.sqrt_spp = (uint32_t)std::sqrt((double)b.samples_per_pixel);
    c.pixel_samples_scale = 1.0 / (double)(c.sqrt_spp * c.sqrt_spp);
    c.recip_sqrt_spp = 1.0 / (double)c.sqrt_spp;
    double theta = b.vertical_fov * (kPi / 180.0);
    double h = std::tan(theta / 2.0);
    double viewport_height = 2.0 * h * b.focus_distance;
    double viewport_width = viewport_height * ((double)c.image_width / (double)c.image_height);
    auto sub = [](const double* p, const double* q, double* o) { for (int i = 0; i < 3; i++) o[i] = p[i] - q[i]; };
    auto norm = [](double* p) {
      double xs = p[0] * p[0], ys = p[1] * p[1], zs = p[2] * p[2];
      double l = std::sqrt(xs + ys + zs);
      for (int i = 0; i < 3; i++) p[i] = p[i] / l;
    };
    auto crossp = [](const double* p, const double* q, double* o) {
      o[0] = p[1] * q[2] - p[2] * q[1];
      o[1] = p[2] * q[0] - p[0] * q[2];
      o[2] = p[0] * q[1] - p[1] * q[0];
    };
    double w[3], u[3], v[3];
    sub(b.look_from, b.look_at, w);
    norm(w);
    crossp(b.up, w, u);
    norm(u);
    crossp(w, u, v);
    double vu[3], vv[3];
    for (int i = 0; i < 3; i++) {
      c.center[i] = b.look_from[i];
      vu[i] = u[i] * viewport_width;      // viewport_width * u
      vv[i] = (-v[i]) * viewport_height;  // viewport_height * -v
      c.du[i] = vu[i] / (double)c.image_width;
      c.dv[i] = vv[i] / (double)c.image_height;
    }
    for (int i = 0; i < 3; i++) {
      double upper_left = c.center[i] - (w[i] * b.focus_distance) - vu[i] / 2.0 - vv[i] / 2.0;
      c.pixel00[i] = upper_left + (c.du[i] + c.dv[i]) * 0.5;
    }
    double defocus_radius = b.focus_distance * std::tan((b.defocus_angle / 2.0) * (kPi / 180.0));
    for (int i = 0; i < 3; i++) {
      c.disk_u[i] = u[i] * defocus_radius;
      c.disk_v[i] = v[i] * defocus_radius;
      c.background[i] = b.background[i];
    }
    c.max_depth = b.max_depth;
    c.defocus_angle = b.defocus_angle;
    has_camera = true;
  }

  // ---- flattening
  struct Flattener {
    const HostScene& H;
    Flat& F;
    uint32_t seq = 0;  // primitives emitted so far, in the reference's depth-first visiting order
    uint64_t medium_samples = 0;  // free-flight sample keys handed out so far (DMedium::sample_base)
    uint32_t boundary_queries = 0;  // inside a medium's boundary: boundary queries per world.hit of that medium (2 x its tests)
    explicit Flattener(const HostScene& h, Flat& f) : H(h), F(f) {}

    void chain_of(nref_t x, std::vector<nref_t>& out) const {
      out.clear();
      if (x == NREF_NONE) return;
      uint32_t first, n;
      if (NREF_KIND(x) == NK_TRANSLATE) { first = F.translates[NREF_INDEX(x)].chain_first; n = F.translates[NREF_INDEX(x)].chain_len; }
      else if (NREF_KIND(x) == NK_SCALE) { first = F.scales[NREF_INDEX(x)].chain_first; n = F.scales[NREF_INDEX(x)].chain_len; }
      else { first = F.rotates[NREF_INDEX(x)].chain_first; n = F.rotates[NREF_INDEX(x)].chain_len; }
      out.assign(F.xchain.begin() + first, F.xchain.begin() + first + n);
    }

    nref_t emit_sphere(const HNode& n, nref_t xform, int leaf_id) {
      DSphere s;
      std::memset(&s, 0, sizeof(s));
      for (int i = 0; i < 3; i++) { s.c[i] = n.a[i]; s.dc[i] = n.a[3 + i]; }
      s.r = n.a[6];
      s.mat = MATK_PACK(n.mat, H.materials[n.mat].kind);
      s.xform = xform;
      s.leaf_id = leaf_id;
      s.seq = seq++;
      if (xform != NREF_NONE && NREF_KIND(xform) == NK_TRANSLATE && F.translates[NREF_INDEX(xform)].chain_len == 1) {
        for (int i = 0; i < 3; i++) s.xoff[i] = F.translates[NREF_INDEX(xform)].off[i];
        s.xoff_only = 1;
      }
      F.spheres.push_back(s);
      return NREF(NK_SPHERE, F.spheres.size() - 1);
    }
    nref_t emit_quad(const HNode& n, nref_t xform, int leaf_id) {
      DQuad q;
      std::memset(&q, 0, sizeof(q));
      for (int i = 0; i < 3; i++) { q.q[i] = n.a[i]; q.u[i] = n.a[3 + i]; q.v[i] = n.a[6 + i]; q.n[i] = n.a[9 + i]; q.w[i] = n.a[12 + i]; }
      q.d = n.a[15];
      q.area = n.a[16];
      q.mat = MATK_PACK(n.mat, H.materials[n.mat].kind);
      q.xform = xform;
      q.leaf_id = leaf_id;
      q.seq = seq++;
      F.quads.push_back(q);
      return NREF(NK_QUAD, F.quads.size() - 1);
    }

    nref_t emit_disc(const HNode& n, nref_t xform, int leaf_id) {
      DDisc d;
      std::memset(&d, 0, sizeof(d));
      for (int i = 0; i < 3; i++) { d.c[i] = n.a[i]; d.n[i] = n.a[3 + i]; }
      d.r = n.a[6];
      d.mat = MATK_PACK(n.mat, H.materials[n.mat].kind);
      d.xform = xform;
      d.leaf_id = leaf_id;
      d.seq = seq++;
      // Disc::new sizes the box as if the normal were along an axis whenever a component exceeds 0.9
      // (disc.rs:25-35); a tilted disc then sticks out of it
      {
        const double len2 = n.a[3] * n.a[3] + n.a[4] * n.a[4] + n.a[5] * n.a[5];
        // the reference's Lambertian weighs a bounce by scattering_pdf / pdf_value, which is |normal| and not 1 on
        // such a disc (lambertian.rs scattering_pdf takes the hit's normal as it is, the cosine pdf a unit one); the
        // fast shading arithmetic takes the ratio as 1, so these scenes are shaded in the reference's arithmetic
        if (!(std::fabs(len2 - 1.0) <= 1e-6)) F.nonunit_normals = 1;
        for (int i = 0; i < 3 && len2 > 0.0; i++) {
          const double reach = d.r * std::sqrt(std::fmax(0.0, 1.0 - n.a[3 + i] * n.a[3 + i] / len2));
          if (n.a[i] - reach < n.box.v[2 * i] || n.a[i] + reach > n.box.v[2 * i + 1]) F.loose_geometry = 1;
        }
      }
      F.discs.push_back(d);
      return NREF(NK_DISC, F.discs.size() - 1);
    }
    nref_t emit_cone(const HNode& n, nref_t xform, int leaf_id) {
      DCone c;
      std::memset(&c, 0, sizeof(c));
      for (int i = 0; i < 3; i++) c.base[i] = n.a[i];
      c.h = n.a[3];
      c.r0 = n.a[4];
      c.r1 = n.a[5];
      c.mat = MATK_PACK(n.mat, H.materials[n.mat].kind);
      c.xform = xform;
      c.leaf_id = leaf_id;
      c.seq = seq++;
      F.cones.push_back(c);
      return NREF(NK_CONE, F.cones.size() - 1);
    }

    // h: host index; xform: innermost transform ancestor; in_boundary: inside a medium's boundary subtree;
    // leaf_override: creation id reported for hits (a box reports its own id for its faces).
    // tests: how often one world.hit of the reference reaches this node (a BoundingVolumeHierarchy over a single
    // object holds it as both children, bvh.rs:22-26, and visits both); only a ConstantMedium cares.
    nref_t emit(int h, nref_t xform, bool in_boundary, int leaf_override, uint32_t tests = 1) {
      const HNode& n = H.nodes[h];
      switch (n.kind) {
        case CHIP_NODE_SPHERE: return emit_sphere(n, xform, leaf_override >= 0 ? leaf_override : n.id);
        case CHIP_NODE_QUAD: return emit_quad(n, xform, leaf_override >= 0 ? leaf_override : n.id);
        case CHIP_NODE_DISC: return emit_disc(n, xform, leaf_override >= 0 ? leaf_override : n.id);
        case CHIP_NODE_CONE_WALL: return emit_cone(n, xform, leaf_override >= 0 ? leaf_override : n.id);
        case CHIP_NODE_GROUP:
        case CHIP_NODE_CONE:
        case CHIP_NODE_BOX: {
          int lo = (n.kind == CHIP_NODE_BOX || n.kind == CHIP_NODE_CONE) ? n.id : leaf_override;
          std::vector<nref_t> refs;
          refs.reserve(n.kids.size());
          for (int k : n.kids) refs.push_back(emit(k, xform, in_boundary, lo, tests));
          DGroup g;
          g.first = (uint32_t)F.gchild.size();
          g.count = (uint32_t)refs.size();
          F.gchild.insert(F.gchild.end(), refs.begin(), refs.end());
          // each child's box, in f32 and rounded outward: what build_wnodes() files the child under when it opens the
          // Group into a culling tree (host only)
          for (int k : n.kids) {
            float b[6];
            set_cull_box(b, H.nodes[k].box);
            F.gbox.insert(F.gbox.end(), b, b + 6);
          }
          F.groups.push_back(g);
          return NREF(NK_GROUP, F.groups.size() - 1);
        }
        case CHIP_NODE_BVH: {
          size_t idx = F.bvh.size();
          F.bvh.push_back(DBvhNode{});
          F.bvh_child_box[0].push_back(H.nodes[n.c0].box);
          F.bvh_child_box[1].push_back(H.nodes[n.c1].box);
          F.bvh_is_root.push_back(n.id >= 0 ? 1 : 0);
          nref_t l = emit(n.c0, xform, in_boundary, leaf_override, n.c1 == n.c0 ? 2 * tests : tests);
          nref_t r = n.c1 == n.c0 ? l : emit(n.c1, xform, in_boundary, leaf_override, tests);
          DBvhNode& b = F.bvh[idx];
          for (int i = 0; i < 6; i++) b.box[i] = n.box.v[i];
          b.left = l;
          b.right = r;
          b.pad[0] = b.pad[1] = 0;
          return NREF(NK_BVH, idx);
        }
        case CHIP_NODE_TRANSLATE:
        case CHIP_NODE_ROTATE:
        case CHIP_NODE_SCALE: {
          std::vector<nref_t> chain;
          chain_of(xform, chain);
          nref_t self;
          size_t idx;
          if (n.kind == CHIP_NODE_TRANSLATE) {
            idx = F.translates.size();
            F.translates.push_back(DTranslate{});
            self = NREF(NK_TRANSLATE, idx);
          } else if (n.kind == CHIP_NODE_ROTATE) {
            idx = F.rotates.size();
            F.rotates.push_back(DRotate{});
            self = NREF(NK_ROTATE, idx);
          } else {
            idx = F.scales.size();
            F.scales.push_back(DScale{});
            self = NREF(NK_SCALE, idx);
          }
          chain.push_back(self);
          uint32_t cf = (uint32_t)F.xchain.size();
          F.xchain.insert(F.xchain.end(), chain.begin(), chain.end());
          if (n.kind == CHIP_NODE_TRANSLATE) {
            DTranslate& t = F.translates[idx];
            for (int i = 0; i < 3; i++) t.off[i] = n.a[i];
            t.parent_xform = xform;
            t.chain_first = cf;
            t.chain_len = (uint32_t)chain.size();
          } else if (n.kind == CHIP_NODE_ROTATE) {
            DRotate& t = F.rotates[idx];
            for (int i = 0; i < 9; i++) t.m[i] = n.a[i];
            t.parent_xform = xform;
            t.chain_first = cf;
            t.chain_len = (uint32_t)chain.size();
          } else {
            DScale& t = F.scales[idx];
            for (int i = 0; i < 9; i++) t.m[i] = t.inv[i] = 0.0;
            for (int i = 0; i < 3; i++) { t.m[4 * i] = n.a[i]; t.inv[4 * i] = n.a[3 + i]; }
            t.parent_xform = xform;
            t.chain_first = cf;
            t.chain_len = (uint32_t)chain.size();
          }
          nref_t c = emit(n.c0, self, in_boundary, leaf_override, tests);
          if (n.kind == CHIP_NODE_TRANSLATE) {
            F.translates[idx].child = c;
            nref_t only = NREF_NONE;
            if (NREF_KIND(c) == NK_SPHERE) only = c;
            else if (NREF_KIND(c) == NK_GROUP && F.groups[NREF_INDEX(c)].count == 1 && NREF_KIND(F.gchild[F.groups[NREF_INDEX(c)].first]) == NK_SPHERE)
              only = F.gchild[F.groups[NREF_INDEX(c)].first];
            F.translates[idx].only_sphere = only;
            if (only != NREF_NONE) {
              const DSphere& sp = F.spheres[NREF_INDEX(only)];
              DTranslate& t = F.translates[idx];
              t.sphere_mat = sp.mat;
              for (int i = 0; i < 3; i++) { t.sc[i] = sp.c[i]; t.sdc[i] = sp.dc[i]; }
              t.sr = sp.r;
            }
          }
          else if (n.kind == CHIP_NODE_ROTATE) F.rotates[idx].child = c;
          else F.scales[idx].child = c;
          // the run of nested instances below this one (stopping before a Translate that is tested in place)
          nref_t ex = self, ec = c;
          for (;;) {
            const uint32_t k = NREF_KIND(ec), i = NREF_INDEX(ec);
            if (k == NK_TRANSLATE && F.translates[i].only_sphere == NREF_NONE) { ex = ec; ec = F.translates[i].child; }
            else if (k == NK_ROTATE) { ex = ec; ec = F.rotates[i].child; }
            else if (k == NK_SCALE) { ex = ec; ec = F.scales[i].child; }
            else break;
          }
          if (n.kind == CHIP_NODE_TRANSLATE) { F.translates[idx].enter_xform = ex; F.translates[idx].enter_child = ec; }
          else if (n.kind == CHIP_NODE_ROTATE) { F.rotates[idx].enter_xform = ex; F.rotates[idx].enter_child = ec; }
          else { F.scales[idx].enter_xform = ex; F.scales[idx].enter_child = ec; }
          return self;
        }
        case CHIP_NODE_MEDIUM: {
          size_t idx = F.media.size();
          F.media.push_back(DMedium{});
          // keys in depth-first order of the expanded graph: a medium node reached along two ways is two media.  A medium
          // inside another medium's boundary (boundary_queries != 0: how many boundary queries one world.hit makes of the
          // enclosing medium, two per test) owns a run of `tests` keys per such query.
          const bool inside_boundary = in_boundary;
          const uint32_t sample_base = (uint32_t)medium_samples;
          medium_samples += (uint64_t)tests * (inside_boundary ? boundary_queries : 1u);
          if (medium_samples >= (1ull << 31)) throw std::runtime_error("too many ConstantMedium samples per ray segment (2^31 at most)");
          const uint32_t outer_queries = boundary_queries;
          boundary_queries = 2u * tests;
          nref_t b = emit(n.c0, xform, true, -1);
          boundary_queries = outer_queries;
          const bool nested = F.media.size() > idx + 1;  // the boundary brought media of its own
          if (nested && inside_boundary) throw std::runtime_error("a ConstantMedium whose boundary holds media, inside another medium's boundary, is not supported (one level of nesting)");
          DMedium& m = F.media[idx];
          m.neg_inv_density = n.a[0];
          m.boundary = b;
          m.mat = MATK_PACK(n.mat, H.materials[n.mat].kind);
          m.xform = xform;
          m.leaf_id = n.id;
          m.tests = tests;
          m.sample_base = sample_base;
          // the common boundary besides a sphere: a box, possibly inside instances that hold nothing else
          m.bx_inner = NREF_NONE;
          m.bq_first = m.bq_count = 0;
          m.nested = nested ? 1u : 0u;
          {
            nref_t inner = NREF_NONE, c = b;
            for (int guard = 0; guard < 64; guard++) {
              const uint32_t k = NREF_KIND(c), i = NREF_INDEX(c);
              if (k == NK_TRANSLATE && F.translates[i].only_sphere == NREF_NONE) { inner = c; c = F.translates[i].child; }
              else if (k == NK_ROTATE) { inner = c; c = F.rotates[i].child; }
              else if (k == NK_SCALE) { inner = c; c = F.scales[i].child; }
              else break;
            }
            if (NREF_KIND(c) == NK_GROUP) {
              const DGroup g = F.groups[NREF_INDEX(c)];
              bool quads_only = g.count >= 1 && g.count <= 6;
              for (uint32_t j = 0; j < g.count && quads_only; j++) quads_only = NREF_KIND(F.gchild[g.first + j]) == NK_QUAD;
              if (quads_only) { m.bx_inner = inner; m.bq_first = g.first; m.bq_count = g.count; }
            }
          }
          // the boundary queries of a medium inside a boundary are answered in place (the traversal of a boundary carries
          // no sub-traversal of its own)
          if (inside_boundary && m.bq_count == 0 && NREF_KIND(b) != NK_SPHERE)
            throw std::runtime_error("a ConstantMedium inside another medium's boundary must itself be bounded by a sphere or a box");
          return NREF(NK_MEDIUM, idx);
        }
      }
      throw std::runtime_error("unknown node kind");
    }

    // One member of the lights node: a leaf with its share of the pdf, or a Group (a ConeFrustum samples as its Group,
    // cone.rs:91-97) whose members follow, each with 1 / len of the Group's share (group.rs:60-78).
    void emit_light_leaf(int h, double weight) {
      const HNode& n = H.nodes[h];
      DLight L;
      if (n.kind == CHIP_NODE_SPHERE) { L.kind = LK_SPHERE; L.index = NREF_INDEX(emit_sphere(n, NREF_NONE, n.id)); }
      else if (n.kind == CHIP_NODE_QUAD) { L.kind = LK_QUAD; L.index = NREF_INDEX(emit_quad(n, NREF_NONE, n.id)); }
      else if (n.kind == CHIP_NODE_DISC) { L.kind = LK_DISC; L.index = NREF_INDEX(emit_disc(n, NREF_NONE, n.id)); }
      else if (n.kind == CHIP_NODE_CONE_WALL) { L.kind = LK_CONE_WALL; L.index = NREF_INDEX(emit_cone(n, NREF_NONE, n.id)); }
      else { L.kind = LK_OTHER; L.index = 0; }  // Node::pdf_value / random defaults (object/mod.rs:59-65)
      L.weight = weight;
      F.lights.push_back(L);
    }
    // Fills tree node `at` (already in F.ltree) for the Group `h` whose share of the pdf is `share`.
    void emit_light_group(int h, size_t at, double share, int depth) {
      if (depth >= CHIP_LIGHT_DEPTH) throw std::runtime_error("Groups inside the lights Group are nested deeper than 16 levels");
      const std::vector<int> kids = H.nodes[h].kids;
      const size_t first = F.ltree.size();
      F.ltree[at] = DLightNode{1u, (uint32_t)first, (uint32_t)kids.size(), 0u};
      F.ltree.resize(first + kids.size(), DLightNode{0u, 0u, 0u, 0u});
      const double weight = share * (1.0 / (double)kids.size());
      for (size_t k = 0; k < kids.size(); k++) {
        const HNode& c = H.nodes[kids[k]];
        if (c.kind == CHIP_NODE_GROUP || c.kind == CHIP_NODE_CONE) {
          emit_light_group(kids[k], first + k, weight, depth + 1);
        } else {
          F.ltree[first + k] = DLightNode{0u, 0u, 0u, (uint32_t)F.lights.size()};
          emit_light_leaf(kids[k], weight);
        }
      }
    }
  };

  void flatten(Flat& F) const {
    if (world < 0) throw std::runtime_error("scene has no world");
    if (materials.size() >= MATK_MAX_MATERIALS) throw std::runtime_error("too many materials (2^21 at most)");
    if (!has_camera) throw std::runtime_error("scene has no camera");
    Flattener fl(*this, F);
    F.root = fl.emit(handle[world], NREF_NONE, false, -1);
    build_flat_list(F);  // (before the lights add their own copies of primitives to the tables)
    if (lights >= 0) {
      F.has_lights = 1;
      const HNode& L = nodes[handle[lights]];
      if (L.kind == CHIP_NODE_GROUP || L.kind == CHIP_NODE_CONE) {  // a ConeFrustum samples as its Group (cone.rs:91-97)
        F.lights_is_group = 1;
        F.ltree.assign(1, DLightNode{1u, 0u, 0u, 0u});
        fl.emit_light_group(handle[lights], 0, 1.0, 0);
      } else {
        F.lights_is_group = 0;
        fl.emit_light_leaf(handle[lights], 1.0);
      }
    }
    reorder_bvh_breadth_first(F);
    build_wnodes(F);
    if (F.bvh.size() >= (1u << 28) || F.spheres.size() >= (1u << 28) || F.quads.size() >= (1u << 28) || F.discs.size() >= (1u << 28) || F.cones.size() >= (1u << 28))
      throw std::runtime_error("scene too large for 28-bit node indices");
  }

  // The flat form: a world of at most CHIP_FLAT_MAX spheres and quads, under Translates and Rotates or under nothing, is
  // also kept as a plain list.  The closest hit does not depend on the order primitives are tried in once exact ties go by
  // the reference's visiting order, and testing a primitive whose enclosing boxes the ray misses finds nothing (the boxes
  // contain it), so a kernel may try the whole list for every ray, all lanes in step, instead of walking hierarchies
  // lane by lane.  Called when the tables hold the world's primitives only.
  static void build_flat_list(Flat& F) {
    F.flat_list.clear();
    const size_t n = F.spheres.size() + F.quads.size();
    if (n == 0 || n > CHIP_FLAT_MAX) return;
    if (!F.media.empty() || !F.discs.empty() || !F.cones.empty() || !F.scales.empty()) return;
    std::vector<DFlatPrim> list;
    for (size_t i = 0; i < F.quads.size(); i++) list.push_back(DFlatPrim{NREF(NK_QUAD, i), F.quads[i].xform, 0u, 0u});
    for (size_t i = 0; i < F.spheres.size(); i++) list.push_back(DFlatPrim{NREF(NK_SPHERE, i), F.spheres[i].xform, 0u, 0u});
    // world space first (NREF_NONE is the largest reference: mapped to 0 for the ordering), then one transform after the other
    std::stable_sort(list.begin(), list.end(), [](const DFlatPrim& a, const DFlatPrim& b) {
      const uint64_t ka = a.xform == NREF_NONE ? 0 : (uint64_t)a.xform + 1, kb = b.xform == NREF_NONE ? 0 : (uint64_t)b.xform + 1;
      return ka < kb;
    });
    nref_t cur = NREF_NONE;
    for (DFlatPrim& e : list) {
      e.enter = e.xform != cur ? 1u : 0u;
      cur = e.xform;
    }
    F.flat_list.swap(list);
  }

  // Renumbers the BVH table breadth-first (world root first, then nested BVHs in order of discovery) so
  // that the leading records are the top of the tree, which the traversal kernel keeps in LDS.
  // Topology, boxes and child order are untouched; only the table indices change.
  static void reorder_bvh_breadth_first(Flat& F) {
    const size_t n = F.bvh.size();
    if (n == 0) return;
    std::vector<uint32_t> order;  // new index -> old index
    order.reserve(n);
    std::vector<uint32_t> newidx(n, 0xFFFFFFFFu);
    auto bfs_from = [&](uint32_t root_old) {
      if (newidx[root_old] != 0xFFFFFFFFu) return;
      size_t head = order.size();
      newidx[root_old] = (uint32_t)order.size();
      order.push_back(root_old);
      while (head < order.size()) {
        const DBvhNode& b = F.bvh[order[head++]];
        for (nref_t c : {b.left, b.right}) {
          if (NREF_KIND(c) == NK_BVH && newidx[NREF_INDEX(c)] == 0xFFFFFFFFu) {
            newidx[NREF_INDEX(c)] = (uint32_t)order.size();
            order.push_back(NREF_INDEX(c));
          }
        }
      }
    };
    if (NREF_KIND(F.root) == NK_BVH) bfs_from(NREF_INDEX(F.root));
    for (uint32_t i = 0; i < n; i++) bfs_from(i);  // BVHs below groups / transforms / media, in table order
    apply_bvh_order(F, order);
  }
  // Renumbers the hierarchy tables: entry i of the new tables is entry order[i] of the old ones; every reference
  // to a hierarchy node follows.
  static void apply_bvh_order(Flat& F, const std::vector<uint32_t>& order) {
    const size_t n = F.bvh.size();
    std::vector<uint32_t> newidx(n, 0);
    for (size_t i = 0; i < n; i++) newidx[order[i]] = (uint32_t)i;
    auto remap = [&](nref_t r) -> nref_t { return NREF_KIND(r) == NK_BVH ? NREF(NK_BVH, newidx[NREF_INDEX(r)]) : r; };
    std::vector<DBvhNode> nb(n);
    for (size_t i = 0; i < n; i++) {
      nb[i] = F.bvh[order[i]];
      nb[i].left = remap(nb[i].left);
      nb[i].right = remap(nb[i].right);
    }
    F.bvh.swap(nb);
    if (F.wnodes.size() == n) {
      std::vector<DWNode> nw(n);
      for (size_t i = 0; i < n; i++) {
        nw[i] = F.wnodes[order[i]];
        for (int k = 0; k < 4; k++) nw[i].child[k] = remap(nw[i].child[k]);
      }
      F.wnodes.swap(nw);
    }
    for (int side = 0; side < 2; side++)
      if (F.bvh_child_box[side].size() == n) {
        std::vector<Box6> cb(n);
        for (size_t i = 0; i < n; i++) cb[i] = F.bvh_child_box[side][order[i]];
        F.bvh_child_box[side].swap(cb);
      }
    if (F.bvh_is_root.size() == n) {
      std::vector<uint8_t> ir(n);
      for (size_t i = 0; i < n; i++) ir[i] = F.bvh_is_root[order[i]];
      F.bvh_is_root.swap(ir);
    }
    F.root = remap(F.root);
    F.ord_root = remap(F.ord_root);
    for (nref_t& c : F.gchild) c = remap(c);
    for (DTranslate& t : F.translates) { t.child = remap(t.child); t.enter_child = remap(t.enter_child); t.ord_child = remap(t.ord_child); }
    for (DRotate& t : F.rotates) { t.child = remap(t.child); t.enter_child = remap(t.enter_child); t.ord_child = remap(t.ord_child); }
    for (DScale& t : F.scales) { t.child = remap(t.child); t.enter_child = remap(t.enter_child); t.ord_child = remap(t.ord_child); }
    for (DMedium& m : F.media) m.boundary = remap(m.boundary);
  }

  // f32 neighbours of an f64 value, rounded away from / towards the box interior.
  static float f32_down(double v) {
    float f = (float)v;
    return (double)f > v ? std::nextafterf(f, -std::numeric_limits<float>::infinity()) : f;
  }
  static float f32_up(double v) {
    float f = (float)v;
    return (double)f < v ? std::nextafterf(f, std::numeric_limits<float>::infinity()) : f;
  }
  // The culling trees of the ordered traversal.  Every BoundingVolumeHierarchy of the scene keeps its
  // reference-built tree in F.bvh (that tree fixes the visiting order, hence which primitive wins an exact
  // tie, and is what the literal traversal walks); for culling, the same member list is re-split by a binned
  // surface-area heuristic, which is free to separate a huge member (the r = 1000 ground sphere) from the
  // small ones instead of cutting the sorted list in half (bvh.rs:16-46).  A culling node stores the f32
  // boxes of up to four children: minima rounded down and maxima up, so each contains the reference's f64
  // box; infinities become +-FLT_MAX so the device's multiply-add slab test never forms inf - inf.  The
  // culling nodes of one hierarchy live in the table slots of that hierarchy's reference nodes, root in
  // the root's slot, top levels first, and in appended slots when those run out (see build_wnodes).
  struct SahLeaf {
    nref_t ref;
    Box6 box;
    double c[3];
  };
  static double box_area(const Box6& b) {
    double dx = b.v[1] - b.v[0], dy = b.v[3] - b.v[2], dz = b.v[5] - b.v[4];
    if (!(dx >= 0.0) || !(dy >= 0.0) || !(dz >= 0.0)) return 0.0;
    return 2.0 * (dx * dy + dy * dz + dz * dx);
  }
  static Box6 box_merge(const Box6& x, const Box6& y) {
    Box6 b;
    for (int a = 0; a < 3; a++) {
      b.v[2 * a] = std::fmin(x.v[2 * a], y.v[2 * a]);
      b.v[2 * a + 1] = std::fmax(x.v[2 * a + 1], y.v[2 * a + 1]);
    }
    return b;
  }
  static void set_cull_box(float* out, const Box6& b) {
    const float big = std::numeric_limits<float>::max();
    // a NaN bound does not constrain (the plane moves out to where every ray passes)
    auto clampf = [&](float v, float nan_as) { return v != v ? nan_as : (v > big ? big : (v < -big ? -big : v)); };
    for (int a = 0; a < 3; a++) {
      out[2 * a] = clampf(f32_down(b.v[2 * a]), -big);
      out[2 * a + 1] = clampf(f32_up(b.v[2 * a + 1]), big);
    }
  }
  // The culling test takes a ray's near plane on each axis by the sign of its direction and never orders the two
  // plane distances itself, so a box whose low planes lie above its high planes is missed by every ray and a box
  // spanning the whole f32 range is hit by every ray.
  static void set_unreachable_box(float* out) {
    const float big = std::numeric_limits<float>::max();
    for (int a = 0; a < 3; a++) { out[2 * a] = big; out[2 * a + 1] = -big; }
  }
  static void set_unbounded_box(float* out) {
    const float big = std::numeric_limits<float>::max();
    for (int a = 0; a < 3; a++) { out[2 * a] = -big; out[2 * a + 1] = big; }
  }
  // Splits leaves[lo, hi) (hi - lo >= 2) and returns the split position.
  static size_t sah_split(std::vector<SahLeaf>& leaves, size_t lo, size_t hi) {
    const int kBins = 16;
    const size_t n = hi - lo;
    double cmin[3] = {kInf, kInf, kInf}, cmax[3] = {-kInf, -kInf, -kInf};
    for (size_t i = lo; i < hi; i++)
      for (int a = 0; a < 3; a++) {
        cmin[a] = std::fmin(cmin[a], leaves[i].c[a]);
        cmax[a] = std::fmax(cmax[a], leaves[i].c[a]);
      }
    double best_cost = kInf;
    int best_axis = -1, best_bin = -1;
    for (int a = 0; a < 3; a++) {
      const double ext = cmax[a] - cmin[a];
      if (!(ext > 0.0) || !std::isfinite(ext)) continue;
      Box6 bb[kBins];
      size_t cnt[kBins] = {0};
      const Box6 none{{kInf, -kInf, kInf, -kInf, kInf, -kInf}};
      for (int k = 0; k < kBins; k++) bb[k] = none;
      const double scale = kBins / ext;
      for (size_t i = lo; i < hi; i++) {
        int k = (int)((leaves[i].c[a] - cmin[a]) * scale);
        k = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
        bb[k] = box_merge(bb[k], leaves[i].box);
        cnt[k]++;
      }
      double right_area[kBins];
      size_t right_cnt[kBins];
      Box6 acc = none;
      size_t c = 0;
      for (int k = kBins - 1; k > 0; k--) {
        acc = box_merge(acc, bb[k]);
        c += cnt[k];
        right_area[k] = box_area(acc);
        right_cnt[k] = c;
      }
      acc = none;
      c = 0;
      for (int k = 0; k + 1 < kBins; k++) {
        acc = box_merge(acc, bb[k]);
        c += cnt[k];
        if (c == 0 || right_cnt[k + 1] == 0) continue;
        double cost = box_area(acc) * (double)c + right_area[k + 1] * (double)right_cnt[k + 1];
        if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = k; }
      }
    }
    if (best_axis < 0) return lo + n / 2;  // coincident centroids: any split will do
    const double ext = cmax[best_axis] - cmin[best_axis];
    const double scale = kBins / ext;
    auto mid = std::partition(leaves.begin() + lo, leaves.begin() + hi, [&](const SahLeaf& l) {
      int k = (int)((l.c[best_axis] - cmin[best_axis]) * scale);
      k = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
      return k <= best_bin;
    });
    size_t m = (size_t)(mid - leaves.begin());
    return (m == lo || m == hi) ? lo + n / 2 : m;
  }
  static void set_wide_child(DWNode& w, int k, nref_t ref, const float* b) {
    w.child[k] = ref;
    w.lo_x[k] = b[0]; w.hi_x[k] = b[1]; w.lo_y[k] = b[2]; w.hi_y[k] = b[3]; w.lo_z[k] = b[4]; w.hi_z[k] = b[5];
  }
  // Binary SAH tree over one hierarchy's members, then folded into nodes of up to four children: a node takes the two
  // children of its binary node and, while it has room, replaces the inner child with the largest box by that
  // child's own two children.
  struct BinNode {
    Box6 box;
    int left = -1, right = -1;  // -1: a leaf holding `ref`
    nref_t ref = NREF_NONE;
  };
  static void init_wnode(DWNode& w) {
    std::memset(&w, 0, sizeof(w));
    float none[6];
    set_unreachable_box(none);
    for (int k = 0; k < 4; k++) set_wide_child(w, k, NREF_NONE, none);  // an unused slot is culled like a missed box
  }
  // The culling trees of the ordered traversal.  Every BoundingVolumeHierarchy gets one over its members, and a Group
  // is opened: its children become members of the tree it sits in (or, for a Group that is the content of an
  // instance or the world itself, of a tree of their own), because the closest hit among a Group's children
  // (group.rs:44-54) does not depend on the order they are tried in once exact ties go by the reference's visiting
  // order.  The ordered traversal thereby never meets a Group, only culling nodes, primitives, instances and media.
  // Trees that need more nodes than their hierarchy has table slots (or have no hierarchy) get slots appended to both
  // tables; the reference-order table's entries there are never visited.
  static void build_wnodes(Flat& F) {
    const size_t n0 = F.bvh.size();
    F.wnodes.resize(n0);
    for (size_t i = 0; i < n0; i++) init_wnode(F.wnodes[i]);
    std::vector<SahLeaf> leaves;
    std::vector<uint32_t> slots, stack;
    std::vector<BinNode> bin;
    struct Job { size_t lo, hi; int node; };
    std::vector<Job> jobs;
    struct WJob { int bin; uint32_t slot; };
    std::vector<WJob> queue;
    auto new_slot = [&]() -> uint32_t {
      DBvhNode b;
      std::memset(&b, 0, sizeof(b));
      b.left = b.right = NREF_NONE;
      F.bvh.push_back(b);
      F.wnodes.push_back(DWNode{});
      init_wnode(F.wnodes.back());
      return (uint32_t)(F.bvh.size() - 1);
    };
    // a member of the tree being built; Groups are opened (explicit work list: Groups nest)
    struct Member { nref_t ref; Box6 box; };
    std::vector<Member> open;
    auto add_member = [&](nref_t ref, const Box6& box) {
      open.assign(1, Member{ref, box});
      while (!open.empty()) {
        const Member m = open.back();
        open.pop_back();
        if (NREF_KIND(m.ref) == NK_GROUP) {
          const DGroup& g = F.groups[NREF_INDEX(m.ref)];
          for (uint32_t j = g.count; j-- > 0;) {
            Member c;
            c.ref = F.gchild[g.first + j];
            for (int a = 0; a < 6; a++) c.box.v[a] = (double)F.gbox[6 * (size_t)(g.first + j) + a];
            open.push_back(c);
          }
          continue;
        }
        if (!(m.box.v[0] <= m.box.v[1])) continue;  // an empty box (an empty hierarchy, bvh.rs:23-26) holds nothing
        SahLeaf l;
        l.ref = m.ref;
        l.box = m.box;
        for (int a = 0; a < 3; a++) l.c[a] = 0.5 * (l.box.v[2 * a] + l.box.v[2 * a + 1]);
        leaves.push_back(l);
      }
    };
    // the tree over `leaves` (at least one), its root in slot slots[0], further nodes in the following slots and,
    // when those run out, in appended ones
    auto build_tree = [&]() {
      const uint32_t root = slots[0];
      if (leaves.size() == 1) {
        // (a hierarchy over a single member holds it as both children and visits it twice, bvh.rs:27-28; the second
        // visit finds what the first one found, and a ConstantMedium's two samples are taken in its one visit,
        // DMedium::tests)
        float b[6];
        set_cull_box(b, leaves[0].box);
        set_wide_child(F.wnodes[root], 0, leaves[0].ref, b);
        return;
      }
      // binary tree, explicit work list (a chain of uneven splits can be as deep as it is long)
      bin.clear();
      bin.push_back(BinNode{});
      jobs.clear();
      jobs.push_back(Job{0, leaves.size(), 0});
      while (!jobs.empty()) {
        const Job job = jobs.back();
        jobs.pop_back();
        Box6 bb = leaves[job.lo].box;
        for (size_t i = job.lo + 1; i < job.hi; i++) bb = box_merge(bb, leaves[i].box);
        bin[job.node].box = bb;
        if (job.hi - job.lo == 1) {
          bin[job.node].ref = leaves[job.lo].ref;
          continue;
        }
        const size_t mid = sah_split(leaves, job.lo, job.hi);
        const int l = (int)bin.size(), r = l + 1;
        bin.push_back(BinNode{});
        bin.push_back(BinNode{});
        bin[job.node].left = l;
        bin[job.node].right = r;
        jobs.push_back(Job{job.lo, mid, l});
        jobs.push_back(Job{mid, job.hi, r});
      }
      // breadth-first fold; wide node k of this tree takes the k-th smallest slot (top levels first)
      queue.clear();
      queue.push_back(WJob{0, root});
      size_t next_slot = 1;
      for (size_t q = 0; q < queue.size(); q++) {
        const WJob job = queue[q];
        int kids[4] = {bin[job.bin].left, bin[job.bin].right, -1, -1};
        int nk = 2;
        while (nk < 4) {
          int best = -1;
          double best_area = -1.0;
          for (int k = 0; k < nk; k++)
            if (bin[kids[k]].left >= 0) {
              const double a = box_area(bin[kids[k]].box);
              if (a > best_area) { best_area = a; best = k; }
            }
          if (best < 0) break;
          const int opened = kids[best];
          kids[best] = bin[opened].left;
          kids[nk++] = bin[opened].right;
        }
        for (int k = 0; k < nk; k++) {
          const BinNode& c = bin[kids[k]];
          nref_t ref;
          if (c.left < 0) {
            ref = c.ref;
          } else {
            const uint32_t slot = next_slot < slots.size() ? slots[next_slot++] : new_slot();
            ref = NREF(NK_BVH, slot);
            queue.push_back(WJob{kids[k], slot});
          }
          float b[6];
          set_cull_box(b, c.box);
          set_wide_child(F.wnodes[job.slot], k, ref, b);  // (indexed anew: new_slot() may have moved the table)
        }
      }
    };
    for (size_t root = 0; root < n0; root++) {
      if (!F.bvh_is_root[root]) continue;
      // members and table slots of this hierarchy: walk its reference nodes
      leaves.clear();
      slots.clear();
      stack.assign(1, (uint32_t)root);
      while (!stack.empty()) {
        uint32_t i = stack.back();
        stack.pop_back();
        slots.push_back(i);
        for (int side = 0; side < 2; side++) {
          const DBvhNode& b = F.bvh[i];
          nref_t c = side == 0 ? b.left : b.right;
          if (side == 1 && b.right == b.left) break;  // a single member is stored twice (bvh.rs:27-28)
          if (NREF_KIND(c) == NK_BVH && !F.bvh_is_root[NREF_INDEX(c)]) { stack.push_back(NREF_INDEX(c)); continue; }
          add_member(c, F.bvh_child_box[side][i]);
        }
      }
      std::sort(slots.begin(), slots.end());
      if (leaves.empty()) continue;  // nothing to hit: the root keeps its four unreachable slots
      build_tree();
    }
    // Groups that are not members of a hierarchy: the content of an instance, or the world
    auto tree_of_group = [&](nref_t ref) -> nref_t {
      if (NREF_KIND(ref) != NK_GROUP) return ref;
      leaves.clear();
      Box6 all{{-kInf, kInf, -kInf, kInf, -kInf, kInf}};
      add_member(ref, all);
      if (leaves.empty()) return ref;  // an empty Group: the ordered traversal passes over it
      slots.assign(1, new_slot());
      build_tree();
      return NREF(NK_BVH, slots[0]);
    };
    for (DTranslate& t : F.translates) t.ord_child = t.only_sphere != NREF_NONE ? t.enter_child : tree_of_group(t.enter_child);
    for (DRotate& t : F.rotates) t.ord_child = tree_of_group(t.enter_child);
    for (DScale& t : F.scales) t.ord_child = tree_of_group(t.enter_child);
    F.ord_root = tree_of_group(F.root);
    const size_t n = F.bvh.size();
    for (int k = 0; k < 2; k++) { F.bvh_child_box[k].clear(); F.bvh_child_box[k].shrink_to_fit(); }
    F.bvh_is_root.clear();
    // The slots that hold culling nodes move to the front of the tables (hierarchy roots keep one index in both
    // tables because both are renumbered alike), top levels first, so that the leading records are the hottest
    // ones: those are what the traversal kernel copies into LDS.
    std::vector<uint32_t> order;
    order.reserve(n);
    for (size_t i = 0; i < n; i++)
      if (F.wnodes[i].child[0] != NREF_NONE) order.push_back((uint32_t)i);
    F.n_wide = (uint32_t)order.size();
    for (size_t i = 0; i < n; i++)
      if (F.wnodes[i].child[0] == NREF_NONE) order.push_back((uint32_t)i);
    apply_bvh_order(F, order);
  }

  // Preorder dump in the order the reference's tree would be walked.
  void dump(int h, int32_t* kinds, int32_t* ids, double* boxes, size_t cap, size_t& count) const {
    const HNode& n = nodes[h];
    if (count < cap) {
      kinds[count] = n.kind;
      ids[count] = n.id;
      for (int i = 0; i < 6; i++) boxes[6 * count + i] = n.box.v[i];
    }
    count++;
    switch (n.kind) {
      case CHIP_NODE_BVH:
        dump(n.c0, kinds, ids, boxes, cap, count);
        dump(n.c1, kinds, ids, boxes, cap, count);
        break;
      case CHIP_NODE_GROUP:
      case CHIP_NODE_CONE:
      case CHIP_NODE_BOX:
        for (int k : n.kids) dump(k, kinds, ids, boxes, cap, count);
        break;
      case CHIP_NODE_TRANSLATE:
      case CHIP_NODE_ROTATE:
      case CHIP_NODE_SCALE:
      case CHIP_NODE_MEDIUM: dump(n.c0, kinds, ids, boxes, cap, count); break;
      default: break;
    }
  }
};

}  // namespace chip
