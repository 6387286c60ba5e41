// TEST INFRASTRUCTURE ONLY — CPU oracle for the caustic path tracer.
//
// This header is a plain f64 C++ restatement of the algorithm of the reference's
// crates/core render path (Camera::render -> ray_color -> Node::hit -> Material /
// Texture / ProbabilityDensityFunction).  It is the checker used by tests/, by
// __graft_entry__.smoke() and by bench.py's cpu_baseline leg.  Nothing in the
// product library links, includes or calls it.
//
// Pinning status: the reference's own executable assertions only cover the math
// value types (doctests in ray.rs, interval.rs, vector.rs, color.rs, matrix.rs,
// axis_aligned_bounding_box.rs, the sphere UV table in object/sphere.rs and the
// RNG range tests in random.rs); those are all re-checked in tests/test_oracle_kat.py.
// Nothing in the reference pins BVH layout, intersections, materials, PDFs or
// images, so above the math-type level this oracle is "parity unpinned": it follows
// the reference source line by line (citations on every function) but there are no
// reference-produced vectors to check it against, and no Rust toolchain here to
// produce them.
//
// Every function cites the reference file:line it restates (paths relative to
// crates/core/src/ unless stated otherwise).  Compile with -ffp-contract=off so
// the floating-point expression order below is what executes.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <stdexcept>
#include <vector>

namespace orc {

constexpr double kInf = std::numeric_limits<double>::infinity();
constexpr double kPi = 3.14159265358979323846;  // f64::consts::PI

// ---------------------------------------------------------------- vector.rs
struct Vec3 {
  double x = 0, y = 0, z = 0;
  Vec3() = default;
  Vec3(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
  double axis(int a) const { return a == 0 ? x : (a == 1 ? y : z); }  // vector.rs:316-322
  double& axis_mut(int a) { return a == 0 ? x : (a == 1 ? y : z); }
  // vector.rs:188-193
  double length_squared() const {
    double xs = x * x, ys = y * y, zs = z * z;
    return xs + ys + zs;
  }
  double length() const { return std::sqrt(length_squared()); }  // vector.rs:170-172
  // vector.rs:211-213
  double dot(const Vec3& o) const { return x * o.x + y * o.y + z * o.z; }
  // vector.rs:234-240
  Vec3 cross(const Vec3& o) const {
    return Vec3(y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x);
  }
  Vec3 unit() const;  // vector.rs:253-255
  // vector.rs:272-275
  bool is_near_zero() const {
    double s = 1e-8;
    return std::fabs(x) < s && std::fabs(y) < s && std::fabs(z) < s;
  }
  Vec3 reflect(const Vec3& n) const;                       // vector.rs:294-296
  Vec3 refract(const Vec3& n, double etai_over_etat) const;  // vector.rs:304-309
};
inline Vec3 operator*(const Vec3& v, double s) { return Vec3(v.x * s, v.y * s, v.z * s); }  // vector.rs:347-357
inline Vec3 operator*(double s, const Vec3& v) { return Vec3(v.x * s, v.y * s, v.z * s); }  // vector.rs:359-369
inline Vec3 operator/(const Vec3& v, double s) { return Vec3(v.x / s, v.y / s, v.z / s); }  // vector.rs:371-381
inline Vec3 operator+(const Vec3& a, const Vec3& b) { return Vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Vec3 operator-(const Vec3& a, const Vec3& b) { return Vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vec3 operator-(const Vec3& a) { return Vec3(-a.x, -a.y, -a.z); }
inline Vec3 Vec3::unit() const { return *this / length(); }
inline Vec3 Vec3::reflect(const Vec3& n) const { return *this - 2.0 * dot(n) * n; }
inline Vec3 Vec3::refract(const Vec3& n, double etai_over_etat) const {
  double cos_theta = std::fmin((-(*this)).dot(n), 1.0);
  Vec3 r_out_perp = etai_over_etat * (*this + cos_theta * n);
  Vec3 r_out_parallel = -std::sqrt(std::fabs(1.0 - r_out_perp.length_squared())) * n;
  return r_out_perp + r_out_parallel;
}

// ---------------------------------------------------------------- color.rs
struct Color {
  double r = 0, g = 0, b = 0;
  Color() = default;
  Color(double r_, double g_, double b_) : r(r_), g(g_), b(b_) {}
};
inline Color operator*(const Color& c, double s) { return Color(c.r * s, c.g * s, c.b * s); }  // color.rs:193-202
inline Color operator*(double s, const Color& c) { return Color(c.r * s, c.g * s, c.b * s); }
inline Color operator*(const Color& a, const Color& b) { return Color(a.r * b.r, a.g * b.g, a.b * b.b); }
inline Color operator/(const Color& c, double s) { return Color(c.r / s, c.g / s, c.b / s); }
inline Color operator+(const Color& a, const Color& b) { return Color(a.r + b.r, a.g + b.g, a.b + b.b); }
// f64::clamp keeps NaN (color.rs:162-168)
inline double rust_clamp(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }
inline Color color_clamp(const Color& c, double lo, double hi) {
  return Color(rust_clamp(c.r, lo, hi), rust_clamp(c.g, lo, hi), rust_clamp(c.b, lo, hi));
}
// color.rs:154-160
inline Color nan_to_zero(const Color& c) {
  return Color(std::isnan(c.r) ? 0.0 : c.r, std::isnan(c.g) ? 0.0 : c.g, std::isnan(c.b) ? 0.0 : c.b);
}
// color.rs:175-177
inline double linear_to_gamma1(double v) { return v > 0.0 ? std::sqrt(v) : 0.0; }
// color.rs:128-134
inline Color linear_to_gamma(const Color& c) {
  return Color(rust_clamp(linear_to_gamma1(c.r), 0.0, 0.999), rust_clamp(linear_to_gamma1(c.g), 0.0, 0.999),
               rust_clamp(linear_to_gamma1(c.b), 0.0, 0.999));
}

// ---------------------------------------------------------------- random.rs
// The Random seam (random.rs:3-7).  The reference's native stream (rand::random,
// random.rs:28-40) is unseeded, so only the distribution is reproducible; the
// helper forms below are the reference's own test double's (random.rs:169-176).
struct Random {
  virtual ~Random() = default;
  virtual double rand() = 0;
  // A uniform addressed by a key instead of by its position in the stream.  ConstantMedium draws its free-flight
  // sample this way (keyed by the medium's creation id), so the sample a medium gets does not depend on the order
  // in which a traversal visits the media.  Generators without random access fall back to the stream.
  virtual double rand_keyed(uint32_t) { return rand(); }
  double rand_interval(double mn, double mx) { return mn + (mx - mn) * rand(); }
  int64_t rand_int_interval(int64_t mn, int64_t mx) {
    int64_t range = mx - mn;
    return mn + (int64_t)(rand() * (double)range);
  }
};

// Value-list generator, the reference's MockRandom (random.rs:139-177).
struct ListRandom : Random {
  std::vector<double> values;
  size_t index = 0;
  double rand() override {
    double v = values[index % values.size()];
    index++;
    return v;
  }
  // random.rs:152-158
  static ListRandom with_length(size_t len) {
    ListRandom r;
    for (size_t i = 0; i < len; i++)
      r.values.push_back((double)((i * 1103515245ull + 12345ull) % 2147483648ull) / 2147483648.0);
    return r;
  }
};

// Counter-based Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy
// as 1, 2, 3", SC'11).  Counter = (pixel, stratum, bounce, block); key = seed.
// One block gives two 53-bit doubles.  This is the render-time stream shared by
// the oracle and the GPU path so that both consume identical uniforms.
inline void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int i = 0; i < 10; i++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

constexpr uint32_t kCameraBounce = 0xFFFFFFFFu;

struct PhiloxRandom : Random {
  uint32_t key[2] = {0, 0};
  uint32_t pixel = 0, stratum = 0, bounce = 0;
  uint32_t draw = 0;        // index of the next uniform within (pixel, stratum, bounce)
  uint32_t cached_block = 0xFFFFFFFFu;
  uint32_t cache[4] = {0, 0, 0, 0};
  void seed(uint64_t s) { key[0] = (uint32_t)s; key[1] = (uint32_t)(s >> 32); }
  void set_path(uint32_t px, uint32_t st) { pixel = px; stratum = st; set_bounce(kCameraBounce); }
  void set_bounce(uint32_t b) { bounce = b; draw = 0; cached_block = 0xFFFFFFFFu; }
  double rand() override {
    uint32_t block = draw >> 1;
    if (block != cached_block) {
      uint32_t ctr[4] = {pixel, stratum, bounce, block};
      philox4x32_10(ctr, key, cache);
      cached_block = block;
    }
    uint32_t lo = cache[(draw & 1) * 2], hi = cache[(draw & 1) * 2 + 1];
    draw++;
    uint64_t u = (((uint64_t)hi << 32) | lo) >> 11;
    return (double)u * (1.0 / 9007199254740992.0);
  }
  // first uniform of block 0x80000000 | key of the current (pixel, stratum, bounce); the stream is not advanced
  double rand_keyed(uint32_t k) override {
    uint32_t ctr[4] = {pixel, stratum, bounce, 0x80000000u | k};
    uint32_t out[4];
    philox4x32_10(ctr, key, out);
    uint64_t u = (((uint64_t)out[1] << 32) | out[0]) >> 11;
    return (double)u * (1.0 / 9007199254740992.0);
  }
};

struct RenderContext {  // lib.rs:34-36
  Random* random = nullptr;
  // Deterministic traversal counters (not in the reference; used to derive the
  // algorithmic bytes per ray segment).
  uint64_t n_bvh_visits = 0, n_prim_tests = 0, n_segments = 0;
  // How ConstantMedium::hit draws its free-flight sample: 0 = the nearest of its keyed samples (the form the device
  // shares, see ConstantMedium); 1 = one draw from the stream per test, literally constant_medium.rs:64 (its outcome
  // depends on the visiting order, so only the reference-order recursion can use it); 2 = a single keyed sample
  // whatever the number of tests, a deliberately wrong form that tests use as a negative control.
  int media_mode = 0;
  // With keyed samples: which boundary query of the enclosing medium is being answered (see ConstantMedium, nested
  // form); 0 outside of any boundary.
  uint32_t medium_ctx = 0;
};

// vector.rs:63-69
inline Vec3 random_interval_vec(Random& r, double mn, double mx) {
  double x = r.rand_interval(mn, mx);
  double y = r.rand_interval(mn, mx);
  double z = r.rand_interval(mn, mx);
  return Vec3(x, y, z);
}
// vector.rs:79-87
inline Vec3 random_unit(Random& r) {
  for (;;) {
    Vec3 p = random_interval_vec(r, -1.0, 1.0);
    double len_sq = p.length_squared();
    if (1e-160 < len_sq && len_sq <= 1.0) return p / std::sqrt(len_sq);
  }
}
// vector.rs:115-126
inline Vec3 random_in_unit_disk(Random& r) {
  for (;;) {
    double x = r.rand_interval(-1.0, 1.0);
    double y = r.rand_interval(-1.0, 1.0);
    Vec3 pt(x, y, 0.0);
    if (pt.length_squared() < 1.0) return pt;
  }
}
// vector.rs:136-146
inline Vec3 random_cosine_direction(Random& r) {
  double r1 = r.rand();
  double r2 = r.rand();
  double phi = 2.0 * kPi * r1;
  double x = std::cos(phi) * std::sqrt(r2);
  double y = std::sin(phi) * std::sqrt(r2);
  double z = std::sqrt(1.0 - r2);
  return Vec3(x, y, z);
}

// ---------------------------------------------------------------- ray.rs
struct Ray {
  Vec3 origin, direction;
  double time = 0.0;
  Ray() = default;
  Ray(const Vec3& o, const Vec3& d, double t = 0.0) : origin(o), direction(d), time(t) {}
  Vec3 at(double t) const { return origin + (t * direction); }  // ray.rs:113-115
};

// ---------------------------------------------------------------- interval.rs
struct Interval {
  double min = kInf, max = -kInf;  // EMPTY, interval.rs:31
  Interval() = default;
  Interval(double a, double b) : min(a), max(b) {}
  // interval.rs:77-82 (f64::min / f64::max)
  static Interval from_intervals(const Interval& a, const Interval& b) {
    return Interval(std::fmin(a.min, b.min), std::fmax(a.max, b.max));
  }
  bool contains(double x) const { return min <= x && x <= max; }   // interval.rs:103-105
  bool surrounds(double x) const { return min < x && x < max; }    // interval.rs:125-127
  Interval expand(double delta) const {                            // interval.rs:148-151
    double padding = delta / 2.0;
    return Interval(min - padding, max + padding);
  }
  double size() const { return max - min; }                        // interval.rs:163-165
  bool is_empty() const { return max <= min; }                     // interval.rs:192-194
  Interval shifted(double rhs) const { return Interval(min + rhs, max + rhs); }  // interval.rs:211-217
};
inline Interval universe() { return Interval(-kInf, kInf); }  // interval.rs:34

// ---------------------------------------------------------------- axis_aligned_bounding_box.rs
struct AABB {
  Interval x, y, z;
  const Interval& axis_interval(int a) const { return a == 0 ? x : (a == 1 ? y : z); }  // aabb.rs:159-165
  // aabb.rs:273-285
  static AABB pad_to_minimums(AABB b) {
    double delta = 0.0001;
    if (b.x.size() < delta) b.x = b.x.expand(delta);
    if (b.y.size() < delta) b.y = b.y.expand(delta);
    if (b.z.size() < delta) b.z = b.z.expand(delta);
    return b;
  }
  static AABB empty() { return pad_to_minimums(AABB{Interval(), Interval(), Interval()}); }  // aabb.rs:41-47
  // aabb.rs:89-107
  static AABB from_points(const Vec3& a, const Vec3& b) {
    AABB r;
    r.x = a.x <= b.x ? Interval(a.x, b.x) : Interval(b.x, a.x);
    r.y = a.y <= b.y ? Interval(a.y, b.y) : Interval(b.y, a.y);
    r.z = a.z <= b.z ? Interval(a.z, b.z) : Interval(b.z, a.z);
    return pad_to_minimums(r);
  }
  // aabb.rs:134-140
  static AABB from_boxes(const AABB& b1, const AABB& b2) {
    AABB r;
    r.x = Interval::from_intervals(b1.x, b2.x);
    r.y = Interval::from_intervals(b1.y, b2.y);
    r.z = Interval::from_intervals(b1.z, b2.z);
    return pad_to_minimums(r);
  }
  // aabb.rs:195-228
  bool hit(const Ray& ray, Interval ray_t) const {
    for (int axis = 0; axis < 3; axis++) {
      const Interval& ax = axis_interval(axis);
      double adinv = 1.0 / ray.direction.axis(axis);
      double t0 = (ax.min - ray.origin.axis(axis)) * adinv;
      double t1 = (ax.max - ray.origin.axis(axis)) * adinv;
      if (t0 < t1) {
        if (t0 > ray_t.min) ray_t.min = t0;
        if (t1 < ray_t.max) ray_t.max = t1;
      } else {
        if (t1 > ray_t.min) ray_t.min = t1;
        if (t0 < ray_t.max) ray_t.max = t0;
      }
      if (ray_t.max <= ray_t.min) return false;
    }
    return true;
  }
  // aabb.rs:246-258 (ties go to the later axis: strict '>')
  int longest_axis() const {
    if (x.size() > y.size()) return x.size() > z.size() ? 0 : 2;
    return y.size() > z.size() ? 1 : 2;
  }
  // aabb.rs:307-313 (no re-padding)
  AABB translated(const Vec3& o) const { return AABB{x.shifted(o.x), y.shifted(o.y), z.shifted(o.z)}; }
};

// ---------------------------------------------------------------- matrix.rs
struct Mat3 {
  double m[3][3];
  // matrix.rs:91-97
  Vec3 mul(const Vec3& v) const {
    return Vec3(m[0][0] * v.x + m[0][1] * v.y + m[0][2] * v.z, m[1][0] * v.x + m[1][1] * v.y + m[1][2] * v.z,
                m[2][0] * v.x + m[2][1] * v.y + m[2][2] * v.z);
  }
};

// ---------------------------------------------------------------- utils/orthonormal_basis.rs
struct ONB {
  Vec3 u, v, w;
  // orthonormal_basis.rs:57-68
  explicit ONB(const Vec3& normal) {
    w = normal.unit();
    Vec3 a = std::fabs(w.x) > 0.9 ? Vec3(0.0, 1.0, 0.0) : Vec3(1.0, 0.0, 0.0);
    v = w.cross(a).unit();
    u = w.cross(v);
  }
  // orthonormal_basis.rs:95-97
  Vec3 transform_to_local(const Vec3& p) const { return (p.x * u) + (p.y * v) + (p.z * w); }
};

// ---------------------------------------------------------------- utils/perlin.rs
struct Perlin {
  Vec3 rand_vec[256];
  int perm_x[256], perm_y[256], perm_z[256];
  // perlin.rs:225-229: i runs 254..=1, target in [0, i); index 255 is never touched.
  static void permute(Random& r, int* p) {
    for (int i = 254; i >= 1; i--) {
      int target = (int)r.rand_int_interval(0, i);
      std::swap(p[i], p[target]);
    }
  }
  // perlin.rs:209-217
  static void generate_perm(Random& r, int* p) {
    for (int i = 0; i < 256; i++) p[i] = i;
    permute(r, p);
  }
  // perlin.rs:60-76
  static Perlin create(Random& r) {
    Perlin p;
    for (int i = 0; i < 256; i++) p.rand_vec[i] = random_unit(r);
    generate_perm(r, p.perm_x);
    generate_perm(r, p.perm_y);
    generate_perm(r, p.perm_z);
    return p;
  }
  // perlin.rs:181-198
  static double trilinear(const Vec3 c[2][2][2], double u, double v, double w) {
    double uu = u * u * (3.0 - 2.0 * u);
    double vv = v * v * (3.0 - 2.0 * v);
    double ww = w * w * (3.0 - 2.0 * w);
    double acc = 0.0;
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 2; j++)
        for (int k = 0; k < 2; k++) {
          Vec3 weight_v(u - (double)i, v - (double)j, w - (double)k);
          acc += ((double)i * uu + (1.0 - (double)i) * (1.0 - uu)) * ((double)j * vv + (1.0 - (double)j) * (1.0 - vv)) *
                 ((double)k * ww + (1.0 - (double)k) * (1.0 - ww)) * c[i][j][k].dot(weight_v);
        }
    return acc;
  }
  // perlin.rs:101-124
  double noise(const Vec3& pt) const {
    double u = pt.x - std::floor(pt.x);
    double v = pt.y - std::floor(pt.y);
    double w = pt.z - std::floor(pt.z);
    int64_t i = (int64_t)std::floor(pt.x);
    int64_t j = (int64_t)std::floor(pt.y);
    int64_t k = (int64_t)std::floor(pt.z);
    Vec3 c[2][2][2];
    for (int di = 0; di < 2; di++)
      for (int dj = 0; dj < 2; dj++)
        for (int dk = 0; dk < 2; dk++) {
          int idx = perm_x[(i + di) & 255] ^ perm_y[(j + dj) & 255] ^ perm_z[(k + dk) & 255];
          c[di][dj][dk] = rand_vec[idx];
        }
    return trilinear(c, u, v, w);
  }
  // perlin.rs:151-163
  double turbulence(const Vec3& pt, uint32_t depth) const {
    double acc = 0.0;
    Vec3 temp_p = pt;
    double weight = 1.0;
    for (uint32_t d = 0; d < depth; d++) {
      acc += weight * noise(temp_p);
      weight *= 0.5;
      temp_p = temp_p * 2.0;
    }
    return std::fabs(acc);
  }
};

// ---------------------------------------------------------------- texture/*.rs
struct Texture {
  virtual ~Texture() = default;
  virtual Color value(double u, double v, const Vec3& pt) const = 0;  // texture/mod.rs:17-19
};
struct SolidColor : Texture {  // texture/solid_color.rs:14-17
  Color c;
  explicit SolidColor(const Color& c_) : c(c_) {}
  Color value(double, double, const Vec3&) const override { return c; }
};
struct CheckerTexture : Texture {  // texture/checker_texture.rs:13-35
  double inv_scale;
  std::shared_ptr<Texture> even, odd;
  CheckerTexture(double scale, std::shared_ptr<Texture> e, std::shared_ptr<Texture> o)
      : inv_scale(1.0 / scale), even(std::move(e)), odd(std::move(o)) {}
  Color value(double u, double v, const Vec3& pt) const override {
    int64_t xi = (int64_t)std::floor(inv_scale * pt.x);
    int64_t yi = (int64_t)std::floor(inv_scale * pt.y);
    int64_t zi = (int64_t)std::floor(inv_scale * pt.z);
    bool is_even = (xi + yi + zi) % 2 == 0;
    return is_even ? even->value(u, v, pt) : odd->value(u, v, pt);
  }
};
struct ImageTexture : Texture {  // texture/image_texture.rs:16-29 + image.rs:58-66
  uint32_t w, h;
  std::vector<uint8_t> rgb;
  ImageTexture(uint32_t w_, uint32_t h_, const uint8_t* data) : w(w_), h(h_), rgb(data, data + (size_t)w_ * h_ * 3) {}
  Color value(double u, double v, const Vec3&) const override {
    u = rust_clamp(u, 0.0, 1.0);
    v = 1.0 - rust_clamp(v, 0.0, 1.0);
    // Rust 'as u32' saturates; u,v are in [0,1] so the products are in range.
    uint32_t i = (uint32_t)(u * (double)w);
    uint32_t j = (uint32_t)(v * (double)h);
    if (i >= w || j >= h) return Color(0.0, 1.0, 1.0);
    const uint8_t* p = &rgb[((size_t)j * w + i) * 3];
    return Color((double)p[0] / 255.0, (double)p[1] / 255.0, (double)p[2] / 255.0);
  }
};
struct PerlinNoiseTexture : Texture {  // texture/perlin_noise.rs:18-22
  Perlin noise;
  double scale;
  Color value(double, double, const Vec3& pt) const override {
    return Color(1.0, 1.0, 1.0) * 0.5 * (1.0 + noise.noise(scale * pt));
  }
};
struct PerlinTurbulenceTexture : Texture {  // texture/perlin_turbulence.rs:20-27
  Perlin noise;
  double scale;
  uint32_t depth;
  Color value(double, double, const Vec3& pt) const override {
    return Color(0.5, 0.5, 0.5) * (1.0 + std::sin(scale * pt.z + 10.0 * noise.turbulence(pt, depth)));
  }
};

// ---------------------------------------------------------------- object/mod.rs
struct Material;
struct HitRecord {  // object/mod.rs:31-39
  Vec3 pt, normal;
  double t = 0, u = 0, v = 0;
  bool front_face = false;
  const Material* material = nullptr;
  int leaf_id = -1;  // creation id of the node that produced the record (not in the reference; for tests)
  // object/mod.rs:44-51
  void set_face_normal(const Ray& ray, const Vec3& outward_normal) {
    front_face = ray.direction.dot(outward_normal) < 0.0;
    normal = front_face ? outward_normal : -outward_normal;
  }
};

enum NodeKind { K_BVH = 0, K_SPHERE = 1, K_QUAD = 2, K_GROUP = 3, K_TRANSLATE = 4, K_ROTATE = 5, K_MEDIUM = 6, K_BOX = 7, K_DISC = 8, K_CONE = 9, K_CONE_WALL = 10, K_SCALE = 11 };

struct Node {  // object/mod.rs:54-68
  int id = -1;  // creation id; -1 for nodes made internally (BVH interior, box faces)
  virtual ~Node() = default;
  virtual int kind() const = 0;
  virtual bool hit(RenderContext& ctx, const Ray& ray, Interval ray_t, HitRecord& rec) const = 0;
  virtual const AABB& bounding_box() const = 0;
  virtual double pdf_value(RenderContext&, const Vec3&, const Vec3&) const { return 0.0; }
  virtual Vec3 random(RenderContext&, const Vec3&) const { return Vec3(1.0, 0.0, 0.0); }
  // Adds v to the number of times one world.hit reaches every ConstantMedium below this node.
  virtual void count_tests(int) {}
  // A scene graph is a DAG (the reference shares Arc<dyn Node>s), and the reference draws a fresh free-flight
  // sample every time a ray reaches a ConstantMedium, whichever way it came.  To address those samples by key
  // (see ConstantMedium) every place a medium occupies in the world needs an identity of its own, so the world is
  // expanded into a tree wherever it holds media: instantiate() returns the node itself for a subtree without
  // media and a copy with instantiated children otherwise; number_media() then hands every medium of the
  // expanded tree, in depth-first order, the keys of its samples.
  virtual bool holds_medium() const { return false; }
  virtual std::shared_ptr<Node> instantiate(const std::shared_ptr<Node>& self) const { return self; }
  virtual void number_media(uint32_t&) {}
};
using NodeP = std::shared_ptr<Node>;

// ---------------------------------------------------------------- probability_density_function/*.rs
struct Pdf {
  virtual ~Pdf() = default;
  virtual double value(RenderContext& ctx, const Vec3& direction) const = 0;
  virtual Vec3 generate(RenderContext& ctx) const = 0;
};
struct CosinePdf : Pdf {  // cosine.rs:10-27
  ONB uvw;
  explicit CosinePdf(const Vec3& w) : uvw(w) {}
  double value(RenderContext&, const Vec3& direction) const override {
    double cosine_theta = direction.unit().dot(uvw.w);
    double v = cosine_theta / kPi;
    return std::fmax(v, 0.0);
  }
  Vec3 generate(RenderContext& ctx) const override { return uvw.transform_to_local(random_cosine_direction(*ctx.random)); }
};
struct SpherePdf : Pdf {  // probability_density_function/sphere.rs:20-26
  double value(RenderContext&, const Vec3&) const override { return 1.0 / (4.0 * kPi); }
  Vec3 generate(RenderContext& ctx) const override { return random_unit(*ctx.random); }
};
struct HittablePdf : Pdf {  // hittable.rs:11-23
  const Node* objects;
  Vec3 origin;
  HittablePdf(const Node* o, const Vec3& org) : objects(o), origin(org) {}
  double value(RenderContext& ctx, const Vec3& direction) const override { return objects->pdf_value(ctx, origin, direction); }
  Vec3 generate(RenderContext& ctx) const override { return objects->random(ctx, origin); }
};
struct MixturePdf : Pdf {  // mixture.rs:20-32
  const Pdf *pdf0, *pdf1;
  MixturePdf(const Pdf* a, const Pdf* b) : pdf0(a), pdf1(b) {}
  double value(RenderContext& ctx, const Vec3& direction) const override {
    double v0 = 0.5 * pdf0->value(ctx, direction);
    double v1 = 0.5 * pdf1->value(ctx, direction);
    return v0 + v1;
  }
  Vec3 generate(RenderContext& ctx) const override {
    if (ctx.random->rand() < 0.5) return pdf0->generate(ctx);
    return pdf1->generate(ctx);
  }
};

// ---------------------------------------------------------------- material/*.rs
struct ScatterResult {  // material/mod.rs:37-45
  Color attenuation;
  bool is_ray = false;
  Ray ray;
  std::unique_ptr<Pdf> pdf;
};
enum MaterialKind { M_LAMBERTIAN = 0, M_METAL = 1, M_DIELECTRIC = 2, M_DIFFUSE_LIGHT = 3, M_ISOTROPIC = 4, M_EMPTY = 5 };
struct Material {  // material/mod.rs:19-35
  virtual ~Material() = default;
  virtual int kind() const = 0;
  virtual bool scatter(RenderContext& ctx, const Ray& r_in, const HitRecord& hit, ScatterResult& out) const = 0;
  virtual Color emitted(const Ray&, const HitRecord&, double, double, const Vec3&) const { return Color(0, 0, 0); }
  virtual double scattering_pdf(RenderContext&, const Ray&, const HitRecord&, const Ray&) const { return 0.0; }
};
struct Lambertian : Material {  // material/lambertian.rs:28-50
  std::shared_ptr<Texture> texture;
  explicit Lambertian(std::shared_ptr<Texture> t) : texture(std::move(t)) {}
  int kind() const override { return M_LAMBERTIAN; }
  bool scatter(RenderContext&, const Ray&, const HitRecord& hit, ScatterResult& out) const override {
    out.attenuation = texture->value(hit.u, hit.v, hit.pt);
    out.is_ray = false;
    out.pdf.reset(new CosinePdf(hit.normal));
    return true;
  }
  double scattering_pdf(RenderContext&, const Ray&, const HitRecord& hit, const Ray& scattered) const override {
    double cos_theta = hit.normal.dot(scattered.direction.unit());
    return cos_theta < 0.0 ? 0.0 : cos_theta / kPi;
  }
};
struct Metal : Material {  // material/metal.rs:19-28
  Color albedo;
  double fuzz;
  Metal(const Color& a, double f) : albedo(a), fuzz(f) {}
  int kind() const override { return M_METAL; }
  bool scatter(RenderContext& ctx, const Ray& r_in, const HitRecord& hit, ScatterResult& out) const override {
    Vec3 reflected = r_in.direction.reflect(hit.normal);
    reflected = reflected.unit() + (fuzz * random_unit(*ctx.random));
    out.attenuation = albedo;
    out.is_ray = true;
    out.ray = Ray(hit.pt, reflected, r_in.time);
    return true;
  }
};
struct Dielectric : Material {  // material/dielectric.rs:16-52
  double refraction_index;
  explicit Dielectric(double ri) : refraction_index(ri) {}
  int kind() const override { return M_DIELECTRIC; }
  static double powi5(double x) {  // f64::powi(5): x^5 by repeated multiplication
    double x2 = x * x;
    double x4 = x2 * x2;
    return x4 * x;
  }
  static double reflectance(double cosine, double ri) {
    double r0 = (1.0 - ri) / (1.0 + ri);
    r0 = r0 * r0;
    return r0 + (1.0 - r0) * powi5(1.0 - cosine);
  }
  bool scatter(RenderContext& ctx, const Ray& r_in, const HitRecord& hit, ScatterResult& out) const override {
    double ri = hit.front_face ? 1.0 / refraction_index : refraction_index;
    Vec3 unit_direction = r_in.direction.unit();
    double cos_theta = std::fmin((-unit_direction).dot(hit.normal), 1.0);
    double sin_theta = std::sqrt(1.0 - cos_theta * cos_theta);
    bool cannot_refract = ri * sin_theta > 1.0;
    // short-circuit: rand() is drawn only when refraction is possible
    Vec3 direction = (cannot_refract || reflectance(cos_theta, ri) > ctx.random->rand())
                         ? unit_direction.reflect(hit.normal)
                         : unit_direction.refract(hit.normal, ri);
    out.attenuation = Color(1.0, 1.0, 1.0);
    out.is_ray = true;
    out.ray = Ray(hit.pt, direction, r_in.time);
    return true;
  }
};
struct DiffuseLight : Material {  // material/diffuse_light.rs:27-44
  std::shared_ptr<Texture> texture;
  explicit DiffuseLight(std::shared_ptr<Texture> t) : texture(std::move(t)) {}
  int kind() const override { return M_DIFFUSE_LIGHT; }
  bool scatter(RenderContext&, const Ray&, const HitRecord&, ScatterResult&) const override { return false; }
  Color emitted(const Ray&, const HitRecord& hit, double u, double v, const Vec3& pt) const override {
    return hit.front_face ? texture->value(u, v, pt) : Color(0, 0, 0);
  }
};
struct Isotropic : Material {  // material/isotropic.rs:28-45
  std::shared_ptr<Texture> texture;
  explicit Isotropic(std::shared_ptr<Texture> t) : texture(std::move(t)) {}
  int kind() const override { return M_ISOTROPIC; }
  bool scatter(RenderContext&, const Ray&, const HitRecord& hit, ScatterResult& out) const override {
    out.attenuation = texture->value(hit.u, hit.v, hit.pt);
    out.is_ray = false;
    out.pdf.reset(new SpherePdf());
    return true;
  }
  double scattering_pdf(RenderContext&, const Ray&, const HitRecord&, const Ray&) const override {
    return 1.0 / (4.0 / kPi);
  }
};
struct EmptyMaterial : Material {  // material/empty.rs:18-27
  int kind() const override { return M_EMPTY; }
  bool scatter(RenderContext&, const Ray&, const HitRecord&, ScatterResult&) const override { return false; }
};

// ---------------------------------------------------------------- object/sphere.rs
struct Sphere : Node {
  Ray center;  // origin + time * direction
  double radius;
  std::shared_ptr<Material> material;
  AABB bbox;
  // sphere.rs:21-29
  Sphere(const Vec3& c, double r, std::shared_ptr<Material> m) : center(c, Vec3(0, 0, 0)), radius(r), material(std::move(m)) {
    Vec3 rv(r, r, r);
    bbox = AABB::from_points(c - rv, c + rv);
  }
  // sphere.rs:31-47
  void set_direction(const Vec3& d) {
    center = Ray(center.origin, d);
    Vec3 rvec(radius, radius, radius);
    AABB box1 = AABB::from_points(center.at(0.0) - rvec, center.at(0.0) + rvec);
    AABB box2 = AABB::from_points(center.at(1.0) - rvec, center.at(1.0) + rvec);
    bbox = AABB::from_boxes(box1, box2);
  }
  // sphere.rs:70-83
  static void get_uv(const Vec3& pt, double& u, double& v) {
    double theta = std::acos(-pt.y);
    double phi = std::atan2(-pt.z, pt.x) + kPi;
    u = phi / (2.0 * kPi);
    v = theta / kPi;
  }
  // sphere.rs:85-95
  static Vec3 random_to_sphere(Random& random, double radius, double distance_squared) {
    double r1 = random.rand();
    double r2 = random.rand();
    double z = 1.0 + r2 * (std::sqrt(1.0 - radius * radius / distance_squared) - 1.0);
    double phi = 2.0 * kPi * r1;
    double x = std::cos(phi) * std::sqrt(1.0 - z * z);
    double y = std::sin(phi) * std::sqrt(1.0 - z * z);
    return Vec3(x, y, z);
  }
  int kind() const override { return K_SPHERE; }
  // sphere.rs:99-137
  bool hit(RenderContext& ctx, const Ray& ray, Interval ray_t, HitRecord& rec) const override {
    ctx.n_prim_tests++;
    Vec3 current_center = center.at(ray.time);
    Vec3 oc = current_center - ray.origin;
    double a = ray.direction.length_squared();
    double h = ray.direction.dot(oc);
    double c = oc.length_squared() - radius * radius;
    double discriminant = h * h - a * c;
    if (discriminant < 0.0) return false;
    double sqrt_discriminant = std::sqrt(discriminant);
    double root = (h - sqrt_discriminant) / a;
    if (!ray_t.surrounds(root)) {
      root = (h + sqrt_discriminant) / a;
      if (!ray_t.surrounds(root)) return false;
    }
    double t = root;
    Vec3 pt = ray.at(t);
    Vec3 outward_normal = (pt - current_center) / radius;
    rec.pt = pt;
    rec.t = t;
    get_uv(outward_normal, rec.u, rec.v);
    rec.material = material.get();
    rec.leaf_id = id;
    rec.set_face_normal(ray, outward_normal);
    return true;
  }
  const AABB& bounding_box() const override { return bbox; }
  // sphere.rs:143-159
  double pdf_value(RenderContext& ctx, const Vec3& origin, const Vec3& direction) const override {
    HitRecord rec;
    if (!hit(ctx, Ray(origin, direction), Interval(0.001, kInf), rec)) return 0.0;
    double dist_squared = (center.at(0.0) - origin).length_squared();
    double cos_theta_max = std::sqrt(1.0 - radius * radius / dist_squared);
    double solid_angle = 2.0 * kPi * (1.0 - cos_theta_max);
    return 1.0 / solid_angle;
  }
  // sphere.rs:161-170
  Vec3 random(RenderContext& ctx, const Vec3& origin) const override {
    Vec3 direction = center.at(0.0) - origin;
    double distance_squared = direction.length_squared();
    ONB uvw(direction);
    return uvw.transform_to_local(random_to_sphere(*ctx.random, radius, distance_squared));
  }
};

// ---------------------------------------------------------------- object/quad.rs
struct Quad : Node {
  Vec3 q, u, v;
  std::shared_ptr<Material> material;
  AABB bbox;
  Vec3 normal;
  double d;
  Vec3 w;
  double area;
  // quad.rs:59-76, 92-96
  Quad(const Vec3& q_, const Vec3& u_, const Vec3& v_, std::shared_ptr<Material> m) : q(q_), u(u_), v(v_), material(std::move(m)) {
    Vec3 n = u.cross(v);
    normal = n.unit();
    d = normal.dot(q);
    w = n / n.dot(n);
    AABB d1 = AABB::from_points(q, q + u + v);
    AABB d2 = AABB::from_points(q + u, q + v);
    bbox = AABB::from_boxes(d1, d2);
    area = n.length();
  }
  int kind() const override { return K_QUAD; }
  // quad.rs:144-183 (+ is_interior quad.rs:112-122)
  bool hit(RenderContext& ctx, const Ray& ray, Interval ray_t, HitRecord& rec) const override {
    ctx.n_prim_tests++;
    double denom = normal.dot(ray.direction);
    if (std::fabs(denom) < 1e-8) return false;
    double t = (d - normal.dot(ray.origin)) / denom;
    if (!ray_t.contains(t)) return false;
    Vec3 intersection = ray.at(t);
    Vec3 planar = intersection - q;
    double alpha = w.dot(planar.cross(v));
    double beta = w.dot(u.cross(planar));
    Interval unit_interval(0.0, 1.0);
    if (!unit_interval.contains(alpha) || !unit_interval.contains(beta)) return false;
    rec.pt = intersection;
    rec.t = t;
    rec.u = alpha;
    rec.v = beta;
    rec.material = material.get();
    rec.leaf_id = id;
    rec.set_face_normal(ray, normal);
    return true;
  }
  const AABB& bounding_box() const override { return bbox; }
  // quad.rs:210-226
  double pdf_value(RenderContext& ctx, const Vec3& origin, const Vec3& direction) const override {
    HitRecord rec;
    if (!hit(ctx, Ray(origin, direction), Interval(0.001, kInf), rec)) return 0.0;
    double distance_squared = rec.t * rec.t * direction.length_squared();
    double cosine = std::fabs(direction.dot(rec.normal) / direction.length());
    return distance_squared / (cosine * area);
  }
  // quad.rs:242-245 (first draw scales u, second scales v)
  Vec3 random(RenderContext& ctx, const Vec3& origin) const override {
    double r1 = ctx.random->rand();
    double r2 = ctx.random->rand();
    Vec3 p = q + (r1 * u) + (r2 * v);
    return p - origin;
  }
};

// ---------------------------------------------------------------- object/group.rs
struct Group : Node {
  std::vector<NodeP> nodes;
  AABB bbox = AABB::empty();  // group.rs:15-20
  // group.rs:30-34
  void push(NodeP n) {
    AABB nb = n->bounding_box();
    nodes.push_back(std::move(n));
    bbox = AABB::from_boxes(bbox, nb);
  }
  int kind() const override { return K_GROUP; }
  void count_tests(int v) override { for (const NodeP& n : nodes) n->count_tests(v); }
  bool holds_medium() const override {
    for (const NodeP& n : nodes)
      if (n->holds_medium()) return true;
    return false;
  }
  NodeP instantiate(const NodeP& self) const override {
    if (!holds_medium()) return self;
    auto c = std::make_shared<Group>(*this);
    for (NodeP& n : c->nodes) n = n->instantiate(n);
    return c;
  }
  void number_media(uint32_t& next) override { for (const NodeP& n : nodes) n->number_media(next); }
  // group.rs:44-54
  bool hit(RenderContext& ctx, const Ray& ray, Interval ray_t, HitRecord& rec) const override {
    bool any = false;
    HitRecord tmp;
    for (const NodeP& n : nodes) {
      if (n->hit(ctx, ray, ray_t, tmp)) {
        ray_t.max = tmp.t;
        rec = tmp;
        any = true;
      }
    }
    return any;
  }
  const AABB& bounding_box() const override { return bbox; }
  // group.rs:60-69
  double pdf_value(RenderContext& ctx, const Vec3& origin, const Vec3& direction) const override {
    double weight = 1.0 / (double)nodes.size();
    double sum = 0.0;
    for (const NodeP& n : nodes) sum += weight * n->pdf_value(ctx, origin, direction);
    return sum;
  }
  // group.rs:71-78
  Vec3 random(RenderContext& ctx, const Vec3& origin) const override {
    if (nodes.empty()) return Vec3(0.0, 1.0, 0.0);
    size_t r = (size_t)ctx.random->rand_int_interval(0, (int64_t)nodes.size());
    return nodes[r]->random(ctx, origin);
  }
};

// ---------------------------------------------------------------- object/box_node.rs
struct BoxPrimitive : Node {
  Group group;
  // box_node.rs:15-74
  BoxPrimitive(const Vec3& a, const Vec3& b, const std::shared_ptr<Material>& m) {
    Vec3 mn(std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z));
    Vec3 mx(std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z));
    Vec3 dx(mx.x - mn.x, 0.0, 0.0);
    Vec3 dy(0.0, mx.y - mn.y, 0.0);
    Vec3 dz(0.0, 0.0, mx.z - mn.z);
    group.push(std::make_shared<Quad>(Vec3(mn.x, mn.y, mx.z), dx, dy, m));   // front
    group.push(std::make_shared<Quad>(Vec3(mx.x, mn.y, mx.z), -dz, dy, m));  // right
    group.push(std::make_shared<Quad>(Vec3(mx.x, mn.y, mn.z), -dx, dy, m));  // back
    group.push(std::make_shared<Quad>(Vec3(mn.x, mn.y, mn.z), dz, dy, m));   // left
    group.push(std::make_shared<Quad>(Vec3(mn.x, mx.y, mx.z), dx, -dz, m));  // top
    group.push(std::make_shared<Quad>(Vec3(mn.x, mn.y, mn.z), dx, dz, m));   // bottom
  }
  int kind() const override { return K_BOX; }
  // box_node.rs:78-80
  bool hit(RenderContext& ctx, const Ray& ray, Interval ray_t, HitRecord& rec) const override {
    bool h = group.hit(ctx, ray, ray_t, rec);
    if (h) rec.leaf_id = id;
    return h;
  }
  const AABB& bounding_box() const override { return group.bounding_box(); }
};

// ---------------------------------------------------------------- object/disc.rs
struct Disc : Node {
  Vec3 center;
  double radius;
  Vec3 normal;
  std::shared_ptr<Material> material;
  AABB bbox;
  // disc.rs:24-54
  Disc(const Vec3& c, double r, const Vec3& n, std::shared_ptr<Material> m) : center(c), radius(r), normal(n), material(std::move(m)) {
    double radius_y = std::fabs(n.y) > 0.9 ? 0.0 : r;
    double radius_x = std::fabs(n.x) > 0.9 ? 0.0 : r;
    double radius_z = std::fabs(n.z) > 0.9 ? 0.0 : r;
    double delta = 1e-4;
    Vec3 ext(radius_x + std::fabs(n.x) * delta, radius_y + std::fabs(n.y) * delta, radius_z + std::fabs(n.z) * delta);
    bbox = AABB::from_points(c - ext, c + ext);
  }
  int kind() const override { return K_DISC; }
  // disc.rs:111-150 (+ get_uv disc.rs:57-66)
  bool hit(RenderContext& ctx, const Ray& ray, Interval ray_t, HitRecord& rec) const override {
    ctx.n_prim_tests++;
    double denominator = ray.direction.dot(normal);
    if (std::fabs(denominator) < 1e-8) return false;
    double t = (center - ray.origin).dot(normal) / denominator;
    if (!ray_t.contains(t)) return false;
    Vec3 pt = ray.at(t);
    Vec3 v = pt - center;
    if (v.length_squared() > radius * radius) return false;
    Vec3 local_pt = pt - center;
    rec.pt = pt;
    rec.t = t;
    rec.u = (local_pt.x / radius + 1.0) * 0.5;
    rec.v = (local_pt.z / radius + 1.0) * 0.5;
    rec.material = material.get();
    rec.leaf_id = id;
    rec.set_face_normal(ray, normal);
    return true;
  }
  const AABB& bounding_box() const override { return bbox; }
  // disc.rs:156-186
  double pdf_value(RenderContext& ctx, const Vec3& origin, const Vec3& direction) const override {
    HitRecord rec;
    if (!hit(ctx, Ray(origin, direction), Interval(1e-4, kInf), rec)) return 0.0;
    double dist_squared = rec.t * rec.t * direction.length_squared();
    double cosine_abs = std::fabs(rec.normal.dot(-direction));
    if (cosine_abs < 1e-8) return 0.0;
    double area = kPi * radius * radius;
    return dist_squared / (cosine_abs * area);
  }
  // disc.rs:188-193 with random_on_disc disc.rs:68-95
  Vec3 random(RenderContext& ctx, const Vec3& origin) const override {
    double r_sq = std::sqrt(ctx.random->rand());
    double phi = 2.0 * kPi * ctx.random->rand();
    double x = r_sq * std::cos(phi) * radius;
    double y = r_sq * std::sin(phi) * radius;
    ONB uvw(normal);
    Vec3 random_local_pt = uvw.u * x + uvw.v * y;
    Vec3 target = center + random_local_pt;
    return target - origin;
  }
};

// ---------------------------------------------------------------- object/cone.rs (the side wall)
struct ConeFrustumWall : Node {
  Vec3 base;
  double height, r0, r1;  // r0 bottom radius, r1 top radius
  std::shared_ptr<Material> material;
  AABB bbox;
  // cone.rs:116-147
  ConeFrustumWall(const Vec3& b, double h, double r1_, double r0_, std::shared_ptr<Material> m)
      : base(b), height(h), r0(r0_), r1(r1_), material(std::move(m)) {
    double max_radius = std::fmax(r0, r1);
    Vec3 min_p(b.x - max_radius, b.y, b.z - max_radius);
    Vec3 max_p(b.x + max_radius, b.y + h, b.z + max_radius);
    bbox = AABB::from_points(min_p, max_p);
  }
  int kind() const override { return K_CONE_WALL; }
  // cone.rs:164-288
  bool hit(RenderContext& ctx, const Ray& ray, Interval ray_t, HitRecord& rec) const override {
    ctx.n_prim_tests++;
    double h = height;
    double base_y = base.y;
    Vec3 local_origin = ray.origin - base;
    double ox = local_origin.x, oy = local_origin.y, oz = local_origin.z;
    double dx = ray.direction.x, dy = ray.direction.y, dz = ray.direction.z;
    double y_local = oy, dy_local = dy;
    double dr = r1 - r0;
    double k = dr / h;
    double a = dx * dx + dz * dz - k * k * dy_local * dy_local;
    double b = 2.0 * (ox * dx + oz * dz - k * k * y_local * dy_local - k * r0 * dy_local);
    double c = ox * ox + oz * oz - (r0 * r0 + 2.0 * k * r0 * y_local + k * k * y_local * y_local);
    if (std::fabs(a) < 1e-8) return false;
    double discriminant = b * b - 4.0 * a * c;
    if (discriminant < 0.0) return false;
    double sqrt_discriminant = std::sqrt(discriminant);
    double t = (-b - sqrt_discriminant) / (2.0 * a);
    if (!ray_t.contains(t)) {
      t = (-b + sqrt_discriminant) / (2.0 * a);
      if (!ray_t.contains(t)) return false;
    }
    Vec3 pt = ray.at(t);
    double hit_y = pt.y;
    double y_min = base_y, y_max = base_y + h;
    if (hit_y < y_min || hit_y > y_max) {
      double t_other = (-b + sqrt_discriminant) / (2.0 * a);
      if (t_other == t) return false;
      if (!ray_t.contains(t_other)) return false;
      Vec3 pt_other = ray.at(t_other);
      if (pt_other.y < y_min || pt_other.y > y_max) return false;
      t = t_other;
    }
    pt = ray.at(t);
    double hit_y_local = pt.y - base_y;
    double hit_radius = r0 + k * hit_y_local;
    Vec3 pt_local = pt - base;
    Vec3 outward_normal = Vec3(pt_local.x, k * hit_radius, pt_local.z).unit();
    // get_uv(pt_local, 0.0, h), cone.rs:149-160
    double phi = std::atan2(pt_local.z, pt_local.x);
    double u = (phi + kPi) / (2.0 * kPi);
    double v = rust_clamp((pt_local.y - 0.0) / h, 0.0, 1.0);
    rec.pt = pt;
    rec.t = t;
    rec.u = u;
    rec.v = v;
    rec.material = material.get();
    rec.leaf_id = id;
    rec.set_face_normal(ray, outward_normal);
    return true;
  }
  const AABB& bounding_box() const override { return bbox; }
  // cone.rs:296-313: 1 / lateral area whatever the direction
  double pdf_value(RenderContext&, const Vec3&, const Vec3&) const override {
    double dr = r1 - r0;
    double l = std::sqrt(height * height + dr * dr);
    double area = kPi * (r0 + r1) * l;
    return area > 1e-8 ? 1.0 / area : 0.0;
  }
  // cone.rs:315-347: returns a point of the wall, not a direction from the origin
  Vec3 random(RenderContext& ctx, const Vec3&) const override {
    double u1 = ctx.random->rand_interval(0.0, 1.0);
    double u2 = ctx.random->rand_interval(0.0, 1.0);
    double phi = 2.0 * kPi * u1;
    double r_rand_sq = r0 * r0 + u2 * (r1 * r1 - r0 * r0);
    double r_rand = std::sqrt(r_rand_sq);
    double y_local = std::fabs(r1 - r0) < 1e-8 ? height * u2 : height / (r1 - r0) * (r_rand - r0);
    Vec3 p_local(r_rand * std::cos(phi), y_local, r_rand * std::sin(phi));
    return p_local + base;
  }
};

// ---------------------------------------------------------------- object/cone.rs (the frustum)
struct ConeFrustum : Node {
  Group group;
  // cone.rs:19-75: top disc, bottom disc (each only if its radius > 1e-4), side wall
  ConeFrustum(const Vec3& base, double height, double top_radius, double bottom_radius, const std::shared_ptr<Material>& m) {
    double y_base = base.y, y_top = base.y + height;
    if (top_radius > 1e-4) group.push(std::make_shared<Disc>(Vec3(base.x, y_top, base.z), top_radius, Vec3(0.0, 1.0, 0.0), m));
    if (bottom_radius > 1e-4) group.push(std::make_shared<Disc>(Vec3(base.x, y_base, base.z), bottom_radius, Vec3(0.0, -1.0, 0.0), m));
    group.push(std::make_shared<ConeFrustumWall>(base, height, top_radius, bottom_radius, m));
  }
  int kind() const override { return K_CONE; }
  bool hit(RenderContext& ctx, const Ray& ray, Interval ray_t, HitRecord& rec) const override {
    bool h = group.hit(ctx, ray, ray_t, rec);
    if (h) rec.leaf_id = id;
    return h;
  }
  const AABB& bounding_box() const override { return group.bounding_box(); }
  double pdf_value(RenderContext& ctx, const Vec3& origin, const Vec3& direction) const override { return group.pdf_value(ctx, origin, direction); }
  Vec3 random(RenderContext& ctx, const Vec3& origin) const override { return group.random(ctx, origin); }
};

// ---------------------------------------------------------------- object/translate.rs
struct Translate : Node {
  NodeP object;
  Vec3 offset;
  AABB bbox;
  // translate.rs:15-22
  Translate(NodeP o, const Vec3& off) : object(std::move(o)), offset(off) { bbox = object->bounding_box().translated(off); }
  int kind() const override { return K_TRANSLATE; }
  void count_tests(int v) override { object->count_tests(v); }
  bool holds_medium() const override { return object->holds_medium(); }
  NodeP instantiate(const NodeP& self) const override {
    if (!holds_medium()) return self;
    auto c = std::make_shared<Translate>(*this);
    c->object = object->instantiate(object);
    return c;
  }
  void number_media(uint32_t& next) override { object->number_media(next); }
  // translate.rs:26-37
  bool hit(RenderContext& ctx, const Ray& ray, Interval ray_t, HitRecord& rec) const override {
    Ray offset_r(ray.origin - offset, ray.direction, ray.time);
    if (!object->hit(ctx, offset_r, ray_t, rec)) return false;
    rec.pt = rec.pt + offset;
    return true;
  }
  const AABB& bounding_box() const override { return bbox; }
};

// ---------------------------------------------------------------- object/rotate.rs
struct Rotate : Node {
  NodeP object;
  Mat3 rot, inv;
  AABB bbox;
  // rotate.rs:19-79
  Rotate(NodeP o, const Vec3& axis_in, double angle_deg) : object(std::move(o)) {
    double radians = angle_deg * (kPi / 180.0);  // f64::to_radians
    // two libm calls as the source has them: a compiler that sees sin and cos of one value calls sincos instead,
    // whose cosine is not always cos()'s to the last place (56.301153430308204 degrees: one ulp apart in glibc)
    volatile double radians_again = radians;
    double sin_theta = std::sin(radians);
    double cos_theta = std::cos(radians_again);
    Vec3 axis = axis_in.unit();
    double x = axis.x, y = axis.y, z = axis.z;
    double omc = 1.0 - cos_theta;
    rot.m[0][0] = cos_theta + x * x * omc;
    rot.m[0][1] = x * y * omc - z * sin_theta;
    rot.m[0][2] = x * z * omc + y * sin_theta;
    rot.m[1][0] = y * x * omc + z * sin_theta;
    rot.m[1][1] = cos_theta + y * y * omc;
    rot.m[1][2] = y * z * omc - x * sin_theta;
    rot.m[2][0] = z * x * omc - y * sin_theta;
    rot.m[2][1] = z * y * omc + x * sin_theta;
    rot.m[2][2] = cos_theta + z * z * omc;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) inv.m[i][j] = rot.m[j][i];
    bbox = compute_bounding_box(object->bounding_box(), rot);
  }
  // rotate.rs:96-131
  static AABB compute_bounding_box(const AABB& ob, const Mat3& rm) {
    Vec3 mn(kInf, kInf, kInf), mx(-kInf, -kInf, -kInf);
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 2; j++)
        for (int k = 0; k < 2; k++) {
          double i_f = (double)i, j_f = (double)j, k_f = (double)k;
          double x = i_f * ob.x.max + (1.0 - i_f) * ob.x.min;
          double y = j_f * ob.y.max + (1.0 - j_f) * ob.y.min;
          double z = k_f * ob.z.max + (1.0 - k_f) * ob.z.min;
          Vec3 rotated = rm.mul(Vec3(x, y, z));
          for (int a = 0; a < 3; a++) {
            mn.axis_mut(a) = std::fmin(mn.axis(a), rotated.axis(a));
            mx.axis_mut(a) = std::fmax(mx.axis(a), rotated.axis(a));
          }
        }
    return AABB::from_points(mn, mx);
  }
  int kind() const override { return K_ROTATE; }
  void count_tests(int v) override { object->count_tests(v); }
  bool holds_medium() const override { return object->holds_medium(); }
  NodeP instantiate(const NodeP& self) const override {
    if (!holds_medium()) return self;
    auto c = std::make_shared<Rotate>(*this);
    c->object = object->instantiate(object);
    return c;
  }
  void number_media(uint32_t& next) override { object->number_media(next); }
  // rotate.rs:135-149
  bool hit(RenderContext& ctx, const Ray& ray, Interval ray_t, HitRecord& rec) const override {
    Ray rotated_r(inv.mul(ray.origin), inv.mul(ray.direction), ray.time);
    if (!object->hit(ctx, rotated_r, ray_t, rec)) return false;
    rec.pt = rot.mul(rec.pt);
    rec.normal = rot.mul(rec.normal);
    return true;
  }
  const AABB& bounding_box() const override { return bbox; }
};

// ---------------------------------------------------------------- object/scale.rs
struct Scale : Node {
  NodeP object;
  Mat3 scale, inv;
  AABB bbox;
  // scale.rs:18-60
  Scale(NodeP o, double sx, double sy, double sz) : object(std::move(o)) {
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) scale.m[i][j] = inv.m[i][j] = 0.0;
    scale.m[0][0] = sx; scale.m[1][1] = sy; scale.m[2][2] = sz;
    inv.m[0][0] = std::fabs(sx) > 1e-9 ? 1.0 / sx : kInf;
    inv.m[1][1] = std::fabs(sy) > 1e-9 ? 1.0 / sy : kInf;
    inv.m[2][2] = std::fabs(sz) > 1e-9 ? 1.0 / sz : kInf;
    // scale.rs:62-91
    const AABB& ob = object->bounding_box();
    Interval ix(std::fmin(ob.x.min * sx, ob.x.max * sx), std::fmax(ob.x.min * sx, ob.x.max * sx));
    Interval iy(std::fmin(ob.y.min * sy, ob.y.max * sy), std::fmax(ob.y.min * sy, ob.y.max * sy));
    Interval iz(std::fmin(ob.z.min * sz, ob.z.max * sz), std::fmax(ob.z.min * sz, ob.z.max * sz));
    bbox = AABB::pad_to_minimums(AABB{ix, iy, iz});
  }
  int kind() const override { return K_SCALE; }
  void count_tests(int v) override { object->count_tests(v); }
  bool holds_medium() const override { return object->holds_medium(); }
  NodeP instantiate(const NodeP& self) const override {
    if (!holds_medium()) return self;
    auto c = std::make_shared<Scale>(*this);
    c->object = object->instantiate(object);
    return c;
  }
  void number_media(uint32_t& next) override { object->number_media(next); }
  // scale.rs:95-117: t is kept; the normal goes through the inverse matrix and is re-normalised
  bool hit(RenderContext& ctx, const Ray& ray, Interval ray_t, HitRecord& rec) const override {
    Ray scaled_r(inv.mul(ray.origin), inv.mul(ray.direction), ray.time);
    if (!object->hit(ctx, scaled_r, ray_t, rec)) return false;
    rec.pt = scale.mul(rec.pt);
    rec.normal = inv.mul(rec.normal);
    rec.normal = rec.normal.unit();
    return true;
  }
  const AABB& bounding_box() const override { return bbox; }
};

// ---------------------------------------------------------------- object/constant_medium.rs
struct ConstantMedium : Node {
  NodeP boundary;
  double neg_inv_density;
  std::shared_ptr<Material> phase_function;
  // constant_medium.rs:19-37
  ConstantMedium(NodeP b, double density, std::shared_ptr<Texture> tex)
      : boundary(std::move(b)), neg_inv_density(-1.0 / density), phase_function(std::make_shared<Isotropic>(std::move(tex))) {}
  int kind() const override { return K_MEDIUM; }
  // The reference draws a fresh free-flight sample on every test and keeps the nearest hit; one world.hit tests
  // a medium more than once when a BoundingVolumeHierarchy holds it as both children (bvh.rs:22-26).  `tests`
  // is that count (set by count_tests from the world root); hit() takes the nearest of `tests` keyed samples
  // every time it is called, which has the same distribution as the reference's sequence of tests and makes
  // the outcome independent of the visiting order.  The samples are addressed by sample_base .. sample_base +
  // tests - 1, numbers no other medium of the (expanded, see Node::instantiate) world shares, so media met along
  // one ray draw independently, as they do from the reference's stream.
  int tests = 0;
  uint32_t sample_base = 0;
  // A boundary may hold media of its own (constant_medium.rs:19-29 takes any node).  Every test of this medium then
  // asks the boundary twice, and each query makes the media inside draw afresh, so the tests cannot be folded into one
  // "nearest of `tests` samples": with keyed samples hit() runs them one after the other over the shrinking interval,
  // as the reference's recursion does, and a medium inside the boundary owns a run of its `tests` keys per query
  // (`queries` = 2 x this medium's tests; query q of test k is context 2k + q, RenderContext::medium_ctx).  One level
  // of nesting, like the device.
  bool nested = false;
  uint32_t queries = 1;  // for a medium inside a boundary: boundary queries per world.hit of the enclosing medium
  static uint32_t& numbering_queries() { static thread_local uint32_t q = 0; return q; }
  void count_tests(int v) override { tests += v; }
  bool holds_medium() const override { return true; }
  NodeP instantiate(const NodeP&) const override {
    auto c = std::make_shared<ConstantMedium>(*this);
    c->tests = 0;
    if (boundary->holds_medium()) {
      c->boundary = boundary->instantiate(boundary);
      c->nested = true;
    }
    return c;
  }
  void number_media(uint32_t& next) override {
    const uint32_t outer = numbering_queries();
    queries = outer ? outer : 1u;
    sample_base = next;
    next += (uint32_t)tests * queries;
    if (nested) {
      if (outer) throw std::runtime_error("a ConstantMedium whose boundary holds media, inside another medium's boundary, is not supported (one level of nesting)");
      boundary->count_tests(1);
      numbering_queries() = 2u * (uint32_t)tests;
      boundary->number_media(next);
      numbering_queries() = outer;
    }
  }
  // constant_medium.rs:41-80
  bool hit(RenderContext& ctx, const Ray& ray, Interval ray_t, HitRecord& rec) const override {
    const bool one_by_one = nested && ctx.media_mode == 0;
    const int reps = one_by_one ? tests : 1;
    const uint32_t outer_ctx = ctx.medium_ctx;
    bool any = false;
    for (int rep = 0; rep < reps; rep++) {
      HitRecord hit1, hit2;
      if (one_by_one) ctx.medium_ctx = 2u * (uint32_t)rep;
      const bool got1 = boundary->hit(ctx, ray, universe(), hit1);
      if (one_by_one) ctx.medium_ctx = 2u * (uint32_t)rep + 1u;
      const bool got2 = got1 && boundary->hit(ctx, ray, Interval(hit1.t + 0.0001, kInf), hit2);
      ctx.medium_ctx = outer_ctx;
      if (!got2) continue;
      if (hit1.t < ray_t.min) hit1.t = ray_t.min;
      if (hit2.t > ray_t.max) hit2.t = ray_t.max;
      if (hit1.t >= hit2.t) continue;
      if (hit1.t < 0.0) hit1.t = 0.0;
      double ray_length = ray.direction.length();
      double distance_inside_boundary = (hit2.t - hit1.t) * ray_length;
      double u;
      if (ctx.media_mode == 1) {
        u = ctx.random->rand();  // constant_medium.rs:64
      } else {
        const uint32_t key0 = sample_base + outer_ctx * (uint32_t)tests;
        u = ctx.random->rand_keyed(one_by_one ? key0 + (uint32_t)rep : key0);
        if (ctx.media_mode == 0 && !one_by_one)
          for (int j = 1; j < tests; j++) u = std::fmax(u, ctx.random->rand_keyed(key0 + (uint32_t)j));
      }
      double hit_distance = neg_inv_density * std::log(u);
      if (hit_distance > distance_inside_boundary) continue;
      double t = hit1.t + hit_distance / ray_length;
      rec.pt = ray.at(t);
      rec.normal = Vec3(1.0, 0.0, 0.0);
      rec.t = t;
      rec.u = 0.0;
      rec.v = 0.0;
      rec.front_face = true;
      rec.material = phase_function.get();
      rec.leaf_id = id;
      ray_t.max = t;
      any = true;
    }
    return any;
  }
  const AABB& bounding_box() const override { return boundary->bounding_box(); }
};

// ---------------------------------------------------------------- object/bounding_volume_hierarchy.rs
struct BVH : Node {
  NodeP left, right;
  AABB bbox;
  // bvh.rs:16-46
  explicit BVH(const std::vector<NodeP>& nodes_in) {
    AABB span = AABB::empty();
    for (const NodeP& n : nodes_in) span = AABB::from_boxes(span, n->bounding_box());
    if (nodes_in.empty()) {
      left = std::make_shared<Group>();
      right = std::make_shared<Group>();
    } else if (nodes_in.size() == 1) {
      left = nodes_in[0];
      right = nodes_in[0];
    } else if (nodes_in.size() == 2) {
      left = nodes_in[0];
      right = nodes_in[1];
    } else {
      int axis = span.longest_axis();
      std::vector<NodeP> nodes = nodes_in;
      // slice::sort_by is stable; comparator is f64::total_cmp on bbox.axis.min (bvh.rs:87-91)
      std::stable_sort(nodes.begin(), nodes.end(), [axis](const NodeP& a, const NodeP& b) {
        return total_less(a->bounding_box().axis_interval(axis).min, b->bounding_box().axis_interval(axis).min);
      });
      size_t mid = nodes.size() / 2;
      left = std::make_shared<BVH>(std::vector<NodeP>(nodes.begin(), nodes.begin() + mid));
      right = std::make_shared<BVH>(std::vector<NodeP>(nodes.begin() + mid, nodes.end()));
    }
    bbox = AABB::from_boxes(left->bounding_box(), right->bounding_box());
  }
  // f64::total_cmp: IEEE totalOrder (-0 < +0, NaNs at the ends)
  static bool total_less(double a, double b) {
    int64_t ia, ib;
    std::memcpy(&ia, &a, 8);
    std::memcpy(&ib, &b, 8);
    ia ^= (int64_t)((uint64_t)(ia >> 63) >> 1);
    ib ^= (int64_t)((uint64_t)(ib >> 63) >> 1);
    return ia < ib;
  }
  int kind() const override { return K_BVH; }
  void count_tests(int v) override { left->count_tests(v); right->count_tests(v); }  // a single object is both children
  bool holds_medium() const override { return left->holds_medium() || right->holds_medium(); }
  NodeP instantiate(const NodeP& self) const override {
    if (!holds_medium()) return self;
    auto c = std::make_shared<BVH>(*this);
    c->left = left->instantiate(left);
    c->right = right == left ? c->left : right->instantiate(right);  // one object held twice stays one object
    return c;
  }
  void number_media(uint32_t& next) override {
    left->number_media(next);
    if (right != left) right->number_media(next);
  }
  // bvh.rs:58-76
  bool hit(RenderContext& ctx, const Ray& ray, Interval ray_t, HitRecord& rec) const override {
    ctx.n_bvh_visits++;
    if (!bbox.hit(ray, ray_t)) return false;
    HitRecord hl, hr;
    bool hit_left = left->hit(ctx, ray, ray_t, hl);
    double t = ray_t.max;
    if (hit_left) t = hl.t;
    bool hit_right = right->hit(ctx, ray, Interval(ray_t.min, t), hr);
    if (hit_right) {
      rec = hr;
      return true;
    }
    if (hit_left) {
      rec = hl;
      return true;
    }
    return false;
  }
  const AABB& bounding_box() const override { return bbox; }
};

// ---------------------------------------------------------------- camera.rs
struct CameraDesc {  // CameraBuilder fields, camera.rs:34-89 (defaults camera.rs:106-120)
  double vertical_fov = 90.0;
  double aspect_ratio = 1.0;
  uint32_t image_width = 100;
  double look_from[3] = {0, 0, 0};
  double look_at[3] = {0, 0, -1};
  double up[3] = {0, 1, 0};
  double defocus_angle = 0.0;
  double focus_distance = 10.0;
  uint32_t samples_per_pixel = 10;
  uint32_t max_depth = 10;
  double background[3] = {0, 0, 0};
};

struct Camera {
  uint32_t image_width, image_height, max_depth, sqrt_spp;
  Vec3 center, pixel00_loc, pixel_delta_u, pixel_delta_v, defocus_disk_u, defocus_disk_v;
  double pixel_samples_scale, defocus_angle, reciprocal_sqrt_spp;
  Color background;
  // camera.rs:126-182
  explicit Camera(const CameraDesc& b) {
    uint32_t ih = (uint32_t)((double)b.image_width / b.aspect_ratio);
    image_height = ih < 1 ? 1 : ih;
    image_width = b.image_width;
    sqrt_spp = (uint32_t)std::sqrt((double)b.samples_per_pixel);
    pixel_samples_scale = 1.0 / (double)(sqrt_spp * sqrt_spp);
    reciprocal_sqrt_spp = 1.0 / (double)sqrt_spp;
    Vec3 look_from(b.look_from[0], b.look_from[1], b.look_from[2]);
    Vec3 look_at(b.look_at[0], b.look_at[1], b.look_at[2]);
    Vec3 up(b.up[0], b.up[1], b.up[2]);
    center = look_from;
    double theta = b.vertical_fov * (kPi / 180.0);
    double h = std::tan(theta / 2.0);
    double viewport_height = 2.0 * h * b.focus_distance;
    double viewport_width = viewport_height * ((double)image_width / (double)image_height);
    Vec3 w = (look_from - look_at).unit();
    Vec3 u = up.cross(w).unit();
    Vec3 v = w.cross(u);
    Vec3 viewport_u = viewport_width * u;
    Vec3 viewport_v = viewport_height * -v;
    pixel_delta_u = viewport_u / (double)image_width;
    pixel_delta_v = viewport_v / (double)image_height;
    Vec3 viewport_upper_left = center - (b.focus_distance * w) - viewport_u / 2.0 - viewport_v / 2.0;
    pixel00_loc = viewport_upper_left + 0.5 * (pixel_delta_u + pixel_delta_v);
    double defocus_radius = b.focus_distance * std::tan((b.defocus_angle / 2.0) * (kPi / 180.0));
    defocus_disk_u = u * defocus_radius;
    defocus_disk_v = v * defocus_radius;
    max_depth = b.max_depth;
    defocus_angle = b.defocus_angle;
    background = Color(b.background[0], b.background[1], b.background[2]);
  }
  // camera.rs:389-394
  Vec3 sample_square_stratified(Random& random, uint32_t s_x, uint32_t s_y) const {
    double px = (((double)s_x + random.rand()) * reciprocal_sqrt_spp) - 0.5;
    double py = (((double)s_y + random.rand()) * reciprocal_sqrt_spp) - 0.5;
    return Vec3(px, py, 0.0);
  }
  // camera.rs:416-419
  Vec3 defocus_disk_sample(Random& random) const {
    Vec3 pt = random_in_unit_disk(random);
    return center + (pt.x * defocus_disk_u) + (pt.y * defocus_disk_v);
  }
  // camera.rs:359-374
  Ray get_ray(RenderContext& ctx, uint32_t x, uint32_t y, uint32_t s_x, uint32_t s_y) const {
    Vec3 offset = sample_square_stratified(*ctx.random, s_x, s_y);
    Vec3 pixel_sample = pixel00_loc + (((double)x + offset.x) * pixel_delta_u) + (((double)y + offset.y) * pixel_delta_v);
    Vec3 ray_origin = defocus_angle <= 0.0 ? center : defocus_disk_sample(*ctx.random);
    Vec3 ray_direction = pixel_sample - ray_origin;
    double ray_time = ctx.random->rand();
    return Ray(ray_origin, ray_direction, ray_time);
  }
  // camera.rs:250-308.  set_bounce keys the counter-based stream on the bounce
  // index (max_depth - depth); all draws of a bounce happen before the recursion.
  Color ray_color(RenderContext& ctx, const Ray& ray, uint32_t depth, const Node& world, const Node* lights) const {
    if (depth == 0) return Color(0, 0, 0);
    if (auto* pr = dynamic_cast<PhiloxRandom*>(ctx.random)) pr->set_bounce(max_depth - depth);
    ctx.n_segments++;
    HitRecord hit;
    if (!world.hit(ctx, ray, Interval(0.001, kInf), hit)) return background;
    Color color_from_emission = hit.material->emitted(ray, hit, hit.u, hit.v, hit.pt);
    ScatterResult sr;
    if (!hit.material->scatter(ctx, ray, hit, sr)) return color_from_emission;
    if (sr.is_ray) return sr.attenuation * ray_color(ctx, sr.ray, depth - 1, world, lights);
    std::unique_ptr<HittablePdf> light_pdf;
    std::unique_ptr<MixturePdf> mix;
    const Pdf* pdf = sr.pdf.get();
    if (lights) {
      light_pdf.reset(new HittablePdf(lights, hit.pt));
      mix.reset(new MixturePdf(light_pdf.get(), sr.pdf.get()));
      pdf = mix.get();
    }
    Ray scattered(hit.pt, pdf->generate(ctx), ray.time);
    double pdf_value = pdf->value(ctx, scattered.direction);
    if (pdf_value < 0.05) return color_from_emission;
    double scattering_pdf = hit.material->scattering_pdf(ctx, ray, hit, scattered);
    Color sample_color = ray_color(ctx, scattered, depth - 1, world, lights);
    Color color_from_scatter = (sr.attenuation * scattering_pdf * sample_color) / pdf_value;
    Color color = color_from_emission + color_from_scatter;
    return color_clamp(color, 0.0, 10.0);
  }
  // One (pixel, stratum) sample: camera.rs:337-338
  Color sample(RenderContext& ctx, uint32_t x, uint32_t y, uint32_t s_x, uint32_t s_y, const Node& world, const Node* lights) const {
    if (auto* pr = dynamic_cast<PhiloxRandom*>(ctx.random)) pr->set_path(y * image_width + x, s_y * sqrt_spp + s_x);
    Ray r = get_ray(ctx, x, y, s_x, s_y);
    return ray_color(ctx, r, max_depth, world, lights);
  }
  // camera.rs:324-345 restricted to the strata {offset, offset+stride, ...}; returns the
  // un-normalised linear sum (the caller scales, zeroes NaNs and gamma-encodes).
  Color render_sum(RenderContext& ctx, uint32_t x, uint32_t y, const Node& world, const Node* lights, uint32_t stratum_offset,
                   uint32_t stratum_stride) const {
    Color pixel_color(0, 0, 0);
    for (uint32_t s_y = 0; s_y < sqrt_spp; s_y++)
      for (uint32_t s_x = 0; s_x < sqrt_spp; s_x++) {
        uint32_t s = s_y * sqrt_spp + s_x;
        if (s % stratum_stride != stratum_offset) continue;
        Color c = sample(ctx, x, y, s_x, s_y, world, lights);
        pixel_color = pixel_color + c;
      }
    return pixel_color;
  }
  // camera.rs:343-344
  Color resolve(const Color& sum) const { return linear_to_gamma(pixel_samples_scale * nan_to_zero(sum)); }
};

// Forward-only restatement of ray_color: each bounce is the per-channel map
// x -> clamp(e + w*x, lo_k, hi_k); the path carries the composition
// x -> clamp(a + b*x, lo, hi).  Used to check on the CPU, in f64, that the
// iterative estimator the GPU uses equals the recursion of camera.rs:250-308.
struct AffineClamp {
  double a[3] = {0, 0, 0}, b[3] = {1, 1, 1}, lo[3], hi[3];
  static constexpr double kBig = 1e30;
  AffineClamp() { for (int i = 0; i < 3; i++) { lo[i] = -kBig; hi[i] = kBig; } }
  void compose(const double e[3], const double w[3], double klo, double khi) {
    for (int i = 0; i < 3; i++) {
      double p = a[i] + b[i] * klo, q = a[i] + b[i] * khi;
      double nlo = rust_clamp(p < q ? p : q, lo[i], hi[i]);
      double nhi = rust_clamp(p < q ? q : p, lo[i], hi[i]);
      a[i] = a[i] + b[i] * e[i];
      b[i] = b[i] * w[i];
      lo[i] = nlo;
      hi[i] = nhi;
    }
  }
  Color apply(const Color& x) const {
    return Color(rust_clamp(a[0] + b[0] * x.r, lo[0], hi[0]), rust_clamp(a[1] + b[1] * x.g, lo[1], hi[1]),
                 rust_clamp(a[2] + b[2] * x.b, lo[2], hi[2]));
  }
};

inline Color ray_color_iterative(const Camera& cam, RenderContext& ctx, Ray ray, const Node& world, const Node* lights) {
  AffineClamp f;
  for (uint32_t bounce = 0;; bounce++) {
    if (bounce == cam.max_depth) return f.apply(Color(0, 0, 0));
    if (auto* pr = dynamic_cast<PhiloxRandom*>(ctx.random)) pr->set_bounce(bounce);
    HitRecord hit;
    if (!world.hit(ctx, ray, Interval(0.001, kInf), hit)) return f.apply(cam.background);
    Color E = hit.material->emitted(ray, hit, hit.u, hit.v, hit.pt);
    ScatterResult sr;
    if (!hit.material->scatter(ctx, ray, hit, sr)) return f.apply(E);
    if (sr.is_ray) {
      double e[3] = {0, 0, 0}, w[3] = {sr.attenuation.r, sr.attenuation.g, sr.attenuation.b};
      f.compose(e, w, -AffineClamp::kBig, AffineClamp::kBig);
      ray = sr.ray;
      continue;
    }
    std::unique_ptr<HittablePdf> light_pdf;
    std::unique_ptr<MixturePdf> mix;
    const Pdf* pdf = sr.pdf.get();
    if (lights) {
      light_pdf.reset(new HittablePdf(lights, hit.pt));
      mix.reset(new MixturePdf(light_pdf.get(), sr.pdf.get()));
      pdf = mix.get();
    }
    Ray scattered(hit.pt, pdf->generate(ctx), ray.time);
    double pdf_value = pdf->value(ctx, scattered.direction);
    if (pdf_value < 0.05) return f.apply(E);
    double sp = hit.material->scattering_pdf(ctx, ray, hit, scattered);
    double e[3] = {E.r, E.g, E.b};
    double w[3] = {sr.attenuation.r * sp / pdf_value, sr.attenuation.g * sp / pdf_value, sr.attenuation.b * sp / pdf_value};
    f.compose(e, w, 0.0, 10.0);
    ray = scattered;
  }
}

}  // namespace orc
