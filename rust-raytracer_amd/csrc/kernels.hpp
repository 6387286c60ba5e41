// gfx950 kernels of the wavefront path tracer: ray generation, closest-hit
// traversal, material-binned shading with cosine / light / mixture importance
// sampling, per-pixel accumulation and the final resolve.
//
// Geometry runs in f64 with the reference's expression order (build with
// -ffp-contract=off), so box / sphere / quad decisions are the same as
// crates/core's for the same ray.  Citations give the reference file:line a
// device function restates (paths relative to crates/core/src/).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_scene.h"

#define CHIP_BLOCK 256
#define CHIP_STACK_LDS 16  // per-lane short stack entries kept in LDS
#define CHIP_STACK_OVF 48  // further entries spill to a per-thread global slab
#ifndef CHIP_TRAV_PROFILE
#define CHIP_TRAV_PROFILE 0  // 1: the node / primitive counters count loop trips per wave instead of visits per lane (lane-utilisation study)
#endif
#ifndef CHIP_SHADE_PRIO
#define CHIP_SHADE_PRIO 1  // wave priority of the shade kernel: its few instructions issue ahead of a co-resident traversal kernel's, so its memory requests leave early
#endif
#ifndef CHIP_REFILL_THRESH_B0
#define CHIP_REFILL_THRESH_B0 1  // the same for camera rays: never, so that a wave always computes 64 camera rays at a time
#endif
#ifndef CHIP_REFILL_THRESH_B0_INSTANCES
#define CHIP_REFILL_THRESH_B0_INSTANCES 32  // ... except among instanced spheres, whose long descents from memory leave the early finishers idle for too long (lattice +4 %)
#endif
// The descent loop of the ordered traversal goes on while at least 1 / DEN of the wave's lanes that still have a ray
// are at culling nodes (see traverse); 0 = until none is.  Measured per scene family: bare spheres gain the most
// from a small share, sphere instances from a larger one, the variant with media a little, and the one for rotated
// quads without media loses from leaving the loop early at all.
#ifndef CHIP_DESCENT_DEN_SPHERES
#define CHIP_DESCENT_DEN_SPHERES 8u
#endif
#ifndef CHIP_DESCENT_DEN_INSTANCES
#define CHIP_DESCENT_DEN_INSTANCES 4u
#endif
#ifndef CHIP_DESCENT_DEN_MEDIA
#define CHIP_DESCENT_DEN_MEDIA 8u
#endif
// Waves per SIMD the shade kernels are register-allocated for.  Lean kernels (spheres, solid colours, no lights): 4
// (128 VGPRs, no spills; one such wave per SIMD still fits beside the sphere traversal's four 96-VGPR waves; 5 and 6
// spill and measure 1-3 % slower).  Full kernel (every material, texture and light kind): whatever it needs (about
// 250).  Its form without image / noise textures needs 130-141 on its own: 4 waves cost it a few spills at bounce 0.
#ifndef CHIP_SHADE_LEAN_WAVES
#define CHIP_SHADE_LEAN_WAVES 4
#endif
#ifndef CHIP_SHADE_FULL_WAVES
#define CHIP_SHADE_FULL_WAVES 1
#endif
#ifndef CHIP_SHADE_PLAIN_WAVES
#define CHIP_SHADE_PLAIN_WAVES 4
#endif
#ifndef CHIP_REFILL_THRESH
#define CHIP_REFILL_THRESH 48  // a traversing wave refills its idle lanes once fewer than this many are busy
#endif
#define CHIP_PI 3.14159265358979323846
#define CHIP_BIG 1e30f
// Per-bounce counter row: [0] live paths, [1 .. MK_COUNT] material queue lengths, then the cursors the
// persistent traversal waves reserve queue chunks through and the literal queue's length.
#define CHIP_CTR_STRIDE 16
#define CHIP_CTR_CURSOR 7
#define CHIP_CTR_LIT 8
#define CHIP_CTR_LIT_CURSOR 9
#define CHIP_CTR_RICH 10  // entries the plain-texture shade pass left to the pass with the image / noise texture code

// ------------------------------------------------------------------ path pool
// The live paths of a bounce are kept dense: entry i of a bounce is one small record per consumer
// (array of structures) at index i, and shading writes the records of the paths that go on to the next
// free indices of the other half of a ping-pong pair.  HBM moves 128-byte granules, so records gathered
// from a sparse set of slots would waste half of every fetch (measured: 64-byte records read and written
// at a stride of two or more reach 2.5 TB/s, dense ones 3.5 to 4.7 TB/s); the path's identity (its index
// in the batch, which fixes its pixel, stratum and random stream) travels in the ray record instead.
struct alignas(64) PathRay {  // what traversal reads and shading writes
  double o[3], d[3], time;
  uint32_t id;   // path index in the batch, CHIP_DEAD_ID for a reserved entry that stayed empty
  uint32_t pad;
};
// In the default shading mode a ray record is 48 bytes: origin (3 x f64), direction as three floats, path id, time (f64),
// one after the other at a 48-byte stride in the same array.  Scattered directions of that mode come from f32
// trigonometry and f32-seeded reciprocals in the first place; rounding them to f32 where they are stored (once, in the
// kernel that makes them: the traversal and the shading of the next bounce both read the rounded value back) takes a
// quarter of the bytes a continuing path moves per bounce.  With CHIP_RENDER_EXACT_SHADING the record is PathRay.
#define CHIP_RAY48_BYTES 48
#define CHIP_DEAD_ID 0xFFFFFFFFu
#define NREF_DEAD NREF(NK_NONE, 0x0FFFFFFE)  // PathHit.node of an empty entry
struct alignas(16) PathHit {  // what traversal writes: closest hit of the current segment
  double t;
  uint32_t node;  // node reference, NREF_NONE = miss
  uint32_t info;  // bits 0..23: the hit primitive's mat word (material index + kind); bits 24..31 (bounce 0 only):
                  // defocus-disk candidates the camera ray consumed
};
#define PATHHIT_TRIES(info) ((info) >> 24)
// The composed radiance map of the path so far, x -> b x before the first bounce that clamps (camera.rs:304) and
// x -> clamp(b x, min(0, bound), max(0, bound)) after it, and the path id.  One end of the clamp interval is always
// zero (the clamp of camera.rs:304 is [0, 10] and every later bounce only scales and clips it), so the record holds the
// other end and one bit instead of two ends per channel.
#define CHIP_MAP_CLAMPED 0x80000000u  // in PathMap::id: a clamping bounce has happened (path ids stay below 2^31)
struct alignas(32) PathMap {
  uint32_t id;
  float b[3], bound[3];
  uint32_t pad;
};
struct PathPool {
  PathRay* ray[2];  // [cap] each: entries of even / odd bounces
  PathMap* st[2];   // [cap] each
  PathHit* hit;     // [cap] by entry of the current bounce
  float4* sample;   // [cap] by path id: finished sample radiance (rgb, w unused)
  uint32_t* q_lit;      // [cap] entries whose ray the literal traversal must trace
  uint32_t* counters;   // [max_depth+2][CHIP_CTR_STRIDE]: [b][0] entries of bounce b, then the cursors of the traversal kernels
  uint32_t* errflag;    // [4]: [0] traversal stack overflow
  unsigned long long* stats;  // [4]: segments, bvh visits, prim tests, spare
  uint32_t* ovf;        // [ovf_slabs][2 * CHIP_STACK_OVF] traversal stack spill (value, distance pairs), one slab per thread of a launch
  uint32_t cap;         // path slots (what `sample` holds and a batch may hold paths)
  uint32_t slack;       // entries the per-bounce arrays (ray, st, hit, q_lit) hold beyond cap: a bounce may carry empty entries beside its paths
  uint32_t ovf_slabs;
};

struct BatchDesc {
  uint32_t rect_x0, rect_y0, rect_w, rect_h;  // pixel rectangle
  uint32_t npix;                              // rect_w * rect_h
  uint32_t n_strata;                          // strata in this batch
  uint32_t stratum_first, stratum_stride;     // global stratum of batch-local sb: first + sb*stride
  uint32_t n_paths;                           // npix * n_strata
  uint32_t npix_magic[2], rect_w_magic[2], sqrt_spp_magic[2];  // ceil(2^64 / divisor), low and high word (0, 0 for a divisor of 1)
  // Order of a stratum's paths over the rectangle: 8 x 8 pixel tiles, tile after tile (tiles_per_row = rect_w / 8), when
  // both sides of the rectangle are multiples of 8; row after row (tiles_per_row = 0) otherwise.  A wave's 64 camera
  // rays then cover a compact patch of the image, and so do the paths that are still alive later.
  uint32_t tiles_per_row, tiles_magic[2];
  uint32_t seed_lo, seed_hi;
  uint32_t reference_order;                   // trace every ray in the reference's own visiting order
  uint32_t ray48;                             // ray records are the 48-byte form (the default shading mode)
};

// ------------------------------------------------------------------ Philox4x32-10 stream
// Counter (pixel, stratum, bounce, block), key = seed; each block yields two
// 53-bit uniforms in [0,1).  Replaces RenderContext.random (lib.rs:34-36).
struct Rng {
  uint32_t k0, k1, pixel, stratum, bounce, draw;
  uint32_t cached_block;
  uint32_t c[4];
  // Which of the enclosing medium's boundary queries is being answered (see the nested form in traverse): a medium met
  // inside another medium's boundary draws from a run of keys of its own per query, 0 everywhere else.
  uint32_t mctx;
  __device__ void init(uint32_t seed_lo, uint32_t seed_hi, uint32_t px, uint32_t st, uint32_t b, uint32_t d) {
    k0 = seed_lo; k1 = seed_hi; pixel = px; stratum = st; bounce = b; draw = d;
    cached_block = 0xFFFFFFFFu;
    mctx = 0;
  }
  __device__ void block(uint32_t blk) {
    uint32_t c0 = pixel, c1 = stratum, c2 = bounce, c3 = blk, a = k0, b = k1;
#pragma unroll
    for (int i = 0; i < 10; i++) {
      // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of separate high and low halves
      const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
      const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
      uint32_t n0 = hi1 ^ c1 ^ a, n2 = hi0 ^ c3 ^ b;
      c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
      a += 0x9E3779B9u;
      b += 0xBB67AE85u;
    }
    c[0] = c0; c[1] = c1; c[2] = c2; c[3] = c3;
    cached_block = blk;
  }
  __device__ double next() {
    uint32_t blk = draw >> 1;
    if (blk != cached_block) block(blk);
    uint32_t lo = (draw & 1) ? c[2] : c[0], hi = (draw & 1) ? c[3] : c[1];
    draw++;
    unsigned long long u = (((unsigned long long)hi << 32) | lo) >> 11;
    return (double)u * (1.0 / 9007199254740992.0);
  }
  // A uniform addressed by a key (block 0x80000000 | key of this path and bounce) instead of by its position in
  // the stream; the stream is not advanced.  Used for the free-flight sample of a ConstantMedium.
  __device__ double keyed(uint32_t key) {
    const uint32_t keep = cached_block, k0c = c[0], k1c = c[1], k2c = c[2], k3c = c[3];
    block(0x80000000u | key);
    const uint32_t lo = c[0], hi = c[1];
    cached_block = keep; c[0] = k0c; c[1] = k1c; c[2] = k2c; c[3] = k3c;
    unsigned long long u = (((unsigned long long)hi << 32) | lo) >> 11;
    return (double)u * (1.0 / 9007199254740992.0);
  }
  // random.rs:169-176 forms
  __device__ double interval(double mn, double mx) { return mn + (mx - mn) * next(); }
};
#define CHIP_CAMERA_BOUNCE 0xFFFFFFFFu
#define CHIP_REJECT_CAP 64  // bound on rejection-sampling loops (P(miss) < 1e-20)

// ------------------------------------------------------------------ small f64 vector helpers (vector.rs)
struct V3 {
  double x, y, z;
};
__device__ inline V3 mk(double x, double y, double z) { return V3{x, y, z}; }
__device__ inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ inline V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
__device__ inline V3 operator*(V3 a, double s) { return V3{a.x * s, a.y * s, a.z * s}; }
__device__ inline V3 operator*(double s, V3 a) { return V3{a.x * s, a.y * s, a.z * s}; }
__device__ inline V3 operator/(V3 a, double s) { return V3{a.x / s, a.y / s, a.z / s}; }
__device__ inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // vector.rs:211-213
__device__ inline V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ inline double len2(V3 a) {  // vector.rs:188-193
  double xs = a.x * a.x, ys = a.y * a.y, zs = a.z * a.z;
  return xs + ys + zs;
}
__device__ inline double len(V3 a) { return sqrt(len2(a)); }
__device__ inline V3 unit(V3 a) { return a / len(a); }
__device__ inline V3 ld3(const double* p) { return V3{p[0], p[1], p[2]}; }
__device__ inline V3 reflect(V3 v, V3 n) { return v - 2.0 * dot(v, n) * n; }  // vector.rs:294-296
__device__ inline V3 refract(V3 uv, V3 n, double eta) {                      // vector.rs:304-309
  double cos_theta = fmin(dot(-uv, n), 1.0);
  V3 perp = eta * (uv + cos_theta * n);
  V3 par = -sqrt(fabs(1.0 - len2(perp))) * n;
  return perp + par;
}
// matrix.rs:91-97 with row-major m; mulT multiplies by the transpose (rotate.rs:52-68)
__device__ inline V3 mat_mul(const double* m, V3 v) {
  return V3{m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z, m[6] * v.x + m[7] * v.y + m[8] * v.z};
}
__device__ inline V3 mat_mulT(const double* m, V3 v) {
  return V3{m[0] * v.x + m[3] * v.y + m[6] * v.z, m[1] * v.x + m[4] * v.y + m[7] * v.z, m[2] * v.x + m[5] * v.y + m[8] * v.z};
}
__device__ inline double fast_rsqrt(double x);
template <bool FAST = false>
__device__ inline V3 random_unit(Rng& r) {  // vector.rs:79-87
  V3 p = mk(1, 0, 0);
  for (int it = 0; it < CHIP_REJECT_CAP; it++) {
    double x = r.interval(-1.0, 1.0);
    double y = r.interval(-1.0, 1.0);
    double z = r.interval(-1.0, 1.0);
    p = mk(x, y, z);
    double l2 = len2(p);
    if (1e-160 < l2 && l2 <= 1.0) return FAST ? p * fast_rsqrt(fmax(l2, 1e-30)) : p / sqrt(l2);
  }
  return unit(p);
}
__device__ inline V3 random_cosine_direction(Rng& r) {  // vector.rs:136-146
  double r1 = r.next();
  double r2 = r.next();
  double phi = 2.0 * CHIP_PI * r1;
  double x = cos(phi) * sqrt(r2);
  double y = sin(phi) * sqrt(r2);
  double z = sqrt(1.0 - r2);
  return mk(x, y, z);
}
struct Onb {  // utils/orthonormal_basis.rs:57-68,95-97
  V3 u, v, w;
  __device__ explicit Onb(V3 normal) {
    w = unit(normal);
    V3 a = fabs(w.x) > 0.9 ? mk(0.0, 1.0, 0.0) : mk(1.0, 0.0, 0.0);
    v = unit(cross(w, a));
    u = cross(w, v);
  }
  __device__ V3 local(V3 p) const { return (p.x * u) + (p.y * v) + (p.z * w); }
};

// ------------------------------------------------------------------ traversal
struct TravRay {
  V3 wo, wd;        // world-space ray (camera.rs:264 passes it to world.hit)
  V3 o, d, inv;     // ray in the current node's space and 1/d (aabb.rs:202)
  double dd;        // direction.length_squared() of d (sphere.rs:102)
  double time;
  // f32 culling form of the ray (ordered traversal): the slab distance of a box plane b is
  // fma(b, m, c) with m ~ 1/d and c ~ -o/d.  Per axis the ray enters a box through its low plane when d > 0 and
  // through its high plane otherwise, so which plane is the near one is a property of the ray: far16 is 16 where
  // the near plane is the high one (a byte offset between a record's low and high plane words, and the index of
  // the near plane in a (low, high) pair when divided by 16).  (m_n, c_n) are biased so that the near plane's
  // distance comes out too small, (m_f, c_f) so that the far plane's comes out too large; capf is the current
  // closest hit rounded up.
  float m_n[3], m_f[3], c_n[3], c_f[3];
  uint32_t far16[3];
  float capf;
  uint32_t best_mat;  // mat word of the closest primitive so far
};
// The traversal kernel's dynamic LDS: [scene cache records][per-lane stacks].  Declared at file scope so
// that device functions index it directly and the compiler emits ds_read / ds_write (a pointer to it
// stored in a struct decays to a generic pointer and is accessed with flat loads).
extern __shared__ uint4 g_lds[];
// Volatile LDS-address-space view of g_lds for the cache reads: keeps them ds_read_b128 (without it the
// compiler merges the "cached / not cached" branches into one generic-pointer load).
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef const volatile __attribute__((address_space(3))) u32x4_t* lds_cv4_t;
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u32x2_t* lds_u2_t;
typedef __attribute__((address_space(3))) unsigned char* lds_b_t;
__device__ inline uint4 lds_read4(uint32_t i) {
  lds_cv4_t p = (lds_cv4_t)g_lds;
  u32x4_t t = p[i];
  uint4 v;
  v.x = t.x; v.y = t.y; v.z = t.z; v.w = t.w;
  return v;
}

struct TravStack {
  lds_u2_t lds;   // this lane's column of the block's LDS stack: (value, entry distance) per entry, one 8-byte word
  uint32_t* ovf;  // this thread's global spill slab
  int sp;
  uint32_t overflow;
  uint32_t redo;    // ordered traversal met a ray it must not handle (see traverse): rerun it literally
  int lds_cap;      // entries kept in LDS
  uint32_t stride8; // bytes between consecutive entries of one lane (8 x threads per block; below 2^24)
};
// Read-only copy of the hottest scene records in LDS: the first n_bvh culling nodes (the tables are in
// breadth-first order, so these are the top of the tree) and the first n_sph spheres.  A sphere is split into
// its four 16-byte components stored in separate arrays (component k of record i at [k * cap + i]), which
// spreads the lanes of a wave over the LDS banks; a culling node keeps its first 112 bytes together (a 112-byte
// stride sends 16 consecutive records to 16 different groups of four banks), so that one address register per axis
// reaches all of it through the instructions' offset fields (see load_wnode).
struct NodeCache {
  uint32_t bvh_off, sph_off;  // first uint4 of the BVH / sphere component arrays in g_lds
  uint32_t n_bvh, n_sph;      // records cached
  uint32_t cap_bvh, cap_sph;  // array strides (records)
};
__device__ inline lds_u2_t st_entry(const TravStack& s) {  // entry s.sp of this lane's column: one 24-bit multiply-add
  return (lds_u2_t)((lds_b_t)s.lds + __umul24((uint32_t)s.sp, s.stride8));
}
struct TravCounters {
  uint32_t bvh, prim;
};

// Entries are (value, entry distance) pairs: in the ordered traversal (WITH_T) a deferred child is dropped
// when it is popped if the closest hit found meanwhile is nearer than the child's box.  Markers and raw
// words are pushed with distance 0 and always survive.
template <bool WITH_T>
__device__ inline void st_push(TravStack& s, uint32_t v, float t = 0.0f) {
  if (s.sp < s.lds_cap) {
    u32x2_t e;
    e.x = v; e.y = __float_as_uint(t);
    (*st_entry(s)) = e;
  } else if (s.sp < s.lds_cap + CHIP_STACK_OVF) {
    s.ovf[2 * (s.sp - s.lds_cap)] = v;
    s.ovf[2 * (s.sp - s.lds_cap) + 1] = __float_as_uint(t);
  } else { s.overflow = 1; return; }
  s.sp++;
}
// Push without a branch in the common case, for the ordered descent's deferred children: the entry is written
// whether or not the lane pushes (`take`), into the next free LDS entry, or into the spare entry behind the LDS part
// (index lds_cap: the kernels allocate one entry more than lds_cap) once the lane is past it, and the stack pointer
// moves only for a lane that pushes.  The spill slab is written under a wave-uniform branch that is rarely taken.
__device__ inline void st_push_sel(TravStack& s, uint32_t v, float t, bool take) {
  u32x2_t e;
  e.x = v; e.y = __float_as_uint(t);
  *(lds_u2_t)((lds_b_t)s.lds + __umul24((uint32_t)min(s.sp, s.lds_cap), s.stride8)) = e;
  if (__any(take && s.sp >= s.lds_cap)) {
    if (take && s.sp >= s.lds_cap) {
      if (s.sp < s.lds_cap + CHIP_STACK_OVF) {
        s.ovf[2 * (s.sp - s.lds_cap)] = v;
        s.ovf[2 * (s.sp - s.lds_cap) + 1] = __float_as_uint(t);
      } else {
        s.overflow = 1;
        take = false;
      }
    }
  }
  s.sp += take ? 1 : 0;
}
__device__ inline uint32_t st_pop_raw(TravStack& s, int base, bool with_t) {
  if (s.sp <= base) return NREF_NONE;
  s.sp--;
  return s.sp < s.lds_cap ? (*st_entry(s)).x : s.ovf[2 * (s.sp - s.lds_cap)];
}
template <bool WITH_T>
__device__ inline uint32_t st_pop(TravStack& s, int base, float capf = 0.0f) {
  for (;;) {
    if (s.sp <= base) return NREF_NONE;
    s.sp--;
    uint32_t v;
    float t = 0.0f;
    if (s.sp < s.lds_cap) {
      const u32x2_t e = (*st_entry(s));
      v = e.x;
      t = __uint_as_float(e.y);
    } else {
      v = s.ovf[2 * (s.sp - s.lds_cap)];
      if (WITH_T) t = __uint_as_float(s.ovf[2 * (s.sp - s.lds_cap) + 1]);
    }
    if (!WITH_T || t <= capf) return v;
  }
}

// Register images of the two records the traversal reads most.
struct BvhRec {
  double box[6];
  nref_t left, right;
};
struct SphereRec {
  double c[3], dc[3], r;
  uint32_t mat;
};
__device__ inline double u2d(uint32_t lo, uint32_t hi) { return __hiloint2double((int)hi, (int)lo); }
template <bool CACHE>
__device__ inline void load_bvh(const DScene& S, const NodeCache& C, uint32_t idx, BvhRec& n) {
  // (the LDS cache holds culling nodes; the literal traversal reads the reference's nodes from memory)
  const uint4* p = reinterpret_cast<const uint4*>(S.bvh + idx);
  const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
  n.box[0] = u2d(a.x, a.y); n.box[1] = u2d(a.z, a.w);
  n.box[2] = u2d(b.x, b.y); n.box[3] = u2d(b.z, b.w);
  n.box[4] = u2d(c.x, c.y); n.box[5] = u2d(c.z, c.w);
  n.left = d.x; n.right = d.y;
}
template <bool CACHE>
__device__ inline void load_sphere(const DScene& S, const NodeCache& C, uint32_t idx, SphereRec& n) {
  uint4 a, b, c, d;
  if (CACHE && idx < C.n_sph) {
    const uint32_t i = C.sph_off + idx;
    a = lds_read4(i); b = lds_read4(i + C.cap_sph); c = lds_read4(i + 2 * C.cap_sph); d = lds_read4(i + 3 * C.cap_sph);
  } else {
    const uint4* p = reinterpret_cast<const uint4*>(S.spheres + idx);
    a = p[0]; b = p[1]; c = p[2]; d = p[3];
  }
  n.c[0] = u2d(a.x, a.y); n.c[1] = u2d(a.z, a.w);
  n.c[2] = u2d(b.x, b.y); n.dc[0] = u2d(b.z, b.w);
  n.dc[1] = u2d(c.x, c.y); n.dc[2] = u2d(c.z, c.w);
  n.r = u2d(d.x, d.y);
  n.mat = d.z;
}

// A culling node as one ray sees it: per axis the four children's near planes and far planes (TravRay::far16 says
// which of the record's low / high plane words is which for this ray).
struct WNodeRec {
  float near_x[4], far_x[4], near_y[4], far_y[4], near_z[4], far_z[4];
  nref_t child[4];
};
__device__ inline void u4_to_f4(const uint4& u, float* f) {
  f[0] = __uint_as_float(u.x); f[1] = __uint_as_float(u.y); f[2] = __uint_as_float(u.z); f[3] = __uint_as_float(u.w);
}
__device__ inline uint4 lds_read4_at(uint32_t byte) {
  typedef const volatile __attribute__((address_space(3))) unsigned char* lds_cvb_t;
  lds_cv4_t p = (lds_cv4_t)((lds_cvb_t)g_lds + byte);
  u32x4_t t = *p;
  uint4 v;
  v.x = t.x; v.y = t.y; v.z = t.z; v.w = t.w;
  return v;
}
// In a 128-byte record the low and high plane words of an axis are 16 bytes apart in one 32-byte aligned pair, so
// the far plane's address is the near plane's with bit 4 flipped; the LDS copy has the same record layout at a
// 112-byte stride.  Global records are addressed with 32-bit byte offsets from the table's base (the host refuses
// tables of 2^25 nodes or more).
template <bool CACHE>
__device__ inline void load_wnode(const DScene& S, const NodeCache& C, uint32_t idx, const TravRay& r, WNodeRec& n) {
  uint4 q[CHIP_WNODE_U4];
  if (CACHE && idx < C.n_bvh) {
    // one multiply for the record, one add and one subtract per axis (near plane at +f, far plane at +16 - f); the
    // rest of each address is a constant that rides in the instruction's offset field
    const uint32_t rec = __umul24(idx, 16u * CHIP_WNODE_U4) + (C.bvh_off << 4);
    const uint32_t nx = rec + r.far16[0], fx = rec - r.far16[0];
    const uint32_t ny = rec + r.far16[1], fy = rec - r.far16[1];
    const uint32_t nz = rec + r.far16[2], fz = rec - r.far16[2];
    q[0] = lds_read4_at(nx); q[1] = lds_read4_at(fx + 16u);
    q[2] = lds_read4_at(ny + 32u); q[3] = lds_read4_at(fy + 48u);
    q[4] = lds_read4_at(nz + 64u); q[5] = lds_read4_at(fz + 80u);
    q[6] = lds_read4_at(rec + 96u);
  } else {
    const unsigned char* p = reinterpret_cast<const unsigned char*>(S.wnodes);
    const uint32_t b = idx << 7;
    const uint32_t nx = b + r.far16[0], ny = b + r.far16[1] + 32u, nz = b + r.far16[2] + 64u;
    q[0] = *reinterpret_cast<const uint4*>(p + nx); q[1] = *reinterpret_cast<const uint4*>(p + (nx ^ 16u));
    q[2] = *reinterpret_cast<const uint4*>(p + ny); q[3] = *reinterpret_cast<const uint4*>(p + (ny ^ 16u));
    q[4] = *reinterpret_cast<const uint4*>(p + nz); q[5] = *reinterpret_cast<const uint4*>(p + (nz ^ 16u));
    q[6] = *reinterpret_cast<const uint4*>(p + (b + 96u));
  }
  u4_to_f4(q[0], n.near_x); u4_to_f4(q[1], n.far_x); u4_to_f4(q[2], n.near_y); u4_to_f4(q[3], n.far_y); u4_to_f4(q[4], n.near_z); u4_to_f4(q[5], n.far_z);
  n.child[0] = q[6].x; n.child[1] = q[6].y; n.child[2] = q[6].z; n.child[3] = q[6].w;
}

// Conservative f32 slab test of one child box against the ray's culling form: [entry, exit] contains
// every t in [tmin, closest] at which the ray is inside the f64 box, so a child with entry > exit can
// hold no hit that would replace the current one.
#define CHIP_CULL_DELTA 9.5367431640625e-07f  // 2^-20: covers the f32 rounding of 1/d, m and the multiply-add
#define CHIP_CULL_TMIN 0.00099999f             // just below the 0.001 ray minimum of camera.rs:264
#define CHIP_CULL_BIG 3.0e38f
__device__ inline float f32_above(double v) {  // an f32 >= v (one ulp of slack)
  const float f = (float)v;
  return fmaf(fabsf(f), 1.1920929e-07f, f);
}
__device__ inline float f32_below(double v) {
  const float f = (float)v;
  return fmaf(-fabsf(f), 1.1920929e-07f, f);
}
// Culling form for the ray currently in r.o / r.d.  Every direction component is finite and well away from zero here
// (has_zero_component() sends the other rays to the literal traversal), so each axis culls and a box with its low planes
// above its high planes (an unused child slot) is missed whatever the ray.
// Returns whether every component's reciprocal slope is an ordinary f32 of moderate size (the f32 counterpart of
// has_zero_component(), which a caller that computes the form anyway can use in its place).
__device__ inline bool ray_set_cull(TravRay& r) {
  const double o[3] = {r.o.x, r.o.y, r.o.z}, d[3] = {r.d.x, r.d.y, r.d.z};
  bool ok = true;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    // The near plane is the low one for d > 0 and the high one for d < 0; either way its distance (plane - o) / d
    // shrinks when |1/d| does, and the far plane's grows when |1/d| does, so the two multipliers do not depend on
    // the sign: only which plane they are applied to does (far16).
    const float invf = __builtin_amdgcn_rcpf((float)d[a]);  // v_rcp_f32 (1 ulp, inside CHIP_CULL_DELTA) instead of a correctly rounded division
    r.m_n[a] = invf * (1.0f - CHIP_CULL_DELTA);
    r.m_f[a] = invf * (1.0f + CHIP_CULL_DELTA);
    // -o * m in f32: the conversion of o and the product are each off by at most 2^-24 of the result, so moving
    // the product by 2^-22 of itself puts it safely on the conservative side
    const float of = (float)o[a];
    const float pn = -(of * r.m_n[a]), pf = -(of * r.m_f[a]);
    r.c_n[a] = fmaf(-fabsf(pn), 2.3841858e-07f, pn);
    r.c_f[a] = fmaf(fabsf(pf), 2.3841858e-07f, pf);
    r.far16[a] = invf > 0.0f ? 0u : 16u;
    ok = ((int)ok & (int)(fabsf(invf) > 1.0e-18f) & (int)(fabsf(invf) < 1.0e18f)) != 0;  // (false for NaN; no short-circuit branches)
  }
  return ok;
}

// aabb.rs:195-228.  The per-axis early return is folded into one final test: the
// running interval only ever narrows and NaN candidates are never stored, so
// "max <= min after some axis" holds at the end iff it held after that axis.
__device__ __noinline__ bool aabb_hit_chain(double b0, double b1, double b2, double b3, double b4, double b5, V3 o, V3 inv, double tmin,
                                             double tmax) {
#define CHIP_SLAB(lo, hi, o, inv)            \
  {                                          \
    double t0 = ((lo) - (o)) * (inv);        \
    double t1 = ((hi) - (o)) * (inv);        \
    if (t0 < t1) {                           \
      if (t0 > tmin) tmin = t0;              \
      if (t1 < tmax) tmax = t1;              \
    } else {                                 \
      if (t1 > tmin) tmin = t1;              \
      if (t0 < tmax) tmax = t0;              \
    }                                        \
  }
  CHIP_SLAB(b0, b1, o.x, inv.x)
  CHIP_SLAB(b2, b3, o.y, inv.y)
  CHIP_SLAB(b4, b5, o.z, inv.z)
#undef CHIP_SLAB
  return !(tmax <= tmin);
}
// Same decision with min/max instructions: for ordered t0, t1 the compare chain above is
// tmin = max(tmin, min(t0,t1)), tmax = min(tmax, max(t0,t1)).  When a slab product is NaN
// (0 * inf: zero direction component with the origin on a slab plane) the chain's
// "comparison false => no update" behaviour differs from min/max, so those lanes take the chain.
// v_min_f64 / v_max_f64 without the operand canonicalisation the compiler adds to fmin / fmax (it cannot
// prove the operands are not signalling NaNs); used only where the operands are known to be ordered.
__device__ inline double vmin_f64(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ inline double vmax_f64(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ inline bool aabb_hit(const double* b, const TravRay& r, double tmin, double tmax) {
  double t0x = (b[0] - r.o.x) * r.inv.x, t1x = (b[1] - r.o.x) * r.inv.x;
  double t0y = (b[2] - r.o.y) * r.inv.y, t1y = (b[3] - r.o.y) * r.inv.y;
  double t0z = (b[4] - r.o.z) * r.inv.z, t1z = (b[5] - r.o.z) * r.inv.z;
  if (__builtin_expect(__builtin_isunordered(t0x, t1x) || __builtin_isunordered(t0y, t1y) || __builtin_isunordered(t0z, t1z), 0))
    return aabb_hit_chain(b[0], b[1], b[2], b[3], b[4], b[5], r.o, r.inv, tmin, tmax);
  tmin = vmax_f64(tmin, vmin_f64(t0x, t1x));
  tmax = vmin_f64(tmax, vmax_f64(t0x, t1x));
  tmin = vmax_f64(tmin, vmin_f64(t0y, t1y));
  tmax = vmin_f64(tmax, vmax_f64(t0y, t1y));
  tmin = vmax_f64(tmin, vmin_f64(t0z, t1z));
  tmax = vmin_f64(tmax, vmax_f64(t0z, t1z));
  return !(tmax <= tmin);
}

// sphere.rs:99-121: nearest root strictly inside (tmin, tmax)
template <typename SphereT>
__device__ inline bool sphere_hit(const SphereT& s, const TravRay& r, double tmin, double tmax, double& t_out) {
  V3 center = ld3(s.c) + (r.time * ld3(s.dc));
  V3 oc = center - r.o;
  double a = r.dd;
  double h = dot(r.d, oc);
  double c = len2(oc) - s.r * s.r;
  double disc = h * h - a * c;
  if (disc < 0.0) return false;
  double sq = sqrt(disc);
  double root = (h - sq) / a;
  if (!(tmin < root && root < tmax)) {
    root = (h + sq) / a;
    if (!(tmin < root && root < tmax)) return false;
  }
  t_out = root;
  return true;
}
// quad.rs:144-169: t inside [tmin, tmax] inclusive, plane coordinates inside [0,1] inclusive
__device__ inline bool quad_hit_od(const DQuad& q, V3 ro, V3 rd, double tmin, double tmax, double& t_out, double& alpha, double& beta);
__device__ inline bool quad_hit(const DQuad& q, const TravRay& r, double tmin, double tmax, double& t_out, double& alpha, double& beta) {
  return quad_hit_od(q, r.o, r.d, tmin, tmax, t_out, alpha, beta);
}
__device__ inline bool quad_hit_od(const DQuad& q, V3 ro, V3 rd, double tmin, double tmax, double& t_out, double& alpha, double& beta) {
  // quad.rs:128-157 without its early returns: every operand is asked for at once (one memory round trip instead of
  // two) and the conditions are combined at the end; the arithmetic that decides is the same, a rejected test merely
  // computes plane coordinates nobody looks at (comparisons with the NaN or infinity of a zero denominator are false)
  const V3 n = ld3(q.n), qq = ld3(q.q), w = ld3(q.w), v = ld3(q.v), u = ld3(q.u);
  const double denom = dot(n, rd);
  const double t = (q.d - dot(n, ro)) / denom;
  const V3 p = ro + (t * rd);
  const V3 planar = p - qq;
  alpha = dot(w, cross(planar, v));
  beta = dot(w, cross(u, planar));
  const bool ok = !(fabs(denom) < 1e-8) && (tmin <= t && t <= tmax) && (0.0 <= alpha && alpha <= 1.0) && (0.0 <= beta && beta <= 1.0);
  if (ok) t_out = t;
  return ok;
}

// disc.rs:111-131: plane hit with t inside [tmin, tmax] inclusive, then the radius test
__device__ inline bool disc_hit(const DDisc& ds, const TravRay& r, double tmin, double tmax, double& t_out) {
  V3 n = ld3(ds.n), c = ld3(ds.c);
  double denominator = dot(r.d, n);
  if (fabs(denominator) < 1e-8) return false;
  double t = dot(c - r.o, n) / denominator;
  if (!(tmin <= t && t <= tmax)) return false;
  V3 pt = r.o + (t * r.d);
  V3 v = pt - c;
  if (len2(v) > ds.r * ds.r) return false;
  t_out = t;
  return true;
}
// cone.rs:164-243 (ConeFrustumWall::hit up to the choice of t)
__device__ inline bool cone_hit(const DCone& cn, const TravRay& r, double tmin, double tmax, double& t_out) {
  const double h = cn.h, r0 = cn.r0, r1 = cn.r1;
  const double base_y = cn.base[1];
  V3 lo = r.o - ld3(cn.base);
  double ox = lo.x, oy = lo.y, oz = lo.z;
  double dx = r.d.x, dy = r.d.y, dz = r.d.z;
  double dr = r1 - r0;
  double k = dr / h;
  double a = dx * dx + dz * dz - k * k * dy * dy;
  double b = 2.0 * (ox * dx + oz * dz - k * k * oy * dy - k * r0 * dy);
  double c = ox * ox + oz * oz - (r0 * r0 + 2.0 * k * r0 * oy + k * k * oy * oy);
  if (fabs(a) < 1e-8) return false;
  double discriminant = b * b - 4.0 * a * c;
  if (discriminant < 0.0) return false;
  double sq = sqrt(discriminant);
  double t = (-b - sq) / (2.0 * a);
  if (!(tmin <= t && t <= tmax)) {
    t = (-b + sq) / (2.0 * a);
    if (!(tmin <= t && t <= tmax)) return false;
  }
  double hit_y = r.o.y + (t * r.d.y);  // the y of ray.at(t)
  double y_min = base_y, y_max = base_y + h;
  if (hit_y < y_min || hit_y > y_max) {
    double t_other = (-b + sq) / (2.0 * a);
    if (t_other == t) return false;
    if (!(tmin <= t_other && t_other <= tmax)) return false;
    double y_other = r.o.y + (t_other * r.d.y);
    if (y_other < y_min || y_other > y_max) return false;
    t = t_other;
  }
  t_out = t;
  return true;
}

// Ordered traversal visits primitives in another order than the reference, so a candidate whose t
// equals the current closest t exactly is decided by what the reference's order would have done:
// the later primitive (depth-first position `seq`) replaces the earlier one iff its own interval test
// is inclusive (quad.rs:152 `contains`), and loses iff it is exclusive (sphere.rs:110 `surrounds`).
__device__ inline uint32_t prim_seq(const DSphere* spheres, const DQuad* quads, const DDisc* discs, const DCone* cones, nref_t p) {
  const uint32_t k = NREF_KIND(p), i = NREF_INDEX(p);
  return k == NK_SPHERE ? spheres[i].seq : (k == NK_QUAD ? quads[i].seq : (k == NK_DISC ? discs[i].seq : cones[i].seq));
}
// (takes the tables rather than the scene: an out-of-line call with the whole DScene would put it on the stack)
__device__ __noinline__ bool tie_wins_tables(const DSphere* spheres, const DQuad* quads, const DDisc* discs, const DCone* cones, nref_t cand, nref_t best) {
  if (best == NREF_NONE) return true;
  // quads, discs and cone walls accept t == max (`contains`); spheres do not (`surrounds`)
  const bool c_incl = NREF_KIND(cand) != NK_SPHERE, b_incl = NREF_KIND(best) != NK_SPHERE;
  return prim_seq(spheres, quads, discs, cones, cand) >= prim_seq(spheres, quads, discs, cones, best) ? c_incl : !b_incl;
}
__device__ inline bool tie_wins(const DScene& S, nref_t cand, nref_t best) { return tie_wins_tables(S.spheres, S.quads, S.discs, S.cones, cand, best); }
// sphere.rs:99-121 with the interval's upper end inclusive (ties are then settled by tie_wins)
template <typename SphereT>
__device__ inline bool sphere_hit_incl(const SphereT& s, const TravRay& r, double tmin, double tmax, double& t_out) {
  V3 center = ld3(s.c) + (r.time * ld3(s.dc));
  V3 oc = center - r.o;
  double a = r.dd;
  double h = dot(r.d, oc);
  double c = len2(oc) - s.r * s.r;
  double disc = h * h - a * c;
  if (disc < 0.0) return false;
  double sq = sqrt(disc);
  double root = (h - sq) / a;
  if (!(tmin < root && root <= tmax)) {
    root = (h + sq) / a;
    if (!(tmin < root && root <= tmax)) return false;
  }
  t_out = root;
  return true;
}

template <bool ORD = false>
__device__ inline bool ray_set_inv(TravRay& r) {
  bool ok = true;
  if (ORD) ok = ray_set_cull(r);
  else r.inv = mk(1.0 / r.d.x, 1.0 / r.d.y, 1.0 / r.d.z);
  r.dd = len2(r.d);
  return ok;
}
// Scene features a traversal / shade instantiation handles (scenes lacking a feature run a leaner variant):
// quads & groups | Translate & Rotate | ConstantMedium | discs, cone walls & Scale
// TF_MEDIA_ANY: some medium's boundary is more than a bare sphere (a box is queried in place, anything else by the out-of-line sub-traversal, which
// costs the whole kernel registers; media bounded by spheres alone run without it)
// TF_NO_QUADROT (with TF_QUADS | TF_XFORM only): groups and Translates but neither quads nor Rotates, the shape of
// instanced-sphere scenes (what the OpenSCAD front end emits for spheres: Translate(Group[Sphere]))
// TF_MEDIA_BOX: some medium is bounded by a box (queried in place, six quad tests), none by anything but a sphere or a box
enum : uint32_t { TF_QUADS = 1, TF_XFORM = 2, TF_MEDIA = 4, TF_EXT = 8, TF_MEDIA_ANY = 16, TF_NO_QUADROT = 32, TF_MEDIA_BOX = 64 };

// translate.rs:28 / rotate.rs:137-138 / scale.rs:97-98
template <bool EXT>
__device__ inline void apply_xform(const DScene& S, nref_t x, V3& o, V3& d) {
  if (NREF_KIND(x) == NK_TRANSLATE) {
    o = o - ld3(S.translates[NREF_INDEX(x)].off);
  } else if (EXT && NREF_KIND(x) == NK_SCALE) {
    const double* m = S.scales[NREF_INDEX(x)].inv;
    o = mat_mul(m, o);
    d = mat_mul(m, d);
  } else {
    const double* m = S.rotates[NREF_INDEX(x)].m;
    o = mat_mulT(m, o);
    d = mat_mulT(m, d);
  }
}
template <bool EXT>
__device__ inline void xform_chain(const DScene& S, nref_t x, uint32_t& first, uint32_t& n) {
  if (x == NREF_NONE) { first = 0; n = 0; }
  else if (NREF_KIND(x) == NK_TRANSLATE) { first = S.translates[NREF_INDEX(x)].chain_first; n = S.translates[NREF_INDEX(x)].chain_len; }
  else if (EXT && NREF_KIND(x) == NK_SCALE) { first = S.scales[NREF_INDEX(x)].chain_first; n = S.scales[NREF_INDEX(x)].chain_len; }
  else { first = S.rotates[NREF_INDEX(x)].chain_first; n = S.rotates[NREF_INDEX(x)].chain_len; }
}
// Ray in the space below transform x: replay x's root->self chain on the world ray.
template <bool EXT>
__device__ inline void ray_to_space(const DScene& S, nref_t x, V3 wo, V3 wd, V3& o, V3& d) {
  o = wo; d = wd;
  uint32_t first, n;
  xform_chain<EXT>(S, x, first, n);
  for (uint32_t i = 0; i < n; i++) apply_xform<EXT>(S, S.xchain[first + i], o, d);
}


// Closest hit below `start` for the ray in r, over (tmin, best_t]; best_t doubles as
// the shrinking interval maximum of bvh.rs:66-70 / group.rs:47-52.
// `node` is the traversal cursor: on entry the node to visit, on return NREF_NONE when the subtree
// is exhausted.  With YIELD the loop also returns (cursor and stack intact, to be resumed) as soon as
// fewer than CHIP_REFILL_THRESH lanes of the wave are still traversing, so the caller can hand the
// idle lanes new rays.
// A zero direction component makes the reference's slab test (aabb.rs:202-228) form 0 * inf = NaN when the
// origin lies exactly on a box plane, and its compare chain then rejects or accepts the box depending on
// the sign of the zero; the culling test of the ordered traversal does not reproduce that, so such rays
// are traced by the literal interpretation instead.
// The same goes for components so small or so large that their f32 reciprocal is not an ordinary number: the culling
// test relies on every axis having a finite non-zero slope.  NaN components fail the comparisons and land here too.
__device__ inline bool has_zero_component(V3 d) {
  const double lo = 1.0e-18, hi = 1.0e18;
  return !(fabs(d.x) > lo && fabs(d.x) < hi && fabs(d.y) > lo && fabs(d.y) < hi && fabs(d.z) > lo && fabs(d.z) < hi);
}

// Takes the ray straight into the space of `inner`, the innermost of a run of nested instances (the instances in
// between have no effect of their own: translate.rs:26-38, rotate.rs:137-150, scale.rs:93-115 only transform the ray
// and hand it on).  Returns true when the ordered traversal has to hand the ray back (a zero direction component).
template <uint32_t F, bool ORD>
__device__ inline bool enter_instances(const DScene& S, nref_t inner, TravRay& r, TravStack& st) {
  ray_to_space<(F & TF_EXT) != 0>(S, inner, r.wo, r.wd, r.o, r.d);
  if (ORD && has_zero_component(r.d)) {
    st.redo = 1;
    return true;
  }
  ray_set_inv<ORD>(r);
  return false;
}


// ORD selects the ordered traversal (nearer child first, f32 child-box culling, ties settled by the
// reference's visiting order); without it the reference's own order and f64 box tests are interpreted
// literally, which the ConstantMedium variant needs (its result depends on the interval it is handed).
template <uint32_t F, bool YIELD, bool CACHE, bool ORD = true>
__device__ void traverse(const DScene& S, const NodeCache& C, nref_t& node, TravRay& r, double tmin, double& best_t, nref_t& best, Rng& rng,
                         TravStack& st, const int base, TravCounters& cnt, int yield_below = CHIP_REFILL_THRESH);

// A medium's boundary subtree (constant_medium.rs:42-45) is traversed on the same stack.  It may hold media of its own
// (constant_medium.rs:19-29 takes any node) as long as theirs are bounded by a sphere or a box, which are queried in
// place: this instantiation carries the medium code without the sub-traversal, so the nesting ends here.
template <bool EXT>
__device__ __noinline__ void traverse_boundary(const DScene& S, nref_t start, TravRay& r, double tmin, double& best_t, nref_t& best,
                                               Rng& rng, TravStack& st, TravCounters& cnt) {
  nref_t node = start;
  NodeCache none{0, 0, 0, 0, 0, 0};
  traverse<TF_QUADS | TF_XFORM | TF_MEDIA | TF_MEDIA_BOX | (EXT ? TF_EXT : 0u), false, false, false>(S, none, node, r, tmin, best_t, best, rng, st, st.sp, cnt);
}

template <uint32_t F, bool YIELD, bool CACHE, bool ORD>
__device__ void traverse(const DScene& S, const NodeCache& C, nref_t& node, TravRay& r, double tmin, double& best_t, nref_t& best, Rng& rng,
                         TravStack& st, const int base, TravCounters& cnt, int yield_below) {
  constexpr uint32_t DEN = F == 0 ? CHIP_DESCENT_DEN_SPHERES : ((F & TF_NO_QUADROT) ? CHIP_DESCENT_DEN_INSTANCES : ((F & TF_MEDIA) ? CHIP_DESCENT_DEN_MEDIA : 0u));
  for (;;) {
    // BVH descent (bvh.rs:58-76): test the node's own box over (tmin, best_t); left first, right deferred.
    if (ORD) {
      // the boxes of up to four children are tested here; the nearest child is visited next, the others are deferred
      // nearest on top.  A child's sort key is its entry distance (positive, so its bits order like the value) with
      // the two lowest bits replaced by the child's slot; the distance a deferred child carries is thereby rounded
      // down, which keeps the pop-time test conservative.
      // The wave stays in the descent loop while enough of its lanes are at culling nodes; once most of them wait
      // at something else it lets go, handles those (the lanes still at culling nodes sit that out) and comes back.
      // Either phase thereby runs with a fair share of the wave's lanes, whereas waiting for the last descending
      // lane leaves most of them idle for most of the loop's trips.
      for (;;) {
        const bool inner = NREF_KIND(node) == NK_BVH;
        if (DEN != 0u) {
          const uint32_t n_inner = (uint32_t)__popcll(__ballot(inner)), n_here = (uint32_t)__popcll(__ballot(1));
          if (n_inner == 0u || n_inner * DEN < n_here) break;
          if (!inner) continue;
        } else if (!inner) {
          break;
        }
        WNodeRec n;
        load_wnode<CACHE>(S, C, NREF_INDEX(node), r, n);
#if CHIP_TRAV_PROFILE  /* counts wave-level loop trips instead of per-lane visits */
        if (__lane_id() == (uint32_t)__ffsll((long long)__ballot(1)) - 1u) cnt.bvh++;
#else
        cnt.bvh++;
#endif
        uint32_t key[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const float x0 = fmaf(n.near_x[k], r.m_n[0], r.c_n[0]), x1 = fmaf(n.far_x[k], r.m_f[0], r.c_f[0]);
          const float y0 = fmaf(n.near_y[k], r.m_n[1], r.c_n[1]), y1 = fmaf(n.far_y[k], r.m_f[1], r.c_f[1]);
          const float z0 = fmaf(n.near_z[k], r.m_n[2], r.c_n[2]), z1 = fmaf(n.far_z[k], r.m_f[2], r.c_f[2]);
          const float en = fmaxf(__builtin_fmaxf(__builtin_fmaxf(x0, y0), z0), CHIP_CULL_TMIN);
          const float ex = fminf(__builtin_fminf(__builtin_fminf(x1, y1), z1), r.capf);
          key[k] = en <= ex ? ((__float_as_uint(en) & ~3u) | (uint32_t)k) : 0xFFFFFFFFu;  // an unused slot never passes
        }
        // sorting network on four keys
#define CHIP_CSWAP(a, b) { const uint32_t lo_ = min(key[a], key[b]), hi_ = max(key[a], key[b]); key[a] = lo_; key[b] = hi_; }
        CHIP_CSWAP(0, 1) CHIP_CSWAP(2, 3) CHIP_CSWAP(0, 2) CHIP_CSWAP(1, 3) CHIP_CSWAP(1, 2)
#undef CHIP_CSWAP
#define CHIP_CHILD(kk) (((kk) & 2u) ? (((kk) & 1u) ? n.child[3] : n.child[2]) : (((kk) & 1u) ? n.child[1] : n.child[0]))
        // (a node without a child in reach has four all-ones keys: the pushes keep nothing and the node is popped)
        if (__any(key[2] != 0xFFFFFFFFu)) {  // three or four children in reach: rare enough to be worth the branch
          st_push_sel(st, CHIP_CHILD(key[3]), __uint_as_float(key[3] & ~3u), key[3] != 0xFFFFFFFFu);
          st_push_sel(st, CHIP_CHILD(key[2]), __uint_as_float(key[2] & ~3u), key[2] != 0xFFFFFFFFu);
        }
        st_push_sel(st, CHIP_CHILD(key[1]), __uint_as_float(key[1] & ~3u), key[1] != 0xFFFFFFFFu);
        node = CHIP_CHILD(key[0]);
        if (key[0] == 0xFFFFFFFFu) node = st_pop<true>(st, base, r.capf);
#undef CHIP_CHILD
      }
    } else {
      while (NREF_KIND(node) == NK_BVH) {
        BvhRec n;
        load_bvh<CACHE>(S, C, NREF_INDEX(node), n);
        cnt.bvh++;
        if (aabb_hit(n.box, r, tmin, best_t)) {
          st_push<false>(st, n.right);
          node = n.left;
        } else {
          node = st_pop<false>(st, base);
        }
      }
    }
    if (node == NREF_NONE) break;
    const uint32_t kind = NREF_KIND(node), idx = NREF_INDEX(node);
    // candidate hit at t: closer wins; ORD settles exact ties by the reference's visiting order
#define CHIP_TAKE(ref, tval, matw)                                            \
  if (!ORD || (tval) < best_t || tie_wins(S, (ref), best)) {                  \
    best_t = (tval);                                                          \
    best = (ref);                                                             \
    r.best_mat = (matw);                                                      \
    if (ORD) r.capf = f32_above(best_t);                                      \
  }
    if (ORD && DEN != 0u && kind == NK_BVH) {
      // still descending: sits this round out
    } else if (kind == NK_SPHERE) {
      double t;
#if CHIP_TRAV_PROFILE
      if (__lane_id() == (uint32_t)__ffsll((long long)__ballot(1)) - 1u) cnt.prim++;
#else
      cnt.prim++;
#endif
      SphereRec sp;
      load_sphere<CACHE>(S, C, idx, sp);
      if (ORD ? sphere_hit_incl(sp, r, tmin, best_t, t) : sphere_hit(sp, r, tmin, best_t, t)) { CHIP_TAKE(node, t, sp.mat) }
    } else if ((F & TF_QUADS) && !(F & TF_NO_QUADROT) && kind == NK_QUAD) {
      double t, al, be;
      cnt.prim++;
      if (quad_hit(S.quads[idx], r, tmin, best_t, t, al, be)) { CHIP_TAKE(node, t, S.quads[idx].mat) }
    } else if ((F & TF_EXT) && kind == NK_DISC) {
      double t;
      cnt.prim++;
      if (disc_hit(S.discs[idx], r, tmin, best_t, t)) { CHIP_TAKE(node, t, S.discs[idx].mat) }
    } else if ((F & TF_EXT) && kind == NK_CONE) {
      double t;
      cnt.prim++;
      if (cone_hit(S.cones[idx], r, tmin, best_t, t)) { CHIP_TAKE(node, t, S.cones[idx].mat) }
    } else if (!ORD && (F & TF_QUADS) && (kind == NK_GROUP || kind == NK_GROUP_CONT)) {
      // group.rs:44-54: children in order, each over the shrinking interval.  Primitive
      // children are tested inline; any other child is descended into, leaving a
      // (position, continuation) pair on the stack.
      uint32_t pos = 0;
      if (kind == NK_GROUP_CONT) pos = st_pop_raw(st, base, ORD);
      const DGroup g = S.groups[idx];
      nref_t descend = NREF_NONE;
      for (; pos < g.count; pos++) {
        nref_t c = S.gchild[g.first + pos];
        uint32_t ck = NREF_KIND(c);
        if (ck == NK_SPHERE) {
          double t;
          cnt.prim++;
          if (ORD ? sphere_hit_incl(S.spheres[NREF_INDEX(c)], r, tmin, best_t, t) : sphere_hit(S.spheres[NREF_INDEX(c)], r, tmin, best_t, t)) {
            CHIP_TAKE(c, t, S.spheres[NREF_INDEX(c)].mat)
          }
        } else if (!(F & TF_NO_QUADROT) && ck == NK_QUAD) {
          double t, al, be;
          cnt.prim++;
          if (quad_hit(S.quads[NREF_INDEX(c)], r, tmin, best_t, t, al, be)) { CHIP_TAKE(c, t, S.quads[NREF_INDEX(c)].mat) }
        } else if ((F & TF_EXT) && ck == NK_DISC) {
          double t;
          cnt.prim++;
          if (disc_hit(S.discs[NREF_INDEX(c)], r, tmin, best_t, t)) { CHIP_TAKE(c, t, S.discs[NREF_INDEX(c)].mat) }
        } else if ((F & TF_EXT) && ck == NK_CONE) {
          double t;
          cnt.prim++;
          if (cone_hit(S.cones[NREF_INDEX(c)], r, tmin, best_t, t)) { CHIP_TAKE(c, t, S.cones[NREF_INDEX(c)].mat) }
        } else {
          if (pos + 1 < g.count) {
            // both words or neither: a marker without its position (or the reverse) would be misread when popped
            if (st.sp + 2 <= st.lds_cap + CHIP_STACK_OVF) {
              st_push<ORD>(st, pos + 1);
              st_push<ORD>(st, NREF(NK_GROUP_CONT, idx));
            } else {
              st.overflow = 1;
            }
          }
          descend = c;
          break;
        }
      }
      if (descend != NREF_NONE) { node = descend; continue; }
    } else if ((F & TF_XFORM) && kind == NK_TRANSLATE) {
      const DTranslate& t = S.translates[idx];
      nref_t only = t.only_sphere;
      V3 toff = ld3(t.off);
      SphereRec sp;
      sp.c[0] = t.sc[0]; sp.c[1] = t.sc[1]; sp.c[2] = t.sc[2];
      // (a scene without moving spheres: the same arithmetic on +0.0, without the record's second half)
      sp.dc[0] = sp.dc[1] = sp.dc[2] = 0.0;
      if (!(F & TF_NO_QUADROT) || S.moving_spheres) {
        if (F & TF_NO_QUADROT) asm volatile("" ::: "memory");  // (keeps these loads on this side of the branch)
        sp.dc[0] = t.sdc[0]; sp.dc[1] = t.sdc[1]; sp.dc[2] = t.sdc[2];
      }
      sp.r = t.sr;
      sp.mat = t.sphere_mat;
      // Instanced-sphere scenes: everything the test below reads is asked for here, before the branch on what the
      // Translate holds (left alone, the compiler loads that word, waits, branches and only then asks for the rest of
      // the same record: two trips to memory one after the other).
      if (F & TF_NO_QUADROT)
        asm volatile("" : "+v"(only), "+v"(toff.x), "+v"(toff.y), "+v"(toff.z), "+v"(sp.c[0]), "+v"(sp.c[1]), "+v"(sp.c[2]), "+v"(sp.dc[0]), "+v"(sp.dc[1]),
                         "+v"(sp.dc[2]), "+v"(sp.r), "+v"(sp.mat));
      if (only != NREF_NONE) {
        // A Translate around a single sphere (translate.rs:26-38 over group.rs:44-54 over sphere.rs:99-121): the same
        // arithmetic on the offset origin, without entering and leaving the instance on the stack.
        // The sphere's geometry rides in the Translate's own record (one 128-byte fetch instead of two dependent ones).
        const V3 outer_o = r.o;
        r.o = r.o - toff;
        const nref_t c = only;
        double tt;
        cnt.prim++;
        if (ORD ? sphere_hit_incl(sp, r, tmin, best_t, tt) : sphere_hit(sp, r, tmin, best_t, tt)) { CHIP_TAKE(c, tt, sp.mat) }
        r.o = outer_o;
      } else {
        st_push<ORD>(st, NREF(NK_EXIT_TRANSLATE, idx));
        if (t.enter_xform == node) {
          r.o = r.o - ld3(t.off);
          if (ORD) ray_set_cull(r);
        } else if (enter_instances<F, ORD>(S, t.enter_xform, r, st)) {
          node = NREF_NONE;
          return;
        }
        node = ORD ? t.ord_child : t.enter_child;
        continue;
      }
    } else if ((F & TF_XFORM) && !(F & TF_NO_QUADROT) && kind == NK_ROTATE) {
      const DRotate& t = S.rotates[idx];
      st_push<ORD>(st, NREF(NK_EXIT_ROTATE, idx));
      if (t.enter_xform != node) {
        if (enter_instances<F, ORD>(S, t.enter_xform, r, st)) {
          node = NREF_NONE;
          return;
        }
        node = ORD ? t.ord_child : t.enter_child;
        continue;
      }
      r.o = mat_mulT(t.m, r.o);
      r.d = mat_mulT(t.m, r.d);
      if (ORD && has_zero_component(r.d)) {  // only an exact cancellation gets here: start over, literally
        st.redo = 1;
        node = NREF_NONE;
        return;
      }
      ray_set_inv<ORD>(r);
      node = ORD ? t.ord_child : t.child;
      continue;
    } else if ((F & TF_EXT) && kind == NK_SCALE) {
      const DScale& t = S.scales[idx];
      st_push<ORD>(st, NREF(NK_EXIT_SCALE, idx));
      if (t.enter_xform != node) {
        if (enter_instances<F, ORD>(S, t.enter_xform, r, st)) {
          node = NREF_NONE;
          return;
        }
        node = ORD ? t.ord_child : t.enter_child;
        continue;
      }
      r.o = mat_mul(t.inv, r.o);
      r.d = mat_mul(t.inv, r.d);
      if (ORD && has_zero_component(r.d)) {  // a zero (or 0 * inf = NaN) component: start over, literally
        st.redo = 1;
        node = NREF_NONE;
        return;
      }
      ray_set_inv<ORD>(r);
      node = ORD ? t.ord_child : t.child;
      continue;
    } else if ((F & TF_XFORM) && (kind == NK_EXIT_TRANSLATE || kind == NK_EXIT_ROTATE || ((F & TF_EXT) && kind == NK_EXIT_SCALE))) {
      nref_t px = kind == NK_EXIT_TRANSLATE ? S.translates[idx].parent_xform
                                            : ((!(F & TF_EXT) || kind == NK_EXIT_ROTATE) ? S.rotates[idx].parent_xform : S.scales[idx].parent_xform);
      ray_to_space<(F & TF_EXT) != 0>(S, px, r.wo, r.wd, r.o, r.d);
      ray_set_inv<ORD>(r);
    } else if ((F & TF_MEDIA) && kind == NK_MEDIUM) {
      // constant_medium.rs:41-80
      const DMedium m = S.media[idx];
      const uint32_t outer_mat = r.best_mat;  // the boundary queries below overwrite it
      bool hit_medium = false;
      if (ORD) r.inv = mk(1.0 / r.d.x, 1.0 / r.d.y, 1.0 / r.d.z);  // the boundary queries run the literal form
      // A medium whose boundary holds media itself: the reference's tests of it (m.tests per world.hit, one after the
      // other over the shrinking interval) are run one by one here, because each of them asks the boundary twice and
      // every such query makes the media inside draw afresh; query q of test k is their context 2k + q.  Any other
      // medium is tested once, with the nearest of its m.tests samples (see below).
      const bool nested = (F & TF_MEDIA_ANY) && m.nested != 0u;
      const uint32_t reps = nested ? m.tests : 1u;
      const uint32_t outer_ctx = rng.mctx;
      for (uint32_t rep = 0; rep < reps; rep++) {
      double t1 = INFINITY, t2 = INFINITY;
      nref_t b1 = NREF_NONE, b2 = NREF_NONE;
      if ((F & (TF_MEDIA_ANY | TF_MEDIA_BOX)) && m.bq_count != 0u) {
        // A box (a Group of quads, bare or under instances that hold nothing else): both queries of constant_medium.rs:42-45
        // from one pass over the quads.  A quad's plane distance does not depend on the interval it is asked over
        // (quad.rs:144-157), and the Group keeps the smallest distance inside the interval (group.rs:44-54), so the
        // first query's answer is the smallest distance of all and the second's the smallest one from t1 + 0.0001 on.
        V3 bo = r.o, bd = r.d;
        if ((F & TF_XFORM) && m.bx_inner != NREF_NONE) ray_to_space<(F & TF_EXT) != 0>(S, m.bx_inner, r.wo, r.wd, bo, bd);
        double tq[6];
#pragma unroll
        for (int i = 0; i < 6; i++) {
          tq[i] = NAN;
          if ((uint32_t)i < m.bq_count) {
            double t, al, be;
            cnt.prim++;
            if (quad_hit_od(S.quads[NREF_INDEX(S.gchild[m.bq_first + i])], bo, bd, -INFINITY, INFINITY, t, al, be)) tq[i] = t;
          }
        }
#pragma unroll
        for (int i = 0; i < 6; i++)
          if (tq[i] < t1) { t1 = tq[i]; b1 = m.boundary; }  // (an entry without a hit is NaN and compares false; a hit's distance is finite)
        if (b1 != NREF_NONE) {
          const double lo2 = t1 + 0.0001;
          cnt.prim += m.bq_count;  // (the reference tests the quads again)
#pragma unroll
          for (int i = 0; i < 6; i++)
            if (tq[i] >= lo2 && (b2 == NREF_NONE || tq[i] < t2)) { t2 = tq[i]; b2 = m.boundary; }
        }
      } else if (!(F & TF_MEDIA_ANY) || NREF_KIND(m.boundary) == NK_SPHERE) {
        // the common boundary, a bare sphere: its two queries (constant_medium.rs:42-45) without the general traversal
        const DSphere& bs = S.spheres[NREF_INDEX(m.boundary)];
        double t;
        cnt.prim++;
        if (sphere_hit(bs, r, -INFINITY, INFINITY, t)) {
          t1 = t;
          b1 = m.boundary;
          cnt.prim++;
          if (sphere_hit(bs, r, t1 + 0.0001, INFINITY, t)) {
            t2 = t;
            b2 = m.boundary;
          }
        }
      } else if (F & TF_MEDIA_ANY) {
        if (nested) rng.mctx = 2u * rep;
        traverse_boundary<(F & TF_EXT) != 0>(S, m.boundary, r, -INFINITY, t1, b1, rng, st, cnt);
        if (nested) rng.mctx = 2u * rep + 1u;
        if (b1 != NREF_NONE) traverse_boundary<(F & TF_EXT) != 0>(S, m.boundary, r, t1 + 0.0001, t2, b2, rng, st, cnt);
        rng.mctx = outer_ctx;
      }
      if (b1 != NREF_NONE) {
        if (b2 != NREF_NONE) {
          if (t1 < tmin) t1 = tmin;
          if (t2 > best_t) t2 = best_t;
          if (!(t1 >= t2)) {
            if (t1 < 0.0) t1 = 0.0;
            double ray_length = len(r.d);
            double inside = (t2 - t1) * ray_length;
            // The reference draws one sample per test of the medium and keeps the nearest hit, and it tests a
            // medium more than once per world.hit when the medium is the only object of a BoundingVolumeHierarchy
            // (m.tests).  The nearest of those exponential samples comes from the largest uniform; taking all of
            // them on every visit makes the visit idempotent and independent of the visiting order.  (A medium met
            // inside another's boundary has such a run of keys per query of the enclosing medium: rng.mctx.)
            const uint32_t key0 = m.sample_base + outer_ctx * m.tests;
            double u = rng.keyed(nested ? key0 + rep : key0);
            if (!nested)
              for (uint32_t j = 1; j < m.tests; j++) u = fmax(u, rng.keyed(key0 + j));
            double hit_distance = m.neg_inv_density * log(u);
            if (!(hit_distance > inside)) {
              best_t = t1 + hit_distance / ray_length;
              best = node;
              hit_medium = true;
            }
          }
        }
      }
      }
      r.best_mat = hit_medium ? m.mat : outer_mat;
      if (ORD && hit_medium) r.capf = f32_above(best_t);
    }
    if (!ORD || DEN == 0u || kind != NK_BVH) node = st_pop<ORD>(st, base, r.capf);
    if (YIELD && ORD && __popcll(__ballot(1)) < yield_below) break;  // (the literal form is too heavy to gain from refills)
  }
#undef CHIP_TAKE
}

// ------------------------------------------------------------------ fast shading arithmetic
// The default shade kernels replace f64 divide / sqrt / sin / cos by an f32 seed plus one f64 Newton
// step (relative error ~1e-14) or by f32 evaluation (sample directions only): scattered directions
// then differ from the reference's arithmetic in the last digits, which only matters statistically.
// CHIP_RENDER_EXACT_SHADING selects the variants that keep the reference's f64 operation order.
__device__ inline double fast_rsqrt(double x) {
  double y = (double)__builtin_amdgcn_rsqf((float)x);  // v_rsq_f32: a seed needs no correct rounding
  return y * (1.5 - 0.5 * x * y * y);
}
__device__ inline double fast_rcp(double x) {
  double y = (double)__builtin_amdgcn_rcpf((float)x);  // v_rcp_f32
  return y * (2.0 - x * y);
}
__device__ inline V3 fast_unit(V3 a) { return a * fast_rsqrt(len2(a)); }

__device__ inline double rclamp(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }  // f64::clamp

struct HitOut {
  V3 pt, n;
  double u, v;
  uint32_t mat;
  int32_t leaf_id;
  bool front;
};

// sphere.rs:70-83
__device__ inline void sphere_uv(V3 p, double& u, double& v) {
  double theta = acos(-p.y);
  double phi = atan2(-p.z, p.x) + CHIP_PI;
  u = phi / (2.0 * CHIP_PI);
  v = theta / CHIP_PI;
}

// Builds the HitRecord (object/mod.rs:31-51) of the winning node: recompute the ray in
// the node's space, the record there (sphere.rs:122-134, quad.rs:159-181,
// constant_medium.rs:70-79), then carry it back out through the enclosing
// Translate / Rotate nodes (translate.rs:34, rotate.rs:145-146).
// want_uv = the material's texture reads (u, v); the sphere's acos / atan2 are skipped otherwise.
template <uint32_t F, bool FAST = false>
__device__ inline void reconstruct(const DScene& S, nref_t leaf, double t, V3 wo, V3 wd, double time, HitOut& h, bool want_uv = true) {
  uint32_t kind = NREF_KIND(leaf), idx = NREF_INDEX(leaf);
  nref_t x = NREF_NONE;
  V3 o = wo, d = wd;
  // a sphere below a single Translate: translate.rs:26-38 from the copy of the offset in the sphere's own record
  const bool lone_translate = (F & TF_XFORM) && kind == NK_SPHERE && S.spheres[idx].xoff_only != 0u;
  if (lone_translate) {
    o = wo - ld3(S.spheres[idx].xoff);
  } else if (F & TF_XFORM) {
    if (kind == NK_SPHERE) x = S.spheres[idx].xform;
    else if (kind == NK_QUAD) x = S.quads[idx].xform;
    else if ((F & TF_EXT) && kind == NK_DISC) x = S.discs[idx].xform;
    else if ((F & TF_EXT) && kind == NK_CONE) x = S.cones[idx].xform;
    else x = S.media[idx].xform;
    ray_to_space<(F & TF_EXT) != 0>(S, x, wo, wd, o, d);
  }
  h.pt = o + (t * d);
  if (kind == NK_SPHERE) {
    const DSphere& s = S.spheres[idx];
    V3 center = ld3(s.c) + (time * ld3(s.dc));
    V3 outward = FAST ? (h.pt - center) * fast_rcp(s.r) : (h.pt - center) / s.r;
    h.u = 0.0;
    h.v = 0.0;
    if (want_uv) sphere_uv(outward, h.u, h.v);
    h.front = dot(d, outward) < 0.0;
    h.n = h.front ? outward : -outward;
    h.mat = MATK_ID(s.mat);
    h.leaf_id = s.leaf_id;
  } else if ((F & TF_QUADS) && kind == NK_QUAD) {
    const DQuad& q = S.quads[idx];
    V3 planar = h.pt - ld3(q.q);
    V3 w = ld3(q.w);
    h.u = dot(w, cross(planar, ld3(q.v)));
    h.v = dot(w, cross(ld3(q.u), planar));
    V3 n = ld3(q.n);
    h.front = dot(d, n) < 0.0;
    h.n = h.front ? n : -n;
    h.mat = MATK_ID(q.mat);
    h.leaf_id = q.leaf_id;
  } else if ((F & TF_EXT) && kind == NK_DISC) {  // disc.rs:133-149
    const DDisc& ds = S.discs[idx];
    V3 local_pt = h.pt - ld3(ds.c);
    h.u = (local_pt.x / ds.r + 1.0) * 0.5;
    h.v = (local_pt.z / ds.r + 1.0) * 0.5;
    V3 n = ld3(ds.n);
    h.front = dot(d, n) < 0.0;
    h.n = h.front ? n : -n;
    h.mat = MATK_ID(ds.mat);
    h.leaf_id = ds.leaf_id;
  } else if ((F & TF_EXT) && kind == NK_CONE) {  // cone.rs:245-287
    const DCone& cn = S.cones[idx];
    double k = (cn.r1 - cn.r0) / cn.h;
    double hit_y_local = h.pt.y - cn.base[1];
    double hit_radius = cn.r0 + k * hit_y_local;
    V3 pt_local = h.pt - ld3(cn.base);
    V3 outward = unit(mk(pt_local.x, k * hit_radius, pt_local.z));
    double phi = atan2(pt_local.z, pt_local.x);
    h.u = (phi + CHIP_PI) / (2.0 * CHIP_PI);
    h.v = rclamp((pt_local.y - 0.0) / cn.h, 0.0, 1.0);
    h.front = dot(d, outward) < 0.0;
    h.n = h.front ? outward : -outward;
    h.mat = MATK_ID(cn.mat);
    h.leaf_id = cn.leaf_id;
  } else if (F & TF_MEDIA) {
    const DMedium& m = S.media[idx];
    h.n = mk(1.0, 0.0, 0.0);
    h.u = 0.0;
    h.v = 0.0;
    h.front = true;
    h.mat = MATK_ID(m.mat);
    h.leaf_id = m.leaf_id;
  }
  if (lone_translate) {
    h.pt = h.pt + ld3(S.spheres[idx].xoff);
  } else if (F & TF_XFORM) {
    uint32_t first, n;
    xform_chain<(F & TF_EXT) != 0>(S, x, first, n);
    for (uint32_t i = n; i > 0; i--) {
      nref_t xf = S.xchain[first + i - 1];
      if (NREF_KIND(xf) == NK_TRANSLATE) {
        h.pt = h.pt + ld3(S.translates[NREF_INDEX(xf)].off);
      } else if ((F & TF_EXT) && NREF_KIND(xf) == NK_SCALE) {  // scale.rs:104-113
        const DScale& sc = S.scales[NREF_INDEX(xf)];
        h.pt = mat_mul(sc.m, h.pt);
        h.n = unit(mat_mul(sc.inv, h.n));
      } else {
        const double* m = S.rotates[NREF_INDEX(xf)].m;
        h.pt = mat_mul(m, h.pt);
        h.n = mat_mul(m, h.n);
      }
    }
  }
}

// ------------------------------------------------------------------ textures (texture/*.rs, utils/perlin.rs)
__device__ inline double perlin_noise(const DPerlin& P, V3 pt) {  // perlin.rs:101-124,181-198
  double fx = floor(pt.x), fy = floor(pt.y), fz = floor(pt.z);
  double u = pt.x - fx, v = pt.y - fy, w = pt.z - fz;
  long long i = (long long)fx, j = (long long)fy, k = (long long)fz;
  double uu = u * u * (3.0 - 2.0 * u);
  double vv = v * v * (3.0 - 2.0 * v);
  double ww = w * w * (3.0 - 2.0 * w);
  double acc = 0.0;
  for (int di = 0; di < 2; di++)
    for (int dj = 0; dj < 2; dj++)
      for (int dk = 0; dk < 2; dk++) {
        int idx = P.perm_x[(i + di) & 255] ^ P.perm_y[(j + dj) & 255] ^ P.perm_z[(k + dk) & 255];
        V3 c = ld3(&P.rand_vec[3 * idx]);
        V3 wv = mk(u - (double)di, v - (double)dj, w - (double)dk);
        acc += ((double)di * uu + (1.0 - (double)di) * (1.0 - uu)) * ((double)dj * vv + (1.0 - (double)dj) * (1.0 - vv)) *
               ((double)dk * ww + (1.0 - (double)dk) * (1.0 - ww)) * dot(c, wv);
      }
  return acc;
}
__device__ inline double perlin_turbulence(const DPerlin& P, V3 pt, uint32_t depth) {  // perlin.rs:151-163
  double acc = 0.0, weight = 1.0;
  V3 p = pt;
  for (uint32_t i = 0; i < depth; i++) {
    acc += weight * perlin_noise(P, p);
    weight *= 0.5;
    p = p * 2.0;
  }
  return fabs(acc);
}
// PLAIN: the scene's textures are solid colours and checkers only (told by the host), so the image and noise code, with
// the registers its table look-ups and sine need, is left out of the kernel.
template <bool PLAIN>
__device__ inline V3 tex_value(const DScene& S, int tex, double u, double v, V3 pt) {
  for (int guard = 0; guard < 64; guard++) {
    const DTexture& T = S.textures[tex];
    if (PLAIN && T.kind != TK_SOLID && T.kind != TK_CHECKER) return mk(0, 0, 0);  // (not reached: the host checked)
    switch (T.kind) {
      case TK_SOLID: return mk(T.p[0], T.p[1], T.p[2]);  // solid_color.rs:14-17
      case TK_CHECKER: {                                  // checker_texture.rs:22-35
        long long xi = (long long)floor(T.p[0] * pt.x);
        long long yi = (long long)floor(T.p[0] * pt.y);
        long long zi = (long long)floor(T.p[0] * pt.z);
        tex = ((xi + yi + zi) % 2 == 0) ? (int)T.a : (int)T.b;
        break;
      }
      case TK_IMAGE: {  // image_texture.rs:16-29, image.rs:58-66
        if (PLAIN) return mk(0, 0, 0);
        double uc = rclamp(u, 0.0, 1.0);
        double vc = 1.0 - rclamp(v, 0.0, 1.0);
        uint32_t i = (uint32_t)(uc * (double)T.a);
        uint32_t j = (uint32_t)(vc * (double)T.b);
        if (i >= T.a || j >= T.b) return mk(0.0, 1.0, 1.0);
        const uint8_t* p = S.texels + T.c + ((size_t)j * T.a + i) * 3;
        return mk((double)p[0] / 255.0, (double)p[1] / 255.0, (double)p[2] / 255.0);
      }
      case TK_NOISE: {  // perlin_noise.rs:18-22
        if (PLAIN) return mk(0, 0, 0);
        double s = 0.5 * (1.0 + perlin_noise(S.perlins[T.a], T.p[0] * pt));
        return mk(s, s, s);
      }
      default: {  // TK_TURBULENCE, perlin_turbulence.rs:20-27
        if (PLAIN) return mk(0, 0, 0);
        double s = 0.5 * (1.0 + sin(T.p[0] * pt.z + 10.0 * perlin_turbulence(S.perlins[T.a], pt, T.b)));
        return mk(s, s, s);
      }
    }
  }
  return mk(0, 0, 0);
}

// ------------------------------------------------------------------ light sampling (quad.rs:210-245, sphere.rs:85-95,143-170, group.rs:60-78)
template <bool EXT>
__device__ inline double light_pdf_one(const DScene& S, const DLight& L, V3 origin, V3 dir) {
  TravRay r;
  r.o = origin; r.d = dir; r.time = 0.0;  // Ray::new -> time 0
  r.dd = len2(dir);
  if (L.kind == LK_SPHERE) {
    const DSphere& s = S.spheres[L.index];
    double t;
    if (!sphere_hit(s, r, 0.001, INFINITY, t)) return 0.0;
    double dist2 = len2(ld3(s.c) - origin);
    double cos_theta_max = sqrt(1.0 - s.r * s.r / dist2);
    double solid_angle = 2.0 * CHIP_PI * (1.0 - cos_theta_max);
    return 1.0 / solid_angle;
  } else if (L.kind == LK_QUAD) {
    const DQuad& q = S.quads[L.index];
    double t, al, be;
    if (!quad_hit(q, r, 0.001, INFINITY, t, al, be)) return 0.0;
    V3 n = ld3(q.n);
    V3 hn = dot(dir, n) < 0.0 ? n : -n;
    double distance_squared = t * t * len2(dir);
    double cosine = fabs(dot(dir, hn) / len(dir));
    return distance_squared / (cosine * q.area);
  }
  if (EXT && L.kind == LK_DISC) {  // disc.rs:156-186
    const DDisc& ds = S.discs[L.index];
    double t;
    if (!disc_hit(ds, r, 1e-4, INFINITY, t)) return 0.0;
    V3 n = ld3(ds.n);
    V3 hn = dot(dir, n) < 0.0 ? n : -n;
    double dist_squared = t * t * len2(dir);
    double cosine_abs = fabs(dot(hn, -dir));
    if (cosine_abs < 1e-8) return 0.0;
    double area = CHIP_PI * ds.r * ds.r;
    return dist_squared / (cosine_abs * area);
  }
  if (EXT && L.kind == LK_CONE_WALL) {  // cone.rs:296-313: 1 / lateral area whatever the direction
    const DCone& cn = S.cones[L.index];
    double dr = cn.r1 - cn.r0;
    double l = sqrt(cn.h * cn.h + dr * dr);
    double area = CHIP_PI * (cn.r0 + cn.r1) * l;
    return area > 1e-8 ? 1.0 / area : 0.0;
  }
  return 0.0;  // Node::pdf_value default, object/mod.rs:59-61
}
template <bool EXT>
__device__ inline V3 light_random_one(const DScene& S, const DLight& L, V3 origin, Rng& rng) {
  if (L.kind == LK_SPHERE) {
    const DSphere& s = S.spheres[L.index];
    V3 direction = ld3(s.c) - origin;
    double distance_squared = len2(direction);
    Onb uvw(direction);
    double r1 = rng.next();
    double r2 = rng.next();
    double z = 1.0 + r2 * (sqrt(1.0 - s.r * s.r / distance_squared) - 1.0);
    double phi = 2.0 * CHIP_PI * r1;
    double x = cos(phi) * sqrt(1.0 - z * z);
    double y = sin(phi) * sqrt(1.0 - z * z);
    return uvw.local(mk(x, y, z));
  } else if (L.kind == LK_QUAD) {
    const DQuad& q = S.quads[L.index];
    double r1 = rng.next();
    double r2 = rng.next();
    V3 p = ld3(q.q) + (r1 * ld3(q.u)) + (r2 * ld3(q.v));
    return p - origin;
  }
  if (EXT && L.kind == LK_DISC) {  // disc.rs:68-95,188-193
    const DDisc& ds = S.discs[L.index];
    double r_sq = sqrt(rng.next());
    double phi = 2.0 * CHIP_PI * rng.next();
    double x = r_sq * cos(phi) * ds.r;
    double y = r_sq * sin(phi) * ds.r;
    Onb uvw(ld3(ds.n));
    V3 random_local_pt = (uvw.u * x) + (uvw.v * y);
    V3 target = ld3(ds.c) + random_local_pt;
    return target - origin;
  }
  if (EXT && L.kind == LK_CONE_WALL) {  // cone.rs:315-347: a point of the wall, not a direction from the origin
    const DCone& cn = S.cones[L.index];
    double u1 = 0.0 + (1.0 - 0.0) * rng.next();
    double u2 = 0.0 + (1.0 - 0.0) * rng.next();
    double phi = 2.0 * CHIP_PI * u1;
    double r_rand_sq = cn.r0 * cn.r0 + u2 * (cn.r1 * cn.r1 - cn.r0 * cn.r0);
    double r_rand = sqrt(r_rand_sq);
    double y_local = fabs(cn.r1 - cn.r0) < 1e-8 ? cn.h * u2 : cn.h / (cn.r1 - cn.r0) * (r_rand - cn.r0);
    V3 p_local = mk(r_rand * cos(phi), y_local, r_rand * sin(phi));
    return p_local + ld3(cn.base);
  }
  return mk(1.0, 0.0, 0.0);  // Node::random default, object/mod.rs:63-65
}
template <bool EXT>
__device__ inline double lights_pdf(const DScene& S, V3 origin, V3 dir) {
  if (!S.lights_is_group) return light_pdf_one<EXT>(S, S.lights[0], origin, dir);
  // group.rs:60-69 over the leaves of the lights Group, each with its share (1 / len for a direct member; members of a
  // Group nested inside carry the product of the shares on the way, which regroups the reference's nested sums)
  double sum = 0.0;
  for (uint32_t i = 0; i < S.n_lights; i++) sum += S.lights[i].weight * light_pdf_one<EXT>(S, S.lights[i], origin, dir);
  return sum;
}
template <bool EXT>
__device__ inline V3 lights_random(const DScene& S, V3 origin, Rng& rng) {
  if (!S.lights_is_group) return light_random_one<EXT>(S, S.lights[0], origin, rng);
  // group.rs:71-78: one uniform pick per Group on the way down; an empty Group answers (0, 1, 0)
  uint32_t node = 0;
  for (int depth = 0; depth < CHIP_LIGHT_DEPTH; depth++) {
    const DLightNode g = S.ltree[node];
    if (g.count == 0) return mk(0.0, 1.0, 0.0);
    uint32_t i = (uint32_t)(long long)(rng.next() * (double)g.count);  // rand_int_interval(0, len)
    if (i >= g.count) i = g.count - 1;
    const DLightNode c = S.ltree[g.first + i];
    if (!c.is_group) return light_random_one<EXT>(S, S.lights[c.leaf], origin, rng);
    node = g.first + i;
  }
  return mk(0.0, 1.0, 0.0);  // (not reached: the host refuses deeper nesting)
}

// ------------------------------------------------------------------ queue helpers (wave64 ballot + prefix popcount)
__device__ inline void queue_push(uint32_t* q, uint32_t* counter, bool pred, uint32_t value) {
  unsigned long long mask = __ballot(pred);
  if (mask == 0) return;
  uint32_t lane = __lane_id();
  uint32_t leader = __ffsll((long long)mask) - 1;
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
  base = __shfl(base, leader);
  if (pred) {
    uint32_t rank = __popcll(mask & ((1ull << lane) - 1ull));
    q[base + rank] = value;
  }
}

__device__ inline float fclampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }  // keeps NaN like f64::clamp

// The radiance map of a path that is still alive has no constant term: a material that scatters emits nothing
// (material/mod.rs default `emitted`), and the one that emits (diffuse_light.rs:28-43) ends the path.
struct Affine {
  float b[3], bound[3];
  uint32_t clamped;  // 0 or CHIP_MAP_CLAMPED
};
__device__ inline float affine_lo(const Affine& f, int c) { return f.clamped ? fminf(0.f, f.bound[c]) : -CHIP_BIG; }
__device__ inline float affine_hi(const Affine& f, int c) { return f.clamped ? fmaxf(0.f, f.bound[c]) : CHIP_BIG; }
// One bounce of camera.rs:274-304 as the per-channel map x -> w x (specular, camera.rs:274-276: klo = -CHIP_BIG) or
// x -> clamp(w x, 0, khi) (the PDF branch, camera.rs:304: klo = 0), composed on the right of the path's map.  The image of
// [0, khi] under x -> b x is the interval between 0 and b khi, clipped by the clamp the path already carries.
__device__ inline void affine_compose(Affine& f, const float w[3], float klo, float khi) {
  const bool clamps = klo > -CHIP_BIG;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    if (clamps) f.bound[c] = fclampf(f.b[c] * khi, affine_lo(f, c), affine_hi(f, c));
    f.b[c] = f.b[c] * w[c];
  }
  if (clamps) f.clamped = CHIP_MAP_CLAMPED;
}
__device__ inline float affine_apply(const Affine& f, int c, float x) {
  const float v = f.b[c] * x;
  return f.clamped ? fclampf(v, fminf(0.f, f.bound[c]), fmaxf(0.f, f.bound[c])) : v;
}
__device__ inline void finish_path(const PathPool& P, uint32_t slot, const Affine& f, const float x[3]) {
  // (the id travelled through memory: one that does not name a slot of the pool is reported, not written through)
  if (slot < P.cap) P.sample[slot] = make_float4(affine_apply(f, 0, x[0]), affine_apply(f, 1, x[1]), affine_apply(f, 2, x[2]), 0.f);
  else atomicOr(&P.errflag[1], 1u);
}
// 16-byte accesses of a ray record
__device__ inline void ray_load(const PathRay* base, uint32_t entry, bool r48, V3& o, V3& d, double& time, uint32_t& id) {
  if (r48) {
    const uint4* p = reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(base) + (size_t)entry * CHIP_RAY48_BYTES);
    const uint4 a = p[0], b = p[1], c = p[2];
    o = mk(u2d(a.x, a.y), u2d(a.z, a.w), u2d(b.x, b.y));
    d = mk((double)__uint_as_float(b.z), (double)__uint_as_float(b.w), (double)__uint_as_float(c.x));
    id = c.y;
    time = u2d(c.z, c.w);
  } else {
    const PathRay* rec = base + entry;
    const double2* p = reinterpret_cast<const double2*>(rec);
    double2 a = p[0], b = p[1], c = p[2];
    const uint4 e = reinterpret_cast<const uint4*>(rec)[3];
    o = mk(a.x, a.y, b.x);
    d = mk(b.y, c.x, c.y);
    time = u2d(e.x, e.y);
    id = e.z;
  }
}
// Marks entry `entry` of a ray array empty (an entry reserved for a path that then ended).
__device__ inline void ray_mark_dead(PathRay* base, uint32_t entry, bool r48) {
  if (r48) reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(base) + (size_t)entry * CHIP_RAY48_BYTES)[2] = make_uint4(0u, CHIP_DEAD_ID, 0u, 0u);
  else reinterpret_cast<uint4*>(base + entry)[3] = make_uint4(0u, 0u, CHIP_DEAD_ID, 0u);
}

// ---- quad-cooperative record access
// A lane that reads its own 64-byte record with four 16-byte loads touches the same cache line four times
// in a row, and its stores leave the CU as four partial-line writes.  Instead the four lanes of a quad
// move one whole record per instruction (lane c takes 16-byte chunk c), one record of the quad after the
// other, and the chunks change hands through a per-wave LDS staging buffer, so that every lane ends up
// with (or starts from) the chunks of its own record.  All 64 lanes must be active in these calls; a
// lane without a record passes CHIP_NO_SLOT.
// Lane 4q + m owns wave entry 16m + q, so that access j of a quad (the record of quad lane j) covers wave
// entries 16j .. 16j + 15 across the wave: consecutive entries, consecutive records.
#define CHIP_NO_SLOT 0xFFFFFFFFu
#define CHIP_XPOSE_U4 260  // uint4 per wave: 64 records of 4 chunks plus one chunk of padding per 16 records
template <int CTRL>
__device__ inline uint32_t dpp_quad(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
template <int J>
__device__ inline uint32_t quad_bcast(uint32_t v) { return dpp_quad<J * 0x55>(v); }
struct QuadRecord {
  uint32_t w[4][4];  // [chunk][dword]
};
// Staging position (in uint4) of chunk 0 of wave entry e: the padding makes both the quad-wise and the
// lane-wise accesses conflict-free (16 consecutive lanes always cover the 16 bank groups).
__device__ inline uint32_t xpose_at(uint32_t e) { return 4u * e + (e >> 4); }
// Ranks the lanes of a wave that pass `take` within a shared counter, in the order in which the
// quad-cooperative stores visit them (quad lane 0 of every quad first, then quad lane 1, ...), so that
// each of the four store instructions writes a contiguous run of records.  All lanes must be active.
__device__ inline uint32_t quad_order_rank(bool take, uint32_t lane, uint32_t* counter) {
  const unsigned long long m = __ballot(take);
  const unsigned long long g0 = 0x1111111111111111ull;
  const uint32_t c0 = (uint32_t)__popcll(m & g0), c1 = (uint32_t)__popcll(m & (g0 << 1)), c2 = (uint32_t)__popcll(m & (g0 << 2));
  const uint32_t j = lane & 3u;
  const uint32_t before = j == 0 ? 0u : (j == 1 ? c0 : (j == 2 ? c0 + c1 : c0 + c1 + c2));
  const uint32_t within = (uint32_t)__popcll(m & (g0 << j) & ((1ull << lane) - 1ull));
  uint32_t base = 0;
  if (m) {
    if (lane == 0) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = __builtin_amdgcn_readfirstlane(base);
  }
  return base + before + within;
}

// Loads the first NCH 16-byte chunks of record `slot` (STRIDE bytes apart) of every lane.
template <int STRIDE, int NCH>
__device__ inline void quad_load(const void* base, uint32_t slot, uint32_t lane, uint4* stage, QuadRecord& r) {
  const uint32_t c = lane & 3, q = lane >> 2;
  uint32_t s[4] = {quad_bcast<0>(slot), quad_bcast<1>(slot), quad_bcast<2>(slot), quad_bcast<3>(slot)};
  asm volatile("" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]));  // (keeps the cross-lane reads here, where every lane is active: see quad_pair_slots)
  uint4 v[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    v[j] = make_uint4(0, 0, 0, 0);
    if (s[j] != CHIP_NO_SLOT && (NCH == 4 || c < NCH))
      v[j] = *reinterpret_cast<const uint4*>(static_cast<const char*>(base) + (size_t)s[j] * STRIDE + c * 16);
  }
#pragma unroll
  for (int j = 0; j < 4; j++) stage[xpose_at(16u * j + q) + c] = v[j];
  __builtin_amdgcn_wave_barrier();
  const uint32_t own = xpose_at(16u * c + q);
#pragma unroll
  for (int k = 0; k < NCH; k++) {
    const uint4 t = stage[own + k];
    r.w[k][0] = t.x; r.w[k][1] = t.y; r.w[k][2] = t.z; r.w[k][3] = t.w;
  }
  __builtin_amdgcn_wave_barrier();
}
// The same in two steps, for a caller that has work to do while the loads are in flight: the loads ...
template <int STRIDE, int NCH>
__device__ inline void quad_load_issue(const void* base, uint32_t slot, uint32_t lane, uint4 (&v)[4]) {
  const uint32_t c = lane & 3;
  uint32_t s[4] = {quad_bcast<0>(slot), quad_bcast<1>(slot), quad_bcast<2>(slot), quad_bcast<3>(slot)};
  asm volatile("" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]));  // (keeps the cross-lane reads here, where every lane is active: see quad_pair_slots)
  // (each load under its own condition, as in quad_load: measured faster than loads without branches, and the
  // branches keep the compiler from moving the loads down to their first use)
#pragma unroll
  for (int j = 0; j < 4; j++) {
    v[j] = make_uint4(0, 0, 0, 0);
    if (s[j] != CHIP_NO_SLOT && (NCH == 4 || c < NCH))
      v[j] = *reinterpret_cast<const uint4*>(static_cast<const char*>(base) + (size_t)s[j] * STRIDE + c * 16);
  }
}
// ... and the exchange through the staging buffer that hands every lane the chunks of its own record.
template <int NCH>
__device__ inline void quad_load_finish(const uint4 (&v)[4], uint32_t lane, uint4* stage, QuadRecord& r) {
  const uint32_t c = lane & 3, q = lane >> 2;
#pragma unroll
  for (int j = 0; j < 4; j++) stage[xpose_at(16u * j + q) + c] = v[j];
  __builtin_amdgcn_wave_barrier();
  const uint32_t own = xpose_at(16u * c + q);
#pragma unroll
  for (int k = 0; k < NCH; k++) {
    const uint4 t = stage[own + k];
    r.w[k][0] = t.x; r.w[k][1] = t.y; r.w[k][2] = t.z; r.w[k][3] = t.w;
  }
  __builtin_amdgcn_wave_barrier();
}
// Stores the first NCH chunks of the record of every lane whose slot is not CHIP_NO_SLOT.
template <int STRIDE, int NCH>
__device__ inline void quad_store(void* base, uint32_t slot, uint32_t lane, uint4* stage, const QuadRecord& r) {
  const uint32_t c = lane & 3, q = lane >> 2;
  const uint32_t own = xpose_at(16u * c + q);
#pragma unroll
  for (int k = 0; k < NCH; k++) stage[own + k] = make_uint4(r.w[k][0], r.w[k][1], r.w[k][2], r.w[k][3]);
  __builtin_amdgcn_wave_barrier();
  uint32_t s[4] = {quad_bcast<0>(slot), quad_bcast<1>(slot), quad_bcast<2>(slot), quad_bcast<3>(slot)};
  asm volatile("" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]));  // (keeps the cross-lane reads here, where every lane is active: see quad_pair_slots)
#pragma unroll
  for (int j = 0; j < 4; j++) {
    if (s[j] != CHIP_NO_SLOT && (NCH == 4 || c < NCH))
      *reinterpret_cast<uint4*>(static_cast<char*>(base) + (size_t)s[j] * STRIDE + c * 16) = stage[xpose_at(16u * j + q) + c];
  }
  __builtin_amdgcn_wave_barrier();
}
// The same for 32-byte records (two chunks): one instruction moves two records per quad, lanes 0 and 1 of the quad the
// chunks of one, lanes 2 and 3 those of the next, so two instructions instead of four cover the wave.
// The slots of the two records a lane helps to move: those of quad lanes 0 and 2 for the lower half of the quad, of quad
// lanes 1 and 3 for the upper half.  All four cross-lane reads happen here, with every lane of the wave active, and the
// empty asm pins them to this point: written as `upper ? quad_bcast<1>(slot) : quad_bcast<0>(slot)` the compiler turns the
// selection into branches and moves each read into its branch, where the lane it reads from is switched off and reads as 0.
__device__ inline void quad_pair_slots(uint32_t slot, bool upper, uint32_t& s0, uint32_t& s1) {
  uint32_t b0 = quad_bcast<0>(slot), b1 = quad_bcast<1>(slot), b2 = quad_bcast<2>(slot), b3 = quad_bcast<3>(slot);
  asm volatile("" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
  s0 = upper ? b1 : b0;
  s1 = upper ? b3 : b2;
}
__device__ inline void quad_load32_issue(const void* base, uint32_t slot, uint32_t lane, uint4 (&v)[2]) {
  const uint32_t c = lane & 1u, hi = lane & 2u;
  uint32_t s0, s1;
  quad_pair_slots(slot, hi != 0u, s0, s1);
  v[0] = make_uint4(0, 0, 0, 0);
  v[1] = make_uint4(0, 0, 0, 0);
  if (s0 != CHIP_NO_SLOT) v[0] = *reinterpret_cast<const uint4*>(static_cast<const char*>(base) + (size_t)s0 * 32 + c * 16);
  if (s1 != CHIP_NO_SLOT) v[1] = *reinterpret_cast<const uint4*>(static_cast<const char*>(base) + (size_t)s1 * 32 + c * 16);
}
__device__ inline void quad_load32_finish(const uint4 (&v)[2], uint32_t lane, uint4* stage, QuadRecord& r) {
  const uint32_t c = lane & 1u, m = (lane >> 1) & 1u, q = lane >> 2;
  stage[xpose_at(16u * m + q) + c] = v[0];
  stage[xpose_at(16u * (2u + m) + q) + c] = v[1];
  __builtin_amdgcn_wave_barrier();
  const uint32_t own = xpose_at(16u * (lane & 3u) + q);
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const uint4 t = stage[own + k];
    r.w[k][0] = t.x; r.w[k][1] = t.y; r.w[k][2] = t.z; r.w[k][3] = t.w;
  }
  __builtin_amdgcn_wave_barrier();
}
__device__ inline void quad_load32(const void* base, uint32_t slot, uint32_t lane, uint4* stage, QuadRecord& r) {
  uint4 v[2];
  quad_load32_issue(base, slot, lane, v);
  quad_load32_finish(v, lane, stage, r);
}
__device__ inline void quad_store32(void* base, uint32_t slot, uint32_t lane, uint4* stage, const QuadRecord& r) {
  const uint32_t c = lane & 1u, hi = lane & 2u, m = (lane >> 1) & 1u, q = lane >> 2;
  const uint32_t own = xpose_at(16u * (lane & 3u) + q);
#pragma unroll
  for (int k = 0; k < 2; k++) stage[own + k] = make_uint4(r.w[k][0], r.w[k][1], r.w[k][2], r.w[k][3]);
  __builtin_amdgcn_wave_barrier();
  uint32_t s0, s1;
  quad_pair_slots(slot, hi != 0u, s0, s1);
  if (s0 != CHIP_NO_SLOT) *reinterpret_cast<uint4*>(static_cast<char*>(base) + (size_t)s0 * 32 + c * 16) = stage[xpose_at(16u * m + q) + c];
  if (s1 != CHIP_NO_SLOT) *reinterpret_cast<uint4*>(static_cast<char*>(base) + (size_t)s1 * 32 + c * 16) = stage[xpose_at(16u * (2u + m) + q) + c];
  __builtin_amdgcn_wave_barrier();
}
static_assert(sizeof(PathMap) == 32, "the 32-byte record accessors move the radiance map");
__device__ inline void d2u(double v, uint32_t& lo, uint32_t& hi) {
  lo = (uint32_t)__double2loint(v);
  hi = (uint32_t)__double2hiint(v);
}
__device__ inline void ray_unpack(const QuadRecord& q, V3& o, V3& d, double& time, uint32_t& id) {
  o = mk(u2d(q.w[0][0], q.w[0][1]), u2d(q.w[0][2], q.w[0][3]), u2d(q.w[1][0], q.w[1][1]));
  d = mk(u2d(q.w[1][2], q.w[1][3]), u2d(q.w[2][0], q.w[2][1]), u2d(q.w[2][2], q.w[2][3]));
  time = u2d(q.w[3][0], q.w[3][1]);
  id = q.w[3][2];
}
__device__ inline void ray_pack(QuadRecord& q, V3 o, V3 d, double time, uint32_t id) {
  d2u(o.x, q.w[0][0], q.w[0][1]); d2u(o.y, q.w[0][2], q.w[0][3]);
  d2u(o.z, q.w[1][0], q.w[1][1]); d2u(d.x, q.w[1][2], q.w[1][3]);
  d2u(d.y, q.w[2][0], q.w[2][1]); d2u(d.z, q.w[2][2], q.w[2][3]);
  d2u(time, q.w[3][0], q.w[3][1]); q.w[3][2] = id; q.w[3][3] = 0;
}
// the 48-byte form: [o.x o.y] [o.z | d.x d.y as floats] [d.z as float, id | time]
__device__ inline void ray_unpack48(const QuadRecord& q, V3& o, V3& d, double& time, uint32_t& id) {
  o = mk(u2d(q.w[0][0], q.w[0][1]), u2d(q.w[0][2], q.w[0][3]), u2d(q.w[1][0], q.w[1][1]));
  d = mk((double)__uint_as_float(q.w[1][2]), (double)__uint_as_float(q.w[1][3]), (double)__uint_as_float(q.w[2][0]));
  id = q.w[2][1];
  time = u2d(q.w[2][2], q.w[2][3]);
}
__device__ inline void ray_pack48(QuadRecord& q, V3 o, V3 d, double time, uint32_t id) {
  d2u(o.x, q.w[0][0], q.w[0][1]); d2u(o.y, q.w[0][2], q.w[0][3]);
  d2u(o.z, q.w[1][0], q.w[1][1]);
  q.w[1][2] = __float_as_uint((float)d.x); q.w[1][3] = __float_as_uint((float)d.y);
  q.w[2][0] = __float_as_uint((float)d.z); q.w[2][1] = id;
  d2u(time, q.w[2][2], q.w[2][3]);
}
__device__ inline void affine_unpack(const QuadRecord& q, Affine& f, uint32_t& id) {
  id = q.w[0][0] & ~CHIP_MAP_CLAMPED; f.clamped = q.w[0][0] & CHIP_MAP_CLAMPED;
  f.b[0] = __uint_as_float(q.w[0][1]); f.b[1] = __uint_as_float(q.w[0][2]); f.b[2] = __uint_as_float(q.w[0][3]);
  f.bound[0] = __uint_as_float(q.w[1][0]); f.bound[1] = __uint_as_float(q.w[1][1]); f.bound[2] = __uint_as_float(q.w[1][2]);
}
__device__ inline void affine_pack(QuadRecord& q, const Affine& f, uint32_t id) {
  q.w[0][0] = id | f.clamped; q.w[0][1] = __float_as_uint(f.b[0]); q.w[0][2] = __float_as_uint(f.b[1]); q.w[0][3] = __float_as_uint(f.b[2]);
  q.w[1][0] = __float_as_uint(f.bound[0]); q.w[1][1] = __float_as_uint(f.bound[1]); q.w[1][2] = __float_as_uint(f.bound[2]); q.w[1][3] = 0u;
}
__device__ inline void affine_identity(Affine& f) {
#pragma unroll
  for (int c = 0; c < 3; c++) { f.b[c] = 1.f; f.bound[c] = 0.f; }
  f.clamped = 0u;
}
// The radiance map of every lane's entry of `bounce` (the identity before the first bounce, never stored).
template <bool B0>
__device__ inline void quad_load_map(const PathPool& P, uint32_t entry, uint32_t lane, uint32_t bounce, uint4* stage, Affine& f, uint32_t& id) {
  if (B0) {
    affine_identity(f);
    id = entry;  // at bounce 0 entry i is path i
  } else {
    QuadRecord q;
    quad_load32(P.st[bounce & 1], entry, lane, stage, q);
    affine_unpack(q, f, id);
  }
}

// n / d for a batch-constant divisor d through its precomputed reciprocal c = ceil(2^64 / d): the high 64
// bits of c * n are exactly floor(n / d) for every 32-bit n (c = 0 stands for d = 1).
__host__ __device__ inline uint32_t magic_div(uint32_t n, const uint32_t c[2]) {
  if ((c[0] | c[1]) == 0u) return n;
  const unsigned long long lo = (unsigned long long)c[0] * n, hi = (unsigned long long)c[1] * n;
  return (uint32_t)((hi + (lo >> 32)) >> 32);
}
// Position inside the batch's rectangle of the p-th path of a stratum (see BatchDesc::tiles_per_row).
__device__ inline void rect_pixel(const BatchDesc& B, uint32_t p, uint32_t& xl, uint32_t& yl) {
  if (B.tiles_per_row) {
    const uint32_t t = p >> 6, i = p & 63u;
    const uint32_t trow = magic_div(t, B.tiles_magic), tcol = t - trow * B.tiles_per_row;
    xl = tcol * 8u + (i & 7u);
    yl = trow * 8u + (i >> 3);
  } else {
    yl = magic_div(p, B.rect_w_magic);
    xl = p - yl * B.rect_w;
  }
}
__device__ inline void slot_to_path(const BatchDesc& B, const DCamera& cam, uint32_t slot, uint32_t& x, uint32_t& y, uint32_t& pixel,
                                    uint32_t& stratum) {
  const uint32_t sb = magic_div(slot, B.npix_magic), p = slot - sb * B.npix;
  uint32_t xl, yl;
  rect_pixel(B, p, xl, yl);
  y = B.rect_y0 + yl;
  x = B.rect_x0 + xl;
  pixel = y * cam.image_width + x;
  stratum = B.stratum_first + sb * B.stratum_stride;
}

// camera.rs:359-374,389-394,416-419 with vector.rs:115-126
// `tries` is the number of defocus-disk candidates the rejection loop consumed (vector.rs:115-126).  On entry 0 means
// "run the loop" (the count is returned); a count from an earlier evaluation of the same ray makes this one go
// straight to the accepted candidate, which in a wave saves the trips the slowest lane needs.
__device__ inline void camera_ray(const DCamera& cam, Rng& rng, uint32_t x, uint32_t y, uint32_t s_x, uint32_t s_y, V3& o, V3& d,
                                  double& time, uint32_t& tries) {
  double px = (((double)s_x + rng.next()) * cam.recip_sqrt_spp) - 0.5;
  double py = (((double)s_y + rng.next()) * cam.recip_sqrt_spp) - 0.5;
  V3 pixel_sample = ld3(cam.pixel00) + (((double)x + px) * ld3(cam.du)) + (((double)y + py) * ld3(cam.dv));
  if (cam.defocus_angle <= 0.0) {
    o = ld3(cam.center);
  } else {
    double dx = 0, dy = 0;
    if (tries != 0) {
      rng.draw += 2u * (tries - 1u);
      dx = rng.interval(-1.0, 1.0);
      dy = rng.interval(-1.0, 1.0);
    } else {
      for (int it = 0; it < CHIP_REJECT_CAP; it++) {
        dx = rng.interval(-1.0, 1.0);
        dy = rng.interval(-1.0, 1.0);
        tries++;
        double xs = dx * dx, ys = dy * dy, zs = 0.0 * 0.0;
        if (xs + ys + zs < 1.0) break;
      }
    }
    o = ld3(cam.center) + (dx * ld3(cam.disk_u)) + (dy * ld3(cam.disk_v));
  }
  d = pixel_sample - o;
  time = rng.next();
}

// The camera ray of path `slot` of a batch (camera.rs:359-374).  Camera rays are never stored: the
// traversal and the shading of bounce 0 both compute them from the path's Philox stream, which costs a
// few hundred VALU operations against 192 bytes of HBM traffic per path.
__device__ inline void path_camera_ray(const DScene& S, const BatchDesc& B, uint32_t slot, V3& o, V3& d, double& time, uint32_t& tries) {
  uint32_t x, y, pixel, stratum;
  slot_to_path(B, S.cam, slot, x, y, pixel, stratum);
  Rng rng;
  rng.init(B.seed_lo, B.seed_hi, pixel, stratum, CHIP_CAMERA_BOUNCE, 0);
  const uint32_t s_y = magic_div(stratum, B.sqrt_spp_magic), s_x = stratum - s_y * S.cam.sqrt_spp;
  camera_ray(S.cam, rng, x, y, s_x, s_y, o, d, time, tries);
}
// Starts a batch: every path is live at bounce 0 (entry i is path i; nothing is stored for it yet).
__global__ void k_begin_batch(PathPool P, BatchDesc B) {
  if (blockIdx.x == 0 && threadIdx.x == 0) P.counters[0] = B.n_paths;
}

// One world.hit (camera.rs:264) per live path: closest-hit traversal; stores (t, node) per path.
// Waves are persistent: they reserve chunks of the bounce's live queue through a shared cursor.
// ORD instantiations run the ordered traversal and hand the rays it does not take (zero direction
// component) to the literal queue; the !ORD instantiations replay the reference's order, either for the
// whole live queue (ConstantMedium scenes, CHIP_RENDER_REFERENCE_ORDER) or, with lit_pass, for that
// literal queue.
#ifndef CHIP_EXTEND_WAVES
#define CHIP_EXTEND_WAVES 4
#endif
// What the traversal kernel keeps in LDS besides the per-lane stacks.
struct CacheDesc {
  uint32_t n_bvh, n_sph;     // leading BVH nodes / spheres copied into LDS (0 = no cache)
  uint32_t stack_entries;    // per-lane stack entries kept in LDS
};
// Threads per block of the traversal kernel: with the LDS scene cache one block owns a whole CU's LDS.  Twelve
// waves (768 threads) leave each of them 168 VGPRs, which the variants with rotated quads (whose test keeps a whole
// quad record in registers) and with media need to stay out of scratch memory; the other variants fit (or nearly
// fit) sixteen waves' 128 and run faster with the extra waves than without the few spills.
#ifndef CHIP_EXT_BLOCK_QUADROT
#define CHIP_EXT_BLOCK_QUADROT 768
#endif
#ifndef CHIP_EXT_BLOCK_MEDIA
#define CHIP_EXT_BLOCK_MEDIA 768
#endif
__host__ __device__ constexpr int extend_block(uint32_t F, bool CACHE) {
  return CACHE ? ((F & TF_MEDIA_ANY) ? 512 : ((F & TF_MEDIA) ? CHIP_EXT_BLOCK_MEDIA : (((F & TF_XFORM) && !(F & TF_NO_QUADROT)) ? CHIP_EXT_BLOCK_QUADROT : 1024))) : CHIP_BLOCK;
}
// Waves per SIMD the traversal kernel is register-allocated for.  The LDS-cached sphere-scene kernel asks for 6
// (at most 80 VGPRs) although only 4 of its own waves fit beside its LDS: the registers it leaves free are
// what lets two shade waves per SIMD of the other batch run next to it.
#ifndef CHIP_EXT0_WAVES
#define CHIP_EXT0_WAVES 5
#endif
__host__ __device__ constexpr int extend_min_waves(uint32_t F, bool CACHE) {
  return CACHE ? (F == 0 ? CHIP_EXT0_WAVES : 1) : ((F & TF_MEDIA) ? 1 : ((F & TF_XFORM) ? 2 : CHIP_EXTEND_WAVES));
}
// B0: the instantiation for bounce 0 (camera rays are computed, not loaded), kept apart so that the
// camera code does not cost the other bounces registers.
template <uint32_t F, bool CACHE, bool ORD, bool B0>
__global__ void __launch_bounds__(extend_block(F, CACHE), extend_min_waves(F, CACHE)) k_extend(DScene S, PathPool P, BatchDesc B, uint32_t bounce_arg, CacheDesc cd, uint32_t lit_pass) {
  const uint32_t bounce = B0 ? 0u : bounce_arg;
  // LDS layout: [7 x n_bvh uint4: culling nodes][4 x n_sph uint4: spheres][(stack_entries + 1) x blockDim (value, distance) pairs]
  const NodeCache C{0u, (uint32_t)CHIP_WNODE_U4 * cd.n_bvh, cd.n_bvh, cd.n_sph, cd.n_bvh, cd.n_sph};
  uint2* lds_stack = reinterpret_cast<uint2*>(g_lds + (CHIP_WNODE_U4 * cd.n_bvh + 4u * cd.n_sph));
  uint32_t* ctr = P.counters + (size_t)bounce * CHIP_CTR_STRIDE;
  const uint32_t n_live = lit_pass ? ctr[CHIP_CTR_LIT] : ctr[0];
  if ((size_t)blockIdx.x * 64u >= n_live) return;  // the grid is sized from an upper bound: surplus blocks leave at once
  if (CACHE) {
    for (uint32_t i = threadIdx.x; i < cd.n_bvh; i += blockDim.x) {
      const uint4* p = reinterpret_cast<const uint4*>(S.wnodes + i);
#pragma unroll
      for (int k = 0; k < CHIP_WNODE_U4; k++) g_lds[(uint32_t)CHIP_WNODE_U4 * i + k] = p[k];
    }
    const uint32_t so = (uint32_t)CHIP_WNODE_U4 * cd.n_bvh;
    for (uint32_t i = threadIdx.x; i < cd.n_sph; i += blockDim.x) {
      const uint4* p = reinterpret_cast<const uint4*>(S.spheres + i);
      g_lds[so + i] = p[0]; g_lds[so + cd.n_sph + i] = p[1]; g_lds[so + 2 * cd.n_sph + i] = p[2]; g_lds[so + 3 * cd.n_sph + i] = p[3];
    }
    __syncthreads();
  }
  const PathRay* ray_in = P.ray[bounce & 1];
  uint32_t* cursor = lit_pass ? &ctr[CHIP_CTR_LIT_CURSOR] : &ctr[CHIP_CTR_CURSOR];
  const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane = __lane_id();
  TravStack st;
  st.lds = (lds_u2_t)lds_stack + threadIdx.x;
  st.ovf = P.ovf + (size_t)gtid * (2 * CHIP_STACK_OVF);
  st.overflow = 0;
  st.redo = 0;
  st.sp = 0;
  st.lds_cap = (int)cd.stack_entries;
  st.stride8 = 8u * blockDim.x;
  TravCounters cnt{0, 0};
  uint32_t nseg = 0;
  bool has = false;  // this lane holds a ray (traversing or finished)
  uint32_t entry = 0;
  TravRay r;
  r.wo = r.wd = r.o = r.d = r.inv = mk(0, 0, 0);
  r.time = 0;
  double best_t = INFINITY;
  nref_t best = NREF_NONE, node = NREF_NONE;
  Rng rng;
  rng.init(B.seed_lo, B.seed_hi, 0, 0, bounce, 0);
  // chunk of the queue a wave reserves at a time: large enough to keep the shared cursor cold,
  // small enough to leave every wave of the grid some work
  const uint32_t n_waves = gridDim.x * (blockDim.x / 64);
  uint32_t chunk = (n_live / (n_waves * 4u)) & ~63u;
  chunk = chunk < 64u ? 64u : (chunk > 2048u ? 2048u : chunk);
  uint32_t w_next = 0, w_end = 0;  // wave-uniform
  bool exhausted = false;
  for (;;) {
    // 1. lanes whose traversal finished store (t, node); the HitRecord is rebuilt where it is consumed
    if (has && node == NREF_NONE) {
      if (ORD && st.redo) {  // handed back by the ordered traversal: goes to the literal queue instead
        st.redo = 0;
        P.q_lit[atomicAdd(&ctr[CHIP_CTR_LIT], 1u)] = entry;
      } else {
        nseg++;
        PathHit ph;
        ph.t = best_t;
        ph.node = best;
        ph.info = (r.best_mat & 0xFFFFFFu) | (B0 ? (rng.draw << 24) : 0u);
        *reinterpret_cast<uint4*>(P.hit + entry) = *reinterpret_cast<const uint4*>(&ph);
      }
      has = false;
    }
    // 2. idle lanes take the next entries of the wave's current chunk of the queue; a new chunk is
    //    reserved from the shared cursor (one atomic per chunk) when the current one is used up
    {
      const unsigned long long need = __ballot(!has);
      if (need) {
        if (w_next >= w_end && !exhausted) {
          uint32_t base = 0;
          if (lane == 0) base = atomicAdd(cursor, chunk);
          base = __shfl(base, 0);
          w_next = base;
          w_end = min(base + chunk, n_live);
          if (base >= n_live) { exhausted = true; w_end = w_next; }
        }
        const uint32_t rank = __popcll(need & ((1ull << lane) - 1ull));
        const uint32_t avail = w_end - w_next;
        if (!has && rank < avail) {
          entry = lit_pass ? P.q_lit[w_next + rank] : w_next + rank;
          uint32_t id = entry;  // at bounce 0 entry i is path i
          uint32_t tries = 0;
          if (B0) path_camera_ray(S, B, entry, r.wo, r.wd, r.time, tries);
          else ray_load(ray_in, entry, B.ray48 != 0u, r.wo, r.wd, r.time, id);
          const bool dead = !B0 && id == CHIP_DEAD_ID;  // reserved by the last shading pass but left empty
          if (dead) *reinterpret_cast<uint4*>(P.hit + entry) = make_uint4(0u, 0u, NREF_DEAD, 0u);
          // the culling form is computed first: it also tells whether the ordered traversal can take the ray
          r.o = r.wo; r.d = r.wd;
          const bool ordinary = ray_set_inv<ORD>(r);
#ifdef CHIP_TEST_HAND_OVER  // (test build: every CHIP_TEST_HAND_OVER-th ray takes the hand-over to the literal traversal)
          const bool to_literal = !dead && ORD && (B.reference_order || !ordinary || entry % CHIP_TEST_HAND_OVER == 0u);
#else
          const bool to_literal = !dead && ORD && (B.reference_order || !ordinary);
#endif
          if (ORD) {  // the literal queue takes these rays (appended with one atomic per wave)
            const unsigned long long ml = __ballot(to_literal);
            if (ml) {
              const uint32_t leader = (uint32_t)__ffsll((long long)ml) - 1u;
              uint32_t lbase = 0;
              if (lane == leader) lbase = atomicAdd(&ctr[CHIP_CTR_LIT], (uint32_t)__popcll(ml));
              lbase = __shfl(lbase, leader);
              if (to_literal) P.q_lit[lbase + (uint32_t)__popcll(ml & ((1ull << lane) - 1ull))] = entry;
            }
          }
          if (!to_literal && !dead) {
            r.capf = INFINITY;
            if (F & TF_MEDIA) {
              uint32_t x, y, pixel, stratum;
              slot_to_path(B, S.cam, id, x, y, pixel, stratum);
              rng.init(B.seed_lo, B.seed_hi, pixel, stratum, bounce, 0);
            }
            if (B0) rng.draw = tries;  // rides to the hit record: the bounce-0 shade pass rebuilds the camera ray from it
            best_t = INFINITY;
            best = NREF_NONE;
            node = ORD ? S.ord_root : S.root;
            st.sp = 0;
            has = true;
          }
        }
        w_next += min((uint32_t)__popcll(need), avail);
      }
    }
    if (!__any(has) && exhausted) break;
    // 3. traverse until the rays are done or too few lanes are still busy
    if (has) traverse<F, true, CACHE, ORD>(S, C, node, r, 0.001, best_t, best, rng, st, 0, cnt,
                                           (exhausted && w_next >= w_end) ? 0 : (B0 ? ((F & TF_NO_QUADROT) ? CHIP_REFILL_THRESH_B0_INSTANCES : CHIP_REFILL_THRESH_B0) : CHIP_REFILL_THRESH));
  }
  if (st.overflow) atomicOr(&P.errflag[0], 1u);
  // per-wave reduction of the traversal counters
  for (int off = 32; off > 0; off >>= 1) {
    cnt.bvh += __shfl_down(cnt.bvh, off);
    cnt.prim += __shfl_down(cnt.prim, off);
    nseg += __shfl_down(nseg, off);
  }
  if (lane == 0 && nseg) {
    atomicAdd(&P.stats[0], (unsigned long long)nseg);
    atomicAdd(&P.stats[1], (unsigned long long)cnt.bvh);
    atomicAdd(&P.stats[2], (unsigned long long)cnt.prim);
  }
}


// ------------------------------------------------------------------ compact form for small sphere scenes
// A scene of bare spheres whose whole culling tree and sphere table are LDS-resident, with fewer than 1023 of either
// (the RTIOW scenes), is traversed with 11-bit node references: bit 10 says "sphere", the low bits are the index,
// 0x7FF stands for none.  A child's sort key then carries the reference itself in its low 11 bits (the culling
// record's child words are stored as such tags), under the entry distance truncated to its upper 21 bits: picking the
// nearest child is one AND instead of a three-select pick by slot number, a deferred child is pushed as its key, one
// 4-byte stack word, and a popped word is dropped when its (rounded-down) distance lies beyond the closest hit.
// Same visits up to the order of children whose distances agree to 12 bits, same hits, same ties (tie_wins).
#define CHIP_SM_SPHERE 0x400u
#define CHIP_SM_NONE 0x7FFu
#define CHIP_SM_MAX 1023u  // nodes / spheres such a scene may hold (index 0x3FF with the sphere bit would read as none)
typedef __attribute__((address_space(3))) uint32_t* lds_u1_t;
// The stack lives in LDS only: a ray whose descent would need more entries than that is flagged and traced by the
// literal pass instead (the caller hands it over like a ray with a zero direction component), which keeps every
// push free of branches.
struct SmallStack {
  lds_u1_t lds;      // this lane's column: one 4-byte word per entry
  int sp, lds_cap;
  uint32_t stride4;  // bytes between consecutive entries of one lane
  uint32_t overflow;
};
__host__ __device__ inline uint32_t small_ref(nref_t x) {
  if (x == NREF_NONE) return CHIP_SM_NONE;
  return (NREF_KIND(x) == NK_SPHERE ? CHIP_SM_SPHERE : 0u) | (NREF_INDEX(x) & 0x3FFu);
}
// Writes the key into the next free entry and keeps it only if it names a child (a culled child's key is all ones):
// no branch.  The caller has made sure that up to three entries are free (sm_room).
__device__ inline void sm_push(SmallStack& s, uint32_t key) {
  *(lds_u1_t)((lds_b_t)s.lds + __umul24((uint32_t)s.sp, s.stride4)) = key;
  s.sp += key != 0xFFFFFFFFu ? 1 : 0;
}
__device__ inline void sm_room(SmallStack& s) {
  const int lim = s.lds_cap - 3;
  s.overflow |= s.sp > lim ? 1u : 0u;
  s.sp = min(s.sp, lim);
}
__device__ inline uint32_t sm_pop(SmallStack& s, float capf) {
  for (;;) {
    if (s.sp <= 0) return CHIP_SM_NONE;
    s.sp--;
    const uint32_t key = *(lds_u1_t)((lds_b_t)s.lds + __umul24((uint32_t)s.sp, s.stride4));
    if (__uint_as_float(key & ~CHIP_SM_NONE) <= capf) return key & CHIP_SM_NONE;
  }
}
// The ordered closest-hit loop of traverse<0, ...> on compact references (see there for the descent scheduling).
template <bool YIELD>
__device__ void traverse_small(const DScene& S, const NodeCache& C, uint32_t& node, TravRay& r, double tmin, double& best_t, nref_t& best,
                               SmallStack& st, TravCounters& cnt, int yield_below) {
  constexpr uint32_t DEN = CHIP_DESCENT_DEN_SPHERES;
  for (;;) {
    for (;;) {
      const bool inner = node < CHIP_SM_SPHERE;
      const uint32_t n_inner = (uint32_t)__popcll(__ballot(inner)), n_here = (uint32_t)__popcll(__ballot(1));
      if (n_inner == 0u || n_inner * DEN < n_here) break;
      if (!inner) continue;
      const uint32_t rec = __umul24(node, 16u * CHIP_WNODE_U4) + (C.bvh_off << 4);
      const uint32_t nx = rec + r.far16[0], fx = rec - r.far16[0];
      const uint32_t ny = rec + r.far16[1], fy = rec - r.far16[1];
      const uint32_t nz = rec + r.far16[2], fz = rec - r.far16[2];
      const uint4 qnx = lds_read4_at(nx), qfx = lds_read4_at(fx + 16u), qny = lds_read4_at(ny + 32u), qfy = lds_read4_at(fy + 48u),
                  qnz = lds_read4_at(nz + 64u), qfz = lds_read4_at(fz + 80u), tag = lds_read4_at(rec + 96u);
      cnt.bvh++;
      float near_x[4], far_x[4], near_y[4], far_y[4], near_z[4], far_z[4];
      u4_to_f4(qnx, near_x); u4_to_f4(qfx, far_x); u4_to_f4(qny, near_y); u4_to_f4(qfy, far_y); u4_to_f4(qnz, near_z); u4_to_f4(qfz, far_z);
      const uint32_t tg[4] = {tag.x, tag.y, tag.z, tag.w};
      uint32_t key[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const float x0 = fmaf(near_x[k], r.m_n[0], r.c_n[0]), x1 = fmaf(far_x[k], r.m_f[0], r.c_f[0]);
        const float y0 = fmaf(near_y[k], r.m_n[1], r.c_n[1]), y1 = fmaf(far_y[k], r.m_f[1], r.c_f[1]);
        const float z0 = fmaf(near_z[k], r.m_n[2], r.c_n[2]), z1 = fmaf(far_z[k], r.m_f[2], r.c_f[2]);
        const float en = fmaxf(__builtin_fmaxf(__builtin_fmaxf(x0, y0), z0), CHIP_CULL_TMIN);
        const float ex = fminf(__builtin_fminf(__builtin_fminf(x1, y1), z1), r.capf);
        key[k] = en <= ex ? ((__float_as_uint(en) & ~CHIP_SM_NONE) | tg[k]) : 0xFFFFFFFFu;
      }
#define CHIP_CSWAP(a, b) { const uint32_t lo_ = min(key[a], key[b]), hi_ = max(key[a], key[b]); key[a] = lo_; key[b] = hi_; }
      CHIP_CSWAP(0, 1) CHIP_CSWAP(2, 3) CHIP_CSWAP(0, 2) CHIP_CSWAP(1, 3) CHIP_CSWAP(1, 2)
#undef CHIP_CSWAP
      // (a node without a child in reach has four all-ones keys: the pushes keep nothing and the next node is "none")
      sm_room(st);
      if (__any(key[2] != 0xFFFFFFFFu)) {  // three or four children in reach: rare enough to be worth the branch
        sm_push(st, key[3]);
        sm_push(st, key[2]);
      }
      sm_push(st, key[1]);
      node = key[0] & CHIP_SM_NONE;
      if (node == CHIP_SM_NONE) node = sm_pop(st, r.capf);
    }
    if (node == CHIP_SM_NONE) break;
    if (node >= CHIP_SM_SPHERE) {  // (a lane still at a culling node sits this round out)
      const uint32_t idx = node & 0x3FFu;
      cnt.prim++;
      SphereRec sp;
      {  // every sphere of such a scene is in LDS: no "cached?" test, no path to memory
        const uint32_t i = C.sph_off + idx;
        const uint4 a = lds_read4(i), b = lds_read4(i + C.cap_sph), c = lds_read4(i + 2 * C.cap_sph), d = lds_read4(i + 3 * C.cap_sph);
        sp.c[0] = u2d(a.x, a.y); sp.c[1] = u2d(a.z, a.w);
        sp.c[2] = u2d(b.x, b.y); sp.dc[0] = u2d(b.z, b.w);
        sp.dc[1] = u2d(c.x, c.y); sp.dc[2] = u2d(c.z, c.w);
        sp.r = u2d(d.x, d.y);
        sp.mat = d.z;
      }
      double t;
      if (sphere_hit_incl(sp, r, tmin, best_t, t)) {
        const nref_t ref = NREF(NK_SPHERE, idx);
        if (t < best_t || tie_wins(S, ref, best)) {
          best_t = t;
          best = ref;
          r.best_mat = sp.mat;
          r.capf = f32_above(best_t);
        }
      }
      node = sm_pop(st, r.capf);
    }
    if (YIELD && __popcll(__ballot(1)) < yield_below) break;
  }
}

// k_extend<0, true, true, B0> on compact references: the same queue handling (chunks, refills, literal hand-over),
// the LDS copy of a culling record with its child words turned into tags, 4-byte stack words.
template <bool B0>
__global__ void __launch_bounds__(extend_block(0, true), extend_min_waves(0, true)) k_extend_small(DScene S, PathPool P, BatchDesc B, uint32_t bounce_arg, CacheDesc cd) {
  const uint32_t bounce = B0 ? 0u : bounce_arg;
  // LDS layout: [7 x n_bvh uint4: culling records][4 x n_sph uint4: spheres][stack_entries x blockDim 4-byte words]
  const NodeCache C{0u, (uint32_t)CHIP_WNODE_U4 * cd.n_bvh, cd.n_bvh, cd.n_sph, cd.n_bvh, cd.n_sph};
  uint32_t* ctr = P.counters + (size_t)bounce * CHIP_CTR_STRIDE;
  const uint32_t n_live = ctr[0];
  if ((size_t)blockIdx.x * 64u >= n_live) return;
  for (uint32_t i = threadIdx.x; i < cd.n_bvh; i += blockDim.x) {
    const uint4* p = reinterpret_cast<const uint4*>(S.wnodes + i);
#pragma unroll
    for (int k = 0; k < 6; k++) g_lds[(uint32_t)CHIP_WNODE_U4 * i + k] = p[k];
    const uint4 ch = p[6];
    g_lds[(uint32_t)CHIP_WNODE_U4 * i + 6u] = make_uint4(small_ref(ch.x), small_ref(ch.y), small_ref(ch.z), small_ref(ch.w));
  }
  const uint32_t so = (uint32_t)CHIP_WNODE_U4 * cd.n_bvh;
  for (uint32_t i = threadIdx.x; i < cd.n_sph; i += blockDim.x) {
    const uint4* p = reinterpret_cast<const uint4*>(S.spheres + i);
    g_lds[so + i] = p[0]; g_lds[so + cd.n_sph + i] = p[1]; g_lds[so + 2 * cd.n_sph + i] = p[2]; g_lds[so + 3 * cd.n_sph + i] = p[3];
  }
  __syncthreads();
  const PathRay* ray_in = P.ray[bounce & 1];
  uint32_t* cursor = &ctr[CHIP_CTR_CURSOR];
  const uint32_t lane = __lane_id();
  SmallStack st;
  st.lds = (lds_u1_t)(g_lds + (CHIP_WNODE_U4 * cd.n_bvh + 4u * cd.n_sph)) + threadIdx.x;
  st.overflow = 0;
  st.sp = 0;
  st.lds_cap = (int)cd.stack_entries;
  st.stride4 = 4u * blockDim.x;
  TravCounters cnt{0, 0};
  uint32_t nseg = 0;
  bool has = false;
  uint32_t entry = 0, draws = 0;
  TravRay r;
  r.wo = r.wd = r.o = r.d = r.inv = mk(0, 0, 0);
  r.time = 0;
  double best_t = INFINITY;
  nref_t best = NREF_NONE;
  uint32_t node = CHIP_SM_NONE;
  const uint32_t root = small_ref(S.ord_root);
  const uint32_t n_waves = gridDim.x * (blockDim.x / 64);
  uint32_t chunk = (n_live / (n_waves * 4u)) & ~63u;
  chunk = chunk < 64u ? 64u : (chunk > 2048u ? 2048u : chunk);
  uint32_t w_next = 0, w_end = 0;
  bool exhausted = false;
  for (;;) {
    if (has && node == CHIP_SM_NONE) {
      if (st.overflow) {  // the descent ran out of stack: the literal pass traces this ray
        st.overflow = 0;
        P.q_lit[atomicAdd(&ctr[CHIP_CTR_LIT], 1u)] = entry;
      } else {
        nseg++;
        PathHit ph;
        ph.t = best_t;
        ph.node = best;
        ph.info = (r.best_mat & 0xFFFFFFu) | (B0 ? (draws << 24) : 0u);
        *reinterpret_cast<uint4*>(P.hit + entry) = *reinterpret_cast<const uint4*>(&ph);
      }
      has = false;
    }
    {
      const unsigned long long need = __ballot(!has);
      if (need) {
        if (w_next >= w_end && !exhausted) {
          uint32_t base = 0;
          if (lane == 0) base = atomicAdd(cursor, chunk);
          base = __shfl(base, 0);
          w_next = base;
          w_end = min(base + chunk, n_live);
          if (base >= n_live) { exhausted = true; w_end = w_next; }
        }
        const uint32_t rank = __popcll(need & ((1ull << lane) - 1ull));
        const uint32_t avail = w_end - w_next;
        if (!has && rank < avail) {
          entry = w_next + rank;
          uint32_t id = entry;
          uint32_t tries = 0;
          if (B0) path_camera_ray(S, B, entry, r.wo, r.wd, r.time, tries);
          else ray_load(ray_in, entry, B.ray48 != 0u, r.wo, r.wd, r.time, id);
          const bool dead = !B0 && id == CHIP_DEAD_ID;
          if (dead) *reinterpret_cast<uint4*>(P.hit + entry) = make_uint4(0u, 0u, NREF_DEAD, 0u);
          r.o = r.wo; r.d = r.wd;
          const bool ordinary = ray_set_inv<true>(r);
#ifdef CHIP_TEST_HAND_OVER
          const bool to_literal = !dead && (B.reference_order || !ordinary || entry % CHIP_TEST_HAND_OVER == 0u);
#else
          const bool to_literal = !dead && (B.reference_order || !ordinary);
#endif
          const unsigned long long ml = __ballot(to_literal);
          if (ml) {
            const uint32_t leader = (uint32_t)__ffsll((long long)ml) - 1u;
            uint32_t lbase = 0;
            if (lane == leader) lbase = atomicAdd(&ctr[CHIP_CTR_LIT], (uint32_t)__popcll(ml));
            lbase = __shfl(lbase, leader);
            if (to_literal) P.q_lit[lbase + (uint32_t)__popcll(ml & ((1ull << lane) - 1ull))] = entry;
          }
          if (!to_literal && !dead) {
            r.capf = INFINITY;
            if (B0) draws = tries;
            best_t = INFINITY;
            best = NREF_NONE;
            node = root;
            st.sp = 0;
            has = true;
          }
        }
        w_next += min((uint32_t)__popcll(need), avail);
      }
    }
    if (!__any(has) && exhausted) break;
    if (has) traverse_small<true>(S, C, node, r, 0.001, best_t, best, st, cnt, (exhausted && w_next >= w_end) ? 0 : (B0 ? CHIP_REFILL_THRESH_B0 : CHIP_REFILL_THRESH));
  }
  for (int off = 32; off > 0; off >>= 1) {
    cnt.bvh += __shfl_down(cnt.bvh, off);
    cnt.prim += __shfl_down(cnt.prim, off);
    nseg += __shfl_down(nseg, off);
  }
  if (lane == 0 && nseg) {
    atomicAdd(&P.stats[0], (unsigned long long)nseg);
    atomicAdd(&P.stats[1], (unsigned long long)cnt.bvh);
    atomicAdd(&P.stats[2], (unsigned long long)cnt.prim);
  }
}

// ------------------------------------------------------------------ flat form for scenes of a few primitives
// A world of a few spheres and quads (DScene::flat_list: the Cornell box and its kin) is not walked but tried whole:
// every lane takes one ray through the same list, in step, so the primitive's record comes through scalar loads and
// no lane waits for another one's node kind (the hierarchy walk of such a scene runs at a third of its lanes).  The
// tests and the arithmetic are those of the ordered traversal: inclusive intervals, exact ties settled by the
// reference's visiting order, the ray taken into an instance's space by replaying its transform chain; a ray with a
// direction component that is zero (in the world or in an instance's space) goes to the literal pass, as it does
// there, because the reference's box tests decide by the sign of such a zero.
#define CHIP_FLAT_BLOCK 256
// The scene tables are read-only while a kernel runs, but the compiler cannot know that (the kernel also stores through
// pointers) and would fetch a record with one vector load per lane although every lane reads the same one.  Read through
// a pointer to the constant address space, a record at a wave-uniform index comes through the scalar unit into scalar
// registers, once per wave.
#define CHIP_KCONST(T, ptr) (reinterpret_cast<const __attribute__((address_space(4))) T*>(reinterpret_cast<uintptr_t>(ptr)))
// ray_to_space for Translates and Rotates, on scalar loads
__device__ inline void flat_to_space(const DScene& S, nref_t x, V3 wo, V3 wd, V3& o, V3& d) {
  o = wo; d = wd;
  uint32_t first, n;
  if (NREF_KIND(x) == NK_TRANSLATE) { first = CHIP_KCONST(DTranslate, S.translates)[NREF_INDEX(x)].chain_first; n = CHIP_KCONST(DTranslate, S.translates)[NREF_INDEX(x)].chain_len; }
  else { first = CHIP_KCONST(DRotate, S.rotates)[NREF_INDEX(x)].chain_first; n = CHIP_KCONST(DRotate, S.rotates)[NREF_INDEX(x)].chain_len; }
  for (uint32_t i = 0; i < n; i++) {
    const nref_t c = CHIP_KCONST(nref_t, S.xchain)[first + i];
    if (NREF_KIND(c) == NK_TRANSLATE) {
      const auto* t = CHIP_KCONST(DTranslate, S.translates) + NREF_INDEX(c);
      o = o - mk(t->off[0], t->off[1], t->off[2]);
    } else {
      const auto* t = CHIP_KCONST(DRotate, S.rotates) + NREF_INDEX(c);
      double m[9];
#pragma unroll
      for (int k = 0; k < 9; k++) m[k] = t->m[k];
      o = mat_mulT(m, o);
      d = mat_mulT(m, d);
    }
  }
}
template <bool B0>
__global__ void __launch_bounds__(CHIP_FLAT_BLOCK) k_extend_flat(DScene S, PathPool P, BatchDesc B, uint32_t bounce_arg) {
  const uint32_t bounce = B0 ? 0u : bounce_arg;
  uint32_t* ctr = P.counters + (size_t)bounce * CHIP_CTR_STRIDE;
  const uint32_t n_live = ctr[0];
  const PathRay* ray_in = P.ray[bounce & 1];
  const bool r48 = B.ray48 != 0u;
  const uint32_t n_flat = S.n_flat;
  uint32_t nseg = 0, ntest = 0;
  for (uint32_t entry = blockIdx.x * blockDim.x + threadIdx.x; entry < n_live; entry += gridDim.x * blockDim.x) {
    TravRay r;
    uint32_t id = entry, tries = 0;
    if (B0) path_camera_ray(S, B, entry, r.wo, r.wd, r.time, tries);
    else ray_load(ray_in, entry, r48, r.wo, r.wd, r.time, id);
    if (!B0 && id == CHIP_DEAD_ID) {  // reserved by the last shading pass but left empty
      *reinterpret_cast<uint4*>(P.hit + entry) = make_uint4(0u, 0u, NREF_DEAD, 0u);
      continue;
    }
    bool literal = has_zero_component(r.wd);
#ifdef CHIP_TEST_HAND_OVER
    literal = literal || entry % CHIP_TEST_HAND_OVER == 0u;
#endif
    r.o = r.wo; r.d = r.wd;
    r.dd = len2(r.d);
    double best_t = INFINITY;
    nref_t best = NREF_NONE;
    uint32_t best_mat = 0;
    for (uint32_t i = 0; i < n_flat; i++) {
      const auto* fe = CHIP_KCONST(DFlatPrim, S.flat_list) + i;
      DFlatPrim e;
      e.ref = fe->ref; e.xform = fe->xform; e.enter = fe->enter;
      if (e.enter) {
        flat_to_space(S, e.xform, r.wo, r.wd, r.o, r.d);
        literal = literal || has_zero_component(r.d);
        r.dd = len2(r.d);
      }
      const uint32_t idx = NREF_INDEX(e.ref);
      double t;
      bool hit;
      uint32_t matw;
      if (NREF_KIND(e.ref) == NK_SPHERE) {
        const auto* rec = CHIP_KCONST(DSphere, S.spheres) + idx;
        SphereRec sp;
#pragma unroll
        for (int k = 0; k < 3; k++) { sp.c[k] = rec->c[k]; sp.dc[k] = rec->dc[k]; }
        sp.r = rec->r;
        sp.mat = rec->mat;
        hit = sphere_hit_incl(sp, r, 0.001, best_t, t);
        matw = sp.mat;
      } else {
        const auto* rec = CHIP_KCONST(DQuad, S.quads) + idx;
        DQuad q;
#pragma unroll
        for (int k = 0; k < 3; k++) { q.q[k] = rec->q[k]; q.u[k] = rec->u[k]; q.v[k] = rec->v[k]; q.n[k] = rec->n[k]; q.w[k] = rec->w[k]; }
        q.d = rec->d;
        q.mat = rec->mat;
        double al, be;
        hit = quad_hit_od(q, r.o, r.d, 0.001, best_t, t, al, be);
        matw = q.mat;
      }
      if (hit && (t < best_t || tie_wins(S, e.ref, best))) {
        best_t = t;
        best = e.ref;
        best_mat = matw;
      }
    }
    if (literal) {
      P.q_lit[atomicAdd(&ctr[CHIP_CTR_LIT], 1u)] = entry;
    } else {
      nseg++;
      ntest += n_flat;
      PathHit ph;
      ph.t = best_t;
      ph.node = best;
      ph.info = (best_mat & 0xFFFFFFu) | (B0 ? (tries << 24) : 0u);
      *reinterpret_cast<uint4*>(P.hit + entry) = *reinterpret_cast<const uint4*>(&ph);
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    ntest += __shfl_down(ntest, off);
    nseg += __shfl_down(nseg, off);
  }
  if (__lane_id() == 0 && nseg) {
    atomicAdd(&P.stats[0], (unsigned long long)nseg);
    atomicAdd(&P.stats[2], (unsigned long long)ntest);
  }
}

// One hit of one path: Material::emitted / scatter / scattering_pdf of material kind KIND plus the PDF
// branch of camera.rs:278-305.  The segment's ray (ro, rd, rtime) and closest hit (ph) come from the
// caller.  Returns true with the next ray and the bounce's map x -> clamp(xw x, klo, khi) if the material
// scattered, false with the radiance xw that ends the path otherwise; the caller composes either with the
// path's radiance map, which therefore need not occupy registers here.
// LEAN is the variant for scenes made only of spheres with solid-colour textures and no lights (no
// transforms, quads, media, image / noise / checker textures or light sampling to carry in registers);
// LEAN with EXT is the same for spheres under Translate / Rotate instances.
// shade_material: the material part, on a HitRecord the caller has built, with the path's random stream named by
// (pixel, stratum); shade_hit: the same after rebuilding the HitRecord from (ray, t, node) and the path slot.
template <uint32_t KIND, bool FAST, bool LEAN, bool EXT, bool PLAIN>
__device__ inline bool shade_material(const DScene& S, const BatchDesc& B, uint32_t bounce, uint32_t pixel, uint32_t stratum, const DMaterial& M,
                                      const HitOut& hrec, V3 rd, float (&xw)[3], float& klo_out, float& khi_out, V3& next_o, V3& next_d) {
  bool cont = false;
  xw[0] = 0.f; xw[1] = 0.f; xw[2] = 0.f;
  const bool has_lights = !LEAN && S.has_lights;
  const bool front = hrec.front;
  const V3 pt = hrec.pt;
  if (KIND == MK_DIFFUSE_LIGHT) {
    // diffuse_light.rs:28-43: no scatter; emits on the front face only
    if (front) {
      V3 e = LEAN ? ld3(M.p) : tex_value<PLAIN>(S, M.tex, hrec.u, hrec.v, pt);
      xw[0] = (float)e.x; xw[1] = (float)e.y; xw[2] = (float)e.z;
    }
  } else if (KIND == MK_EMPTY) {
    // empty.rs:18-27 + default emitted: zero radiance
  } else {
    const V3 nrm = hrec.n;
    Rng rng;
    rng.init(B.seed_lo, B.seed_hi, pixel, stratum, bounce, 0);  // (the traversal draws nothing from the stream: media samples are keyed)
    V3 ndir;
    bool terminal = false;
    float w[3] = {0.f, 0.f, 0.f};
    float klo = -CHIP_BIG, khi = CHIP_BIG;
    if (KIND == MK_METAL) {  // metal.rs:20-28
      V3 reflected = reflect(rd, nrm);
      ndir = (FAST ? fast_unit(reflected) : unit(reflected)) + (M.p[3] * random_unit<FAST>(rng));
      w[0] = (float)M.p[0]; w[1] = (float)M.p[1]; w[2] = (float)M.p[2];
    } else if (KIND == MK_DIELECTRIC) {  // dielectric.rs:30-52
      double ri = front ? 1.0 / M.p[0] : M.p[0];
      V3 ud = FAST ? fast_unit(rd) : unit(rd);
      double cos_theta = fmin(dot(-ud, nrm), 1.0);
      double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
      bool cannot_refract = ri * sin_theta > 1.0;
      bool refl = cannot_refract;
      if (!refl) {
        double r0 = (1.0 - ri) / (1.0 + ri);
        r0 = r0 * r0;
        double x1 = 1.0 - cos_theta;
        double x2 = x1 * x1;
        double x4 = x2 * x2;
        double reflectance = r0 + (1.0 - r0) * (x4 * x1);
        refl = reflectance > rng.next();
      }
      ndir = refl ? reflect(ud, nrm) : refract(ud, nrm, ri);
      w[0] = 1.f; w[1] = 1.f; w[2] = 1.f;
    } else {
      // Lambertian (lambertian.rs:29-49) / Isotropic (isotropic.rs:29-44): PDF branch, camera.rs:278-305
      V3 atten = LEAN ? ld3(M.p) : tex_value<PLAIN>(S, M.tex, hrec.u, hrec.v, pt);  // solid_color.rs:14-17 (colour copied into the material)
      bool use_light = false;
      if (has_lights) use_light = rng.next() < 0.5;  // mixture.rs:26-32
      if (FAST) {
        // same estimator with the fast arithmetic above; the hit normal is already unit length
        V3 ow = nrm, ou = nrm, ov = nrm;
        double local_z = 0.0;
        if (KIND == MK_LAMBERTIAN) {
          V3 a = fabs(ow.x) > 0.9 ? mk(0.0, 1.0, 0.0) : mk(1.0, 0.0, 0.0);
          ov = fast_unit(cross(ow, a));
          ou = cross(ow, ov);
        }
        if (use_light) {
          ndir = lights_random<EXT>(S, pt, rng);
        } else if (KIND == MK_LAMBERTIAN) {
          float r1 = (float)rng.next(), r2 = (float)rng.next();
          float sn, cs;
          sincospif(2.0f * r1, &sn, &cs);
          float sq = sqrtf(r2);
          local_z = (double)sqrtf(1.0f - r2);
          ndir = ((double)(cs * sq) * ou) + ((double)(sn * sq) * ov) + (local_z * ow);
        } else {
          ndir = random_unit<true>(rng);
        }
        double cos_n = 0.0;  // cosine between the scattered direction and the normal
        if (KIND == MK_LAMBERTIAN) cos_n = (use_light || has_lights) ? dot(fast_unit(ndir), ow) : local_z;
        double mat_pdf = KIND == MK_LAMBERTIAN ? fmax(cos_n * (1.0 / CHIP_PI), 0.0) : 1.0 / (4.0 * CHIP_PI);
        double pv = mat_pdf;
        if (has_lights) pv = 0.5 * lights_pdf<EXT>(S, pt, ndir) + 0.5 * mat_pdf;
        if (pv < 0.05) {
          terminal = true;
        } else {
          double sp = KIND == MK_LAMBERTIAN ? (cos_n < 0.0 ? 0.0 : cos_n * (1.0 / CHIP_PI)) : 1.0 / (4.0 / CHIP_PI);
          double k = (KIND == MK_LAMBERTIAN && !has_lights) ? 1.0 : sp * fast_rcp(pv);  // cosine sampling alone: sp == pv
          w[0] = (float)(atten.x * k); w[1] = (float)(atten.y * k); w[2] = (float)(atten.z * k);
          klo = 0.f; khi = 10.f;
        }
      } else {
      Onb uvw(nrm);                                    // cosine.rs:10-14 (unused for Isotropic)
      if (use_light) ndir = lights_random<EXT>(S, pt, rng);
      else if (KIND == MK_LAMBERTIAN) ndir = uvw.local(random_cosine_direction(rng));  // cosine.rs:24-27
      else ndir = random_unit<false>(rng);                                            // probability_density_function/sphere.rs:24-26
      double mat_pdf;
      if (KIND == MK_LAMBERTIAN) mat_pdf = fmax(dot(unit(ndir), uvw.w) / CHIP_PI, 0.0);  // cosine.rs:18-22
      else mat_pdf = 1.0 / (4.0 * CHIP_PI);                                              // sphere.rs:20-22
      double pv = mat_pdf;
      if (has_lights) {  // mixture.rs:20-24
        double v0 = 0.5 * lights_pdf<EXT>(S, pt, ndir);
        double v1 = 0.5 * mat_pdf;
        pv = v0 + v1;
      }
      if (pv < 0.05) {  // camera.rs:291-293: returns the emission (zero for these materials)
        terminal = true;
      } else {
        double sp;
        if (KIND == MK_LAMBERTIAN) {
          double cos_theta = dot(nrm, unit(ndir));
          sp = cos_theta < 0.0 ? 0.0 : cos_theta / CHIP_PI;
        } else {
          sp = 1.0 / (4.0 / CHIP_PI);  // isotropic.rs:43
        }
        w[0] = (float)(atten.x * sp / pv); w[1] = (float)(atten.y * sp / pv); w[2] = (float)(atten.z * sp / pv);
        klo = 0.f; khi = 10.f;  // camera.rs:304
      }
      }
    }
    if (!terminal) {  // (a terminal path returns the emission, zero for these materials: camera.rs:291-293)
      xw[0] = w[0]; xw[1] = w[1]; xw[2] = w[2];
      klo_out = klo; khi_out = khi;
      next_o = pt;  // the caller stores the map and the next ray (its time never changes, camera.rs:287)
      next_d = ndir;
      cont = true;
    }
  }
  return cont;
}
template <uint32_t KIND, bool FAST, bool LEAN, bool EXT, bool PLAIN>
__device__ inline bool shade_hit(const DScene& S, const BatchDesc& B, uint32_t bounce, uint32_t slot, const PathHit& ph, V3 ro, V3 rd,
                                 double rtime, float (&xw)[3], float& klo_out, float& khi_out, V3& next_o, V3& next_d) {
  // HitRecord of the segment's closest hit (object/mod.rs:31-51), rebuilt from (ray, t, node)
  const DMaterial& M = S.materials[MATK_ID(ph.info)];
  HitOut hrec;
  reconstruct<LEAN ? (EXT ? TF_XFORM : 0u) : (TF_QUADS | TF_XFORM | TF_MEDIA | (EXT ? TF_EXT : 0u)), FAST>(S, ph.node, ph.t, ro, rd, rtime, hrec, !LEAN && !PLAIN && (M.flags & MF_NEEDS_UV) != 0);
  uint32_t x_ = 0, y_ = 0, pixel = 0, stratum = 0;
  if (KIND != MK_DIFFUSE_LIGHT && KIND != MK_EMPTY) slot_to_path(B, S.cam, slot, x_, y_, pixel, stratum);
  return shade_material<KIND, FAST, LEAN, EXT, PLAIN>(S, B, bounce, pixel, stratum, M, hrec, rd, xw, klo_out, khi_out, next_o, next_d);
}

// Material-binned shading of one bounce.  A block takes a tile of the bounce's entries, bins them by the
// material kind of their hit in LDS (a counting sort; misses and empty entries last), and shades the tile
// kind after kind, so a wave runs one material's code except where two kinds meet.  Misses take the
// background (camera.rs:264-266).  The tile reserves room for every entry that can scatter in the next
// bounce's arrays with one global atomic, the paths that go on write their new ray and radiance map
// there, and the few reserved entries left over (paths ended by camera.rs:291-293) are marked empty.
#ifndef CHIP_SHADE_TILE
#define CHIP_SHADE_TILE 2048  // (512 / 1024 / 2048 / 4096 measured on the sphere scene: shading 283 / 267 / 247 / 252 ms per frame over both streams)
#endif
// B0: the instantiation for bounce 0, whose entries are the batch's paths in order with their camera rays
// recomputed and the identity as radiance map; kept apart so that neither variant pays registers for the other.
// `pass`: which entries of the bounce this launch shades.
// SHADE_LIT_QUEUE (bounce 0 only): the entries are those the literal queue of the bounce names, for the form of bounce 0
// in which the traversal kernel shades its own hits (k_first_small) and only the rays it handed to the literal pass are
// left.
enum : uint32_t { SHADE_ALL = 0, SHADE_PLAIN_ONLY = 1, SHADE_RICH_ONLY = 2, SHADE_LIT_QUEUE = 3 };
// TBL: the scene's record tables are small enough to sit in LDS (a Cornell box: 13 primitives, 6 materials, 2 lights).
// Shading a hit follows a chain of dependent reads (hit -> primitive -> its transforms -> material -> texture -> light list
// -> light's primitive), each a trip to L2 when the tables are in memory; a block of this form copies them into LDS
// first and shades from a DScene whose table pointers name the copies, so every link of the chain is an LDS read.
#define CHIP_SHADE_TBL_BYTES 12288u
struct ShadeTables {  // records per table (the host checked that they fit CHIP_SHADE_TBL_BYTES, see shade_tables_fit)
  uint32_t n_spheres, n_quads, n_translates, n_rotates, n_xchain, n_media, n_materials, n_textures, n_lights, n_ltree;
  uint32_t n_discs, n_cones, n_scales;  // (scenes with these run the EXT instantiation)
};
__host__ __device__ inline uint32_t tbl_up(uint32_t v, uint32_t a) { return (v + a - 1u) & ~(a - 1u); }
// byte offsets of the tables inside the LDS copy, each at its record type's alignment; returns the total
__host__ __device__ inline uint32_t shade_tables_layout(const ShadeTables& n, uint32_t (&off)[13]) {
  uint32_t at = 0;
  off[0] = at = tbl_up(at, alignof(DSphere)); at += n.n_spheres * (uint32_t)sizeof(DSphere);
  off[1] = at = tbl_up(at, alignof(DTranslate)); at += n.n_translates * (uint32_t)sizeof(DTranslate);
  off[2] = at = tbl_up(at, 16u); at += n.n_quads * (uint32_t)sizeof(DQuad);
  off[3] = at = tbl_up(at, 16u); at += n.n_rotates * (uint32_t)sizeof(DRotate);
  off[4] = at = tbl_up(at, 16u); at += n.n_media * (uint32_t)sizeof(DMedium);
  off[5] = at = tbl_up(at, 16u); at += n.n_materials * (uint32_t)sizeof(DMaterial);
  off[6] = at = tbl_up(at, 16u); at += n.n_textures * (uint32_t)sizeof(DTexture);
  off[7] = at = tbl_up(at, 16u); at += n.n_lights * (uint32_t)sizeof(DLight);
  off[8] = at = tbl_up(at, 16u); at += n.n_ltree * (uint32_t)sizeof(DLightNode);
  off[9] = at = tbl_up(at, 16u); at += n.n_xchain * (uint32_t)sizeof(nref_t);
  off[10] = at = tbl_up(at, 16u); at += n.n_discs * (uint32_t)sizeof(DDisc);
  off[11] = at = tbl_up(at, 16u); at += n.n_cones * (uint32_t)sizeof(DCone);
  off[12] = at = tbl_up(at, 16u); at += n.n_scales * (uint32_t)sizeof(DScale);
  return tbl_up(at, 16u);
}
template <typename T>
__device__ inline const T* shade_table_copy(unsigned char* lds, uint32_t off, const T* src, uint32_t n) {
  static_assert(sizeof(T) % 4 == 0, "tables are copied word by word");
  const uint32_t words = n * (uint32_t)(sizeof(T) / 4);
  uint32_t* d = reinterpret_cast<uint32_t*>(lds + off);
  const uint32_t* g = reinterpret_cast<const uint32_t*>(src);
  for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) d[i] = g[i];
  return reinterpret_cast<const T*>(lds + off);
}
template <bool FAST, bool LEAN, bool EXT, bool B0, bool PLAIN = false, bool TBL = false>
__global__ void __launch_bounds__(CHIP_BLOCK, (LEAN && FAST) ? CHIP_SHADE_LEAN_WAVES : (LEAN ? 1 : (PLAIN ? CHIP_SHADE_PLAIN_WAVES : CHIP_SHADE_FULL_WAVES))) k_shade_tiles(DScene S_arg, PathPool P, BatchDesc B, uint32_t bounce_arg, uint32_t pass, ShadeTables tbl) {
  const uint32_t bounce = B0 ? 0u : bounce_arg;
  __shared__ __attribute__((aligned(128))) unsigned char s_tbl[TBL ? CHIP_SHADE_TBL_BYTES : 16];
  DScene S_lds = S_arg;
  if (TBL) {
    uint32_t off[13];
    shade_tables_layout(tbl, off);
    S_lds.spheres = shade_table_copy(s_tbl, off[0], S_arg.spheres, tbl.n_spheres);
    S_lds.translates = shade_table_copy(s_tbl, off[1], S_arg.translates, tbl.n_translates);
    S_lds.quads = shade_table_copy(s_tbl, off[2], S_arg.quads, tbl.n_quads);
    S_lds.rotates = shade_table_copy(s_tbl, off[3], S_arg.rotates, tbl.n_rotates);
    S_lds.media = shade_table_copy(s_tbl, off[4], S_arg.media, tbl.n_media);
    S_lds.materials = shade_table_copy(s_tbl, off[5], S_arg.materials, tbl.n_materials);
    S_lds.textures = shade_table_copy(s_tbl, off[6], S_arg.textures, tbl.n_textures);
    S_lds.lights = shade_table_copy(s_tbl, off[7], S_arg.lights, tbl.n_lights);
    S_lds.ltree = shade_table_copy(s_tbl, off[8], S_arg.ltree, tbl.n_ltree);
    S_lds.xchain = shade_table_copy(s_tbl, off[9], S_arg.xchain, tbl.n_xchain);
    if (EXT) {
      S_lds.discs = shade_table_copy(s_tbl, off[10], S_arg.discs, tbl.n_discs);
      S_lds.cones = shade_table_copy(s_tbl, off[11], S_arg.cones, tbl.n_cones);
      S_lds.scales = shade_table_copy(s_tbl, off[12], S_arg.scales, tbl.n_scales);
    }
    __syncthreads();
  }
  const DScene& S = TBL ? S_lds : S_arg;
#if CHIP_SHADE_PRIO
  __builtin_amdgcn_s_setprio(CHIP_SHADE_PRIO);
#endif
  const PathRay* ray_in = P.ray[bounce & 1];
  PathRay* ray_out = P.ray[(bounce + 1) & 1];
  PathMap* st_out = P.st[(bounce + 1) & 1];
  uint32_t* ctr = P.counters + (size_t)bounce * CHIP_CTR_STRIDE;
  uint32_t* ctr_next = P.counters + (size_t)(bounce + 1) * CHIP_CTR_STRIDE;
  // entry i of this launch is q_lit[i]: the rays the fused first bounce left to the literal pass, or the hits the
  // plain-texture pass left to this one (it lists them in the queue, which the traversal is done with by then, so that
  // this pass goes over those few entries instead of the whole bounce)
  const bool rich_queue = !LEAN && pass == SHADE_RICH_ONLY;
  const bool via_queue = (B0 && pass == SHADE_LIT_QUEUE) || rich_queue;
  const uint32_t n_live = rich_queue ? ctr[CHIP_CTR_RICH] : (via_queue ? ctr[CHIP_CTR_LIT] : ctr[0]);
  const bool last_bounce = bounce + 1 >= S.cam.max_depth;
  constexpr uint32_t NB = MK_COUNT + 3;        // material kinds, misses, empty entries, entries left to the other pass
  __shared__ uint16_t s_order[CHIP_SHADE_TILE];  // tile-local entry numbers, sorted by bin
  __shared__ uint16_t s_rank[CHIP_SHADE_TILE];
  __shared__ uint8_t s_kind[CHIP_SHADE_TILE];
  __shared__ uint32_t s_cnt[NB], s_off[NB + 1], s_live, s_base, s_reserved, s_qbase;
  __shared__ uint4 s_stage[CHIP_BLOCK / 64][CHIP_XPOSE_U4];
  uint4* stage = s_stage[threadIdx.x >> 6];
  // entries per tile: up to CHIP_SHADE_TILE, but small enough that a short queue still spreads over the grid
  uint32_t tile = (n_live / gridDim.x) & ~(uint32_t)(CHIP_BLOCK - 1);
  tile = tile < CHIP_BLOCK ? CHIP_BLOCK : (tile > CHIP_SHADE_TILE ? CHIP_SHADE_TILE : tile);
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t lane_entry = 16u * (lane & 3u) + (lane >> 2);
  for (uint32_t tile0 = blockIdx.x * tile; tile0 < n_live; tile0 += gridDim.x * tile) {
    if (threadIdx.x < NB) s_cnt[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_live = 0;
    __syncthreads();
    const uint32_t tile_n = min(tile, n_live - tile0);
    for (uint32_t k = threadIdx.x; k < tile_n; k += CHIP_BLOCK) {
      const uint32_t e = via_queue ? P.q_lit[tile0 + k] : tile0 + k;
      const uint2 hw = *reinterpret_cast<const uint2*>(&P.hit[e].node);
      uint32_t kind = hw.x == NREF_NONE ? MK_COUNT : (hw.x == NREF_DEAD ? MK_COUNT + 1 : MATK_KIND(hw.y));
      if (!LEAN && pass == SHADE_PLAIN_ONLY) {
        // two launches share the bounce: hits on materials with an image / noise texture go to the second one (the
        // full kernel), everything else, misses included, to this one (the kernel without that code)
        const bool rich = kind < MK_COUNT && (S.materials[MATK_ID(hw.y)].flags & MF_RICH_TEX) != 0;
        if (rich) kind = MK_COUNT + 2;
      }
      s_kind[k] = (uint8_t)kind;
      s_rank[k] = (uint16_t)atomicAdd(&s_cnt[kind], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t acc = 0;
      for (uint32_t b = 0; b < NB; b++) { s_off[b] = acc; acc += s_cnt[b]; }
      s_off[NB] = acc;
      // room for every entry whose material can scatter (camera.rs:278)
      const uint32_t reserve = last_bounce ? 0u : s_cnt[MK_LAMBERTIAN] + s_cnt[MK_METAL] + s_cnt[MK_DIELECTRIC] + s_cnt[MK_ISOTROPIC];
      s_reserved = reserve;
      s_base = reserve ? atomicAdd(&ctr_next[0], reserve) : 0u;
      s_qbase = (!LEAN && pass == SHADE_PLAIN_ONLY && s_cnt[NB - 1]) ? atomicAdd(&ctr[CHIP_CTR_RICH], s_cnt[NB - 1]) : 0u;
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < tile_n; k += CHIP_BLOCK) s_order[s_off[s_kind[k]] + s_rank[k]] = (uint16_t)k;
    __syncthreads();
    if (!LEAN && pass == SHADE_PLAIN_ONLY)  // the entries left to the other pass, listed for it
      for (uint32_t i = threadIdx.x; i < s_cnt[NB - 1]; i += CHIP_BLOCK) P.q_lit[s_qbase + i] = tile0 + s_order[s_off[NB - 1] + i];
    // One loop per bin, so that each material's code is register-allocated on its own.  A wave takes 64
    // consecutive entries of the bin per trip; lane 4q + m shades entry 16m + q of them, which makes load j
    // of the quad-cooperative accesses cover entries 16j .. 16j + 15.  The trip count is wave-uniform.
#define CHIP_SHADE_BIN(KIND)                                                                                     \
  {                                                                                                              \
    const uint32_t bin_end = __builtin_amdgcn_readfirstlane(s_off[KIND + 1]);                                    \
    _Pragma("clang loop unroll(disable)")                                                                        \
    for (uint32_t k0 = __builtin_amdgcn_readfirstlane(s_off[KIND] + (threadIdx.x & ~63u)); k0 < bin_end; k0 += CHIP_BLOCK) { \
      const uint32_t k = k0 + lane_entry;                                                                        \
      const bool valid = k < bin_end;                                                                            \
      uint32_t entry = valid ? tile0 + s_order[k] : CHIP_NO_SLOT;                                                \
      if (via_queue && valid) entry = P.q_lit[entry];                                                            \
      PathHit ph;                                                                                                \
      ph.t = 0.0; ph.node = NREF_NONE; ph.info = 0;                                                              \
      if (valid) *reinterpret_cast<uint4*>(&ph) = *reinterpret_cast<const uint4*>(P.hit + entry);                \
      V3 ro = mk(0, 0, 0), rd = mk(0, 0, 0), no = mk(0, 0, 0), nd = mk(0, 0, 0);                                  \
      double rtime = 0.0;                                                                                        \
      uint32_t id = entry;                                                                                       \
      if (B0) {                                                                                                  \
        uint32_t tries = PATHHIT_TRIES(ph.info);                                                                 \
        ph.info &= 0xFFFFFFu;                                                                                    \
        if (valid) path_camera_ray(S, B, entry, ro, rd, rtime, tries);                                           \
      } else {                                                                                                   \
        QuadRecord q;                                                                                            \
        if constexpr (FAST) {                                                                                    \
          quad_load<CHIP_RAY48_BYTES, 3>(ray_in, entry, lane, stage, q);                                         \
          ray_unpack48(q, ro, rd, rtime, id);                                                                    \
        } else {                                                                                                 \
          quad_load<(int)sizeof(PathRay), 4>(ray_in, entry, lane, stage, q);                                     \
          ray_unpack(q, ro, rd, rtime, id);                                                                      \
        }                                                                                                        \
      }                                                                                                          \
      float xw[3] = {0.f, 0.f, 0.f}, klo = 0.f, khi = 0.f;                                                       \
      bool cont = false;                                                                                         \
      /* The lean forms have registers to spare: their radiance-map loads are issued before the material code and  \
         travel while it chases the hit's sphere and material records (one memory round trip less per trip).  The \
         other forms load the map after the material code, which is where their registers are scarce. */         \
      constexpr bool EARLY_MAP = LEAN && FAST && !B0;                                                            \
      uint4 mapv[2];                                                                                             \
      if (EARLY_MAP) quad_load32_issue(P.st[bounce & 1], entry, lane, mapv);                                     \
      if (valid) cont = shade_hit<KIND, FAST, LEAN, EXT, PLAIN>(S, B, bounce, id, ph, ro, rd, rtime, xw, klo, khi, no, nd); \
      Affine f;                                                                                                  \
      if (EARLY_MAP) {                                                                                           \
        QuadRecord qm;                                                                                           \
        quad_load32_finish(mapv, lane, stage, qm);                                                                \
        affine_unpack(qm, f, id);                                                                                \
      } else {                                                                                                   \
        quad_load_map<B0>(P, entry, lane, bounce, stage, f, id);                                                 \
      }                                                                                                          \
      if (valid) {                                                                                               \
        if (cont) {                                                                                              \
          affine_compose(f, xw, klo, khi);                                                                       \
          xw[0] = 0.f; xw[1] = 0.f; xw[2] = 0.f;                                                                 \
          cont = !last_bounce;  /* camera.rs:259-261 */                                                          \
        }                                                                                                        \
        if (!cont) finish_path(P, id, f, xw);                                                                    \
      }                                                                                                          \
      const uint32_t out_rank = quad_order_rank(cont, lane, &s_live);  /* every lane takes part */               \
      const uint32_t out = cont ? s_base + out_rank : CHIP_NO_SLOT;                                              \
      {                                                                                                          \
        QuadRecord q;                                                                                            \
        if constexpr (FAST) {                                                                                    \
          ray_pack48(q, no, nd, rtime, id);                                                                      \
          quad_store<CHIP_RAY48_BYTES, 3>(ray_out, out, lane, stage, q);                                         \
        } else {                                                                                                 \
          ray_pack(q, no, nd, rtime, id);                                                                        \
          quad_store<(int)sizeof(PathRay), 4>(ray_out, out, lane, stage, q);                                     \
        }                                                                                                        \
        affine_pack(q, f, id);                                                                                   \
        quad_store32(st_out, out, lane, stage, q);                                        \
      }                                                                                                          \
    }                                                                                                            \
  }
    CHIP_SHADE_BIN(MK_LAMBERTIAN)
    CHIP_SHADE_BIN(MK_METAL)
    CHIP_SHADE_BIN(MK_DIELECTRIC)
    CHIP_SHADE_BIN(MK_DIFFUSE_LIGHT)
    CHIP_SHADE_BIN(MK_ISOTROPIC)
    CHIP_SHADE_BIN(MK_EMPTY)
#undef CHIP_SHADE_BIN
    {  // misses, camera.rs:264-266
      const uint32_t bin_end = __builtin_amdgcn_readfirstlane(s_off[MK_COUNT + 1]);
      for (uint32_t k0 = __builtin_amdgcn_readfirstlane(s_off[MK_COUNT] + (threadIdx.x & ~63u)); k0 < bin_end; k0 += CHIP_BLOCK) {
        const uint32_t k = k0 + lane_entry;
        const bool valid = k < bin_end;
        uint32_t entry = valid ? tile0 + s_order[k] : CHIP_NO_SLOT;
        if (via_queue && valid) entry = P.q_lit[entry];
        uint32_t id = entry;
        Affine f;
        quad_load_map<B0>(P, entry, lane, bounce, stage, f, id);  // a miss needs nothing of its ray record
        float bg[3] = {(float)S.cam.background[0], (float)S.cam.background[1], (float)S.cam.background[2]};
        if (valid) finish_path(P, id, f, bg);
      }
    }
    __syncthreads();
    // reserved entries that no path took
    for (uint32_t k = s_live + threadIdx.x; k < s_reserved; k += CHIP_BLOCK)
      ray_mark_dead(ray_out, s_base + k, FAST);
    __syncthreads();
  }
}

// ------------------------------------------------------------------ bounce 0 in one kernel (small sphere scenes)
// Camera rays of one wave come from an 8 x 8 pixel tile: they finish their traversal at about the same time and mostly
// hit the same few spheres, so the wave-at-a-time form loses little to divergence, and shading them where they were
// traced saves what the two-kernel form pays at bounce 0: the hit record's trip through memory, the second evaluation of
// the camera ray (three to six Philox blocks) and the binning pass.  k_first_small is k_extend_small<true> followed, per
// wave, by the lean shade code (spheres with solid colours, no lights) on the 64 hits; the paths that go on write their
// ray and radiance map as entries of bounce 1.  Rays the compact traversal does not take (a zero direction component,
// a descent deeper than its stack) go to the literal queue as before; the literal pass traces them and one launch of the
// bounce-0 shade kernel over that queue (SHADE_LIT_QUEUE) shades them with the same code.
//
// Entries of bounce 1 are handed out to the waves in ranges of `out_range` entries taken from the bounce's entry counter
// (one atomic per range instead of one per wave and trip); a wave's outputs fill its range from the front, the next
// range is taken when one is used up (an output run may straddle two ranges), and what is left of a wave's last range
// when the kernel ends is marked empty, like the entries the shade kernel reserves for paths that then end.  With
// out_range = 0 a wave takes exactly what it needs per trip (small launches, where the counter is not contended and
// the pool has no room to spare).
template <bool FAST>
__global__ void __launch_bounds__(extend_block(0, true), 4) k_first_small(DScene S, PathPool P, BatchDesc B, CacheDesc cd, uint32_t out_range) {
  const NodeCache C{0u, (uint32_t)CHIP_WNODE_U4 * cd.n_bvh, cd.n_bvh, cd.n_sph, cd.n_bvh, cd.n_sph};
  uint32_t* ctr = P.counters;
  uint32_t* ctr_next = P.counters + CHIP_CTR_STRIDE;
  const uint32_t n_live = ctr[0];
  if ((size_t)blockIdx.x * 64u >= n_live) return;
  for (uint32_t i = threadIdx.x; i < cd.n_bvh; i += blockDim.x) {
    const uint4* p = reinterpret_cast<const uint4*>(S.wnodes + i);
#pragma unroll
    for (int k = 0; k < 6; k++) g_lds[(uint32_t)CHIP_WNODE_U4 * i + k] = p[k];
    const uint4 ch = p[6];
    g_lds[(uint32_t)CHIP_WNODE_U4 * i + 6u] = make_uint4(small_ref(ch.x), small_ref(ch.y), small_ref(ch.z), small_ref(ch.w));
  }
  const uint32_t so = (uint32_t)CHIP_WNODE_U4 * cd.n_bvh;
  for (uint32_t i = threadIdx.x; i < cd.n_sph; i += blockDim.x) {
    const uint4* p = reinterpret_cast<const uint4*>(S.spheres + i);
    g_lds[so + i] = p[0]; g_lds[so + cd.n_sph + i] = p[1]; g_lds[so + 2 * cd.n_sph + i] = p[2]; g_lds[so + 3 * cd.n_sph + i] = p[3];
  }
  __syncthreads();
  uint32_t* cursor = &ctr[CHIP_CTR_CURSOR];
  const uint32_t lane = __lane_id();
  const unsigned long long lanes_below = (1ull << lane) - 1ull;
  SmallStack st;
  st.lds = (lds_u1_t)(g_lds + (CHIP_WNODE_U4 * cd.n_bvh + 4u * cd.n_sph)) + threadIdx.x;
  st.overflow = 0;
  st.sp = 0;
  st.lds_cap = (int)cd.stack_entries;
  st.stride4 = 4u * blockDim.x;
  TravCounters cnt{0, 0};
  uint32_t nseg = 0;
  const uint32_t root = small_ref(S.ord_root);
  const uint32_t n_waves = gridDim.x * (blockDim.x / 64);
  uint32_t chunk = (n_live / (n_waves * 4u)) & ~63u;
  chunk = chunk < 64u ? 64u : (chunk > 2048u ? 2048u : chunk);
  uint32_t w_next = 0, w_end = 0;                // wave-uniform: the chunk of bounce-0 entries (= paths) in hand
  uint32_t o_next = 0, o_end = 0;                // wave-uniform: the range of bounce-1 entries in hand
  const bool last_bounce = 1u >= S.cam.max_depth;  // camera.rs:259-261
  PathRay* ray_out = P.ray[1];
  PathMap* st_out = P.st[1];
  for (;;) {
    if (w_next >= w_end) {
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(cursor, chunk);
      base = __builtin_amdgcn_readfirstlane(base);
      if (base >= n_live) break;
      w_next = base;
      w_end = min(base + chunk, n_live);
    }
    // 1. the next 64 paths of the chunk, one per lane: camera ray (camera.rs:359-374)
    const uint32_t entry = w_next + lane;
    const bool mine = entry < w_end;
    w_next = min(w_next + 64u, w_end);
    TravRay r;
    r.wo = r.wd = r.o = r.d = r.inv = mk(0, 0, 0);
    r.time = 0;
    r.best_mat = 0;
    uint32_t pixel = 0, stratum = 0;
    bool has = false;
    double best_t = INFINITY;
    nref_t best = NREF_NONE;
    uint32_t node = CHIP_SM_NONE;
    bool to_literal = false;
    if (mine) {
      uint32_t x, y;
      slot_to_path(B, S.cam, entry, x, y, pixel, stratum);
      Rng rng;
      rng.init(B.seed_lo, B.seed_hi, pixel, stratum, CHIP_CAMERA_BOUNCE, 0);
      const uint32_t s_y = magic_div(stratum, B.sqrt_spp_magic), s_x = stratum - s_y * S.cam.sqrt_spp;
      uint32_t tries = 0;
      camera_ray(S.cam, rng, x, y, s_x, s_y, r.wo, r.wd, r.time, tries);
      r.o = r.wo; r.d = r.wd;
      const bool ordinary = ray_set_inv<true>(r);
#ifdef CHIP_TEST_HAND_OVER
      to_literal = B.reference_order || !ordinary || entry % CHIP_TEST_HAND_OVER == 0u;
#else
      to_literal = B.reference_order || !ordinary;
#endif
      if (!to_literal) {
        r.capf = INFINITY;
        node = root;
        st.sp = 0;
        has = true;
      }
    }
    // 2. closest hit; the whole wave stays until its last ray is done
    if (has) traverse_small<true>(S, C, node, r, 0.001, best_t, best, st, cnt, 0);
    if (has && st.overflow) {  // the descent ran out of stack: the literal pass traces this ray
      st.overflow = 0;
      has = false;
      to_literal = true;
    }
    {
      const unsigned long long ml = __ballot(to_literal);
      if (ml) {
        uint32_t lbase = 0;
        if (lane == 0) lbase = atomicAdd(&ctr[CHIP_CTR_LIT], (uint32_t)__popcll(ml));
        lbase = __builtin_amdgcn_readfirstlane(lbase);
        if (to_literal) P.q_lit[lbase + (uint32_t)__popcll(ml & lanes_below)] = entry;
      }
    }
    nseg += has ? 1u : 0u;
    // 3. shading (material/*.rs through shade_material; a miss takes the background, camera.rs:264-266)
    const uint32_t kind = !has ? 0xFFu : (best == NREF_NONE ? (uint32_t)MK_COUNT : MATK_KIND(r.best_mat));
    float xw[3] = {0.f, 0.f, 0.f}, klo = 0.f, khi = 0.f;
    bool cont = false;
    V3 no = mk(0, 0, 0), nd = mk(0, 0, 0);
    HitOut hrec;
    hrec.pt = hrec.n = mk(0, 0, 0);
    hrec.u = hrec.v = 0.0;
    hrec.front = false;
    hrec.mat = 0;
    hrec.leaf_id = -1;
    if (has && best != NREF_NONE) {
      // HitRecord of the closest sphere (sphere.rs:122-134, object/mod.rs:44-51) from its LDS copy; the same arithmetic as
      // reconstruct<0, FAST>, which the shade kernel runs for the rays the literal pass traced
      const uint32_t i = C.sph_off + NREF_INDEX(best);
      const uint4 a = lds_read4(i), b = lds_read4(i + C.cap_sph), c = lds_read4(i + 2 * C.cap_sph), d = lds_read4(i + 3 * C.cap_sph);
      const V3 c0 = mk(u2d(a.x, a.y), u2d(a.z, a.w), u2d(b.x, b.y)), dc = mk(u2d(b.z, b.w), u2d(c.x, c.y), u2d(c.z, c.w));
      const double radius = u2d(d.x, d.y);
      hrec.pt = r.wo + (best_t * r.wd);
      const V3 center = c0 + (r.time * dc);
      const V3 outward = FAST ? (hrec.pt - center) * fast_rcp(radius) : (hrec.pt - center) / radius;
      hrec.front = dot(r.wd, outward) < 0.0;
      hrec.n = hrec.front ? outward : -outward;
      hrec.mat = MATK_ID(r.best_mat);
    }
#define CHIP_FIRST_KIND(KIND)                                                                                          \
  if (__any(kind == KIND)) {                                                                                           \
    if (kind == KIND)                                                                                                  \
      cont = shade_material<KIND, FAST, true, false, false>(S, B, 0u, pixel, stratum, S.materials[hrec.mat], hrec, r.wd, xw, klo, khi, no, nd); \
  }
    CHIP_FIRST_KIND(MK_LAMBERTIAN)
    CHIP_FIRST_KIND(MK_METAL)
    CHIP_FIRST_KIND(MK_DIELECTRIC)
    CHIP_FIRST_KIND(MK_DIFFUSE_LIGHT)
    CHIP_FIRST_KIND(MK_ISOTROPIC)
#undef CHIP_FIRST_KIND
    if (kind == MK_COUNT) {
      xw[0] = (float)S.cam.background[0]; xw[1] = (float)S.cam.background[1]; xw[2] = (float)S.cam.background[2];
    }
    Affine f;
    affine_identity(f);
    if (has) {
      if (cont) {
        affine_compose(f, xw, klo, khi);
        xw[0] = 0.f; xw[1] = 0.f; xw[2] = 0.f;
        cont = !last_bounce;
      }
      if (!cont) finish_path(P, entry, f, xw);  // (at bounce 0 entry i is path i)
    }
    // 4. the paths that go on become entries of bounce 1
    const unsigned long long mc = __ballot(cont);
    if (mc) {
      const uint32_t k = (uint32_t)__popcll(mc);
      uint32_t base1 = o_next, n1 = k, base2 = 0;
      if (out_range == 0u) {
        uint32_t g = 0;
        if (lane == 0) g = atomicAdd(&ctr_next[0], k);
        base1 = __builtin_amdgcn_readfirstlane(g);
      } else if (o_end - o_next >= k) {
        o_next += k;
      } else {
        n1 = o_end - o_next;
        uint32_t g = 0;
        if (lane == 0) g = atomicAdd(&ctr_next[0], out_range);
        base2 = __builtin_amdgcn_readfirstlane(g);
        o_next = base2 + (k - n1);
        o_end = base2 + out_range;
      }
      if (cont) {
        const uint32_t rank = (uint32_t)__popcll(mc & lanes_below);
        const uint32_t out = rank < n1 ? base1 + rank : base2 + (rank - n1);
        if constexpr (FAST) {
          uint4* pr = reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(ray_out) + (size_t)out * CHIP_RAY48_BYTES);
          uint4 q0, q1, q2;
          d2u(no.x, q0.x, q0.y); d2u(no.y, q0.z, q0.w);
          d2u(no.z, q1.x, q1.y); q1.z = __float_as_uint((float)nd.x); q1.w = __float_as_uint((float)nd.y);
          q2.x = __float_as_uint((float)nd.z); q2.y = entry; d2u(r.time, q2.z, q2.w);
          pr[0] = q0; pr[1] = q1; pr[2] = q2;
        } else {
          uint4* pr = reinterpret_cast<uint4*>(ray_out + out);
          uint4 q0, q1, q2, q3;
          d2u(no.x, q0.x, q0.y); d2u(no.y, q0.z, q0.w);
          d2u(no.z, q1.x, q1.y); d2u(nd.x, q1.z, q1.w);
          d2u(nd.y, q2.x, q2.y); d2u(nd.z, q2.z, q2.w);
          d2u(r.time, q3.x, q3.y); q3.z = entry; q3.w = 0u;
          pr[0] = q0; pr[1] = q1; pr[2] = q2; pr[3] = q3;
        }
        uint4* pm = reinterpret_cast<uint4*>(st_out + out);
        pm[0] = make_uint4(entry | f.clamped, __float_as_uint(f.b[0]), __float_as_uint(f.b[1]), __float_as_uint(f.b[2]));
        pm[1] = make_uint4(__float_as_uint(f.bound[0]), __float_as_uint(f.bound[1]), __float_as_uint(f.bound[2]), 0u);
      }
    }
  }
  // what is left of the wave's last range holds no path
  for (uint32_t i = o_next + lane; i < o_end; i += 64u) ray_mark_dead(ray_out, i, FAST);
  for (int off = 32; off > 0; off >>= 1) {
    cnt.bvh += __shfl_down(cnt.bvh, off);
    cnt.prim += __shfl_down(cnt.prim, off);
    nseg += __shfl_down(nseg, off);
  }
  if (lane == 0 && nseg) {
    atomicAdd(&P.stats[0], (unsigned long long)nseg);
    atomicAdd(&P.stats[1], (unsigned long long)cnt.bvh);
    atomicAdd(&P.stats[2], (unsigned long long)cnt.prim);
  }
}

// ------------------------------------------------------------------ the tail of a batch in one kernel (small sphere scenes)
// From a dozen bounces on a batch holds too few paths to fill the device, and every further bounce still costs a round
// of dependent launches: the last forty bounces of a batch take about 8 ms however few paths are left.  k_tail_small
// takes the entries of one bounce, once they are few enough, and follows every one of those paths to its end: a lane
// traces its ray (traverse_small, as k_extend_small does), shades the hit where it stands (the lean material code, as
// k_first_small does), keeps the scattered ray and the composed radiance map in registers, and goes on with the next
// bounce; a lane whose path has ended takes the next entry of the queue.  A path's random stream is keyed by (pixel,
// stratum, bounce), and in the default shading mode the scattered direction is rounded to the floats a ray record would
// have held, so every sample is the bit pattern the bounce-by-bounce kernels produce.
// A ray the compact traversal does not take (a zero direction component, a descent deeper than its stack) is traced
// right here by the literal form (tail_literal: the reference's own tree and order, its stack in the thread's spill slab).
// The host enqueues the kernel ahead of a bounce's ordinary launches without knowing the live count: a launch that finds
// more than `max_live` entries returns at once, and one that takes the bounce sets its entry count to zero when its last
// block is done, so that the launches that follow find nothing to do.
#define CHIP_CTR_TAIL_DONE 11
// (takes the two tables and the root rather than the scene: an out-of-line call with the whole DScene would put it, and
// with it the caller's every access to it, into scratch memory)
__device__ __noinline__ void tail_literal(const DBvhNode* bvh, const DSphere* spheres, nref_t scene_root, TravRay& r, double& best_t, nref_t& best,
                                          uint32_t* slab, TravCounters& cnt, uint32_t* errflag) {
  DScene S{};
  S.bvh = bvh;
  S.spheres = spheres;
  S.root = scene_root;
  TravStack ts;
  ts.lds = (lds_u2_t)0;  // (never addressed: no entry is kept in LDS)
  ts.ovf = slab;
  ts.sp = 0;
  ts.overflow = 0;
  ts.redo = 0;
  ts.lds_cap = 0;
  ts.stride8 = 0;
  r.o = r.wo; r.d = r.wd;
  ray_set_inv<false>(r);
  best_t = INFINITY;
  best = NREF_NONE;
  nref_t node = S.root;
  Rng rng;
  rng.init(0, 0, 0, 0, 0, 0);
  const NodeCache none{0, 0, 0, 0, 0, 0};
  traverse<0, false, false, false>(S, none, node, r, 0.001, best_t, best, rng, ts, 0, cnt);
  if (ts.overflow) atomicOr(errflag, 1u);
}
template <bool FAST>
__global__ void __launch_bounds__(extend_block(0, true), 4) k_tail_small(DScene S, PathPool P, BatchDesc B, uint32_t bounce0, CacheDesc cd, uint32_t max_live) {
  uint32_t* ctr = P.counters + (size_t)bounce0 * CHIP_CTR_STRIDE;
  const uint32_t n_live = ctr[0];
  if (n_live == 0u || n_live > max_live) return;  // (every block of the launch reads the same count: see the end of the kernel)
  const NodeCache C{0u, (uint32_t)CHIP_WNODE_U4 * cd.n_bvh, cd.n_bvh, cd.n_sph, cd.n_bvh, cd.n_sph};
  for (uint32_t i = threadIdx.x; i < cd.n_bvh; i += blockDim.x) {
    const uint4* p = reinterpret_cast<const uint4*>(S.wnodes + i);
#pragma unroll
    for (int k = 0; k < 6; k++) g_lds[(uint32_t)CHIP_WNODE_U4 * i + k] = p[k];
    const uint4 ch = p[6];
    g_lds[(uint32_t)CHIP_WNODE_U4 * i + 6u] = make_uint4(small_ref(ch.x), small_ref(ch.y), small_ref(ch.z), small_ref(ch.w));
  }
  const uint32_t so = (uint32_t)CHIP_WNODE_U4 * cd.n_bvh;
  for (uint32_t i = threadIdx.x; i < cd.n_sph; i += blockDim.x) {
    const uint4* p = reinterpret_cast<const uint4*>(S.spheres + i);
    g_lds[so + i] = p[0]; g_lds[so + cd.n_sph + i] = p[1]; g_lds[so + 2 * cd.n_sph + i] = p[2]; g_lds[so + 3 * cd.n_sph + i] = p[3];
  }
  __syncthreads();
  const PathRay* ray_in = P.ray[bounce0 & 1];
  const PathMap* map_in = P.st[bounce0 & 1];
  uint32_t* cursor = &ctr[CHIP_CTR_CURSOR];
  const uint32_t lane = __lane_id();
  const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x;
  SmallStack st;
  st.lds = (lds_u1_t)(g_lds + (CHIP_WNODE_U4 * cd.n_bvh + 4u * cd.n_sph)) + threadIdx.x;
  st.overflow = 0;
  st.sp = 0;
  st.lds_cap = (int)cd.stack_entries;
  st.stride4 = 4u * blockDim.x;
  TravCounters cnt{0, 0};
  uint32_t nseg = 0;
  const uint32_t root = small_ref(S.ord_root);
  const uint32_t max_depth = S.cam.max_depth;
  // per-lane path state
  bool has = false;
  uint32_t id = 0, bnc = bounce0;
  Affine f;
  affine_identity(f);
  TravRay r;
  r.wo = r.wd = r.o = r.d = r.inv = mk(0, 0, 0);
  r.time = 0;
  r.best_mat = 0;
  double best_t = INFINITY;
  nref_t best = NREF_NONE;
  uint32_t node = CHIP_SM_NONE;
  uint32_t w_next = 0, w_end = 0;
  bool exhausted = false;
  // Takes the ray in r.wo / r.wd to the start of its traversal (or flags it for the literal form)
#define CHIP_TAIL_START(hand_over)                                        \
  {                                                                       \
    r.o = r.wo; r.d = r.wd;                                               \
    const bool ordinary_ = ray_set_inv<true>(r);                          \
    r.capf = INFINITY;                                                    \
    best_t = INFINITY;                                                    \
    best = NREF_NONE;                                                     \
    st.sp = 0;                                                            \
    st.overflow = (!ordinary_ || (hand_over)) ? 1u : 0u;                  \
    node = st.overflow ? CHIP_SM_NONE : root;                             \
    has = true;                                                           \
  }
  for (;;) {
    // 1. rays whose traversal is over: those the compact form gave up on are traced literally first
    const bool fin = has && node == CHIP_SM_NONE;
    {
      const bool lit = fin && st.overflow != 0u;
      if (__any(lit)) {
        if (lit) tail_literal(S.bvh, S.spheres, S.root, r, best_t, best, P.ovf + (size_t)gtid * (2 * CHIP_STACK_OVF), cnt, &P.errflag[0]);
      }
      st.overflow = fin ? 0u : st.overflow;
    }
    // 2. their hits are shaded where they stand (material/*.rs through shade_material; a miss takes the background)
    if (__any(fin)) {
      nseg += fin ? 1u : 0u;
      const uint32_t kind = !fin ? 0xFFu : (best == NREF_NONE ? (uint32_t)MK_COUNT : MATK_KIND(r.best_mat));
      float xw[3] = {0.f, 0.f, 0.f}, klo = 0.f, khi = 0.f;
      bool cont = false;
      V3 no = mk(0, 0, 0), nd = mk(0, 0, 0);
      HitOut hrec;
      hrec.pt = hrec.n = mk(0, 0, 0);
      hrec.u = hrec.v = 0.0;
      hrec.front = false;
      hrec.mat = 0;
      hrec.leaf_id = -1;
      uint32_t pixel = 0, stratum = 0;
      if (fin && best != NREF_NONE) {
        // HitRecord of the closest sphere (sphere.rs:122-134, object/mod.rs:44-51) from its LDS copy: reconstruct<0, FAST>'s arithmetic
        const uint32_t i = C.sph_off + NREF_INDEX(best);
        const uint4 a = lds_read4(i), b = lds_read4(i + C.cap_sph), c = lds_read4(i + 2 * C.cap_sph), d = lds_read4(i + 3 * C.cap_sph);
        const V3 c0 = mk(u2d(a.x, a.y), u2d(a.z, a.w), u2d(b.x, b.y)), dc = mk(u2d(b.z, b.w), u2d(c.x, c.y), u2d(c.z, c.w));
        const double radius = u2d(d.x, d.y);
        hrec.pt = r.wo + (best_t * r.wd);
        const V3 center = c0 + (r.time * dc);
        const V3 outward = FAST ? (hrec.pt - center) * fast_rcp(radius) : (hrec.pt - center) / radius;
        hrec.front = dot(r.wd, outward) < 0.0;
        hrec.n = hrec.front ? outward : -outward;
        hrec.mat = MATK_ID(r.best_mat);
        uint32_t x_, y_;
        slot_to_path(B, S.cam, id, x_, y_, pixel, stratum);
      }
#define CHIP_TAIL_KIND(KIND)                                                                                           \
  if (__any(kind == KIND)) {                                                                                           \
    if (kind == KIND)                                                                                                  \
      cont = shade_material<KIND, FAST, true, false, false>(S, B, bnc, pixel, stratum, S.materials[hrec.mat], hrec, r.wd, xw, klo, khi, no, nd); \
  }
      CHIP_TAIL_KIND(MK_LAMBERTIAN)
      CHIP_TAIL_KIND(MK_METAL)
      CHIP_TAIL_KIND(MK_DIELECTRIC)
      CHIP_TAIL_KIND(MK_DIFFUSE_LIGHT)
      CHIP_TAIL_KIND(MK_ISOTROPIC)
#undef CHIP_TAIL_KIND
      if (kind == MK_COUNT) {
        xw[0] = (float)S.cam.background[0]; xw[1] = (float)S.cam.background[1]; xw[2] = (float)S.cam.background[2];
      }
      if (fin) {
        if (cont) {
          affine_compose(f, xw, klo, khi);
          xw[0] = 0.f; xw[1] = 0.f; xw[2] = 0.f;
          cont = !(bnc + 1u >= max_depth);  // camera.rs:259-261
        }
        if (!cont) {
          finish_path(P, id, f, xw);
          has = false;
        } else {
          // the next segment of the same path: the ray a record of the next bounce would have held
          r.wo = no;
          if (FAST) r.wd = mk((double)(float)nd.x, (double)(float)nd.y, (double)(float)nd.z);
          else r.wd = nd;
          bnc++;
          CHIP_TAIL_START(false)
        }
      }
    }
    // 3. lanes without a path take the next entries of the queue
    {
      const unsigned long long need = __ballot(!has);
      if (need && !exhausted) {
        if (w_next >= w_end) {
          uint32_t base = 0;
          if (lane == 0) base = atomicAdd(cursor, 64u);
          base = __builtin_amdgcn_readfirstlane(base);
          w_next = base;
          w_end = min(base + 64u, n_live);
          if (base >= n_live) { exhausted = true; w_end = w_next; }
        }
        const uint32_t rank = __popcll(need & ((1ull << lane) - 1ull));
        const uint32_t avail = w_end - w_next;
        if (!has && rank < avail) {
          const uint32_t entry = w_next + rank;
          uint32_t rid = 0;
          ray_load(ray_in, entry, B.ray48 != 0u, r.wo, r.wd, r.time, rid);
          if (rid != CHIP_DEAD_ID) {  // (an entry reserved by the last shading pass but left empty is passed over)
            const uint4* pm = reinterpret_cast<const uint4*>(map_in + entry);
            const uint4 m0 = pm[0], m1 = pm[1];
            id = m0.x & ~CHIP_MAP_CLAMPED;
            f.clamped = m0.x & CHIP_MAP_CLAMPED;
            f.b[0] = __uint_as_float(m0.y); f.b[1] = __uint_as_float(m0.z); f.b[2] = __uint_as_float(m0.w);
            f.bound[0] = __uint_as_float(m1.x); f.bound[1] = __uint_as_float(m1.y); f.bound[2] = __uint_as_float(m1.z);
            bnc = bounce0;
#ifdef CHIP_TEST_HAND_OVER
            CHIP_TAIL_START(entry % CHIP_TEST_HAND_OVER == 0u)
#else
            CHIP_TAIL_START(false)
#endif
          }
        }
        w_next += min((uint32_t)__popcll(need), avail);
      }
    }
    if (!__any(has) && exhausted) break;
    // 4. closest hits, until too few lanes are still at it
    //    (a lane that finishes goes on with its path's next ray, so the wave comes back for it as it would for a refill;
    //    once the queue is used up, when a quarter of the paths the wave still holds are waiting)
    const int busy_thresh = exhausted ? (int)((uint32_t)__popcll(__ballot(has)) * 3u / 4u) : CHIP_REFILL_THRESH;
    if (has && node != CHIP_SM_NONE) traverse_small<true>(S, C, node, r, 0.001, best_t, best, st, cnt, busy_thresh);
  }
#undef CHIP_TAIL_START
  for (int off = 32; off > 0; off >>= 1) {
    cnt.bvh += __shfl_down(cnt.bvh, off);
    cnt.prim += __shfl_down(cnt.prim, off);
    nseg += __shfl_down(nseg, off);
  }
  if (lane == 0 && nseg) {
    atomicAdd(&P.stats[0], (unsigned long long)nseg);
    atomicAdd(&P.stats[1], (unsigned long long)cnt.bvh);
    atomicAdd(&P.stats[2], (unsigned long long)cnt.prim);
  }
  // the last block to finish takes the bounce's entries off the books: the launches that follow find none
  __syncthreads();
  if (threadIdx.x == 0) {
    if (atomicAdd(&ctr[CHIP_CTR_TAIL_DONE], 1u) == gridDim.x - 1u) ctr[0] = 0u;
  }
}

// Adds this batch's finished samples into the rectangle's f32 radiance sum, strata in
// ascending order per pixel (camera.rs:335-341's `pixel_color += sample`).
__global__ void __launch_bounds__(CHIP_BLOCK) k_accumulate(PathPool P, BatchDesc B, float* sum_rgb) {
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < B.npix; p += gridDim.x * blockDim.x) {
    // the running sums take one sample after the other, so the result does not depend on where the batches were cut
    uint32_t xl, yl;
    rect_pixel(B, p, xl, yl);
    const size_t q = (size_t)yl * B.rect_w + xl;  // the pixel's place in the band's row-major sums
    float r = sum_rgb[q * 3 + 0], g = sum_rgb[q * 3 + 1], b = sum_rgb[q * 3 + 2];
    for (uint32_t sb = 0; sb < B.n_strata; sb++) {
      const float4 v = P.sample[(size_t)sb * B.npix + p];
      r += v.x;
      g += v.y;
      b += v.z;
    }
    sum_rgb[q * 3 + 0] = r;
    sum_rgb[q * 3 + 1] = g;
    sum_rgb[q * 3 + 2] = b;
  }
}

// camera.rs:343-344 (scale, NaN -> 0 on the sum, gamma = sqrt, clamp to [0, 0.999]) and the
// 8-bit quantisation of crates/cli/src/main.rs:193-198 (u8_scale 255.999) or crates/wasm/src/lib.rs:269-276 (255.0).
__global__ void __launch_bounds__(CHIP_BLOCK) k_resolve(const float* sum_rgb, size_t n_values, double scale, float* out_gamma,
                                                        uint8_t* out_rgb8, double u8_scale) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_values; i += (size_t)gridDim.x * blockDim.x) {
    double v = (double)sum_rgb[i];
    if (v != v) v = 0.0;
    v = scale * v;
    double g = v > 0.0 ? sqrt(v) : 0.0;
    g = rclamp(g, 0.0, 0.999);
    if (out_gamma) out_gamma[i] = (float)g;
    if (out_rgb8) out_rgb8[i] = (uint8_t)(g * u8_scale);
  }
}

// Test hook: closest hit + HitRecord for caller-supplied rays.
struct DHitOut {
  double t, pt[3], normal[3], u, v;
  int32_t hit, leaf_id, front_face, material_kind;
};
template <uint32_t F>
__global__ void __launch_bounds__(CHIP_BLOCK) k_trace_rays(DScene S, PathPool P, uint32_t seed_lo, uint32_t seed_hi, uint32_t n,
                                                           const double* o, const double* d, const double* tm, DHitOut* out) {
  __shared__ uint2 lds_stack[(CHIP_STACK_LDS + 1) * CHIP_BLOCK];  // (+ the spare entry of st_push_sel)
  const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x;
  TravStack st;
  st.lds = (lds_u2_t)lds_stack + threadIdx.x;
  st.ovf = P.ovf + (size_t)gtid * (2 * CHIP_STACK_OVF);
  st.overflow = 0;
  st.redo = 0;
  st.lds_cap = CHIP_STACK_LDS;
  st.stride8 = 8u * CHIP_BLOCK;
  const NodeCache none{0, 0, 0, 0, 0, 0};
  TravCounters cnt{0, 0};
  for (uint32_t i = gtid; i < n; i += gridDim.x * blockDim.x) {
    TravRay r;
    r.wo = mk(o[3 * (size_t)i], o[3 * (size_t)i + 1], o[3 * (size_t)i + 2]);
    r.wd = mk(d[3 * (size_t)i], d[3 * (size_t)i + 1], d[3 * (size_t)i + 2]);
    r.time = tm ? tm[i] : 0.0;
    r.o = r.wo; r.d = r.wd;
    constexpr bool ORD = true;
    // a moving sphere's box covers times 0..1 (sphere.rs:36-47); outside of them it leaves its box, and only the
    // reference's own visiting order says what such a ray finds
    bool literal = !ORD || S.literal_only || has_zero_component(r.wd) || !(r.time >= 0.0 && r.time <= 1.0);
    if (literal) ray_set_inv<false>(r);
    else ray_set_inv<true>(r);
    r.capf = INFINITY;
    Rng rng;
    rng.init(seed_lo, seed_hi, i, 0, 0, 0);
    double best_t = INFINITY;
    nref_t best = NREF_NONE;
    st.sp = 0;
    st.redo = 0;
    nref_t node = literal ? S.root : S.ord_root;
    if constexpr (ORD) {
      if (!literal) traverse<F, false, false, true>(S, none, node, r, 0.001, best_t, best, rng, st, 0, cnt);
      if (st.redo) {  // handed back by the ordered traversal: start over in the reference's order
        st.redo = 0;
        literal = true;
        r.o = r.wo; r.d = r.wd;
        ray_set_inv<false>(r);
        best_t = INFINITY;
        best = NREF_NONE;
        st.sp = 0;
        node = S.root;
      }
      if (literal) traverse<F, false, false, false>(S, none, node, r, 0.001, best_t, best, rng, st, 0, cnt);
    } else {
      traverse<F, false, false, false>(S, none, node, r, 0.001, best_t, best, rng, st, 0, cnt);
    }
    DHitOut h;
    h.t = 0; h.u = 0; h.v = 0;
    for (int k = 0; k < 3; k++) { h.pt[k] = 0; h.normal[k] = 0; }
    h.hit = 0; h.leaf_id = -1; h.front_face = 0; h.material_kind = -1;
    if (best != NREF_NONE) {
      HitOut ho;
      reconstruct<F>(S, best, best_t, r.wo, r.wd, r.time, ho);
      h.hit = 1;
      h.t = best_t;
      h.pt[0] = ho.pt.x; h.pt[1] = ho.pt.y; h.pt[2] = ho.pt.z;
      h.normal[0] = ho.n.x; h.normal[1] = ho.n.y; h.normal[2] = ho.n.z;
      h.u = ho.u; h.v = ho.v;
      h.leaf_id = ho.leaf_id;
      h.front_face = ho.front ? 1 : 0;
      h.material_kind = (int32_t)S.materials[ho.mat].kind;
    }
    out[i] = h;
  }
  if (st.overflow) atomicOr(&P.errflag[0], 1u);
}

// Test hook: camera rays for (x, y, s_x, s_y) tuples.
__global__ void __launch_bounds__(CHIP_BLOCK) k_camera_rays(DScene S, uint32_t seed_lo, uint32_t seed_hi, uint32_t n, const uint32_t* xyss,
                                                            double* rays) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    uint32_t x = xyss[4 * (size_t)i], y = xyss[4 * (size_t)i + 1], sx = xyss[4 * (size_t)i + 2], sy = xyss[4 * (size_t)i + 3];
    Rng rng;
    rng.init(seed_lo, seed_hi, y * S.cam.image_width + x, sy * S.cam.sqrt_spp + sx, CHIP_CAMERA_BOUNCE, 0);
    V3 o, d;
    double time;
    uint32_t tries = 0;
    camera_ray(S.cam, rng, x, y, sx, sy, o, d, time, tries);
    double* r = rays + 7 * (size_t)i;
    r[0] = o.x; r[1] = o.y; r[2] = o.z; r[3] = d.x; r[4] = d.y; r[5] = d.z; r[6] = time;
  }
}
