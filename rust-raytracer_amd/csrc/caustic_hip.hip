// libcaustic_hip.so: the C ABI of include/caustic_hip.h over the host scene graph
// (scene_host.hpp) and the gfx950 kernels (kernels.hpp).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "kernels.hpp"
#include "scene_host.hpp"

using namespace chip;

static thread_local std::string g_err;
static int fail(const std::string& msg) {
  g_err = msg;
  return -1;
}
#define HIP_TRY(expr)                                                                           \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(e_));       \
  } while (0)

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  hipError_t upload(const std::vector<T>& v) {
    release();
    n = v.size();
    size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    hipError_t e = hipMalloc((void**)&p, bytes);
    if (e != hipSuccess) return e;
    if (n) e = hipMemcpy(p, v.data(), n * sizeof(T), hipMemcpyHostToDevice);
    return e;
  }
  hipError_t alloc(size_t count) {
    release();
    n = count;
    return hipMalloc((void**)&p, std::max<size_t>(count, 1) * sizeof(T));
  }
  // keeps the buffer when it is already large enough
  hipError_t reserve(size_t count) { return (p && n >= count) ? hipSuccess : alloc(count); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
};

// The calling thread's current device is put back when a library call returns.
struct DeviceGuard {
  int prev = -1;
  DeviceGuard() {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};
// Events made during one render call; destroyed on every way out of it (after the device has gone idle when the
// call fails midway, so that no queued work still refers to them).
struct EventBag {
  std::vector<hipEvent_t> evs;
  bool failed = false;
  hipError_t make(hipEvent_t* out, bool timing) {
    hipError_t e = timing ? hipEventCreate(out) : hipEventCreateWithFlags(out, hipEventDisableTiming);
    if (e == hipSuccess) evs.push_back(*out);
    else *out = nullptr;
    return e;
  }
  ~EventBag() {
    if (failed) (void)hipDeviceSynchronize();
    for (hipEvent_t e : evs) (void)hipEventDestroy(e);
  }
};

// ---- scratch of the entry point that allocates nothing (chip_render_accumulate_noalloc)
// One path pool of CHIP_NOALLOC_PATHS slots with its counters, spill slab, error flags and statistics, laid out
// in a file-scope device array.  Every GPU the library runs on holds its own copy of the array at its own
// address; host code never uses the array's name as a pointer, only the address hipGetSymbolAddress returned
// for the device at hand (DeviceState::noalloc_base).
#define CHIP_NOALLOC_PATHS 32768u
#define CHIP_NOALLOC_COUNTER_ROWS 256u
#define CHIP_NOALLOC_MAX_BLOCK 1024u  // largest traversal block (extend_block)
#define CHIP_NOALLOC_OVF_SLABS (CHIP_NOALLOC_PATHS + CHIP_NOALLOC_MAX_BLOCK)
static_assert(extend_block(0, true) <= (int)CHIP_NOALLOC_MAX_BLOCK && extend_block(TF_QUADS | TF_XFORM, true) <= (int)CHIP_NOALLOC_MAX_BLOCK &&
                  extend_block(TF_MEDIA, true) <= (int)CHIP_NOALLOC_MAX_BLOCK && CHIP_BLOCK <= (int)CHIP_NOALLOC_MAX_BLOCK,
              "the static spill slab must cover a grid rounded up to the largest traversal block");
struct NoAllocLayout {
  PathRay ray[2 * CHIP_NOALLOC_PATHS];
  PathMap st[2 * CHIP_NOALLOC_PATHS];
  PathHit hit[CHIP_NOALLOC_PATHS];
  float4 sample[CHIP_NOALLOC_PATHS];
  uint32_t qlit[CHIP_NOALLOC_PATHS];
  uint32_t counters[CHIP_NOALLOC_COUNTER_ROWS * CHIP_CTR_STRIDE];
  // one spill slab per thread of a traversal launch: the grid covers the live paths rounded up to whole blocks, so a
  // launch over a full pool has up to one block less one thread more than slots (launch_extend_b checks the bound)
  uint32_t ovf[(size_t)CHIP_NOALLOC_OVF_SLABS * 2 * CHIP_STACK_OVF];
  uint32_t err[4];
  unsigned long long stats[4];
};
static __device__ __align__(256) unsigned char scratch[sizeof(NoAllocLayout)];

// Path pools.  A render of several batches runs two batches at a time, each in its own pool on its own
// stream, so that one batch's traversal (VALU-bound) overlaps the other's shading (HBM-bound).
struct PoolBufs {
  bool ready = false;
  PathPool P{};
  DevBuf<PathRay> ray;
  DevBuf<PathHit> hit;
  DevBuf<PathMap> st;
  DevBuf<float4> sample;
  DevBuf<uint32_t> qlit, counters, ovf;
  uint32_t counters_rows = 0;
  size_t slack = 0;  // entries the per-bounce arrays hold beyond P.cap (see PathPool::slack)
  void release() {
    ray.release(); hit.release(); st.release(); sample.release();
    qlit.release(); counters.release(); ovf.release();
    ready = false;
    counters_rows = 0;
    P = PathPool{};
  }
};
// What a batch in flight needs besides its pool: made once per device at commit, so a render call creates none of it.
struct LaneRes {
  uint32_t* h_live = nullptr;  // pinned: live-count readbacks (ring of four)
  hipStream_t stream = nullptr;
  hipEvent_t ev_rb[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_join = nullptr;
};
static constexpr int kMaxLanes = 4;

// Everything a scene keeps on one GPU: its tables, path pools, streams, events, pinned words and frame buffers.
struct DeviceState {
  int device = -1;
  int n_cus = 256;
  bool uploaded = false;
  DevBuf<DBvhNode> d_bvh;
  DevBuf<DWNode> d_wnodes;
  DevBuf<DSphere> d_spheres;
  DevBuf<DQuad> d_quads;
  DevBuf<DGroup> d_groups;
  DevBuf<nref_t> d_gchild;
  DevBuf<DTranslate> d_translates;
  DevBuf<DRotate> d_rotates;
  DevBuf<DScale> d_scales;
  DevBuf<DDisc> d_discs;
  DevBuf<DCone> d_cones;
  DevBuf<DMedium> d_media;
  DevBuf<nref_t> d_xchain;
  DevBuf<DMaterial> d_materials;
  DevBuf<DTexture> d_textures;
  DevBuf<DPerlin> d_perlins;
  DevBuf<uint8_t> d_texels;
  DevBuf<DLight> d_lights;
  DevBuf<DLightNode> d_ltree;
  DevBuf<DFlatPrim> d_flat;
  DScene dscene;
  PoolBufs pools[kMaxLanes];
  LaneRes lanes[kMaxLanes];
  bool res_ready = false;
  DevBuf<uint32_t> p_err;
  DevBuf<unsigned long long> p_stats;
  uint32_t* h_pinned = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_fork = nullptr;
  hipStream_t multi_stream = nullptr;  // chip_render_frame_multi: the stream this replica's render forks from and its sums leave on
  DevBuf<float> frame_sum, frame_stage;  // chip_render_frame's radiance sums; a peer's sums on their way into them
  DevBuf<uint8_t> frame_rgb8;
  unsigned char* noalloc_base = nullptr;  // this device's copy of `scratch`
  PathPool noalloc_pool{};

  void release_tables() {
    d_bvh.release(); d_wnodes.release(); d_spheres.release(); d_quads.release(); d_groups.release(); d_gchild.release();
    d_translates.release(); d_rotates.release(); d_scales.release(); d_discs.release(); d_cones.release(); d_media.release(); d_xchain.release(); d_materials.release();
    d_textures.release(); d_perlins.release(); d_texels.release(); d_lights.release(); d_ltree.release(); d_flat.release();
    uploaded = false;
  }
  // (with this state's device current)
  void release_all() {
    release_tables();
    for (int k = 0; k < kMaxLanes; k++) {
      pools[k].release();
      LaneRes& L = lanes[k];
      if (L.h_live) (void)hipHostFree(L.h_live);
      if (L.stream) (void)hipStreamDestroy(L.stream);
      for (int j = 0; j < 4; j++)
        if (L.ev_rb[j]) (void)hipEventDestroy(L.ev_rb[j]);
      if (L.ev_join) (void)hipEventDestroy(L.ev_join);
      L = LaneRes{};
    }
    p_err.release();
    p_stats.release();
    if (h_pinned) (void)hipHostFree(h_pinned);
    h_pinned = nullptr;
    for (hipEvent_t* e : {&ev0, &ev1, &ev_fork}) {
      if (*e) (void)hipEventDestroy(*e);
      *e = nullptr;
    }
    frame_sum.release(); frame_stage.release(); frame_rgb8.release();
    if (multi_stream) (void)hipStreamDestroy(multi_stream);
    multi_stream = nullptr;
    res_ready = false;
    noalloc_base = nullptr;
    noalloc_pool = PathPool{};
  }
};

struct chip_scene {
  HostScene host;
  bool committed = false;
  Flat flat;
  DeviceState dev;                                      // the device chip_commit ran on
  std::vector<std::unique_ptr<DeviceState>> replicas;   // chip_commit_devices: one per entry of its device list

  void release_device() {
    DeviceGuard guard;
    if (dev.device >= 0 && hipSetDevice(dev.device) == hipSuccess) dev.release_all();
    for (auto& r : replicas)
      if (r && r->device >= 0 && hipSetDevice(r->device) == hipSuccess) r->release_all();
    replicas.clear();
    committed = false;
  }
};

// Smallest traversal variant that covers the node families present in the flattened scene.
static uint32_t scene_features(const Flat& F) {
  const uint32_t ext = (!F.discs.empty() || !F.cones.empty() || !F.scales.empty()) ? (TF_QUADS | TF_XFORM | TF_EXT) : 0u;
  if (!F.media.empty()) {
    uint32_t f = TF_QUADS | TF_XFORM | TF_MEDIA | ext;
    // boundaries: bare spheres and boxes are queried in place; anything else needs the variant with the sub-traversal
    for (const DMedium& m : F.media) {
      if (m.bq_count != 0) f |= TF_MEDIA_BOX;
      else if (NREF_KIND(m.boundary) != NK_SPHERE) f |= TF_MEDIA_ANY;
    }
    if (ext) f |= TF_MEDIA_ANY;  // media together with discs / cones / Scale: only the variant with everything exists
    if (f & TF_MEDIA_ANY) f &= ~(uint32_t)TF_MEDIA_BOX;  // (that variant queries boxes in place as well)
    return f;
  }
  if (ext) return ext;
  if (!F.translates.empty() || !F.rotates.empty())
    return TF_QUADS | TF_XFORM | ((F.quads.empty() && F.rotates.empty()) ? TF_NO_QUADROT : 0u);
  if (!F.quads.empty() || !F.groups.empty()) return TF_QUADS;
  return 0;
}

static const uint32_t kMaxGridBlocksDef = 2048;
static const uint32_t kMaxGridBlocks = 2048;  // 256 CUs x 8 resident 256-thread blocks

// Largest path pools a render may use, all batches in flight together.  Every bounce of a batch is one round of kernel
// launches, and the late bounces of a batch carry too few paths to fill the device, so batches are made as large as the
// frame and the memory allow: up to 2^30 path slots (241 B each, 259 GB of an MI355X's 288 GiB; ensure_pool() takes less
// when less is free), at most 2^29 per pool.  The 1080p / 1024 spp frame then runs as four batches of 256 strata; with
// the earlier cap of 2^28 slots (16 batches) it took 425 ms instead of 347.
static const size_t kPoolPathsMaxEach = (size_t)1 << 29;
static size_t pool_capacity_max() {
  const char* e = getenv("CHIP_POOL_PATHS");
  if (e && *e) {
    long long v = atoll(e);
    if (v >= 4096) return (size_t)v;
  }
  return (size_t)1 << 30;
}
// (per-bounce arrays hold 1/16 more entries than the pool has path slots: PathPool::slack)
static const size_t kBytesPerPathSlot = (2 * sizeof(PathRay) + sizeof(PathHit) + 2 * sizeof(PathMap) + sizeof(uint32_t)) * 17 / 16 + sizeof(float4);

// The small per-device resources of a scene (with the device current): streams, events, pinned words, error and
// statistics words, and the address of this device's copy of the no-allocation scratch.
static int ensure_resources(DeviceState& D) {
  if (D.res_ready) return 0;
  HIP_TRY(D.p_err.reserve(4));
  HIP_TRY(D.p_stats.reserve(4));
  if (!D.h_pinned) HIP_TRY(hipHostMalloc((void**)&D.h_pinned, 64 * sizeof(uint32_t), hipHostMallocDefault));
  for (int k = 0; k < kMaxLanes; k++) {
    LaneRes& L = D.lanes[k];
    if (!L.h_live) HIP_TRY(hipHostMalloc((void**)&L.h_live, 16 * sizeof(uint32_t), hipHostMallocDefault));
    if (!L.stream) HIP_TRY(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
    for (int j = 0; j < 4; j++)
      if (!L.ev_rb[j]) HIP_TRY(hipEventCreateWithFlags(&L.ev_rb[j], hipEventDisableTiming));
    if (!L.ev_join) HIP_TRY(hipEventCreateWithFlags(&L.ev_join, hipEventDisableTiming));
  }
  if (!D.ev0) HIP_TRY(hipEventCreate(&D.ev0));
  if (!D.ev1) HIP_TRY(hipEventCreate(&D.ev1));
  if (!D.ev_fork) HIP_TRY(hipEventCreateWithFlags(&D.ev_fork, hipEventDisableTiming));
  if (!D.multi_stream) HIP_TRY(hipStreamCreateWithFlags(&D.multi_stream, hipStreamNonBlocking));
  {
    void* base = nullptr;
    HIP_TRY(hipGetSymbolAddress(&base, HIP_SYMBOL(scratch)));
    if (!base) return fail("hipGetSymbolAddress returned no address for the no-allocation scratch");
    D.noalloc_base = static_cast<unsigned char*>(base);
    NoAllocLayout* L = reinterpret_cast<NoAllocLayout*>(base);  // address arithmetic only: the host never dereferences it
    PathPool& P = D.noalloc_pool;
    P.ray[0] = L->ray; P.ray[1] = L->ray + CHIP_NOALLOC_PATHS;
    P.st[0] = L->st; P.st[1] = L->st + CHIP_NOALLOC_PATHS;
    P.hit = L->hit; P.sample = L->sample; P.q_lit = L->qlit;
    P.counters = L->counters; P.errflag = L->err; P.stats = L->stats; P.ovf = L->ovf;
    P.cap = CHIP_NOALLOC_PATHS;
    P.slack = 0;
    P.ovf_slabs = CHIP_NOALLOC_OVF_SLABS;
  }
  D.res_ready = true;
  return 0;
}

// Makes pool `which` hold min(want_paths, max_paths, what fits in free device memory) path slots.
// Replicas of one device allocate one after the other (a process-wide lock), each keeping to its share of the free memory.
static std::mutex g_pool_mutex;
static int ensure_pool(const chip_scene* s, DeviceState& D, int which, size_t want_paths, size_t max_paths, int mem_share = 1) {
  std::lock_guard<std::mutex> lock(g_pool_mutex);
  PoolBufs& pb = D.pools[which];
  uint32_t rows = s->host.cam.max_depth + 2;
  size_t cap = std::min(std::max<size_t>(want_paths, 4096), std::max<size_t>(max_paths, 4096));
  if (!(pb.ready && pb.P.cap >= cap)) {
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    if (pb.ready) free_b += (size_t)pb.P.cap * kBytesPerPathSlot;
    cap = std::min(cap, std::max<size_t>(4096, (size_t)(0.8 * (double)free_b / (double)std::max(mem_share, 1)) / kBytesPerPathSlot));
  }
  if (ensure_resources(D)) return -1;
  if (pb.ready && pb.P.cap >= cap && pb.counters_rows >= rows) return 0;
  if (!pb.ready) HIP_TRY(pb.ovf.alloc((size_t)kMaxGridBlocks * CHIP_BLOCK * CHIP_STACK_OVF * 2));
  if (!pb.ready || pb.P.cap < cap) {
    pb.ray.release(); pb.hit.release(); pb.st.release(); pb.sample.release();
    pb.qlit.release();
    pb.ready = false;
    // the arrays indexed by a bounce's entries have room for the empty entries a bounce may hold beside its live paths
    pb.slack = cap / 16;
    HIP_TRY(pb.ray.alloc(2 * (cap + pb.slack)));
    HIP_TRY(pb.hit.alloc(cap + pb.slack));
    HIP_TRY(pb.st.alloc(2 * (cap + pb.slack)));
    HIP_TRY(pb.sample.alloc(cap));
    HIP_TRY(pb.qlit.alloc(cap + pb.slack));
    pb.P.cap = (uint32_t)cap;
    pb.P.slack = (uint32_t)pb.slack;
  }
  if (pb.counters_rows < rows) {
    HIP_TRY(pb.counters.alloc((size_t)rows * CHIP_CTR_STRIDE));
    pb.counters_rows = rows;
  }
  PathPool& P = pb.P;
  P.ray[0] = pb.ray.p; P.ray[1] = pb.ray.p + (P.cap + pb.slack); P.st[0] = pb.st.p; P.st[1] = pb.st.p + (P.cap + pb.slack);
  P.hit = pb.hit.p; P.sample = pb.sample.p; P.q_lit = pb.qlit.p;
  P.counters = pb.counters.p; P.errflag = D.p_err.p; P.stats = D.p_stats.p; P.ovf = pb.ovf.p;
  P.ovf_slabs = kMaxGridBlocks * CHIP_BLOCK;
  pb.ready = true;
  return 0;
}

// ceil(2^64 / d) for magic_div (0 for d = 1, where the quotient is the dividend)
static void set_magic(uint32_t out[2], uint32_t d) {
  unsigned long long c = 0;
  if (d > 1) c = ~0ull / d + 1;  // floor((2^64 - 1) / d) + 1 = ceil(2^64 / d) for every d that does not divide 2^64 ...
  if (d > 1 && (d & (d - 1)) == 0) c = (1ull << 63) / (d >> 1);  // ... and for the powers of two exactly 2^64 / d
  out[0] = (uint32_t)c;
  out[1] = (uint32_t)(c >> 32);
}
static uint32_t grid_for(size_t n) {
  size_t b = (n + CHIP_BLOCK - 1) / CHIP_BLOCK;
  if (b < 1) b = 1;
  if (b > kMaxGridBlocks) b = kMaxGridBlocks;
  return (uint32_t)b;
}

struct RenderPlan {
  uint32_t xmin, xmax, ymin, ymax, rw, rh;
  uint32_t s_first, s_stride, s_count;
};

static int plan_render(chip_scene* s, const chip_render_opts* o, RenderPlan& pl) {
  if (!s->committed) return fail("scene not committed");
  const DCamera& cam = s->host.cam;
  pl.xmin = o->xmin; pl.xmax = o->xmax; pl.ymin = o->ymin; pl.ymax = o->ymax;
  if (pl.xmax == 0 && pl.ymax == 0) { pl.xmin = 0; pl.ymin = 0; pl.xmax = cam.image_width; pl.ymax = cam.image_height; }
  if (pl.xmax > cam.image_width || pl.ymax > cam.image_height || pl.xmin >= pl.xmax || pl.ymin >= pl.ymax)
    return fail("render rectangle is empty or outside the image");
  pl.rw = pl.xmax - pl.xmin;
  pl.rh = pl.ymax - pl.ymin;
  uint32_t stride = o->stratum_stride ? o->stratum_stride : 1;
  if (o->stratum_offset >= stride) return fail("stratum_offset must be < stratum_stride");
  uint32_t total = cam.sqrt_spp * cam.sqrt_spp;
  pl.s_first = o->stratum_offset;
  pl.s_stride = stride;
  pl.s_count = pl.s_first < total ? (total - pl.s_first + stride - 1) / stride : 0;
  return 0;
}

// Traversal launch: picks the variant for the scene's node families and, when the scene has a BVH,
// the LDS-cached form (one persistent block per CU; top of the BVH and leading spheres in LDS).
struct ExtendPlan {
  bool cache;
  bool small;  // bare spheres, everything LDS-resident, few enough for 11-bit references: the compact form (k_extend_small)
  CacheDesc cd;
  uint32_t block, lds_bytes;
};
static const uint32_t kLdsBytes = 160 * 1024;
// Tuning knobs read from the environment, once per render call (tests change them between calls of one process).
struct EnvKnobs {
  bool no_lds_cache = false, no_small = false, no_flat = false;
  int lds_stack = 0;          // 0 = the variant's default
  uint32_t small_stack = 16;  // stack words per lane of the compact form
  uint32_t small_queue = 16384;
  static EnvKnobs read() {
    EnvKnobs k;
    if (const char* e = getenv("CHIP_NO_LDS_CACHE")) k.no_lds_cache = *e == '1';
    if (const char* e = getenv("CHIP_NO_SMALL_FORM")) k.no_small = *e == '1';
    if (const char* e = getenv("CHIP_NO_FLAT_FORM")) k.no_flat = *e == '1';
    if (const char* e = getenv("CHIP_LDS_STACK")) { int v = atoi(e); if (v >= 2 && v <= 16) k.lds_stack = v; }
    if (const char* e = getenv("CHIP_SMALL_STACK")) { int v = atoi(e); if (v >= 4 && v <= 24) k.small_stack = (uint32_t)v; }
    if (const char* e = getenv("CHIP_SMALL_QUEUE")) { if (*e) k.small_queue = (uint32_t)atol(e); }
    return k;
  }
};
// uncached: the plain 256-thread form (literal passes)
static ExtendPlan plan_extend(const chip_scene* s, const EnvKnobs& env, uint32_t feat, bool uncached) {
  ExtendPlan p;
  p.cache = !uncached && !s->flat.bvh.empty() && !env.no_lds_cache;
  if (p.cache) {
    p.block = (uint32_t)extend_block(feat, true);
    // (value, entry distance) pairs per lane kept in LDS; the rest of a deep descent goes to the spill slab.  Deep
    // trees read from memory (instanced spheres) gain more from two further entries than from the 140 nodes of
    // cache they cost; the sphere scenes' trees are shallow
    p.cd.stack_entries = (feat & TF_NO_QUADROT) ? 10 : 8;
    if (env.lds_stack) p.cd.stack_entries = (uint32_t)env.lds_stack;
    const uint32_t stack_bytes = (p.cd.stack_entries + 1) * 8 * p.block;  // (+ the spare entry of st_push_sel)
    const uint32_t node_bytes = 16 * CHIP_WNODE_U4, sphere_bytes = 64;
    uint32_t budget = kLdsBytes - stack_bytes;
    p.cd.n_bvh = (uint32_t)std::min<size_t>(s->flat.n_wide, budget / node_bytes);
    budget -= p.cd.n_bvh * node_bytes;
    p.cd.n_sph = (uint32_t)std::min<size_t>(s->flat.spheres.size(), budget / sphere_bytes);
    p.lds_bytes = node_bytes * p.cd.n_bvh + sphere_bytes * p.cd.n_sph + stack_bytes;
    // the compact form: planned on its own (4-byte stack words, the whole scene cached or not at all)
    const uint32_t small_entries = env.small_stack;
    const size_t small_bytes = node_bytes * s->flat.n_wide + sphere_bytes * s->flat.spheres.size() + (size_t)small_entries * 4 * p.block;
    p.small = !env.no_small && feat == 0 && s->flat.n_wide <= CHIP_SM_MAX && s->flat.spheres.size() <= CHIP_SM_MAX && small_bytes <= kLdsBytes;
    if (p.small) {
      p.cd.stack_entries = small_entries;
      p.cd.n_bvh = (uint32_t)s->flat.n_wide;
      p.cd.n_sph = (uint32_t)s->flat.spheres.size();
      p.lds_bytes = (uint32_t)small_bytes;
    }
  } else {
    p.small = false;
    p.block = CHIP_BLOCK;
    p.cd.stack_entries = CHIP_STACK_LDS;
    p.cd.n_bvh = p.cd.n_sph = 0;
    p.lds_bytes = (CHIP_STACK_LDS + 1) * 8 * CHIP_BLOCK;
  }
  return p;
}
// The launch shapes of one render call: planned once, used by every bounce.
struct ExtendPlans {
  uint32_t feat;         // the feature set whose instantiation runs (extend_variant)
  ExtendPlan lit;        // the literal passes: the plain 256-thread form
  ExtendPlan full;       // long queues: the LDS-cached form when the scene has one
  ExtendPlan shortq;     // short queues (the late bounces)
  uint32_t small_queue;  // queues below this many entries are short
  bool flat;             // the world is tried as a list of primitives (k_extend_flat)
};
template <uint32_t F, bool CACHE, bool ORD, bool B0>
static int launch_extend_b(const DeviceState& D, const PathPool& P, const ExtendPlan& p, uint32_t n_live_upper, hipStream_t st, const BatchDesc& B,
                           uint32_t bounce, uint32_t lit_pass) {
  // the instantiation may use all of a CU's LDS: told to the runtime once per device
  static std::atomic<uint64_t> attr_done{0};
  const uint64_t bit = 1ull << ((unsigned)D.device & 63u);
  if (!(attr_done.load(std::memory_order_acquire) & bit)) {
    HIP_TRY(hipFuncSetAttribute((const void*)k_extend<F, CACHE, ORD, B0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
    attr_done.fetch_or(bit, std::memory_order_release);
  }
  // never launch a shape the instantiation was not compiled for
  if (p.block != (uint32_t)extend_block(F, CACHE)) return fail("internal: traversal launch planned for another kernel variant");
  uint32_t blocks = (n_live_upper + p.block - 1) / p.block;
  uint32_t max_blocks = CACHE ? (uint32_t)D.n_cus : kMaxGridBlocksDef;
  blocks = std::max(1u, std::min(blocks, max_blocks));
  // A literal pass that is expected to find its queue empty is one wave: on a device full of the other batch's
  // kernels a single wave finds a free slot sooner than a block of four, and it drains a queue of any length anyway.
  uint32_t threads = p.block;
  uint32_t lds_bytes = p.lds_bytes;
  if (!CACHE && lit_pass && n_live_upper <= 1u) {
    threads = 64u;
    lds_bytes = (p.cd.stack_entries + 1u) * 8u * threads;
  }
  // every thread of the launch owns a spill slab: the grid must not be larger than the pool's slab array
  if ((size_t)blocks * threads > P.ovf_slabs) return fail("internal: traversal launch larger than the pool's spill slabs");
  hipLaunchKernelGGL((k_extend<F, CACHE, ORD, B0>), dim3(blocks), dim3(threads), lds_bytes, st, D.dscene, P, B, bounce, p.cd, lit_pass);
  return 0;
}
template <bool B0>
static int launch_extend_small(const DeviceState& D, const PathPool& P, const ExtendPlan& p, uint32_t n_live_upper, hipStream_t st, const BatchDesc& B,
                               uint32_t bounce) {
  static std::atomic<uint64_t> attr_done{0};
  const uint64_t bit = 1ull << ((unsigned)D.device & 63u);
  if (!(attr_done.load(std::memory_order_acquire) & bit)) {
    HIP_TRY(hipFuncSetAttribute((const void*)k_extend_small<B0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
    attr_done.fetch_or(bit, std::memory_order_release);
  }
  if (p.block != (uint32_t)extend_block(0, true)) return fail("internal: traversal launch planned for another kernel variant");
  uint32_t blocks = (n_live_upper + p.block - 1) / p.block;
  blocks = std::max(1u, std::min(blocks, (uint32_t)D.n_cus));
  hipLaunchKernelGGL((k_extend_small<B0>), dim3(blocks), dim3(p.block), p.lds_bytes, st, D.dscene, P, B, bounce, p.cd);
  return 0;
}
// Bounce 0 of a small sphere scene with lean shading, traced and shaded by one kernel (k_first_small).
// waste_out: upper bound on the empty entries the launch may leave in bounce 1's arrays.
static int launch_first_small(const DeviceState& D, const PathPool& P, const ExtendPlan& p, uint32_t n_live_upper, hipStream_t st, const BatchDesc& B,
                              bool fast, uint32_t* waste_out) {
  static std::atomic<uint64_t> attr_done{0};
  const uint64_t bit = 1ull << ((unsigned)D.device & 63u);
  if (!(attr_done.load(std::memory_order_acquire) & bit)) {
    HIP_TRY(hipFuncSetAttribute((const void*)k_first_small<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
    HIP_TRY(hipFuncSetAttribute((const void*)k_first_small<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
    attr_done.fetch_or(bit, std::memory_order_release);
  }
  if (p.block != (uint32_t)extend_block(0, true)) return fail("internal: traversal launch planned for another kernel variant");
  uint32_t blocks = (n_live_upper + p.block - 1) / p.block;
  blocks = std::max(1u, std::min(blocks, (uint32_t)D.n_cus));
  // Entries of bounce 1 are handed to the waves in ranges (one atomic per range); every wave may leave part of its last
  // range empty, so the ranges are sized to keep that within the slack the pool's arrays have beyond `cap` entries.
  // Small launches take exactly what they need per wave and trip.
  const uint32_t n_waves = blocks * (p.block / 64u);
  uint32_t range = 0;
  if (n_live_upper >= (1u << 22) && P.slack >= n_live_upper / 16u) {
    range = (n_live_upper / (n_waves * 16u)) & ~63u;
    range = std::min(range, 2048u);
    if (range < 64u) range = 0;
  }
  *waste_out = range * n_waves;
  if (fast) hipLaunchKernelGGL((k_first_small<true>), dim3(blocks), dim3(p.block), p.lds_bytes, st, D.dscene, P, B, p.cd, range);
  else hipLaunchKernelGGL((k_first_small<false>), dim3(blocks), dim3(p.block), p.lds_bytes, st, D.dscene, P, B, p.cd, range);
  return 0;
}
// The tail of a batch in one kernel (k_tail_small), enqueued ahead of a bounce's ordinary launches: it takes the bounce
// (and every bounce after it) if the bounce holds at most max_live entries, and does nothing otherwise.
static int launch_tail_small(const DeviceState& D, const PathPool& P, const ExtendPlan& p, hipStream_t st, const BatchDesc& B, uint32_t bounce, bool fast,
                             uint32_t max_live) {
  static std::atomic<uint64_t> attr_done{0};
  const uint64_t bit = 1ull << ((unsigned)D.device & 63u);
  if (!(attr_done.load(std::memory_order_acquire) & bit)) {
    HIP_TRY(hipFuncSetAttribute((const void*)k_tail_small<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
    HIP_TRY(hipFuncSetAttribute((const void*)k_tail_small<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
    attr_done.fetch_or(bit, std::memory_order_release);
  }
  if (p.block != (uint32_t)extend_block(0, true)) return fail("internal: traversal launch planned for another kernel variant");
  uint32_t blocks = (std::min<uint32_t>(max_live, P.cap) + p.block - 1) / p.block;
  blocks = std::max(1u, std::min(blocks, (uint32_t)D.n_cus));
  // the literal form inside the kernel keeps its stack in the thread's spill slab: no more threads than slabs (the static
  // pool of the no-allocation entry point has few)
  blocks = std::min<uint32_t>(blocks, (uint32_t)(P.ovf_slabs / p.block));
  if (blocks == 0u) return 0;
  if (fast) hipLaunchKernelGGL((k_tail_small<true>), dim3(blocks), dim3(p.block), p.lds_bytes, st, D.dscene, P, B, bounce, p.cd, max_live);
  else hipLaunchKernelGGL((k_tail_small<false>), dim3(blocks), dim3(p.block), p.lds_bytes, st, D.dscene, P, B, bounce, p.cd, max_live);
  return 0;
}
template <uint32_t F, bool CACHE, bool ORD>
static int launch_extend_t(const DeviceState& D, const PathPool& P, const ExtendPlan& p, uint32_t n_live_upper, hipStream_t st, const BatchDesc& B,
                           uint32_t bounce, uint32_t lit_pass) {
  if (bounce == 0) return launch_extend_b<F, CACHE, ORD, true>(D, P, p, n_live_upper, st, B, bounce, lit_pass);
  return launch_extend_b<F, CACHE, ORD, false>(D, P, p, n_live_upper, st, B, bounce, lit_pass);
}
// The feature set whose traversal instantiation runs a scene with features `feat`: its own when one was compiled,
// else the one with every feature.
static uint32_t extend_variant(uint32_t feat) {
  switch (feat) {
    case 0:
    case TF_QUADS:
    case TF_QUADS | TF_XFORM:
    case TF_QUADS | TF_XFORM | TF_NO_QUADROT:
    case TF_QUADS | TF_XFORM | TF_EXT:
    case TF_QUADS | TF_XFORM | TF_MEDIA:
    case TF_QUADS | TF_XFORM | TF_MEDIA | TF_MEDIA_BOX:
    case TF_QUADS | TF_XFORM | TF_MEDIA | TF_MEDIA_ANY: return feat;
    default: return TF_QUADS | TF_XFORM | TF_MEDIA | TF_EXT | TF_MEDIA_ANY;
  }
}
static ExtendPlans plan_extends(const chip_scene* s, uint32_t scene_feat) {
  const EnvKnobs env = EnvKnobs::read();
  ExtendPlans X;
  // the launch shape (threads per block, LDS split) must be planned for the instantiation that runs
  X.feat = extend_variant(scene_feat);
  X.lit = plan_extend(s, env, X.feat, true);
  X.full = plan_extend(s, env, X.feat, false);
  // A short queue (the late bounces) is not worth a CU-filling block and its copy of the scene tables: it runs the
  // plain 256-thread form, whose blocks also find room sooner beside the other batch's kernels (+2 % on the sphere
  // and Cornell scenes).  Not for instanced spheres: their deep trees want the cached top levels even then.
  X.shortq = (X.feat & TF_NO_QUADROT) ? X.full : X.lit;
  X.small_queue = env.small_queue;
  // a few spheres and quads: the flat form (CHIP_NO_FLAT_FORM=1 keeps the hierarchy walk: tests compare the two)
  // (bare spheres walk their LDS-resident tree well: only a handful of them is faster as a list)
  X.flat = !env.no_flat && !s->flat.flat_list.empty() && !s->flat.loose_geometry && (X.feat & ~(TF_QUADS | TF_XFORM)) == 0u &&
           (X.feat != 0u || s->flat.flat_list.size() <= 8);
  return X;
}
// Whether bounce 0 runs in one kernel (k_first_small): small scenes of bare spheres (the compact form of the traversal)
// with lean shading (solid colours, no lights).  CHIP_NO_FUSED_FIRST=1 selects the two-kernel form (read per call: tests
// compare the two).
static bool fused_first_possible(const ExtendPlans& X, bool lean) {
  const char* e = getenv("CHIP_NO_FUSED_FIRST");
  return lean && X.feat == 0 && X.full.cache && X.full.small && !(e && *e == '1');
}
// One bounce of traversal: the ordered traversal followed by a literal pass over the rays it handed back (normally none; all of
// them in a reference_order render).
// lit_expected: whether the literal queue is expected to hold rays (it sizes the literal pass's grid only: one block
// drains a queue of any length, slowly).
// fused_first (bounce 0 only; 0 = off, 1 = exact shading, 2 = fast shading): the traversal kernel also shades its hits.
static int launch_extend(const DeviceState& D, const PathPool& P, const ExtendPlans& X, uint32_t n_live_upper, hipStream_t st,
                         const BatchDesc& B, uint32_t bounce, bool lit_expected, int fused_first = 0, uint32_t* waste_out = nullptr) {
  const uint32_t feat = X.feat;
  const ExtendPlan& lit = X.lit;
  const ExtendPlan& p = n_live_upper < X.small_queue ? X.shortq : X.full;
  const uint32_t lit_upper = B.reference_order ? n_live_upper : (lit_expected ? std::min<uint32_t>(n_live_upper, 64u * CHIP_BLOCK) : 1u);
#define CHIP_EXT(FEAT)                                                                                   \
  {                                                                                                      \
    int rc_ = p.cache ? launch_extend_t<FEAT, true, true>(D, P, p, n_live_upper, st, B, bounce, 0)       \
                      : launch_extend_t<FEAT, false, true>(D, P, p, n_live_upper, st, B, bounce, 0);     \
    if (rc_) return rc_;                                                                                 \
    return launch_extend_t<FEAT, false, false>(D, P, lit, lit_upper, st, B, bounce, 1);                  \
  }
  if (fused_first) {
    if (!(bounce == 0 && feat == 0 && X.full.cache && X.full.small) || !waste_out) return fail("internal: fused first bounce planned for a scene it does not cover");
    int rc_ = launch_first_small(D, P, X.full, n_live_upper, st, B, fused_first == 2, waste_out);
    if (rc_) return rc_;
    return launch_extend_t<0, false, false>(D, P, lit, lit_upper, st, B, bounce, 1);
  }
  if (X.flat && !B.reference_order) {
    if (D.dscene.n_flat == 0u || D.dscene.n_flat > CHIP_FLAT_MAX) return fail("internal: flat traversal planned for a scene without a primitive list");
    uint32_t blocks = (n_live_upper + CHIP_FLAT_BLOCK - 1) / CHIP_FLAT_BLOCK;
    blocks = std::max(1u, std::min(blocks, (uint32_t)D.n_cus * 32u));
    if (bounce == 0) hipLaunchKernelGGL((k_extend_flat<true>), dim3(blocks), dim3(CHIP_FLAT_BLOCK), 0, st, D.dscene, P, B, bounce);
    else hipLaunchKernelGGL((k_extend_flat<false>), dim3(blocks), dim3(CHIP_FLAT_BLOCK), 0, st, D.dscene, P, B, bounce);
    switch (feat) {  // the literal pass over the rays handed back (normally none)
      case 0: return launch_extend_t<0, false, false>(D, P, lit, lit_upper, st, B, bounce, 1);
      case TF_QUADS: return launch_extend_t<TF_QUADS, false, false>(D, P, lit, lit_upper, st, B, bounce, 1);
      case TF_QUADS | TF_XFORM: return launch_extend_t<TF_QUADS | TF_XFORM, false, false>(D, P, lit, lit_upper, st, B, bounce, 1);
      default: return fail("internal: flat traversal planned for a scene it does not cover");
    }
  }
  if (feat == 0 && p.cache && p.small) {
    int rc_ = bounce == 0 ? launch_extend_small<true>(D, P, p, n_live_upper, st, B, bounce) : launch_extend_small<false>(D, P, p, n_live_upper, st, B, bounce);
    if (rc_) return rc_;
    return launch_extend_t<0, false, false>(D, P, lit, lit_upper, st, B, bounce, 1);
  }
  switch (feat) {
    case 0: CHIP_EXT(0)
    case TF_QUADS: CHIP_EXT(TF_QUADS)
    case TF_QUADS | TF_XFORM: CHIP_EXT(TF_QUADS | TF_XFORM)
    case TF_QUADS | TF_XFORM | TF_NO_QUADROT: CHIP_EXT(TF_QUADS | TF_XFORM | TF_NO_QUADROT)
    case TF_QUADS | TF_XFORM | TF_EXT: CHIP_EXT(TF_QUADS | TF_XFORM | TF_EXT)
    case TF_QUADS | TF_XFORM | TF_MEDIA: CHIP_EXT(TF_QUADS | TF_XFORM | TF_MEDIA)
    case TF_QUADS | TF_XFORM | TF_MEDIA | TF_MEDIA_BOX: CHIP_EXT(TF_QUADS | TF_XFORM | TF_MEDIA | TF_MEDIA_BOX)
    case TF_QUADS | TF_XFORM | TF_MEDIA | TF_MEDIA_ANY: CHIP_EXT(TF_QUADS | TF_XFORM | TF_MEDIA | TF_MEDIA_ANY)
    default: CHIP_EXT(TF_QUADS | TF_XFORM | TF_MEDIA | TF_EXT | TF_MEDIA_ANY)
  }
#undef CHIP_EXT
}

// Adds the rectangle's radiance sums for the planned strata into d_sum (device, rw*rh*3 floats).
// The strata are cut into batches of as many paths as a pool holds.  With more than one batch, two
// batches are in flight at a time, each in its own pool on its own stream; their kernels are enqueued
// bounce by bounce in turn, so the device overlaps the traversal of one batch with the shading of the
// other.  Batches add their samples to d_sum in batch order (chained events), which keeps the f32 sums
// independent of how the device interleaved them.
// With `noalloc` the call uses the device's static scratch pool on the caller's stream (one batch at a time) and
// the streams, events and pinned words made at commit: it allocates nothing.
// mem_share: how many concurrent render calls share this device's free memory (replicas of one device).
static int render_into(chip_scene* s, DeviceState& D, const chip_render_opts* o, const RenderPlan& pl, float* d_sum, hipStream_t stream,
                       chip_stats* stats, bool noalloc, int mem_share = 1) {
  DeviceGuard guard;
  HIP_TRY(hipSetDevice(D.device));
  if (!D.uploaded) return fail("scene not committed on this device");
  const DScene& S = D.dscene;
  const DCamera& cam = s->host.cam;
  const bool time_kernels = !noalloc && (o->flags & CHIP_RENDER_TIME_KERNELS) != 0;
  // (a Disc with a normal that is not of unit length: the fast arithmetic's unit-normal shortcuts do not hold)
  const bool exact_shading = (o->flags & CHIP_RENDER_EXACT_SHADING) != 0 || s->flat.nonunit_normals != 0;
  const uint32_t feat = scene_features(s->flat);
  const ExtendPlans xplans = plan_extends(s, feat);
  // spheres with solid colours and no light sampling: the lean shade kernels
  bool solid_only = !s->flat.has_lights;
  for (const DTexture& t : s->host.textures) solid_only = solid_only && t.kind == TK_SOLID;
  const bool lean = feat == 0 && solid_only;
  // the same under Translate / Rotate instances (instanced spheres, nothing else to intersect)
  // solid colours and checkers only: the full shade kernel without its image / noise texture code
  bool plain_textures = true;
  for (const DTexture& t : s->host.textures) plain_textures = plain_textures && (t.kind == TK_SOLID || t.kind == TK_CHECKER);
  bool all_rich = !s->host.materials.empty();
  for (const DMaterial& m : s->host.materials) all_rich = all_rich && (m.flags & MF_RICH_TEX) != 0;
  const bool fused_first_ok = fused_first_possible(xplans, lean);
  // bounces of at most this many entries are finished by k_tail_small (0 = never; small sphere scenes with lean shading)
  uint32_t tail_paths = 1u << 22;
  bool tail_paths_given = false;
  if (const char* e = getenv("CHIP_TAIL_PATHS")) { if (*e) { tail_paths = (uint32_t)std::max(0LL, std::min(atoll(e), 1LL << 27)); tail_paths_given = true; } }
  static const uint32_t kTailFirstBounce = 1;  // (bounce 0 has no entries in memory: its paths are the batch)
  const bool lean_xform = solid_only && (feat & ~TF_NO_QUADROT) == (TF_QUADS | TF_XFORM) && s->flat.quads.empty() && s->flat.media.empty();
  // scenes of a few primitives, materials and lights: the shade kernel works from LDS copies of the record tables
  // (CHIP_NO_SHADE_TABLES=1 keeps them in memory: tests compare the two)
  ShadeTables shade_tbl{};
  bool shade_tbl_ok = false;
  if (!lean && !lean_xform && plain_textures) {
    const Flat& F = s->flat;
    shade_tbl = ShadeTables{(uint32_t)F.spheres.size(), (uint32_t)F.quads.size(), (uint32_t)F.translates.size(), (uint32_t)F.rotates.size(),
                            (uint32_t)F.xchain.size(), (uint32_t)F.media.size(), (uint32_t)s->host.materials.size(), (uint32_t)s->host.textures.size(),
                            (uint32_t)F.lights.size(), (uint32_t)F.ltree.size(), (uint32_t)F.discs.size(), (uint32_t)F.cones.size(), (uint32_t)F.scales.size()};
    const size_t records = F.spheres.size() + F.quads.size() + F.translates.size() + F.rotates.size() + F.xchain.size() + F.media.size() +
                           s->host.materials.size() + s->host.textures.size() + F.lights.size() + F.ltree.size() + F.discs.size() + F.cones.size() +
                           F.scales.size();
    uint32_t off[13];
    const char* e = getenv("CHIP_NO_SHADE_TABLES");
    shade_tbl_ok = records < 4096 && shade_tables_layout(shade_tbl, off) <= CHIP_SHADE_TBL_BYTES && !(e && *e == '1');
  }

  // pools: two when the render is large enough to be worth splitting
  int n_lanes = 1;
  size_t cap = 0;
  if (noalloc) {
    if (!D.res_ready || !D.noalloc_base) return fail("scene not committed on this device");
    if (cam.max_depth + 2 > CHIP_NOALLOC_COUNTER_ROWS) return fail("max_depth is beyond what the no-allocation entry point supports (254)");
    if ((size_t)pl.rw * pl.rh > CHIP_NOALLOC_PATHS) return fail("render rectangle holds more pixels than the no-allocation entry point supports (32768)");
    cap = CHIP_NOALLOC_PATHS;
  } else {
    const size_t want_total = (size_t)pl.rw * pl.rh * std::max<uint32_t>(pl.s_count, 1);
    const char* e_streams = getenv("CHIP_STREAMS");
    size_t overlap_min = (size_t)1 << 24;  // renders smaller than this are not worth splitting
    if (const char* e_min = getenv("CHIP_OVERLAP_MIN_PATHS")) overlap_min = (size_t)std::max<long long>(0, atoll(e_min));
    int lanes_wanted = 2;  // batches in flight (CHIP_STREAMS overrides: 1 = no overlap)
    if (e_streams && *e_streams) lanes_wanted = std::max(1, std::min(kMaxLanes, atoi(e_streams)));
    if (want_total < overlap_min) lanes_wanted = 1;
    n_lanes = (int)std::min<uint32_t>((uint32_t)lanes_wanted, std::max<uint32_t>(pl.s_count, 1));
    const size_t max_each = std::min(pool_capacity_max() / (size_t)n_lanes, kPoolPathsMaxEach);
    const size_t want_each = (size_t)pl.rw * pl.rh * ((std::max<uint32_t>(pl.s_count, 1) + n_lanes - 1) / n_lanes);
    for (int k = 0; k < n_lanes; k++)
      if (ensure_pool(s, D, k, want_each, max_each, mem_share)) return -1;
    cap = D.pools[0].P.cap;
    for (int k = 1; k < n_lanes; k++) cap = std::min<size_t>(cap, D.pools[k].P.cap);
    if (pl.rw > cap) return fail("render rectangle is wider than the path pool");
  }
  uint32_t* const d_err = noalloc ? D.noalloc_pool.errflag : D.p_err.p;
  unsigned long long* const d_stats = noalloc ? D.noalloc_pool.stats : D.p_stats.p;

  // batches: bands of rows (only when one stratum of the rectangle exceeds a pool) x groups of strata
  struct Batch {
    BatchDesc B;
    uint32_t by;
  };
  std::vector<Batch> batches;
  const uint32_t band_rows = std::min<uint32_t>(pl.rh, (uint32_t)(cap / pl.rw));
  for (uint32_t by = 0; by < pl.rh; by += band_rows) {
    const uint32_t bh = std::min(band_rows, pl.rh - by);
    const uint32_t npix = pl.rw * bh;
    uint32_t spb = std::max<uint32_t>(1, (uint32_t)(cap / npix));
    if (o->strata_per_batch) {
      spb = std::min(spb, o->strata_per_batch);
    } else if (n_lanes >= 2 && pl.s_count > spb) {
      // batches of equal size, as many as a multiple of the batches in flight: no lane is left with one batch more
      uint32_t nb = (pl.s_count + spb - 1) / spb;
      nb = (nb + (uint32_t)n_lanes - 1) / (uint32_t)n_lanes * (uint32_t)n_lanes;
      spb = (pl.s_count + nb - 1) / nb;
    }
    for (uint32_t sb0 = 0; sb0 < pl.s_count; sb0 += spb) {
      Batch bt;
      BatchDesc& B = bt.B;
      B.rect_x0 = pl.xmin; B.rect_y0 = pl.ymin + by; B.rect_w = pl.rw; B.rect_h = bh;
      B.npix = npix;
      set_magic(B.npix_magic, npix);
      set_magic(B.rect_w_magic, pl.rw);
      set_magic(B.sqrt_spp_magic, cam.sqrt_spp);
      {
        static const bool linear = [] { const char* e = getenv("CHIP_LINEAR_PIXELS"); return e && *e == '1'; }();
        B.tiles_per_row = (!linear && pl.rw % 8u == 0 && bh % 8u == 0) ? pl.rw / 8u : 0u;
        set_magic(B.tiles_magic, B.tiles_per_row ? B.tiles_per_row : 1u);
      }
      B.n_strata = std::min(spb, pl.s_count - sb0);
      B.stratum_first = pl.s_first + sb0 * pl.s_stride;
      B.stratum_stride = pl.s_stride;
      B.n_paths = npix * B.n_strata;
      B.seed_lo = (uint32_t)o->seed;
      B.seed_hi = (uint32_t)(o->seed >> 32);
      B.reference_order = ((o->flags & CHIP_RENDER_REFERENCE_ORDER) || S.literal_only) ? 1u : 0u;
      B.ray48 = exact_shading ? 0u : 1u;  // (the kernels of the default shading mode read and write the 48-byte ray record)
      bt.by = by;
      batches.push_back(bt);
    }
  }

  struct Lane {
    const PathPool* P = nullptr;
    LaneRes* res = nullptr;
    hipStream_t st = nullptr;
    bool active = false;
    bool traced = false;  // all bounces of the batch are enqueued; only the accumulation is still to come
    size_t batch = 0;
    uint32_t bounce = 0, live_upper = 0;
    bool lit_expected = true;  // until a readback says the literal queue stayed empty
    // live-count readbacks in flight: the host runs ahead of the device and only learns a bounce's live
    // count (which sizes later grids and ends the batch early) when the copy has landed
    uint32_t rb_head = 0, rb_count = 0;
    std::vector<hipEvent_t> kev;  // (trace begin, trace end = shade begin, shade end) per bounce
  };
  Lane lanes[kMaxLanes];
  const bool forked = n_lanes >= 2 && batches.size() >= 2;
  const int n_active_lanes = forked ? (int)std::min<size_t>((size_t)n_lanes, batches.size()) : 1;
  // A render of no more batches than are in flight at once (one rank's share of a multi-GPU frame, a small frame) has no
  // later batch to run beside a tail: the kernel that takes the tail then runs on a device it cannot fill, so it takes
  // over later (an eighth of the headline frame's strata: 47.8 ms with 4 M entries, 45.6 ms with 2 M)
  if (!tail_paths_given && batches.size() <= (size_t)n_active_lanes) tail_paths = 1u << 21;
  EventBag bag;  // events of this call only (kernel timing, batch-order chain)
#define RENDER_TRY(expr)                                                                        \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) {                                                                     \
      bag.failed = true;                                                                        \
      return fail(std::string(#expr) + ": " + hipGetErrorString(e_));                           \
    }                                                                                           \
  } while (0)
  std::vector<hipEvent_t> ev_acc(forked ? batches.size() : 0, nullptr);  // batch i has added its samples to d_sum
  RENDER_TRY(hipMemsetAsync(d_err, 0, 4 * sizeof(uint32_t), stream));
  RENDER_TRY(hipMemsetAsync(d_stats, 0, 4 * sizeof(unsigned long long), stream));
  RENDER_TRY(hipEventRecord(D.ev0, stream));
  if (forked) RENDER_TRY(hipEventRecord(D.ev_fork, stream));
  for (int k = 0; k < n_active_lanes; k++) {
    lanes[k].P = noalloc ? &D.noalloc_pool : &D.pools[k].P;
    lanes[k].res = &D.lanes[k];
    lanes[k].st = forked ? D.lanes[k].stream : stream;
    if (forked) RENDER_TRY(hipStreamWaitEvent(lanes[k].st, D.ev_fork, 0));
  }
  uint64_t n_paths_total = 0, n_trace_launches = 0;
  size_t next_batch = 0;

  auto finish_batch = [&](Lane& L) -> int {
    const Batch& bt = batches[L.batch];
    // the sums take the batches in order: a batch whose predecessor (on another stream) has not been
    // enqueued to its end yet waits on the host and is asked again on the next round
    L.traced = true;
    if (forked && L.batch > 0 && !ev_acc[L.batch - 1]) return 0;
    if (forked && L.batch > 0) RENDER_TRY(hipStreamWaitEvent(L.st, ev_acc[L.batch - 1], 0));
    hipLaunchKernelGGL(k_accumulate, dim3(grid_for(bt.B.npix)), dim3(CHIP_BLOCK), 0, L.st, *L.P, bt.B, d_sum + (size_t)bt.by * pl.rw * 3);
    RENDER_TRY(hipGetLastError());
    if (forked) {
      RENDER_TRY(bag.make(&ev_acc[L.batch], false));
      RENDER_TRY(hipEventRecord(ev_acc[L.batch], L.st));
    }
    L.active = false;
    L.traced = false;
    return 0;
  };
  static const uint32_t shade_grid_max = [] {
    const char* e = getenv("CHIP_SHADE_GRID");
    long v = e && *e ? atol(e) : 0;
    return v >= 1 ? (uint32_t)v : kMaxGridBlocks;
  }();
  // Enqueues the lane's batch from its current bounce up to the next live-count readback (or its end).
  auto advance = [&](Lane& L) -> int {
    const Batch& bt = batches[L.batch];
    const BatchDesc& B = bt.B;
    const PathPool& P = *L.P;
    auto stamp = [&]() -> int {
      hipEvent_t e;
      RENDER_TRY(bag.make(&e, true));
      L.kev.push_back(e);
      RENDER_TRY(hipEventRecord(e, L.st));
      return 0;
    };
    while (L.bounce < cam.max_depth) {
      const uint32_t bounce = L.bounce;
      const uint32_t grid_full = std::min(grid_for(L.live_upper), shade_grid_max);
      if (time_kernels && stamp()) return -1;
      // bounce 0 of a small sphere scene with lean shading: one kernel traces and shades; the shade launch below then
      // only takes the rays that went through the literal queue (normally none: one block)
      const bool fused = bounce == 0 && fused_first_ok && !B.reference_order;
      // once few paths are left, one kernel follows them to their ends (it declines while the bounce holds more)
      if (tail_paths && fused_first_ok && !B.reference_order && bounce >= kTailFirstBounce && L.live_upper > 0) {
        if (launch_tail_small(D, P, xplans.full, L.st, B, bounce, !exact_shading, tail_paths)) {
          bag.failed = true;
          return -1;
        }
        if (L.live_upper <= tail_paths && (size_t)P.ovf_slabs >= xplans.full.block) {
          // the bounce cannot hold more than the kernel takes: the batch ends here, with no launch left to enqueue
          if (time_kernels && (stamp() || stamp())) return -1;
          n_trace_launches++;
          L.bounce = cam.max_depth;
          break;
        }
      }
      uint32_t waste = 0;
      if (launch_extend(D, P, xplans, L.live_upper, L.st, B, bounce, L.lit_expected, fused ? (exact_shading ? 1 : 2) : 0, &waste)) {
        bag.failed = true;
        return -1;
      }
      if (time_kernels && stamp()) return -1;
      n_trace_launches++;
      const uint32_t shade_pass_all = fused ? (uint32_t)SHADE_LIT_QUEUE : (uint32_t)SHADE_ALL;
      const uint32_t grid = fused ? (L.lit_expected ? std::min(grid_full, 64u) : 1u) : grid_full;
      if (fused) L.live_upper += waste;  // (bounce 1 may hold that many empty entries beside its paths)
      {
        const bool ext = (feat & TF_EXT) != 0;
#define CHIP_SHADE(FAST, LEAN, EXT, PASS)                                                                                          \
  do {                                                                                                                         \
    if (bounce == 0) hipLaunchKernelGGL((k_shade_tiles<FAST, LEAN, EXT, true>), dim3(grid), dim3(CHIP_BLOCK), 0, L.st, S, P, B, bounce, PASS, ShadeTables{}); \
    else hipLaunchKernelGGL((k_shade_tiles<FAST, LEAN, EXT, false>), dim3(grid), dim3(CHIP_BLOCK), 0, L.st, S, P, B, bounce, PASS, ShadeTables{}); \
  } while (0)
        // the full kernel's form without image / noise texture code: everything (SHADE_ALL) when the scene has no such
        // texture, else everything but the hits on materials that have one, which a second launch of the full kernel takes
#define CHIP_SHADE_PLAIN(FAST, EXT, PASS)                                                                                         \
  do {                                                                                                                         \
    if (bounce == 0) hipLaunchKernelGGL((k_shade_tiles<FAST, false, EXT, true, true>), dim3(grid), dim3(CHIP_BLOCK), 0, L.st, S, P, B, bounce, PASS, ShadeTables{}); \
    else hipLaunchKernelGGL((k_shade_tiles<FAST, false, EXT, false, true>), dim3(grid), dim3(CHIP_BLOCK), 0, L.st, S, P, B, bounce, PASS, ShadeTables{}); \
  } while (0)
        // the same with the scene's record tables copied into LDS (scenes of a few primitives, materials and lights)
#define CHIP_SHADE_PLAIN_TBL(FAST, EXT, PASS)                                                                                     \
  do {                                                                                                                         \
    if (bounce == 0) hipLaunchKernelGGL((k_shade_tiles<FAST, false, EXT, true, true, true>), dim3(grid), dim3(CHIP_BLOCK), 0, L.st, S, P, B, bounce, PASS, shade_tbl); \
    else hipLaunchKernelGGL((k_shade_tiles<FAST, false, EXT, false, true, true>), dim3(grid), dim3(CHIP_BLOCK), 0, L.st, S, P, B, bounce, PASS, shade_tbl); \
  } while (0)
        if (lean) {
          if (exact_shading) CHIP_SHADE(false, true, false, shade_pass_all); else CHIP_SHADE(true, true, false, shade_pass_all);
        } else if (lean_xform) {
          if (exact_shading) CHIP_SHADE(false, true, true, SHADE_ALL); else CHIP_SHADE(true, true, true, SHADE_ALL);
        } else if (plain_textures && shade_tbl_ok) {
          if (ext) { if (exact_shading) CHIP_SHADE_PLAIN_TBL(false, true, SHADE_ALL); else CHIP_SHADE_PLAIN_TBL(true, true, SHADE_ALL); }
          else { if (exact_shading) CHIP_SHADE_PLAIN_TBL(false, false, SHADE_ALL); else CHIP_SHADE_PLAIN_TBL(true, false, SHADE_ALL); }
        } else if (plain_textures) {
          if (ext) { if (exact_shading) CHIP_SHADE_PLAIN(false, true, SHADE_ALL); else CHIP_SHADE_PLAIN(true, true, SHADE_ALL); }
          else { if (exact_shading) CHIP_SHADE_PLAIN(false, false, SHADE_ALL); else CHIP_SHADE_PLAIN(true, false, SHADE_ALL); }
        } else if (all_rich) {  // every material carries an image / noise texture: the full kernel takes the whole bounce
          if (ext) { if (exact_shading) CHIP_SHADE(false, false, true, SHADE_ALL); else CHIP_SHADE(true, false, true, SHADE_ALL); }
          else { if (exact_shading) CHIP_SHADE(false, false, false, SHADE_ALL); else CHIP_SHADE(true, false, false, SHADE_ALL); }
        } else if (ext) {
          if (exact_shading) CHIP_SHADE_PLAIN(false, true, SHADE_PLAIN_ONLY); else CHIP_SHADE_PLAIN(true, true, SHADE_PLAIN_ONLY);
          // (the second pass goes over the hits the first one listed for it: a fraction of the bounce, and a grid to match)
          const uint32_t grid_all = grid;
          { const uint32_t grid = std::max(1u, grid_all / 4u);
            if (exact_shading) CHIP_SHADE(false, false, true, SHADE_RICH_ONLY); else CHIP_SHADE(true, false, true, SHADE_RICH_ONLY); }
        } else {
          if (exact_shading) CHIP_SHADE_PLAIN(false, false, SHADE_PLAIN_ONLY); else CHIP_SHADE_PLAIN(true, false, SHADE_PLAIN_ONLY);
          const uint32_t grid_all = grid;
          { const uint32_t grid = std::max(1u, grid_all / 4u);
            if (exact_shading) CHIP_SHADE(false, false, false, SHADE_RICH_ONLY); else CHIP_SHADE(true, false, false, SHADE_RICH_ONLY); }
        }
#undef CHIP_SHADE_PLAIN_TBL
#undef CHIP_SHADE_PLAIN
#undef CHIP_SHADE
      }
      if (time_kernels && stamp()) return -1;
      L.bounce++;
      // every few bounces learn how many paths are still alive (sizes the grids, stops the loop early)
      if (L.bounce < cam.max_depth && (bounce % 4) == 3) {
        const uint32_t j = (L.rb_head + L.rb_count) % 4;
        RENDER_TRY(hipMemcpyAsync(L.res->h_live + j, P.counters + (size_t)L.bounce * CHIP_CTR_STRIDE, sizeof(uint32_t), hipMemcpyDeviceToHost, L.st));
        // ... and how many rays the bounce just enqueued handed to the literal pass (normally none)
        RENDER_TRY(hipMemcpyAsync(L.res->h_live + 4 + j, P.counters + (size_t)bounce * CHIP_CTR_STRIDE + CHIP_CTR_LIT, sizeof(uint32_t), hipMemcpyDeviceToHost, L.st));
        RENDER_TRY(hipEventRecord(L.res->ev_rb[j], L.st));
        L.rb_count++;
        return 0;
      }
    }
    return finish_batch(L);
  };
  // Takes the readbacks that have landed (waiting for the oldest when `must` or when the ring is full).
  auto poll_readbacks = [&](Lane& L, bool must) -> int {
    while (L.rb_count) {
      const uint32_t j = L.rb_head;
      if (must || L.rb_count >= 3) {
        RENDER_TRY(hipEventSynchronize(L.res->ev_rb[j]));
      } else {
        hipError_t q = hipEventQuery(L.res->ev_rb[j]);
        if (q == hipErrorNotReady) break;
        RENDER_TRY(q);
      }
      must = false;
      L.live_upper = std::min(L.live_upper, L.res->h_live[j]);
      L.lit_expected = L.res->h_live[4 + j] != 0;
      L.rb_head = (L.rb_head + 1) % 4;
      L.rb_count--;
    }
    return 0;
  };

  for (;;) {
    bool any = false;
    for (int k = 0; k < n_active_lanes; k++) {
      Lane& L = lanes[k];
      if (!L.active) {
        if (next_batch >= batches.size()) continue;
        L.batch = next_batch++;
        L.active = true;
        L.bounce = 0;
        L.lit_expected = true;
        const Batch& bt = batches[L.batch];
        const PathPool& P = *L.P;
        n_paths_total += bt.B.n_paths;
        RENDER_TRY(hipMemsetAsync(P.counters, 0, (size_t)(cam.max_depth + 2) * CHIP_CTR_STRIDE * sizeof(uint32_t), L.st));
        hipLaunchKernelGGL(k_begin_batch, dim3(1), dim3(64), 0, L.st, P, bt.B);
        // a path writes its sample when it ends; with max_depth 0 no path ever starts (camera.rs:259-261)
        if (cam.max_depth == 0) RENDER_TRY(hipMemsetAsync(P.sample, 0, (size_t)bt.B.n_paths * sizeof(float4), L.st));
        L.live_upper = bt.B.n_paths;
      }
      any = true;
      if (poll_readbacks(L, false)) return -1;
      if (L.traced || L.live_upper == 0) {
        if (finish_batch(L)) return -1;
      } else if (advance(L)) {
        return -1;
      }
      if (!L.active) {  // the batch is fully enqueued: its readbacks are of no further use
        // (a copy still in flight lands before the next batch's copy into the same slot: one stream, in order)
        L.rb_head = 0;
        L.rb_count = 0;
      }
    }
    if (!any) break;
  }
  if (forked) {
    for (int k = 0; k < n_active_lanes; k++) {
      RENDER_TRY(hipEventRecord(lanes[k].res->ev_join, lanes[k].st));
      RENDER_TRY(hipStreamWaitEvent(stream, lanes[k].res->ev_join, 0));
    }
  }
  RENDER_TRY(hipEventRecord(D.ev1, stream));
  RENDER_TRY(hipMemcpyAsync(D.h_pinned + 8, d_err, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
  RENDER_TRY(hipMemcpyAsync(D.h_pinned + 16, d_stats, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
  RENDER_TRY(hipStreamSynchronize(stream));
  RENDER_TRY(hipGetLastError());
  float ms = 0.f;
  double ms_trace = 0.0, ms_shade = 0.0;
  RENDER_TRY(hipEventElapsedTime(&ms, D.ev0, D.ev1));
  for (int k = 0; k < n_active_lanes; k++) {
    const std::vector<hipEvent_t>& kev = lanes[k].kev;
    for (size_t i = 0; i + 2 < kev.size(); i += 3) {
      float a = 0.f, b = 0.f;
      (void)hipEventElapsedTime(&a, kev[i], kev[i + 1]);
      (void)hipEventElapsedTime(&b, kev[i + 1], kev[i + 2]);
      ms_trace += a;
      ms_shade += b;
    }
  }
#undef RENDER_TRY
  if (D.h_pinned[8] & 1u) return fail("traversal stack overflow (scene nesting deeper than the device stack)");
  if (D.h_pinned[9] & 1u) return fail("internal: a path record named a path outside the batch");
  if (stats) {
    const unsigned long long* st = (const unsigned long long*)(D.h_pinned + 16);
    stats->paths = n_paths_total;
    stats->segments = st[0];
    stats->bvh_visits = st[1];
    stats->prim_tests = st[2];
    stats->ms_total = ms;
    stats->ms_trace = ms_trace;  // summed over both streams when two batches overlap
    stats->ms_shade = ms_shade;
    stats->ms_other = 0.0;
    stats->trace_launches = n_trace_launches;
    stats->batches = batches.size();
  }
  return 0;
}

// Flattened tables -> device `D.device` (current on entry).
static int upload_scene(chip_scene* s, DeviceState& D, const Flat& F) {
  D.uploaded = false;
  {
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, D.device));
    D.n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  if (ensure_resources(D)) return -1;
  // the traversal addresses culling nodes by 32-bit byte offsets of 128 bytes each
  if (F.wnodes.size() >= ((size_t)1 << 25)) return fail("scene has 2^25 or more culling nodes");
  HIP_TRY(D.d_bvh.upload(F.bvh));
  HIP_TRY(D.d_wnodes.upload(F.wnodes));
  HIP_TRY(D.d_spheres.upload(F.spheres));
  HIP_TRY(D.d_quads.upload(F.quads));
  HIP_TRY(D.d_groups.upload(F.groups));
  HIP_TRY(D.d_gchild.upload(F.gchild));
  HIP_TRY(D.d_translates.upload(F.translates));
  HIP_TRY(D.d_rotates.upload(F.rotates));
  HIP_TRY(D.d_scales.upload(F.scales));
  HIP_TRY(D.d_discs.upload(F.discs));
  HIP_TRY(D.d_cones.upload(F.cones));
  HIP_TRY(D.d_media.upload(F.media));
  HIP_TRY(D.d_xchain.upload(F.xchain));
  HIP_TRY(D.d_materials.upload(s->host.materials));
  HIP_TRY(D.d_textures.upload(s->host.textures));
  HIP_TRY(D.d_perlins.upload(s->host.perlins));
  HIP_TRY(D.d_texels.upload(s->host.texels));
  HIP_TRY(D.d_lights.upload(F.lights));
  HIP_TRY(D.d_ltree.upload(F.ltree));
  HIP_TRY(D.d_flat.upload(F.flat_list));
  DScene& T = D.dscene;
  T.bvh = D.d_bvh.p; T.wnodes = D.d_wnodes.p; T.spheres = D.d_spheres.p; T.quads = D.d_quads.p; T.groups = D.d_groups.p; T.gchild = D.d_gchild.p;
  T.translates = D.d_translates.p; T.rotates = D.d_rotates.p; T.scales = D.d_scales.p; T.discs = D.d_discs.p; T.cones = D.d_cones.p;
  T.media = D.d_media.p; T.xchain = D.d_xchain.p;
  T.materials = D.d_materials.p; T.textures = D.d_textures.p; T.perlins = D.d_perlins.p; T.texels = D.d_texels.p;
  T.lights = D.d_lights.p;
  T.ltree = D.d_ltree.p;
  T.flat_list = D.d_flat.p;
  T.n_flat = (uint32_t)F.flat_list.size();
  T.root = F.root;
  T.ord_root = F.ord_root;
  T.n_lights = (uint32_t)F.lights.size();
  T.has_lights = F.has_lights;
  T.lights_is_group = F.lights_is_group;
  T.literal_only = F.loose_geometry;
  T.moving_spheres = 0;
  for (const DSphere& sp : F.spheres)
    for (int i = 0; i < 3; i++) {
      uint64_t bits;
      memcpy(&bits, &sp.dc[i], 8);
      if (bits != 0) T.moving_spheres = 1;
    }
  T.cam = s->host.cam;
  D.uploaded = true;
  return 0;
}

// a += src[0] + src[1] + ... over n floats, each element's additions in list order (the sums of other GPUs added to
// this one's: the same bits whichever GPU finished first)
#define CHIP_ADD_MAX 16
struct AddList {
  const float* src[CHIP_ADD_MAX];
  uint32_t n;
};
__global__ void __launch_bounds__(CHIP_BLOCK) k_add_sums(float* a, AddList L, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float v = a[i];
    for (uint32_t k = 0; k < L.n; k++) v += L.src[k][i];
    a[i] = v;
  }
}

extern "C" {

const char* chip_last_error(void) { return g_err.c_str(); }

int chip_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return fail(std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  return n;
}

chip_scene* chip_scene_create(void) { return new chip_scene(); }
void chip_scene_destroy(chip_scene* s) {
  if (!s) return;
  s->release_device();
  delete s;
}

#define NEED_SCENE if (!s) return fail("null scene")
#define CHECK_TEX(id) if ((id) < 0 || (size_t)(id) >= s->host.textures.size()) return fail("bad texture id")
#define CHECK_MAT(id) if ((id) < 0 || (size_t)(id) >= s->host.materials.size()) return fail("bad material id")
#define CHECK_NODE(id) if ((id) < 0 || (size_t)(id) >= s->host.handle.size()) return fail("bad node id")

static int add_tex(chip_scene* s, const DTexture& t) {
  s->host.textures.push_back(t);
  s->committed = false;
  return (int)s->host.textures.size() - 1;
}
static int add_mat(chip_scene* s, uint32_t kind, int tex, double p0, double p1, double p2, double p3) {
  DMaterial m;
  m.kind = kind;
  m.tex = tex;
  m.p[0] = p0; m.p[1] = p1; m.p[2] = p2; m.p[3] = p3;
  m.flags = 0;
  m.pad = 0;
  s->host.materials.push_back(m);
  s->committed = false;
  return (int)s->host.materials.size() - 1;
}
static DTexture solid_tex(double r, double g, double b) {
  DTexture t;
  memset(&t, 0, sizeof(t));
  t.kind = TK_SOLID;
  t.p[0] = r; t.p[1] = g; t.p[2] = b;
  return t;
}

int chip_tex_solid(chip_scene* s, double r, double g, double b) {
  NEED_SCENE;
  return add_tex(s, solid_tex(r, g, b));
}
int chip_tex_checker(chip_scene* s, double scale, int even, int odd) {
  NEED_SCENE;
  CHECK_TEX(even);
  CHECK_TEX(odd);
  DTexture t;
  memset(&t, 0, sizeof(t));
  t.kind = TK_CHECKER;
  t.a = (uint32_t)even;
  t.b = (uint32_t)odd;
  t.p[0] = 1.0 / scale;  // checker_texture.rs:15
  return add_tex(s, t);
}
int chip_tex_image(chip_scene* s, uint32_t w, uint32_t h, const uint8_t* rgb) {
  NEED_SCENE;
  if (!rgb || w == 0 || h == 0) return fail("bad image");
  DTexture t;
  memset(&t, 0, sizeof(t));
  t.kind = TK_IMAGE;
  t.a = w;
  t.b = h;
  t.c = (uint32_t)s->host.texels.size();
  s->host.texels.insert(s->host.texels.end(), rgb, rgb + (size_t)w * h * 3);
  return add_tex(s, t);
}
static int add_perlin(chip_scene* s, const double* rand_vec, const int32_t* px, const int32_t* py, const int32_t* pz) {
  DPerlin p;
  memcpy(p.rand_vec, rand_vec, sizeof(p.rand_vec));
  for (int i = 0; i < 256; i++) {
    if ((uint32_t)px[i] > 255 || (uint32_t)py[i] > 255 || (uint32_t)pz[i] > 255) return fail("perlin permutation entry out of range");
    p.perm_x[i] = px[i]; p.perm_y[i] = py[i]; p.perm_z[i] = pz[i];
  }
  s->host.perlins.push_back(p);
  return (int)s->host.perlins.size() - 1;
}
int chip_tex_perlin_noise(chip_scene* s, double scale, const double* rand_vec, const int32_t* px, const int32_t* py, const int32_t* pz) {
  NEED_SCENE;
  if (!rand_vec || !px || !py || !pz) return fail("null perlin table");
  int pi = add_perlin(s, rand_vec, px, py, pz);
  if (pi < 0) return -1;
  DTexture t;
  memset(&t, 0, sizeof(t));
  t.kind = TK_NOISE;
  t.a = (uint32_t)pi;
  t.p[0] = scale;
  return add_tex(s, t);
}
int chip_tex_perlin_turbulence(chip_scene* s, double scale, uint32_t depth, const double* rand_vec, const int32_t* px, const int32_t* py,
                               const int32_t* pz) {
  NEED_SCENE;
  if (!rand_vec || !px || !py || !pz) return fail("null perlin table");
  int pi = add_perlin(s, rand_vec, px, py, pz);
  if (pi < 0) return -1;
  DTexture t;
  memset(&t, 0, sizeof(t));
  t.kind = TK_TURBULENCE;
  t.a = (uint32_t)pi;
  t.b = depth;
  t.p[0] = scale;
  return add_tex(s, t);
}

int chip_lambertian_color(chip_scene* s, double r, double g, double b) {
  NEED_SCENE;
  int t = add_tex(s, solid_tex(r, g, b));
  return add_mat(s, MK_LAMBERTIAN, t, 0, 0, 0, 0);
}
int chip_lambertian_tex(chip_scene* s, int tex) {
  NEED_SCENE;
  CHECK_TEX(tex);
  return add_mat(s, MK_LAMBERTIAN, tex, 0, 0, 0, 0);
}
int chip_metal(chip_scene* s, double r, double g, double b, double fuzz) {
  NEED_SCENE;
  return add_mat(s, MK_METAL, -1, r, g, b, fuzz);
}
int chip_dielectric(chip_scene* s, double ri) {
  NEED_SCENE;
  return add_mat(s, MK_DIELECTRIC, -1, ri, 0, 0, 0);
}
int chip_diffuse_light_color(chip_scene* s, double r, double g, double b) {
  NEED_SCENE;
  int t = add_tex(s, solid_tex(r, g, b));
  return add_mat(s, MK_DIFFUSE_LIGHT, t, 0, 0, 0, 0);
}
int chip_diffuse_light_tex(chip_scene* s, int tex) {
  NEED_SCENE;
  CHECK_TEX(tex);
  return add_mat(s, MK_DIFFUSE_LIGHT, tex, 0, 0, 0, 0);
}
int chip_empty_material(chip_scene* s) {
  NEED_SCENE;
  return add_mat(s, MK_EMPTY, -1, 0, 0, 0, 0);
}

int chip_sphere(chip_scene* s, const double* c, double r, int mat) {
  NEED_SCENE;
  CHECK_MAT(mat);
  s->committed = false;
  return s->host.nodes[s->host.make_sphere(c, r, mat, true)].id;
}
int chip_sphere_set_direction(chip_scene* s, int node, const double* d) {
  NEED_SCENE;
  CHECK_NODE(node);
  int h = s->host.handle[node];
  if (s->host.nodes[h].kind != CHIP_NODE_SPHERE) return fail("node is not a sphere");
  if (s->host.nodes[h].referenced) return fail("sphere already referenced");
  s->host.sphere_set_direction(h, d);
  s->committed = false;
  return 0;
}
int chip_quad(chip_scene* s, const double* q, const double* u, const double* v, int mat) {
  NEED_SCENE;
  CHECK_MAT(mat);
  s->committed = false;
  return s->host.nodes[s->host.make_quad(q, u, v, mat, true)].id;
}
int chip_box(chip_scene* s, const double* a, const double* b, int mat) {
  NEED_SCENE;
  CHECK_MAT(mat);
  s->committed = false;
  return s->host.nodes[s->host.make_box(a, b, mat)].id;
}
static int to_host_ids(chip_scene* s, const int32_t* ids, int n, std::vector<int>& out) {
  if (n < 0 || (n > 0 && !ids)) return fail("bad node list");
  out.resize(n);
  for (int i = 0; i < n; i++) {
    CHECK_NODE(ids[i]);
    out[i] = s->host.handle[ids[i]];
  }
  return 0;
}
int chip_group(chip_scene* s, const int32_t* ids, int n) {
  NEED_SCENE;
  std::vector<int> v;
  if (to_host_ids(s, ids, n, v)) return -1;
  s->committed = false;
  return s->host.nodes[s->host.make_group(v, CHIP_NODE_GROUP, true)].id;
}
int chip_bvh(chip_scene* s, const int32_t* ids, int n) {
  NEED_SCENE;
  std::vector<int> v;
  if (to_host_ids(s, ids, n, v)) return -1;
  s->committed = false;
  return s->host.nodes[s->host.make_bvh(v, true)].id;
}
int chip_translate(chip_scene* s, int node, const double* off) {
  NEED_SCENE;
  CHECK_NODE(node);
  s->committed = false;
  return s->host.nodes[s->host.make_translate(s->host.handle[node], off)].id;
}
int chip_rotate(chip_scene* s, int node, const double* axis, double deg) {
  NEED_SCENE;
  CHECK_NODE(node);
  s->committed = false;
  return s->host.nodes[s->host.make_rotate(s->host.handle[node], axis, deg)].id;
}
int chip_scale(chip_scene* s, int node, double sx, double sy, double sz) {
  NEED_SCENE;
  CHECK_NODE(node);
  s->committed = false;
  return s->host.nodes[s->host.make_scale(s->host.handle[node], sx, sy, sz)].id;
}
int chip_disc(chip_scene* s, const double* center, double radius, const double* normal, int mat) {
  NEED_SCENE;
  CHECK_MAT(mat);
  if (!center || !normal) return fail("null argument");
  s->committed = false;
  return s->host.nodes[s->host.make_disc(center, radius, normal, mat, true)].id;
}
int chip_cone_frustum(chip_scene* s, const double* base, double height, double top_radius, double bottom_radius, int mat) {
  NEED_SCENE;
  CHECK_MAT(mat);
  if (!base) return fail("null argument");
  s->committed = false;
  return s->host.nodes[s->host.make_cone_frustum(base, height, top_radius, bottom_radius, mat)].id;
}
int chip_constant_medium_color(chip_scene* s, int boundary, double density, double r, double g, double b) {
  NEED_SCENE;
  CHECK_NODE(boundary);
  int t = add_tex(s, solid_tex(r, g, b));
  int m = add_mat(s, MK_ISOTROPIC, t, 0, 0, 0, 0);  // isotropic.rs:21-25
  return s->host.nodes[s->host.make_medium(s->host.handle[boundary], density, m)].id;
}
int chip_constant_medium_tex(chip_scene* s, int boundary, double density, int tex) {
  NEED_SCENE;
  CHECK_NODE(boundary);
  CHECK_TEX(tex);
  int m = add_mat(s, MK_ISOTROPIC, tex, 0, 0, 0, 0);  // isotropic.rs:17-19
  return s->host.nodes[s->host.make_medium(s->host.handle[boundary], density, m)].id;
}

int64_t chip_scene_from_flat(chip_scene* s, const chip_flat_scene* flat, int32_t* node_ids) {
  NEED_SCENE;
  if (!flat || (flat->n_nodes && !flat->nodes) || (flat->n_members && !flat->members)) return fail("null argument");
  if (flat->n_nodes > (size_t)INT32_MAX) return fail("too many records");
  std::vector<int32_t> made(flat->n_nodes, -1);
  std::vector<int32_t> group;
  char msg[160];
  for (size_t i = 0; i < flat->n_nodes; i++) {
    const chip_flat_node& R = flat->nodes[i];
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
      case CHIP_NODE_SPHERE:
        id = chip_sphere(s, R.v, R.v[3], R.mat);
        if (id >= 0 && (R.flags & CHIP_FLAT_MOVING) && chip_sphere_set_direction(s, id, R.v + 4) < 0) id = -1;
        break;
      case CHIP_NODE_QUAD: id = chip_quad(s, R.v, R.v + 3, R.v + 6, R.mat); break;
      case CHIP_NODE_BOX: id = chip_box(s, R.v, R.v + 3, R.mat); break;
      case CHIP_NODE_DISC: id = chip_disc(s, R.v, R.v[3], R.v + 4, R.mat); break;
      case CHIP_NODE_CONE: id = chip_cone_frustum(s, R.v, R.v[3], R.v[4], R.v[5], R.mat); break;
      case CHIP_NODE_GROUP:
      case CHIP_NODE_BVH: {
        if (R.first < 0 || R.count < 0 || (size_t)R.first + (size_t)R.count > flat->n_members) {
          snprintf(msg, sizeof msg, "flat record %zu: member range outside the member list", i);
          return fail(msg);
        }
        group.resize((size_t)R.count);
        for (int32_t j = 0; j < R.count; j++) group[(size_t)j] = resolve(flat->members[(size_t)R.first + (size_t)j]);
        if (!bad_ref) id = R.kind == CHIP_NODE_GROUP ? chip_group(s, group.data(), R.count) : chip_bvh(s, group.data(), R.count);
        break;
      }
      case CHIP_NODE_TRANSLATE:
      case CHIP_NODE_ROTATE:
      case CHIP_NODE_SCALE:
      case CHIP_NODE_MEDIUM: {
        const int32_t child = resolve(R.first);
        if (bad_ref) break;
        if (R.kind == CHIP_NODE_TRANSLATE) id = chip_translate(s, child, R.v);
        else if (R.kind == CHIP_NODE_ROTATE) id = chip_rotate(s, child, R.v, R.v[3]);
        else if (R.kind == CHIP_NODE_SCALE) id = chip_scale(s, child, R.v[0], R.v[1], R.v[2]);
        else if (R.flags & CHIP_FLAT_MEDIUM_TEX) id = chip_constant_medium_tex(s, child, R.v[0], R.mat);
        else id = chip_constant_medium_color(s, child, R.v[0], R.v[1], R.v[2], R.v[3]);
        break;
      }
      default:
        snprintf(msg, sizeof msg, "flat record %zu: kind %d has no constructor", i, (int)R.kind);
        return fail(msg);
    }
    if (bad_ref) {
      snprintf(msg, sizeof msg, "flat record %zu refers to a record that is not before it", i);
      return fail(msg);
    }
    if (id < 0) {
      snprintf(msg, sizeof msg, "flat record %zu: %s", i, chip_last_error());
      return fail(msg);
    }
    made[i] = id;
  }
  if (node_ids) std::copy(made.begin(), made.end(), node_ids);
  return (int64_t)flat->n_nodes;
}

int chip_scene_set_world(chip_scene* s, int node) {
  NEED_SCENE;
  CHECK_NODE(node);
  s->host.world = node;
  s->committed = false;
  return 0;
}
int chip_scene_set_lights(chip_scene* s, int node) {
  NEED_SCENE;
  if (node >= 0) CHECK_NODE(node);
  s->host.lights = node < 0 ? -1 : node;
  s->committed = false;
  return 0;
}
int chip_scene_set_camera(chip_scene* s, const chip_camera_desc* d) {
  NEED_SCENE;
  if (!d) return fail("null camera desc");
  if (d->image_width == 0) return fail("image_width must be > 0");
  if (d->samples_per_pixel == 0) return fail("samples_per_pixel must be > 0");
  s->host.build_camera(*d);
  s->committed = false;
  return 0;
}

int chip_commit(chip_scene* s) {
  NEED_SCENE;
  s->committed = false;  // a commit that fails midway leaves the scene uncommitted, not pointing at freed tables
  Flat F;
  try {
    s->host.flatten(F);
  } catch (const std::exception& e) {
    return fail(e.what());
  }
  if (F.has_lights && F.lights.empty() && !F.lights_is_group) return fail("internal: empty light list");
  // a material needs (u, v) only when an image texture is reachable from its texture (checker sub-textures included)
  {
    const std::vector<DTexture>& T = s->host.textures;
    std::vector<int> uv(T.size(), -1);
    std::function<bool(int, int)> needs = [&](int t, int depth) -> bool {
      if (t < 0 || (size_t)t >= T.size() || depth > 64) return true;
      if (uv[t] >= 0) return uv[t] != 0;
      bool r = T[t].kind == TK_IMAGE || (T[t].kind == TK_CHECKER && (needs((int)T[t].a, depth + 1) || needs((int)T[t].b, depth + 1)));
      uv[t] = r ? 1 : 0;
      return r;
    };
    std::vector<int> rich_of(T.size(), -1);
    std::function<bool(int, int)> rich = [&](int t, int depth) -> bool {
      if (t < 0 || (size_t)t >= T.size() || depth > 64) return true;
      if (rich_of[t] >= 0) return rich_of[t] != 0;
      bool r = (T[t].kind != TK_SOLID && T[t].kind != TK_CHECKER) ||
               (T[t].kind == TK_CHECKER && (rich((int)T[t].a, depth + 1) || rich((int)T[t].b, depth + 1)));
      rich_of[t] = r ? 1 : 0;
      return r;
    };
    for (DMaterial& m : s->host.materials) {
      m.flags = (m.tex >= 0 && needs(m.tex, 0)) ? MF_NEEDS_UV : 0u;
      if (m.tex >= 0 && rich(m.tex, 0)) m.flags |= MF_RICH_TEX;
      // a solid-colour texture's value rides in the material record (read by the lean shade kernels)
      if ((m.kind == MK_LAMBERTIAN || m.kind == MK_DIFFUSE_LIGHT || m.kind == MK_ISOTROPIC) && m.tex >= 0 && (size_t)m.tex < T.size() &&
          T[m.tex].kind == TK_SOLID)
        for (int c = 0; c < 3; c++) m.p[c] = T[m.tex].p[c];
    }
  }
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  // replicas made by an earlier chip_commit_devices describe the previous scene
  if (!s->replicas.empty()) {
    DeviceGuard guard;
    for (auto& r : s->replicas)
      if (r && r->device >= 0 && hipSetDevice(r->device) == hipSuccess) r->release_all();
    s->replicas.clear();
  }
  // committing under another current device than last time: what was made on the old device goes
  if (s->dev.device >= 0 && s->dev.device != dev) {
    DeviceGuard guard;
    if (hipSetDevice(s->dev.device) == hipSuccess) s->dev.release_all();
  }
  s->dev.device = dev;
  if (upload_scene(s, s->dev, F)) return -1;
  s->flat = std::move(F);
  s->committed = true;
  return 0;
}

int chip_commit_devices(chip_scene* s, const int* devices, int n) {
  NEED_SCENE;
  if (!s->committed) return fail("chip_commit_devices: commit the scene first");
  if (n < 1 || n > 64 || !devices) return fail("chip_commit_devices: need 1 to 64 devices");
  int count = 0;
  HIP_TRY(hipGetDeviceCount(&count));
  for (int i = 0; i < n; i++)
    if (devices[i] < 0 || devices[i] >= count) return fail("chip_commit_devices: no such device");
  DeviceGuard guard;
  for (auto& r : s->replicas)
    if (r && r->device >= 0 && hipSetDevice(r->device) == hipSuccess) r->release_all();
  s->replicas.clear();
  for (int i = 0; i < n; i++) {
    std::unique_ptr<DeviceState> r(new DeviceState());
    r->device = devices[i];
    HIP_TRY(hipSetDevice(devices[i]));
    int rc = upload_scene(s, *r, s->flat);
    s->replicas.push_back(std::move(r));  // (kept also when the upload failed: what it allocated is released with the others)
    if (rc) return -1;
  }
  // the sums travel from every replica's device to the first one's: direct peer access where the machine offers it
  for (int i = 1; i < n; i++) {
    if (devices[i] == devices[0]) continue;
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, devices[0], devices[i]) == hipSuccess && can) {
      HIP_TRY(hipSetDevice(devices[0]));
      hipError_t e = hipDeviceEnablePeerAccess(devices[i], 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail(std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
      (void)hipGetLastError();
    }
  }
  return 0;
}

int chip_image_size(chip_scene* s, uint32_t* w, uint32_t* h) {
  NEED_SCENE;
  if (!s->host.has_camera) return fail("scene has no camera");
  if (w) *w = s->host.cam.image_width;
  if (h) *h = s->host.cam.image_height;
  return 0;
}
int chip_sqrt_spp(chip_scene* s) {
  NEED_SCENE;
  if (!s->host.has_camera) return fail("scene has no camera");
  return (int)s->host.cam.sqrt_spp;
}
int chip_scene_traits(chip_scene* s) {
  NEED_SCENE;
  if (!s->committed) return fail("scene not committed");
  int bits = 0;
  if (s->flat.nonunit_normals) bits |= CHIP_SCENE_EXACT_SHADING_FORCED;
  if (s->flat.loose_geometry) bits |= CHIP_SCENE_REFERENCE_ORDER_FORCED;
  const uint32_t feat = scene_features(s->flat);
  bool solid_only = !s->flat.has_lights;
  for (const DTexture& t : s->host.textures) solid_only = solid_only && t.kind == TK_SOLID;
  const ExtendPlans X = plan_extends(s, feat);
  if (fused_first_possible(X, feat == 0 && solid_only) && !s->flat.loose_geometry) bits |= CHIP_SCENE_FUSED_FIRST_BOUNCE;
  return bits;
}
int chip_camera_derived(chip_scene* s, double* out) {
  NEED_SCENE;
  if (!s->host.has_camera) return fail("scene has no camera");
  const DCamera& c = s->host.cam;
  const double* vs[6] = {c.center, c.pixel00, c.du, c.dv, c.disk_u, c.disk_v};
  for (int i = 0; i < 6; i++)
    for (int k = 0; k < 3; k++) out[3 * i + k] = vs[i][k];
  out[18] = c.pixel_samples_scale;
  out[19] = c.recip_sqrt_spp;
  out[20] = c.defocus_angle;
  out[21] = c.sqrt_spp;
  out[22] = c.image_width;
  out[23] = c.image_height;
  return 0;
}

int chip_render_accumulate_device(chip_scene* s, const chip_render_opts* opts, float* d_sum_rgb, void* stream, chip_stats* stats) {
  NEED_SCENE;
  if (!opts || !d_sum_rgb) return fail("null argument");
  RenderPlan pl;
  if (plan_render(s, opts, pl)) return -1;
  return render_into(s, s->dev, opts, pl, d_sum_rgb, (hipStream_t)stream, stats, false);
}

int chip_render_accumulate_noalloc(chip_scene* s, const chip_render_opts* opts, float* d_sum_rgb, void* stream, chip_stats* stats) {
  NEED_SCENE;
  if (!opts || !d_sum_rgb) return fail("null argument");
  RenderPlan pl;
  if (plan_render(s, opts, pl)) return -1;
  return render_into(s, s->dev, opts, pl, d_sum_rgb, (hipStream_t)stream, stats, true);
}
uint32_t chip_noalloc_max_pixels(void) { return CHIP_NOALLOC_PATHS; }

int chip_resolve_device(chip_scene* s, const float* d_sum_rgb, size_t npixels, float* d_out_gamma, uint8_t* d_out_rgb8, void* stream) {
  return chip_resolve_device_flags(s, d_sum_rgb, npixels, d_out_gamma, d_out_rgb8, 0, stream);
}
static int resolve_on(chip_scene* s, DeviceState& D, const float* d_sum_rgb, size_t npixels, float* d_out_gamma, uint8_t* d_out_rgb8, uint32_t flags,
                      hipStream_t stream) {
  DeviceGuard guard;
  HIP_TRY(hipSetDevice(D.device));
  hipLaunchKernelGGL(k_resolve, dim3(grid_for(npixels * 3)), dim3(CHIP_BLOCK), 0, stream, d_sum_rgb, npixels * 3, s->host.cam.pixel_samples_scale,
                     d_out_gamma, d_out_rgb8, (flags & CHIP_RENDER_QUANTISE_255) ? 255.0 : 255.999);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(stream));
  return 0;
}
int chip_resolve_device_flags(chip_scene* s, const float* d_sum_rgb, size_t npixels, float* d_out_gamma, uint8_t* d_out_rgb8, uint32_t flags,
                              void* stream) {
  NEED_SCENE;
  if (!s->host.has_camera) return fail("scene has no camera");
  if (!s->committed) return fail("scene not committed");
  if (!d_sum_rgb) return fail("null argument");
  return resolve_on(s, s->dev, d_sum_rgb, npixels, d_out_gamma, d_out_rgb8, flags, (hipStream_t)stream);
}

// The sums of `D` (device buffer, npix * 3 floats) to the host and / or resolved to 8 bits.
static int frame_outputs(chip_scene* s, DeviceState& D, size_t npix, uint32_t flags, float* out_sum_rgb, uint8_t* out_rgb8) {
  DeviceGuard guard;
  HIP_TRY(hipSetDevice(D.device));
  if (out_sum_rgb) HIP_TRY(hipMemcpy(out_sum_rgb, D.frame_sum.p, npix * 3 * sizeof(float), hipMemcpyDeviceToHost));
  if (out_rgb8) {
    HIP_TRY(D.frame_rgb8.reserve(npix * 3));
    if (resolve_on(s, D, D.frame_sum.p, npix, nullptr, D.frame_rgb8.p, flags, nullptr)) return -1;
    HIP_TRY(hipMemcpy(out_rgb8, D.frame_rgb8.p, npix * 3, hipMemcpyDeviceToHost));
  }
  return 0;
}

int chip_render_frame(chip_scene* s, const chip_render_opts* opts, float* out_sum_rgb, uint8_t* out_rgb8, chip_stats* stats) {
  NEED_SCENE;
  if (!opts) return fail("null argument");
  RenderPlan pl;
  if (plan_render(s, opts, pl)) return -1;
  DeviceState& D = s->dev;
  DeviceGuard guard;
  HIP_TRY(hipSetDevice(D.device));
  const size_t npix = (size_t)pl.rw * pl.rh;
  HIP_TRY(D.frame_sum.reserve(npix * 3));  // kept between calls (grows only)
  HIP_TRY(hipMemsetAsync(D.frame_sum.p, 0, npix * 3 * sizeof(float), nullptr));
  if (render_into(s, D, opts, pl, D.frame_sum.p, nullptr, stats, false)) return -1;
  return frame_outputs(s, D, npix, opts->flags, out_sum_rgb, out_rgb8);
}

int chip_render_frame_multi(chip_scene* s, const chip_render_opts* opts, float* out_sum_rgb, uint8_t* out_rgb8, chip_stats* stats) {
  NEED_SCENE;
  if (!opts) return fail("null argument");
  const int n = (int)s->replicas.size();
  if (n < 1) return fail("chip_render_frame_multi: call chip_commit_devices first");
  RenderPlan pl;
  if (plan_render(s, opts, pl)) return -1;
  const size_t npix = (size_t)pl.rw * pl.rh;
  const uint32_t stride0 = opts->stratum_stride ? opts->stratum_stride : 1;
  DeviceGuard guard;
  // How the call is dealt to the replicas.  With at least as many strata as replicas, replica g takes every n-th of
  // the strata this call covers (crates/cli/src/main.rs:119-163 deals blocks of pixels to threads; the samples of a
  // pixel are just as independent, camera.rs:335-341) and the replicas' sums are added.  With fewer strata than
  // replicas (a progressive pass, a low-spp preview) that would leave GPUs idle, so the rectangle is cut into bands of
  // rows instead, as the reference cuts the image into blocks (main.rs:83-106): replica g renders every stratum of
  // band g, and the bands are copied side by side, with no addition at all.
  const bool by_rows = pl.s_count < (uint32_t)n && pl.rh >= 2;
  const int n_used = by_rows ? (int)std::min<uint32_t>((uint32_t)n, pl.rh) : n;
  auto band_row = [&](int g) -> uint32_t {  // first row of band g, on multiples of 8 rows where the rectangle allows
    if (g >= n_used) return pl.rh;
    const uint32_t unit = pl.rh >= 8u * (uint32_t)n_used ? 8u : 1u;
    return (uint32_t)((uint64_t)(pl.rh / unit) * (uint32_t)g / (uint32_t)n_used) * unit;
  };
  DeviceState& D0 = *s->replicas[0];
  // replicas per device: the path pools of replicas that share a device share its free memory
  std::vector<int> share((size_t)n, 1);
  for (int g = 0; g < n; g++) {
    int c = 0;
    for (int h = 0; h < n_used; h++) c += s->replicas[(size_t)h]->device == s->replicas[(size_t)g]->device ? 1 : 0;
    share[(size_t)g] = std::max(c, 1);
  }
  for (int g = 0; g < n_used; g++) {
    DeviceState& D = *s->replicas[(size_t)g];
    HIP_TRY(hipSetDevice(D.device));
    if (ensure_resources(D)) return -1;
    HIP_TRY(D.frame_sum.reserve(npix * 3));
    HIP_TRY(hipMemsetAsync(D.frame_sum.p, 0, npix * 3 * sizeof(float), D.multi_stream));
    HIP_TRY(hipStreamSynchronize(D.multi_stream));  // (other replicas write into the first one's buffer in the row-band mode)
  }
  HIP_TRY(hipSetDevice(D0.device));
  // a staging slot on the first device for every replica that lives on another one (strata mode)
  std::vector<const float*> src((size_t)n_used, nullptr);
  if (!by_rows) {
    size_t slots = 0;
    for (int g = 1; g < n_used; g++) slots += s->replicas[(size_t)g]->device != D0.device ? 1 : 0;
    if (slots) HIP_TRY(D0.frame_stage.reserve(slots * npix * 3));
    size_t k = 0;
    for (int g = 0; g < n_used; g++) {
      DeviceState& D = *s->replicas[(size_t)g];
      src[(size_t)g] = (g == 0 || D.device == D0.device) ? D.frame_sum.p : D0.frame_stage.p + (k++) * npix * 3;
    }
  }
  std::vector<int> rc((size_t)n_used, 0);
  std::vector<std::string> err((size_t)n_used);
  std::vector<chip_stats> st((size_t)n_used);
  // One host thread per replica: it renders on the replica's own stream and, as soon as its sums are complete, sends
  // them to the first device from that same stream, so the transfers of replicas that finish early run while the
  // others still render.
  std::vector<std::thread> workers;
  for (int g = 0; g < n_used; g++)
    workers.emplace_back([&, g]() {
      DeviceState& D = *s->replicas[(size_t)g];
      auto bad = [&](const std::string& what) {
        rc[(size_t)g] = -1;
        err[(size_t)g] = what;
      };
      chip_render_opts o = *opts;
      RenderPlan plg = pl;
      if (by_rows) {
        const uint32_t r0 = band_row(g), r1 = band_row(g + 1);
        if (r1 <= r0) return;
        plg.ymin = pl.ymin + r0; plg.ymax = pl.ymin + r1; plg.rh = r1 - r0;
      } else {
        o.stratum_offset = opts->stratum_offset + (uint32_t)g * stride0;
        o.stratum_stride = stride0 * (uint32_t)n_used;
        if (plan_render(s, &o, plg)) return bad(g_err);
      }
      if (render_into(s, D, &o, plg, D.frame_sum.p, D.multi_stream, &st[(size_t)g], false, share[(size_t)g])) return bad(g_err);  // (the message lives in this thread)
      hipError_t e = hipSetDevice(D.device);
      if (e == hipSuccess && by_rows && g > 0) {
        // band g of the first device's frame, straight from this replica's band buffer
        float* dst = D0.frame_sum.p + (size_t)band_row(g) * pl.rw * 3;
        const size_t bytes = (size_t)plg.rh * pl.rw * 3 * sizeof(float);
        e = D.device == D0.device ? hipMemcpyAsync(dst, D.frame_sum.p, bytes, hipMemcpyDeviceToDevice, D.multi_stream)
                                  : hipMemcpyPeerAsync(dst, D0.device, D.frame_sum.p, D.device, bytes, D.multi_stream);
      } else if (e == hipSuccess && !by_rows && src[(size_t)g] != D.frame_sum.p) {
        e = hipMemcpyPeerAsync(const_cast<float*>(src[(size_t)g]), D0.device, D.frame_sum.p, D.device, npix * 3 * sizeof(float), D.multi_stream);
      }
      if (e == hipSuccess) e = hipStreamSynchronize(D.multi_stream);
      if (e != hipSuccess) bad(std::string("gathering the sums: ") + hipGetErrorString(e));
    });
  for (std::thread& t : workers) t.join();
  for (int g = 0; g < n_used; g++)
    if (rc[(size_t)g]) return fail("replica " + std::to_string(g) + ": " + err[(size_t)g]);
  HIP_TRY(hipSetDevice(D0.device));
  if (!by_rows && n_used > 1) {
    // one pass adds the replicas' sums in replica order, so the result does not depend on which finished first
    AddList L;
    for (int g0 = 1; g0 < n_used; g0 += CHIP_ADD_MAX) {
      L.n = (uint32_t)std::min(CHIP_ADD_MAX, n_used - g0);
      for (uint32_t k = 0; k < L.n; k++) L.src[k] = src[(size_t)g0 + k];
      hipLaunchKernelGGL(k_add_sums, dim3(grid_for(npix * 3)), dim3(CHIP_BLOCK), 0, D0.multi_stream, D0.frame_sum.p, L, npix * 3);
      HIP_TRY(hipGetLastError());
    }
  }
  HIP_TRY(hipStreamSynchronize(D0.multi_stream));
  if (stats) {
    *stats = st[0];
    for (int g = 1; g < n_used; g++) {
      stats->paths += st[(size_t)g].paths; stats->segments += st[(size_t)g].segments; stats->bvh_visits += st[(size_t)g].bvh_visits;
      stats->prim_tests += st[(size_t)g].prim_tests; stats->trace_launches += st[(size_t)g].trace_launches; stats->batches += st[(size_t)g].batches;
      stats->ms_total = std::max(stats->ms_total, st[(size_t)g].ms_total);
      stats->ms_trace += st[(size_t)g].ms_trace; stats->ms_shade += st[(size_t)g].ms_shade;
    }
  }
  return frame_outputs(s, D0, npix, opts->flags, out_sum_rgb, out_rgb8);
}

int chip_render_frame_blocks(chip_scene* s, const chip_render_opts* opts, uint32_t block_size, uint32_t passes, chip_block_fn on_block, void* user,
                             chip_stats* stats) {
  NEED_SCENE;
  if (!opts || !on_block) return fail("null argument");
  if (block_size == 0) return fail("block_size must be > 0");
  RenderPlan pl;
  if (plan_render(s, opts, pl)) return -1;
  DeviceState& D = s->dev;
  DeviceGuard guard;
  HIP_TRY(hipSetDevice(D.device));
  const size_t npix = (size_t)pl.rw * pl.rh;
  HIP_TRY(D.frame_sum.reserve(npix * 3));
  HIP_TRY(D.frame_rgb8.reserve(npix * 3));
  HIP_TRY(hipMemsetAsync(D.frame_sum.p, 0, npix * 3 * sizeof(float), nullptr));
  const uint32_t total = pl.s_count;
  passes = std::max<uint32_t>(1, std::min<uint32_t>(passes, std::max<uint32_t>(total, 1)));
  std::vector<uint8_t> rgb8(npix * 3), blockbuf;
  chip_stats sum{};
  uint32_t done = 0;
  for (uint32_t p = 0; p < passes; p++) {
    // pass p adds strata [p * total / passes, (p + 1) * total / passes) of this call's list to the running sums
    const uint32_t i0 = (uint32_t)((uint64_t)p * total / passes), i1 = (uint32_t)((uint64_t)(p + 1) * total / passes);
    RenderPlan sub = pl;
    sub.s_first = pl.s_first + i0 * pl.s_stride;
    sub.s_count = i1 - i0;
    chip_stats st{};
    if (sub.s_count && render_into(s, D, opts, sub, D.frame_sum.p, nullptr, &st, false)) return -1;
    done = i1;
    sum.paths += st.paths; sum.segments += st.segments; sum.bvh_visits += st.bvh_visits; sum.prim_tests += st.prim_tests;
    sum.ms_total += st.ms_total; sum.ms_trace += st.ms_trace; sum.ms_shade += st.ms_shade; sum.trace_launches += st.trace_launches;
    sum.batches += st.batches;
    // camera.rs:343-344 with the samples taken so far (after the last pass of a call that covers every stratum this is
    // pixel_samples_scale itself, so the final blocks are chip_render_frame's bytes)
    const double scale = 1.0 / (double)std::max<uint32_t>(done, 1);
    hipLaunchKernelGGL(k_resolve, dim3(grid_for(npix * 3)), dim3(CHIP_BLOCK), 0, nullptr, D.frame_sum.p, npix * 3, scale, (float*)nullptr, D.frame_rgb8.p,
                       (opts->flags & CHIP_RENDER_QUANTISE_255) ? 255.0 : 255.999);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(rgb8.data(), D.frame_rgb8.p, npix * 3, hipMemcpyDeviceToHost));
    // the blocks of crates/cli/src/main.rs:83-106 (rows of block_size x block_size tiles, clipped at the image edge),
    // handed over in the order its workers pop them: last pushed first (main.rs:132)
    const uint32_t nbx = (pl.rw + block_size - 1) / block_size, nby = (pl.rh + block_size - 1) / block_size;
    for (uint32_t b = nbx * nby; b-- > 0;) {
      const uint32_t bx = b % nbx, by = b / nbx;
      const uint32_t x0 = bx * block_size, x1 = std::min(x0 + block_size, pl.rw), y0 = by * block_size, y1 = std::min(y0 + block_size, pl.rh);
      blockbuf.resize((size_t)(x1 - x0) * (y1 - y0) * 3);
      for (uint32_t y = y0; y < y1; y++)
        std::memcpy(blockbuf.data() + (size_t)(y - y0) * (x1 - x0) * 3, rgb8.data() + ((size_t)y * pl.rw + x0) * 3, (size_t)(x1 - x0) * 3);
      on_block(user, pl.xmin + x0, pl.xmin + x1, pl.ymin + y0, pl.ymin + y1, blockbuf.data(), done, total);
    }
  }
  if (stats) *stats = sum;
  return 0;
}

// ---- PNG (crates/cli/src/main.rs:188 saves the frame with the image crate; this writer emits the same pixels as an
// 8-bit RGB PNG whose zlib stream uses stored blocks, which every PNG reader accepts)
static uint32_t crc32_update(uint32_t crc, const uint8_t* p, size_t n) {
  static uint32_t table[256];
  static bool ready = false;
  if (!ready) {
    for (uint32_t i = 0; i < 256; i++) {
      uint32_t c = i;
      for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      table[i] = c;
    }
    ready = true;
  }
  for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
  return crc;
}
static void put_be32(std::vector<uint8_t>& v, uint32_t x) {
  v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}
static void put_chunk(std::vector<uint8_t>& out, const char* type, const std::vector<uint8_t>& data) {
  put_be32(out, (uint32_t)data.size());
  const size_t at = out.size();
  out.insert(out.end(), type, type + 4);
  out.insert(out.end(), data.begin(), data.end());
  put_be32(out, crc32_update(0xFFFFFFFFu, out.data() + at, out.size() - at) ^ 0xFFFFFFFFu);
}
int chip_write_png(const char* path, uint32_t width, uint32_t height, const uint8_t* rgb8) {
  if (!path || !rgb8 || width == 0 || height == 0) return fail("chip_write_png: bad argument");
  if ((uint64_t)width * height * 3 + height > 0x7FFFFFFFull) return fail("chip_write_png: image too large");
  std::vector<uint8_t> raw;  // scanlines, each behind filter type 0
  raw.reserve((size_t)height * (1 + (size_t)width * 3));
  for (uint32_t y = 0; y < height; y++) {
    raw.push_back(0);
    raw.insert(raw.end(), rgb8 + (size_t)y * width * 3, rgb8 + (size_t)(y + 1) * width * 3);
  }
  std::vector<uint8_t> z = {0x78, 0x01};  // zlib header: deflate, 32 K window, no preset dictionary
  uint32_t a = 1, b = 0;                  // Adler-32 of the raw bytes
  for (size_t pos = 0; pos < raw.size();) {
    const size_t n = std::min<size_t>(65535, raw.size() - pos);
    z.push_back(pos + n == raw.size() ? 1 : 0);  // stored block, BFINAL on the last one
    z.push_back((uint8_t)n); z.push_back((uint8_t)(n >> 8));
    z.push_back((uint8_t)~n); z.push_back((uint8_t)(~n >> 8));
    z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
    for (size_t i = 0; i < n; i++) {
      a = (a + raw[pos + i]) % 65521u;
      b = (b + a) % 65521u;
    }
    pos += n;
  }
  put_be32(z, (b << 16) | a);
  std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  std::vector<uint8_t> ihdr;
  put_be32(ihdr, width);
  put_be32(ihdr, height);
  ihdr.insert(ihdr.end(), {8, 2, 0, 0, 0});  // 8 bits per channel, colour type 2 (RGB), deflate, adaptive filtering, no interlace
  put_chunk(out, "IHDR", ihdr);
  put_chunk(out, "IDAT", z);
  put_chunk(out, "IEND", {});
  FILE* f = fopen(path, "wb");
  if (!f) return fail(std::string("chip_write_png: cannot open ") + path);
  const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
  if (fclose(f) != 0 || !ok) return fail(std::string("chip_write_png: write to ") + path + " failed");
  return 0;
}

int chip_trace_rays(chip_scene* s, uint64_t seed, size_t n, const double* o, const double* d, const double* tm, chip_hit* out) {
  NEED_SCENE;
  if (!s->committed) return fail("scene not committed");
  if (n == 0) return 0;
  if (!o || !d || !out) return fail("null argument");
  if (n > 0xFFFFFFFFull) return fail("too many rays");
  static_assert(sizeof(chip_hit) == sizeof(DHitOut), "hit layout");
  DeviceState& D = s->dev;
  DeviceGuard guard;
  HIP_TRY(hipSetDevice(D.device));
  if (ensure_pool(s, D, 0, 0, 4096)) return -1;
  DevBuf<double> d_o, d_d, d_t;
  DevBuf<DHitOut> d_out;
  int rc = 0;
  hipError_t e;
  do {
    if ((e = d_o.alloc(3 * n)) != hipSuccess) break;
    if ((e = d_d.alloc(3 * n)) != hipSuccess) break;
    if ((e = d_out.alloc(n)) != hipSuccess) break;
    if ((e = hipMemcpy(d_o.p, o, 3 * n * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess) break;
    if ((e = hipMemcpy(d_d.p, d, 3 * n * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess) break;
    if (tm) {
      if ((e = d_t.alloc(n)) != hipSuccess) break;
      if ((e = hipMemcpy(d_t.p, tm, n * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess) break;
    }
    if ((e = hipMemset(D.p_err.p, 0, 4 * sizeof(uint32_t))) != hipSuccess) break;
#define CHIP_TRACE(FEAT)                                                                                                         \
  hipLaunchKernelGGL(k_trace_rays<FEAT>, dim3(grid_for(n)), dim3(CHIP_BLOCK), 0, nullptr, D.dscene, D.pools[0].P, (uint32_t)seed, \
                     (uint32_t)(seed >> 32), (uint32_t)n, d_o.p, d_d.p, tm ? d_t.p : nullptr, d_out.p)
    switch (extend_variant(scene_features(s->flat))) {
      case 0: CHIP_TRACE(0); break;
      case TF_QUADS: CHIP_TRACE(TF_QUADS); break;
      case TF_QUADS | TF_XFORM: CHIP_TRACE(TF_QUADS | TF_XFORM); break;
      case TF_QUADS | TF_XFORM | TF_NO_QUADROT: CHIP_TRACE(TF_QUADS | TF_XFORM | TF_NO_QUADROT); break;
      case TF_QUADS | TF_XFORM | TF_MEDIA: CHIP_TRACE(TF_QUADS | TF_XFORM | TF_MEDIA); break;
      case TF_QUADS | TF_XFORM | TF_MEDIA | TF_MEDIA_BOX: CHIP_TRACE(TF_QUADS | TF_XFORM | TF_MEDIA | TF_MEDIA_BOX); break;
      case TF_QUADS | TF_XFORM | TF_MEDIA | TF_MEDIA_ANY: CHIP_TRACE(TF_QUADS | TF_XFORM | TF_MEDIA | TF_MEDIA_ANY); break;
      case TF_QUADS | TF_XFORM | TF_EXT: CHIP_TRACE(TF_QUADS | TF_XFORM | TF_EXT); break;
      default: CHIP_TRACE(TF_QUADS | TF_XFORM | TF_MEDIA | TF_EXT | TF_MEDIA_ANY); break;
    }
#undef CHIP_TRACE
    if ((e = hipGetLastError()) != hipSuccess) break;
    if ((e = hipDeviceSynchronize()) != hipSuccess) break;
    if ((e = hipMemcpy(out, d_out.p, n * sizeof(DHitOut), hipMemcpyDeviceToHost)) != hipSuccess) break;
    uint32_t err[4];
    if ((e = hipMemcpy(err, D.p_err.p, sizeof(err), hipMemcpyDeviceToHost)) != hipSuccess) break;
    if (err[0] & 1u) rc = fail("traversal stack overflow (scene nesting deeper than the device stack)");
  } while (0);
  d_o.release(); d_d.release(); d_t.release(); d_out.release();
  if (e != hipSuccess) return fail(std::string("chip_trace_rays: ") + hipGetErrorString(e));
  return rc;
}

int chip_camera_rays(chip_scene* s, uint64_t seed, size_t n, const uint32_t* xyss, double* rays) {
  NEED_SCENE;
  if (!s->committed) return fail("scene not committed");
  if (n == 0) return 0;
  if (!xyss || !rays) return fail("null argument");
  if (n > 0xFFFFFFFFull) return fail("too many rays");
  DeviceGuard guard;
  HIP_TRY(hipSetDevice(s->dev.device));
  DevBuf<uint32_t> d_in;
  DevBuf<double> d_out;
  hipError_t e;
  do {
    if ((e = d_in.alloc(4 * n)) != hipSuccess) break;
    if ((e = d_out.alloc(7 * n)) != hipSuccess) break;
    if ((e = hipMemcpy(d_in.p, xyss, 4 * n * sizeof(uint32_t), hipMemcpyHostToDevice)) != hipSuccess) break;
    hipLaunchKernelGGL(k_camera_rays, dim3(grid_for(n)), dim3(CHIP_BLOCK), 0, nullptr, s->dev.dscene, (uint32_t)seed, (uint32_t)(seed >> 32),
                       (uint32_t)n, d_in.p, d_out.p);
    if ((e = hipGetLastError()) != hipSuccess) break;
    if ((e = hipDeviceSynchronize()) != hipSuccess) break;
    e = hipMemcpy(rays, d_out.p, 7 * n * sizeof(double), hipMemcpyDeviceToHost);
  } while (0);
  d_in.release(); d_out.release();
  if (e != hipSuccess) return fail(std::string("chip_camera_rays: ") + hipGetErrorString(e));
  return 0;
}

int64_t chip_dump_tree(chip_scene* s, int32_t* kinds, int32_t* ids, double* boxes, size_t cap) {
  NEED_SCENE;
  if (s->host.world < 0) return fail("scene has no world");
  size_t count = 0;
  s->host.dump(s->host.handle[s->host.world], kinds, ids, boxes, cap, count);
  return (int64_t)count;
}

// Flattens the scene on the host (no device is touched) and checks the culling trees of the ordered traversal against
// the reference-order structure they were derived from.
int64_t chip_check_culling_trees(chip_scene* s) {
  NEED_SCENE;
  Flat F;
  try {
    s->host.flatten(F);
  } catch (const std::exception& e) {
    return fail(e.what());
  }
  // what a walk meets: primitives, media (their boundaries are not entered) and sphere-only Translates, as (kind, index)
  auto is_unit = [&](nref_t r) {
    const uint32_t k = NREF_KIND(r);
    return k == NK_SPHERE || k == NK_QUAD || k == NK_DISC || k == NK_CONE || k == NK_MEDIUM ||
           (k == NK_TRANSLATE && F.translates[NREF_INDEX(r)].only_sphere != NREF_NONE);
  };
  std::vector<nref_t> ref_units, ord_units, work;
  // reference order: hierarchies by their two children, Groups by their child lists, instances by `child`
  std::vector<nref_t> rwork;
  auto ref_walk = [&](nref_t start, std::vector<nref_t>& units) -> const char* {
    rwork.assign(1, start);
    while (!rwork.empty()) {
      const nref_t r = rwork.back();
      rwork.pop_back();
      if (r == NREF_NONE) continue;
      const uint32_t k = NREF_KIND(r), i = NREF_INDEX(r);
      if (is_unit(r)) { units.push_back(r); continue; }
      if (k == NK_BVH) {
        if (i >= F.bvh.size()) return "reference node index out of range";
        rwork.push_back(F.bvh[i].left);
        if (F.bvh[i].right != F.bvh[i].left) rwork.push_back(F.bvh[i].right);  // a single member is held twice (bvh.rs:27-28)
      } else if (k == NK_GROUP) {
        for (uint32_t j = 0; j < F.groups[i].count; j++) rwork.push_back(F.gchild[F.groups[i].first + j]);
      } else if (k == NK_TRANSLATE) rwork.push_back(F.translates[i].child);
      else if (k == NK_ROTATE) rwork.push_back(F.rotates[i].child);
      else if (k == NK_SCALE) rwork.push_back(F.scales[i].child);
      else return "unexpected node kind in the reference-order structure";
    }
    return nullptr;
  };
  if (const char* why = ref_walk(F.root, ref_units)) return fail(why);
  // ordered traversal: culling nodes by their child words, instances by `ord_child` from the innermost instance of
  // their run (enter_xform); it must meet no Group that holds anything a ray can hit
  size_t n_nodes = 0;
  work.assign(1, F.ord_root);
  while (!work.empty()) {
    const nref_t r = work.back();
    work.pop_back();
    if (r == NREF_NONE) continue;
    const uint32_t k = NREF_KIND(r), i = NREF_INDEX(r);
    if (is_unit(r)) { ord_units.push_back(r); continue; }
    if (k == NK_BVH) {
      if (i >= F.wnodes.size()) return fail("culling node index out of range");
      n_nodes++;
      const DWNode& w = F.wnodes[i];
      // nodes with children lead the table (they are what the traversal kernel copies into LDS); the root of a
      // hierarchy that holds nothing may lie behind them
      if (i >= F.n_wide && (w.child[0] != NREF_NONE || w.child[1] != NREF_NONE || w.child[2] != NREF_NONE || w.child[3] != NREF_NONE))
        return fail("a culling node with children lies outside the leading part of the table");
      for (int c = 0; c < 4; c++) {
        if (w.child[c] == NREF_NONE) {
          if (!(w.lo_x[c] > w.hi_x[c] && w.lo_y[c] > w.hi_y[c] && w.lo_z[c] > w.hi_z[c])) return fail("an unused child slot does not hold an inverted box");
          continue;
        }
        if (NREF_KIND(w.child[c]) == NK_BVH) {  // the child node's own children lie inside the box this node holds for it
          if (NREF_INDEX(w.child[c]) >= F.wnodes.size()) return fail("culling node index out of range");
          const DWNode& v = F.wnodes[NREF_INDEX(w.child[c])];
          for (int d = 0; d < 4; d++) {
            if (v.child[d] == NREF_NONE) continue;
            if (v.lo_x[d] < w.lo_x[c] || v.hi_x[d] > w.hi_x[c] || v.lo_y[d] < w.lo_y[c] || v.hi_y[d] > w.hi_y[c] || v.lo_z[d] < w.lo_z[c] ||
                v.hi_z[d] > w.hi_z[c])
              return fail("a culling node's box does not contain its grandchildren's");
          }
        }
        work.push_back(w.child[c]);
      }
    } else if (k == NK_GROUP) {
      // the ordered traversal passes over a Group, so only Groups that reach nothing a ray can hit may be left
      // (empty ones, or ones holding empty Groups and hierarchies without members, bvh.rs:23-26)
      std::vector<nref_t> inside;
      if (const char* why = ref_walk(r, inside)) return fail(why);
      if (!inside.empty()) return fail("the ordered traversal would meet a Group that holds something");
    } else if (k == NK_TRANSLATE || k == NK_ROTATE || k == NK_SCALE) {
      const nref_t x = k == NK_TRANSLATE ? F.translates[i].enter_xform : (k == NK_ROTATE ? F.rotates[i].enter_xform : F.scales[i].enter_xform);
      const uint32_t xk = NREF_KIND(x), xi = NREF_INDEX(x);
      work.push_back(xk == NK_TRANSLATE ? F.translates[xi].ord_child : (xk == NK_ROTATE ? F.rotates[xi].ord_child : F.scales[xi].ord_child));
    } else {
      return fail("unexpected node kind in the culling trees");
    }
  }
  std::sort(ref_units.begin(), ref_units.end());
  std::sort(ord_units.begin(), ord_units.end());
  if (std::adjacent_find(ord_units.begin(), ord_units.end()) != ord_units.end()) return fail("the culling trees reach a primitive twice");
  if (ref_units != ord_units) return fail("the culling trees do not reach the primitives the reference-order structure reaches");
  (void)n_nodes;
  return (int64_t)ord_units.size();
}

}  // extern "C"
