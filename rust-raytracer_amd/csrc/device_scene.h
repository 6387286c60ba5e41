// Flattened scene records shared by the host flattener and the gfx950 kernels.
// All geometry is f64 (the reference computes in f64 and hit/miss decisions must
// agree with it); radiance state and the framebuffer are f32.
#pragma once
#include <stddef.h>
#include <stdint.h>

// A node reference: kind in the top 4 bits, index into that kind's table below.
typedef uint32_t nref_t;
enum : uint32_t {
  NK_BVH = 0,
  NK_SPHERE = 1,
  NK_QUAD = 2,
  NK_GROUP = 3,
  NK_TRANSLATE = 4,
  NK_ROTATE = 5,
  NK_MEDIUM = 6,
  NK_EXIT_TRANSLATE = 7,  // traversal-stack marker: leave a Translate
  NK_EXIT_ROTATE = 8,     // traversal-stack marker: leave a Rotate
  NK_GROUP_CONT = 9,      // traversal-stack marker: continue a Group at the child index stored below it
  NK_DISC = 10,
  NK_CONE = 11,           // the side wall of a cone frustum
  NK_SCALE = 12,
  NK_EXIT_SCALE = 13,     // traversal-stack marker: leave a Scale
  NK_NONE = 15,
};
#define NREF(kind, idx) ((nref_t)(((uint32_t)(kind) << 28) | (uint32_t)(idx)))
#define NREF_KIND(r) ((r) >> 28)
#define NREF_INDEX(r) ((r)&0x0FFFFFFFu)
#define NREF_NONE NREF(NK_NONE, 0x0FFFFFFF)

// The `mat` word of a primitive record: material index in bits 0..20, that material's kind (MK_*) in
// bits 21..23, so the traversal can tag a hit with its material kind without touching the material table.
#define MATK_PACK(id, kind) ((uint32_t)(id) | ((uint32_t)(kind) << 21))
#define MATK_ID(x) ((x)&0x1FFFFFu)
#define MATK_KIND(x) (((x) >> 21) & 7u)
#define MATK_MAX_MATERIALS (1u << 21)

// One BVH node: its own box (tested on entry, as bounding_volume_hierarchy.rs:59 does)
// and its two children in reference order (left is always visited first).
struct alignas(64) DBvhNode {
  double box[6];  // xmin,xmax,ymin,ymax,zmin,zmax
  nref_t left, right;
  uint32_t pad[2];
};

// Culling node of the ordered traversal: up to four children with their boxes rounded outward to f32, one 128-byte
// record (the granule HBM moves), planes stored component-wise so that the four slab tests of an axis read one
// 16-byte word per plane.  Used only for culling: a child is skipped when its box cannot contain a hit closer than the
// current one, so these boxes only have to be conservative.  The nodes of one BoundingVolumeHierarchy live in table
// slots of that hierarchy's DBvhNodes, its root in the root's slot, so a reference to a hierarchy means the same
// index in both tables.  Unused child slots hold NREF_NONE.
struct alignas(128) DWNode {
  float lo_x[4], hi_x[4], lo_y[4], hi_y[4], lo_z[4], hi_z[4];
  nref_t child[4];
  uint32_t pad[4];
};
#define CHIP_WNODE_U4 7  // 16-byte words of a DWNode the traversal reads

struct alignas(128) DSphere {
  double c[3];   // center at time 0
  double dc[3];  // center motion per unit time
  double r;
  uint32_t mat;  // MATK_PACK
  nref_t xform;  // innermost Translate/Rotate ancestor, or NREF_NONE
  int32_t leaf_id;
  uint32_t seq;  // position of this primitive in the reference's depth-first visiting order
  // When all that stands between the sphere and the world is one Translate (xform, with no transform above it): that
  // Translate's offset, so that rebuilding the hit record reads this record only (instanced-sphere scenes hold a
  // million of each, scattered over memory).
  double xoff[3];
  uint32_t xoff_only;  // xoff is valid
  uint32_t pad;
};

struct alignas(16) DQuad {
  double q[3], u[3], v[3], n[3], w[3];
  double d, area;
  uint32_t mat;
  nref_t xform;
  int32_t leaf_id;
  uint32_t seq;  // as DSphere::seq
};

struct alignas(8) DDisc {  // object/disc.rs
  double c[3], n[3], r;
  uint32_t mat;  // MATK_PACK
  nref_t xform;
  int32_t leaf_id;
  uint32_t seq;
};

struct alignas(8) DCone {  // ConeFrustumWall, object/cone.rs:107-114
  double base[3], h, r0, r1;  // r0 bottom radius, r1 top radius
  uint32_t mat;  // MATK_PACK
  nref_t xform;
  int32_t leaf_id;
  uint32_t seq;
};

struct DGroup {
  uint32_t first, count;  // children are gchild[first .. first+count)
};

struct alignas(128) DTranslate {
  double off[3];
  // With only_sphere (the sphere, when all the Translate holds is one sphere, directly or as a Group's only child; else
  // NREF_NONE): that sphere's geometry and mat word, so that testing the instance reads this one 128-byte record instead
  // of two records one after the other; and of this record its first half only, unless some sphere of the scene moves
  // (each lane of a wave reads a record of its own, so a test costs what the number of 16-byte words it asks for costs).
  double sc[3], sr;
  nref_t only_sphere;
  uint32_t sphere_mat;
  double sdc[3];
  nref_t child;
  nref_t parent_xform;             // innermost transform ancestor of this node, or NREF_NONE
  uint32_t chain_first, chain_len;  // this node's root->self transform chain in xchain[]
  // Where a traversal that enters this node continues: through a run of nested instances (each the other's only
  // content, e.g. Translate(Rotate(Box))) straight into the innermost one's space (enter_xform) and child.
  nref_t enter_xform, enter_child;
  // Where the ordered traversal continues instead of enter_child: the same node, unless that is a Group, whose
  // children it finds in a culling tree of their own.
  nref_t ord_child;
};
static_assert(sizeof(DTranslate) == 128 && offsetof(DTranslate, sdc) == 64, "a fused instance test reads the first 64 bytes");

struct alignas(8) DRotate {
  double m[9];  // row-major rotation; the inverse is its transpose (rotate.rs:52-68)
  nref_t child;
  nref_t parent_xform;
  uint32_t chain_first, chain_len;
  nref_t enter_xform, enter_child, ord_child;  // see DTranslate
  uint32_t pad;
};

struct alignas(8) DScale {
  double m[9], inv[9];  // row-major scale matrix and the reference's inverse (scale.rs:18-46)
  nref_t child;
  nref_t parent_xform;
  uint32_t chain_first, chain_len;
  nref_t enter_xform, enter_child, ord_child;  // see DTranslate
  uint32_t pad;
};

struct alignas(8) DMedium {
  double neg_inv_density;
  nref_t boundary;
  uint32_t mat;  // Isotropic material
  nref_t xform;
  int32_t leaf_id;
  uint32_t tests;  // times one world.hit of the reference tests this medium, each with a fresh free-flight sample
  // Those samples are the keyed uniforms sample_base .. sample_base + tests - 1 of the path's bounce: every record
  // (a medium node instanced in several places of the scene graph gives one record per place) owns its own keys, so
  // the media a ray meets draw independently, as they do from the reference's stream (constant_medium.rs:64).
  uint32_t sample_base;
  // A boundary that is a Group of up to six quads (a box, box_node.rs:15-74), bare or inside a run of nested instances
  // (translate / rotate / scale around nothing else), is queried in place: bq_count quads listed in gchild[bq_first ..],
  // in the space of transform bx_inner (NREF_NONE: the medium's own space).  bq_count == 0: any other boundary.
  nref_t bx_inner;
  uint32_t bq_first, bq_count;
  // The boundary holds media itself (constant_medium.rs:19-29 takes any node): each of the reference's `tests` tests of
  // this medium asks the boundary twice, and every such query makes the media inside draw afresh.  A medium inside such
  // a boundary owns tests x (2 x the enclosing medium's tests) keys: run c of them serves the enclosing medium's query c.
  uint32_t nested;
};

enum : uint32_t { MK_LAMBERTIAN = 0, MK_METAL = 1, MK_DIELECTRIC = 2, MK_DIFFUSE_LIGHT = 3, MK_ISOTROPIC = 4, MK_EMPTY = 5, MK_COUNT = 6 };
enum : uint32_t {
  MF_NEEDS_UV = 1,  // the material's texture tree contains an image texture (reads u, v)
  MF_RICH_TEX = 2,  // ... an image, noise or turbulence texture: shaded by the kernel that carries that code
};
struct alignas(8) DMaterial {
  uint32_t kind;
  int32_t tex;  // Lambertian / DiffuseLight / Isotropic texture id
  double p[4];  // Metal: albedo rgb, fuzz; Dielectric: refraction index
  uint32_t flags;
  uint32_t pad;
};

enum : uint32_t { TK_SOLID = 0, TK_CHECKER = 1, TK_IMAGE = 2, TK_NOISE = 3, TK_TURBULENCE = 4 };
struct alignas(8) DTexture {
  uint32_t kind;
  uint32_t a, b;  // checker: even, odd texture ids; image: width, height; perlin: table index, turbulence depth
  uint32_t c;     // image: byte offset into texels
  double p[3];    // solid: rgb; checker: inv_scale; perlin: scale
};

struct DPerlin {
  double rand_vec[256 * 3];
  int32_t perm_x[256], perm_y[256], perm_z[256];
};

enum : uint32_t { LK_SPHERE = 1, LK_QUAD = 2, LK_DISC = 3, LK_CONE_WALL = 4, LK_OTHER = 0 };
struct DLight {
  uint32_t kind;
  uint32_t index;  // into spheres / quads / discs / cones
  // Share of this light in the light list's pdf (group.rs:60-69): 1 / len for a member of the lights Group, the product
  // of the shares along the way for a member of a Group nested inside it.
  double weight;
};
// The lights node as a tree, for drawing a sample the way group.rs:71-78 does (one uniform pick per Group on the way down):
// node 0 is the lights Group itself; a group's children are nodes first .. first + count - 1; a leaf names its DLight.
#define CHIP_LIGHT_DEPTH 16  // Groups nested inside the lights Group, at most
struct DLightNode {
  uint32_t is_group;
  uint32_t first, count;
  uint32_t leaf;
};

// A scene of a few primitives as a list (the flat form of the traversal): every sphere and quad of the world with the
// transform it sits under, primitives of one transform in a row.  `enter` is set on the first primitive of each
// such run: the ray is taken into that space there.
#define CHIP_FLAT_MAX 32
struct alignas(16) DFlatPrim {
  nref_t ref;
  nref_t xform;  // innermost transform ancestor, or NREF_NONE
  uint32_t enter;
  uint32_t pad;
};

struct DCamera {  // Camera, camera.rs:202-231
  double center[3], pixel00[3], du[3], dv[3], disk_u[3], disk_v[3];
  double background[3];
  double pixel_samples_scale, recip_sqrt_spp, defocus_angle;
  uint32_t image_width, image_height, max_depth, sqrt_spp;
};

struct DScene {
  const DBvhNode* bvh;
  const DWNode* wnodes;
  const DSphere* spheres;
  const DQuad* quads;
  const DGroup* groups;
  const nref_t* gchild;
  const DTranslate* translates;
  const DRotate* rotates;
  const DScale* scales;
  const DDisc* discs;
  const DCone* cones;
  const DMedium* media;
  const nref_t* xchain;
  const DMaterial* materials;
  const DTexture* textures;
  const DPerlin* perlins;
  const uint8_t* texels;
  const DLight* lights;
  const DLightNode* ltree;
  const DFlatPrim* flat_list;  // n_flat entries; n_flat == 0: the scene has no flat form
  uint32_t n_flat;
  nref_t root;
  nref_t ord_root;  // the ordered traversal's starting node (differs from root when that is a Group)
  uint32_t n_lights;
  uint32_t has_lights;       // SceneData.lights is Some
  uint32_t lights_is_group;  // lights root is a Group (one member is picked per sample)
  // Some primitive reaches outside its own bounding box (a Disc whose normal is tilted but has a component above 0.9,
  // disc.rs:25-35): the reference then finds it through whichever enclosing boxes the ray happens to cross, so every
  // ray replays the reference's visiting order instead of culling by child boxes.
  uint32_t literal_only;
  // Some sphere's centre moves (sphere.rs:28-45).  When none does, every DSphere::dc and DTranslate::sdc is +0.0 and
  // the traversal of instanced spheres leaves those words unread.
  uint32_t moving_spheres;
  DCamera cam;
};
