/* Plain C11 translation unit: include/caustic_hip.h must compile as C, and the layout of every struct that crosses
 * the ABI is pinned here.  The same numbers are asserted in bindings/rust/caustic-hip-sys/src/lib.rs (const
 * assertions) and checked against the ctypes structures of rust-raytracer_amd/bindings.py by
 * tests/test_host_logic.py::test_abi_layouts_agree. */
#include <stddef.h>

#include "caustic_hip.h"

#define SIZE(T, n) _Static_assert(sizeof(T) == (n), "sizeof " #T)
#define OFF(T, f, n) _Static_assert(offsetof(T, f) == (n), "offsetof " #T "." #f)

SIZE(chip_camera_desc, 144);
OFF(chip_camera_desc, vertical_fov, 0);
OFF(chip_camera_desc, aspect_ratio, 8);
OFF(chip_camera_desc, image_width, 16);
OFF(chip_camera_desc, reserved0, 20);
OFF(chip_camera_desc, look_from, 24);
OFF(chip_camera_desc, look_at, 48);
OFF(chip_camera_desc, up, 72);
OFF(chip_camera_desc, defocus_angle, 96);
OFF(chip_camera_desc, focus_distance, 104);
OFF(chip_camera_desc, samples_per_pixel, 112);
OFF(chip_camera_desc, max_depth, 116);
OFF(chip_camera_desc, background, 120);

SIZE(chip_render_opts, 40);
OFF(chip_render_opts, seed, 0);
OFF(chip_render_opts, stratum_offset, 8);
OFF(chip_render_opts, stratum_stride, 12);
OFF(chip_render_opts, xmin, 16);
OFF(chip_render_opts, xmax, 20);
OFF(chip_render_opts, ymin, 24);
OFF(chip_render_opts, ymax, 28);
OFF(chip_render_opts, strata_per_batch, 32);
OFF(chip_render_opts, flags, 36);

SIZE(chip_hit, 88);
OFF(chip_hit, t, 0);
OFF(chip_hit, pt, 8);
OFF(chip_hit, normal, 32);
OFF(chip_hit, u, 56);
OFF(chip_hit, v, 64);
OFF(chip_hit, hit, 72);
OFF(chip_hit, leaf_id, 76);
OFF(chip_hit, front_face, 80);
OFF(chip_hit, material_kind, 84);

SIZE(chip_stats, 80);
OFF(chip_stats, paths, 0);
OFF(chip_stats, segments, 8);
OFF(chip_stats, bvh_visits, 16);
OFF(chip_stats, prim_tests, 24);
OFF(chip_stats, ms_total, 32);
OFF(chip_stats, ms_trace, 40);
OFF(chip_stats, ms_shade, 48);
OFF(chip_stats, ms_other, 56);
OFF(chip_stats, trace_launches, 64);
OFF(chip_stats, batches, 72);

SIZE(chip_flat_node, 96);
OFF(chip_flat_node, kind, 0);
OFF(chip_flat_node, mat, 4);
OFF(chip_flat_node, first, 8);
OFF(chip_flat_node, count, 12);
OFF(chip_flat_node, flags, 16);
OFF(chip_flat_node, reserved, 20);
OFF(chip_flat_node, v, 24);

SIZE(chip_flat_scene, 32);
OFF(chip_flat_scene, nodes, 0);
OFF(chip_flat_scene, n_nodes, 8);
OFF(chip_flat_scene, members, 16);
OFF(chip_flat_scene, n_members, 24);

_Static_assert(CHIP_RENDER_TIME_KERNELS == 1 && CHIP_RENDER_EXACT_SHADING == 2 && CHIP_RENDER_REFERENCE_ORDER == 4 && CHIP_RENDER_QUANTISE_255 == 8, "render flags");
_Static_assert(CHIP_NODE_BVH == 0 && CHIP_NODE_SPHERE == 1 && CHIP_NODE_QUAD == 2 && CHIP_NODE_GROUP == 3 && CHIP_NODE_TRANSLATE == 4 && CHIP_NODE_ROTATE == 5 &&
                   CHIP_NODE_MEDIUM == 6 && CHIP_NODE_BOX == 7 && CHIP_NODE_DISC == 8 && CHIP_NODE_CONE == 9 && CHIP_NODE_CONE_WALL == 10 && CHIP_NODE_SCALE == 11,
               "node kinds");
_Static_assert(CHIP_FLAT_MOVING == 1 && CHIP_FLAT_MEDIUM_TEX == 2, "flat flags");

int abi_check_translation_unit_is_not_empty;
