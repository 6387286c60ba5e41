/* A native caller of libcaustic_hip.so in plain C, with the shape of the reference's command-line program
 * (crates/cli/src/main.rs:29-199): build the scene through the per-node constructors, render the frame block by
 * block into an image buffer, save it as a PNG.  The scene is the RTIOW final scene of
 * crates/cli/src/scene/random_spheres.rs:11-100 with its random draws taken from a seeded splitmix64 stream in the
 * source's draw order (the reference's own stream is unseeded), i.e. the same constructor calls with the same
 * arguments as the package's Python scene builder makes, so both callers hold the same scene.
 *
 * usage: cli_caller WIDTH SPP DEPTH RENDER_SEED PASSES OUT.png
 * Prints one line "paths=... segments=... WxH" on success; exit status 1 with the library's message otherwise.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "caustic_hip.h"

#define BLOCK_SIZE 10 /* crates/cli/src/main.rs:27 */

static uint64_t rng_state;
static void rng_seed(uint64_t seed) { rng_state = seed * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull; }
static double rng_rand(void) { /* uniform in [0, 1), 53 bits (the contract of crates/core/src/random.rs:3-7) */
  rng_state += 0x9E3779B97F4A7C15ull;
  uint64_t z = rng_state;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}
static double rng_interval(double lo, double hi) { return lo + (hi - lo) * rng_rand(); }

static int die(const char* what) {
  fprintf(stderr, "cli_caller: %s: %s\n", what, chip_last_error());
  return 1;
}
#define TRY(expr) do { if ((expr) < 0) return die(#expr); } while (0)

struct frame {
  uint32_t w, h;
  uint8_t* rgb;
  uint32_t blocks, last_done, total;
};
/* crates/cli/src/main.rs:165-182: a finished block's pixels go into the image */
static void on_block(void* user, uint32_t xmin, uint32_t xmax, uint32_t ymin, uint32_t ymax, const uint8_t* rgb8, uint32_t strata_done,
                     uint32_t strata_total) {
  struct frame* f = (struct frame*)user;
  for (uint32_t y = ymin; y < ymax; y++)
    memcpy(f->rgb + ((size_t)y * f->w + xmin) * 3, rgb8 + (size_t)(y - ymin) * (xmax - xmin) * 3, (size_t)(xmax - xmin) * 3);
  f->blocks++;
  f->last_done = strata_done;
  f->total = strata_total;
}

int main(int argc, char** argv) {
  if (argc != 7) {
    fprintf(stderr, "usage: %s WIDTH SPP DEPTH RENDER_SEED PASSES OUT.png\n", argv[0]);
    return 2;
  }
  const uint32_t width = (uint32_t)atoi(argv[1]), spp = (uint32_t)atoi(argv[2]), depth = (uint32_t)atoi(argv[3]);
  const uint64_t render_seed = strtoull(argv[4], NULL, 10);
  const uint32_t passes = (uint32_t)atoi(argv[5]);
  const char* out_path = argv[6];

  chip_scene* s = chip_scene_create();
  if (!s) return die("chip_scene_create");
  rng_seed(1);
  static int32_t world[512];
  int n = 0, mat, node;
  const double ground_c[3] = {0.0, -1000.0, 0.0};
  TRY(mat = chip_lambertian_color(s, 0.5, 0.5, 0.5));
  TRY(world[n++] = chip_sphere(s, ground_c, 1000.0, mat));
  for (int a = -11; a < 11; a++)
    for (int b = -11; b < 11; b++) {
      const double choose_mat = rng_rand();
      const double cx = (double)a + 0.9 * rng_rand();
      const double cz = (double)b + 0.9 * rng_rand();
      const double center[3] = {cx, 0.2, cz};
      const double dx = cx - 4.0, dy = 0.2 - 0.2, dz = cz - 0.0;
      if (sqrt(dx * dx + dy * dy + dz * dz) > 0.9) {
        if (choose_mat < 0.8) {
          const double c1r = rng_rand(), c1g = rng_rand(), c1b = rng_rand();
          const double c2r = rng_rand(), c2g = rng_rand(), c2b = rng_rand();
          TRY(mat = chip_lambertian_color(s, c1r * c2r, c1g * c2g, c1b * c2b));
          TRY(node = chip_sphere(s, center, 0.2, mat));
          const double dir[3] = {0.0, rng_interval(0.0, 0.5), 0.0};
          TRY(chip_sphere_set_direction(s, node, dir));
          world[n++] = node;
        } else if (choose_mat < 0.95) {
          const double r = rng_interval(0.5, 1.0), g = rng_interval(0.5, 1.0), bl = rng_interval(0.5, 1.0);
          const double fuzz = rng_interval(0.0, 0.5);
          TRY(mat = chip_metal(s, r, g, bl, fuzz));
          TRY(world[n++] = chip_sphere(s, center, 0.2, mat));
        } else {
          TRY(mat = chip_dielectric(s, 1.5));
          TRY(world[n++] = chip_sphere(s, center, 0.2, mat));
        }
      }
    }
  const double c_glass[3] = {0.0, 1.0, 0.0}, c_diffuse[3] = {-4.0, 1.0, 0.0}, c_metal[3] = {4.0, 1.0, 0.0};
  TRY(mat = chip_dielectric(s, 1.5));
  TRY(world[n++] = chip_sphere(s, c_glass, 1.0, mat));
  TRY(mat = chip_lambertian_color(s, 0.4, 0.2, 0.1));
  TRY(world[n++] = chip_sphere(s, c_diffuse, 1.0, mat));
  TRY(mat = chip_metal(s, 0.7, 0.6, 0.5, 0.0));
  TRY(world[n++] = chip_sphere(s, c_metal, 1.0, mat));
  TRY(node = chip_bvh(s, world, n));
  TRY(chip_scene_set_world(s, node));
  TRY(chip_scene_set_lights(s, -1));
  chip_camera_desc cam;
  memset(&cam, 0, sizeof cam);
  cam.vertical_fov = 20.0;
  cam.aspect_ratio = 16.0 / 9.0;
  cam.image_width = width;
  cam.look_from[0] = 13.0; cam.look_from[1] = 2.0; cam.look_from[2] = 3.0;
  cam.up[1] = 1.0;
  cam.defocus_angle = 0.6;
  cam.focus_distance = 10.0;
  cam.samples_per_pixel = spp;
  cam.max_depth = depth;
  cam.background[0] = 0.7; cam.background[1] = 0.8; cam.background[2] = 1.0;
  TRY(chip_scene_set_camera(s, &cam));
  TRY(chip_commit(s));

  struct frame f;
  memset(&f, 0, sizeof f);
  TRY(chip_image_size(s, &f.w, &f.h));
  f.rgb = (uint8_t*)calloc((size_t)f.w * f.h, 3);
  if (!f.rgb) return 1;
  chip_render_opts o;
  memset(&o, 0, sizeof o);
  o.seed = render_seed;
  o.stratum_stride = 1;
  chip_stats st;
  memset(&st, 0, sizeof st);
  TRY(chip_render_frame_blocks(s, &o, BLOCK_SIZE, passes, on_block, &f, &st));
  if (f.last_done != f.total || f.blocks == 0) {
    fprintf(stderr, "cli_caller: the last pass did not cover every stratum\n");
    return 1;
  }
  TRY(chip_write_png(out_path, f.w, f.h, f.rgb)); /* crates/cli/src/main.rs:188 */
  printf("paths=%llu segments=%llu %ux%u blocks=%u\n", (unsigned long long)st.paths, (unsigned long long)st.segments, f.w, f.h, f.blocks);
  free(f.rgb);
  chip_scene_destroy(s);
  return 0;
}
