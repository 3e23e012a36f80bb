/* oracle.h — CPU restatement of racer-tracer's render path, in plain C, f64.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped path (include/,
 * racer-tracer_amd/) may include, link or call anything in this directory;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do,
 * and only as the checker / the timed CPU baseline.
 *
 * Pinning status (see DESIGN.md "Oracle"): the Rust reference cannot be
 * built or run here (no cargo/rustc; needs a window and a key press; is
 * unseeded), and its own tests hold only four Vec3 operator checks
 * (racer-tracer/src/vec3.rs:446-503).  Those are replayed in
 * tests/test_oracle_reference_vectors.py together with everything the
 * reference itself PRODUCED, its five 600x600 SavePng screenshots
 * (the PNGs under assets/, committed as values in tests/golden/reference_assets.json):
 * sky pixels of three_balls.png / noise_and_textures.png exactly; 4x4 and 8x8
 * block means of clown.png (< 0.001), three_balls.png (< 0.004) and of
 * noise_and_textures.png outside its randomly seeded Perlin sphere (< 0.01);
 * the position of both lights and of the background in emissive.png;
 * layout, hue and the saturated light value of cornell_box.png.  What no
 * reference output covers (Noise shading values, Box/RotateY/Translate,
 * MovingSphere, Reinhard/Hable, the tile stream) is PARITY UNPINNED by the
 * reference and held by hand-derived known answers and
 * review.
 *
 * The POD scene/camera/param structs are the ones of include/rt_abi.h, so
 * oracle and device consume literally the same input bytes.
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stdint.h>
#include "../include/rt_abi.h"
#include "../include/rt_rng.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- RNG (include/rt_rng.h contract) ---- */
void orc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], int rounds, uint32_t out[4]);
/* d0 (which=0) or d1 (which=1) of the addressed block */
double orc_rng_double(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t segment,
                      uint32_t purpose, uint32_t block, int which);
/* e_0, e_1, e_2 of the addressed block (the three 42-bit draws of a random_in_unit_sphere candidate) */
void orc_rng_triple(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t segment,
                    uint32_t purpose, uint32_t block, double e[3]);

/* ---- Vec3 operators exactly as vec3.rs defines them (for the reference's
 *      own unit tests and the KATs) ---- */
void orc_vec3_add(const double a[3], const double b[3], double out[3]);
void orc_vec3_sub(const double a[3], const double b[3], double out[3]);
void orc_vec3_mul(const double a[3], const double b[3], double out[3]);
void orc_vec3_scale(const double a[3], double s, double out[3]);
void orc_vec3_div(const double a[3], double s, double out[3]);
void orc_reflect(const double v[3], const double n[3], double out[3]);
void orc_refract(const double uv[3], const double n[3], double ratio, double out[3]);
double orc_schlick(double cosine, double refraction_index);
void orc_sphere_uv(const double n[3], double *u, double *v);

/* ---- camera.rs:196-234 ---- */
void orc_camera_new(const double look_from[3], const double look_at[3],
                    const double scene_up[3], double vfov, double aperture,
                    double focus_distance, double aspect_ratio, double time_a,
                    double time_b, RtCamera *out);

/* ---- geometry ---- */
typedef struct OrcHit {
    double point[3];
    double normal[3];
    double t;
    double u, v;
    int32_t front_face;
    int32_t material;
    int32_t obj_id;
    int32_t _pad;
} OrcHit;

/* One primitive incl. its RotateY/Translate wrappers; returns 1 on hit. */
int orc_hit_primitive(const RtPrimitive *prim, const double origin[3], const double dir[3],
                      double t_min, double t_max, OrcHit *out);
/* ... with the ray's time (ray.rs:26-28), which only MovingSphere reads */
int orc_hit_primitive_time(const RtPrimitive *prim, const double origin[3], const double dir[3], double time,
                           double t_min, double t_max, OrcHit *out);
/* AABB of a primitive incl. wrappers, as geometry_creation.rs builds it
 * (RotateY's box reproduces rotate_y.rs:66-90 bit for bit, bugs included). */
void orc_primitive_aabb(const RtPrimitive *prim, double out_min[3], double out_max[3]);
int orc_aabb_hit(const double bmin[3], const double bmax[3], const double origin[3],
                 const double dir[3], double t_min, double t_max);

/* Closest hit over the whole scene.  use_bvh = 1: build the reference's
 * median-split BVH (bvh_node.rs:31-82) and walk it like Node::hit
 * (bvh_node.rs:112-132); 0: linear scan (shared_scene.rs semantics). */
typedef struct OrcScene OrcScene;
OrcScene *orc_scene_build(const RtSceneDesc *desc, int use_bvh, uint64_t seed);
void orc_scene_free(OrcScene *s);
int orc_scene_hit(const OrcScene *s, const double origin[3], const double dir[3],
                  double t_min, double t_max, OrcHit *out);
int orc_scene_hit_time(const OrcScene *s, const double origin[3], const double dir[3], double time,
                       double t_min, double t_max, OrcHit *out);

/* ---- textures / background ---- */
void orc_texture_value(const RtSceneDesc *desc, int32_t texture, double u, double v,
                       const double p[3], double out[3]);
void orc_background_color(const RtBackground *bg, const double dir[3], double out[3]);
double orc_perlin_noise(const RtPerlin *perlin, const double p[3]);
double orc_perlin_turbulence(const RtPerlin *perlin, const double p[3], int depth);

/* ---- the render path: cpu.rs:26-131 + renderer.rs:41-90 ----
 * out_rgb: width*height*3 f64, row 0 = top, = sqrt(sum/samples) like
 * cpu.rs:52.  n_threads <= 0 means "all online cores".  Honours
 * params->strip_* (unowned rows are left untouched).  *segments (may be
 * NULL) receives the number of scene.hit evaluations.  Returns RtError. */
int orc_render(const RtSceneDesc *desc, const RtCamera *camera, const RtRenderParams *params,
               int n_threads, int use_bvh, double *out_rgb, uint64_t *segments);
/* Tile grid of CpuRenderer::prepare_threads (cpu.rs:73-115):
 * out[i] = {x, y, width, height}, column-major; returns the tile count. */
int orc_tile_grid(int width, int height, int tiles_w, int tiles_h, int32_t *out, int max_tiles);
/* One sample of one pixel (for unit tests): radiance of ray_color. */
void orc_sample_radiance(const RtSceneDesc *desc, const OrcScene *scene, const RtCamera *camera,
                         const RtRenderParams *params, int px, int py, int sample,
                         double out[3], int *n_segments);
double orc_pixel_u(const RtRenderParams *params, int px, int py);
void orc_sample_radiance_u(const RtSceneDesc *desc, const OrcScene *scene, const RtCamera *camera,
                           const RtRenderParams *params, int px, int py, int sample, double u,
                           double out[3], int *n_segments);
int orc_online_cores(void);

/* ---- post: image_buffer.rs:135-170, tone_map/{aces,hable,reinhard,none}.rs, image_action/png.rs:19-31 ---- */
enum OrcToneMapKind { ORC_TM_NONE = 0, ORC_TM_REINHARD = 1, ORC_TM_HABLE = 2, ORC_TM_ACES = 3 };
typedef struct OrcToneMap {
    int32_t kind;
    int32_t _pad;
    double max_white;        /* Reinhard (default 25) */
    double hable[6];         /* shoulder, linear_strength, linear_angle, toe_strength,
                                toe_numerator, toe_denominator */
    double exposure_bias;    /* Hable (default 2) */
    double linear_white;     /* Hable (default 11.2) */
    double aces_in[9];       /* row-major */
    double aces_out[9];
} OrcToneMap;
void orc_tone_map_defaults(int kind, OrcToneMap *out);
void orc_tone_map_apply(const OrcToneMap *tm, const double *rgb_in, double *rgb_out, size_t n_pixels);
void orc_pack_rgba8(const double *rgb, size_t n_pixels, uint8_t *out_rgba);

#ifdef __cplusplus
}
#endif
#endif
