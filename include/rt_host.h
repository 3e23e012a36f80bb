/* rt_host.h — C ABI of the host layer around the render path: what
 * racer-tracer's `run()` does before and after it calls the renderer
 * (racer-tracer/src/main.rs:71-129 and :148-158).
 *
 *   before: read config.yml + CLI overrides (config.rs:30-67), load the scene
 *           (scene/yml.rs), merge scene camera over config camera
 *           (camera.rs:403-435), build the camera (camera.rs:196-234), pick
 *           the tone map (main.rs:84-86), and describe the scene as the PODs
 *           of rt_abi.h;
 *   after:  tone-map the renderer's tiles (image_buffer.rs:135-170), pack
 *           them to RGBA8 and save `<SHA-256>.png` (image_action/png.rs).
 *
 * Same library (libracer_tracer_amd.so), plain C types only.
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include "rt_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct RthSession RthSession; /* opaque: Config + SceneLoadData + flattened scene + camera */

/* image_action values (config.rs:95-100) */
enum RthImageAction { RTH_IMAGE_ACTION_NONE = 0, RTH_IMAGE_ACTION_SAVE_PNG = 1 };
/* tone map kinds (config.rs:136-167) */
enum RthToneMap { RTH_TONE_MAP_NONE = 0, RTH_TONE_MAP_REINHARD = 1, RTH_TONE_MAP_HABLE = 2, RTH_TONE_MAP_ACES = 3 };

/* `Config::try_from(Args)` + `loader.load()` + camera/tone-map wiring
 * (main.rs:73-110).  scene_override = the value of -s/--scene (a .yml path,
 * "sandbox" or "random"; NULL = the config's loader); image_action_override =
 * the value of --image-action ("png"/"none"; NULL = the config's).
 * Returns the reference's error codes (error.rs:71-97). */
int rth_session_open(const char *config_path, const char *scene_override,
                     const char *image_action_override, uint64_t seed, RthSession **out);
void rth_session_close(RthSession *s);

const RtSceneDesc *rth_session_scene(const RthSession *s);  /* valid until close */
const RtCamera *rth_session_camera(const RthSession *s);
/* screen + `render:` block (+ seed) as RtRenderParams; preview != 0 gives the
 * `preview:` block instead (config.rs:180-186). */
int rth_session_params(const RthSession *s, int preview, RtRenderParams *out);
int rth_session_image_action(const RthSession *s);
int rth_session_tone_map_kind(const RthSession *s);
const char *rth_session_image_output_dir(const RthSession *s); /* NULL when unset */
/* The session's tone map (scene's, else config's: main.rs:84-86) with its resolved
 * parameters, in the form rt_post_rgba8_device takes. */
int rth_session_tone_map(const RthSession *s, RtToneMap *out);

/* ScreenBuffer::update's per-pixel tone map (image_buffer.rs:150) over
 * n_pixels RGB triples; in and out may alias. */
int rth_tone_map(const RthSession *s, const double *rgb_in, double *rgb_out, size_t n_pixels);
/* png.rs:21-31 packing: (c * 255.0) as u32, RGBA big-endian bytes, alpha 255. */
int rth_pack_rgba8(const double *rgb, size_t n_pixels, uint8_t *out_rgba);
/* SavePng::action (png.rs:13-61) on an already tone-mapped frame: writes
 * `<dir>/<UPPER-HEX SHA-256 of the RGBA bytes>.png`; the path is copied to
 * out_path (capacity cap).  dir = NULL uses the session's image_output_dir;
 * when that is unset the call does nothing and returns RT_OK with "" (png.rs:56-59). */
int rth_save_png(const RthSession *s, const double *rgb, int width, int height, const char *dir,
                 char *out_path, size_t cap);

/* Stand-alone pieces, exposed for tests and for callers that bring their own
 * loader: Camera::new (camera.rs:196-234) with aspect = width / height. */
int rth_camera_new(const double look_from[3], const double look_at[3], double vfov, double aperture,
                   double focus_distance, int width, int height, RtCamera *out);
/* Decode an image file the way TextureImage::try_new does (RGBA8, row 0 =
 * top).  *rgba is malloc'ed; free with rth_free. */
int rth_decode_image(const char *path, uint8_t **rgba, int *width, int *height);
void rth_free(void *p);
/* Upper-hex SHA-256 (65 bytes incl. NUL) */
int rth_sha256_hex(const uint8_t *data, size_t len, char out[65]);

const char *rth_last_error_message(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_HOST_H */
