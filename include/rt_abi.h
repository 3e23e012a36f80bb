/* rt_abi.h — C ABI of the MI355X render path (libracer_tracer_amd.so).
 *
 * This is the drop-in boundary for racer-tracer's renderer: everything the
 * reference does between `Renderer::render` being called and the
 * `ImageBufferEvent::BufferUpdate` tiles coming out
 * (racer-tracer/src/renderer.rs:101-107, renderer/cpu.rs:26-131) happens
 * behind the functions declared here.  The signatures use plain C types
 * only (pointers, sizes, PODs) so that a Rust `impl Renderer` can bind them
 * with `extern "C"` unchanged; INTEGRATION.md shows that binding.
 *
 * Every struct is a flattened ("described") form of a reference trait
 * object; the comment on each one names the Rust type it replaces.
 * All floating point is f64 like the reference (vec3.rs:13-16).
 */
#ifndef RT_ABI_H
#define RT_ABI_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 5

/* ------------------------------------------------------------------ errors
 * 0 = Ok.  1..22 are exactly the reference's exit codes
 * (racer-tracer/src/error.rs:71-97).  Codes >= 100 are new, GPU-side. */
enum RtError {
    RT_OK = 0,
    RT_ERR_FAILED_TO_CREATE_WINDOW = 1,
    RT_ERR_FAILED_TO_UPDATE_WINDOW = 2,
    RT_ERR_CONFIGURATION = 3,
    RT_ERR_UNKNOWN_MATERIAL = 4,
    RT_ERR_FAILED_TO_ACQUIRE_LOCK = 5,
    RT_ERR_EXIT_EVENT = 6,
    RT_ERR_CANCEL_EVENT = 7,
    RT_ERR_IMAGE_SAVE = 8,
    RT_ERR_SCENE_LOAD = 9,
    RT_ERR_ARGUMENT_PARSING = 10,
    RT_ERR_KEY = 11,
    RT_ERR_CREATE_LOG = 12,
    RT_ERR_RECEIVE = 13,
    RT_ERR_SEND = 14,
    RT_ERR_ACTION_PROTOCOL = 15,
    RT_ERR_BUS_WRITE = 16,
    RT_ERR_BUS_READ = 17,
    RT_ERR_BUS_UPDATE = 18,
    RT_ERR_BUS_TIMEOUT = 19,
    RT_ERR_NO_OBJECT_WITH_ID = 20,
    RT_ERR_FAILED_TO_OPEN_IMAGE = 21,
    RT_ERR_FAILED_TO_PARSE = 22,
    /* new */
    RT_ERR_NO_DEVICE = 100,      /* no HIP device / runtime unusable */
    RT_ERR_HIP = 101,            /* a HIP call failed; see rt_last_error_message */
    RT_ERR_INVALID_ARGUMENT = 102,
    RT_ERR_UNSUPPORTED = 103,    /* scene uses something the device path lacks */
    RT_ERR_OUT_OF_MEMORY = 104
};

/* ---------------------------------------------------------------- textures
 * Flattened `dyn Texture` (texture.rs:8-10). */
enum RtTextureKind {
    RT_TEX_SOLID_COLOR = 0, /* texture/solid_color.rs:24-28  : color            */
    RT_TEX_CHECKERED = 1,   /* texture/checkered.rs:32-42    : tex_even, tex_odd */
    RT_TEX_IMAGE = 2,       /* texture/image.rs:28-51        : image            */
    RT_TEX_NOISE = 3        /* texture/noise.rs:26-33        : color, scale, depth, perlin */
};

typedef struct RtTexture {
    int32_t kind;
    int32_t tex_even; /* Checkered: index of `texture_a` (used when sines >= 0) */
    int32_t tex_odd;  /* Checkered: index of `texture_b` (used when sines <  0) */
    int32_t image;    /* Image: index into RtSceneDesc.images                   */
    int32_t perlin;   /* Noise: index into RtSceneDesc.perlins                  */
    int32_t depth;    /* Noise: turbulence octaves                              */
    double color[3];  /* SolidColor / Noise colour                              */
    double scale;     /* Noise scale                                            */
} RtTexture;

/* Decoded image of a TextureImage: RGBA8, row-major, row 0 = top
 * (what `image::open(..).into_rgba8()` yields, texture/image.rs:18-24). */
typedef struct RtImage {
    const uint8_t *rgba;
    int32_t width;
    int32_t height;
} RtImage;

/* One `Perlin` (texture/noise.rs:36-55): 256 unit gradient vectors and the
 * three permutation tables (identity in the reference, noise.rs:121-130). */
typedef struct RtPerlin {
    double ranvec[256][3];
    int32_t perm_x[256];
    int32_t perm_y[256];
    int32_t perm_z[256];
} RtPerlin;

/* --------------------------------------------------------------- materials
 * Flattened `dyn Material` (material.rs:10-15). */
enum RtMaterialKind {
    RT_MAT_LAMBERTIAN = 0,   /* material/lambertian.rs:26-38   */
    RT_MAT_METAL = 1,        /* material/metal.rs:26-43        */
    RT_MAT_DIELECTRIC = 2,   /* material/dialectric.rs:25-55   */
    RT_MAT_DIFFUSE_LIGHT = 3 /* material/diffuse_light.rs:25-37 */
};

typedef struct RtMaterial {
    int32_t kind;
    int32_t texture;         /* index into textures (unused for Dielectric) */
    double fuzz;             /* Metal                                       */
    double refraction_index; /* Dielectric                                  */
} RtMaterial;

/* --------------------------------------------------------------- primitives
 * Flattened `SceneObject` + `dyn HittableSceneObject` (scene.rs:24-49).
 * The YAML loader can wrap an object in at most one RotateY and then at
 * most one Translate (scene/yml.rs:401-439), so the instance chain is two
 * optional fields instead of a tree. */
enum RtPrimitiveKind {
    RT_PRIM_SPHERE = 0,  /* geometry/sphere.rs  : p = cx, cy, cz, radius        */
    RT_PRIM_XY_RECT = 1, /* geometry/xy_rect.rs : p = x0, x1, y0, y1, k         */
    RT_PRIM_XZ_RECT = 2, /* geometry/xz_rect.rs : p = x0, x1, z0, z1, k         */
    RT_PRIM_YZ_RECT = 3, /* geometry/yz_rect.rs : p = y0, y1, z0, z1, k         */
    RT_PRIM_BOX = 4,     /* geometry/box.rs     : p = min xyz, max xyz          */
    RT_PRIM_MOVING_SPHERE = 5 /* geometry/moving_sphere.rs : p = cx, cy, cz (at time_a), radius;
                                 center_b = centre at time_b; linear in ray.time()      */
};

enum RtPrimitiveFlags {
    RT_PRIM_HAS_ROTATE_Y = 1, /* geometry/rotate_y.rs, applied first (inner) */
    RT_PRIM_HAS_TRANSLATE = 2 /* geometry/translate.rs, applied second (outer) */
};

typedef struct RtPrimitive {
    int32_t kind;
    int32_t material;
    int32_t flags;
    int32_t obj_id;      /* scene.rs:51,60 (informational; not read by the path) */
    double p[6];
    double rot_sin;      /* sin/cos of radians(degrees), rotate_y.rs:19-28 */
    double rot_cos;
    double translate[3]; /* translate.rs:13-16 */
    double center_b[3];  /* MovingSphere.pos_b (moving_sphere.rs:20-25) */
    double time_a;       /* MovingSphere.time_a / time_b: 0 and 1 in the reference */
    double time_b;       /*   (scene/random.rs:55)                                 */
} RtPrimitive;

/* -------------------------------------------------------------- background
 * Flattened `dyn BackgroundColor` (background_color.rs:3-5). */
enum RtBackgroundKind {
    RT_BG_SKY = 0,  /* background_color.rs:27-33: (1-t)*top + t*bottom */
    RT_BG_SOLID = 1 /* background_color.rs:45-48: `top` holds the colour */
};

typedef struct RtBackground {
    int32_t kind;
    int32_t _pad;
    double top[3];
    double bottom[3];
} RtBackground;

/* ------------------------------------------------------------------- scene
 * `SceneLoadData.objects` + `.background` (scene.rs:109-114) in POD form.
 * The library copies everything it needs in rt_scene_create; the caller
 * may free the arrays afterwards. */
typedef struct RtSceneDesc {
    const RtPrimitive *primitives;
    int32_t n_primitives;
    const RtMaterial *materials;
    int32_t n_materials;
    const RtTexture *textures;
    int32_t n_textures;
    const RtImage *images;
    int32_t n_images;
    const RtPerlin *perlins;
    int32_t n_perlins;
    RtBackground background;
} RtSceneDesc;

/* ------------------------------------------------------------------ camera
 * The 14 fields of `CameraSharedData` (camera.rs:56-72), passed per call
 * because the camera changes between renders (main.rs:178-189). */
typedef struct RtCamera {
    double origin[3];
    double upper_left_corner[3];
    double forward[3];
    double right[3];
    double up[3];
    double horizontal[3];
    double vertical[3];
    double vfov;
    double viewport_width;
    double viewport_height;
    double lens_radius;
    double focus_distance;
    double time_a;
    double time_b;
} RtCamera;

/* ------------------------------------------------------------------ params
 * `Image` (image.rs:3-8) + `RenderConfig` (config.rs:75-82) + the seed the
 * reference does not have (util.rs:9-23 is OS-seeded). */
typedef struct RtRenderParams {
    int32_t width;      /* screen.width  (>= 2: cpu.rs:36 divides by width - 1;   */
    int32_t height;     /* screen.height  >= 2: cpu.rs:40 divides by height - 1)  */
    int32_t samples;    /* render.samples   */
    int32_t max_depth;  /* render.max_depth */
    int32_t tiles_w;    /* render.num_threads_width  (tile grid of rt_render) */
    int32_t tiles_h;    /* render.num_threads_height */
    uint64_t seed;      /* key of the counter-based RNG (rt_rng.h) */
    /* Row ownership for multi-GPU renders: this call renders the image rows
     * r with (r / strip_rows) % strip_count == strip_index and leaves the
     * others untouched.  strip_count <= 1 means "all rows". */
    int32_t strip_rows;
    int32_t strip_count;
    int32_t strip_index;
    /* `scale` of the preview renderer (config.rs:81, renderer/cpu_scaled.rs): 0 or 1
     * = every pixel (CpuRenderer).  scale > 1 = CpuRendererScaled: only the top-left
     * pixel of each scale_w x scale_h block is traced and the block is filled with it,
     * where scale_w = largest divisor <= scale of (width / tiles_w) and likewise for
     * rows (cpu_scaled.rs:18-41); what is left over at the right/bottom edge stays
     * (0,0,0) (cpu_scaled.rs:50-52).  Not combinable with strips. */
    int32_t scale;
} RtRenderParams;

/* ---------------------------------------------------------------- tone map
 * Flattened `dyn ToneMap` (tone_map.rs:14-16) with the parameters its factory
 * resolves (tone_map.rs:18-66); matrices are row-major. */
enum RtToneMapKind {
    RT_TM_NONE = 0,     /* tone_map/none.rs     */
    RT_TM_REINHARD = 1, /* tone_map/reinhard.rs : max_white                       */
    RT_TM_HABLE = 2,    /* tone_map/hable.rs    : hable[6], exposure_bias, linear_white */
    RT_TM_ACES = 3      /* tone_map/aces.rs     : aces_in, aces_out               */
};

typedef struct RtToneMap {
    int32_t kind;
    int32_t _pad;
    double max_white;
    double hable[6]; /* shoulder_strength, linear_strength, linear_angle, toe_strength, toe_numerator, toe_denominator */
    double exposure_bias;
    double linear_white;
    double aces_in[9];
    double aces_out[9];
} RtToneMap;

typedef struct RtScene RtScene; /* opaque; owned by the library */

/* ------------------------------------------------------------ scene options
 * Implementation choices rt_scene_create makes by itself, exposed so that a
 * caller (the parity tests above all) can pin them.  Every combination renders
 * the same picture: closest-hit semantics are those of bvh_node.rs:112-132 in
 * all of them and the pooled and v1 kernels implement the same contract.  The
 * library reads NO environment variable (developer builds with
 * -DRT_DEVELOPER_KNOBS aside). */
enum RtClosestHit {
    RT_HIT_AUTO = 0,   /* linear loop up to 48 primitives, BVH above */
    RT_HIT_LINEAR = 1, /* brute force over the primitive table (it lives in LDS: a few hundred primitives at most) */
    RT_HIT_BVH = 2     /* skip-link BVH */
};
enum RtTraceKernel {
    RT_KERNEL_POOL = 0, /* k_trace_pool_f64: persistent waves over a pool of paths (default) */
    RT_KERNEL_V1 = 1    /* k_trace_f64: lane = pixel; the simple second implementation */
};
/* Arithmetic of the trace kernels.  The reference divides (vec3.rs:79-85 unit_vector, sphere.rs:52,
 * xy_rect.rs:31) and multiplies by 1/x (vec3.rs:279-301) in IEEE f64 without FMA contraction.
 *   RT_ARITH_FAST (default): one reciprocal / reciprocal square root per ray from the hardware seed, refined to
 *     <= 1 ulp, FMA contraction on.  Against the reference's arithmetic (the oracle) the shipped scenes agree to
 *     ~1e-13 per channel; a scene that amplifies rounding (thousands of small mirrors: a bounce multiplies a
 *     direction error by distance / radius) can flip the odd hit, i.e. a few pixels in ten thousand beyond 1e-3.
 *     Scenes with boxes, wrappers, moving spheres or more than 48 primitives add their per-pixel sums in 64-bit fixed
 *     point (quantum: the scene's largest emission / background component x 2^-52 per sample — a pixel whose radiance
 *     is of that order, black for every purpose, comes out up to 4e-8 from the f64 sum's value), which makes the frame
 *     independent of how the GPU schedules the work; that needs a bound on a sample's radiance, so a scene with a
 *     colour above 1 (or below 0, or not finite) on a scattering material is rendered as RT_ARITH_REFERENCE.
 *   RT_ARITH_REFERENCE: the reference's own operations (IEEE divisions, sqrt + three divisions, no contraction);
 *     holds the 1e-3 per-channel tolerance on such scenes too; ~25 % slower.  Same kernels, same draws, same
 *     closest-hit rule: only the last bits of the arithmetic differ. */
enum RtArithmetic {
    RT_ARITH_FAST = 0,
    RT_ARITH_REFERENCE = 1
};
/* How rt_render_frame_multi_device collects this scene's strips in the output buffer.
 *   RT_GATHER_AUTO (default): a scene on the output's device renders straight into the output; every other one
 *     renders into its own frame and sends its strips by peer copies.
 *   RT_GATHER_STAGED: this scene always takes the second road, also on the output's device — a switch for tests, so
 *     that ONE card executes the staging + strided-copy code that several cards run.  Same frame either way. */
enum RtGather {
    RT_GATHER_AUTO = 0,
    RT_GATHER_STAGED = 1
};
typedef struct RtSceneOptions {
    int32_t closest_hit; /* RtClosestHit  */
    int32_t kernel;      /* RtTraceKernel */
    int32_t arithmetic;  /* RtArithmetic  */
    int32_t gather;      /* RtGather      */
    int32_t _reserved[4]; /* must be 0 */
} RtSceneOptions;

/* Statistics of the last render on a scene (path segments = ray_color
 * levels actually evaluated; feeds the roofline figure of bench.py). */
typedef struct RtRenderStats {
    uint64_t samples;        /* primary rays traced               */
    uint64_t segments;       /* sum over samples of path length   */
    double kernel_ms;        /* HIP-event time of the trace kernel(s), this call */
    double resolve_ms;       /* HIP-event time of the resolve kernel, this call  */
    int32_t kernel_launches; /* trace-kernel launches in this call */
    int32_t _pad;
} RtRenderStats;

/* Tile callback = one `ImageBufferEvent::BufferUpdate{rgb,r,c,width,height}`
 * (image_buffer.rs:62-71): `rgb` is width*height*3 f64, row-major, already
 * divided by samples and sqrt'd, NOT tone-mapped (cpu.rs:52,64-70).
 * `rgb` is only valid during the call. */
typedef void (*RtTileCallback)(void *user, const double *rgb, int32_t r, int32_t c,
                               int32_t width, int32_t height);

/* --------------------------------------------------------------- functions
 * Threading (renderer.rs:101: `trait Renderer: Send + Sync`, called from one rayon worker at a
 * time, main.rs:163-199): the library keeps no global state besides the thread-local error text.
 * Different RtScene objects may be used from different threads concurrently; calls on the SAME
 * RtScene must be serialised by the caller (the reference's render task does: one render at a
 * time, scene rebuilt between renders).  Callbacks run on the calling thread. */

/* ABI version of the loaded library (== RT_ABI_VERSION of its build). */
int rt_abi_version(void);

/* Number of usable HIP devices (0 when there is none or the runtime is
 * unusable).  Never fails. */
int rt_device_count(void);

/* Upload a scene to a device.  Replaces handing `&dyn Hittable` +
 * `&dyn BackgroundColor` to the renderer (renderer.rs:92-99); re-doable
 * because the reference rebuilds its BVH between renders (main.rs:178). */
int rt_scene_create(const RtSceneDesc *desc, int device, RtScene **out);
/* Same with explicit options (NULL = defaults = rt_scene_create). */
int rt_scene_create_ex(const RtSceneDesc *desc, int device, const RtSceneOptions *options, RtScene **out);
void rt_scene_destroy(RtScene *scene);
/* rt_scene_destroy keeps what a render allocated (slices, frames, pinned host memory, counters, streams, events) in a
 * per-device cache of at most two sets, and the next rt_scene_create on that device takes a set over: the reference
 * rebuilds its scene on every object event (main.rs:174-189), and a rebuilt scene's first render then costs what any
 * render costs (measured: 4.7 -> 1.6 ms for a 1080p preview, profiles/r04_scene_create.txt).  This gives the cached
 * memory back (e.g. before the process goes on to something else); scenes that are alive are not touched. */
void rt_release_cached_buffers(void);

/* Replaces `CpuRenderer::render` (renderer/cpu.rs:118-131) with the whole
 * frame as ONE BufferUpdate (legal: renderer/image.rs:56-62 does the same).
 * out_rgb: caller-owned HOST memory, width*height*3 f64, row-major, row 0 =
 * top; gamma-encoded (sqrt(sum/samples)), not tone-mapped, not clamped.
 * The launch writes finished pixels into pinned memory itself, band of rows by
 * band of rows, and the call copies a finished band into out_rgb while the GPU
 * renders the next: the frame is in out_rgb a fraction of a millisecond after
 * the last wave ends (no resolve launch, no device-to-host copy afterwards). */
int rt_render_frame(RtScene *scene, const RtCamera *camera, const RtRenderParams *params,
                    double *out_rgb);

/* Same, but out_rgb_device is DEVICE memory of the scene's device and the
 * work is enqueued on `hip_stream` (a hipStream_t; NULL = default stream)
 * without synchronising: the caller (or a collective on the same stream)
 * orders against it.  Rows not owned under params->strip_* are not written. */
int rt_render_frame_device(RtScene *scene, const RtCamera *camera,
                           const RtRenderParams *params, double *out_rgb_device,
                           void *hip_stream);

/* Cancel hook of rt_render_ex / rt_render_multi = `do_cancel` (renderer.rs:25-30):
 * returns non-zero once the render should stop.  The reference's
 * `RenderData.cancel_event` is an `Option<&synchronoise::SignalEvent>` polled with
 * `wait_timeout(Duration::ZERO)`; a binding passes a function that does exactly
 * that.  Called on the calling thread only, between
 * launches, before every tile callback and every few tens of microseconds while
 * the call waits for the GPU. */
typedef int (*RtCancelCallback)(void *cancel_user);

/* Replaces `CpuRenderer::render` with the reference's own tile stream:
 * tiles_w x tiles_h tiles in column-major order with the last row/column
 * absorbing remainders (cpu.rs:73-115), one callback per tile — tiles_w *
 * tiles_h of them, an empty tile (more tile rows than image rows) included,
 * with width or height 0 — on the calling thread.  Delivery is progressive: ONE launch renders the frame tile
 * column by tile column and writes finished pixels straight into pinned host
 * memory; the callbacks of a finished column run while the GPU works on the
 * next, so tiles arrive during the render as the reference's do (cpu.rs:64-70);
 * their pixels are bit-identical to rt_render_frame's.  `cancel` (may be NULL)
 * is polled like `do_cancel` (renderer.rs:25-30) before every callback and
 * while the call waits for the GPU: once it is non-zero the waves in flight
 * stop at their next work item, the call returns RT_OK and emits nothing
 * further; tiles delivered before stay delivered (cpu.rs:55-62).  Passing a
 * flag costs nothing while it stays zero.  If it is already set on entry the
 * call returns RT_ERR_CANCEL_EVENT (cpu.rs:82-85).
 * params->strip_count > 1 is refused (RT_ERR_INVALID_ARGUMENT): tiles are finished pieces of the
 * frame.  rt_scene_last_stats after this call: kernel_ms spans the launch. */
int rt_render(RtScene *scene, const RtCamera *camera, const RtRenderParams *params,
              RtTileCallback callback, void *user, const volatile int *cancel);

/* rt_render with the cancel hook as a FUNCTION (NULL = never cancelled): the form a
 * binding of the reference uses, whose cancel event is an object, not an int in memory. */
int rt_render_ex(RtScene *scene, const RtCamera *camera, const RtRenderParams *params,
                 RtTileCallback callback, void *user, RtCancelCallback cancelled, void *cancel_user);

/* What the reference does to a finished tile downstream of the renderer, on
 * the device: ScreenBuffer::update's tone map (image_buffer.rs:147-153) and
 * SavePng's packing `(c * 255.0) as u32 -> (r << 24 | g << 16 | b << 8 | 255)`
 * as big-endian bytes (image_action/png.rs:21-31), fused in one pass over a
 * frame produced by rt_render_frame_device.  rgb_device: n_pixels*3 f64;
 * rgba_device: n_pixels*4 bytes; mapped_device: n_pixels*3 f64 receiving the
 * tone-mapped floats, or NULL.  Enqueued on `hip_stream`, no synchronisation.
 * Arithmetic is unfused IEEE f64, so the bytes equal rth_tone_map +
 * rth_pack_rgba8 on the host. */
int rt_post_rgba8_device(RtScene *scene, const RtToneMap *tone_map, const double *rgb_device,
                         size_t n_pixels, uint8_t *rgba_device, double *mapped_device, void *hip_stream);

/* rt_render_frame + rt_post_rgba8_device + copy: out_rgba is HOST memory,
 * width*height*4 bytes (what SavePng hashes and encodes).  Whole frames only
 * (params->strip_count > 1 is refused): with several GPUs gather the strips of
 * rt_render_frame_device first and pack on the gathering rank with rt_post_rgba8_device. */
int rt_render_frame_rgba8(RtScene *scene, const RtCamera *camera, const RtRenderParams *params,
                          const RtToneMap *tone_map, uint8_t *out_rgba);

/* ---------------------------------------------------------- several devices
 * The reference shards one frame over its rayon pool by tile inside ONE call
 * (renderer/cpu.rs:118-131); these do the same over the GPUs of one process.
 * scenes[i] are RtScene objects of the SAME description created on the devices
 * that should take part (a device may appear twice: two scenes on it share it).
 * The frame is cut into strips of `strip_rows` rows (0 = 8), strip j belongs to
 * scenes[j % n_scenes]; every device traces its strips concurrently on its own
 * stream and the finished pixels are collected
 *   - rt_render_frame_multi:        in out_rgb (HOST memory): every device writes
 *     its finished pixels over its own PCIe link into one pinned frame, which the
 *     call copies to out_rgb band by band while the devices render on;
 *   - rt_render_frame_multi_device: in out_rgb_device, memory of scenes[0]'s
 *     device, by peer copies over xGMI (one strided copy per device on the
 *     SOURCE device's stream behind its resolve pass; a single process needs no
 *     RCCL rendezvous for that — the multi-PROCESS path gathers with RCCL,
 *     racer-tracer_amd/strips.py).  Synchronises before returning;
 *   - rt_render_multi:              as rt_render_ex's tile stream: a tile column
 *     is handed to the callback as soon as EVERY device has finished its strips
 *     of it (same order, same tiles, same cancel behaviour as rt_render_ex), so
 *     a several-GPU `impl Renderer` keeps progressive delivery and cancel.
 * params->strip_* must be unset (the call sets them per device) and
 * params->scale <= 1.  The frame is bit-identical to rt_render_frame's for
 * every n_scenes when strip_rows is a multiple of 8 (the RNG is addressed by the
 * global pixel index and strips are then whole 8-row item tiles); other strip
 * heights regroup the pixels that share a tile's sums and agree to rounding.
 * rt_scene_last_stats(scenes[i]) afterwards gives device i's share. */
int rt_render_frame_multi(RtScene *const *scenes, int n_scenes, const RtCamera *camera,
                          const RtRenderParams *params, int strip_rows, double *out_rgb);
int rt_render_frame_multi_device(RtScene *const *scenes, int n_scenes, const RtCamera *camera,
                                 const RtRenderParams *params, int strip_rows, double *out_rgb_device);
int rt_render_multi(RtScene *const *scenes, int n_scenes, const RtCamera *camera, const RtRenderParams *params,
                    int strip_rows, RtTileCallback callback, void *user, RtCancelCallback cancelled, void *cancel_user);

/* Stats of the most recent render call on this scene (synchronises the
 * stream of that call first). */
int rt_scene_last_stats(RtScene *scene, RtRenderStats *out);

/* Static description of an error code; never NULL. */
const char *rt_strerror(int code);
/* Thread-local detail of the last failure in this library ("" if none). */
const char *rt_last_error_message(void);

#ifdef __cplusplus
}
#endif

#endif /* RT_ABI_H */
