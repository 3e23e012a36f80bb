// rt_device_types.h — device-side tables of a scene, as laid out in HBM.
//
// These are the upload forms of include/rt_abi.h's PODs: same content,
// re-packed so that (a) the closest-hit loop reads one 192-byte record per
// primitive with a wave-uniform index (scalar loads) and (b) what shading
// needs about a material — kind, fuzz, ior, and the colour when its texture
// is a plain SolidColor — sits in one record, so the common case costs a
// single per-lane gather.
#pragma once
#include <stdint.h>

namespace rtdev {

struct alignas(16) Material { // 64 B
    int32_t kind;             // RtMaterialKind
    int32_t texture;          // index into textures
    int32_t tex_kind;         // RtTextureKind of `texture` (-1 for Dielectric)
    int32_t needs_uv;         // texture tree contains an Image texture
    double fuzz;
    double ior;
    // the texture's colour when tex_kind == SolidColor; for a Dielectric (no texture) the three quotients
    // dialectric.rs:26,17-19 forms on every hit, formed once at upload with the same IEEE divisions:
    // color[0] = 1 / ior, color[1] = ((1 - 1/ior) / (1 + 1/ior))^2 (front face), color[2] = ((1 - ior) / (1 + ior))^2
    double color[3];
    double _pad;
};
static_assert(sizeof(Material) == 64, "Material must be 64 bytes");

struct alignas(16) Prim { // 192 B
    double p[6];          // sphere: c.xyz, r | rect: a0,a1,b0,b1,k | box: min.xyz,max.xyz
    double rot_sin, rot_cos;
    double tr[3];
    int32_t kind;         // RtPrimitiveKind
    int32_t flags;        // RtPrimitiveFlags
    int32_t material;
    int32_t _pad0;
    double inv_radius;    // sphere: 1.0 / radius (sphere.rs:61 divides via reciprocal)
    double radius2;       // sphere: radius * radius (sphere.rs:43), formed once at upload
    double _pad1;
    // The primitive's material, copied in at upload (scene.rs:74-76: a SceneObject owns its
    // material): shading the winning primitive is then ONE round of per-lane loads instead of
    // the dependent chain primitive -> material index -> material record.
    Material mat;
};
static_assert(sizeof(Prim) == 192, "Prim must be 192 bytes");

struct alignas(16) Texture { // 64 B
    int32_t kind;
    int32_t tex_even, tex_odd;
    int32_t image, perlin, depth;
    int32_t _pad[2];
    union {
        double color[3]; // SolidColor / Noise
        struct {         // Image: a copy of images[image], so that a lookup is ONE dependent load (the texel), not
            const uint8_t *rgba; // texture -> image record -> texel (the texture table sits in LDS in the pooled kernel)
            int32_t width, height;
        } img;
    };
    double scale;
};
static_assert(sizeof(Texture) == 64, "Texture must be 64 bytes");

struct Image {
    const uint8_t *rgba; // device pointer, RGBA8 row-major, row 0 = top
    int32_t width, height;
};

struct Perlin { // gradient table only: perm tables are folded into `index`
    double ranvec[256][3];
    int32_t perm_x[256], perm_y[256], perm_z[256];
};

// 32 B; depth-first order with a skip link (rt_bvh.h).  The box is for CULLING only, so it is kept in single
// precision: coordinates relative to the root box's centre (TraceArgs.bvh_center), rounded outward and padded by
// 2^-19 of the scene's extent — more than the f32 slab test of a ray whose origin lies inside the root box can be
// off by (rt_bvh_slab.h; closest_hit_bvh clips the ray to the root box in f64 first).  Primitives are tested in f64 as
// everywhere.  The two planes of an axis sit side by side, so that one v_pk_fma_f32 forms both distances.
// The array ends with a SENTINEL at index n: an all-space box that is a leaf without primitives — the node every
// finished walk arrives at through its last skip link and "hits", so the descent loop has ONE way out (a leaf was
// entered) and no `i < n` test per step.
struct alignas(16) BvhNode {
    float lohi[6];        // mn.x, mx.x, mn.y, mx.y, mn.z, mx.z
    int32_t skip;         // next node when this subtree is finished or missed (n = the sentinel)
    int32_t first_count;  // leaf: (first primitive of the leaf << 3) | number of primitives (1..7); inner: 0; sentinel: kSentinel
    float mn(int k) const { return lohi[2 * k]; }
    float mx(int k) const { return lohi[2 * k + 1]; }
    static constexpr int32_t kSentinel = 1 << 30; // a leaf (non-zero) of zero primitives
};
static_assert(sizeof(BvhNode) == 32, "BvhNode must be 32 bytes");

// What the BVH walk needs to TEST a leaf primitive, 64 B instead of the 128 B of a Prim record it would touch: a
// lane's leaf tests are gathers from global memory (the table is in leaf order; `random`: 40 % of the walk), and
// a gather costs the L1 one pass per lane and instruction.  tag 0: an unwrapped Sphere or MovingSphere whose
// centre is c0 + ((time - leaf_time_a) * leaf_inv_dt) * dc (moving_sphere.rs:37-39; dc = 0 for a Sphere), with
// the scene-wide time interval of TraceArgs; tag 1: anything else — the walk then reads the Prim record.
struct alignas(64) LeafGeo {
    double c0[3], radius2;
    double dc[3];
    int64_t tag;
};
static_assert(sizeof(LeafGeo) == 64, "LeafGeo must be 64 bytes");

// dynamic LDS of the pooled kernel's lens-disk samples per block: 4 waves x NBUF batches x 64 entries x (x, y);
// NBUF = 1 in the BVH variants, 2 elsewhere (rt_trace_pool_kernel.hip: WaveLds)
inline size_t pool_lens_lds_bytes(bool bvh) { return (size_t)4 * (bvh ? 1 : 2) * 64 * 2 * sizeof(double); }
// ... and of the batches' ray times, when the scene has a MovingSphere: 4 waves x NBUF batches x 64 entries
inline size_t pool_time_lds_bytes(bool bvh) { return (size_t)4 * (bvh ? 1 : 2) * 64 * sizeof(double); }

struct Camera { // what get_ray reads (camera.rs:326-337)
    double origin[3], ulc[3], right[3], up[3], horizontal[3], vertical[3];
    double lens_radius;
    double time_a, time_b; // ray time = R(time_a, time_b), camera.rs:335 (read by MovingSphere only)
};

struct Background {
    int32_t kind, _pad;
    double top[3], bottom[3];
};

// Slots of the statistics buffer TraceArgs.segments[RT_STAT_SLOTS]: path segments, primary rays started
// (counted where a path is handed out), and — written by a -DRT_PROFILE_REGIONS build only — shader-clock
// cycles per region of the path loop, wall-clock marks of the first/last wave, and two histograms of the
// lanes tracing per iteration (bins of 8: 0, 1-8, ..., 57-64): while the item's pool still has paths to
// hand out, and after it ran dry (the item's tail).
enum {
    RT_STAT_SEGMENTS = 0,
    RT_STAT_SAMPLES = 1,
    RT_STAT_REGIONS = 2,      // 16 regions
    RT_STAT_WALL = 18,        // first start, last start, first end, last end
    RT_STAT_LANES_BODY = 22,  // 9 bins
    RT_STAT_LANES_TAIL = 31,  // 9 bins
    RT_STAT_NOISE = 40,       // wave-iterations with a Noise lookup, lookups; BVH nodes visited, leaf primitives tested
    RT_STAT_REGION_LANES = 44, // 16 regions: cycles x active lanes at the region's closing marker
    RT_STAT_SLOTS = 60
};

enum { RT_MAX_CHUNKS = 64 }; // a frame's samples are cut into at most this many chunks (slices of `partial`)
enum { RT_MAX_REGIONS = 32 }; // a launch's items are queued region by region (TraceArgs.regions)

// One region of a launch: a rectangle of 8x8 item tiles whose items (tile x chunk, chunk-major inside the region)
// are queued together, so that the region's pixels are FINISHED together: a tile column of rt_render's tile stream,
// a band of rows of rt_render_frame.  A launch with one region is the plain whole-grid, chunk-major queue.
struct Region {
    uint32_t item_begin; // first item of the region (regions are listed in queue order)
    int32_t tx0, ntx;    // tile columns [tx0, tx0 + ntx)
    int32_t ty0, nty;    // tile rows    [ty0, ty0 + nty)  (rows of the launch's OWNED-row grid)
};

// Kernel argument block (passed by value -> kernarg segment, scalar loads).
struct TraceArgs {
    const Prim *prims;
    const Texture *textures;
    const Image *images;
    const Perlin *perlins;
    int32_t n_prims, n_materials, n_textures, n_images, n_perlins;
    int32_t perlin_identity;     // every Perlin's permutation tables are the identity (always so in the reference)
    int32_t perlin_in_lds;       // the first table's gradients (6 KB) are staged at the end of the dynamic LDS
    int32_t width, height;       // full image
    int32_t samples, max_depth;
    int32_t sample_begin, sample_end; // this launch accumulates samples [begin, end)
    int32_t strip_rows, strip_count, strip_index;
    int32_t owned_rows;          // number of image rows this launch covers
    uint32_t seed_lo, seed_hi;
    Camera cam;
    Background bg;
    double *accum;               // [height][width][3] running sums (owned rows only written)
    unsigned long long *segments; // global segment counter
    // --- pooled kernel (k_trace_pool_f64) ---
    // Work item = (8x8 pixel tile, chunk of samples); items are numbered
    // chunk-major: item = chunk * n_tiles + tile.  Each item writes the sums of
    // its samples to partial[chunk_base + chunk][pixel]; k_resolve adds the
    // chunks in index order, so the result does not depend on scheduling.
    double *partial;             // [n_chunks_total][slice_rows][width][3]: the rows of the launch's OWNED-row grid only
    int32_t slice_rows;          // rows per slice = owned_rows (a rank's eighth of a C5 frame keeps 0.53 GB of slices, not 4.2)
    unsigned int *queue;         // item counter (zeroed before the launch)
    // CANCEL.  A word in pinned host memory (or NULL when the call cannot be cancelled): the lane that fetches a wave's
    // next item also reads it, and non-zero means "the queue is dry".  The host raises it with a plain store — no
    // command-processor packet, no fill kernel, nothing that has to find room on a GPU that a persistent grid fills:
    // hipStreamWriteValue32 turned out to be a blit KERNEL, which only lands while a SIMD has registers to spare (it
    // never did beside the 128-VGPR variants, nor while another share's launch waited on the same device).
    const unsigned int *cancel_flag;
    uint32_t n_items;
    int32_t n_chunks;            // chunks in this launch
    int32_t chunk_base;          // index of this launch's first chunk in `partial`
    int32_t chunk_samples;       // samples per full-length chunk (informational; the kernel reads chunk_start)
    // Chunk c of the frame covers the samples [chunk_start[c], chunk_start[c + 1]): full-length chunks first, then a
    // taper of ever shorter ones, so that the last items of a launch are small (rt_api.hip: chunk_plan).
    int32_t chunk_start[RT_MAX_CHUNKS + 1];
    int32_t tiles_x, n_tiles;    // 8x8 tiles over width x owned_rows
    // cpu.rs:36,40 divide by (W-1) and (H-1); the pooled kernel multiplies by these
    double inv_width_m1, inv_height_m1;
    // Preview renderer (cpu_scaled.rs): grid cell (gx, gy) is pixel (gx*step_x, gy*step_y);
    // cover_w = grid width * step_x.  step_x = step_y = 1 and cover_w = width otherwise.
    int32_t step_x, step_y, cover_w, cover_h;
    // Regions of the item queue (above).  n_regions <= 1: item = chunk * n_tiles + tile over the whole grid.
    int32_t n_regions;
    Region regions[RT_MAX_REGIONS];
    // DELIVERY (rt_render, rt_render_frame and their multi-device forms).  When `deliver_out` is set the launch
    // finishes its own pixels: the wave that completes the LAST chunk of a tile (tile_done, one counter per tile,
    // re-armed to 0 by that wave) adds the tile's slices in chunk order — k_resolve_chunks_f64's arithmetic —
    // and writes sqrt(sum / samples) straight to `deliver_out`, which may be pinned HOST memory mapped into the
    // device: no resolve launch, no copy engine, nothing that needs a free compute unit beside the persistent
    // grid.  Layout of deliver_out: tile columns of `deliver_col_step` pixels (the last one takes the remainder,
    // cpu.rs:97-109), each stored as [height][column width][3] — so a tile of rt_render's stream is a contiguous
    // run — or the plain [height][width][3] frame when deliver_cols == 1.  The wave that finishes a region's last
    // tile (region_done) publishes `deliver_serial` in deliver_flags[region] (pinned host memory), which the
    // calling thread polls.
    double *deliver_out;
    unsigned int *tile_done;      // [n_tiles of the owned-row grid], zero between launches
    unsigned int *region_done;    // [RT_MAX_REGIONS], zero between launches
    unsigned int *deliver_flags;  // [RT_MAX_REGIONS] in pinned host memory
    uint32_t deliver_serial;
    int32_t deliver_col_step, deliver_cols;
    int32_t total_chunks;         // chunks of the frame (== n_chunks of a single-launch render)
    // BVH (scenes with more primitives than the brute-force loop is good for)
    const BvhNode *bvh_nodes;
    // trees too large for LDS: eight copies of the node array (n_bvh_nodes + 1 entries each), one per sign octant of the
    // ray direction, in which the child that lies first along such a ray is the first child (rt_bvh.cpp); NULL otherwise
    const BvhNode *bvh_nodes_ordered;
    double bvh_root_mn[3], bvh_root_mx[3]; // the root box in f64 (padded like the node boxes)
    double bvh_center[3];                  // origin of the node boxes' coordinates
    const int32_t *bvh_prim_index;
    int32_t n_bvh_nodes;   // without the sentinel that follows them in the array
    int32_t bvh_lds_nodes; // == n_bvh_nodes + 1 (the sentinel is staged too) when the node array is staged in dynamic LDS, else 0
    const LeafGeo *leaf_geo;             // per primitive of the (leaf-ordered) table
    double leaf_time_a, leaf_inv_dt;     // MovingSphere.time_a and 1 / (time_b - time_a) of the tag-0 records
    // Linear-loop variants: the device table is grouped — untransformed XY rects first, then XZ, then YZ
    // (rect_end[a] is the end of group a), then untransformed spheres (sphere_end), then boxes with or without
    // wrappers (box_end), then everything else (moving spheres, wrapped rects and spheres) — so that the
    // closest-hit loop runs one straight-line test per group instead of a scalar switch on the kind of every record.
    int32_t rect_end[3];
    int32_t sphere_end;
    int32_t box_end;
    int32_t lens_lds;      // the camera has an aperture: the lens-disk samples of the batches sit at the end of dynamic LDS
    // FIXED-POINT SUMS (the pooled variants that keep two items in flight: rt_trace_pool_kernel.hip, OVERLAP).  sum_scale =
    // 2^k: a finished sample's radiance T is added to its pixel's sum as the INTEGER round(T * sum_scale) (< 2^52: the host
    // derives k from a bound on the scene's radiance — every attenuation <= 1, emission and background <= E — and on the
    // samples of a chunk, so the 64-bit sums cannot overflow), and the item's sum is that integer, rounded once, times
    // sum_unscale = 2^-k.  Integer sums do not depend on the order of their terms, which is what lets a wave start its next
    // item while the last paths of the previous one are still in flight without the frame depending on scheduling.  The
    // price is an ABSOLUTE quantum of E 2^-52 per sample where a double has a relative one: a pixel whose radiance is of
    // that order — black, for every purpose — comes out up to sqrt(E 2^-53) (4e-8 for E = 16) from the f64 sum's value.
    // The other variants, and every RT_ARITH_REFERENCE kernel, add doubles in the order the samples finish, one item at a time.
    double sum_scale, sum_unscale;
    int32_t time_lds;  // the scene has a MovingSphere: the ray times of the batches sit behind the lens samples in dynamic LDS
    int32_t _pad_sum;
    int32_t dbg[4];        // developer knobs (env RT_DBG0..3), 0 in production
};

} // namespace rtdev
