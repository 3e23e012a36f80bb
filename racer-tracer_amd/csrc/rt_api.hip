// rt_api.hip — implementation of the C ABI declared in include/rt_abi.h.
//
// Host code only (HIP runtime calls); the kernels live in
// rt_trace_kernel.hip.  Nothing here falls back to a CPU renderer: without a
// usable HIP device every entry point returns RT_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <cmath>
#include <atomic>
#include <mutex>
#include <new>
#include <string>
#include <vector>
#include <thread>
#include <chrono>
#include "rt_bvh.h"
#include "rt_scene.h"

// The trace kernels exist twice (rt_trace_common.h: ARITHMETIC): RT_ARITH_FAST, and RT_ARITH_REFERENCE behind *_exact.
#define RT_DECLARE_LAUNCHERS(SUFFIX)                                                                                       \
    extern "C" hipError_t rtdev_launch_trace##SUFFIX(const rtdev::TraceArgs *args, int prims_class, int textured,         \
                                                     int specular, hipStream_t stream);                                   \
    extern "C" hipError_t rtdev_launch_resolve##SUFFIX(const double *accum, double *out, int width, int height,           \
                                                       int strip_rows, int strip_count, int strip_index, int samples,     \
                                                       hipStream_t stream);                                               \
    extern "C" int rtdev_pool_blocks_per_cu##SUFFIX(int prims_class, int textured, int specular, int bvh, size_t dyn_lds); \
    extern "C" hipError_t rtdev_launch_trace_pool##SUFFIX(const rtdev::TraceArgs *args, int prims_class, int textured,    \
                                                          int specular, int bvh, unsigned blocks, hipStream_t stream);    \
    extern "C" hipError_t rtdev_launch_resolve_chunks##SUFFIX(const double *partial, double *out, int width, int height,  \
                                                              int n_chunks, int slice_rows, int strip_rows, int strip_count, \
                                                              int strip_index, int step_x, int step_y, int cover_w,       \
                                                              int cover_h, int out_col_step, int out_cols, int samples,   \
                                                              hipStream_t stream);
RT_DECLARE_LAUNCHERS()
RT_DECLARE_LAUNCHERS(_exact)
extern "C" hipError_t rtdev_launch_post_rgba8(const RtToneMap *tm, const double *rgb, size_t n_pixels, uint8_t *rgba,
                                              double *mapped, hipStream_t stream);

thread_local std::string g_last_error;

int rtapi::fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}
using rtapi::Cancel;
using rtapi::Delivery;
using rtapi::DevBuf;
using rtapi::fail;


namespace {

int validate_desc(const RtSceneDesc *d) {
    if (!d) return fail(RT_ERR_INVALID_ARGUMENT, "scene description is NULL");
    if (d->n_primitives < 0 || d->n_materials < 0 || d->n_textures < 0 || d->n_images < 0 || d->n_perlins < 0)
        return fail(RT_ERR_INVALID_ARGUMENT, "negative table size");
    if ((d->n_primitives && !d->primitives) || (d->n_materials && !d->materials) ||
        (d->n_textures && !d->textures) || (d->n_images && !d->images) || (d->n_perlins && !d->perlins))
        return fail(RT_ERR_INVALID_ARGUMENT, "NULL table with non-zero size");
    for (int i = 0; i < d->n_textures; ++i) {
        const RtTexture &t = d->textures[i];
        switch (t.kind) {
        case RT_TEX_SOLID_COLOR: break;
        case RT_TEX_CHECKERED:
            for (int c : {t.tex_even, t.tex_odd}) {
                if (c < 0 || c >= d->n_textures)
                    return fail(RT_ERR_SCENE_LOAD, "Checkered texture " + std::to_string(i) + " names a missing texture");
                if (d->textures[c].kind == RT_TEX_CHECKERED) // scene/yml.rs:212-243 resolves one level only
                    return fail(RT_ERR_UNSUPPORTED, "Checkered texture of a Checkered texture");
            }
            break;
        case RT_TEX_IMAGE:
            if (t.image < 0 || t.image >= d->n_images) return fail(RT_ERR_INVALID_ARGUMENT, "texture image index out of range");
            break;
        case RT_TEX_NOISE:
            if (t.perlin < 0 || t.perlin >= d->n_perlins) return fail(RT_ERR_INVALID_ARGUMENT, "texture perlin index out of range");
            if (t.depth < 0) return fail(RT_ERR_INVALID_ARGUMENT, "negative noise depth");
            break;
        default: return fail(RT_ERR_INVALID_ARGUMENT, "unknown texture kind");
        }
    }
    for (int i = 0; i < d->n_images; ++i)
        if (!d->images[i].rgba || d->images[i].width <= 0 || d->images[i].height <= 0)
            return fail(RT_ERR_FAILED_TO_OPEN_IMAGE, "image " + std::to_string(i) + " is empty");
    for (int i = 0; i < d->n_materials; ++i) {
        const RtMaterial &m = d->materials[i];
        if (m.kind < RT_MAT_LAMBERTIAN || m.kind > RT_MAT_DIFFUSE_LIGHT)
            return fail(RT_ERR_UNKNOWN_MATERIAL, "unknown material kind");
        if (m.kind != RT_MAT_DIELECTRIC && (m.texture < 0 || m.texture >= d->n_textures))
            return fail(RT_ERR_SCENE_LOAD, "material " + std::to_string(i) + " names a missing texture");
    }
    for (int i = 0; i < d->n_primitives; ++i) {
        const RtPrimitive &p = d->primitives[i];
        if (p.kind < RT_PRIM_SPHERE || p.kind > RT_PRIM_MOVING_SPHERE) return fail(RT_ERR_INVALID_ARGUMENT, "unknown primitive kind");
        if (p.kind == RT_PRIM_MOVING_SPHERE && (p.flags & (RT_PRIM_HAS_ROTATE_Y | RT_PRIM_HAS_TRANSLATE)))
            return fail(RT_ERR_UNSUPPORTED, "a MovingSphere cannot be wrapped in RotateY/Translate");
        if (p.material < 0 || p.material >= d->n_materials)
            return fail(RT_ERR_UNKNOWN_MATERIAL, "primitive " + std::to_string(i) + " names a missing material");
    }
    if (d->background.kind != RT_BG_SKY && d->background.kind != RT_BG_SOLID)
        return fail(RT_ERR_INVALID_ARGUMENT, "unknown background kind");
    return RT_OK;
}

bool texture_reads_uv(const RtSceneDesc *d, int ti) {
    const RtTexture &t = d->textures[ti];
    if (t.kind == RT_TEX_IMAGE) return true;
    if (t.kind == RT_TEX_CHECKERED)
        return d->textures[t.tex_even].kind == RT_TEX_IMAGE || d->textures[t.tex_odd].kind == RT_TEX_IMAGE;
    return false;
}

template <class T> int upload(DevBuf<T> &buf, const std::vector<T> &host) {
    RT_HIP(buf.alloc(host.size()));
    if (!host.empty()) RT_HIP(hipMemcpy(buf.ptr, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    return RT_OK;
}

} // namespace

int rtapi::check_params(const RtCamera *camera, const RtRenderParams *p) {
    if (!camera || !p) return fail(RT_ERR_INVALID_ARGUMENT, "camera/params is NULL");
    // cpu.rs:36,40 divide by (W - 1) and (H - 1): a one-pixel dimension is a division by zero in the
    // reference (inf/NaN rays, an undefined picture); it is refused here instead of imitated
    if (p->width < 2 || p->height < 2) return fail(RT_ERR_INVALID_ARGUMENT, "width and height must be at least 2");
    if (p->samples <= 0) return fail(RT_ERR_INVALID_ARGUMENT, "samples must be positive");
    if (p->max_depth < 0 || p->max_depth >= (1 << 24)) return fail(RT_ERR_INVALID_ARGUMENT, "max_depth out of range");
    if ((uint64_t)p->width * (uint64_t)p->height > 0xFFFFFFFFull) return fail(RT_ERR_INVALID_ARGUMENT, "image too large for the pixel counter");
    if (p->scale < 0) return fail(RT_ERR_INVALID_ARGUMENT, "scale must not be negative");
    if (p->scale > 1 && p->strip_count > 1) return fail(RT_ERR_INVALID_ARGUMENT, "the preview scale cannot be combined with strips");
    if (p->strip_count > 1) {
        if (p->strip_rows <= 0) return fail(RT_ERR_INVALID_ARGUMENT, "strip_rows must be positive when strip_count > 1");
        if (p->strip_index < 0 || p->strip_index >= p->strip_count) return fail(RT_ERR_INVALID_ARGUMENT, "strip_index out of range");
    }
    return RT_OK;
}
using rtapi::check_params;

namespace {

// SAMPLE CHUNKS.  A work item of the pooled kernel is a tile x a chunk of its samples, and the order in which a
// pixel's samples are summed follows the chunk boundaries, so they depend on the sample count ONLY (never on
// tiling, strips, batches or the device): the frame is bit-identical for every GPU count.
// Returns the start sample of every chunk plus the total (size = chunks + 1).
// * About sixteen full-length chunks per frame: every item ends in a tail of ~20 iterations in which its last
//   deep paths die out at a handful of lanes (7.5 % of C3's iterations with chunks of 32), so long chunks pay -
//   until items become too few and too long for the end of a launch to balance, which a rank's share of a
//   multi-GPU frame reaches first.  Measured on the 1080p frames with the taper below in place
//   (tools/perf_ab.sh RT_POOL_CHUNK=.., tools/strip_share.py), ms per frame / slowest of 8 shares: C3 (1024 spp)
//   full chunks of 44: 94.4 / 13.4, 64: 93.4 / 13.6, 88: 92.9 / 13.8, 128: 92.5 / 14.4 (round 1's fixed 32 without
//   taper: 96.6 / 13.9); C2 (256 spp) 16: 18.5, 24: 18.3, 32: 18.6; C4 (512 spp) 24: 59.8, 32: 59.4, 44: 59.4, 64: 59.8.
//   spp / 16, at least 24, serves one GPU and eight.
// * The last one to two chunk lengths of samples are cut into ever shorter chunks (halving down to 4 samples): items
//   are queued chunk-major, so a launch ends on small items and its waves finish together.
std::vector<int> chunk_plan(int samples) {
    int full = ((samples + 15) / 16 + 3) / 4 * 4;
    if (full < 24) full = 24;
#ifdef RT_DEVELOPER_KNOBS // changes the summation order: never in the product build
    if (const char *k = getenv("RT_POOL_CHUNK"))
        if (atoi(k) > 0) full = atoi(k);
#endif
    std::vector<int> starts;
    int at = 0;
    while (samples - at >= 2 * full && (int)starts.size() < rtdev::RT_MAX_CHUNKS - 8) {
        starts.push_back(at);
        at += full;
    }
#ifdef RT_DEVELOPER_KNOBS
    const bool taper = getenv("RT_POOL_NO_TAPER") == nullptr;
#else
    const bool taper = true;
#endif
    while (samples - at > 8 && taper) {
        starts.push_back(at);
        const int rest = samples - at;
        at += rest >= 2 * full ? full : (rest / 2 + 3) / 4 * 4; // more than 2 x full only when the chunk table is full
        if ((int)starts.size() >= rtdev::RT_MAX_CHUNKS - 1) break;
    }
    if (at < samples) starts.push_back(at);
    starts.push_back(samples);
    return starts;
}

void fill_args(const RtScene *s, const RtCamera *c, const RtRenderParams *p, rtdev::TraceArgs &a) {
    memset(&a, 0, sizeof a);
    a.prims = s->prims.ptr;
    a.textures = s->textures.ptr;
    a.images = s->images.ptr;
    a.perlins = s->perlins.ptr;
    a.n_prims = s->n_prims;
    a.n_materials = s->n_materials;
    a.n_textures = s->n_textures;
    a.n_images = s->n_images;
    a.n_perlins = s->n_perlins;
    a.perlin_identity = s->perlin_identity;
    a.perlin_in_lds = s->textured && s->n_perlins > 0 && s->perlin_identity;
    a.width = p->width;
    a.height = p->height;
    a.samples = p->samples;
    a.max_depth = p->max_depth;
    a.sample_begin = 0;
    a.sample_end = p->samples;
    if (p->strip_count > 1) {
        a.strip_rows = p->strip_rows;
        a.strip_count = p->strip_count;
        a.strip_index = p->strip_index;
        a.owned_rows = rtapi::owned_rows_of(p);
    } else {
        a.strip_rows = p->height;
        a.strip_count = 1;
        a.strip_index = 0;
        a.owned_rows = p->height;
    }
    a.step_x = a.step_y = 1;
    a.cover_w = p->width;
    a.cover_h = p->height;
    if (p->scale > 1) { // CpuRendererScaled::new (cpu_scaled.rs:33-41) + raytrace's scaled grid (:50-52)
        auto highest_divisible = [](int value, int div) { // cpu_scaled.rs:18-24
            while (value % div != 0) --div;
            return div;
        };
        const int tw = p->tiles_w > 0 ? p->tiles_w : 1, th = p->tiles_h > 0 ? p->tiles_h : 1;
        a.step_x = highest_divisible(p->width / tw, p->scale);
        a.step_y = highest_divisible(p->height / th, p->scale);
        a.cover_w = (p->width / a.step_x) * a.step_x;
        a.cover_h = (p->height / a.step_y) * a.step_y;
        a.owned_rows = p->height / a.step_y; // grid rows
    }
    a.seed_lo = (uint32_t)(p->seed & 0xffffffffull);
    a.seed_hi = (uint32_t)(p->seed >> 32);
    a.inv_width_m1 = 1.0 / (double)(p->width - 1);
    a.inv_height_m1 = 1.0 / (double)(p->height - 1);
    for (int k = 0; k < 3; ++k) {
        a.cam.origin[k] = c->origin[k];
        a.cam.ulc[k] = c->upper_left_corner[k];
        a.cam.right[k] = c->right[k];
        a.cam.up[k] = c->up[k];
        a.cam.horizontal[k] = c->horizontal[k];
        a.cam.vertical[k] = c->vertical[k];
    }
    a.cam.lens_radius = c->lens_radius;
    a.lens_lds = c->lens_radius != 0.0;
    a.time_lds = s->has_moving;
    a.cam.time_a = c->time_a;
    a.cam.time_b = c->time_b;
    a.bg = s->bg;
    a.accum = s->accum.ptr;
    a.segments = s->segments.ptr;
    a.bvh_nodes = s->bvh_nodes.ptr;
    a.bvh_nodes_ordered = s->bvh_nodes_in_lds ? nullptr : s->bvh_nodes_ordered.ptr;
    a.bvh_prim_index = s->bvh_prim_index.ptr;
    a.n_bvh_nodes = s->n_bvh_nodes;
    for (int k = 0; k < 3; ++k) {
        a.bvh_root_mn[k] = s->bvh_root_mn[k];
        a.bvh_root_mx[k] = s->bvh_root_mx[k];
        a.bvh_center[k] = s->bvh_center[k];
    }
    a.bvh_lds_nodes = s->bvh_nodes_in_lds ? s->n_bvh_nodes + 1 : 0; // the sentinel behind the tree's nodes is staged too
    a.leaf_geo = s->leaf_geo.ptr;
    a.leaf_time_a = s->leaf_time_a;
    a.leaf_inv_dt = s->leaf_inv_dt;
    for (int g = 0; g < 3; ++g) a.rect_end[g] = s->rect_end[g];
    a.sphere_end = s->sphere_end;
    a.box_end = s->box_end;
    // Fixed-point sums (rt_device_types.h: sum_scale): a sample's radiance is at most radiance_bound < 2^e, so
    // T * 2^(52 - e) < 2^52 — what the kernel's conversion can hold — and 2048 of them, the samples of the longest chunk
    // (or the scale halves), stay below 2^63.
    if (!s->exact && !s->use_v1 && s->radiance_bound > 0.0) {
        int e = 0;
        // radiance_bound < 2^e, with room for the last bits a sample may exceed the bound by (a sky blend or a Noise factor
        // an ulp above 1, twenty bounces deep): a bound within 1e-6 of the power of two takes the next one
        if (frexp(s->radiance_bound, &e) > 1.0 - 1e-6) ++e;
        const std::vector<int> plan = chunk_plan(p->samples);
        int longest = 1;
        for (size_t k = 0; k + 1 < plan.size(); ++k) longest = std::max(longest, plan[k + 1] - plan[k]);
        for (; longest > 2048; longest = (longest + 1) / 2) ++e;
        a.sum_scale = ldexp(1.0, 52 - e);
        a.sum_unscale = ldexp(1.0, e - 52);
    }
#ifdef RT_DEVELOPER_KNOBS // throw-away kernel knobs of the developer build (tools/perf_ab.sh)
    for (int k = 0; k < 4; ++k) {
        char name[16];
        snprintf(name, sizeof name, "RT_DBG%d", k);
        if (const char *v = getenv(name)) a.dbg[k] = atoi(v);
    }
#endif
}

} // namespace

int rtapi::chunk_count(int samples) { return (int)chunk_plan(samples).size() - 1; }

int rtapi::owned_rows_of(const RtRenderParams *p) {
    if (p->strip_count <= 1) return p->height;
    int owned_strips = 0; // strips j with first row (j*count + index)*rows inside the image
    for (long long j = 0; (j * p->strip_count + p->strip_index) * (long long)p->strip_rows < p->height; ++j) ++owned_strips;
    return owned_strips * p->strip_rows;
}

// What a pooled-kernel launch of these parameters needs in device memory, allocated now.  enqueue_render does the same
// when it finds a buffer too small; a call over SEVERAL shares reserves for all of them before it launches the first
// (rt_deliver.hip, rt_multi.hip): hipMalloc waits for the device's running kernels, so a share that allocated inside its
// enqueue held the calling thread until the shares launched before it had finished — no cancel poll, no band copied
// meanwhile (measured with two scenes on one card: 122 ms of a 4096-spp frame before the hook was looked at).
int rtapi::reserve_render_buffers(RtScene *s, const RtRenderParams *p, bool delivering) {
    if (s->use_v1) return RT_OK;
    RT_HIP(hipSetDevice(s->device));
    const int owned = p->scale > 1 ? p->height : owned_rows_of(p); // (the preview's grid is smaller: an upper bound)
    const size_t slice_elems = (size_t)p->width * (size_t)owned * 3;
    const size_t chunks = (size_t)chunk_plan(p->samples).size() - 1;
    if (s->partial.count < slice_elems * chunks) RT_HIP(s->partial.alloc(slice_elems * chunks));
    if (s->queue.count < 1) RT_HIP(s->queue.alloc(1));
    if (delivering) {
        const size_t n_tiles = (size_t)((p->width + 7) / 8) * (size_t)((owned + 7) / 8);
        if (s->tile_done.count < n_tiles) {
            RT_HIP(s->tile_done.alloc(n_tiles));
            s->deliver_dirty = true;
        }
        if (s->region_done.count < (size_t)rtdev::RT_MAX_REGIONS) {
            RT_HIP(s->region_done.alloc((size_t)rtdev::RT_MAX_REGIONS));
            s->deliver_dirty = true;
        }
    }
    return RT_OK;
}

int rtapi::enqueue_render(RtScene *s, const RtCamera *camera, const RtRenderParams *p, double *out_device,
                          hipStream_t stream, int batch, const Cancel &cancel, const Delivery *delivery, int out_col_step,
                          int out_cols) {
    RT_HIP(hipSetDevice(s->device));
    size_t n = (size_t)p->width * (size_t)p->height * 3;
    rtdev::TraceArgs a;
    fill_args(s, camera, p, a);
    if (batch <= 0 || batch > p->samples) batch = p->samples;
    int launches = 0;
    if (s->use_v1) {
        if (p->scale > 1) return fail(RT_ERR_UNSUPPORTED, "the v1 kernel has no preview mode");
        if (delivery || out_cols > 1) return fail(RT_ERR_UNSUPPORTED, "the v1 kernel does not deliver its own pixels");
        if (s->accum.count < n) RT_HIP(s->accum.alloc(n));
        a.accum = s->accum.ptr;
        RT_HIP(hipMemsetAsync(s->segments.ptr, 0, rtdev::RT_STAT_SLOTS * sizeof(unsigned long long), stream));
        RT_HIP(hipEventRecord(s->ev_begin, stream));
        for (int b = 0; b < p->samples; b += batch) {
            if (cancel.raised()) return RT_ERR_CANCEL_EVENT;
            a.sample_begin = b;
            a.sample_end = b + batch < p->samples ? b + batch : p->samples;
            RT_HIP((s->exact ? rtdev_launch_trace_exact : rtdev_launch_trace)(&a, s->prims_class, s->textured, s->specular, stream));
            ++launches;
            if (cancel.armed()) RT_HIP(hipStreamSynchronize(stream)); // so the next poll is meaningful
        }
        RT_HIP(hipEventRecord(s->ev_traced, stream));
        RT_HIP((s->exact ? rtdev_launch_resolve_exact : rtdev_launch_resolve)(s->accum.ptr, out_device, p->width, p->height, a.strip_rows, a.strip_count,
                                    a.strip_index, p->samples, stream));
        RT_HIP(hipEventRecord(s->ev_resolved, stream));
    } else {
        // Work items = 8x8 tiles x sample chunks.
        a.tiles_x = (a.cover_w / a.step_x + 7) / 8; // grid cells per row
        a.n_tiles = a.tiles_x * ((a.owned_rows + 7) / 8);
        // Sample chunks (chunk_plan above).  Sample batches are cut on chunk boundaries, so batching changes
        // nothing either.
        const std::vector<int> starts = chunk_plan(p->samples);
        const int total_chunks = (int)starts.size() - 1;
        for (int c = 0; c <= total_chunks; ++c) a.chunk_start[c] = starts[(size_t)c];
        a.chunk_samples = starts[1] - starts[0];
        a.total_chunks = total_chunks;
        struct Launch {
            int first_chunk, n_chunks;
        };
        std::vector<Launch> plan;
        for (int c = 0; c < total_chunks;) {
            Launch l{c, 0};
            while (c < total_chunks && (l.n_chunks == 0 || starts[(size_t)c] - starts[(size_t)l.first_chunk] < batch)) {
                ++l.n_chunks;
                ++c;
            }
            plan.push_back(l);
        }
        if (delivery) { // one launch, its items queued region by region, finishing its own pixels
            if (plan.size() != 1 || p->scale > 1) return fail(RT_ERR_UNSUPPORTED, "a delivering launch is one whole-frame launch");
            if (delivery->regions.empty() || (int)delivery->regions.size() > rtdev::RT_MAX_REGIONS)
                return fail(RT_ERR_INVALID_ARGUMENT, "bad region list");
            if (s->tile_done.count < (size_t)a.n_tiles) {
                RT_HIP(s->tile_done.alloc((size_t)a.n_tiles));
                s->deliver_dirty = true;
            }
            if (s->region_done.count < (size_t)rtdev::RT_MAX_REGIONS) {
                RT_HIP(s->region_done.alloc((size_t)rtdev::RT_MAX_REGIONS));
                s->deliver_dirty = true;
            }
            if (s->deliver_dirty) { // fresh buffers, or a launch that was cut short (cancel, error): counters back to zero
                RT_HIP(hipMemsetAsync(s->tile_done.ptr, 0, s->tile_done.count * sizeof(unsigned int), stream));
                RT_HIP(hipMemsetAsync(s->region_done.ptr, 0, s->region_done.count * sizeof(unsigned int), stream));
            }
            s->deliver_dirty = true; // until every region has been published (rt_deliver.hip clears it)
            a.n_regions = (int)delivery->regions.size();
            uint64_t at = 0;
            for (int r = 0; r < a.n_regions; ++r) {
                rtdev::Region reg = delivery->regions[(size_t)r];
                if (reg.ntx <= 0 || reg.nty <= 0 || reg.tx0 < 0 || reg.ty0 < 0 || reg.tx0 + reg.ntx > a.tiles_x ||
                    (reg.ty0 + reg.nty) * a.tiles_x > a.n_tiles)
                    return fail(RT_ERR_INVALID_ARGUMENT, "region outside the tile grid");
                reg.item_begin = (uint32_t)at;
                at += (uint64_t)reg.ntx * (uint64_t)reg.nty * (uint64_t)total_chunks;
                a.regions[r] = reg;
            }
            if (at != (uint64_t)a.n_tiles * (uint64_t)total_chunks) return fail(RT_ERR_INVALID_ARGUMENT, "the regions do not tile the grid");
            a.deliver_out = delivery->out;
            a.tile_done = s->tile_done.ptr;
            a.region_done = s->region_done.ptr;
            a.deliver_flags = s->host_flags;
            a.deliver_serial = delivery->serial;
            a.deliver_col_step = delivery->col_step;
            a.deliver_cols = delivery->cols;
        }
        // a call whose caller polls a cancel hook: the waves read the scene's cancel word with every item they fetch
        if (cancel.armed() || (delivery && delivery->cancellable)) {
            s->host_flags[rtdev::RT_MAX_REGIONS] = 0u;
            a.cancel_flag = s->host_flags + rtdev::RT_MAX_REGIONS;
        }
        const int n_batches = (int)plan.size();
        // slices hold the launch's owned rows only (the kernel compacts rows: owned_rows, tile_py0)
        a.slice_rows = a.owned_rows;
        const size_t slice_elems = (size_t)p->width * (size_t)a.slice_rows * 3;
        if (s->partial.count < slice_elems * (size_t)total_chunks) RT_HIP(s->partial.alloc(slice_elems * (size_t)total_chunks));
        if (s->queue.count < (size_t)n_batches) RT_HIP(s->queue.alloc((size_t)n_batches));
        a.partial = s->partial.ptr;
        RT_HIP(hipMemsetAsync(s->segments.ptr, 0, rtdev::RT_STAT_SLOTS * sizeof(unsigned long long), stream));
#ifdef RT_PROFILE_REGIONS
        RT_HIP(hipMemsetAsync(s->segments.ptr + rtdev::RT_STAT_WALL + 0, 0xff, sizeof(unsigned long long), stream)); // min slots
        RT_HIP(hipMemsetAsync(s->segments.ptr + rtdev::RT_STAT_WALL + 2, 0xff, sizeof(unsigned long long), stream));
#endif
        RT_HIP(hipMemsetAsync(s->queue.ptr, 0, sizeof(unsigned int) * s->queue.count, stream));
        // A tree that lives in LDS is walked in ONE fixed child order: the order that suits the rays starting at this
        // camera (rt_bvh.h: order_bvh_for_origin).  Re-emitted when the camera has moved — microseconds for the few hundred
        // nodes LDS holds — and copied in stream order, i.e. behind whatever launch of this scene still walks the old array.
        if (s->use_bvh && s->bvh_nodes_in_lds && !s->bvh_host.nodes.empty() &&
            (!s->bvh_is_ordered || camera->origin[0] != s->bvh_ordered_for[0] || camera->origin[1] != s->bvh_ordered_for[1] ||
             camera->origin[2] != s->bvh_ordered_for[2])) {
            s->bvh_upload_slot ^= 1; // (two host copies in turn: the one a pending copy may still read is left alone)
            std::vector<rtdev::BvhNode> &arr = s->bvh_ordered_nodes[s->bvh_upload_slot];
            arr = rtdev::order_bvh_for_origin(s->bvh_host, camera->origin);
            RT_HIP(hipMemcpyAsync(s->bvh_nodes.ptr, arr.data(), arr.size() * sizeof(rtdev::BvhNode), hipMemcpyHostToDevice, stream));
            for (int k = 0; k < 3; ++k) s->bvh_ordered_for[k] = camera->origin[k];
            s->bvh_is_ordered = true;
        }
        RT_HIP(hipEventRecord(s->ev_begin, stream));
        // a slice is only written for the pixels a launch covers; unowned rows are skipped by the resolve
        int chunks_done = 0;
        for (const Launch &l : plan) {
            if (cancel.raised()) return RT_ERR_CANCEL_EVENT;
            a.sample_begin = starts[(size_t)l.first_chunk];
            a.sample_end = starts[(size_t)(l.first_chunk + l.n_chunks)];
            a.n_chunks = l.n_chunks;
            a.chunk_base = l.first_chunk;
            a.n_items = (uint32_t)a.n_chunks * (uint32_t)a.n_tiles;
            if ((uint64_t)a.n_chunks * (uint64_t)a.n_tiles >= 0x40000000ull) // the item counter's top bit is the cancel poison
                return fail(RT_ERR_UNSUPPORTED, "more than 2^30 work items in one launch");
            a.queue = s->queue.ptr + launches;
            unsigned blocks = (unsigned)(s->num_cus * (a.lens_lds ? s->pool_blocks_per_cu_lens : s->pool_blocks_per_cu));
            unsigned needed = (a.n_items + 3) / 4;
            if (blocks > needed) blocks = needed;
            RT_HIP((s->exact ? rtdev_launch_trace_pool_exact : rtdev_launch_trace_pool)(&a, s->prims_class, s->textured, s->specular, s->use_bvh, blocks, stream));
            chunks_done += a.n_chunks;
            ++launches;
        }
        RT_HIP(hipEventRecord(s->ev_traced, stream));
        if (!delivery)
            RT_HIP((s->exact ? rtdev_launch_resolve_chunks_exact : rtdev_launch_resolve_chunks)(s->partial.ptr, out_device, p->width, p->height, chunks_done, a.slice_rows, a.strip_rows,
                                               a.strip_count, a.strip_index, a.step_x, a.step_y, a.cover_w, a.cover_h,
                                               out_col_step, out_cols, p->samples, stream));
        RT_HIP(hipEventRecord(s->ev_resolved, stream));
        s->last_chunks = chunks_done;
    }
    s->last_stream = stream;
    s->has_stats = true;
    s->last_launches = launches;
    return RT_OK;
}
using rtapi::enqueue_render;

// Block until `ev` has happened; with a cancel hook, poll it meanwhile and give up
// (RT_ERR_CANCEL_EVENT) as soon as it is raised.
int rtapi::wait_event(hipEvent_t ev, const Cancel &cancel) {
    if (!cancel.armed()) {
        RT_HIP(hipEventSynchronize(ev));
        return RT_OK;
    }
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) return RT_OK;
        if (e != hipErrorNotReady) return fail(RT_ERR_HIP, std::string("hipEventQuery: ") + hipGetErrorString(e));
        (void)hipGetLastError(); // hipErrorNotReady is sticky in hipGetLastError otherwise
        if (cancel.raised()) return RT_ERR_CANCEL_EVENT;
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}

// Ends the pool launches of the current call early.  Two ways at once:
// * the scene's CANCEL WORD in pinned host memory (TraceArgs.cancel_flag), which the lane that fetches a wave's next item
//   reads: a CPU store, seen by every wave at its next item — the way that always works;
// * every item counter of the call becomes 2^31 (enqueue_render keeps a launch below 2^30 items, so no count of further
//   hand-outs wraps it), written with hipStreamWriteValue32 on a third stream behind ev_begin (which the render stream
//   records BEHIND its own clearing of the counters, so a cancel raised right after the enqueue cannot be erased).
//   Rounds 2 and 3 relied on this one alone and took it for a command-processor write; it is a small kernel of the
//   runtime's, which lands only while a SIMD has registers to spare: beside the 80-VGPR cornell variant (the one the
//   cancel test rendered) within an item's time, beside a 128-VGPR variant — four waves x 128 = the whole file — or
//   while another share's launch waits on the same device, not before the launch has ended (round 4 found it when the
//   cancel tests began to count the rays a cancelled launch had started instead of taking its time).  It still covers
//   the v1 kernel's launches and anything not yet started.
int rtapi::poison_queue_begin(RtScene *s) {
    // 1. the cancel word in pinned memory, which every wave reads with its next item: a CPU store, lands at once
    if (s->host_flags) {
        reinterpret_cast<volatile unsigned int *>(s->host_flags)[rtdev::RT_MAX_REGIONS] = 1u;
        std::atomic_thread_fence(std::memory_order_seq_cst);
    }
    // 2. the item counters themselves, for a launch that has not started yet or a caller without the word (the v1
    //    kernel's batches); this write is a small kernel of the runtime's and lands when it finds room
    RT_HIP(hipSetDevice(s->device));
    RT_HIP(hipStreamWaitEvent(s->stream_ctl, s->ev_begin, 0));
    bool by_cp = true;
    for (size_t i = 0; i < s->queue.count && by_cp; ++i)
        by_cp = hipStreamWriteValue32(s->stream_ctl, s->queue.ptr + i, 0x80000000u, 0) == hipSuccess;
    if (!by_cp) { // a runtime without stream memory operations: a fill kernel does it, once it finds room
        (void)hipGetLastError();
        RT_HIP(hipMemsetD32Async((hipDeviceptr_t)s->queue.ptr, (int)0x80000000u, s->queue.count, s->stream_ctl));
    }
    return RT_OK;
}

int rtapi::poison_queue(RtScene *s) {
    const int rc = poison_queue_begin(s);
    if (rc != RT_OK) return rc;
    RT_HIP(hipStreamSynchronize(s->stream_ctl));
    return RT_OK;
}

// ---------------------------------------------------------------------------------------------- render-buffer cache
// Everything a render call allocates on first use, kept per device across rt_scene_destroy / rt_scene_create.
// Measured on 1 x MI355X at 1080p (tools/time_scene_create.py, profiles/r04_scene_create.txt): rt_scene_create itself
// is 0.3 - 1.4 ms, but the first render of a new scene paid 3 ms of hipMalloc / hipHostMalloc (50 MB pinned frame,
// slices, counters) and the destroy before it 1 - 4 ms of hipFree / hipHostFree — on every object event of the
// reference's interactive loop.  At most two sets per device are kept (two scenes alive at a time is the pattern of
// `rebuild, then drop the old one`); rt_release_cached_buffers gives the memory back.
namespace {
struct RenderBuffers {
    int device = -1;
    DevBuf<double> partial, accum, frame;
    DevBuf<unsigned int> queue, tile_done, region_done;
    DevBuf<uint8_t> rgba;
    DevBuf<unsigned long long> segments;
    double *host_frame = nullptr;
    size_t host_frame_count = 0;
    unsigned int *host_flags = nullptr;
    uint32_t deliver_serial = 0; // the flags still hold the serials this scene published: the counter moves on
    bool deliver_dirty = false;
    hipStream_t stream = nullptr, stream_ctl = nullptr;
    hipEvent_t ev_begin = nullptr, ev_traced = nullptr, ev_resolved = nullptr;
    void free_all() {
        if (device < 0) return;
        (void)hipSetDevice(device);
        partial.release();
        accum.release();
        frame.release();
        queue.release();
        tile_done.release();
        region_done.release();
        rgba.release();
        segments.release();
        if (host_frame) (void)hipHostFree(host_frame);
        if (host_flags) (void)hipHostFree(host_flags);
        if (ev_begin) (void)hipEventDestroy(ev_begin);
        if (ev_traced) (void)hipEventDestroy(ev_traced);
        if (ev_resolved) (void)hipEventDestroy(ev_resolved);
        if (stream) (void)hipStreamDestroy(stream);
        if (stream_ctl) (void)hipStreamDestroy(stream_ctl);
        device = -1;
    }
};
std::mutex g_cache_mutex;
std::vector<RenderBuffers> g_cache;
const size_t kCachedSetsPerDevice = 2;

template <class T> void move_buf(DevBuf<T> &to, DevBuf<T> &from) {
    to.ptr = from.ptr;
    to.count = from.count;
    from.ptr = nullptr;
    from.count = 0;
}
void move_render_buffers(RenderBuffers &b, RtScene *s, bool to_scene) {
#define RT_MOVE(member) do { if (to_scene) move_buf(s->member, b.member); else move_buf(b.member, s->member); } while (0)
    RT_MOVE(partial);
    RT_MOVE(accum);
    RT_MOVE(frame);
    RT_MOVE(queue);
    RT_MOVE(tile_done);
    RT_MOVE(region_done);
    RT_MOVE(rgba);
    RT_MOVE(segments);
#undef RT_MOVE
#define RT_SWAP(member) do { if (to_scene) s->member = b.member; else b.member = s->member; } while (0)
    RT_SWAP(host_frame);
    RT_SWAP(host_frame_count);
    RT_SWAP(host_flags);
    RT_SWAP(deliver_serial);
    RT_SWAP(deliver_dirty);
    RT_SWAP(stream);
    RT_SWAP(stream_ctl);
    RT_SWAP(ev_begin);
    RT_SWAP(ev_traced);
    RT_SWAP(ev_resolved);
#undef RT_SWAP
    if (!to_scene) {
        s->host_frame = nullptr;
        s->host_frame_count = 0;
        s->host_flags = nullptr;
        s->stream = s->stream_ctl = nullptr;
        s->ev_begin = s->ev_traced = s->ev_resolved = nullptr;
    }
}
// rt_scene_destroy: the scene's render buffers go to the cache (or are freed when the device's slots are taken or the
// scene never got as far as creating its streams)
void render_cache_put(RtScene *s) {
    RenderBuffers b;
    b.device = s->device;
    move_render_buffers(b, s, false);
    const bool complete = b.stream && b.stream_ctl && b.ev_begin && b.ev_traced && b.ev_resolved && b.host_flags && b.segments.ptr;
    if (complete) {
        std::lock_guard<std::mutex> lock(g_cache_mutex);
        size_t held = 0;
        for (const RenderBuffers &c : g_cache) held += c.device == b.device;
        if (held < kCachedSetsPerDevice) {
            g_cache.push_back(b);
            return;
        }
    }
    b.free_all();
}
// rt_scene_create: take over a cached set of this device (the one with the largest slices), if there is one
bool render_cache_take(RtScene *s) {
    RenderBuffers b;
    {
        std::lock_guard<std::mutex> lock(g_cache_mutex);
        int best = -1;
        for (size_t i = 0; i < g_cache.size(); ++i)
            if (g_cache[i].device == s->device && (best < 0 || g_cache[i].partial.count > g_cache[(size_t)best].partial.count)) best = (int)i;
        if (best < 0) return false;
        b = g_cache[(size_t)best];
        g_cache.erase(g_cache.begin() + best);
    }
    move_render_buffers(b, s, true);
    return true;
}
} // namespace

extern "C" {

int rt_abi_version(void) { return RT_ABI_VERSION; }

int rt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *rt_last_error_message(void) { return g_last_error.c_str(); }

const char *rt_strerror(int code) {
    switch (code) {
    case RT_OK: return "Ok";
    case RT_ERR_FAILED_TO_CREATE_WINDOW: return "Failed to create window";
    case RT_ERR_FAILED_TO_UPDATE_WINDOW: return "Failed to update window";
    case RT_ERR_CONFIGURATION: return "Config Error";
    case RT_ERR_UNKNOWN_MATERIAL: return "Unknown Material";
    case RT_ERR_FAILED_TO_ACQUIRE_LOCK: return "Failed to acquire lock";
    case RT_ERR_EXIT_EVENT: return "Exit event";
    case RT_ERR_CANCEL_EVENT: return "Cancel event";
    case RT_ERR_IMAGE_SAVE: return "Image save error";
    case RT_ERR_SCENE_LOAD: return "Scene failed to load";
    case RT_ERR_ARGUMENT_PARSING: return "Argument parsing Error";
    case RT_ERR_KEY: return "Key callback failed";
    case RT_ERR_CREATE_LOG: return "Failed to create log";
    case RT_ERR_RECEIVE: return "Failed to recieve data";
    case RT_ERR_SEND: return "Failed to send data";
    case RT_ERR_ACTION_PROTOCOL: return "Action protocol error";
    case RT_ERR_BUS_WRITE: return "Failed to write data to bus";
    case RT_ERR_BUS_READ: return "Failed to read data from bus";
    case RT_ERR_BUS_UPDATE: return "Failed to update databus";
    case RT_ERR_BUS_TIMEOUT: return "Bus timeout error";
    case RT_ERR_NO_OBJECT_WITH_ID: return "No object with id";
    case RT_ERR_FAILED_TO_OPEN_IMAGE: return "Failed to open image";
    case RT_ERR_FAILED_TO_PARSE: return "Failed to parse into a vector";
    case RT_ERR_NO_DEVICE: return "No usable HIP device";
    case RT_ERR_HIP: return "HIP runtime error";
    case RT_ERR_INVALID_ARGUMENT: return "Invalid argument";
    case RT_ERR_UNSUPPORTED: return "Unsupported scene feature";
    case RT_ERR_OUT_OF_MEMORY: return "Out of memory";
    default: return "Unknown error";
    }
}

void rt_scene_destroy(RtScene *s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    if (s->stream_ctl) (void)hipStreamSynchronize(s->stream_ctl);
    for (uint8_t *p : s->image_pixels)
        if (p) (void)hipFree(p);
    s->prims.release();
    s->textures.release();
    s->images.release();
    s->perlins.release();
    s->bvh_nodes.release();
    s->bvh_nodes_ordered.release();
    s->bvh_prim_index.release();
    s->leaf_geo.release();
    // what a render allocates — slices, frames, pinned memory, counters, streams, events — outlives the scene: the
    // reference rebuilds its scene on every object event (main.rs:174-189), and the next rt_scene_create on this
    // device takes these over instead of paying hipMalloc / hipHostMalloc again (render_cache below)
    render_cache_put(s);
    delete s;
}

void rt_release_cached_buffers(void) {
    std::vector<RenderBuffers> all;
    {
        std::lock_guard<std::mutex> lock(g_cache_mutex);
        all.swap(g_cache);
    }
    for (RenderBuffers &b : all) b.free_all();
}

int rt_scene_create(const RtSceneDesc *d, int device, RtScene **out) { return rt_scene_create_ex(d, device, nullptr, out); }

namespace {
int scene_create(const RtSceneDesc *d, int device, const RtSceneOptions *options, RtScene **out) {
    int rc = validate_desc(d);
    if (rc != RT_OK) return rc;
    RtSceneOptions opt;
    memset(&opt, 0, sizeof opt);
    if (options) opt = *options;
    if (opt.closest_hit < RT_HIT_AUTO || opt.closest_hit > RT_HIT_BVH) return fail(RT_ERR_INVALID_ARGUMENT, "unknown closest_hit option");
    if (opt.kernel < RT_KERNEL_POOL || opt.kernel > RT_KERNEL_V1) return fail(RT_ERR_INVALID_ARGUMENT, "unknown kernel option");
    if (opt.arithmetic < RT_ARITH_FAST || opt.arithmetic > RT_ARITH_REFERENCE) return fail(RT_ERR_INVALID_ARGUMENT, "unknown arithmetic option");
    if (opt.gather < RT_GATHER_AUTO || opt.gather > RT_GATHER_STAGED) return fail(RT_ERR_INVALID_ARGUMENT, "unknown gather option");
    for (int32_t r : opt._reserved)
        if (r != 0) return fail(RT_ERR_INVALID_ARGUMENT, "reserved option fields must be 0");
    int n_dev = rt_device_count();
    if (n_dev <= 0) return fail(RT_ERR_NO_DEVICE, "no HIP device is visible to this process");
    if (device < 0 || device >= n_dev) return fail(RT_ERR_INVALID_ARGUMENT, "device index out of range");
    RT_HIP(hipSetDevice(device));

    RtScene *s = new (std::nothrow) RtScene();
    if (!s) return fail(RT_ERR_OUT_OF_MEMORY, "host allocation failed");
    s->device = device;
    s->exact = opt.arithmetic == RT_ARITH_REFERENCE;
    s->gather_staged = opt.gather == RT_GATHER_STAGED;
    struct Guard { // destroy the half-built scene on any early return
        RtScene *s;
        ~Guard() { if (s) rt_scene_destroy(s); }
    } guard{s};

    std::vector<rtdev::Prim> prims((size_t)d->n_primitives);
    for (int i = 0; i < d->n_primitives; ++i) {
        const RtPrimitive &p = d->primitives[i];
        rtdev::Prim &q = prims[(size_t)i];
        memset(&q, 0, sizeof q);
        for (int k = 0; k < 6; ++k) q.p[k] = p.p[k];
        q.rot_sin = p.rot_sin;
        q.rot_cos = p.rot_cos;
        for (int k = 0; k < 3; ++k) q.tr[k] = p.translate[k];
        q.kind = p.kind;
        q.flags = p.flags & (RT_PRIM_HAS_ROTATE_Y | RT_PRIM_HAS_TRANSLATE);
        // an absent wrapper is the identity on the device (box_t subtracts the offset unconditionally)
        if (!(q.flags & RT_PRIM_HAS_TRANSLATE)) q.tr[0] = q.tr[1] = q.tr[2] = 0.0;
        if (!(q.flags & RT_PRIM_HAS_ROTATE_Y)) {
            q.rot_sin = 0.0;
            q.rot_cos = 1.0;
        }
        q.material = p.material;
        q.inv_radius = (p.kind == RT_PRIM_SPHERE || p.kind == RT_PRIM_MOVING_SPHERE) ? 1.0 / p.p[3] : 0.0;
        q.radius2 = p.p[3] * p.p[3];
        if (p.kind == RT_PRIM_MOVING_SPHERE) { // device packing: tr = pos_b - pos_a, rot_sin = time_a, rot_cos = 1/(time_b - time_a)
            for (int k = 0; k < 3; ++k) q.tr[k] = p.center_b[k] - p.p[k];
            q.rot_sin = p.time_a;
            q.rot_cos = 1.0 / (p.time_b - p.time_a);
        }
    }
    std::vector<rtdev::Material> materials((size_t)d->n_materials);
    for (int i = 0; i < d->n_materials; ++i) {
        const RtMaterial &m = d->materials[i];
        rtdev::Material &q = materials[(size_t)i];
        memset(&q, 0, sizeof q);
        q.kind = m.kind;
        q.texture = m.texture;
        q.tex_kind = -1;
        q.fuzz = m.fuzz;
        q.ior = m.refraction_index;
        if (m.kind == RT_MAT_DIELECTRIC) { // rt_device_types.h: the per-hit quotients, once
            const double ior = m.refraction_index;
            q.color[0] = 1.0 / ior;
            const double front = (1.0 - q.color[0]) / (1.0 + q.color[0]), back = (1.0 - ior) / (1.0 + ior);
            q.color[1] = front * front;
            q.color[2] = back * back;
        }
        if (m.kind != RT_MAT_DIELECTRIC) {
            const RtTexture &t = d->textures[m.texture];
            q.tex_kind = t.kind;
            q.needs_uv = texture_reads_uv(d, m.texture) ? 1 : 0;
            for (int k = 0; k < 3; ++k) q.color[k] = t.color[k];
        }
    }
    for (rtdev::Prim &q : prims) q.mat = materials[(size_t)q.material]; // the only device copy of a material
    // What a finished sample can be at most (RtScene.radiance_bound): the product of its path's attenuations times what the
    // path ran into.  Attenuations are texture values (lambertian.rs:36, metal.rs:40) or 1 (dialectric.rs:26) — a
    // SolidColor's colour, a Noise colour times 0.5 (1 + sin) <= the colour, an image texel <= 1 — so with every such colour
    // in [0, 1] the bound is the largest of 1 (renderer.rs:48-55: white at depth 0), the emitted colours
    // (diffuse_light.rs:33-35) and the background's.  A colour outside [0, 1] on a scattering material, or anything
    // negative or not finite, leaves the scene without a bound (0): the pooled kernel then keeps f64 sums.
    {
        bool bounded = true;
        double bound = 1.0;
        auto colours_of = [&](int ti, double &hi, double &lo) { // over the texture and, for a Checkered, its two sides
            auto one = [&](const RtTexture &t) {
                if (t.kind == RT_TEX_IMAGE) {
                    hi = std::max(hi, 1.0);
                    lo = std::min(lo, 0.0);
                    return;
                }
                if (t.kind == RT_TEX_CHECKERED) return;
                for (int k = 0; k < 3; ++k) {
                    if (!std::isfinite(t.color[k])) bounded = false;
                    hi = std::max(hi, t.color[k]);
                    lo = std::min(lo, t.color[k]);
                }
            };
            const RtTexture &t = d->textures[ti];
            one(t);
            if (t.kind == RT_TEX_CHECKERED) {
                one(d->textures[t.tex_even]);
                one(d->textures[t.tex_odd]);
                // (a Checkered inside a Checkered is not evaluated further by the kernels: texture_value_deferred returns its colour field)
                for (int side : {t.tex_even, t.tex_odd})
                    if (d->textures[side].kind == RT_TEX_CHECKERED)
                        for (int k = 0; k < 3; ++k) {
                            if (!std::isfinite(d->textures[side].color[k])) bounded = false;
                            hi = std::max(hi, d->textures[side].color[k]);
                            lo = std::min(lo, d->textures[side].color[k]);
                        }
            }
        };
        for (int i = 0; i < d->n_materials; ++i) {
            const RtMaterial &m = d->materials[i];
            if (m.kind == RT_MAT_DIELECTRIC) continue;
            double hi = 0.0, lo = 0.0;
            colours_of(m.texture, hi, lo);
            if (lo < 0.0) bounded = false;
            if (m.kind == RT_MAT_DIFFUSE_LIGHT) bound = std::max(bound, hi);
            else if (hi > 1.0) bounded = false;
        }
        for (int k = 0; k < 3; ++k)
            for (double c : {d->background.top[k], d->background.bottom[k]}) {
                if (!std::isfinite(c) || c < 0.0) bounded = false;
                bound = std::max(bound, c);
            }
        s->radiance_bound = bounded && std::isfinite(bound) && bound < 0x1p40 ? bound : 0.0;
    }
    std::vector<rtdev::Image> images((size_t)d->n_images);
    s->image_pixels.assign((size_t)d->n_images, nullptr);
    for (int i = 0; i < d->n_images; ++i) {
        size_t bytes = (size_t)d->images[i].width * (size_t)d->images[i].height * 4;
        RT_HIP(hipMalloc((void **)&s->image_pixels[(size_t)i], bytes));
        RT_HIP(hipMemcpy(s->image_pixels[(size_t)i], d->images[i].rgba, bytes, hipMemcpyHostToDevice));
        images[(size_t)i].rgba = s->image_pixels[(size_t)i];
        images[(size_t)i].width = d->images[i].width;
        images[(size_t)i].height = d->images[i].height;
    }
    std::vector<rtdev::Texture> textures((size_t)d->n_textures);
    for (int i = 0; i < d->n_textures; ++i) {
        const RtTexture &t = d->textures[i];
        rtdev::Texture &q = textures[(size_t)i];
        memset(&q, 0, sizeof q);
        q.kind = t.kind;
        q.tex_even = t.tex_even;
        q.tex_odd = t.tex_odd;
        q.image = t.image;
        q.perlin = t.perlin;
        q.depth = t.depth;
        for (int k = 0; k < 3; ++k) q.color[k] = t.color[k];
        q.scale = t.scale;
        if (t.kind == RT_TEX_IMAGE) { // the device record of its image, embedded (rt_device_types.h)
            q.img.rgba = images[(size_t)t.image].rgba;
            q.img.width = images[(size_t)t.image].width;
            q.img.height = images[(size_t)t.image].height;
        }
    }
    std::vector<rtdev::Perlin> perlins((size_t)d->n_perlins);
    for (int i = 0; i < d->n_perlins; ++i) {
        static_assert(sizeof(rtdev::Perlin) == sizeof(RtPerlin), "Perlin layouts must match");
        memcpy(&perlins[(size_t)i], &d->perlins[i], sizeof(RtPerlin));
        for (int k = 0; k < 256; ++k)
            if (d->perlins[i].perm_x[k] != k || d->perlins[i].perm_y[k] != k || d->perlins[i].perm_z[k] != k) s->perlin_identity = 0;
    }
    bool only_rects = true, only_spheres = true;
    for (const rtdev::Prim &q : prims) {
        bool is_rect = q.kind == RT_PRIM_XY_RECT || q.kind == RT_PRIM_XZ_RECT || q.kind == RT_PRIM_YZ_RECT;
        if (q.flags || !is_rect) only_rects = false;
        if (q.flags || q.kind != RT_PRIM_SPHERE) only_spheres = false;
    }
    s->prims_class = only_rects ? 0 : (only_spheres ? 1 : 2);
    for (const rtdev::Prim &q : prims)
        if (q.kind == RT_PRIM_MOVING_SPHERE) s->has_moving = 1;
    for (const rtdev::Material &q : materials) {
        if (q.kind == RT_MAT_METAL || q.kind == RT_MAT_DIELECTRIC) s->specular = 1;
        if (q.kind != RT_MAT_DIELECTRIC && q.tex_kind != RT_TEX_SOLID_COLOR) s->textured = 1;
    }
    // The linear loop costs ~35 VALU instructions per primitive with scalar loads and
    // no divergence; the BVH walk ~25 node visits plus leaf tests with per-lane loads.
    // They cross at a few dozen primitives (clown.yml, 23 spheres, is still linear).
    const int kBvhThreshold = 48;
    s->use_bvh = d->n_primitives > kBvhThreshold;
    if (opt.closest_hit != RT_HIT_AUTO) s->use_bvh = opt.closest_hit == RT_HIT_BVH && d->n_primitives > 0;
    // The RT_ARITH_FAST copies of the pooled variants that keep two items in flight (any primitive kind, BVH:
    // rt_trace_pool_kernel.hip, OVERLAP) have fixed-point sums only: a scene without a radiance bound is rendered by their
    // RT_ARITH_REFERENCE copies (f64 sums, one item per wave at a time, the reference's own divisions: ~25 % slower).
    if (s->radiance_bound == 0.0 && opt.kernel == RT_KERNEL_POOL && (s->use_bvh || s->prims_class == 2)) s->exact = true;
    // the linear-loop variants keep the whole primitive table in LDS
    if (!s->use_bvh && (size_t)d->n_primitives * sizeof(rtdev::Prim) > 120 * 1024)
        return fail(RT_ERR_UNSUPPORTED, "RT_HIT_LINEAR: the primitive table does not fit in LDS");
    if (!s->use_bvh) { // linear loop: group the table (rt_device_types.h: rect_end, sphere_end); the order inside a group is kept
        std::vector<rtdev::Prim> sorted;
        sorted.reserve(prims.size());
        auto group_of = [](const rtdev::Prim &q) {
            if (q.flags == 0 && q.kind == RT_PRIM_XY_RECT) return 0;
            if (q.flags == 0 && q.kind == RT_PRIM_XZ_RECT) return 1;
            if (q.flags == 0 && q.kind == RT_PRIM_YZ_RECT) return 2;
            if (q.flags == 0 && q.kind == RT_PRIM_SPHERE) return 3;
            if (q.kind == RT_PRIM_BOX) return 4; // bare or wrapped
            return 5;
        };
        for (int g = 0; g < 6; ++g) {
            for (const rtdev::Prim &q : prims)
                if (group_of(q) == g) sorted.push_back(q);
            if (g < 3) s->rect_end[g] = (int)sorted.size();
            if (g == 3) s->sphere_end = (int)sorted.size();
            if (g == 4) s->box_end = (int)sorted.size();
        }
        prims.swap(sorted);
    }
    if (s->use_bvh) {
        // Primitives per leaf.  A leaf primitive costs a lane four times what a node costs (its record comes from
        // global memory, the node from LDS; `random`: 40 % of the walk for 5.8 tests against 26.5 nodes per
        // segment), so leaves of three beat leaves of four (66.1 -> 62.1 ms) — as long as the larger node array
        // does not cost the variant a block per CU (leaves of two: 70.6 ms with three blocks instead of four).
        const size_t lds_other = (s->textured && d->n_perlins > 0 && s->perlin_identity ? sizeof(double) * 256 * 3 : 0) +
                                 (s->has_moving ? rtdev::pool_time_lds_bytes(true) : 0);
        auto blocks_with = [&](const rtdev::BvhBuild &b) {
            const size_t bytes = b.nodes.size() * sizeof(rtdev::BvhNode);
            return (s->exact ? rtdev_pool_blocks_per_cu_exact : rtdev_pool_blocks_per_cu)(s->prims_class, s->textured, s->specular, 1, (bytes <= 32 * 1024 ? bytes : 0) + lds_other);
        };
        // (more than 2048 primitives: at most four to a leaf, the node array cannot fit LDS — the direction-ordered copies are
        // wanted, built in the same pass)
        const bool surely_large = d->n_primitives > 2048;
        rtdev::BvhBuild bvh = rtdev::build_bvh(d->primitives, d->n_primitives, 4, surely_large);
        int max_leaf = 0;
#ifdef RT_DEVELOPER_KNOBS
        if (const char *k = getenv("RT_BVH_LEAF")) max_leaf = atoi(k);
#endif
        if (max_leaf > 0) {
            bvh = rtdev::build_bvh(d->primitives, d->n_primitives, max_leaf);
        } else if (bvh.nodes.size() * sizeof(rtdev::BvhNode) <= 32 * 1024) { // the nodes live in LDS
            rtdev::BvhBuild three = rtdev::build_bvh(d->primitives, d->n_primitives, 3);
            if (three.nodes.size() * sizeof(rtdev::BvhNode) <= 32 * 1024 && blocks_with(three) == blocks_with(bvh)) bvh = std::move(three);
        }
        if (bvh.nodes.size() * sizeof(rtdev::BvhNode) > 32 * 1024) {
            // the nodes stay in global memory, where a step is two dependent loads and every node not visited counts: the
            // eight direction-ordered copies (rt_bvh.cpp; 8 x 32 B per node)
            bool ordered = true;
#ifdef RT_DEVELOPER_KNOBS
            if (const char *k = getenv("RT_BVH_ORDERED")) ordered = atoi(k) != 0;
#endif
            if (ordered) {
                if (bvh.ordered.empty()) bvh = rtdev::build_bvh(d->primitives, d->n_primitives, max_leaf > 0 ? max_leaf : 4, true);
                if ((rc = upload(s->bvh_nodes_ordered, bvh.ordered)) != RT_OK) return rc;
            }
        }
        if ((rc = upload(s->bvh_nodes, bvh.nodes)) != RT_OK) return rc;
        if (bvh.nodes.size() * sizeof(rtdev::BvhNode) <= 32 * 1024) { // the nodes live in LDS: kept for the per-camera child order (enqueue_render)
            s->bvh_host.nodes = bvh.nodes;
            for (int k = 0; k < 3; ++k) s->bvh_host.center[k] = bvh.center[k];
        }
        if ((rc = upload(s->bvh_prim_index, bvh.prim_index)) != RT_OK) return rc;
        s->n_bvh_nodes = (int)bvh.nodes.size() - 1; // the array ends with the sentinel (rt_device_types.h: BvhNode)
        for (int k = 0; k < 3; ++k) {
            s->bvh_root_mn[k] = bvh.root_mn[k];
            s->bvh_root_mx[k] = bvh.root_mx[k];
            s->bvh_center[k] = bvh.center[k];
        }
        // the device table is stored in leaf order, so a leaf is a contiguous run of records
        std::vector<rtdev::Prim> ordered(prims.size());
        for (size_t j = 0; j < bvh.prim_index.size(); ++j) ordered[j] = prims[(size_t)bvh.prim_index[j]];
        prims.swap(ordered);
        // the compact records the walk tests leaves with; the first MovingSphere sets the scene-wide time interval
        std::vector<rtdev::LeafGeo> geo(prims.size());
        bool have_interval = false;
        for (size_t j = 0; j < prims.size(); ++j) {
            const rtdev::Prim &q = prims[j];
            rtdev::LeafGeo &g = geo[j];
            memset(&g, 0, sizeof g);
            g.tag = 1;
            if (q.flags != 0 || (q.kind != RT_PRIM_SPHERE && q.kind != RT_PRIM_MOVING_SPHERE)) continue;
            if (q.kind == RT_PRIM_MOVING_SPHERE) {
                // a MovingSphere with time_a == time_b degenerates by itself in the reference (moving_sphere.rs:37-39:
                // 0/0); its 1 / (time_b - time_a) = inf must not become the scene-wide interval, where it would turn
                // the centre of every plain Sphere (dc = 0) into inf * 0 = NaN: it keeps the general path (tag 1)
                if (!std::isfinite(q.rot_cos)) continue;
                if (!have_interval) {
                    s->leaf_time_a = q.rot_sin;
                    s->leaf_inv_dt = q.rot_cos;
                    have_interval = true;
                }
                if (q.rot_sin != s->leaf_time_a || q.rot_cos != s->leaf_inv_dt) continue; // another interval: general path
                for (int k = 0; k < 3; ++k) g.dc[k] = q.tr[k];
            }
            for (int k = 0; k < 3; ++k) g.c0[k] = q.p[k];
            g.radius2 = q.radius2;
            g.tag = 0;
        }
        if ((rc = upload(s->leaf_geo, geo)) != RT_OK) return rc;
    }
    if ((rc = upload(s->prims, prims)) != RT_OK) return rc;
    if ((rc = upload(s->textures, textures)) != RT_OK) return rc;
    if ((rc = upload(s->images, images)) != RT_OK) return rc;
    if ((rc = upload(s->perlins, perlins)) != RT_OK) return rc;
    s->n_prims = d->n_primitives;
    s->n_materials = d->n_materials;
    s->n_textures = d->n_textures;
    s->n_images = d->n_images;
    s->n_perlins = d->n_perlins;
    s->bg.kind = d->background.kind;
    for (int k = 0; k < 3; ++k) {
        s->bg.top[k] = d->background.top[k];
        s->bg.bottom[k] = d->background.bottom[k];
    }
    s->use_v1 = opt.kernel == RT_KERNEL_V1;
    RT_HIP(hipDeviceGetAttribute(&s->num_cus, hipDeviceAttributeMultiprocessorCount, device));
    s->bvh_nodes_in_lds = s->use_bvh && (size_t)(s->n_bvh_nodes + 1) * sizeof(rtdev::BvhNode) <= 32 * 1024;
#ifdef RT_DEVELOPER_KNOBS
    if (const char *k = getenv("RT_BVH_LDS")) s->bvh_nodes_in_lds = s->bvh_nodes_in_lds && atoi(k) != 0;
#endif
    // dynamic LDS of the variant: the BVH node array, or the primitive table of the linear-loop variants
    const size_t dyn_lds = (s->use_bvh ? (s->bvh_nodes_in_lds ? (size_t)(s->n_bvh_nodes + 1) * sizeof(rtdev::BvhNode) : 0)
                                       : (size_t)s->n_prims * sizeof(rtdev::Prim) + (s->textured ? (size_t)s->n_textures * sizeof(rtdev::Texture) : 0)) +
                           (s->textured && s->n_perlins > 0 && s->perlin_identity ? sizeof(double) * 256 * 3 : 0) +
                           (s->has_moving ? rtdev::pool_time_lds_bytes(s->use_bvh != 0) : 0);
    s->pool_blocks_per_cu = (s->exact ? rtdev_pool_blocks_per_cu_exact : rtdev_pool_blocks_per_cu)(s->prims_class, s->textured, s->specular, s->use_bvh, dyn_lds);
    s->pool_blocks_per_cu_lens = (s->exact ? rtdev_pool_blocks_per_cu_exact : rtdev_pool_blocks_per_cu)(s->prims_class, s->textured, s->specular, s->use_bvh,
                                                          dyn_lds + rtdev::pool_lens_lds_bytes(s->use_bvh != 0));
#ifdef RT_DEVELOPER_KNOBS // occupancy experiments
    if (const char *k = getenv("RT_POOL_BLOCKS_PER_CU"))
        if (atoi(k) > 0) s->pool_blocks_per_cu = s->pool_blocks_per_cu_lens = atoi(k);
#endif
    if (!render_cache_take(s)) { // nothing of this device's to take over: the first scene, or more than the cache holds
        RT_HIP(s->segments.alloc(rtdev::RT_STAT_SLOTS)); // rt_device_types.h: RT_STAT_*
        RT_HIP(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
        RT_HIP(hipEventCreate(&s->ev_begin));
        RT_HIP(hipEventCreate(&s->ev_traced));
        RT_HIP(hipEventCreate(&s->ev_resolved));
        RT_HIP(hipHostMalloc((void **)&s->host_flags, (rtdev::RT_MAX_REGIONS + 1) * sizeof(unsigned int),
                             hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent));
        memset(s->host_flags, 0, (rtdev::RT_MAX_REGIONS + 1) * sizeof(unsigned int));
        RT_HIP(hipStreamCreateWithFlags(&s->stream_ctl, hipStreamNonBlocking));
    }
    RT_HIP(hipMemsetAsync(s->segments.ptr, 0, rtdev::RT_STAT_SLOTS * sizeof(unsigned long long), s->stream));
    guard.s = nullptr;
    *out = s;
    return RT_OK;
}
} // namespace

int rt_scene_create_ex(const RtSceneDesc *d, int device, const RtSceneOptions *options, RtScene **out) {
    if (!out) return fail(RT_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    try { // nothing may unwind through the C ABI (std::vector / std::string / the BVH build allocate)
        return scene_create(d, device, options, out);
    } catch (const std::bad_alloc &) {
        return fail(RT_ERR_OUT_OF_MEMORY, "host allocation failed while building the scene");
    } catch (const std::exception &e) {
        return fail(RT_ERR_INVALID_ARGUMENT, std::string("scene build failed: ") + e.what());
    }
}

int rt_render_frame_device(RtScene *s, const RtCamera *camera, const RtRenderParams *p, double *out_dev,
                           void *hip_stream) {
    if (!s || !out_dev) return fail(RT_ERR_INVALID_ARGUMENT, "scene/out is NULL");
    int rc = check_params(camera, p);
    if (rc != RT_OK) return rc;
    return enqueue_render(s, camera, p, out_dev, (hipStream_t)hip_stream, 0, Cancel());
}

int rt_post_rgba8_device(RtScene *s, const RtToneMap *tm, const double *rgb_device, size_t n_pixels,
                         uint8_t *rgba_device, double *mapped_device, void *hip_stream) {
    if (!s || !tm || !rgb_device || !rgba_device) return fail(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    if (tm->kind < RT_TM_NONE || tm->kind > RT_TM_ACES) return fail(RT_ERR_INVALID_ARGUMENT, "unknown tone map kind");
    RT_HIP(hipSetDevice(s->device));
    RT_HIP(rtdev_launch_post_rgba8(tm, rgb_device, n_pixels, rgba_device, mapped_device, (hipStream_t)hip_stream));
    return RT_OK;
}

int rt_render_frame_rgba8(RtScene *s, const RtCamera *camera, const RtRenderParams *p, const RtToneMap *tm,
                          uint8_t *out_rgba) {
    if (!s || !tm || !out_rgba) return fail(RT_ERR_INVALID_ARGUMENT, "scene/tone_map/out is NULL");
    int rc = check_params(camera, p);
    if (rc != RT_OK) return rc;
    if (p->strip_count > 1) // the packed frame is a whole picture; gather strips with rt_render_frame_device, then rt_post_rgba8_device
        return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_frame_rgba8 packs the whole frame: strip ownership is not supported here");
    RT_HIP(hipSetDevice(s->device));
    const size_t px = (size_t)p->width * (size_t)p->height;
    if (s->frame.count < px * 3) RT_HIP(s->frame.alloc(px * 3));
    if (s->rgba.count < px * 4) RT_HIP(s->rgba.alloc(px * 4));
    rc = enqueue_render(s, camera, p, s->frame.ptr, s->stream, 0, Cancel());
    if (rc != RT_OK) return rc;
    rc = rt_post_rgba8_device(s, tm, s->frame.ptr, px, s->rgba.ptr, nullptr, s->stream);
    if (rc != RT_OK) return rc;
    RT_HIP(hipStreamSynchronize(s->stream));
    RT_HIP(hipMemcpy(out_rgba, s->rgba.ptr, px * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_scene_last_stats(RtScene *s, RtRenderStats *out) {
    if (!s || !out) return fail(RT_ERR_INVALID_ARGUMENT, "scene/out is NULL");
    memset(out, 0, sizeof *out);
    if (!s->has_stats) return RT_OK;
    RT_HIP(hipSetDevice(s->device));
    RT_HIP(hipEventSynchronize(s->ev_resolved));
    float ms_trace = 0.f, ms_resolve = 0.f;
    RT_HIP(hipEventElapsedTime(&ms_trace, s->ev_begin, s->ev_traced));
    RT_HIP(hipEventElapsedTime(&ms_resolve, s->ev_traced, s->ev_resolved));
    unsigned long long counters[rtdev::RT_STAT_SLOTS]; // rt_device_types.h: RT_STAT_*
    RT_HIP(hipMemcpy(counters, s->segments.ptr, sizeof counters, hipMemcpyDeviceToHost));
    const unsigned long long segs = counters[rtdev::RT_STAT_SEGMENTS];
#ifdef RT_PROFILE_REGIONS
    {
        const unsigned long long *c = counters;
        static const char *names[16] = {"item setup", "batches", "hand-out + primary ray", "closest hit", "miss / material",
                                        "sampler", "scatter + accumulate", "item end", "hit record", "texture, step 1",
                                        "Noise rounds", "BVH: descent to a leaf", "BVH: leaf primitives", "-", "-", "-"};
        double total = 0;
        for (int k = 0; k < 16; ++k) total += (double)c[rtdev::RT_STAT_REGIONS + k];
        for (int k = 0; k < 13; ++k)
            if (k < 11 || c[rtdev::RT_STAT_REGIONS + k])
            fprintf(stderr, "region %-24s %6.2f %%  (%.3g wave-cycles, %4.1f lanes active at its closing marker)\n", names[k],
                    100.0 * (double)c[rtdev::RT_STAT_REGIONS + k] / total, (double)c[rtdev::RT_STAT_REGIONS + k],
                    c[rtdev::RT_STAT_REGIONS + k] ? (double)c[rtdev::RT_STAT_REGION_LANES + k] / (double)c[rtdev::RT_STAT_REGIONS + k] : 0.0);
        if (c[rtdev::RT_STAT_NOISE])
            fprintf(stderr, "region noise lookups: %.3g wave-iterations with one, %.1f lanes each on average\n", (double)c[rtdev::RT_STAT_NOISE],
                    (double)c[rtdev::RT_STAT_NOISE + 1] / (double)c[rtdev::RT_STAT_NOISE]);
        if (c[rtdev::RT_STAT_NOISE + 2])
            fprintf(stderr, "region BVH walk: %.1f nodes visited and %.1f leaf primitives tested per segment\n",
                    (double)c[rtdev::RT_STAT_NOISE + 2] / (double)segs, (double)c[rtdev::RT_STAT_NOISE + 3] / (double)segs);
        // lanes tracing per iteration: while the pool has paths to hand out / in the item's tail
        for (int part = 0; part < 2; ++part) {
            const unsigned long long *h = c + (part ? rtdev::RT_STAT_LANES_TAIL : rtdev::RT_STAT_LANES_BODY);
            double iters = 0, lanes = 0;
            for (int k = 0; k < 9; ++k) {
                iters += (double)h[k];
                lanes += (double)h[k] * (k == 0 ? 0.0 : 8.0 * k - 3.5); // bin centre
            }
            fprintf(stderr, "region lanes tracing, %s: %.4g iterations, mean %.1f lanes; bins 0|1-8|..|57-64:", part ? "item tail (pool dry)" : "pool not dry  ",
                    iters, iters > 0 ? lanes / iters : 0.0);
            for (int k = 0; k < 9; ++k) fprintf(stderr, " %.1f%%", iters > 0 ? 100.0 * (double)h[k] / iters : 0.0);
            fprintf(stderr, "\n");
        }
        // 100 MHz wall clock: when did the first/last wave start and end (last launch of the call)
        const unsigned long long *w = c + rtdev::RT_STAT_WALL;
        fprintf(stderr, "region waves: last start +%.3f ms, first end +%.3f ms, last end +%.3f ms after the first start\n",
                (double)(w[1] - w[0]) * 1e-5, (double)(w[2] - w[0]) * 1e-5, (double)(w[3] - w[0]) * 1e-5);
    }
#endif
    out->samples = counters[rtdev::RT_STAT_SAMPLES]; // counted on the device where a path is handed out
    out->segments = segs;
    out->kernel_ms = ms_trace;
    out->resolve_ms = ms_resolve;
    out->kernel_launches = s->last_launches;
    return RT_OK;
}

} // extern "C"
