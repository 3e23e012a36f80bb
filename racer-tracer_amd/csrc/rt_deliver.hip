// rt_deliver.hip — the entry points that hand finished pixels to the HOST: rt_render_frame, rt_render /
// rt_render_ex (the reference's tile stream) and their several-device forms rt_render_frame_multi / rt_render_multi.
//
// What the reference does here: CpuRenderer::render shards the frame into tiles, every tile is traced on a rayon
// worker and sent to the writer the moment it is finished (racer-tracer/src/renderer/cpu.rs:64-70,118-131), with
// do_cancel polled per tile row (cpu.rs:55, renderer.rs:25-30).
//
// Here ONE persistent launch per device renders the whole frame (its share of it with several devices) and
// DELIVERS ITS OWN PIXELS (rt_device_types.h: TraceArgs.deliver_out): the items are queued region by region — a
// region is a tile column of the stream, or a band of rows of a whole-frame call — the wave that completes the last
// sample chunk of an 8x8 item tile sums the tile's slices and writes sqrt(sum / samples) straight into pinned host
// memory (mapped into every device, laid out so that a tile of the stream is one contiguous run), and the wave that
// completes a region publishes it in a host-visible flag.  The calling thread only polls flags, runs the callbacks
// (or copies finished bands into the caller's frame) while the GPU works on the next region, and polls the cancel
// hook.  No resolve launch, no copy engine, no second stream: nothing has to find a free compute unit beside the
// persistent grid, which is what the round-2 form (ten windowed launches on two streams + a strided copy per column)
// paid 15 % of the frame for.  Pixels are bit-identical to the two-pass path (same slices, same sum order).
//
// Host code only.
#include <hip/hip_runtime.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>
#include "rt_scene.h"

using rtapi::Cancel;
using rtapi::Delivery;
using rtapi::fail;

namespace {

inline int up8(int x) { return (x + 7) & ~7; }

// One device's part of a delivering call.
struct Share {
    RtScene *scene = nullptr;
    RtRenderParams params;
    Delivery delivery;
    int next_region = 0; // regions [0, next_region) have been seen published
    bool launched = false;
};

// Pinned frame + flags of the call live on shares[0]'s scene.
int ensure_host_frame(RtScene *s, size_t doubles) {
    if (s->host_frame_count >= doubles) return RT_OK;
    RT_HIP(hipSetDevice(s->device));
    if (s->host_frame) (void)hipHostFree(s->host_frame);
    s->host_frame = nullptr;
    s->host_frame_count = 0;
    RT_HIP(hipHostMalloc((void **)&s->host_frame, doubles * sizeof(double),
                         hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent));
    s->host_frame_count = doubles;
    return RT_OK;
}

// Blocks until region `r` of `sh` has been published.  RT_ERR_CANCEL_EVENT when the hook is raised first.
int wait_region(Share &sh, int r, const Cancel &cancel) {
    RtScene *s = sh.scene;
    const volatile unsigned int *flags = s->host_flags;
    const uint32_t serial = sh.delivery.serial;
    for (unsigned spins = 1;; ++spins) {
        if (flags[r] == serial) {
            std::atomic_thread_fence(std::memory_order_acquire);
            return RT_OK;
        }
        if (cancel.raised()) return RT_ERR_CANCEL_EVENT;
        if ((spins & 127u) == 0) { // has the launch ended (or failed) without publishing?
            (void)hipSetDevice(s->device);
            const hipError_t e = hipEventQuery(s->ev_traced);
            if (e == hipSuccess) {
                if (flags[r] == serial) continue;
                return fail(RT_ERR_HIP, "the launch ended without publishing region " + std::to_string(r));
            }
            if (e != hipErrorNotReady) return fail(RT_ERR_HIP, std::string("hipEventQuery: ") + hipGetErrorString(e));
            (void)hipGetLastError();
        }
        if (spins < 64u) std::this_thread::yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
}

// Ends everything in flight (cancel or error) and leaves the scenes reusable.
void abort_shares(std::vector<Share> &shares) {
    for (Share &sh : shares) // every share's poison is on its way before any of them is waited for
        if (sh.launched) (void)rtapi::poison_queue_begin(sh.scene);
    for (Share &sh : shares)
        if (sh.launched) {
            (void)hipSetDevice(sh.scene->device);
            (void)hipStreamSynchronize(sh.scene->stream_ctl);
            (void)hipStreamSynchronize(sh.scene->stream);
        }
    (void)hipGetLastError();
}

int finish_shares(std::vector<Share> &shares) {
    int rc = RT_OK;
    for (Share &sh : shares) {
        if (!sh.launched) continue;
        if (hipSetDevice(sh.scene->device) != hipSuccess || hipStreamSynchronize(sh.scene->stream) != hipSuccess) {
            if (rc == RT_OK) rc = fail(RT_ERR_HIP, "stream synchronisation failed");
        } else {
            sh.scene->deliver_dirty = false; // every region was published: the counters are back at zero
        }
    }
    return rc;
}

int check_scenes(RtScene *const *scenes, int n) {
    if (!scenes || n <= 0) return fail(RT_ERR_INVALID_ARGUMENT, "no scenes");
    for (int i = 0; i < n; ++i)
        if (!scenes[i]) return fail(RT_ERR_INVALID_ARGUMENT, "scenes[" + std::to_string(i) + "] is NULL");
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j)
            if (scenes[i] == scenes[j]) return fail(RT_ERR_INVALID_ARGUMENT, "the same RtScene is listed twice (create one per share)");
    return RT_OK;
}

// shares[i].params: the caller's parameters for one scene, strips dealt out for several
int make_shares(RtScene *const *scenes, int n, const RtRenderParams *p, int strip_rows, std::vector<Share> &shares) {
    if (n > 1) {
        if (p->strip_count > 1) return fail(RT_ERR_INVALID_ARGUMENT, "params->strip_* must be unset: the call assigns strips itself");
        if (p->scale > 1) return fail(RT_ERR_INVALID_ARGUMENT, "the preview scale cannot be combined with strips");
        if (strip_rows < 0) return fail(RT_ERR_INVALID_ARGUMENT, "strip_rows must not be negative");
        if (strip_rows == 0) strip_rows = 8;
    }
    shares.assign((size_t)n, Share());
    for (int i = 0; i < n; ++i) {
        Share &sh = shares[(size_t)i];
        sh.scene = scenes[i];
        sh.params = *p;
        if (n > 1) {
            sh.params.strip_rows = strip_rows;
            sh.params.strip_count = n;
            sh.params.strip_index = i;
        }
    }
    return RT_OK;
}

int launch_shares(std::vector<Share> &shares, const RtCamera *camera, bool cancellable = false) {
    for (Share &sh : shares) { // every allocation of the call before its first launch (rt_api.hip: reserve_render_buffers)
        if (sh.delivery.regions.empty()) continue;
        const int rc = rtapi::reserve_render_buffers(sh.scene, &sh.params, true);
        if (rc != RT_OK) return rc;
    }
    for (Share &sh : shares) {
        sh.delivery.cancellable = cancellable;
        // A share that owns no rows (more shares than strips: height 16 over three scenes; a strip_index whose first
        // strip lies below the image) has nothing to launch and nothing to wait for: it counts as published, and the
        // rows of the frame stay as they are — what the two-pass path does with n_items == 0.
        if (sh.delivery.regions.empty()) continue;
        sh.delivery.serial = ++sh.scene->deliver_serial;
        if (sh.delivery.serial == 0) sh.delivery.serial = ++sh.scene->deliver_serial; // 0 is the flags' idle value
        const int rc = rtapi::enqueue_render(sh.scene, camera, &sh.params, nullptr, sh.scene->stream, 0, Cancel(), &sh.delivery);
        if (rc != RT_OK) return rc;
        sh.launched = true;
    }
    return RT_OK;
}

// ------------------------------------------------------------------------------------------------ whole frames
// Regions = bands of tile rows; the calling thread copies a band's rows from the pinned frame into the caller's
// (pageable) frame while the GPU renders the next band, so only the last band's copy is exposed.
int deliver_frame(RtScene *const *scenes, int n, const RtCamera *camera, const RtRenderParams *p, int strip_rows,
                  double *out_rgb) {
    std::vector<Share> shares;
    int rc = make_shares(scenes, n, p, strip_rows, shares);
    if (rc != RT_OK) return rc;
    const size_t row_doubles = (size_t)p->width * 3;
    rc = ensure_host_frame(scenes[0], row_doubles * (size_t)p->height);
    if (rc != RT_OK) return rc;
    const int tiles_x = (p->width + 7) / 8;
    int most_bands = 0;
    for (Share &sh : shares) {
        const int tile_rows = (rtapi::owned_rows_of(&sh.params) + 7) / 8;
        // How many bands?  The launch ends on the LAST band's last chunks — (tiles of the band) x (a few chunks) items for
        // thousands of resident waves — so a small last band costs an unbalanced tail of about one item's duration,
        // while a large one costs the copy of its rows after the GPU is done.  Items are short when a frame has many
        // chunks (C3: 19 chunks, 0.2 ms items: 32 bands, +0.8 ms in all) and long when it has few (64 spp of a
        // 485-sphere scene: 4 chunks, 1.6 ms items: 32 bands cost +3.5 ms, 8 bands +0.7): two bands per chunk.
        int bands = tile_rows / 2; // at least two tile rows each
        const int by_chunks = 2 * rtapi::chunk_count(sh.params.samples);
        if (bands > by_chunks) bands = by_chunks < 4 ? 4 : by_chunks;
        if (bands > rtdev::RT_MAX_REGIONS) bands = rtdev::RT_MAX_REGIONS;
        if (bands > tile_rows) bands = tile_rows;
        if (bands < 1) bands = 1;
        for (int b = 0; b < bands; ++b) {
            rtdev::Region reg;
            reg.item_begin = 0;
            reg.tx0 = 0;
            reg.ntx = tiles_x;
            reg.ty0 = (int)((long long)tile_rows * b / bands);
            reg.nty = (int)((long long)tile_rows * (b + 1) / bands) - reg.ty0;
            if (reg.nty > 0) sh.delivery.regions.push_back(reg);
        }
        sh.delivery.out = scenes[0]->host_frame;
        sh.delivery.col_step = p->width;
        sh.delivery.cols = 1;
        most_bands = (int)sh.delivery.regions.size() > most_bands ? (int)sh.delivery.regions.size() : most_bands;
    }
    rc = launch_shares(shares, camera);
    for (int b = 0; b < most_bands && rc == RT_OK; ++b)
        for (Share &sh : shares) {
            if (b >= (int)sh.delivery.regions.size()) continue;
            rc = wait_region(sh, b, Cancel());
            if (rc != RT_OK) break;
            const rtdev::Region &reg = sh.delivery.regions[(size_t)b];
            const int owned = rtapi::owned_rows_of(&sh.params);
            int vr = reg.ty0 * 8, vr_end = (reg.ty0 + reg.nty) * 8;
            if (vr_end > owned) vr_end = owned;
            while (vr < vr_end) { // runs of consecutive image rows (a strip, or the whole band without strips)
                const int row = rtapi::owned_row_to_image_row(&sh.params, vr);
                int run = 1;
                while (vr + run < vr_end && rtapi::owned_row_to_image_row(&sh.params, vr + run) == row + run) ++run;
                if (row < p->height) {
                    const int rows = row + run <= p->height ? run : p->height - row;
                    memcpy(out_rgb + (size_t)row * row_doubles, scenes[0]->host_frame + (size_t)row * row_doubles,
                           (size_t)rows * row_doubles * sizeof(double));
                }
                vr += run;
            }
        }
    if (rc != RT_OK) {
        abort_shares(shares);
        return rc;
    }
    return finish_shares(shares);
}

// ------------------------------------------------------------------------------------------------- tile stream
int deliver_tiles(RtScene *const *scenes, int n, const RtCamera *camera, const RtRenderParams *p, int strip_rows,
                  RtTileCallback callback, void *user, const Cancel &cancel) {
    std::vector<Share> shares;
    int rc = make_shares(scenes, n, p, strip_rows, shares);
    if (rc != RT_OK) return rc;
    // cpu.rs:73-115 tile grid, column-major, remainders in the last row/column
    const int width_step = p->width / p->tiles_w, height_step = p->height / p->tiles_h;
    auto column_x = [&](int ws) { return width_step * ws; };
    auto column_w = [&](int ws) { return ws == p->tiles_w - 1 ? p->width - width_step * ws : width_step; };
    rc = ensure_host_frame(scenes[0], (size_t)p->width * (size_t)p->height * 3);
    if (rc != RT_OK) return rc;
    const double *frame = scenes[0]->host_frame;
    // One callback per tile of tile column `ws`, top to bottom.  The pinned frame holds column ws as
    // [height][w][3] behind the columns before it, so a tile is a contiguous run of it.
    auto emit_column = [&](int ws) {
        const int x = column_x(ws), w = column_w(ws);
        const double *col = frame + (size_t)p->height * (size_t)x * 3;
        for (int hs = 0; hs < p->tiles_h; ++hs) {
            if (cancel.raised()) return false;
            const int y = height_step * hs;
            const int h = hs == p->tiles_h - 1 ? p->height - y : height_step;
            // (an empty tile — more tile rows than image rows — is still ONE BufferUpdate, as in the reference, whose
            // raytrace sends the buffer of every SubImage, empty or not: cpu.rs:64-70)
            callback(user, col + (size_t)y * (size_t)(w > 0 ? w : 0) * 3, y, x, w > 0 ? w : 0, h > 0 ? h : 0);
        }
        return true;
    };
    // A pixel's sum depends on the order its samples meet in LDS, i.e. on which 8x8 item tile it sits in, so the
    // regions are cut on the whole-frame item grid: the window of tile columns [a, b) is the pixel range
    // [up8(x_a), up8(x_b)) (first from 0, last to the image edge) and holds exactly the item tiles of the
    // whole-frame render; tile column k is complete once the windows up to its own are (up8(x_k+1) >= x_k+1).
    // More than RT_MAX_REGIONS tile columns share regions.  A window may be empty.
    auto window_begin = [&](int ws) {
        if (ws <= 0) return 0;
        if (ws >= p->tiles_w) return p->width;
        const int x = up8(column_x(ws));
        return x < p->width ? x : p->width;
    };
    const int per_region = (p->tiles_w + rtdev::RT_MAX_REGIONS - 1) / rtdev::RT_MAX_REGIONS;
    std::vector<int> columns_after; // columns_after[r] = tile columns [.., this) are complete once region r is
    std::vector<rtdev::Region> regions;
    for (int a = 0; a < p->tiles_w; a += per_region) {
        const int b = a + per_region < p->tiles_w ? a + per_region : p->tiles_w;
        const int x0 = window_begin(a), x1 = window_begin(b);
        if (x1 > x0) {
            rtdev::Region reg;
            reg.item_begin = 0;
            reg.tx0 = x0 / 8;
            reg.ntx = (x1 + 7) / 8 - reg.tx0;
            reg.ty0 = 0;
            reg.nty = 0; // per share below
            regions.push_back(reg);
            columns_after.push_back(b);
        } else if (!columns_after.empty()) {
            columns_after.back() = b; // nothing of its own to wait for
        }
    }
    if (regions.empty()) return fail(RT_ERR_INVALID_ARGUMENT, "empty frame");
    columns_after.back() = p->tiles_w;
    for (Share &sh : shares) {
        const int tile_rows = (rtapi::owned_rows_of(&sh.params) + 7) / 8;
        if (tile_rows > 0) sh.delivery.regions = regions; // (none: the share owns no rows and is not launched)
        for (rtdev::Region &reg : sh.delivery.regions) reg.nty = tile_rows;
        sh.delivery.out = scenes[0]->host_frame;
        sh.delivery.col_step = width_step;
        sh.delivery.cols = p->tiles_w;
    }
    rc = launch_shares(shares, camera, cancel.armed());
    bool cancelled = false;
    int emitted = 0;
    for (size_t r = 0; r < regions.size() && rc == RT_OK && !cancelled; ++r) {
        for (Share &sh : shares) {
            if (!sh.launched) continue; // owns no rows
            rc = wait_region(sh, (int)r, cancel);
            if (rc != RT_OK) break;
        }
        if (rc != RT_OK) break;
        for (; emitted < columns_after[r] && !cancelled; ++emitted) cancelled = !emit_column(emitted);
    }
    if (rc == RT_ERR_CANCEL_EVENT) {
        cancelled = true;
        rc = RT_OK;
    }
    if (rc != RT_OK || cancelled) { // cpu.rs:55-62: a cancelled render returns Ok(()), nothing further is written
        abort_shares(shares);
        return rc;
    }
    return finish_shares(shares);
}

// The two-pass path behind the tile stream where the delivering launch does not apply (preview scale, the v1 kernel, a
// tile grid wider than the image): the tiles are cut from the finished frame.  The pooled kernel's resolve pass writes
// the stream's tile-column layout itself (k_resolve_chunks_f64), ONE copy brings the frame into the scene's pinned
// buffer and the callbacks read it in place — the reference's interactive mode renders a preview on every camera move
// (interactive.rs:196-267), so what this path costs beyond the 0.5 ms of device work is the frame rate of the window
// (1080p preview: 14.3 ms with a pageable frame, a zero-initialised staging vector and a host-side repack; now 1.44).
// The v1 kernel is traced in sample batches with a synchronisation after each, so that the hook is polled about as
// often as the reference polls it per tile row (cpu.rs:55), and its plain frame is repacked per column on the host.
int tiles_from_frame(RtScene *s, const RtCamera *camera, const RtRenderParams *p, RtTileCallback callback, void *user,
                     const Cancel &cancel) {
    const size_t n = (size_t)p->width * (size_t)p->height * 3;
    if (s->frame.count < n) RT_HIP(s->frame.alloc(n));
    int rc = ensure_host_frame(s, n);
    if (rc != RT_OK) return rc;
    int batch = 0;
    if (cancel.armed() && s->use_v1) { // at most 32 launches, at least 16 samples each
        batch = (p->samples + 31) / 32;
        if (batch < 16) batch = 16;
        if (batch > p->samples) batch = p->samples;
    }
    const int width_step = p->width / p->tiles_w, height_step = p->height / p->tiles_h;
    const bool column_layout = !s->use_v1 && width_step > 0 && p->tiles_w > 1; // written by the resolve pass itself
    rc = rtapi::enqueue_render(s, camera, p, s->frame.ptr, s->stream, batch, cancel, nullptr, column_layout ? width_step : 0,
                               column_layout ? p->tiles_w : 1);
    hipEvent_t copied = s->ev_resolved; // re-recorded behind the copy: rt_scene_last_stats reads ev_traced -> ev_resolved (+ the copy)
    if (rc == RT_OK) {
        hipError_t e = hipMemcpyAsync(s->host_frame, s->frame.ptr, n * sizeof(double), hipMemcpyDeviceToHost, s->stream);
        if (e == hipSuccess) e = hipEventRecord(copied, s->stream);
        if (e != hipSuccess) { // the kernels are in flight on the scene's buffers: drain before the error goes up
            (void)hipStreamSynchronize(s->stream);
            return fail(RT_ERR_HIP, std::string("tiles_from_frame: ") + hipGetErrorString(e));
        }
    }
    if (rc == RT_OK) rc = rtapi::wait_event(copied, cancel);
    if (rc == RT_ERR_CANCEL_EVENT) { // cpu.rs:55-62: return Ok, no tile written
        rc = s->use_v1 ? RT_OK : rtapi::poison_queue(s);
        (void)hipStreamSynchronize(s->stream);
        return rc;
    }
    if (rc != RT_OK) {
        (void)hipStreamSynchronize(s->stream);
        return rc;
    }
    RT_HIP(hipStreamSynchronize(s->stream));
    std::vector<double> column; // v1 / single-column grids: one column of the plain frame, repacked
    for (int ws = 0; ws < p->tiles_w; ++ws) {
        const int x = width_step * ws, w = ws == p->tiles_w - 1 ? p->width - x : width_step;
        const double *col;
        if (column_layout) {
            col = s->host_frame + (size_t)p->height * (size_t)x * 3;
        } else if (w == p->width || w <= 0) {
            col = s->host_frame; // (w == 0: an empty tile column — more tile columns than pixels — reads nothing)
        } else {
            column.resize((size_t)w * (size_t)p->height * 3);
            for (int r = 0; r < p->height; ++r)
                memcpy(&column[(size_t)r * w * 3], s->host_frame + ((size_t)r * p->width + x) * 3, (size_t)w * 3 * sizeof(double));
            col = column.data();
        }
        for (int hs = 0; hs < p->tiles_h; ++hs) {
            if (cancel.raised()) return RT_OK;
            const int y = height_step * hs;
            const int h = hs == p->tiles_h - 1 ? p->height - y : height_step;
            callback(user, col + (size_t)y * (size_t)(w > 0 ? w : 0) * 3, y, x, w > 0 ? w : 0, h > 0 ? h : 0); // empty tiles are sent too (cpu.rs:64-70)
        }
    }
    return RT_OK;
}

int render_tiles(RtScene *const *scenes, int n, const RtCamera *camera, const RtRenderParams *p, int strip_rows,
                 RtTileCallback callback, void *user, const Cancel &cancel) {
    int rc = check_scenes(scenes, n);
    if (rc != RT_OK) return rc;
    if (!callback) return fail(RT_ERR_INVALID_ARGUMENT, "callback is NULL");
    rc = rtapi::check_params(camera, p);
    if (rc != RT_OK) return rc;
    if (p->tiles_w <= 0 || p->tiles_h <= 0) return fail(RT_ERR_INVALID_ARGUMENT, "tile grid must be positive");
    if (p->strip_count > 1) // a tile of the stream is a finished piece of the frame; row ownership is for rt_render_frame*
        return fail(RT_ERR_INVALID_ARGUMENT, "the tile stream delivers whole tiles: params->strip_* is not supported here");
    if (cancel.raised()) return RT_ERR_CANCEL_EVENT; // cpu.rs:82-85: prepare_threads fails with CancelEvent
    bool delivering = p->scale <= 1 && p->width / p->tiles_w > 0;
    for (int i = 0; i < n; ++i) delivering = delivering && !scenes[i]->use_v1;
    if (delivering) return deliver_tiles(scenes, n, camera, p, strip_rows, callback, user, cancel);
    if (n > 1) return fail(RT_ERR_UNSUPPORTED, "rt_render_multi: the preview scale and the v1 kernel render on one device");
    RT_HIP(hipSetDevice(scenes[0]->device));
    return tiles_from_frame(scenes[0], camera, p, callback, user, cancel);
}

template <class F> int guarded(const char *what, F f) { // nothing may unwind through the C ABI
    try {
        return f();
    } catch (const std::bad_alloc &) {
        return fail(RT_ERR_OUT_OF_MEMORY, std::string(what) + ": host allocation failed");
    } catch (const std::exception &e) {
        return fail(RT_ERR_INVALID_ARGUMENT, std::string(what) + ": " + e.what());
    }
}

} // namespace

extern "C" {

int rt_render_frame(RtScene *s, const RtCamera *camera, const RtRenderParams *p, double *out_rgb) {
    if (!s || !out_rgb) return fail(RT_ERR_INVALID_ARGUMENT, "scene/out is NULL");
    int rc = rtapi::check_params(camera, p);
    if (rc != RT_OK) return rc;
    return guarded("rt_render_frame", [&]() -> int {
        RT_HIP(hipSetDevice(s->device));
        if (s->use_v1 || p->scale > 1) { // two-pass path: resolve kernel + one copy
            const size_t n = (size_t)p->width * (size_t)p->height * 3;
            if (s->frame.count < n) RT_HIP(s->frame.alloc(n));
            int rc2 = rtapi::enqueue_render(s, camera, p, s->frame.ptr, s->stream, 0, Cancel());
            if (rc2 != RT_OK) {
                (void)hipStreamSynchronize(s->stream);
                return rc2;
            }
            RT_HIP(hipStreamSynchronize(s->stream));
            if (p->strip_count > 1) { // the owned rows only; the rest of out_rgb stays untouched
                const size_t row_bytes = (size_t)p->width * 3 * sizeof(double);
                for (int r = 0; r < p->height; ++r)
                    if ((r / p->strip_rows) % p->strip_count == p->strip_index)
                        RT_HIP(hipMemcpy(out_rgb + (size_t)r * p->width * 3, s->frame.ptr + (size_t)r * p->width * 3, row_bytes, hipMemcpyDeviceToHost));
            } else {
                RT_HIP(hipMemcpy(out_rgb, s->frame.ptr, n * sizeof(double), hipMemcpyDeviceToHost));
            }
            return RT_OK;
        }
        RtScene *scenes[1] = {s};
        return deliver_frame(scenes, 1, camera, p, 0, out_rgb);
    });
}

int rt_render_frame_multi(RtScene *const *scenes, int n_scenes, const RtCamera *camera, const RtRenderParams *params,
                          int strip_rows, double *out_rgb) {
    int rc = check_scenes(scenes, n_scenes);
    if (rc != RT_OK) return rc;
    if (!out_rgb) return fail(RT_ERR_INVALID_ARGUMENT, "out is NULL");
    rc = rtapi::check_params(camera, params);
    if (rc != RT_OK) return rc;
    if (params->strip_count > 1) return fail(RT_ERR_INVALID_ARGUMENT, "params->strip_* must be unset: the call assigns strips itself");
    if (params->scale > 1) return fail(RT_ERR_INVALID_ARGUMENT, "the preview scale cannot be combined with strips");
    for (int i = 0; i < n_scenes; ++i)
        if (scenes[i]->use_v1) return fail(RT_ERR_UNSUPPORTED, "rt_render_frame_multi needs the pooled kernel");
    return guarded("rt_render_frame_multi", [&] { return deliver_frame(scenes, n_scenes, camera, params, strip_rows, out_rgb); });
}

int rt_render(RtScene *s, const RtCamera *camera, const RtRenderParams *p, RtTileCallback callback, void *user,
              const volatile int *cancel) {
    Cancel c;
    c.flag = cancel;
    RtScene *scenes[1] = {s};
    return guarded("rt_render", [&] { return render_tiles(scenes, 1, camera, p, 0, callback, user, c); });
}

int rt_render_ex(RtScene *s, const RtCamera *camera, const RtRenderParams *p, RtTileCallback callback, void *user,
                 RtCancelCallback cancelled, void *cancel_user) {
    Cancel c;
    c.fn = cancelled;
    c.user = cancel_user;
    RtScene *scenes[1] = {s};
    return guarded("rt_render_ex", [&] { return render_tiles(scenes, 1, camera, p, 0, callback, user, c); });
}

int rt_render_multi(RtScene *const *scenes, int n_scenes, const RtCamera *camera, const RtRenderParams *p, int strip_rows,
                    RtTileCallback callback, void *user, RtCancelCallback cancelled, void *cancel_user) {
    Cancel c;
    c.fn = cancelled;
    c.user = cancel_user;
    return guarded("rt_render_multi", [&] { return render_tiles(scenes, n_scenes, camera, p, strip_rows, callback, user, c); });
}

} // extern "C"
