/* render.c — CpuRenderer restated: tile grid, pixel/sample loops, one
 * worker per core pulling tiles (rayon's into_par_iter stand-in).
 *
 * TEST INFRASTRUCTURE (see oracle.h).
 * Follows racer-tracer/src/renderer/cpu.rs:26-131.
 */
#include "oracle.h"
#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

int orc_online_cores(void) {
    long n = sysconf(_SC_NPROCESSORS_ONLN);
    return n > 0 ? (int)n : 1;
}

/* cpu.rs:73-115: width_step = W / tiles_w, column-major (ws outer, hs inner),
 * last column/row absorb the remainder. */
int orc_tile_grid(int width, int height, int tiles_w, int tiles_h, int32_t *out, int max_tiles) {
    if (tiles_w <= 0 || tiles_h <= 0) return 0;
    int width_step = width / tiles_w;
    int height_step = height / tiles_h;
    int n = 0;
    for (int ws = 0; ws < tiles_w; ++ws)
        for (int hs = 0; hs < tiles_h; ++hs) {
            if (n >= max_tiles) return n;
            out[4 * n + 0] = width_step * ws;
            out[4 * n + 1] = height_step * hs;
            out[4 * n + 2] = ws == tiles_w - 1 ? width - width_step * ws : width_step;
            out[4 * n + 3] = hs == tiles_h - 1 ? height - height_step * hs : height_step;
            ++n;
        }
    return n;
}

typedef struct {
    const RtSceneDesc *desc;
    const OrcScene *scene;
    const RtCamera *camera;
    const RtRenderParams *params;
    const int32_t *tiles;
    int n_tiles;
    atomic_int next_tile;
    atomic_ullong segments;
    double *out_rgb;
} Job;

static int row_owned(const RtRenderParams *p, int row) {
    if (p->strip_count <= 1 || p->strip_rows <= 0) return 1;
    return (row / p->strip_rows) % p->strip_count == p->strip_index;
}

/* cpu.rs:26-71 `raytrace` for one tile; the BufferUpdate write (cpu.rs:64-70)
 * becomes a store into the caller's frame at (r = y, c = x). */
/* cpu_scaled.rs:18-24 */
static int get_highest_divdable(int value, int div) {
    if (div < 1) div = 1;
    while (value % div != 0) --div;
    return div;
}

/* cpu_scaled.rs:45-98 `CpuRendererScaled::raytrace`: trace the top-left pixel of
 * every scale_w x scale_h block and fill the block with it; pixels of the tile
 * that no whole block covers keep Vec3::default() (cpu_scaled.rs:52). */
static void raytrace_tile_scaled(Job *job, const int32_t *tile, int scale_w, int scale_h) {
    const RtRenderParams *p = job->params;
    int x0 = tile[0], y0 = tile[1], w = tile[2], h = tile[3];
    int scaled_width = w / scale_w, scaled_height = h / scale_h;
    unsigned long long segs = 0;
    for (int r = 0; r < h; ++r)
        memset(job->out_rgb + 3 * ((size_t)(y0 + r) * (size_t)p->width + (size_t)x0), 0, sizeof(double) * 3 * (size_t)w);
    for (int row = 0; row < scaled_height; ++row)
        for (int column = 0; column < scaled_width; ++column) {
            int px = x0 + column * scale_w, py = y0 + row * scale_h;
            double u = orc_pixel_u(p, px, py);
            double color[3] = { 0.0, 0.0, 0.0 };
            for (int s = 0; s < p->samples; ++s) {
                double c[3];
                int n;
                orc_sample_radiance_u(job->desc, job->scene, job->camera, p, px, py, s, u, c, &n);
                color[0] += c[0]; color[1] += c[1]; color[2] += c[2];
                segs += (unsigned long long)n;
            }
            double scale = 1.0 / (double)p->samples;
            double out[3] = { sqrt(scale * color[0]), sqrt(scale * color[1]), sqrt(scale * color[2]) };
            for (int sh = 0; sh < scale_h; ++sh)
                for (int sw = 0; sw < scale_w; ++sw)
                    memcpy(job->out_rgb + 3 * ((size_t)(py + sh) * (size_t)p->width + (size_t)(px + sw)), out, sizeof out);
        }
    atomic_fetch_add(&job->segments, segs);
}

static void raytrace_tile(Job *job, const int32_t *tile) {
    const RtRenderParams *p = job->params;
    int x0 = tile[0], y0 = tile[1], w = tile[2], h = tile[3];
    if (p->scale > 1) { /* cpu_scaled.rs:33-41 */
        int tw = p->tiles_w > 0 ? p->tiles_w : 1, th = p->tiles_h > 0 ? p->tiles_h : 1;
        raytrace_tile_scaled(job, tile, get_highest_divdable(p->width / tw, p->scale),
                             get_highest_divdable(p->height / th, p->scale));
        return;
    }
    unsigned long long segs = 0;
    for (int row = 0; row < h; ++row) {
        if (!row_owned(p, y0 + row)) continue;
        for (int column = 0; column < w; ++column) {
            double u = orc_pixel_u(p, x0 + column, y0 + row); /* cpu.rs:35-36 */
            double color[3] = { 0.0, 0.0, 0.0 };
            for (int s = 0; s < p->samples; ++s) {
                double c[3];
                int n;
                orc_sample_radiance_u(job->desc, job->scene, job->camera, p, x0 + column, y0 + row, s, u, c, &n);
                color[0] += c[0]; color[1] += c[1]; color[2] += c[2]; /* vec3.rs:38-42 */
                segs += (unsigned long long)n;
            }
            /* vec3.rs:119-125 scale_sqrt */
            double scale = 1.0 / (double)p->samples;
            double *px = job->out_rgb + 3 * ((size_t)(y0 + row) * (size_t)p->width + (size_t)(x0 + column));
            px[0] = sqrt(scale * color[0]);
            px[1] = sqrt(scale * color[1]);
            px[2] = sqrt(scale * color[2]);
        }
    }
    atomic_fetch_add(&job->segments, segs);
}

static void *worker(void *arg) {
    Job *job = (Job *)arg;
    for (;;) {
        int t = atomic_fetch_add(&job->next_tile, 1);
        if (t >= job->n_tiles) break;
        raytrace_tile(job, job->tiles + 4 * t);
    }
    return NULL;
}

int orc_render(const RtSceneDesc *desc, const RtCamera *camera, const RtRenderParams *params,
               int n_threads, int use_bvh, double *out_rgb, uint64_t *segments) {
    if (!desc || !camera || !params || !out_rgb) return RT_ERR_INVALID_ARGUMENT;
    if (params->width <= 0 || params->height <= 0 || params->samples <= 0 || params->max_depth < 0)
        return RT_ERR_INVALID_ARGUMENT;
    int tw = params->tiles_w > 0 ? params->tiles_w : 1;
    int th = params->tiles_h > 0 ? params->tiles_h : 1;
    if (tw > params->width) tw = params->width;
    if (th > params->height) th = params->height;
    int32_t *tiles = (int32_t *)malloc(sizeof(int32_t) * 4 * (size_t)tw * (size_t)th);
    if (!tiles) return RT_ERR_OUT_OF_MEMORY;
    OrcScene *scene = orc_scene_build(desc, use_bvh, params->seed);
    Job job;
    memset(&job, 0, sizeof job);
    job.desc = desc;
    job.scene = scene;
    job.camera = camera;
    job.params = params;
    job.tiles = tiles;
    job.n_tiles = orc_tile_grid(params->width, params->height, tw, th, tiles, tw * th);
    atomic_init(&job.next_tile, 0);
    atomic_init(&job.segments, 0);
    job.out_rgb = out_rgb;

    if (n_threads <= 0) n_threads = orc_online_cores();
    if (n_threads > job.n_tiles) n_threads = job.n_tiles;
    if (n_threads <= 1) {
        worker(&job);
    } else {
        pthread_t *th_ids = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
        int started = 0;
        for (int i = 0; i < n_threads; ++i)
            if (pthread_create(&th_ids[started], NULL, worker, &job) == 0) ++started;
        if (started == 0) worker(&job);
        for (int i = 0; i < started; ++i) pthread_join(th_ids[i], NULL);
        free(th_ids);
    }
    if (segments) *segments = (uint64_t)atomic_load(&job.segments);
    orc_scene_free(scene);
    free(tiles);
    return RT_OK;
}
