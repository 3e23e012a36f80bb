// rt_multi.hip — one frame on several GPUs of ONE process, collected in device 0's HBM (include/rt_abi.h:
// rt_render_frame_multi_device; the host-output forms rt_render_frame_multi / rt_render_multi live in rt_deliver.hip).
//
// The reference shards a frame over its rayon pool by tile inside one
// `CpuRenderer::render` call (racer-tracer/src/renderer/cpu.rs:118-131).  Here
// the shards are 8-row strips, strip j -> scenes[j % n] (interleaved: cheap sky
// rows and expensive floor rows spread evenly), every device traces its strips
// on its own stream at the same time, and the finished strips are collected
// in device 0's HBM by peer copies over xGMI — one strided copy per device —
// enqueued on the SOURCE device's stream right behind its resolve pass.
// One process owns every device here, so plain peer copies (SDMA engines over
// the xGMI links) do the gather; RCCL's rendezvous buys nothing without a
// second process.  The multi-process form of the same gather is
// racer-tracer_amd/strips.py (torch.distributed, backend "nccl" = RCCL).
//
// Host code only.
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include "rt_scene.h"

using rtapi::fail;

namespace {

struct Share {
    RtScene *scene;
    RtRenderParams params;
    double *target; // where this device's resolve pass writes (a full-frame sized buffer; owned rows only)
    bool in_place;  // target IS the output buffer: no copy
};

int render_multi(RtScene *const *scenes, int n, const RtCamera *camera, const RtRenderParams *p, int strip_rows,
                 double *out) {
    if (!scenes || n <= 0) return fail(RT_ERR_INVALID_ARGUMENT, "no scenes");
    for (int i = 0; i < n; ++i)
        if (!scenes[i]) return fail(RT_ERR_INVALID_ARGUMENT, "scenes[" + std::to_string(i) + "] is NULL");
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j)
            if (scenes[i] == scenes[j]) return fail(RT_ERR_INVALID_ARGUMENT, "the same RtScene is listed twice (create one per share)");
    if (!out) return fail(RT_ERR_INVALID_ARGUMENT, "out is NULL");
    int rc = rtapi::check_params(camera, p);
    if (rc != RT_OK) return rc;
    if (p->strip_count > 1) return fail(RT_ERR_INVALID_ARGUMENT, "params->strip_* must be unset: the call assigns strips itself");
    if (p->scale > 1) return fail(RT_ERR_INVALID_ARGUMENT, "the preview scale cannot be combined with strips");
    if (strip_rows < 0) return fail(RT_ERR_INVALID_ARGUMENT, "strip_rows must not be negative");
    if (strip_rows == 0) strip_rows = 8;

    const size_t row_elems = (size_t)p->width * 3;
    const size_t n_elems = row_elems * (size_t)p->height;
    const int n_strips = (p->height + strip_rows - 1) / strip_rows;
    const int dst_device = scenes[0]->device;

    std::vector<Share> shares((size_t)n);
    // whatever way this function is left, no stream is still writing into the scenes' buffers afterwards
    struct Drain {
        std::vector<RtScene *> launched;
        ~Drain() {
            for (RtScene *s : launched)
                if (hipSetDevice(s->device) == hipSuccess) (void)hipStreamSynchronize(s->stream);
        }
    } drain;
    // 0. every allocation of the call before its first launch (hipMalloc waits for the device's running kernels)
    for (int i = 0; i < n; ++i) {
        Share &sh = shares[(size_t)i];
        sh.scene = scenes[i];
        sh.params = *p;
        if (n > 1) {
            sh.params.strip_rows = strip_rows;
            sh.params.strip_count = n;
            sh.params.strip_index = i;
        }
        RT_HIP(hipSetDevice(sh.scene->device));
        sh.in_place = sh.scene->device == dst_device && !sh.scene->gather_staged;
        if (!sh.in_place && sh.scene->frame.count < n_elems) RT_HIP(sh.scene->frame.alloc(n_elems));
        rc = rtapi::reserve_render_buffers(sh.scene, &sh.params, false);
        if (rc != RT_OK) return rc;
    }
    // 1. every device starts tracing before any copy is enqueued (a copy to pageable host memory
    //    blocks the calling thread)
    for (int i = 0; i < n; ++i) {
        Share &sh = shares[(size_t)i];
        RT_HIP(hipSetDevice(sh.scene->device));
        // RT_GATHER_STAGED (RtSceneOptions.gather, a test switch): the share renders into its own staging frame and is
        // copied even when it sits on the output's device, so that ONE card runs what several run
        sh.target = sh.in_place ? out : sh.scene->frame.ptr;
        rc = rtapi::enqueue_render(sh.scene, camera, &sh.params, sh.target, sh.scene->stream, 0, rtapi::Cancel());
        drain.launched.push_back(sh.scene);
        if (rc != RT_OK) return rc;
    }
    // 2. every other device sends its own strips to device 0, behind its resolve pass on its own stream
    if (n > 1) {
        for (int i = 0; i < n; ++i) {
            const int dev = scenes[i]->device;
            if (dev == dst_device) continue; // (a staged share on the output's device copies within the device)
            RT_HIP(hipSetDevice(dev));
            int can = 0;
            RT_HIP(hipDeviceCanAccessPeer(&can, dev, dst_device));
            if (can) {
                const hipError_t e = hipDeviceEnablePeerAccess(dst_device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
                    return fail(RT_ERR_HIP, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
                (void)hipGetLastError();
            } // without peer access hipMemcpyPeerAsync stages through the host by itself
        }
    }
    int first_error = RT_OK;
    for (int i = 0; i < n && first_error == RT_OK; ++i) {
        const Share &sh = shares[(size_t)i];
        if (sh.in_place) continue;
        if (hipSetDevice(sh.scene->device) != hipSuccess) {
            first_error = fail(RT_ERR_HIP, "hipSetDevice failed");
            break;
        }
        const int src_device = sh.scene->device;
        hipStream_t st = sh.scene->stream;
        auto peer = [&](size_t off_elems, size_t bytes) {
            return bytes == 0 ? hipSuccess : hipMemcpyPeerAsync(out + off_elems, dst_device, sh.target + off_elems, src_device, bytes, st);
        };
        hipError_t e = hipSuccess;
        if (n == 1) { // one share (on another device than the output, or staged): the whole frame in one copy
            e = peer(0, n_elems * sizeof(double));
        } else {
            // Share i owns the strips i, i + n, i + 2n, ...: `full` whole strips, pitch n * strip_rows rows apart — ONE
            // strided copy (34 strips per device at C5) — and possibly one short strip at the bottom of the image.
            const size_t strip_bytes = (size_t)strip_rows * row_elems * sizeof(double);
            const size_t pitch = strip_bytes * (size_t)n;
            int full = 0, tail_rows = 0, tail_row0 = 0;
            for (int j = i; j < n_strips; j += n) {
                const int r0 = j * strip_rows;
                if (r0 + strip_rows <= p->height) ++full;
                else {
                    tail_rows = p->height - r0;
                    tail_row0 = r0;
                }
            }
            const size_t first = (size_t)i * (size_t)strip_rows * row_elems;
            if (full > 0) {
                e = hipMemcpy2DAsync(out + first, pitch, sh.target + first, pitch, strip_bytes, (size_t)full, hipMemcpyDeviceToDevice, st);
                if (e != hipSuccess) { // a runtime that refuses a strided copy between these two devices: strip by strip
                    (void)hipGetLastError();
                    e = hipSuccess;
                    for (int k = 0; k < full && e == hipSuccess; ++k) e = peer(first + (size_t)k * (pitch / sizeof(double)), strip_bytes);
                }
            }
            if (e == hipSuccess && tail_rows > 0) e = peer((size_t)tail_row0 * row_elems, (size_t)tail_rows * row_elems * sizeof(double));
        }
        if (e != hipSuccess) first_error = fail(RT_ERR_HIP, std::string("strip copy: ") + hipGetErrorString(e));
    }
    // 3. the frame is complete when every stream has drained
    for (int i = 0; i < n; ++i) {
        if (hipSetDevice(scenes[i]->device) != hipSuccess || hipStreamSynchronize(scenes[i]->stream) != hipSuccess)
            if (first_error == RT_OK) first_error = fail(RT_ERR_HIP, "stream synchronisation failed on share " + std::to_string(i));
    }
    return first_error;
}

} // namespace

extern "C" {

int rt_render_frame_multi_device(RtScene *const *scenes, int n_scenes, const RtCamera *camera,
                                 const RtRenderParams *params, int strip_rows, double *out_rgb_device) {
    try {
        return render_multi(scenes, n_scenes, camera, params, strip_rows, out_rgb_device);
    } catch (const std::exception &e) {
        return fail(RT_ERR_OUT_OF_MEMORY, std::string("rt_render_frame_multi_device: ") + e.what());
    }
}

} // extern "C"
