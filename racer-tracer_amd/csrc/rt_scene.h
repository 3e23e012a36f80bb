// rt_scene.h — what the host-side translation units of the library share: the RtScene object
// behind include/rt_abi.h's opaque handle, the error helpers and the one function that enqueues a
// render (rt_api.hip).  Private to the library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "rt_device_types.h"
#include "rt_bvh.h"
#include "../../include/rt_abi.h"

namespace rtapi {

// sets the thread-local text behind rt_last_error_message and returns `code`
int fail(int code, const std::string &msg);

#define RT_HIP(call)                                                                            \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return rtapi::fail(RT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));  \
    } while (0)

template <class T> struct DevBuf {
    T *ptr = nullptr;
    size_t count = 0;
    hipError_t alloc(size_t n) {
        release();
        if (n == 0) return hipSuccess;
        hipError_t e = hipMalloc((void **)&ptr, n * sizeof(T));
        if (e == hipSuccess) count = n;
        return e;
    }
    void release() {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        count = 0;
    }
};

// How a caller asks for cancellation: the flag of rt_render, the callback of rt_render_ex, or neither.
// Polled by the calling thread only.
struct Cancel {
    const volatile int *flag = nullptr;
    int (*fn)(void *) = nullptr;
    void *user = nullptr;
    bool armed() const { return flag != nullptr || fn != nullptr; }
    bool raised() const { return (flag && *flag) || (fn && fn(user) != 0); }
};

// Where a launch's finished pixels go when the launch delivers them itself (TraceArgs.deliver_*; rt_deliver.hip).
struct Delivery {
    double *out = nullptr;              // device-visible address of the output (pinned host memory or HBM)
    int col_step = 0, cols = 1;         // tile-column layout; cols == 1: plain [H][W][3] with col_step == width
    std::vector<rtdev::Region> regions; // non-empty rectangles of item tiles in queue order (item_begin is filled in)
    uint32_t serial = 0;                // value published in the scene's host_flags[region]
    bool cancellable = false;           // the caller polls a cancel hook while this launch runs: the waves read the scene's cancel word
};

} // namespace rtapi

struct RtScene {
    int device = 0;
    rtapi::DevBuf<rtdev::Prim> prims;
    rtapi::DevBuf<rtdev::Texture> textures;
    rtapi::DevBuf<rtdev::Image> images;
    rtapi::DevBuf<rtdev::Perlin> perlins;
    std::vector<uint8_t *> image_pixels; // device copies of the RGBA8 texels
    int n_prims = 0, n_materials = 0, n_textures = 0, n_images = 0, n_perlins = 0;
    int perlin_identity = 1; // all permutation tables are the identity (noise.rs:121-130 never shuffles them)
    rtdev::Background bg;
    // kernel specialisation (rt_trace_kernel.hip): 0 rects only, 1 spheres only, 2 anything
    int prims_class = 2;
    int rect_end[3] = {0, 0, 0}; // linear loop: ends of the XY / XZ / YZ rect groups of the (grouped) device table
    int sphere_end = 0;          // ... and of the plain-sphere group behind them
    int box_end = 0;             // ... and of the boxes (wrapped or not) behind those
    int textured = 0; // some material's texture is not a plain SolidColor
    // upper bound of a finished sample's radiance (every attenuation in [0, 1]; emission, background and the depth-0 white
    // below it), or 0 when the scene has none (rt_api.hip: scene_create): what sizes the pooled kernel's fixed-point sums
    double radiance_bound = 0.0;
    int specular = 0; // some material is Metal or Dielectric
    int has_moving = 0; // some primitive is a MovingSphere (the only reader of a ray's time)

    // closest hit: linear loop for small scenes, skip-link BVH (rt_bvh.h) above kBvhThreshold primitives
    int use_bvh = 0;
    rtapi::DevBuf<rtdev::BvhNode> bvh_nodes;
    rtapi::DevBuf<rtdev::BvhNode> bvh_nodes_ordered; // eight direction-ordered copies (trees that do not fit LDS)
    rtapi::DevBuf<int32_t> bvh_prim_index;
    rtapi::DevBuf<rtdev::LeafGeo> leaf_geo; // what a leaf test reads (rt_device_types.h)
    double leaf_time_a = 0.0, leaf_inv_dt = 1.0;
    int n_bvh_nodes = 0;
    double bvh_root_mn[3] = {0, 0, 0}, bvh_root_mx[3] = {0, 0, 0}, bvh_center[3] = {0, 0, 0};
    bool bvh_nodes_in_lds = false; // node array (32 B each) staged in dynamic LDS when <= 32 KiB
    // the tree on the host, for trees that live in LDS: before a launch the node array is re-emitted with every node's
    // children nearest-to-the-camera first (rt_bvh.h: order_bvh_for_origin) whenever the camera has moved
    rtdev::BvhBuild bvh_host;
    double bvh_ordered_for[3] = {0, 0, 0};
    bool bvh_is_ordered = false;
    std::vector<rtdev::BvhNode> bvh_ordered_nodes[2]; // host copies of the array last uploaded and the one before
    int bvh_upload_slot = 0;

    // pooled kernel (default): persistent grid = CUs x resident blocks of the variant
    bool use_v1 = false;   // RtSceneOptions.kernel == RT_KERNEL_V1: the lane-per-pixel kernel
    bool exact = false;    // RtSceneOptions.arithmetic == RT_ARITH_REFERENCE: the *_exact copy of the trace kernels
    bool gather_staged = false; // RtSceneOptions.gather == RT_GATHER_STAGED (rt_multi.hip)
    int num_cus = 0, pool_blocks_per_cu = 1;
    int pool_blocks_per_cu_lens = 1; // ... when the camera has an aperture (its lens samples take dynamic LDS)
    rtapi::DevBuf<double> partial;       // [chunks][H][W][3] per-chunk sums
    rtapi::DevBuf<unsigned int> queue;   // one item counter per launch of a render call
    int last_chunks = 0;

    rtapi::DevBuf<double> accum;  // running sums, W*H*3 (v1 kernel)
    rtapi::DevBuf<double> frame;  // resolved frame for the host-output entry points
    rtapi::DevBuf<uint8_t> rgba;  // packed frame of rt_render_frame_rgba8
    rtapi::DevBuf<unsigned long long> segments;
    hipStream_t stream = nullptr; // used by the host-output entry points
    hipEvent_t ev_begin = nullptr, ev_traced = nullptr, ev_resolved = nullptr;
    // Delivery (rt_deliver.hip): the launch writes finished pixels straight into `host_frame` — pinned, portable host
    // memory mapped into every device, so several scenes (devices) can fill one frame — and publishes finished regions
    // in `host_flags`; tile_done / region_done are the device counters behind that (zero between launches).
    double *host_frame = nullptr;
    size_t host_frame_count = 0;
    unsigned int *host_flags = nullptr; // [RT_MAX_REGIONS] published regions + [1] the cancel word the waves read (TraceArgs.cancel_flag)
    rtapi::DevBuf<unsigned int> tile_done, region_done;
    uint32_t deliver_serial = 0;
    bool deliver_dirty = false; // a delivering launch was cut short: the counters must be cleared before the next one
    // cancel: the stream whose command processor overwrites the launches' item counters (rt_api.hip: poison_queue)
    hipStream_t stream_ctl = nullptr;
    hipStream_t last_stream = nullptr;
    bool has_stats = false;
    int last_launches = 0;
};

namespace rtapi {
int check_params(const RtCamera *camera, const RtRenderParams *p);
// Enqueue trace + resolve on `stream` (two-pass path: the resolve kernel writes out_device), or — with a Delivery —
// ONE delivering launch that finishes its own pixels (out_device is ignored).  `cancel` is polled between the
// sample batches of the v1 kernel only.  Returns RT_ERR_CANCEL_EVENT when cancelled (callers map that to RT_OK).
// out_col_step / out_cols: layout the RESOLVE pass of the two-pass path writes (rt_trace_pool_kernel.hip:
// k_resolve_chunks_f64): the plain frame, or the tile stream's column layout.
int enqueue_render(RtScene *s, const RtCamera *camera, const RtRenderParams *p, double *out_device,
                   hipStream_t stream, int batch, const Cancel &cancel, const Delivery *delivery = nullptr,
                   int out_col_step = 0, int out_cols = 1);
// Allocate now what enqueue_render would allocate for these parameters (calls over several shares: before the first launch).
int reserve_render_buffers(RtScene *s, const RtRenderParams *p, bool delivering);
// Block until `ev` has happened, polling `cancel` meanwhile (RT_ERR_CANCEL_EVENT as soon as it is raised).
int wait_event(hipEvent_t ev, const Cancel &cancel);
// Ends the pool launches in flight on `s` early: every item counter becomes 2^31 (rt_api.hip).
int poison_queue(RtScene *s);
// The same without waiting for the write to land: several shares are poisoned side by side (rt_deliver.hip: abort_shares) —
// a share whose launch still waits behind another share's on the SAME device only reaches its ev_begin when that one ends,
// and waiting for it before poisoning the next would let the next run to its end.
int poison_queue_begin(RtScene *s);
// Image row of row `vr` of the launch's owned-row grid (identity without strips).
inline int owned_row_to_image_row(const RtRenderParams *p, int vr) {
    if (p->strip_count <= 1) return vr;
    return ((vr / p->strip_rows) * p->strip_count + p->strip_index) * p->strip_rows + vr % p->strip_rows;
}
// Sample chunks a frame of `samples` samples per pixel is cut into (a function of spp only: rt_api.hip: chunk_plan).
int chunk_count(int samples);
// Rows of the owned-row grid of a render with these parameters (a multiple of strip_rows with strips).
int owned_rows_of(const RtRenderParams *p);
} // namespace rtapi
