// rt_scene.h — what the host-side translation units of the library share: the RtScene object
// behind include/rt_abi.h's opaque handle, the error helpers and the one function that enqueues a
// render (rt_api.hip).  Private to the library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "rt_device_types.h"
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

// Column window of a progressive render: pixel columns [x0, x0 + width) of every
// row, `index` of `count` windows of one rt_render call (the segment counter and the
// begin event belong to the first, the item-counter slots to all of them).
struct Window {
    int x0 = 0, width = 0; // width 0 = the whole frame
    int index = 0, count = 1;
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
    int textured = 0; // some material's texture is not a plain SolidColor
    int specular = 0; // some material is Metal or Dielectric

    // closest hit: linear loop for small scenes, skip-link BVH (rt_bvh.h) above kBvhThreshold primitives
    int use_bvh = 0;
    rtapi::DevBuf<rtdev::BvhNode> bvh_nodes;
    rtapi::DevBuf<int32_t> bvh_prim_index;
    rtapi::DevBuf<rtdev::LeafGeo> leaf_geo; // what a leaf test reads (rt_device_types.h)
    double leaf_time_a = 0.0, leaf_inv_dt = 1.0;
    int n_bvh_nodes = 0;
    double bvh_root_mn[3] = {0, 0, 0}, bvh_root_mx[3] = {0, 0, 0}, bvh_center[3] = {0, 0, 0};
    bool bvh_nodes_in_lds = false; // node array (32 B each) staged in dynamic LDS when <= 32 KiB

    // pooled kernel (default): persistent grid = CUs x resident blocks of the variant
    bool use_v1 = false;   // RtSceneOptions.kernel == RT_KERNEL_V1: the lane-per-pixel kernel
    int num_cus = 0, pool_blocks_per_cu = 1;
    int pool_blocks_per_cu_lens = 1; // ... when the camera has an aperture (its lens samples take dynamic LDS)
    rtapi::DevBuf<double> partial;       // [chunks][H][W][3] per-chunk sums
    rtapi::DevBuf<unsigned int> queue;   // one item counter per launch of a render call
    int last_chunks = 0;

    rtapi::DevBuf<double> accum;  // running sums, W*H*3 (v1 kernel)
    rtapi::DevBuf<double> frame;  // resolved frame for the host-output entry points
    rtapi::DevBuf<uint8_t> rgba;  // packed frame of rt_render_frame_rgba8
    rtapi::DevBuf<unsigned long long> segments;
    hipStream_t stream = nullptr; // used by rt_render_frame / rt_render
    hipStream_t stream2 = nullptr; // rt_render: tile columns alternate between the two, so a column's ramp-up fills the CUs its predecessor's tail leaves idle
    hipEvent_t ev_begin = nullptr, ev_traced = nullptr, ev_resolved = nullptr;
    // rt_render's progressive delivery: two pinned column buffers [height][column width][3]
    // and the events that say a column's copy has landed
    double *pinned[2] = {nullptr, nullptr};
    size_t pinned_count[2] = {0, 0};
    hipEvent_t ev_column[2] = {nullptr, nullptr};
    // rt_render's cancel: the stream whose command processor overwrites the launches' item counters (rt_api.hip: poison_queue)
    hipStream_t stream_ctl = nullptr;
    hipStream_t last_stream = nullptr;
    bool has_stats = false;
    int last_launches = 0;
};

namespace rtapi {
int check_params(const RtCamera *camera, const RtRenderParams *p);
// Enqueue trace (in sample batches, polling `cancel` between them) + resolve on `stream`.
// Returns RT_ERR_CANCEL_EVENT when cancelled (callers map that to RT_OK).
int enqueue_render(RtScene *s, const RtCamera *camera, const RtRenderParams *p, double *out_device,
                   hipStream_t stream, int batch, const volatile int *cancel, const Window &win = Window());
} // namespace rtapi
