// rt_bvh_slab.h — the single-precision slab test of the BVH walk, as ONE piece of code for the device
// (rt_trace_common.h: closest_hit_bvh) and for the CPU model of the walk (tools/sim/bvh_sim.cpp, run by
// tests/test_bvh_builder_cpu.py): the host's fmaf / fminf / fmaxf are the IEEE operations v_fma_f32 / v_min_f32 /
// v_max_f32 perform, so the model tests the device's arithmetic, not a restatement of it.
//
// The node boxes only CULL (primitives are tested in f64), so they are f32 boxes around the root's centre, padded by
// 2^-20 of the scene's extent and rounded outward (rt_bvh.cpp); the ray is clipped to the root box in f64 first.
#pragma once
#include <math.h>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define RT_SLAB_FN __host__ __device__ inline
#else
#define RT_SLAB_FN inline
#endif

namespace rtdev {

struct SlabRay {       // a ray prepared for the f32 slab tests
    float ivx, ivy, ivz; // 1 / direction, clamped to a finite magnitude
    float oix, oiy, oiz; // (origin - centre) * iv
};

// 1/d as the slab test wants it.  A direction component of exactly 0 (an axis-parallel ray) has 1/d = +-inf, and
// (float) of a reciprocal beyond 3.4e38 overflows to it: mn * inf - o * inf is then inf - inf = NaN for one plane of the
// slab while the other is +-inf, and fmax(-inf, NaN) = -inf makes t_far = -inf — the node, the ROOT included, is culled
// and the ray silently misses the whole scene.  With the reciprocal clamped to +-2^64 the planes of that slab land at
// +-(huge) with the right signs (the ray is inside the slab or outside it for every t that matters) and nothing
// overflows for scene extents below 2^63.  fminf / fmaxf also turn a NaN reciprocal into a finite one.
RT_SLAB_FN float slab_finite(float v) { return fminf(fmaxf(v, -0x1p64f), 0x1p64f); }

RT_SLAB_FN SlabRay slab_ray(float ofx, float ofy, float ofz, double inv_dx, double inv_dy, double inv_dz) {
    SlabRay r;
    r.ivx = slab_finite((float)inv_dx);
    r.ivy = slab_finite((float)inv_dy);
    r.ivz = slab_finite((float)inv_dz);
    r.oix = ofx * r.ivx;
    r.oiy = ofy * r.ivy;
    r.oiz = ofz * r.ivz;
    return r;
}

// Does the ray's window [tmin_f, best_f] (seen from the clipped origin, already rounded outward) overlap the box?
// `slack` is the relative slack of the interval test (2^-20: the f32 plane distances are good to 2^-23 relative).
// On the device the minima and maxima are written as the instructions themselves: through fminf / fmaxf the compiler
// re-canonicalises the loop-carried window ends (v_max_f32 x, x) on EVERY node, two of the step's 26 vector
// instructions, because it cannot prove them free of signalling NaNs.  v_min / v_max return the non-NaN operand like
// fminf / fmaxf do, so the host form below is the same function.
RT_SLAB_FN bool slab_hit(const float mn[3], const float mx[3], const SlabRay &r, float tmin_f, float best_f, float slack) {
    const float ax = fmaf(mn[0], r.ivx, -r.oix), bx = fmaf(mx[0], r.ivx, -r.oix);
    const float ay = fmaf(mn[1], r.ivy, -r.oiy), by = fmaf(mx[1], r.ivy, -r.oiy);
    const float az = fmaf(mn[2], r.ivz, -r.oiz), bz = fmaf(mx[2], r.ivz, -r.oiz);
#if defined(__HIP_DEVICE_COMPILE__)
    float nx, ny, nz, fx, fy, fz, t_near, t_far;
    asm("v_min_f32 %0, %1, %2" : "=v"(nx) : "v"(ax), "v"(bx));
    asm("v_max_f32 %0, %1, %2" : "=v"(fx) : "v"(ax), "v"(bx));
    asm("v_min_f32 %0, %1, %2" : "=v"(ny) : "v"(ay), "v"(by));
    asm("v_max_f32 %0, %1, %2" : "=v"(fy) : "v"(ay), "v"(by));
    asm("v_min_f32 %0, %1, %2" : "=v"(nz) : "v"(az), "v"(bz));
    asm("v_max_f32 %0, %1, %2" : "=v"(fz) : "v"(az), "v"(bz));
    asm("v_max_f32 %0, %1, %2" : "=v"(nz) : "v"(nz), "v"(tmin_f));
    asm("v_min_f32 %0, %1, %2" : "=v"(fz) : "v"(fz), "v"(best_f));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t_near) : "v"(nx), "v"(ny), "v"(nz));
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(t_far) : "v"(fx), "v"(fy), "v"(fz));
#else
    const float t_near = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), tmin_f));
    const float t_far = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), best_f));
#endif
    return t_near <= fmaf(fabsf(t_far), slack, t_far);
}

} // namespace rtdev
