// rt_bvh_slab.h — the single-precision slab test of the BVH walk, as ONE piece of code for the device
// (rt_trace_common.h: closest_hit_bvh) and for the CPU model of the walk (tools/sim/bvh_sim.cpp, run by
// tests/test_bvh_builder_cpu.py): the host's fmaf / fminf / fmaxf are the IEEE operations v_fma_f32 / v_min_f32 /
// v_max_f32 perform, so the model tests the device's arithmetic, not a restatement of it.
//
// The node boxes only CULL (primitives are tested in f64), so they are f32 boxes around the root's centre, padded by
// 2^-19 of the scene's extent and rounded outward (rt_bvh.cpp); the ray is clipped to the root box in f64 first.
#pragma once
#include <math.h>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define RT_SLAB_FN __host__ __device__ inline
#else
#define RT_SLAB_FN inline
#endif

namespace rtdev {

struct SlabRay {       // a ray prepared for the f32 slab tests: six consecutive floats = the register pairs (ivx, ivy), (ivz, nox), (noy, noz)
    float ivx, ivy, ivz; // 1 / direction, clamped to a finite magnitude
    float nox, noy, noz; // -(origin - centre) * iv
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
    r.nox = -(ofx * r.ivx);
    r.noy = -(ofy * r.ivy);
    r.noz = -(ofz * r.ivz);
    return r;
}

// Does the ray's window [tmin_f, best_f] (seen from the clipped origin, already rounded outward) overlap the box
// lohi = {mn.x, mx.x, mn.y, mx.y, mn.z, mx.z} (BvhNode's layout)?
// No slack in the comparison: every plane distance is off by less than the box's own padding moves it.  With E the
// scene's extent, |coordinate| <= E and |origin - centre| <= E after the clip to the root box; the fma rounds once
// (2^-24 of a result below 2 E |iv|), (origin - centre) and iv were each rounded to f32 once (2^-24 relative): under
// 6 x 2^-24 E |iv| in all, and the planes are padded by 2^-19 E (rt_bvh.cpp), i.e. 2^-19 E |iv| in t — three times
// that.  So a ray that meets the unpadded box in exact arithmetic has t_near <= t_far here.  (Rounds 2 and 3 carried a
// relative slack of 2^-20 on t_far instead, one v_fma_f32 per node on top of a 2^-20 E padding.)
// On the device (closest_hit_bvh) the six plane distances are three v_pk_fma_f32 — each half a true fma, like fmaf —
// and the minima and maxima are written as the instructions themselves: through fminf / fmaxf the compiler
// re-canonicalises the loop-carried window ends (v_max_f32 x, x) on EVERY node because it cannot prove them free of
// signalling NaNs.  v_min / v_max return the non-NaN operand like fminf / fmaxf do, so this host form is the same function.
RT_SLAB_FN bool slab_hit(const float lohi[6], const SlabRay &r, float tmin_f, float best_f) {
    const float ax = fmaf(lohi[0], r.ivx, r.nox), bx = fmaf(lohi[1], r.ivx, r.nox);
    const float ay = fmaf(lohi[2], r.ivy, r.noy), by = fmaf(lohi[3], r.ivy, r.noy);
    const float az = fmaf(lohi[4], r.ivz, r.noz), bz = fmaf(lohi[5], r.ivz, r.noz);
    const float t_near = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), tmin_f));
    const float t_far = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), best_f));
    return t_near <= t_far;
}

} // namespace rtdev
