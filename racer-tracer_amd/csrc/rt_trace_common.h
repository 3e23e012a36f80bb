// rt_trace_common.h — device-side building blocks shared by the trace
// kernels: Vec3 maths, the addressed Philox RNG (include/rt_rng.h), primitive
// intersection with the RotateY/Translate wrappers, hit records, textures.
// Formulas follow SURVEY.md App. A; each function cites the Rust it mirrors.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rt_device_types.h"
#include "rt_bvh_slab.h"
#include "../../include/rt_abi.h"
#include "../../include/rt_rng.h"

// ARITHMETIC.  The trace kernels exist twice in the library (RtSceneOptions.arithmetic):
//   RT_ARITH_FAST      — reciprocals and reciprocal square roots once per ray from the hardware seeds (<= 1 ulp), FMA
//                        contraction on: the default;
//   RT_ARITH_REFERENCE — the reference's own operations: IEEE divisions (vec3.rs:279-301 multiplies by 1/x,
//                        sphere.rs:52 and xy_rect.rs:31 divide), sqrt + three divisions for unit_vector
//                        (vec3.rs:79-85), no FMA contraction.  Compiled from the SAME sources with -DRT_EXACT_DIV
//                        -ffp-contract=off into namespace rtdev_exact and launchers named *_exact.
// Everything that is not arithmetic (types, tables, TraceArgs) is shared: namespace rtdev.
#ifdef RT_EXACT_DIV
#define RT_KNS rtdev_exact
#define RT_LAUNCHER(name) name##_exact
#else
#define RT_KNS rtdev_fast
#define RT_LAUNCHER(name) name
#endif

namespace rtdev {
enum { PRIMS_RECTS = 0, PRIMS_SPHERES = 1, PRIMS_ANY = 2 };
}

namespace RT_KNS {
using namespace rtdev;

// Scene tables are never written while a trace kernel runs.  Reading them
// through the constant address space lets the backend use scalar loads (s_load,
// K$) whenever the index is wave-uniform — the closest-hit loop — instead of
// per-lane vector loads with a vmcnt wait per primitive.
#define RT_CONSTANT __attribute__((address_space(4)))
__device__ __forceinline__ Prim load_prim_uniform(const Prim *table, int i) {
    Prim p;
    __builtin_memcpy(&p, (const RT_CONSTANT Prim *)(table + i), sizeof(Prim));
    return p;
}

// The kernel's own argument block, re-read where it is used.  Camera and
// background are 41 doubles that only the regeneration and miss blocks read;
// held in SGPRs across the path loop they are spilled to VGPR lanes and come
// back one v_readlane at a time (measured: ~80 of ~580 VALU instructions per
// loop iteration).  Reading them through a laundered kernarg pointer keeps the
// scalar loads (K$ hits) next to their use instead.
__device__ __forceinline__ const RT_CONSTANT TraceArgs *kernargs_here() {
    const RT_CONSTANT TraceArgs *p = (const RT_CONSTANT TraceArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p)); // opaque per use: the loads cannot be hoisted out of the loop
    return p;
}

// Wave votes and lane shuffles without the library's wrappers.  HIP's ballot(int) turns its predicate into an integer and
// back (v_cndmask_b32 0 / 1, v_cmp_ne_u32: two vector instructions in front of every ballot — a dozen ballots an
// iteration); the builtin takes the comparison's own lane mask.  __shfl adds the lane's row base for widths below 64
// (v_and_or_b32), which a wave-wide shuffle does not need.
__device__ __forceinline__ uint64_t ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ int shfl_i(int v, int src_lane) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }
__device__ __forceinline__ double shfl_d(double v, int src_lane) {
    const long long bits = __double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(uint32_t)bits);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(uint32_t)((unsigned long long)bits >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// ------------------------------------------------------------------ vec3
struct d3 {
    double x, y, z;
};
__device__ __forceinline__ d3 ld3(const RT_CONSTANT double *p) { return d3{p[0], p[1], p[2]}; }
__device__ __forceinline__ d3 mk(double x, double y, double z) { return d3{x, y, z}; }
__device__ __forceinline__ d3 ld3(const double *p) { return d3{p[0], p[1], p[2]}; }
__device__ __forceinline__ d3 operator+(d3 a, d3 b) { return d3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ d3 operator-(d3 a, d3 b) { return d3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ d3 operator-(d3 a) { return d3{-a.x, -a.y, -a.z}; }
__device__ __forceinline__ d3 operator*(d3 a, d3 b) { return d3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ d3 operator*(d3 a, double s) { return d3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ d3 operator*(double s, d3 a) { return d3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ double dot(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ double len2(d3 a) { return dot(a, a); }
// vec3.rs:79-85 divides by the length
__device__ __forceinline__ d3 unit(d3 a) {
    double l = sqrt(len2(a));
    return d3{a.x / l, a.y / l, a.z / l};
}
// 1/x and 1/sqrt(x) to ~1 ulp: hardware seed + two Newton steps (the same
// refinement the compiler's full f64 division uses, without its scaling and
// fix-up instructions).  Used where the reference divides several values by
// one denominator; results differ from true division by an ulp or two.
// -DRT_EXACT_DIV (the RT_ARITH_REFERENCE copy of the kernels, compiled -ffp-contract=off): the
// reference's own operations instead — IEEE division (vec3.rs:279-301 multiplies by 1/x, sphere.rs:52
// and xy_rect.rs:31 divide) and sqrt + three divisions for unit_vector (vec3.rs:79-85) — so that a
// caller can have the reference's arithmetic, and a parity test can tell an arithmetic difference from a
// traversal defect.
__device__ __forceinline__ double rcp_f64(double x) {
#ifdef RT_EXACT_DIV
    return 1.0 / x;
#endif
    // v_rcp_f64 is good to 2^-25 (measured, tools/microbench/refine_accuracy.hip), so ONE step of third order —
    // r0 (1 + e + e^2), e = 1 - x r0 — leaves 2^-75 before rounding: at most 0.5 ulp from 1/x and correctly
    // rounded for 99.97 % of arguments, exactly what two Newton steps gave, with three fma instead of four
    const double r0 = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r0, 1.0);
    const double r = fma(r0, fma(e, e, e), r0);
    // x = 0 or inf turns the refinement into NaN: v_div_fixup_f64 puts the true quotient's inf / 0 back (one
    // instruction; it hands every other quotient through unchanged)
    return __builtin_amdgcn_div_fixup(r, x, 1.0);
}
__device__ __forceinline__ double rsqrt_f64(double x) {
#ifdef RT_EXACT_DIV
    return 1.0 / sqrt(x);
#endif
    // one step of third order from the 2^-25 seed: y (1 + e/2 + 3 e^2/8), e = 1 - x y^2.  Measured at most
    // 0.99 ulp from 1/sqrt(x) (87 % correctly rounded) in five operations; the two coupled Goldschmidt steps
    // this replaces took eight and reached 1.64 ulp (80 %) (tools/microbench/refine_accuracy.hip)
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-y, x * y, 1.0);
    return fma(y, e * fma(e, 0.375, 0.5), y);
}
// sqrt(x) for x >= 0 to ~1 ulp from the reciprocal-square-root seed: the library's correctly rounded sqrt is
// this plus a range-scaling prologue/epilogue for arguments near the ends of the exponent range (18
// instructions against 11), and the sphere test (sphere.rs:47) takes one square root per candidate sphere.
__device__ __forceinline__ double sqrt_fast(double x) {
#ifdef RT_EXACT_DIV
    return sqrt(x);
#endif
    // rsq(0) = inf would turn the products below into NaN; with the seed taken at max(x, 2^-1000) every product
    // of x = 0 is an exact 0 (one v_max_f64 instead of a compare and two selects; arguments in (0, 2^-1000),
    // 150 orders of magnitude below any discriminant this kernel forms, would come out wrong)
    const double y = __builtin_amdgcn_rsq(fmax(x, 0x1p-1000));
    double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    return fma(fma(-g, g, x), h, g);
}
// vec3.rs:79-85 unit_vector with one reciprocal square root instead of a
// square root and three divisions
__device__ __forceinline__ d3 unit_fast(d3 a) {
#ifdef RT_EXACT_DIV
    return unit(a);
#endif
    double inv = rsqrt_f64(len2(a));
    return d3{a.x * inv, a.y * inv, a.z * inv};
}
// num / den where `inv` = 1/den was formed once per ray (the reference divides: xy_rect.rs:31, sphere.rs:52)
__device__ __forceinline__ double div_by(double num, double den, double inv) {
#ifdef RT_EXACT_DIV
    return num / den;
#else
    return num * inv;
#endif
}
__device__ __forceinline__ d3 rcp3(d3 a) { return d3{rcp_f64(a.x), rcp_f64(a.y), rcp_f64(a.z)}; }
__device__ __forceinline__ double comp(d3 a, int axis) { return axis == 0 ? a.x : (axis == 1 ? a.y : a.z); }

// ------------------------------------------------------------------- RNG
struct u4 {
    uint32_t a, b, c, d;
};

// gfx950's three-input bitwise op (truth table 0x96 = a ^ b ^ c) folds the two
// xors of every Philox half-round into one VALU instruction.
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
}

__device__ __forceinline__ u4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                            uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < RT_PHILOX_ROUNDS; ++r) {
        uint64_t p0 = (uint64_t)RT_PHILOX_M0 * c0;
        uint64_t p1 = (uint64_t)RT_PHILOX_M1 * c2;
        uint32_t n0 = xor3((uint32_t)(p1 >> 32), c1, k0);
        uint32_t n2 = xor3((uint32_t)(p0 >> 32), c3, k1);
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += RT_PHILOX_W0;
        k1 += RT_PHILOX_W1;
    }
    return u4{c0, c1, c2, c3};
}

// (((u64)hi << 32 | lo) >> 11) * 2^-53, exactly (two exact conversions, exact sum)
__device__ __forceinline__ double u53(uint32_t hi, uint32_t lo) {
    uint32_t top = hi >> 11;
    uint32_t low = (hi << 21) | (lo >> 11);
    return fma((double)top, 0x1p-21, (double)low * 0x1p-53);
}

// random_double_range(-1, 1) of the same draw: -1 + 2 * u53(hi, lo).  Every
// intermediate is exactly representable (multiples of 2^-52 below 1 in
// magnitude), so this equals the oracle's a + (b - a) * d bit for bit.
__device__ __forceinline__ double sym53(uint32_t hi, uint32_t lo) {
    // k = the top 53 bits of hi:lo; the value is k * 2^-52 - 1.  With b = the top bit of k and m = its low 52 bits, the
    // double D = 1.m (exponent of [1, 2), mantissa m — bit operations only) is 1 + m * 2^-52, and k * 2^-52 - 1 =
    // b + m * 2^-52 - 1 = D - (2 - b): ONE f64 subtraction (exact: both operands are multiples of 2^-52 below 2 in
    // magnitude) instead of two integer-to-double conversions and two fma (17 SIMD cycles instead of 23 per coordinate).
    const uint32_t h = hi >> 11;                                   // bit 20: b; bits 19..0: the top of m
    const uint32_t d_hi = 0x3FF00000u | (h & 0x000FFFFFu);
    const uint32_t d_lo = (hi << 21) | (lo >> 11);                 // the low 32 bits of m
    const uint32_t c_hi = 0x40000000u - (h & 0x00100000u);         // 2.0, or 1.0 (0x3FF00000) when b is set
    const double D = __longlong_as_double((long long)(((unsigned long long)d_hi << 32) | d_lo));
    const double c = __longlong_as_double((long long)((unsigned long long)c_hi << 32));
    return D - c;
}

// The three coordinates of a random_in_unit_sphere candidate from ONE block (rt_rng.h, RT_RNG_SCATTER): coordinate j is
// random_double_range(-1, 1) of the 42-bit uniform d_j = U_j * 2^-42, U_j = out[j] << 10 | (out[3] >> 10 j & 0x3FF).
// D = 2 * (1 + U * 2^-42) is assembled from bits (exponent of [2, 4), mantissa U << 10) and -1 + 2 d = D - 3, exactly.
__device__ __forceinline__ double sym42(uint32_t w, uint32_t t_at_10) { // t_at_10: the ten extra bits at bits 10..19
    const uint32_t d_hi = 0x40000000u | (w >> 12);
    const uint32_t d_lo = (w << 20) | t_at_10;
    return __longlong_as_double((long long)(((unsigned long long)d_hi << 32) | d_lo)) - 3.0;
}
__device__ __forceinline__ d3 sphere_candidate(const u4 &b) {
    return mk(sym42(b.a, (b.d << 10) & 0x000FFC00u), sym42(b.b, b.d & 0x000FFC00u), sym42(b.c, (b.d >> 10) & 0x000FFC00u));
}

struct PathRng {
    uint32_t pixel, sample, k0, k1;
    __device__ __forceinline__ u4 block(uint32_t segment, uint32_t purpose, uint32_t blk) const {
        return philox4x32(pixel, sample, (segment << 8) | purpose, blk, k0, k1);
    }
};

// vec3.rs:424-430 random_in_unit_sphere under the addressed-draw contract
__device__ __forceinline__ d3 random_in_unit_sphere(const PathRng &rng, uint32_t segment) {
    for (uint32_t i = 0;; ++i) {
        d3 p = sphere_candidate(rng.block(segment, RT_RNG_SCATTER, i));
        if (len2(p) >= 1.0) continue;
        return p;
    }
}

// --------------------------------------------------------------- geometry
struct Hit {
    d3 point, normal;
    double u, v;
    bool front;
    bool uv_approx; // u, v come from sphere_uv_f32: good to UV_EPS_U / UV_EPS_V, to be certified by whoever reads them
};

__device__ __forceinline__ void set_face_normal(Hit &h, d3 dir, d3 outward) { // geometry.rs:49-56
    h.front = dot(dir, outward) < 0.0;
    // `front ? outward : -outward` as ONE select of a sign mask and three xors of the upper halves (negation is the sign
    // bit, whatever the value): four vector instructions where three negate-and-select pairs are six
    const uint32_t flip = h.front ? 0u : 0x80000000u;
    auto with_sign = [flip](double x) {
        return __longlong_as_double((long long)((unsigned long long)__double_as_longlong(x) ^ ((unsigned long long)flip << 32)));
    };
    h.normal = mk(with_sign(outward.x), with_sign(outward.y), with_sign(outward.z));
}

// The same for the rects' outward normal +e_axis (xy_rect.rs:43-45 and its siblings): dot(dir, e_axis) is
// dir[axis] itself and a hit implies dir[axis] != 0, so the normal is e_axis with the opposite of that sign —
// three sign-bit operations and three selects instead of a dot product, a compare and six selects.  (The zero
// components are +0 on either face; the reference's negated (0, 0, 1) has -0 there, which no sum or product
// downstream can tell apart unless its other operand is an exact zero too.)
__device__ __forceinline__ void set_face_normal_axis(Hit &h, d3 dir, int axis) {
#ifdef RT_EXACT_DIV
    set_face_normal(h, dir, mk(axis == 0 ? 1.0 : 0.0, axis == 1 ? 1.0 : 0.0, axis == 2 ? 1.0 : 0.0));
#else
    auto against = [](double dk) { // +1.0 for dk < 0, -1.0 for dk > 0
        const uint32_t hi = 0xBFF00000u ^ ((uint32_t)((unsigned long long)__double_as_longlong(dk) >> 32) & 0x80000000u);
        return __longlong_as_double((long long)((unsigned long long)hi << 32));
    };
    h.normal = mk(axis == 0 ? against(dir.x) : 0.0, axis == 1 ? against(dir.y) : 0.0, axis == 2 ? against(dir.z) : 0.0);
    h.front = h.normal.x + h.normal.y + h.normal.z > 0.0; // only Dialectric reads it
#endif
}

__device__ __forceinline__ d3 rot_fwd(d3 a, double s, double c) { // rotate_y.rs:42-46
    return mk(c * a.x - s * a.z, a.y, s * a.x + c * a.z);
}
__device__ __forceinline__ d3 rot_back(d3 a, double s, double c) { // rotate_y.rs:55-59
    return mk(c * a.x + s * a.z, a.y, -s * a.x + c * a.z);
}

// One axis-aligned rect in its own frame; `axis` = constant axis.
// xy_rect.rs:29-40 / xz_rect.rs / yz_rect.rs
// EARLY_OUT: leave after the t-range test when no lane passes it.  That pays for free-standing
// rects (cornell's walls: -4 % without it) and costs for the six sides of a box, where the
// predicate form saves a wave twelve stops to test `exec` (cornell_box_boxes +6 %).
template <bool EARLY_OUT>
__device__ __forceinline__ bool rect_t(int axis, double a0, double a1, double b0, double b1, double k,
                                       d3 o, d3 d, d3 inv_d, double t_min, double t_max, double &t_out) {
    int ia = axis == 0 ? 1 : 0;
    int ib = axis == 2 ? 1 : 2;
    const double t = div_by(k - comp(o, axis), comp(d, axis), comp(inv_d, axis)); // xy_rect.rs:31 divides; inv_d = 1/d per ray
    if (EARLY_OUT) {
        if (t < t_min || t > t_max) return false;
        const double a = comp(o, ia) + t * comp(d, ia);
        const double b = comp(o, ib) + t * comp(d, ib);
        if (a < a0 || a > a1 || b < b0 || b > b1) return false;
        t_out = t;
        return true;
    }
    const double a = comp(o, ia) + t * comp(d, ia);
    const double b = comp(o, ib) + t * comp(d, ib);
    // the two early returns as one predicate (same comparisons, so NaNs fall the same way)
    const bool miss = (t < t_min) | (t > t_max) | (a < a0) | (a > a1) | (b < b0) | (b > b1);
    t_out = t;
    return !miss;
}

// The free-standing rect test of the linear loop, WRITTEN OUT (RT_ARITH_FAST): rect_t<true> above and the update of the
// closest hit, `if (hit) { best_t = t; best = i; }`, instruction for instruction — the same subtraction, product, two fma
// and six comparisons (so the same t, the same decisions, NaNs included) — but with the comparisons as v_cmpx, which
// narrow the exec mask themselves.  As the compiler builds it every rect costs 8 scalar instructions (three s_and_b64 of
// the comparison masks, two s_and_saveexec, two s_or exec, a branch) around its 12 vector ones, and the scalar unit — one
// per CU, shared by its four SIMDs — is what the rects-only variant waits for once the vector pipes are full
// (tools/bb_profile.py: 234 scalar instructions per iteration, 77 of them here).  This form has two, and the early out.
template <int AXIS>
__device__ __forceinline__ void rect_closest_update(double a0, double a1, double b0, double b1, double k, d3 o, d3 d, d3 inv_d,
                                                    double t_min, double &best_t, int &best, int i) {
    constexpr int IA = AXIS == 0 ? 1 : 0, IB = AXIS == 2 ? 1 : 2;
    double t, a, b;
    unsigned long long saved;
    asm volatile("v_add_f64 %[t], %[k], -%[oc]\n\t"
                 "v_mul_f64 %[t], %[inv], %[t]\n\t"          // xy_rect.rs:31 (k - o) / d with the ray's reciprocal
                 "s_mov_b64 %[saved], exec\n\t"
                 "v_cmpx_ngt_f64 %[tmin], %[t]\n\t"          // xy_rect.rs:32  not (t < t_min) ...
                 "v_cmpx_ngt_f64 %[t], %[bt]\n\t"            //                ... and not (t > t_max)
                 "s_cbranch_execz 1f\n\t"                    // no lane reaches the plane inside the range: the early out
                 "v_fma_f64 %[a], %[da], %[t], %[oa]\n\t"    // xy_rect.rs:35-36
                 "v_fma_f64 %[b], %[db], %[t], %[ob]\n\t"
                 "v_cmpx_ngt_f64 %[a0], %[a]\n\t"            // xy_rect.rs:37  not (a < a0), not (a > a1), not (b < b0), not (b > b1)
                 "v_cmpx_nlt_f64 %[a1], %[a]\n\t"
                 "v_cmpx_ngt_f64 %[b0], %[b]\n\t"
                 "v_cmpx_nlt_f64 %[b1], %[b]\n\t"
                 "v_mov_b64 %[bt], %[t]\n\t"                 // the lanes still enabled have a closer hit
                 "v_mov_b32 %[best], %[i]\n"
                 "1:\n\t"
                 "s_mov_b64 exec, %[saved]"
                 : [t] "=&v"(t), [a] "=&v"(a), [b] "=&v"(b), [saved] "=&s"(saved), [bt] "+v"(best_t), [best] "+v"(best)
                 : [k] "s"(k), [oc] "v"(comp(o, AXIS)), [inv] "v"(comp(inv_d, AXIS)), [tmin] "s"(t_min), [da] "v"(comp(d, IA)),
                   [oa] "v"(comp(o, IA)), [db] "v"(comp(d, IB)), [ob] "v"(comp(o, IB)), [a0] "s"(a0), [a1] "s"(a1), [b0] "s"(b0),
                   [b1] "s"(b1), [i] "s"(i)
                 : "vcc");
}

// box.rs:22-71 side s of a Boxx: axis and (a0,a1,b0,b1,k)
__device__ __forceinline__ void box_side(const double *p, int s, int &axis, double &a0, double &a1,
                                         double &b0, double &b1, double &k) {
    axis = 2 - (s >> 1);
    if (axis == 2) { a0 = p[0]; a1 = p[3]; b0 = p[1]; b1 = p[4]; k = (s & 1) ? p[2] : p[5]; }
    else if (axis == 1) { a0 = p[0]; a1 = p[3]; b0 = p[2]; b1 = p[5]; k = (s & 1) ? p[1] : p[4]; }
    else { a0 = p[1]; a1 = p[4]; b0 = p[2]; b1 = p[5]; k = (s & 1) ? p[0] : p[3]; }
}

// sphere.rs:39-59: nearest root of the half-b quadratic in [t_min, t_max]
__device__ __forceinline__ bool sphere_t(d3 center, double radius2, d3 o, d3 d, double inv_a, double t_min, double t_max,
                                         double &t_out) {
    const d3 oc = o - center;
    const double a = len2(d);
    const double half_b = dot(oc, d);
    const double c = len2(oc) - radius2;
    const double disc = half_b * half_b - a * c;
    if (disc < 0.0) return false;
    const double sqrtd = sqrt_fast(disc);
    double root = div_by(-half_b - sqrtd, a, inv_a); // sphere.rs:52 divides by a
    if (root < t_min || t_max < root) {
        root = div_by(-half_b + sqrtd, a, inv_a);
        if (root < t_min || t_max < root) return false;
    }
    t_out = root;
    return true;
}

// box.rs:82-101 — a Boxx is a list of six rects (box.rs:22-71), hit() keeps the side with the smallest t in
// [t_min, t_max] whose in-plane point lies inside the rect's inclusive bounds.  For an axis-aligned box those six tests
// are the three-slab test read twice: the ray is inside the slab pair of axis a for t between its two plane distances,
// inside the box for t in [t_enter, t_exit] = [max of the three nearer, min of the three farther], and the side
// hit() returns is the entry side when t_enter lies in the range, else the exit side (origin inside the box, or closer
// than t_min in front of it).  The six plane distances are the rect tests' own (k - o) / d, so a hit's t is the value
// the six-rect form returns, bit for bit; what differs is how "inside the rect" is decided for rays that graze an edge
// of the box to within rounding: by comparing plane distances here, by comparing the in-plane point with the bounds
// there (ties and such grazing rays are open, SURVEY B-15).  12 subtract/multiplies, 10 min/max and the side code
// instead of six rects of 10 vector instructions and a select chain each.
// `side` = box.rs's side index (box_side above) of the pair's first plane: 2 * (2 - axis).
// The RT_ARITH_REFERENCE kernels keep the six rect tests (the reference's own comparisons).
__device__ __forceinline__ bool box_slab_t(const double *p, d3 o, d3 d, d3 inv_d, double t_min, double t_max, double &t_out,
                                           int &side) {
    const double x_mn = div_by(p[0] - o.x, d.x, inv_d.x), x_mx = div_by(p[3] - o.x, d.x, inv_d.x);
    const double y_mn = div_by(p[1] - o.y, d.y, inv_d.y), y_mx = div_by(p[4] - o.y, d.y, inv_d.y);
    const double z_mn = div_by(p[2] - o.z, d.z, inv_d.z), z_mx = div_by(p[5] - o.z, d.z, inv_d.z);
    // (fmin / fmax drop a NaN — 0 * inf, origin exactly on a plane of a slab the ray runs parallel to — like the root
    // clip of the BVH walk: the slab then constrains nothing, as for the rect tests, whose NaN comparisons all fail)
    const double x_near = fmin(x_mn, x_mx), x_far = fmax(x_mn, x_mx);
    const double y_near = fmin(y_mn, y_mx), y_far = fmax(y_mn, y_mx);
    const double z_near = fmin(z_mn, z_mx), z_far = fmax(z_mn, z_mx);
    const double t_enter = fmax(fmax(x_near, y_near), z_near);
    const double t_exit = fmin(fmin(x_far, y_far), z_far);
    const bool entry = t_enter >= t_min;
    const double t = entry ? t_enter : t_exit;
    // Which pair of planes t belongs to: the axis whose nearer (farther) distance it equals.  Only the axis reaches the
    // hit record (box_side: the rect's normal axis and in-plane bounds; which of the pair's two planes it was shows in
    // nothing but t itself), so `side` is the even index of the pair.  Both candidates without a branch: as control
    // flow the backend built three nested exec-mask regions per box around four moves.
    const int side_in = t_enter == z_near ? 0 : (t_enter == y_near ? 2 : 4);
    const int side_out = t_exit == z_far ? 0 : (t_exit == y_far ? 2 : 4);
    side = entry ? side_in : side_out;
    t_out = t;
    return t_enter <= t_exit && t >= t_min && t <= t_max;
}

// Nearest t of primitive P in [t_min, t_max], wrappers applied
// (translate.rs:31, rotate_y.rs:39-48).  aux = box side.
template <int PRIMS>
// inv_d = 1/d (component-wise) and inv_a = 1/|d|^2 are computed once per ray.
// `time` is the ray's time (ray.rs:26-28); only MovingSphere reads it.
__device__ __forceinline__ bool prim_t(const Prim &P, d3 o, d3 d, d3 inv_d, double inv_a, double time, double t_min,
                                       double t_max, double &t_out, int &aux) {
    aux = 0;
    if (PRIMS == PRIMS_RECTS) { // untransformed rects only: kind picks the axis
        int kind = P.kind;
        if (kind == RT_PRIM_XY_RECT) return rect_t<true>(2, P.p[0], P.p[1], P.p[2], P.p[3], P.p[4], o, d, inv_d, t_min, t_max, t_out);
        if (kind == RT_PRIM_XZ_RECT) return rect_t<true>(1, P.p[0], P.p[1], P.p[2], P.p[3], P.p[4], o, d, inv_d, t_min, t_max, t_out);
        return rect_t<true>(0, P.p[0], P.p[1], P.p[2], P.p[3], P.p[4], o, d, inv_d, t_min, t_max, t_out);
    }
    if (PRIMS == PRIMS_ANY) {
        if (P.flags & RT_PRIM_HAS_TRANSLATE) o = o - ld3(P.tr);
        if (P.flags & RT_PRIM_HAS_ROTATE_Y) {
            o = rot_fwd(o, P.rot_sin, P.rot_cos);
            d = rot_fwd(d, P.rot_sin, P.rot_cos);
            inv_d.x = rcp_f64(d.x); // (d.y is untouched, and the rotation preserves |d|, so inv_a stands)
            inv_d.z = rcp_f64(d.z);
        }
    }
    switch (PRIMS == PRIMS_SPHERES ? (int)RT_PRIM_SPHERE : P.kind) {
    case RT_PRIM_MOVING_SPHERE: // moving_sphere.rs:37-39,49-69: same quadratic around the centre at `time`
    case RT_PRIM_SPHERE: { // sphere.rs:39-59
        d3 center = ld3(P.p);
        if (PRIMS == PRIMS_ANY && P.kind == RT_PRIM_MOVING_SPHERE)
            center = center + ((time - P.rot_sin) * P.rot_cos) * ld3(P.tr); // tr = pos_b - pos_a, rot_* = time_a, 1/(time_b - time_a)
        return sphere_t(center, P.radius2, o, d, inv_a, t_min, t_max, t_out);
    }
    case RT_PRIM_XY_RECT: return rect_t<true>(2, P.p[0], P.p[1], P.p[2], P.p[3], P.p[4], o, d, inv_d, t_min, t_max, t_out);
    case RT_PRIM_XZ_RECT: return rect_t<true>(1, P.p[0], P.p[1], P.p[2], P.p[3], P.p[4], o, d, inv_d, t_min, t_max, t_out);
    case RT_PRIM_YZ_RECT: return rect_t<true>(0, P.p[0], P.p[1], P.p[2], P.p[3], P.p[4], o, d, inv_d, t_min, t_max, t_out);
    default: { // box.rs:82-101
#ifndef RT_EXACT_DIV
        return box_slab_t(P.p, o, d, inv_d, t_min, t_max, t_out, aux);
#endif
        bool any = false;
        double closest = t_max;
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            int axis;
            double a0, a1, b0, b1, k, t;
            box_side(P.p, s, axis, a0, a1, b0, b1, k);
            if (rect_t<false>(axis, a0, a1, b0, b1, k, o, d, inv_d, t_min, closest, t)) {
                closest = t;
                aux = s;
                any = true;
            }
        }
        t_out = closest;
        return any;
    }
    }
}

// A record of the linear loop's BOX group (TraceArgs.box_end): a Boxx, bare or inside RotateY and / or Translate
// (translate.rs:31, rotate_y.rs:39-48; the YAML loader's only wrapped objects in the shipped scenes, sandbox.rs:39-80).
// The record is wave-uniform: its flags steer scalar branches, sin / cos / offset and the six bounds are SGPR operands.
// The ray is moved into the box's frame once (an unwrapped box's offset is (0, 0, 0): o - 0 is o), not once per side.
__device__ __forceinline__ bool box_t(const Prim &P, d3 o, d3 d, d3 inv_d, double t_min, double t_max, double &t_out, int &side) {
#ifdef RT_EXACT_DIV
    return prim_t<PRIMS_ANY>(P, o, d, inv_d, 0.0, 0.0, t_min, t_max, t_out, side);
#else
#ifndef RT_BOX_LOADS_LAZY
    // everything the test reads of the record is requested before the first use: one scalar-load wait per box instead of
    // three (offset + flags, sin / cos inside the rotated arm, the bounds): cornell_box_boxes +0.8 % at four waves per SIMD, +1.2 % at five
    asm volatile("; box record requested" ::"s"(P.p[0]), "s"(P.p[1]), "s"(P.p[2]), "s"(P.p[3]), "s"(P.p[4]), "s"(P.p[5]), "s"(P.rot_sin),
                 "s"(P.rot_cos), "s"(P.tr[0]), "s"(P.tr[1]), "s"(P.tr[2]), "s"(P.flags));
#endif
    o = o - ld3(P.tr);
    if (P.flags & RT_PRIM_HAS_ROTATE_Y) {
        o = rot_fwd(o, P.rot_sin, P.rot_cos);
        d = rot_fwd(d, P.rot_sin, P.rot_cos);
        inv_d.x = rcp_f64(d.x);
        inv_d.z = rcp_f64(d.z);
    }
    return box_slab_t(P.p, o, d, inv_d, t_min, t_max, t_out, side);
#endif
}

// Closest hit through the skip-link BVH (rt_bvh.h): one integer of traversal
// state per lane, no stack.  Same acceptance rule as the brute-force loop
// (t in [t_min, best_t], later equal hit wins), so only exact ties can differ.
//
// "while-while" shape: every lane first walks inner nodes until it stands on a
// leaf whose box it hits (or has left the tree), THEN the wave tests leaf
// primitives together.  Mixing the two in one loop body makes a wave pay the
// primitive tests on almost every step (measured: LABNOTES.md, round-3 section 4.6).
// The node array a ray of direction d walks when the tree lives in global memory: the copy ordered for d's sign octant
// (rt_bvh.cpp), or the one array there is.  (Chosen by the CALLER of closest_hit_bvh: selecting inside it would merge the
// LDS and the global pointer into a flat one and turn the LDS walk's ds_read into flat loads — `random` 39 -> 51 ms.)
__device__ __forceinline__ const BvhNode *bvh_nodes_for(const TraceArgs &A, d3 d) {
    if (A.bvh_nodes_ordered == nullptr) return A.bvh_nodes;
    const int oct = (d.x < 0.0 ? 1 : 0) | (d.y < 0.0 ? 2 : 0) | (d.z < 0.0 ? 4 : 0);
    return A.bvh_nodes_ordered + (size_t)oct * (size_t)(A.n_bvh_nodes + 1);
}

struct NoMark { // profile hook of closest_hit_bvh: the regions build passes one that books cycles
    __device__ __forceinline__ void operator()(int) const {}
};
template <int PRIMS, class Mark = NoMark>
__device__ __forceinline__ void closest_hit_bvh(const TraceArgs &A, const BvhNode *nodes, d3 o, d3 d, d3 inv_d,
                                                double inv_a, double time, double t_min, double &best_t, int &best,
                                                int &best_aux, unsigned *walk_stats = nullptr, Mark mark = Mark()) {
    // CULLING IN SINGLE PRECISION.  The node boxes only decide which primitives get tested (in f64, as everywhere),
    // so they are f32 boxes around the root's centre, 32 B per node.  The ray is first clipped to the root box in
    // f64: from there its origin is within the scene's extent E of the centre, the f32 plane distances below are
    // off by a few 2^-24 E in the plane position — the boxes are padded by 2^-20 E (rt_bvh.cpp) — and by 2^-23
    // relative in t, for which the interval test carries a 2^-20 relative slack.
    const RT_CONSTANT TraceArgs *K = kernargs_here();
    double t0 = 0.0; // ray parameter of the clipped origin
    {
        const double ax = (K->bvh_root_mn[0] - o.x) * inv_d.x, bx = (K->bvh_root_mx[0] - o.x) * inv_d.x;
        const double ay = (K->bvh_root_mn[1] - o.y) * inv_d.y, by = (K->bvh_root_mx[1] - o.y) * inv_d.y;
        const double az = (K->bvh_root_mn[2] - o.z) * inv_d.z, bz = (K->bvh_root_mx[2] - o.z) * inv_d.z;
        // fmin/fmax drop NaNs (0 * inf on a slab boundary), which keeps the test conservative
        const double t_enter = fmax(fmax(fmin(ax, bx), fmin(ay, by)), fmin(az, bz));
        const double t_exit = fmin(fmin(fmax(ax, bx), fmax(ay, by)), fmin(fmax(az, bz), best_t));
        if (!(fmax(t_enter, t_min) <= t_exit)) return; // misses the scene's bounds
        if (t_enter > 0.0) t0 = t_enter;
    }
    const SlabRay sr = slab_ray((float)(fma(t0, d.x, o.x) - K->bvh_center[0]), (float)(fma(t0, d.y, o.y) - K->bvh_center[1]),
                                (float)(fma(t0, d.z, o.z) - K->bvh_center[2]), inv_d.x, inv_d.y, inv_d.z); // rt_bvh_slab.h
    const float slack = 0x1p-20f; // of the WINDOW ends only (f64 -> f32, rounded outward); the box test itself carries none
    // the window [t_min, best_t] seen from the clipped origin, rounded outward
    const float tmin_f = (float)(t_min - t0) - fabsf((float)(t_min - t0)) * slack - 0x1p-126f;
    auto far_of = [&](double bt) { // upper end of the window; never below its lower end, so that the sentinel is always "hit"
        const float f = (float)(bt - t0);
        return fmaxf(f + fabsf(f) * slack, tmin_f);
    };
    float best_f = far_of(best_t);
    if (!(tmin_f <= best_f)) return; // NaN somewhere in the ray: nothing can be hit (and the walk below relies on the order)
    // the ray's six constants as the three register pairs v_pk_fma_f32 reads (SlabRay's order)
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 p01 = {sr.ivx, sr.ivy}, p23 = {sr.ivz, sr.nox}, p45 = {sr.noy, sr.noz};
    int i = 0;
    const int n = A.n_bvh_nodes;
    while (i < n) {
        int fc = 0;
        // DESCENT: skip / step down until a leaf is entered.  The array ends with a sentinel — an all-space leaf without
        // primitives at index n, where every finished walk arrives through its last skip link — so the loop has ONE
        // exit and no `i < n` test per step.  It ends for every lane: the index grows with every step (i + 1, or a skip
        // link, which points forward), and at index n the test below cannot fail — the sentinel's planes are at -inf /
        // +inf (or NaN, which v_min / v_max drop), leaving t_near = tmin_f and t_far = best_f, which are ordered and not
        // NaN (checked above; far_of keeps best_f there).  The step is what the walk is made of (58 % of the
        // BVH variant's vector instructions, profiles/r04_random_opcode_hist.txt), so it is written out: both planes of
        // an axis in one v_pk_fma_f32 (the node keeps them side by side), min / max as the instructions, no slack
        // fma (rt_bvh_slab.h has the error budget), the leaf test on the raw first_count word.
        for (;;) {
            if (walk_stats) ++walk_stats[0]; // profile build: nodes visited
            const uint4 *raw = reinterpret_cast<const uint4 *>(&nodes[i]);
            const uint4 q0 = raw[0], q1 = raw[1]; // the whole 32-byte node in two 128-bit reads BEFORE the test
            const f2 qx = {__uint_as_float(q0.x), __uint_as_float(q0.y)}, qy = {__uint_as_float(q0.z), __uint_as_float(q0.w)},
                     qz = {__uint_as_float(q1.x), __uint_as_float(q1.y)};
            f2 tx, ty, tz;
            float nx, ny, nz, fx, fy, fz, t_near, t_far;
            // (lo, hi) * (iv, iv) + (no, no): op_sel picks the half of each 64-bit source for the low result, op_sel_hi for the high one
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(tx) : "v"(qx), "v"(p01), "v"(p23)); // x: ivx = p01.lo, nox = p23.hi
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "=v"(ty) : "v"(qy), "v"(p01), "v"(p45)); // y: ivy = p01.hi, noy = p45.lo
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(tz) : "v"(qz), "v"(p23), "v"(p45)); // z: ivz = p23.lo, noz = p45.hi
            // (one statement: between separate asm statements the compiler puts an s_nop for hazards it cannot rule out)
            asm("v_min_f32 %0, %8, %9\n\tv_max_f32 %1, %8, %9\n\t"
                "v_min_f32 %2, %10, %11\n\tv_max_f32 %3, %10, %11\n\t"
                "v_min_f32 %4, %12, %13\n\tv_max_f32 %5, %12, %13\n\t"
                "v_max_f32 %4, %4, %14\n\tv_min_f32 %5, %5, %15\n\t"
                "v_max3_f32 %6, %0, %2, %4\n\tv_min3_f32 %7, %1, %3, %5"
                : "=&v"(nx), "=&v"(fx), "=&v"(ny), "=&v"(fy), "=&v"(nz), "=&v"(fz), "=&v"(t_near), "=&v"(t_far)
                : "v"(tx.x), "v"(tx.y), "v"(ty.x), "v"(ty.y), "v"(tz.x), "v"(tz.y), "v"(tmin_f), "v"(best_f));
            const bool hit = !(t_near > t_far); // (a NaN, which the window's ends exclude, would step on: the walk still ends)
            fc = (int)q1.w;
            i = hit ? i + 1 : (int)q1.z; // inner: first child; leaf: its skip link is i + 1 as well
            if (hit && fc != 0) break; // a leaf (or the sentinel) was entered
        }
        const int count = fc & 7, first = fc >> 3; // the sentinel: no primitives, and i = n + 1 ends the walk
        mark(11); // profile build: the descent
        bool improved = false;
        for (int k = 0; k < count; ++k) { // the leaf's primitives (stored contiguously in leaf order)
            if (walk_stats) ++walk_stats[1]; // profile build: primitives tested
            const int pi = first + k;
            double t;
            int aux = 0;
            bool hit;
            // The compact record is read and the sphere test run WITHOUT asking the tag first: behind `if (G.tag == 0)`
            // a leaf primitive cost two dependent round trips to global memory (the tag, then the record), and the leaf
            // tests were 30 % of the `random` frame's wave time for 14 % of its vector instructions
            // (profiles/r04_region_cycles.txt).  A record of another kind (tag 1) holds zeros here: its sphere test is
            // discarded and the general routine runs for it — in scenes of spheres, never.
            const LeafGeo &G = A.leaf_geo[pi];
            const long long tag = G.tag;
            const d3 center = ld3(G.c0) + ((time - A.leaf_time_a) * A.leaf_inv_dt) * ld3(G.dc); // sphere.rs:39-59 / moving_sphere.rs:37-39,49-69
            hit = sphere_t(center, G.radius2, o, d, inv_a, t_min, best_t, t) && tag == 0;
            if (tag != 0) hit = prim_t<PRIMS>(A.prims[pi], o, d, inv_d, inv_a, time, t_min, best_t, t, aux);
            if (hit) {
                best_t = t;
                best = pi;
                best_aux = aux;
                improved = true;
            }
        }
        if (improved) best_f = far_of(best_t);
        mark(12); // profile build: the leaf
    }
}

// sphere.rs:20-27; out of line: acos/atan2 are large and only image textures read u,v.  Returned BY VALUE:
// with reference parameters the caller's whole Hit record became addressable and lived in scratch memory,
// written on every hit of every textured variant.
struct UV {
    double u, v;
};
__device__ __noinline__ UV sphere_uv(d3 outward) {
    const double PI = 3.14159265358979323846;
    double theta = acos(-outward.y);
    double phi = atan2(-outward.z, outward.x) + PI;
#ifdef RT_EXACT_DIV
    return UV{phi / (2.0 * PI), theta / PI};
#else
    return UV{phi * (1.0 / (2.0 * PI)), theta * (1.0 / PI)}; // sphere.rs:25-26 divide; one rounding apart at most
#endif
}

// sphere.rs:20-27 in SINGLE precision, for the one consumer of a sphere's (u, v): the nearest-texel lookup of an image
// texture (texture/image.rs:28-51), which only asks which cell of a w x h grid (u, v) falls into.  acos + atan2 in f64 are
// ~250 vector instructions at 4 cycles, run for the handful of lanes of a wave that hit the textured sphere (C4: 6 % of
// the frame); in f32 they are ~70 at 2 cycles.  The result is within UV_EPS_U / UV_EPS_V of the exact pair — inputs
// rounded to f32 (6e-8 relative), acosf / atan2f good to a few ulp of pi (5e-7), the rest in f64 — PROVIDED the point is
// not near a pole, where acos' derivative 1 / sqrt(1 - y^2) amplifies the input rounding: the caller keeps the f64 form
// for 1 - y^2 < 2^-10.  image_texel() certifies every use: when a cell boundary lies within the bound of the value it
// recomputes (u, v) exactly, so the texel — hence the frame — is the one the f64 form selects, bit for bit.
#define RT_UV_EPS_U 5e-7
#define RT_UV_EPS_V 2e-6
__device__ __forceinline__ UV sphere_uv_f32(d3 outward) {
    const float theta = acosf(-(float)outward.y);
    const float phi = atan2f(-(float)outward.z, (float)outward.x);
    return UV{((double)phi + 3.14159265358979323846) * (1.0 / (2.0 * 3.14159265358979323846)),
              (double)theta * (1.0 / 3.14159265358979323846)};
}

// Rebuild the HitRecord of the winning primitive (geometry.rs:17-57).
// FAST_UV: a plain sphere's (u, v) may come from sphere_uv_f32 (Hit.uv_approx says so); the pooled kernel's texture code
// certifies it, the v1 kernel keeps the f64 form.
template <int PRIMS, bool TEXTURED, bool FAST_UV = false>
__device__ __forceinline__ Hit prim_hit_record(const Prim &P, d3 o, d3 d, double time, double t, int aux, bool want_uv) {
    d3 oo = o, dd = d;
    const int flags = PRIMS == PRIMS_ANY ? P.flags : 0;
    if (flags & RT_PRIM_HAS_TRANSLATE) oo = oo - ld3(P.tr);
    if (flags & RT_PRIM_HAS_ROTATE_Y) {
        oo = rot_fwd(oo, P.rot_sin, P.rot_cos);
        dd = rot_fwd(dd, P.rot_sin, P.rot_cos);
    }
    Hit h;
    h.point = oo + t * dd; // ray.rs:30-32
    h.u = 0.0;
    h.v = 0.0;
    h.uv_approx = false;
    const int kind = PRIMS == PRIMS_SPHERES ? (int)RT_PRIM_SPHERE : P.kind;
    if (PRIMS == PRIMS_ANY && kind == RT_PRIM_MOVING_SPHERE) {
        const d3 center = ld3(P.p) + ((time - P.rot_sin) * P.rot_cos) * ld3(P.tr);
        d3 outward = (h.point - center) * P.inv_radius; // moving_sphere.rs:75
        if (TEXTURED && want_uv) { // moving_sphere.rs:76: uv of the POINT (SURVEY B-19)
            const UV uv = sphere_uv(h.point);
            h.u = uv.u;
            h.v = uv.v;
        }
        set_face_normal(h, dd, outward);
    } else if (PRIMS != PRIMS_RECTS && kind == RT_PRIM_SPHERE) {
        d3 outward = (h.point - ld3(P.p)) * P.inv_radius; // sphere.rs:61
        if (TEXTURED && want_uv) {
#ifndef RT_EXACT_DIV
            // unwrapped sphere away from the poles: f32 now, certified (or redone in f64) where it is used
            if (FAST_UV && flags == 0 && fma(-outward.y, outward.y, 1.0) >= 0x1p-10) {
                const UV uv = sphere_uv_f32(outward);
                h.u = uv.u;
                h.v = uv.v;
                h.uv_approx = true;
            } else
#endif
            {
                const UV uv = sphere_uv(outward);
                h.u = uv.u;
                h.v = uv.v;
            }
        }
        set_face_normal(h, dd, outward);
    } else {
        int axis;
        double a0, a1, b0, b1, k;
        if (PRIMS == PRIMS_ANY && kind == RT_PRIM_BOX) {
            box_side(P.p, aux, axis, a0, a1, b0, b1, k);
        } else {
            axis = kind == RT_PRIM_XY_RECT ? 2 : (kind == RT_PRIM_XZ_RECT ? 1 : 0);
            a0 = P.p[0]; a1 = P.p[1]; b0 = P.p[2]; b1 = P.p[3];
        }
        if (TEXTURED && want_uv) { // xy_rect.rs:41-42
            int ia = axis == 0 ? 1 : 0;
            int ib = axis == 2 ? 1 : 2;
            h.u = (comp(h.point, ia) - a0) / (a1 - a0);
            h.v = (comp(h.point, ib) - b0) / (b1 - b0);
        }
        set_face_normal_axis(h, dd, axis);
    }
    if (flags & RT_PRIM_HAS_ROTATE_Y) { // rotate_y.rs:52-63 (face test vs the rotated ray)
        h.point = rot_back(h.point, P.rot_sin, P.rot_cos);
        set_face_normal(h, dd, rot_back(h.normal, P.rot_sin, P.rot_cos));
    }
    if (flags & RT_PRIM_HAS_TRANSLATE) { // translate.rs:34-37 (re-runs set_face_normal)
        h.point = h.point + ld3(P.tr);
        set_face_normal(h, d, h.normal);
    }
    return h;
}

// --------------------------------------------------------------- textures
// IDENTITY: the three permutation tables are the identity.  They always are in the reference —
// Perlin::permute loops over `(count - 1)..0`, an empty range (noise.rs:121-130, SURVEY B-9) — so
// rt_scene_create checks the uploaded tables and the lattice hash becomes (i ^ j ^ k) & 255 without
// three table reads per corner; other tables take the general path.
template <bool IDENTITY>
__device__ __forceinline__ double perlin_noise(const Perlin &pl, d3 p) { // noise.rs:57-96
    double fx = floor(p.x), fy = floor(p.y), fz = floor(p.z);
    double u = p.x - fx, v = p.y - fy, w = p.z - fz;
    int i = (int)fx, j = (int)fy, k = (int)fz; // saturating on AMDGCN like Rust's `as i32`
    double uu = u * u * (3.0 - 2.0 * u);
    double vv = v * v * (3.0 - 2.0 * v);
    double ww = w * w * (3.0 - 2.0 * w);
    // noise.rs:84-91 weights corner (di, dj, dk) with (di*uu + (1-di)*(1-uu)) * (dj*vv + ...) * (dk*ww + ...);
    // with di in {0, 1} that sum is exactly uu or (1 - uu) (x*1 + y*0 = x for finite y), so the
    // factors are formed once per axis and multiplied in the same order
    const double fu[2] = {1.0 - uu, uu}, fv[2] = {1.0 - vv, vv}, fw[2] = {1.0 - ww, ww};
    double accum = 0.0;
#pragma unroll
    for (int di = 0; di < 2; ++di)
#pragma unroll
        for (int dj = 0; dj < 2; ++dj) {
            const double fuv = fu[di] * fv[dj];
#pragma unroll
            for (int dk = 0; dk < 2; ++dk) {
                int index;
                if (IDENTITY)
                    index = (i + di) ^ (j + dj) ^ (k + dk);
                else
                    index = pl.perm_x[(uint32_t)(i + di) & 255u] ^ pl.perm_y[(uint32_t)(j + dj) & 255u] ^
                            pl.perm_z[(uint32_t)(k + dk) & 255u];
                const double *g = pl.ranvec[index & 255];
                d3 weight = mk(u - di, v - dj, w - dk);
                accum += fuv * fw[dk] * dot(ld3(g), weight);
            }
        }
    return accum;
}

template <bool IDENTITY>
__device__ __forceinline__ double perlin_turbulence(const Perlin &pl, d3 p, int depth) { // noise.rs:98-109
    double accum = 0.0, weight = 1.0;
    for (int o = 0; o < depth; ++o) {
        accum += weight * perlin_noise<IDENTITY>(pl, p);
        weight *= 0.5;
        p = p * 2.0;
    }
    return fabs(accum);
}

__device__ __forceinline__ double clamp01(double x) { return x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x); }

// sin(x) for |x| up to ~1e6 with a 3-part Cody-Waite reduction by pi/2 (each
// part has 33 significant bits, so k * part is exact for k < 2^20) and the
// classic degree-13/14 kernels on [-pi/4, pi/4].  About 35 f64 instructions, no
// table, no scratch — unlike the general-range library sin, whose Payne-Hanek
// path forces 400+ bytes of scratch on every wave of the kernel.  Within ~1 ulp
// of the reference's libm sin; checkered.rs only uses the sign of a product of
// sines and noise.rs feeds sin(scale*z + 10*turb) of scene-sized arguments.
__device__ __forceinline__ double sin_lean(double x) {
    const double kd = rint(x * 0.63661977236758134308); // 2/pi
    double r = fma(-kd, 1.57079632673412561417e+00, x);
    r = fma(-kd, 6.07710050630396597660e-11, r);
    r = fma(-kd, 2.02226624871116645580e-21, r);
    const int q = (int)kd;
    const double z = r * r;
    // sine kernel
    const double ps = fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
                                        2.75573137070700676789e-06), -1.98412698298579493134e-04),
                          8.33333333332248946124e-03);
    const double sn = fma(z * r, fma(z, ps, -1.66666666666666324348e-01), r);
    // cosine kernel
    const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
                                               -2.75573143513906633035e-07), 2.48015872894767294178e-05),
                                 -1.38888888888741095749e-03), 4.16666666666666019037e-02);
    const double cs = 1.0 - fma(0.5, z, -(z * z) * pc);
    const double v = (q & 1) ? cs : sn;
    return (q & 2) ? -v : v;
}

// Sign of sin(x) (-1, 0, +1) from sin_lean's own argument reduction: with x = k*pi/2 + r,
// |r| <= pi/4, sin(x) is sin(r), cos(r), -sin(r), -cos(r) for k mod 4 = 0..3; cos(r) > 0 and
// sin(r) has the sign of r, so no polynomial is needed.  checkered.rs:36-41 only asks whether
// the product of three sines is negative.
__device__ __forceinline__ int sin_sign(double x) {
    const double kd = rint(x * 0.63661977236758134308); // 2/pi
    double r = fma(-kd, 1.57079632673412561417e+00, x);
    r = fma(-kd, 6.07710050630396597660e-11, r);
    r = fma(-kd, 2.02226624871116645580e-21, r);
    const int q = (int)kd;
    const int s = (q & 1) ? 1 : (r > 0.0 ? 1 : (r < 0.0 ? -1 : 0));
    return (q & 2) ? -s : s;
}

// checkered.rs:36-41's whole question in one go: is sin(a) * sin(b) * sin(c) < 0 ?  That is: no factor is zero and an
// odd number of them is negative.  Per factor (sin_sign above): it is zero iff k is even and r == 0, negative iff
// bit 1 of k differs from (k even and r < 0).  The three signs are combined as BITS — the sign bit of r, bits 0 and 1
// of k — instead of three -1/0/+1 integers built with compares and selects and then multiplied (18 vector instructions
// per factor, 12 now; the same predicate, so the same texture side for every point).
__device__ __forceinline__ bool sines_product_negative(double a, double b, double c) {
#ifndef RT_EXACT_DIV
    // sin(x) < 0 exactly when floor(x / pi) is odd, and sin(x) = 0 for a double x only at x = 0: the predicate is the
    // parity of three floors.  x * (1 / pi) in one rounded product is off by 2^-52 relative, so it names the wrong
    // half-period only for an x within |x| 2.2e-16 of a multiple of pi — for the |x| <= 1e4 of a checkered surface a
    // chance of 1e-12 per factor, where the three-term Cody-Waite reduction below is exact to 1e-21 k.  That buys 35 of the
    // 53 vector instructions this test ran on EVERY iteration of noise_and_textures (its ground is the checker):
    // 4.3 % of C4's vector instructions (profiles/r04_c4_opcode_hist.txt).  RT_ARITH_REFERENCE keeps the reduction.
    const double inv_pi = 0.31830988618379067154;
    const uint32_t odd = (uint32_t)(int)floor(a * inv_pi) ^ (uint32_t)(int)floor(b * inv_pi) ^ (uint32_t)(int)floor(c * inv_pi);
    const bool zero = a * b * c == 0.0; // a factor sin(0) = 0 makes the product 0, which is not < 0 (|x| < 1e-100 aside)
    return !zero && (odd & 1u) != 0u;
#else
    uint32_t negative = 0;
    bool zero = false;
    const double xs[3] = {a, b, c};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double kd = rint(xs[k] * 0.63661977236758134308); // 2/pi
        double r = fma(-kd, 1.57079632673412561417e+00, xs[k]);
        r = fma(-kd, 6.07710050630396597660e-11, r);
        r = fma(-kd, 2.02226624871116645580e-21, r);
        const uint32_t q = (uint32_t)(int)kd;
        const uint32_t r_neg = (uint32_t)((unsigned long long)__double_as_longlong(r) >> 63); // r < 0 (r is never -0 with r != 0 ... and -0 counts as zero below)
        zero = zero || ((q & 1u) == 0u && r == 0.0);
        negative ^= (q >> 1) ^ (~q & r_neg); // bit 0: this factor is negative (when it is not zero)
    }
    return !zero && (negative & 1u) != 0u;
#endif
}

// Texture::value for everything that is not a plain SolidColor.
// lds_perlin: LDS copy of A.perlins[0], or nullptr.
// `textures`: the texture table (A.textures, or the pooled kernel's LDS copy of it).
__device__ __forceinline__ d3 texture_value_full(const TraceArgs &A, const Perlin *lds_perlin, const Texture *textures, int ti,
                                                 double u, double v, d3 p) {
    const Texture *T = &textures[ti];
    if (T->kind == RT_TEX_CHECKERED) { // checkered.rs:32-42
        // sines = sin(10x) * sin(10y) * sin(10z) < 0  <=>  an odd number of negative factors, none zero
        const int sines = sin_sign(p.x * 10.0) * sin_sign(p.y * 10.0) * sin_sign(p.z * 10.0);
        T = &textures[sines < 0 ? T->tex_odd : T->tex_even];
    }
    const int kind = T->kind;
    if (kind == RT_TEX_IMAGE) { // texture/image.rs:28-51
        const Image img = A.images[T->image];
        const double uu = clamp01(u);
        const double vv = 1.0 - clamp01(v);
        double i = uu * (double)img.width;
        double j = vv * (double)img.height;
        if (i >= (double)img.width) i = (double)img.width - 1.0;
        if (j >= (double)img.height) j = (double)img.height - 1.0;
        const uint32_t xi = (uint32_t)i, yj = (uint32_t)j; // saturating, NaN -> 0
        const uchar4 px = reinterpret_cast<const uchar4 *>(img.rgba)[(size_t)yj * (size_t)img.width + xi];
        const double s = 1.0 / 255.0;
        return mk((double)px.x * s, (double)px.y * s, (double)px.z * s);
    }
    if (kind == RT_TEX_NOISE) { // noise.rs:26-33
        // The pooled kernel stages the first table's gradients in LDS when the permutations are the
        // identity (the reference's case); anything else reads its table from global memory.
        double turb;
        if (lds_perlin != nullptr && T->perlin == 0) turb = perlin_turbulence<true>(*lds_perlin, p, T->depth);
        else turb = perlin_turbulence<false>(A.perlins[T->perlin], p, T->depth);
        const double f = 1.0 + sin_lean(T->scale * p.z + 10.0 * turb);
        return (ld3(T->color) * 0.5) * f;
    }
    return ld3(T->color);
}

// Texture::value in two steps for the pooled kernel.  Step 1 (per lane, inside the divergent shading
// code): everything except the Perlin turbulence of a Noise texture; for a Noise texture the lane only
// notes WHICH one (`noise_tex`, after a Checkered parent has picked its side) and returns zeros.
// Step 2 (whole wave, coop_noise_turbulence in rt_trace_pool_kernel.hip) evaluates the turbulence of all
// noted lookups together, and noise_colour() finishes noise.rs:26-33.
// texture/image.rs:28-51: the cell of a width x height grid that (u, v) selects (v flipped, both clamped)
__device__ __forceinline__ void image_cell(double u, double v, int width, int height, uint32_t &xi, uint32_t &yj) {
    const double uu = clamp01(u);
    const double vv = 1.0 - clamp01(v);
    double i = uu * (double)width;
    double j = vv * (double)height;
    if (i >= (double)width) i = (double)width - 1.0;
    if (j >= (double)height) j = (double)height - 1.0;
    xi = (uint32_t)i; // saturating, NaN -> 0
    yj = (uint32_t)j;
}
// Is a cell boundary within +-eps_u / +-eps_v of (u, v)?  (Only then can an approximate pair select another cell
// than the exact one; the clamps are monotone, so they cannot create a difference that the unclamped grid does not have.)
__device__ __forceinline__ bool image_cell_uncertain(double u, double v, int width, int height) {
    const double iu = u * (double)width, jv = v * (double)height;
    const double fu = iu - floor(iu), fv = jv - floor(jv);
    const double mu = RT_UV_EPS_U * (double)width, mv = RT_UV_EPS_V * (double)height;
    return fu < mu || fu > 1.0 - mu || fv < mv || fv > 1.0 - mv;
}

// uv_approx: (u, v) are sphere_uv_f32's; `exact_uv()` recomputes them in f64 (called only when a lookup cannot be certified)
template <class ExactUV>
__device__ __forceinline__ d3 texture_value_deferred(const TraceArgs &A, const Texture *textures, int ti, double u, double v,
                                                     d3 p, int &noise_tex, bool uv_approx, ExactUV exact_uv) {
    const Texture *T = &textures[ti];
    if (T->kind == RT_TEX_CHECKERED) { // checkered.rs:32-42
        ti = sines_product_negative(p.x * 10.0, p.y * 10.0, p.z * 10.0) ? T->tex_odd : T->tex_even;
        T = &textures[ti];
    }
    const int kind = T->kind;
    if (kind == RT_TEX_IMAGE) { // texture/image.rs:28-51
        const int width = T->img.width, height = T->img.height;
        if (uv_approx && image_cell_uncertain(u, v, width, height)) { // a cell boundary within the f32 pair's error: redo in f64
            const UV e = exact_uv();
            u = e.u;
            v = e.v;
        }
        uint32_t xi, yj;
        image_cell(u, v, width, height, xi, yj);
        const uchar4 px = reinterpret_cast<const uchar4 *>(T->img.rgba)[(size_t)yj * (size_t)width + xi];
        const double s = 1.0 / 255.0;
        return mk((double)px.x * s, (double)px.y * s, (double)px.z * s);
    }
    if (kind == RT_TEX_NOISE) {
        noise_tex = ti;
        return mk(0.0, 0.0, 0.0);
    }
    return ld3(T->color);
}

// noise.rs:26-33 once the turbulence is known: color * 0.5 * (1 + sin(scale * p.z + 10 * turb))
__device__ __forceinline__ d3 noise_colour(const Texture &T, d3 p, double turb) {
    const double f = 1.0 + sin_lean(T.scale * p.z + 10.0 * turb);
    return (ld3(T.color) * 0.5) * f;
}

template <bool TEXTURED>
__device__ __forceinline__ d3 texture_value(const TraceArgs &A, const Perlin *lds_perlin, const Texture *textures,
                                            const Material &M, double u, double v, d3 p) {
    if (!TEXTURED || M.tex_kind == RT_TEX_SOLID_COLOR) return ld3(M.color); // solid_color.rs:24-28
    return texture_value_full(A, lds_perlin, textures, M.texture, u, v, p);
}

} // namespace RT_KNS
