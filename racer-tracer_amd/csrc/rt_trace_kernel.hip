// rt_trace_kernel.hip — the per-pixel path-tracing hot path on gfx950 (f64).
//
// What it replaces: CpuRenderer::raytrace's pixel x sample loop
// (racer-tracer/src/renderer/cpu.rs:26-71) with the recursive ray_color
// (renderer.rs:41-90) flattened into an iterative, register-resident form:
//
//   * one lane owns one pixel and keeps {ray, throughput, pixel sum} in
//     VGPRs; when its path ends it immediately REGENERATES the next sample
//     of its pixel, so a wave never waits for its longest path and no ray
//     state ever travels through HBM;
//   * every random draw is addressed by (pixel, sample, segment, purpose,
//     iteration) under the contract of include/rt_rng.h, so lanes carry no
//     generator state and dead draws are skipped;
//   * the closest-hit loop walks the primitive table with a wave-uniform
//     index (scalar loads from the kernarg pointer), brute force, which for
//     the <= few dozen primitives of the reference's scenes beats a BVH walk
//     with its divergent stack.
//
// Arithmetic follows SURVEY.md App. A (the same formulas oracle/trace.c
// restates); fused multiply-adds and reciprocal-multiply for the shared
// 1/|d|^2 are the only liberties, worth a few ulps.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rt_device_types.h"
#include "../../include/rt_abi.h"
#include "../../include/rt_rng.h"

namespace rtdev {

enum { PRIMS_RECTS = 0, PRIMS_SPHERES = 1, PRIMS_ANY = 2 };

// ------------------------------------------------------------------ vec3
struct d3 {
    double x, y, z;
};
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
__device__ __forceinline__ double comp(d3 a, int axis) { return axis == 0 ? a.x : (axis == 1 ? a.y : a.z); }

// ------------------------------------------------------------------- RNG
struct u4 {
    uint32_t a, b, c, d;
};

__device__ __forceinline__ u4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                            uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)RT_PHILOX_M0 * c0;
        uint64_t p1 = (uint64_t)RT_PHILOX_M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
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

struct PathRng {
    uint32_t pixel, sample, k0, k1;
    __device__ __forceinline__ u4 block(uint32_t segment, uint32_t purpose, uint32_t blk) const {
        return philox4x32_10(pixel, sample, (segment << 8) | purpose, blk, k0, k1);
    }
};

// vec3.rs:424-430 random_in_unit_sphere under the addressed-draw contract
__device__ __forceinline__ d3 random_in_unit_sphere(const PathRng &rng, uint32_t segment) {
    for (uint32_t i = 0;; ++i) {
        u4 b0 = rng.block(segment, RT_RNG_SCATTER, 2 * i);
        u4 b1 = rng.block(segment, RT_RNG_SCATTER, 2 * i + 1);
        d3 p = mk(fma(2.0, u53(b0.a, b0.b), -1.0), fma(2.0, u53(b0.c, b0.d), -1.0),
                  fma(2.0, u53(b1.a, b1.b), -1.0));
        if (len2(p) >= 1.0) continue;
        return p;
    }
}

// --------------------------------------------------------------- geometry
struct Hit {
    d3 point, normal;
    double u, v;
    bool front;
};

__device__ __forceinline__ void set_face_normal(Hit &h, d3 dir, d3 outward) { // geometry.rs:49-56
    h.front = dot(dir, outward) < 0.0;
    h.normal = h.front ? outward : -outward;
}

__device__ __forceinline__ d3 rot_fwd(d3 a, double s, double c) { // rotate_y.rs:42-46
    return mk(c * a.x - s * a.z, a.y, s * a.x + c * a.z);
}
__device__ __forceinline__ d3 rot_back(d3 a, double s, double c) { // rotate_y.rs:55-59
    return mk(c * a.x + s * a.z, a.y, -s * a.x + c * a.z);
}

// One axis-aligned rect in its own frame; `axis` = constant axis.
// xy_rect.rs:29-40 / xz_rect.rs / yz_rect.rs
__device__ __forceinline__ bool rect_t(int axis, double a0, double a1, double b0, double b1, double k,
                                       d3 o, d3 d, double t_min, double t_max, double &t_out) {
    int ia = axis == 0 ? 1 : 0;
    int ib = axis == 2 ? 1 : 2;
    double t = (k - comp(o, axis)) / comp(d, axis);
    if (t < t_min || t > t_max) return false;
    double a = comp(o, ia) + t * comp(d, ia);
    double b = comp(o, ib) + t * comp(d, ib);
    if (a < a0 || a > a1 || b < b0 || b > b1) return false;
    t_out = t;
    return true;
}

// box.rs:22-71 side s of a Boxx: axis and (a0,a1,b0,b1,k)
__device__ __forceinline__ void box_side(const double *p, int s, int &axis, double &a0, double &a1,
                                         double &b0, double &b1, double &k) {
    axis = 2 - (s >> 1);
    if (axis == 2) { a0 = p[0]; a1 = p[3]; b0 = p[1]; b1 = p[4]; k = (s & 1) ? p[2] : p[5]; }
    else if (axis == 1) { a0 = p[0]; a1 = p[3]; b0 = p[2]; b1 = p[5]; k = (s & 1) ? p[1] : p[4]; }
    else { a0 = p[1]; a1 = p[4]; b0 = p[2]; b1 = p[5]; k = (s & 1) ? p[0] : p[3]; }
}

// Nearest t of primitive P in [t_min, t_max], wrappers applied
// (translate.rs:31, rotate_y.rs:39-48).  aux = box side.
template <int PRIMS>
__device__ __forceinline__ bool prim_t(const Prim &P, d3 o, d3 d, double t_min, double t_max,
                                       double &t_out, int &aux) {
    aux = 0;
    if (PRIMS == PRIMS_RECTS) { // untransformed rects only: kind picks the axis
        int kind = P.kind;
        if (kind == RT_PRIM_XY_RECT) return rect_t(2, P.p[0], P.p[1], P.p[2], P.p[3], P.p[4], o, d, t_min, t_max, t_out);
        if (kind == RT_PRIM_XZ_RECT) return rect_t(1, P.p[0], P.p[1], P.p[2], P.p[3], P.p[4], o, d, t_min, t_max, t_out);
        return rect_t(0, P.p[0], P.p[1], P.p[2], P.p[3], P.p[4], o, d, t_min, t_max, t_out);
    }
    if (PRIMS == PRIMS_ANY) {
        if (P.flags & RT_PRIM_HAS_TRANSLATE) o = o - ld3(P.tr);
        if (P.flags & RT_PRIM_HAS_ROTATE_Y) {
            o = rot_fwd(o, P.rot_sin, P.rot_cos);
            d = rot_fwd(d, P.rot_sin, P.rot_cos);
        }
    }
    switch (PRIMS == PRIMS_SPHERES ? (int)RT_PRIM_SPHERE : P.kind) {
    case RT_PRIM_SPHERE: { // sphere.rs:39-59
        d3 oc = o - ld3(P.p);
        double a = len2(d);
        double half_b = dot(oc, d);
        double c = len2(oc) - P.p[3] * P.p[3];
        double disc = half_b * half_b - a * c;
        if (disc < 0.0) return false;
        double sqrtd = sqrt(disc);
        double root = (-half_b - sqrtd) / a;
        if (root < t_min || t_max < root) {
            root = (-half_b + sqrtd) / a;
            if (root < t_min || t_max < root) return false;
        }
        t_out = root;
        return true;
    }
    case RT_PRIM_XY_RECT: return rect_t(2, P.p[0], P.p[1], P.p[2], P.p[3], P.p[4], o, d, t_min, t_max, t_out);
    case RT_PRIM_XZ_RECT: return rect_t(1, P.p[0], P.p[1], P.p[2], P.p[3], P.p[4], o, d, t_min, t_max, t_out);
    case RT_PRIM_YZ_RECT: return rect_t(0, P.p[0], P.p[1], P.p[2], P.p[3], P.p[4], o, d, t_min, t_max, t_out);
    default: { // box.rs:82-101
        bool any = false;
        double closest = t_max;
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            int axis;
            double a0, a1, b0, b1, k, t;
            box_side(P.p, s, axis, a0, a1, b0, b1, k);
            if (rect_t(axis, a0, a1, b0, b1, k, o, d, t_min, closest, t)) {
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

// sphere.rs:20-27; out of line: acos/atan2 are large and only image textures read u,v
__device__ __noinline__ void sphere_uv(d3 outward, double &u, double &v) {
    const double PI = 3.14159265358979323846;
    double theta = acos(-outward.y);
    double phi = atan2(-outward.z, outward.x) + PI;
    u = phi / (2.0 * PI);
    v = theta / PI;
}

// Rebuild the HitRecord of the winning primitive (geometry.rs:17-57).
template <int PRIMS, bool TEXTURED>
__device__ __forceinline__ Hit prim_hit_record(const Prim &P, d3 o, d3 d, double t, int aux, bool want_uv) {
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
    const int kind = PRIMS == PRIMS_SPHERES ? (int)RT_PRIM_SPHERE : P.kind;
    if (PRIMS != PRIMS_RECTS && kind == RT_PRIM_SPHERE) {
        d3 outward = (h.point - ld3(P.p)) * P.inv_radius; // sphere.rs:61
        if (TEXTURED && want_uv) sphere_uv(outward, h.u, h.v);
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
        set_face_normal(h, dd, mk(axis == 0 ? 1.0 : 0.0, axis == 1 ? 1.0 : 0.0, axis == 2 ? 1.0 : 0.0));
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
__device__ __forceinline__ double perlin_noise(const Perlin &pl, d3 p) { // noise.rs:57-96
    double fx = floor(p.x), fy = floor(p.y), fz = floor(p.z);
    double u = p.x - fx, v = p.y - fy, w = p.z - fz;
    int i = (int)fx, j = (int)fy, k = (int)fz; // saturating on AMDGCN like Rust's `as i32`
    double uu = u * u * (3.0 - 2.0 * u);
    double vv = v * v * (3.0 - 2.0 * v);
    double ww = w * w * (3.0 - 2.0 * w);
    double accum = 0.0;
#pragma unroll
    for (int di = 0; di < 2; ++di)
#pragma unroll
        for (int dj = 0; dj < 2; ++dj)
#pragma unroll
            for (int dk = 0; dk < 2; ++dk) {
                int index = pl.perm_x[(uint32_t)(i + di) & 255u] ^ pl.perm_y[(uint32_t)(j + dj) & 255u] ^
                            pl.perm_z[(uint32_t)(k + dk) & 255u];
                const double *g = pl.ranvec[index & 255];
                d3 weight = mk(u - di, v - dj, w - dk);
                accum += (di * uu + (1 - di) * (1.0 - uu)) * (dj * vv + (1 - dj) * (1.0 - vv)) *
                         (dk * ww + (1 - dk) * (1.0 - ww)) * dot(ld3(g), weight);
            }
    return accum;
}

__device__ __forceinline__ double perlin_turbulence(const Perlin &pl, d3 p, int depth) { // noise.rs:98-109
    double accum = 0.0, weight = 1.0;
    for (int o = 0; o < depth; ++o) {
        accum += weight * perlin_noise(pl, p);
        weight *= 0.5;
        p = p * 2.0;
    }
    return fabs(accum);
}

__device__ __forceinline__ double clamp01(double x) { return x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x); }

__device__ __noinline__ d3 texture_value_slow(const TraceArgs &A, int ti, double u, double v, d3 p) {
    Texture T = A.textures[ti];
    if (T.kind == RT_TEX_CHECKERED) { // checkered.rs:32-42
        double sines = sin(p.x * 10.0) * sin(p.y * 10.0) * sin(p.z * 10.0);
        T = A.textures[sines < 0.0 ? T.tex_odd : T.tex_even];
    }
    if (T.kind == RT_TEX_IMAGE) { // texture/image.rs:28-51
        Image img = A.images[T.image];
        double uu = clamp01(u);
        double vv = 1.0 - clamp01(v);
        double i = uu * (double)img.width;
        double j = vv * (double)img.height;
        if (i >= (double)img.width) i = (double)img.width - 1.0;
        if (j >= (double)img.height) j = (double)img.height - 1.0;
        uint32_t xi = (uint32_t)i, yj = (uint32_t)j; // saturating, NaN -> 0
        uchar4 px = reinterpret_cast<const uchar4 *>(img.rgba)[(size_t)yj * (size_t)img.width + xi];
        double s = 1.0 / 255.0;
        return mk((double)px.x * s, (double)px.y * s, (double)px.z * s);
    }
    if (T.kind == RT_TEX_NOISE) { // noise.rs:26-33
        double f = 1.0 + sin(T.scale * p.z + 10.0 * perlin_turbulence(A.perlins[T.perlin], p, T.depth));
        return (ld3(T.color) * 0.5) * f;
    }
    return ld3(T.color);
}

template <bool TEXTURED>
__device__ __forceinline__ d3 texture_value(const TraceArgs &A, const Material &M, double u, double v, d3 p) {
    if (!TEXTURED || M.tex_kind == RT_TEX_SOLID_COLOR) return ld3(M.color); // solid_color.rs:24-28
    return texture_value_slow(A, M.texture, u, v, p);
}

// ------------------------------------------------------------- the kernel
// 256 threads = 4 waves; each wave owns an 8x8 pixel tile, each block a
// 16x16 one.  Rows are "owned rows": with strips (multi-GPU) the launch
// covers only this rank's rows, compacted.
//
// Template flags specialise the kernel to what the scene contains (chosen on
// the host at launch), which keeps the register footprint of the common
// cases small:
//   PRIMS    : PRIMS_RECTS (untransformed rects only), PRIMS_SPHERES
//              (untransformed spheres only), PRIMS_ANY
//   TEXTURED : some material's texture is not a plain SolidColor
//   SPECULAR : some material is Metal or Dielectric
template <int PRIMS, bool TEXTURED, bool SPECULAR>
__global__ __launch_bounds__(256) void k_trace_f64(const TraceArgs A) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int tiles_x = (A.width + 15) >> 4;
    const int bx = blockIdx.x % tiles_x;
    const int by = blockIdx.x / tiles_x;
    const int px = bx * 16 + (wave & 1) * 8 + (lane & 7);
    const int vrow = by * 16 + (wave >> 1) * 8 + (lane >> 3);
    int py = vrow;
    if (A.strip_count > 1) // owned row -> image row
        py = ((vrow / A.strip_rows) * A.strip_count + A.strip_index) * A.strip_rows + vrow % A.strip_rows;
    const bool in_image = px < A.width && vrow < A.owned_rows && py < A.height;

    PathRng rng;
    rng.pixel = (uint32_t)py * (uint32_t)A.width + (uint32_t)px;
    rng.k0 = A.seed_lo;
    rng.k1 = A.seed_hi;

    // cpu.rs:35-36: one horizontal jitter per pixel
    rng.sample = RT_RNG_SAMPLE_PIXEL;
    u4 bj = rng.block(0, RT_RNG_PIXEL, 0);
    const double u = ((double)px + u53(bj.a, bj.b)) / (double)(A.width - 1);

    d3 sum = mk(0.0, 0.0, 0.0);
    d3 o = sum, d = sum, T = sum;
    int s = in_image ? A.sample_begin : A.sample_end;
    uint32_t seg = 0;
    bool alive = false;
    unsigned int n_segments = 0;

    while (s < A.sample_end) {
        if (!alive) {
            // cpu.rs:39-40 + camera.rs:326-337
            rng.sample = (uint32_t)s;
            u4 bc = rng.block(0, RT_RNG_CAMERA, 0);
            double v = ((double)py + u53(bc.a, bc.b)) / (double)(A.height - 1);
            d3 offset = mk(0.0, 0.0, 0.0);
            if (A.cam.lens_radius != 0.0) { // aperture 0: the disk is multiplied by 0 -> skip its draws
                double rx, ry;
                for (uint32_t i = 0;; ++i) { // util.rs:25-39
                    u4 b = rng.block(0, RT_RNG_LENS, i);
                    rx = fma(2.0, u53(b.a, b.b), -1.0);
                    ry = fma(2.0, u53(b.c, b.d), -1.0);
                    if (rx * rx + ry * ry >= 1.0) continue;
                    break;
                }
                rx *= A.cam.lens_radius;
                ry *= A.cam.lens_radius;
                offset = ld3(A.cam.right) * rx + ld3(A.cam.up) * ry;
            }
            d3 co = ld3(A.cam.origin);
            o = co + offset;
            d = ld3(A.cam.ulc) + u * ld3(A.cam.horizontal) - v * ld3(A.cam.vertical) - co - offset;
            // ray time (camera.rs:335) is drawn by the oracle but no primitive in scope reads it
            T = mk(1.0, 1.0, 1.0);
            seg = 0;
            alive = true;
        }

        d3 contrib; // what this path adds to the pixel if it ends here
        bool ended;
        if (A.max_depth <= 0) { // renderer.rs:48-55 with max_depth 0: white, nothing traced
            contrib = T;
            ended = true;
        } else {
            ++n_segments;
            // closest hit, t in [0.001, inf) (renderer.rs:58)
            double best_t = __builtin_inf();
            int best = -1, best_aux = 0;
            for (int i = 0; i < A.n_prims; ++i) {
                double t;
                int aux;
                if (prim_t<PRIMS>(A.prims[i], o, d, 0.001, best_t, t, aux)) {
                    best_t = t;
                    best = i;
                    best_aux = aux;
                }
            }
            if (best < 0) { // background_color.rs:27-33 / :45-48
                d3 bgc = ld3(A.bg.top);
                if (A.bg.kind == RT_BG_SKY) {
                    double t = 0.5 * (d.y / sqrt(len2(d)) + 1.0);
                    bgc = (1.0 - t) * ld3(A.bg.top) + t * ld3(A.bg.bottom);
                }
                contrib = T * bgc;
                ended = true;
            } else {
                const Prim &P = A.prims[best];
                const Material &M = A.materials[P.material];
                Hit h = prim_hit_record<PRIMS, TEXTURED>(P, o, d, best_t, best_aux, M.needs_uv != 0);
                if (M.kind == RT_MAT_DIFFUSE_LIGHT) { // diffuse_light.rs:25-37
                    contrib = T * texture_value<TEXTURED>(A, M, h.u, h.v, h.point);
                    ended = true;
                } else if (M.kind == RT_MAT_LAMBERTIAN) { // lambertian.rs:26-38
                    d3 dir = h.normal + unit(random_in_unit_sphere(rng, seg));
                    if (fabs(dir.x) < 1e-8 && fabs(dir.y) < 1e-8 && fabs(dir.z) < 1e-8) dir = h.normal;
                    T = T * texture_value<TEXTURED>(A, M, h.u, h.v, h.point);
                    o = h.point;
                    d = dir;
                    ended = false;
                } else if (SPECULAR && M.kind == RT_MAT_METAL) { // metal.rs:26-43
                    d3 ud = unit(d);
                    d3 dir = ud - (2.0 * dot(ud, h.normal)) * h.normal;
                    if (M.fuzz != 0.0) dir = dir + M.fuzz * random_in_unit_sphere(rng, seg);
                    if (dot(dir, h.normal) < 0.0) {
                        contrib = mk(0.0, 0.0, 0.0);
                        ended = true;
                    } else {
                        T = T * texture_value<TEXTURED>(A, M, h.u, h.v, h.point);
                        o = h.point;
                        d = dir;
                        ended = false;
                    }
                } else if (SPECULAR) { // dialectric.rs:25-55
                    double ratio = h.front ? 1.0 / M.ior : M.ior;
                    d3 ud = unit(d);
                    double cos_theta = fmin(dot(-ud, h.normal), 1.0);
                    double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
                    bool reflect_it = ratio * sin_theta > 1.0;
                    if (!reflect_it) { // the draw happens only when refraction is possible
                        double r0 = (1.0 - ratio) / (1.0 + ratio);
                        r0 = r0 * r0;
                        double m = 1.0 - cos_theta;
                        double m2 = m * m;
                        double refl = r0 + (1.0 - r0) * (m2 * m2 * m);
                        u4 b = rng.block(seg, RT_RNG_DIELECTRIC, 0);
                        reflect_it = refl > u53(b.a, b.b);
                    }
                    d3 dir;
                    if (reflect_it) {
                        dir = ud - (2.0 * dot(ud, h.normal)) * h.normal;
                    } else { // vec3.rs:416-422
                        d3 perp = ratio * (ud + cos_theta * h.normal);
                        dir = perp + (-sqrt(fabs(1.0 - len2(perp)))) * h.normal;
                    }
                    o = h.point;
                    d = dir;
                    ended = false;
                } else { // unreachable: the host picks SPECULAR whenever such a material exists
                    contrib = mk(0.0, 0.0, 0.0);
                    ended = true;
                }
                // renderer.rs:48-55: the recursion's next level has depth 0 -> white
                if (!ended && (int)++seg >= A.max_depth) {
                    contrib = T;
                    ended = true;
                }
            }
        }
        if (ended) {
            sum = sum + contrib; // vec3.rs:38-42 Color::add, in sample order
            ++s;
            alive = false;
        }
    }

    if (in_image) {
        double *px_out = A.accum + 3 * (size_t)rng.pixel;
        if (A.sample_begin == 0) {
            px_out[0] = sum.x; px_out[1] = sum.y; px_out[2] = sum.z;
        } else {
            px_out[0] += sum.x; px_out[1] += sum.y; px_out[2] += sum.z;
        }
    }
    // one atomic per wave for the segment statistic
    unsigned long long total = n_segments;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) total += __shfl_down(total, off, 64);
    if (lane == 0 && total) atomicAdd(A.segments, total);
}

// vec3.rs:119-125 scale_sqrt over the owned rows: out = sqrt(accum / samples)
__global__ __launch_bounds__(256) void k_resolve_f64(const double *__restrict__ accum, double *__restrict__ out,
                                                     int width, int height, int strip_rows, int strip_count,
                                                     int strip_index, double scale) {
    size_t n = (size_t)width * (size_t)height * 3;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (strip_count > 1) {
            int row = (int)(i / ((size_t)width * 3));
            if ((row / strip_rows) % strip_count != strip_index) continue;
        }
        out[i] = sqrt(scale * accum[i]);
    }
}

} // namespace rtdev

namespace {
template <int PRIMS, bool TEXTURED, bool SPECULAR>
void launch_variant(const rtdev::TraceArgs &a, unsigned blocks, hipStream_t stream) {
    hipLaunchKernelGGL((rtdev::k_trace_f64<PRIMS, TEXTURED, SPECULAR>), dim3(blocks), dim3(256), 0, stream, a);
}
template <int PRIMS>
void launch_prims(const rtdev::TraceArgs &a, bool textured, bool specular, unsigned blocks, hipStream_t stream) {
    if (textured) {
        if (specular) launch_variant<PRIMS, true, true>(a, blocks, stream);
        else launch_variant<PRIMS, true, false>(a, blocks, stream);
    } else {
        if (specular) launch_variant<PRIMS, false, true>(a, blocks, stream);
        else launch_variant<PRIMS, false, false>(a, blocks, stream);
    }
}
} // namespace

// prims_class: rtdev::PRIMS_*; textured/specular: scene feature flags (see k_trace_f64)
extern "C" hipError_t rtdev_launch_trace(const rtdev::TraceArgs *args, int prims_class, int textured,
                                         int specular, hipStream_t stream) {
    int tiles_x = (args->width + 15) / 16;
    int tiles_y = (args->owned_rows + 15) / 16;
    if (tiles_x <= 0 || tiles_y <= 0) return hipSuccess;
    unsigned blocks = (unsigned)(tiles_x * tiles_y);
    switch (prims_class) {
    case rtdev::PRIMS_RECTS: launch_prims<rtdev::PRIMS_RECTS>(*args, textured != 0, specular != 0, blocks, stream); break;
    case rtdev::PRIMS_SPHERES: launch_prims<rtdev::PRIMS_SPHERES>(*args, textured != 0, specular != 0, blocks, stream); break;
    default: launch_prims<rtdev::PRIMS_ANY>(*args, textured != 0, specular != 0, blocks, stream); break;
    }
    return hipGetLastError();
}

extern "C" hipError_t rtdev_launch_resolve(const double *accum, double *out, int width, int height,
                                           int strip_rows, int strip_count, int strip_index, int samples,
                                           hipStream_t stream) {
    size_t n = (size_t)width * (size_t)height * 3;
    unsigned blocks = (unsigned)((n + 255) / 256);
    if (blocks > 2048u) blocks = 2048u;
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(rtdev::k_resolve_f64, dim3(blocks), dim3(256), 0, stream, accum, out, width, height,
                       strip_rows, strip_count, strip_index, 1.0 / (double)samples);
    return hipGetLastError();
}
