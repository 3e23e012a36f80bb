/* trace.c — scene model + recursive ray_color of the reference, in C/f64.
 *
 * TEST INFRASTRUCTURE (see oracle.h).  Each function names the Rust it
 * restates.  Operation ORDER follows the Rust source (e.g. `Vec3 / f64`
 * multiplies by the reciprocal while `unit_vector` divides) and the file is
 * compiled with -ffp-contract=off so that, given the same random draws,
 * this computes what the reference computes in f64.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define PI 3.14159265358979323846264338327950288 /* std::f64::consts::PI */

/* ------------------------------------------------------------------ Vec3 */
typedef struct { double x, y, z; } V3;

static inline V3 v3(double x, double y, double z) { V3 r = { x, y, z }; return r; }
static inline V3 v3p(const double *p) { return v3(p[0], p[1], p[2]); }
static inline void v3out(V3 a, double *o) { o[0] = a.x; o[1] = a.y; o[2] = a.z; }
/* vec3.rs operator impls (component-wise) */
static inline V3 add(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline V3 sub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline V3 mulv(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline V3 scale(V3 a, double s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline V3 neg(V3 a) { return v3(-a.x, -a.y, -a.z); }
/* vec3.rs:279-301: Vec3 / f64 == (1.0 / rhs) * self */
static inline V3 divs(V3 a, double s) { double t = 1.0 / s; return v3(t * a.x, t * a.y, t * a.z); }
/* vec3.rs:165-167 */
static inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* vec3.rs:169-175 */
static inline V3 cross(V3 a, V3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* vec3.rs:91-93 */
static inline double length_squared(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
/* vec3.rs:87-89 */
static inline double length(V3 a) { return sqrt(length_squared(a)); }
/* vec3.rs:79-85: true division by the length */
static inline V3 unit_vector(V3 a) { double l = length(a); return v3(a.x / l, a.y / l, a.z / l); }
/* vec3.rs:127-130 */
static inline int near_zero(V3 a) {
    const double s = 1e-8;
    return fabs(a.x) < s && fabs(a.y) < s && fabs(a.z) < s;
}
/* vec3.rs:412-414: v1 - 2.0 * v1.dot(v2) * v2 */
static inline V3 reflect(V3 v, V3 n) { return sub(v, scale(n, 2.0 * dot(v, n))); }
/* vec3.rs:416-422 */
static inline V3 refract(V3 uv, V3 n, double etai_over_etat) {
    double cos_theta = fmin(dot(neg(uv), n), 1.0);
    V3 r_out_perp = scale(add(uv, scale(n, cos_theta)), etai_over_etat);
    V3 r_out_parallel = scale(n, -sqrt(fabs(1.0 - length_squared(r_out_perp))));
    return add(r_out_perp, r_out_parallel);
}
/* ray.rs:30-32 */
static inline V3 ray_at(V3 o, V3 d, double t) { return add(o, scale(d, t)); }

void orc_vec3_add(const double a[3], const double b[3], double out[3]) { v3out(add(v3p(a), v3p(b)), out); }
void orc_vec3_sub(const double a[3], const double b[3], double out[3]) { v3out(sub(v3p(a), v3p(b)), out); }
void orc_vec3_mul(const double a[3], const double b[3], double out[3]) { v3out(mulv(v3p(a), v3p(b)), out); }
void orc_vec3_scale(const double a[3], double s, double out[3]) { v3out(scale(v3p(a), s), out); }
void orc_vec3_div(const double a[3], double s, double out[3]) { v3out(divs(v3p(a), s), out); }
void orc_reflect(const double v[3], const double n[3], double out[3]) { v3out(reflect(v3p(v), v3p(n)), out); }
void orc_refract(const double uv[3], const double n[3], double ratio, double out[3]) {
    v3out(refract(v3p(uv), v3p(n), ratio), out);
}

/* ------------------------------------------------------------------- RNG
 * Path-addressed draws (include/rt_rng.h). */
typedef struct {
    uint64_t seed;
    uint32_t pixel;
    uint32_t sample;
} PathRng;

static inline double draw(const PathRng *r, uint32_t segment, uint32_t purpose, uint32_t block, int which) {
    return orc_rng_double(r->seed, r->pixel, r->sample, segment, purpose, block, which);
}
/* util.rs:14-17 with the contract's range map */
static inline double range(double a, double b, double d) { return a + (b - a) * d; }

/* util.rs:25-39 random_in_unit_disk */
static V3 random_in_unit_disk(const PathRng *r) {
    for (uint32_t i = 0;; ++i) {
        V3 p = v3(range(-1.0, 1.0, draw(r, 0, RT_RNG_LENS, i, 0)),
                  range(-1.0, 1.0, draw(r, 0, RT_RNG_LENS, i, 1)), 0.0);
        if (length_squared(p) >= 1.0) continue;
        return p;
    }
}
/* vec3.rs:424-430 random_in_unit_sphere */
static V3 random_in_unit_sphere(const PathRng *r, uint32_t segment) {
    for (uint32_t i = 0;; ++i) {
        double e[3]; /* Vec3::random_range(-1, 1): the candidate's three draws share block i */
        orc_rng_triple(r->seed, r->pixel, r->sample, segment, RT_RNG_SCATTER, i, e);
        V3 p = v3(range(-1.0, 1.0, e[0]), range(-1.0, 1.0, e[1]), range(-1.0, 1.0, e[2]));
        if (length_squared(p) >= 1.0) continue;
        return p;
    }
}
/* vec3.rs:442-444 */
static V3 random_unit_vector(const PathRng *r, uint32_t segment) {
    return unit_vector(random_in_unit_sphere(r, segment));
}

/* ------------------------------------------------------------------ AABB */
/* aabb.rs:42-59 (`hit`, the variant the BVH uses) */
int orc_aabb_hit(const double bmin[3], const double bmax[3], const double origin[3],
                 const double dir[3], double t_min, double t_max) {
    for (int a = 0; a < 3; ++a) {
        double inv_d = 1.0 / dir[a];
        double t0 = (bmin[a] - origin[a]) * inv_d;
        double t1 = (bmax[a] - origin[a]) * inv_d;
        if (inv_d < 0.0) { double tmp = t0; t0 = t1; t1 = tmp; }
        double lo = t0 > t_min ? t0 : t_min;
        double hi = t1 < t_max ? t1 : t_max;
        if (hi <= lo) return 0;
    }
    return 1;
}

/* aabb.rs:10-26 */
static void aabb_new(V3 a, V3 b, double mn[3], double mx[3]) {
    mn[0] = fmin(a.x, b.x); mn[1] = fmin(a.y, b.y); mn[2] = fmin(a.z, b.z);
    mx[0] = fmax(a.x, b.x); mx[1] = fmax(a.y, b.y); mx[2] = fmax(a.z, b.z);
}

/* ------------------------------------------------------------- primitives */
/* geometry.rs:49-56 */
static void set_face_normal(OrcHit *h, V3 dir, V3 outward) {
    h->front_face = dot(dir, outward) < 0.0;
    V3 n = h->front_face ? outward : neg(outward);
    v3out(n, h->normal);
}

/* sphere.rs:20-27 */
void orc_sphere_uv(const double n[3], double *u, double *v) {
    double theta = acos(-n[1]);
    double phi = atan2(-n[2], n[0]) + PI;
    *u = phi / (2.0 * PI);
    *v = theta / PI;
}

/* sphere.rs:31-68 */
static int hit_sphere(const double *p, V3 o, V3 d, double t_min, double t_max, OrcHit *h) {
    V3 center = v3(p[0], p[1], p[2]);
    double radius = p[3];
    V3 oc = sub(o, center);
    double a = length_squared(d);
    double half_b = dot(oc, d);
    double c = length_squared(oc) - radius * radius;
    double discriminant = half_b * half_b - a * c;
    if (discriminant < 0.0) return 0;
    double sqrtd = sqrt(discriminant);
    double root = (-half_b - sqrtd) / a;
    if (root < t_min || t_max < root) {
        root = (-half_b + sqrtd) / a;
        if (root < t_min || t_max < root) return 0;
    }
    V3 point = ray_at(o, d, root);
    V3 outward = divs(sub(point, center), radius);
    double on[3];
    v3out(outward, on);
    orc_sphere_uv(on, &h->u, &h->v);
    v3out(point, h->point);
    h->t = root;
    set_face_normal(h, d, outward);
    return 1;
}

/* xy_rect.rs:21-48, xz_rect.rs:21-49, yz_rect.rs:21-49.  `axis` is the
 * constant axis (2, 1, 0); a/b are the in-plane axes in declaration order. */
static int hit_rect(int axis, const double *p, V3 o, V3 d, double t_min, double t_max, OrcHit *h) {
    double oo[3] = { o.x, o.y, o.z }, dd[3] = { d.x, d.y, d.z };
    int ia, ib;
    if (axis == 2) { ia = 0; ib = 1; } else if (axis == 1) { ia = 0; ib = 2; } else { ia = 1; ib = 2; }
    double a0 = p[0], a1 = p[1], b0 = p[2], b1 = p[3], k = p[4];
    double t = (k - oo[axis]) / dd[axis];
    if (t < t_min || t > t_max) return 0;
    double a = oo[ia] + t * dd[ia];
    double b = oo[ib] + t * dd[ib];
    if (a < a0 || a > a1 || b < b0 || b > b1) return 0;
    h->u = (a - a0) / (a1 - a0);
    h->v = (b - b0) / (b1 - b0);
    v3out(ray_at(o, d, t), h->point);
    h->t = t;
    V3 outward = v3(axis == 0 ? 1.0 : 0.0, axis == 1 ? 1.0 : 0.0, axis == 2 ? 1.0 : 0.0);
    set_face_normal(h, d, outward);
    return 1;
}

/* box.rs:22-71 (side order) and :82-101 (closest-hit loop) */
static int hit_box(const double *p, V3 o, V3 d, double t_min, double t_max, OrcHit *h) {
    double mnx = p[0], mny = p[1], mnz = p[2], mxx = p[3], mxy = p[4], mxz = p[5];
    const double sides[6][5] = {
        { mnx, mxx, mny, mxy, mxz }, /* xy @ max.z */
        { mnx, mxx, mny, mxy, mnz }, /* xy @ min.z */
        { mnx, mxx, mnz, mxz, mxy }, /* xz @ max.y */
        { mnx, mxx, mnz, mxz, mny }, /* xz @ min.y */
        { mny, mxy, mnz, mxz, mxx }, /* yz @ max.x */
        { mny, mxy, mnz, mxz, mnx }, /* yz @ min.x */
    };
    static const int axes[6] = { 2, 2, 1, 1, 0, 0 };
    int any = 0;
    double closest = t_max;
    for (int s = 0; s < 6; ++s) {
        OrcHit tmp;
        if (hit_rect(axes[s], sides[s], o, d, t_min, closest, &tmp)) {
            closest = tmp.t;
            *h = tmp;
            any = 1;
        }
    }
    return any;
}

/* moving_sphere.rs:41-87: the centre moves linearly with the ray's time; uv
 * comes from the hit POINT, not the normal (moving_sphere.rs:76, SURVEY B-19) */
static int hit_moving_sphere(const RtPrimitive *prim, V3 o, V3 d, double time, double t_min, double t_max, OrcHit *h) {
    V3 pos_a = v3(prim->p[0], prim->p[1], prim->p[2]);
    double radius = prim->p[3];
    /* moving_sphere.rs:37-39 */
    V3 center = add(pos_a, scale(sub(v3p(prim->center_b), pos_a), (time - prim->time_a) / (prim->time_b - prim->time_a)));
    V3 oc = sub(o, center);
    double a = length_squared(d);
    double half_b = dot(oc, d);
    double c = length_squared(oc) - radius * radius;
    double discriminant = half_b * half_b - a * c;
    if (discriminant < 0.0) return 0;
    double sqrtd = sqrt(discriminant);
    double root = (-half_b - sqrtd) / a;
    if (root < t_min || t_max < root) {
        root = (-half_b + sqrtd) / a;
        if (root < t_min || t_max < root) return 0;
    }
    V3 point = ray_at(o, d, root);
    V3 outward = divs(sub(point, center), radius);
    double pt[3];
    v3out(point, pt);
    orc_sphere_uv(pt, &h->u, &h->v);
    v3out(point, h->point);
    h->t = root;
    set_face_normal(h, d, outward);
    return 1;
}

static int hit_bare(const RtPrimitive *prim, V3 o, V3 d, double time, double t_min, double t_max, OrcHit *h) {
    switch (prim->kind) {
    case RT_PRIM_MOVING_SPHERE: return hit_moving_sphere(prim, o, d, time, t_min, t_max, h);
    case RT_PRIM_SPHERE: return hit_sphere(prim->p, o, d, t_min, t_max, h);
    case RT_PRIM_XY_RECT: return hit_rect(2, prim->p, o, d, t_min, t_max, h);
    case RT_PRIM_XZ_RECT: return hit_rect(1, prim->p, o, d, t_min, t_max, h);
    case RT_PRIM_YZ_RECT: return hit_rect(0, prim->p, o, d, t_min, t_max, h);
    case RT_PRIM_BOX: return hit_box(prim->p, o, d, t_min, t_max, h);
    default: return 0;
    }
}

/* rotate_y.rs:31-64 around the bare primitive */
static int hit_rotated(const RtPrimitive *prim, V3 o, V3 d, double time, double t_min, double t_max, OrcHit *h) {
    if (!(prim->flags & RT_PRIM_HAS_ROTATE_Y)) return hit_bare(prim, o, d, time, t_min, t_max, h);
    double s = prim->rot_sin, c = prim->rot_cos;
    V3 ro = v3(c * o.x - s * o.z, o.y, s * o.x + c * o.z);
    V3 rd = v3(c * d.x - s * d.z, d.y, s * d.x + c * d.z);
    if (!hit_bare(prim, ro, rd, time, t_min, t_max, h)) return 0;
    V3 pt = v3p(h->point), n = v3p(h->normal);
    V3 wp = v3(c * pt.x + s * pt.z, pt.y, -s * pt.x + c * pt.z);
    V3 wn = v3(c * n.x + s * n.z, n.y, -s * n.x + c * n.z);
    v3out(wp, h->point);
    /* rotate_y.rs:62: face test against the ROTATED (object-space) ray */
    set_face_normal(h, rd, wn);
    return 1;
}

/* translate.rs:24-41 around (rotated) primitive */
int orc_hit_primitive(const RtPrimitive *prim, const double origin[3], const double dir[3],
                      double t_min, double t_max, OrcHit *out) {
    return orc_hit_primitive_time(prim, origin, dir, 0.0, t_min, t_max, out);
}

int orc_hit_primitive_time(const RtPrimitive *prim, const double origin[3], const double dir[3], double time,
                           double t_min, double t_max, OrcHit *out) {
    V3 o = v3p(origin), d = v3p(dir);
    int hit;
    if (prim->flags & RT_PRIM_HAS_TRANSLATE) {
        V3 off = v3p(prim->translate);
        V3 mo = sub(o, off);
        hit = hit_rotated(prim, mo, d, time, t_min, t_max, out);
        if (hit) {
            v3out(add(v3p(out->point), off), out->point);
            /* translate.rs:36: set_face_normal with the already-flipped normal */
            set_face_normal(out, d, v3p(out->normal));
        }
    } else {
        hit = hit_rotated(prim, o, d, time, t_min, t_max, out);
    }
    if (hit) {
        out->material = prim->material;
        out->obj_id = prim->obj_id;
    }
    return hit;
}

/* geometry_creation.rs + each create_bounding_box */
void orc_primitive_aabb(const RtPrimitive *prim, double mn[3], double mx[3]) {
    const double *p = prim->p;
    V3 pos; /* SceneObject.pos of the bare primitive */
    switch (prim->kind) {
    case RT_PRIM_SPHERE: /* sphere.rs:72-77 */
        pos = v3(p[0], p[1], p[2]);
        aabb_new(sub(pos, v3(p[3], p[3], p[3])), add(pos, v3(p[3], p[3], p[3])), mn, mx);
        break;
    case RT_PRIM_XY_RECT: /* xy_rect.rs:50-55 */
        pos = v3(p[0], p[2], p[4]);
        aabb_new(v3(p[0], p[2], p[4] - 0.0001), v3(p[1], p[3], p[4] + 0.0001), mn, mx);
        break;
    case RT_PRIM_XZ_RECT:
        pos = v3(p[0], p[4], p[2]);
        aabb_new(v3(p[0], p[4] - 0.0001, p[2]), v3(p[1], p[4] + 0.0001, p[3]), mn, mx);
        break;
    case RT_PRIM_YZ_RECT:
        pos = v3(p[4], p[0], p[2]);
        aabb_new(v3(p[4] - 0.0001, p[0], p[2]), v3(p[4] + 0.0001, p[1], p[3]), mn, mx);
        break;
    case RT_PRIM_MOVING_SPHERE: { /* moving_sphere.rs:93-106: union of the boxes at pos_a and pos_b */
        double mn2[3], mx2[3];
        V3 r3 = v3(p[3], p[3], p[3]), pb = v3p(prim->center_b);
        pos = v3(p[0], p[1], p[2]);
        aabb_new(sub(pos, r3), add(pos, r3), mn, mx);
        aabb_new(sub(pb, r3), add(pb, r3), mn2, mx2);
        for (int a = 0; a < 3; ++a) {
            mn[a] = fmin(mn[a], mn2[a]);
            mx[a] = fmax(mx[a], mx2[a]);
        }
        break;
    }
    default: /* box.rs:103-110, geometry_creation.rs:95-103 */
        pos = v3(p[0], p[1], p[2]);
        aabb_new(v3(p[0], p[1], p[2]), v3(p[3], p[4], p[5]), mn, mx);
        break;
    }
    if (prim->flags & RT_PRIM_HAS_ROTATE_Y) {
        /* rotate_y.rs:66-90, arithmetic slips included (`+ z`, and min/max
         * shifted by pos inside the corner loop). */
        double s = prim->rot_sin, c = prim->rot_cos;
        V3 lo = v3(DBL_MAX, DBL_MAX, DBL_MAX), hi = v3(-DBL_MAX, -DBL_MAX, -DBL_MAX);
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                for (int k = 0; k < 2; ++k) {
                    double x = (double)i * mx[0] + (double)(1 - i) * mn[0];
                    double y = (double)j * mx[1] + (double)(1 - j) * mn[1];
                    double z = (double)k * mx[2] + (double)(1 - k) * mn[2];
                    double new_x = c * x + s + z;
                    double new_z = -s * x + c * z;
                    lo = v3(fmin(lo.x, new_x), fmin(lo.y, y), fmin(lo.z, new_z));
                    hi = v3(fmax(hi.x, new_x), fmax(hi.y, y), fmax(hi.z, new_z));
                    lo = add(lo, pos);
                    hi = add(hi, pos);
                }
        aabb_new(lo, hi, mn, mx);
    }
    if (prim->flags & RT_PRIM_HAS_TRANSLATE) { /* translate.rs:43-46 */
        V3 off = v3p(prim->translate);
        aabb_new(add(v3p(mn), off), add(v3p(mx), off), mn, mx);
    }
}

/* -------------------------------------------------------------------- BVH */
typedef struct BvhNode {
    int leaf_prim; /* >= 0: leaf */
    int left, right;
    double mn[3], mx[3];
} BvhNode;

struct OrcScene {
    const RtSceneDesc *desc;
    int use_bvh;
    BvhNode *nodes;
    int n_nodes;
    int root;
    uint64_t seed;
    uint32_t build_counter;
};

/* bvh_node.rs:31-82.  `idx` holds primitive indices of this subtree. */
static int bvh_build(OrcScene *s, int *idx, int n, double (*pmn)[3], double (*pmx)[3]) {
    /* random axis in {0,1}: bvh_node.rs:32 (exclusive upper bound 2) */
    int axis = orc_rng_double(s->seed, s->build_counter++, RT_RNG_SAMPLE_TABLE, 0, RT_RNG_BVH, 0, 0) < 0.5 ? 0 : 1;
    int me = s->n_nodes++;
    BvhNode *node = &s->nodes[me];
    if (n == 1) {
        node->leaf_prim = idx[0];
        node->left = node->right = -1;
        memcpy(node->mn, pmn[idx[0]], sizeof node->mn);
        memcpy(node->mx, pmx[idx[0]], sizeof node->mx);
        return me;
    }
    /* sort by bbox min on the axis (for n == 2 the reference orders the pair
     * with the same comparator, bvh_node.rs:46-50) */
    for (int i = 1; i < n; ++i) { /* stable insertion sort, "a.min < b.min" */
        int v = idx[i], j = i;
        while (j > 0 && pmn[v][axis] < pmn[idx[j - 1]][axis]) { idx[j] = idx[j - 1]; --j; }
        idx[j] = v;
    }
    int mid = n / 2;
    int l, r;
    if (n == 2) {
        /* bvh_node.rs:46-50: (0,1) if cmp(o0,o1) else (1,0); after the stable
         * sort above idx[0] is the smaller or, on a tie, the original first,
         * whereas the reference puts the original SECOND first on a tie.
         * Ties only change topology (SURVEY B-15). */
        l = bvh_build(s, idx, 1, pmn, pmx);
        r = bvh_build(s, idx + 1, 1, pmn, pmx);
    } else {
        l = bvh_build(s, idx, mid, pmn, pmx);
        r = bvh_build(s, idx + mid, n - mid, pmn, pmx);
    }
    node = &s->nodes[me]; /* (no realloc, but keep the habit) */
    node->leaf_prim = -1;
    node->left = l;
    node->right = r;
    for (int a = 0; a < 3; ++a) { /* aabb.rs:95-114 */
        node->mn[a] = fmin(s->nodes[l].mn[a], s->nodes[r].mn[a]);
        node->mx[a] = fmax(s->nodes[l].mx[a], s->nodes[r].mx[a]);
    }
    return me;
}

OrcScene *orc_scene_build(const RtSceneDesc *desc, int use_bvh, uint64_t seed) {
    OrcScene *s = (OrcScene *)calloc(1, sizeof *s);
    s->desc = desc;
    s->use_bvh = use_bvh && desc->n_primitives > 0;
    s->seed = seed;
    if (s->use_bvh) {
        int n = desc->n_primitives;
        s->nodes = (BvhNode *)calloc((size_t)(2 * n), sizeof(BvhNode));
        double (*pmn)[3] = malloc(sizeof(double[3]) * (size_t)n);
        double (*pmx)[3] = malloc(sizeof(double[3]) * (size_t)n);
        int *idx = malloc(sizeof(int) * (size_t)n);
        for (int i = 0; i < n; ++i) {
            idx[i] = i;
            orc_primitive_aabb(&desc->primitives[i], pmn[i], pmx[i]);
        }
        s->root = bvh_build(s, idx, n, pmn, pmx);
        free(pmn); free(pmx); free(idx);
    }
    return s;
}

void orc_scene_free(OrcScene *s) {
    if (!s) return;
    free(s->nodes);
    free(s);
}

/* bvh_node.rs:112-132 */
static int bvh_hit(const OrcScene *s, int ni, const double o[3], const double d[3], double time,
                   double t_min, double t_max, OrcHit *out) {
    const BvhNode *node = &s->nodes[ni];
    if (!orc_aabb_hit(node->mn, node->mx, o, d, t_min, t_max)) return 0;
    if (node->leaf_prim >= 0)
        return orc_hit_primitive_time(&s->desc->primitives[node->leaf_prim], o, d, time, t_min, t_max, out);
    OrcHit l;
    if (bvh_hit(s, node->left, o, d, time, t_min, t_max, &l)) {
        OrcHit r;
        if (bvh_hit(s, node->right, o, d, time, t_min, l.t, &r)) *out = r; else *out = l;
        return 1;
    }
    return bvh_hit(s, node->right, o, d, time, t_min, t_max, out);
}

int orc_scene_hit(const OrcScene *s, const double origin[3], const double dir[3],
                  double t_min, double t_max, OrcHit *out) {
    return orc_scene_hit_time(s, origin, dir, 0.0, t_min, t_max, out);
}

int orc_scene_hit_time(const OrcScene *s, const double origin[3], const double dir[3], double time,
                       double t_min, double t_max, OrcHit *out) {
    if (s->desc->n_primitives <= 0) return 0; /* SURVEY B-18: reference would hang */
    if (s->use_bvh) return bvh_hit(s, s->root, origin, dir, time, t_min, t_max, out);
    /* linear scan with shrinking t_max (shared_scene.rs) */
    int any = 0;
    double closest = t_max;
    for (int i = 0; i < s->desc->n_primitives; ++i) {
        OrcHit tmp;
        if (orc_hit_primitive_time(&s->desc->primitives[i], origin, dir, time, t_min, closest, &tmp)) {
            closest = tmp.t;
            *out = tmp;
            any = 1;
        }
    }
    return any;
}

/* --------------------------------------------------------------- textures */
/* noise.rs:79-96 */
static double perlin_interp(V3 c[2][2][2], double u, double v, double w) {
    double uu = u * u * (3.0 - 2.0 * u);
    double vv = v * v * (3.0 - 2.0 * v);
    double ww = w * w * (3.0 - 2.0 * w);
    double accum = 0.0;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int k = 0; k < 2; ++k) {
                V3 weight = v3(u - (double)i, v - (double)j, w - (double)k);
                accum += ((double)i * uu + (1.0 - (double)i) * (1.0 - uu))
                       * ((double)j * vv + (1.0 - (double)j) * (1.0 - vv))
                       * ((double)k * ww + (1.0 - (double)k) * (1.0 - ww))
                       * dot(c[i][j][k], weight);
            }
    return accum;
}

/* Rust `f64 as i32`: saturating, NaN -> 0 */
static int32_t f64_as_i32(double x) {
    if (x != x) return 0;
    if (x >= 2147483647.0) return INT32_MAX;
    if (x <= -2147483648.0) return INT32_MIN;
    return (int32_t)x;
}

/* noise.rs:57-77 */
double orc_perlin_noise(const RtPerlin *pl, const double p[3]) {
    double fx = floor(p[0]), fy = floor(p[1]), fz = floor(p[2]);
    double u = p[0] - fx, v = p[1] - fy, w = p[2] - fz;
    int32_t i = f64_as_i32(fx), j = f64_as_i32(fy), k = f64_as_i32(fz);
    V3 c[2][2][2];
    for (int di = 0; di < 2; ++di)
        for (int dj = 0; dj < 2; ++dj)
            for (int dk = 0; dk < 2; ++dk) {
                /* i + di wraps like Rust release arithmetic; & 255 keeps it in range */
                int32_t index = pl->perm_x[(uint32_t)(i + di) & 255u]
                              ^ pl->perm_y[(uint32_t)(j + dj) & 255u]
                              ^ pl->perm_z[(uint32_t)(k + dk) & 255u];
                c[di][dj][dk] = v3p(pl->ranvec[index & 255]);
            }
    return perlin_interp(c, u, v, w);
}

/* noise.rs:98-109 */
double orc_perlin_turbulence(const RtPerlin *pl, const double p[3], int depth) {
    double accum = 0.0, weight = 1.0;
    double tp[3] = { p[0], p[1], p[2] };
    for (int o = 0; o < depth; ++o) {
        accum += weight * orc_perlin_noise(pl, tp);
        weight *= 0.5;
        tp[0] *= 2.0; tp[1] *= 2.0; tp[2] *= 2.0;
    }
    return fabs(accum);
}

static double clamp01(double x) { /* f64::clamp(0.0, 1.0); NaN stays NaN */
    if (x < 0.0) return 0.0;
    if (x > 1.0) return 1.0;
    return x;
}

void orc_texture_value(const RtSceneDesc *desc, int32_t ti, double u, double v,
                       const double p[3], double out[3]) {
    const RtTexture *t = &desc->textures[ti];
    switch (t->kind) {
    case RT_TEX_SOLID_COLOR: /* solid_color.rs:24-28 */
        out[0] = t->color[0]; out[1] = t->color[1]; out[2] = t->color[2];
        return;
    case RT_TEX_CHECKERED: { /* checkered.rs:32-42, checker_size = 10 */
        double sines = sin(p[0] * 10.0) * sin(p[1] * 10.0) * sin(p[2] * 10.0);
        orc_texture_value(desc, sines < 0.0 ? t->tex_odd : t->tex_even, u, v, p, out);
        return;
    }
    case RT_TEX_IMAGE: { /* texture/image.rs:28-51 */
        const RtImage *img = &desc->images[t->image];
        double uu = clamp01(u);
        double vv = 1.0 - clamp01(v);
        double i = uu * (double)img->width;
        double j = vv * (double)img->height;
        if (i >= (double)img->width) i = (double)img->width - 1.0;
        if (j >= (double)img->height) j = (double)img->height - 1.0;
        uint32_t xi = (i != i || i <= 0.0) ? 0u : (uint32_t)i; /* `as u32` */
        uint32_t yj = (j != j || j <= 0.0) ? 0u : (uint32_t)j;
        const uint8_t *px = img->rgba + 4 * ((size_t)yj * (size_t)img->width + xi);
        double color_scale = 1.0 / 255.0;
        out[0] = (double)px[0] * color_scale;
        out[1] = (double)px[1] * color_scale;
        out[2] = (double)px[2] * color_scale;
        return;
    }
    default: { /* noise.rs:26-33 */
        const RtPerlin *pl = &desc->perlins[t->perlin];
        double f = 1.0 + sin(t->scale * p[2] + 10.0 * orc_perlin_turbulence(pl, p, t->depth));
        V3 c = scale(scale(v3p(t->color), 0.5), f);
        v3out(c, out);
        return;
    }
    }
}

/* background_color.rs:27-33, :45-48 */
void orc_background_color(const RtBackground *bg, const double dir[3], double out[3]) {
    if (bg->kind == RT_BG_SOLID) {
        out[0] = bg->top[0]; out[1] = bg->top[1]; out[2] = bg->top[2];
        return;
    }
    V3 unit = unit_vector(v3p(dir));
    double t = 0.5 * (unit.y + 1.0);
    V3 c = add(scale(v3p(bg->top), 1.0 - t), scale(v3p(bg->bottom), t));
    v3out(c, out);
}

/* -------------------------------------------------------------- materials */
/* dialectric.rs:17-22 */
double orc_schlick(double cosine, double refraction_index) {
    double r0 = (1.0 - refraction_index) / (1.0 + refraction_index);
    r0 = r0 * r0;
    return r0 + (1.0 - r0) * pow(1.0 - cosine, 5.0);
}

/* Material::scatter; returns 1 and fills (dir, attenuation) when the ray
 * continues.  The new ray's origin is rec->point and it inherits `time`. */
static int scatter(const RtSceneDesc *desc, const RtMaterial *m, V3 ray_dir, const OrcHit *rec,
                   const PathRng *rng, uint32_t segment, V3 *out_dir, V3 *attenuation) {
    V3 normal = v3p(rec->normal);
    double att[3];
    switch (m->kind) {
    case RT_MAT_LAMBERTIAN: { /* lambertian.rs:26-38 */
        V3 dir = add(normal, random_unit_vector(rng, segment));
        if (near_zero(dir)) dir = normal;
        *out_dir = dir;
        orc_texture_value(desc, m->texture, rec->u, rec->v, rec->point, att);
        *attenuation = v3p(att);
        return 1;
    }
    case RT_MAT_METAL: { /* metal.rs:26-43 */
        V3 reflected = reflect(unit_vector(ray_dir), normal);
        V3 dir = add(reflected, scale(random_in_unit_sphere(rng, segment), m->fuzz));
        if (dot(dir, normal) < 0.0) return 0;
        *out_dir = dir;
        orc_texture_value(desc, m->texture, rec->u, rec->v, rec->point, att);
        *attenuation = v3p(att);
        return 1;
    }
    case RT_MAT_DIELECTRIC: { /* dialectric.rs:25-55 */
        double ratio = rec->front_face ? 1.0 / m->refraction_index : m->refraction_index;
        V3 unit = unit_vector(ray_dir);
        double cos_theta = fmin(dot(neg(unit), normal), 1.0);
        double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
        int cannot_refract = ratio * sin_theta > 1.0;
        /* `||` short-circuits: the draw happens only when refraction is possible */
        if (cannot_refract || orc_schlick(cos_theta, ratio) > draw(rng, segment, RT_RNG_DIELECTRIC, 0, 0))
            *out_dir = reflect(unit, normal);
        else
            *out_dir = refract(unit, normal, ratio);
        *attenuation = v3(1.0, 1.0, 1.0);
        return 1;
    }
    default: /* diffuse_light.rs:25-31 */
        return 0;
    }
}

/* material.rs:12-14 default, diffuse_light.rs:33-36 */
static V3 color_emitted(const RtSceneDesc *desc, const RtMaterial *m, const OrcHit *rec) {
    if (m->kind != RT_MAT_DIFFUSE_LIGHT) return v3(0.0, 0.0, 0.0);
    double e[3];
    orc_texture_value(desc, m->texture, rec->u, rec->v, rec->point, e);
    return v3p(e);
}

/* ---------------------------------------------------------------- ray_color
 * renderer.rs:41-90, recursive like the original; only `.rgb` is kept
 * (cpu.rs:41-49 discards the rest).  `depth` counts down from max_depth;
 * segment index for the RNG = max_depth - depth. */
typedef struct {
    const RtSceneDesc *desc;
    const OrcScene *scene;
    int max_depth;
    uint64_t segments;
} TraceCtx;

static V3 ray_color(TraceCtx *ctx, V3 o, V3 d, double time, int depth, const PathRng *rng) {
    if (depth == 0) return v3(1.0, 1.0, 1.0); /* renderer.rs:48-55 */
    double oo[3], dd[3];
    v3out(o, oo); v3out(d, dd);
    OrcHit rec;
    ctx->segments++;
    if (!orc_scene_hit_time(ctx->scene, oo, dd, time, 0.001, INFINITY, &rec)) {
        double bg[3];
        orc_background_color(&ctx->desc->background, dd, bg);
        return v3p(bg);
    }
    const RtMaterial *m = &ctx->desc->materials[rec.material];
    V3 emitted = color_emitted(ctx->desc, m, &rec);
    V3 sdir, att;
    if (!scatter(ctx->desc, m, d, &rec, rng, (uint32_t)(ctx->max_depth - depth), &sdir, &att))
        return emitted;
    /* every Material::scatter builds its ray with ray.time() (e.g. lambertian.rs:35) */
    V3 deeper = ray_color(ctx, v3p(rec.point), sdir, time, depth - 1, rng);
    return add(emitted, mulv(att, deeper));
}

/* camera.rs:326-337 */
static void get_ray(const RtCamera *cam, double u, double v, const PathRng *rng, V3 *o, V3 *d,
                    double *time) {
    V3 rd = scale(random_in_unit_disk(rng), cam->lens_radius);
    V3 offset = add(scale(v3p(cam->right), rd.x), scale(v3p(cam->up), rd.y));
    V3 origin = v3p(cam->origin);
    *o = add(origin, offset);
    *d = sub(sub(sub(add(v3p(cam->upper_left_corner), scale(v3p(cam->horizontal), u)),
                     scale(v3p(cam->vertical), v)),
                 origin),
             offset);
    *time = range(cam->time_a, cam->time_b, draw(rng, 0, RT_RNG_CAMERA, 0, 1));
}

/* cpu.rs:35-36: the horizontal jitter a pixel shares between its samples */
double orc_pixel_u(const RtRenderParams *params, int px, int py) {
    uint32_t pixel = (uint32_t)py * (uint32_t)params->width + (uint32_t)px;
    PathRng prng = { params->seed, pixel, RT_RNG_SAMPLE_PIXEL };
    return ((double)px + draw(&prng, 0, RT_RNG_PIXEL, 0, 0)) / (double)(params->width - 1);
}

/* One pass of the sample loop body, cpu.rs:39-49 */
void orc_sample_radiance_u(const RtSceneDesc *desc, const OrcScene *scene, const RtCamera *camera,
                           const RtRenderParams *params, int px, int py, int sample, double u,
                           double out[3], int *n_segments) {
    uint32_t pixel = (uint32_t)py * (uint32_t)params->width + (uint32_t)px;
    PathRng rng = { params->seed, pixel, (uint32_t)sample };
    double v = ((double)py + draw(&rng, 0, RT_RNG_CAMERA, 0, 0)) / (double)(params->height - 1);
    V3 o, d;
    double time;
    get_ray(camera, u, v, &rng, &o, &d, &time);
    TraceCtx ctx = { desc, scene, params->max_depth, 0 };
    V3 c = ray_color(&ctx, o, d, time, params->max_depth, &rng);
    v3out(c, out);
    if (n_segments) *n_segments = (int)ctx.segments;
}

void orc_sample_radiance(const RtSceneDesc *desc, const OrcScene *scene, const RtCamera *camera,
                         const RtRenderParams *params, int px, int py, int sample,
                         double out[3], int *n_segments) {
    orc_sample_radiance_u(desc, scene, camera, params, px, py, sample,
                          orc_pixel_u(params, px, py), out, n_segments);
}

/* camera.rs:196-234 */
void orc_camera_new(const double look_from[3], const double look_at[3],
                    const double scene_up[3], double vfov, double aperture,
                    double focus_distance, double aspect_ratio, double time_a,
                    double time_b, RtCamera *out) {
    double h = tan((vfov * PI / 180.0) / 2.0); /* util.rs:5-7 */
    double viewport_height = 2.0 * h;
    double viewport_width = aspect_ratio * viewport_height;
    V3 from = v3p(look_from);
    V3 forward = unit_vector(sub(from, v3p(look_at)));
    V3 right = unit_vector(cross(v3p(scene_up), forward));
    V3 up = cross(forward, right);
    V3 horizontal = scale(right, focus_distance * viewport_width);
    V3 vertical = scale(up, focus_distance * viewport_height);
    V3 ulc = sub(sub(add(from, divs(vertical, 2.0)), divs(horizontal, 2.0)),
                 scale(forward, focus_distance));
    memset(out, 0, sizeof *out);
    v3out(from, out->origin);
    v3out(ulc, out->upper_left_corner);
    v3out(forward, out->forward);
    v3out(right, out->right);
    v3out(up, out->up);
    v3out(horizontal, out->horizontal);
    v3out(vertical, out->vertical);
    out->vfov = vfov;
    out->viewport_width = viewport_width;
    out->viewport_height = viewport_height;
    out->lens_radius = aperture * 0.5;
    out->focus_distance = focus_distance;
    out->time_a = time_a;
    out->time_b = time_b;
}
