// bvh_rotate.cpp — what would a better TREE buy the skip-link walk of the `random` scene?  (CPU model, review item of
// round 3: "measure what tree quality buys — re-braiding / reinsertion passes".)
//
// Takes the tree rt_bvh.cpp builds (SAH sweep, <= max_leaf primitives per leaf), measures its SAH cost and the boxes /
// primitives a ray touches in the device's fixed-order walk (every node whose parent was hit, left subtree before right:
// what the skip links encode), then improves the topology with
//   (1) tree rotations (Kensler 2008): at every inner node, swap a child with a grandchild of the other side whenever
//       that shrinks the surface area of the node that changes, passes until nothing improves;
//   (2) reinsertion (Bittner et al. 2013, the simple form): take the subtrees with the largest area out and put each
//       back where it increases the tree's cost least (branch-and-bound over the induced cost);
// and measures again.  Rays: primary rays of the scene camera, then one diffuse bounce from their hit points.
//   g++ -O2 -std=c++17 -Iinclude -Iracer-tracer_amd -o racer-tracer_amd/build/bvh_rotate tools/sim/bvh_rotate.cpp \
//       racer-tracer_amd/build/product/rt_bvh.o -Lracer-tracer_amd/lib -lracer_tracer_amd -Wl,-rpath,$PWD/racer-tracer_amd/lib
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <random>
#include <vector>
#include "rt_abi.h"
#include "rt_host.h"
#include "csrc/rt_bvh.h"

struct Ray { double o[3], d[3], time; };
struct Node {
    double mn[3], mx[3];
    int l = -1, r = -1, parent = -1;
    std::vector<int> prims; // leaf
    bool leaf() const { return l < 0; }
};
static double area(const double *mn, const double *mx) {
    const double dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    return 2.0 * (dx * dy + dy * dz + dz * dx);
}
static double area(const Node &n) { return area(n.mn, n.mx); }
static void unite(const Node &a, const Node &b, double *mn, double *mx) {
    for (int k = 0; k < 3; ++k) { mn[k] = std::min(a.mn[k], b.mn[k]); mx[k] = std::max(a.mx[k], b.mx[k]); }
}

struct Tree {
    std::vector<Node> n;
    int root = 0;
    void refit_up(int i) {
        for (; i >= 0; i = n[i].parent)
            if (!n[i].leaf()) unite(n[n[i].l], n[n[i].r], n[i].mn, n[i].mx);
    }
    // expected cost, in box tests, per ray that hits the box of node `top` (the whole tree by default; the ground sphere of
    // the random scene makes the root's box a thousand times the field's, so the field's subtree is reported on its own)
    double sah(int top = -1, double c_box = 1.0, double c_prim = 2.0) const {
        if (top < 0) top = root;
        double cost = 0;
        std::function<void(int)> go = [&](int i) {
            cost += c_box * area(n[i]);
            if (n[i].leaf()) cost += c_prim * area(n[i]) * (double)n[i].prims.size();
            else { go(n[i].l); go(n[i].r); }
        };
        go(top);
        return cost / area(n[top]);
    }
};

static bool hit_prim(const RtPrimitive &p, const Ray &r, double tmax, double &t) {
    double c[3];
    for (int k = 0; k < 3; ++k) c[k] = p.p[k];
    if (p.kind == RT_PRIM_MOVING_SPHERE) {
        const double f = (r.time - p.time_a) / (p.time_b - p.time_a);
        for (int k = 0; k < 3; ++k) c[k] += f * (p.center_b[k] - p.p[k]);
    }
    const double oc[3] = {r.o[0] - c[0], r.o[1] - c[1], r.o[2] - c[2]};
    const double a = r.d[0] * r.d[0] + r.d[1] * r.d[1] + r.d[2] * r.d[2];
    const double hb = oc[0] * r.d[0] + oc[1] * r.d[1] + oc[2] * r.d[2];
    const double cc = oc[0] * oc[0] + oc[1] * oc[1] + oc[2] * oc[2] - p.p[3] * p.p[3];
    const double disc = hb * hb - a * cc;
    if (disc < 0) return false;
    const double sq = std::sqrt(disc);
    double root = (-hb - sq) / a;
    if (root < 0.001 || root > tmax) {
        root = (-hb + sq) / a;
        if (root < 0.001 || root > tmax) return false;
    }
    t = root;
    return true;
}
static bool hit_box(const Node &n, const Ray &r, double tmax) {
    double t0 = 0.001, t1 = tmax;
    for (int k = 0; k < 3; ++k) {
        const double inv = 1.0 / r.d[k];
        const double a = (n.mn[k] - r.o[k]) * inv, b = (n.mx[k] - r.o[k]) * inv;
        t0 = std::fmax(t0, std::fmin(a, b));
        t1 = std::fmin(t1, std::fmax(a, b));
    }
    return t0 <= t1;
}

struct Count { long boxes = 0, prims = 0, rays = 0; };
static double walk(const Tree &T, const RtSceneDesc *d, const Ray &r, Count &c, int &best) { // the skip-link walk's visits
    double bt = INFINITY;
    best = -1;
    ++c.rays;
    std::function<void(int)> go = [&](int i) {
        ++c.boxes;
        if (!hit_box(T.n[i], r, bt)) return;
        if (T.n[i].leaf()) {
            for (int pi : T.n[i].prims) {
                ++c.prims;
                double t;
                if (hit_prim(d->primitives[pi], r, bt, t)) { bt = t; best = pi; }
            }
            return;
        }
        go(T.n[i].l);
        go(T.n[i].r);
    };
    go(T.root);
    return bt;
}

// ---- (1) rotations
static int rotate_pass(Tree &T) {
    int done = 0;
    std::function<void(int)> go = [&](int i) {
        Node &N = T.n[i];
        if (N.leaf()) return;
        go(N.l);
        go(N.r);
        // candidates: swap child A (= l or r) with a grandchild G under the OTHER child B: B's box becomes union(A, other grandchild)
        double best_gain = 1e-12;
        int best_a = -1, best_g = -1;
        for (int side = 0; side < 2; ++side) {
            const int a = side ? N.r : N.l, b = side ? N.l : N.r;
            if (T.n[b].leaf()) continue;
            for (int gs = 0; gs < 2; ++gs) {
                const int g = gs ? T.n[b].r : T.n[b].l, other = gs ? T.n[b].l : T.n[b].r;
                double mn[3], mx[3];
                unite(T.n[a], T.n[other], mn, mx);
                const double gain = area(T.n[b]) - area(mn, mx); // the node's own box (union of everything below) does not change
                if (gain > best_gain) { best_gain = gain; best_a = a; best_g = g; }
            }
        }
        if (best_a >= 0) {
            const int a = best_a, g = best_g, b = T.n[g].parent;
            (N.l == a ? N.l : N.r) = g;
            (T.n[b].l == g ? T.n[b].l : T.n[b].r) = a;
            T.n[g].parent = i;
            T.n[a].parent = b;
            unite(T.n[T.n[b].l], T.n[T.n[b].r], T.n[b].mn, T.n[b].mx);
            ++done;
        }
    };
    go(T.root);
    return done;
}

// ---- (2) reinsertion of the subtrees with the largest boxes
static int reinsert_pass(Tree &T, double fraction) {
    std::vector<int> cand;
    for (int i = 0; i < (int)T.n.size(); ++i)
        if (i != T.root && T.n[i].parent >= 0 && T.n[i].parent != T.root) cand.push_back(i);
    std::sort(cand.begin(), cand.end(), [&](int a, int b) { return area(T.n[a]) > area(T.n[b]); });
    cand.resize((size_t)(cand.size() * fraction));
    int moved = 0;
    for (int x : cand) {
        const int p = T.n[x].parent;
        if (p < 0 || p == T.root) continue;
        const int gp = T.n[p].parent, sib = T.n[p].l == x ? T.n[p].r : T.n[p].l;
        const double before = T.sah();
        // take x out: its sibling takes the parent's place
        (T.n[gp].l == p ? T.n[gp].l : T.n[gp].r) = sib;
        T.n[sib].parent = gp;
        T.refit_up(gp);
        // best place: the node y (not inside x) for which making a new parent of (y, x) costs least: area(union) + the growth of y's ancestors
        double best = INFINITY;
        int best_y = -1;
        std::function<void(int, double)> search = [&](int y, double induced) {
            double mn[3], mx[3];
            unite(T.n[y], T.n[x], mn, mx);
            const double direct = area(mn, mx);
            if (induced + direct < best) { best = induced + direct; best_y = y; }
            const double grow = direct - area(T.n[y]);
            if (!T.n[y].leaf() && induced + grow + area(T.n[x]) < best) {
                search(T.n[y].l, induced + grow);
                search(T.n[y].r, induced + grow);
            }
        };
        search(T.root, 0.0);
        // node p is re-used as the new parent of (best_y, x)
        const int y = best_y, yp = T.n[y].parent;
        T.n[p].l = y;
        T.n[p].r = x;
        T.n[p].parent = yp;
        if (yp >= 0) (T.n[yp].l == y ? T.n[yp].l : T.n[yp].r) = p;
        else T.root = p;
        T.n[y].parent = p;
        T.n[x].parent = p;
        T.refit_up(p);
        moved += T.sah() < before - 1e-12;
    }
    return moved;
}

int main(int argc, char **argv) {
    RthSession *session = nullptr;
    if (rth_session_open("scenes/config_c2.yml", "random", nullptr, 1, &session) != RT_OK) {
        fprintf(stderr, "%s\n", rth_last_error_message());
        return 1;
    }
    const RtSceneDesc *d = rth_session_scene(session);
    const RtCamera *cam = rth_session_camera(session);
    const int max_leaf = argc > 1 ? atoi(argv[1]) : 3;
    rtdev::BvhBuild bvh = rtdev::build_bvh(d->primitives, d->n_primitives, max_leaf);
    const int n = (int)bvh.nodes.size() - 1;
    Tree T;
    T.n.resize((size_t)n);
    { // topology from the skip links, boxes in f64 from the primitives
        std::function<void(int, int)> go = [&](int i, int parent) {
            Node &N = T.n[i];
            N.parent = parent;
            const int fc = bvh.nodes[i].first_count;
            if (fc & 7) {
                for (int k = 0; k < (fc & 7); ++k) N.prims.push_back(bvh.prim_index[(fc >> 3) + k]);
                for (int k = 0; k < 3; ++k) { N.mn[k] = INFINITY; N.mx[k] = -INFINITY; }
                for (int pi : N.prims) {
                    double mn[3], mx[3];
                    rtdev::primitive_bounds(d->primitives[pi], mn, mx);
                    for (int k = 0; k < 3; ++k) { N.mn[k] = std::min(N.mn[k], mn[k]); N.mx[k] = std::max(N.mx[k], mx[k]); }
                }
                return;
            }
            N.l = i + 1;
            N.r = bvh.nodes[i + 1].skip;
            go(N.l, i);
            go(N.r, i);
            unite(T.n[N.l], T.n[N.r], N.mn, N.mx);
        };
        go(0, -1);
    }
    // rays
    const int W = 480, H = 270;
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    std::vector<Ray> primary, bounce;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            Ray r;
            const double u = (x + U(rng)) / (W - 1), v = (y + U(rng)) / (H - 1);
            for (int k = 0; k < 3; ++k) {
                r.o[k] = cam->origin[k];
                r.d[k] = cam->upper_left_corner[k] + u * cam->horizontal[k] - v * cam->vertical[k] - cam->origin[k];
            }
            r.time = U(rng);
            primary.push_back(r);
        }
    {
        Count c;
        for (const Ray &r : primary) {
            int best;
            const double t = walk(T, d, r, c, best);
            if (best < 0) continue;
            Ray b;
            double nrm[3], len = 0;
            const RtPrimitive &P = d->primitives[best];
            double ctr[3] = {P.p[0], P.p[1], P.p[2]};
            if (P.kind == RT_PRIM_MOVING_SPHERE)
                for (int k = 0; k < 3; ++k) ctr[k] += (r.time - P.time_a) / (P.time_b - P.time_a) * (P.center_b[k] - P.p[k]);
            for (int k = 0; k < 3; ++k) { b.o[k] = r.o[k] + t * r.d[k]; nrm[k] = (b.o[k] - ctr[k]) / P.p[3]; }
            for (;;) { // lambertian.rs: normal + random unit vector
                double s[3] = {2 * U(rng) - 1, 2 * U(rng) - 1, 2 * U(rng) - 1};
                len = s[0] * s[0] + s[1] * s[1] + s[2] * s[2];
                if (len >= 1 || len < 1e-12) continue;
                for (int k = 0; k < 3; ++k) b.d[k] = nrm[k] + s[k] / std::sqrt(len);
                break;
            }
            b.time = r.time;
            bounce.push_back(b);
        }
    }
    auto measure = [&](const char *what) {
        Count cp, cb;
        int best;
        for (const Ray &r : primary) walk(T, d, r, cp, best);
        for (const Ray &r : bounce) walk(T, d, r, cb, best);
        int field = T.root; // the subtree without the dominating primitive: descend while one child is a single leaf
        while (!T.n[field].leaf() && (T.n[T.n[field].l].leaf() || T.n[T.n[field].r].leaf()) && area(T.n[field]) > 1e5)
            field = T.n[T.n[field].l].leaf() ? T.n[field].r : T.n[field].l;
        printf("%-44s SAH cost %7.2f (the field's subtree: %6.2f) | primary rays: %5.2f boxes, %4.2f primitives | bounce rays: %5.2f boxes, %4.2f primitives\n", what, T.sah(), T.sah(field),
               (double)cp.boxes / cp.rays, (double)cp.prims / cp.rays, (double)cb.boxes / cb.rays, (double)cb.prims / cb.rays);
    };
    printf("random scene, %d primitives, %d nodes, <= %d primitives per leaf; %zu primary and %zu bounce rays\n", d->n_primitives, n, max_leaf,
           primary.size(), bounce.size());
    measure("as built (SAH sweep, rt_bvh.cpp)");
    int total = 0;
    for (int pass = 0; pass < 32; ++pass) {
        const int k = rotate_pass(T);
        total += k;
        if (k == 0) break;
    }
    char line[96];
    snprintf(line, sizeof line, "+ tree rotations (%d applied)", total);
    measure(line);
    int moved = 0;
    for (int pass = 0; pass < 4; ++pass) moved += reinsert_pass(T, 0.25);
    snprintf(line, sizeof line, "+ reinsertion of the largest subtrees (%d kept)", moved);
    measure(line);
    total = 0;
    for (int pass = 0; pass < 32; ++pass) {
        const int k = rotate_pass(T);
        total += k;
        if (k == 0) break;
    }
    snprintf(line, sizeof line, "+ rotations again (%d applied)", total);
    measure(line);
    { // (3) no new topology at all: every inner node's children in the order of their boxes' distance from the CAMERA (what
      // a fixed-order walk wants for primary rays; indifferent to bounce rays)
        auto dist2 = [&](const Node &nd) {
            double s = 0;
            for (int k = 0; k < 3; ++k) {
                const double o = cam->origin[k], dlt = o < nd.mn[k] ? nd.mn[k] - o : (o > nd.mx[k] ? o - nd.mx[k] : 0.0);
                s += dlt * dlt;
            }
            return s;
        };
        int swapped = 0;
        for (Node &nd : T.n)
            if (!nd.leaf() && dist2(T.n[nd.r]) < dist2(T.n[nd.l])) { std::swap(nd.l, nd.r); ++swapped; }
        snprintf(line, sizeof line, "children nearest-to-camera first (%d swapped)", swapped);
        measure(line);
    }
    { // (4) the PRODUCT's routine (rt_bvh.h: order_bvh_for_origin, what enqueue_render uploads for this camera): the skip-link
      // walk over its array must find the linear scan's closest hit for every ray, and touch fewer boxes than over the
      // builder's own order
        auto skip_walk = [&](const std::vector<rtdev::BvhNode> &arr, const Ray &r, Count &c, int &best) {
            double bt = INFINITY;
            best = -1;
            ++c.rays;
            int i = 0;
            while (i < n) {
                ++c.boxes;
                Node box;
                for (int k = 0; k < 3; ++k) { box.mn[k] = (double)arr[(size_t)i].mn(k) + bvh.center[k]; box.mx[k] = (double)arr[(size_t)i].mx(k) + bvh.center[k]; }
                if (hit_box(box, r, bt)) {
                    const int fc = arr[(size_t)i].first_count;
                    for (int k = 0; k < (fc & 7); ++k) {
                        ++c.prims;
                        const int pi = bvh.prim_index[(size_t)((fc >> 3) + k)];
                        double t;
                        if (hit_prim(d->primitives[pi], r, bt, t)) { bt = t; best = pi; }
                    }
                    i = i + 1;
                } else {
                    i = arr[(size_t)i].skip;
                }
            }
            return bt;
        };
        const std::vector<rtdev::BvhNode> ordered = rtdev::order_bvh_for_origin(bvh, cam->origin);
        long mismatches = 0;
        Count c0p, c0b, c1p, c1b;
        for (int pass = 0; pass < 2; ++pass)
            for (const Ray &r : pass ? bounce : primary) {
                int b0, b1, bl = -1;
                const double t0 = skip_walk(bvh.nodes, r, pass ? c0b : c0p, b0), t1 = skip_walk(ordered, r, pass ? c1b : c1p, b1);
                double bt = INFINITY;
                for (int i = 0; i < d->n_primitives; ++i) {
                    double t;
                    if (hit_prim(d->primitives[i], r, bt, t)) { bt = t; bl = i; }
                }
                mismatches += !(t0 == bt && t1 == bt) || (bl >= 0 && (b0 < 0 || b1 < 0));
            }
        printf("order_bvh_for_origin(camera): %ld rays whose closest hit differs from the linear scan's; boxes per primary ray %.2f -> %.2f, per bounce ray %.2f -> %.2f; "
               "leaf primitives %.2f -> %.2f, %.2f -> %.2f\n", mismatches, (double)c0p.boxes / c0p.rays, (double)c1p.boxes / c1p.rays, (double)c0b.boxes / c0b.rays,
               (double)c1b.boxes / c1b.rays, (double)c0p.prims / c0p.rays, (double)c1p.prims / c1p.rays, (double)c0b.prims / c0b.rays, (double)c1b.prims / c1b.rays);
    }
    rth_session_close(session);
    return 0;
}
