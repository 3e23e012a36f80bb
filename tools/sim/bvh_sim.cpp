// bvh_sim.cpp — CPU model of two closest-hit walks over the `random` scene's BVH: how many boxes and primitives does
// a ray touch, and how many steps does a WAVE of 64 rays take (it pays for its slowest lane)?
//   (a) the skip-link walk of rt_trace_common.h: closest_hit_bvh (one box per step, fixed child order)
//   (b) an ordered walk with a stack over the same tree: both children's boxes per step, nearer child first
// Rays: primary rays of the scene camera in scanline order, then one bounce in a cosine-ish random direction.
//   g++ -O2 -std=c++17 -Iinclude -Iracer-tracer_amd -o racer-tracer_amd/build/bvh_sim tools/sim/bvh_sim.cpp \
//       racer-tracer_amd/build/product/rt_bvh.o -Lracer-tracer_amd/lib -lracer_tracer_amd -Wl,-rpath,$PWD/racer-tracer_amd/lib
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "rt_abi.h"
#include "rt_host.h"
#include "csrc/rt_bvh.h"
#include "csrc/rt_bvh_slab.h"

struct Ray { double o[3], d[3], time; };
struct Tree { // binary tree recovered from the skip-link array
    std::vector<int> left, right; // inner: child node indices; leaf: -1
};

// The DEVICE's box test (rt_bvh_slab.h: the same code closest_hit_bvh runs): the ray is clipped to the root box in
// f64, then every node is tested in f32 around the root's centre.  `tnear` (f64, for child ordering only) is a
// by-product the device walk does not need.
struct DevRay {
    rtdev::SlabRay sr;
    double t0;
    float tmin_f;
    bool misses_root;
};
static DevRay prepare(const rtdev::BvhBuild &bvh, const Ray &r) { // closest_hit_bvh's prologue
    DevRay q;
    double t_enter = -INFINITY, t_exit = INFINITY;
    for (int k = 0; k < 3; ++k) {
        const double inv = 1.0 / r.d[k];
        const double a = (bvh.root_mn[k] - r.o[k]) * inv, b = (bvh.root_mx[k] - r.o[k]) * inv;
        t_enter = std::fmax(t_enter, std::fmin(a, b)); // fmin / fmax drop NaNs like the device's
        t_exit = std::fmin(t_exit, std::fmax(a, b));
    }
    q.misses_root = !(std::fmax(t_enter, 0.001) <= t_exit);
    q.t0 = t_enter > 0.0 ? t_enter : 0.0;
    q.sr = rtdev::slab_ray((float)(std::fma(q.t0, r.d[0], r.o[0]) - bvh.center[0]), (float)(std::fma(q.t0, r.d[1], r.o[1]) - bvh.center[1]),
                           (float)(std::fma(q.t0, r.d[2], r.o[2]) - bvh.center[2]), 1.0 / r.d[0], 1.0 / r.d[1], 1.0 / r.d[2]);
    const float slack = 0x1p-20f;
    q.tmin_f = (float)(0.001 - q.t0) - std::fabs((float)(0.001 - q.t0)) * slack - 0x1p-126f;
    return q;
}
static const rtdev::BvhBuild *g_bvh = nullptr;
static bool hit_box(const rtdev::BvhNode &n, const double c[3], const Ray &r, double tmax, double &tnear) {
    const DevRay q = prepare(*g_bvh, r);
    if (q.misses_root) return false;
    const float slack = 0x1p-20f;
    const float f = (float)(tmax - q.t0);
    const float best_f = f + std::fabs(f) * slack;
    double t0 = 0.001;
    for (int k = 0; k < 3; ++k) { // entry distance in f64, for ordering children only
        const double inv = 1.0 / r.d[k];
        const double a = ((double)n.mn(k) + c[k] - r.o[k]) * inv, b = ((double)n.mx(k) + c[k] - r.o[k]) * inv;
        t0 = std::fmax(t0, std::fmin(a, b));
    }
    tnear = t0;
    return rtdev::slab_hit(n.lohi, q.sr, q.tmin_f, best_f);
}
static bool hit_prim(const RtPrimitive &p, const Ray &r, double tmax, double &t) {
    double c[3];
    for (int k = 0; k < 3; ++k) c[k] = p.p[k];
    if (p.kind == RT_PRIM_MOVING_SPHERE) {
        const double f = (r.time - p.time_a) / (p.time_b - p.time_a);
        for (int k = 0; k < 3; ++k) c[k] += f * (p.center_b[k] - p.p[k]);
    }
    double oc[3] = {r.o[0] - c[0], r.o[1] - c[1], r.o[2] - c[2]};
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

struct Count { long boxes = 0, prims = 0, steps = 0; };

int main(int argc, char **argv) {
    RthSession *session = nullptr;
    if (rth_session_open("scenes/config_c2.yml", "random", nullptr, 1, &session) != RT_OK) {
        fprintf(stderr, "%s\n", rth_last_error_message());
        return 1;
    }
    const RtSceneDesc *d = rth_session_scene(session);
    const RtCamera *cam = rth_session_camera(session);
    const int max_leaf = argc > 1 ? atoi(argv[1]) : 4;
    rtdev::BvhBuild bvh = rtdev::build_bvh(d->primitives, d->n_primitives, max_leaf, true); // with the eight direction-ordered arrays
    g_bvh = &bvh;
    printf("max %d primitives per leaf: ", max_leaf);
    const int n = (int)bvh.nodes.size() - 1; // the array ends with the walk's sentinel (rt_device_types.h: BvhNode), not a node of the tree
    Tree tree;
    tree.left.assign(n, -1);
    tree.right.assign(n, -1);
    int leaves = 0, max_depth = 0;
    { // children of inner node i: i + 1 and skip(i + 1)
        std::vector<std::pair<int, int>> st = {{0, 1}};
        while (!st.empty()) {
            auto [i, depth] = st.back();
            st.pop_back();
            max_depth = std::max(max_depth, depth);
            if (bvh.nodes[i].first_count & 7) { ++leaves; continue; }
            tree.left[i] = i + 1;
            tree.right[i] = bvh.nodes[i + 1].skip;
            st.push_back({tree.left[i], depth + 1});
            st.push_back({tree.right[i], depth + 1});
        }
    }
    printf("%d primitives, %d nodes (%d leaves, %d inner), depth %d\n", d->n_primitives, n, leaves, n - leaves, max_depth);

    const int W = 480, H = 270;
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    std::vector<Ray> rays;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            Ray r;
            const double u = (x + U(rng)) / (W - 1), v = (y + U(rng)) / (H - 1);
            for (int k = 0; k < 3; ++k) {
                r.o[k] = cam->origin[k];
                r.d[k] = cam->upper_left_corner[k] + u * cam->horizontal[k] - v * cam->vertical[k] - cam->origin[k];
            }
            r.time = U(rng);
            rays.push_back(r);
        }
    auto closest_linear = [&](const Ray &r, int &best) {
        double bt = INFINITY;
        best = -1;
        for (int i = 0; i < d->n_primitives; ++i) {
            double t;
            if (hit_prim(d->primitives[i], r, bt, t)) { bt = t; best = i; }
        }
        return bt;
    };
    auto walk_skip = [&](const Ray &r, Count &c, int &best) {
        double bt = INFINITY;
        best = -1;
        int i = 0, steps = 0;
        while (i < n) {
            ++c.boxes; ++steps;
            double tn;
            if (hit_box(bvh.nodes[i], bvh.center, r, bt, tn)) {
                const int fc = bvh.nodes[i].first_count;
                for (int k = 0; k < (fc & 7); ++k) {
                    ++c.prims;
                    const int pi = bvh.prim_index[(fc >> 3) + k];
                    double t;
                    if (hit_prim(d->primitives[pi], r, bt, t)) { bt = t; best = pi; }
                }
                i = i + 1;
            } else {
                i = bvh.nodes[i].skip;
            }
        }
        return steps;
    };
    // the same walk over the copy of the node array that is ordered for the ray's direction octant (rt_bvh.cpp; what the
    // device does for trees that stay in global memory)
    auto walk_skip_ordered = [&](const Ray &r, Count &c, int &best) {
        const int oct = (r.d[0] < 0.0 ? 1 : 0) | (r.d[1] < 0.0 ? 2 : 0) | (r.d[2] < 0.0 ? 4 : 0);
        const rtdev::BvhNode *arr = bvh.ordered.data() + (size_t)oct * (size_t)(n + 1);
        double bt = INFINITY;
        best = -1;
        int i = 0, steps = 0;
        while (i < n) {
            ++c.boxes; ++steps;
            double tn;
            if (hit_box(arr[i], bvh.center, r, bt, tn)) {
                const int fc = arr[i].first_count;
                for (int k = 0; k < (fc & 7); ++k) {
                    ++c.prims;
                    const int pi = bvh.prim_index[(fc >> 3) + k];
                    double t;
                    if (hit_prim(d->primitives[pi], r, bt, t)) { bt = t; best = pi; }
                }
                i = i + 1;
            } else {
                i = arr[i].skip;
            }
        }
        return steps;
    };
    // The skip-link walk as the sequence of phases a lane of the device's while-while loop goes through:
    // (descent steps until a leaf is entered, primitives of that leaf) ... (descent steps until the tree is left, 0).
    auto walk_trace = [&](const Ray &r, std::vector<std::pair<int, int>> &phases) {
        double bt = INFINITY;
        int i = 0, steps = 0;
        phases.clear();
        while (i < n) {
            ++steps;
            double tn;
            if (hit_box(bvh.nodes[i], bvh.center, r, bt, tn)) {
                const int fc = bvh.nodes[i].first_count;
                if (fc & 7) {
                    phases.push_back({steps, fc & 7});
                    steps = 0;
                    for (int k = 0; k < (fc & 7); ++k) {
                        double t;
                        if (hit_prim(d->primitives[bvh.prim_index[(fc >> 3) + k]], r, bt, t)) bt = t;
                    }
                }
                i = i + 1;
            } else {
                i = bvh.nodes[i].skip;
            }
        }
        phases.push_back({steps, 0});
    };
    // What would lanes that fetch a NEW ray when theirs is through buy?  A wave of 64 lanes in lockstep, while-while:
    // a descent round costs `cd` (every lane that is not at a leaf takes one step), a leaf round `cl` per primitive (the
    // wave runs max-count rounds).  keep_in_lanes: each lane owns one ray and the wave ends when its slowest ray does
    // (the kernel today); refill: a lane whose ray is through takes the next ray of the queue at the next loop top
    // (a wavefront walk over rays that wait in memory), `cf` per loop top at which somebody fetches.
    // straggle: the descent rounds of an outer round stop once no more than this many lanes are still descending (they go on
    // in the next outer round while the others test their leaves); 0 = every lane reaches its leaf first (the kernel today)
    auto wave_cost = [&](const std::vector<std::vector<std::pair<int, int>>> &traces, bool refill, double cd, double cl, double cf,
                         double &lane_use, int straggle = 0) {
        struct Lane { int ray = -1; size_t phase = 0; int left = 0; };
        std::vector<Lane> lanes(64);
        size_t next = 0;
        double cost = 0, useful = 0;
        auto assign = [&](Lane &L) {
            L.ray = (int)next++;
            L.phase = 0;
            L.left = traces[(size_t)L.ray][0].first;
        };
        while (true) {
            bool fetched = false, any = false;
            for (Lane &L : lanes) {
                if (L.ray < 0 && next < traces.size() && (refill || std::all_of(lanes.begin(), lanes.end(), [](const Lane &x) { return x.ray < 0; }))) {
                    // without refill a new group of 64 starts only when the whole wave is idle
                }
            }
            const bool all_idle = std::all_of(lanes.begin(), lanes.end(), [](const Lane &x) { return x.ray < 0; });
            for (Lane &L : lanes)
                if (L.ray < 0 && next < traces.size() && (refill || all_idle)) { assign(L); fetched = true; }
            for (const Lane &L : lanes) any = any || L.ray >= 0;
            if (!any) break;
            if (fetched && refill) cost += cf;
            // descent rounds until every lane with a ray stands at a leaf (or has left the tree)
            while (true) {
                int walking = 0;
                for (const Lane &L : lanes) walking += L.ray >= 0 && L.left > 0;
                if (walking == 0 || (walking <= straggle && walking < 64)) {
                    int at_leaf = 0;
                    for (const Lane &L : lanes) at_leaf += L.ray >= 0 && L.left == 0;
                    if (walking == 0 || at_leaf > 0) break;
                }
                for (Lane &L : lanes)
                    if (L.ray >= 0 && L.left > 0) --L.left;
                cost += cd;
                useful += cd * walking / 64.0;
            }
            // leaf rounds (lanes still descending sit them out)
            int rounds = 0;
            double lane_rounds = 0;
            for (Lane &L : lanes)
                if (L.ray >= 0 && L.left == 0) {
                    const int c = traces[(size_t)L.ray][L.phase].second;
                    rounds = std::max(rounds, c);
                    lane_rounds += c;
                }
            cost += cl * rounds;
            useful += cl * lane_rounds / 64.0;
            for (Lane &L : lanes)
                if (L.ray >= 0 && L.left == 0) {
                    const auto &tr = traces[(size_t)L.ray];
                    if (tr[L.phase].second == 0) { L.ray = -1; continue; } // left the tree
                    ++L.phase;
                    L.left = tr[L.phase].first;
                    if (L.left == 0 && tr[L.phase].second == 0) L.ray = -1;
                }
        }
        lane_use = useful / cost;
        return cost / (double)traces.size();
    };
    // (c) skip links, two nodes per step: node i and its layout successor i + 1 are fetched and tested together; when
    //     the walk goes on at i + 1 anyway (i is a hit inner node, or a missed node whose skip link is i + 1) the
    //     second result is used at once.  No stack, no new layout.
    auto walk_two = [&](const Ray &r, Count &c, int &best) {
        double bt = INFINITY;
        best = -1;
        int i = 0, steps = 0;
        auto leaf = [&](int j) {
            const int fc = bvh.nodes[j].first_count;
            for (int k = 0; k < (fc & 7); ++k) {
                ++c.prims;
                const int pi = bvh.prim_index[(fc >> 3) + k];
                double t;
                if (hit_prim(d->primitives[pi], r, bt, t)) { bt = t; best = pi; }
            }
        };
        while (i < n) {
            ++steps;
            double tn;
            c.boxes += i + 1 < n ? 2 : 1;
            const bool ha = hit_box(bvh.nodes[i], bvh.center, r, bt, tn);
            const bool hb = i + 1 < n && hit_box(bvh.nodes[i + 1], bvh.center, r, bt, tn);
            const bool a_leaf = (bvh.nodes[i].first_count & 7) != 0;
            if (ha && a_leaf) { leaf(i); i = i + 1; continue; }
            const bool next_is_b = ha || bvh.nodes[i].skip == i + 1;
            if (!next_is_b) { i = bvh.nodes[i].skip; continue; }
            if (i + 1 >= n) break;
            if (!hb) { i = bvh.nodes[i + 1].skip; continue; }
            if (bvh.nodes[i + 1].first_count & 7) { leaf(i + 1); i = i + 2; continue; }
            i = i + 2;
        }
        return steps;
    };
    auto walk_ordered = [&](const Ray &r, Count &c, int &best, int &max_sp) {
        double bt = INFINITY;
        best = -1;
        int stack[64], sp = 0, cur = 0, steps = 0;
        auto leaf = [&](int i) {
            const int fc = bvh.nodes[i].first_count;
            for (int k = 0; k < (fc & 7); ++k) {
                ++c.prims;
                const int pi = bvh.prim_index[(fc >> 3) + k];
                double t;
                if (hit_prim(d->primitives[pi], r, bt, t)) { bt = t; best = pi; }
            }
        };
        double tn;
        ++c.boxes;
        if (!hit_box(bvh.nodes[0], bvh.center, r, bt, tn)) return 1;
        for (;;) {
            ++steps;
            if (tree.left[cur] < 0) {
                leaf(cur);
                if (sp == 0) break;
                cur = stack[--sp];
                continue;
            }
            const int L = tree.left[cur], R = tree.right[cur];
            double tl, tr;
            c.boxes += 2;
            const bool hl = hit_box(bvh.nodes[L], bvh.center, r, bt, tl), hr = hit_box(bvh.nodes[R], bvh.center, r, bt, tr);
            if (hl && hr) {
                const bool lfirst = tl <= tr;
                stack[sp++] = lfirst ? R : L;
                max_sp = std::max(max_sp, sp);
                cur = lfirst ? L : R;
            } else if (hl || hr) {
                cur = hl ? L : R;
            } else {
                if (sp == 0) break;
                cur = stack[--sp];
            }
        }
        return steps;
    };
    { // axis-parallel rays: direction components of exactly 0 (1/d = inf).  The f32 slab test once turned these into
      // inf - inf = NaN and culled the root (rt_bvh_slab.h: slab_finite); every hit must be the linear scan's.
        std::uniform_real_distribution<double> X(-12.0, 12.0), Y(0.05, 2.5);
        long differ = 0, hits = 0;
        const int n_axis = 12000;
        for (int k = 0; k < n_axis; ++k) {
            Ray r;
            r.o[0] = X(rng); r.o[1] = Y(rng); r.o[2] = X(rng);
            const int axis = k % 3, kind = (k / 3) % 4;
            for (int j = 0; j < 3; ++j) r.d[j] = 0.0;
            r.d[axis] = (k & 8) ? 1.0 : -1.0;                        // kind 0: two zero components
            if (kind == 1) r.d[(axis + 1) % 3] = U(rng) - 0.5;       // one zero component
            if (kind == 2) r.d[(axis + 2) % 3] = -0.0;               // a negative zero among them
            if (kind == 3) { r.d[axis] = 1e-300; r.d[(axis + 1) % 3] = (k & 8) ? 1.0 : -1.0; } // 1/d beyond the f32 range
            if (k % 7 == 0) { r.o[0] = std::round(r.o[0]); r.o[2] = std::round(r.o[2]); }      // origins ON lattice planes
            r.time = U(rng);
            Count c;
            int bs, bl;
            walk_skip(r, c, bs);
            closest_linear(r, bl);
            differ += bs != bl;
            hits += bl >= 0;
        }
        printf("axis-parallel: %d rays, %ld hit something | closest hits differ: %ld\n", n_axis, hits, differ);
    }
    // NOTE the ordered walk re-tests a popped node's box implicitly never: a popped far child may have become
    // prunable (best_t shrank); count that variant too
    for (int bounce = 0; bounce < 3; ++bounce) {
        Count a, b, c2, e;
        long wave_a = 0, wave_b = 0, wave_c = 0, wave_e = 0, mismatches = 0;
        int max_sp = 0;
        std::vector<Ray> next;
        for (size_t w = 0; w < rays.size(); w += 64) {
            int ma = 0, mb = 0, mc = 0, me = 0;
            for (size_t j = w; j < std::min(rays.size(), w + 64); ++j) {
                int ba, bb;
                ma = std::max(ma, walk_skip(rays[j], a, ba));
                { int be; me = std::max(me, walk_skip_ordered(rays[j], e, be)); mismatches += be != ba; }
                mb = std::max(mb, walk_ordered(rays[j], b, bb, max_sp));
                mismatches += ba != bb;
                { int bc; mc = std::max(mc, walk_two(rays[j], c2, bc)); mismatches += bc != ba; }
                if (ba >= 0) { // bounce: origin at the hit, direction normal + random unit vector
                    int bl;
                    const double t = closest_linear(rays[j], bl);
                    mismatches += bl != ba;
                    const RtPrimitive &p = d->primitives[ba];
                    Ray r;
                    double c[3], nrm[3], len = 0;
                    for (int k = 0; k < 3; ++k) c[k] = p.p[k];
                    if (p.kind == RT_PRIM_MOVING_SPHERE) {
                        const double f = (rays[j].time - p.time_a) / (p.time_b - p.time_a);
                        for (int k = 0; k < 3; ++k) c[k] += f * (p.center_b[k] - p.p[k]);
                    }
                    for (int k = 0; k < 3; ++k) {
                        r.o[k] = rays[j].o[k] + t * rays[j].d[k];
                        nrm[k] = (r.o[k] - c[k]) / p.p[3];
                    }
                    double v[3];
                    do {
                        len = 0;
                        for (int k = 0; k < 3; ++k) { v[k] = 2 * U(rng) - 1; len += v[k] * v[k]; }
                    } while (len >= 1 || len == 0);
                    for (int k = 0; k < 3; ++k) r.d[k] = nrm[k] + v[k] / std::sqrt(len);
                    r.time = rays[j].time;
                    next.push_back(r);
                }
            }
            wave_a += ma;
            wave_b += mb;
            wave_c += mc;
            wave_e += me;
        }
        const double nr = (double)rays.size(), nw = std::ceil(nr / 64);
        printf("bounce %d: %zu rays | skip-link: %.1f boxes %.1f prims per ray, %.1f steps per wave | ordered+stack: %.1f boxes %.1f prims per ray, "
               "%.1f steps per wave (2 boxes each), stack depth %d | skip-link, two per step: %.1f boxes, %.1f steps per wave | "
               "skip-link over the direction-ordered array: %.1f boxes %.1f prims per ray, %.1f steps per wave | closest hits differ: %ld\n",
               bounce, rays.size(), a.boxes / nr, a.prims / nr, wave_a / nw, b.boxes / nr, b.prims / nr, wave_b / nw, max_sp, c2.boxes / nr, wave_c / nw,
               e.boxes / nr, e.prims / nr, wave_e / nw, mismatches);
        { // lockstep model of a wave over this bounce's rays: today's walk against one whose lanes fetch new rays
            std::vector<std::vector<std::pair<int, int>>> traces(rays.size());
            for (size_t j = 0; j < rays.size(); ++j) walk_trace(rays[j], traces[j]);
            // costs in SIMD cycles per wave: a descent step is ~24 32-bit vector instructions, a leaf primitive ~40 f64 ones
            const double cd = 24 * 2.3, cl = 40 * 4.2, cf = 60 * 2.3;
            double use_a = 0, use_b = 0;
            const double a1 = wave_cost(traces, false, cd, cl, cf, use_a), b1 = wave_cost(traces, true, cd, cl, cf, use_b);
            printf("bounce %d, wave model: rays kept in lanes %.0f cycles per ray (%.0f %% of lane-cycles useful) | lanes refilled from a queue %.0f (%.0f %%) | x %.2f\n",
                   bounce, a1, 100 * use_a, b1, 100 * use_b, a1 / b1);
            for (int straggle : {8, 16, 32}) {
                double ua = 0, ub = 0;
                const double a2 = wave_cost(traces, false, cd, cl, cf, ua, straggle), b2 = wave_cost(traces, true, cd, cl, cf, ub, straggle);
                printf("bounce %d, wave model, leaves tested once <= %d lanes still descend: kept in lanes %.0f (%.0f %%) | refilled %.0f (%.0f %%)\n",
                       bounce, straggle, a2, 100 * ua, b2, 100 * ub);
            }
        }
        rays.swap(next);
    }
    rth_session_close(session);
    return 0;
}
