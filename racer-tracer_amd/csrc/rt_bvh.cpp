// rt_bvh.cpp — see rt_bvh.h.
#include "rt_bvh.h"
#include <algorithm>
#include <cfloat>
#include <cmath>

namespace rtdev {

void primitive_bounds(const RtPrimitive &p, double mn[3], double mx[3]) {
    double lo[3], hi[3];
    switch (p.kind) {
    case RT_PRIM_SPHERE:
    case RT_PRIM_MOVING_SPHERE: {
        double r = std::fabs(p.p[3]);
        for (int k = 0; k < 3; ++k) {
            lo[k] = p.p[k] - r;
            hi[k] = p.p[k] + r;
            if (p.kind == RT_PRIM_MOVING_SPHERE) { // swept between the two centres
                lo[k] = std::min(lo[k], p.center_b[k] - r);
                hi[k] = std::max(hi[k], p.center_b[k] + r);
            }
        }
        break;
    }
    case RT_PRIM_XY_RECT: lo[0] = p.p[0]; hi[0] = p.p[1]; lo[1] = p.p[2]; hi[1] = p.p[3]; lo[2] = hi[2] = p.p[4]; break;
    case RT_PRIM_XZ_RECT: lo[0] = p.p[0]; hi[0] = p.p[1]; lo[2] = p.p[2]; hi[2] = p.p[3]; lo[1] = hi[1] = p.p[4]; break;
    case RT_PRIM_YZ_RECT: lo[1] = p.p[0]; hi[1] = p.p[1]; lo[2] = p.p[2]; hi[2] = p.p[3]; lo[0] = hi[0] = p.p[4]; break;
    default:
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::min(p.p[k], p.p[3 + k]);
            hi[k] = std::max(p.p[k], p.p[3 + k]);
        }
        break;
    }
    for (int k = 0; k < 3; ++k)
        if (lo[k] > hi[k]) std::swap(lo[k], hi[k]);
    if (p.flags & RT_PRIM_HAS_ROTATE_Y) { // object -> world: x' = c x + s z, z' = -s x + c z (rotate_y.rs:55-59)
        double s = p.rot_sin, c = p.rot_cos;
        double nlo[3] = {DBL_MAX, lo[1], DBL_MAX}, nhi[3] = {-DBL_MAX, hi[1], -DBL_MAX};
        for (int i = 0; i < 2; ++i)
            for (int k = 0; k < 2; ++k) {
                double x = i ? hi[0] : lo[0], z = k ? hi[2] : lo[2];
                double wx = c * x + s * z, wz = -s * x + c * z;
                nlo[0] = std::min(nlo[0], wx); nhi[0] = std::max(nhi[0], wx);
                nlo[2] = std::min(nlo[2], wz); nhi[2] = std::max(nhi[2], wz);
            }
        for (int k = 0; k < 3; ++k) { lo[k] = nlo[k]; hi[k] = nhi[k]; }
    }
    if (p.flags & RT_PRIM_HAS_TRANSLATE)
        for (int k = 0; k < 3; ++k) { lo[k] += p.translate[k]; hi[k] += p.translate[k]; }
    for (int k = 0; k < 3; ++k) { // inflate: the slab test must never be stricter than the primitive's own test
        double pad = 1e-9 * std::max({std::fabs(lo[k]), std::fabs(hi[k]), hi[k] - lo[k], 1e-3});
        mn[k] = lo[k] - pad;
        mx[k] = hi[k] + pad;
    }
}

namespace {

struct Node64 { // the builder's own node: f64 box, links as on the device
    double mn[3], mx[3];
    int32_t skip, first, count;
    int32_t axis = -1; // inner node: the axis its children were sorted along (-1: a chain of full leaves, no order to speak of)
};
struct Build64 {
    std::vector<Node64> nodes;
    std::vector<int32_t> prim_index;
};

struct Item {
    double mn[3], mx[3], centroid[3];
    int32_t prim;
};

// Emits the subtree over items[begin, end) in depth-first order; returns its root index.
int emit(Build64 &out, std::vector<Item> &items, int begin, int end, int max_leaf) {
    int me = (int)out.nodes.size();
    out.nodes.emplace_back();
    double mn[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, mx[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
    double cmn[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, cmx[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
    for (int i = begin; i < end; ++i)
        for (int k = 0; k < 3; ++k) {
            mn[k] = std::min(mn[k], items[(size_t)i].mn[k]);
            mx[k] = std::max(mx[k], items[(size_t)i].mx[k]);
            cmn[k] = std::min(cmn[k], items[(size_t)i].centroid[k]);
            cmx[k] = std::max(cmx[k], items[(size_t)i].centroid[k]);
        }
    int n = end - begin;
    bool leaf = n <= max_leaf;
    int axis = 0;
    if (!leaf) {
        for (int k = 1; k < 3; ++k)
            if (cmx[k] - cmn[k] > cmx[axis] - cmn[axis]) axis = k;
        if (!(cmx[axis] - cmn[axis] > 0.0)) leaf = n <= 64; // coincident centroids: splitting cannot separate them
    }
    if (leaf && n > max_leaf) { // degenerate pile: chain of full leaves under one box
        leaf = false;
        axis = -1;
    }
    Node64 node;
    for (int k = 0; k < 3; ++k) { node.mn[k] = mn[k]; node.mx[k] = mx[k]; }
    if (leaf) {
        node.first = (int32_t)out.prim_index.size();
        node.count = n;
        for (int i = begin; i < end; ++i) out.prim_index.push_back(items[(size_t)i].prim);
        node.skip = me + 1;
        out.nodes[(size_t)me] = node;
        return me;
    }
    int mid = begin + n / 2;
    int best_axis_of_split = axis;
    if (axis >= 0) {
        // Surface-area heuristic, full sweep over the three axes: minimise
        // area(L) * |L| + area(R) * |R|.  A plain median split is badly wrong when one
        // primitive dwarfs the rest (the r = 1000 ground sphere of the random scene
        // dragged a scene-sized box through eight levels); SAH isolates it at the root.
        auto half_area = [](const double *lo, const double *hi) {
            double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
            return dx * dy + dy * dz + dz * dx;
        };
        double best_cost = DBL_MAX;
        int best_axis = axis, best_mid = mid;
        std::vector<double> right_area((size_t)n);
        for (int ax = 0; ax < 3; ++ax) {
            std::sort(items.begin() + begin, items.begin() + end, [ax](const Item &a, const Item &b) {
                if (a.centroid[ax] != b.centroid[ax]) return a.centroid[ax] < b.centroid[ax];
                return a.prim < b.prim; // deterministic on ties
            });
            double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
            for (int i = end - 1; i > begin; --i) { // right_area[k] = area of items[begin + k, end)
                for (int k = 0; k < 3; ++k) {
                    lo[k] = std::min(lo[k], items[(size_t)i].mn[k]);
                    hi[k] = std::max(hi[k], items[(size_t)i].mx[k]);
                }
                right_area[(size_t)(i - begin)] = half_area(lo, hi);
            }
            double llo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, lhi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
            for (int i = begin; i < end - 1; ++i) {
                for (int k = 0; k < 3; ++k) {
                    llo[k] = std::min(llo[k], items[(size_t)i].mn[k]);
                    lhi[k] = std::max(lhi[k], items[(size_t)i].mx[k]);
                }
                int nl = i - begin + 1;
                double cost = half_area(llo, lhi) * nl + right_area[(size_t)nl] * (n - nl);
                if (cost < best_cost) {
                    best_cost = cost;
                    best_axis = ax;
                    best_mid = begin + nl;
                }
            }
        }
        std::sort(items.begin() + begin, items.begin() + end, [best_axis](const Item &a, const Item &b) {
            if (a.centroid[best_axis] != b.centroid[best_axis]) return a.centroid[best_axis] < b.centroid[best_axis];
            return a.prim < b.prim;
        });
        mid = best_mid;
        best_axis_of_split = best_axis;
    } else {
        mid = begin + max_leaf;
    }
    node.first = -1;
    node.count = 0;
    node.axis = axis >= 0 ? best_axis_of_split : -1;
    out.nodes[(size_t)me] = node;
    emit(out, items, begin, mid, max_leaf);
    emit(out, items, mid, end, max_leaf);
    out.nodes[(size_t)me].skip = (int32_t)out.nodes.size();
    return me;
}

// x rounded to float toward -inf / +inf
float round_down(double x) {
    float f = (float)x;
    return (double)f > x ? std::nextafterf(f, -INFINITY) : f;
}
float round_up(double x) {
    float f = (float)x;
    return (double)f < x ? std::nextafterf(f, INFINITY) : f;
}

} // namespace

BvhBuild build_bvh(const RtPrimitive *prims, int n_prims, int max_leaf, bool ordered) {
    if (max_leaf < 1) max_leaf = 1;
    if (max_leaf > 7) max_leaf = 7; // BvhNode.first_count keeps the count in three bits
    BvhBuild out;
    for (int k = 0; k < 3; ++k) out.root_mn[k] = out.root_mx[k] = out.center[k] = 0.0;
    if (n_prims <= 0) return out;
    std::vector<Item> items((size_t)n_prims);
    for (int i = 0; i < n_prims; ++i) {
        Item &it = items[(size_t)i];
        primitive_bounds(prims[i], it.mn, it.mx);
        for (int k = 0; k < 3; ++k) it.centroid[k] = 0.5 * (it.mn[k] + it.mx[k]);
        it.prim = i;
    }
    Build64 tree;
    tree.nodes.reserve((size_t)n_prims);
    emit(tree, items, 0, n_prims, max_leaf);
    out.prim_index = tree.prim_index;
    // Device form: f32 boxes around the root's centre.  A ray is clipped to the root box in f64 before the walk, so
    // its origin is at most `extent` from the centre and the f32 slab test is off by a few 2^-24 * extent in the
    // plane positions: the boxes are padded by 2^-19 * extent and rounded outward (rt_bvh_slab.h has the error budget).
    const Node64 &root = tree.nodes[0];
    double extent = 0.0;
    for (int k = 0; k < 3; ++k) {
        out.center[k] = 0.5 * (root.mn[k] + root.mx[k]);
        extent = std::max(extent, 0.5 * (root.mx[k] - root.mn[k]));
    }
    const double pad = std::ldexp(extent, -19);
    for (int k = 0; k < 3; ++k) {
        out.root_mn[k] = root.mn[k] - pad;
        out.root_mx[k] = root.mx[k] + pad;
    }
    out.nodes.resize(tree.nodes.size() + 1);
    for (size_t i = 0; i < tree.nodes.size(); ++i) {
        const Node64 &n = tree.nodes[i];
        BvhNode &q = out.nodes[i];
        for (int k = 0; k < 3; ++k) {
            q.lohi[2 * k] = round_down(n.mn[k] - out.center[k] - pad);
            q.lohi[2 * k + 1] = round_up(n.mx[k] - out.center[k] + pad);
        }
        q.skip = n.skip;
        q.first_count = n.count > 0 ? (n.first << 3) | n.count : 0;
    }
    // the sentinel (rt_device_types.h: BvhNode): all of space, a leaf without primitives, one step past the end
    BvhNode &end = out.nodes.back();
    for (int k = 0; k < 3; ++k) {
        end.lohi[2 * k] = -INFINITY;
        end.lohi[2 * k + 1] = INFINITY;
    }
    end.skip = (int32_t)out.nodes.size();
    end.first_count = BvhNode::kSentinel;
    if (ordered) {
        // DIRECTION-ORDERED COPIES.  The walk visits a node's children in memory order (first child = i + 1), so ONE array
        // serves rays of one direction well: the child that lies first along the ray shrinks best_t before its sibling
        // is asked.  Eight arrays, one per sign octant of the ray direction: at a node whose children were split along
        // axis a, the lower child comes first when the ray travels towards +a, the upper one otherwise.  Same nodes, same
        // boxes, same leaves (they keep pointing into the one leaf-ordered primitive table): only the order and the skip
        // links differ.  Array 0 (all components positive) is the array above.
        const size_t n = tree.nodes.size();
        std::vector<int32_t> right((size_t)n, -1); // second child of inner node i in the builder's order: skip of its first child
        for (size_t i = 0; i < n; ++i)
            if (tree.nodes[i].count == 0) right[i] = tree.nodes[i + 1].skip;
        out.ordered.assign(8 * (n + 1), BvhNode());
        for (int oct = 0; oct < 8; ++oct) {
            BvhNode *arr = out.ordered.data() + (size_t)oct * (n + 1);
            size_t next = 0;
            // iterative depth-first emission; `fix` remembers which emitted node's skip link waits for the subtree's end
            struct Frame { int32_t node; int32_t emitted_at; int stage; };
            std::vector<Frame> stack;
            stack.push_back(Frame{0, -1, 0});
            while (!stack.empty()) {
                Frame &f = stack.back();
                const Node64 &nd = tree.nodes[(size_t)f.node];
                if (f.stage == 0) {
                    f.emitted_at = (int32_t)next;
                    arr[next] = out.nodes[(size_t)f.node]; // box and first_count as in array 0
                    ++next;
                    if (nd.count > 0) { // leaf
                        arr[(size_t)f.emitted_at].skip = (int32_t)next;
                        stack.pop_back();
                        continue;
                    }
                    f.stage = 1;
                    const bool flip = nd.axis >= 0 && ((oct >> nd.axis) & 1) != 0; // the ray travels towards -axis: upper child first
                    const int32_t a = f.node + 1, b = right[(size_t)f.node];
                    const int32_t first = flip ? b : a, second = flip ? a : b;
                    // (second is pushed first so that `first` is processed next)
                    stack.push_back(Frame{second, -1, 0});
                    stack.push_back(Frame{first, -1, 0});
                } else { // both subtrees are out: everything behind them is where a miss of this node continues
                    arr[(size_t)f.emitted_at].skip = (int32_t)next;
                    stack.pop_back();
                }
            }
            arr[n] = out.nodes.back(); // the sentinel
        }
    }
    return out;
}

std::vector<BvhNode> order_bvh_for_origin(const BvhBuild &b, const double origin[3]) {
    const size_t n = b.nodes.empty() ? 0 : b.nodes.size() - 1; // without the sentinel
    std::vector<BvhNode> arr(b.nodes.size());
    if (n == 0) return b.nodes;
    auto dist2 = [&](const BvhNode &q) { // squared distance from the origin to the (f32, centre-relative) box; 0 inside
        double s = 0.0;
        for (int k = 0; k < 3; ++k) {
            const double o = origin[k] - b.center[k], lo = q.mn(k), hi = q.mx(k);
            const double d = o < lo ? lo - o : (o > hi ? o - hi : 0.0);
            s += d * d;
        }
        return s;
    };
    size_t next = 0;
    struct Frame { int32_t node; int32_t emitted_at; int stage; };
    std::vector<Frame> stack;
    stack.push_back(Frame{0, -1, 0});
    while (!stack.empty()) { // depth-first emission, as for the direction-ordered copies of build_bvh
        Frame &f = stack.back();
        const BvhNode &nd = b.nodes[(size_t)f.node];
        if (f.stage == 0) {
            f.emitted_at = (int32_t)next;
            arr[next] = nd;
            ++next;
            if (nd.first_count != 0) { // leaf
                arr[(size_t)f.emitted_at].skip = (int32_t)next;
                stack.pop_back();
                continue;
            }
            f.stage = 1;
            const int32_t first_child = f.node + 1, second_child = b.nodes[(size_t)first_child].skip;
            const bool flip = dist2(b.nodes[(size_t)second_child]) < dist2(b.nodes[(size_t)first_child]);
            const int32_t first = flip ? second_child : first_child, second = flip ? first_child : second_child;
            stack.push_back(Frame{second, -1, 0}); // (pushed first, so that `first` is emitted next)
            stack.push_back(Frame{first, -1, 0});
        } else {
            arr[(size_t)f.emitted_at].skip = (int32_t)next;
            stack.pop_back();
        }
    }
    arr[n] = b.nodes.back(); // the sentinel
    return arr;
}

} // namespace rtdev
