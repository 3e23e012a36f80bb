// rt_bvh.h — bounding-volume hierarchy for scenes too large for the
// brute-force closest-hit loop.
//
// The reference walks a binary BVH with one object per leaf, random split
// axis in {x, y} and both children always visited (bvh_node.rs:31-132).  Only
// its RESULT is part of the contract — the closest hit in [t_min, t_max],
// topology-free except for exact ties (SURVEY B-15) — so this one is built for
// the GPU instead: surface-area-heuristic splits (full sweep over three axes),
// up to four primitives per leaf stored contiguously in leaf order, nodes in
// depth-first order with a skip link, so a lane walks it with one integer of
// state and no stack:
//
//     i = 0;  while (i < n) { if (ray hits node i) { test its primitives (leaf); i = i + 1 or skip } else i = skip[i]; }
//
// Boxes are the TRUE bounds of the primitives (not rotate_y.rs:66-90's
// mis-sized ones), inflated so the slab test can never reject a primitive
// whose own intersection routine accepts the ray; on the device they are
// single-precision culling boxes around the root's centre (BvhNode).
#pragma once
#include <stdint.h>
#include <vector>
#include "../../include/rt_abi.h"
#include "rt_device_types.h"

namespace rtdev {

struct BvhBuild {
    std::vector<BvhNode> nodes;      // compact device form (rt_device_types.h); the LAST entry is the sentinel, not a node of the tree
    std::vector<BvhNode> ordered;    // optional: eight arrays of nodes.size() entries, one per sign octant of the ray direction (rt_bvh.cpp)
    std::vector<int32_t> prim_index; // leaves refer to ranges of this list
    double root_mn[3], root_mx[3];   // root box, f64, padded like the node boxes
    double center[3];                // the node boxes are relative to this point
};

// Host-side build over the ABI primitives (wrappers and motion included in the bounds).
BvhBuild build_bvh(const RtPrimitive *prims, int n_prims, int max_leaf = 4, bool ordered = false); // max_leaf: primitives per leaf, 1..7; ordered: also the eight direction-ordered arrays
// The node array (sentinel included) with every inner node's children in the order of their boxes' distance from
// `origin`, nearest first — the order a fixed-order walk wants for rays that START there: all primary rays of a camera.
// Same nodes, boxes and leaves as b.nodes; only the order and the skip links differ (tools/sim/bvh_rotate.cpp: the
// `random` scene's primary rays touch 28.1 boxes and 4.6 leaf primitives instead of 32.8 and 6.3; bounce rays the same).
std::vector<BvhNode> order_bvh_for_origin(const BvhBuild &b, const double origin[3]);
// True bounds of one primitive incl. RotateY / Translate / motion.
void primitive_bounds(const RtPrimitive &p, double mn[3], double mx[3]);

} // namespace rtdev
