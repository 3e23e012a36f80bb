// walk_refill.hip — what would lanes that fetch a NEW ray when theirs is through buy the BVH walk?
//
// The BVH variants of the trace kernel are issue-bound at ~21 of 64 lanes per vector instruction: a wave's lanes each
// own one path and the walk runs until the wave's slowest ray is through.  tools/sim/bvh_sim.cpp MODELS
// a walk whose lanes are refilled from a ray queue (a wavefront tracer's trace kernel) at 1.4 - 1.6 x on bounce rays.
// This MEASURES it: the same skip-link walk (rt_bvh_slab.h's f32 culling boxes in LDS, f64 sphere tests, while-while)
// over the `random` scene's own bounce rays, once with 64 rays pinned to the 64 lanes of a wave until all are through
// (mode 0, the kernel today), once with every lane fetching the next ray of a global queue as soon as its own is
// through (mode 1), optionally testing leaves as soon as few lanes still descend (mode 2).
// Nothing here is product code; it answers whether a wavefront redesign of the BVH variants would pay.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -Iinclude -Iracer-tracer_amd -o build/walk_refill tools/microbench/walk_refill.hip \
//         racer-tracer_amd/build/product/rt_bvh.o -Lracer-tracer_amd/lib -lracer_tracer_amd -Wl,-rpath,$PWD/racer-tracer_amd/lib
#include <hip/hip_runtime.h>
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

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Ray { double o[3], d[3], time, pad; };                 // 64 B
struct Sphere { double c0[3], r2, dc[3], inv_dt_unused; };    // 64 B: centre at time t = c0 + t * dc (time interval [0, 1])
struct Hit { double t; int prim, pad; };

struct Args {
    const rtdev::BvhNode *nodes;
    const Sphere *spheres; // in leaf order
    const Ray *rays;
    Hit *hits;
    unsigned int *queue;
    unsigned long long *stats; // [4] wave-level counts: rounds, descent iterations, leaf iterations, prologues
    int n_nodes, n_rays, mode, straggle, fetch_min;
    int mode_arg;  // mode 4: helpers are also handed work every mode_arg descent steps (0: only between rounds)
    int own_range; // 1: every wave takes rays from its OWN contiguous share (a counter in a register, no atomic at all)
    double root_mn[3], root_mx[3], center[3];
};

__device__ __forceinline__ int lane_rank(uint64_t mask) {
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// 1/x like the trace kernel forms it (rt_trace_common.h: rcp_f64): hardware seed + one third-order step + fix-up
__device__ __forceinline__ double rcp_fast(double x) {
    const double r0 = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r0, 1.0);
    return __builtin_amdgcn_div_fixup(fma(r0, fma(e, e, e), r0), x, 1.0);
}

template <bool PREFETCH> __global__ __launch_bounds__(256, 4) void k_walk(const Args A) {
    extern __shared__ __align__(16) unsigned char lds[];
    rtdev::BvhNode *nodes = reinterpret_cast<rtdev::BvhNode *>(lds);
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(A.nodes);
        uint4 *dst = reinterpret_cast<uint4 *>(lds);
        for (int i = threadIdx.x; i < A.n_nodes * 2; i += 256) dst[i] = src[i];
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const int n = A.n_nodes;
    // per-lane ray state
    int ray = -1, i = n, best = -1;
    double ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0, time = 0, inv_a = 0, best_t = 0, t0 = 0;
    rtdev::SlabRay sr = {};
    float tmin_f = 0.f, best_f = 0.f;
    const float slack = 0x1p-20f;
    bool queue_dry = false;
    // mode 3: the prefetched next ray of this lane, and the request for indices that is in flight
    int next_index = -1;
    Ray next_ray = {};
    bool asked = false, want = false;
    unsigned asked_base = 0, asked_count = 0, want_rank = 0;
    unsigned n_rounds = 0, n_descents = 0, n_leaves = 0, n_starts = 0; // wave-uniform: what the SIMD issues, whatever the lanes
    // where the next rays come from: the global queue (one device-scope atomic per fetch, all waves on ONE address), or the
    // wave's own share of the ray array
    const unsigned n_waves = gridDim.x * (blockDim.x >> 6), wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const unsigned share = ((unsigned)A.n_rays + n_waves - 1) / n_waves;
    unsigned own_next = wave * share;
    const unsigned own_end = min(own_next + share, (unsigned)A.n_rays);
    auto take = [&](unsigned count) -> unsigned { // first index of `count` rays; indices >= the returned limit do not exist
        if (A.own_range) {
            const unsigned base = own_next;
            own_next += count;
            return base;
        }
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(A.queue, count);
        return base; // lane 0's value; read with readfirstlane where it is used
    };
    const unsigned limit = A.own_range ? own_end : (unsigned)A.n_rays;
    unsigned guard = 0; // every wave leaves after a bounded number of rounds whatever happens
    for (;;) {
        auto start = [&](int index, const Ray &R) { // closest_hit_bvh's prologue for a fresh ray
            ray = index;
            ox = R.o[0]; oy = R.o[1]; oz = R.o[2];
            dx = R.d[0]; dy = R.d[1]; dz = R.d[2];
            time = R.time;
            inv_a = rcp_fast(dx * dx + dy * dy + dz * dz);
            best_t = __builtin_inf();
            best = -1;
            i = 0;
            // clip to the root box in f64, then f32 around the root's centre
            const double ix = rcp_fast(dx), iy = rcp_fast(dy), iz = rcp_fast(dz);
            const double ax = (A.root_mn[0] - ox) * ix, bx = (A.root_mx[0] - ox) * ix;
            const double ay = (A.root_mn[1] - oy) * iy, by = (A.root_mx[1] - oy) * iy;
            const double az = (A.root_mn[2] - oz) * iz, bz = (A.root_mx[2] - oz) * iz;
            const double t_enter = fmax(fmax(fmin(ax, bx), fmin(ay, by)), fmin(az, bz));
            const double t_exit = fmin(fmin(fmax(ax, bx), fmax(ay, by)), fmax(az, bz));
            if (!(fmax(t_enter, 0.001) <= t_exit)) i = n; // misses the scene
            t0 = t_enter > 0.0 ? t_enter : 0.0;
            sr = rtdev::slab_ray((float)(fma(t0, dx, ox) - A.center[0]), (float)(fma(t0, dy, oy) - A.center[1]),
                                 (float)(fma(t0, dz, oz) - A.center[2]), ix, iy, iz);
            tmin_f = (float)(0.001 - t0) - fabsf((float)(0.001 - t0)) * slack - 0x1p-126f;
            best_f = __builtin_inff();
        };
        if constexpr (PREFETCH) {
            // PREFETCHED refill, what a real trace kernel would do: every lane keeps its NEXT ray loaded beside the one it
            // walks, so a lane whose ray is through goes on at once; the queue counter is asked one round ahead of the
            // loads and the loads a ray ahead of their use, so neither latency is waited for.
            n_starts += __ballot(ray < 0 && next_index >= 0) != 0;
            if (ray < 0 && next_index >= 0) { // 1. take the prefetched ray
                start(next_index, next_ray);
                next_index = -1;
            }
            if (asked) { // 2. the indices asked for LAST round have arrived: load those rays
                const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane((int)asked_base);
                if (base + asked_count >= limit) queue_dry = true;
                if (want) {
                    const unsigned mine = base + want_rank;
                    if (mine < limit) {
                        next_index = (int)mine;
                        next_ray = A.rays[mine];
                    }
                    want = false;
                }
                asked = false;
            }
            const uint64_t empty = __ballot(next_index < 0); // 3. ask for indices for the empty prefetch slots
            const int n_empty = __popcll(empty);
            if (!queue_dry && (n_empty >= A.fetch_min || n_empty == 64)) {
                asked_base = take((unsigned)n_empty);
                asked_count = (unsigned)n_empty;
                asked = true;
                if (next_index < 0) {
                    want = true;
                    want_rank = (unsigned)lane_rank(empty);
                }
            }
            if (__ballot(ray >= 0 || next_index >= 0) == 0 && !asked) break;
        } else {
        // ---- hand out rays
        const bool idle = ray < 0;
        const uint64_t idle_mask = __ballot(idle);
        const int n_idle = __popcll(idle_mask);
        // a fetch is a round trip to a global counter and a 64-byte load per lane: refilling lanes one by one makes the walk
        // wait for memory every round (measured: 2.4 x SLOWER than pinned rays), so lanes are refilled fetch_min at a time
        const bool fetch = !queue_dry && (A.mode == 0 ? n_idle == 64 : (n_idle >= A.fetch_min || n_idle == 64));
        if (fetch) {
            ++n_starts;
            const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane((int)take((unsigned)n_idle));
            if (base + (unsigned)n_idle >= limit) queue_dry = true;
            const unsigned mine = base + (unsigned)lane_rank(idle_mask);
            if (idle && mine < limit) start((int)mine, A.rays[mine]);
        }
        if (__ballot(ray >= 0) == 0) break; // queue dry and nothing in flight
        }
        if (++guard > (1u << 22)) break;
        ++n_rounds;
        // ---- descent: until every lane with a ray stands at a leaf or has left the tree (or few still descend)
        int count = 0, first = 0;
        for (;;) {
            const bool walking = ray >= 0 && i < n && count == 0;
            const int n_walking = __popcll(__ballot(walking));
            if (n_walking == 0) break;
            if (A.mode == 2 && n_walking <= A.straggle && __ballot(ray >= 0 && count > 0) != 0) break;
            ++n_descents;
            if (walking) {
                const uint4 *raw = reinterpret_cast<const uint4 *>(&nodes[i]);
                const uint4 q0 = raw[0], q1 = raw[1];
                const float lohi[6] = {__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z), __uint_as_float(q0.w), __uint_as_float(q1.x), __uint_as_float(q1.y)};
                const int skip = (int)q1.z, fc = (int)q1.w;
                const bool hit = rtdev::slab_hit(lohi, sr, tmin_f, best_f);
                i = hit ? i + 1 : skip;
                count = hit ? (fc & 7) : 0;
                first = fc >> 3;
            }
        }
        // ---- leaves
        for (int k = 0; __ballot(ray >= 0 && k < count) != 0; ++k) {
            ++n_leaves;
            if (ray >= 0 && k < count) {
                const Sphere S = A.spheres[first + k];
                const double cx = S.c0[0] + time * S.dc[0], cy = S.c0[1] + time * S.dc[1], cz = S.c0[2] + time * S.dc[2];
                const double px = ox - cx, py = oy - cy, pz = oz - cz;
                const double a = dx * dx + dy * dy + dz * dz;
                const double hb = px * dx + py * dy + pz * dz;
                const double c = px * px + py * py + pz * pz - S.r2;
                const double disc = hb * hb - a * c;
                if (disc >= 0.0) {
                    const double sq = sqrt(disc);
                    double root = (-hb - sq) * inv_a;
                    if (root < 0.001 || best_t < root) root = (-hb + sq) * inv_a;
                    if (!(root < 0.001 || best_t < root)) {
                        best_t = root;
                        best = first + k;
                        const float f = (float)(best_t - t0);
                        best_f = f + fabsf(f) * slack;
                    }
                }
            }
        }
        // ---- rays that have left the tree are done
        if (ray >= 0 && i >= n && count == 0) {
            A.hits[ray] = Hit{best_t, best, 0};
            ray = -1;
        }
        // (a lane that broke out of the descent at a leaf has count > 0 handled above and goes on descending next round)
    }
    if (lane == 0) {
        atomicAdd(&A.stats[0], (unsigned long long)n_rounds);
        atomicAdd(&A.stats[1], (unsigned long long)n_descents);
        atomicAdd(&A.stats[2], (unsigned long long)n_leaves);
        atomicAdd(&A.stats[3], (unsigned long long)n_starts);
    }
}


// ---- mode 4: 64 rays pinned to a wave, and lanes whose ray is through HELP the lanes that are not
// A skip-link walk is a walk over a RANGE of the node array: [i, e) with e = the array's end.  The node after node i's
// subtree is nodes[i].skip, so a lane at node i whose range goes on past that subtree can hand [skip, e) to an idle lane
// (a copy of its ray, its closest hit so far as the helper's culling bound) and keep [i, skip) — whole subtrees on both
// sides, no stack, and the closest hit is the minimum over everybody who walked a part of the ray's range (merged
// through LDS once the wave is through).  The same hits as the undivided walk: every primitive is tested by the same
// code against the same ray; only which lane does it changes.
template <typename T> __device__ __forceinline__ T take_from(T v, int src) {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "");
    if constexpr (sizeof(T) == 4) {
        return __builtin_bit_cast(T, __shfl(__builtin_bit_cast(int, v), src, 64));
    } else {
        const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
        const unsigned lo = (unsigned)__shfl((int)(unsigned)u, src, 64), hi = (unsigned)__shfl((int)(unsigned)(u >> 32), src, 64);
        return __builtin_bit_cast(T, ((unsigned long long)hi << 32) | lo);
    }
}

__global__ __launch_bounds__(256, 4) void k_walk_donate(const Args A) {
    extern __shared__ __align__(16) unsigned char lds[];
    rtdev::BvhNode *nodes = reinterpret_cast<rtdev::BvhNode *>(lds);
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(A.nodes);
        uint4 *dst = reinterpret_cast<uint4 *>(lds);
        for (int i = threadIdx.x; i < A.n_nodes * 2; i += 256) dst[i] = src[i];
    }
    // per wave behind the nodes: the rays' merged closest hits and the table that pairs idle lanes with donors
    struct Merge { unsigned long long t_bits[64]; unsigned prim[64]; int donor[64]; };
    Merge &M = reinterpret_cast<Merge *>(lds + (size_t)A.n_nodes * sizeof(rtdev::BvhNode))[threadIdx.x >> 6];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int n = A.n_nodes;
    const float slack = 0x1p-20f;
    const unsigned n_waves = gridDim.x * (blockDim.x >> 6), wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const unsigned share = ((unsigned)A.n_rays + n_waves - 1) / n_waves;
    unsigned own_next = wave * share;
    const unsigned own_end = min(own_next + share, (unsigned)A.n_rays);
    unsigned n_rounds = 0, n_descents = 0, n_leaves = 0, n_gifts = 0;
    for (; own_next < own_end; own_next += 64) {
        const unsigned mine = own_next + (unsigned)lane;
        const bool have = mine < own_end;
        const Ray R = A.rays[have ? mine : own_next];
        double ox = R.o[0], oy = R.o[1], oz = R.o[2], dx = R.d[0], dy = R.d[1], dz = R.d[2], time = R.time;
        double inv_a = rcp_fast(dx * dx + dy * dy + dz * dz);
        double best_t = __builtin_inf();
        int best = -1, i = 0, e = n, owner = lane;
        const double ix = rcp_fast(dx), iy = rcp_fast(dy), iz = rcp_fast(dz);
        const double ax = (A.root_mn[0] - ox) * ix, bx = (A.root_mx[0] - ox) * ix;
        const double ay = (A.root_mn[1] - oy) * iy, by = (A.root_mx[1] - oy) * iy;
        const double az = (A.root_mn[2] - oz) * iz, bz = (A.root_mx[2] - oz) * iz;
        const double t_enter = fmax(fmax(fmin(ax, bx), fmin(ay, by)), fmin(az, bz));
        const double t_exit = fmin(fmin(fmax(ax, bx), fmax(ay, by)), fmax(az, bz));
        if (!have || !(fmax(t_enter, 0.001) <= t_exit)) i = n; // no ray, or it misses the scene
        double t0 = t_enter > 0.0 ? t_enter : 0.0;
        rtdev::SlabRay sr = rtdev::slab_ray((float)(fma(t0, dx, ox) - A.center[0]), (float)(fma(t0, dy, oy) - A.center[1]),
                                            (float)(fma(t0, dz, oz) - A.center[2]), ix, iy, iz);
        float tmin_f = (float)(0.001 - t0) - fabsf((float)(0.001 - t0)) * slack - 0x1p-126f;
        float best_f = __builtin_inff();
        M.t_bits[lane] = 0x7ff0000000000000ull; // +inf
        M.prim[lane] = 0xffffffffu;
        // A lane's (best_t, best) is a hit of the ray of `owner` if best >= 0.  t > 0, so its bits order like the value.
        auto hand_in = [&](bool who) {
            const bool found = who && best >= 0;
            if (found) atomicMin(&M.t_bits[owner], (unsigned long long)__double_as_longlong(best_t));
            // whoever holds the minimum names its primitive (an earlier, farther hit's name is overwritten)
            if (found && M.t_bits[owner] == (unsigned long long)__double_as_longlong(best_t)) M.prim[owner] = (unsigned)best;
        };
        // ---- lanes without work take a part of somebody's range (callable wherever no lane stands at a leaf it has
        // not tested yet... or does: a lane with count > 0 neither gives nor takes)
        auto donate = [&](int count_now) {
            const uint64_t has_range = __ballot(i < e);
            const uint64_t free_lanes = __ballot(!(i < e) && count_now == 0);
            const int n_idle = __popcll(free_lanes);
            if (n_idle < A.fetch_min) return;
            int m = n;
            if (i < e && count_now == 0) m = (int)reinterpret_cast<const uint4 *>(&nodes[i])[1].z; // the node after node i's subtree
            const bool can = i < e && count_now == 0 && m < e && e - m >= A.straggle;
            const uint64_t donors = __ballot(can);
            if (donors == 0) return;
            (void)has_range;
            const int n_don = __popcll(donors);
            if (can) M.donor[lane_rank(donors)] = lane;
            const bool is_free = !(i < e) && count_now == 0;
            const int r = lane_rank(free_lanes);
            const bool take = is_free && r < n_don;
            const bool gives = can && lane_rank(donors) < n_idle;
            const int src = take ? M.donor[r] : lane;
            const double g_ox = take_from(ox, src), g_oy = take_from(oy, src), g_oz = take_from(oz, src);
            const double g_dx = take_from(dx, src), g_dy = take_from(dy, src), g_dz = take_from(dz, src);
            const double g_time = take_from(time, src), g_inv_a = take_from(inv_a, src), g_best_t = take_from(best_t, src);
            const double g_t0 = take_from(t0, src);
            const float g_ivx = take_from(sr.ivx, src), g_ivy = take_from(sr.ivy, src), g_ivz = take_from(sr.ivz, src);
            const float g_oix = take_from(sr.oix, src), g_oiy = take_from(sr.oiy, src), g_oiz = take_from(sr.oiz, src);
            const float g_tmin = take_from(tmin_f, src), g_bestf = take_from(best_f, src);
            const int g_e = take_from(e, src), g_m = take_from(m, src), g_owner = take_from(owner, src);
            hand_in(take); // what a helper found for the ray it worked on before goes to that ray's owner first
            if (take) {
                best = -1;
                ox = g_ox; oy = g_oy; oz = g_oz; dx = g_dx; dy = g_dy; dz = g_dz;
                time = g_time; inv_a = g_inv_a; best_t = g_best_t; t0 = g_t0;
                sr.ivx = g_ivx; sr.ivy = g_ivy; sr.ivz = g_ivz; sr.oix = g_oix; sr.oiy = g_oiy; sr.oiz = g_oiz;
                tmin_f = g_tmin; best_f = g_bestf;
                i = g_m; e = g_e; owner = g_owner;
            }
            if (gives) e = m;
            n_gifts += (unsigned)min(n_don, n_idle);
        };
        for (unsigned guard = 0; guard < (1u << 20); ++guard) {
            const uint64_t busy = __ballot(i < e);
            if (busy == 0) break;
            ++n_rounds;
            donate(0);
            // ---- descent
            int count = 0, first = 0;
            unsigned step = 0;
            for (;;) {
                const bool walking = i < e && count == 0;
                if (__ballot(walking) == 0) break;
                ++n_descents;
                if (walking) {
                    const uint4 *raw = reinterpret_cast<const uint4 *>(&nodes[i]);
                    const uint4 q0 = raw[0], q1 = raw[1];
                    const float lohi[6] = {__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z), __uint_as_float(q0.w), __uint_as_float(q1.x), __uint_as_float(q1.y)};
                    const int skip = (int)q1.z, fc = (int)q1.w;
                    const bool hit = rtdev::slab_hit(lohi, sr, tmin_f, best_f);
                    i = hit ? i + 1 : skip;
                    count = hit ? (fc & 7) : 0;
                    first = fc >> 3;
                }
                if (A.mode_arg > 0 && (++step % (unsigned)A.mode_arg) == 0) donate(count);
            }
            // ---- leaves
            for (int k = 0; __ballot(k < count) != 0; ++k) {
                ++n_leaves;
                if (k < count) {
                    const Sphere S = A.spheres[first + k];
                    const double cx = S.c0[0] + time * S.dc[0], cy = S.c0[1] + time * S.dc[1], cz = S.c0[2] + time * S.dc[2];
                    const double px = ox - cx, py = oy - cy, pz = oz - cz;
                    const double a = dx * dx + dy * dy + dz * dz;
                    const double hb = px * dx + py * dy + pz * dz;
                    const double c = px * px + py * py + pz * pz - S.r2;
                    const double disc = hb * hb - a * c;
                    if (disc >= 0.0) {
                        const double sq = sqrt(disc);
                        double root = (-hb - sq) * inv_a;
                        if (root < 0.001 || best_t < root) root = (-hb + sq) * inv_a;
                        if (!(root < 0.001 || best_t < root)) {
                            best_t = root;
                            best = first + k;
                            const float f = (float)(best_t - t0);
                            best_f = f + fabsf(f) * slack;
                        }
                    }
                }
            }
        }
        // ---- merge: the closest hit of a ray is the minimum over the lanes that walked parts of its range
        hand_in(true);
        if (have) {
            const unsigned long long tb = M.t_bits[lane];
            A.hits[mine] = Hit{__longlong_as_double((long long)tb), (int)M.prim[lane], 0};
        }
    }
    if (lane == 0) {
        atomicAdd(&A.stats[0], (unsigned long long)n_rounds);
        atomicAdd(&A.stats[1], (unsigned long long)n_descents);
        atomicAdd(&A.stats[2], (unsigned long long)n_leaves);
        atomicAdd(&A.stats[3], (unsigned long long)n_gifts);
    }
}

struct HostScene {
    std::vector<Sphere> spheres; // leaf order
    rtdev::BvhBuild bvh;
};

static bool hit_sphere(const Sphere &s, const Ray &r, double tmax, double &t) {
    const double c[3] = {s.c0[0] + r.time * s.dc[0], s.c0[1] + r.time * s.dc[1], s.c0[2] + r.time * s.dc[2]};
    const double p[3] = {r.o[0] - c[0], r.o[1] - c[1], r.o[2] - c[2]};
    const double a = r.d[0] * r.d[0] + r.d[1] * r.d[1] + r.d[2] * r.d[2];
    const double hb = p[0] * r.d[0] + p[1] * r.d[1] + p[2] * r.d[2];
    const double cc = p[0] * p[0] + p[1] * p[1] + p[2] * p[2] - s.r2;
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

int main(int argc, char **argv) {
    RthSession *session = nullptr;
    if (rth_session_open("scenes/config_c2.yml", "random", nullptr, 1, &session) != RT_OK) {
        printf("%s\n", rth_last_error_message());
        return 1;
    }
    const RtSceneDesc *d = rth_session_scene(session);
    const RtCamera *cam = rth_session_camera(session);
    HostScene hs;
    hs.bvh = rtdev::build_bvh(d->primitives, d->n_primitives, 3);
    for (int pi : hs.bvh.prim_index) {
        const RtPrimitive &p = d->primitives[pi];
        Sphere s = {};
        for (int k = 0; k < 3; ++k) {
            s.c0[k] = p.p[k];
            s.dc[k] = p.kind == RT_PRIM_MOVING_SPHERE ? p.center_b[k] - p.p[k] : 0.0;
        }
        s.r2 = p.p[3] * p.p[3];
        hs.spheres.push_back(s);
    }
    // rays: primary rays of the scene's camera, then one diffuse bounce off whatever they hit (like bvh_sim)
    const int W = 960, H = 540;
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    std::vector<Ray> primary, bounce;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            Ray r = {};
            const double u = (x + U(rng)) / (W - 1), v = (y + U(rng)) / (H - 1);
            for (int k = 0; k < 3; ++k) {
                r.o[k] = cam->origin[k];
                r.d[k] = cam->upper_left_corner[k] + u * cam->horizontal[k] - v * cam->vertical[k] - cam->origin[k];
            }
            r.time = U(rng);
            primary.push_back(r);
        }
    for (const Ray &r : primary) {
        double bt = INFINITY;
        int best = -1;
        for (size_t i = 0; i < hs.spheres.size(); ++i) {
            double t;
            if (hit_sphere(hs.spheres[i], r, bt, t)) { bt = t; best = (int)i; }
        }
        if (best < 0) continue;
        const Sphere &s = hs.spheres[(size_t)best];
        Ray b = {};
        double nrm[3], v[3], len;
        for (int k = 0; k < 3; ++k) {
            b.o[k] = r.o[k] + bt * r.d[k];
            nrm[k] = (b.o[k] - (s.c0[k] + r.time * s.dc[k])) / std::sqrt(s.r2);
        }
        do {
            len = 0;
            for (int k = 0; k < 3; ++k) { v[k] = 2 * U(rng) - 1; len += v[k] * v[k]; }
        } while (len >= 1 || len == 0);
        for (int k = 0; k < 3; ++k) b.d[k] = nrm[k] + v[k] / std::sqrt(len);
        b.time = r.time;
        bounce.push_back(b);
    }
    printf("random scene: %d spheres, %zu nodes; %zu primary rays, %zu bounce rays\n", d->n_primitives, hs.bvh.nodes.size(), primary.size(), bounce.size());

    Args a = {};
    rtdev::BvhNode *d_nodes;
    Sphere *d_spheres;
    CK(hipMalloc((void **)&d_nodes, hs.bvh.nodes.size() * sizeof(rtdev::BvhNode)));
    CK(hipMemcpy(d_nodes, hs.bvh.nodes.data(), hs.bvh.nodes.size() * sizeof(rtdev::BvhNode), hipMemcpyHostToDevice));
    CK(hipMalloc((void **)&d_spheres, hs.spheres.size() * sizeof(Sphere)));
    CK(hipMemcpy(d_spheres, hs.spheres.data(), hs.spheres.size() * sizeof(Sphere), hipMemcpyHostToDevice));
    CK(hipMalloc((void **)&a.queue, sizeof(unsigned)));
    CK(hipMalloc((void **)&a.stats, 4 * sizeof(unsigned long long)));
    a.nodes = d_nodes;
    a.spheres = d_spheres;
    a.n_nodes = (int)hs.bvh.nodes.size();
    for (int k = 0; k < 3; ++k) { a.root_mn[k] = hs.bvh.root_mn[k]; a.root_mx[k] = hs.bvh.root_mx[k]; a.center[k] = hs.bvh.center[k]; }
    const size_t lds = hs.bvh.nodes.size() * sizeof(rtdev::BvhNode);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int set = 0; set < 2; ++set) {
        std::vector<Ray> rays = set == 0 ? primary : bounce;
        const size_t base_n = rays.size();
        while (rays.size() < 4000000) rays.insert(rays.end(), rays.begin(), rays.begin() + (long)base_n); // a few million rays
        Ray *d_rays;
        Hit *d_hits;
        CK(hipMalloc((void **)&d_rays, rays.size() * sizeof(Ray)));
        CK(hipMemcpy(d_rays, rays.data(), rays.size() * sizeof(Ray), hipMemcpyHostToDevice));
        CK(hipMalloc((void **)&d_hits, rays.size() * sizeof(Hit)));
        a.rays = d_rays;
        a.hits = d_hits;
        a.n_rays = (int)rays.size();
        std::vector<Hit> ref;
        struct Case { int mode, straggle, fetch_min, own; const char *name; int arg = 0; };
        const Case cases[] = {{0, 0, 64, 0, "64 rays pinned to a wave's lanes (today)"},
                              {1, 0, 1, 0, "lanes refilled one by one"},
                              {1, 0, 8, 0, "refilled once 8 lanes are idle"},
                              {1, 0, 16, 0, "refilled once 16 lanes are idle"},
                              {1, 0, 32, 0, "refilled once 32 lanes are idle"},
                              {2, 8, 16, 0, "16 idle; leaves once <= 8 lanes descend"},
                              {3, 0, 1, 0, "PREFETCHED next ray per lane, one by one"},
                              {3, 0, 16, 0, "prefetched, asked once 16 slots are empty"},
                              {0, 0, 64, 1, "OWN SHARE per wave, no atomic: pinned"},
                              {1, 0, 1, 1, "own share: refilled one by one"},
                              {1, 0, 8, 1, "own share: refilled once 8 lanes are idle"},
                              {1, 0, 16, 1, "own share: refilled once 16 lanes are idle"},
                              {1, 0, 32, 1, "own share: refilled once 32 lanes are idle"},
                              {2, 8, 16, 1, "own share: 16 idle; leaves once <= 8 descend"},
                              {3, 0, 1, 1, "own share: prefetched, one by one"},
                              {3, 0, 16, 1, "own share: prefetched, 16 slots"},
                              {4, 8, 65, 1, "the helpers' kernel with no lane ever helping"},
                              {4, 8, 8, 1, "own share: pinned + HELPERS at 8 idle, parts >= 8 nodes"},
                              {4, 8, 16, 1, "pinned + helpers at 16 idle, parts >= 8 nodes"},
                              {4, 16, 16, 1, "pinned + helpers at 16 idle, parts >= 16 nodes"},
                              {4, 4, 24, 1, "pinned + helpers at 24 idle, parts >= 4 nodes"},
                              {4, 16, 32, 1, "pinned + helpers at 32 idle, parts >= 16 nodes"},
                              {4, 32, 16, 1, "pinned + helpers at 16 idle, parts >= 32 nodes"},
                              {4, 8, 16, 1, "helpers at 16 idle, also every 4 descent steps", 4},
                              {4, 8, 16, 1, "helpers at 16 idle, also every 2 descent steps", 2},
                              {4, 8, 8, 1, "helpers at 8 idle, also every 4 descent steps", 4},
                              {4, 4, 8, 1, "helpers at 8 idle, parts >= 4, every 2 steps", 2},
                              {4, 8, 24, 1, "helpers at 24 idle, also every 4 descent steps", 4},
                              {4, 8, 32, 1, "helpers at 32 idle, also every 3 descent steps", 3}};
        for (const Case &c : cases) {
            a.mode = c.mode;
            a.straggle = c.straggle;
            a.fetch_min = c.fetch_min;
            a.own_range = c.own;
            a.mode_arg = c.arg;
            float best_ms = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemset(a.queue, 0, sizeof(unsigned)));
                CK(hipMemset(a.stats, 0, 4 * sizeof(unsigned long long)));
                CK(hipMemset(d_hits, 0xff, rays.size() * sizeof(Hit)));
                CK(hipEventRecord(e0));
                if (a.mode == 4) hipLaunchKernelGGL(k_walk_donate, dim3(256 * 4), dim3(256), lds + 4 * (64 * 8 + 64 * 4 + 64 * 4), 0, a);
                else if (a.mode == 3) hipLaunchKernelGGL(k_walk<true>, dim3(256 * 4), dim3(256), lds, 0, a);
                else hipLaunchKernelGGL(k_walk<false>, dim3(256 * 4), dim3(256), lds, 0, a);
                CK(hipGetLastError());
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                best_ms = std::min(best_ms, ms);
            }
            std::vector<Hit> got(rays.size());
            CK(hipMemcpy(got.data(), d_hits, rays.size() * sizeof(Hit), hipMemcpyDeviceToHost));
            long differ = 0;
            if (c.mode == 0 && !c.own) ref = got;
            else for (size_t k = 0; k < got.size(); ++k) {
                const bool bad = got[k].prim != ref[k].prim || got[k].t != ref[k].t;
                if (bad && differ < 4) printf("  ray %zu: prim %d t %.17g, pinned walk prim %d t %.17g\n", k, got[k].prim, got[k].t, ref[k].prim, ref[k].t);
                differ += bad;
            }
            if (differ) printf("  %ld of %zu rays differ\n", differ, got.size());
            printf("%-8s rays | %-44s %7.2f ms  %6.2f G rays/s%s\n", set == 0 ? "primary" : "bounce", c.name, best_ms,
                   rays.size() / best_ms / 1e6, (c.mode == 0 && !c.own) ? "" : (differ ? "  RESULTS DIFFER" : "  same hits"));
            unsigned long long st[4];
            CK(hipMemcpy(st, a.stats, sizeof st, hipMemcpyDeviceToHost));
            const double per = 64.0 / (double)rays.size(); // wave-level issues per 64 rays: 1 wave with no idle lane walks 64 rays
            printf("           per 64 rays: %6.1f rounds, %6.1f descent steps, %6.1f leaf tests, %5.2f prologues issued\n", st[0] * per,
                   st[1] * per, st[2] * per, st[3] * per);
        }
        CK(hipFree(d_rays));
        CK(hipFree(d_hits));
    }
    rth_session_close(session);
    return 0;
}
