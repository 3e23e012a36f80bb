// rt_trace_pool_kernel.hip — the default trace kernel (f64): persistent waves
// over a queue of (8x8 pixel tile, sample chunk) items, lanes drawing paths
// from the item's pool, wave-cooperative rejection sampling.
//
// Replaces CpuRenderer::raytrace's pixel x sample loop
// (racer-tracer/src/renderer/cpu.rs:26-71) and the recursive ray_color
// (renderer.rs:41-90).  The ideas, all measured against the v1 kernel
// (rt_trace_kernel.hip) on cornell_box 1080p:
//
// 1. PERSISTENT WAVES + ITEM QUEUE.  The grid is sized to what is resident
//    (CUs x blocks per CU); every wave pulls items from one atomic counter
//    until the queue is dry.  An item is a tile x a chunk of its samples; the
//    chunk plan (rt_api.hip: chunk_plan, a function of spp only) has long chunks
//    first and a taper of short ones last, so a launch ends on small items.
//
// 2. LANES ARE NOT PIXELS.  Inside an item the 64 lanes share a pool of
//    64 x chunk (pixel, sample) paths: whenever lanes have no path in flight a
//    ballot + prefix count hands them the next pool entries.  No lane waits for
//    the slowest pixel of its tile; ray state (origin, direction, throughput)
//    lives in VGPRs and never touches HBM.  Finished samples are added to the
//    tile's 64 pixel sums in LDS (ds_add_f64); the item's sums go to its own
//    slice partial[chunk][pixel], and k_resolve adds slices in index order, so
//    the frame is deterministic and independent of scheduling.
//
// 3. COOPERATIVE REJECTION SAMPLING.  random_in_unit_sphere (vec3.rs:424-430)
//    loops until a candidate falls inside the sphere; per lane that is 1.9
//    iterations on average but a wave pays for its slowest lane (~6.4).
//    Because every draw is ADDRESSED (include/rt_rng.h) — candidate i of a
//    path is a pure function of (pixel, sample, segment, i): ONE Philox block,
//    three 42-bit coordinates — any lane can evaluate any lane's candidate.  Each round the lanes that still need a
//    sample post a request in LDS, the 64 lanes split into groups that test
//    consecutive candidates of one request each, and a ballot picks the first
//    accepted candidate in stream order.  ~3 rounds instead of ~6.4, identical
//    values (the accepted candidate is exactly the one the sequential loop
//    would have stopped at).
//
// 4. REGENERATION BATCHES.  Pool entries leave in index order, so the camera samples
//    (vertical jitter, ray time, lens disk) of 64 consecutive entries are drawn by the
//    whole wave into LDS ahead of the hand-out instead of by the ~16 lanes that start a
//    path in a given iteration.
//
// 5. UNIFORMS ARE RE-READ WHERE THEY ARE USED (kernargs_here): camera and background would
//    otherwise sit in SGPRs across the path loop and be spilled to VGPR lanes.
//
// 6. NO SCALAR SWITCH IN THE CLOSEST-HIT LOOP.  The device table is grouped by kind (plain XY / XZ / YZ
//    rects, plain spheres, the rest); each group gets one straight-line test with the plane a compile-time
//    constant.  With a wave-uniform index the `match` on a record's kind was scalar control flow, ~25 SALU
//    instructions and half a dozen branches around 12 vector instructions per rect.
//
// 7. THE ONE EXPENSIVE TEXTURE IS EVALUATED BY THE WHOLE WAVE (coop_noise_turbulence): lanes only note
//    their Noise lookup while shading; eight lookups per round take eight lanes each, one octave per lane.
//
// 8. A PATH CARRIES ORIGIN, DIRECTION, THROUGHPUT AND NOTHING ELSE WHILE IT WAITS.  A hit's point overwrites the ray
//    origin, its attenuation is multiplied into the throughput at once, a Lambertian hit's normal overwrites the ray
//    direction (the incoming one is dead), and a finished path's contribution is formed in the throughput.  Every
//    value kept beside those cost registers and, where the material arms meet, a copy per arm.
//
// 9. TWO ITEMS IN FLIGHT (the variants with any primitive kind or a BVH).  The lanes an item's last paths no longer need
//    start the NEXT item's paths; per-pixel sums are 64-bit fixed-point integers, so the order in which samples arrive —
//    which now depends on what else the wave traces — cannot change a bit of the frame (SUMS AND OVERLAPPED ITEMS below).
//
// The launch is VALU-throughput bound with the CU's one scalar ALU close behind (scalar_busy 0.60 on C3: count scalar
// instructions before vector ones; DESIGN.md 4.2 / 6 and LABNOTES.md have the counters, the per-region cycle profile of the
// -DRT_PROFILE_REGIONS build — which adds 1.8 KB of LDS per block and can cost a variant at a granule edge a block per CU —
// and the variants that were measured and dropped).
#include <cstddef>
#include <type_traits>
#include "rt_trace_common.h"

#ifndef RT_OCC_TEX
#define RT_OCC_TEX 4 // textured spheres-only / rects-only variants: 40 KB of LDS per block still fits four (C4 +3 %)
#endif
#ifndef RT_OCC_TEX_BVH
#define RT_OCC_TEX_BVH 4 // textured BVH variants: 32-byte nodes and no Perlin table in static LDS leave room for four blocks (random scene +20 %)
#endif
#ifndef RT_OCC_TEX_ANY
#define RT_OCC_TEX_ANY 4 // textured linear-loop variants with any primitive kind (emissive.yml 24.5 -> 22.8 ms against three)
#endif
#ifndef RT_OCC_SPEC
#define RT_OCC_SPEC 5
#endif
#ifndef RT_OCC_ANY
#define RT_OCC_ANY 5 // untextured linear-loop variants with any primitive kind (cornell_box_boxes: 4 -> 5 waves per SIMD 20.7 -> 19.3 ms)
#endif
#ifndef RT_OCC_PLAIN
#define RT_OCC_PLAIN 7 // rects-only / spheres-only, no textures, no specular materials: 72 VGPRs, seven blocks per CU when LDS allows (C3 73.4 ->
                       // 72.0 ms against six).  Six needed 80 of the 79 the allocator wanted until the path state was re-set between
                       // items (its live ranges then end at the path loop's exit: 73); the bound keeps the allocator there.
#endif
namespace RT_KNS {

// -DRT_PROFILE_REGIONS: developer build that accumulates the shader-clock cycles each wave
// spends per region of the path loop into A.segments[RT_STAT_REGIONS..]; rt_scene_last_stats prints them
// (tools/region_profile.sh).  Not part of the product build.
#ifdef RT_PROFILE_REGIONS
// rt_t_[0..15] region cycles, [16] the previous marker, [17..34] the two lane histograms, [35..36] noise counts
#define RT_REGION_DECL                                                          \
    __shared__ unsigned long long rt_t_all_[4][56];                             \
    unsigned long long *rt_t_ = rt_t_all_[threadIdx.x >> 6];                    \
    if ((threadIdx.x & 63) < 56) rt_t_[threadIdx.x & 63] = 0;                   \
    if ((threadIdx.x & 63) == 16) rt_t_[16] = __builtin_readcyclecounter();   \
    const unsigned long long rt_wave_start_ = wall_clock64();
// usable inside divergent code: the first ACTIVE lane books the time since the previous marker
#define RT_REGION(k)                                                            \
    do {                                                                        \
        const unsigned long long act_ = ballot(1);                            \
        if (lane_rank(act_) == 0) {                                             \
            const unsigned long long now_ = __builtin_readcyclecounter();       \
            rt_t_[k] += now_ - rt_t_[16];                                       \
            rt_t_[40 + (k)] += (now_ - rt_t_[16]) * (unsigned long long)__popcll(act_); \
            rt_t_[16] = now_;                                                   \
        }                                                                       \
    } while (0)
/* lanes tracing this iteration (wave-uniform n), booked while the pool has entries / in the item's tail */ \
#define RT_LANES(n, tail)                                                       \
    do {                                                                        \
        const int n_ = (n); /* the ballot needs every lane */                   \
        if (lane == 0) rt_t_[17 + ((tail) ? 9 : 0) + (n_ + 7) / 8] += 1;        \
    } while (0)
#define RT_REGION_FLUSH                                                         \
    if (lane < 16) atomicAdd(A.segments + RT_STAT_REGIONS + lane, rt_t_[lane]); \
    if (lane < 18) atomicAdd(A.segments + RT_STAT_LANES_BODY + lane, rt_t_[17 + lane]); \
    if (lane < 4) atomicAdd(A.segments + RT_STAT_NOISE + lane, rt_t_[35 + lane]); \
    if (lane < 16) atomicAdd(A.segments + RT_STAT_REGION_LANES + lane, rt_t_[40 + lane]); \
    if (lane == 0) { /* wall clock (100 MHz) of the first/last wave start and end */ \
        const unsigned long long end_ = wall_clock64();                         \
        atomicMin(A.segments + RT_STAT_WALL + 0, rt_wave_start_);               \
        atomicMax(A.segments + RT_STAT_WALL + 1, rt_wave_start_);               \
        atomicMin(A.segments + RT_STAT_WALL + 2, end_);                         \
        atomicMax(A.segments + RT_STAT_WALL + 3, end_);                         \
    }
#else
#define RT_REGION_DECL
#ifdef RT_ISA_MARKERS // `make isa`: the region boundaries as comments in the assembly listing (tools/isa_regions.py)
#define RT_REGION(k) asm volatile("; ==== end of region " #k);
#else
#define RT_REGION(k)
#endif
#define RT_LANES(n, tail)
#define RT_REGION_FLUSH
#endif

__device__ __forceinline__ int lane_rank(uint64_t mask) { // set bits of `mask` below this lane
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Request slots of the wave-cooperative Perlin turbulence (coop_noise_turbulence): eight lookups per
// round, eight lanes each.
struct NoiseSlots {
    double p[8][3];    // hit point of the lookup
    int depth[8];      // its texture's octave count (noise.rs:28)
    int perlin[8];     // ... and gradient table
    double term[8][8]; // 0.5^o * noise(2^o * p) of the round's octaves, summed in order by the requester
};
// A request of the cooperative samplers: {pixel, sample, segment, next candidate}
struct alignas(16) Req4 { // one 128-bit LDS access
    uint32_t x, y, z, w;
};
// The sampler's request slots and the turbulence slots are never live at the same time (the Noise rounds of
// an iteration end before its sampler rounds begin, and both finish with what they posted), so they share
// their LDS: the textured variants — the BVH ones above all, whose node array already fills LDS to
// three blocks per CU — cost no more LDS than the plain ones.
// (A round of the samplers that has more than 32 requests open gives every lane its own candidate and touches no slot, so 32
// slots are all they use; variants without textures have no Noise lookups to make room for.)
template <bool TEXTURED> union SamplerScratch {
    Req4 req[32];
    NoiseSlots noise;
};
template <> union SamplerScratch<false> {
    Req4 req[32];
};

// What the END of an item needs of it (finish_item), parked while the item's last paths are in flight and the wave already
// hands out the next item's.
struct alignas(16) ItemInfo {
    int chunk, tx, ty, tile_py0, region, reg_tiles, rows_aligned, _pad;
};

// Per-wave scratch in LDS.  HAS_TIME: the scene has MovingSpheres (PRIMS_ANY variants).
// NBUF: batches of camera samples kept (2, or 1 for the BVH variants, whose node array wants the LDS).
// The lens-disk samples of a batch (16 B per entry) are NOT in here: only a camera with an aperture draws them, and
// without their 8 KB per block the plain variants fit six blocks per CU instead of five (cornell 85.2 -> 82.1 ms).
// They live behind everything else in DYNAMIC LDS, which the launch sizes by the camera (TraceArgs.lens_lds).
// OVERLAP: the variant keeps two items in flight (see SUMS AND OVERLAPPED ITEMS in the kernel): two sets of sums, and — to pay
// for them — the per-pixel `base` vector is replaced by the jitter it is made of.
template <bool TEXTURED, int NBUF, bool OVERLAP> struct WaveLds {
    // per pixel of the CURRENT item's tile (the one entries are handed out of): upper_left_corner + u * horizontal with the
    // pixel's ONE horizontal jitter u = (px + ju) / (W - 1) (cpu.rs:35-36, camera.rs:331) — or, OVERLAP, just u: the
    // hand-out then forms the vector (three fma and a read of `ulc` per iteration for 1 KB)
    double base[OVERLAP ? 1 : 64][3];
    double u[OVERLAP ? 64 : 1];
    // per-pixel radiance sums: doubles — or, OVERLAP, 64-bit fixed-point integers (TraceArgs.sum_scale) of the (up to) two
    // items in flight: slot s, pixel p at [s * 64 + p]
    double sum[OVERLAP ? 128 : 64][3];
    SamplerScratch<TEXTURED> scratch; // cooperative sampler requests / Noise lookups
    uint8_t pix_of[64]; // pool slot -> lane-order pixel index, for tiles cut by the image edge (current item)
    // Camera samples of the pool entries, drawn 64 entries at a time by the WHOLE wave
    // (prepare_batch below): entry w sits in slot w & 63 of buffer (w >> 6) & (NBUF - 1).
    double v[NBUF][64];        // (py + jv) / (H - 1)                      cpu.rs:39-40
    ItemInfo info[2];
    // camera.upper_left_corner, for the hand-out: v_fma_f64 reads ONE scalar operand, so of upper_left_corner + u * horizontal
    // one vector has to come from vector registers — copied there with six v_mov_b32 per iteration, or read from here
    double ulc[3];
};

// vec3.rs:424-430 for every lane with `need`, evaluated by the whole wave.
// Must be called by all 64 lanes (wave-uniform control flow).  Runs at most
// `max_rounds` rounds: a lane whose request is still open afterwards returns
// false and keeps `base` (its first untested candidate), so the search resumes
// at the same stream position in the next call.
__device__ __forceinline__ bool coop_random_in_unit_sphere(bool need, uint32_t pixel, uint32_t sample, uint32_t seg,
                                                           uint32_t &base, uint32_t k0, uint32_t k1, int lane,
                                                           Req4 *req, int max_rounds, d3 &result) {
    bool have = false;
    uint64_t pending = ballot(need);
    for (int round = 0; round < max_rounds && pending != 0; ++round) {
        const int n = __popcll(pending);
        // a round costs the whole wave ~70 instructions; past the first it only runs while enough requests are
        // open to be worth that (the others resume next iteration): C2 +2.2 %, C3 +0.3 % (thresholds 2..16 and one to
        // six rounds measured again with one-block candidates: 8 and two rounds stay)
        if (round > 0 && n < 8) break;
        // group size q = 2^lg, the largest power of two with n * q <= 64
        const int lg = n > 32 ? 0 : (n > 16 ? 1 : (n > 8 ? 2 : (n > 4 ? 3 : (n > 2 ? 4 : (n > 1 ? 5 : 6)))));
        if (lg == 0) { // more than 32 requests: one candidate each, so every lane tests its OWN (no LDS, no shuffle)
            if (need) {
                // `result` only means something to a lane that returns true: a lane that still needs a sample
                // may take every candidate it looks at, rejected ones included, without a select
                result = sphere_candidate(philox4x32(pixel, sample, (seg << 8) | RT_RNG_SCATTER, base, k0, k1));
                if (len2(result) < 1.0) {
                    need = false;
                    have = true;
                } else {
                    base += 1u;
                }
            }
            pending = ballot(need);
            continue;
        }
        const int rank = lane_rank(pending);
        if (need) req[rank] = Req4{pixel, sample, seg, base};
        const int j = lane >> lg;              // request served by this lane
        const int c = lane & ((1 << lg) - 1);  // candidate offset inside the group
        const bool serving = j < n;
        const Req4 r = req[serving ? j : 0];
        const uint32_t i = r.w + (uint32_t)c;  // candidate index = its block (rt_rng.h)
        const d3 p = sphere_candidate(philox4x32(r.x, r.y, (r.z << 8) | RT_RNG_SCATTER, i, k0, k1));
        const uint64_t accepted = ballot(serving && len2(p) < 1.0);
        // first accepted candidate of my own request, in stream order
        const int first = need ? (rank << lg) : 0;
        const uint64_t width_mask = lg == 6 ? ~0ull : ((1ull << (1 << lg)) - 1ull);
        const uint64_t mine = need ? ((accepted >> first) & width_mask) : 0ull;
        const bool got = mine != 0;
        const int src = got ? first + __ffsll((unsigned long long)mine) - 1 : lane;
        const double rx = shfl_d(p.x, src), ry = shfl_d(p.y, src), rz = shfl_d(p.z, src);
        if (need) result = mk(rx, ry, rz); // (its own candidate's coordinates when it got none: see above)
        if (got) {
            need = false;
            have = true;
        } else if (need) {
            base += 1u << lg;
        }
        pending = ballot(need);
    }
    return have;
}

// util.rs:25-39 random_in_unit_disk for every lane with `need`, same scheme: one
// Philox block per candidate (block i -> x, y), first accepted candidate in stream
// order.  Runs until every request is settled (a fresh path needs its ray now).
__device__ __forceinline__ void coop_random_in_unit_disk(bool need, uint32_t pixel, uint32_t sample, uint32_t k0,
                                                         uint32_t k1, int lane, Req4 *req, double &out_x,
                                                         double &out_y) {
    uint32_t base = 0;
    uint64_t pending = ballot(need);
    while (pending != 0) {
        const int n = __popcll(pending);
        const int lg = n > 32 ? 0 : (n > 16 ? 1 : (n > 8 ? 2 : (n > 4 ? 3 : (n > 2 ? 4 : (n > 1 ? 5 : 6)))));
        if (lg == 0) { // one candidate per request: every lane tests its own (the usual first round of a batch)
            if (need) {
                const u4 b = philox4x32(pixel, sample, RT_RNG_LENS, base, k0, k1);
                const double x = sym53(b.a, b.b), y = sym53(b.c, b.d);
                if (x * x + y * y < 1.0) {
                    out_x = x;
                    out_y = y;
                    need = false;
                } else {
                    base += 1u;
                }
            }
            pending = ballot(need);
            continue;
        }
        const int rank = lane_rank(pending);
        if (need) req[rank] = Req4{pixel, sample, 0u, base};
        const int j = lane >> lg;
        const int c = lane & ((1 << lg) - 1);
        const bool serving = j < n;
        const Req4 r = req[serving ? j : 0];
        const u4 b = philox4x32(r.x, r.y, RT_RNG_LENS, r.w + (uint32_t)c, k0, k1);
        const double x = sym53(b.a, b.b), y = sym53(b.c, b.d);
        const uint64_t accepted = ballot(serving && x * x + y * y < 1.0);
        const int first = need ? (rank << lg) : 0;
        const uint64_t width_mask = lg == 6 ? ~0ull : ((1ull << (1 << lg)) - 1ull);
        const uint64_t mine = need ? ((accepted >> first) & width_mask) : 0ull;
        const bool got = mine != 0;
        const int src = got ? first + __ffsll((unsigned long long)mine) - 1 : lane;
        const double rx = shfl_d(x, src), ry = shfl_d(y, src);
        if (got) {
            out_x = rx;
            out_y = ry;
            need = false;
        } else if (need) {
            base += 1u << lg;
        }
        pending = ballot(need);
    }
}

// SHADING SORTED BY COST: the one expensive texture.  A marble lookup (noise.rs:26-33, :98-109) is seven
// octaves of eight gradient dots, ~900 VALU instructions, and in a wave of 64 paths it is typically wanted by
// a handful of lanes at a time (noise_and_textures: <= 4 lanes in 60 % of the iterations that have one), so
// evaluated where the hit is shaded it runs at ~6 % lane use.  Octaves are independent until they are
// summed, so the wave evaluates them side by side instead: lanes note their lookup during shading
// (texture_value_deferred), then — all 64 lanes, wave-uniform control flow — eight lookups per round take
// eight lanes each, lane o of a group computes octave o (0.5^o * noise(2^o p), both scalings exact), and
// the requester adds the terms in octave order like the reference's loop.  One round costs about what ONE
// octave costs.  Every lookup goes through this code, so a value never depends on which lanes sat next to it.
// Returns turb(p, depth) (noise.rs:98-109) to the lanes with `need`; must be called by the whole wave.
__device__ __forceinline__ double coop_noise_turbulence(bool need, d3 p, int depth, int perlin, const TraceArgs &A,
                                                        const Perlin *lds_perlin, int lane, NoiseSlots &S) {
    double turb = 0.0;
    const uint64_t pending = ballot(need);
    const int n = __popcll(pending);
    const int rank = lane_rank(pending);
    const int j = lane >> 3, o8 = lane & 7; // this lane works on request j of the round, octave o8 (+ 8 per pass)
    for (int r0 = 0; r0 < n; r0 += 8) {
        const bool mine = need && rank >= r0 && rank < r0 + 8;
        const int slot = rank - r0;
        if (mine) {
            S.p[slot][0] = p.x;
            S.p[slot][1] = p.y;
            S.p[slot][2] = p.z;
            S.depth[slot] = depth;
            S.perlin[slot] = perlin;
        }
        const bool serving = r0 + j < n;
        const int dj = serving ? S.depth[j] : 0;
        double accum = 0.0; // noise.rs:99
        for (int ob = 0; ballot(serving && ob < dj) != 0; ob += 8) { // one pass unless a texture has more than 8 octaves
            const int o = ob + o8;
            double term = 0.0;
            if (serving && o < dj) {
                // noise.rs:103-106: temp_p doubles and weight halves per octave — powers of two, exact
                const d3 q = mk(__builtin_ldexp(S.p[j][0], o), __builtin_ldexp(S.p[j][1], o), __builtin_ldexp(S.p[j][2], o));
                const int pi = S.perlin[j];
                const double nz = (lds_perlin != nullptr && pi == 0) ? perlin_noise<true>(*lds_perlin, q)
                                                                     : perlin_noise<false>(A.perlins[pi], q);
                term = __builtin_ldexp(nz, -o);
            }
            S.term[j][o8] = term;
            if (mine) { // accum += weight * noise(temp_p), in octave order (absent octaves are +0.0)
#pragma unroll
                for (int k = 0; k < 8; ++k) accum += S.term[slot][k];
            }
        }
        if (mine) turb = fabs(accum); // noise.rs:108
    }
    return turb;
}

// DELIVERY: what a delivering launch does at the end of an item (rt_device_types.h: TraceArgs.deliver_out).  Out of
// line on purpose: inlined, its address arithmetic and loads raised the register count of the whole path loop
// (80 -> 89 VGPRs in the plain variants, i.e. five blocks per CU instead of six).
__device__ __noinline__ void deliver_item(const RT_CONSTANT TraceArgs *K_in, double sum0, double sum1, double sum2, bool my_valid, int chunk,
                                          int tx, int ty, int tile_py0, bool rows_aligned, int region, uint32_t reg_tiles) {
    const int lane = threadIdx.x & 63;
    const RT_CONSTANT TraceArgs *K;
    // K: the launch's own argument block in the kernarg segment, handed in by the kernel (a by-reference copy of `A` would
    // live in scratch, and llvm.amdgcn.kernarg.segment.ptr is NULL outside a kernel).  Function arguments travel in
    // VGPRs; readfirstlane tells the backend that this one is wave-uniform, so its fields come by scalar loads.
    {
        const uint64_t bits = (uint64_t)K_in;
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bits);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(bits >> 32));
        K = (const RT_CONSTANT TraceArgs *)(((uint64_t)hi << 32) | lo);
    }
    chunk = __builtin_amdgcn_readfirstlane(chunk);
    tx = __builtin_amdgcn_readfirstlane(tx);
    ty = __builtin_amdgcn_readfirstlane(ty);
    tile_py0 = __builtin_amdgcn_readfirstlane(tile_py0);
    region = __builtin_amdgcn_readfirstlane(region);
    reg_tiles = (uint32_t)__builtin_amdgcn_readfirstlane((int)reg_tiles);
    {
        const size_t slice = (size_t)K->slice_rows * (size_t)K->width * 3; // a slice holds the launch's owned rows only
        // (pixel coordinates are formed again here rather than kept in registers across the path loop)
        const int out_px = (tx * 8 + (lane & 7)) * K->step_x;
        int out_py = tile_py0 + (lane >> 3) * K->step_y;
        if (!rows_aligned) {
            const int vrow = ty * 8 + (lane >> 3);
            out_py = ((vrow / K->strip_rows) * K->strip_count + K->strip_index) * K->strip_rows + vrow % K->strip_rows;
        }
        const size_t in_slice = ((size_t)(ty * 8 + (lane >> 3)) * (size_t)K->width + (size_t)out_px) * 3;
        double *dst = K->partial + (size_t)(K->chunk_base + chunk) * slice + in_slice;
        // ---- DELIVERY: the launch finishes its own pixels (rt_device_types.h: TraceArgs.deliver_out).
        // The slices cross waves inside ONE launch here, and an XCD's L2 is not coherent with its seven neighbours':
        // a device-scope release fence per item (L2 write-back + invalidate) was measured first and costs C2 12 %
        // (profiles/r03_fence_probe.txt).  Instead the slice stores and loads are device-scope relaxed atomics —
        // plain stores / loads with the sc1 bit, which write through to / read from the level all XCDs share — and
        // s_waitcnt orders this wave's stores before its bump of the tile's counter: no cache maintenance at all.
        if (my_valid) {
            __hip_atomic_store(dst + 0, sum0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dst + 1, sum1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dst + 2, sum2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the stores have been acknowledged at device scope
        const int tile_id = ty * K->tiles_x + tx;
        uint32_t chunks_before = 0;
        if (lane == 0) chunks_before = __hip_atomic_fetch_add(K->tile_done + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        chunks_before = (uint32_t)__builtin_amdgcn_readfirstlane((int)chunks_before);
        if ((int)chunks_before != K->total_chunks - 1) return;
        // Every chunk of this tile has landed: vec3.rs:119-125 scale_sqrt over the slices in chunk order —
        // exactly k_resolve_chunks_f64's sum, so the pixels are bit-identical to the two-pass path.
        if (my_valid) {
            const double *src = K->partial + in_slice;
            double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
#pragma unroll 1
            for (int c = 0; c < K->total_chunks; ++c) {
                acc0 += __hip_atomic_load(src + (size_t)c * slice + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                acc1 += __hip_atomic_load(src + (size_t)c * slice + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                acc2 += __hip_atomic_load(src + (size_t)c * slice + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const double scale = 1.0 / (double)K->samples; // what rtdev_launch_resolve_chunks passes
            // tile-column layout of the output (one column = the plain frame): column `col` starts at pixel column
            // col * step and is stored as [height][its width][3] behind the columns before it
            int col = out_px / K->deliver_col_step;
            if (col > K->deliver_cols - 1) col = K->deliver_cols - 1;
            const int col_x = col * K->deliver_col_step;
            const int col_w = col == K->deliver_cols - 1 ? K->width - col_x : K->deliver_col_step;
            double *out = K->deliver_out + ((size_t)K->height * (size_t)col_x + (size_t)out_py * (size_t)col_w + (size_t)(out_px - col_x)) * 3;
            out[0] = sqrt(scale * acc0);
            out[1] = sqrt(scale * acc1);
            out[2] = sqrt(scale * acc2);
        }
        if (lane == 0) __hip_atomic_store(K->tile_done + tile_id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // re-armed for the next launch on this scene (written through, like the slices)
        // The pixels may sit in HOST memory: release them at system scope before this tile is counted (once per
        // tile, 1/19 of the items on C3), and publish the region behind an acquire of the other waves' releases.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        uint32_t tiles_before = 0;
        if (lane == 0) tiles_before = __hip_atomic_fetch_add(K->region_done + region, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tiles_before = (uint32_t)__builtin_amdgcn_readfirstlane((int)tiles_before);
        if (tiles_before == reg_tiles - 1u) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (lane == 0) {
                __hip_atomic_store(K->region_done + region, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(K->deliver_flags + region, K->deliver_serial, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// BVH: closest hit through the skip-link hierarchy instead of the linear loop
// (instantiated for PRIMS_ANY only; chosen for scenes with many primitives).
template <int PRIMS, bool TEXTURED, bool SPECULAR, bool BVH>
__global__ __launch_bounds__(256, TEXTURED ? (PRIMS == PRIMS_ANY ? (BVH ? RT_OCC_TEX_BVH : RT_OCC_TEX_ANY) : RT_OCC_TEX) : (BVH ? 4 : (PRIMS == PRIMS_ANY ? RT_OCC_ANY : (SPECULAR ? RT_OCC_SPEC : RT_OCC_PLAIN)))) void k_trace_pool_f64(const TraceArgs A) {
    // Two batches of camera samples stay ahead of the hand-out, so that it may straddle a batch
    // boundary; the BVH variants keep one (their node array wants the LDS: three resident blocks
    // instead of two on the `random` scene) and a hand-out stops at the end of its batch.
    constexpr int NBUF = BVH ? 1 : 2;
    // Two items in flight where it pays: the variants whose iterations are long (any primitive kind, BVH).  The rects-only
    // and spheres-only variants gain under 1 % from it (their tail iterations are cheap: no hand-out, no batches, one sampler
    // round) and lose 2 % to the bookkeeping, so they keep one item at a time and f64 sums.
#if defined(RT_EXACT_DIV) || defined(RT_NO_OVERLAP)
    constexpr bool OVERLAP = false;
#else
    constexpr bool OVERLAP = BVH || PRIMS == PRIMS_ANY;
#endif
    __shared__ WaveLds<TEXTURED, NBUF, OVERLAP> lds_all[4];
    // Dynamic LDS of a block: [BVH nodes | primitive table + texture table][Perlin gradients]; the host sizes it
    // (pool_dynamic_lds below) and says what is in it.
    extern __shared__ __align__(16) unsigned char dyn_lds[];
    // The gradients of the first Perlin table (6 KB) are staged in LDS once per block when the
    // permutation tables are the identity (always, in the reference: noise.rs:121-130): the 56
    // random gradient fetches of a marble lookup then hit LDS instead of the vector memory
    // path, and the lattice hash needs no table.  Scenes without a Noise texture do not pay the 6 KB.
    struct PerlinGradients {
        double ranvec[256][3];
    };
    static_assert(offsetof(Perlin, ranvec) == 0, "the gradients lead the Perlin record");
    const Perlin *lds_perlin = nullptr;
    if (TEXTURED) {
        if (A.perlin_in_lds) {
            const size_t at = BVH ? (size_t)A.bvh_lds_nodes * sizeof(BvhNode)
                                  : (size_t)A.n_prims * sizeof(Prim) + (size_t)A.n_textures * sizeof(Texture);
            const uint64_t *src = reinterpret_cast<const uint64_t *>(A.perlins);
            uint64_t *dst = reinterpret_cast<uint64_t *>(dyn_lds + at);
            for (int i = threadIdx.x; i < (int)(sizeof(PerlinGradients) / 8); i += 256) dst[i] = src[i];
            lds_perlin = reinterpret_cast<const Perlin *>(dyn_lds + at); // only .ranvec is read (IDENTITY path)
        }
        __syncthreads();
    }
    // the lens-disk samples of this wave's batches: [NBUF][64][2] doubles per wave behind the tables above (see WaveLds)
    double (*lens)[64][2] = nullptr;
    if (A.lens_lds) {
        size_t at = BVH ? (size_t)A.bvh_lds_nodes * sizeof(BvhNode)
                        : (size_t)A.n_prims * sizeof(Prim) + (TEXTURED ? (size_t)A.n_textures * sizeof(Texture) : 0);
        if (TEXTURED && A.perlin_in_lds) at += sizeof(PerlinGradients);
        lens = reinterpret_cast<double (*)[64][2]>(dyn_lds + at) + (threadIdx.x >> 6) * NBUF;
    }
    // ... and the ray times of the batches, [NBUF][64] doubles per wave behind those: only a scene with a MovingSphere reads a
    // ray's time (ray.rs:26-28), and without their 1 KB per wave more blocks fit a CU
    double (*ray_times)[64] = nullptr;
    if (PRIMS == PRIMS_ANY && A.time_lds) {
        size_t at = BVH ? (size_t)A.bvh_lds_nodes * sizeof(BvhNode)
                        : (size_t)A.n_prims * sizeof(Prim) + (TEXTURED ? (size_t)A.n_textures * sizeof(Texture) : 0);
        if (TEXTURED && A.perlin_in_lds) at += sizeof(PerlinGradients);
        if (A.lens_lds) at += (size_t)4 * NBUF * 64 * 2 * sizeof(double); // rt_device_types.h: pool_lens_lds_bytes
        ray_times = reinterpret_cast<double (*)[64]>(dyn_lds + at) + (threadIdx.x >> 6) * NBUF;
    }
    // BVH nodes are staged in dynamic LDS when they fit (the host sets bvh_lds_nodes):
    // a traversal step is a dependent load, and ~100 steps at
    // L2 latency with 3-4 waves per SIMD is what bounds the big-scene variants.
    const BvhNode *lds_nodes = nullptr;
    if (BVH && A.bvh_lds_nodes > 0) {
        const uint4 *src = reinterpret_cast<const uint4 *>(A.bvh_nodes);
        uint4 *dst = reinterpret_cast<uint4 *>(dyn_lds);
        for (int i = threadIdx.x; i < A.bvh_lds_nodes * (int)(sizeof(BvhNode) / 16); i += 256) dst[i] = src[i];
        lds_nodes = reinterpret_cast<const BvhNode *>(dyn_lds);
        __syncthreads();
    }
    // The linear-loop variants stage the whole primitive table (192 B each, materials included)
    // in dynamic LDS instead: the closest-hit loop keeps reading it through the scalar cache, but
    // the per-lane fetch of the WINNING primitive for shading is then an LDS read, not a trip
    // through the vector memory path in the middle of every iteration.
    const Prim *lds_prims = nullptr;
    if (!BVH) {
        const uint4 *src = reinterpret_cast<const uint4 *>(A.prims);
        uint4 *dst = reinterpret_cast<uint4 *>(dyn_lds);
        for (int i = threadIdx.x; i < A.n_prims * (int)(sizeof(Prim) / 16); i += 256) dst[i] = src[i];
        lds_prims = reinterpret_cast<const Prim *>(dyn_lds);
        __syncthreads();
    }
    // ... and the texture table (64 B each) behind it: a Checkered lookup is a dependent chain
    // texture -> sub-texture
    const Texture *lds_textures = nullptr;
    if (!BVH && TEXTURED) {
        const uint4 *src = reinterpret_cast<const uint4 *>(A.textures);
        uint4 *dst = reinterpret_cast<uint4 *>(dyn_lds + (size_t)A.n_prims * sizeof(Prim));
        for (int i = threadIdx.x; i < A.n_textures * (int)(sizeof(Texture) / 16); i += 256) dst[i] = src[i];
        lds_textures = reinterpret_cast<const Texture *>(dyn_lds + (size_t)A.n_prims * sizeof(Prim));
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const int lane_of_wave = lane;
    WaveLds<TEXTURED, NBUF, OVERLAP> &L = lds_all[threadIdx.x >> 6];
    unsigned int n_segments = 0, n_started = 0;
    if (OVERLAP && lane < 3) L.ulc[lane] = A.cam.ulc[lane];

    // SUMS AND OVERLAPPED ITEMS.  When the pool of an item is dry its last paths still take a dozen iterations to end, with
    // ever fewer lanes tracing (`random`: 14 % of a wave's iterations ran at 14 lanes, cornell_box_boxes 13 % at 19, C4
    // 6 % at 19; profiles/r04_region_cycles.txt).  With per-pixel sums kept as doubles the wave has to sit that tail out: the
    // order in which an item's samples reach its sums — hence their rounding — must not depend on what else the wave is
    // doing, or the frame would depend on which wave drew which item.  The OVERLAP variants keep the sums as INTEGERS
    // instead (TraceArgs.sum_scale: the scene's radiance is bounded, so a sample fits a 52-bit fixed-point number) — exact,
    // so their order does not matter — and draw the next item the moment the pool runs dry: two items are in flight, the
    // lanes the old one frees start the new one's paths (cornell_box_boxes -5 %, `random` -2.5 % of the frame time; the
    // frame is then also the same for strips that cut the item tiles: tests/test_gpu_overlap.py).  The other variants,
    // and every RT_ARITH_REFERENCE kernel, add doubles and start the next item after the last path.  (A scene without a
    // radiance bound — a colour above 1 on a scattering material — is given to the RT_ARITH_REFERENCE copy by the host.)
    //
    // The wave-uniform flags of this bookkeeping are bits of ONE integer and the tests on them integer compares: as `bool`s
    // they lived in 64-bit lane masks, which the path loop's scalar register pressure sent to VGPR lanes — a dozen v_readlane
    // per iteration.  They are also made opaque where the path loop tests them: a loop-invariant test is hoisted out of the
    // loop AS A LANE MASK, with the same fate.
    constexpr bool FIXED_SUMS = OVERLAP;
#ifdef RT_EXACT_DIV
    const uint32_t sum_scale_hi = 0u;
#else
    const uint32_t sum_scale_hi = (uint32_t)((unsigned long long)__double_as_longlong(A.sum_scale) >> 32); // 2^k: the lower half is 0
#endif
    // ---- the item entries are handed out of (slot `cur` of L.sum / L.info); the other slot may hold an item whose
    // pool is dry and whose last paths are still in flight (`draining`)
    // HAVE: slot `cur` holds an item (its pool may be dry); CUR: cur << 6, the slot bit as it sits in a lane's `spix`
    // POOL_DRY (two-item variants): the hand-out has just taken the pool's last entries
    enum : uint32_t { HAVE = 1u, DRAINING = 2u, QUEUE_DRY = 4u, POOL_DRY = 8u, CUR = 64u };
    uint32_t state = 0u;
    int smp0 = 0, n_valid = 0;
    int tile_py0 = 0; // image row of the current tile's first row; -1: strips that cut tiles (the batches then take the long road)
    uint32_t total = 0, next = 0, n_batches = 0, batches_done = 0;
    // `next` at which the next batch of camera samples is due (batch b when next >= (b - (NBUF - 1)) * 64: NBUF batches stay
    // ahead of the hand-out), or ~0 when the item's batches are all drawn: ONE scalar compare per iteration decides
    uint32_t batch_due = ~0u;
    uint32_t my_pixel = 0; // image index of this lane's pixel of the current item's tile

    RT_REGION_DECL
    // Draws the wave's next item and sets slot `cur` up for it; false when the queue is dry.  (All 64 lanes.)
    auto start_item = [&]() -> bool {
        // (an opaque copy of the lane index: whatever of the code below depends on the lane alone would otherwise be computed
        // once, in front of the loops, and parked in scratch memory across them — the plain variants have none)
        int lane = lane_of_wave;
        asm volatile("" : "+v"(lane));
        uint32_t item = 0;
        if (lane == 0) {
            item = atomicAdd(A.queue, 1u);
            // The host's cancel word (rt_device_types.h: cancel_flag).  Reads of pinned host memory are a scarce resource —
            // every item of every wave asking cost rt_render 30 % on C3 (615 000 items, ~28 M such reads per second is
            // what the link gives) — so only the wave that draws an item whose number is a multiple of 32 asks, and when
            // the word is up it poisons the launch's item counter ITSELF: every other wave's next atomicAdd then returns
            // "queue dry".  On C3 an item is drawn every 0.12 us, so somebody asks every 4 us.
            const unsigned int *cf = kernargs_here()->cancel_flag;
            if (cf != nullptr && (item & 31u) == 0u && item < A.n_items &&
                __hip_atomic_load(cf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) {
                atomicOr(A.queue, 0x80000000u);
                item = 0x80000000u;
            }
        }
        item = (uint32_t)__builtin_amdgcn_readfirstlane((int)item);
        if (item >= A.n_items) return false;
        // item -> (region, chunk, tile): regions in queue order, chunk-major inside a region (rt_device_types.h: Region)
        int region = 0, reg_tx0 = 0, reg_ty0 = 0, reg_ntx = A.tiles_x;
        uint32_t reg_tiles = (uint32_t)A.n_tiles, local = item;
        if (A.n_regions > 1) {
            const RT_CONSTANT TraceArgs *K = kernargs_here();
            while (region + 1 < A.n_regions && item >= K->regions[region + 1].item_begin) ++region;
            local = item - K->regions[region].item_begin;
            reg_tx0 = K->regions[region].tx0;
            reg_ntx = K->regions[region].ntx;
            reg_ty0 = K->regions[region].ty0;
            reg_tiles = (uint32_t)reg_ntx * (uint32_t)K->regions[region].nty;
        }
        const uint32_t chunk = local / reg_tiles;
        const uint32_t tile = local - chunk * reg_tiles;
        const int ty = reg_ty0 + (int)(tile / (uint32_t)reg_ntx);
        const int tx = reg_tx0 + (int)(tile % (uint32_t)reg_ntx);
        const int cur = (int)(state >> 6) & 1;
        int n_smp; // the chunk's samples (chunk = index within this launch)
        {
            const RT_CONSTANT TraceArgs *K = kernargs_here();
            smp0 = K->chunk_start[A.chunk_base + (int)chunk];
            n_smp = K->chunk_start[A.chunk_base + (int)chunk + 1] - smp0;
        }

        // ---- this lane's pixel of the tile (used for the pool table and the hand-out's pixel index)
        // (step_x/step_y > 1: the preview renderer, cpu_scaled.rs — the grid cell is the
        // top-left pixel of a block, the resolve pass fills the block)
        // Image row of the tile's first row.  With strips of a multiple of 8 rows (what every multi-GPU host here
        // uses) a tile lies inside ONE strip, so the owned-row -> image-row map is one scalar division per item;
        // other strip heights take the per-lane division.
        const bool rows_aligned = A.strip_count <= 1 || (A.strip_rows & 7) == 0;
        tile_py0 = ty * 8 * A.step_y;
        if (A.strip_count > 1 && rows_aligned) {
            const int q = (ty * 8) / A.strip_rows; // wave-uniform
            tile_py0 = (q * A.strip_count + A.strip_index) * A.strip_rows + (ty * 8 - q * A.strip_rows);
        }
        const int my_px = (tx * 8 + (lane & 7)) * A.step_x;
        const int my_vrow = ty * 8 + (lane >> 3);
        int my_py = tile_py0 + (lane >> 3) * A.step_y;
        if (!rows_aligned)
            my_py = ((my_vrow / A.strip_rows) * A.strip_count + A.strip_index) * A.strip_rows + my_vrow % A.strip_rows;
        const bool my_valid = my_px < A.cover_w && my_vrow < A.owned_rows && my_py < A.height;
        my_pixel = (uint32_t)my_py * (uint32_t)A.width + (uint32_t)my_px;
        const uint64_t valid_mask = ballot(my_valid);
        n_valid = __popcll(valid_mask);
        {
            PathRng prng{my_pixel, RT_RNG_SAMPLE_PIXEL, A.seed_lo, A.seed_hi};
            u4 bj = prng.block(0, RT_RNG_PIXEL, 0);
            const RT_CONSTANT TraceArgs *K = kernargs_here();
            const double u = div_by((double)my_px + u53(bj.a, bj.b), (double)(A.width - 1), K->inv_width_m1); // cpu.rs:35-36
            if (OVERLAP) {
                L.u[lane] = u;
            } else {
                const d3 base = ld3(K->cam.ulc) + u * ld3(K->cam.horizontal); // camera.rs:331, first two terms
                L.base[lane][0] = base.x;
                L.base[lane][1] = base.y;
                L.base[lane][2] = base.z;
            }
            // fixed-point sums: a sample arrives as the bit pattern of (T * scale + 2^52), i.e. 0x433 << 52 plus the
            // integer; the n_smp patterns' exponent fields are taken off here, once, instead of masked off every sample
            const unsigned long long zero = FIXED_SUMS ? 0ull - (unsigned long long)n_smp * 0x4330000000000000ull : 0ull;
            unsigned long long *sum = reinterpret_cast<unsigned long long *>(L.sum[cur * 64 + lane]);
            sum[0] = zero;
            sum[1] = zero;
            sum[2] = zero;
            if (my_valid) L.pix_of[lane_rank(valid_mask)] = lane;
            if (lane == 0) L.info[cur] = ItemInfo{(int)chunk, tx, ty, tile_py0, region, (int)reg_tiles, rows_aligned ? 1 : 0, 0};
        }
        if (!rows_aligned) tile_py0 = -1;
        total = (uint32_t)n_valid * (uint32_t)n_smp; // paths in this item's pool
        next = 0;
        n_batches = (total + 63u) >> 6;
        batches_done = 0;
        batch_due = n_batches > 0u ? 0u : ~0u;
        return true;
    };
    // The item of slot `s` has no path left: its sums go to its own slice of `partial` (or the launch finishes the tile's
    // pixels itself: deliver_item).  (All 64 lanes.)
    auto finish_item = [&](int s) {
        int lane = lane_of_wave;
        asm volatile("" : "+v"(lane));
        const ItemInfo I = L.info[s];
        const int chunk = __builtin_amdgcn_readfirstlane(I.chunk), f_tx = __builtin_amdgcn_readfirstlane(I.tx),
                  f_ty = __builtin_amdgcn_readfirstlane(I.ty), f_py0 = __builtin_amdgcn_readfirstlane(I.tile_py0);
        const bool f_aligned = __builtin_amdgcn_readfirstlane(I.rows_aligned) != 0;
        const int f_px = (f_tx * 8 + (lane & 7)) * A.step_x;
        const int f_vrow = f_ty * 8 + (lane >> 3);
        int f_py = f_py0 + (lane >> 3) * A.step_y;
        if (!f_aligned) f_py = ((f_vrow / A.strip_rows) * A.strip_count + A.strip_index) * A.strip_rows + f_vrow % A.strip_rows;
        const bool f_valid = f_px < A.cover_w && f_vrow < A.owned_rows && f_py < A.height;
        double s0 = L.sum[s * 64 + lane][0], s1 = L.sum[s * 64 + lane][1], s2 = L.sum[s * 64 + lane][2];
        if (FIXED_SUMS) { // the integers, rounded ONCE to the nearest double; the scale is a power of two
            const double unscale = kernargs_here()->sum_unscale;
            s0 = (double)(unsigned long long)__double_as_longlong(s0) * unscale;
            s1 = (double)(unsigned long long)__double_as_longlong(s1) * unscale;
            s2 = (double)(unsigned long long)__double_as_longlong(s2) * unscale;
        }
        if (A.deliver_out == nullptr) {
            if (f_valid) {
                // slice row = row of the launch's owned-row grid (a share's slices hold its own rows only)
                const size_t in_slice = (size_t)f_vrow * (size_t)A.width + (size_t)f_px;
                double *dst = A.partial + ((size_t)(A.chunk_base + chunk) * (size_t)A.slice_rows * (size_t)A.width + in_slice) * 3;
                dst[0] = s0;
                dst[1] = s1;
                dst[2] = s2;
            }
        } else {
            deliver_item(kernargs_here(), s0, s1, s2, f_valid, chunk, f_tx, f_ty, f_py0, f_aligned, __builtin_amdgcn_readfirstlane(I.region),
                         (uint32_t)__builtin_amdgcn_readfirstlane(I.reg_tiles));
        }
    };
    // Pool entry w of the current item = (pixel w % n_valid of the tile, sample smp0 + w / n_valid).
    auto entry_of = [&](uint32_t w, int &py_out, uint32_t &pixel_out, uint32_t &sample_out) { // (all 64 lanes: a lane shuffle)
        int s_off, pix;
        if (n_valid == 64) {
            pix = (int)(w & 63u);
            s_off = (int)(w >> 6);
        } else {
            s_off = (int)(w / (uint32_t)n_valid);
            pix = L.pix_of[w - (uint32_t)s_off * (uint32_t)n_valid];
        }
        // lane p of the wave holds the image index of pixel p of the tile (my_pixel); the image row follows from the
        // tile's first row — or, with strips that cut tiles (tile_py0 < 0), from the parked tile row
        pixel_out = (uint32_t)shfl_i((int)my_pixel, pix);
        py_out = tile_py0 + (pix >> 3) * A.step_y;
        if (tile_py0 < 0) {
            const int vrow = __builtin_amdgcn_readfirstlane(L.info[(state >> 6) & 1u].ty) * 8 + (pix >> 3);
            py_out = ((vrow / A.strip_rows) * A.strip_count + A.strip_index) * A.strip_rows + vrow % A.strip_rows;
        }
        sample_out = (uint32_t)(smp0 + s_off);
    };
    // REGENERATION BATCHES.  A lane that starts a new path needs the path's camera
    // sample: the vertical jitter and ray time (one Philox block) and the lens disk (a
    // rejection loop).  Only ~16 of 64 lanes start a path in a given iteration, so doing
    // this at the hand-out runs ~120 instructions at 25 % lane use.  Entries leave the pool
    // in index order, so the wave instead draws the samples of 64 consecutive entries at
    // once — every lane busy — into LDS, two batches ahead of `next`.  (The buffers belong to the current item: an item
    // that still has paths in flight when the next one starts has handed out all its entries.)
    auto prepare_batch = [&](uint32_t b) {
        int lane = lane_of_wave; // (opaque, like start_item's)
        asm volatile("" : "+v"(lane));
        const RT_CONSTANT TraceArgs *K = kernargs_here();
        const uint32_t w = b * 64u + (uint32_t)lane;
        const bool in_pool = w < total;
        int py_b = 0;
        uint32_t pixel_b = 0, sample_b = 0;
        entry_of(w, py_b, pixel_b, sample_b); // (entries behind the pool's end: values nobody reads)
        const u4 bc = philox4x32(pixel_b, sample_b, RT_RNG_CAMERA, 0u, A.seed_lo, A.seed_hi);
        const int buf = (int)(b & (uint32_t)(NBUF - 1));
        L.v[buf][lane] = div_by((double)py_b + u53(bc.a, bc.b), (double)(A.height - 1), K->inv_height_m1); // cpu.rs:39-40
        // camera.rs:335: the ray's time, second double of the same block (MovingSphere reads it)
        if (PRIMS == PRIMS_ANY && ray_times != nullptr) ray_times[buf][lane] = K->cam.time_a + (K->cam.time_b - K->cam.time_a) * u53(bc.c, bc.d);
        // camera.rs:327: aperture 0 multiplies the disk by 0, so its draws are dead and skipped
        if (K->cam.lens_radius != 0.0) {
            double lx = 0.0, ly = 0.0;
            coop_random_in_unit_disk(in_pool, pixel_b, sample_b, A.seed_lo, A.seed_hi, lane, L.scratch.req, lx, ly);
            lens[buf][lane][0] = lx * K->cam.lens_radius; // camera.rs:327 `lens_radius * random_in_unit_disk()`: formed here, by all
            lens[buf][lane][1] = ly * K->cam.lens_radius; // 64 lanes, not at the hand-out by the quarter of them that start a path
        }
    };

    // ---- path state of this lane (it outlives the items: a lane's path may belong to either slot)
    bool alive = false;
    int spix = 0;         // slot of the item the current path belongs to << 6 | its pixel of that item's tile (lane order)
    PathRng rng{0, 0, A.seed_lo, A.seed_hi};
    d3 o = mk(0, 0, 0), d = o, T = o;
    uint32_t seg = 0;
    double ray_time = 0.0; // ray.rs:26-28; scattered rays inherit it (e.g. lambertian.rs:35)
    // A lane whose hit needs a random_in_unit_sphere sample that the wave has not
    // found yet stays `waiting` (it keeps its hit below and skips tracing) until a
    // later iteration's sampler rounds reach its accepted candidate.
    bool waiting = false;
    uint32_t cand_base = 0;   // first untested candidate of the open request
    bool is_lambert = false;  // material of the open hit (else Metal)
    // the open hit: its point takes the ray origin's place (`o` is dead once the hit record exists) and its
    // attenuation goes into T at once, so neither is carried as extra state while the lane waits
    d3 hit_normal = o;
    double fuzz = 0.0;

    for (;;) {
        // ---- the items in flight (wave-uniform; out here, not in the path loop below: with this code inside it the path
        // state was copied from register to register around it in every iteration — C3 5.78 -> 6.84 vector instructions
        // per segment)
        if ((state & (HAVE | DRAINING)) == HAVE && next >= total) { // the pool is dry: what is in flight of it drains in the other slot
            state ^= HAVE | DRAINING | (OVERLAP ? CUR : 0u);
            state &= ~POOL_DRY;
            total = next = n_batches = batches_done = 0; // (no pool: the hand-out below finds nothing to do)
            batch_due = ~0u;
        }
        // (without OVERLAP there is one slot, and whatever is in flight belongs to the draining item)
        if ((state & DRAINING) != 0u && ballot(alive && (!OVERLAP || (((uint32_t)spix ^ state) & CUR) != 0u)) == 0) { // the last path of the draining item has ended
            finish_item(OVERLAP ? (int)((state ^ CUR) >> 6) & 1 : 0);
            state &= ~DRAINING;
        }
        if ((state & (HAVE | QUEUE_DRY)) == 0u && (FIXED_SUMS || (state & DRAINING) == 0u)) {
            state |= start_item() ? HAVE : QUEUE_DRY;
            if (OVERLAP && (state & HAVE) != 0u && total == 0u) state |= POOL_DRY; // (an item without a pixel: nothing to hand out)
        }
        if ((state & (HAVE | DRAINING)) == 0u) break; // the queue is dry and nothing is in flight
        RT_REGION(0); // item setup / end
        if (!OVERLAP) {
            // one item at a time: no path is in flight here.  Saying so — every lane's path state is set anew — ends the
            // state's live ranges at the loop's exit: across the item code above they would otherwise hold their registers
            // (the plain variants: 77 -> 80 VGPRs and 64 bytes of scratch).
            alive = false;
            waiting = false;
            is_lambert = false;
            spix = 0;
            rng.pixel = rng.sample = 0;
            o = d = T = hit_normal = mk(0, 0, 0);
            seg = cand_base = 0;
            ray_time = fuzz = 0.0;
        }
        // ---- the path loop: until the pool runs dry or the draining item's last path ends
        for (;;) {
        // ---- camera samples for the entries about to leave the pool (whole wave, see above)
        while (next >= batch_due) {
            prepare_batch(batches_done++);
            batch_due = batches_done >= n_batches ? ~0u : (batches_done < (uint32_t)NBUF ? 0u : (batches_done - (uint32_t)(NBUF - 1)) << 6);
        }
        RT_REGION(1); // batches
        // ---- hand pool entries to the lanes without a path (ballot + prefix count)
        if (next < total) {
            const uint64_t idle = ballot(!alive);
            const uint32_t w = next + (uint32_t)lane_rank(idle);
            // entries whose camera samples are in LDS: all of them with two buffers, the current batch with one
            const uint32_t ready = NBUF == 2 ? total : min(total, batches_done << 6);
            next = min(next + (uint32_t)__popcll(idle), ready);
            if (OVERLAP && next >= total) state |= POOL_DRY;
            // entry_of(w) without the pixel arithmetic: lane p of the wave holds pixel p's index in the image
            // (my_pixel), so the entry's comes by a lane shuffle — which every lane has to take part in, hence
            // out here (the lanes that have a path compute an entry nobody reads)
            int s_off, pix_new;
            if (n_valid == 64) {
                pix_new = (int)(w & 63u);
                s_off = (int)(w >> 6);
            } else {
                s_off = (int)(w / (uint32_t)n_valid);
                pix_new = L.pix_of[w - (uint32_t)s_off * (uint32_t)n_valid];
            }
            const uint32_t pixel_new = (uint32_t)shfl_i((int)my_pixel, pix_new);
            if (!alive && w < ready) { // cpu.rs:39-40 + camera.rs:326-337
                spix = (int)(state & CUR) | pix_new;
                rng.pixel = pixel_new;
                rng.sample = (uint32_t)(smp0 + s_off);
                const RT_CONSTANT TraceArgs *K = kernargs_here();
                const int buf = (int)((w >> 6) & (uint32_t)(NBUF - 1)), slot = (int)(w & 63u);
                const double v = L.v[buf][slot];
                const d3 co = ld3(K->cam.origin);
                o = co;
                if (OVERLAP) d = (ld3(L.ulc) + L.u[pix_new] * ld3(K->cam.horizontal)) - v * ld3(K->cam.vertical) - co; // camera.rs:331
                else d = ld3(L.base[pix_new]) - v * ld3(K->cam.vertical) - co;
                const double lr = K->cam.lens_radius;
                if (lr != 0.0) {
                    const d3 offset = ld3(K->cam.right) * lens[buf][slot][0] + ld3(K->cam.up) * lens[buf][slot][1];
                    o = co + offset;
                    d = d - offset;
                }
                if (PRIMS == PRIMS_ANY && ray_times != nullptr) ray_time = ray_times[buf][slot];
                T = mk(1.0, 1.0, 1.0);
                seg = 0;
                alive = true;
                ++n_started;
            }
        }
        RT_REGION(2); // hand-out + primary ray
        // (one item at a time: the loop ends when the pool is dry and nothing is in flight — a scalar test on a ballot.
        // The OVERLAP variants leave at the bottom instead.)
        if (!OVERLAP && ballot(alive) == 0) break;
        // ---- one ray_color level for every lane with a path
        // A path that ends here adds its throughput T (times what it ran into) to its pixel: T is dead
        // afterwards, so the product is formed in place.
        bool ended = false;
        bool scattered = false;  // the path got a new ray this iteration (depth check below)
        bool finish = false;     // Lambertian / Metal hit whose direction can be completed now
        int noise_tex = -1;      // Noise texture this lane's hit wants (evaluated by the whole wave below)
        RT_LANES(__popcll(ballot(alive && !waiting)), next >= total);
        if (alive && !waiting) {
            int max_depth = A.max_depth;
            asm volatile("" : "+s"(max_depth)); // (opaque: hoisted out of the loop the test lives in a lane mask that is spilled)
            if (max_depth <= 0) { // renderer.rs:48-55 with max_depth 0
                ended = true;
            } else {
                ++n_segments;
                double best_t = __builtin_inf(); // closest hit, t in [0.001, inf) (renderer.rs:58)
                int best = -1, best_aux = 0;
                const d3 inv_d = rcp3(d);
                const double inv_a = PRIMS == PRIMS_RECTS ? 0.0 : rcp_f64(len2(d));
                if (BVH) {
#ifdef RT_PROFILE_REGIONS
                    unsigned walk[2] = {0, 0};
                    unsigned *walk_stats = walk;
#else
                    unsigned *walk_stats = nullptr;
#endif
#ifdef RT_PROFILE_REGIONS
                    auto walk_mark = [&](int k) { RT_REGION(k); };
#else
                    const NoMark walk_mark;
#endif
                    if (lds_nodes != nullptr)
                        closest_hit_bvh<PRIMS>(A, lds_nodes, o, d, inv_d, inv_a, ray_time, 0.001, best_t, best, best_aux, walk_stats, walk_mark);
                    else
                        closest_hit_bvh<PRIMS>(A, bvh_nodes_for(A, d), o, d, inv_d, inv_a, ray_time, 0.001, best_t, best, best_aux, walk_stats, walk_mark);
#ifdef RT_PROFILE_REGIONS
                    atomicAdd(&rt_t_[37], (unsigned long long)walk[0]); // per-lane totals (LDS atomics)
                    atomicAdd(&rt_t_[38], (unsigned long long)walk[1]);
#endif
                } else {
                    auto test = [&](const Prim &P, int i) {
                        double t;
                        int aux;
                        if (prim_t<PRIMS>(P, o, d, inv_d, inv_a, ray_time, 0.001, best_t, t, aux)) {
                            best_t = t;
                            best = i;
                            best_aux = aux;
                        }
                    };
                    // two records per scalar-load wait: the table's latency is paid n/2 times, not n
                    // (C3 +2.8 %, C2 +3.8 %; three per wait run out of SGPRs and lose it again)
                    if (PRIMS != PRIMS_SPHERES) {
                        // The table is grouped (rect_end, sphere_end): one straight-line test per group, the plane a
                        // compile-time constant, instead of a scalar switch on the kind of every record.
                        auto test_plane = [&](auto axis, const Prim &P, int i) {
#ifndef RT_EXACT_DIV
                            rect_closest_update<decltype(axis)::value>(P.p[0], P.p[1], P.p[2], P.p[3], P.p[4], o, d, inv_d, 0.001, best_t, best, i);
                            return;
#endif
                            double t;
                            if (rect_t<true>(decltype(axis)::value, P.p[0], P.p[1], P.p[2], P.p[3], P.p[4], o, d, inv_d, 0.001, best_t, t)) {
                                best_t = t;
                                best = i;
                            }
                        };
                        // ONE pointer runs through the groups (they follow each other in the table): an address formed from the
                        // index — a 64-bit multiply-add, four scalar instructions — for every group's first record and every odd
                        // one out was a tenth of the rects-only variant's scalar instructions
                        const Prim *rec = A.prims;
                        int i = 0;
                        auto group = [&](auto axis, int end) {
                            for (; i + 1 < end; i += 2, rec += 2) {
                                const Prim pa = load_prim_uniform(rec, 0), pb = load_prim_uniform(rec, 1);
                                test_plane(axis, pa, i);
                                test_plane(axis, pb, i + 1);
                            }
                            if (i < end) {
                                test_plane(axis, load_prim_uniform(rec, 0), i);
                                ++i;
                                ++rec;
                            }
                        };
                        // the group bounds are re-read from the kernel arguments HERE (kernargs_here): hoisted out of the
                        // path loop their emptiness tests sit in SGPR pairs that the allocator parks in VGPR lanes, and
                        // every iteration pays a v_readlane (4.3 SIMD cycles of the saturated vector pipe) per half of them
                        const RT_CONSTANT TraceArgs *KB = kernargs_here();
                        const int end_xy = KB->rect_end[0], end_xz = KB->rect_end[1], end_yz = KB->rect_end[2];
                        group(std::integral_constant<int, 2>(), end_xy);  // XY
                        group(std::integral_constant<int, 1>(), end_xz);  // XZ
                        group(std::integral_constant<int, 0>(), end_yz);  // YZ
                        if (PRIMS == PRIMS_ANY) {
                            for (; i < A.sphere_end; ++i, ++rec) { // plain spheres: sphere.rs:39-59, no switch, no wrapper
                                double t;
                                int aux;
                                if (prim_t<PRIMS_SPHERES>(load_prim_uniform(rec, 0), o, d, inv_d, inv_a, ray_time, 0.001, best_t, t, aux)) {
                                    best_t = t;
                                    best = i;
                                    best_aux = 0;
                                }
                            }
                            for (; i < KB->box_end; ++i, ++rec) { // boxes, bare or wrapped: box.rs:82-101 as three slabs, no switch
                                double t;
                                int side;
                                if (box_t(load_prim_uniform(rec, 0), o, d, inv_d, 0.001, best_t, t, side)) {
                                    best_t = t;
                                    best = i;
                                    best_aux = side;
                                }
                            }
                            for (; i < A.n_prims; ++i, ++rec) test(load_prim_uniform(rec, 0), i); // moving spheres, wrapped rects and spheres
                        }
                    } else {
                        const Prim *rec = A.prims;
                        int i = 0;
                        for (; i + 1 < A.n_prims; i += 2, rec += 2) {
                            const Prim pa = load_prim_uniform(rec, 0), pb = load_prim_uniform(rec, 1);
                            test(pa, i);
                            test(pb, i + 1);
                        }
                        if (i < A.n_prims) test(load_prim_uniform(rec, 0), i);
                    }
                }
                RT_REGION(3); // closest hit
                if (best < 0) { // background_color.rs:27-33 / :45-48
                    const RT_CONSTANT TraceArgs *K = kernargs_here();
                    d3 bgc = ld3(K->bg.top);
                    if (K->bg.kind == RT_BG_SKY) {
                        const double t = 0.5 * (unit_fast(d).y + 1.0);
                        bgc = (1.0 - t) * bgc + t * ld3(K->bg.bottom);
                    }
                    T = T * bgc;
                    ended = true;
                } else {
                    const Prim &P = BVH ? A.prims[best] : lds_prims[best];
                    const Material &M = P.mat;
                    const Hit h = prim_hit_record<PRIMS, TEXTURED, true>(P, o, d, ray_time, best_t, best_aux, M.needs_uv != 0);
                    const int kind = M.kind;
                    RT_REGION(8); // hit record
#ifdef RT_PROFILE_REGIONS
                    { // how many lanes of an iteration look up a Noise texture together?
                        const int n_noise = __popcll(ballot(TEXTURED && M.tex_kind == RT_TEX_NOISE));
                        if (n_noise > 0 && lane_rank(ballot(1)) == 0) {
                            rt_t_[35] += 1;
                            rt_t_[36] += (unsigned long long)n_noise;
                        }
                    }
#endif
                    // Texture::value once for every material that has one (light, Lambertian, Metal):
                    // one copy of the texture code, shared by the lanes of all three
                    // (variants without SPECULAR hold lights and Lambertians only: the host picks SPECULAR
                    // whenever a Metal or Dialectric exists)
                    if (!SPECULAR || kind != RT_MAT_DIELECTRIC) {
                        d3 tex;
                        if (!TEXTURED || M.tex_kind == RT_TEX_SOLID_COLOR) tex = ld3(M.color); // solid_color.rs:24-28
                        else tex = texture_value_deferred(A, BVH ? A.textures : lds_textures, M.texture, h.u, h.v, h.point, noise_tex, h.uv_approx,
                                                          // sphere.rs:20-27 in f64 from the outward normal (the face normal, un-flipped: exact)
                                                          [&] { return sphere_uv(h.front ? h.normal : -h.normal); });
                        // emission (the path ends) or attenuation, one copy for all three; a Noise colour arrives below
                        if (!TEXTURED || noise_tex < 0) T = T * tex;
                    }
                    RT_REGION(9); // texture, step 1
                    // With four arms, what every material does with the hit goes before the switch: a value
                    // assigned in one arm only costs every arm a copy where they meet (C2 -1.7 %; with the two
                    // arms of the other variants the same hoist costs C3 1 %).  A light ends the path, the point
                    // of a Noise light is read below; hit_normal and fuzz are Metal's, dead for the others.
                    const d3 d_in = d;
                    if (SPECULAR) {
                        o = h.point;
                        cand_base = 0;
                        hit_normal = h.normal;
                        fuzz = M.fuzz;
                    }
                    if (kind == RT_MAT_DIFFUSE_LIGHT) { // diffuse_light.rs:25-37
                        ended = true;
                        if (!SPECULAR && TEXTURED) o = h.point;
                    } else if (!SPECULAR || kind == RT_MAT_LAMBERTIAN) { // lambertian.rs:26-38 (direction below)
                        if (!SPECULAR) {
                            o = h.point;
                            cand_base = 0;
                        }
                        d = h.normal; // the incoming direction is dead: lambertian.rs:27 starts from the normal
                        is_lambert = true;
                        waiting = true;
                    } else if (SPECULAR && kind == RT_MAT_METAL) { // metal.rs:26-43
                        const d3 ud = unit_fast(d_in);
                        d = ud - (2.0 * dot(ud, h.normal)) * h.normal; // metal.rs:30 reflect(); the fuzz term follows below
                        is_lambert = false;
                        waiting = fuzz != 0.0; // fuzz 0 multiplies the sample by 0: its draws are dead
                        finish = !waiting;
                    } else { // dialectric.rs:25-55
                        const double ratio = h.front ? M.color[0] : M.ior; // 1 / ior, divided at upload
                        const d3 ud = unit_fast(d_in);
                        const double cos_theta = fmin(dot(-ud, h.normal), 1.0);
                        const double sin_theta = sqrt_fast(1.0 - cos_theta * cos_theta);
                        bool reflect_it = ratio * sin_theta > 1.0;
                        if (!reflect_it) { // the draw happens only when refraction is possible
                            const double r0 = h.front ? M.color[1] : M.color[2]; // ((1 - ratio) / (1 + ratio))^2, at upload
                            const double m = 1.0 - cos_theta;
                            const double m2 = m * m;
                            const double refl = r0 + (1.0 - r0) * (m2 * m2 * m);
                            const u4 b = rng.block(seg, RT_RNG_DIELECTRIC, 0);
                            reflect_it = refl > u53(b.a, b.b);
                        }
                        if (reflect_it) {
                            d = ud - (2.0 * dot(ud, h.normal)) * h.normal;
                        } else { // vec3.rs:416-422
                            const d3 perp = ratio * (ud + cos_theta * h.normal);
                            d = perp + (-sqrt_fast(fabs(1.0 - len2(perp)))) * h.normal;
                        }
                        scattered = true;
                    }
                }
            }
        }

        // ---- the wave evaluates the open rejection loops together (all 64 lanes arrive
        // here).  A request still open afterwards resumes next iteration, which costs its lane
        // one idle pass.  On cornell-like scenes (many requests, cheap iterations) two rounds
        // settle ~90 % and a third costs more than the idle lanes it saves; where an iteration
        // is expensive (textures, glass) or requests are few, up to four rounds pay: the loop
        // stops as soon as nothing is pending (measured: C2 +1.4 %, C4 +2.7 %, C3 -9 % with 3).
        RT_REGION(4); // miss / material
        if constexpr (TEXTURED) { // the Noise lookups of this iteration, by the whole wave (all 64 lanes arrive here)
            const bool lookup = noise_tex >= 0;
            if (ballot(lookup) != 0) {
                const Texture *tt = BVH ? A.textures : lds_textures;
                const double turb = coop_noise_turbulence(lookup, o, lookup ? tt[noise_tex].depth : 0,
                                                          lookup ? tt[noise_tex].perlin : 0, A, lds_perlin, lane, L.scratch.noise);
                if (lookup) {
                    const d3 tex = noise_colour(tt[noise_tex], o, turb);
                    T = T * tex; // DiffuseLight's emission (the path has ended) or Lambertian / Metal attenuation
                }
            }
        }
        RT_REGION(10); // Noise rounds
        d3 sph = mk(0.0, 0.0, 0.0); // (left uninitialised, three moves fewer per iteration cost the plain variants 16 bytes of scratch)
        if (coop_random_in_unit_sphere(waiting, rng.pixel, rng.sample, seg, cand_base, A.seed_lo, A.seed_hi, lane,
                                       L.scratch.req, (TEXTURED || SPECULAR) ? 4 : 2, sph)) {
            waiting = false;
            finish = true;
        }

        RT_REGION(5); // sampler
        if (finish) {
            if (is_lambert) { // lambertian.rs:27-33
                const d3 dir = d + unit_fast(sph); // d holds the normal since the hit
                // vec3.rs:127-130 near_zero keeps the normal: once in 10^23 samples, so the wave branches
                // around the six selects it would otherwise issue every time
                const bool near_zero = fabs(dir.x) < 1e-8 && fabs(dir.y) < 1e-8 && fabs(dir.z) < 1e-8;
                const d3 normal = d;
                d = dir;
                if (ballot(near_zero) != 0) {
                    asm volatile("; near_zero (keeps the compiler from turning the branch back into selects)");
                    if (near_zero) d = normal;
                }
                scattered = true;
            } else if (SPECULAR) { // metal.rs:31-42: d holds the reflected direction since the hit
                if (fuzz != 0.0) d = d + fuzz * sph;
                if (dot(d, hit_normal) < 0.0) {
                    T = mk(0.0, 0.0, 0.0);
                    ended = true;
                } else {
                    scattered = true;
                }
            }
        }
        // renderer.rs:48-55: the recursion's next level has depth 0 -> white
        if (scattered && (int)++seg >= A.max_depth) ended = true;
        if (alive && ended) { // vec3.rs:38-42 Color::add into the pixel's sum
            // (the scale is a power of two: its upper half lives in ONE scalar register across the loop — a scalar load and its
            // wait here, in every iteration, showed in the frame time)
            if (FIXED_SUMS) {
                const uint32_t scale_hi = sum_scale_hi;
                // T * scale + 2^52 (one fma: the product is exact, the sum rounds T to a multiple of 1 / scale) has the
                // integer in its mantissa; integer adds commute exactly (start_item has taken the exponent fields off)
                const double scale = __longlong_as_double((long long)((unsigned long long)scale_hi << 32));
                unsigned long long *sum = reinterpret_cast<unsigned long long *>(L.sum[spix]);
                atomicAdd(sum + 0, (unsigned long long)__double_as_longlong(fma(T.x, scale, 0x1p52)));
                atomicAdd(sum + 1, (unsigned long long)__double_as_longlong(fma(T.y, scale, 0x1p52)));
                atomicAdd(sum + 2, (unsigned long long)__double_as_longlong(fma(T.z, scale, 0x1p52)));
            } else {
                atomicAdd(&L.sum[spix][0], T.x);
                atomicAdd(&L.sum[spix][1], T.y);
                atomicAdd(&L.sum[spix][2], T.z);
            }
            alive = false;
        }
        RT_REGION(6); // scatter + accumulate
        // ---- ONE way out of the path loop, down here (a second `break`, or one inside an else-arm of the hand-out, had the
        // compiler merge the lanes' alive / waiting / material masks under the exec mask at two more joins: +20 scalar
        // instructions per iteration on a kernel whose scalar unit is busy 70 % of the time): an item event is due — the pool
        // has run dry and the other slot is free to take what is in flight of it, or the draining item's last path has ended.
        // ("Nothing in flight" is one of the two: a pool that is not dry feeds the lanes in the next iteration.)
        if (OVERLAP) {
            uint32_t st = state;
            asm volatile("" : "+s"(st)); // (opaque: `state` does not change in this loop)
            if ((st & (DRAINING | POOL_DRY)) != 0u) { // (one scalar test in the usual iteration)
                uint64_t go = 0ull; // (a scalar 64-bit value, not a bool: see the flags above)
                if ((st & DRAINING) != 0u) go = ballot(alive && (((uint32_t)spix ^ st) & CUR) != 0u);
                if (go == 0ull) break;
            }
        }
        }
    }
    RT_REGION(7); // item end
    RT_REGION_FLUSH
    unsigned long long total_segments = n_segments; // one atomic per wave for the statistic
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) total_segments += __shfl_down(total_segments, off, 64);
    if (lane == 0 && total_segments) atomicAdd(A.segments + RT_STAT_SEGMENTS, total_segments);
    unsigned long long started = n_started; // primary rays (RtRenderStats.samples)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) started += __shfl_down(started, off, 64);
    if (lane == 0 && started) atomicAdd(A.segments + RT_STAT_SAMPLES, started);
}

// vec3.rs:119-125 scale_sqrt over the owned rows: out = sqrt(sum over chunks / samples),
// chunks added in index order.  With step_x/step_y > 1 (preview renderer) every pixel takes
// the value of its block's top-left pixel and pixels outside the covered area are (0,0,0)
// (cpu_scaled.rs:50-52, :80-89).  `out` is the plain [height][width][3] frame (out_cols == 1) or the tile-column layout
// of the tile stream — column c of out_col_step pixels (the last takes the remainder, cpu.rs:97-109) stored as
// [height][column width][3] behind the columns before it — so that a tile of rt_render is one contiguous run.
__global__ __launch_bounds__(256) void k_resolve_chunks_f64(const double *__restrict__ partial, double *__restrict__ out,
                                                            int width, int height, int n_chunks, int slice_rows, int strip_rows,
                                                            int strip_count, int strip_index, int step_x, int step_y,
                                                            int cover_w, int cover_h, int out_col_step, int out_cols,
                                                            double scale) {
    // a slice holds the launch's OWNED rows only (TraceArgs.slice_rows): with strips the pass walks those rows and maps
    // each to its image row; the preview's slice rows are grid rows, read by every pixel of a block
    const size_t slice = (size_t)width * (size_t)slice_rows * 3;
    const size_t n = (size_t)width * (size_t)(strip_count > 1 ? slice_rows : height) * 3;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        size_t src = i, dst = i;
        if (strip_count > 1 || step_x > 1 || step_y > 1 || out_cols > 1) {
            const size_t pixel = i / 3;
            const int ch = (int)(i - pixel * 3);
            const int r = (int)(pixel / (size_t)width), col = (int)(pixel - (size_t)r * (size_t)width);
            int row = r;
            if (strip_count > 1) { // r is a row of the owned-row grid
                row = ((r / strip_rows) * strip_count + strip_index) * strip_rows + r % strip_rows;
                if (row >= height) continue;
            }
            dst = ((size_t)row * (size_t)width + (size_t)col) * 3 + ch;
            if (out_cols > 1) {
                int c = col / out_col_step;
                if (c > out_cols - 1) c = out_cols - 1;
                const int col_x = c * out_col_step;
                const int col_w = c == out_cols - 1 ? width - col_x : out_col_step;
                dst = ((size_t)height * (size_t)col_x + (size_t)row * (size_t)col_w + (size_t)(col - col_x)) * 3 + ch;
            }
            if (col >= cover_w || row >= cover_h) {
                out[dst] = 0.0;
                continue;
            }
            src = ((size_t)(r / step_y) * (size_t)width + (size_t)(col - col % step_x)) * 3 + ch;
        }
        double acc = 0.0;
        for (int c = 0; c < n_chunks; ++c) acc += partial[(size_t)c * slice + src];
        out[dst] = sqrt(scale * acc);
    }
}

} // namespace RT_KNS

#if defined(RT_BB_COUNT) && !defined(RT_EXACT_DIV)
// -DRT_BB_COUNT (tools/bb_build.sh, never the product): the basic-block counters that tools/bb_instrument.py's inserted
// instructions add to, and the host call that reads them.  Nothing in the C++ below touches the array on the device.
enum { RT_BB_MAX = 8192 };
extern "C" {
__device__ __attribute__((used)) unsigned long long rt_bb_counts[RT_BB_MAX];
int rtdev_bb_counts(unsigned long long *out, int n, int reset) {
    if (n > RT_BB_MAX) n = RT_BB_MAX;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(rt_bb_counts), (size_t)n * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        static unsigned long long zeros[RT_BB_MAX];
        if (hipMemcpyToSymbol(HIP_SYMBOL(rt_bb_counts), zeros, sizeof zeros) != hipSuccess) return -1;
    }
    return n;
}
}
#endif

namespace {
// One table entry per compiled variant: PRIMS x TEXTURED x SPECULAR with the
// linear closest-hit loop, plus PRIMS_ANY x TEXTURED x SPECULAR with the BVH.
template <int PRIMS, bool TEXTURED, bool SPECULAR, bool BVH> struct PoolVariant {
    static void launch(const rtdev::TraceArgs &a, unsigned blocks, hipStream_t stream) {
        const size_t dyn = (BVH ? (size_t)a.bvh_lds_nodes * sizeof(rtdev::BvhNode)
                                : (size_t)a.n_prims * sizeof(rtdev::Prim) + (TEXTURED ? (size_t)a.n_textures * sizeof(rtdev::Texture) : 0)) +
                           (TEXTURED && a.perlin_in_lds ? sizeof(double) * 256 * 3 : 0) +
                           (a.lens_lds ? rtdev::pool_lens_lds_bytes(BVH) : 0) + (a.time_lds ? rtdev::pool_time_lds_bytes(BVH) : 0);
        hipLaunchKernelGGL((RT_KNS::k_trace_pool_f64<PRIMS, TEXTURED, SPECULAR, BVH>), dim3(blocks), dim3(256), dyn, stream, a);
    }
    static int blocks_per_cu(size_t dyn_lds) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, RT_KNS::k_trace_pool_f64<PRIMS, TEXTURED, SPECULAR, BVH>, 256, dyn_lds) != hipSuccess)
            return 1;
        // The calculator divides the CU's 160 KB by the block's LDS bytes; the hardware hands LDS out in granules of 1280
        // bytes, 128 to a CU (tools/microbench/lds_fit.hip: 23 168 bytes per block already leave 6 blocks where the
        // calculator says 7).  A persistent grid one block per CU too large runs that block when the others are done.
        hipFuncAttributes attr;
        if (hipFuncGetAttributes(&attr, (const void *)RT_KNS::k_trace_pool_f64<PRIMS, TEXTURED, SPECULAR, BVH>) == hipSuccess) {
            const size_t granules = (attr.sharedSizeBytes + dyn_lds + 1279) / 1280;
            if (granules > 0 && (int)(128 / granules) < n) n = (int)(128 / granules);
        }
        return n < 1 ? 1 : n;
    }
};

// Calls F::template run<Variant>() for the variant the flags select.
template <class F> auto dispatch_variant(int prims_class, bool textured, bool specular, bool bvh, F f) {
    using namespace rtdev;
#define RT_PICK(P, B)                                                                           \
    (textured ? (specular ? f(PoolVariant<P, true, true, B>()) : f(PoolVariant<P, true, false, B>())) \
              : (specular ? f(PoolVariant<P, false, true, B>()) : f(PoolVariant<P, false, false, B>())))
    if (bvh) return RT_PICK(PRIMS_ANY, true);
    if (prims_class == PRIMS_RECTS) return RT_PICK(PRIMS_RECTS, false);
    if (prims_class == PRIMS_SPHERES) return RT_PICK(PRIMS_SPHERES, false);
    return RT_PICK(PRIMS_ANY, false);
#undef RT_PICK
}
} // namespace

// Resident blocks per CU of the variant (the persistent grid is CUs x this).
extern "C" int RT_LAUNCHER(rtdev_pool_blocks_per_cu)(int prims_class, int textured, int specular, int bvh, size_t dyn_lds) {
    return dispatch_variant(prims_class, textured != 0, specular != 0, bvh != 0,
                            [dyn_lds](auto v) { return decltype(v)::blocks_per_cu(dyn_lds); });
}

extern "C" hipError_t RT_LAUNCHER(rtdev_launch_trace_pool)(const rtdev::TraceArgs *args, int prims_class, int textured, int specular,
                                              int bvh, unsigned blocks, hipStream_t stream) {
    if (blocks == 0 || args->n_items == 0) return hipSuccess;
    dispatch_variant(prims_class, textured != 0, specular != 0, bvh != 0, [&](auto v) {
        decltype(v)::launch(*args, blocks, stream);
        return 0;
    });
    return hipGetLastError();
}

extern "C" hipError_t RT_LAUNCHER(rtdev_launch_resolve_chunks)(const double *partial, double *out, int width, int height, int n_chunks,
                                                               int slice_rows, int strip_rows, int strip_count, int strip_index, int step_x, int step_y,
                                                               int cover_w, int cover_h, int out_col_step, int out_cols, int samples,
                                                               hipStream_t stream) {
    if (out_cols <= 1 || out_col_step <= 0) {
        out_cols = 1;
        out_col_step = width;
    }
    size_t n = (size_t)width * (size_t)(strip_count > 1 ? slice_rows : height) * 3;
    unsigned blocks = (unsigned)((n + 255) / 256);
    if (blocks > 4096u) blocks = 4096u;
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(RT_KNS::k_resolve_chunks_f64, dim3(blocks), dim3(256), 0, stream, partial, out, width, height,
                       n_chunks, slice_rows, strip_rows, strip_count, strip_index, step_x, step_y, cover_w, cover_h, out_col_step, out_cols,
                       1.0 / (double)samples);
    return hipGetLastError();
}
