// rt_trace_kernel.hip — v1 of the per-pixel path-tracing kernel (f64): one
// lane owns one pixel and regenerates its next sample as soon as a path ends
// (see DESIGN.md 4.2).  Kept as the simple, order-exact reference form of the
// device path (per-pixel sums in the oracle's sample order) and for A/B
// measurements against the pooled kernel of rt_trace_pool_kernel.hip, which is
// the default.  Selected with RtSceneOptions.kernel = RT_KERNEL_V1 (rt_scene_create_ex).
#include "rt_trace_common.h"

namespace RT_KNS {

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
    double ray_time = 0.0; // ray.rs:26-28; scattered rays inherit it
    bool alive = false;
    unsigned int n_segments = 0, n_started = 0;

    while (s < A.sample_end) {
        if (!alive) {
            ++n_started;
            // cpu.rs:39-40 + camera.rs:326-337
            rng.sample = (uint32_t)s;
            u4 bc = rng.block(0, RT_RNG_CAMERA, 0);
            double v = ((double)py + u53(bc.a, bc.b)) / (double)(A.height - 1);
            d3 offset = mk(0.0, 0.0, 0.0);
            if (A.cam.lens_radius != 0.0) { // aperture 0: the disk is multiplied by 0 -> skip its draws
                double rx, ry;
                for (uint32_t i = 0;; ++i) { // util.rs:25-39
                    u4 b = rng.block(0, RT_RNG_LENS, i);
                    rx = sym53(b.a, b.b);
                    ry = sym53(b.c, b.d);
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
            ray_time = A.cam.time_a + (A.cam.time_b - A.cam.time_a) * u53(bc.c, bc.d); // camera.rs:335
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
            const d3 inv_d = rcp3(d);
            const double inv_a = PRIMS == PRIMS_RECTS ? 0.0 : rcp_f64(len2(d));
            for (int i = 0; i < A.n_prims; ++i) {
                double t;
                int aux;
                if (prim_t<PRIMS>(load_prim_uniform(A.prims, i), o, d, inv_d, inv_a, ray_time, 0.001, best_t, t, aux)) {
                    best_t = t;
                    best = i;
                    best_aux = aux;
                }
            }
            if (best < 0) { // background_color.rs:27-33 / :45-48
                d3 bgc = ld3(A.bg.top);
                if (A.bg.kind == RT_BG_SKY) {
                    double t = 0.5 * (unit_fast(d).y + 1.0);
                    bgc = (1.0 - t) * ld3(A.bg.top) + t * ld3(A.bg.bottom);
                }
                contrib = T * bgc;
                ended = true;
            } else {
                const Prim &P = A.prims[best];
                const Material &M = P.mat;
                Hit h = prim_hit_record<PRIMS, TEXTURED>(P, o, d, ray_time, best_t, best_aux, M.needs_uv != 0);
                if (M.kind == RT_MAT_DIFFUSE_LIGHT) { // diffuse_light.rs:25-37
                    contrib = T * texture_value<TEXTURED>(A, nullptr, A.textures, M, h.u, h.v, h.point);
                    ended = true;
                } else if (M.kind == RT_MAT_LAMBERTIAN) { // lambertian.rs:26-38
                    d3 dir = h.normal + unit_fast(random_in_unit_sphere(rng, seg));
                    if (fabs(dir.x) < 1e-8 && fabs(dir.y) < 1e-8 && fabs(dir.z) < 1e-8) dir = h.normal;
                    T = T * texture_value<TEXTURED>(A, nullptr, A.textures, M, h.u, h.v, h.point);
                    o = h.point;
                    d = dir;
                    ended = false;
                } else if (SPECULAR && M.kind == RT_MAT_METAL) { // metal.rs:26-43
                    d3 ud = unit_fast(d);
                    d3 dir = ud - (2.0 * dot(ud, h.normal)) * h.normal;
                    if (M.fuzz != 0.0) dir = dir + M.fuzz * random_in_unit_sphere(rng, seg);
                    if (dot(dir, h.normal) < 0.0) {
                        contrib = mk(0.0, 0.0, 0.0);
                        ended = true;
                    } else {
                        T = T * texture_value<TEXTURED>(A, nullptr, A.textures, M, h.u, h.v, h.point);
                        o = h.point;
                        d = dir;
                        ended = false;
                    }
                } else if (SPECULAR) { // dialectric.rs:25-55
                    double ratio = h.front ? 1.0 / M.ior : M.ior;
                    d3 ud = unit_fast(d);
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
    if (lane == 0 && total) atomicAdd(A.segments + RT_STAT_SEGMENTS, total);
    unsigned long long started = n_started; // primary rays (RtRenderStats.samples)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) started += __shfl_down(started, off, 64);
    if (lane == 0 && started) atomicAdd(A.segments + RT_STAT_SAMPLES, started);
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

} // namespace RT_KNS

namespace {
template <int PRIMS, bool TEXTURED, bool SPECULAR>
void launch_variant(const rtdev::TraceArgs &a, unsigned blocks, hipStream_t stream) {
    hipLaunchKernelGGL((RT_KNS::k_trace_f64<PRIMS, TEXTURED, SPECULAR>), dim3(blocks), dim3(256), 0, stream, a);
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
extern "C" hipError_t RT_LAUNCHER(rtdev_launch_trace)(const rtdev::TraceArgs *args, int prims_class, int textured,
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

extern "C" hipError_t RT_LAUNCHER(rtdev_launch_resolve)(const double *accum, double *out, int width, int height,
                                           int strip_rows, int strip_count, int strip_index, int samples,
                                           hipStream_t stream) {
    size_t n = (size_t)width * (size_t)height * 3;
    unsigned blocks = (unsigned)((n + 255) / 256);
    if (blocks > 2048u) blocks = 2048u;
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(RT_KNS::k_resolve_f64, dim3(blocks), dim3(256), 0, stream, accum, out, width, height,
                       strip_rows, strip_count, strip_index, 1.0 / (double)samples);
    return hipGetLastError();
}
