// The CPU oracle (oracle/*.c) under -fsanitize=address,undefined: every shipped scene and the procedural one rendered at
// 32x18x2 through both closest-hit routines and several thread counts, strips, and every known-answer entry point of
// oracle.h called once (tests/test_oracle_sanitizers.py builds and runs this; sanitizers are CPU-only on this pool).
// The scenes come through the C++ host layer, which is compiled into the same sanitised binary.
// argv[1] = repository root.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "rt_host.h"
#include "../oracle/oracle.h"

static int g_bad = 0;
static void check(bool ok, const char *what) {
    if (!ok) {
        printf("FAILED: %s\n", what);
        ++g_bad;
    }
}
static bool all_finite(const std::vector<double> &v) {
    for (double x : v)
        if (!std::isfinite(x)) return false;
    return true;
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const std::string root = argv[1];
    const char *scenes[] = {"three_balls", "cornell_box", "noise_and_textures", "emissive", "clown", "two_balls", "cornell_box_boxes", "random"};
    for (const char *sc : scenes) {
        const std::string cfg = root + "/scenes/config_c2.yml";
        const std::string scene = std::string(sc) == "random" ? std::string("random") : root + "/scenes/" + sc + ".yml";
        RthSession *s = nullptr;
        int rc = rth_session_open(cfg.c_str(), scene.c_str(), nullptr, 3, &s);
        if (rc) {
            printf("%s: session rc=%d %s\n", sc, rc, rth_last_error_message());
            ++g_bad;
            continue;
        }
        const RtSceneDesc *d = rth_session_scene(s);
        const RtCamera *cam = rth_session_camera(s);
        RtRenderParams p;
        rth_session_params(s, 0, &p);
        p.width = 32, p.height = 18, p.samples = 2, p.max_depth = 6;
        std::vector<double> a((size_t)p.width * p.height * 3, -1.0), b(a), c(a);
        uint64_t seg_a = 0, seg_b = 0;
        rc = orc_render(d, cam, &p, 2, 0, a.data(), &seg_a);       // linear scan, two threads
        check(rc == 0 && all_finite(a) && seg_a > 0, "orc_render (linear)");
        rc = orc_render(d, cam, &p, 0, 1, b.data(), &seg_b);       // the reference-shaped BVH, all cores
        check(rc == 0 && all_finite(b) && seg_b > 0, "orc_render (bvh)");
        RtRenderParams strip = p;                                  // rows of one share only
        strip.strip_rows = 4, strip.strip_count = 3, strip.strip_index = 1;
        rc = orc_render(d, cam, &strip, 1, 0, c.data(), nullptr);
        check(rc == 0, "orc_render (strips)");
        for (int r = 0; r < p.height; ++r)
            for (int k = 0; k < p.width * 3; ++k) {
                const double got = c[(size_t)r * p.width * 3 + k];
                const bool own = (r / 4) % 3 == 1;
                if (own ? got != a[(size_t)r * p.width * 3 + k] : got != -1.0) {
                    check(false, "strip rows equal the whole frame's, other rows untouched");
                    r = p.height;
                    break;
                }
            }
        // single samples through a prebuilt scene, both routines
        for (int use_bvh = 0; use_bvh < 2; ++use_bvh) {
            OrcScene *os = orc_scene_build(d, use_bvh, p.seed);
            double rad[3];
            int nseg = 0;
            orc_sample_radiance(d, os, cam, &p, 7, 5, 1, rad, &nseg);
            check(std::isfinite(rad[0] + rad[1] + rad[2]) && nseg >= 0, "orc_sample_radiance");
            orc_sample_radiance_u(d, os, cam, &p, 7, 5, 1, orc_pixel_u(&p, 7, 5), rad, &nseg);
            OrcHit h;
            const double o[3] = {cam->origin[0], cam->origin[1], cam->origin[2]};
            const double dir[3] = {cam->forward[0], cam->forward[1], cam->forward[2]};
            (void)orc_scene_hit(os, o, dir, 0.001, INFINITY, &h);
            (void)orc_scene_hit_time(os, o, dir, 0.5, 0.001, INFINITY, &h);
            orc_scene_free(os);
        }
        // every primitive on its own, its box, and every texture
        for (int i = 0; i < d->n_primitives && i < 64; ++i) {
            OrcHit h;
            const double o[3] = {cam->origin[0], cam->origin[1], cam->origin[2]};
            const double dir[3] = {d->primitives[i].p[0] - o[0], d->primitives[i].p[1] - o[1], d->primitives[i].p[2] - o[2]};
            (void)orc_hit_primitive(&d->primitives[i], o, dir, 0.001, INFINITY, &h);
            (void)orc_hit_primitive_time(&d->primitives[i], o, dir, 0.25, 0.001, INFINITY, &h);
            double mn[3], mx[3];
            orc_primitive_aabb(&d->primitives[i], mn, mx);
            (void)orc_aabb_hit(mn, mx, o, dir, 0.001, INFINITY);
        }
        for (int t = 0; t < d->n_textures; ++t) {
            const double pt[3] = {1.25, -0.5, 3.0};
            double col[3];
            orc_texture_value(d, t, 0.3, 0.9, pt, col);
            check(std::isfinite(col[0] + col[1] + col[2]), "orc_texture_value");
            orc_texture_value(d, t, -2.0, 7.0, pt, col); // clamped u, v
        }
        for (int k = 0; k < d->n_perlins; ++k) {
            const double pt[3] = {-3.7, 100.25, 0.0};
            check(std::isfinite(orc_perlin_noise(&d->perlins[k], pt)) && std::isfinite(orc_perlin_turbulence(&d->perlins[k], pt, 7)), "perlin");
        }
        const double up[3] = {0, 1, 0};
        double bg[3];
        orc_background_color(&d->background, up, bg);
        printf("%s ok: %d primitives, %llu / %llu segments\n", sc, d->n_primitives, (unsigned long long)seg_a, (unsigned long long)seg_b);
        rth_session_close(s);
    }
    // ---- known-answer entry points
    {
        const uint32_t ctr[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, key[2] = {0xffffffffu, 0xffffffffu};
        uint32_t out[4];
        orc_philox4x32(ctr, key, 10, out);
        check(out[0] == 0x408f276du && out[1] == 0x41c83b0eu && out[2] == 0xa20bc7c6u && out[3] == 0x6d5451fdu, "Random123 philox4x32-10 vector");
        orc_philox4x32(ctr, key, 7, out);
        const double d0 = orc_rng_double(1, 2, 3, 4, RT_RNG_CAMERA, 0, 0), d1 = orc_rng_double(1, 2, 3, 4, RT_RNG_CAMERA, 0, 1);
        check(d0 >= 0 && d0 < 1 && d1 >= 0 && d1 < 1, "orc_rng_double in [0, 1)");
        double e[3];
        orc_rng_triple(9, 8, 7, 6, RT_RNG_SCATTER, 5, e);
        check(e[0] >= 0 && e[0] < 1 && e[1] >= 0 && e[1] < 1 && e[2] >= 0 && e[2] < 1, "orc_rng_triple in [0, 1)");
        const double a[3] = {1, 2, 3}, b[3] = {4, 5, 6};
        double r[3];
        orc_vec3_add(a, b, r);
        check(r[0] == 5 && r[1] == 7 && r[2] == 9, "vec3 add (vec3.rs:446-503)");
        orc_vec3_sub(a, b, r);
        orc_vec3_mul(a, b, r);
        orc_vec3_scale(a, 2.0, r);
        orc_vec3_div(a, 2.0, r);
        check(r[0] == 0.5 && r[1] == 1.0 && r[2] == 1.5, "vec3 div");
        const double n[3] = {0, 1, 0}, v[3] = {0.6, -0.8, 0};
        orc_reflect(v, n, r);
        check(r[0] == 0.6 && r[1] == 0.8, "reflect");
        orc_refract(v, n, 1.0 / 1.5, r);
        check(std::isfinite(r[0] + r[1] + r[2]) && orc_schlick(0.5, 1.5) > 0, "refract / schlick");
        double u, vv;
        orc_sphere_uv(n, &u, &vv);
        check(u >= 0 && u <= 1 && vv >= 0 && vv <= 1, "sphere_uv");
        RtCamera cam;
        const double from[3] = {13, 2, 3}, at[3] = {0, 0, 0}, up[3] = {0, 1, 0};
        orc_camera_new(from, at, up, 20.0, 0.1, 10.0, 16.0 / 9.0, 0.0, 1.0, &cam);
        check(std::isfinite(cam.horizontal[0]) && cam.lens_radius == 0.05, "camera_new");
        int32_t tiles[4 * 200];
        check(orc_tile_grid(1920, 1080, 10, 10, tiles, 200) == 100 && orc_tile_grid(7, 5, 3, 2, tiles, 200) == 6, "tile grid (cpu.rs:73-115)");
        check(orc_tile_grid(1920, 1080, 10, 10, tiles, 5) <= 100, "tile grid with a short output array");
        check(orc_online_cores() >= 1, "online cores");
        std::vector<double> rgb((size_t)16 * 3), mapped(rgb.size());
        for (size_t i = 0; i < rgb.size(); ++i) rgb[i] = (double)i * 0.37 - 0.5; // negative, > 1 and huge values included
        rgb[5] = 1e30;
        for (int kind = ORC_TM_NONE; kind <= ORC_TM_ACES; ++kind) {
            OrcToneMap tm;
            orc_tone_map_defaults(kind, &tm);
            orc_tone_map_apply(&tm, rgb.data(), mapped.data(), 16);
            uint8_t bytes[16 * 4];
            orc_pack_rgba8(mapped.data(), 16, bytes);
            check(bytes[3] == 255, "pack rgba8 alpha");
        }
    }
    printf(g_bad ? "oracle sanitize run: %d check(s) failed\n" : "oracle sanitize run: all checks passed\n", g_bad);
    return g_bad ? 1 : 0;
}
