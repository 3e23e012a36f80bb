// session.cpp — the C ABI of include/rt_host.h over the C++ host classes.
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <string>
#include "../../include/rt_host.h"
#include "camera.h"
#include "config.h"
#include "error.h"
#include "image_action.h"
#include "image_io.h"
#include "scene.h"
#include "tone_map.h"

using namespace rthost;

struct RthSession {
    Config config;
    SceneLoadData data;
    std::unique_ptr<FlatScene> flat;
    std::unique_ptr<ToneMap> tone_map;
    int tone_map_kind = RTH_TONE_MAP_NONE;
    RtCamera camera;
    uint64_t seed = 1;
};

namespace {
thread_local std::string g_host_error;

template <class F> int guarded(F &&f) {
    try {
        g_host_error.clear();
        return f();
    } catch (const TracerError &e) {
        g_host_error = e.what();
        return e.code();
    } catch (const std::bad_alloc &) {
        g_host_error = "out of memory";
        return RT_ERR_OUT_OF_MEMORY;
    } catch (const std::exception &e) {
        g_host_error = e.what();
        return RT_ERR_INVALID_ARGUMENT;
    }
}

int tone_map_kind_of(const ToneMap &t) {
    std::string n = t.name();
    if (n == "Reinhard") return RTH_TONE_MAP_REINHARD;
    if (n == "Hable") return RTH_TONE_MAP_HABLE;
    if (n == "Aces") return RTH_TONE_MAP_ACES;
    return RTH_TONE_MAP_NONE;
}
} // namespace

extern "C" {

const char *rth_last_error_message(void) { return g_host_error.c_str(); }

int rth_session_open(const char *config_path, const char *scene_override, const char *image_action_override,
                     uint64_t seed, RthSession **out) {
    return guarded([&]() -> int {
        if (!config_path || !out) throw TracerError(RT_ERR_INVALID_ARGUMENT, "config_path/out is NULL");
        *out = nullptr;
        Args args;
        args.config = config_path;
        if (scene_override) args.scene = std::string(scene_override);
        if (image_action_override) args.image_action = image_action_from_str(image_action_override);
        args.seed = seed;
        auto s = std::make_unique<RthSession>();
        s->seed = seed;
        s->config = config_try_from(args);                        // main.rs:325
        s->data = make_loader(s->config.loader, seed)->load();    // main.rs:74-81
        // main.rs:84-86: the scene's tone map, else the config's
        s->tone_map = s->data.tone_map ? std::move(s->data.tone_map) : make_tone_map(s->config.tone_map);
        s->tone_map_kind = tone_map_kind_of(*s->tone_map);
        // main.rs:95-110
        CameraData cd = CameraData::merge(s->data.camera.value_or(CameraConfig()), s->config.camera);
        if (s->config.screen.width == 0 || s->config.screen.height == 0)
            throw TracerError::Configuration(config_path, "screen.width and screen.height must be positive");
        double aspect = (double)s->config.screen.width / (double)s->config.screen.height; // image.rs:10-17
        s->camera = camera_new(CameraInitData{cd.pos, cd.look_at, Vec3(0.0, 1.0, 0.0), cd.vfov, cd.aperture,
                                              cd.focus_distance, aspect, 0.0, 1.0});
        s->flat = flatten_scene(s->data);
        *out = s.release();
        return RT_OK;
    });
}

void rth_session_close(RthSession *s) { delete s; }

const RtSceneDesc *rth_session_scene(const RthSession *s) { return s ? &s->flat->desc : nullptr; }
const RtCamera *rth_session_camera(const RthSession *s) { return s ? &s->camera : nullptr; }

int rth_session_params(const RthSession *s, int preview, RtRenderParams *out) {
    return guarded([&]() -> int {
        if (!s || !out) throw TracerError(RT_ERR_INVALID_ARGUMENT, "session/out is NULL");
        const RenderConfig &r = preview ? s->config.preview : s->config.render;
        memset(out, 0, sizeof *out);
        out->width = (int32_t)s->config.screen.width;
        out->height = (int32_t)s->config.screen.height;
        out->samples = (int32_t)r.samples;
        out->max_depth = (int32_t)r.max_depth;
        out->tiles_w = (int32_t)r.num_threads_width;
        out->tiles_h = (int32_t)r.num_threads_height;
        out->seed = s->seed;
        // the preview renderer is CpuRendererScaled and reads `scale`; the final one is
        // CpuRenderer and ignores it (config.rs:203-207, renderer.rs:109-116)
        out->scale = preview ? (int32_t)r.scale : 0;
        return RT_OK;
    });
}

int rth_session_image_action(const RthSession *s) {
    return s && s->config.image_action == ImageActionConfig::SavePng ? RTH_IMAGE_ACTION_SAVE_PNG : RTH_IMAGE_ACTION_NONE;
}
int rth_session_tone_map_kind(const RthSession *s) { return s ? s->tone_map_kind : RTH_TONE_MAP_NONE; }
const char *rth_session_image_output_dir(const RthSession *s) {
    return s && s->config.image_output_dir ? s->config.image_output_dir->c_str() : nullptr;
}

int rth_session_tone_map(const RthSession *s, RtToneMap *out) {
    return guarded([&]() -> int {
        if (!s || !out) throw TracerError(RT_ERR_INVALID_ARGUMENT, "session/out is NULL");
        *out = s->tone_map->describe();
        return RT_OK;
    });
}

int rth_tone_map(const RthSession *s, const double *in, double *out, size_t n) {
    return guarded([&]() -> int {
        if (!s || !in || !out) throw TracerError(RT_ERR_INVALID_ARGUMENT, "NULL argument");
        for (size_t i = 0; i < n; ++i) { // image_buffer.rs:147-153
            Color c = s->tone_map->tone_map(Color(in[3 * i], in[3 * i + 1], in[3 * i + 2]));
            out[3 * i] = c.x();
            out[3 * i + 1] = c.y();
            out[3 * i + 2] = c.z();
        }
        return RT_OK;
    });
}

int rth_pack_rgba8(const double *rgb, size_t n, uint8_t *out) {
    return guarded([&]() -> int {
        if (!rgb || !out) throw TracerError(RT_ERR_INVALID_ARGUMENT, "NULL argument");
        pack_rgba8(rgb, n, out);
        return RT_OK;
    });
}

int rth_save_png(const RthSession *s, const double *rgb, int width, int height, const char *dir, char *out_path,
                 size_t cap) {
    return guarded([&]() -> int {
        if (!rgb || width <= 0 || height <= 0) throw TracerError(RT_ERR_INVALID_ARGUMENT, "empty frame");
        if (out_path && cap) out_path[0] = '\0';
        std::optional<std::string> d;
        if (dir) d = std::string(dir);
        else if (s && s->config.image_output_dir) d = s->config.image_output_dir;
        if (!d) return RT_OK; // png.rs:56-59: no output directory -> skip
        std::string path = save_png(rgb, width, height, *d);
        if (out_path && cap) {
            strncpy(out_path, path.c_str(), cap - 1);
            out_path[cap - 1] = '\0';
        }
        return RT_OK;
    });
}

int rth_camera_new(const double look_from[3], const double look_at[3], double vfov, double aperture,
                   double focus_distance, int width, int height, RtCamera *out) {
    return guarded([&]() -> int {
        if (!look_from || !look_at || !out || width <= 0 || height <= 0)
            throw TracerError(RT_ERR_INVALID_ARGUMENT, "bad camera arguments");
        *out = camera_new(CameraInitData{Vec3(look_from[0], look_from[1], look_from[2]),
                                         Vec3(look_at[0], look_at[1], look_at[2]), Vec3(0.0, 1.0, 0.0), vfov,
                                         aperture, focus_distance, (double)width / (double)height, 0.0, 1.0});
        return RT_OK;
    });
}

int rth_decode_image(const char *path, uint8_t **rgba, int *width, int *height) {
    return guarded([&]() -> int {
        if (!path || !rgba || !width || !height) throw TracerError(RT_ERR_INVALID_ARGUMENT, "NULL argument");
        std::vector<uint8_t> px;
        std::string why;
        if (!decode_image_rgba8(path, px, *width, *height, why)) throw TracerError::FailedToOpenImage(path, why);
        *rgba = (uint8_t *)malloc(px.size());
        if (!*rgba) throw std::bad_alloc();
        memcpy(*rgba, px.data(), px.size());
        return RT_OK;
    });
}

void rth_free(void *p) { free(p); }

int rth_sha256_hex(const uint8_t *data, size_t len, char out[65]) {
    return guarded([&]() -> int {
        if ((!data && len) || !out) throw TracerError(RT_ERR_INVALID_ARGUMENT, "NULL argument");
        std::string h = sha256_hex_upper(data, len);
        memcpy(out, h.c_str(), 65);
        return RT_OK;
    });
}

} // extern "C"
