// main.cpp — `racer-tracer-amd`: headless stand-in for the reference binary
// with the same flags (racer-tracer/src/config.rs:12-28):
//     -c/--config <file.yml>   (default ./config.yml, env CONFIG)
//     -s/--scene  <file.yml|sandbox|random>
//     --image-action <png|none>
// plus --seed, --device and --devices N (the frame is sharded over N GPUs of this
// process the way the reference shards it over its rayon pool, cpu.rs:118-131).  The reference opens a window and renders when R
// is released (scene_controller/interactive.rs:83-86); this renders the final
// image once and exits, which is what `--image-action png` is for.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/rt_host.h"
#include "config.h"
#include "error.h"

namespace {

struct ScreenBuffer { // image_buffer.rs:104-170: tone-map each tile, keep the frame
    RthSession *session;
    int width, height;
    std::vector<double> buffer;
};

void on_tile(void *user, const double *rgb, int32_t r, int32_t c, int32_t w, int32_t h) {
    ScreenBuffer *sb = static_cast<ScreenBuffer *>(user);
    std::vector<double> mapped((size_t)w * (size_t)h * 3);
    rth_tone_map(sb->session, rgb, mapped.data(), (size_t)w * (size_t)h);
    for (int row = 0; row < h; ++row)
        memcpy(&sb->buffer[((size_t)(r + row) * (size_t)sb->width + (size_t)c) * 3], &mapped[(size_t)row * (size_t)w * 3],
               (size_t)w * 3 * sizeof(double));
}

} // namespace

int main(int argc, char **argv) {
    rthost::Args args;
    try {
        args = rthost::Args::parse(argc, argv);
    } catch (const rthost::TracerError &e) {
        fprintf(stderr, "%s\n", e.what());
        return e.code();
    }
    if (args.help) {
        printf("racer-tracer-amd [-c config.yml] [-s scene.yml|sandbox] [--image-action png|none] [--seed N] [--device N] [--devices N]\n");
        return 0;
    }
    RthSession *session = nullptr;
    const char *action = nullptr;
    if (args.image_action) action = *args.image_action == rthost::ImageActionConfig::SavePng ? "png" : "none";
    int rc = rth_session_open(args.config.c_str(), args.scene ? args.scene->c_str() : nullptr, action, args.seed, &session);
    if (rc != RT_OK) {
        fprintf(stderr, "%s\n", rth_last_error_message());
        return rc;
    }
    RtRenderParams params;
    rth_session_params(session, 0, &params);
    if (args.devices < 1) {
        fprintf(stderr, "--devices must be at least 1\n");
        rth_session_close(session);
        return RT_ERR_ARGUMENT_PARSING;
    }
    std::vector<RtScene *> scenes;
    auto destroy_scenes = [&] {
        for (RtScene *s : scenes) rt_scene_destroy(s);
    };
    for (int k = 0; k < args.devices; ++k) { // one upload of the (tiny) scene per device
        RtScene *scene = nullptr;
        rc = rt_scene_create(rth_session_scene(session), args.device + k, &scene);
        if (rc != RT_OK) {
            fprintf(stderr, "%s: %s\n", rt_strerror(rc), rt_last_error_message());
            destroy_scenes();
            rth_session_close(session);
            return rc;
        }
        scenes.push_back(scene);
    }
    ScreenBuffer sb{session, params.width, params.height, std::vector<double>((size_t)params.width * (size_t)params.height * 3, 0.0)};
    fprintf(stderr, "Rendering image...\n"); // interactive.rs:229
    auto t0 = std::chrono::steady_clock::now();
    // the reference's tile stream (cpu.rs:64-70): every finished tile goes through ScreenBuffer::update's tone map;
    // with several devices a tile column arrives once every device has finished its strips of it
    if (scenes.size() == 1) {
        rc = rt_render(scenes[0], rth_session_camera(session), &params, on_tile, &sb, nullptr);
    } else {
        rc = rt_render_multi(scenes.data(), (int)scenes.size(), rth_session_camera(session), &params, 0, on_tile, &sb, nullptr, nullptr);
    }
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (rc != RT_OK) {
        fprintf(stderr, "%s: %s\n", rt_strerror(rc), rt_last_error_message());
    } else {
        RtRenderStats st{};
        for (RtScene *s : scenes) { // every device's share
            RtRenderStats one;
            rt_scene_last_stats(s, &one);
            st.samples += one.samples;
            st.segments += one.segments;
            st.kernel_ms = one.kernel_ms > st.kernel_ms ? one.kernel_ms : st.kernel_ms;
        }
        fprintf(stderr, "It took %.3f seconds to render the image. (%.1f Msamples/s, %.2f segments/sample, kernel %.1f ms)\n",
                secs, (double)st.samples / secs / 1e6, st.samples ? (double)st.segments / (double)st.samples : 0.0, st.kernel_ms);
        if (rth_session_image_action(session) == RTH_IMAGE_ACTION_SAVE_PNG) { // main.rs:153-156
            char path[4096];
            fprintf(stderr, "Saving image...\n");
            rc = rth_save_png(session, sb.buffer.data(), params.width, params.height, nullptr, path, sizeof path);
            if (rc != RT_OK) fprintf(stderr, "%s\n", rth_last_error_message());
            else if (path[0]) fprintf(stderr, "Saved image to: %s\n", path);
            else fprintf(stderr, "No output directory for saving pngs. Skipping.\n");
        }
    }
    destroy_scenes();
    rth_session_close(session);
    return rc;
}
