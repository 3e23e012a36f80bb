#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include "rt_host.h"
// Exercises the C++ host layer (YAML, scene loader, JPEG decode, tone map, PNG + SHA-256, the
// procedural scene, error paths) in a -fsanitize=address,undefined build; tests/test_host_sanitizers.py
// compiles and runs it.  argv[1] = repository root, argv[2] = scratch directory.
int main(int argc, char **argv) {
    if (argc < 3) return 2;
    const char *root = argv[1];
    const char *scratch = argv[2];
    const char *scenes[] = {"three_balls", "cornell_box", "noise_and_textures", "emissive", "clown", "two_balls", "cornell_box_boxes"};
    for (const char *sc : scenes) {
        std::string cfg = std::string(root) + "/scenes/config_c3.yml", scene = std::string(root) + "/scenes/" + sc + ".yml";
        RthSession *s = nullptr;
        int rc = rth_session_open(cfg.c_str(), scene.c_str(), nullptr, 1, &s);
        printf("%s rc=%d %s\n", sc, rc, rc ? rth_last_error_message() : "");
        if (rc) continue;
        const RtSceneDesc *d = rth_session_scene(s);
        printf("  prims %d mats %d tex %d img %d perlin %d\n", d->n_primitives, d->n_materials, d->n_textures, d->n_images, d->n_perlins);
        RtRenderParams p;
        rth_session_params(s, 0, &p);
        std::vector<double> rgb((size_t)64 * 36 * 3, 0.5), out(rgb.size());
        rth_tone_map(s, rgb.data(), out.data(), 64 * 36);
        char hex[512];
        rc = rth_save_png(s, out.data(), 64, 36, scratch, hex, sizeof hex);
        printf("  png rc=%d\n", rc);
        rth_session_close(s);
    }
    { // procedural scene
        std::string cfg = std::string(root) + "/scenes/config_c1.yml";
        RthSession *s = nullptr;
        int rc = rth_session_open(cfg.c_str(), "random", nullptr, 7, &s);
        printf("random rc=%d prims %d\n", rc, rc ? 0 : rth_session_scene(s)->n_primitives);
        if (!rc) rth_session_close(s);
    }
    // error paths
    RthSession *s = nullptr;
    printf("missing config rc=%d\n", rth_session_open("/nonexistent.yml", nullptr, nullptr, 1, &s));
    printf("missing scene rc=%d\n", rth_session_open((std::string(root) + "/scenes/config_c3.yml").c_str(), "/nonexistent_scene.yml", nullptr, 1, &s));
    uint8_t *px = nullptr; int w = 0, h = 0;
    int rc = rth_decode_image((std::string(root) + "/resources/images/earthmap.jpg").c_str(), &px, &w, &h);
    printf("decode jpg rc=%d %dx%d\n", rc, w, h);
    rth_free(px);
    printf("decode garbage rc=%d\n", rth_decode_image((std::string(root) + "/README.md").c_str(), &px, &w, &h));
    return 0;
}
