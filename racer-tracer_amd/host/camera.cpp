// camera.cpp — see camera.h.
#include "camera.h"
#include <cmath>
#include <cstring>

namespace rthost {

CameraData CameraData::merge(const CameraConfig &a, const CameraConfig &b) { // camera.rs:403-435
    CameraData d;
    d.vfov = a.vfov ? *a.vfov : b.vfov.value_or(20.0);
    d.aperture = a.aperture ? *a.aperture : b.aperture.value_or(0.0);
    d.focus_distance = a.focus_distance ? *a.focus_distance : b.focus_distance.value_or(1000.0);
    d.pos = a.pos ? *a.pos : b.pos.value_or(Vec3(0.0, 0.0, 0.0));
    d.look_at = a.look_at ? *a.look_at : b.look_at.value_or(Vec3(0.0, 0.0, -1.0));
    d.speed = a.speed ? *a.speed : b.speed.value_or(0.0002);
    d.sensitivity = a.sensitivity ? *a.sensitivity : b.sensitivity.value_or(0.001);
    return d;
}

RtCamera camera_new(const CameraInitData &init) { // camera.rs:196-234
    const double PI = 3.14159265358979323846264338327950288;
    double h = std::tan((init.vfov * PI / 180.0) / 2.0); // util.rs:5-7
    double viewport_height = 2.0 * h;
    double viewport_width = init.aspect_ratio * viewport_height;

    Vec3 forward = (init.look_from - init.look_at).unit_vector();
    Vec3 right = init.scene_up.cross(forward).unit_vector();
    Vec3 up = forward.cross(right);

    Vec3 horizontal = init.focus_distance * viewport_width * right;
    Vec3 vertical = init.focus_distance * viewport_height * up;
    Vec3 upper_left_corner = init.look_from + vertical / 2.0 - horizontal / 2.0 - init.focus_distance * forward;

    RtCamera c;
    memset(&c, 0, sizeof c);
    for (int k = 0; k < 3; ++k) {
        c.origin[k] = init.look_from[k];
        c.upper_left_corner[k] = upper_left_corner[k];
        c.forward[k] = forward[k];
        c.right[k] = right[k];
        c.up[k] = up[k];
        c.horizontal[k] = horizontal[k];
        c.vertical[k] = vertical[k];
    }
    c.vfov = init.vfov;
    c.viewport_width = viewport_width;
    c.viewport_height = viewport_height;
    c.lens_radius = init.aperture * 0.5;
    c.focus_distance = init.focus_distance;
    c.time_a = init.time_a;
    c.time_b = init.time_b;
    return c;
}

} // namespace rthost
