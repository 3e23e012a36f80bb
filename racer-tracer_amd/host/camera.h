// camera.h — CameraData::merge and Camera::new
// (racer-tracer/src/camera.rs:403-464, :196-234).  `get_ray` (camera.rs:326-337)
// runs inside the kernel; the host only builds the 14 shared fields.
#pragma once
#include "../../include/rt_abi.h"
#include "config.h"
#include "vec3.h"

namespace rthost {

struct CameraData { // camera.rs:393-401
    double vfov, aperture, focus_distance;
    Vec3 pos, look_at;
    double speed, sensitivity;
    // scene values win over config values; then the defaults of camera.rs:437-463
    static CameraData merge(const CameraConfig &data1, const CameraConfig &data2);
};

struct CameraInitData { // camera.rs:180-190
    Vec3 look_from, look_at, scene_up;
    double vfov, aperture, focus_distance, aspect_ratio, time_a, time_b;
};

RtCamera camera_new(const CameraInitData &init); // camera.rs:196-234

} // namespace rthost
