// vec3.h — host-side Vec3 (racer-tracer/src/vec3.rs:13-16), the operators the
// host layer needs (camera set-up, tone maps, Perlin table).  `operator/`
// multiplies by the reciprocal like vec3.rs:279-301; unit_vector() divides
// like vec3.rs:79-85.
#pragma once
#include <cmath>

namespace rthost {

struct Vec3 {
    double pos[3] = {0.0, 0.0, 0.0};
    Vec3() = default;
    Vec3(double x, double y, double z) : pos{x, y, z} {}
    double x() const { return pos[0]; }
    double y() const { return pos[1]; }
    double z() const { return pos[2]; }
    double operator[](int i) const { return pos[i]; }
    double &operator[](int i) { return pos[i]; }
    double length_squared() const { return pos[0] * pos[0] + pos[1] * pos[1] + pos[2] * pos[2]; }
    double length() const { return std::sqrt(length_squared()); }
    Vec3 unit_vector() const {
        double len = length();
        return Vec3(pos[0] / len, pos[1] / len, pos[2] / len);
    }
    double dot(const Vec3 &v) const { return pos[0] * v.pos[0] + pos[1] * v.pos[1] + pos[2] * v.pos[2]; }
    Vec3 cross(const Vec3 &v) const {
        return Vec3(pos[1] * v.pos[2] - pos[2] * v.pos[1], pos[2] * v.pos[0] - pos[0] * v.pos[2],
                    pos[0] * v.pos[1] - pos[1] * v.pos[0]);
    }
};
using Color = Vec3;

inline Vec3 operator+(const Vec3 &a, const Vec3 &b) { return Vec3(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
inline Vec3 operator-(const Vec3 &a, const Vec3 &b) { return Vec3(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
inline Vec3 operator-(const Vec3 &a) { return Vec3(-a[0], -a[1], -a[2]); }
inline Vec3 operator*(const Vec3 &a, const Vec3 &b) { return Vec3(a[0] * b[0], a[1] * b[1], a[2] * b[2]); }
inline Vec3 operator*(const Vec3 &a, double s) { return Vec3(a[0] * s, a[1] * s, a[2] * s); }
inline Vec3 operator*(double s, const Vec3 &a) { return Vec3(a[0] * s, a[1] * s, a[2] * s); }
inline Vec3 operator+(const Vec3 &a, double s) { return Vec3(a[0] + s, a[1] + s, a[2] + s); }
inline Vec3 operator-(const Vec3 &a, double s) { return Vec3(a[0] - s, a[1] - s, a[2] - s); }
inline Vec3 operator/(const Vec3 &a, double s) { return (1.0 / s) * a; }

} // namespace rthost
