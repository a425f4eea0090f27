// f64 vector arithmetic for gfx950 device code, with the operation order of the
// reference's Vec3 (src/vec3.rs). Built with -ffp-contract=off: every + - * / sqrt
// below is one IEEE binary64 operation, so results are bit-identical to the CPU's.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtd {

#define RT_D __device__ __forceinline__

struct V3 {
    double x, y, z;
};

RT_D V3 v3(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_D V3 v3p(const double* p) { return v3(p[0], p[1], p[2]); }

// Tables staged in LDS are reached through address-space-3 pointers, so that the compiler emits ds_read (which
// only waits on the LDS counter) instead of flat loads (which wait on the global-memory counter as well).
#define RT_LDS __attribute__((address_space(3)))
RT_D V3 v3p(const RT_LDS double* p) { return v3(p[0], p[1], p[2]); }
RT_D V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }        // src/vec3.rs:88-98
RT_D V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }        // :132-141
RT_D V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }                             // :241-250
RT_D V3 operator*(V3 a, double s) { return v3(a.x * s, a.y * s, a.z * s); }          // :181-190
RT_D V3 operator/(V3 a, double s) { return v3(a.x / s, a.y / s, a.z / s); }          // :212-221
RT_D double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }            // :174-179 (`Vec3 * Vec3`)
RT_D V3 elemul(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }           // :224-230
RT_D V3 cross(V3 a, V3 b) {                                                          // :232-238
    return v3(a.y * b.z - a.z * b.y, b.x * a.z - a.x * b.z, a.x * b.y - a.y * b.x);
}
RT_D double squared_length(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }       // :27-29
RT_D double length(V3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }         // :277-279
RT_D V3 at(V3 o, V3 d, double t) { return o + d * t; }                               // Ray::at, src/ray.rs:10-12
RT_D double comp(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

RT_D V3 reflect(V3 v, V3 n) { return v - n * 2.0 * dot(v, n); }                      // :77-79
RT_D V3 refract(V3 uv, V3 n, double etai_over_etat) {                                // :81-86
    const double cos_theta = dot(-uv, n);
    const V3 r_out_prep = (uv + n * cos_theta) * etai_over_etat;
    const V3 r_out_parallel = n * (-sqrt(fabs(1.0 - squared_length(r_out_prep))));
    return r_out_prep + r_out_parallel;
}

}  // namespace rtd
