// TEST INFRASTRUCTURE ONLY — CPU oracle for the per-pixel ray_color path.
//
// A C++ restatement of the reference's algorithm (Mighty-A/RayTracer, Rust), f64
// throughout, compiled with -ffp-contract=off so that + - * / sqrt round exactly
// as the Rust binary's do. Every function cites the reference lines it follows.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
// this; the product (raytracer_amd/) never includes, links or calls it.
//
// What pins it. The Rust toolchain is absent, so the binary cannot be run, and the reference's only tests are
// the 21 Vec3 operator tests (src/vec3.rs:297-441). Pinned: those tests; the resolve known answer
// (0.52,0.80,0.92) -> [183,228,244] (src/main.rs:153,216-224); and, end to end, the one output of the binary the
// reference holds: img/kaleidoscope.png, its scene 11 at 800x800, which a render of this oracle reproduces to
// 0.8 levels RMSE after 4x4 averaging (camera, row convention, Box6, RotateY, Translate, Triangle, Metal,
// background, depth-50 recursion, resolve). PARITY UNPINNED for what that picture does not exercise: Sphere,
// MovingSphere, the rects on their own, ConstantMedium, Lambertian / Dielectric / DiffuseLight / Isotropic, the
// textures and Perlin noise, and the BVH over more than four objects - the reference has no test, fixture or
// output for them; trust there is source-following only.
//
// Deliberate deviations from the reference, identical in the HIP path:
//  * RNG: counter-based Philox keyed by (pixel, sample, bounce, stream) instead
//    of thread_rng (philox.hpp). random_int follows the "Ray Tracing: The Next
//    Week" definition int(random_double(min, max+1)).
//  * MovingSphere::hit and ConstantMedium::hit leave rec.u/rec.v stale in the
//    reference (src/hittable.rs:226-232, src/constant_medium.rs:75-87); here
//    they are set to 0.
//  * Vec3::unit() of a zero vector panics in the reference (src/vec3.rs:285-286);
//    here scatter treats such a path as absorbed.
//  * aabb_mode TIGHT (what the GPU runs) replaces the reference's three
//    independent 1-D overlap tests (src/aabb.rs:15-30) by a carried slab test
//    on boxes rounded outward to f32 (FBox::hit); REFERENCE mode keeps
//    src/aabb.rs as is. Both give the same closest hit whenever boxes bound
//    their contents.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <memory>
#include <stdexcept>
#include <vector>

#include "lbvh.hpp"
#include "philox.hpp"

namespace orc {

constexpr double INF = std::numeric_limits<double>::infinity();  // src/rtweekend.rs:3
constexpr double PI = 3.14159265358979323846264338327950288;      // src/rtweekend.rs:4
constexpr uint32_t NONE = 0xFFFFFFFFu;

enum AabbMode : int { AABB_REFERENCE = 0, AABB_TIGHT = 1, AABB_PER_AXIS_F32 = 2 };

enum PrimKind : int { K_SPHERE = 0, K_MSPHERE, K_RECT, K_BOX6, K_TRIANGLE, K_MEDIUM, K_NKINDS };

struct RayRecord;  // harvested rays, see api.cpp

struct Tls {
    RngCtx rng;
    uint32_t pixel = 0, sample = 0, bounce = 0;
    int aabb_mode = AABB_REFERENCE;
    double carried_t0 = -std::numeric_limits<double>::infinity(), carried_t1 = std::numeric_limits<double>::infinity();  // SceneCtx's
    uint64_t n_box = 0;              // BVHNode box tests
    uint64_t n_prim[K_NKINDS] = {};  // primitive hit() calls by kind
    uint64_t n_rays = 0;             // world.hit queries issued by ray_color
    bool in_boundary = false;        // inside the boundary queries of a ConstantMedium
    bool deferred_pass = false;      // World::hit is putting the deferred media to the ray (GPU modes)
    std::vector<RayRecord>* recorder = nullptr;
    uint32_t record_max_bounce = 0;
};
inline thread_local Tls tls;

// ---------------------------------------------------------------- rtweekend.rs
inline double degrees_to_radians(double degrees) { return degrees * PI / 180.0; }  // :6-8
inline double random_double(double min, double max) {                              // :11-14
    return min + (max - min) * tls.rng.next_unit();
}
inline int random_int(int min, int max) {  // :17-20 ([min, max])
    return (int)random_double((double)min, (double)(max + 1));
}
inline double clamp(double x, double min, double max) {  // :22-30
    if (x < min) return min;
    if (x > max) return max;
    return x;
}

// --------------------------------------------------------------------- vec3.rs
struct Vec3 {
    double x, y, z;
    Vec3() : x(0), y(0), z(0) {}
    Vec3(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
    static Vec3 ones() { return Vec3(1.0, 1.0, 1.0); }
    static Vec3 zero() { return Vec3(0.0, 0.0, 0.0); }
    double squared_length() const { return x * x + y * y + z * z; }     // :27-29
    double length() const { return std::sqrt(x * x + y * y + z * z); }  // :277-279
    static Vec3 random(double min, double max) {                        // :31-37
        double a = random_double(min, max);
        double b = random_double(min, max);
        double c = random_double(min, max);
        return Vec3(a, b, c);
    }
    double operator[](int i) const { return i == 1 ? y : (i == 2 ? z : x); }  // :252-262
    double& at(int i) {                                                       // :264-275
        if (i == 0) return x;
        if (i == 1) return y;
        if (i == 2) return z;
        throw std::out_of_range("Vec3 index");
    }
    Vec3 unit() const {  // :283-294 (panics on zero length)
        double len = length();
        if (len == 0.0) throw std::domain_error("unit() of zero vector");
        return Vec3(x / len, y / len, z / len);
    }
    static Vec3 elemul(Vec3 a, Vec3 b) { return Vec3(a.x * b.x, a.y * b.y, a.z * b.z); }  // :224-230
    static Vec3 cross(Vec3 a, Vec3 b) {                                                   // :232-238
        return Vec3(a.y * b.z - a.z * b.y, b.x * a.z - a.x * b.z, a.x * b.y - a.y * b.x);
    }
};
using Color = Vec3;
using Point = Vec3;
inline Vec3 operator+(Vec3 a, Vec3 b) { return Vec3(a.x + b.x, a.y + b.y, a.z + b.z); }  // :88-98
inline Vec3 operator+(Vec3 a, double b) { return Vec3(a.x + b, a.y + b, a.z + b); }      // :100-110
inline Vec3 operator-(Vec3 a, Vec3 b) { return Vec3(a.x - b.x, a.y - b.y, a.z - b.z); }  // :132-141
inline Vec3 operator-(Vec3 a, double b) { return Vec3(a.x - b, a.y - b, a.z - b); }      // :143-152
inline double operator*(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }    // :174-179 (dot!)
inline Vec3 operator*(Vec3 a, double b) { return Vec3(a.x * b, a.y * b, a.z * b); }      // :181-190
inline Vec3 operator/(Vec3 a, double b) { return Vec3(a.x / b, a.y / b, a.z / b); }      // :212-221
inline Vec3 operator-(Vec3 a) { return Vec3(-a.x, -a.y, -a.z); }                         // :241-250
inline Vec3& operator+=(Vec3& a, Vec3 b) { a = a + b; return a; }                        // :112-120
inline Vec3& operator+=(Vec3& a, double b) { a = a + b; return a; }                      // :122-130
inline Vec3& operator-=(Vec3& a, Vec3 b) { a = a - b; return a; }                        // :154-162
inline Vec3& operator-=(Vec3& a, double b) { a = a - b; return a; }                      // :164-172
inline Vec3& operator*=(Vec3& a, Vec3 b) { a = Vec3::elemul(a, b); return a; }           // :192-200
inline Vec3& operator*=(Vec3& a, double b) { a = a * b; return a; }                      // :202-210
inline bool operator==(Vec3 a, Vec3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

inline Vec3 random_in_unit_sphere() {  // :40-48
    for (;;) {
        Vec3 p = Vec3::random(-1.0, 1.0);
        if (p.squared_length() >= 1.0) continue;
        return p;
    }
}
inline Vec3 random_unit_vector() {  // :50-55
    double a = random_double(0.0, 2.0 * PI);
    double z = random_double(-1.0, 1.0);
    double r = std::sqrt(1.0 - z * z);
    return Vec3(r * std::cos(a), r * std::sin(a), z);
}
inline Vec3 random_in_unit_disk() {  // :67-75
    for (;;) {
        double a = random_double(-1.0, 1.0);
        double b = random_double(-1.0, 1.0);
        Vec3 p(a, b, 0.0);
        if (p.squared_length() >= 1.0) continue;
        return p;
    }
}
inline Vec3 reflect(const Vec3& v, const Vec3& n) { return v - n * 2.0 * (v * n); }  // :77-79
inline Vec3 refract(Vec3 uv, Vec3 n, double etai_over_etat) {                        // :81-86
    double cos_theta = (-uv) * n;
    Vec3 r_out_prep = (uv + n * cos_theta) * etai_over_etat;
    Vec3 r_out_parallel = n * (-std::sqrt(std::fabs(1.0 - r_out_prep.squared_length())));
    return r_out_prep + r_out_parallel;
}

// ---------------------------------------------------------------------- ray.rs
struct Ray {  // :3-7
    Vec3 orig, dire;
    double tm = 0.0;
    Vec3 at(double t) const { return orig + dire * t; }  // :10-12
};

// --------------------------------------------------------------------- aabb.rs
inline float f32_down(double v) {
    float f = (float)v;
    if ((double)f > v) f = std::nextafterf(f, -std::numeric_limits<float>::infinity());
    return f;
}
inline float f32_up(double v) {
    float f = (float)v;
    if ((double)f < v) f = std::nextafterf(f, std::numeric_limits<float>::infinity());
    return f;
}

struct AABB {  // :5-8
    Point _min, _max;
    AABB() : _min(Point::ones()), _max(Point::ones()) {}
    AABB(const Point& a, const Point& b) : _min(a), _max(b) {}

    // :15-30, the reference's test: three independent 1-D overlaps, tmin/tmax NOT carried.
    bool hit(const Ray& r, double tmin, double tmax) const {
        for (int a = 0; a < 3; ++a) {
            double inv_d = 1.0 / r.dire[a];
            double t0 = (_min[a] - r.orig[a]) * inv_d;
            double t1 = (_max[a] - r.orig[a]) * inv_d;
            if (inv_d < 0.0) std::swap(t0, t1);
            t0 = t0 > tmin ? t0 : tmin;
            t1 = t1 < tmax ? t1 : tmax;
            if (t1 <= t0) return false;
        }
        return true;
    }
};

// Box as the flattened BVH stores it: f32, rounded outward; a box that is flat on
// some axis in f64 (never hit by src/aabb.rs:25 for finite slopes) is stored
// inverted so that no ray passes it.
struct FBox {
    float lo[3], hi[3];
    static FBox from(const AABB& b) {
        FBox f;
        bool flat = false;
        for (int a = 0; a < 3; ++a) {
            if (b._min[a] == b._max[a]) flat = true;
            f.lo[a] = f32_down(b._min[a]);
            f.hi[a] = f32_up(b._max[a]);
        }
        if (flat)
            for (int a = 0; a < 3; ++a) {
                f.lo[a] = std::numeric_limits<float>::infinity();
                f.hi[a] = -std::numeric_limits<float>::infinity();
            }
        return f;
    }
    // The "tight" rule shared bit for bit with the HIP kernels: near/far plane picked by the sign of 1/d (the
    // reference's swap, src/aabb.rs:20-22), t = fma(plane, 1/d, -(o * 1/d)) per plane, the interval carried across
    // the three axes with NaN terms ignored, pass unless hi <= lo; a NaN tmin or tmax passes (src/aabb.rs:23-25).
    bool hit(const Ray& r, double tmin, double tmax) const {
        if (tmin != tmin || tmax != tmax) return true;
        double lo_t = tmin, hi_t = tmax;
        for (int a = 0; a < 3; ++a) {
            const double inv_d = 1.0 / r.dire[a];
            const double noi = -(r.orig[a] * inv_d);
            const bool neg = inv_d < 0.0;
            const double t0 = std::fma((double)(neg ? hi[a] : lo[a]), inv_d, noi);
            const double t1 = std::fma((double)(neg ? lo[a] : hi[a]), inv_d, noi);
            lo_t = std::fmax(t0, lo_t);
            hi_t = std::fmin(t1, hi_t);
        }
        return !(hi_t <= lo_t);
    }
    // The reference's per-axis rule (src/aabb.rs:15-30: tmin / tmax not carried between axes, t = (plane - origin) *
    // (1 / d)) on the f32 planes: what the HIP kernels run on scenes whose boxes may not bound their contents.
    bool hit_per_axis(const Ray& r, double tmin, double tmax) const {
        if (tmin != tmin || tmax != tmax) return true;
        for (int a = 0; a < 3; ++a) {
            const double inv_d = 1.0 / r.dire[a];
            const bool neg = inv_d < 0.0;
            const double t0 = ((double)(neg ? hi[a] : lo[a]) - r.orig[a]) * inv_d;
            const double t1 = ((double)(neg ? lo[a] : hi[a]) - r.orig[a]) * inv_d;
            if (std::fmin(t1, tmax) <= std::fmax(t0, tmin)) return false;
        }
        return true;
    }
};

inline AABB surrounding_box(const AABB& box0, const AABB& box1) {  // :33-50 (f64::min/max ignore NaN)
    Point small(std::fmin(box0._min.x, box1._min.x), std::fmin(box0._min.y, box1._min.y),
                std::fmin(box0._min.z, box1._min.z));
    Point big(std::fmax(box0._max.x, box1._max.x), std::fmax(box0._max.y, box1._max.y),
              std::fmax(box0._max.z, box1._max.z));
    return AABB(small, big);
}

// ------------------------------------------------------------------ texture.rs
struct Texture {  // :8-10
    virtual ~Texture() = default;
    virtual Color value(double u, double v, const Point& p) const = 0;
};
using TexPtr = std::shared_ptr<Texture>;

struct SolidColor : Texture {  // :12-26
    Color color_value;
    explicit SolidColor(Color c) : color_value(c) {}
    Color value(double, double, const Point&) const override { return color_value; }
};

struct CheckerTexture : Texture {  // :28-51
    TexPtr odd, even;
    CheckerTexture(Color c1, Color c2)
        : odd(std::make_shared<SolidColor>(c2)), even(std::make_shared<SolidColor>(c1)) {}
    Color value(double u, double v, const Point& p) const override {
        double sines = std::sin(p.x * 10.0) * std::sin(10.0 * p.y) * std::sin(10.0 * p.z);
        if (sines < 0.0) return odd->value(u, v, p);
        return even->value(u, v, p);
    }
};

// ------------------------------------------------------------------- perlin.rs
struct Perlin {
    static constexpr int POINT_COUNT = 256;  // :4
    Vec3 ran_vec[POINT_COUNT];
    int perm_x[POINT_COUNT], perm_y[POINT_COUNT], perm_z[POINT_COUNT];

    static void permute(int* tmp, int n) {  // :14-20
        for (int k = 0; k < n - 1; ++k) {
            int i = n - 1 - k;
            int target = random_int(0, i);
            std::swap(tmp[i], tmp[target]);
        }
    }
    static void perlin_generate_perm(int* tmp) {  // :21-30
        for (int i = 0; i < POINT_COUNT; ++i) tmp[i] = i;
        permute(tmp, POINT_COUNT);
    }
    Perlin() {  // :31-42
        for (int i = 0; i < POINT_COUNT; ++i) ran_vec[i] = Vec3::random(-1.0, 1.0).unit();
        perlin_generate_perm(perm_x);
        perlin_generate_perm(perm_y);
        perlin_generate_perm(perm_z);
    }
    static double perlin_interp(const Vec3 c[2][2][2], double u, double v, double w) {  // :44-63
        double accum = 0.0;
        double uu = u * u * (3.0 - 2.0 * u);
        double vv = v * v * (3.0 - 2.0 * v);
        double ww = w * w * (3.0 - 2.0 * w);
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                for (int k = 0; k < 2; ++k) {
                    Vec3 weight_v(u - (double)i, v - (double)j, w - (double)k);
                    accum += ((double)i * uu + (1.0 - (double)i) * (1.0 - uu)) *
                             ((double)j * vv + (1.0 - (double)j) * (1.0 - vv)) *
                             ((double)k * ww + (1.0 - (double)k) * (1.0 - ww)) * (c[i][j][k] * weight_v);
                }
        return accum;
    }
    static int f2i(double f) {  // Rust `as i32`: saturating, NaN -> 0
        if (f != f) return 0;
        if (f <= -2147483648.0) return INT32_MIN;
        if (f >= 2147483647.0) return INT32_MAX;
        return (int)f;
    }
    double noise(const Point& p) const {  // :66-89
        double uu = p.x - std::floor(p.x);
        double vv = p.y - std::floor(p.y);
        double ww = p.z - std::floor(p.z);
        // wrapping add as in release-mode Rust
        uint32_t i = (uint32_t)f2i(std::floor(p.x));
        uint32_t j = (uint32_t)f2i(std::floor(p.y));
        uint32_t k = (uint32_t)f2i(std::floor(p.z));
        Vec3 c_array[2][2][2];
        for (uint32_t di = 0; di < 2; ++di)
            for (uint32_t dj = 0; dj < 2; ++dj)
                for (uint32_t dk = 0; dk < 2; ++dk)
                    c_array[di][dj][dk] =
                        ran_vec[perm_x[(i + di) & 255] ^ perm_y[(j + dj) & 255] ^ perm_z[(k + dk) & 255]];
        return perlin_interp(c_array, uu, vv, ww);
    }
    double turb(const Point& p, int depth) const {  // :91-103
        double accum = 0.0;
        Point temp_p = p;
        double weight = 1.0;
        for (int i = 0; i < depth; ++i) {
            accum += weight * noise(temp_p);
            weight *= 0.5;
            temp_p *= 2.0;
        }
        return std::fabs(accum);
    }
};

struct NoiseTexture : Texture {  // texture.rs:53-73
    Perlin noise;
    double scale;
    explicit NoiseTexture(double sc) : noise(), scale(sc) {}
    Color value(double, double, const Point& p) const override {
        return Color::ones() * 0.5 * (1.0 + std::sin(scale * p.z + 10.0 * noise.turb(p * scale, 7)));
    }
};

struct ImageTexture : Texture {  // texture.rs:75-114; texels are handed in decoded (RGB u8)
    uint32_t width, height;
    std::vector<uint8_t> img;
    ImageTexture(uint32_t w, uint32_t h, const uint8_t* rgb) : width(w), height(h), img(rgb, rgb + (size_t)w * h * 3) {}
    static uint32_t f2u(double f) {  // Rust `as u32`: saturating, NaN -> 0
        if (!(f > 0.0)) return 0;
        if (f >= 4294967295.0) return 0xFFFFFFFFu;
        return (uint32_t)f;
    }
    Color value(double u, double v, const Point&) const override {
        u = clamp(u, 0.0, 1.0);
        v = 1.0 - clamp(v, 0.0, 1.0);
        uint32_t i = f2u(u * (double)width);
        uint32_t j = f2u(v * (double)height);
        if (i >= width) i = width - 1;
        if (j >= height) j = height - 1;
        double color_scale = 1.0 / 255.0;
        const uint8_t* px = &img[((size_t)j * width + i) * 3];
        return Color(color_scale * (double)px[0], color_scale * (double)px[1], color_scale * (double)px[2]);
    }
};

// ----------------------------------------------------------------- material.rs
struct HitRecord;
struct Material {  // :8-20
    uint32_t mat_id = NONE;
    virtual ~Material() = default;
    virtual bool scatter(const Ray& r_in, const HitRecord& rec, Color& attenuation, Ray& scattered) const = 0;
    virtual Color emitted(double, double, const Point&) const { return Color(0.0, 0.0, 0.0); }
};
using MatPtr = std::shared_ptr<Material>;

// ---------------------------------------------------------------- hittable.rs
struct HitRecord {  // :11-19 (+ identification of the winning object for parity checks)
    Point p;
    Vec3 normal;
    const Material* mat_ptr = nullptr;
    double t = 0.0, u = 0.0, v = 0.0;
    bool front_face = false;
    uint32_t prim_id = NONE;  // creation-order id of the object that produced the hit
    uint32_t side = 0;        // Box6 side 0..5 in src/box6.rs:22-71 order
    uint32_t inst_id = NONE;  // creation-order id of the innermost Translate/RotateY around it

    void set_face_normal(const Ray& r, const Vec3& outward_normal) {  // :33-40
        front_face = (r.dire * outward_normal) < 0.0;
        if (front_face) normal = outward_normal;
        else normal = -outward_normal;
    }
};

// Creation-order id counters of the scene being built (one per thread at a time).
struct SceneCtx {
    uint32_t next_prim = 0, next_inst = 0, next_medium = 0, next_mat = 0;
    // the ray times for which every moving sphere of the scene is within its own motion interval (see BVHNode)
    double carried_t0 = -std::numeric_limits<double>::infinity(), carried_t1 = std::numeric_limits<double>::infinity();
};
inline thread_local SceneCtx* cur_ctx = nullptr;
inline uint32_t take_prim_id() { return cur_ctx ? cur_ctx->next_prim++ : 0; }
inline uint32_t take_inst_id() { return cur_ctx ? cur_ctx->next_inst++ : 0; }
inline uint32_t take_medium_id() { return cur_ctx ? cur_ctx->next_medium++ : 0; }

struct Hittable {  // :43-46
    virtual ~Hittable() = default;
    virtual bool hit(const Ray& r, double t_min, double t_max, HitRecord& rec) const = 0;
    virtual bool bounding_box(double t0, double t1, AABB& output_box) const = 0;
    // called when BVHNode::new makes this object a direct child of a node built over (t0, t1); see CullBox
    // many_boxes: the list the tree was made from holds at least K_MANY_BOXES boxes of six sides
    virtual void adopted_by_bvh(double, double, bool /*many_boxes*/) const {}
    virtual bool is_box6() const { return false; }
    virtual const Hittable& unwrap() const { return *this; }  // the object a chain of Translate / RotateY wraps
    virtual bool one_entry() const { return false; }          // a single primitive
    // For the covering boxes of BVHNode (below): the hull [lo, hi] of the motion intervals of the moving spheres at or
    // below this object is widened to hold them, and a box that holds the object for every ray time in [t0, t1],
    // computed from the objects themselves (false: not known, behind a Translate / RotateY).
    virtual void motion_hull(double& /*lo*/, double& /*hi*/) const {}
    virtual bool bounds_over(double t0, double t1, AABB& out) const { return bounding_box(t0, t1, out); }
    // Does the box this object presents to a tree built over an interval that covers its motion hull hold everything
    // below it? Not when a tree below (or a RotateY, whose box is taken over (0, 1), :321) was built over an interval the
    // motion of its own content reaches out of: a BVHNode's box is the union of its children's stored boxes
    // (src/bvh.rs:80-88, :113-116), so every box above a box that does not hold does not hold either.
    virtual bool holds_content() const { return true; }
};
using HitPtr = std::shared_ptr<Hittable>;

// Not in the reference: the product's flattener puts a box entry in front of every Translate / RotateY that is a
// direct child of a BVHNode (the box the wrapper presented to BVHNode::new), so that rays which pass the node's box
// but miss the wrapper's own are not moved into its space and back. Pure culling. The oracle does the same in its
// two GPU box modes, counted as a box test, so that traversal work stays comparable; the reference mode does not.
struct CullBox {
    mutable FBox fbox;
    mutable bool has = false;
    void adopt(const Hittable& self, double t0, double t1, bool primitive = false) const {
        AABB b;
        // a wrapper: not around a single primitive, and only around content that does not move and whose own boxes hold
        // it (the box is one the reference never tests: it may only cull what no ray of any time can hit); a primitive
        // (a box of six sides that is a leaf of a tree): always
        double lo = std::numeric_limits<double>::infinity(), hi = -lo;
        self.motion_hull(lo, hi);
        has = (primitive || (!self.unwrap().one_entry() && !(lo <= hi) && self.holds_content())) && self.bounding_box(t0, t1, b);
        if (!has) return;
        // padded (2^-30 of the box's size and position) before it is rounded outward, so that no face of the object
        // coincides with a face of the box: a hit on such a face at exactly the running closest t would otherwise be
        // culled, where the reference accepts it
        for (int a = 0; a < 3; ++a) {
            const double scale = std::fmax(std::fmax(std::fabs(b._min[a]), std::fabs(b._max[a])), b._max[a] - b._min[a]);
            const double pad = std::fmax(scale, 1e-30) * 0x1p-30;
            fbox.lo[a] = f32_down(b._min[a] - pad);
            fbox.hi[a] = f32_up(b._max[a] + pad);
        }
    }
    bool culls(const Ray& r, double t_min, double t_max) const {
        if (!has || tls.aabb_mode == AABB_REFERENCE || tls.in_boundary) return false;
        ++tls.n_box;
        return !(tls.aabb_mode == AABB_TIGHT ? fbox.hit(r, t_min, t_max) : fbox.hit_per_axis(r, t_min, t_max));
    }
};

inline void get_sphere_uv(const Vec3& p, double& u, double& v) {  // :64-69
    double phi = std::atan2(p.z, p.x);
    double theta = std::asin(p.y);
    u = 1.0 - (phi + PI) / (2.0 * PI);
    v = (theta + PI / 2.0) / PI;
}

struct Sphere : Hittable {  // :48-122
    bool one_entry() const override { return true; }
    Point center;
    double radius;
    MatPtr mat_ptr;
    uint32_t id;
    Sphere(Point c, double r, MatPtr m) : center(c), radius(r), mat_ptr(std::move(m)), id(take_prim_id()) {}
    bool hit(const Ray& r, double t_min, double t_max, HitRecord& rec) const override {
        ++tls.n_prim[K_SPHERE];
        Vec3 oc = r.orig - center;
        double a = r.dire.squared_length();
        double half_b = oc * r.dire;
        double c = oc.squared_length() - radius * radius;
        double discriminant = half_b * half_b - a * c;
        if (discriminant > 0.0) {
            double root = std::sqrt(discriminant);
            double temp = (-half_b - root) / a;
            if (temp < t_max && temp > t_min) return accept(r, temp, rec);
            temp = (-half_b + root) / a;
            if (temp < t_max && temp > t_min) return accept(r, temp, rec);
        }
        return false;
    }
    bool accept(const Ray& r, double temp, HitRecord& rec) const {  // :84-94 / :99-109
        rec.t = temp;
        rec.p = r.at(rec.t);
        Vec3 outward_normal = (rec.p - center) / radius;
        rec.set_face_normal(r, outward_normal);
        get_sphere_uv((rec.p - center) / radius, rec.u, rec.v);
        rec.mat_ptr = mat_ptr.get();
        rec.prim_id = id; rec.side = 0; rec.inst_id = NONE;
        return true;
    }
    bool bounding_box(double, double, AABB& output_box) const override {  // :115-121
        output_box = AABB(center - Vec3(radius, radius, radius), center + Vec3(radius, radius, radius));
        return true;
    }
};

struct HittableList : Hittable {  // :124-185
    std::vector<HitPtr> objects;
    void clear() { objects.clear(); }
    void add(HitPtr object) { objects.push_back(std::move(object)); }
    bool hit(const Ray& r, double t_min, double t_max, HitRecord& rec) const override {  // :151-164
        HitRecord temp_rec = rec;
        bool hit_anything = false;
        double closest_so_far = t_max;
        for (const auto& object : objects) {
            if (object->hit(r, t_min, closest_so_far, temp_rec)) {
                hit_anything = true;
                closest_so_far = temp_rec.t;
                rec = temp_rec;
            }
        }
        return hit_anything;
    }
    bool bounding_box(double t0, double t1, AABB& output_box) const override {  // :166-184
        if (objects.empty()) return false;
        AABB temp_box;
        bool first_box = true;
        for (const auto& object : objects) {
            if (!object->bounding_box(t0, t1, temp_box)) return false;
            output_box = first_box ? temp_box : surrounding_box(output_box, temp_box);
            first_box = false;
        }
        return true;
    }
    void motion_hull(double& lo, double& hi) const override { for (const auto& o : objects) o->motion_hull(lo, hi); }
    bool holds_content() const override {
        for (const auto& o : objects) if (!o->holds_content()) return false;
        return true;
    }
    bool bounds_over(double t0, double t1, AABB& out) const override {
        if (objects.empty()) return false;
        AABB temp;
        bool first = true;
        for (const auto& o : objects) {
            if (!o->bounds_over(t0, t1, temp)) return false;
            out = first ? temp : surrounding_box(out, temp);
            first = false;
        }
        return true;
    }
};

struct MovingSphere : Hittable {  // :187-260
    bool one_entry() const override { return true; }
    Point center0, center1;
    double time0, time1, radius;
    MatPtr mat_ptr;
    uint32_t id;
    MovingSphere(Point cen0, Point cen1, double t0, double t1, double r, MatPtr m)
        : center0(cen0), center1(cen1), time0(t0), time1(t1), radius(r), mat_ptr(std::move(m)), id(take_prim_id()) {
        if (cur_ctx && (cen0.x != cen1.x || cen0.y != cen1.y || cen0.z != cen1.z)) {
            cur_ctx->carried_t0 = std::fmax(cur_ctx->carried_t0, std::fmin(t0, t1));
            cur_ctx->carried_t1 = std::fmin(cur_ctx->carried_t1, std::fmax(t0, t1));
        }
    }
    Point center(double time) const {  // :208-211
        return center0 + (center1 - center0) * ((time - time0) / (time1 - time0));
    }
    bool hit(const Ray& r, double t_min, double t_max, HitRecord& rec) const override {  // :215-246
        ++tls.n_prim[K_MSPHERE];
        Vec3 oc = r.orig - center(r.tm);
        double a = r.dire.squared_length();
        double half_b = oc * r.dire;
        double c = oc.squared_length() - radius * radius;
        double discriminant = half_b * half_b - a * c;
        if (discriminant > 0.0) {
            double root = std::sqrt(discriminant);
            double temp = (-half_b - root) / a;
            if (temp < t_max && temp > t_min) return accept(r, temp, rec);
            temp = (-half_b + root) / a;
            if (temp < t_max && temp > t_min) return accept(r, temp, rec);
        }
        return false;
    }
    bool accept(const Ray& r, double temp, HitRecord& rec) const {
        rec.t = temp;
        rec.p = r.at(rec.t);
        Vec3 outward_normal = (rec.p - center(r.tm)) / radius;
        rec.set_face_normal(r, outward_normal);
        rec.u = 0.0; rec.v = 0.0;  // stale in the reference; defined as 0 here
        rec.mat_ptr = mat_ptr.get();
        rec.prim_id = id; rec.side = 0; rec.inst_id = NONE;
        return true;
    }
    bool bounding_box(double t0, double t1, AABB& output_box) const override {  // :248-259
        AABB box0(center(t0) - Vec3(radius, radius, radius), center(t0) + Vec3(radius, radius, radius));
        AABB box1(center(t1) - Vec3(radius, radius, radius), center(t1) + Vec3(radius, radius, radius));
        output_box = surrounding_box(box0, box1);
        return true;
    }
    void motion_hull(double& lo, double& hi) const override {
        if (center0.x != center1.x || center0.y != center1.y || center0.z != center1.z) {
            lo = std::fmin(lo, std::fmin(time0, time1));
            hi = std::fmax(hi, std::fmax(time0, time1));
        }
    }
};

struct Translate : Hittable {  // :262-304
    HitPtr ptr;
    Vec3 offset;
    uint32_t id;
    Translate(HitPtr p, const Vec3& displacement) : ptr(std::move(p)), offset(displacement), id(take_inst_id()) {}
    CullBox cull;
    const Hittable& unwrap() const override { return ptr->unwrap(); }
    void motion_hull(double& lo, double& hi) const override { ptr->motion_hull(lo, hi); }
    bool holds_content() const override { return ptr->holds_content(); }
    bool bounds_over(double, double, AABB&) const override { return false; }
    void adopted_by_bvh(double t0, double t1, bool) const override { cull.adopt(*this, t0, t1); }
    bool hit(const Ray& r, double t_min, double t_max, HitRecord& rec) const override {  // :277-292
        if (cull.culls(r, t_min, t_max)) return false;
        Ray moved_r{r.orig - offset, r.dire, r.tm};
        if (!ptr->hit(moved_r, t_min, t_max, rec)) return false;
        rec.p += offset;
        Vec3 normal = rec.normal;
        rec.set_face_normal(moved_r, normal);
        if (rec.inst_id == NONE) rec.inst_id = id;
        return true;
    }
    bool bounding_box(double t0, double t1, AABB& output_box) const override {  // :294-303
        if (!ptr->bounding_box(t0, t1, output_box)) return false;
        output_box = AABB(output_box._min + offset, output_box._max + offset);
        return true;
    }
};

struct RotateY : Hittable {  // :306-392
    HitPtr ptr;
    double sin_theta, cos_theta;
    bool hasbox;
    AABB bbox;
    uint32_t id;
    RotateY(HitPtr p, double angle) : ptr(std::move(p)), id(take_inst_id()) {  // :315-350
        double radians = degrees_to_radians(angle);
        // Two libm calls, as the reference writes them (radians.sin(), radians.cos(), :317-319). Left alone GCC merges
        // them into one sincos() call, and glibc's sincos differs from its sin / cos in the last bit for about one angle
        // in a thousand: enough to move a grazing hit inside the wrapper by hundreds of units in the last place. The
        // product's host library (built by clang) makes the two calls.
        sin_theta = std::sin(radians);
        volatile double radians_again = radians;
        cos_theta = std::cos(radians_again);
        Point min(INF, INF, INF);
        Point max(-INF, -INF, -INF);
        hasbox = ptr->bounding_box(0.0, 1.0, bbox);
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                for (int k = 0; k < 2; ++k) {
                    double x = (double)i * bbox._max.x + (double)(1 - i) * bbox._min.x;
                    double y = (double)j * bbox._max.y + (double)(1 - j) * bbox._min.y;
                    double z = (double)k * bbox._max.z + (double)(1 - k) * bbox._min.z;
                    double newx = cos_theta * x + sin_theta * z;
                    double newz = -sin_theta * x + cos_theta * z;
                    Vec3 tester(newx, y, newz);
                    for (int c = 0; c < 3; ++c) {
                        min.at(c) = std::fmin(min[c], tester[c]);
                        max.at(c) = std::fmax(max[c], tester[c]);
                    }
                }
        bbox = AABB(min, max);
    }
    CullBox cull;
    const Hittable& unwrap() const override { return ptr->unwrap(); }
    void motion_hull(double& lo, double& hi) const override { ptr->motion_hull(lo, hi); }
    bool holds_content() const override {
        double lo = std::numeric_limits<double>::infinity(), hi = -lo;
        ptr->motion_hull(lo, hi);
        return ptr->holds_content() && (!(lo <= hi) || (lo >= 0.0 && hi <= 1.0));
    }
    bool bounds_over(double, double, AABB&) const override { return false; }
    void adopted_by_bvh(double t0, double t1, bool) const override { cull.adopt(*this, t0, t1); }
    bool hit(const Ray& r, double t_min, double t_max, HitRecord& rec) const override {  // :354-387
        if (cull.culls(r, t_min, t_max)) return false;
        Vec3 orig = r.orig;
        Vec3 dire = r.dire;
        orig.x = cos_theta * r.orig[0] - sin_theta * r.orig[2];
        orig.z = sin_theta * r.orig[0] + cos_theta * r.orig[2];
        dire.x = cos_theta * r.dire[0] - sin_theta * r.dire[2];
        dire.z = sin_theta * r.dire[0] + cos_theta * r.dire[2];
        Ray rotated_r{orig, dire, r.tm};
        if (!ptr->hit(rotated_r, t_min, t_max, rec)) return false;
        Vec3 p = rec.p;
        Vec3 normal = rec.normal;
        p.x = cos_theta * rec.p[0] + sin_theta * rec.p[2];
        p.z = -sin_theta * rec.p[0] + cos_theta * rec.p[2];
        normal.x = cos_theta * rec.normal[0] + sin_theta * rec.normal[2];
        normal.z = -sin_theta * rec.normal[0] + cos_theta * rec.normal[2];
        rec.p = p;
        rec.set_face_normal(rotated_r, normal);
        if (rec.inst_id == NONE) rec.inst_id = id;
        return true;
    }
    bool bounding_box(double, double, AABB& output_box) const override {  // :388-391
        output_box = bbox;
        return hasbox;
    }
};

// ------------------------------------------------------------------- aarect.rs
// One struct for the three orientations: axis 0 = XYRect (:7-56, plane z = k),
// 1 = XZRect (:58-107, plane y = k), 2 = YZRect (:109-158, plane x = k).
struct AARect : Hittable {
    bool one_entry() const override { return true; }
    int axis;
    double a0, a1, b0, b1, k;
    MatPtr mp;
    uint32_t id, side;
    AARect(int axis_, double a0_, double a1_, double b0_, double b1_, double k_, MatPtr mat, uint32_t id_, uint32_t side_)
        : axis(axis_), a0(a0_), a1(a1_), b0(b0_), b1(b1_), k(k_), mp(std::move(mat)), id(id_), side(side_) {}
    bool hit(const Ray& r, double t0, double t1, HitRecord& rec) const override {
        ++tls.n_prim[K_RECT];
        double ok, dk, oa, da, ob, db;
        Vec3 outward_normal;
        if (axis == 0) { ok = r.orig.z; dk = r.dire.z; oa = r.orig.x; da = r.dire.x; ob = r.orig.y; db = r.dire.y; outward_normal = Vec3(0.0, 0.0, 1.0); }
        else if (axis == 1) { ok = r.orig.y; dk = r.dire.y; oa = r.orig.x; da = r.dire.x; ob = r.orig.z; db = r.dire.z; outward_normal = Vec3(0.0, 1.0, 0.0); }
        else { ok = r.orig.x; dk = r.dire.x; oa = r.orig.y; da = r.dire.y; ob = r.orig.z; db = r.dire.z; outward_normal = Vec3(1.0, 0.0, 0.0); }
        double t = (k - ok) / dk;                                   // :31 / :82 / :133
        if (t < t0 || t > t1) return false;                         // :32
        double a = oa + t * da;                                     // :35-36
        double b = ob + t * db;
        if (a < a0 || a > a1 || b < b0 || b > b1) return false;      // :37
        rec.u = (a - a0) / (a1 - a0);                               // :40-41
        rec.v = (b - b0) / (b1 - b0);
        rec.t = t;
        rec.set_face_normal(r, outward_normal);                     // :43-44
        rec.mat_ptr = mp.get();
        rec.p = r.at(t);                                            // :46
        rec.prim_id = id; rec.side = side; rec.inst_id = NONE;
        return true;
    }
    bool bounding_box(double, double, AABB& output_box) const override {  // :49-55 / :100-106 / :151-157
        if (axis == 0) output_box = AABB(Point(a0, b0, k - 0.0001), Point(a1, b1, k + 0.0001));
        else if (axis == 1) output_box = AABB(Point(a0, k - 0.0001, b0), Point(a1, k + 0.0001, b1));
        else output_box = AABB(Point(k - 0.0001, a0, b0), Point(k + 0.0001, a1, b1));
        return true;
    }
};
inline HitPtr XYRect(double x0, double x1, double y0, double y1, double k, MatPtr m) {
    return std::make_shared<AARect>(0, x0, x1, y0, y1, k, std::move(m), take_prim_id(), 0);
}
inline HitPtr XZRect(double x0, double x1, double z0, double z1, double k, MatPtr m) {
    return std::make_shared<AARect>(1, x0, x1, z0, z1, k, std::move(m), take_prim_id(), 0);
}
inline HitPtr YZRect(double y0, double y1, double z0, double z1, double k, MatPtr m) {
    return std::make_shared<AARect>(2, y0, y1, z0, z1, k, std::move(m), take_prim_id(), 0);
}

// --------------------------------------------------------------------- box6.rs
struct Box6 : Hittable {  // :9-85; sides share the box's id and carry their index
    bool one_entry() const override { return true; }
    Point box_min, box_max;
    HittableList sides;
    uint32_t id;
    Box6(const Point& p0, const Point& p1, MatPtr ptr) : box_min(p0), box_max(p1), id(take_prim_id()) {
        sides.add(std::make_shared<AARect>(0, p0.x, p1.x, p0.y, p1.y, p1.z, ptr, id, 0));  // :22-29
        sides.add(std::make_shared<AARect>(0, p0.x, p1.x, p0.y, p1.y, p0.z, ptr, id, 1));  // :30-37
        sides.add(std::make_shared<AARect>(1, p0.x, p1.x, p0.z, p1.z, p1.y, ptr, id, 2));  // :39-46
        sides.add(std::make_shared<AARect>(1, p0.x, p1.x, p0.z, p1.z, p0.y, ptr, id, 3));  // :47-54
        sides.add(std::make_shared<AARect>(2, p0.y, p1.y, p0.z, p1.z, p1.x, ptr, id, 4));  // :56-63
        sides.add(std::make_shared<AARect>(2, p0.y, p1.y, p0.z, p1.z, p0.x, ptr, id, 5));  // :64-71
    }
    CullBox cull;
    bool is_box6() const override { return true; }
    void adopted_by_bvh(double t0, double t1, bool many_boxes) const override {
        if (many_boxes) cull.adopt(*this, t0, t1, true); else cull.has = false;
    }
    bool hit(const Ray& r, double t_min, double t_max, HitRecord& rec) const override {  // :78-80
        if (cull.culls(r, t_min, t_max)) return false;
        ++tls.n_prim[K_BOX6];
        return sides.hit(r, t_min, t_max, rec);
    }
    bool bounding_box(double, double, AABB& output_box) const override {  // :81-84
        output_box = AABB(box_min, box_max);
        return true;
    }
};

// ----------------------------------------------------------------- triangle.rs
struct Triangle : Hittable {  // :7-84 (constructor argument order point1, point2, point0: :15)
    bool one_entry() const override { return true; }
    Point point1, point2, point0;
    MatPtr mp;
    uint32_t id;
    Triangle(Point p1, Point p2, Point p0, MatPtr m) : point1(p1), point2(p2), point0(p0), mp(std::move(m)), id(take_prim_id()) {}
    bool hit(const Ray& r, double t_min, double t_max, HitRecord& rec) const override {  // :26-69
        ++tls.n_prim[K_TRIANGLE];
        Vec3 edge1 = point1 - point0;
        Vec3 edge2 = point2 - point0;
        Vec3 pvec = Vec3::cross(r.dire, edge2);
        double det = pvec * edge1;
        const double EPSILON = 1e-8;
        if (det < EPSILON && det > -EPSILON) return false;
        double inv_det = 1.0 / det;
        Vec3 tvec = r.orig - point0;
        double u = tvec * pvec * inv_det;
        if (u < 0.0 || u > 1.0) return false;
        Vec3 qvec = Vec3::cross(tvec, edge1);
        double v = r.dire * qvec * inv_det;
        if (v < 0.0 || u + v > 1.0) return false;
        double t = edge2 * qvec * inv_det;
        if (t < t_min || t > t_max) return false;
        Vec3 outward_normal = Vec3::cross(edge1, edge2).unit();
        rec.set_face_normal(r, outward_normal);
        rec.u = u;
        rec.v = v;
        rec.t = t;
        rec.p = r.at(t);
        rec.mat_ptr = mp.get();
        rec.prim_id = id; rec.side = 0; rec.inst_id = NONE;
        return true;
    }
    bool bounding_box(double, double, AABB& output_box) const override {  // :70-83
        double min_x = std::fmin(std::fmin(point0.x, point1.x), point2.x);
        double min_y = std::fmin(std::fmin(point0.y, point1.y), point2.y);
        double min_z = std::fmin(std::fmin(point0.z, point1.z), point2.z);
        double max_x = std::fmax(std::fmax(point0.x, point1.x), point2.x);
        double max_y = std::fmax(std::fmax(point0.y, point1.y), point2.y);
        double max_z = std::fmax(std::fmax(point0.z, point1.z), point2.z);
        output_box = AABB(Point(min_x, min_y, min_z), Point(max_x, max_y, max_z));
        return true;
    }
};

// ----------------------------------------------------------------- material.rs
struct Lambertian : Material {  // :22-51
    TexPtr albedo;
    explicit Lambertian(Color a) : albedo(std::make_shared<SolidColor>(a)) {}
    explicit Lambertian(TexPtr a) : albedo(std::move(a)) {}  // struct literal, e.g. src/main.rs:250
    bool scatter(const Ray& r_in, const HitRecord& rec, Color& attenuation, Ray& scattered) const override {
        Vec3 scatter_direction = rec.normal + random_unit_vector();
        scattered = Ray{rec.p, scatter_direction, r_in.tm};
        attenuation = albedo->value(rec.u, rec.v, rec.p);
        return true;
    }
};

struct Metal : Material {  // :53-84
    Color albedo;
    double fuzz;
    Metal(const Color& a, double f) : albedo(a), fuzz(f < 1.0 ? f : 1.0) {}
    bool scatter(const Ray& r_in, const HitRecord& rec, Color& attenuation, Ray& scattered) const override {
        if (r_in.dire.length() == 0.0) return false;  // reference would panic
        Vec3 reflected = reflect(r_in.dire.unit(), rec.normal);
        scattered = Ray{rec.p, reflected + random_in_unit_sphere() * fuzz, r_in.tm};
        attenuation = albedo;
        return scattered.dire * rec.normal > 0.0;
    }
};

inline double schlick(double cosine, double ref_idx) {  // :144-148; powi(5) = x * ((x*x)*(x*x))
    double r0 = (1.0 - ref_idx) / (1.0 + ref_idx);
    r0 = r0 * r0;
    double x = 1.0 - cosine;
    double x2 = x * x;
    double x5 = x * (x2 * x2);
    return r0 + (1.0 - r0) * x5;
}

struct Dielectric : Material {  // :86-142
    double ref_idx;
    explicit Dielectric(double ri) : ref_idx(ri) {}
    bool scatter(const Ray& r_in, const HitRecord& rec, Color& attenuation, Ray& scattered) const override {
        attenuation = Color(1.0, 1.0, 1.0);
        double etai_over_etat = rec.front_face ? 1.0 / ref_idx : ref_idx;
        if (r_in.dire.length() == 0.0) return false;  // reference would panic
        Vec3 unit_direction = r_in.dire.unit();
        double cos_theta = std::fmin((-unit_direction) * rec.normal, 1.0);
        double sin_theta = std::sqrt(1.0 - cos_theta * cos_theta);
        if (etai_over_etat * sin_theta > 1.0) {
            Vec3 reflected = reflect(unit_direction, rec.normal);
            scattered = Ray{rec.p, reflected, r_in.tm};
            return true;
        }
        double reflect_prob = schlick(cos_theta, etai_over_etat);
        if (random_double(0.0, 1.0) < reflect_prob) {
            Vec3 reflected = reflect(unit_direction, rec.normal);
            scattered = Ray{rec.p, reflected, r_in.tm};
            return true;
        }
        Vec3 refracted = refract(unit_direction, rec.normal, etai_over_etat);
        scattered = Ray{rec.p, refracted, r_in.tm};
        return true;
    }
};

struct DiffuseLight : Material {  // :150-179
    TexPtr emit;
    explicit DiffuseLight(TexPtr a) : emit(std::move(a)) {}
    explicit DiffuseLight(Color c) : emit(std::make_shared<SolidColor>(c)) {}
    bool scatter(const Ray&, const HitRecord&, Color&, Ray&) const override { return false; }
    Color emitted(double u, double v, const Point& p) const override { return emit->value(u, v, p); }
};

struct Isotropic : Material {  // :181-211
    TexPtr albedo;
    explicit Isotropic(TexPtr a) : albedo(std::move(a)) {}
    explicit Isotropic(Color c) : albedo(std::make_shared<SolidColor>(c)) {}
    bool scatter(const Ray& r_in, const HitRecord& rec, Color& attenuation, Ray& scattered) const override {
        scattered = Ray{rec.p, random_in_unit_sphere(), r_in.tm};
        attenuation = albedo->value(rec.u, rec.v, rec.p);
        return true;
    }
};

// ---------------------------------------------------------- constant_medium.rs
struct ConstantMedium : Hittable {  // :10-94
    HitPtr boundary;
    MatPtr phase_function;
    double neg_inv_density;
    uint32_t id, rng_id;
    // Not in the reference: the product takes media in world space whose boundary is one sphere out of its entry stream
    // and asks them before everything else (World::hit below; collect_deferred_media decides which). The GPU modes do
    // the same, so that traversal work stays comparable. Bit m set: deferred in box mode m.
    mutable uint32_t deferred_modes = 0;
    ConstantMedium(HitPtr b, double d, TexPtr a)
        : boundary(std::move(b)), phase_function(std::make_shared<Isotropic>(std::move(a))), neg_inv_density(-1.0 / d),
          id(take_prim_id()), rng_id(take_medium_id()) {}
    ConstantMedium(HitPtr b, double d, Color c)
        : boundary(std::move(b)), phase_function(std::make_shared<Isotropic>(c)), neg_inv_density(-1.0 / d),
          id(take_prim_id()), rng_id(take_medium_id()) {}
    bool hit(const Ray& r, double t_min, double t_max, HitRecord& rec) const override {  // :34-89
        if (((deferred_modes >> tls.aabb_mode) & 1u) && !tls.deferred_pass) return false;  // World::hit has asked it already
        ++tls.n_prim[K_MEDIUM];
        HitRecord rec1, rec2;
        struct Inside { bool was; Inside() : was(tls.in_boundary) { tls.in_boundary = true; } ~Inside() { tls.in_boundary = was; } };
        {
            Inside scope;
            if (!boundary->hit(r, -INF, INF, rec1)) return false;
            if (!boundary->hit(r, rec1.t + 0.0001, INF, rec2)) return false;
        }
        if (rec1.t < t_min) rec1.t = t_min;
        if (rec2.t > t_max) rec2.t = t_max;
        if (rec1.t > rec2.t) return false;
        if (rec1.t < 0.0) rec1.t = 0.0;
        double ray_length = r.dire.length();
        double distance_inside_boundary = (rec2.t - rec1.t) * ray_length;
        // the one draw of this call, keyed by (pixel, sample, bounce, medium) so that the
        // closest hit does not depend on traversal order
        tls.rng.set(tls.pixel, tls.sample, tls.bounce, STREAM_MEDIUM0 + rng_id);
        double hit_distance = neg_inv_density * std::log10(random_double(0.0, 1.0));  // :69 (log10!)
        if (hit_distance > distance_inside_boundary) return false;
        rec.t = rec1.t + hit_distance / ray_length;
        rec.p = r.at(rec.t);
        rec.normal = Vec3(1.0, 0.0, 0.0);
        rec.front_face = true;
        rec.u = 0.0; rec.v = 0.0;  // stale in the reference; defined as 0 here
        rec.mat_ptr = phase_function.get();
        rec.prim_id = id; rec.side = 0; rec.inst_id = NONE;
        return true;
    }
    bool bounding_box(double t0, double t1, AABB& output_box) const override {  // :91-93
        return boundary->bounding_box(t0, t1, output_box);
    }
    void motion_hull(double& lo, double& hi) const override { boundary->motion_hull(lo, hi); }
    bool holds_content() const override { return boundary->holds_content(); }
    bool bounds_over(double t0, double t1, AABB& out) const override { return boundary->bounds_over(t0, t1, out); }
};

// ---------------------------------------------------------------------- bvh.rs
struct BVHNode : Hittable {  // :10-15
    HitPtr left, right;
    AABB _box;
    FBox fbox;
    // Not in the reference; what the product's flattener does in scenes whose boxes may not bound their contents
    // (AABB_PER_AXIS_F32). `holds`: every moving sphere below moves within the interval the node was built over and
    // every tree / wrapper below holds its own content (holds_content), so the
    // box holds its content and is tested with the carried interval. Otherwise the box is tested with the reference's
    // per-axis rule, behind a covering box over the whole motion of what is below (carried interval, a box test of its
    // own): a ray that misses the covering box cannot hit anything below. Both only for rays whose time lies within the
    // motion interval of every moving sphere of the scene (tls.carried_t0 .. t1): outside it MovingSphere::center
    // extrapolates (src/hittable.rs:208-211) and no box built over the motion holds; such a ray gets the reference's
    // per-axis test on every node and passes every covering box.
    bool holds = false, has_cover = false;
    bool content_held = false;  // what a tree above sees (a linear tree is walked without marks: holds stays false there)
    bool holds_content() const override { return content_held; }
    bool interval_covers_motion(double time0, double time1, double& lo, double& hi) const {
        lo = std::numeric_limits<double>::infinity(); hi = -lo;
        motion_hull(lo, hi);
        // the flattener keeps the node's interval as f32, rounded inward
        return !(lo <= hi) || (lo >= (double)f32_up(time0) && hi <= (double)f32_down(time1));
    }
    FBox cover;
    // A node whose box contains the world's box (the root, and in final_scene the two nodes above the fog's sphere of
    // radius 5000) is passed by every ray that starts inside the world; the flattener gives such a node no entry, and
    // the GPU modes here skip its test (mark_world_boxes below). Skipping a box that holds its content changes no hit.
    bool no_entry = false;
    // Everything below is a deferred medium (carried-interval mode): the product's stream has no entry for such a node
    bool pruned = false;
    void motion_hull(double& lo, double& hi) const override {
        left->motion_hull(lo, hi);
        if (right.get() != left.get()) right->motion_hull(lo, hi);
    }
    bool bounds_over(double t0, double t1, AABB& out) const override {
        AABB a, b;
        if (!left->bounds_over(t0, t1, a)) return false;
        if (right.get() == left.get()) { out = a; return true; }
        if (!right->bounds_over(t0, t1, b)) return false;
        out = surrounding_box(a, b);
        return true;
    }
    // after left, right and _box are set by BVHNode::new over (time0, time1)
    void find_cover(double time0, double time1) {
        double lo, hi;
        holds = interval_covers_motion(time0, time1, lo, hi) && left->holds_content() && (right.get() == left.get() || right->holds_content());
        content_held = holds;
        AABB all;
        has_cover = !holds && bounds_over(lo, hi, all);
        if (!has_cover) return;
        for (int a = 0; a < 3; ++a) {  // padded like CullBox
            const double scale = std::fmax(std::fmax(std::fabs(all._min[a]), std::fabs(all._max[a])), all._max[a] - all._min[a]);
            const double pad = std::fmax(scale, 1e-30) * 0x1p-30;
            cover.lo[a] = f32_down(all._min[a] - pad);
            cover.hi[a] = f32_up(all._max[a] + pad);
        }
    }

    bool hit(const Ray& r, double t_min, double t_max, HitRecord& rec) const override {  // :101-111
        if (pruned && tls.aabb_mode == AABB_TIGHT) return false;
        if (no_entry && tls.aabb_mode != AABB_REFERENCE) {
            bool hit_left = left->hit(r, t_min, t_max, rec);
            if (right.get() == left.get()) return hit_left;
            bool hit_right = right->hit(r, t_min, hit_left ? rec.t : t_max, rec);
            return hit_left || hit_right;
        }
        const bool in_motion = r.tm >= tls.carried_t0 && r.tm <= tls.carried_t1;
        if (tls.aabb_mode == AABB_PER_AXIS_F32 && has_cover && !tls.in_boundary) {
            ++tls.n_box;
            if (in_motion && !cover.hit(r, t_min, t_max)) return false;
        }
        ++tls.n_box;
        if (tls.aabb_mode == AABB_TIGHT) {
            if (!fbox.hit(r, t_min, t_max)) return false;
        } else if (tls.aabb_mode == AABB_PER_AXIS_F32) {
            if (!(holds && in_motion ? fbox.hit(r, t_min, t_max) : fbox.hit_per_axis(r, t_min, t_max))) return false;
        } else {
            if (!_box.hit(r, t_min, t_max)) return false;
        }
        bool hit_left = left->hit(r, t_min, t_max, rec);
        if (right.get() == left.get()) return hit_left;  // span-1 node (:39-44): the second call is idempotent
        bool hit_right = right->hit(r, t_min, hit_left ? rec.t : t_max, rec);
        return hit_left || hit_right;
    }
    bool bounding_box(double, double, AABB& output_box) const override {  // :113-116
        output_box = _box;
        return true;
    }
};

// Marks the nodes that get no entry: reached from the world's root through BVHNodes and HittableLists only (world
// space, outside any medium boundary), holding their content, not flat on an axis, and containing the root's box.
inline void mark_world_boxes(const HitPtr& obj, const AABB& world) {
    if (auto* list = dynamic_cast<HittableList*>(obj.get())) {
        for (const HitPtr& o : list->objects) mark_world_boxes(o, world);
        return;
    }
    auto* node = dynamic_cast<BVHNode*>(obj.get());
    if (!node) return;
    bool flat = false, contains = true;
    for (int a = 0; a < 3; ++a) {
        flat = flat || node->_box._min[a] == node->_box._max[a];
        contains = contains && node->_box._min[a] <= world._min[a] && node->_box._max[a] >= world._max[a];
    }
    node->no_entry = node->holds && !flat && contains;
    mark_world_boxes(node->left, world);
    if (node->right.get() != node->left.get()) mark_world_boxes(node->right, world);
}

// The media that are asked before the walk: ConstantMedium objects in world space (reached from the root through
// HittableLists and BVHNodes, not through wrappers) whose boundary is one Sphere, the first K_MAX_DEFERRED_MEDIA in
// walk order. Where boxes may not bound their contents (AABB_PER_AXIS_F32) only those that every ray reaches: below
// nodes that get no entry.
constexpr size_t K_MAX_DEFERRED_MEDIA = 2;
inline void collect_deferred_media(const HitPtr& obj, bool reached_by_all, std::vector<const ConstantMedium*>& tight,
                                   std::vector<const ConstantMedium*>& per_axis) {
    if (auto* list = dynamic_cast<HittableList*>(obj.get())) {
        for (const HitPtr& o : list->objects) collect_deferred_media(o, reached_by_all, tight, per_axis);
    } else if (auto* node = dynamic_cast<BVHNode*>(obj.get())) {
        const bool all = reached_by_all && node->no_entry;
        collect_deferred_media(node->left, all, tight, per_axis);
        if (node->right.get() != node->left.get()) collect_deferred_media(node->right, all, tight, per_axis);
    } else if (auto* medium = dynamic_cast<ConstantMedium*>(obj.get())) {
        if (!dynamic_cast<Sphere*>(medium->boundary.get())) return;
        if (tight.size() < K_MAX_DEFERRED_MEDIA && !(medium->deferred_modes & (1u << AABB_TIGHT))) {
            medium->deferred_modes |= 1u << AABB_TIGHT;
            tight.push_back(medium);
        }
        if (reached_by_all && per_axis.size() < K_MAX_DEFERRED_MEDIA && !(medium->deferred_modes & (1u << AABB_PER_AXIS_F32))) {
            medium->deferred_modes |= 1u << AABB_PER_AXIS_F32;
            per_axis.push_back(medium);
        }
    }
}
// Nodes with nothing but deferred media below (carried-interval mode) have no entry in the product's stream.
inline bool prune_emptied_nodes(const HitPtr& obj) {
    if (auto* list = dynamic_cast<HittableList*>(obj.get())) {
        bool empty = !list->objects.empty();
        for (const HitPtr& o : list->objects) empty = prune_emptied_nodes(o) && empty;
        return empty;
    }
    if (auto* node = dynamic_cast<BVHNode*>(obj.get())) {
        const bool l = prune_emptied_nodes(node->left);
        const bool r = node->right.get() == node->left.get() ? l : prune_emptied_nodes(node->right);
        node->pruned = l && r;
        return node->pruned;
    }
    if (auto* medium = dynamic_cast<ConstantMedium*>(obj.get())) return (medium->deferred_modes & (1u << AABB_TIGHT)) != 0;
    return false;
}

// world.hit (src/color.rs:41) as the product searches the world: in the GPU modes the deferred media are asked first, as
// if they stood first in the world's list, then the reference's walk with their hit as the closest so far. The medium's
// candidate t does not depend on t_max (only its acceptance does) and its random number is keyed by the medium, so the
// closest of all candidates comes out as in the reference's order, except when the medium's t and another object's t
// agree to the last bits.
struct World : Hittable {
    HitPtr root;
    std::vector<const ConstantMedium*> deferred_tight, deferred_per_axis;
    explicit World(HitPtr r) : root(std::move(r)) {
        if (auto* node = dynamic_cast<BVHNode*>(root.get())) mark_world_boxes(root, node->_box);
        collect_deferred_media(root, true, deferred_tight, deferred_per_axis);
        (void)prune_emptied_nodes(root);
    }
    bool hit(const Ray& r, double t_min, double t_max, HitRecord& rec) const override {
        if (tls.aabb_mode == AABB_REFERENCE) return root->hit(r, t_min, t_max, rec);
        bool any = false;
        for (const ConstantMedium* m : tls.aabb_mode == AABB_TIGHT ? deferred_tight : deferred_per_axis) {
            tls.deferred_pass = true;
            const bool h = m->hit(r, t_min, any ? rec.t : t_max, rec);
            tls.deferred_pass = false;
            any = any || h;
        }
        // like HittableList::hit (src/hittable.rs:151-164): the record is only written by what is hit within the range
        HitRecord temp = rec;
        if (root->hit(r, t_min, any ? rec.t : t_max, temp)) { rec = temp; any = true; }
        return any;
    }
    bool bounding_box(double t0, double t1, AABB& output_box) const override { return root->bounding_box(t0, t1, output_box); }
};

// Boxes of six sides that are leaves of a tree get a culling box of their own (CullBox) when the list the tree is made
// from holds at least this many of them; the product's flattener applies the same rule.
constexpr size_t K_MANY_BOXES = 8;
inline thread_local bool bvh_many_boxes = false;

inline bool box_less(const HitPtr& a, const HitPtr& b, int axis) {  // box_compare :119-134 (times 0,0)
    AABB box_a, box_b;
    if (!a->bounding_box(0.0, 0.0, box_a) || !b->bounding_box(0.0, 0.0, box_b)) throw std::runtime_error("no bounding box");
    return box_a._min[axis] < box_b._min[axis];
}

// Split-axis policy (not in the reference): false = the reference's random axis, true = the axis on which the box
// minima of the node's objects spread the widest (ties: the lowest axis).
inline thread_local bool bvh_widest_axis = false;

inline int widest_axis(const std::vector<HitPtr>& objects, size_t start, size_t end) {
    double lo[3], hi[3];
    for (int a = 0; a < 3; ++a) { lo[a] = std::numeric_limits<double>::infinity(); hi[a] = -lo[a]; }
    for (size_t i = start; i < end; ++i) {
        AABB b;
        if (!objects[i]->bounding_box(0.0, 0.0, b)) throw std::runtime_error("no bounding box");
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], b._min[a]); hi[a] = std::max(hi[a], b._min[a]); }
    }
    int best = 0;
    for (int a = 1; a < 3; ++a) if (hi[a] - lo[a] > hi[best] - lo[best]) best = a;
    return best;
}

inline std::shared_ptr<BVHNode> new_from_list(std::vector<HitPtr>& objects, size_t start, size_t end, double time0,
                                              double time1) {  // :17-91
    int axis = random_int(0, 2);
    size_t object_span = end - start;
    if (bvh_widest_axis && object_span > 1) axis = widest_axis(objects, start, end);
    auto temp = std::make_shared<BVHNode>();
    if (object_span == 1) {
        temp->left = objects[start];
        temp->right = objects[start];
    } else if (object_span == 2) {
        // Ordering::Greater swaps; Less and Equal keep the order (:46-68)
        if (box_less(objects[start + 1], objects[start], axis)) {
            temp->left = objects[start + 1];
            temp->right = objects[start];
        } else {
            temp->left = objects[start];
            temp->right = objects[start + 1];
        }
    } else {
        // slice::sort_by is a stable sort (:70)
        std::stable_sort(objects.begin() + start, objects.begin() + end,
                         [axis](const HitPtr& a, const HitPtr& b) { return box_less(a, b, axis); });
        size_t mid = start + object_span / 2;
        temp->left = new_from_list(objects, start, mid, time0, time1);
        temp->right = new_from_list(objects, mid, end, time0, time1);
    }
    AABB box_left, box_right;
    if (!temp->left->bounding_box(time0, time1, box_left) || !temp->right->bounding_box(time0, time1, box_right))
        throw std::runtime_error("no bounding box");  // :83-87
    temp->_box = surrounding_box(box_left, box_right);
    temp->fbox = FBox::from(temp->_box);
    temp->find_cover(time0, time1);
    if (object_span <= 2) {  // its children are objects of the list, not nodes
        temp->left->adopted_by_bvh(time0, time1, bvh_many_boxes);
        if (temp->right.get() != temp->left.get()) temp->right->adopted_by_bvh(time0, time1, bvh_many_boxes);
    }
    return temp;
}

// Tree policy (not in the reference): true = lists of two or more objects get the linear BVH of lbvh.hpp.
inline thread_local bool bvh_lbvh = false;

// the object tree that a pre-order token sequence describes; `at` advances past the subtree
inline HitPtr lbvh_subtree(const std::vector<RtNode>& tokens, size_t& at, const std::vector<HitPtr>& objects, double time0,
                           double time1) {
    const RtNode& t = tokens.at(at++);
    if (t.meta == RT_BVH_TOKEN_LEAF) return objects.at(t.link);
    auto node = std::make_shared<BVHNode>();
    node->left = lbvh_subtree(tokens, at, objects, time0, time1);
    node->right = lbvh_subtree(tokens, at, objects, time0, time1);
    if (at != t.link) throw std::runtime_error("malformed tree tokens");
    AABB box_left, box_right;
    if (!node->left->bounding_box(time0, time1, box_left) || !node->right->bounding_box(time0, time1, box_right))
        throw std::runtime_error("no bounding box");
    node->_box = surrounding_box(box_left, box_right);
    node->fbox = FBox::from(node->_box);
    for (int a = 0; a < 3; ++a)
        if (node->fbox.lo[a] != t.lo[a] || node->fbox.hi[a] != t.hi[a]) throw std::runtime_error("token box differs from the node's");
    return node;
}

inline std::shared_ptr<BVHNode> lbvh_from_list(const std::vector<HitPtr>& objects, double time0, double time1) {
    std::vector<double> boxes(objects.size() * 6);
    for (size_t i = 0; i < objects.size(); ++i) {
        AABB b;
        if (!objects[i]->bounding_box(time0, time1, b)) throw std::runtime_error("no bounding box");
        for (int a = 0; a < 3; ++a) { boxes[6 * i + a] = b._min[a]; boxes[6 * i + 3 + a] = b._max[a]; }
    }
    size_t n_box6 = 0;
    for (const HitPtr& o : objects) n_box6 += o->is_box6() ? 1 : 0;
    for (const HitPtr& o : objects) o->adopted_by_bvh(time0, time1, n_box6 >= K_MANY_BOXES);
    const std::vector<RtNode> tokens = orc_lbvh::build_tokens(boxes.data(), objects.size());
    size_t at = 0;
    std::shared_ptr<BVHNode> root = std::static_pointer_cast<BVHNode>(lbvh_subtree(tokens, at, objects, time0, time1));
    double lo, hi;
    root->content_held = root->interval_covers_motion(time0, time1, lo, hi);
    for (const HitPtr& o : objects) root->content_held = root->content_held && o->holds_content();
    return root;
}

inline std::shared_ptr<BVHNode> BVHNode_new(HittableList& list, double time0, double time1) {  // :94-97
    if (list.objects.empty()) throw std::runtime_error("BVHNode::new on an empty list");
    struct ManyBoxes {  // for the builders below, restored on the way out (trees nest: a list may hold a tree)
        bool was;
        explicit ManyBoxes(const std::vector<HitPtr>& objects) : was(bvh_many_boxes) {
            size_t n = 0;
            for (const HitPtr& o : objects) n += o->is_box6() ? 1 : 0;
            bvh_many_boxes = n >= K_MANY_BOXES;
        }
        ~ManyBoxes() { bvh_many_boxes = was; }
    } scope(list.objects);
    if (bvh_lbvh && list.objects.size() >= 2) {
        // the scene stream moves on as if the reference builder had drawn its axis per node (:24), so that whatever the
        // scene draws afterwards is the same under either tree policy
        struct Nodes { static uint64_t of(size_t n) { return n <= 2 ? 1 : 1 + of(n / 2) + of(n - n / 2); } };
        for (uint64_t k = Nodes::of(list.objects.size()); k > 0; --k) (void)random_int(0, 2);
        return lbvh_from_list(list.objects, time0, time1);
    }
    return new_from_list(list.objects, 0, list.objects.size(), time0, time1);
}

// ------------------------------------------------------------------- camera.rs
struct CameraInfo { double vfov, aspect_ratio, aperture, focus_dist; };  // :19-24
struct Camera {                                                          // :6-17
    Point orig, lower_left_corner;
    Vec3 horizontal, vertical, u, v, _w;
    double lens_radius, time0, time1;
    Camera() = default;
    Camera(Point lookfrom, Point lookat, Vec3 vup, CameraInfo info, double t0, double t1) {  // :27-59
        double theta = degrees_to_radians(info.vfov);
        double h = std::tan(theta / 2.0);
        double viewport_height = 2.0 * h;
        double viewport_width = info.aspect_ratio * viewport_height;
        Vec3 w = (lookfrom - lookat).unit();
        u = Vec3::cross(vup, w).unit();
        v = Vec3::cross(w, u);
        _w = w;
        orig = lookfrom;
        horizontal = u * info.focus_dist * viewport_width;
        vertical = v * info.focus_dist * viewport_height;
        lower_left_corner = orig - horizontal / 2.0 - vertical / 2.0 - w * info.focus_dist;
        lens_radius = info.aperture / 2.0;
        time0 = t0;
        time1 = t1;
    }
    Ray get_ray(double s, double t) const {  // :61-71
        tls.rng.set(tls.pixel, tls.sample, 0, STREAM_LENS);
        Vec3 rd = random_in_unit_disk() * lens_radius;
        Vec3 offset = u * rd.x + v * rd.y;
        tls.rng.set(tls.pixel, tls.sample, 0, STREAM_TIME);
        double tm = random_double(time0, time1);
        return Ray{orig + offset, lower_left_corner + horizontal * s + vertical * t - orig - offset, tm};
    }
};

// -------------------------------------------------------------------- color.rs
struct RayRecord {
    Ray r;
    uint32_t pixel, sample, bounce;
};

inline Color ray_color(const Ray& r, const Color& background, const Hittable& world, int depth, int max_depth) {  // :34-63
    HitRecord rec;
    if (depth <= 0) return Color(0.0, 0.0, 0.0);
    tls.bounce = (uint32_t)(max_depth - depth);
    if (tls.recorder && tls.bounce <= tls.record_max_bounce) tls.recorder->push_back(RayRecord{r, tls.pixel, tls.sample, tls.bounce});
    ++tls.n_rays;
    if (!world.hit(r, 0.0001, INF, rec)) return background;
    Ray scattered{Vec3::ones(), Vec3::ones(), 0.0};
    Color attenuation(0.0, 0.0, 0.0);
    Color emitted = rec.mat_ptr->emitted(rec.u, rec.v, rec.p);
    tls.rng.set(tls.pixel, tls.sample, tls.bounce, STREAM_SCATTER);
    if (!rec.mat_ptr->scatter(r, rec, attenuation, scattered)) return emitted;
    return emitted + Vec3::elemul(attenuation, ray_color(scattered, background, world, depth - 1, max_depth));
}

// One path sample of pixel (x, y): src/main.rs:206-210.
inline Color sample_pixel(const Camera& cam, const Hittable& world, const Color& background, uint32_t width,
                          uint32_t height, uint32_t x, uint32_t y, uint32_t s, int max_depth) {
    tls.pixel = y * width + x;
    tls.sample = s;
    tls.rng.set(tls.pixel, tls.sample, 0, STREAM_JITTER);
    double ju = random_double(0.0, 1.0);
    double jv = random_double(0.0, 1.0);
    double u = ((double)x + ju) / (double)(width - 1);
    double v = ((double)(height - y) + jv) / (double)(height - 1);  // row off-by-one kept (:207-208)
    Ray r = cam.get_ray(u, v);
    return ray_color(r, background, world, max_depth, max_depth);
}

// src/main.rs:212-225 (= color.rs:18-31): gamma 2, clamp, saturating u8 cast.
inline uint8_t f2u8(double f) {
    if (!(f > 0.0)) return 0;
    if (f >= 255.0) return 255;
    return (uint8_t)f;
}
inline void resolve_pixel(const double sum[3], uint64_t samples_per_pixel, uint8_t out[3]) {
    double scale = 1.0 / (double)samples_per_pixel;
    for (int c = 0; c < 3; ++c) {
        double r = std::sqrt(scale * sum[c]);
        out[c] = f2u8(clamp(r, 0.0, 0.999) * 255.0);
    }
}

}  // namespace orc
