// Host-side scene object model: the reference's three traits and their
// constructors restated in C++, with `hit`/`scatter`/`value` removed (those run on
// the GPU) and one added method per trait that writes the object into the flat
// tables of include/rt/abi.h:
//     Hittable::flatten   <- trait Hittable  (src/hittable.rs:43-46)
//     Material::describe  <- trait Material  (src/material.rs:8-20)
//     Texture::describe   <- trait Texture   (src/texture.rs:8-10)
// bounding_box keeps the reference's exact arithmetic because BVHNode::new
// (src/bvh.rs:17-98) sorts and bounds with it.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <limits>
#include <future>
#include <memory>
#include <unordered_map>
#include <stdexcept>
#include <vector>

#include "../../../include/rt/abi.h"
#include "../rng.h"

namespace rth {

constexpr double INF = std::numeric_limits<double>::infinity();
constexpr double PI = 3.14159265358979323846264338327950288;  // src/rtweekend.rs:4

// ---- math (src/vec3.rs) ----------------------------------------------------
struct Vec3 {
    double x = 0, y = 0, z = 0;
    Vec3() = default;
    Vec3(double a, double b, double c) : x(a), y(b), z(c) {}
    double operator[](int i) const { return i == 1 ? y : (i == 2 ? z : x); }  // src/vec3.rs:252-262
    double& at(int i) { return i == 0 ? x : (i == 1 ? y : z); }
    double squared_length() const { return x * x + y * y + z * z; }     // :27-29
    double length() const { return std::sqrt(x * x + y * y + z * z); }  // :277-279
    Vec3 unit() const {                                                  // :283-294
        double len = length();
        if (len == 0.0) throw std::domain_error("unit() of a zero-length vector");
        return Vec3(x / len, y / len, z / len);
    }
};
using Point = Vec3;
using Color = Vec3;
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator*(Vec3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3 operator/(Vec3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline Vec3 elemul(Vec3 a, Vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }  // :224-230
inline Vec3 cross(Vec3 a, Vec3 b) {                                                // :232-238
    return {a.y * b.z - a.z * b.y, b.x * a.z - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline void put3(double* d, Vec3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }

// ---- per-scene construction state ------------------------------------------
// The scene stream of the counter-based RNG and the creation-order id counters.
struct SceneCtx {
    rt::Rng rng;
    uint32_t next_prim = 0, next_inst = 0, next_medium = 0;
    explicit SceneCtx(uint64_t seed) { rng.set_scene(seed); }
};
inline thread_local SceneCtx* g_ctx = nullptr;
struct CtxScope {
    SceneCtx* prev;
    explicit CtxScope(SceneCtx* c) : prev(g_ctx) { g_ctx = c; }
    ~CtxScope() { g_ctx = prev; }
};
inline SceneCtx& ctx() {
    if (!g_ctx) throw std::logic_error("scene objects must be created inside a scene context");
    return *g_ctx;
}
inline double random_double(double min, double max) { return ctx().rng.uniform(min, max); }  // src/rtweekend.rs:11-14
inline int random_int(int min, int max) { return (int)random_double((double)min, (double)(max + 1)); }  // :17-20
inline Vec3 random_vec3(double min, double max) {  // Vec3::random, src/vec3.rs:31-37
    double a = random_double(min, max), b = random_double(min, max), c = random_double(min, max);
    return {a, b, c};
}
inline double degrees_to_radians(double degrees) { return degrees * PI / 180.0; }  // src/rtweekend.rs:6-8

// ---- boxes (src/aabb.rs) ----------------------------------------------------
struct AABB {
    Point _min{1, 1, 1}, _max{1, 1, 1};
};
inline AABB surrounding_box(const AABB& a, const AABB& b) {  // :33-50
    AABB r;
    r._min = {std::fmin(a._min.x, b._min.x), std::fmin(a._min.y, b._min.y), std::fmin(a._min.z, b._min.z)};
    r._max = {std::fmax(a._max.x, b._max.x), std::fmax(a._max.y, b._max.y), std::fmax(a._max.z, b._max.z)};
    return r;
}
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

// ---- flat table builder ----------------------------------------------------
// object address -> table index. Open addressing with linear probing: a 10 M-triangle scene makes 10 M insertions,
// which cost seconds in a node-based map.
class PtrIndexMap {
    struct Slot { const void* key; uint32_t value; };
    std::vector<Slot> slots_;
    size_t used_ = 0;
    static size_t hash(const void* p) { return (size_t)(((uint64_t)(uintptr_t)p >> 4) * 0x9E3779B97F4A7C15ull >> 20); }
    Slot* probe(const void* key) {
        const size_t mask = slots_.size() - 1;
        size_t i = hash(key) & mask;
        while (slots_[i].key && slots_[i].key != key) i = (i + 1) & mask;
        return &slots_[i];
    }
    void grow() {
        std::vector<Slot> old;
        old.swap(slots_);
        slots_.assign(old.empty() ? 1024 : old.size() * 4, Slot{nullptr, 0});
        for (const Slot& s : old)
            if (s.key) *probe(s.key) = s;
    }

public:
    using iterator = Slot*;
    struct Found { const void* first; uint32_t second; };
    iterator end() { return nullptr; }
    iterator find(const void* key) {
        if (slots_.empty()) return nullptr;
        Slot* s = probe(key);
        return s->key ? s : nullptr;
    }
    uint32_t& operator[](const void* key) {
        if ((used_ + 1) * 2 > slots_.size()) grow();
        Slot* s = probe(key);
        if (!s->key) { s->key = key; s->value = 0; ++used_; }
        return s->value;
    }
    void clear() { std::vector<Slot>().swap(slots_); used_ = 0; }
};

inline std::atomic<uint64_t> g_flat_generation{0};

struct Flat {
    const uint64_t gen = ++g_flat_generation;  // tells this flattening apart from every other one (see Hittable)
    std::vector<RtNode> nodes;
    std::vector<RtSphere> spheres;
    std::vector<RtMovingSphere> msphere;
    std::vector<RtRect> rects;
    std::vector<RtBox6> boxes;
    std::vector<RtTriangle> triangles;
    std::vector<RtXform> xforms;
    std::vector<RtMedium> media;
    std::vector<RtMaterial> materials;
    std::vector<RtTexture> textures;
    std::vector<RtPerlin> perlins;
    std::vector<RtImage> images;
    std::vector<uint8_t> texels;
    PtrIndexMap seen;  // object -> table index, so shared objects are stored once
    std::vector<uint32_t> xf_stack;
    bool in_medium = false;
    // the time interval every BVHNode around the object being flattened was built over (their intersection), and
    // whether some object moves outside it: its boxes then do not bound it (RT_DESC_BOXES_MAY_NOT_BOUND)
    double cover_t0 = -std::numeric_limits<double>::infinity(), cover_t1 = std::numeric_limits<double>::infinity();
    bool boxes_may_not_bound = false;
    // the box of the first BVHNode flattened, the world's: every ray that starts on something in the scene starts
    // inside it (see BVHNode::flatten)
    AABB world_box;
    bool have_world_box = false;

    uint32_t push_node(uint32_t kind, uint32_t index, uint32_t link = 0) {
        if (index >= (1u << 28)) throw std::length_error("table index exceeds 28 bits");
        RtNode n{};
        n.meta = kind | (index << 4);
        n.link = link;
        nodes.push_back(n);
        return (uint32_t)nodes.size() - 1;
    }
};

struct Texture {
    virtual ~Texture() = default;
    virtual uint32_t describe(Flat& f) const = 0;
};
using TexPtr = std::shared_ptr<Texture>;

struct SolidColor : Texture {  // src/texture.rs:12-26
    Color color_value;
    explicit SolidColor(Color c) : color_value(c) {}
    uint32_t describe(Flat& f) const override {
        auto it = f.seen.find(this);
        if (it != f.seen.end()) return it->value;
        RtTexture t{};
        t.kind = RT_T_SOLID;
        put3(t.a, color_value);
        f.textures.push_back(t);
        return f.seen[this] = (uint32_t)f.textures.size() - 1;
    }
};

struct CheckerTexture : Texture {  // src/texture.rs:28-51: even = c1, odd = c2
    Color even, odd;
    CheckerTexture(Color c1, Color c2) : even(c1), odd(c2) {}
    uint32_t describe(Flat& f) const override {
        auto it = f.seen.find(this);
        if (it != f.seen.end()) return it->value;
        RtTexture t{};
        t.kind = RT_T_CHECKER;
        put3(t.a, even);
        put3(t.b, odd);
        f.textures.push_back(t);
        return f.seen[this] = (uint32_t)f.textures.size() - 1;
    }
};

struct Perlin {  // src/perlin.rs:6-42
    RtPerlin tab;
    static void permute(int32_t* tmp, int n) {  // :14-20
        for (int k = 0; k < n - 1; ++k) {
            int i = n - 1 - k;
            int target = random_int(0, i);
            std::swap(tmp[i], tmp[target]);
        }
    }
    static void generate_perm(int32_t* tmp) {  // :21-30
        for (int i = 0; i < 256; ++i) tmp[i] = i;
        permute(tmp, 256);
    }
    Perlin() {  // :31-42
        for (int i = 0; i < 256; ++i) put3(tab.ran_vec[i], random_vec3(-1.0, 1.0).unit());
        generate_perm(tab.perm_x);
        generate_perm(tab.perm_y);
        generate_perm(tab.perm_z);
    }
};

struct NoiseTexture : Texture {  // src/texture.rs:53-73
    Perlin noise;
    double scale;
    explicit NoiseTexture(double sc) : noise(), scale(sc) {}
    uint32_t describe(Flat& f) const override {
        auto it = f.seen.find(this);
        if (it != f.seen.end()) return it->value;
        f.perlins.push_back(noise.tab);
        RtTexture t{};
        t.kind = RT_T_NOISE;
        t.aux = (uint32_t)f.perlins.size() - 1;
        t.a[0] = scale;
        f.textures.push_back(t);
        return f.seen[this] = (uint32_t)f.textures.size() - 1;
    }
};

struct ImageTexture : Texture {  // src/texture.rs:75-90; takes decoded RGB8 rows instead of a file name
    uint32_t width, height;
    std::vector<uint8_t> rgb;
    ImageTexture(const uint8_t* p, uint32_t w, uint32_t h) : width(w), height(h), rgb(p, p + (size_t)w * h * 3) {
        if (w == 0 || h == 0) throw std::invalid_argument("empty image texture");
    }
    uint32_t describe(Flat& f) const override {
        auto it = f.seen.find(this);
        if (it != f.seen.end()) return it->value;
        RtImage im{};
        im.width = width;
        im.height = height;
        im.offset = f.texels.size();
        f.texels.insert(f.texels.end(), rgb.begin(), rgb.end());
        f.images.push_back(im);
        RtTexture t{};
        t.kind = RT_T_IMAGE;
        t.aux = (uint32_t)f.images.size() - 1;
        f.textures.push_back(t);
        return f.seen[this] = (uint32_t)f.textures.size() - 1;
    }
};

struct Material {
    virtual ~Material() = default;
    virtual RtMaterial fill(Flat& f) const = 0;
    uint32_t describe(Flat& f) const {
        auto it = f.seen.find(this);
        if (it != f.seen.end()) return it->value;
        RtMaterial m = fill(f);
        f.materials.push_back(m);
        return f.seen[this] = (uint32_t)f.materials.size() - 1;
    }
};
using MatPtr = std::shared_ptr<Material>;

struct Lambertian : Material {  // src/material.rs:22-32
    TexPtr albedo;
    explicit Lambertian(Color a) : albedo(std::make_shared<SolidColor>(a)) {}
    explicit Lambertian(TexPtr a) : albedo(std::move(a)) {}
    RtMaterial fill(Flat& f) const override { RtMaterial m{}; m.kind = RT_M_LAMBERTIAN; m.tex = albedo->describe(f); return m; }
};
struct Metal : Material {  // src/material.rs:53-65 (fuzz clamped to <= 1 at :62)
    Color albedo;
    double fuzz;
    Metal(Color a, double fz) : albedo(a), fuzz(fz < 1.0 ? fz : 1.0) {}
    RtMaterial fill(Flat&) const override { RtMaterial m{}; m.kind = RT_M_METAL; put3(m.albedo, albedo); m.param = fuzz; return m; }
};
struct Dielectric : Material {  // src/material.rs:86-94
    double ref_idx;
    explicit Dielectric(double ri) : ref_idx(ri) {}
    RtMaterial fill(Flat&) const override { RtMaterial m{}; m.kind = RT_M_DIELECTRIC; m.param = ref_idx; return m; }
};
struct DiffuseLight : Material {  // src/material.rs:150-164
    TexPtr emit;
    explicit DiffuseLight(TexPtr a) : emit(std::move(a)) {}
    explicit DiffuseLight(Color c) : emit(std::make_shared<SolidColor>(c)) {}
    RtMaterial fill(Flat& f) const override { RtMaterial m{}; m.kind = RT_M_DIFFUSE_LIGHT; m.tex = emit->describe(f); return m; }
};
struct Isotropic : Material {  // src/material.rs:181-193
    TexPtr albedo;
    explicit Isotropic(TexPtr a) : albedo(std::move(a)) {}
    explicit Isotropic(Color c) : albedo(std::make_shared<SolidColor>(c)) {}
    RtMaterial fill(Flat& f) const override { RtMaterial m{}; m.kind = RT_M_ISOTROPIC; m.tex = albedo->describe(f); return m; }
};

struct Hittable {
    virtual ~Hittable() = default;
    virtual bool bounding_box(double t0, double t1, AABB& out) const = 0;
    virtual void flatten(Flat& f) const = 0;
    // Called when BVHNode::new makes this object a direct child of a node built over (t0, t1). Translate / RotateY
    // keep the box they presented then (see CullBox).
    // many_boxes: the list the tree was made from holds at least kManyBoxes boxes of six sides
    virtual void adopted_by_bvh(double, double, bool /*many_boxes*/) const {}
    virtual bool is_box6() const { return false; }
    // the object a chain of Translate / RotateY wraps, and whether an object is one primitive (one stream entry)
    virtual const Hittable& unwrap() const { return *this; }
    virtual bool one_entry() const { return false; }
    // The hull of the motion intervals of the moving spheres at or below this object: [lo, hi] is widened to hold it
    // (lo = +inf, hi = -inf: nothing below moves).
    virtual void motion_hull(double& /*lo*/, double& /*hi*/) const {}
    // A box that holds the object for every ray time in [t0, t1], computed from the objects themselves. Unlike
    // bounding_box it does not stop at a BVHNode's stored box (which holds its content over the interval the node was
    // built with only). false: not known (behind a Translate / RotateY).
    virtual bool bounds_over(double t0, double t1, AABB& out) const { return bounding_box(t0, t1, out); }
    // Does the box this object presents to a tree built over an interval that covers its motion hull hold everything
    // below it? Not when a tree (or a RotateY, whose box is taken over (0, 1) at construction, src/hittable.rs:321) below
    // was built over an interval that the motion of its own content reaches out of: the box of a BVHNode is the union of
    // its children's stored boxes (src/bvh.rs:80-88, :113-116), so a box that does not hold makes every box above it
    // one that does not hold either, whatever interval those were built over.
    virtual bool holds_content() const { return true; }

protected:
    // Where this object's row went in the flattening under way, so that an object reached twice (shared Arc in the
    // reference) is stored once. Kept in the object instead of a map keyed by address: with 10 M triangles the map is
    // half a gigabyte of randomly accessed memory and costs seconds. One flattening at a time per object.
    mutable uint64_t flat_gen_ = 0;
    mutable uint32_t flat_idx_ = 0;
    bool flat_lookup(const Flat& f, uint32_t& idx) const { idx = flat_idx_; return flat_gen_ == f.gen; }
    uint32_t flat_store(const Flat& f, uint32_t idx) const { flat_gen_ = f.gen; flat_idx_ = idx; return idx; }
};
using HitPtr = std::shared_ptr<Hittable>;

// The reference tests no box of its own before entering a Translate / RotateY that is a leaf of a BVH: a ray that
// passes the parent node's box (the union with the sibling's) is moved into the wrapper's space and back even when
// it misses the wrapper's own box. Entering and leaving a wrapper are the most expensive entries of the stream and
// the least coherent ones, so the flattener puts a box entry in front of such a wrapper when it holds more than one
// primitive: the box the wrapper presented to BVHNode::new. Pure culling: the closest hit is the same.
inline bool cull_boxes_enabled() {
    static const bool off = std::getenv("RT_NO_CULL_BOXES") != nullptr;  // developer knob for A/B measurements
    return !off;
}
inline double cull_pad(double mn, double mx) {
    const double scale = std::fmax(std::fmax(std::fabs(mn), std::fabs(mx)), mx - mn);
    return std::fmax(scale, 1e-30) * 0x1p-30;
}
struct CullBox {
    mutable AABB box;
    mutable bool has = false;
    // Not around a single primitive: that is tested in one step inside its wrappers, without entering and leaving
    // them, and a box in front of it only adds a step (measured on the Cornell box: 15 % slower with one).
    // Only around content that does not move and whose own boxes hold it: the box the wrapper presented to BVHNode::new
    // is a box the reference never tests, so it may only cull what cannot be hit, for rays of every time.
    void adopt(const Hittable& self, double t0, double t1) const {
        double lo = std::numeric_limits<double>::infinity(), hi = -lo;
        self.motion_hull(lo, hi);
        has = !self.unwrap().one_entry() && !(lo <= hi) && self.holds_content() && self.bounding_box(t0, t1, box);
    }
    void adopt_primitive(const Hittable& self, double t0, double t1) const { has = self.bounding_box(t0, t1, box); }
    // the box entry in front of the wrapper's entries; returns its position or NONE
    uint32_t open(Flat& f) const {
        if (!has || f.in_medium || !cull_boxes_enabled()) return 0xFFFFFFFFu;
        const uint32_t at = f.push_node(RT_N_INNER, 0);
        // Padded before it is rounded outward: no face of the object may coincide with a face of this box, or a hit
        // on that face at exactly the running closest t (two objects hit at the same t, where the reference lets the
        // later one win) would be culled by the test's `far <= near`. The pad is relative to the box (2^-30 of its
        // size and position): a million times the rounding noise of the f64 box test, and below one unit in the last
        // place of the f32 planes, so that the box stays as tight as f32 allows. Tight matters: a ray that leaves a face
        // of a box of six sides starts inside that box's padded culling box, and is only spared the box's six plane tests
        // if it is out of the padding before tmin = 0.0001 (with 2^-20 it never was: every such ray tested the box it
        // had just left).
        for (int a = 0; a < 3; ++a) {
            const double pad = cull_pad(box._min[a], box._max[a]);
            f.nodes[at].lo[a] = f32_down(box._min[a] - pad);
            f.nodes[at].hi[a] = f32_up(box._max[a] + pad);
        }
        return at;
    }
    static void close(Flat& f, uint32_t at) { if (at != 0xFFFFFFFFu) f.nodes[at].link = (uint32_t)f.nodes.size(); }
};

struct Sphere : Hittable {  // src/hittable.rs:48-62, 115-121
    bool one_entry() const override { return true; }
    Point center;
    double radius;
    MatPtr mat_ptr;
    uint32_t id;
    Sphere(Point c, double r, MatPtr m) : center(c), radius(r), mat_ptr(std::move(m)), id(ctx().next_prim++) {}
    bool bounding_box(double, double, AABB& out) const override {
        out._min = center - Vec3(radius, radius, radius);
        out._max = center + Vec3(radius, radius, radius);
        return true;
    }
    void flatten(Flat& f) const override {
        uint32_t idx;
        if (!flat_lookup(f, idx)) {
            RtSphere s{};
            put3(s.c, center); s.r = radius; s.mat = mat_ptr->describe(f); s.prim_id = id;
            f.spheres.push_back(s);
            idx = flat_store(f, (uint32_t)f.spheres.size() - 1);
        }
        f.push_node(RT_N_SPHERE, idx);
    }
};

struct MovingSphere : Hittable {  // src/hittable.rs:187-212, 248-259
    bool one_entry() const override { return true; }
    Point center0, center1;
    double time0, time1, radius;
    MatPtr mat_ptr;
    uint32_t id;
    MovingSphere(Point c0, Point c1, double t0, double t1, double r, MatPtr m)
        : center0(c0), center1(c1), time0(t0), time1(t1), radius(r), mat_ptr(std::move(m)), id(ctx().next_prim++) {}
    Point center(double time) const { return center0 + (center1 - center0) * ((time - time0) / (time1 - time0)); }
    bool bounding_box(double t0, double t1, AABB& out) const override {
        AABB b0, b1;
        b0._min = center(t0) - Vec3(radius, radius, radius); b0._max = center(t0) + Vec3(radius, radius, radius);
        b1._min = center(t1) - Vec3(radius, radius, radius); b1._max = center(t1) + Vec3(radius, radius, radius);
        out = surrounding_box(b0, b1);
        return true;
    }
    bool moves() const { return center0.x != center1.x || center0.y != center1.y || center0.z != center1.z; }
    void motion_hull(double& lo, double& hi) const override {
        if (moves()) { lo = std::fmin(lo, std::fmin(time0, time1)); hi = std::fmax(hi, std::fmax(time0, time1)); }
    }
    void flatten(Flat& f) const override {
        // MovingSphere::bounding_box(t0, t1) bounds the sphere for times in [t0, t1] only (src/hittable.rs:248-259)
        const bool moves = this->moves();
        if (moves && (std::fmin(time0, time1) < f.cover_t0 || std::fmax(time0, time1) > f.cover_t1)) f.boxes_may_not_bound = true;
        uint32_t idx;
        if (!flat_lookup(f, idx)) {
            RtMovingSphere s{};
            put3(s.c0, center0); put3(s.c1, center1); s.t0 = time0; s.t1 = time1; s.r = radius;
            s.mat = mat_ptr->describe(f); s.prim_id = id;
            f.msphere.push_back(s);
            idx = flat_store(f, (uint32_t)f.msphere.size() - 1);
        }
        f.push_node(RT_N_MSPHERE, idx);
    }
};

// XYRect / XZRect / YZRect (src/aarect.rs:7-27, 58-78, 109-129) as one class; axis 0 = XY, 1 = XZ, 2 = YZ.
struct AARect : Hittable {
    bool one_entry() const override { return true; }
    int axis;
    double a0, a1, b0, b1, k;
    MatPtr mp;
    uint32_t id;
    AARect(int axis_, double a0_, double a1_, double b0_, double b1_, double k_, MatPtr m)
        : axis(axis_), a0(a0_), a1(a1_), b0(b0_), b1(b1_), k(k_), mp(std::move(m)), id(ctx().next_prim++) {}
    bool bounding_box(double, double, AABB& out) const override {  // :49-55, 100-106, 151-157
        if (axis == 0) { out._min = {a0, b0, k - 0.0001}; out._max = {a1, b1, k + 0.0001}; }
        else if (axis == 1) { out._min = {a0, k - 0.0001, b0}; out._max = {a1, k + 0.0001, b1}; }
        else { out._min = {k - 0.0001, a0, b0}; out._max = {k + 0.0001, a1, b1}; }
        return true;
    }
    void flatten(Flat& f) const override {
        uint32_t idx;
        if (!flat_lookup(f, idx)) {
            RtRect r{};
            r.a0 = a0; r.a1 = a1; r.b0 = b0; r.b1 = b1; r.k = k; r.axis = (uint32_t)axis; r.mat = mp->describe(f); r.prim_id = id;
            f.rects.push_back(r);
            idx = flat_store(f, (uint32_t)f.rects.size() - 1);
        }
        f.push_node(RT_N_RECT, idx);
    }
};
inline HitPtr XYRect(double x0, double x1, double y0, double y1, double k, MatPtr m) { return std::make_shared<AARect>(0, x0, x1, y0, y1, k, std::move(m)); }
inline HitPtr XZRect(double x0, double x1, double z0, double z1, double k, MatPtr m) { return std::make_shared<AARect>(1, x0, x1, z0, z1, k, std::move(m)); }
inline HitPtr YZRect(double y0, double y1, double z0, double z1, double k, MatPtr m) { return std::make_shared<AARect>(2, y0, y1, z0, z1, k, std::move(m)); }

struct Box6 : Hittable {  // src/box6.rs:9-85; the six rects of :22-71 are implied by p0, p1
    bool one_entry() const override { return true; }
    Point box_min, box_max;
    MatPtr mat;
    uint32_t id;
    Box6(Point p0, Point p1, MatPtr m) : box_min(p0), box_max(p1), mat(std::move(m)), id(ctx().next_prim++) {}
    bool bounding_box(double, double, AABB& out) const override { out._min = box_min; out._max = box_max; return true; }
    CullBox cull;
    bool is_box6() const override { return true; }
    void adopted_by_bvh(double t0, double t1, bool many_boxes) const override {
        if (many_boxes) cull.adopt_primitive(*this, t0, t1); else cull.has = false;
    }
    void flatten(Flat& f) const override {
        const uint32_t cull_at = cull.open(f);
        uint32_t idx;
        if (!flat_lookup(f, idx)) {
            RtBox6 b{};
            put3(b.p0, box_min); put3(b.p1, box_max); b.mat = mat->describe(f); b.prim_id = id;
            f.boxes.push_back(b);
            idx = flat_store(f, (uint32_t)f.boxes.size() - 1);
        }
        f.push_node(RT_N_BOX6, idx);
        CullBox::close(f, cull_at);
    }
};

struct Triangle : Hittable {  // src/triangle.rs:7-23, 70-83 (argument order point1, point2, point0)
    bool one_entry() const override { return true; }
    Point point1, point2, point0;
    MatPtr mp;
    uint32_t id;
    Triangle(Point p1, Point p2, Point p0, MatPtr m) : point1(p1), point2(p2), point0(p0), mp(std::move(m)), id(ctx().next_prim++) {}
    bool bounding_box(double, double, AABB& out) const override {
        out._min = {std::fmin(std::fmin(point0.x, point1.x), point2.x), std::fmin(std::fmin(point0.y, point1.y), point2.y),
                    std::fmin(std::fmin(point0.z, point1.z), point2.z)};
        out._max = {std::fmax(std::fmax(point0.x, point1.x), point2.x), std::fmax(std::fmax(point0.y, point1.y), point2.y),
                    std::fmax(std::fmax(point0.z, point1.z), point2.z)};
        return true;
    }
    void flatten(Flat& f) const override {
        uint32_t idx;
        if (!flat_lookup(f, idx)) {
            RtTriangle t{};
            put3(t.p0, point0);
            put3(t.e1, point1 - point0);  // edge1, src/triangle.rs:27
            put3(t.e2, point2 - point0);  // edge2, :28
            t.mat = mp->describe(f); t.prim_id = id;
            f.triangles.push_back(t);
            idx = flat_store(f, (uint32_t)f.triangles.size() - 1);
        }
        f.push_node(RT_N_TRIANGLE, idx);
    }
};

struct HittableList : Hittable {  // src/hittable.rs:124-185
    std::vector<HitPtr> objects;
    void add(HitPtr o) { objects.push_back(std::move(o)); }
    void clear() { objects.clear(); }
    bool bounding_box(double t0, double t1, AABB& out) const override {
        if (objects.empty()) return false;
        AABB temp;
        bool first = true;
        for (const auto& o : objects) {
            if (!o->bounding_box(t0, t1, temp)) return false;
            out = first ? temp : surrounding_box(out, temp);
            first = false;
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
    // HittableList::hit (:151-164) keeps a running closest t over its objects in order, which is what
    // walking their entries front to back does.
    void flatten(Flat& f) const override { for (const auto& o : objects) o->flatten(f); }
};

inline void push_xform(Flat& f, const RtXform& proto, const Hittable& child) {
    if (f.xf_stack.size() >= 4) throw std::length_error("more than 4 nested Translate/RotateY wrappers");
    RtXform x = proto;
    const uint32_t self = (uint32_t)f.xforms.size();
    x.depth = (uint32_t)f.xf_stack.size() + 1;
    for (uint32_t i = 0; i < 4; ++i) x.anc[i] = RT_NONE;
    for (size_t i = 0; i < f.xf_stack.size(); ++i) x.anc[i] = f.xf_stack[i];
    x.anc[x.depth - 1] = self;
    f.xforms.push_back(x);
    f.push_node(RT_N_XF_ENTER, self);
    f.xf_stack.push_back(self);
    child.flatten(f);
    f.xf_stack.pop_back();
    f.push_node(RT_N_XF_EXIT, self);
}

struct Translate : Hittable {  // src/hittable.rs:262-304
    HitPtr ptr;
    Vec3 offset;
    uint32_t id;
    Translate(HitPtr p, Vec3 displacement) : ptr(std::move(p)), offset(displacement), id(ctx().next_inst++) {}
    bool bounding_box(double t0, double t1, AABB& out) const override {
        if (!ptr->bounding_box(t0, t1, out)) return false;
        out._min = out._min + offset;
        out._max = out._max + offset;
        return true;
    }
    CullBox cull;
    const Hittable& unwrap() const override { return ptr->unwrap(); }
    void motion_hull(double& lo, double& hi) const override { ptr->motion_hull(lo, hi); }
    bool holds_content() const override { return ptr->holds_content(); }
    bool bounds_over(double, double, AABB&) const override { return false; }
    void adopted_by_bvh(double t0, double t1, bool) const override { cull.adopt(*this, t0, t1); }
    void flatten(Flat& f) const override {
        RtXform x{};
        x.kind = 0; x.inst_id = id; put3(x.offset, offset);
        const uint32_t at = cull.open(f);
        push_xform(f, x, *ptr);
        CullBox::close(f, at);
    }
};

struct RotateY : Hittable {  // src/hittable.rs:306-392
    HitPtr ptr;
    double sin_theta, cos_theta;
    bool hasbox;
    AABB bbox;
    uint32_t id;
    RotateY(HitPtr p, double angle) : ptr(std::move(p)), id(ctx().next_inst++) {  // :315-350
        double radians = degrees_to_radians(angle);
        sin_theta = std::sin(radians);
        cos_theta = std::cos(radians);
        Point mn(INF, INF, INF), mx(-INF, -INF, -INF);
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
                        mn.at(c) = std::fmin(mn[c], tester[c]);
                        mx.at(c) = std::fmax(mx[c], tester[c]);
                    }
                }
        bbox._min = mn;
        bbox._max = mx;
    }
    bool bounding_box(double, double, AABB& out) const override { out = bbox; return hasbox; }
    CullBox cull;
    const Hittable& unwrap() const override { return ptr->unwrap(); }
    void motion_hull(double& lo, double& hi) const override { ptr->motion_hull(lo, hi); }
    bool holds_content() const override {  // the box is taken over (0, 1) whatever tree the wrapper ends up in (:321)
        double lo = std::numeric_limits<double>::infinity(), hi = -lo;
        ptr->motion_hull(lo, hi);
        return ptr->holds_content() && (!(lo <= hi) || (lo >= 0.0 && hi <= 1.0));
    }
    bool bounds_over(double, double, AABB&) const override { return false; }
    void adopted_by_bvh(double t0, double t1, bool) const override { cull.adopt(*this, t0, t1); }
    void flatten(Flat& f) const override {
        RtXform x{};
        x.kind = 1; x.inst_id = id; x.sin_theta = sin_theta; x.cos_theta = cos_theta;
        const uint32_t at = cull.open(f);
        push_xform(f, x, *ptr);
        CullBox::close(f, at);
    }
};

struct ConstantMedium : Hittable {  // src/constant_medium.rs:10-31, 91-93
    HitPtr boundary;
    MatPtr phase_function;
    double neg_inv_density;
    uint32_t id, rng_id;
    ConstantMedium(HitPtr b, double d, TexPtr a)
        : boundary(std::move(b)), phase_function(std::make_shared<Isotropic>(std::move(a))), neg_inv_density(-1.0 / d),
          id(ctx().next_prim++), rng_id(ctx().next_medium++) {}
    ConstantMedium(HitPtr b, double d, Color c)
        : boundary(std::move(b)), phase_function(std::make_shared<Isotropic>(c)), neg_inv_density(-1.0 / d),
          id(ctx().next_prim++), rng_id(ctx().next_medium++) {}
    bool bounding_box(double t0, double t1, AABB& out) const override { return boundary->bounding_box(t0, t1, out); }
    void motion_hull(double& lo, double& hi) const override { boundary->motion_hull(lo, hi); }
    bool holds_content() const override { return boundary->holds_content(); }
    bool bounds_over(double t0, double t1, AABB& out) const override { return boundary->bounds_over(t0, t1, out); }
    void flatten(Flat& f) const override {
        if (f.in_medium) throw std::invalid_argument("a ConstantMedium inside the boundary of another is not supported");
        if (rng_id >= 0xFFF0u) throw std::length_error("too many ConstantMedium objects");
        uint32_t idx;
        if (!flat_lookup(f, idx)) {
            RtMedium m{};
            m.neg_inv_density = neg_inv_density; m.mat = phase_function->describe(f); m.rng_id = rng_id; m.prim_id = id;
            f.media.push_back(m);
            idx = flat_store(f, (uint32_t)f.media.size() - 1);
        }
        const uint32_t at = f.push_node(RT_N_MEDIUM, idx);
        f.in_medium = true;
        boundary->flatten(f);
        f.in_medium = false;
        f.nodes[at].link = (uint32_t)f.nodes.size();
    }
};

// Narrows the time interval the flattener knows the enclosing trees were built over, for the duration of a subtree.
struct CoverScope {
    Flat& f;
    double t0, t1;
    CoverScope(Flat& f_, float time0, float time1) : f(f_), t0(f_.cover_t0), t1(f_.cover_t1) {
        f.cover_t0 = std::fmax(f.cover_t0, (double)time0);
        f.cover_t1 = std::fmin(f.cover_t1, (double)time1);
    }
    ~CoverScope() { f.cover_t0 = t0; f.cover_t1 = t1; }
};

struct BVHNode : Hittable {  // src/bvh.rs:10-15
    HitPtr left, right;
    AABB _box;
    // the interval BVHNode::new was given (src/bvh.rs:94); f32 is enough to tell (0, 0.1) from (0, 1), rounded inward
    float time0 = -std::numeric_limits<float>::infinity(), time1 = std::numeric_limits<float>::infinity();
    void set_times(double t0, double t1) { time0 = f32_up(t0); time1 = f32_down(t1); }
    bool bounding_box(double, double, AABB& out) const override { out = _box; return true; }  // :113-116
    // the hull of the subtree's motion intervals, computed once
    mutable bool hull_known = false;
    mutable double hull_lo = 0.0, hull_hi = 0.0;
    void motion_hull(double& lo, double& hi) const override {
        if (!hull_known) {
            hull_lo = std::numeric_limits<double>::infinity(); hull_hi = -hull_lo;
            if (left) left->motion_hull(hull_lo, hull_hi);
            if (right && right.get() != left.get()) right->motion_hull(hull_lo, hull_hi);
            hull_known = true;
        }
        lo = std::fmin(lo, hull_lo); hi = std::fmax(hi, hull_hi);
    }
    bool bounds_over(double t0, double t1, AABB& out) const override {
        if (!left) return false;
        AABB a, b;
        if (!left->bounds_over(t0, t1, a)) return false;
        if (right.get() == left.get()) { out = a; return true; }
        if (!right->bounds_over(t0, t1, b)) return false;
        out = surrounding_box(a, b);
        return true;
    }
    // Does the stored box hold everything below for every ray time the moving spheres below are defined for? It was
    // built over (time0, time1) (src/bvh.rs:83-88) as the union of what its children presented: a MovingSphere below
    // whose motion interval reaches outside sticks out of it (the reference's random_scene: tree over (0, 0.1), spheres
    // moving over (0, 1), src/main.rs:315), and so does the content of a child whose own box does not hold (a nested
    // tree built over a shorter interval), whatever this node's interval.
    virtual bool children_hold() const { return left->holds_content() && (right.get() == left.get() || right->holds_content()); }
    mutable bool covered_known = false, covered_value = false;
    bool covered() const {
        if (!covered_known) {
            double lo = std::numeric_limits<double>::infinity(), hi = -lo;
            motion_hull(lo, hi);
            covered_value = (!(lo <= hi) || (lo >= (double)time0 && hi <= (double)time1)) && children_hold();
            covered_known = true;
        }
        return covered_value;
    }
    bool holds_content() const override { return covered(); }
    void flatten(Flat& f) const override {
        CoverScope cover(f, time0, time1);
        // A box that holds its content is tested with the carried interval (RT_INNER_CARRIED), also in a scene where
        // others do not. One that may not is tested with the reference's per-axis rule, as the reference does, and gets
        // a box in front of it that does hold the content over its whole motion (carried interval): a ray that misses
        // that box cannot hit anything below, whatever the reference's own test says, and most rays do miss it, where
        // the per-axis rule on its own lets nearly every ray through.
        const bool holds = covered();
        // A box that contains the world's box is passed by every ray that starts inside the world (all of them but
        // camera rays of a camera placed outside, and those it only costs the work it would have saved): the test can
        // tell nothing, so the node gets no entry and its children take its place, as those of a HittableList would.
        // Skipping a box that holds its content never changes a hit. Only in world space, and not for a box that is
        // flat on an axis (the reference never hits that one, src/aabb.rs:25).
        bool is_flat = false;
        for (int a = 0; a < 3; ++a) is_flat = is_flat || (_box._min[a] == _box._max[a]);
        if (!f.have_world_box) { f.world_box = _box; f.have_world_box = true; }  // the first node flattened is the world's root
        if (holds && !is_flat && f.xf_stack.empty() && !f.in_medium && cull_boxes_enabled()) {
            bool contains = true;
            for (int a = 0; a < 3; ++a) contains = contains && _box._min[a] <= f.world_box._min[a] && _box._max[a] >= f.world_box._max[a];
            if (contains) {
                left->flatten(f);
                if (right.get() != left.get()) right->flatten(f);
                return;
            }
        }
        uint32_t cover_at = 0xFFFFFFFFu;
        if (!holds && !f.in_medium && cull_boxes_enabled()) {
            AABB all;
            if (bounds_over(hull_lo, hull_hi, all)) {
                cover_at = f.push_node(RT_N_INNER, RT_INNER_COVER);
                for (int a = 0; a < 3; ++a) {
                    const double pad = cull_pad(all._min[a], all._max[a]);  // see CullBox::open
                    f.nodes[cover_at].lo[a] = f32_down(all._min[a] - pad);
                    f.nodes[cover_at].hi[a] = f32_up(all._max[a] + pad);
                }
            }
        }
        const uint32_t at = f.push_node(RT_N_INNER, holds ? RT_INNER_CARRIED : 0u);
        RtNode& n = f.nodes[at];
        // A box that is flat on an axis is never hit by the reference's test (src/aabb.rs:25, t1 <= t0);
        // rounding it outward to f32 could give it thickness, so it is stored inverted instead.
        bool flat = false;
        for (int a = 0; a < 3; ++a) flat = flat || (_box._min[a] == _box._max[a]);
        for (int a = 0; a < 3; ++a) {
            n.lo[a] = flat ? std::numeric_limits<float>::infinity() : f32_down(_box._min[a]);
            n.hi[a] = flat ? -std::numeric_limits<float>::infinity() : f32_up(_box._max[a]);
        }
        left->flatten(f);
        // A span-1 node holds the same object twice (src/bvh.rs:39-44); hitting it a second time with
        // t_max = rec.t changes nothing, so it is emitted once.
        if (right.get() != left.get()) right->flatten(f);
        f.nodes[at].link = (uint32_t)f.nodes.size();
        if (cover_at != 0xFFFFFFFFu) f.nodes[cover_at].link = (uint32_t)f.nodes.size();
    }
    static std::shared_ptr<BVHNode> create(HittableList& list, double time0, double time1);  // BVHNode::new, :94-97
};

// box_compare (src/bvh.rs:119-134) keys on bounding_box(0, 0)._min[axis].
inline double box_key(const HitPtr& o, int axis) {
    AABB b;
    if (!o->bounding_box(0.0, 0.0, b)) throw std::runtime_error("object without a bounding box in BVHNode::new");  // :123-125
    return b._min[axis];
}

// Split-axis policy of the builder below. false: the reference's random axis. true: the axis on which the
// box minima of the node's objects spread the widest (ties: the lowest axis).
inline thread_local bool g_bvh_widest_axis = false;

inline int widest_axis(const std::vector<HitPtr>& objects, size_t start, size_t end) {
    double lo[3], hi[3];
    for (int a = 0; a < 3; ++a) { lo[a] = std::numeric_limits<double>::infinity(); hi[a] = -lo[a]; }
    for (size_t i = start; i < end; ++i) {
        AABB b;
        if (!objects[i]->bounding_box(0.0, 0.0, b)) throw std::runtime_error("object without a bounding box in BVHNode::new");
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], b._min[a]); hi[a] = std::max(hi[a], b._min[a]); }
    }
    int best = 0;
    for (int a = 1; a < 3; ++a) if (hi[a] - lo[a] > hi[best] - lo[best]) best = a;
    return best;
}

// BVHNode::new draws one random_int per node, in depth-first order (src/bvh.rs:24). The scene stream is
// counter-based, so the draw of every node is known from its position alone and large subtrees can be built by
// separate threads while the tree stays the one a sequential build gives.
inline uint64_t rng_position(const rt::Rng& r) { return (uint64_t)r.k | ((uint64_t)r.k_hi << 32); }
inline void rng_seek(rt::Rng& r, uint64_t draw) {
    r.k = (uint32_t)draw;
    r.k_hi = (uint32_t)(draw >> 32);
    if (r.k & 1u) {  // the second draw of a block: its first half was consumed, the second waits in `spare`
        const rt::Philox4 o = rt::philox4x32_10(r.a, r.b, r.c, (r.k >> 1) | (r.k_hi << 31), r.k0, r.k1);
        r.spare = ((uint64_t)o.v[3] << 32) | o.v[2];
    }
}

// Boxes of six sides that are leaves of a tree get a culling box of their own (see CullBox) when the tree's list holds
// at least this many of them (the 400 ground boxes of final_scene, src/main.rs:591-612): a lone box among other
// objects is mostly hit by the rays that reach it, and the extra entry then only costs (10 M-triangle mesh: -9 %).
constexpr size_t kManyBoxes = 8;
inline bool many_boxes_in(const std::vector<HitPtr>& objects) {
    size_t n = 0;
    for (const HitPtr& o : objects) if (o->is_box6() && ++n >= kManyBoxes) return true;
    return false;
}

struct BvhBuild {
    std::vector<HitPtr>& objects;
    double time0, time1;
    rt::Rng base;                                  // the scene stream (position irrelevant)
    bool widest;
    bool many_boxes = false;
    std::unordered_map<size_t, uint64_t> draws;    // nodes (= draws) of a subtree over n objects

    uint64_t count_draws(size_t n) {
        if (n <= 2) return 1;
        auto it = draws.find(n);
        if (it != draws.end()) return it->second;
        const uint64_t d = 1 + count_draws(n / 2) + count_draws(n - n / 2);
        draws.emplace(n, d);
        return d;
    }
    uint64_t draws_of(size_t n) const { return n <= 2 ? 1 : draws.at(n); }

    // src/bvh.rs:17-91; `draw` = index of this node's random_int in the scene stream
    std::shared_ptr<BVHNode> build(size_t start, size_t end, uint64_t draw, int fork_levels) const {
        rt::Rng r = base;
        rng_seek(r, draw);
        int axis = (int)r.uniform(0.0, 3.0);  // random_int(0, 2), drawn for every node, also span 1 and 2 (:24)
        const size_t span = end - start;
        if (widest && span > 1) axis = widest_axis(objects, start, end);
        auto node = std::make_shared<BVHNode>();
        if (span == 1) {
            node->left = node->right = objects[start];
        } else if (span == 2) {
            if (box_key(objects[start + 1], axis) < box_key(objects[start], axis)) {  // Ordering::Greater (:54-60)
                node->left = objects[start + 1];
                node->right = objects[start];
            } else {
                node->left = objects[start];
                node->right = objects[start + 1];
            }
        } else {
            // slice::sort_by(comparator) is a stable sort (:70). Keys are extracted once per level; a stable
            // sort by key gives the same permutation as the comparator-based one.
            std::vector<std::pair<double, HitPtr>> keyed;
            keyed.reserve(span);
            for (size_t i = start; i < end; ++i) keyed.emplace_back(box_key(objects[i], axis), std::move(objects[i]));
            std::stable_sort(keyed.begin(), keyed.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
            for (size_t i = 0; i < span; ++i) objects[start + i] = std::move(keyed[i].second);
            keyed = {};
            const size_t mid = start + span / 2;
            const uint64_t d_left = draw + 1, d_right = d_left + draws_of(mid - start);
            if (fork_levels > 0 && span >= 65536) {
                auto left = std::async(std::launch::async, [&]() { return build(start, mid, d_left, fork_levels - 1); });
                node->right = build(mid, end, d_right, fork_levels - 1);
                node->left = left.get();
            } else {
                node->left = build(start, mid, d_left, 0);
                node->right = build(mid, end, d_right, 0);
            }
        }
        AABB bl, br;
        if (!node->left->bounding_box(time0, time1, bl) || !node->right->bounding_box(time0, time1, br))
            throw std::runtime_error("object without a bounding box in BVHNode::new");  // :83-87
        node->_box = surrounding_box(bl, br);
        node->set_times(time0, time1);
        if (span <= 2) {  // its children are objects of the list, not nodes
            node->left->adopted_by_bvh(time0, time1, many_boxes);
            if (node->right.get() != node->left.get()) node->right->adopted_by_bvh(time0, time1, many_boxes);
        }
        return node;
    }
};

// A failure of the library's own C ABI underneath a host call: carries the RtError code through the unwinding.
struct AbiError : std::runtime_error {
    int code;
    AbiError(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

// Tree policy of BVHNode::create. false: the host recursion above. true: lists of two or more objects get the tree
// that rt_bvh_build makes on the current HIP device.
inline thread_local bool g_bvh_device = false;

struct DeviceBvh : BVHNode {  // left / right stay empty: the tree is made on the device when the scene is flattened
    std::vector<HitPtr> objects;
    double t0 = 0.0, t1 = 0.0;
    mutable double ms_device = 0.0;
    void motion_hull(double& lo, double& hi) const override { for (const auto& o : objects) o->motion_hull(lo, hi); }
    bool children_hold() const override {
        for (const auto& o : objects) if (!o->holds_content()) return false;
        return true;
    }
    bool bounds_over(double a, double b, AABB& out) const override {
        AABB temp;
        for (size_t i = 0; i < objects.size(); ++i) {
            if (!objects[i]->bounds_over(a, b, temp)) return false;
            out = i ? surrounding_box(out, temp) : temp;
        }
        return !objects.empty();
    }

    void boxes_of(std::vector<double>& boxes, AABB* whole) const {
        boxes.resize(objects.size() * 6);
        for (size_t i = 0; i < objects.size(); ++i) {
            AABB b;
            if (!objects[i]->bounding_box(t0, t1, b)) throw std::runtime_error("object without a bounding box in BVHNode::new");
            for (int a = 0; a < 3; ++a) { boxes[6 * i + a] = b._min[a]; boxes[6 * i + 3 + a] = b._max[a]; }
            if (whole) *whole = i ? surrounding_box(*whole, b) : b;  // min / max are exact: the same box as the tree's root
        }
    }
    DeviceBvh(const std::vector<HitPtr>& list, double time0, double time1) : objects(list), t0(time0), t1(time1) {
        set_times(time0, time1);
        if (objects.size() < 2) throw std::invalid_argument("a tree built on the device needs at least two objects");
        if (objects.size() > RT_BVH_MAX_OBJECTS) throw std::invalid_argument("too many objects for one tree");
        std::vector<double> boxes;
        boxes_of(boxes, &_box);
        const bool many_boxes = many_boxes_in(objects);
        for (const HitPtr& o : objects) o->adopted_by_bvh(time0, time1, many_boxes);
        // what rt_bvh_build refuses is refused here, where the reference's BVHNode::new would run (src/bvh.rs:94)
        for (size_t i = 0; i < objects.size(); ++i)
            for (int a = 0; a < 3; ++a) {
                const double mn = boxes[6 * i + a], mx = boxes[6 * i + 3 + a], c = (mn + mx) * 0.5;
                if (mn != mn || mx != mx || c != c || std::isinf(c)) throw std::invalid_argument("a bounding box holds NaN or has no finite centre");
            }
        int n_dev = rt_device_count();
        if (n_dev < 1) throw AbiError(RT_E_DEVICE, "building a tree on the device needs a HIP device; there is no CPU path");
    }
    // Every object is flattened on its own, in list order (the order they were created in: sequential memory), into
    // a side array; the device builds the tree over the boxes and puts each object's entries where its leaf stands
    // (rt_bvh_build_stream). A walk over the tokens on the host would visit the objects in Morton order, one cache
    // miss per object: 2 s for 10 M triangles.
    void flatten(Flat& f) const override {
        CoverScope cover(f, time0, time1);
        if (!f.have_world_box) { f.world_box = _box; f.have_world_box = true; }
        std::vector<RtNode> mine;
        mine.swap(f.nodes);  // the entries before this tree; f.nodes collects the objects' entries meanwhile
        std::vector<uint32_t> offsets(objects.size() + 1);
        try {
            for (size_t i = 0; i < objects.size(); ++i) {
                offsets[i] = (uint32_t)f.nodes.size();
                objects[i]->flatten(f);
                if (f.nodes.size() >= 0xFFFFFFFFull) throw std::length_error("too many entries");
            }
        } catch (...) { mine.swap(f.nodes); throw; }
        offsets[objects.size()] = (uint32_t)f.nodes.size();
        std::vector<RtNode> obj_entries;
        obj_entries.swap(f.nodes);
        mine.swap(f.nodes);
        std::vector<double> boxes;
        boxes_of(boxes, nullptr);
        const size_t base = f.nodes.size(), n_out = objects.size() - 1 + obj_entries.size();
        if (base + n_out >= 0xFFFFFFFFull) throw std::length_error("too many entries");
        f.nodes.resize(base + n_out);
        const auto t_begin = std::chrono::steady_clock::now();
        const int rc = rt_bvh_build_stream(boxes.data(), objects.size(), obj_entries.data(), offsets.data(), f.nodes.data() + base, &ms_device);
        if (rc != RT_OK) throw AbiError(rc, rt_last_error());
        if (std::getenv("RT_HOST_TIMING"))
            std::fprintf(stderr, "DeviceBvh: %zu objects, rt_bvh_build_stream %.3f s (tree on the device %.2f ms)\n", objects.size(),
                         std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(), ms_device);
        if (base)  // links of the new entries count from where this tree starts
            for (size_t k = base; k < f.nodes.size(); ++k) {
                const uint32_t kind = f.nodes[k].meta & 15u;
                if (kind == RT_N_INNER || kind == RT_N_MEDIUM) f.nodes[k].link += (uint32_t)base;
            }
    }
};

inline std::shared_ptr<BVHNode> BVHNode::create(HittableList& list, double time0, double time1) {
    if (list.objects.empty()) throw std::invalid_argument("BVHNode::new on an empty list");
    if (g_bvh_device && list.objects.size() >= 2) {
        // the scene stream moves on as if the host recursion had drawn its axis per node, so that whatever the scene
        // draws afterwards is the same under either tree policy
        BvhBuild counter{list.objects, time0, time1, ctx().rng, false, false, {}};
        rng_seek(ctx().rng, rng_position(ctx().rng) + counter.count_draws(list.objects.size()));
        return std::make_shared<DeviceBvh>(list.objects, time0, time1);
    }
    BvhBuild b{list.objects, time0, time1, ctx().rng, g_bvh_widest_axis, many_boxes_in(list.objects), {}};
    const uint64_t first = rng_position(ctx().rng);
    const uint64_t total = b.count_draws(list.objects.size());
    int fork_levels = 4;  // up to 16 subtrees at a time; RT_BVH_FORK_LEVELS=0 builds on the calling thread alone
    if (const char* e = std::getenv("RT_BVH_FORK_LEVELS")) fork_levels = std::max(0, std::min(8, std::atoi(e)));
    auto root = b.build(0, list.objects.size(), first, fork_levels);
    rng_seek(ctx().rng, first + total);
    return root;
}

// ---- camera (src/camera.rs:27-59) -------------------------------------------
inline RtCamera camera_new(Point lookfrom, Point lookat, Vec3 vup, double vfov, double aspect_ratio, double aperture,
                           double focus_dist, double t0, double t1) {
    const double theta = degrees_to_radians(vfov);
    const double h = std::tan(theta / 2.0);
    const double viewport_height = 2.0 * h;
    const double viewport_width = aspect_ratio * viewport_height;
    const Vec3 w = (lookfrom - lookat).unit();
    const Vec3 u = cross(vup, w).unit();
    const Vec3 v = cross(w, u);
    const Vec3 horizontal = u * focus_dist * viewport_width;
    const Vec3 vertical = v * focus_dist * viewport_height;
    const Vec3 llc = lookfrom - horizontal / 2.0 - vertical / 2.0 - w * focus_dist;
    RtCamera c{};
    put3(c.orig, lookfrom); put3(c.lower_left_corner, llc); put3(c.horizontal, horizontal); put3(c.vertical, vertical);
    put3(c.u, u); put3(c.v, v);
    c.lens_radius = aperture / 2.0; c.time0 = t0; c.time1 = t1;
    return c;
}

}  // namespace rth
